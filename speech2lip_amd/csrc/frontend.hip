// Small pixel-invariant / frame-invariant kernels that feed the fused MLP:
//   audio_encode_kernel   -- TalkingFace.audio_merge_forward        (tf_nerf.py:197-213, :91-109)
//   frame_vectors_kernel  -- frame-only halves of layer 0 / skip     (tf_nerf.py:247, :252-258, :269-281, :434-442)
//   pixel_tables_kernel   -- Embedder + pixel-only halves            (tf_nerf.py:404-425, :252, :269)
//   embed_rows_kernel     -- rows [u,v,a64] -> x[128] for the general rgb_forward path
// All are latency/launch-bound VALU kernels (<1 % of a clip's time); weights are read through
// transposed copies so that consecutive threads read consecutive addresses.
#include "s2l_common.h"

namespace s2l {

constexpr int kFB = 4;  // frames per block in the audio / frame-vector kernels
constexpr int kDedupMinWindows = 2048;      // s2l_audio_encode: from this many windows on, copies of window 0 are detected on the device (below)

__device__ inline float lrelu(float x) { return x > 0.f ? x : 0.02f * x; }

// Conv1d(k=3, stride=2, pad=1) + LeakyReLU(0.02) over kFB frames held in LDS.
// xin [kFB][CIN][TIN], yout [kFB][COUT][TIN/2], wT [CIN][3][COUT].
template <int CIN, int COUT, int TIN, int FB = kFB, int U = 8>
__device__ inline void conv_stage(const float* __restrict__ wT, const float* __restrict__ b, const float* xin,
                                  float* yout) {
  constexpr int TOUT = TIN / 2;
  for (int item = threadIdx.x; item < FB * COUT * TOUT; item += blockDim.x) {
    const int o = item % COUT;
    const int tau = (item / COUT) % TOUT;
    const int fb = item / (COUT * TOUT);
    const float* x = xin + fb * CIN * TIN;
    float acc = b[o];
    // (x8: 24 weight loads in flight per thread instead of 3 -- the chain of fmas keeps its order, so the bits do; one frame per call is a
    //  chain of dependent L2 round trips: 22 -> 11 us for the encoder at F = 1)
    // Of the three taps t = 2 tau - 1, 2 tau, 2 tau + 1 only the first can fall into the padding (tau = 0; 2 tau + 1 <= TIN - 1 always): that
    // step is SKIPPED, as before, but by a select on a loop-invariant flag instead of a lane-dependent branch around a load -- the branch
    // cost 60 - 100 cycles per step in the stages with more than one tau per wave (3.5 + 1.3 + 2.5 us of a 15-us call; same bits).
    static_assert(TIN == 2 * TOUT, "stride 2, k = 3, pad 1");
    const bool first_ok = tau > 0;
    const int t0 = first_ok ? 2 * tau - 1 : 0;
#pragma unroll U
    for (int c = 0; c < CIN; ++c) {
      const float v = fmaf(wT[(c * 3 + 0) * COUT + o], x[c * TIN + t0], acc);
      acc = first_ok ? v : acc;
      acc = fmaf(wT[(c * 3 + 1) * COUT + o], x[c * TIN + 2 * tau], acc);
      acc = fmaf(wT[(c * 3 + 2) * COUT + o], x[c * TIN + 2 * tau + 1], acc);
    }
    yout[(fb * COUT + o) * TOUT + tau] = lrelu(acc);
  }
  __syncthreads();
}

// FB frames per block: 4 for clips; 1 when a call has fewer than 4 frames (the reference's one-frame-per-call mode: a block of 4 with
// one valid frame spent 3/4 of its dependent load + fma chains on copies of it -- 40 us of a 180-us single-frame render).  The
// arithmetic per output is the same chain in both.
// DEDUP (calls of many windows): the reference's per-frame driver hands the encoder the SAME window once per pixel (inference.py:144, 151:
// `auds.tile(...)`, 4 096 - 16 384 copies).  feat[0] then already holds f(window 0) (audio_encode_lds_kernel ran on it first, the same
// function bit for bit): a block whose windows all equal window 0 BITWISE copies that row instead of recomputing it -- the same output
// by definition of a function; a block with any other window runs the stages as always.  (Row 0 may be rewritten by block 0 while others
// read it: with the bits it already has.)
template <int FB, bool DEDUP = false>
__global__ __launch_bounds__(256) void audio_encode_kernel(const float* __restrict__ packed,
                                                          const float* __restrict__ windows,
                                                          float* __restrict__ feat, int64_t n) {
  __shared__ float x0[FB * 29 * 16];
  __shared__ float y1[FB * 32 * 8];
  __shared__ float y2[FB * 32 * 4];
  __shared__ float y3[FB * 64 * 2];
  __shared__ float y4[FB * 64];
  __shared__ float f1[FB * 64];
  const int64_t f0 = (int64_t)blockIdx.x * FB;
  int differs = 0;
  // windows [f][t 16][c 29] -> x0 [fb][c][t]   (the permute of tf_nerf.py:207)
  for (int i = threadIdx.x; i < FB * 16 * 29; i += blockDim.x) {
    const int fb = i / (16 * 29), r = i - fb * 16 * 29;
    const int t = r / 29, c = r - t * 29;
    const int64_t f = f0 + fb < n ? f0 + fb : n - 1;
    const float v = windows[f * 16 * 29 + r];
    x0[(fb * 29 + c) * 16 + t] = v;
    if (DEDUP) differs |= __float_as_uint(v) != __float_as_uint(windows[r]);
  }
  if (DEDUP) {
    if (!__syncthreads_or(differs)) {
      const int fb = threadIdx.x >> 6, o = threadIdx.x & 63;
      if (fb < FB && f0 + fb < n && f0 + fb > 0) feat[(f0 + fb) * 64 + o] = feat[o];
      return;
    }
  } else {
    __syncthreads();
  }
  conv_stage<29, 32, 16, FB>(packed + OFF_C0W, packed + OFF_C0B, x0, y1);
  conv_stage<32, 32, 8, FB>(packed + OFF_C2W, packed + OFF_C2B, y1, y2);
  conv_stage<32, 64, 4, FB>(packed + OFF_C4W, packed + OFF_C4B, y2, y3);
  conv_stage<64, 64, 2, FB>(packed + OFF_C6W, packed + OFF_C6B, y3, y4);
  {  // Linear(64,64) + LeakyReLU, Linear(64,64): thread = (frame, output)
    const int fb = threadIdx.x >> 6, o = threadIdx.x & 63;
    const bool mine = fb < FB;      // (FB = 1: one wave works, the others only keep the barrier)
    float acc = packed[OFF_F0B + o];
    if (mine) {
#pragma unroll 16
      for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F0W + k * 64 + o], y4[fb * 64 + k], acc);
      f1[fb * 64 + o] = lrelu(acc);
    }
    __syncthreads();
    acc = packed[OFF_F2B + o];
    if (mine) {
#pragma unroll 16
      for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F2W + k * 64 + o], f1[fb * 64 + k], acc);
      if (f0 + fb < n) feat[(f0 + fb) * 64 + o] = acc;
    }
  }
}

// The encoder for calls of fewer than kFB frames (the reference's one-frame-per-call mode, inference.py:129-159): one frame per block, and
// the block first brings ALL of the encoder's weights -- one contiguous 131-KB section of the packed blob -- into LDS with 16-byte loads
// from every thread (one L2 round trip for the lot), then runs the same stages on them.  audio_encode_kernel<1> walked six stages of
// dependent "load a weight from L2, fma" chains: 22.8 us per frame; the chains are the same here (same operands, same order: the same
// bits) with LDS latencies in them.
constexpr int kEncFloats = (int)(OFF_G0 - OFF_C0W);
constexpr int kEncLds = (kEncFloats + 29 * 16 + 32 * 8 + 32 * 4 + 64 * 2 + 64 + 64) * 4;
static_assert(OFF_W0T % 4 == 0 && OFF_W5AT % 4 == 0, "16-byte loads of the frame vectors' second stage");
static_assert(OFF_C0W % 4 == 0 && kEncFloats % 4 == 0 && kEncLds <= 160 * 1024, "the encoder's section is whole 16-byte pieces and fits the LDS");
// The six stages of one frame on the LDS copy of the encoder's weights; returns output o = threadIdx.x & 63 of the final layer in the first
// wave (`mine`); the other waves only keep the barriers.  dbg: pricing builds' way out (S2L_ENC_STOP).
__device__ inline float encode_frame_in_lds(const float* __restrict__ packed, const float* __restrict__ window, float* enc_smem, float* dbg) {
  float* const wl = enc_smem;                       // the section [OFF_C0W, OFF_G0) of `packed`
  float* const x0 = wl + kEncFloats;
  float* const y1 = x0 + 29 * 16;
  float* const y2 = y1 + 32 * 8;
  float* const y3 = y2 + 32 * 4;
  float* const y4 = y3 + 64 * 2;
  float* const f1 = y4 + 64;
  {      // every thread's 32 pieces requested before the first is stored: one round trip, not eight (2.8 -> 1.3 us)
    const f4* src = reinterpret_cast<const f4*>(packed + OFF_C0W);
    f4* dst = reinterpret_cast<f4*>(wl);
    constexpr int kPer = (kEncFloats / 4 + 255) / 256;
    f4 r[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = threadIdx.x + 256 * k;
      r[k] = src[i < kEncFloats / 4 ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < kEncFloats / 4) dst[i] = r[k];
    }
  }
  for (int i = threadIdx.x; i < 16 * 29; i += 256) {      // window [t 16][c 29] -> x0 [c][t]
    const int t = i / 29, c = i - t * 29;
    x0[c * 16 + t] = window[i];
  }
  __syncthreads();
  const float* w = wl - OFF_C0W;                      // so that the blob's offsets address the copy
#ifndef S2L_ENC_UNROLL
#define S2L_ENC_UNROLL 8
#endif
#ifndef S2L_ENC_FC_UNROLL
#define S2L_ENC_FC_UNROLL 16
#endif
#ifdef S2L_ENC_STOP      // pricing builds (tools/dev): leave after the staging (0) or after stage S2L_ENC_STOP
#define S2L_ENC_STOP_AT(k) if (S2L_ENC_STOP == (k)) { if (threadIdx.x < 64 && dbg) dbg[threadIdx.x] = y1[threadIdx.x] + y2[threadIdx.x] + y3[threadIdx.x] + y4[threadIdx.x]; return 0.f; }
#else
#define S2L_ENC_STOP_AT(k)
#endif
  S2L_ENC_STOP_AT(0)
  conv_stage<29, 32, 16, 1, S2L_ENC_UNROLL>(w + OFF_C0W, w + OFF_C0B, x0, y1);
  S2L_ENC_STOP_AT(1)
  conv_stage<32, 32, 8, 1, S2L_ENC_UNROLL>(w + OFF_C2W, w + OFF_C2B, y1, y2);
  S2L_ENC_STOP_AT(2)
  conv_stage<32, 64, 4, 1, S2L_ENC_UNROLL>(w + OFF_C4W, w + OFF_C4B, y2, y3);
  S2L_ENC_STOP_AT(3)
  conv_stage<64, 64, 2, 1, S2L_ENC_UNROLL>(w + OFF_C6W, w + OFF_C6B, y3, y4);
  S2L_ENC_STOP_AT(4)
  const int o = threadIdx.x & 63;
  const bool mine = threadIdx.x < 64;
  float acc = w[OFF_F0B + o];
  if (mine) {
#pragma unroll S2L_ENC_FC_UNROLL
    for (int k = 0; k < 64; ++k) acc = fmaf(w[OFF_F0W + k * 64 + o], y4[k], acc);
    f1[o] = lrelu(acc);
  }
  __syncthreads();
  acc = w[OFF_F2B + o];
  if (mine) {
#pragma unroll S2L_ENC_FC_UNROLL
    for (int k = 0; k < 64; ++k) acc = fmaf(w[OFF_F2W + k * 64 + o], f1[k], acc);
  }
  return acc;
}

__global__ __launch_bounds__(256) void audio_encode_lds_kernel(const float* __restrict__ packed, const float* __restrict__ windows,
                                                              float* __restrict__ feat, int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float enc_smem[];
  const int64_t f = blockIdx.x;
  const float acc = encode_frame_in_lds(packed, windows + f * 16 * 29, enc_smem, feat + f * 64);
#ifndef S2L_ENC_STOP
  if (threadIdx.x < 64) feat[f * 64 + threadIdx.x] = acc;
#endif
}

// ---- audio encoder backward (training) --------------------------------------------------------
// Gradient sections of one block's partial, torch layouts, in state-dict order.
constexpr int kAG_C0W = 0, kAG_C0B = kAG_C0W + 32 * 29 * 3, kAG_C2W = kAG_C0B + 32, kAG_C2B = kAG_C2W + 32 * 32 * 3,
              kAG_C4W = kAG_C2B + 32, kAG_C4B = kAG_C4W + 64 * 32 * 3, kAG_C6W = kAG_C4B + 64,
              kAG_C6B = kAG_C6W + 64 * 64 * 3, kAG_F0W = kAG_C6B + 64, kAG_F0B = kAG_F0W + 64 * 64,
              kAG_F2W = kAG_F0B + 64, kAG_F2B = kAG_F2W + 64 * 64, kAudioGradFloats = kAG_F2B + 64;

__device__ inline float lrelu_slope(float post) { return post > 0.f ? 1.f : 0.02f; }

// Backward of one Conv1d(k3,s2,p1)+LeakyReLU stage over kFB frames in LDS.
//   xin [kFB][CIN][TIN] = stage input (post-activation of the previous stage)
//   gy  [kFB][COUT][TIN/2] = gradient w.r.t. this stage's PRE-activation
//   -> dw [COUT][CIN][3], db [COUT] (block partials); gx [kFB][CIN][TIN] = gradient w.r.t. the previous
//      stage's pre-activation (skipped when gx == nullptr)
template <int CIN, int COUT, int TIN>
__device__ inline void conv_stage_bwd(const float* __restrict__ wT, const float* xin, const float* gy, float* dw, float* db,
                                      float* gx, int nvalid) {
  constexpr int TOUT = TIN / 2;
  for (int item = threadIdx.x; item < COUT * CIN * 3; item += blockDim.x) {
    const int k = item % 3, c = (item / 3) % CIN, o = item / (3 * CIN);
    float acc = 0.f;
    for (int fb = 0; fb < nvalid; ++fb)
#pragma unroll
      for (int tau = 0; tau < TOUT; ++tau) {
        const int t = 2 * tau + k - 1;
        if (t >= 0 && t < TIN) acc = fmaf(gy[(fb * COUT + o) * TOUT + tau], xin[(fb * CIN + c) * TIN + t], acc);
      }
    dw[item] = acc;
  }
  for (int o = threadIdx.x; o < COUT; o += blockDim.x) {
    float acc = 0.f;
    for (int fb = 0; fb < nvalid; ++fb)
      for (int tau = 0; tau < TOUT; ++tau) acc += gy[(fb * COUT + o) * TOUT + tau];
    db[o] = acc;
  }
  if (gx) {
    for (int item = threadIdx.x; item < kFB * CIN * TIN; item += blockDim.x) {
      const int t = item % TIN, c = (item / TIN) % CIN, fb = item / (TIN * CIN);
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int num = t + 1 - k;                 // t = 2*tau + k - 1
        if (num >= 0 && (num & 1) == 0 && (num >> 1) < TOUT) {
          const int tau = num >> 1;
          for (int o = 0; o < COUT; ++o) acc = fmaf(wT[(c * 3 + k) * COUT + o], gy[(fb * COUT + o) * TOUT + tau], acc);
        }
      }
      gx[item] = acc * lrelu_slope(xin[item]);
    }
  }
  __syncthreads();
}

// Recomputes the encoder forward for kFB frames in LDS, then back-propagates dfeat [B,64] to every
// encoder parameter; writes this block's partial sums (reduced over blocks by reduce_rows_kernel).
// 1024 threads (round 5; 256 before: 130 us per call, a chain of strided item loops -- every item is still computed by one thread in the
// same order, so the sums are the same bits); the (frame, output) sections below use the first 256.
__global__ __launch_bounds__(1024) void audio_backward_kernel(const float* __restrict__ packed,
                                                            const float* __restrict__ windows,
                                                            const float* __restrict__ dfeat, float* __restrict__ partial,
                                                            int64_t n) {
  __shared__ float x0[kFB * 29 * 16];
  __shared__ float y1[kFB * 32 * 8], g1[kFB * 32 * 8];
  __shared__ float y2[kFB * 32 * 4], g2[kFB * 32 * 4];
  __shared__ float y3[kFB * 64 * 2], g3[kFB * 64 * 2];
  __shared__ float y4[kFB * 64], g4[kFB * 64];
  __shared__ float f1[kFB * 64], gf1[kFB * 64], gout[kFB * 64];
  const int64_t f0 = (int64_t)blockIdx.x * kFB;
  const int nvalid = (int)((n - f0) < kFB ? (n - f0) : kFB);
  float* P = partial + (int64_t)blockIdx.x * kAudioGradFloats;
  for (int i = threadIdx.x; i < kFB * 16 * 29; i += blockDim.x) {
    const int fb = i / (16 * 29), r = i - fb * 16 * 29;
    const int t = r / 29, c = r - t * 29;
    const int64_t f = f0 + fb < n ? f0 + fb : n - 1;
    x0[(fb * 29 + c) * 16 + t] = windows[f * 16 * 29 + r];
  }
  __syncthreads();
  conv_stage<29, 32, 16>(packed + OFF_C0W, packed + OFF_C0B, x0, y1);
  conv_stage<32, 32, 8>(packed + OFF_C2W, packed + OFF_C2B, y1, y2);
  conv_stage<32, 64, 4>(packed + OFF_C4W, packed + OFF_C4B, y2, y3);
  conv_stage<64, 64, 2>(packed + OFF_C6W, packed + OFF_C6B, y3, y4);
  const int fb = (threadIdx.x >> 6) & 3, o = threadIdx.x & 63;
  const bool fo = threadIdx.x < 256;      // the threads of the (frame, output) sections
  if (fo) {
    float acc = packed[OFF_F0B + o];
    for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F0W + k * 64 + o], y4[fb * 64 + k], acc);
    f1[fb * 64 + o] = lrelu(acc);
    gout[fb * 64 + o] = fb < nvalid ? dfeat[(f0 + fb) * 64 + o] : 0.f;
  }
  __syncthreads();
  // Linear(64,64) #2: y = W2 f1 + b2.  packed F2W is [in k][out o].
  for (int item = threadIdx.x; item < 64 * 64; item += blockDim.x) {
    const int oo = item >> 6, kk = item & 63;
    float acc = 0.f;
    for (int b = 0; b < nvalid; ++b) acc = fmaf(gout[b * 64 + oo], f1[b * 64 + kk], acc);
    P[kAG_F2W + item] = acc;
  }
  if (threadIdx.x < 64) {
    float acc = 0.f;
    for (int b = 0; b < nvalid; ++b) acc += gout[b * 64 + threadIdx.x];
    P[kAG_F2B + threadIdx.x] = acc;
  }
  if (fo) {
    float acc = 0.f;   // d f1[fb][o] = sum_out W2[out][o] gout[fb][out], then through the LeakyReLU
    for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F2W + o * 64 + k], gout[fb * 64 + k], acc);
    gf1[fb * 64 + o] = acc * lrelu_slope(f1[fb * 64 + o]);
  }
  __syncthreads();
  // Linear(64,64) #1: f1_pre = W1 y4 + b1
  for (int item = threadIdx.x; item < 64 * 64; item += blockDim.x) {
    const int oo = item >> 6, kk = item & 63;
    float acc = 0.f;
    for (int b = 0; b < nvalid; ++b) acc = fmaf(gf1[b * 64 + oo], y4[b * 64 + kk], acc);
    P[kAG_F0W + item] = acc;
  }
  if (threadIdx.x < 64) {
    float acc = 0.f;
    for (int b = 0; b < nvalid; ++b) acc += gf1[b * 64 + threadIdx.x];
    P[kAG_F0B + threadIdx.x] = acc;
  }
  if (fo) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F0W + o * 64 + k], gf1[fb * 64 + k], acc);
    g4[fb * 64 + o] = acc * lrelu_slope(y4[fb * 64 + o]);   // gradient w.r.t. conv6 pre-activation (T = 1)
  }
  __syncthreads();
  conv_stage_bwd<64, 64, 2>(packed + OFF_C6W, y3, g4, P + kAG_C6W, P + kAG_C6B, g3, nvalid);
  conv_stage_bwd<32, 64, 4>(packed + OFF_C4W, y2, g3, P + kAG_C4W, P + kAG_C4B, g2, nvalid);
  conv_stage_bwd<32, 32, 8>(packed + OFF_C2W, y1, g2, P + kAG_C2W, P + kAG_C2B, g1, nvalid);
  conv_stage_bwd<29, 32, 16>(packed + OFF_C0W, x0, g1, P + kAG_C0W, P + kAG_C0B, nullptr, nvalid);
}

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                         int n_blocks, int n_elems) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elems) return;
  float s = 0.f;
  for (int b = 0; b < n_blocks; ++b) s += partial[(int64_t)b * n_elems + e];
  out[e] = s;
}

// q0[f] = W0 (Wa a_f + Wt PE(idx_f) + bsum0) + b0 ; q5[f] likewise with the skip projections.
// 256 threads: thread n owns output feature n for kFB frames.
// FB frames per block: kFB for clips, 1 for calls of fewer than kFB frames (as audio_encode_kernel: the same fma chains; with two
// accumulators instead of eight the compiler keeps 32 pairs of weight loads in flight: 21 -> 9 us at one frame per call).
template <int FB>
__global__ __launch_bounds__(256) void frame_vectors_kernel(const float* __restrict__ packed,
                                                           const float* __restrict__ feat,
                                                           const int64_t* __restrict__ frame_idx,
                                                           float* __restrict__ q0, float* __restrict__ q5, int64_t n) {
  constexpr int kFB = FB;      // (shadows the namespace constant inside this kernel)
  constexpr int kUnroll = FB == 1 ? 32 : 8;
  __shared__ float a[kFB][64];
  __shared__ float pe[kFB][20];
  __shared__ float s0[kFB][256];
  __shared__ float s5[kFB][256];
  const int64_t f0 = (int64_t)blockIdx.x * kFB;
  const int tid = threadIdx.x;
  if ((tid >> 6) < kFB) {
    const int fb = tid >> 6, k = tid & 63;
    const int64_t f = f0 + fb < n ? f0 + fb : n - 1;
    a[fb][k] = feat[f * 64 + k];
    if (k < 20) {
      // PositionalEncodingTime: pe[2i] = sin(pos*div_i), pe[2i+1] = cos(pos*div_i), pos = float(idx)
      const float pos = (float)frame_idx[f];
      const float arg = __fmul_rn(pos, packed[OFF_DIV + (k >> 1)]);
      pe[fb][k] = (k & 1) ? cosf(arg) : sinf(arg);
    }
  }
  __syncthreads();
  float acc0[kFB], acc5[kFB];
#pragma unroll
  for (int fb = 0; fb < kFB; ++fb) {
    acc0[fb] = packed[OFF_BSUM0 + tid];
    acc5[fb] = packed[OFF_BSUM5 + tid];
  }
  // (x8: eight pairs of weight loads in flight per thread; the fma chains keep their order.  One frame per call: 28 -> 21 us)
#pragma unroll kUnroll
  for (int k = 0; k < 64; ++k) {
    const float w0 = packed[OFF_WAT + k * 256 + tid], w5 = packed[OFF_WAST + k * 256 + tid];
#pragma unroll
    for (int fb = 0; fb < kFB; ++fb) {
      acc0[fb] = fmaf(w0, a[fb][k], acc0[fb]);
      acc5[fb] = fmaf(w5, a[fb][k], acc5[fb]);
    }
  }
#pragma unroll kUnroll
  for (int k = 0; k < 20; ++k) {
    const float w0 = packed[OFF_WTT + k * 256 + tid], w5 = packed[OFF_WTST + k * 256 + tid];
#pragma unroll
    for (int fb = 0; fb < kFB; ++fb) {
      acc0[fb] = fmaf(w0, pe[fb][k], acc0[fb]);
      acc5[fb] = fmaf(w5, pe[fb][k], acc5[fb]);
    }
  }
#pragma unroll
  for (int fb = 0; fb < kFB; ++fb) {
    s0[fb][tid] = acc0[fb];
    s5[fb][tid] = acc5[fb];
    acc0[fb] = packed[OFF_B0 + tid];
    acc5[fb] = packed[OFF_B5 + tid];
  }
  __syncthreads();
#pragma unroll kUnroll
  for (int k = 0; k < 256; ++k) {
    const float w0 = packed[OFF_W0T + k * 256 + tid], w5 = packed[OFF_W5AT + k * 256 + tid];
#pragma unroll
    for (int fb = 0; fb < kFB; ++fb) {
      acc0[fb] = fmaf(w0, s0[fb][k], acc0[fb]);
      acc5[fb] = fmaf(w5, s5[fb][k], acc5[fb]);
    }
  }
#pragma unroll
  for (int fb = 0; fb < kFB; ++fb)
    if (f0 + fb < n) {
      q0[(f0 + fb) * 256 + tid] = acc0[fb];
      q5[(f0 + fb) * 256 + tid] = acc5[fb];
    }
}

// One frame per call (fewer than kFB frames): frame_vectors_kernel<1> is one workgroup pulling 684 KB of weights through one CU in two
// dependent stages (9 us).  Here a frame's 2 x 256 outputs are split over kFVSplit workgroups: each one repeats stage 1 (every workgroup
// needs all 256 sums of both kinds; 172 KB, now from L2 lines its neighbours fetch too) and computes 32 + 32 outputs of stage 2, whose 64 KB
// of weights all 256 threads request into LDS BEFORE stage 1 starts (16 16-byte loads per thread in one round trip, hidden behind stage 1).
// Every output is the same chain of fmas in the same order on the same operands: the same bits.
constexpr int kFVSplit = 8;
__global__ __launch_bounds__(256) void frame_vectors_split_kernel(const float* __restrict__ packed, const float* __restrict__ feat,
                                                                 const int64_t* __restrict__ frame_idx, float* __restrict__ q0,
                                                                 float* __restrict__ q5, int64_t n) {
  constexpr int kOut = 256 / kFVSplit;                // outputs of each kind per workgroup
  __shared__ __attribute__((aligned(16))) float w0s[256 * kOut];      // [k][kOut]: the slice of W0T / W5AT
  __shared__ __attribute__((aligned(16))) float w5s[256 * kOut];
  __shared__ float a[64];
  __shared__ float pe[20];
  __shared__ float s0[256];
  __shared__ float s5[256];
  const int64_t f = blockIdx.x;
  const int j = blockIdx.y, tid = threadIdx.x;
  // stage 2's weights: the slice [k 256][kOut] of each matrix as 16-byte pieces, piece t + 256 i to thread t (lane-linear in LDS)
  constexpr int kPieces = kOut / 4;                   // per row
  f4 r0[kPieces], r5[kPieces];
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = tid + 256 * i, k = idx / kPieces, pc = idx % kPieces;
    r0[i] = *reinterpret_cast<const f4*>(packed + OFF_W0T + k * 256 + j * kOut + 4 * pc);
    r5[i] = *reinterpret_cast<const f4*>(packed + OFF_W5AT + k * 256 + j * kOut + 4 * pc);
  }
  if (tid < 64) {
    a[tid] = feat[f * 64 + tid];
    if (tid < 20) {
      const float pos = (float)frame_idx[f];
      const float arg = __fmul_rn(pos, packed[OFF_DIV + (tid >> 1)]);
      pe[tid] = (tid & 1) ? cosf(arg) : sinf(arg);
    }
  }
  __syncthreads();
  float acc0 = packed[OFF_BSUM0 + tid], acc5 = packed[OFF_BSUM5 + tid];
#pragma unroll
  for (int k = 0; k < 64; ++k) {      // (fully unrolled: the 128 + 40 weight loads of this stage leave together)
    acc0 = fmaf(packed[OFF_WAT + k * 256 + tid], a[k], acc0);
    acc5 = fmaf(packed[OFF_WAST + k * 256 + tid], a[k], acc5);
  }
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    acc0 = fmaf(packed[OFF_WTT + k * 256 + tid], pe[k], acc0);
    acc5 = fmaf(packed[OFF_WTST + k * 256 + tid], pe[k], acc5);
  }
  s0[tid] = acc0;
  s5[tid] = acc5;
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    *reinterpret_cast<f4*>(w0s + 4 * (tid + 256 * i)) = r0[i];
    *reinterpret_cast<f4*>(w5s + 4 * (tid + 256 * i)) = r5[i];
  }
  __syncthreads();
  if (tid < 2 * kOut) {      // threads 0 .. kOut-1: q0's outputs, kOut .. 2 kOut-1: q5's
    const bool five = tid >= kOut;
    const int o = five ? tid - kOut : tid;
    const float* ws = five ? w5s : w0s;
    const float* sv = five ? s5 : s0;
    float acc = packed[(five ? OFF_B5 : OFF_B0) + j * kOut + o];
#pragma unroll 32
    for (int k = 0; k < 256; ++k) acc = fmaf(ws[k * kOut + o], sv[k], acc);
    (five ? q5 : q0)[f * 256 + j * kOut + o] = acc;
  }
}

// One frame per call, one launch: the encoder and the frame vectors of a frame in the same workgroups (s2l_frame_front).  Each of the
// kFVSplit workgroups of a frame runs the WHOLE encoder itself (the same 8 us whether one workgroup does it or eight, and no launch
// boundary behind it: ~3 us of a 50-us call), then its share of the frame vectors exactly as frame_vectors_split_kernel; the second stage's
// weights wait in registers through the encoder and land in the LDS the encoder's weights have left.  Workgroup 0 of the frame also
// writes the feature row.  Same chains as the two kernels: the same bits.
__global__ __launch_bounds__(256) void frame_front_kernel(const float* __restrict__ packed, const float* __restrict__ windows,
                                                         const int64_t* __restrict__ frame_idx, float* __restrict__ feat,
                                                         float* __restrict__ q0, float* __restrict__ q5, int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float enc_smem[];
  constexpr int kOut = 256 / kFVSplit, kPieces = kOut / 4;
  static_assert(2 * 256 * kOut <= kEncFloats, "the second stage's slices fit where the encoder's weights were");
  float* const w0s = enc_smem;                      // (after the encoder is done with its weights)
  float* const w5s = enc_smem + 256 * kOut;
  __shared__ float a[64];
  __shared__ float pe[20];
  __shared__ float s0[256];
  __shared__ float s5[256];
  const int64_t f = blockIdx.x;
  const int j = blockIdx.y, tid = threadIdx.x;
  f4 r0[kPieces], r5[kPieces];
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = tid + 256 * i, k = idx / kPieces, pc = idx % kPieces;
    r0[i] = *reinterpret_cast<const f4*>(packed + OFF_W0T + k * 256 + j * kOut + 4 * pc);
    r5[i] = *reinterpret_cast<const f4*>(packed + OFF_W5AT + k * 256 + j * kOut + 4 * pc);
  }
  const float fv = encode_frame_in_lds(packed, windows + f * 16 * 29, enc_smem, nullptr);
  if (tid < 64) {
    a[tid] = fv;
    if (feat && j == 0) feat[f * 64 + tid] = fv;
    if (tid < 20) {
      const float pos = (float)frame_idx[f];
      const float arg = __fmul_rn(pos, packed[OFF_DIV + (tid >> 1)]);
      pe[tid] = (tid & 1) ? cosf(arg) : sinf(arg);
    }
  }
  __syncthreads();      // (also: every wave is past its last read of the encoder's weights)
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    *reinterpret_cast<f4*>(w0s + 4 * (tid + 256 * i)) = r0[i];
    *reinterpret_cast<f4*>(w5s + 4 * (tid + 256 * i)) = r5[i];
  }
  float acc0 = packed[OFF_BSUM0 + tid], acc5 = packed[OFF_BSUM5 + tid];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    acc0 = fmaf(packed[OFF_WAT + k * 256 + tid], a[k], acc0);
    acc5 = fmaf(packed[OFF_WAST + k * 256 + tid], a[k], acc5);
  }
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    acc0 = fmaf(packed[OFF_WTT + k * 256 + tid], pe[k], acc0);
    acc5 = fmaf(packed[OFF_WTST + k * 256 + tid], pe[k], acc5);
  }
  s0[tid] = acc0;
  s5[tid] = acc5;
  __syncthreads();
  if (tid < 2 * kOut) {
    const bool five = tid >= kOut;
    const int o = five ? tid - kOut : tid;
    const float* ws = five ? w5s : w0s;
    const float* sv = five ? s5 : s0;
    float acc = packed[(five ? OFF_B5 : OFF_B0) + j * kOut + o];
#pragma unroll 32
    for (int k = 0; k < 256; ++k) acc = fmaf(ws[k * kOut + o], sv[k], acc);
    (five ? q5 : q0)[f * 256 + j * kOut + o] = acc;
  }
}

// Embedder(10, 2): [u, v, sin(u), sin(v), cos(u), cos(v), sin(2u), sin(2v), ..., cos(512v)].
// The product x*freq is exact (power-of-two scale), sinf/cosf are the accurate OCML versions
// (arguments reach 512: never the fast __sinf intrinsics).
__device__ inline float embed_feature(float u, float v, int i) {
  if (i < 2) return i == 0 ? u : v;
  const int blk = (i - 2) >> 1;             // 0..19: (freq, fn)
  const float x = ((i & 1) ? v : u) * (float)(1 << (blk >> 1));
  return (blk & 1) ? cosf(x) : sinf(x);
}

constexpr int kPB = 16;  // pixels per block in pixel_tables_kernel == pixels per render tile

__global__ __launch_bounds__(256) void pixel_tables_kernel(const float* __restrict__ packed,
                                                          const float* __restrict__ coords, float* __restrict__ p0,
                                                          float* __restrict__ p5, int64_t hw) {
  __shared__ float e[kPB][44];
  __shared__ float s0[kPB][256];
  __shared__ float s5[kPB][256];
  const int64_t pbase = (int64_t)blockIdx.x * kPB;
  const int tid = threadIdx.x;
  for (int i = tid; i < kPB * kEmb; i += blockDim.x) {
    const int pb = i / kEmb, k = i - pb * kEmb;
    const int64_t p = pbase + pb < hw ? pbase + pb : hw - 1;
    e[pb][k] = embed_feature(coords[2 * p], coords[2 * p + 1], k);
  }
  __syncthreads();
  float acc0[kPB], acc5[kPB];
#pragma unroll
  for (int pb = 0; pb < kPB; ++pb) acc0[pb] = acc5[pb] = 0.f;
  for (int k = 0; k < kEmb; ++k) {
    const float w0 = packed[OFF_WUVT + k * 256 + tid], w5 = packed[OFF_WUVST + k * 256 + tid];
#pragma unroll
    for (int pb = 0; pb < kPB; ++pb) {
      acc0[pb] = fmaf(w0, e[pb][k], acc0[pb]);
      acc5[pb] = fmaf(w5, e[pb][k], acc5[pb]);
    }
  }
#pragma unroll
  for (int pb = 0; pb < kPB; ++pb) {
    s0[pb][tid] = acc0[pb];
    s5[pb][tid] = acc5[pb];
    acc0[pb] = acc5[pb] = 0.f;
  }
  __syncthreads();
  for (int k = 0; k < 256; ++k) {
    const float w0 = packed[OFF_W0T + k * 256 + tid], w5 = packed[OFF_W5AT + k * 256 + tid];
#pragma unroll
    for (int pb = 0; pb < kPB; ++pb) {
      acc0[pb] = fmaf(w0, s0[pb][k], acc0[pb]);
      acc5[pb] = fmaf(w5, s5[pb][k], acc5[pb]);
    }
  }
  // tile layout of render.hip: [pixel group][mb 16][q 4][px 16][4]; feature tid = mb*16 + q*4 + r.
  // kPB == 16 == one pixel group per block; rows past hw repeat the last pixel (never stored).
  const int mb = tid >> 4, qq = (tid >> 2) & 3, r = tid & 3;
  float* d0 = p0 + (int64_t)blockIdx.x * 4096 + ((mb * 4 + qq) * 16) * 4 + r;
  float* d5 = p5 + (int64_t)blockIdx.x * 4096 + ((mb * 4 + qq) * 16) * 4 + r;
#pragma unroll
  for (int pb = 0; pb < kPB; ++pb) {
    d0[pb * 4] = acc0[pb];
    d5[pb * 4] = acc5[pb];
  }
}

// General rows: x[row] = [E(u,v) (42) | a (64) | PE(t) (20) | 0 0].  128 threads per row pair.
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ packed,
                                                        const float* __restrict__ uv_audio, float time_pos,
                                                        float* __restrict__ x, int64_t n_rows) {
  const int64_t row = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 7);
  const int k = threadIdx.x & 127;
  if (row >= n_rows) return;
  const float* r = uv_audio + row * 66;
  float v;
  if (k < kEmb) v = embed_feature(r[0], r[1], k);
  else if (k < kEmb + kAud) v = r[2 + (k - kEmb)];
  else if (k < kEmb + kAud + kTime) {
    const int i = k - kEmb - kAud;
    const float arg = __fmul_rn(time_pos, packed[OFF_DIV + (i >> 1)]);
    v = (i & 1) ? cosf(arg) : sinf(arg);
  } else v = 0.f;
  x[row * kGenK + k] = v;
}

}  // namespace s2l

extern "C" int s2l_audio_encode(const float* packed, const float* windows, float* feat, int64_t n, s2l_stream_t stream) {
  if (n < 0) return S2L_E_SIZE;
  if (n == 0) return S2L_OK;
  if (!packed || !windows || !feat) return S2L_E_NULL;
  if (n < s2l::kFB) {
    static s2l::LdsOptIn flag;
    int dev = 0, n_cu = 0;
    int rc = s2l::current_device_cus(&dev, &n_cu);
    if (rc) return rc;
    if ((rc = s2l::ensure_dynamic_lds(reinterpret_cast<const void*>(s2l::audio_encode_lds_kernel), s2l::kEncLds, flag, dev))) return rc;
    hipLaunchKernelGGL(s2l::audio_encode_lds_kernel, dim3((unsigned)n), dim3(256), s2l::kEncLds, static_cast<hipStream_t>(stream), packed,
                       windows, feat, n);
  } else {
    const int64_t blocks = (n + s2l::kFB - 1) / s2l::kFB;
    if (n >= s2l::kDedupMinWindows) {      // many windows: window 0's feature first, then blocks that only hold copies of window 0 copy it
      static s2l::LdsOptIn flag;
      int dev = 0, n_cu = 0;
      int rc = s2l::current_device_cus(&dev, &n_cu);
      if (rc) return rc;
      if ((rc = s2l::ensure_dynamic_lds(reinterpret_cast<const void*>(s2l::audio_encode_lds_kernel), s2l::kEncLds, flag, dev))) return rc;
      hipLaunchKernelGGL(s2l::audio_encode_lds_kernel, dim3(1), dim3(256), s2l::kEncLds, static_cast<hipStream_t>(stream), packed, windows, feat,
                         (int64_t)1);
      hipLaunchKernelGGL((s2l::audio_encode_kernel<s2l::kFB, true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                         packed, windows, feat, n);
    } else {
      hipLaunchKernelGGL(s2l::audio_encode_kernel<s2l::kFB>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                         packed, windows, feat, n);
    }
  }
  return (int)hipGetLastError();
}

// Encoder + frame vectors of a call of fewer than four frames in ONE launch (TalkingFace.render_clip's one-frame-per-call path, inference.py:129-159):
// windows [n,16,29], frame_idx [n] -> q0 / q5 [n,256] and, if feat != null, the features [n,64] -- the bits of s2l_audio_encode followed by
// s2l_frame_vectors.  n >= 4: S2L_E_SIZE (clips use the two entry points: their kernels serve four frames per workgroup).
extern "C" int s2l_frame_front(const float* packed, const float* windows, const int64_t* frame_idx, float* feat, float* q0, float* q5,
                               int64_t n, s2l_stream_t stream) {
  if (n < 0 || n >= s2l::kFB) return S2L_E_SIZE;
  if (n == 0) return S2L_OK;
  if (!packed || !windows || !frame_idx || !q0 || !q5) return S2L_E_NULL;
  static s2l::LdsOptIn flag;
  int dev = 0, n_cu = 0;
  int rc = s2l::current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  if ((rc = s2l::ensure_dynamic_lds(reinterpret_cast<const void*>(s2l::frame_front_kernel), s2l::kEncLds, flag, dev))) return rc;
  hipLaunchKernelGGL(s2l::frame_front_kernel, dim3((unsigned)n, s2l::kFVSplit), dim3(256), s2l::kEncLds, static_cast<hipStream_t>(stream), packed,
                     windows, frame_idx, feat, q0, q5, n);
  return (int)hipGetLastError();
}

extern "C" int s2l_frame_vectors(const float* packed, const float* feat, const int64_t* frame_idx, float* q0, float* q5,
                                 int64_t n, s2l_stream_t stream) {
  if (n < 0) return S2L_E_SIZE;
  if (n == 0) return S2L_OK;
  if (!packed || !feat || !frame_idx || !q0 || !q5) return S2L_E_NULL;
  if (n < s2l::kFB) {
#ifdef S2L_FV_ONE_BLOCK
    hipLaunchKernelGGL(s2l::frame_vectors_kernel<1>, dim3((unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), packed, feat,
                       frame_idx, q0, q5, n);
#else
    hipLaunchKernelGGL(s2l::frame_vectors_split_kernel, dim3((unsigned)n, s2l::kFVSplit), dim3(256), 0, static_cast<hipStream_t>(stream), packed,
                       feat, frame_idx, q0, q5, n);
#endif
  } else {
    const int64_t blocks = (n + s2l::kFB - 1) / s2l::kFB;
    hipLaunchKernelGGL(s2l::frame_vectors_kernel<s2l::kFB>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       packed, feat, frame_idx, q0, q5, n);
  }
  return (int)hipGetLastError();
}

extern "C" int s2l_pixel_tables(const float* packed, const float* coords, float* p0, float* p5, int64_t hw,
                                s2l_stream_t stream) {
  if (hw < 0) return S2L_E_SIZE;
  if (hw == 0) return S2L_OK;
  if (!packed || !coords || !p0 || !p5) return S2L_E_NULL;
  if (s2l::misaligned16(p0) || s2l::misaligned16(p5)) return S2L_E_ALIGN;
  const int64_t blocks = (hw + s2l::kPB - 1) / s2l::kPB;
  hipLaunchKernelGGL(s2l::pixel_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     packed, coords, p0, p5, hw);
  return (int)hipGetLastError();
}

extern "C" int64_t s2l_audio_grad_floats(void) { return s2l::kAudioGradFloats; }

// Encoder backward: windows [B,16,29], dfeat [B,64] -> grads [s2l_audio_grad_floats()] = the 12 encoder
// tensors in state-dict order and torch layout (encoder_conv.{0,2,4,6}.{weight,bias}, encoder_fc1.{0,2}.*).
// work: ceil(B/4) * s2l_audio_grad_floats() floats.
extern "C" int s2l_audio_backward(const float* packed, const float* windows, const float* dfeat, float* work, float* grads,
                                  int64_t n, s2l_stream_t stream) {
  if (n <= 0) return S2L_E_SIZE;
  if (!packed || !windows || !dfeat || !work || !grads) return S2L_E_NULL;
  const int blocks = (int)((n + s2l::kFB - 1) / s2l::kFB);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(s2l::audio_backward_kernel, dim3(blocks), dim3(1024), 0, st, packed, windows, dfeat, work, n);
  hipLaunchKernelGGL(s2l::reduce_rows_kernel, dim3((s2l::kAudioGradFloats + 255) / 256), dim3(256), 0, st, work, grads,
                     blocks, (int)s2l::kAudioGradFloats);
  return (int)hipGetLastError();
}

namespace s2l {
int launch_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                      hipStream_t st) {
  const int64_t blocks = (n_rows + 1) / 2;
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, uv_audio, (float)time_index, x,
                     n_rows);
  return (int)hipGetLastError();
}
}  // namespace s2l

// x [N,128] = [E(uv) 42 | audio 64 | PE(time) 20 | 0 0] for arbitrary rows: the first half of s2l_rgb_forward on its own, for
// callers that keep x and the activations for a backward pass (speech2lip_amd.autograd).
extern "C" int s2l_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                              s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed || !uv_audio || !x) return S2L_E_NULL;
  if (s2l::misaligned16(packed) || s2l::misaligned16(x)) return S2L_E_ALIGN;
  return s2l::launch_embed_rows(packed, uv_audio, time_index, x, n_rows, static_cast<hipStream_t>(stream));
}
