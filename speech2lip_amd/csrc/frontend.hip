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

__device__ inline float lrelu(float x) { return x > 0.f ? x : 0.02f * x; }

// Conv1d(k=3, stride=2, pad=1) + LeakyReLU(0.02) over kFB frames held in LDS.
// xin [kFB][CIN][TIN], yout [kFB][COUT][TIN/2], wT [CIN][3][COUT].
template <int CIN, int COUT, int TIN>
__device__ inline void conv_stage(const float* __restrict__ wT, const float* __restrict__ b, const float* xin,
                                  float* yout) {
  constexpr int TOUT = TIN / 2;
  for (int item = threadIdx.x; item < kFB * COUT * TOUT; item += blockDim.x) {
    const int o = item % COUT;
    const int tau = (item / COUT) % TOUT;
    const int fb = item / (COUT * TOUT);
    const float* x = xin + fb * CIN * TIN;
    float acc = b[o];
    for (int c = 0; c < CIN; ++c) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int t = 2 * tau + k - 1;
        if (t >= 0 && t < TIN) acc = fmaf(wT[(c * 3 + k) * COUT + o], x[c * TIN + t], acc);
      }
    }
    yout[(fb * COUT + o) * TOUT + tau] = lrelu(acc);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void audio_encode_kernel(const float* __restrict__ packed,
                                                          const float* __restrict__ windows,
                                                          float* __restrict__ feat, int64_t n) {
  __shared__ float x0[kFB * 29 * 16];
  __shared__ float y1[kFB * 32 * 8];
  __shared__ float y2[kFB * 32 * 4];
  __shared__ float y3[kFB * 64 * 2];
  __shared__ float y4[kFB * 64];
  __shared__ float f1[kFB * 64];
  const int64_t f0 = (int64_t)blockIdx.x * kFB;
  // windows [f][t 16][c 29] -> x0 [fb][c][t]   (the permute of tf_nerf.py:207)
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
  {  // Linear(64,64) + LeakyReLU, Linear(64,64): thread = (frame, output)
    const int fb = threadIdx.x >> 6, o = threadIdx.x & 63;
    float acc = packed[OFF_F0B + o];
    for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F0W + k * 64 + o], y4[fb * 64 + k], acc);
    f1[fb * 64 + o] = lrelu(acc);
    __syncthreads();
    acc = packed[OFF_F2B + o];
    for (int k = 0; k < 64; ++k) acc = fmaf(packed[OFF_F2W + k * 64 + o], f1[fb * 64 + k], acc);
    if (f0 + fb < n) feat[(f0 + fb) * 64 + o] = acc;
  }
}

// q0[f] = W0 (Wa a_f + Wt PE(idx_f) + bsum0) + b0 ; q5[f] likewise with the skip projections.
// 256 threads: thread n owns output feature n for kFB frames.
__global__ __launch_bounds__(256) void frame_vectors_kernel(const float* __restrict__ packed,
                                                           const float* __restrict__ feat,
                                                           const int64_t* __restrict__ frame_idx,
                                                           float* __restrict__ q0, float* __restrict__ q5, int64_t n) {
  __shared__ float a[kFB][64];
  __shared__ float pe[kFB][20];
  __shared__ float s0[kFB][256];
  __shared__ float s5[kFB][256];
  const int64_t f0 = (int64_t)blockIdx.x * kFB;
  const int tid = threadIdx.x;
  {
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
  for (int k = 0; k < 64; ++k) {
    const float w0 = packed[OFF_WAT + k * 256 + tid], w5 = packed[OFF_WAST + k * 256 + tid];
#pragma unroll
    for (int fb = 0; fb < kFB; ++fb) {
      acc0[fb] = fmaf(w0, a[fb][k], acc0[fb]);
      acc5[fb] = fmaf(w5, a[fb][k], acc5[fb]);
    }
  }
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
  const int64_t blocks = (n + s2l::kFB - 1) / s2l::kFB;
  hipLaunchKernelGGL(s2l::audio_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     packed, windows, feat, n);
  return (int)hipGetLastError();
}

extern "C" int s2l_frame_vectors(const float* packed, const float* feat, const int64_t* frame_idx, float* q0, float* q5,
                                 int64_t n, s2l_stream_t stream) {
  if (n < 0) return S2L_E_SIZE;
  if (n == 0) return S2L_OK;
  if (!packed || !feat || !frame_idx || !q0 || !q5) return S2L_E_NULL;
  const int64_t blocks = (n + s2l::kFB - 1) / s2l::kFB;
  hipLaunchKernelGGL(s2l::frame_vectors_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     packed, feat, frame_idx, q0, q5, n);
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

namespace s2l {
int launch_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                      hipStream_t st) {
  const int64_t blocks = (n_rows + 1) / 2;
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, uv_audio, (float)time_index, x,
                     n_rows);
  return (int)hipGetLastError();
}
}  // namespace s2l
