// Post-fusion U-Net (the first "next" row after the hot path): eval-mode
// SimpleUnetLight (src/face_simple/models/SimpleUnetLight.py:16-111), the network
// TalkingFace.post_fusion2_onlylip_light applies to the composite (tf_nerf.py:387).
//
//   x1 = DoubleConv(3->64)(x)                      @ H x W
//   x2 = DoubleConv(64->128)(maxpool2(x1))         @ H/2
//   x3 = DoubleConv(128->128)(maxpool2(x2))        @ H/4
//   y  = DoubleConv(256->64, mid 128)(cat[x2, up2(x3)])
//   y  = DoubleConv(128->64, mid 64)(cat[x1, up2(y)])
//   out = conv1x1(64->3)(y)
// DoubleConv = (conv3x3 no bias -> BatchNorm -> ReLU) x 2; BatchNorm is folded into the weights at
// pack time (eval mode: running statistics).  up2 = bilinear x2 with align_corners=True, zero-padded
// to the skip's size.  Activations are NHWC fp32 in HBM.
//
// conv3x3 is an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32): a workgroup computes a 16x16
// pixel tile x 64 output channels; input channels go through LDS 16 at a time (18x18 halo tile,
// 20.7 KB) together with that chunk's weights in A-operand order (9 taps x 4 M-blocks, 36.9 KB);
// per tap a wave issues 4 + 4 ds_read_b128 for 64 MFMAs.  57.6 KB of LDS per workgroup -> two
// workgroups per CU overlap one another's staging.  The concat is virtual (two input pointers), the
// final 1x1 convolution is fused into the last conv's epilogue.
#include "s2l_common.h"
#ifdef S2L_WITH_REFERENCE_KERNELS
#include "conv16.h"      // (the generated-assembly split convolution: libs2l_hip_ref.so only)
#endif
#include "convh.h"
#include "unet_layout.h"

namespace s2l {

struct UnetTensors {
  const float* w[10];
  const float* gamma[10];
  const float* beta[10];
  const float* mean[10];
  const float* var[10];
  const float* outw;
  const float* outb;
};

// one thread per packed weight element of a 3x3 layer (layer >= 1): BatchNorm folded in
// eps < 0: RAW weights (no BatchNorm fold) for the train-mode network, whose BatchNorm uses batch statistics
// The pack kernels take the layer from blockIdx.y (+ 1) and the form from blockIdx.z: ONE launch packs all nine 3x3 layers, forward and
// transposed (a training net re-packs after every optimizer step: 36 launches of a few microseconds each per iteration before).
constexpr int64_t kPackMaxFloats = 2 * 16 * kChunkFloats;      // the largest layer (256 -> 128); smaller layers' surplus blocks return
__device__ __forceinline__ void unet_pack_conv(UnetTensors t, int layer, float* __restrict__ packed, float eps) {
  const int cin = kUnetConvs[layer].cin, cout = kUnetConvs[layer].cout;
  const int64_t n = (int64_t)(cout / 64) * (cin / 16) * kChunkFloats;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int ks = e & 3, lane = (e >> 2) & 63, mb = (e >> 8) & 3;
  const int tap = (int)((e >> 10) % 9);
  const int64_t chunk = e / kChunkFloats;
  const int cc = (int)(chunk % (cin / 16)), ct = (int)(chunk / (cin / 16));
  const int co = ct * 64 + mb * 16 + (lane & 15);
  const int ci = cc * 16 + 4 * (lane >> 4) + ks;
  const float scale = eps < 0.f ? 1.f : t.gamma[layer][co] / sqrtf(t.var[layer][co] + eps);
  packed[unet_w_off(layer) + e] = t.w[layer][((int64_t)co * cin + ci) * 9 + tap] * scale;
}

// the transposed twin: output rows = the layer's INPUT channels, k = its OUTPUT channels, tap t reads forward tap 8 - t
__device__ __forceinline__ void unet_pack_conv_T(UnetTensors t, int layer, float* __restrict__ packed, float eps) {
  const int cin = kUnetConvs[layer].cin, cout = kUnetConvs[layer].cout;      // forward roles
  const int64_t n = (int64_t)(cin / 64) * (cout / 16) * kChunkFloats;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int ks = e & 3, lane = (e >> 2) & 63, mb = (e >> 8) & 3;
  const int tap = (int)((e >> 10) % 9);
  const int64_t chunk = e / kChunkFloats;
  const int cc = (int)(chunk % (cout / 16)), ct = (int)(chunk / (cout / 16));
  const int ci = ct * 64 + mb * 16 + (lane & 15);          // row of the transposed GEMM = forward input channel
  const int co = cc * 16 + 4 * (lane >> 4) + ks;           // k = forward output channel
  const float scale = eps < 0.f ? 1.f : t.gamma[layer][co] / sqrtf(t.var[layer][co] + eps);
  packed[unet_wT_off(layer) + e] = t.w[layer][((int64_t)co * cin + ci) * 9 + (8 - tap)] * scale;
}

// grid (blocks of the largest layer, 9 layers, 2 forms)
__global__ void unet_pack_convs_kernel(UnetTensors t, float* __restrict__ packed, float eps) {
  if (blockIdx.z == 0) unet_pack_conv(t, blockIdx.y + 1, packed, eps);
  else unet_pack_conv_T(t, blockIdx.y + 1, packed, eps);
}

__global__ void unet_pack_misc(UnetTensors t, float* __restrict__ packed, float eps) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt = gridDim.x * blockDim.x;
  for (int i = tid; i < 64 * 27; i += nt) {   // first conv, plain [co][ci*9 + tap], folded
    const int co = i / 27;
    packed[unet_w_off(0) + i] = t.w[0][i] * (eps < 0.f ? 1.f : t.gamma[0][co] / sqrtf(t.var[0][co] + eps));
  }
  int64_t off = kUnetBiasOff;
  for (int l = 0; l < 10; ++l) {
    for (int co = tid; co < kUnetConvs[l].cout; co += nt) {
      const float scale = eps < 0.f ? 0.f : t.gamma[l][co] / sqrtf(t.var[l][co] + eps);
      packed[off + co] = eps < 0.f ? 0.f : t.beta[l][co] - t.mean[l][co] * scale;
    }
    off += kUnetConvs[l].cout;
  }
  for (int i = tid; i < 192; i += nt) packed[kUnetOutW + i] = t.outw[i];
  for (int i = tid; i < 4; i += nt) packed[kUnetOutB + i] = i < 3 ? t.outb[i] : 0.f;
}

// ---- bf16 operand form of the 3x3 layers 1..9 (training chain in the precision BASELINE config 5 names) -------------------
// conv3x3_bf16_kernel runs the same implicit GEMM on v_mfma_f32_32x32x16_bf16: weights and the staged input tile are bf16,
// accumulation, bias / ReLU / gate and every tensor in HBM stay fp32.  A chunk = (64 output channels, 32 input channels):
// 9 taps x 2 k-steps x 2 M-blocks x 64 lanes x 8 bf16; lane l of an A quad holds W[row 32 mb + (l & 31)][k 16 s + 8 (l >> 5) + j].
// Split-bf16 ("bf16x3") operand form of the forward 3x3 layers: every fp32 operand x is carried as hi = bf16(x) and
// lo = bf16(x - hi) (x - hi - lo is <= 2^-17 |x|), and a product a b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-16 relative operand error instead of bf16's 2^-8, three MFMAs at 16x the
// fp32-MFMA rate.  A chunk = (64 output channels, 16 input channels) in the SAME LDS shape as the plain bf16 chunk:
// [tap 9][part: hi, lo][M-block 2][lane 64][8 bf16], lane l of an A quad holds W[row 32 mb + (l & 31)][k 8 (l >> 5) + j].
__host__ __device__ constexpr int64_t unet_w16x3_off(int layer) {
  int64_t off = 0;
  for (int l = 1; l < layer; ++l) off += (int64_t)(kUnetConvs[l].cout / 64) * (kUnetConvs[l].cin / 16) * kChunk16Halves;
  return off;
}
constexpr int64_t kUnetPacked16x3Halves = unet_w16x3_off(10);

__device__ __forceinline__ uint16_t to_bf16(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }   // round to nearest even

// one thread per packed bf16 element; transposed = the input-gradient form (rows = forward input channels, taps mirrored)
// grid (blocks of the largest layer, 9 layers, 2 forms)
__global__ void unet_pack_conv16(UnetTensors t, uint16_t* __restrict__ packed16, float eps) {
  const int layer = blockIdx.y + 1, transposed = blockIdx.z;
  const int cin = kUnetConvs[layer].cin, cout = kUnetConvs[layer].cout;
  const int rows = transposed ? cin : cout, kdim = transposed ? cout : cin;
  const int64_t n = (int64_t)(rows / 64) * (kdim / 32) * kChunk16Halves;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int j = e & 7, lane = (e >> 3) & 63, mb = (e >> 9) & 1, ks = (e >> 10) & 1;
  const int tap = (int)((e >> 11) % 9);
  const int64_t chunk = e / kChunk16Halves;
  const int cc = (int)(chunk % (kdim / 32)), ct = (int)(chunk / (kdim / 32));
  const int row = ct * 64 + mb * 32 + (lane & 31), k = cc * 32 + ks * 16 + 8 * (lane >> 5) + j;
  const int co = transposed ? k : row, ci = transposed ? row : k;
  const float scale = eps < 0.f ? 1.f : t.gamma[layer][co] / sqrtf(t.var[layer][co] + eps);
  packed16[(transposed ? unet_wT16_off(layer) : unet_w16_off(layer)) + e] =
      to_bf16(t.w[layer][((int64_t)co * cin + ci) * 9 + (transposed ? 8 - tap : tap)] * scale);
}

typedef _Float16 h2v_ __attribute__((ext_vector_type(2)));
// split form (forward only): one thread per packed element, hi and lo parts of the BatchNorm-folded fp32 weight
__global__ void unet_pack_conv16x3(UnetTensors t, uint16_t* __restrict__ packed, float eps) {      // grid (blocks of the largest layer, 9 layers)
  const int layer = blockIdx.y + 1;
  const int cin = kUnetConvs[layer].cin, cout = kUnetConvs[layer].cout;
  const int64_t n = (int64_t)(cout / 64) * (cin / 16) * kChunk16Halves;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int j = e & 7, lane = (e >> 3) & 63, mb = (e >> 9) & 1, part = (e >> 10) & 1;
  const int tap = (int)((e >> 11) % 9);
  const int64_t chunk = e / kChunk16Halves;
  const int cc = (int)(chunk % (cin / 16)), ct = (int)(chunk / (cin / 16));
  const int co = ct * 64 + mb * 32 + (lane & 31), ci = cc * 16 + 8 * (lane >> 5) + j;
  const float scale = t.gamma[layer][co] / sqrtf(t.var[layer][co] + eps);
  const float w = t.w[layer][((int64_t)co * cin + ci) * 9 + tap] * scale;      // the fp32 kernel's folded weight, bit for bit
  // (parts as IEEE halves, round toward zero: the same v_cvt_pkrtz_f16_f32 the kernels apply to the activations)
  const uint32_t hi2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(w, 0.f));
  const float hif = (float)__builtin_bit_cast(h2v_, hi2)[0];
  h2v_ lov;
  lov[0] = (_Float16)(w - hif);      // (the lo part rounds to nearest: unbiased)
  lov[1] = (_Float16)0.f;
  const uint32_t lo2 = __builtin_bit_cast(uint32_t, lov);
  packed[unet_w16x3_off(layer) + e] = (uint16_t)((part == 0 ? hi2 : lo2) & 0xffffu);
}

// ---- first conv: 3 -> 64 on MFMA (0.5 % of the FLOPs, but 64 MB of output per frame: write-bound) -----------------------
// D[co][pixel] = sum_k W[co][k] in[k][pixel], k = c*9 + tap (27, zero-padded to 28 = 7 k-steps of v_mfma_f32_16x16x4_f32): the
// same fma chain, in the same order, as a scalar loop over k.  A wave owns 16 consecutive pixels per iteration: lane
// (q = lane>>4, px = lane&15) gathers the 7 inputs k = 4j + q of its pixel (28 loads per pixel instead of 27 per output-channel
// quad), the 28 weight operands per lane stay in registers, and each lane stores 4 x 16 bytes (4 consecutive channels per M-block).
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int H, int W) {
  // A workgroup owns a 16-row x 64-column tile (grid: column tiles, row tiles, frames); its 18 x 66 x 3 input values go through LDS once
  // (coalesced) and the 7 per-lane inputs of a 16-pixel group are LDS reads: as 28 gathered global loads per pixel group the kernel was
  // bound by the texture path (0.38 ms per 16 frames at 500 x 500 against 0.2 of output writes).  Same fma chains: the same bits.
  __shared__ float tile[18 * 66 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, px = lane & 15;
  const int64_t frame = blockIdx.z;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  const float* xf = x + frame * (int64_t)H * W * 3;
  float* yf = y + frame * (int64_t)H * W * 64;
  {      // (the thread's 14 tile values are fetched together, then stored: one conditional load after the other is latency-bound)
    float tv[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int r = i / 198, rem = i - r * 198, cx = rem / 3;
      const int gy = y0 - 1 + r, gx = x0 - 1 + cx;
      const bool in = i < 18 * 66 * 3 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float v = xf[in ? ((int64_t)gy * W + gx) * 3 + (rem - cx * 3) : 0];
      tv[k] = in ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < 18 * 66 * 3) tile[i] = tv[k];
    }
  }
  // A operands: lane holds W[co = 16 mb + px][k = 4j + q]
  float wa[4][7];
  f4 bias[4];
  int koff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int k = 4 * j + q, c = k / 9, t = k - 9 * c;
    koff[j] = k < 27 ? ((t / 3) * 66 + t % 3) * 3 + c : -1;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int k = 4 * j + q;
      wa[mb][j] = k < 27 ? w[(16 * mb + px) * 27 + k] : 0.f;
    }
    bias[mb] = b ? *reinterpret_cast<const f4*>(b + 16 * mb + 4 * q) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * wave + rr, gy = y0 + r;
    if (gy >= H) break;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cx = 16 * u + px, gx = x0 + cx;
      float in[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) in[j] = koff[j] >= 0 ? tile[(r * 66 + cx) * 3 + koff[j]] : 0.f;
      if (x0 + 16 * u >= W) break;      // (wave-uniform)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mb][j], in[j], acc, 0, 0, 0);
        if (gx < W) {
          f4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = b ? fmaxf(acc[e] + bias[mb][e], 0.f) : acc[e];   // b == NULL: raw pre-BatchNorm output
          *reinterpret_cast<f4*>(yf + ((int64_t)gy * W + gx) * 64 + 16 * mb + 4 * q) = o;
        }
      }
    }
  }
}

// ---- conv3x3 implicit GEMM ---------------------------------------------------------------------------
struct ConvArgs {
  const float* inA;   // [F,H,W,CA]
  const float* inB;   // [F,H,W,CB] or null (virtual concat: channels of A first)
  const float* w;     // packed chunks [cout/64][cin/16][kChunkFloats]
  const float* bias;  // [cout]
  float* out;         // [F,H,W,cout]                   (unused when FUSE_OUT)
  const float* outw;  // FUSE_OUT: [3][64], outb [3], out3 [F,H,W,3]
  const float* outb;
  float* out3;
  float* pool;        // or null: MaxPool2d(2) of the output, [F,H/2,W/2,cout], written by the same epilogue
  const float* gate;  // or null: [F,H,W,cout]; the output is zeroed where gate <= 0 (ReLU mask of the backward pass)
  int CA, CB, cout, H, W, tiles_x, tiles_y, n_ct;
  int relu;           // 1: ReLU in the epilogue (forward); 0: linear (input-gradient convolutions)
  const uint16_t* w16;   // conv3x3_bf16_kernel: packed bf16 chunks [cout/64][cin/32][kChunk16Halves]; split form: [cout/64][cin/16][...]
  int split;             // 1: w16 is the split-bf16 (hi, lo) form
  int n_frames_asm;      // conv3x3_asm_kernel: frames of this launch (its grid is the CU count, not the tile count)
#ifdef S2L_EXP_TRACE
  long long* trace;   // experiment builds (tools/trace_conv.py): [workgroup][24] timestamps of this launch
#endif
};

#ifdef S2L_EXP_TRACE
static long long* g_conv_trace = nullptr;     // [launch][8192 workgroups][24]
static int g_conv_launch = 0;
extern "C" int s2l_debug_set_conv_trace(void* p) { g_conv_trace = static_cast<long long*>(p); g_conv_launch = 0; return 0; }
#define CONV_TRACE(slot)                                                                                              \
  do {                                                                                                                \
    if (a.trace && threadIdx.x == 0)                                                                                  \
      a.trace[(blockIdx.x + gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z)) * 24 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#define CONV_TRACE_ARGS(a, grid)                                                                              \
  (a).trace = (g_conv_trace && (int64_t)(grid).x * (grid).y * (grid).z <= 8192) ? g_conv_trace + (int64_t)(g_conv_launch++) * 8192 * 24 : nullptr
#else
#define CONV_TRACE(slot) do { } while (0)
#define CONV_TRACE_ARGS(a, grid) do { } while (0)
#endif

#ifndef S2L_CONV_ASM
#define S2L_CONV_ASM 1
#endif

__device__ __forceinline__ f4 mfma16u(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <bool FUSE_OUT>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_in[18 * 18 * 16];
  __shared__ __attribute__((aligned(16))) float lds_w[kChunkFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, px = lane & 15;
  const int tx = blockIdx.x, ty = blockIdx.y;
  const int ct = blockIdx.z % a.n_ct;
  const int64_t frame = blockIdx.z / a.n_ct;
  const int x0 = tx * 16, y0 = ty * 16;
  const int cin = a.CA + a.CB;
  const int nchunks = cin / 16;
  const float* inA = a.inA + frame * (int64_t)a.H * a.W * a.CA;
  const float* inB = a.inB ? a.inB + frame * (int64_t)a.H * a.W * a.CB : nullptr;

  CONV_TRACE(0);
#ifdef S2L_EXP_TRACE
  if (a.trace && threadIdx.x == 0) {
    long long* t = a.trace + (blockIdx.x + gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z)) * 24;
    t[22] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    t[23] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
  }
#endif
  f4 acc[4][4];   // [M-block][pixel group = tile row 4*wave + g]
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const f4 b = a.bias ? *reinterpret_cast<const f4*>(a.bias + ct * 64 + mb * 16 + 4 * q) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[mb][g] = b;
  }

  // Register-prefetch pipeline over the 16-channel chunks: the global loads of chunk cc+1 are in
  // flight while chunk cc's 576 MFMAs run; they are committed to LDS between two barriers.
  constexpr int kInQuads = 18 * 18 * 4;                 // f4 elements of the halo tile
  constexpr int kInPer = (kInQuads + 255) / 256;        // 6 (the last pass is partial)
  constexpr int kWPer = kChunkFloats / 4 / 256;         // 9
  f4 pin[kInPer], pw[kWPer];
  auto fetch = [&](int cc) {
    const bool fromA = cc * 16 < a.CA;
    const float* src = fromA ? inA : inB;
    const int C = fromA ? a.CA : a.CB;
    const int coff = fromA ? cc * 16 : cc * 16 - a.CA;
#pragma unroll
    for (int k = 0; k < kInPer; ++k) {
      const int i = threadIdx.x + k * 256;
      const int pi = i >> 2, qq = i & 3;
      const int gy = y0 - 1 + pi / 18, gx = x0 - 1 + pi % 18;
      pin[k] = (f4){0.f, 0.f, 0.f, 0.f};
      if (i < kInQuads && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
        pin[k] = *reinterpret_cast<const f4*>(src + ((int64_t)gy * a.W + gx) * C + coff + 4 * qq);
    }
    const f4* wsrc = reinterpret_cast<const f4*>(a.w + ((int64_t)ct * nchunks + cc) * kChunkFloats);
#pragma unroll
    for (int k = 0; k < kWPer; ++k) pw[k] = wsrc[threadIdx.x + k * 256];
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kInPer; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i < kInQuads) *reinterpret_cast<f4*>(lds_in + (i >> 2) * 16 + 4 * (i & 3)) = pin[k];
    }
#pragma unroll
    for (int k = 0; k < kWPer; ++k) reinterpret_cast<f4*>(lds_w)[threadIdx.x + k * 256] = pw[k];
  };
  fetch(0);
  commit();
  __syncthreads();
  CONV_TRACE(1);
  for (int cc = 0; cc < nchunks; ++cc) {
    if (cc + 1 < nchunks) fetch(cc + 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t % 3;
      f4 B4[4], A4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        B4[g] = *reinterpret_cast<const f4*>(lds_in + ((4 * wave + g + dy) * 18 + px + dx) * 16 + 4 * q);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) A4[mb] = *reinterpret_cast<const f4*>(lds_w + ((t * 4 + mb) * 64 + lane) * 4);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[mb][g] = mfma16u(A4[mb][ks], B4[g][ks], acc[mb][g]);
    }
    CONV_TRACE(2 + cc);
    __syncthreads();            // everyone is done reading chunk cc
    if (cc + 1 < nchunks) {
      commit();
      __syncthreads();
    }
  }

  // epilogue: ReLU (bias is already in), store NHWC; D[row = 4q + r -> channel][col = px -> pixel]
  const int gx = x0 + px;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int gy = y0 + 4 * wave + g;
    const bool ok = gy < a.H && gx < a.W;
    const int64_t pix = frame * (int64_t)a.H * a.W + (int64_t)gy * a.W + gx;
    if (FUSE_OUT) {
      // outc (conv1x1 64 -> 3) on the 64 channels of this pixel: lane partial over its 16, reduce over q
      float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h = fmaxf(acc[mb][g][r], 0.f);
          const int c = mb * 16 + 4 * q + r;
#pragma unroll
          for (int o = 0; o < 3; ++o) p3[o] = fmaf(a.outw[o * 64 + c], h, p3[o]);
        }
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        p3[o] += __shfl_xor(p3[o], 16);
        p3[o] += __shfl_xor(p3[o], 32);
      }
      if (q == 0 && ok) {
        float* o3 = a.out3 + pix * 3;
        o3[0] = p3[0] + a.outb[0];
        o3[1] = p3[1] + a.outb[1];
        o3[2] = p3[2] + a.outb[2];
      }
      if (a.out && ok) {   // training: the last hidden activation is kept for the backward pass
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          f4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = fmaxf(acc[mb][g][r], 0.f);
          *reinterpret_cast<f4*>(a.out + pix * a.cout + ct * 64 + mb * 16 + 4 * q) = h;
        }
      }
    } else if (ok) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f4 h;
        float* dst = a.out + pix * a.cout + ct * 64 + mb * 16 + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = a.relu ? fmaxf(acc[mb][g][r], 0.f) : acc[mb][g][r];
        if (a.gate) {
          const f4 gt = *reinterpret_cast<const f4*>(a.gate + (dst - a.out));
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = gt[r] > 0.f ? h[r] : 0.f;
        }
        *reinterpret_cast<f4*>(dst) = h;
      }
    }
  }
  if (!FUSE_OUT && a.pool) {
    // MaxPool2d(2) in the same epilogue: tile rows 4*wave+g pair up as (0,1), (2,3) in registers, columns px, px^1 across lanes
    const int H2 = a.H / 2, W2 = a.W / 2;
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int py2 = (y0 + 4 * wave + 2 * gp) / 2, px2 = gx / 2;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fmaxf(fmaxf(acc[mb][2 * gp][r], acc[mb][2 * gp + 1][r]), 0.f);
          m[r] = fmaxf(v, __shfl_xor(v, 1));
        }
        if (!(px & 1) && py2 < H2 && px2 < W2)
          *reinterpret_cast<f4*>(a.pool + ((frame * H2 + py2) * (int64_t)W2 + px2) * a.cout + ct * 64 + mb * 16 + 4 * q) = m;
      }
    }
  }
  CONV_TRACE(20);
}

// ---- upsampling (MaxPool2d(2) lives in the conv epilogue) ------------------------------------------------------------------------------
// bilinear x2, align_corners=True, then zero padding to (Ho, Wo) (SimpleUnetLight.py:57-66)
// Window form (training on a crop of the frame, s2l_unet_forward_saved_window): x and y are CROPS of the full low-resolution /
// output tensors; the bilinear source position is computed in FULL-frame coordinates (align_corners=True makes the scale depend on
// the full size) and then shifted into the crop, so every value that does not depend on data outside the crop equals the
// full-frame value bit for bit.  Without a window: hf = h, wf = w, Hof = Ho, Wof = Wo, origins 0.
// Elementwise kernels over [F, rows, cols, C] tensors take a (ceil(cols * C/4 / 256), rows, F) grid: a thread finds its channel
// quad and column with one 32-bit division, its row and frame in blockIdx (a linear index would cost three 64-bit divisions).
__device__ __forceinline__ bool quad_of_thread(int cols, int cq, int rows, int& c4, int& x, int& y, int64_t& f, int64_t& i) {
  const unsigned ix = blockIdx.x * 256u + threadIdx.x;
  if (ix >= (unsigned)cols * (unsigned)cq) return false;
  x = (int)(ix / (unsigned)cq);
  c4 = (int)(ix - (unsigned)x * (unsigned)cq);
  y = (int)blockIdx.y;
  f = blockIdx.z;
  i = ((f * rows + y) * (int64_t)cols + x) * cq + c4;
  return true;
}
static dim3 quad_grid(int cols, int C, int rows, int64_t F) {
  return dim3((unsigned)(((int64_t)cols * (C / 4) + 255) / 256), (unsigned)rows, (unsigned)F);
}

struct UpWin {
  int hf, wf, Hof, Wof;     // full low-resolution size, full output size
  int oyi, oxi, oyo, oxo;   // origin of the input crop / output crop inside the full tensors
};
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int C,
                                                       int Ho, int Wo, UpWin win) {
  int c4, xo, yo;
  int64_t f, i;
  if (!quad_of_thread(Wo, C / 4, Ho, c4, xo, yo, f, i)) return;
  const int padT = (win.Hof - 2 * win.hf) / 2, padL = (win.Wof - 2 * win.wf) / 2;
  const int yu = yo + win.oyo - padT, xu = xo + win.oxo - padL;
  f4 o = (f4){0.f, 0.f, 0.f, 0.f};
  if ((unsigned)yu < (unsigned)(2 * win.hf) && (unsigned)xu < (unsigned)(2 * win.wf)) {
    const float sy = 2 * win.hf > 1 ? (float)(win.hf - 1) / (float)(2 * win.hf - 1) : 0.f;
    const float sx = 2 * win.wf > 1 ? (float)(win.wf - 1) / (float)(2 * win.wf - 1) : 0.f;
    const float fy = sy * (float)yu, fx = sx * (float)xu;
    const int y0f = (int)fy, x0f = (int)fx;
    const int y1f = y0f + (y0f < win.hf - 1 ? 1 : 0), x1f = x0f + (x0f < win.wf - 1 ? 1 : 0);
    const float ly = fy - (float)y0f, lx = fx - (float)x0f;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const int y0 = y0f - win.oyi, y1 = y1f - win.oyi, x0 = x0f - win.oxi, x1 = x1f - win.oxi;
    const float* s = x + f * (int64_t)h * w * C + c4 * 4;
    auto at = [&](int yy, int xx) {          // taps outside the crop only feed outputs nobody reads: zero
      return ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? *reinterpret_cast<const f4*>(s + ((int64_t)yy * w + xx) * C)
                                                                        : (f4){0.f, 0.f, 0.f, 0.f};
    };
    const f4 v00 = at(y0, x0), v01 = at(y0, x1), v10 = at(y1, x0), v11 = at(y1, x1);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = hy * (hx * v00[r] + lx * v01[r]) + ly * (hx * v10[r] + lx * v11[r]);
  }
  *reinterpret_cast<f4*>(y + i * 4) = o;
}
static UpWin no_window(int h, int w, int Ho, int Wo) { return UpWin{h, w, Ho, Wo, 0, 0, 0, 0}; }

// ---- the same convolution with bf16 operands (ConvArgs::w16) ---------------------------------------------------------------
// Tile, grid, inputs and epilogue as conv3x3_kernel.  Input channels go through LDS 32 at a
// time: the 18x18 halo tile as bf16 [pixel][32 channels] with an 80-byte pixel stride (conflict-free ds_read_b128 of the 32
// pixels of an N-block), converted from the fp32 activations on the way in, and the chunk's weights in A-operand order
// (36.9 KB).  Per tap and k-step a wave issues 2 + 2 ds_read_b128 for 4 MFMAs of 32x32x16; wave w owns tile rows 4w..4w+3 as
// two N-blocks of 2 rows x 16 pixels.  16x fewer matrix-pipe cycles than the fp32 form: the kernel is bound by staging and HBM.
#ifndef S2L_UEXP
#define S2L_UEXP 0   // tools/ubench experiments on the bf16-operand convolution kernels only (results wrong): 1 no MFMAs, 2 no activation loads, 4 no weight loads, 8 no LDS commits, 16 no epilogue stores, 32 no requests inside the persistent kernel's chunk loop
#endif
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef short bf8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
constexpr int kPix16 = 40;   // halves per halo pixel: 32 channels + 8 of padding

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf2v v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
// The SPLIT form's parts are IEEE halves (round 4; bf16 parts before: same MFMA rate, 11 + 11 instead of 8 + 8 significant bits --
// RMSE vs the exact fp32 kernels 1.1e-5 -> ~1e-6): hi = f16(x) and lo = f16(x - hi), both by v_cvt_pkrtz_f16_f32 (round toward zero:
// x - hi is exact and has x's sign; out-of-range values saturate at 65504, never inf), in EVERY split kernel and in the weight pack.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
// the lo parts round to NEAREST: with both parts rounded toward zero every operand came out 2^-21 too small on average, a bias the ten
// layers add up coherently (measured: 102 dB instead of the bf16 parts' 99; nearest: see bench_unet)
// (the residual of an operand beyond the half range -- hi saturates at 65504, so x - hi can itself exceed it -- is clamped to the range
// first: a value that large is already wrong in this mode, but it must stay FINITE; an inf part would turn the MFMA's sum into NaN.
// One v_med3_f32 per value; in-range values are untouched.)  Valid operand range of the split mode: |x| < 65504 (include/s2l_hip.h).
__device__ __forceinline__ uint32_t pack_f16x2_rne(float lo, float hi) {
  h2v v;
  v[0] = (_Float16)__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f);
  v[1] = (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float f16_lo(uint32_t p) { return (float)__builtin_bit_cast(h2v, p)[0]; }
__device__ __forceinline__ float f16_hi(uint32_t p) { return (float)__builtin_bit_cast(h2v, p)[1]; }
__device__ __forceinline__ f16v mfma32_f16(u4v a, u4v b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
__device__ __forceinline__ f16v mfma32_bf16(u4v a, u4v b, f16v c) {
#if S2L_UEXP & 1
  c[0] += __uint_as_float(a[0] ^ b[0]);      // keeps the operand reads alive
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
#endif
}

// SPLIT: the split-bf16 form (unet_w16x3_off): 16 input channels per chunk, a halo pixel's 80 bytes hold [hi 16 ch | lo 16 ch | pad],
// the chunk's weights [tap][hi, lo][mb][lane][8] -- the same LDS shapes and the same eight ds_read_b128 per tap, but twelve MFMAs:
// acc += A_hi B_hi + A_hi B_lo + A_lo B_hi (the lo x lo term is below 2^-17 of the product and is dropped).
// WAVES = 8 (the split form's default): ONE workgroup of 512 threads per CU owns a 32 x 16-pixel tile (wave w: rows 4 w .. 4 w + 3,
// as before); the chunk's weights go from global memory straight to LDS (LDS-DMA, no registers, no ds_write pass) into one of TWO
// buffers, a whole chunk ahead of their use, and are staged once per eight waves instead of once per four.  (Ablations of the
// 4-wave form, tools/ubench notes in DESIGN 4.6: loads + commits were 2.4 of the 7.0 ms of convolution time in a 16-frame forward
// -- two workgroups per CU do not hide each other's ~2 us of load latency.)
constexpr int kBf16WideLds = 34 * 18 * kPix16 * 2 + 2 * kChunk16Halves * 2;      // 48 960 + 73 728 bytes
template <bool FUSE_OUT, bool SPLIT = false, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void conv3x3_bf16_kernel(ConvArgs a) {
  constexpr int kThreads = 64 * WAVES, kTileH = 4 * WAVES, kHaloH = kTileH + 2;
  constexpr bool kDma = WAVES == 8;
  static_assert(WAVES == 4 || (WAVES == 8 && SPLIT), "the 8-wave form exists for the split operands");
  uint16_t* lds_in;
  uint16_t* lds_w;        // (kDma: buffer 0; buffer 1 follows it)
  if constexpr (kDma) {
    extern __shared__ __attribute__((aligned(16))) char conv16_smem[];
    lds_in = reinterpret_cast<uint16_t*>(conv16_smem);
    lds_w = lds_in + kHaloH * 18 * kPix16;
  } else {
    __shared__ __attribute__((aligned(16))) uint16_t s_in[18 * 18 * kPix16];
    __shared__ __attribute__((aligned(16))) uint16_t s_w[kChunk16Halves];
    lds_in = s_in;
    lds_w = s_w;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, hh = lane >> 5;
  const int tx = blockIdx.x, ty = blockIdx.y;
  const int ct = blockIdx.z % a.n_ct;
  const int64_t frame = blockIdx.z / a.n_ct;
  const int x0 = tx * 16, y0 = ty * kTileH;
  const int cin = a.CA + a.CB;
  constexpr int kCC = SPLIT ? 16 : 32;                  // input channels per chunk
  const int nchunks = cin / kCC;
  const float* inA = a.inA + frame * (int64_t)a.H * a.W * a.CA;
  const float* inB = a.inB ? a.inB + frame * (int64_t)a.H * a.W * a.CB : nullptr;

  f16v acc[2][2];   // [M-block of 32 channels][N-block: tile rows 4 wave + 2 nb, + 1]
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = a.bias ? a.bias[ct * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;

  constexpr int kQ = kCC / 4;                           // f4 (4 fp32 channels) elements per halo pixel
  constexpr int kInQuads = kHaloH * 18 * kQ;
  constexpr int kInPer = (kInQuads + kThreads - 1) / kThreads;      // 11 (SPLIT: 6, 8 waves: 5; the last pass is partial)
  constexpr int kWPer = kDma ? 1 : kChunk16Halves / 8 / 256;         // 9
  f4 pin[kInPer];
  u4v pw[kWPer];
  // 8-wave form: the chunk's 36 KiB of weights as 36 LDS-DMA instructions of 1 KiB (lane l moves bytes [16 l, 16 l + 16)), wave w
  // issues pieces w, w + 8, ...; they land while the current chunk computes and are waited for (vmcnt) before the publishing barrier
  constexpr int kDmaPieces = kChunk16Halves * 2 / 1024, kDmaPer = (kDmaPieces + WAVES - 1) / WAVES;      // 36; 5 per wave (the last partial)
  auto dma_piece = [&](int cc, int j) {      // piece wave + WAVES j of chunk cc's weights
    if constexpr (kDma) {
      const int p = wave + WAVES * j;
      if (p < kDmaPieces) {
        const char* wsrc = reinterpret_cast<const char*>(a.w16 + ((int64_t)ct * nchunks + cc) * kChunk16Halves);
        char* wdst = reinterpret_cast<char*>(lds_w + (cc & 1) * kChunk16Halves);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(wsrc + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) uint32_t*)(wdst + p * 1024), 16, 0, 0);
      }
    }
  };
  auto dma_weights = [&](int cc) {
#pragma unroll
    for (int j = 0; j < kDmaPer; ++j) dma_piece(cc, j);
  };
  // the halo pixel each of this thread's loads reads is the same for every chunk of the tile: its index (or -1: outside the frame /
  // past the halo) is formed once -- re-deriving it per chunk (two divisions, bounds, a 64-bit product per load) was 2.4 k cycles of
  // issue per chunk and wave, next to 4.4 k of matrix work (tools/trace_conv16.py)
  static_assert(kThreads % kQ == 0, "a thread keeps its channel quad across passes");
  int pixoff[kInPer];
#pragma unroll
  for (int k = 0; k < kInPer; ++k) {
    const int i = threadIdx.x + k * kThreads;
    const int pi = i / kQ;
    const int gy = y0 - 1 + pi / 18, gx = x0 - 1 + pi % 18;
    pixoff[k] = (i < kInQuads && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) ? gy * a.W + gx : -1;
  }
  const int c4x4 = 4 * (threadIdx.x % kQ);
  auto fetch_one = [&](int cc, int k) {      // load k of chunk cc's halo tile into its staging register
    const bool fromA = cc * kCC < a.CA;
    const float* src = (fromA ? inA : inB) + (fromA ? cc * kCC : cc * kCC - a.CA) + c4x4;
    const int C = fromA ? a.CA : a.CB;
    pin[k] = (f4){0.f, 0.f, 0.f, 0.f};
    if (!(S2L_UEXP & 2) && pixoff[k] >= 0) pin[k] = *reinterpret_cast<const f4*>(src + (int64_t)pixoff[k] * C);
  };
  auto fetch = [&](int cc) {
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(cc, k);
    if constexpr (kDma) {
      dma_weights(cc);
    } else {
      const u4v* wsrc = reinterpret_cast<const u4v*>(a.w16 + ((int64_t)ct * nchunks + cc) * kChunk16Halves);
#pragma unroll
      for (int k = 0; k < kWPer; ++k) pw[k] = (S2L_UEXP & 4) ? u4v{(uint32_t)cc, 0u, 0u, 0u} : wsrc[threadIdx.x + k * 256];
    }
  };
  auto commit = [&]() {
    if (S2L_UEXP & 8) return;
#pragma unroll
    for (int k = 0; k < kInPer; ++k) {
      const int i = threadIdx.x + k * kThreads;
      if (i < kInQuads) {
        uint2 h;
        if (SPLIT) {      // hi = f16(x), lo = f16(x - hi): the second 16 "channels" of the pixel
          h.x = pack_f16x2(pin[k][0], pin[k][1]);
          h.y = pack_f16x2(pin[k][2], pin[k][3]);
          uint2 l;
          l.x = pack_f16x2_rne(pin[k][0] - f16_lo(h.x), pin[k][1] - f16_hi(h.x));
          l.y = pack_f16x2_rne(pin[k][2] - f16_lo(h.y), pin[k][3] - f16_hi(h.y));
          *reinterpret_cast<uint2*>(lds_in + (i / kQ) * kPix16 + 16 + 4 * (i % kQ)) = l;
        } else {
          h.x = pack_bf16x2(pin[k][0], pin[k][1]);
          h.y = pack_bf16x2(pin[k][2], pin[k][3]);
        }
        *reinterpret_cast<uint2*>(lds_in + (i / kQ) * kPix16 + 4 * (i % kQ)) = h;
      }
    }
    if constexpr (kDma) {
      __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's weight pieces of the next chunk have landed in the other buffer
    } else {
#pragma unroll
      for (int k = 0; k < kWPer; ++k) reinterpret_cast<u4v*>(lds_w)[threadIdx.x + k * 256] = pw[k];
    }
  };
  // halo pixel of this lane's column in N-block nb, before the tap offset
  // N-block nb of wave w: lanes 0..15 -> tile row 4 w + nb, lanes 16..31 -> row 4 w + 2 + nb (columns n & 15): the two rows of a
  // 2x2 pooling window are the SAME lane's two N-blocks, its two columns are lanes n and n ^ 1 (one DPP quad permute) -- with rows
  // (2 nb, 2 nb + 1) per block the pooled copy cost two ds_bpermute per value, 5 k cycles per tile (tools/trace_conv16.py)
  const int pbase = (4 * wave + 2 * (n >> 4)) * 18 + (n & 15);
#ifdef S2L_EXP_TRACE
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define PH(k) do { const long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define PH(k) do { } while (0)
#endif
  CONV_TRACE(0);
  fetch(0);
  commit();
  __syncthreads();
  PH(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    // 8-wave form: a wave issues in order, and every 1-KiB memory instruction waits its turn at the CU's one texture-address path
    // (76 KiB per chunk = 1.2 k cycles of it): issued as a block -- before the MFMAs, or mid-way by half of the waves -- the next
    // chunk's ten memory instructions were 2.4 k cycles in which that wave fed nothing to the matrix pipe (tools/trace_conv16.py).
    // They are issued ONE PER TAP behind the tap's twelve MFMAs instead (taps 0..4), which are queued in the pipe while the
    // address path digests the instruction; the last four taps give the loads time to land before the barrier.
    if (cc + 1 < nchunks && !kDma) fetch(cc + 1);
    PH(1);
    const uint16_t* wcur = kDma ? lds_w + (cc & 1) * kChunk16Halves : lds_w;
    if constexpr (SPLIT) {
      // the operands of tap t + 1 are read while the twelve MFMAs of tap t issue (two register sets): the LDS latency, which
      // the compiler's own schedule exposed 22 times per chunk (reads, lgkmcnt(0), a few MFMAs), hides behind matrix work
      u4v A[2][2][2], B[2][2][2];   // [set][part: hi, lo][block]
      auto load_tap = [&](int t, int set) {
        const int dy = t / 3, dx = t % 3;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) A[set][pt][mb] = reinterpret_cast<const u4v*>(wcur)[((t * 2 + pt) * 2 + mb) * 64 + lane];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            B[set][pt][nb] = *reinterpret_cast<const u4v*>(lds_in + (pbase + (nb + dy) * 18 + dx) * kPix16 + 16 * pt + 8 * hh);
        }
      };
      load_tap(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int cur = t & 1;
        if (t + 1 < 9) load_tap(t + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // smallest terms first: the two cross terms, then hi x hi
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma32_f16(A[cur][1][mb], B[cur][0][nb], acc[mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma32_f16(A[cur][0][mb], B[cur][1][nb], acc[mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma32_f16(A[cur][0][mb], B[cur][0][nb], acc[mb][nb]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kDma) {
          if (cc + 1 < nchunks) {
            if (t < kInPer) fetch_one(cc + 1, t);
            if (t < kDmaPer) dma_piece(cc + 1, t);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t % 3;
      {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u4v A[2], B[2];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) A[mb] = reinterpret_cast<const u4v*>(lds_w)[((t * 2 + ks) * 2 + mb) * 64 + lane];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            B[nb] = *reinterpret_cast<const u4v*>(lds_in + (pbase + (nb + dy) * 18 + dx) * kPix16 + 16 * ks + 8 * hh);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma32_bf16(A[mb], B[nb], acc[mb][nb]);
        }
      }
    }
    }
    PH(2);
    __syncthreads();            // everyone is done reading chunk cc
    PH(3);
    if (cc + 1 < nchunks) {
      commit();
      PH(4);
      __syncthreads();
      PH(5);
    }
  }
#ifdef S2L_EXP_TRACE
  if (a.trace && threadIdx.x == 0) {
    long long* tt = a.trace + (blockIdx.x + gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z)) * 24;
    for (int k = 0; k < 6; ++k) tt[2 + k] = tph[k];
    tt[1] = nchunks;
  }
#endif

  // epilogue: D reg r of lane (n, hh) = channel 32 mb + (r & 3) + 8 (r >> 2) + 4 hh of pixel (row 4 wave + 2 (n >> 4) + nb, col n & 15)
  const int gx = x0 + (n & 15);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int gy = y0 + 4 * wave + 2 * (n >> 4) + nb;
    const bool ok = gy < a.H && gx < a.W;
    const int64_t pix = frame * (int64_t)a.H * a.W + (int64_t)gy * a.W + gx;
    if (FUSE_OUT) {
      // outc (conv1x1 64 -> 3) on the 64 channels of this pixel: lane partial over its 32, the other half sits in lane ^ 32
      float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float h = fmaxf(acc[mb][nb][r], 0.f);
          const int c = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
          for (int o = 0; o < 3; ++o) p3[o] = fmaf(a.outw[o * 64 + c], h, p3[o]);
        }
#pragma unroll
      for (int o = 0; o < 3; ++o) p3[o] += __shfl_xor(p3[o], 32);
      if (hh == 0 && ok) {
        float* o3 = a.out3 + pix * 3;
        o3[0] = p3[0] + a.outb[0];
        o3[1] = p3[1] + a.outb[1];
        o3[2] = p3[2] + a.outb[2];
      }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        f4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (a.relu || FUSE_OUT) ? fmaxf(acc[mb][nb][4 * rr + r], 0.f) : acc[mb][nb][4 * rr + r];
        if (ok && a.out) {     // (FUSE_OUT: the last hidden activation, kept only for the backward pass)
          float* dst = a.out + pix * a.cout + ct * 64 + mb * 32 + 8 * rr + 4 * hh;
          if (a.gate) {
            const f4 gt = *reinterpret_cast<const f4*>(a.gate + (dst - a.out));
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = gt[r] > 0.f ? h[r] : 0.f;
          }
          *reinterpret_cast<f4*>(dst) = h;
        }
      }
  }
  if (!FUSE_OUT && a.pool) {   // MaxPool2d(2): rows = the lane's two N-blocks, columns = lanes n and n ^ 1
    const int H2 = a.H / 2, W2 = a.W / 2;
    const int py2 = (y0 + 4 * wave + 2 * (n >> 4)) / 2, px2 = gx / 2;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        f4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fmaxf(fmaxf(acc[mb][0][4 * rr + r], acc[mb][1][4 * rr + r]), 0.f);
          const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
          m[r] = fmaxf(v, o);
        }
        if (!(n & 1) && py2 < H2 && px2 < W2)
          *reinterpret_cast<f4*>(a.pool + ((frame * H2 + py2) * (int64_t)W2 + px2) * a.cout + ct * 64 + mb * 32 + 8 * rr + 4 * hh) = m;
      }
  }
  CONV_TRACE(20);
}

// ---- the split-bf16 convolution as a PERSISTENT kernel with ONE barrier per chunk (round 3) ------------------------------------
// Same arithmetic, tile (32 x 16 pixels, 64 output channels, eight waves) and accumulation order as
// conv3x3_bf16_kernel<.., SPLIT = true, 8>: the outputs are the same bits (tools/cmp_split_kernels.py, test_unet_split_*).  What
// changed is when memory is touched.  tools/trace_conv16.py on the one-tile-per-workgroup form: of 55-98 k cycles per tile 10 k were
// the exposed loads of the first chunk; every chunk paid barrier -> commit (hi / lo conversion + LDS writes) -> barrier with the
// matrix pipe idle, and ~2.4 k cycles at the first barrier for the NEXT chunk's loads, because __syncthreads drains vmcnt; and with
// all eight waves in step, operand reads issued as a block of eight kept the LDS busy for ~500 cycles per tap in which nobody had
// operands.  Here:
//  * a workgroup walks a contiguous range of (frame, channel tile, y, x) tiles; the chunks of all its tiles form ONE stream;
//  * TWO halo buffers (64 bytes per pixel, no padding: the 16-byte segment index is XORed with (column >> 2) & 3, rows are 20 pixels
//    = 5 x 256 bytes apart, so the sixteen pixels of an operand read cover all 64 banks) and two weight buffers: during chunk g the
//    weights of g + 1 arrive by LDS-DMA, the halo values of g + 1 (requested during g - 1, held in registers) are converted and
//    written behind the MFMAs of taps 5..8, and the values of g + 2 are requested -- one barrier per chunk, no commit phase;
//  * the weight requests are assembly text (the compiler would drain vmcnt before every LDS access that may alias an LDS-DMA it
//    knows about) waited for with a counted vmcnt: loads return in order, so "at most n outstanding" retires everything older
//    than the n newest (a tile's epilogue stores in between only make a wait stricter); so are the requests of the halo values
//    inside the chunk loop (see fetch_one for the three rules that keep their registers from being copied in flight);
//  * the eight operand reads of tap t + 1 are issued one behind each of the first eight MFMAs of tap t.
constexpr int kSwzRow = 20 * 64;                                // bytes between halo rows
constexpr int kSwzIn = 34 * kSwzRow;                            // 43 520 bytes per halo buffer
constexpr int kSplitBias = 2 * kSwzIn + 2 * kChunk16Halves * 2;  // 87 040 + 73 728 = 160 768 bytes, then the layer's biases (<= 256)
constexpr int kSplitLds = kSplitBias + 1024;
template <int N> struct IntC { static constexpr int value = N; };

// SPLIT = false: the same kernel for PLAIN bf16 operands (the training chain: forward with saved activations, the input-gradient
// twins with their ReLU gate, the raw convolutions of train-mode BatchNorm).  A chunk is then 32 input channels = the same 64 bytes
// per halo pixel and the same 36 KiB of weights ([tap][k-step][M-block] where the split form has [tap][part][M-block]): identical
// LDS shapes and operand reads, eight MFMAs per tap instead of twelve.  The halo values travel in HALVES of 16 channels, one per
// staging set (so the register budget is the split form's): during chunk g set 0 (channels 0..15 of g + 1, requested during
// g - 1) is committed behind taps 0..1 and re-requested for g + 2 behind taps 2..4, set 1 behind taps 5..6 and 7..8.
template <bool FUSE_OUT, bool SPLIT = true>
__global__ __launch_bounds__(512, 1) void conv3x3_split_kernel(ConvArgs a) {
  constexpr int kThreads = 512, kQ = 4, kInQuads = 34 * 18 * kQ, kInPer = 5;
  constexpr int kDmaPieces = kChunk16Halves * 2 / 1024, kDmaPer = 5;        // 36 pieces of 1 KiB: waves 0..3 move five, 4..7 four
  static_assert(kInPer * kThreads >= kInQuads && kDmaPer * 8 >= kDmaPieces, "coverage");
  extern __shared__ __attribute__((aligned(16))) char split_smem[];
  char* const lds_in = split_smem;                              // two halo buffers
  const uint16_t* const lds_w = reinterpret_cast<const uint16_t*>(split_smem + 2 * kSwzIn);      // two weight buffers
  const uint32_t lds_w_addr = (uint32_t)(uintptr_t)(split_smem + 2 * kSwzIn);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, hh = lane >> 5;
  constexpr int kCC = SPLIT ? 16 : 32;                          // input channels per chunk
  const int nchunks = (a.CA + a.CB) / kCC;                      // even for every layer of the net (checked by the launcher)
  const int nunits = SPLIT ? nchunks : 2 * nchunks;             // 16-channel staging units per tile
  const int tiles_y = (a.H + 31) / 32;
  const int64_t total = (int64_t)a.tiles_x * tiles_y * a.n_ct * a.n_frames_asm;
  const int tile0 = (int)(total * blockIdx.x / gridDim.x), tile_end = (int)(total * (blockIdx.x + 1) / gridDim.x);
  if (tile0 >= tile_end) return;
#ifdef S2L_EXP_TRACE
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 24] = tlast;
#define PHS(k) do { const long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define PHS(k) do { } while (0)
#endif
  const int n_dma = wave < 4 ? 5 : 4;                           // this wave's weight pieces per chunk

  struct TilePos { int x0, y0, ct; int64_t frame; };
  auto decode = [&](int t) {
    TilePos p;
    p.x0 = (t % a.tiles_x) * 16;
    t /= a.tiles_x;
    p.y0 = (t % tiles_y) * 32;
    t /= tiles_y;
    p.ct = t % a.n_ct;
    p.frame = t / a.n_ct;
    return p;
  };

  // ---- per-thread LDS offsets (independent of the tile) ----
  // commit: quad i = tid + 512 k of the halo tile = pixel i / 4 (row-major 34 x 18), channels 4 (i % 4) .. + 3: 8 bytes of hi at
  // segment (c4 >> 1), 8 bytes of lo at segment 2 + (c4 >> 1) = the hi address ^ 32
  int coff[kInPer];
#pragma unroll
  for (int k = 0; k < kInPer; ++k) {
    const int i = threadIdx.x + k * kThreads, pi = i / kQ, c4 = i % kQ;
    const int row = pi / 18, col = pi % 18;
    coff[k] = row * kSwzRow + col * 64 + ((((c4 >> 1) ^ ((col >> 2) & 3))) << 4) + (c4 & 1) * 8;
  }
  // operand reads: lane (n, hh) of N-block blk, tap (dy, dx), part pt: pixel (row 4 wave + 2 (n >> 4) + blk + dy, col (n & 15) + dx),
  // segment 2 pt + hh; (blk + dy) * kSwzRow is an immediate offset
  int zb[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int col = (n & 15) + dx;
      zb[dx][pt] = (4 * wave + 2 * (n >> 4)) * kSwzRow + col * 64 + (((2 * pt + hh) ^ ((col >> 2) & 3)) << 4);
    }

  // ---- input stream: the next chunk whose halo values are requested ----
  int f_tile = tile0, f_cc = 0;
  int pixoff[kInPer], f_vbits = 0;
  const float *f_inA = nullptr, *f_inB = nullptr;
  const int c4x4 = 4 * (threadIdx.x % kQ);
  auto f_setup = [&]() {
    const TilePos p = decode(f_tile);
    f_vbits = 0;
#pragma unroll
    for (int k = 0; k < kInPer; ++k) {
      const int i = threadIdx.x + k * kThreads;
      const int pi = i / kQ;
      const int gy = p.y0 - 1 + pi / 18, gx = p.x0 - 1 + pi % 18;
      const bool ok = i < kInQuads && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      pixoff[k] = ok ? gy * a.W + gx : 0;                       // (clamped: the load is always issued, the value dropped at commit)
      f_vbits |= (ok ? 1 : 0) << k;
    }
    f_inA = a.inA + p.frame * (int64_t)a.H * a.W * a.CA;
    f_inB = a.inB ? a.inB + p.frame * (int64_t)a.H * a.W * a.CB : nullptr;
  };
  f4 pin[2][kInPer];
#pragma unroll
  for (int k = 0; k < kInPer; ++k) pin[0][k] = pin[1][k] = (f4){0.f, 0.f, 0.f, 0.f};
  int pvalid[2] = {0, 0};
  auto fetch_one = [&](int k, auto SET, auto PLAIN) {
    const bool fromA = f_cc * 16 < a.CA;
    const float* src = (fromA ? f_inA : f_inB) + (fromA ? f_cc * 16 : f_cc * 16 - a.CA) + c4x4;
    const int C = fromA ? a.CA : a.CB;
    // Inside the chunk loop the request is assembly text with a TIED operand: the staging variable is defined in place, so it
    // never leaves its physical register, and nothing reads it until `values_landed` below.  Three rules keep the register
    // allocator from copying a register whose load is still in flight (a copy taken before the data lands and written back later
    // -- ~1 % of random shapes showed wrong pixels in tools/soak_conv_kernels.py while any of them was broken):
    //  (1) every in-loop definition of a staging register is a tied asm (no second register can hold the value);
    //  (2) the prologue's requests are ORDINARY loads (`plain`), so the copies the allocator makes at the loop entry, where
    //      prologue and loop disagree about registers, are behind the compiler's own wait;
    //  (3) the counted wait is an asm WITHOUT register operands, followed by one tied no-op per staging set (a switch over tied
    //      asm statements made the allocator copy the set into temporaries before the wait and back after it).
    // tests/test_abi_and_host.py checks the built code object: no v_mov / v_accvgpr reads a register that an asm load writes.
    const float* addr = src + (int64_t)pixoff[k] * C;
    if constexpr (decltype(PLAIN)::value) {
      pin[decltype(SET)::value][k] = *reinterpret_cast<const f4*>(addr);
    } else {
      f4& dstreg = pin[decltype(SET)::value][k];
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dstreg) : "v"(addr) : "memory");
    }
  };
  auto f_advance = [&](auto SET) {
    pvalid[decltype(SET)::value] = f_vbits;
    if (++f_cc == nunits) {       // (f_cc counts 16-channel units)
      f_cc = 0;
      if (++f_tile < tile_end) f_setup();
    }
  };
  // ---- weight stream: the next chunk whose weights are moved ----
  int w_tile = tile0, w_cc = 0;
  const uint16_t* w_ptr = a.w16 + (int64_t)decode(tile0).ct * nchunks * kChunk16Halves;      // chunk (w_tile, w_cc)
  auto dma_piece = [&](int j, int buf) {
    const int p = wave + 8 * j;
    if (p < kDmaPieces) {
      const char* wsrc = reinterpret_cast<const char*>(w_ptr) + p * 1024 + lane * 16;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_w_addr + (uint32_t)(buf * kChunk16Halves * 2 + p * 1024));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(wsrc) : "memory");
    }
  };
  auto w_advance = [&]() {
    w_ptr += kChunk16Halves;
    if (++w_cc == nchunks) {
      w_cc = 0;
      if (++w_tile < tile_end) w_ptr = a.w16 + (int64_t)decode(w_tile).ct * nchunks * kChunk16Halves;
    }
  };
  auto commit_one = [&](int k, auto SET, int buf) {       // (plain form: set = the chunk's half, channels 16 set .. + 15 = segments 2 set, 2 set + 1)
    constexpr int set = decltype(SET)::value;
    const int i = threadIdx.x + k * kThreads;
    if (i < kInQuads) {
      f4 v = pin[set][k];
      if (!((pvalid[set] >> k) & 1)) v = (f4){0.f, 0.f, 0.f, 0.f};
      uint2 h;
      char* dst = lds_in + buf * kSwzIn;
      if constexpr (SPLIT) {
        h.x = pack_f16x2(v[0], v[1]);
        h.y = pack_f16x2(v[2], v[3]);
        uint2 l;
        l.x = pack_f16x2_rne(v[0] - f16_lo(h.x), v[1] - f16_hi(h.x));
        l.y = pack_f16x2_rne(v[2] - f16_lo(h.y), v[3] - f16_hi(h.y));
        *reinterpret_cast<uint2*>(dst + coff[k]) = h;
        *reinterpret_cast<uint2*>(dst + (coff[k] ^ 32)) = l;
      } else {
        h.x = pack_bf16x2(v[0], v[1]);
        h.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(dst + (coff[k] ^ (set * 32))) = h;
      }
    }
  };
  auto wait_loads = [&](int newer) {      // every vector-memory LOAD except the `newer` most recent ones has landed
    switch (newer) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
  };
  // the counted wait for a staging set + the tied no-op that makes the set's registers depend on it (rule 3)
  auto values_landed = [&](int newer, auto SET) {
    constexpr int set = decltype(SET)::value;
    wait_loads(newer);
    f4 &p0 = pin[set][0], &p1 = pin[set][1], &p2 = pin[set][2], &p3 = pin[set][3], &p4 = pin[set][4];
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4) : : "memory");
  };
  auto barrier_lgkm = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // the layer's biases wait in LDS: read from global memory at every tile start, their ~2 k cycles of latency were exposed
  float* const lds_bias = reinterpret_cast<float*>(split_smem + kSplitBias);
  if (threadIdx.x < 256) lds_bias[threadIdx.x] = (a.bias && (int)threadIdx.x < a.n_ct * 64) ? a.bias[threadIdx.x] : 0.f;

  // ---- prologue: chunk 0 (values + weights) and the values of chunk 1 ----
  const int64_t chunks_total = (int64_t)(tile_end - tile0) * nchunks;      // >= 2
  f_setup();
  if constexpr (SPLIT) {
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(k, IntC<0>{}, IntC<1>{});
    f_advance(IntC<0>{});
#pragma unroll
    for (int j = 0; j < kDmaPer; ++j) dma_piece(j, 0);
    w_advance();
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(k, IntC<1>{}, IntC<1>{});
    f_advance(IntC<1>{});
#pragma unroll
    for (int k = 0; k < kInPer; ++k) commit_one(k, IntC<0>{}, 0);
    wait_loads(kInPer);                                         // chunk 0's weight pieces (older than chunk 1's five value loads)
  } else {
    // both halves of chunk 0, committed at once (the one exposed latency of the workgroup); then both halves of chunk 1
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(k, IntC<0>{}, IntC<1>{});
    f_advance(IntC<0>{});
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(k, IntC<1>{}, IntC<1>{});
    f_advance(IntC<1>{});
#pragma unroll
    for (int j = 0; j < kDmaPer; ++j) dma_piece(j, 0);
    w_advance();
#pragma unroll
    for (int k = 0; k < kInPer; ++k) commit_one(k, IntC<0>{}, 0);
#pragma unroll
    for (int k = 0; k < kInPer; ++k) commit_one(k, IntC<1>{}, 0);
    wait_loads(0);                                              // chunk 0's weight pieces
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(k, IntC<0>{}, IntC<1>{});
    f_advance(IntC<0>{});
#pragma unroll
    for (int k = 0; k < kInPer; ++k) fetch_one(k, IntC<1>{}, IntC<1>{});
    f_advance(IntC<1>{});
  }
  // the compiler waits for the prologue's (ordinary) loads HERE: a load it still counts as pending at the loop entry would make it
  // drain vmcnt at the first use inside the loop, on every pass
  asm volatile("" : "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]), "+v"(pin[0][3]), "+v"(pin[0][4]));
  asm volatile("" : "+v"(pin[1][0]), "+v"(pin[1][1]), "+v"(pin[1][2]), "+v"(pin[1][3]), "+v"(pin[1][4]));
  barrier_lgkm();
  PHS(0);

  int64_t seq = 0;
  f16v acc[2][2];

  auto step = [&](auto SET) {
    constexpr int set = decltype(SET)::value;                   // chunk parity = its halo / weight buffer = the staging set of chunk g + 2
    const bool next1 = seq + 1 < chunks_total;                  // chunk g + 1 exists: its weights move, its values are committed
    const bool next2 = seq + 2 < chunks_total;                  // chunk g + 2 exists: its values are requested into pin[set]
    const uint16_t* wcur = lds_w + set * kChunk16Halves;
    const char* icur = lds_in + set * kSwzIn;
    u4v A[2][2][2], B[2][2][2];   // [register set][part: hi, lo][block]
    // operand read i (0..7) of tap t into register set os, in the order the tap's MFMAs need them: lo weights, hi pixels (first
    // cross term), lo pixels (second), hi weights (third)
    auto read_one = [&](int t, int os, int i) {
      const int dy = t / 3, dx = t % 3;
      const int kind = i >> 1, blk = i & 1;      // 0: A lo, 1: B hi, 2: B lo, 3: A hi
      if (kind == 0 || kind == 3) {
        const int pt = kind == 0 ? 1 : 0;
        A[os][pt][blk] = reinterpret_cast<const u4v*>(wcur)[((t * 2 + pt) * 2 + blk) * 64 + lane];
      } else {
        const int pt = kind == 1 ? 0 : 1;
        B[os][pt][blk] = *reinterpret_cast<const u4v*>(icur + zb[dx][pt] + (blk + dy) * kSwzRow);
      }
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) read_one(0, 0, i);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1;
      if (t == 5 && next1) values_landed((next1 ? n_dma : 0) + (next2 ? kInPer : 0), IntC<set ^ 1>{});      // chunk g + 1's values (requested during g - 1)
#pragma unroll
      for (int m = 0; m < 12; ++m) {      // smallest terms first: lo x hi, hi x lo, hi x hi
        const int g = m >> 2, mb = (m >> 1) & 1, nb = m & 1;
        const int pa = g == 0 ? 1 : 0, pb = g == 1 ? 1 : 0;
        acc[mb][nb] = mfma32_f16(A[cur][pa][mb], B[cur][pb][nb], acc[mb][nb]);
        if (m < 8 && t + 1 < 9) read_one(t + 1, cur ^ 1, m);
        __builtin_amdgcn_sched_barrier(0);
      }
      // behind tap 0..4: two requests each, weights first (they are needed at the end of THIS chunk): D0 D1 | D2 D3 | D4 L0 | L1 L2 | L3 L4
      if (t < 5 && !(S2L_UEXP & 32)) {
        const int r0 = 2 * t, r1 = 2 * t + 1;
        if (r0 < 5) { if (next1) dma_piece(r0, set ^ 1); } else if (next2) fetch_one(r0 - 5, SET, IntC<0>{});
        if (r1 < 5) { if (next1) dma_piece(r1, set ^ 1); } else if (next2) fetch_one(r1 - 5, SET, IntC<0>{});
      }
      // behind tap 5..8: the values of chunk g + 1 -> the other halo buffer (2 + 1 + 1 + 1 quads)
      if (t >= 5 && next1) {
        if (t == 5) commit_one(0, IntC<set ^ 1>{}, set ^ 1);
        commit_one(t - 4, IntC<set ^ 1>{}, set ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PHS(1);
    if (next1) w_advance();
    if (next2) f_advance(SET);
    ++seq;
    PHS(5);
    wait_loads(next2 ? kInPer : 0);                             // the weights of chunk g + 1 have landed (the values of g + 2 may fly)
    PHS(3);
    barrier_lgkm();
    PHS(2);
  };

  // plain bf16: one step per 32-channel chunk; both staging sets turn over inside it
  auto step_plain = [&]() {
    const int buf = (int)(seq & 1);
    const bool next1 = seq + 1 < chunks_total;                  // chunk g + 1 exists: weights move, both halves are committed
    const bool next2 = seq + 2 < chunks_total;                  // chunk g + 2 exists: both halves are requested
    const uint16_t* wcur = lds_w + buf * kChunk16Halves;
    const char* icur = lds_in + buf * kSwzIn;
    u4v A[2][2][2], B[2][2][2];   // [register set][k-step][block]
    auto read_one = [&](int t, int os, int i) {      // read i (0..7): A k-step 0 (2 blocks), B k-step 0, A k-step 1, B k-step 1
      const int dy = t / 3, dx = t % 3;
      const int ks = i >> 2, isB = (i >> 1) & 1, blk = i & 1;
      if (!isB) A[os][ks][blk] = reinterpret_cast<const u4v*>(wcur)[((t * 2 + ks) * 2 + blk) * 64 + lane];
      else B[os][ks][blk] = *reinterpret_cast<const u4v*>(icur + zb[dx][ks] + (blk + dy) * kSwzRow);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) read_one(0, 0, i);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1;
      // set 0 = channels 0..15 of chunk g + 1 (requested during g - 1, before that chunk's set-1 requests: 5 newer loads);
      // set 1 = channels 16..31 (newer: this chunk's weight pieces and set-0 requests)
      if (t == 0 && next1) values_landed(kInPer, IntC<0>{});
      if (t == 5 && next1) values_landed(n_dma + (next2 ? kInPer : 0), IntC<1>{});
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int ks = m >> 2, mb = (m >> 1) & 1, nb = m & 1;
        acc[mb][nb] = mfma32_bf16(A[cur][ks][mb], B[cur][ks][nb], acc[mb][nb]);
        if (t + 1 < 9) read_one(t + 1, cur ^ 1, m);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (next1) {      // commits: set 0 behind taps 0, 1 (3 + 2 quads), set 1 behind taps 5, 6
        if (t == 0) { commit_one(0, IntC<0>{}, buf ^ 1); commit_one(1, IntC<0>{}, buf ^ 1); commit_one(2, IntC<0>{}, buf ^ 1); }
        if (t == 1) { commit_one(3, IntC<0>{}, buf ^ 1); commit_one(4, IntC<0>{}, buf ^ 1); }
        if (t == 5) { commit_one(0, IntC<1>{}, buf ^ 1); commit_one(1, IntC<1>{}, buf ^ 1); commit_one(2, IntC<1>{}, buf ^ 1); }
        if (t == 6) { commit_one(3, IntC<1>{}, buf ^ 1); commit_one(4, IntC<1>{}, buf ^ 1); }
      }
      // requests: D0 D1 | D2 D3 | D4 L0 | L1 L2 | L3 L4 behind taps 0..4 (set 0 <- g + 2), L0' L1' L2' | L3' L4' behind taps 7, 8 (set 1)
      if (t < 5) {
        const int r0 = 2 * t, r1 = 2 * t + 1;
        if (r0 < 5) { if (next1) dma_piece(r0, buf ^ 1); } else if (next2) fetch_one(r0 - 5, IntC<0>{}, IntC<0>{});
        if (r1 < 5) { if (next1) dma_piece(r1, buf ^ 1); } else if (next2) fetch_one(r1 - 5, IntC<0>{}, IntC<0>{});
      }
      if (t == 4 && next2) f_advance(IntC<0>{});
      if (t == 7 && next2) { fetch_one(0, IntC<1>{}, IntC<0>{}); fetch_one(1, IntC<1>{}, IntC<0>{}); fetch_one(2, IntC<1>{}, IntC<0>{}); }
      if (t == 8 && next2) { fetch_one(3, IntC<1>{}, IntC<0>{}); fetch_one(4, IntC<1>{}, IntC<0>{}); }
      __builtin_amdgcn_sched_barrier(0);
    }
    PHS(1);
    if (next1) w_advance();
    if (next2) f_advance(IntC<1>{});
    ++seq;
    PHS(5);
    wait_loads(next2 ? 2 * kInPer : 0);                         // the weights of chunk g + 1 have landed (the values of g + 2 may fly)
    PHS(3);
    barrier_lgkm();
    PHS(2);
  };

  for (int tile = tile0; tile < tile_end; ++tile) {
    const TilePos tp = decode(tile);
    const int x0 = tp.x0, y0 = tp.y0, ct = tp.ct;
    const int64_t frame = tp.frame;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const f4 b = *reinterpret_cast<const f4*>(lds_bias + ct * 64 + mb * 32 + 8 * rr + 4 * hh);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mb][nb][4 * rr + r] = b[r];
        }
    for (int cc = 0; cc < nchunks; cc += 2) {
      if constexpr (SPLIT) {
        step(IntC<0>{});
        step(IntC<1>{});
      } else {
        step_plain();
        step_plain();
      }
    }

    // epilogue (as conv3x3_bf16_kernel): D reg r of lane (n, hh) = channel 32 mb + (r & 3) + 8 (r >> 2) + 4 hh of pixel
    // (row 4 wave + 2 (n >> 4) + nb, col n & 15)
    const int gx = x0 + (n & 15);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int gy = y0 + 4 * wave + 2 * (n >> 4) + nb;
      const bool ok = gy < a.H && gx < a.W;
      const int64_t pix = frame * (int64_t)a.H * a.W + (int64_t)gy * a.W + gx;
      if (FUSE_OUT) {
        float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float h = fmaxf(acc[mb][nb][r], 0.f);
            const int c = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
            for (int o = 0; o < 3; ++o) p3[o] = fmaf(a.outw[o * 64 + c], h, p3[o]);
          }
#pragma unroll
        for (int o = 0; o < 3; ++o) p3[o] += __shfl_xor(p3[o], 32);
        if (hh == 0 && ok) {
          float* o3 = a.out3 + pix * 3;
          o3[0] = p3[0] + a.outb[0];
          o3[1] = p3[1] + a.outb[1];
          o3[2] = p3[2] + a.outb[2];
        }
      }
      if (!a.out) continue;       // (FUSE_OUT: the last hidden activation is kept only for a backward pass)
      // The activation leaves through LDS: as the MFMA leaves it, a store instruction would write 16 bytes to each of 64 different
      // 128-byte lines (lane = pixel); transposed, lane l writes channel quad l & 7 of pixel l >> 3: eight whole lines per
      // instruction.  Staging = the four 1-KiB pieces of weight buffer 1 that THIS wave's LDS-DMA writes (wave + 8 j): the buffer
      // was last read in the chunk that just ended, and the only later writer of these pieces is this wave, after its epilogue.
      // [32 pixels][32 channels] fp32, 128 bytes per pixel, the 16-byte quad index XORed with (pixel >> 1) & 7 (conflict-free both ways).
      char* const stage = split_smem + 2 * kSwzIn + kChunk16Halves * 2 + wave * 1024;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          f4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (a.relu || FUSE_OUT) ? fmaxf(acc[mb][nb][4 * rr + r], 0.f) : acc[mb][nb][4 * rr + r];
          *reinterpret_cast<f4*>(stage + (n >> 3) * 8192 + (n & 7) * 128 + (((2 * rr + hh) ^ ((n >> 1) & 7)) << 4)) = h;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int px = 8 * q + (lane >> 3), chq = lane & 7;
          f4 h = *reinterpret_cast<const f4*>(stage + q * 8192 + (px & 7) * 128 + ((chq ^ ((px >> 1) & 7)) << 4));
          const int sy = y0 + 4 * wave + 2 * (px >> 4) + nb, sx = x0 + (px & 15);
          if (sy < a.H && sx < a.W) {
            const int64_t o = (frame * (int64_t)a.H * a.W + (int64_t)sy * a.W + sx) * a.cout + ct * 64 + mb * 32 + 4 * chq;
            if (a.gate) {      // input-gradient twins: the ReLU mask of the activation that fed the layer
              const f4 gt = *reinterpret_cast<const f4*>(a.gate + o);
#pragma unroll
              for (int r = 0; r < 4; ++r) h[r] = gt[r] > 0.f ? h[r] : 0.f;
            }
            *reinterpret_cast<f4*>(a.out + o) = h;
          }
        }
      }
    }
    PHS(4);
    if (!FUSE_OUT && a.pool) {   // MaxPool2d(2): rows = the lane's two N-blocks, columns = lanes n and n ^ 1
      const int H2 = a.H / 2, W2 = a.W / 2;
      const int py2 = (y0 + 4 * wave + 2 * (n >> 4)) / 2, px2 = gx / 2;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          f4 m;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaxf(fmaxf(acc[mb][0][4 * rr + r], acc[mb][1][4 * rr + r]), 0.f);
            const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
            m[r] = fmaxf(v, o);
          }
          if (!(n & 1) && py2 < H2 && px2 < W2)
            *reinterpret_cast<f4*>(a.pool + ((frame * H2 + py2) * (int64_t)W2 + px2) * a.cout + ct * 64 + mb * 32 + 8 * rr + 4 * hh) = m;
        }
    }
    PHS(0);       // (epilogue + next tile's bias: counted with the prologue)
  }
#ifdef S2L_EXP_TRACE
  if (a.trace && threadIdx.x == 0) {
    long long* tt = a.trace + blockIdx.x * 24;
    for (int k = 0; k < 6; ++k) tt[2 + k] = tph[k];
    tt[1] = chunks_total;
    tt[20] = __builtin_readcyclecounter();
  }
#endif
}

// ---- the forward convolution with its body as one fixed-register assembly text (csrc/gen_conv_body.py: persistent, one wave
// per SIMD, both operands by LDS-DMA into two buffers, the chunk loop nothing but MFMAs, LDS reads and scalar code).  Same tile,
// packed weights, LDS layouts and accumulation order as conv3x3_kernel: bit-identical outputs.  VARIANT 1: also the 2x2-pooled copy; 2: also the network's 1x1 output
// convolution (conv3x3_kernel<true>'s epilogue), the activation itself stored only if a.out is set; 3: no bias, no ReLU, optional
// gate (the input-gradient convolutions and the raw convolutions of the train-mode forward).
constexpr int kConvAsmLds = 2 * (18 * 18 * 16 * 4 + kChunkFloats * 4) + 768;      // two buffers + variant 2's output weights
template <int VARIANT>
__global__ __launch_bounds__(256) void conv3x3_asm_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char conv_smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ldsbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)conv_smem);
#if defined(__HIP_DEVICE_COMPILE__)
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();   // the body loads ConvArgs fields itself (s_load)
#else
  const void* karg = nullptr;   // (host pass of the compiler: never executed)
#endif
  // this workgroup's tiles: a contiguous range of (frame, channel tile, y, x), x fastest
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * a.n_ct * a.n_frames_asm;
  const int tile0 = __builtin_amdgcn_readfirstlane((int)(total * blockIdx.x / gridDim.x));
  const int tile_end = __builtin_amdgcn_readfirstlane((int)(total * (blockIdx.x + 1) / gridDim.x));
  int t = tile0;
  const int tx0 = __builtin_amdgcn_readfirstlane(t % a.tiles_x);
  t /= a.tiles_x;
  const int ty0 = __builtin_amdgcn_readfirstlane(t % a.tiles_y);
  t /= a.tiles_y;
  const int ct0 = __builtin_amdgcn_readfirstlane(t % a.n_ct), frame0 = __builtin_amdgcn_readfirstlane(t / a.n_ct);
  if (VARIANT == 1) {
#include "conv_body_fwd_pool.inc"
  } else if (VARIANT == 2) {
#include "conv_body_fwd_out.inc"
  } else if (VARIANT == 3) {
#include "conv_body_lin.inc"
  } else {
#include "conv_body_fwd.inc"
  }
}

// The assembly kernel takes the launch if it has the epilogue: A | B of equal width, an even number of 16-channel chunks,
// per-frame byte offsets that fit 31 bits, and (bias + ReLU [+ pooling | + fused output]) or (no bias, no ReLU [+ gate]).
static std::atomic<int> g_conv_kernel_kind{0};
static std::atomic<int> g_split_kernel_kind{0};      // 0: conv3x3_split_kernel (persistent), 1: conv3x3_bf16_kernel<.., true, 8> (one tile per workgroup)
extern "C" int s2l_set_unet_split_kernel(int kind) {      // 2: conv16_asm_kernel (csrc/conv16.hip) where it applies, else kind 0
  if (kind != 0 && kind != 1 && kind != 2) return S2L_E_SIZE;
#ifndef S2L_WITH_REFERENCE_KERNELS
  if (kind == 2) return S2L_E_UNSUPPORTED;      // (that form lives in libs2l_hip_ref.so)
#endif
  g_split_kernel_kind.store(kind, std::memory_order_relaxed);
  return S2L_OK;
}
extern "C" int s2l_set_unet_conv_kernel(int kind) {
  if (kind != 0 && kind != 1) return S2L_E_SIZE;
  g_conv_kernel_kind.store(kind, std::memory_order_relaxed);
  return S2L_OK;
}

static int launch_conv_asm(ConvArgs& a, int64_t F, hipStream_t st, bool* launched) {
  *launched = false;
  if (g_conv_kernel_kind.load(std::memory_order_relaxed) == 1) return S2L_OK;
  const bool fwd = a.bias && a.relu && !a.gate && !(a.out3 && a.pool) && (!a.out3 || a.cout == 64);
  const bool lin = !a.bias && !a.relu && !a.out3 && !a.pool && a.out;
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * a.n_ct * F;
  if (!S2L_CONV_ASM || a.w16 || !(fwd || lin) || !(a.CB == 0 || a.CB == a.CA) || (a.CA + a.CB) % 32 != 0 || a.cout % 64 != 0 ||
      (int64_t)(a.H + 2) * (a.W + 2) * std::max(a.CA, a.cout) * 4 >= 0x7fffffff || total >= 0x7fffffff)
    return S2L_OK;
  a.n_frames_asm = (int)F;
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  static LdsOptIn flags[4];
  const int variant = lin ? 3 : a.out3 ? 2 : a.pool ? 1 : 0;
  void (*const kern[4])(ConvArgs) = {conv3x3_asm_kernel<0>, conv3x3_asm_kernel<1>, conv3x3_asm_kernel<2>, conv3x3_asm_kernel<3>};
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern[variant]), kConvAsmLds, flags[variant], dev))) return rc;
  hipLaunchKernelGGL(kern[variant], dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(256), kConvAsmLds, st, a);
  *launched = true;
  return (int)hipGetLastError();
}

// The persistent kernel takes a bf16-operand launch (split or plain) when its chunk count is even and selector 0 is set:
// 32 x 16-pixel tiles, a contiguous tile range per workgroup, one workgroup per CU.
static int launch_conv_persistent(ConvArgs& a, int64_t F, bool fuse_out, bool split, hipStream_t st, bool* launched) {
  *launched = false;
  const int64_t total = (int64_t)a.tiles_x * ((a.H + 31) / 32) * a.n_ct * F;
  const int nchunks = (a.CA + a.CB) / (split ? 16 : 32);
  if (g_split_kernel_kind.load(std::memory_order_relaxed) == 1 || (a.CA + a.CB) % (split ? 16 : 32) != 0 || nchunks % 2 != 0 ||
      (a.CB != 0 && a.CA % 16 != 0) || a.n_ct > 4 || (a.pool && fuse_out) || total >= 0x7fffffff || total == 0)
    return S2L_OK;
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  // plain bf16: a chunk is a third of the split form's matrix work, so the one exposed first-chunk latency of a workgroup only
  // pays off over a few tiles (one 500x500 frame per call = 2 tiles per workgroup measured 1 % slower than one tile per workgroup)
  if (!split && total < 4 * (int64_t)n_cu) return S2L_OK;
  static LdsOptIn pflags[4];
  void (*const kern[4])(ConvArgs) = {conv3x3_split_kernel<false, false>, conv3x3_split_kernel<true, false>,
                                     conv3x3_split_kernel<false, true>, conv3x3_split_kernel<true, true>};
  const int v = (split ? 2 : 0) + (fuse_out ? 1 : 0);
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern[v]), kSplitLds, pflags[v], dev))) return rc;
  a.n_frames_asm = (int)F;
  hipLaunchKernelGGL(kern[v], dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), kSplitLds, st, a);
  *launched = true;
  return (int)hipGetLastError();
}

static int launch_conv(const float* inA, int CA, const float* inB, int CB, const float* packed, int layer, float* out,
                       float* out3, int H, int W, int64_t F, hipStream_t st, float* pool = nullptr, float* keep = nullptr,
                       const uint16_t* packed16 = nullptr, int split = 0) {
  ConvArgs a;
  a.w16 = packed16 ? packed16 + (split ? unet_w16x3_off(layer) : unet_w16_off(layer)) : nullptr;
  a.split = split;
  a.inA = inA; a.inB = inB; a.CA = CA; a.CB = CB;
  a.cout = kUnetConvs[layer].cout;
  a.w = packed + unet_w_off(layer);
  a.bias = packed + unet_b_off(layer);
  a.out = out3 ? keep : out; a.out3 = out3; a.pool = pool; a.gate = nullptr; a.relu = 1;
  a.outw = packed + kUnetOutW; a.outb = packed + kUnetOutB;
  a.H = H; a.W = W;
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
  a.n_ct = a.cout / 64;
  const int64_t gz = F * a.n_ct;
  if (gz > 65535) return S2L_E_SIZE;
  dim3 grid(a.tiles_x, a.tiles_y, (unsigned)gz);
  CONV_TRACE_ARGS(a, grid);
  bool done = false;
  const int rc_asm = launch_conv_asm(a, F, st, &done);
  if (done) return rc_asm;
#ifdef S2L_WITH_REFERENCE_KERNELS
  if (a.w16 && split && !out3 && g_split_kernel_kind.load(std::memory_order_relaxed) == 2) {
    // the generated-assembly split convolution (bias + ReLU; a pooled copy comes from maxpool2_kernel: max of the same fp32 values)
    Conv16Args c;
    c.inA = a.inA; c.inB = a.inB; c.w16 = a.w16; c.bias = a.bias; c.out = a.out; c.CA = a.CA; c.CB = a.CB; c.cout = a.cout;
    c.H = H; c.W = W; c.tiles_x = a.tiles_x; c.tiles_y = a.tiles_y; c.n_ct = a.n_ct; c.relu = 1; c.n_frames = (int)F; c.pool = pool;
    bool launched = false;
    const int rc = launch_conv16_asm(c, st, &launched);
    if (rc || launched) return rc;
  }
#endif
  if (a.w16 && (split || g_split_kernel_kind.load(std::memory_order_relaxed) == 0)) {
    bool launched = false;
    const int rc = launch_conv_persistent(a, F, out3 != nullptr, split != 0, st, &launched);
    if (rc || launched) return rc;
    // the split form without the persistent kernel: one 32 x 16-pixel tile per 512-thread workgroup
    const dim3 wgrid(a.tiles_x, (H + 31) / 32, (unsigned)gz);
    static LdsOptIn wflags[2];
    int dev = 0, n_cu = 0;
    int rc2 = current_device_cus(&dev, &n_cu);
    if (rc2) return rc2;
    void (*const kern)(ConvArgs) = out3 ? conv3x3_bf16_kernel<true, true, 8> : conv3x3_bf16_kernel<false, true, 8>;
    if (split) {
      if ((rc2 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), kBf16WideLds, wflags[out3 ? 1 : 0], dev))) return rc2;
      hipLaunchKernelGGL(kern, wgrid, dim3(512), kBf16WideLds, st, a);
      return (int)hipGetLastError();
    }
  }
  if (a.w16 && split) return S2L_E_SIZE;      // (not reached)
  if (out3 && a.w16) hipLaunchKernelGGL(conv3x3_bf16_kernel<true>, grid, dim3(256), 0, st, a);
  else if (out3) hipLaunchKernelGGL(conv3x3_kernel<true>, grid, dim3(256), 0, st, a);
  else if (a.w16) hipLaunchKernelGGL(conv3x3_bf16_kernel<false>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv3x3_kernel<false>, grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}


// ---- input gradient of the frozen eval-mode network (training: the sync loss and the face photometric loss reach the lip MLP
// through it, training.py:436-459, 491-559; the network is frozen and in eval mode once the sync loss is on, train.py:188-197) ---
// dz -> dx of a 3x3 convolution is the same implicit GEMM with the transposed, tap-mirrored chunks (unet_pack_conv_T), no
// bias, no ReLU, and the ReLU mask of the activation that FED the layer applied in the epilogue (`gate`).
static int launch_conv_dgrad(const float* dz, const float* packed, int layer, float* dx, const float* gate, int H, int W,
                             int64_t F, hipStream_t st, const uint16_t* packed16 = nullptr) {
  ConvArgs a;
  a.w16 = packed16 ? packed16 + unet_wT16_off(layer) : nullptr;
  a.split = 0;
  a.inA = dz; a.inB = nullptr; a.CA = kUnetConvs[layer].cout; a.CB = 0;
  a.cout = kUnetConvs[layer].cin;
  a.w = packed + unet_wT_off(layer);
  a.bias = nullptr;
  a.out = dx; a.out3 = nullptr; a.pool = nullptr; a.gate = gate; a.relu = 0;
  a.outw = a.outb = nullptr;
  a.H = H; a.W = W;
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
  a.n_ct = a.cout / 64;
  const int64_t gz = F * a.n_ct;
  if (gz > 65535) return S2L_E_SIZE;
  const dim3 grid(a.tiles_x, a.tiles_y, (unsigned)gz);
  CONV_TRACE_ARGS(a, grid);
  bool done = false;
  const int rc_asm = launch_conv_asm(a, F, st, &done);
  if (done) return rc_asm;
  if (a.w16) {
    bool launched = false;
    const int rc = launch_conv_persistent(a, F, false, false, st, &launched);
    if (rc || launched) return rc;
    hipLaunchKernelGGL(conv3x3_bf16_kernel<false>, grid, dim3(256), 0, st, a);
  }
  else hipLaunchKernelGGL(conv3x3_kernel<false>, grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

// z9[p][c] = (sum_o outw[o][c] * d_out[p][o]) * (y9[p][c] > 0): adjoint of the fused 1x1 output convolution + ReLU mask
__global__ __launch_bounds__(256) void outc_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ outw,
                                                      const float* __restrict__ y9, float* __restrict__ z9, int64_t n_quads) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_quads) return;
  const int c4 = (int)(i & 15) * 4;
  const int64_t p = i >> 4;
  const float d0 = d_out[p * 3], d1 = d_out[p * 3 + 1], d2 = d_out[p * 3 + 2];
  const f4 y = *reinterpret_cast<const f4*>(y9 + p * 64 + c4);
  f4 z;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = fmaf(outw[128 + c4 + r], d2, fmaf(outw[64 + c4 + r], d1, outw[c4 + r] * d0));
    z[r] = y[r] > 0.f ? v : 0.f;
  }
  *reinterpret_cast<f4*>(z9 + p * 64 + c4) = z;
}

// dx[c](y,x) = sum_{co,t} W0[co][c][t] * z0[co](y - (t/3 - 1), x - (t%3 - 1)): input gradient of the first convolution
// (64 -> 3 channels; 0.5 % of the FLOPs, VALU work).  A workgroup owns a 16x16-pixel tile, one pixel per thread; z0 passes
// through LDS 16 channels at a time (18x18 halo, 80-byte pixel pitch: the 16 pixels of a row read conflict-free), so every z0
// value is fetched from memory once instead of nine times -- the first version read its taps straight from global memory and
// was bound by the texture path (144 uncoalesced 16-byte loads per pixel: 1.4 ms per 40-frame window against 0.4 ms of
// FMAs).  The weights are wave-uniform: re-ordered once per call (conv_first_bwd_weights) so that the 12 of a (chunk, tap,
// channel quad) are contiguous, they reach the FMAs as scalar operands (s_load), not through LDS.
constexpr int kFirstBwdPitch = 20;      // floats per halo pixel in LDS
__global__ __launch_bounds__(256) void conv_first_bwd_weights(const float* __restrict__ w, float* __restrict__ ws) {
  const int i = blockIdx.x * 256 + threadIdx.x;      // ws[((kc * 9 + t) * 4 + k4) * 12 + c * 4 + j] = W0[16 kc + 4 k4 + j][c][t]
  if (i >= 4 * 9 * 4 * 12) return;
  const int j = i & 3, c = (i >> 2) % 3, k4 = (i / 12) & 3, t = (i / 48) % 9, kc = i / 432;
  ws[i] = w[(16 * kc + 4 * k4 + j) * 27 + c * 9 + t];
}
__global__ __launch_bounds__(256) void conv_first_bwd_kernel(const float* __restrict__ z0, const float* __restrict__ ws,
                                                            float* __restrict__ dx, int H, int W) {
  __shared__ __attribute__((aligned(16))) float halo[18 * 18 * kFirstBwdPitch];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
  const int64_t frame = blockIdx.z;
  const float* zf = z0 + frame * (int64_t)H * W * 64;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int kc = 0; kc < 4; ++kc) {
    if (kc) __syncthreads();                  // everyone is done with the chunk before
#pragma unroll
    for (int it = 0; it < 6; ++it) {          // 324 halo pixels x 4 quads
      const int idx = threadIdx.x + 256 * it;
      const int p = idx >> 2, quad = idx & 3;
      const int gy = y0 - 1 + p / 18, gx = x0 - 1 + p % 18;
      if (p < 18 * 18) {
        f4 v = (f4){0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
          v = *reinterpret_cast<const f4*>(zf + ((int64_t)gy * W + gx) * 64 + 16 * kc + 4 * quad);
        *reinterpret_cast<f4*>(halo + p * kFirstBwdPitch + 4 * quad) = v;
      }
    }
    __syncthreads();
    const float* wk = ws + kc * 432;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* src = halo + ((ty + 2 - t / 3) * 18 + tx + 2 - t % 3) * kFirstBwdPitch;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const f4 v = *reinterpret_cast<const f4*>(src + 4 * k4);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* ww = wk + (t * 4 + k4) * 12 + c * 4;
          acc[c] = fmaf(v[3], ww[3], fmaf(v[2], ww[2], fmaf(v[1], ww[1], fmaf(v[0], ww[0], acc[c]))));
        }
      }
    }
  }
  const int px = x0 + tx, py = y0 + ty;
  if (px < W && py < H) {
    float* o = dx + (frame * (int64_t)H * W + (int64_t)py * W + px) * 3;
    o[0] = acc[0];
    o[1] = acc[1];
    o[2] = acc[2];
  }
}
static void launch_conv_first_bwd(const float* z0, const float* w0, float* scratch1728, float* dx, int H, int W, int64_t F,
                                  hipStream_t st) {
  hipLaunchKernelGGL(conv_first_bwd_weights, dim3(7), dim3(256), 0, st, w0, scratch1728);
  hipLaunchKernelGGL(conv_first_bwd_kernel, dim3((unsigned)((W + 15) / 16), (unsigned)((H + 15) / 16), (unsigned)F), dim3(256), 0, st, z0,
                     scratch1728, dx, H, W);
}

// Adjoint of upsample2_kernel as a gather (deterministic): input pixel (yi, xi) collects from the output pixels whose
// bilinear taps include it; then the ReLU mask of the activation that was upsampled.  g: [F,Ho,Wo,ldg], channels
// [coff, coff + C) of it; act, z: [F,h,w,C].
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ g, int ldg, int coff,
                                                           const float* __restrict__ act, float* __restrict__ z, int h, int w, int C,
                                                           int Ho, int Wo, UpWin win) {
  int c4, xi, yi;
  int64_t f, i;
  if (!quad_of_thread(w, C / 4, h, c4, xi, yi, f, i)) return;
  const int yif = yi + win.oyi, xif = xi + win.oxi;          // position in the full low-resolution tensor
  const int padT = (win.Hof - 2 * win.hf) / 2, padL = (win.Wof - 2 * win.wf) / 2;
  const float sy = 2 * win.hf > 1 ? (float)(win.hf - 1) / (float)(2 * win.hf - 1) : 0.f;
  const float sx = 2 * win.wf > 1 ? (float)(win.wf - 1) / (float)(2 * win.wf - 1) : 0.f;
  f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
  // (the column weights of the eight candidate columns once: they do not depend on the row; same taps, same order, same bits)
  float wxs[8];
  int xos[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int xu = 2 * xif - 3 + k;
    const float fx = sx * (float)xu;
    const int x0 = (int)fx, x1 = x0 + (x0 < win.wf - 1 ? 1 : 0);
    const float lx = fx - (float)x0;
    const float wx = (x0 == xif ? 1.f - lx : 0.f) + (x1 == xif ? lx : 0.f);
    xos[k] = xu + padL - win.oxo;
    wxs[k] = (xu >= 0 && xu <= 2 * win.wf - 1 && (unsigned)xos[k] < (unsigned)Wo) ? wx : 0.f;
  }
  for (int yu = max(2 * yif - 3, 0); yu <= min(2 * yif + 4, 2 * win.hf - 1); ++yu) {
    const float fy = sy * (float)yu;
    const int y0 = (int)fy, y1 = y0 + (y0 < win.hf - 1 ? 1 : 0);
    const float ly = fy - (float)y0;
    const float wy = (y0 == yif ? 1.f - ly : 0.f) + (y1 == yif ? ly : 0.f);
    const int yo = yu + padT - win.oyo;                       // row of the (cropped) gradient tensor
    if (wy == 0.f || (unsigned)yo >= (unsigned)Ho) continue;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (wxs[k] == 0.f) continue;
      const f4 v = *reinterpret_cast<const f4*>(g + ((f * Ho + yo) * (int64_t)Wo + xos[k]) * ldg + coff + c4 * 4);
      const float ww = wy * wxs[k];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(ww, v[r], acc[r]);
    }
  }
  const f4 a = *reinterpret_cast<const f4*>(act + i * 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = a[r] > 0.f ? acc[r] : 0.f;
  *reinterpret_cast<f4*>(z + i * 4) = acc;
}

// z = (g_skip + MaxPool2d(2)^T(g_pool)) * (x > 0) at the resolution of x: g_skip = channels [0, C) of gcat [F,H,W,ldg] (the
// skip half of the decoder's concat gradient), g_pool [F,H/2,W/2,C] routed to the FIRST maximum of each 2x2 window in scan
// order (ATen's max_pool2d backward), x / pooled = the forward's activation and its pooled copy.
__global__ __launch_bounds__(256) void pool_bwd_add_kernel(const float* __restrict__ gcat, int ldg, const float* __restrict__ gpool,
                                                          const float* __restrict__ x, const float* __restrict__ pooled,
                                                          float* __restrict__ z, int H, int W, int C) {
  int cq4, xx, yy;
  int64_t f, i;
  if (!quad_of_thread(W, C / 4, H, cq4, xx, yy, f, i)) return;
  const int c4 = cq4 * 4;
  const int H2 = H / 2, W2 = W / 2;
  const int64_t pix = (f * H + yy) * (int64_t)W + xx;
  const f4 xv = *reinterpret_cast<const f4*>(x + pix * C + c4);
  f4 o = *reinterpret_cast<const f4*>(gcat + pix * ldg + c4);
  const int py = yy >> 1, px = xx >> 1;
  if (py < H2 && px < W2) {
    const int64_t pp = ((f * H2 + py) * (int64_t)W2 + px) * C + c4;
    const f4 pv = *reinterpret_cast<const f4*>(pooled + pp);
    const f4 gv = *reinterpret_cast<const f4*>(gpool + pp);
    const int k = (yy & 1) * 2 + (xx & 1);          // position in the window's scan order
    bool earlier[4] = {false, false, false, false};
    for (int j = 0; j < k; ++j) {
      const f4 ev = *reinterpret_cast<const f4*>(x + ((f * H + 2 * py + (j >> 1)) * (int64_t)W + 2 * px + (j & 1)) * C + c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) earlier[r] |= ev[r] == pv[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (xv[r] == pv[r] && !earlier[r]) o[r] += gv[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = xv[r] > 0.f ? o[r] : 0.f;
  *reinterpret_cast<f4*>(z + i * 4) = o;
}

// buffers of the saved forward and the backward scratch (floats per frame at H x W: 320 p1 + 640 p2 + 384 p4 saved,
// 256 p1 + 768 p2 + 384 p4 scratch)
struct UnetSaved {
  float *a0, *x1, *uu, *a8, *y9, *p1, *a2, *x2, *u3, *a6, *u1, *p2, *a4, *x3;
};
static UnetSaved unet_saved(float* w, int64_t p1, int64_t p2, int64_t p4) {
  UnetSaved s;
  s.a0 = w;               s.x1 = s.a0 + p1 * 64;  s.uu = s.x1 + p1 * 64;  s.a8 = s.uu + p1 * 64;  s.y9 = s.a8 + p1 * 64;
  s.p1 = s.y9 + p1 * 64;  s.a2 = s.p1 + p2 * 64;  s.x2 = s.a2 + p2 * 128; s.u3 = s.x2 + p2 * 128; s.a6 = s.u3 + p2 * 128;
  s.u1 = s.a6 + p2 * 128; s.p2 = s.u1 + p2 * 64;  s.a4 = s.p2 + p4 * 128; s.x3 = s.a4 + p4 * 128;
  return s;
}

// ================================================================================================================================
// TRAIN mode: the post-fusion U-Net as the reference runs it until `it > 100000` (train.py:188-197) -- every
// BatchNorm2d normalises with the statistics of the batch and updates its running statistics, the weights receive gradients.
// Forward per 3x3 layer: z = conv(a_prev, W) with the RAW weights (same implicit-GEMM kernel, no bias, no ReLU) -> per-channel
// mean / biased variance of z over F*H*W (two-stage, fixed order) -> a = relu(gamma (z - mean) / sqrt(var + eps) + beta)
// (+ MaxPool2d(2)).  Backward per layer: gy = dL/da * (a > 0) (produced by the same kernels as the eval-mode input gradient),
// s1 = sum gy, s2 = sum gy * zhat -> dz = gamma * invstd * (gy - s1/n - zhat * s2/n), dgamma = s2, dbeta = s1 -> weight
// gradient dW[co][ci][tap] = sum_pixels dz[co](p) a_prev[ci](p + tap) (conv_wgrad_kernel, MFMA) and dz -> da_prev (transposed
// chunks).  nn.BatchNorm2d's running update: momentum 0.1, UNBIASED variance.
namespace {
constexpr int kStatBlocks = 1024;   // row blocks of the per-channel reductions (4 per CU keeps the loads in flight)

// partial[blk][0][c] = sum z, partial[blk][1][c] = sum z^2 over the block's pixels; 256 threads = (256 / C) pixel lanes x C
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ z, int C, int64_t n_pix, int64_t per_block,
                                                           float* __restrict__ partial) {
  __shared__ float red[2][256];
  // blockIdx.y = statistics group (1 group: the whole batch, nn.BatchNorm2d; F groups: one per frame -- the reference hands the
  // net one frame per call, so a batch of its calls is F independent normalisations); n_pix = pixels of ONE group
  z += (int64_t)blockIdx.y * n_pix * C;
  partial += (int64_t)blockIdx.y * gridDim.x * 2 * C;
  const int lanes = 256 / C, c = threadIdx.x % C, pl = threadIdx.x / C;
  const int64_t p0 = (int64_t)blockIdx.x * per_block;
  int64_t p1 = p0 + per_block;
  if (p1 > n_pix) p1 = n_pix;
  float s = 0.f, ss = 0.f;
  int64_t p = p0 + pl;
  for (; p + 3 * lanes < p1; p += 4 * lanes) {          // four independent loads in flight per thread
    const float v0 = z[p * C + c], v1 = z[(p + lanes) * C + c], v2 = z[(p + 2 * lanes) * C + c], v3 = z[(p + 3 * lanes) * C + c];
    s += v0; ss = fmaf(v0, v0, ss);
    s += v1; ss = fmaf(v1, v1, ss);
    s += v2; ss = fmaf(v2, v2, ss);
    s += v3; ss = fmaf(v3, v3, ss);
  }
  for (; p < p1; p += lanes) {
    const float v = z[p * C + c];
    s += v;
    ss = fmaf(v, v, ss);
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int l = 0; l < lanes; ++l) {
      a += red[0][l * C + threadIdx.x];
      b += red[1][l * C + threadIdx.x];
    }
    partial[((int64_t)blockIdx.x * 2) * C + threadIdx.x] = a;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

// per channel: batch mean, biased variance -> st[0..C) = scale = gamma * invstd, st[C..2C) = shift = beta - mean * scale,
// st[2C..3C) = mean, st[3C..4C) = invstd; running statistics updated in place (momentum, unbiased variance)
// one 64-lane block per channel: lane t sums the partials of blocks t, t + 64, ... in fp64, then a fixed-shape butterfly
__device__ __forceinline__ void channel_totals(const float* __restrict__ partial, int n_blocks, int C, int c, double* s0, double* s1) {
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < n_blocks; k += 64) {
    a += (double)partial[((int64_t)k * 2) * C + c];
    b += (double)partial[((int64_t)k * 2 + 1) * C + c];
  }
  for (int o = 32; o; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  *s0 = a;
  *s1 = b;
}

__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ partial, int n_blocks, int C, double n, float eps,
                                                        float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        float* __restrict__ st, int groups) {
  const int c = blockIdx.x;
  // groups in order: the running statistics move once per group, as they do over the reference's successive one-frame calls
  for (int g = 0; g < groups; ++g) {
    double s, ss;
    channel_totals(partial + (int64_t)g * n_blocks * 2 * C, n_blocks, C, c, &s, &ss);
    if (threadIdx.x != 0) continue;
    float* stg = st + g * 512;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * invstd;
    stg[c] = scale;
    stg[C + c] = beta[c] - (float)mean * scale;
    stg[2 * C + c] = (float)mean;
    stg[3 * C + c] = invstd;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
    }
  }
}

// the same per group with the groups in PARALLEL (grid C x groups), the running statistics left to bn_running_kernel: with forty
// frames per launch the in-order loop above was 80 us of sequential 1024-term reductions per layer.  Each block leaves the two
// numbers its group contributes to the running statistics in its own slots of the partial buffer (row 0 of its group; nobody else
// reads this channel's column).
__global__ __launch_bounds__(64) void bn_finalize_groups_kernel(float* __restrict__ partial, int n_blocks, int C, double n, float eps,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ st) {
  const int c = blockIdx.x, g = blockIdx.y;
  float* pg = partial + (int64_t)g * n_blocks * 2 * C;
  double s, ss;
  channel_totals(pg, n_blocks, C, c, &s, &ss);
  if (threadIdx.x != 0) return;
  float* stg = st + g * 512;
  const double mean = s / n;
  double var = ss / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float scale = gamma[c] * invstd;
  stg[c] = scale;
  stg[C + c] = beta[c] - (float)mean * scale;
  stg[2 * C + c] = (float)mean;
  stg[3 * C + c] = invstd;
  pg[c] = (float)mean;
  pg[C + c] = (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
}
// running statistics: one momentum update per group, in group (= frame) order; thread = channel
__global__ __launch_bounds__(128) void bn_running_kernel(const float* __restrict__ partial, int n_blocks, int C, int groups, float momentum,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float rm = running_mean[c], rv = running_var[c];
  for (int g = 0; g < groups; ++g) {
    const float* pg = partial + (int64_t)g * n_blocks * 2 * C;
    rm = (1.f - momentum) * rm + momentum * pg[c];
    rv = (1.f - momentum) * rv + momentum * pg[C + c];
  }
  running_mean[c] = rm;
  running_var[c] = rv;
}

// a = relu(z * scale + shift); pool != NULL: also MaxPool2d(2) of a ([F,H/2,W/2,C]), one thread per (pixel, channel quad)
__global__ __launch_bounds__(256) void bn_relu_kernel(const float* __restrict__ z, const float* __restrict__ st, float* __restrict__ a,
                                                     int C, int64_t n_quads, int64_t group_quads) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_quads) return;
  st += (i / group_quads) * 512;
  const int c4 = (int)(i % (C / 4)) * 4;
  const f4 v = *reinterpret_cast<const f4*>(z + i * 4);
  const f4 sc = *reinterpret_cast<const f4*>(st + c4), sh = *reinterpret_cast<const f4*>(st + C + c4);
  f4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = fmaxf(fmaf(v[r], sc[r], sh[r]), 0.f);
  *reinterpret_cast<f4*>(a + i * 4) = o;
}
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ a, float* __restrict__ p, int H, int W, int C,
                                                      int64_t n_quads) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_quads) return;
  const int cq = C / 4, H2 = H / 2, W2 = W / 2;
  const int c4 = (int)(i % cq) * 4;
  int64_t q = i / cq;
  const int x = (int)(q % W2);
  q /= W2;
  const int y = (int)(q % H2);
  const int64_t f = q / H2;
  const float* s = a + ((f * H + 2 * y) * (int64_t)W + 2 * x) * C + c4;
  const f4 v0 = *reinterpret_cast<const f4*>(s), v1 = *reinterpret_cast<const f4*>(s + C);
  const f4 v2 = *reinterpret_cast<const f4*>(s + (int64_t)W * C), v3 = *reinterpret_cast<const f4*>(s + (int64_t)W * C + C);
  f4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = fmaxf(fmaxf(v0[r], v1[r]), fmaxf(v2[r], v3[r]));
  *reinterpret_cast<f4*>(p + i * 4) = o;
}

// out[p][o] = sum_c w[o][c] a[p][c] + b[o]: the 1x1 output convolution on its own (eval mode fuses it into the last conv)
__global__ __launch_bounds__(256) void outc_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b,
                                                  float* __restrict__ out, int64_t n_pix) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  float acc[3] = {0.f, 0.f, 0.f};
  const f4* s = reinterpret_cast<const f4*>(a + p * 64);
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const f4 v = s[k];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[o] = fmaf(w[o * 64 + 4 * k + r], v[r], acc[o]);
  }
  out[p * 3] = acc[0] + b[0];
  out[p * 3 + 1] = acc[1] + b[1];
  out[p * 3 + 2] = acc[2] + b[2];
}

// BatchNorm backward, stage 1: partial[blk][0][c] = sum gy, partial[blk][1][c] = sum gy * zhat  (zhat = (z - mean) * invstd)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ z,
                                                           const float* __restrict__ st, int C, int64_t n_pix, int64_t per_block,
                                                           float* __restrict__ partial) {
  __shared__ float red[2][256];
  gy += (int64_t)blockIdx.y * n_pix * C;       // (blockIdx.y = statistics group, as in channel_stats_kernel)
  z += (int64_t)blockIdx.y * n_pix * C;
  st += blockIdx.y * 512;
  partial += (int64_t)blockIdx.y * gridDim.x * 2 * C;
  const int lanes = 256 / C, c = threadIdx.x % C, pl = threadIdx.x / C;
  const float mean = st[2 * C + c], invstd = st[3 * C + c];
  const int64_t p0 = (int64_t)blockIdx.x * per_block;
  int64_t p1 = p0 + per_block;
  if (p1 > n_pix) p1 = n_pix;
  float s1 = 0.f, s2 = 0.f;
  int64_t p = p0 + pl;
  for (; p + lanes < p1; p += 2 * lanes) {               // two rows (four loads) in flight per thread
    const float g0 = gy[p * C + c], z0 = z[p * C + c], g1 = gy[(p + lanes) * C + c], z1 = z[(p + lanes) * C + c];
    s1 += g0; s2 = fmaf(g0, (z0 - mean) * invstd, s2);
    s1 += g1; s2 = fmaf(g1, (z1 - mean) * invstd, s2);
  }
  for (; p < p1; p += lanes) {
    const float g = gy[p * C + c];
    s1 += g;
    s2 = fmaf(g, (z[p * C + c] - mean) * invstd, s2);
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int l = 0; l < lanes; ++l) {
      a += red[0][l * C + threadIdx.x];
      b += red[1][l * C + threadIdx.x];
    }
    partial[((int64_t)blockIdx.x * 2) * C + threadIdx.x] = a;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}
// stage 2: totals -> dgamma = s2, dbeta = s1, and the two per-channel means the elementwise stage needs (sums[0..C) = s1/n,
// sums[C..2C) = s2/n)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int n_blocks, int C, double n,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ sums, const float* __restrict__ st_raw) {
  const int c = blockIdx.x, g = blockIdx.y;      // (per group: its own sums; dgamma / dbeta are per-group scratch then)
  double s1, s2;
  channel_totals(partial + (int64_t)g * n_blocks * 2 * C, n_blocks, C, c, &s1, &s2);
  if (threadIdx.x != 0) return;
  // st_raw: the blocks hold sum g and sum g z (a convolution's backward statistics): sum g zhat = invstd (sum g z - mean sum g) is formed here
  if (st_raw) s2 = (s2 - (double)st_raw[g * 512 + 2 * C + c] * s1) * (double)st_raw[g * 512 + 3 * C + c];
  dgamma[g * 256 + c] = (float)s2;
  dbeta[g * 256 + c] = (float)s1;
  sums[g * 256 + c] = (float)(s1 / n);
  sums[g * 256 + C + c] = (float)(s2 / n);
}
// per-group dgamma / dbeta ([group][256]: 128 + 128) -> their sums over the groups, in group order (a TRAINING net whose frames are
// separate statistics groups: the batch's parameter gradient is the sum of the per-call gradients)
__global__ __launch_bounds__(128) void sum_groups_kernel(const float* __restrict__ per_group, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        int C, int groups) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < groups; ++g) {
    a += per_group[g * 256 + c];
    b += per_group[g * 256 + 128 + c];
  }
  dgamma[c] = a;
  dbeta[c] = b;
}
// stage 3 (in place): dz = scale * (gy - s1/n - zhat * s2/n)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ st,
                                                          const float* __restrict__ sums, int C, int64_t n_quads, int64_t group_quads) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_quads) return;
  st += (i / group_quads) * 512;
  sums += (i / group_quads) * 256;
  const int c4 = (int)(i % (C / 4)) * 4;
  f4 g = *reinterpret_cast<const f4*>(gy + i * 4);
  const f4 v = *reinterpret_cast<const f4*>(z + i * 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float zhat = (v[r] - st[2 * C + c4 + r]) * st[3 * C + c4 + r];
    g[r] = st[c4 + r] * (g[r] - sums[c4 + r] - zhat * sums[C + c4 + r]);
  }
  *reinterpret_cast<f4*>(gy + i * 4) = g;
}

// ---- weight gradient of a 3x3 convolution on MFMA: dW[co][ci][t] = sum_{f,y,x} dz[f,y,x,co] * a[f, y + t/3 - 1, x + t%3 - 1, ci].
// GEMM per tap with K = pixels: A[i = co][k = pixel] = dz, B[k = pixel][j = ci] = the shifted input.  Workgroup = (64 output
// channels, 16 input channels, all 9 taps), wave w owns output channels 16 w .. 16 w + 15 (9 accumulators of 4 registers).
// K runs over chunks of 64 consecutive pixels of one image row; a chunk stages dz [64 px][64 co] and the 3 x 66-pixel halo of
// the input [3][66][16 ci] in LDS.  The chunks are dealt round-robin to gridDim.z workgroups (split K); their partial results
// are summed in a fixed order by wgrad_reduce_lanes_kernel.
struct WgradArgs {
  const float* dz;     // [F,H,W,cout]
  const float* inA;    // [F,H,W,CA]
  const float* inB;    // [F,H,W,CB] or null (virtual concat, channels of A first)
  float* partial;      // [S][cout][cin][9]
  int CA, CB, cout, H, W, F, chunks_x;
  int64_t n_chunks;
};
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_dz[64 * 68];
  __shared__ __attribute__((aligned(16))) float lds_a[3 * 66 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, i16 = lane & 15;
  const int ct = blockIdx.x, cc = blockIdx.y;
  const int cin = a.CA + a.CB;
  const bool fromA = cc * 16 < a.CA;
  const float* in = fromA ? a.inA : a.inB;
  const int Cin = fromA ? a.CA : a.CB;
  const int coff = fromA ? cc * 16 : cc * 16 - a.CA;
  f4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
  // register-prefetch pipeline over the chunks: the global loads of the next chunk are in flight while this chunk's 144 MFMAs
  // per wave run; they are committed to LDS between two barriers (as conv3x3_kernel does for its channel chunks)
  f4 pd[4], pa[4];
  auto fetch = [&](int64_t ch) {
    const int cx = (int)(ch % a.chunks_x);
    const int64_t row = ch / a.chunks_x;                  // f * H + y
    const int y = (int)(row % a.H);
    const int64_t f = row / a.H;
    const int x0 = cx * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                         // dz tile: 64 px x 64 co = 1024 quads
      const int idx = threadIdx.x + 256 * k;
      const int px = idx >> 4, c4 = idx & 15;
      pd[k] = (f4){0.f, 0.f, 0.f, 0.f};
      if (x0 + px < a.W) pd[k] = *reinterpret_cast<const f4*>(a.dz + ((row * a.W) + x0 + px) * a.cout + ct * 64 + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                         // input halo: 3 rows x 66 px x 16 ci = 792 quads
      const int idx = threadIdx.x + 256 * k;
      const int c4 = idx & 3, xx = (idx >> 2) % 66, dy = idx / (66 * 4);
      const int gy = y + dy - 1, gx = x0 + xx - 1;
      pa[k] = (f4){0.f, 0.f, 0.f, 0.f};
      if (idx < 3 * 66 * 4 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
        pa[k] = *reinterpret_cast<const f4*>(in + ((f * a.H + gy) * (int64_t)a.W + gx) * Cin + coff + 4 * c4);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = threadIdx.x + 256 * k;
      *reinterpret_cast<f4*>(lds_dz + (idx >> 4) * 68 + 4 * (idx & 15)) = pd[k];
      if (idx < 3 * 66 * 4) *reinterpret_cast<f4*>(lds_a + (idx >> 2) * 16 + 4 * (idx & 3)) = pa[k];
    }
  };
  if ((int64_t)blockIdx.z < a.n_chunks) {
    fetch(blockIdx.z);
    commit();
  }
  __syncthreads();
  for (int64_t ch = blockIdx.z; ch < a.n_chunks; ch += gridDim.z) {
    const bool more = ch + gridDim.z < a.n_chunks;
    if (more) fetch(ch + gridDim.z);
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const float av = lds_dz[(4 * s + q) * 68 + 16 * wave + i16];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = mfma16u(av, lds_a[((t / 3) * 66 + 4 * s + q + t % 3) * 16 + i16], acc[t]);
    }
    __syncthreads();            // everyone is done reading this chunk
    if (more) {
      commit();
      __syncthreads();
    }
  }
  // D[row = 4 q + r -> co][col = i16 -> ci]
  float* p = a.partial + (int64_t)blockIdx.z * a.cout * cin * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[((int64_t)(ct * 64 + 16 * wave + 4 * q + r) * cin + cc * 16 + i16) * 9 + t] = acc[t][r];
}
// The same weight gradient with bf16 OPERANDS (the precision of the training chain's other convolutions; fp32 accumulation, fp32
// tensors in memory): dW[co][ci][t] on v_mfma_f32_16x16x32_bf16, K = 32 pixels per instruction.  Workgroup = (64 output channels, 32
// input channels, 9 taps), wave w owns output channels 16 w .. + 15 (9 x 2 accumulators of 4 registers).  A 16-bit MFMA operand is 8
// CONSECUTIVE k per lane, i.e. 8 consecutive pixels of one channel -- NHWC has them C floats apart -- so the transposition happens in
// registers on the way into LDS: a thread fetches 8 (dz) or 10 (input, with the two neighbours the tap shifts need) consecutive pixels
// of one channel quad, converts, and writes 16-byte pixel octets: dzT [co 64][px 64], aT [dy 3][dx 3][ci 32][px 64] -- one copy of the
// input rows PER dx, so that every tap's operand is an aligned ds_read_b128 (a one-pixel shift of a packed octet would not be).
// K chunks, split K and the partial layout as conv_wgrad_kernel (wgrad_reduce_lanes_kernel sums them).  16 x the matrix rate of the
// fp32 form: the kernel is bound by reading dz (cin / 32 times) and the input (cout / 64 times).
constexpr int kWgPitch = 72;      // halves per LDS row: 64 pixels + 8 of padding (144 bytes: rows stay 16-byte aligned)
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t lds_dz[64 * kWgPitch];
  __shared__ __attribute__((aligned(16))) uint16_t lds_a[9 * 32 * kWgPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kg = lane >> 4, i16 = lane & 15;
  const int ct = blockIdx.x, cc = blockIdx.y;      // cc: 32-channel block of the (virtually concatenated) input
  const int cin = a.CA + a.CB;
  const bool fromA = cc * 32 < a.CA;
  const float* in = fromA ? a.inA : a.inB;
  const int Cin = fromA ? a.CA : a.CB;
  const int coff = fromA ? cc * 32 : cc * 32 - a.CA;
  f4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = (f4){0.f, 0.f, 0.f, 0.f};
  // fetch tasks.  dz: thread < 128 = (pixel octet pg = tid >> 4, channel quad cq = tid & 15): 8 pixels x 4 channels.
  // input: thread < 192 = (row dy = tid / 64, pixel octet pg = (tid >> 3) & 7, channel quad cq = tid & 7): 10 pixels x 4 channels.
  f4 pd[8], pa[10];
  const bool has_dz = threadIdx.x < 128, has_a = threadIdx.x < 192;
  const int dpg = threadIdx.x >> 4, dcq = threadIdx.x & 15;
  const int ady = threadIdx.x >> 6, apg = (threadIdx.x >> 3) & 7, acq = threadIdx.x & 7;
  auto fetch = [&](int64_t ch) {
    const int cx = (int)(ch % a.chunks_x);
    const int64_t row = ch / a.chunks_x;                  // f * H + y
    const int y = (int)(row % a.H);
    const int64_t f = row / a.H;
    const int x0 = cx * 64;
    if (has_dz) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int x = x0 + 8 * dpg + k;
        pd[k] = (f4){0.f, 0.f, 0.f, 0.f};
        if (x < a.W) pd[k] = *reinterpret_cast<const f4*>(a.dz + (row * a.W + x) * a.cout + ct * 64 + 4 * dcq);
      }
    }
    if (has_a) {
      const int gy = y + ady - 1;
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int gx = x0 + 8 * apg + k - 1;
        pa[k] = (f4){0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
          pa[k] = *reinterpret_cast<const f4*>(in + ((f * a.H + gy) * (int64_t)a.W + gx) * Cin + coff + 4 * acq);
      }
    }
  };
  auto commit = [&]() {
    if (has_dz) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(pd[2 * k][c], pd[2 * k + 1][c]);
        *reinterpret_cast<u4v*>(lds_dz + (4 * dcq + c) * kWgPitch + 8 * dpg) = o;
      }
    }
    if (has_a) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          u4v o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(pa[dx + 2 * k][c], pa[dx + 2 * k + 1][c]);
          *reinterpret_cast<u4v*>(lds_a + ((ady * 3 + dx) * 32 + 4 * acq + c) * kWgPitch + 8 * apg) = o;
        }
    }
  };
  if ((int64_t)blockIdx.z < a.n_chunks) {
    fetch(blockIdx.z);
    commit();
  }
  __syncthreads();
  for (int64_t ch = blockIdx.z; ch < a.n_chunks; ch += gridDim.z) {
    const bool more = ch + gridDim.z < a.n_chunks;
    if (more) fetch(ch + gridDim.z);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {      // 32 pixels per k-step; lane (row / column i16, pixel octet kg)
      const u4v av = *reinterpret_cast<const u4v*>(lds_dz + (16 * wave + i16) * kWgPitch + 32 * ks + 8 * kg);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const u4v bv = *reinterpret_cast<const u4v*>(lds_a + (t * 32 + 16 * cb + i16) * kWgPitch + 32 * ks + 8 * kg);
          acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, av), __builtin_bit_cast(bf8v, bv), acc[t][cb], 0, 0, 0);
        }
    }
    __syncthreads();            // everyone is done reading this chunk
    if (more) {
      commit();
      __syncthreads();
    }
  }
  // D[row = 4 kg + r -> co][col = i16 -> ci]
  float* p = a.partial + (int64_t)blockIdx.z * a.cout * cin * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        p[((int64_t)(ct * 64 + 16 * wave + 4 * kg + r) * cin + cc * 32 + 16 * cb + i16) * 9 + t] = acc[t][cb][r];
}
// sum of the split-K partials in a fixed order: 64 elements per workgroup, four threads per element take every fourth partial
// PL lanes of a block share an element: lane pl sums partials pl, pl + PL, ... in that order, the PL sums are added in lane order (fixed
// orders: deterministic).  PL = 4 (256 threads) for the 3x3 layers, PL = 16 (1024 threads) for the small tensors of the first and the
// output layer, whose ~1000 .. 2000 partials made the four-lane form a 125-us serial loop.
template <int PL>
__global__ __launch_bounds__(64 * PL) void wgrad_reduce_lanes_kernel(const float* __restrict__ partial, float* __restrict__ out, int n_parts,
                                                                    int64_t n) {
  __shared__ float red[PL][64];
  const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < n)
    for (int b = pl; b < n_parts; b += PL) s += partial[(int64_t)b * n + e];
  red[pl][el] = s;
  __syncthreads();
  if (pl == 0 && e < n) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < PL; ++k) t += red[k][el];
    out[e] = t;
  }
}

// the same sums for partials laid out [part][tap 9][n / 9] (conv_wgrad_h_kernel: a wave's store is then 128 contiguous bytes per row instead of
// 64 pieces 36 bytes apart); out stays [n / 9][9]
template <int PL>
__global__ __launch_bounds__(64 * PL) void wgrad_reduce_taps_kernel(const float* __restrict__ partial, float* __restrict__ out, int n_parts,
                                                                   int64_t n) {
  __shared__ float red[PL][64];
  const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < n)
    for (int b = pl; b < n_parts; b += PL) s += partial[(int64_t)b * n + e];
  red[pl][el] = s;
  __syncthreads();
  if (pl == 0 && e < n) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < PL; ++k) t += red[k][el];
    const int64_t per_tap = n / 9;
    out[(e % per_tap) * 9 + e / per_tap] = t;
  }
}

// first convolution (3 input channels): partial[blk][co][c*9 + t] = sum over the block's pixels of dz[p][co] x[neighbour t of p][c], on
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chains): K = pixels, four per instruction; A = dz^T (lane (co within its 16-block, pixel q)),
// B = the pixel's 27 inputs padded to 32 (lane (k within its 16-block, pixel q): one gathered value), 4 x 2 accumulators per wave.
// (As a loop of 27 scalar-operand FMAs per pixel this kernel took 1.3 ms per 8 frames, 10 x the time of its HBM traffic.)
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                              float* __restrict__ partial, int H, int W, int64_t n_pix,
                                                              int64_t per_block) {
  __shared__ float red[4][64 * 27];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, i16 = lane & 15;
  f4 acc[4][2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = (f4){0.f, 0.f, 0.f, 0.f};
  // this lane's two input taps: k = 16 kb + i16 = c * 9 + t
  int kc[2], kdy[2], kdx[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int k = 16 * kb + i16, c = k / 9, t = k - 9 * c;
    kc[kb] = k < 27 ? c : -1;
    kdy[kb] = t / 3 - 1;
    kdx[kb] = t % 3 - 1;
  }
  const int64_t p0 = (int64_t)blockIdx.x * per_block;
  int64_t p1 = p0 + per_block;
  if (p1 > n_pix) p1 = n_pix;
  for (int64_t pb = p0 + 4 * wave; pb < p1; pb += 16) {      // four pixels per wave and step
    const int64_t p = pb + q;
    const bool live = p < p1;
    const int xx = live ? (int)(p % W) : 0;
    const int64_t r = live ? p / W : 0;
    const int yy = (int)(r % H);
    const int64_t f = r / H;
    float av[4], bv[2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) av[mb] = live ? dz[p * 64 + 16 * mb + i16] : 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int gy = yy + kdy[kb], gx = xx + kdx[kb];
      const bool ok = live && kc[kb] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      bv[kb] = ok ? x[((f * H + gy) * (int64_t)W + gx) * 3 + kc[kb]] : 0.f;
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) acc[mb][kb] = mfma16u(av[mb], bv[kb], acc[mb][kb]);
  }
  // D[row = 4 q + r -> co = 16 mb + 4 q + r][col = i16 -> k = 16 kb + i16]
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kb + i16;
        if (k < 27) red[wave][(16 * mb + 4 * q + r) * 27 + k] = acc[mb][kb][r];
      }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 27; i += 256)
    partial[(int64_t)blockIdx.x * 64 * 27 + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}
// d outc.weight [3][64] = sum_p d_out[p][o] a9[p][c]; d outc.bias [3] = sum_p d_out[p][o]: partial[blk][3*64 + 3]
__global__ __launch_bounds__(256) void outc_wgrad_kernel(const float* __restrict__ d_out, const float* __restrict__ a9,
                                                        float* __restrict__ partial, int64_t n_pix, int64_t per_block) {
  __shared__ float red[4][196];
  const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
  float acc[3] = {0.f, 0.f, 0.f}, bs[3] = {0.f, 0.f, 0.f};
  const int64_t p0 = (int64_t)blockIdx.x * per_block;
  int64_t p1 = p0 + per_block;
  if (p1 > n_pix) p1 = n_pix;
  for (int64_t p = p0 + pl; p < p1; p += 4) {
    const float v = a9[p * 64 + c];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float d = d_out[p * 3 + o];
      acc[o] = fmaf(d, v, acc[o]);
      bs[o] += d;
    }
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) red[pl][o * 64 + c] = acc[o];
  if (c < 3) red[pl][192 + c] = bs[c];
  __syncthreads();
  if (threadIdx.x < 195) partial[(int64_t)blockIdx.x * 195 + threadIdx.x] =
      ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

struct TrainBufs {
  float *z[10], *act[10];     // pre-BatchNorm outputs and post-ReLU activations of the ten convolutions
  float *p1, *p2, *u3, *uu;   // pooled / up-sampled inputs
  float* st;                  // [10][4][128]: scale, shift, mean, invstd
};
static TrainBufs train_bufs(float* w, int64_t p1, int64_t p2, int64_t p4) {
  const int64_t pl[3] = {p1, p2, p4};
  TrainBufs b;
  for (int l = 0; l < 10; ++l) {
    b.z[l] = w;
    w += pl[kLvl[l]] * kUnetConvs[l].cout;
    b.act[l] = w;
    w += pl[kLvl[l]] * kUnetConvs[l].cout;
  }
  b.p1 = w;  w += p2 * 64;
  b.p2 = w;  w += p4 * 128;
  b.u3 = w;  w += p2 * 128;
  b.uu = w;  w += p1 * 64;
  b.st = w;
  return b;
}
static int64_t train_saved_floats(int64_t p1, int64_t p2, int64_t p4, int64_t groups = 1) {
  const int64_t pl[3] = {p1, p2, p4};
  int64_t n = 0;
  for (int l = 0; l < 10; ++l) n += 2 * pl[kLvl[l]] * kUnetConvs[l].cout;
  return n + p2 * 64 + p4 * 128 + p2 * 128 + p1 * 64 + 10 * 512 * groups;      // st: [10][groups][4][128]
}
// offsets of the gradients in the flat output of s2l_unet_train_backward: per layer conv.weight, bn.weight, bn.bias; then
// outc.conv.weight [3,64], outc.conv.bias [3]
static int64_t grad_off(int layer) {
  int64_t off = 0;
  for (int l = 0; l < layer; ++l) off += (int64_t)kUnetConvs[l].cout * kUnetConvs[l].cin * 9 + 2 * kUnetConvs[l].cout;
  return off;
}
}  // namespace

}  // namespace s2l

using namespace s2l;

extern "C" int64_t s2l_unet_packed_floats(void) { return kUnetPackedFloats; }

// Workspace floats for n_frames frames of height x width (all intermediate activations).
extern "C" int64_t s2l_unet_work_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 0) return 0;
  const int64_t p1 = (int64_t)height * width, p2 = (int64_t)(height / 2) * (width / 2),
                p4 = (int64_t)(height / 4) * (width / 4);
  // t64a, x1, t64b (H) | pool1(64), t128a, x2, up1in(128), t128b, u1(64) (H/2) | pool2, t128c, x3 (H/4)
  return n_frames * (p1 * (64 + 64 + 64) + p2 * (64 + 128 + 128 + 128 + 128 + 64) + p4 * (128 + 128 + 128));
}

// tensors_host: 52 DEVICE pointers: for each of the ten 3x3 convs (execution order, see weights.UNET_CONVS)
// {conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var}, then outc.conv.weight, outc.conv.bias.
extern "C" int s2l_unet_pack(const float* const* tensors_host, float bn_eps, float* packed, s2l_stream_t stream) {
  if (!tensors_host || !packed) return S2L_E_NULL;
  UnetTensors t;
  for (int l = 0; l < 10; ++l) {
    for (int k = 0; k < 5; ++k)
      if (!tensors_host[l * 5 + k]) return S2L_E_NULL;
    t.w[l] = tensors_host[l * 5];
    t.gamma[l] = tensors_host[l * 5 + 1];
    t.beta[l] = tensors_host[l * 5 + 2];
    t.mean[l] = tensors_host[l * 5 + 3];
    t.var[l] = tensors_host[l * 5 + 4];
  }
  if (!tensors_host[50] || !tensors_host[51]) return S2L_E_NULL;
  t.outw = tensors_host[50];
  t.outb = tensors_host[51];
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(unet_pack_convs_kernel, dim3((unsigned)((kPackMaxFloats + 255) / 256), 9, 2), dim3(256), 0, st, t, packed, bn_eps);
  hipLaunchKernelGGL(unet_pack_misc, dim3(16), dim3(256), 0, st, t, packed, bn_eps);
  return (int)hipGetLastError();
}

// x [F,H,W,3] NHWC -> out [F,H,W,3].  H, W >= 4.  work: s2l_unet_work_floats(H, W, F) floats.
static int unet_forward_impl(const float* packed, const uint16_t* packed16, int split, const float* x, float* work, float* out,
                             int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (height < 4 || width < 4 || n_frames < 0) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!packed || !x || !work || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(work)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int H = height, W = width, H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2;
  const int64_t F = n_frames, p1 = (int64_t)H * W * F, p2 = (int64_t)H2 * W2 * F, p4 = (int64_t)H4 * W4 * F;
  float* t64a = work;              float* x1 = t64a + p1 * 64;      float* t64b = x1 + p1 * 64;
  float* pool1 = t64b + p1 * 64;   float* t128a = pool1 + p2 * 64;  float* x2 = t128a + p2 * 128;
  float* up1in = x2 + p2 * 128;    float* t128b = up1in + p2 * 128; float* u1 = t128b + p2 * 128;
  float* pool2 = u1 + p2 * 64;     float* t128c = pool2 + p4 * 128; float* x3 = t128c + p4 * 128;
  int rc;
  if (F > 65535) return S2L_E_SIZE;
  hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)((H + 15) / 16), (unsigned)F), dim3(256), 0, st, x,
                     packed + unet_w_off(0), packed + unet_b_off(0), t64a, H, W);
  if ((rc = launch_conv(t64a, 64, nullptr, 0, packed, 1, x1, nullptr, H, W, F, st, pool1, nullptr, packed16, split))) return rc;   // + MaxPool2d(2)
  if ((rc = launch_conv(pool1, 64, nullptr, 0, packed, 2, t128a, nullptr, H2, W2, F, st, nullptr, nullptr, packed16, split))) return rc;
  if ((rc = launch_conv(t128a, 128, nullptr, 0, packed, 3, x2, nullptr, H2, W2, F, st, pool2, nullptr, packed16, split))) return rc;   // + MaxPool2d(2)
  if ((rc = launch_conv(pool2, 128, nullptr, 0, packed, 4, t128c, nullptr, H4, W4, F, st, nullptr, nullptr, packed16, split))) return rc;
  if ((rc = launch_conv(t128c, 128, nullptr, 0, packed, 5, x3, nullptr, H4, W4, F, st, nullptr, nullptr, packed16, split))) return rc;
  hipLaunchKernelGGL(upsample2_kernel, quad_grid(W2, 128, H2, F), dim3(256), 0, st, x3, up1in, H4, W4, 128, H2, W2,
                     no_window(H4, W4, H2, W2));
  if ((rc = launch_conv(x2, 128, up1in, 128, packed, 6, t128b, nullptr, H2, W2, F, st, nullptr, nullptr, packed16, split))) return rc;
  if ((rc = launch_conv(t128b, 128, nullptr, 0, packed, 7, u1, nullptr, H2, W2, F, st, nullptr, nullptr, packed16, split))) return rc;
  hipLaunchKernelGGL(upsample2_kernel, quad_grid(W, 64, H, F), dim3(256), 0, st, u1, t64a, H2, W2, 64, H, W,
                     no_window(H2, W2, H, W));   // t64a is free again: it becomes up(u1)
  if ((rc = launch_conv(x1, 64, t64a, 64, packed, 8, t64b, nullptr, H, W, F, st, nullptr, nullptr, packed16, split))) return rc;
  if ((rc = launch_conv(t64b, 64, nullptr, 0, packed, 9, nullptr, out, H, W, F, st, nullptr, nullptr, packed16, split))) return rc;
  return (int)hipGetLastError();
}

extern "C" int s2l_unet_forward(const float* packed, const uint16_t* packed16, const float* x, float* work, float* out, int height,
                                int width, int64_t n_frames, s2l_stream_t stream) {
  return unet_forward_impl(packed, packed16, 0, x, work, out, height, width, n_frames, stream);
}
// The same network with the nine 3x3 layers in the split-bf16 operand form (s2l_unet_pack16x3): fp32-grade results (~1e-6 of the
// output scale) at a multiple of the fp32-MFMA rate.
extern "C" int s2l_unet_forward_split(const float* packed, const uint16_t* packed16x3, const float* x, float* work, float* out,
                                      int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (!packed16x3) return S2L_E_NULL;
  if (misaligned16(packed16x3)) return S2L_E_ALIGN;
  return unet_forward_impl(packed, packed16x3, 1, x, work, out, height, width, n_frames, stream);
}

// ---- training: forward that keeps every activation, and the input gradient ----------------------------------------------
extern "C" int64_t s2l_unet_saved_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 0) return 0;
  const int64_t p1 = (int64_t)height * width, p2 = (int64_t)(height / 2) * (width / 2), p4 = (int64_t)(height / 4) * (width / 4);
  return n_frames * (p1 * 320 + p2 * 640 + p4 * 384);
}
extern "C" int64_t s2l_unet_backward_work_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 0) return 0;
  const int64_t p1 = (int64_t)height * width, p2 = (int64_t)(height / 2) * (width / 2), p4 = (int64_t)(height / 4) * (width / 4);
  return n_frames * (p1 * 256 + p2 * 768 + p4 * 384);
}

// A window [origin_y, origin_y + height) x [origin_x, origin_x + width) of a full_h x full_w frame (origins multiples of 4, so
// that the 2x2 pooling windows and the quarter-resolution grid line up with the full frame's): the crop is processed as a frame of
// its own, except that the bilinear up-samplings take their source positions from the FULL frame's geometry.  Every output
// (and, in the backward, every gradient) whose dependency cone stays inside the crop equals the full-frame value bit for bit; the
// cone has a radius of <= 32 pixels (2 + 4 + 8 through the encoder, 4 + 4 + 2 + 2 back up, pooling alignment).  Values nearer
// than that to a crop edge that is not a frame edge are NOT the full-frame values: the caller must not use them.
static int unet_window(int height, int width, int full_h, int full_w, int oy, int ox) {
  if (full_h < height || full_w < width || oy < 0 || ox < 0 || oy + height > full_h || ox + width > full_w) return S2L_E_GEOMETRY;
  if ((oy & 3) || (ox & 3)) return S2L_E_GEOMETRY;
  // a crop that stops short of the frame's bottom / right edge must have a size that pools evenly
  if ((oy + height != full_h && (height & 3)) || (ox + width != full_w && (width & 3))) return S2L_E_GEOMETRY;
  return 0;
}

extern "C" int s2l_unet_forward_saved_window(const float* packed, const uint16_t* packed16, const float* x, float* saved, float* out,
                                             int height, int width, int full_h, int full_w, int origin_y, int origin_x,
                                             int64_t n_frames, s2l_stream_t stream) {
  if (height < 4 || width < 4 || n_frames < 0) return S2L_E_SIZE;
  { const int rcw = unet_window(height, width, full_h, full_w, origin_y, origin_x); if (rcw) return rcw; }
  if (n_frames == 0) return S2L_OK;
  if (!packed || !x || !saved || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(saved)) return S2L_E_ALIGN;
  if (n_frames > 65535) return S2L_E_SIZE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int H = height, W = width, H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2;
  const int64_t F = n_frames, p1 = (int64_t)H * W * F, p2 = (int64_t)H2 * W2 * F, p4 = (int64_t)H4 * W4 * F;
  const UnetSaved s = unet_saved(saved, p1, p2, p4);
  int rc;
  hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)((H + 15) / 16), (unsigned)F), dim3(256), 0, st, x,
                     packed + unet_w_off(0), packed + unet_b_off(0), s.a0, H, W);
  if ((rc = launch_conv(s.a0, 64, nullptr, 0, packed, 1, s.x1, nullptr, H, W, F, st, s.p1, nullptr, packed16))) return rc;
  if ((rc = launch_conv(s.p1, 64, nullptr, 0, packed, 2, s.a2, nullptr, H2, W2, F, st, nullptr, nullptr, packed16))) return rc;
  if ((rc = launch_conv(s.a2, 128, nullptr, 0, packed, 3, s.x2, nullptr, H2, W2, F, st, s.p2, nullptr, packed16))) return rc;
  if ((rc = launch_conv(s.p2, 128, nullptr, 0, packed, 4, s.a4, nullptr, H4, W4, F, st, nullptr, nullptr, packed16))) return rc;
  if ((rc = launch_conv(s.a4, 128, nullptr, 0, packed, 5, s.x3, nullptr, H4, W4, F, st, nullptr, nullptr, packed16))) return rc;
  const UpWin w21 = UpWin{(full_h / 2) / 2, (full_w / 2) / 2, full_h / 2, full_w / 2, origin_y / 4, origin_x / 4, origin_y / 2, origin_x / 2};
  const UpWin w10 = UpWin{full_h / 2, full_w / 2, full_h, full_w, origin_y / 2, origin_x / 2, origin_y, origin_x};
  hipLaunchKernelGGL(upsample2_kernel, quad_grid(W2, 128, H2, F), dim3(256), 0, st, s.x3, s.u3, H4, W4, 128, H2, W2,
                     w21);
  if ((rc = launch_conv(s.x2, 128, s.u3, 128, packed, 6, s.a6, nullptr, H2, W2, F, st, nullptr, nullptr, packed16))) return rc;
  if ((rc = launch_conv(s.a6, 128, nullptr, 0, packed, 7, s.u1, nullptr, H2, W2, F, st, nullptr, nullptr, packed16))) return rc;
  hipLaunchKernelGGL(upsample2_kernel, quad_grid(W, 64, H, F), dim3(256), 0, st, s.u1, s.uu, H2, W2, 64, H, W,
                     w10);
  if ((rc = launch_conv(s.x1, 64, s.uu, 64, packed, 8, s.a8, nullptr, H, W, F, st, nullptr, nullptr, packed16))) return rc;
  if ((rc = launch_conv(s.a8, 64, nullptr, 0, packed, 9, nullptr, out, H, W, F, st, nullptr, s.y9, packed16))) return rc;
  return (int)hipGetLastError();
}

extern "C" int s2l_unet_forward_saved(const float* packed, const float* x, float* saved, float* out, int height, int width,
                                      int64_t n_frames, s2l_stream_t stream) {
  return s2l_unet_forward_saved_window(packed, nullptr, x, saved, out, height, width, height, width, 0, 0, n_frames, stream);
}

// d_out [F,H,W,3] -> d_x [F,H,W,3], from the activations s2l_unet_forward_saved kept; work: s2l_unet_backward_work_floats.
extern "C" int s2l_unet_backward_window(const float* packed, const uint16_t* packed16, const float* saved, const float* d_out,
                                        float* work, float* d_x, int height, int width, int full_h, int full_w, int origin_y,
                                        int origin_x, int64_t n_frames, s2l_stream_t stream) {
  if (height < 4 || width < 4 || n_frames < 0) return S2L_E_SIZE;
  { const int rcw = unet_window(height, width, full_h, full_w, origin_y, origin_x); if (rcw) return rcw; }
  if (n_frames == 0) return S2L_OK;
  if (!packed || !saved || !d_out || !work || !d_x) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(saved) || misaligned16(work)) return S2L_E_ALIGN;
  if (n_frames > 65535) return S2L_E_SIZE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int H = height, W = width, H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2;
  const int64_t F = n_frames, p1 = (int64_t)H * W * F, p2 = (int64_t)H2 * W2 * F, p4 = (int64_t)H4 * W4 * F;
  const UnetSaved s = unet_saved(const_cast<float*>(saved), p1, p2, p4);
  float* zA = work;              float* zB = zA + p1 * 64;      float* gcat8 = zB + p1 * 64;                      // @H
  float* z7 = gcat8 + p1 * 128;  float* z6 = z7 + p2 * 64;      float* gcat6 = z6 + p2 * 128;                     // @H/2
  float* z3 = gcat6 + p2 * 256;  float* z2 = z3 + p2 * 128;     float* gp1 = z2 + p2 * 128;
  float* z5 = gp1 + p2 * 64;     float* z4 = z5 + p4 * 128;     float* gp2 = z4 + p4 * 128;                       // @H/4
  auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  int rc;
  hipLaunchKernelGGL(outc_bwd_kernel, blocks(p1 * 16), dim3(256), 0, st, d_out, packed + kUnetOutW, s.y9, zA, p1 * 16);   // z9
  if ((rc = launch_conv_dgrad(zA, packed, 9, zB, s.a8, H, W, F, st, packed16))) return rc;                                          // z8
  if ((rc = launch_conv_dgrad(zB, packed, 8, gcat8, nullptr, H, W, F, st, packed16))) return rc;                                    // [g_x1 | g_uu]
  const UpWin w21 = UpWin{(full_h / 2) / 2, (full_w / 2) / 2, full_h / 2, full_w / 2, origin_y / 4, origin_x / 4, origin_y / 2, origin_x / 2};
  const UpWin w10 = UpWin{full_h / 2, full_w / 2, full_h, full_w, origin_y / 2, origin_x / 2, origin_y, origin_x};
  hipLaunchKernelGGL(upsample2_bwd_kernel, quad_grid(W2, 64, H2, F), dim3(256), 0, st, gcat8, 128, 64, s.u1, z7, H2, W2, 64,
                     H, W, w10);
  if ((rc = launch_conv_dgrad(z7, packed, 7, z6, s.a6, H2, W2, F, st, packed16))) return rc;
  if ((rc = launch_conv_dgrad(z6, packed, 6, gcat6, nullptr, H2, W2, F, st, packed16))) return rc;                                  // [g_x2 | g_u3]
  hipLaunchKernelGGL(upsample2_bwd_kernel, quad_grid(W4, 128, H4, F), dim3(256), 0, st, gcat6, 256, 128, s.x3, z5, H4, W4, 128,
                     H2, W2, w21);
  if ((rc = launch_conv_dgrad(z5, packed, 5, z4, s.a4, H4, W4, F, st, packed16))) return rc;
  if ((rc = launch_conv_dgrad(z4, packed, 4, gp2, nullptr, H4, W4, F, st, packed16))) return rc;
  hipLaunchKernelGGL(pool_bwd_add_kernel, quad_grid(W2, 128, H2, F), dim3(256), 0, st, gcat6, 256, gp2, s.x2, s.p2, z3,
                     H2, W2, 128);
  if ((rc = launch_conv_dgrad(z3, packed, 3, z2, s.a2, H2, W2, F, st, packed16))) return rc;
  if ((rc = launch_conv_dgrad(z2, packed, 2, gp1, nullptr, H2, W2, F, st, packed16))) return rc;
  hipLaunchKernelGGL(pool_bwd_add_kernel, quad_grid(W, 64, H, F), dim3(256), 0, st, gcat8, 128, gp1, s.x1, s.p1, zA,
                     H, W, 64);   // z1
  if ((rc = launch_conv_dgrad(zA, packed, 1, zB, s.a0, H, W, F, st, packed16))) return rc;                                          // z0
  launch_conv_first_bwd(zB, packed + unet_w_off(0), gcat8, d_x, H, W, F, st);      // (gcat8 is dead: >= 2048 floats of scratch)
  return (int)hipGetLastError();
}

extern "C" int s2l_unet_backward(const float* packed, const float* saved, const float* d_out, float* work, float* d_x,
                                 int height, int width, int64_t n_frames, s2l_stream_t stream) {
  return s2l_unet_backward_window(packed, nullptr, saved, d_out, work, d_x, height, width, height, width, 0, 0, n_frames, stream);
}

// ---- TRAIN mode entry points ------------------------------------------------------------------------------------------------
static int unet_table(const float* const* th, UnetTensors& t) {
  if (!th) return S2L_E_NULL;
  for (int l = 0; l < 10; ++l) {
    for (int k = 0; k < 5; ++k)
      if (!th[l * 5 + k]) return S2L_E_NULL;
    t.w[l] = th[l * 5]; t.gamma[l] = th[l * 5 + 1]; t.beta[l] = th[l * 5 + 2]; t.mean[l] = th[l * 5 + 3]; t.var[l] = th[l * 5 + 4];
  }
  if (!th[50] || !th[51]) return S2L_E_NULL;
  t.outw = th[50]; t.outb = th[51];
  return 0;
}

extern "C" int64_t s2l_unet_packed16_halves(void) { return kUnetPacked16Halves; }

// bf16 operand blob of the nine 3x3 layers (forward and input-gradient form) for the `packed16` argument of the window entry
// points; same tensor table and BatchNorm fold as s2l_unet_pack
extern "C" int s2l_unet_pack16(const float* const* tensors_host, float bn_eps, uint16_t* packed16, s2l_stream_t stream) {
  if (!packed16) return S2L_E_NULL;
  if (misaligned16(packed16)) return S2L_E_ALIGN;
  UnetTensors t;
  const int rc = unet_table(tensors_host, t);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int64_t n_max = 2 * 8 * kChunk16Halves;      // (256 -> 128; the count is the same both ways)
  hipLaunchKernelGGL(unet_pack_conv16, dim3((unsigned)((n_max + 255) / 256), 9, 2), dim3(256), 0, st, t, packed16, bn_eps);
  return (int)hipGetLastError();
}


extern "C" int64_t s2l_unet_packed16x3_halves(void) { return kUnetPacked16x3Halves; }
extern "C" int s2l_unet_pack16x3(const float* const* tensors_host, float bn_eps, uint16_t* packed16x3, s2l_stream_t stream) {
  if (!packed16x3) return S2L_E_NULL;
  if (misaligned16(packed16x3)) return S2L_E_ALIGN;
  UnetTensors t;
  const int rc = unet_table(tensors_host, t);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int64_t n_max = 2 * 16 * kChunk16Halves;
  hipLaunchKernelGGL(unet_pack_conv16x3, dim3((unsigned)((n_max + 255) / 256), 9), dim3(256), 0, st, t, packed16x3, bn_eps);
  return (int)hipGetLastError();
}

// RAW (un-folded) weights in the chunk layout, forward and transposed: what the train-mode network multiplies with.  Same table
// and blob size as s2l_unet_pack; re-run after every optimizer step.
extern "C" int s2l_unet_pack_raw(const float* const* tensors_host, float* packed, s2l_stream_t stream) {
  if (!packed) return S2L_E_NULL;
  UnetTensors t;
  const int rc = unet_table(tensors_host, t);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(unet_pack_convs_kernel, dim3((unsigned)((kPackMaxFloats + 255) / 256), 9, 2), dim3(256), 0, st, t, packed, -1.f);
  hipLaunchKernelGGL(unet_pack_misc, dim3(16), dim3(256), 0, st, t, packed, -1.f);
  return (int)hipGetLastError();
}

extern "C" int64_t s2l_unet_train_saved_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 0) return 0;
  return train_saved_floats((int64_t)height * width * n_frames, (int64_t)(height / 2) * (width / 2) * n_frames,
                            (int64_t)(height / 4) * (width / 4) * n_frames);
}
extern "C" int64_t s2l_unet_train_work_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 0) return 0;
  const int64_t p1 = (int64_t)height * width, p2 = (int64_t)(height / 2) * (width / 2), p4 = (int64_t)(height / 4) * (width / 4);
  // gradient scratch of the eval-mode backward | split-K partials of the weight gradients (<= 32 x 256*128*9) | reduction partials
  return n_frames * (p1 * 256 + p2 * 768 + p4 * 384) + 32 * (int64_t)256 * 128 * 9 + kStatBlocks * 2 * 128 + 4096;
}
extern "C" int64_t s2l_unet_grad_floats(void) { return grad_off(10) + 192 + 3; }
// the per-frame-statistics pair (s2l_unet_train_forward_frames / _backward_frames)
extern "C" int64_t s2l_unet_train_frames_saved_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 1) return 0;
  return train_saved_floats((int64_t)height * width * n_frames, (int64_t)(height / 2) * (width / 2) * n_frames,
                            (int64_t)(height / 4) * (width / 4) * n_frames, n_frames);
}
extern "C" int64_t s2l_unet_train_frames_scratch_floats(int64_t n_frames) { return n_frames < 1 ? 0 : n_frames * kStatBlocks * 2 * 128; }
extern "C" int64_t s2l_unet_train_frames_work_floats(int height, int width, int64_t n_frames) {
  if (height < 4 || width < 4 || n_frames < 1) return 0;
  return s2l_unet_train_work_floats(height, width, n_frames) + (n_frames - 1) * ((int64_t)kStatBlocks * 2 * 128 + 512);
}

static void run_stats(const float* z, int C, int64_t n_pix, float* partial, hipStream_t st, int* n_blocks, int groups = 1) {
  const int64_t per = (n_pix + kStatBlocks - 1) / kStatBlocks;      // n_pix: pixels of one group
  *n_blocks = (int)((n_pix + per - 1) / per);
  hipLaunchKernelGGL(channel_stats_kernel, dim3(*n_blocks, groups), dim3(256), 0, st, z, C, n_pix, per, partial);
}

// x [F,H,W,3] -> out [F,H,W,3] with batch statistics; running_mean / running_var of the ten BatchNorm layers (entries 3 and 4 of
// each layer's five pointers in `tensors_host`, the s2l_unet_pack table) are UPDATED IN PLACE when update_running != 0.
// saved: s2l_unet_train_saved_floats floats (kept for s2l_unet_train_backward); scratch: at least 256*2*128 floats.
// packed16_raw != NULL: the 3x3 layers 1..9 take bf16 operands (s2l_unet_pack16 with bn_eps < 0: the raw weights) on
// v_mfma_f32_32x32x16_bf16 -- fp32 accumulation, fp32 tensors, fp32 statistics -- as the eval-mode chain does in the bf16 step
// frames_are_groups: every frame is its own statistics group (F successive one-frame calls of the reference in one set of launches:
// scratch F x 262144 floats, saved with F statistics blocks per layer); else the batch is one group (nn.BatchNorm2d on [F,C,H,W]).
static int unet_train_forward_impl(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                   float bn_eps, float momentum, int update_running, const float* x, float* saved, float* scratch,
                                   float* out, int height, int width, int64_t n_frames, s2l_stream_t stream,
                                   bool frames_are_groups = false) {
  if (height < 4 || width < 4 || n_frames <= 0 || n_frames > 65535) return S2L_E_SIZE;
  if (!packed_raw || !x || !saved || !scratch || !out) return S2L_E_NULL;
  if (misaligned16(packed_raw) || misaligned16(saved) || misaligned16(packed16_raw)) return S2L_E_ALIGN;
  UnetTensors t;
  int rc = unet_table(tensors_host, t);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int H = height, W = width, H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2;
  const int64_t F = n_frames, p1 = (int64_t)H * W * F, p2 = (int64_t)H2 * W2 * F, p4 = (int64_t)H4 * W4 * F;
  const int64_t pl[3] = {p1, p2, p4};
  const int groups = frames_are_groups ? (int)F : 1;
  const int hh[3] = {H, H2, H4}, ww[3] = {W, W2, W4};
  const TrainBufs b = train_bufs(saved, p1, p2, p4);
  auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  // input of each convolution: (A, CA, B, CB)
  const float* inA[10] = {x, b.act[0], b.p1, b.act[2], b.p2, b.act[4], b.act[3], b.act[6], b.act[1], b.act[8]};
  const float* inB[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, b.u3, nullptr, b.uu, nullptr};
  const int cA[10] = {3, 64, 64, 128, 128, 128, 128, 128, 64, 64}, cB[10] = {0, 0, 0, 0, 0, 0, 128, 0, 64, 0};
  for (int l = 0; l < 10; ++l) {
    const int lv = kLvl[l], C = kUnetConvs[l].cout;
    if (l == 0) {
      hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)((H + 15) / 16), (unsigned)F), dim3(256), 0, st, x,
                         packed_raw + unet_w_off(0), (const float*)nullptr, b.z[0], H, W);
    } else {
      ConvArgs a;
      a.w16 = packed16_raw ? packed16_raw + unet_w16_off(l) : nullptr;
      a.split = 0;
      a.inA = inA[l]; a.inB = inB[l]; a.CA = cA[l]; a.CB = cB[l]; a.cout = C;
      a.w = packed_raw + unet_w_off(l); a.bias = nullptr; a.out = b.z[l]; a.out3 = nullptr; a.pool = nullptr; a.gate = nullptr; a.relu = 0;
      a.outw = a.outb = nullptr; a.H = hh[lv]; a.W = ww[lv];
      a.tiles_x = (a.W + 15) / 16; a.tiles_y = (a.H + 15) / 16; a.n_ct = C / 64;
      if (F * a.n_ct > 65535) return S2L_E_SIZE;
      const dim3 grid(a.tiles_x, a.tiles_y, (unsigned)(F * a.n_ct));
      bool done = false;
      if ((rc = launch_conv_asm(a, F, st, &done))) return rc;       // (declines bf16 operands)
      if (done) {}
      else if (a.w16) {
        bool pl = false;
        if ((rc = launch_conv_persistent(a, F, false, false, st, &pl))) return rc;
        if (!pl) hipLaunchKernelGGL(conv3x3_bf16_kernel<false>, grid, dim3(256), 0, st, a);
      }
      else hipLaunchKernelGGL(conv3x3_kernel<false>, grid, dim3(256), 0, st, a);
    }
    int nb = 0;
    const int64_t gpix = pl[lv] / groups;      // pixels per statistics group
    run_stats(b.z[l], C, gpix, scratch, st, &nb, groups);
    float* stl = b.st + (int64_t)l * 512 * groups;
    if (groups == 1) {
      hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, st, scratch, nb, C, (double)gpix, bn_eps, momentum, t.gamma[l],
                         t.beta[l], update_running ? const_cast<float*>(t.mean[l]) : nullptr,
                         update_running ? const_cast<float*>(t.var[l]) : nullptr, stl, groups);
    } else {
      hipLaunchKernelGGL(bn_finalize_groups_kernel, dim3(C, groups), dim3(64), 0, st, scratch, nb, C, (double)gpix, bn_eps, t.gamma[l],
                         t.beta[l], stl);
      if (update_running)
        hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(128), 0, st, scratch, nb, C, groups, momentum,
                           const_cast<float*>(t.mean[l]), const_cast<float*>(t.var[l]));
    }
    hipLaunchKernelGGL(bn_relu_kernel, blocks(pl[lv] * C / 4), dim3(256), 0, st, b.z[l], stl, b.act[l], C, pl[lv] * C / 4,
                       gpix * C / 4);
    if (l == 1) hipLaunchKernelGGL(maxpool2_kernel, blocks(p2 * 16), dim3(256), 0, st, b.act[1], b.p1, H, W, 64, p2 * 16);
    if (l == 3) hipLaunchKernelGGL(maxpool2_kernel, blocks(p4 * 32), dim3(256), 0, st, b.act[3], b.p2, H2, W2, 128, p4 * 32);
    if (l == 5)
      hipLaunchKernelGGL(upsample2_kernel, quad_grid(W2, 128, H2, F), dim3(256), 0, st, b.act[5], b.u3, H4, W4, 128, H2, W2,
                     no_window(H4, W4, H2, W2));
    if (l == 7)
      hipLaunchKernelGGL(upsample2_kernel, quad_grid(W, 64, H, F), dim3(256), 0, st, b.act[7], b.uu, H2, W2, 64, H, W,
                     no_window(H2, W2, H, W));
  }
  hipLaunchKernelGGL(outc_kernel, blocks(p1), dim3(256), 0, st, b.act[9], t.outw, t.outb, out, p1);
  return (int)hipGetLastError();
}
extern "C" int s2l_unet_train_forward(const float* packed_raw, const float* const* tensors_host, float bn_eps, float momentum,
                                      int update_running, const float* x, float* saved, float* scratch, float* out, int height,
                                      int width, int64_t n_frames, s2l_stream_t stream) {
  return unet_train_forward_impl(packed_raw, nullptr, tensors_host, bn_eps, momentum, update_running, x, saved, scratch, out, height,
                                 width, n_frames, stream);
}
// F frames = F successive one-frame train-mode calls of the reference (tf_nerf.py:387 inside train_stage1's batch-1 calls) in one set
// of launches: every frame is normalised with its own statistics, the running statistics move once per frame IN FRAME ORDER.  Same
// arithmetic per frame as F calls with n_frames = 1: the same bits.  packed16_raw: NULL = exact fp32 convolutions.
extern "C" int s2l_unet_train_forward_frames(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                             float bn_eps, float momentum, int update_running, const float* x, float* saved,
                                             float* scratch, float* out, int height, int width, int64_t n_frames, s2l_stream_t stream) {
  return unet_train_forward_impl(packed_raw, packed16_raw, tensors_host, bn_eps, momentum, update_running, x, saved, scratch, out,
                                 height, width, n_frames, stream, true);
}
extern "C" int s2l_unet_train_forward_bf16(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                           float bn_eps, float momentum, int update_running, const float* x, float* saved,
                                           float* scratch, float* out, int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (!packed16_raw) return S2L_E_NULL;
  return unet_train_forward_impl(packed_raw, packed16_raw, tensors_host, bn_eps, momentum, update_running, x, saved, scratch, out,
                                 height, width, n_frames, stream);
}

// d_out [F,H,W,3] -> d_x [F,H,W,3] (may be NULL) and grads (s2l_unet_grad_floats floats: per layer conv.weight [cout,cin,3,3],
// bn.weight [cout], bn.bias [cout] in execution order, then outc.conv.weight [3,64], outc.conv.bias [3]).
static int unet_train_backward_impl(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                    const float* x, const float* saved, const float* d_out, float* work, float* d_x, float* grads,
                                    int height, int width, int64_t n_frames, s2l_stream_t stream, bool frames_are_groups = false) {
  if (height < 4 || width < 4 || n_frames <= 0 || n_frames > 65535) return S2L_E_SIZE;
  if (misaligned16(packed16_raw)) return S2L_E_ALIGN;
  if (!packed_raw || !x || !saved || !d_out || !work) return S2L_E_NULL;
  if (!grads && !d_x) return S2L_E_NULL;      // nothing to compute
  if (misaligned16(packed_raw) || misaligned16(saved) || misaligned16(work)) return S2L_E_ALIGN;
  UnetTensors t;
  int rc = unet_table(tensors_host, t);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int H = height, W = width, H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2;
  const int64_t F = n_frames, p1 = (int64_t)H * W * F, p2 = (int64_t)H2 * W2 * F, p4 = (int64_t)H4 * W4 * F;
  const int64_t pl[3] = {p1, p2, p4};
  // grads == NULL: a FROZEN net in train-mode BatchNorm (the reference's loop after it > 100000: Trainer.train_step's model.train()
  // undoes the .eval() of train.py:195): only the input gradient is wanted -- the weight-gradient GEMMs (a third of the pass) are
  // skipped; the BatchNorm backward still needs its two per-channel sums, which land in scratch behind the reduction partials
  const bool want_params = grads != nullptr;
  const int hh[3] = {H, H2, H4}, ww[3] = {W, W2, W4};
  const TrainBufs b = train_bufs(const_cast<float*>(saved), p1, p2, p4);
  float* zA = work;              float* zB = zA + p1 * 64;      float* gcat8 = zB + p1 * 64;                      // @H
  float* z7 = gcat8 + p1 * 128;  float* z6 = z7 + p2 * 64;      float* gcat6 = z6 + p2 * 128;                     // @H/2
  float* z3 = gcat6 + p2 * 256;  float* z2 = z3 + p2 * 128;     float* gp1 = z2 + p2 * 128;
  float* z5 = gp1 + p2 * 64;     float* z4 = z5 + p4 * 128;     float* gp2 = z4 + p4 * 128;                       // @H/4
  float* wpart = gp2 + p4 * 128;                                 // split-K partials of the weight gradients
  const int groups = frames_are_groups ? (int)F : 1;
  float* rpart = wpart + 32 * (int64_t)256 * 128 * 9;            // reduction partials + per-channel means, per statistics group
  float* sums = rpart + (int64_t)kStatBlocks * 2 * 128 * groups;
  auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  const float* inA[10] = {x, b.act[0], b.p1, b.act[2], b.p2, b.act[4], b.act[3], b.act[6], b.act[1], b.act[8]};
  const float* inB[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, b.u3, nullptr, b.uu, nullptr};
  const int cA[10] = {3, 64, 64, 128, 128, 128, 128, 128, 64, 64}, cB[10] = {0, 0, 0, 0, 0, 0, 128, 0, 64, 0};

  // BatchNorm backward (in place: gy -> dz) + the layer's weight gradient
  auto layer_grads = [&](int l, float* gy) {
    const int lv = kLvl[l], C = kUnetConvs[l].cout, cin = kUnetConvs[l].cin;
    const int64_t n = pl[lv] / groups;      // pixels per statistics group
    const int64_t per = (n + kStatBlocks - 1) / kStatBlocks;
    const int nb = (int)((n + per - 1) / per);
    const float* stl = b.st + (int64_t)l * 512 * groups;
    float* g = want_params ? grads + grad_off(l) : nullptr;
    // (sums: per group 2 x 128 means; frozen net: dgamma / dbeta go to 256 floats of scratch per group behind them)
    // per-frame groups: every group leaves its own dgamma / dbeta in scratch; a training net's are their sums over the groups
    const bool via_scratch = !want_params || groups > 1;
    float* dgamma = via_scratch ? sums + 256 * groups : g + (int64_t)C * cin * 9;
    float* dbeta = via_scratch ? dgamma + 128 : dgamma + C;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb, groups), dim3(256), 0, st, gy, b.z[l], stl, C, n, per, rpart);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C, groups), dim3(64), 0, st, rpart, nb, C, (double)n, dgamma, dbeta, sums, (const float*)nullptr);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, blocks(pl[lv] * C / 4), dim3(256), 0, st, gy, b.z[l], stl, sums, C, pl[lv] * C / 4,
                       n * C / 4);
    if (!want_params) return;
    if (groups > 1)
      hipLaunchKernelGGL(sum_groups_kernel, dim3(1), dim3(128), 0, st, dgamma, g + (int64_t)C * cin * 9, g + (int64_t)C * cin * 9 + C, C, groups);
    if (l == 0) {
      const int64_t perw = (p1 + 1023) / 1024;
      const int nbw = (int)((p1 + perw - 1) / perw);
      hipLaunchKernelGGL(conv_first_wgrad_kernel, dim3(nbw), dim3(256), 0, st, gy, x, wpart, H, W, p1, perw);
      hipLaunchKernelGGL(wgrad_reduce_lanes_kernel<16>, dim3(27), dim3(1024), 0, st, wpart, g, nbw, (int64_t)64 * 27);
    } else {
      WgradArgs a;
      a.dz = gy; a.inA = inA[l]; a.inB = inB[l]; a.CA = cA[l]; a.CB = cB[l]; a.cout = C;
      a.H = hh[lv]; a.W = ww[lv]; a.F = (int)F; a.chunks_x = (a.W + 63) / 64;
      a.n_chunks = F * a.H * a.chunks_x;
      a.partial = wpart;
      const bool w16 = packed16_raw != nullptr;      // bf16 precision: bf16 operands for the weight gradient too (32 input channels per workgroup)
      const int tiles = (C / 64) * (cin / (w16 ? 32 : 16));
      // split K so that ~2048 workgroups exist (8 per CU), as far as the partial buffer reaches: it holds 32 partials of the
      // largest layer (256 x 128), i.e. 256 of a 64 x 64 one -- with a flat cap of 32 the two 64 -> 64 layers at full
      // resolution ran on 128 workgroups (0.30 of the MFMA peak; 0.6 with this)
      int S = (int)((2048 + tiles - 1) / tiles);
      const int cap = (int)(32 * (int64_t)256 * 128 / ((int64_t)C * cin));
      if (S > cap) S = cap;
      if (S > a.n_chunks) S = (int)a.n_chunks;
      if (w16) hipLaunchKernelGGL(conv_wgrad_bf16_kernel, dim3(C / 64, cin / 32, S), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(conv_wgrad_kernel, dim3(C / 64, cin / 16, S), dim3(256), 0, st, a);
      const int64_t ne = (int64_t)C * cin * 9;
      hipLaunchKernelGGL(wgrad_reduce_lanes_kernel<4>, dim3((unsigned)((ne + 63) / 64)), dim3(256), 0, st, wpart, g, S, ne);
    }
  };

  // output convolution: gy9 = outc^T d_out * (a9 > 0); d outc.weight / bias
  if (want_params) {
    const int64_t perw = (p1 + 2047) / 2048;
    const int nbw = (int)((p1 + perw - 1) / perw);
    hipLaunchKernelGGL(outc_wgrad_kernel, dim3(nbw), dim3(256), 0, st, d_out, b.act[9], wpart, p1, perw);
    hipLaunchKernelGGL(wgrad_reduce_lanes_kernel<16>, dim3(4), dim3(1024), 0, st, wpart, grads + grad_off(10), nbw, (int64_t)195);
  }
  hipLaunchKernelGGL(outc_bwd_kernel, blocks(p1 * 16), dim3(256), 0, st, d_out, t.outw, b.act[9], zA, p1 * 16);
  layer_grads(9, zA);
  if ((rc = launch_conv_dgrad(zA, packed_raw, 9, zB, b.act[8], H, W, F, st, packed16_raw))) return rc;
  layer_grads(8, zB);
  if ((rc = launch_conv_dgrad(zB, packed_raw, 8, gcat8, nullptr, H, W, F, st, packed16_raw))) return rc;                                // [g_x1 | g_uu]
  hipLaunchKernelGGL(upsample2_bwd_kernel, quad_grid(W2, 64, H2, F), dim3(256), 0, st, gcat8, 128, 64, b.act[7], z7, H2, W2, 64,
                     H, W, no_window(H2, W2, H, W));
  layer_grads(7, z7);
  if ((rc = launch_conv_dgrad(z7, packed_raw, 7, z6, b.act[6], H2, W2, F, st, packed16_raw))) return rc;
  layer_grads(6, z6);
  if ((rc = launch_conv_dgrad(z6, packed_raw, 6, gcat6, nullptr, H2, W2, F, st, packed16_raw))) return rc;                              // [g_x2 | g_u3]
  hipLaunchKernelGGL(upsample2_bwd_kernel, quad_grid(W4, 128, H4, F), dim3(256), 0, st, gcat6, 256, 128, b.act[5], z5, H4, W4, 128,
                     H2, W2, no_window(H4, W4, H2, W2));
  layer_grads(5, z5);
  if ((rc = launch_conv_dgrad(z5, packed_raw, 5, z4, b.act[4], H4, W4, F, st, packed16_raw))) return rc;
  layer_grads(4, z4);
  if ((rc = launch_conv_dgrad(z4, packed_raw, 4, gp2, nullptr, H4, W4, F, st, packed16_raw))) return rc;
  hipLaunchKernelGGL(pool_bwd_add_kernel, quad_grid(W2, 128, H2, F), dim3(256), 0, st, gcat6, 256, gp2, b.act[3], b.p2, z3,
                     H2, W2, 128);
  layer_grads(3, z3);
  if ((rc = launch_conv_dgrad(z3, packed_raw, 3, z2, b.act[2], H2, W2, F, st, packed16_raw))) return rc;
  layer_grads(2, z2);
  if ((rc = launch_conv_dgrad(z2, packed_raw, 2, gp1, nullptr, H2, W2, F, st, packed16_raw))) return rc;
  hipLaunchKernelGGL(pool_bwd_add_kernel, quad_grid(W, 64, H, F), dim3(256), 0, st, gcat8, 128, gp1, b.act[1], b.p1, zA,
                     H, W, 64);
  layer_grads(1, zA);
  if ((rc = launch_conv_dgrad(zA, packed_raw, 1, zB, b.act[0], H, W, F, st, packed16_raw))) return rc;
  layer_grads(0, zB);
  if (d_x) launch_conv_first_bwd(zB, packed_raw + unet_w_off(0), gcat8, d_x, H, W, F, st);      // (gcat8 is dead by now)
  return (int)hipGetLastError();
}
extern "C" int s2l_unet_train_backward(const float* packed_raw, const float* const* tensors_host, const float* x, const float* saved,
                                       const float* d_out, float* work, float* d_x, float* grads, int height, int width,
                                       int64_t n_frames, s2l_stream_t stream) {
  return unet_train_backward_impl(packed_raw, nullptr, tensors_host, x, saved, d_out, work, d_x, grads, height, width, n_frames, stream);
}
// the input gradients of layers 1..9 with bf16 operands (the transposed half of the same s2l_unet_pack16(bn_eps < 0) blob);
// BatchNorm backward, weight gradients and the first layer stay fp32
// input gradient of s2l_unet_train_forward_frames (a frozen net: no parameter gradients), per-frame statistics terms
extern "C" int s2l_unet_train_backward_frames(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                              const float* x, const float* saved, const float* d_out, float* work, float* d_x,
                                              int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (!d_x) return S2L_E_NULL;
  return unet_train_backward_impl(packed_raw, packed16_raw, tensors_host, x, saved, d_out, work, d_x, nullptr, height, width, n_frames,
                                  stream, true);
}
// the same for a net that still TRAINS (it <= 100000): also the parameter gradients of the F one-frame calls, summed (grads as
// s2l_unet_train_backward; d_x may be NULL)
extern "C" int s2l_unet_train_backward_frames_grads(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                                    const float* x, const float* saved, const float* d_out, float* work, float* d_x,
                                                    float* grads, int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (!grads) return S2L_E_NULL;
  return unet_train_backward_impl(packed_raw, packed16_raw, tensors_host, x, saved, d_out, work, d_x, grads, height, width, n_frames, stream,
                                  true);
}
extern "C" int s2l_unet_train_backward_bf16(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                            const float* x, const float* saved, const float* d_out, float* work, float* d_x,
                                            float* grads, int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (!packed16_raw) return S2L_E_NULL;
  return unet_train_backward_impl(packed_raw, packed16_raw, tensors_host, x, saved, d_out, work, d_x, grads, height, width, n_frames,
                                  stream);
}

#include "unet_half.inc"
