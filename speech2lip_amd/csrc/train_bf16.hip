// bf16 mode of the lip-MLP training step (BASELINE config 5 names bf16): forward with saved
// activations, backward dz chain and weight-gradient GEMMs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32
// master weights and fp32 gradients.  Same mathematics as rows.hip / train.hip (the fp32 parity mode); operands, saved
// activations and saved gradients are bf16, which makes every one of these kernels HBM-bound instead of MFMA-bound
// (9.7 GB of saved state per direction for 64 frames at 96x96 instead of 38 GB).
//
// Layouts: s2l_bf16.h.  A workgroup (4 waves, one per SIMD) owns 256 rows of the batch; a wave owns 64 (two groups of 32
// = two B operands per A read).  A layer's weights pass through LDS in four 2-block stages; the next stage's global loads
// are in flight in registers while the current one computes (one barrier per stage).  Activations stay in registers
// between layers (kfeat16 trick); per 32-feature block the epilogue adds bias, applies ReLU, records the ReLU bit masks
// (64-bit ballots, 32 B per row per layer) and converts to bf16 twice: into the next layer's B operands and, through a
// 4.5 KiB per-wave LDS transposer, into the [feature][64 rows] tiles the weight-gradient GEMM consumes, stored with
// full-line 1 KiB wave stores.
#include "s2l_common.h"
#include "s2l_bf16.h"

namespace s2l {
namespace b16 {

typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfp2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk2(float lo, float hi) {  // round-to-nearest-even pair
  bfp2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint16_t bf1(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ f16v mfma32(u4 a, u4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

struct Tab16 {
  const float* t[S2L_NUM_TENSORS];
};

// ---- packing ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hidden_w(const Tab16& tab, int layer, int out_f, int in_f) {  // pts_linears[layer], layer 1..7
  const float* w = tab.t[S2L_T_PTS0_W + 2 * layer];
  return layer == 5 ? w[(int64_t)out_f * 512 + 256 + in_f] : w[(int64_t)out_f * 256 + in_f];
}

__global__ void pack_bf16_kernel(Tab16 tab, const float* __restrict__ pf, uint16_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PACKED_HALVES) return;
  float v = 0.f;
  auto slab_idx = [](int e, int& t, int& lane, int& j) { t = e >> 9, lane = (e >> 3) & 63, j = e & 7; };
  int t, lane, j;
  if (i < OFF_BWD_U0) {
    const int s = (int)(i / kStageF), o = (int)(i % kStageF);
    const int L = s >> 2, q = s & 3;
    if (o < 2 * kSlabX) {
      if (s == 31) {  // output layer in H format
        slab_idx(o, t, lane, j);
        const int row = lane & 31;
        if (row < 3) v = tab.t[S2L_T_OUT_W][row * 256 + kfeat16(t, lane >> 5, j)];
      } else if (L == 0 || L == 5) {
        const int R = 2 * q + o / kSlabX;
        slab_idx(o % kSlabX, t, lane, j);
        v = pf[(L == 0 ? OFF_G0 : OFF_G5) + (int64_t)(32 * R + (lane & 31)) * kGenK + 16 * t + 8 * (lane >> 5) + j];
      }
    } else if (L != 0) {
      const int o2 = o - 2 * kSlabX;
      const int R = 2 * q + o2 / kSlabH;
      slab_idx(o2 % kSlabH, t, lane, j);
      v = hidden_w(tab, L, 32 * R + (lane & 31), kfeat16(t, lane >> 5, j));
    }
  } else if (i < OFF_BWD_H) {
    const int e = (int)(i - OFF_BWD_U0);
    const int R = e / kSlabU0;
    lane = (e >> 3) & 63, j = e & 7;
    const int k = 8 * (lane >> 5) + j;
    if (k < 3) v = tab.t[S2L_T_OUT_W][k * 256 + 32 * R + (lane & 31)];
  } else {
    const int e = (int)(i - OFF_BWD_H);
    const int u = e / kStageB, o = e % kStageB;
    slab_idx(o % kSlabH, t, lane, j);
    if (bwd_stage_is_audio(u)) {
      const int R = o / kSlabH;
      v = pf[(u == 8 ? OFF_G5 : OFF_G0) + (int64_t)kfeat16(t, lane >> 5, j) * kGenK + kEmb + 32 * R + (lane & 31)];
    } else {
      const int R = 2 * bwd_stage_quarter(u) + o / kSlabH;
      v = hidden_w(tab, bwd_stage_layer(u), kfeat16(t, lane >> 5, j), 32 * R + (lane & 31));
    }
  }
  dst[i] = bf1(v);
}

// ---- shared pieces of the forward / backward kernels ----------------------------------------------------------------------
constexpr int kTrStride = 144;                  // bytes per feature row of the transposer: 64 rows bf16 + 16 pad
constexpr int kTrBytes = 32 * kTrStride;        // per wave
constexpr int kLdsW = 2 * kStageF * 2;          // two stage buffers, bytes
constexpr int kLdsFwd = kLdsW + 4 * kTrBytes + (8 * 256 + 4) * 4;

struct Stage {
  u4 x[4], h[8];   // 16 B pieces of the X part (16 KiB) and the H part (32 KiB), piece = tid + 256 k
};
__device__ __forceinline__ void stage_gload(Stage& st, const uint16_t* src, int tid, bool ld_x, bool ld_h) {
  const u4* p = reinterpret_cast<const u4*>(src) + tid;
  if (ld_x) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st.x[k] = p[256 * k];
  }
  if (ld_h) {
#pragma unroll
    for (int k = 0; k < 8; ++k) st.h[k] = p[1024 + 256 * k];
  }
}
__device__ __forceinline__ void stage_lstore(const Stage& st, uint16_t* dst, int tid, bool ld_x, bool ld_h) {
  u4* p = reinterpret_cast<u4*>(dst) + tid;
  if (ld_x) {
#pragma unroll
    for (int k = 0; k < 4; ++k) p[256 * k] = st.x[k];
  }
  if (ld_h) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p[1024 + 256 * k] = st.h[k];
  }
}

// Write one finished 32-feature block (both row groups, already bf16 pairs) through the wave's transposer to the
// [feature][64 rows] tile: vals[g][a][0] = features (8a+4hh+0, +1), vals[g][a][1] = (+2, +3) of row 32g + n.
__device__ __forceinline__ void tile_store(const uint32_t (&vals)[2][4][2], char* tr, uint16_t* gdst, int lane) {
  const int n = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int fl = 8 * a + 4 * hh + 2 * d;
        uint16_t* p = reinterpret_cast<uint16_t*>(tr + fl * kTrStride) + 32 * g + n;
        p[0] = (uint16_t)(vals[g][a][d] & 0xffffu);
        p[kTrStride / 2] = (uint16_t)(vals[g][a][d] >> 16);
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  u4* g4 = reinterpret_cast<u4*>(gdst);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = lane + 64 * k;
    g4[p] = *reinterpret_cast<const u4*>(tr + (p >> 3) * kTrStride + (p & 7) * 16);
  }
  __builtin_amdgcn_wave_barrier();
}

struct FwdArgs {
  const uint16_t* wb;
  const float* pf;
  const float* x;
  uint16_t* hT;
  uint64_t* masks;
  float* rgb;
  int64_t n_rows, layer_stride, mask_layer_stride;
  int n_tiles;
};

__global__ __launch_bounds__(256, 1) void fwd_bf16_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* wbuf = reinterpret_cast<uint16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5;
  char* tr = smem + kLdsW + wave * kTrBytes;
  float* bias = reinterpret_cast<float*>(smem + kLdsW + 4 * kTrBytes);
  for (int i = tid; i < 8 * 256; i += 256) {
    const int L = i >> 8, f = i & 255;
    bias[i] = L == 0 ? a.pf[OFF_BG0 + f] : L == 5 ? a.pf[OFF_BG5 + f] : a.pf[OFF_BIAS + (L - 1) * 256 + f];
  }
  if (tid < 4) bias[2048 + tid] = a.pf[OFF_BOUT + tid];

  Stage st;
  stage_gload(st, a.wb + OFF_FWD, tid, true, false);
  stage_lstore(st, wbuf, tid, true, false);
  __syncthreads();

  u4 bx[2][8], bcur[2][16], bnext[2][16];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int t = 0; t < 16; ++t) bcur[g][t] = u4{0u, 0u, 0u, 0u};

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t w0 = (int64_t)tile * kWgRows + 64 * wave;
    const int64_t tile64 = (int64_t)tile * 4 + wave;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int64_t row = w0 + 32 * g + n;
      const bool ok = row < a.n_rows;
      const f4* xr = reinterpret_cast<const f4*>(a.x + (ok ? row : 0) * kGenK + 8 * hh);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f4 lo = xr[4 * t], hi = xr[4 * t + 1];
        if (!ok) lo = hi = f4{0.f, 0.f, 0.f, 0.f};
        bx[g][t] = u4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
      }
    }
    for (int L = 0; L < 8; ++L) {
      const bool use_x = L == 0 || L == 5, use_h = L != 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = 4 * L + q, nxt = (s + 1) & 31;
        const int nL = nxt >> 2;
        const bool nx = nL == 0 || nL == 5 || nxt == 31, nh = nL != 0;
        stage_gload(st, a.wb + OFF_FWD + (int64_t)nxt * kStageF, tid, nx, nh);
        const uint16_t* wl = wbuf + (q & 1) * kStageF;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const int R = 2 * q + which;
          f16v acc[2];
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
          if (use_x) {
            const u4* ax = reinterpret_cast<const u4*>(wl + which * kSlabX) + lane;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const u4 av = ax[64 * t];
              acc[0] = mfma32(av, bx[0][t], acc[0]);
              acc[1] = mfma32(av, bx[1][t], acc[1]);
            }
          }
          if (use_h) {
            const u4* ah = reinterpret_cast<const u4*>(wl + 2 * kSlabX + which * kSlabH) + lane;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              const u4 av = ah[64 * t];
              acc[0] = mfma32(av, bcur[0][t], acc[0]);
              acc[1] = mfma32(av, bcur[1][t], acc[1]);
            }
          }
          // epilogue: bias, ReLU, masks, bf16; features 32R + 8a + 4hh + c
          uint32_t vals[2][4][2];
          uint64_t mymask = 0;
          const float* bl = bias + L * 256 + 32 * R + 4 * hh;
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            const f4 bv = *reinterpret_cast<const f4*>(bl + 8 * a4);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              float v[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                v[c] = fmaxf(acc[g][4 * a4 + c] + bv[c], 0.f);
                const uint64_t b = __ballot(v[c] > 0.f);
                if (lane == g * 16 + 4 * a4 + c) mymask = b;
              }
              vals[g][a4][0] = pk2(v[0], v[1]);
              vals[g][a4][1] = pk2(v[2], v[3]);
            }
          }
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            bnext[g][2 * R] = u4{vals[g][0][0], vals[g][0][1], vals[g][1][0], vals[g][1][1]};
            bnext[g][2 * R + 1] = u4{vals[g][2][0], vals[g][2][1], vals[g][3][0], vals[g][3][1]};
          }
          tile_store(vals, tr, a.hT + L * a.layer_stride + tile64 * (256 * kTileRows) + (32 * R) * kTileRows, lane);
          if (lane < 32) a.masks[L * a.mask_layer_stride + tile64 * 256 + R * 32 + lane] = mymask;
        }
        if (s == 31) {  // output layer on h7 (= bnext), weights in the X part of this stage
          const u4* ao = reinterpret_cast<const u4*>(wl) + lane;
          f16v acc[2];
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const u4 av = ao[64 * t];
            acc[0] = mfma32(av, bnext[0][t], acc[0]);
            acc[1] = mfma32(av, bnext[1][t], acc[1]);
          }
          if (hh == 0) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int64_t row = w0 + 32 * g + n;
              if (row < a.n_rows) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.rgb[row * 3 + c] = acc[g][c] + bias[2048 + c];
              }
            }
          }
        }
        stage_lstore(st, wbuf + ((q + 1) & 1) * kStageF, tid, nx, nh);
        __syncthreads();
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 16; ++t) bcur[g][t] = bnext[g][t];
    }
  }
}

// ---- backward dz chain ------------------------------------------------------------------------------------------------------
// g_7 = (Wout^T drgb) . m_7;  g_{l-1} = (W_l^T g_l) . m_{l-1} for l = 7..1 (l = 5: the h_4 half of pts_linears[5]);
// d audio = G5[:, audio]^T g_5 + G0[:, audio]^T g_0.  Every g_l is stored as a [feature][64 rows] bf16 tile (dzT) for the
// weight-gradient GEMMs; the masks are the forward's ballots, read back as wave-uniform SGPR pairs (one v_cndmask per value).
constexpr int kLdsBwdW = 2 * kStageB * 2;
constexpr int kLdsBwd = kLdsBwdW + 8 * kSlabU0 * 2 + 4 * kTrBytes;

struct StageB {
  u4 h[8];
};

struct BwdArgs {
  const uint16_t* wb;
  const float* drgb;
  const uint64_t* masks;
  uint16_t* dzT;
  float* dxa;
  int64_t n_rows, layer_stride, mask_layer_stride;
  int n_tiles;
};

__device__ __forceinline__ float mask_sel(float v, uint64_t m) {
  float o;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(o) : "v"(v), "s"(m));
  return o;
}

// masked gradient block -> bf16 pairs; mrow = the 32 ballots of this (layer, 64-row tile, R): one coalesced 256-byte load,
// then each word is broadcast into an SGPR pair with v_readlane
__device__ __forceinline__ void mask_block(const f16v (&acc)[2], const uint64_t* __restrict__ mrow, int lane,
                                           uint32_t (&vals)[2][4][2]) {
  const uint64_t mv = mrow[lane & 31];
  const uint32_t mlo = (uint32_t)mv, mhi = (uint32_t)(mv >> 32);
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = g * 16 + 4 * a4 + c;
        const uint64_t m = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(mlo, idx) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, idx) << 32);
        v[c] = mask_sel(acc[g][4 * a4 + c], m);
      }
      vals[g][a4][0] = pk2(v[0], v[1]);
      vals[g][a4][1] = pk2(v[2], v[3]);
    }
}

__global__ __launch_bounds__(256, 1) void bwd_bf16_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* wbuf = reinterpret_cast<uint16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5;
  u4* u0 = reinterpret_cast<u4*>(smem + kLdsBwdW);
  char* tr = smem + kLdsBwdW + 8 * kSlabU0 * 2 + wave * kTrBytes;
  for (int i = tid; i < 8 * 64; i += 256) u0[i] = reinterpret_cast<const u4*>(a.wb + OFF_BWD_U0)[i];

  StageB st;
  auto gload = [&](int u) {
    const u4* p = reinterpret_cast<const u4*>(a.wb + OFF_BWD_H + (int64_t)u * kStageB) + tid;
#pragma unroll
    for (int k = 0; k < 8; ++k) st.h[k] = p[256 * k];
  };
  auto lstore = [&](int u) {
    u4* p = reinterpret_cast<u4*>(wbuf + (u & 1) * kStageB) + tid;
#pragma unroll
    for (int k = 0; k < 8; ++k) p[256 * k] = st.h[k];
  };
  gload(0);
  lstore(0);
  __syncthreads();

  u4 bcur[2][16], bnext[2][16];
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t w0 = (int64_t)tile * kWgRows + 64 * wave;
    const int64_t tile64 = (int64_t)tile * 4 + wave;
    // drgb as a K = 16 B operand: k = 8 hh + j, k < 3 used
    u4 b0[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int64_t row = w0 + 32 * g + n;
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
      if (hh == 0 && row < a.n_rows) d0 = a.drgb[row * 3], d1 = a.drgb[row * 3 + 1], d2 = a.drgb[row * 3 + 2];
      b0[g] = u4{pk2(d0, d1), pk2(d2, 0.f), 0u, 0u};
    }
    f16v acc_a[2][2];   // d audio: [audio block][row group]
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_a[R][g][r] = 0.f;

    // U0: g_7
#pragma unroll
    for (int R = 0; R < 8; ++R) {
      f16v acc[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        acc[g] = mfma32(u0[R * 64 + lane], b0[g], acc[g]);
      }
      uint32_t vals[2][4][2];
      mask_block(acc, a.masks + 7 * a.mask_layer_stride + tile64 * 256 + R * 32, lane, vals);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bcur[g][2 * R] = u4{vals[g][0][0], vals[g][0][1], vals[g][1][0], vals[g][1][1]};
        bcur[g][2 * R + 1] = u4{vals[g][2][0], vals[g][2][1], vals[g][3][0], vals[g][3][1]};
      }
      tile_store(vals, tr, a.dzT + 7 * a.layer_stride + tile64 * (256 * kTileRows) + (32 * R) * kTileRows, lane);
    }

    int u = 0;
    auto audio_stage = [&]() {   // one stage: two 32-dim blocks of G[:, audio]^T against bcur
      const int nxt = u + 1 == kBwdStages ? 0 : u + 1;
      gload(nxt);
      const uint16_t* wl = wbuf + (u & 1) * kStageB;
#pragma unroll
      for (int R = 0; R < 2; ++R) {
        const u4* ah = reinterpret_cast<const u4*>(wl + R * kSlabH) + lane;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const u4 av = ah[64 * t];
          acc_a[R][0] = mfma32(av, bcur[0][t], acc_a[R][0]);
          acc_a[R][1] = mfma32(av, bcur[1][t], acc_a[R][1]);
        }
      }
      lstore(nxt);
      __syncthreads();
      u = nxt;
    };

    for (int l = 7; l >= 1; --l) {   // W_l^T g_l -> g_{l-1}
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nxt = u + 1 == kBwdStages ? 0 : u + 1;
        gload(nxt);
        const uint16_t* wl = wbuf + (u & 1) * kStageB;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const int R = 2 * q + which;
          f16v acc[2];
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
          const u4* ah = reinterpret_cast<const u4*>(wl + which * kSlabH) + lane;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const u4 av = ah[64 * t];
            acc[0] = mfma32(av, bcur[0][t], acc[0]);
            acc[1] = mfma32(av, bcur[1][t], acc[1]);
          }
          uint32_t vals[2][4][2];
          mask_block(acc, a.masks + (l - 1) * a.mask_layer_stride + tile64 * 256 + R * 32, lane, vals);
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            bnext[g][2 * R] = u4{vals[g][0][0], vals[g][0][1], vals[g][1][0], vals[g][1][1]};
            bnext[g][2 * R + 1] = u4{vals[g][2][0], vals[g][2][1], vals[g][3][0], vals[g][3][1]};
          }
          tile_store(vals, tr, a.dzT + (l - 1) * a.layer_stride + tile64 * (256 * kTileRows) + (32 * R) * kTileRows, lane);
        }
        lstore(nxt);
        __syncthreads();
        u = nxt;
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 16; ++t) bcur[g][t] = bnext[g][t];
      if (l == 6) audio_stage();   // bcur = g_5
    }
    audio_stage();                 // bcur = g_0; u wraps to 0 for the next tile
    // d audio [row][64]: lane holds dims 32R + 8a + 4hh + c of row 32g + n
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int64_t row = w0 + 32 * g + n;
      if (row < a.n_rows) {
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4)
            *reinterpret_cast<f4*>(a.dxa + row * kAud + 32 * R + 8 * a4 + 4 * hh) =
                f4{acc_a[R][g][4 * a4], acc_a[R][g][4 * a4 + 1], acc_a[R][g][4 * a4 + 2], acc_a[R][g][4 * a4 + 3]};
      }
    }
  }
}

}  // namespace b16
}  // namespace s2l

using namespace s2l;
using namespace s2l::b16;

extern "C" int64_t s2l_bf16_packed_halves(void) { return PACKED_HALVES; }
extern "C" int64_t s2l_bf16_rows_padded(int64_t n_rows) { return (n_rows + kWgRows - 1) / kWgRows * kWgRows; }

extern "C" int s2l_pack_bf16(const float* const* tensors_host, const float* packed_f32, uint16_t* packed_bf16,
                             s2l_stream_t stream) {
  if (!tensors_host || !packed_f32 || !packed_bf16) return S2L_E_NULL;
  if (misaligned16(packed_bf16)) return S2L_E_ALIGN;
  Tab16 tab;
  for (int i = 0; i < S2L_NUM_TENSORS; ++i) {
    if (!tensors_host[i]) return S2L_E_NULL;
    tab.t[i] = tensors_host[i];
  }
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((PACKED_HALVES + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), tab, packed_f32, packed_bf16);
  return (int)hipGetLastError();
}

static int n_cu_of_device() {
  static int cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cache[dev]) {
    hipDeviceProp_t p;
    cache[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return cache[dev];
}

extern "C" int s2l_train_forward_bf16(const uint16_t* packed_bf16, const float* packed_f32, const float* x, uint16_t* hT,
                                      uint64_t* masks, float* rgb, int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed_bf16 || !packed_f32 || !x || !hT || !masks || !rgb) return S2L_E_NULL;
  if (misaligned16(packed_bf16) || misaligned16(x) || misaligned16(hT) || misaligned16(masks)) return S2L_E_ALIGN;
  static bool attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kLdsFwd);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  FwdArgs a;
  const int64_t np = s2l_bf16_rows_padded(n_rows);
  a.wb = packed_bf16, a.pf = packed_f32, a.x = x, a.hT = hT, a.masks = masks, a.rgb = rgb;
  a.n_rows = n_rows, a.layer_stride = np * 256, a.mask_layer_stride = np / 64 * 256;
  a.n_tiles = (int)(np / kWgRows);
  const int grid = a.n_tiles < n_cu_of_device() ? a.n_tiles : n_cu_of_device();
  hipLaunchKernelGGL(fwd_bf16_kernel, dim3(grid), dim3(256), kLdsFwd, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

extern "C" int s2l_train_backward_bf16(const uint16_t* packed_bf16, const float* drgb, const uint64_t* masks, uint16_t* dzT,
                                       float* dxa, int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed_bf16 || !drgb || !masks || !dzT || !dxa) return S2L_E_NULL;
  if (misaligned16(packed_bf16) || misaligned16(dzT) || misaligned16(dxa) || misaligned16(masks)) return S2L_E_ALIGN;
  static bool attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kLdsBwd);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  BwdArgs a;
  const int64_t np = s2l_bf16_rows_padded(n_rows);
  a.wb = packed_bf16, a.drgb = drgb, a.masks = masks, a.dzT = dzT, a.dxa = dxa;
  a.n_rows = n_rows, a.layer_stride = np * 256, a.mask_layer_stride = np / 64 * 256;
  a.n_tiles = (int)(np / kWgRows);
  const int grid = a.n_tiles < n_cu_of_device() ? a.n_tiles : n_cu_of_device();
  hipLaunchKernelGGL(bwd_bf16_kernel, dim3(grid), dim3(256), kLdsBwd, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
