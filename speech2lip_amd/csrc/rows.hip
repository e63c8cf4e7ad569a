// General-row MLP on the LDS-DMA weight ring: forward (optionally saving h0..h7) and the backward
// dz chain.  Same MFMA scheme and ring protocol as render.hip (read its header first); here every
// ring step is a tile-independent 16 KiB weight block, so the DMA program is a pure function of the
// step number (modulo the steps per tile) and crosses tile boundaries for free.
//
//   forward, 129 steps per tile of 128 rows, in consumption order (`fwd_step_src`):
//     [0,8)    G0 = W0 [Wuv|Wa|Wt|0]   K=128: two 8 KiB M-blocks per step
//     [8,72)   pts_linears 1..4
//     [72,80)  G5 = W5a [Wuv'|Wa'|Wt'|0]: the skip half of pts_linears[5] ...
//     [80,96)  ... then its 16 slabs W5[:, 256:]
//     [96,129) pts_linears 6, 7 and the output block
//   backward, 120 steps: for k = 7..1: [k==5: G5[:, audio]^T, 4 slabs] W_{k-1}^T (16 slabs); then
//   G0[:, audio]^T (4 slabs).
// Rows (x, saved activations, dz) move through ordinary global loads/stores issued by the compiler;
// they only make the hand-counted vmcnt waits more conservative (see render.hip).
#include "s2l_common.h"
#include <algorithm>
#include <atomic>

namespace s2l {

constexpr int kRing = 9;
constexpr int kDepth = kRing - 1;
constexpr int kSlabBytes = kSlab * 4;
constexpr int kSlabQuads = kSlabBytes / 16;
constexpr int kLdsBiasFloats = kHidden * kW + 4 + 2 * kW;   // OFF_BIAS, OFF_BOUT, OFF_BG0, OFF_BG5 are contiguous
constexpr int kLdsBytes = kRing * kSlabBytes + kLdsBiasFloats * 4;
static_assert(OFF_BG0 == OFF_BOUT + 4 && OFF_BG5 == OFF_BG0 + kW, "bias blocks must be contiguous");
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
constexpr int kFwdSteps = 8 + 64 + 8 + 16 + 33;   // 129
constexpr int kBwdSteps = 7 * 16 + 8;             // 120

__device__ __forceinline__ f4 mfma16r(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void dma_4k_r(const char* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_r() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// float offset (into the packed blob) of the block consumed at forward consumption index c in [0,129)
__device__ __forceinline__ int64_t fwd_step_src(int c) {
  // 0..7 G0 | 8..71 W slabs 0..63 | 72..79 G5 | 80..95 W slabs 64..79 | 96..128 W slabs 80..112
  const int64_t w = OFF_WMLP + (int64_t)(c < 72 ? c - 8 : c - 16) * kSlab;
  const int64_t g0 = OFF_WG0 + (int64_t)c * kSlab;
  const int64_t g5 = OFF_WG5 + (int64_t)(c - 72) * kSlab;
  return c < 8 ? g0 : (c >= 72 && c < 80) ? g5 : w;
}
// backward consumption index c in [0,120): k=7:[0,16) k=6:[16,32) k=5: G5AT [32,36) then W4^T [36,52)
// k=4:[52,68) k=3:[68,84) k=2:[84,100) k=1:[100,116) then G0AT [116,120)
__device__ __forceinline__ int64_t bwd_step_src(int c) {
  const int layer = c < 32 ? 6 - (c >> 4) : c < 36 ? 0 : 4 - ((c - 36) >> 4);   // transposed hidden layer index
  const int mb = c < 32 ? (c & 15) : (c - 36) & 15;
  const int64_t w = OFF_WMLPT + (int64_t)(layer * 16 + mb) * kSlab;
  const int64_t g5 = OFF_G5AT + (int64_t)(c - 32) * kSlab;
  const int64_t g0 = OFF_G0AT + (int64_t)(c - 116) * kSlab;
  return (c >= 32 && c < 36) ? g5 : c >= 116 ? g0 : w;
}

template <int G, bool BWD>
struct RowCtx {
  const float* packed;
  const f4* ring;
  uint32_t lds_base;
  int lane, wave, q, px;
  int cur;    // ring buffer being consumed
  int step;   // consumption index within the tile, of the block in `cur`
  const char* lane_src;

  __device__ __forceinline__ void issue(int c_abs, int buf) const {
    constexpr int kSteps = BWD ? kBwdSteps : kFwdSteps;
    int c = c_abs;
    c = c >= kSteps ? c - kSteps : c;
    const int64_t off = BWD ? bwd_step_src(c) : fwd_step_src(c);
    dma_4k_r(lane_src + off * 4, lds_base + buf * kSlabBytes + wave * 4096);
  }
  // publish the next block, retire the current one and refill its buffer
  __device__ __forceinline__ void advance() {
    constexpr int kSteps = BWD ? kBwdSteps : kFwdSteps;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt_r<4 * (kDepth - 1)>();
    asm volatile("s_barrier" ::: "memory");
    issue(step + kRing, cur);
    cur = cur + 1 == kRing ? 0 : cur + 1;
    step = step + 1 == kSteps ? 0 : step + 1;
  }
  __device__ __forceinline__ const f4* slab() const { return ring + cur * kSlabQuads + lane; }
};

// One 16-quad block: acc_lo gets quads 0..NQ-1 against in_lo, (K=128 blocks hold two M-blocks of 8
// quads: then quads 8..15 go to acc_hi against the same operand).  A quads are prefetched two ahead
// into three rotating register sets; the block hand-off happens at quad 14.
template <int G, bool BWD, int KQ, typename InT>
__device__ __forceinline__ void consume_block(RowCtx<G, BWD>& c, const InT& in, f4 (&acc_lo)[G], f4 (&acc_hi)[G], f4& w0,
                                              f4& w1) {
  const f4* sl = c.slab();
#pragma unroll
  for (int jq = 0; jq < 16; ++jq) {
    if (jq == 14) {
      c.advance();
      sl = c.slab() - 16 * 64;
    }
    const f4 w2 = sl[(jq + 2) * 64];
    __builtin_amdgcn_sched_barrier(0);
    const int j4 = KQ == 16 ? jq : (jq & 7);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (KQ == 16 || jq < 8) acc_lo[g] = mfma16r(w0[jj], in[g][j4 * 4 + jj], acc_lo[g]);
        else acc_hi[g] = mfma16r(w0[jj], in[g][j4 * 4 + jj], acc_hi[g]);
      }
    __builtin_amdgcn_sched_barrier(0);
    w0 = w1;
    w1 = w2;
  }
}

struct RowsFwdArgs {
  const float* packed;
  const float* x;     // [N,128]
  float* out;         // [N,3]
  float* hsave;       // optional [8][N][256]
  int64_t total;
  int ntiles;
};

template <int G>
__global__ __launch_bounds__(256) void rows_fwd_kernel(RowsFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;
  RowCtx<G, false> c{a.packed, reinterpret_cast<const f4*>(smem), (uint32_t)(uintptr_t)smem, lane, wave, q, px, 0, 0,
                     reinterpret_cast<const char*>(a.packed) + wave * 4096 + lane * 16};
  const float* lds_b = reinterpret_cast<const float*>(smem + kRing * kSlabBytes) + 4 * q;
  const float* lds_bout = reinterpret_cast<const float*>(smem + kRing * kSlabBytes) + kHidden * kW;
  const float* lds_bg0 = lds_b + kHidden * kW + 4;
  const float* lds_bg5 = lds_bg0 + kW;
#pragma unroll 1
  for (int s = 0; s < kDepth; ++s) c.issue(s, s);
  for (int i = threadIdx.x; i < kLdsBiasFloats; i += 256)
    reinterpret_cast<float*>(smem + kRing * kSlabBytes)[i] = a.packed[OFF_BIAS + i];
  // block 0 landed and published (and the bias block written by every wave); top the ring up
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vmcnt_r<4 * (kDepth - 1)>();
  asm volatile("s_barrier" ::: "memory");
  c.issue(kDepth, kDepth);

  float in[G][64];
  f4 acc[G][16];
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int64_t row[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t n = ((int64_t)tile * 4 + wave) * (G * 16) + g * 16 + px;
      live[g] = n < a.total;
      row[g] = live[g] ? n : a.total - 1;
    }
    auto load_x = [&](float (&xin)[G][32]) {   // kin(j, q) = 32*q + j
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f4* xr = reinterpret_cast<const f4*>(a.x + row[g] * kGenK + 32 * q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f4 v = xr[i];
#pragma unroll
          for (int r = 0; r < 4; ++r) xin[g][i * 4 + r] = v[r];
        }
      }
    };
    auto init_bias = [&](const float* b) {
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        const f4 v = *reinterpret_cast<const f4*>(b + mb * 16);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = v;
      }
    };
    auto relu_to_in = [&](int k) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) {
          f4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = in[g][mb * 4 + r] = fmaxf(acc[g][mb][r], 0.f);
          if (a.hsave && live[g])
            *reinterpret_cast<f4*>(a.hsave + ((int64_t)k * a.total + row[g]) * kW + mb * 16 + 4 * q) = h;
        }
    };
    // K=128 product with a folded matrix: 8 blocks of two M-blocks
    auto fold_layer = [&](f4& w0, f4& w1) {
      float xin[G][32];
      load_x(xin);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        f4 lo[G], hi[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          lo[g] = acc[g][2 * s];
          hi[g] = acc[g][2 * s + 1];
        }
        consume_block<G, false, 8>(c, xin, lo, hi, w0, w1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          acc[g][2 * s] = lo[g];
          acc[g][2 * s + 1] = hi[g];
        }
      }
    };

    f4 w0 = c.slab()[0], w1 = c.slab()[64];
    // h0 = relu(G0 x + c0)
    init_bias(lds_bg0);
    fold_layer(w0, w1);
    relu_to_in(0);
    for (int layer = 0; layer < kHidden; ++layer) {
      if (layer == 4) {
        // pts_linears[5] on cat([skip, h4]): G5 x + c5 first, then the 16 slabs of W5[:, 256:]
        init_bias(lds_bg5);
        fold_layer(w0, w1);
      } else {
        init_bias(lds_b + layer * kW);
      }
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        f4 lo[G];
#pragma unroll
        for (int g = 0; g < G; ++g) lo[g] = acc[g][mb];
        consume_block<G, false, 16>(c, in, lo, lo, w0, w1);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = lo[g];
      }
      relu_to_in(layer + 1);
    }
    // output_linear
    f4 rgb[G];
    {
      const f4 b = *reinterpret_cast<const f4*>(lds_bout);
#pragma unroll
      for (int g = 0; g < G; ++g) rgb[g] = b;
      consume_block<G, false, 16>(c, in, rgb, rgb, w0, w1);
    }
    if (q == 0) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (live[g]) {
          float* o = a.out + row[g] * 3;
          o[0] = rgb[g][0];
          o[1] = rgb[g][1];
          o[2] = rgb[g][2];
        }
    }
  }
  wait_vmcnt_r<0>();
}

struct RowsBwdArgs {
  const float* packed;
  const float* drgb;    // [N,3]
  const float* hsave;   // [8][N][256]
  float* dzsave;        // [8][N][256]
  float* dxa;           // [N,64]
  int64_t total;
  int ntiles;
};

template <int G>
__global__ __launch_bounds__(256) void rows_bwd_kernel(RowsBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;
  RowCtx<G, true> c{a.packed, reinterpret_cast<const f4*>(smem), (uint32_t)(uintptr_t)smem, lane, wave, q, px, 0, 0,
                    reinterpret_cast<const char*>(a.packed) + wave * 4096 + lane * 16};
#pragma unroll 1
  for (int s = 0; s < kDepth; ++s) c.issue(s, s);
  wait_vmcnt_r<4 * (kDepth - 1)>();
  asm volatile("s_barrier" ::: "memory");
  c.issue(kDepth, kDepth);

  float in[G][64];
  f4 acc[G][16];
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int64_t row[G];
    bool live[G];
    f4 dxa[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t n = ((int64_t)tile * 4 + wave) * (G * 16) + g * 16 + px;
      live[g] = n < a.total;
      row[g] = live[g] ? n : a.total - 1;
#pragma unroll
      for (int m = 0; m < 4; ++m) dxa[g][m] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    auto mask_store = [&](int k) {   // dz_k = dh_k * (h_k > 0): stored and moved to the operand registers
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) {
          const int64_t off = ((int64_t)k * a.total + row[g]) * kW + mb * 16 + 4 * q;
          const f4 h = *reinterpret_cast<const f4*>(a.hsave + off);
          f4 d;
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r] = in[g][mb * 4 + r] = h[r] > 0.f ? acc[g][mb][r] : 0.f;
          if (live[g]) *reinterpret_cast<f4*>(a.dzsave + off) = d;
        }
    };
    auto audio_cols = [&](f4& w0, f4& w1) {   // dxa += G[:, audio]^T dz: 4 slabs, M = 4 blocks
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f4 lo[G];
#pragma unroll
        for (int g = 0; g < G; ++g) lo[g] = dxa[g][m];
        consume_block<G, true, 16>(c, in, lo, lo, w0, w1);
#pragma unroll
        for (int g = 0; g < G; ++g) dxa[g][m] = lo[g];
      }
    };

    // dh7 = Wout^T drgb: K = 4 (3 used), one MFMA per M-block, operands straight from global
    {
      float b[G];
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = q < 3 ? a.drgb[row[g] * 3 + q] : 0.f;
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        const float w = a.packed[OFF_WOUTT + mb * 64 + lane];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = mfma16r(w, b[g], (f4){0.f, 0.f, 0.f, 0.f});
      }
    }
    mask_store(7);
    f4 w0 = c.slab()[0], w1 = c.slab()[64];
    for (int k = 7; k >= 1; --k) {
      if (k == 5) audio_cols(w0, w1);
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        f4 lo[G];
#pragma unroll
        for (int g = 0; g < G; ++g) lo[g] = (f4){0.f, 0.f, 0.f, 0.f};
        consume_block<G, true, 16>(c, in, lo, lo, w0, w1);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = lo[g];
      }
      mask_store(k - 1);
    }
    audio_cols(w0, w1);
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (live[g])
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<f4*>(a.dxa + row[g] * 64 + m * 16 + 4 * q) = dxa[g][m];
  }
  wait_vmcnt_r<0>();
}

// The feature-split tile of the forward (gen_rows_fs_body.py): 16 rows per workgroup, the four waves own 64 features each and exchange the
// activations through LDS -- for calls of a few thousand rows (one frame of the reference's per-frame driver), where the column form above
// leaves most CUs idle behind 100-us chains.  Same MFMA chains per output: the same bits.  No saved activations (inference only).
constexpr int kRfsRingPerWave = 8 * 4096;
constexpr int kRfsExchange = 4 * kRfsRingPerWave;
constexpr int kRfsLdsBytes = kRfsExchange + 2 * 16384;
static_assert(kRfsLdsBytes == 160 * 1024, "gen_rows_fs_body.py: LDS_BYTES");
static_assert(OFF_WG5 == OFF_WG0 + 16 * (kSlab / 2) && OFF_BOUT == OFF_BIAS + kHidden * kW && OFF_BG0 == OFF_BOUT + 4 && OFF_BG5 == OFF_BG0 + kW,
              "gen_rows_fs_body.py walks the folded matrices and the bias blocks by these strides");
__global__ __launch_bounds__(256) void rows_fs_kernel(RowsFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * kRfsRingPerWave);
  const int tile0 = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * blockIdx.x / gridDim.x));
  const int tile_end = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * (blockIdx.x + 1) / gridDim.x));
  const int total = __builtin_amdgcn_readfirstlane((int)a.total);
  const float* wb = a.packed + OFF_WMLP + (int64_t)wave * 4 * kSlab;            // the wave's four slabs of a layer (64 contiguous KiB)
  const float* wout = a.packed + OFF_WOUT;
  const float* wg0 = a.packed + OFF_WG0 + (int64_t)wave * 4 * (kSlab / 2);      // ... and its four 8-KiB M-blocks of the folded matrices
  const float* wg5 = a.packed + OFF_WG5 + (int64_t)wave * 4 * (kSlab / 2);
  const float* biasp = a.packed + OFF_BIAS;
#include "rows_fs_body.inc"
}

static int rows_grid(int ntiles, const void* kernel, LdsOptIn& flags, int* grid) {
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  if ((rc = ensure_dynamic_lds(kernel, kLdsBytes, flags, dev))) return rc;
  *grid = ntiles < n_cu ? ntiles : n_cu;
  return 0;
}

template <int G>
static int launch_rows_fwd_g(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st) {
  const int64_t ntiles = (n_rows + G * 64 - 1) / (G * 64);
  if (ntiles > 0x7fffffff) return S2L_E_SIZE;
  RowsFwdArgs a{packed, x, out, hsave, n_rows, (int)ntiles};
  int grid = 0;
  static LdsOptIn flags;
  int rc = rows_grid(a.ntiles, reinterpret_cast<const void*>(&rows_fwd_kernel<G>), flags, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL((rows_fwd_kernel<G>), dim3(grid), dim3(256), kLdsBytes, st, a);
  return (int)hipGetLastError();
}

static std::atomic<int> g_rows_kernel{0};      // 0: choose per call, 1: the column form, 2: the feature-split tile (s2l_set_rows_kernel)

static int launch_rows_fs(const float* packed, const float* x, float* out, int64_t n_rows, hipStream_t st) {
  const int64_t ntiles = (n_rows + 15) / 16;
  if (ntiles > 0x7fffffff || n_rows * 512 >= 0x7fffffffLL) return S2L_E_SIZE;      // (the body addresses x by 32-bit byte offsets)
  RowsFwdArgs a{packed, x, out, nullptr, n_rows, (int)ntiles};
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  static LdsOptIn flags;
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&rows_fs_kernel), kRfsLdsBytes, flags, dev))) return rc;
  hipLaunchKernelGGL(rows_fs_kernel, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kRfsLdsBytes, st, a);
  return (int)hipGetLastError();
}

int launch_rows_fwd(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st) {
  // G = 2: 128-row tiles (x operands, 32 registers per group, live beside both 64-register arrays).  One frame of the reference's
  // per-frame call (inference.py:158: 9 216 rows at 96x96) is only 72 such tiles for 256 CUs: when the 128-row tiles do not fill
  // the chip, 64-row tiles (G = 1: half the MFMAs per weight slab, twice the workgroups) finish sooner.  Same per-row arithmetic.
  int dev = 0, n_cu = 0;
  const int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  // A few thousand rows (one frame of the per-frame driver: 4 096 at 64 x 64, 9 216 at 96 x 96; the ensemble's 4 x 9 216): rounds of 16-row tiles
  // whose waves split the features (~37 us each) against rounds of 64- / 128-row column tiles (~140 / ~270 us): whichever finishes first.  Long
  // calls stay with the column form (a weight slab serves 32 rows per wave there: the least L2 traffic per row).
  const int kind = g_rows_kernel.load(std::memory_order_relaxed);
  if (!hsave && n_rows * 512 < 0x7fffffffLL && kind != 1) {
    auto rounds = [&](int64_t rows_per_tile) { return (double)(((n_rows + rows_per_tile - 1) / rows_per_tile + n_cu - 1) / n_cu); };
    const double t_fs = 37.0 * rounds(16), t_col = std::min(140.0 * rounds(64), 270.0 * rounds(128));
    if (kind == 2 || t_fs < 0.95 * t_col) return launch_rows_fs(packed, x, out, n_rows, st);
  }
  if ((n_rows + 127) / 128 < n_cu) return launch_rows_fwd_g<1>(packed, x, out, hsave, n_rows, st);
  return launch_rows_fwd_g<2>(packed, x, out, hsave, n_rows, st);
}

int launch_rows_bwd(const float* packed, const float* drgb, const float* hsave, float* dzsave, float* dxa, int64_t n_rows,
                    hipStream_t st) {
  constexpr int G = 2;
  const int64_t ntiles = (n_rows + G * 64 - 1) / (G * 64);
  if (ntiles > 0x7fffffff) return S2L_E_SIZE;
  RowsBwdArgs a{packed, drgb, hsave, dzsave, dxa, n_rows, (int)ntiles};
  int grid = 0;
  static LdsOptIn flags;
  int rc = rows_grid(a.ntiles, reinterpret_cast<const void*>(&rows_bwd_kernel<G>), flags, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL((rows_bwd_kernel<G>), dim3(grid), dim3(256), kLdsBytes, st, a);
  return (int)hipGetLastError();
}

}  // namespace s2l

// Which kernel runs the general-row forward when no activations are saved: 0 (default) chosen per call by the row count, 1 always the column
// form (64- / 128-row tiles, a wave per 16 rows), 2 always the feature-split tile (16 rows per workgroup).  Same bits (a test and A/B aid).
extern "C" int s2l_set_rows_kernel(int kind) {
  if (kind < 0 || kind > 2) return S2L_E_SIZE;
  s2l::g_rows_kernel.store(kind, std::memory_order_relaxed);
  return S2L_OK;
}

// s2l_rgb_forward: the exact drop-in for TalkingFace.rgb_forward (tf_nerf.py:225-285) on arbitrary [N,66] rows: embed the
// rows (frontend.hip), then the general-row MLP above.  The clip renderer (s2l_render_lip) lives in render.hip.
namespace s2l {
int launch_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                      hipStream_t st);
}

extern "C" int s2l_rgb_forward(const float* packed, const float* uv_audio, int64_t time_index, float* xbuf, float* out,
                               int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed || !uv_audio || !xbuf || !out) return S2L_E_NULL;
  if (s2l::misaligned16(packed) || s2l::misaligned16(xbuf)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = s2l::launch_embed_rows(packed, uv_audio, time_index, xbuf, n_rows, st);
  if (rc) return rc;
  return s2l::launch_rows_fwd(packed, xbuf, out, nullptr, n_rows, st);
}
