// Fused implicit-function MLP (the hot kernel).
//
// Replaces TalkingFace.rgb_forward (tf_nerf.py:225-285) and, in table mode, the whole per-frame
// driver loop inference.py:140-159: one launch renders every pixel of every frame of a clip.
//
// Structure:
//   * one wave owns G groups of 16 (pixel,frame) samples and ALL 256 features of them;
//   * every 256x256 layer is D[feature][sample] = W[feature][k] * H[k][sample] on
//     v_mfma_f32_16x16x4_f32 (exact fp32, 157 TFLOP/s peak): A = weights, B = activations;
//   * the weights are packed (pack.hip) in the k-order in which the PREVIOUS layer's accumulator
//     registers hold the features, so a layer's D registers are the next layer's B operands
//     as they stand: activations never leave the register file, ReLU is one v_max per register;
//   * layer 0 and the skip half of layer 5 are affine in (pixel-only) + (frame-only) terms and
//     arrive as tables p0/p5 [HW,256] and q0/q5 [F,256], or, in GENERAL mode
//     (arbitrary rows, training-time ensemble), as a K=128 MFMA product with folded matrices.
#include "s2l_common.h"

namespace s2l {

struct MlpArgs {
  const float* packed;
  const float* p0;   // table mode [HW,256]            | general mode: x [N,128]
  const float* p5;
  const float* q0;   // [F,256]
  const float* q5;
  float* out;        // [N,3]
  int64_t total;     // samples = F*HW (table) or rows (general)
  int hw;
};

__device__ inline f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One full layer: acc[g][mb] += W[mb-block] . in[g]   (NJ4*16 = K)
template <int G, int NJ4>
__device__ __forceinline__ void gemm_layer(const f4* __restrict__ wl, int lane, f4 (&acc)[G][16],
                                           const float (&in)[G][NJ4 * 4]) {
#pragma unroll
  for (int mb = 0; mb < 16; ++mb) {
#pragma unroll
    for (int j4 = 0; j4 < NJ4; ++j4) {
      const f4 a = wl[(mb * NJ4 + j4) * 64 + lane];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = mfma16(a[jj], in[g][j4 * 4 + jj], acc[g][mb]);
      }
    }
  }
}

template <int G, int NW, bool GENERAL>
__global__ __launch_bounds__(NW * 64) void mlp_kernel(MlpArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = lane >> 4, px = lane & 15;
  const int64_t nbase = ((int64_t)blockIdx.x * NW + wave) * (G * 16);
  if (nbase >= a.total) return;
  const float* __restrict__ packed = a.packed;

  float in[G][64];
  f4 acc[G][16];
  int64_t pix[G], frm[G];

#pragma unroll
  for (int g = 0; g < G; ++g) {
    int64_t n = nbase + g * 16 + px;
    n = n < a.total ? n : a.total - 1;
    if constexpr (GENERAL) {
      pix[g] = n;
      frm[g] = 0;
    } else {
      frm[g] = n / a.hw;
      pix[g] = n - frm[g] * a.hw;
    }
  }

  if constexpr (GENERAL) {
    // h0 = relu(M0 x + c0), x rows [128] read as 8 quads per lane: kin(j, q) = 32*q + j
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const f4 b = *reinterpret_cast<const f4*>(packed + OFF_BG0 + mb * 16 + 4 * q);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g][mb] = b;
    }
    {
      float xin[G][32];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f4* xr = reinterpret_cast<const f4*>(a.p0 + pix[g] * kGenK + 32 * q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f4 v = xr[i];
#pragma unroll
          for (int r = 0; r < 4; ++r) xin[g][i * 4 + r] = v[r];
        }
      }
      gemm_layer<G, 8>(reinterpret_cast<const f4*>(packed + OFF_WG0), lane, acc, xin);
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(acc[g][mb][r], 0.f);
  } else {
    // h0 = relu(p0[pixel] + q0[frame])
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f4* P = reinterpret_cast<const f4*>(a.p0 + pix[g] * kW + 4 * q);
      const f4* Q = reinterpret_cast<const f4*>(a.q0 + frm[g] * kW + 4 * q);
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        const f4 s = P[mb * 4] + Q[mb * 4];
#pragma unroll
        for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(s[r], 0.f);
      }
    }
  }

  for (int layer = 0; layer < kHidden; ++layer) {
    if (layer == 4) {
      // pts_linears[5] on cat([skip, h4]): the skip half initialises the accumulator
      if constexpr (GENERAL) {
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) {
          const f4 b = *reinterpret_cast<const f4*>(packed + OFF_BG5 + mb * 16 + 4 * q);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g][mb] = b;
        }
        float xin[G][32];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const f4* xr = reinterpret_cast<const f4*>(a.p0 + pix[g] * kGenK + 32 * q);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f4 v = xr[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) xin[g][i * 4 + r] = v[r];
          }
        }
        gemm_layer<G, 8>(reinterpret_cast<const f4*>(packed + OFF_WG5), lane, acc, xin);
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const f4* P = reinterpret_cast<const f4*>(a.p5 + pix[g] * kW + 4 * q);
          const f4* Q = reinterpret_cast<const f4*>(a.q5 + frm[g] * kW + 4 * q);
#pragma unroll
          for (int mb = 0; mb < 16; ++mb) acc[g][mb] = P[mb * 4] + Q[mb * 4];
        }
      }
    } else {
      const float* bias = packed + OFF_BIAS + layer * kW + 4 * q;
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        const f4 b = *reinterpret_cast<const f4*>(bias + mb * 16);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = b;
      }
    }
    gemm_layer<G, 16>(reinterpret_cast<const f4*>(packed + OFF_WMLP) + (int64_t)layer * 16 * 16 * 64, lane, acc, in);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int mb = 0; mb < 16; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(acc[g][mb][r], 0.f);
  }

  // output_linear (3 rows, zero-padded to one 16-row M-block); no activation (tf_nerf.py:283)
  f4 rgb[G];
  {
    const f4 b = *reinterpret_cast<const f4*>(packed + OFF_BOUT);
#pragma unroll
    for (int g = 0; g < G; ++g) rgb[g] = b;
    const f4* wl = reinterpret_cast<const f4*>(packed + OFF_WOUT);
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) {
      const f4 w = wl[j4 * 64 + lane];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int g = 0; g < G; ++g) rgb[g] = mfma16(w[jj], in[g][j4 * 4 + jj], rgb[g]);
    }
  }
  if (q == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t n = nbase + g * 16 + px;
      if (n < a.total) {
        float* o = a.out + n * 3;
        o[0] = rgb[g][0];
        o[1] = rgb[g][1];
        o[2] = rgb[g][2];
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Table-mode kernel, v2: the 113 weight slabs (7 layers x 16 M-blocks + the output block, 16 KiB
// each, contiguous in the packed blob) stream L2 -> LDS through an NBUF-deep ring filled by
// LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass); all 4 waves of the workgroup
// consume the same slab, each for its own 48 samples.  One s_barrier per slab:
//     wait own DMA parts of slab s  ->  barrier  ->  issue DMA of slab s+NBUF-1  ->  64x3 MFMAs
// The barrier both publishes slab s (every wave waited for its own quarter) and retires the
// buffer of slab s-1, which is the one the new DMA overwrites.  DMA is issued from inline asm, so
// hipcc neither counts nor drains it (cdna_hip_programming.md §5.7); vmcnt is counted by hand.
constexpr int kRing = 8;                 // slabs in the LDS ring (8 x 16 KiB = 128 KiB)
constexpr int kDepth = kRing - 1;        // slabs in flight ahead of the one being consumed
constexpr int kSlabBytes = kSlab * 4;    // 16384
constexpr int kNumSlabs = kHidden * 16 + 1;
constexpr int kBiasFloats = kHidden * kW + 4;   // OFF_BIAS .. OFF_BOUT+4 are contiguous in the blob
constexpr int kLdsBytes = kRing * kSlabBytes + kBiasFloats * 4;
static_assert(OFF_WOUT == OFF_WMLP + int64_t(kHidden) * 16 * kSlab, "slabs must be contiguous");
static_assert(OFF_BOUT == OFF_BIAS + kHidden * kW, "bias block must be contiguous");
static_assert(16 % kRing == 0, "ring index must be a compile-time function of the M-block");

// Each wave moves one quarter (4 x 1 KiB) of a slab.  gsrc = this lane's source address of the
// first KiB (wave-quarter base + lane*16); lds_dst = wave-uniform LDS byte address of that KiB.
__device__ __forceinline__ void dma_quarter_slab(const char* gsrc, uint32_t lds_dst) {
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
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wg_barrier() {
#ifndef S2L_EXP_NO_BARRIER   // timing experiment only (results invalid without the barrier)
  asm volatile("s_barrier" ::: "memory");
#endif
}

template <int G>
__device__ __forceinline__ void mfma_quad(const f4& w, const float (&in)[G][64], int j4, f4 (&acc)[G]) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma16(w[jj], in[g][j4 * 4 + jj], acc[g]);
}

template <int G>
__device__ __forceinline__ void mfma_quad_mb(const f4& w, const float (&in)[G][64], int j4, f4 (&acc)[G][16], int mb) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][mb] = mfma16(w[jj], in[g][j4 * 4 + jj], acc[g][mb]);
}

#ifdef S2L_EXP_TRACE   // experiment build only: per-workgroup phase timestamps (s_memtime)
__device__ long long* g_trace = nullptr;
#define S2L_TRACE(slot)                                                              \
  do {                                                                               \
    if (g_trace && threadIdx.x == 0) g_trace[(int64_t)blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define S2L_TRACE(slot) do { } while (0)
#endif

template <int G>
__global__ __launch_bounds__(256) void mlp_ring_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  S2L_TRACE(0);
#ifdef S2L_EXP_TRACE
  if (g_trace && threadIdx.x == 0) g_trace[(int64_t)blockIdx.x * 16 + 14] = wall_clock64();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;
  const int64_t nbase = ((int64_t)blockIdx.x * 4 + wave) * (G * 16);
  const float* __restrict__ packed = a.packed;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  const float* lds_bias = reinterpret_cast<const float*>(smem + kRing * kSlabBytes) + 4 * q;
  const f4* ring = reinterpret_cast<const f4*>(smem) + lane;   // this lane's A-operand quads

  // this wave's quarter of slab 0: global source (per lane) and LDS destination (uniform)
  const char* gq = reinterpret_cast<const char*>(packed + OFF_WMLP) + wave * 4096 + lane * 16;
  const uint32_t lq = lds_base + wave * 4096;
  auto issue = [&](int slab, int buf) {   // slab index clamps to the last one: tail DMAs are dummies
#ifdef S2L_EXP_NO_DMA
    if (slab >= kRing) return;
#endif
    const int sl = slab < kNumSlabs ? slab : kNumSlabs - 1;
    dma_quarter_slab(gq + (int64_t)sl * kSlabBytes, lq + buf * kSlabBytes);
  };
#pragma unroll
  for (int s = 0; s < kDepth; ++s) issue(s, s);

  // biases -> LDS (ordinary loads; nothing of the ring is read before the first barrier)
  for (int i = threadIdx.x; i < kBiasFloats; i += 256)
    reinterpret_cast<float*>(smem + kRing * kSlabBytes)[i] = packed[OFF_BIAS + i];

  float in[G][64];
  f4 acc[G][16];
  int pix[G], frm[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    int64_t n = nbase + g * 16 + px;
    n = n < a.total ? n : a.total - 1;
    frm[g] = (int)(n / a.hw);
    pix[g] = (int)(n - (int64_t)frm[g] * a.hw);
  }
  // h0 = relu(p0[pixel] + q0[frame])
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const f4* P = reinterpret_cast<const f4*>(a.p0 + (int64_t)pix[g] * kW + 4 * q);
    const f4* Q = reinterpret_cast<const f4*>(a.q0 + (int64_t)frm[g] * kW + 4 * q);
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const f4 s = P[mb * 4] + Q[mb * 4];
#pragma unroll
      for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(s[r], 0.f);
    }
  }

  // slab 0: landed + published; then keep the ring full
  S2L_TRACE(1);
  wait_vmcnt<4 * (kDepth - 1)>();
  wg_barrier();
  S2L_TRACE(2);
  issue(kDepth, kDepth % kRing);
  // A-operand quads are prefetched TWO ahead into three rotating register sets: a ds_read never
  // overwrites registers that MFMAs issued in the last ~400 cycles are still reading (the WAR
  // interlock on a 2-set rotation costs ~40 cycles per quad = 11 % of the kernel).
  f4 w0 = ring[0], w1 = ring[64];

  for (int layer = 0; layer < kHidden; ++layer) {
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const f4* sl = ring + (mb % kRing) * (kSlabBytes / 16);
      const f4* sn = ring + ((mb + 1) % kRing) * (kSlabBytes / 16);
      {
#ifdef S2L_EXP_NO_BIAS
        const f4 b = (f4){0.f, 0.f, 0.f, 0.f};
#else
        const f4 b = *reinterpret_cast<const f4*>(lds_bias + layer * kW + mb * 16);
#endif
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = b;
      }
#pragma unroll
      for (int j4 = 0; j4 < 16; ++j4) {
#ifndef S2L_EXP_NO_BOUNDARY
        if (j4 == 14) {
          // quads 14 and 15 are in registers: this wave is done READING slab s.  Publish slab s+1,
          // retire slab s, refill its buffer; the next two reads come from slab s+1.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wait_vmcnt<4 * (kDepth - 1)>();
          wg_barrier();
          issue(layer * 16 + mb + 1 + kDepth, mb % kRing);
        }
#endif
#ifdef S2L_EXP_NO_DSREAD
        f4 w2 = w0;
        asm volatile("" : "+v"(w2));
#else
        const f4 w2 = j4 < 14 ? sl[(j4 + 2) * 64] : sn[(j4 - 14) * 64];
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad_mb<G>(w0, in, j4, acc, mb);
        __builtin_amdgcn_sched_barrier(0);
        w0 = w1;
        w1 = w2;
      }
    }
    if (layer == 4) {
      // pts_linears[5] on cat([skip, h4]): add the skip half p5[pixel] + q5[frame] (q5 carries b5;
      // the packed bias row of this layer is zero).  `in` is dead here, so the loads are free.
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f4* P = reinterpret_cast<const f4*>(a.p5 + (int64_t)pix[g] * kW + 4 * q);
        const f4* Q = reinterpret_cast<const f4*>(a.q5 + (int64_t)frm[g] * kW + 4 * q);
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) {
          const f4 s = P[mb * 4] + Q[mb * 4];
#pragma unroll
          for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(acc[g][mb][r] + s[r], 0.f);
        }
      }
    } else {
#ifdef S2L_EXP_NO_EPILOGUE   // timing experiment only: skip the per-layer ReLU/move (keeps acc alive)
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) asm volatile("" ::"a"(acc[g][mb]));
#else
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mb = 0; mb < 16; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(acc[g][mb][r], 0.f);
#endif
    }
    S2L_TRACE(3 + layer);
  }

  // output_linear: slab 112 (ring buffer 0, already published; w0/w1 = its quads 0/1); no activation
  f4 rgb[G];
  {
    const f4 b = *reinterpret_cast<const f4*>(smem + kRing * kSlabBytes + kHidden * kW * 4);
#pragma unroll
    for (int g = 0; g < G; ++g) rgb[g] = b;
    const f4* sl = ring + ((kNumSlabs - 1) % kRing) * (kSlabBytes / 16);
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) {
      f4 w2 = w1;
      if (j4 < 14) w2 = sl[(j4 + 2) * 64];
      __builtin_amdgcn_sched_barrier(0);
      mfma_quad<G>(w0, in, j4, rgb);
      __builtin_amdgcn_sched_barrier(0);
      w0 = w1;
      w1 = w2;
    }
  }
  if (q == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t n = nbase + g * 16 + px;
      if (n < a.total) {
        float* o = a.out + n * 3;
        o[0] = rgb[g][0];
        o[1] = rgb[g][1];
        o[2] = rgb[g][2];
      }
    }
  }
  S2L_TRACE(10);
  wait_vmcnt<0>();   // tail (dummy) DMAs must land before the workgroup's LDS is released
  S2L_TRACE(11);
#ifdef S2L_EXP_TRACE
  if (g_trace && threadIdx.x == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_trace[(int64_t)blockIdx.x * 16 + 12] = hwid;
    g_trace[(int64_t)blockIdx.x * 16 + 13] = xcc;
    g_trace[(int64_t)blockIdx.x * 16 + 15] = wall_clock64();
  }
#endif
}

#ifdef S2L_EXP_TRACE
extern "C" int s2l_debug_set_trace(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p));
}
#endif

static int launch_mlp_ring(const MlpArgs& a, hipStream_t st) {
  constexpr int G = 3;
  const int64_t per_block = (int64_t)G * 16 * 4;
  const int64_t blocks = (a.total + per_block - 1) / per_block;
  if (blocks > 0x7fffffff) return S2L_E_SIZE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ring_kernel<G>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((mlp_ring_kernel<G>), dim3((unsigned)blocks), dim3(256), kLdsBytes, st, a);
  return (int)hipGetLastError();
}

template <int G, int NW, bool GENERAL>
static int launch_mlp(const MlpArgs& a, hipStream_t st) {
  const int64_t per_block = (int64_t)G * 16 * NW;
  const int64_t blocks = (a.total + per_block - 1) / per_block;
  if (blocks > 0x7fffffff) return S2L_E_SIZE;
  hipLaunchKernelGGL((mlp_kernel<G, NW, GENERAL>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, a);
  return (int)hipGetLastError();
}

int launch_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                      hipStream_t st);

}  // namespace s2l

extern "C" int s2l_render_lip(const float* packed, const float* p0, const float* p5, const float* q0, const float* q5,
                              float* out, int64_t hw, int64_t n_frames, s2l_stream_t stream) {
  if (hw <= 0 || hw > 0x7fffffff || n_frames < 0) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!packed || !p0 || !p5 || !q0 || !q5 || !out) return S2L_E_NULL;
  if (s2l::misaligned16(packed) || s2l::misaligned16(p0) || s2l::misaligned16(p5) || s2l::misaligned16(q0) ||
      s2l::misaligned16(q5))
    return S2L_E_ALIGN;
  s2l::MlpArgs a{packed, p0, p5, q0, q5, out, hw * n_frames, (int)hw};
  return s2l::launch_mlp_ring(a, static_cast<hipStream_t>(stream));
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
  s2l::MlpArgs a{packed, xbuf, nullptr, nullptr, nullptr, out, n_rows, 1};
  return s2l::launch_mlp<2, 4, true>(a, st);
}
