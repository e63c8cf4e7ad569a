// Fused clip renderer (the hot kernel): s2l_render_lip.
//
// Replaces the per-frame driver loop inference.py:140-159 and TalkingFace.rgb_forward
// (tf_nerf.py:225-285) for whole clips: one persistent launch renders every pixel of every frame.
//
//   * tile = 16 consecutive pixels x 12 consecutive frames = 192 samples; a workgroup = 4 waves
//     (one per SIMD, 512 registers each); wave w owns frames 3w..3w+2 of the tile, i.e. three
//     groups of 16 samples, and ALL 256 features of them;
//   * every 256x256 layer is D[feature][sample] = W[feature][k] * H[k][sample] on
//     v_mfma_f32_16x16x4_f32 (exact fp32): A = weights, B = activations, C/D = 192 AGPRs;
//     weights are packed (pack.hip, kfeat order) so that a layer's D registers ARE the next
//     layer's B operands: activations never leave the register file;
//   * layer 0 and the skip half of pts_linears[5] are affine in (pixel-only) + (frame-only)
//     terms: h0 = relu(p0[pix] + q0[frm]), h5 = relu(W5b h4 + p5[pix] + q5[frm]);
//   * EVERYTHING a tile reads arrives through one LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4 from inline asm: no VGPRs, no ds_write pass): per tile 117 ring
//     steps of 16 KiB -- q0 rows, p0 rows, 80 weight slabs, q5 rows, p5 rows, 33 weight slabs --
//     issued kDepth steps ahead of consumption, across tile boundaries (the next tile's tables
//     land while the current tile's last layers run), so the kernel has no other global loads;
//   * one s_barrier per ring step: "wait own DMA quarter of step s+1 -> barrier -> refill the
//     buffer of step s".  The barrier publishes step s+1 and retires step s.
#include "s2l_common.h"

#ifndef S2L_RENDER_ASM
#define S2L_RENDER_ASM 1
#endif
#ifndef S2L_RENDER_CONV
#define S2L_RENDER_CONV 0
#endif
#ifndef S2L_RENDER_G
#define S2L_RENDER_G 3
#endif

namespace s2l {

struct RenderArgs {
  const float* packed;
  const float* p0t;   // [NPG][16 mb][4 q][16 px][4]  (s2l_pixel_tables)
  const float* p5t;
  const float* q0;    // [F][256]
  const float* q5;
  float* out;         // [F][HW][3]
  int hw, nframes;
  int npg, ntiles;    // pixel groups of 16, tiles = npg * ceil(F/12)
};

constexpr int kRing = 9;                       // 16 KiB steps resident in LDS
constexpr int kDepth = kRing - 1;              // steps in flight ahead of the one being consumed
constexpr int kSlabBytes = kSlab * 4;          // 16384
constexpr int kSlabQuads = kSlabBytes / 16;    // f4 elements per step
constexpr int kSteps = 117;                    // ring steps per tile
constexpr int kStepQ0 = 0, kStepP0 = 1, kStepW0 = 2, kStepQ5 = 82, kStepP5 = 83, kStepW5 = 84;
constexpr int kTileFrames = 12, kTilePixels = 16;
constexpr int kBiasFloats = kHidden * kW + 4;  // OFF_BIAS .. OFF_BOUT+4 are contiguous in the blob
constexpr int kLdsBytes = kRing * kSlabBytes + kBiasFloats * 4;
static_assert(OFF_WOUT == OFF_WMLP + int64_t(kHidden) * 16 * kSlab, "weight slabs must be contiguous");
static_assert(OFF_BOUT == OFF_BIAS + kHidden * kW, "bias block must be contiguous");
static_assert(kLdsBytes + 64 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One wave moves 4 x 1 KiB.  gsrc = this lane's source address of the first KiB (+ lane*16);
// lds_dst = wave-uniform LDS byte address of that KiB; the four KiB are contiguous on both sides.
__device__ __forceinline__ void dma_4k(const char* gsrc, uint32_t lds_dst) {
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

__device__ __forceinline__ void dma_1k(const char* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// ReLU as ONE v_max_f32.  fmaxf(x, 0.f) compiles to two (hipcc first canonicalises a possibly-signalling NaN with v_max x, x);
// with 192 values per wave and layer that second instruction sits on the critical path between two layers.  Same result for
// every input the hardware max accepts (max(x, 0) in IEEE mode).
__device__ __forceinline__ float relu1(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef S2L_EXP_TRACE   // experiment build only: per-tile phase timestamps (s_memtime)
__device__ long long* g_trace = nullptr;
#define S2L_TRACE(tile, slot)                                                                      \
  do {                                                                                             \
    if (g_trace && threadIdx.x == 0) g_trace[(int64_t)(tile) * 16 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define S2L_TRACE(tile, slot) do { } while (0)
#endif

template <int G>
__global__ __launch_bounds__(12 / G * 64) void render_tiles_kernel(RenderArgs a) {
  constexpr int kThreads = 12 / G * 64;   // 12 frames per tile, G frames per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  const float* lds_bias = reinterpret_cast<const float*>(smem + kRing * kSlabBytes) + 4 * q;
  const f4* ring = reinterpret_cast<const f4*>(smem);

  // ---- producer side -------------------------------------------------------------------------
  // The DMA program runs kDepth steps ahead of consumption: when the consumer leaves step c of a
  // tile, step c+kRing is issued into the buffer that c occupied (steps >= kSteps belong to this
  // workgroup's next tile).  Which step that is follows from the consumer's position, so the hot
  // loop needs no program counter and no branches: only 4 of the 117 issues per tile are tables.
  const char* wsrc = reinterpret_cast<const char*>(a.packed + OFF_WMLP) + wave * 4096 + lane * 16;
  auto dst_of = [&](int buf) { return lds_base + buf * kSlabBytes + wave * 4096; };
  const bool mover = wave < 4;   // the ring is filled by the first four waves (one per SIMD), 4 KiB each per step
  auto issue_w = [&](int ws, int buf) { if (mover) dma_4k(wsrc + (int64_t)ws * kSlabBytes, dst_of(buf)); };
  auto issue_q = [&](const float* qtab, int tile, int buf) {
    // 16 rows of q (12 used): row r = frame fg*12 + r, clamped; this wave moves rows 4w..4w+3
    if (!mover) return;
    const char* qb = reinterpret_cast<const char*>(qtab) + lane * 16;
    const int f0 = (tile / a.npg) * kTileFrames + wave * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = f0 + i;
      f = f < a.nframes ? f : a.nframes - 1;
      dma_1k(qb + (int64_t)f * 1024, dst_of(buf) + i * 1024);
    }
  };
  auto issue_p = [&](const float* ptab, int tile, int buf) {
    if (mover) dma_4k(reinterpret_cast<const char*>(ptab) + wave * 4096 + lane * 16 + (int64_t)(tile % a.npg) * kSlabBytes,
           dst_of(buf));
  };
  // prime: steps 0..kDepth-1 of the first tile = q0, p0, weight slabs 0..kDepth-3
  issue_q(a.q0, blockIdx.x, 0);
  issue_p(a.p0t, blockIdx.x, 1);
#pragma unroll
  for (int s = 2; s < kDepth; ++s) issue_w(s - kStepW0, s);

  // biases -> LDS (ordinary loads; the ring is not read before the first barrier)
  for (int i = threadIdx.x; i < kBiasFloats; i += kThreads)
    reinterpret_cast<float*>(smem + kRing * kSlabBytes)[i] = a.packed[OFF_BIAS + i];

  // ---- consumer side ---------------------------------------------------------------------------
  int cur = 0;   // ring buffer of the step being consumed
  // Publish the next step and retire the current one; returns the retired buffer, which the
  // caller refills.  Precondition: every LDS read this wave issued on the current step has
  // returned (callers wait lgkmcnt(0) first).
  auto advance = [&]() {
    wait_vmcnt<4 * (kDepth - 1)>();          // own quarter of the next step has landed
    asm volatile("s_barrier" ::: "memory");  // everyone's quarter landed; everyone left the current step
    const int retired = cur;
    cur = cur + 1 == kRing ? 0 : cur + 1;
    return retired;
  };
  auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  // weight slab issued when the consumer leaves step c (c + kRing is a weight step)
  auto ws_after = [&](int c) {
    const int i = c + kRing;
    return i - kStepW0 - (i >= kStepW5 ? 2 : 0) - (i >= kSteps ? kSteps - 2 : 0);
  };

  float in[G][64];
  f4 acc[G][16];

  // in[g] = base[g] (+ acc) + q rows of the three frames this wave owns; then + p rows and ReLU.
  // Step layout: q step = 16 rows x 1 KiB; p step = [mb][q][px] f4, i.e. lane-linear per M-block.
  auto add_q = [&](bool with_acc) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f4* row = ring + cur * kSlabQuads + (wave * G + g) * 64 + q;
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) {
        const f4 v = row[mb * 4];
#pragma unroll
        for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = with_acc ? acc[g][mb][r] + v[r] : v[r];
      }
    }
  };
  auto add_p_relu = [&]() {
    const f4* pl = ring + cur * kSlabQuads + lane;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const f4 v = pl[mb * 64];
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = relu1(in[g][mb * 4 + r] + v[r]);
    }
  };

  // first step (q0 of the first tile) landed and published (and the bias block written by every
  // wave); top the ring up to kDepth in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vmcnt<4 * (kDepth - 1)>();
  asm volatile("s_barrier" ::: "memory");
  issue_w(kDepth - kStepW0, kDepth);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    S2L_TRACE(tile, 0);
    const int next_tile = tile + (int)gridDim.x < a.ntiles ? tile + (int)gridDim.x : a.ntiles - 1;
    // h0 = relu(p0[pixel] + q0[frame])
    add_q(false);
    lgkm0();
    issue_w(ws_after(kStepQ0), advance());
    add_p_relu();
    lgkm0();
    issue_w(ws_after(kStepP0), advance());
    S2L_TRACE(tile, 1);

    // A-operand quads are prefetched two ahead into three rotating register sets
    f4 w0 = ring[cur * kSlabQuads + lane], w1 = ring[cur * kSlabQuads + 64 + lane];

    for (int layer = 0; layer < kHidden; ++layer) {
      const int cbase = kStepW0 + 16 * layer + (layer >= 5 ? 2 : 0);   // ring step of this layer's first slab
      // One slab = one M-block of 16 output features: 16 k-quads x 4 k-steps x G MFMAs.  `convert` (last slab of a layer whose
      // output goes straight to the next layer): once k-quad j has been issued, block j of `in` is dead and block j of the
      // accumulators has been final since slab j, so in[.][4j..4j+3] = relu(acc[.][j]) is written in place, one value behind
      // each MFMA of k-quad j+1 -- the accumulator reads and the v_max ride in the shadow of the matrix pipe instead of standing
      // between two layers.  Only block 15 (which this slab produces) is left for the end.
      auto slab = [&](const int mb, const bool convert) {
        const f4* sl = ring + cur * kSlabQuads + lane;
        {
          const f4 b = *reinterpret_cast<const f4*>(lds_bias + layer * kW + mb * 16);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g][mb] = b;
        }
#pragma unroll
        for (int j4 = 0; j4 < 16; ++j4) {
          if (j4 == 14) {
            // quads 14 and 15 are in registers: done reading this slab.  The next two reads come
            // from the next step (garbage, and discarded, when that step is not a weight slab).
            lgkm0();
            const int buf = advance();
            sl = ring + cur * kSlabQuads + lane - 16 * 64;
            // refill: a weight slab, except the four table steps per tile (static positions)
            if ((mb == 7 || mb == 8) && layer == 4) {
              if (mb == 7) issue_q(a.q5, tile, buf); else issue_p(a.p5t, tile, buf);
            } else if ((mb == 8 || mb == 9) && layer == 6) {
              if (mb == 8) issue_q(a.q0, next_tile, buf); else issue_p(a.p0t, next_tile, buf);
            } else {
              issue_w(ws_after(cbase + mb), buf);
            }
          }
          const f4 w2 = sl[(j4 + 2) * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < G; ++g) {
              acc[g][mb] = mfma16(w0[jj], in[g][j4 * 4 + jj], acc[g][mb]);
              if (convert && j4 >= 1) {
                in[g][(j4 - 1) * 4 + jj] = relu1(acc[g][j4 - 1][jj]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
          w0 = w1;
          w1 = w2;
        }
      };
#pragma unroll
      for (int mb = 0; mb < 15; ++mb) slab(mb, false);
      slab(15, S2L_RENDER_CONV != 0);   // (for layer 4 the converted values are overwritten below: harmless, and free in the shadow)
      if (layer == 4) {
        // pts_linears[5] on cat([skip, h4]): + q5[frame] + p5[pixel] (q5 carries b5; bias row 4 is 0)
        add_q(true);
        lgkm0();
        issue_w(ws_after(kStepQ5), advance());
        add_p_relu();
        lgkm0();
        issue_w(ws_after(kStepP5), advance());
        w0 = ring[cur * kSlabQuads + lane];
        w1 = ring[cur * kSlabQuads + 64 + lane];
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int mb = S2L_RENDER_CONV ? 15 : 0; mb < 16; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = relu1(acc[g][mb][r]);
      }
      S2L_TRACE(tile, 2 + layer);
    }

    // output_linear (3 rows zero-padded to one M-block); no activation (tf_nerf.py:283)
    f4 rgb[G];
    {
      const f4 b = *reinterpret_cast<const f4*>(smem + kRing * kSlabBytes + kHidden * kW * 4);
#pragma unroll
      for (int g = 0; g < G; ++g) rgb[g] = b;
      const f4* sl = ring + cur * kSlabQuads + lane;
#pragma unroll
      for (int j4 = 0; j4 < 16; ++j4) {
        f4 w2 = w1;
        if (j4 < 14) w2 = sl[(j4 + 2) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int g = 0; g < G; ++g) rgb[g] = mfma16(w0[jj], in[g][j4 * 4 + jj], rgb[g]);
        __builtin_amdgcn_sched_barrier(0);
        w0 = w1;
        w1 = w2;
      }
      lgkm0();
      issue_w(ws_after(kSteps - 1), advance());   // publishes the next tile's q0 step
    }
    const int pixel = (tile % a.npg) * kTilePixels + px;
    const int frame0 = (tile / a.npg) * kTileFrames + wave * G;
    if (q == 0 && pixel < a.hw) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (frame0 + g < a.nframes) {
          float* o = a.out + ((int64_t)(frame0 + g) * a.hw + pixel) * 3;
          o[0] = rgb[g][0];
          o[1] = rgb[g][1];
          o[2] = rgb[g][2];
        }
      }
    }
    S2L_TRACE(tile, 9);
  }
  wait_vmcnt<0>();   // run-ahead (dummy) DMAs must land before the workgroup's LDS is released
}

#if S2L_RENDER_ASM
// The same kernel with its body as one fixed-register assembly text (csrc/gen_render_body.py explains why and how).
__global__ __launch_bounds__(256) void render_tiles_asm_kernel(RenderArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < kBiasFloats; i += 256)
    reinterpret_cast<float*>(smem + kRing * kSlabBytes)[i] = a.packed[OFF_BIAS + i];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
  const int grid = (int)gridDim.x, tile0 = (int)blockIdx.x;
  const int gdiv = __builtin_amdgcn_readfirstlane(grid / a.npg), gmod = __builtin_amdgcn_readfirstlane(grid % a.npg);
  const int fg0 = __builtin_amdgcn_readfirstlane(tile0 / a.npg), pg0 = __builtin_amdgcn_readfirstlane(tile0 % a.npg);
  const int fgl = __builtin_amdgcn_readfirstlane((a.ntiles - 1) / a.npg), pgl = __builtin_amdgcn_readfirstlane((a.ntiles - 1) % a.npg);
  const float* wsrc = a.packed + OFF_WMLP;
  const uint32_t lane16 = lds0 + lane * 16, dmaoff = wave * 4096 + lane * 16;
  const uint32_t biasaddr = lds0 + kRing * kSlabBytes + 16 * q, boutaddr = lds0 + kRing * kSlabBytes + kHidden * kW * 4;
  const uint32_t qaddr = lds0 + wave * 3072 + q * 16;
#include "render_body.inc"
}
#endif

#ifdef S2L_EXP_TRACE
extern "C" int s2l_debug_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }
#endif

}  // namespace s2l

// Workgroups the persistent renderer may occupy on each device (0 = one per CU).  A multi-GPU host that overlaps the RCCL
// all-gather of chunk c with the render of chunk c+1 sets this to CUs - k on its device: the renderer fills a CU completely
// (151 KiB LDS, every register), so RCCL's workgroups can only run on CUs it leaves free.  Per device, atomics: a host with
// one thread per GPU can set and use its own limit without racing the others.
static std::atomic<int> g_render_cu_limit[s2l::kMaxDevices];
extern "C" int s2l_set_render_cus(int n_workgroups) {
  if (n_workgroups < 0) return S2L_E_SIZE;
  int dev = 0, n_cu = 0;
  const int rc = s2l::current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  g_render_cu_limit[dev].store(n_workgroups, std::memory_order_relaxed);
  return S2L_OK;
}

extern "C" int s2l_render_lip(const float* packed, const float* p0, const float* p5, const float* q0, const float* q5,
                              float* out, int64_t hw, int64_t n_frames, s2l_stream_t stream) {
  using namespace s2l;
  if (hw <= 0 || hw > (1 << 24) || n_frames < 0 || n_frames > (1 << 24)) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!packed || !p0 || !p5 || !q0 || !q5 || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(p0) || misaligned16(p5) || misaligned16(q0) || misaligned16(q5))
    return S2L_E_ALIGN;
  constexpr int G = S2L_RENDER_G;
  RenderArgs a;
  a.packed = packed; a.p0t = p0; a.p5t = p5; a.q0 = q0; a.q5 = q5; a.out = out;
  a.hw = (int)hw; a.nframes = (int)n_frames;
  a.npg = (int)((hw + kTilePixels - 1) / kTilePixels);
  const int64_t ntiles = (int64_t)a.npg * ((n_frames + kTileFrames - 1) / kTileFrames);
  if (ntiles > 0x7fffffff) return S2L_E_SIZE;
  a.ntiles = (int)ntiles;

  // per-device one-time setup (CU count, >64 KiB dynamic-LDS opt-in), thread-safe: s2l_common.h
  static LdsOptIn lds_flags;
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&render_tiles_kernel<G>), kLdsBytes, lds_flags, dev))) return rc;
  const int limit = g_render_cu_limit[dev].load(std::memory_order_relaxed);
  if (limit > 0 && limit < n_cu) n_cu = limit;
  // persistent: one workgroup per CU (151 KiB of LDS and 4 x 512 registers fill a CU)
  const int grid = a.ntiles < n_cu ? a.ntiles : n_cu;
#if S2L_RENDER_ASM
  static LdsOptIn lds_flags_asm;
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&render_tiles_asm_kernel), kLdsBytes + 64, lds_flags_asm, dev))) return rc;
  hipLaunchKernelGGL(render_tiles_asm_kernel, dim3(grid), dim3(256), kLdsBytes + 64, static_cast<hipStream_t>(stream), a);
#else
  hipLaunchKernelGGL((render_tiles_kernel<G>), dim3(grid), dim3(12 / G * 64), kLdsBytes, static_cast<hipStream_t>(stream), a);
#endif
  return (int)hipGetLastError();
}
