// Fused clip renderer (the hot kernel): s2l_render_lip.
//
// Replaces the per-frame driver loop inference.py:140-159 and TalkingFace.rgb_forward
// (tf_nerf.py:225-285) for whole clips: one persistent launch renders every pixel of every frame.
//
//   * tile = 16 consecutive pixels x 12 consecutive frames = 192 samples; a workgroup = 4 waves
//     (one per SIMD, 512 registers each); wave w owns frames 3w..3w+2 of the tile, i.e. three
//     groups of 16 samples, and ALL 256 features of them;
//   * every 256x256 layer is D[feature][sample] = W[feature][k] * H[k][sample] on
//     v_mfma_f32_16x16x4_f32 (exact fp32): A = weights, B = activations (192 VGPRs), C/D = 192 AGPRs;
//     weights are packed (pack.hip, kfeat order) so that a layer's D registers ARE the next
//     layer's B operands: activations never leave the register file;
//   * layer 0 and the skip half of pts_linears[5] are affine in (pixel-only) + (frame-only)
//     terms: h0 = relu(p0[pix] + q0[frm]), h5 = relu(W5b h4 + p5[pix] + q5[frm]);
//   * EVERYTHING a tile reads arrives through one LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no ds_write pass): per tile 117 ring steps of 16 KiB --
//     q0 rows, p0 rows, 80 weight slabs, q5 rows, p5 rows, 33 weight slabs -- issued 8 steps ahead of
//     consumption, across tile boundaries (the next tile's tables land while the current tile's last
//     layers run), so the kernel has no other global loads;
//   * one s_barrier per ring step: "wait own DMA quarter of step s+1 -> barrier -> refill the
//     buffer of step s".  The barrier publishes step s+1 and retires step s.
//
// The kernel BODY is one fixed-register assembly text, written at build time by csrc/gen_render_body.py (which also says
// why: with one wave per SIMD, VALU instructions never overlap the wave's own MFMAs, so the schedule has to be owned, not
// suggested).  The C++ below only unpacks the arguments.  The C++ version of the same loop, kept until commit 62292d8,
// produced the same bits at 0.905-0.915 of the fp32-MFMA peak; this one runs at 0.94-0.95 (the rest is mostly clock: at full
// load the part settles at 2.29-2.33 GHz, 0.975 of the cycles are MFMA cycles).
#include "s2l_common.h"

namespace s2l {

struct RenderArgs {
  const float* packed;
  const float* p0t;   // [NPG][16 mb][4 q][16 px][4]  (s2l_pixel_tables)
  const float* p5t;
  const float* q0;    // [F][256]
  const float* q5;
  float* out;         // [F][HW][3]
  int hw, nframes;
  int npg, ntiles;    // pixel groups of 16 in the image; tiles = ceil(npg / PGT) * nfg
  int nfg;            // frame groups = ceil(F / FT)
};

// Tile shapes (csrc/gen_render_body.py, VARIANTS): 4 waves x G groups of 16 samples = PGT pixel groups x FT frames.
//   kLong   3 groups, 1 x 12: the shape for clips (its generated text is pinned);
//   kWide   3 groups, 12 x 1: one frame per tile -- no wasted frame slots when F is not a multiple of 12;
//   kSingle 1 group,   4 x 1: 64 samples per tile, so that ONE frame spreads over the whole chip (the reference's per-frame mode).
// Frames are bit-identical whichever shape rendered them (every sample column sees the same MFMA sequence).
//   kFeat   the feature-split tile (render_fs_kernel, gen_render_fs_body.py): 16 samples per tile, the four waves own 64 features each and
//           exchange the activations through LDS: a tile is 7 x 256 MFMAs per wave instead of 7 x 1024 -- ONE frame in a third of the time.
enum RenderShape { kLong = 0, kWide = 1, kSingle = 2, kFeat = 3 };
struct ShapeDims { int g, pgt, ft; };
__host__ __device__ constexpr ShapeDims shape_dims(int shape) {
  return shape == kLong ? ShapeDims{3, 1, 12} : shape == kWide ? ShapeDims{3, 12, 1} : shape == kSingle ? ShapeDims{1, 4, 1} : ShapeDims{1, 1, 1};
}

constexpr int kRing = 9;                       // 16 KiB steps resident in LDS (gen_render_body.py: KRING)
constexpr int kSlabBytes = kSlab * 4;          // 16384
constexpr int kTilePixels = 16;
constexpr int kBiasFloats = kHidden * kW + 4;  // OFF_BIAS .. OFF_BOUT+4 are contiguous in the blob
// ring + bias block + 76 B (the body prefetches "the next slab's bias" once past the block's end: unused values; the scratch
// below is 16-byte aligned) + 1 KiB of scratch per wave (accumulators -> B registers between two layers)
constexpr int kScratchOff = kRing * kSlabBytes + (kBiasFloats * 4 + 64 + 15) / 16 * 16;
constexpr int kLdsBytes = kScratchOff + 4 * 1024;
static_assert(OFF_WOUT == OFF_WMLP + int64_t(kHidden) * 16 * kSlab, "weight slabs must be contiguous");
static_assert(OFF_BOUT == OFF_BIAS + kHidden * kW, "bias block must be contiguous");
static_assert(kHidden == 7 && kSlabBytes == 16384, "gen_render_body.py is written for 7 MFMA layers of 16 KiB slabs");
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

#ifdef S2L_EXP_TRACE   // experiment build only (tools/trace_tiles.py, generator run with S2L_RENDER_TRACE=1): per-tile timestamps
__device__ long long* g_trace = nullptr;
extern "C" int s2l_debug_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }
#endif

template <int SHAPE>
__global__ __launch_bounds__(256) void render_tiles_kernel(RenderArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // biases -> LDS (ordinary stores; the body waits for them before its first barrier)
  for (int i = threadIdx.x; i < kBiasFloats; i += 256)
    reinterpret_cast<float*>(smem + kRing * kSlabBytes)[i] = a.packed[OFF_BIAS + i];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;      // k-subgroup and sample of this lane in every 16x16x4 MFMA
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);   // this wave's quarter of ring buffer 0
  // this workgroup's tiles: a contiguous range in pixel-group-major order, tile t = (pixel-group block t / nfg, frame group t % nfg)
  const int nfg = __builtin_amdgcn_readfirstlane(a.nfg);
  const int npgm1 = __builtin_amdgcn_readfirstlane(a.npg - 1);      // (the wide shapes clamp their table rows to the last group)
  (void)npgm1;
  const int tile0 = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * blockIdx.x / gridDim.x));
  const int tile_end = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * (blockIdx.x + 1) / gridDim.x));
  const int fg0 = __builtin_amdgcn_readfirstlane(tile0 % nfg), pg0 = __builtin_amdgcn_readfirstlane(tile0 / nfg);
  const float* wsrc = a.packed + OFF_WMLP;
  const uint32_t lane16 = lds0 + lane * 16;                                  // A quads and p rows are lane-linear
  const uint32_t dmaoff = wave * 4096 + lane * 16;                           // this lane's 16 B of a 16 KiB step
  const uint32_t biasaddr = lds0 + kRing * kSlabBytes + 16 * q;              // bias of features 16 mb + 4 q .. + 3
  const uint32_t boutaddr = lds0 + kRing * kSlabBytes + kHidden * kW * 4;    // output-layer bias (rows >= 3 are zero)
  // q rows: the long shape's wave owns frames (rows) 3 wave + g; the one-frame shapes read row 0
  const uint32_t qaddr = lds0 + (SHAPE == kLong ? wave * 3072 : 0) + q * 16;
  const uint32_t scraddr = lds0 + kScratchOff + wave * 1024 + lane * 16;     // this wave's scratch
  if constexpr (SHAPE == kLong) {
#include "render_body.inc"
  } else if constexpr (SHAPE == kWide) {
#include "render_body_wide.inc"
  } else {
#include "render_body_single.inc"
  }
}

// ---- the feature-split tile (csrc/gen_render_fs_body.py has the design): 16 samples per tile; wave w owns features 64 w .. 64 w + 63
constexpr int kFsRingPerWave = 8 * 4096;                       // the wave's private ring of eight 4-KiB pieces of its own slabs
constexpr int kFsExchange = 4 * kFsRingPerWave;                // two 16-KiB activation blocks [M-block][lane][4] behind the rings
constexpr int kFsLdsBytes = kFsExchange + 2 * 16384;
static_assert(kFsLdsBytes == 160 * 1024, "gen_render_fs_body.py: LDS_BYTES");
__global__ __launch_bounds__(256) void render_fs_kernel(RenderArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * kFsRingPerWave);
  const int nfg = __builtin_amdgcn_readfirstlane(a.nfg);
  const int tile0 = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * blockIdx.x / gridDim.x));
  const int tile_end = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * (blockIdx.x + 1) / gridDim.x));
  const int fg0 = __builtin_amdgcn_readfirstlane(tile0 % nfg), pg0 = __builtin_amdgcn_readfirstlane(tile0 / nfg);
  const float* wb = a.packed + OFF_WMLP + (int64_t)wave * 4 * kSlab;      // the wave's four slabs of layer 0 (64 contiguous KiB per layer)
  const float* wout = a.packed + OFF_WOUT;
  const float* biasp = a.packed + OFF_BIAS;
#include "render_fs_body.inc"
}

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

// 0 = choose per call (below), 1 + shape = always that shape (tests, A/B runs)
static std::atomic<int> g_render_shape{0};
extern "C" int s2l_set_render_shape(int mode) {
  if (mode < 0 || mode > 4) return S2L_E_SIZE;
  g_render_shape.store(mode, std::memory_order_relaxed);
  return S2L_OK;
}

// (shared with the split-half renderer of render16.hip)
extern "C" int s2l_render_shape_mode(void) { return g_render_shape.load(std::memory_order_relaxed); }
extern "C" int s2l_render_cu_limit(int dev) { return (dev >= 0 && dev < s2l::kMaxDevices) ? g_render_cu_limit[dev].load(std::memory_order_relaxed) : 0; }

// The shape with the smallest estimated time on n_cu CUs: rounds of tiles over the persistent grid x the cost of a tile (a G = 1
// tile streams the same 113 weight slabs for a third of the MFMAs: ~0.36 of a G = 3 tile; the wide shape's 22 extra table steps
// cost ~3 %).  Ties go to the long shape (pinned text, least table traffic).
static int pick_render_shape(int64_t npg, int64_t n_frames, int n_cu) {
  double best = 0;
  int pick = s2l::kLong;
  // (measured, in units of the long tile's ~306 us: a single tile 0.38 at one frame per call; a feature-split tile 34 us while a workgroup
  //  has <= 3 of them, 42 us in steady state -- 256 CUs x 4 waves then stream 1.8 MB of weights per tile from L2 side by side with the MFMAs)
  const double cost[4] = {1.0, 1.03, 0.36, 0.112};
  for (int shp = 0; shp < 4; ++shp) {
    const s2l::ShapeDims d = s2l::shape_dims(shp);
    const int64_t tiles = ((npg + d.pgt - 1) / d.pgt) * ((n_frames + d.ft - 1) / d.ft);
    // a persistent workgroup owns a contiguous range of ceil / floor(tiles / grid) tiles: the longest range sets the time
    const int64_t grid = tiles < n_cu ? tiles : n_cu;
    const int64_t rounds = (tiles + grid - 1) / grid;
    const double t = shp == s2l::kFeat && rounds > 3 ? 3 * cost[shp] + (double)(rounds - 3) * 0.137 : (double)rounds * cost[shp];
    if (shp == 0 || t < best * 0.97) best = t, pick = shp;
  }
  return pick;
}

extern "C" int s2l_render_lip(const float* packed, const float* p0, const float* p5, const float* q0, const float* q5,
                              float* out, int64_t hw, int64_t n_frames, s2l_stream_t stream) {
  using namespace s2l;
  if (hw <= 0 || hw > (1 << 24) || n_frames < 0 || n_frames > (1 << 24)) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!packed || !p0 || !p5 || !q0 || !q5 || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(p0) || misaligned16(p5) || misaligned16(q0) || misaligned16(q5))
    return S2L_E_ALIGN;
  RenderArgs a;
  a.packed = packed; a.p0t = p0; a.p5t = p5; a.q0 = q0; a.q5 = q5; a.out = out;
  a.hw = (int)hw; a.nframes = (int)n_frames;
  a.npg = (int)((hw + kTilePixels - 1) / kTilePixels);

  // per-device one-time setup (CU count, >64 KiB dynamic-LDS opt-in), thread-safe: s2l_common.h
  static LdsOptIn lds_flags[4];
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  const int limit = g_render_cu_limit[dev].load(std::memory_order_relaxed);
  if (limit > 0 && limit < n_cu) n_cu = limit;
  const int forced = g_render_shape.load(std::memory_order_relaxed);
  const int shape = forced ? forced - 1 : pick_render_shape(a.npg, n_frames, n_cu);
  const ShapeDims d = shape_dims(shape);
  a.nfg = (int)((n_frames + d.ft - 1) / d.ft);
  const int64_t ntiles = (int64_t)((a.npg + d.pgt - 1) / d.pgt) * a.nfg;
  if (ntiles > 0x7fffffff) return S2L_E_SIZE;
  a.ntiles = (int)ntiles;
  void (*const kern[4])(RenderArgs) = {render_tiles_kernel<kLong>, render_tiles_kernel<kWide>, render_tiles_kernel<kSingle>, render_fs_kernel};
  const int lds_bytes = shape == kFeat ? kFsLdsBytes : kLdsBytes;
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern[shape]), lds_bytes, lds_flags[shape], dev))) return rc;
  // persistent: one workgroup per CU (151 - 160 KiB of LDS and 4 x 512 registers fill a CU)
  const int grid = a.ntiles < n_cu ? a.ntiles : n_cu;
  hipLaunchKernelGGL(kern[shape], dim3(grid), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
