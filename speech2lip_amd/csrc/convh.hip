// The 3x3 convolution of the U-Net's training chain on HALF-WIDTH tensors (bf16 NHWC activations and gradients in HBM) as a
// generated-assembly kernel: csrc/gen_convh_body.py has the design, the register map and the schedule.  Replaces, inside
// `SimpleUnetLight.forward` / its autograd backward in TRAIN mode (SimpleUnetLight.py:16-111 applied at tf_nerf.py:387 by the frozen
// net of training.py:436-459 after `it > 100000`), one `nn.Conv2d(3x3, padding=1, bias=False)` or its input gradient.
#include "s2l_common.h"
#include "convh.h"
#include <atomic>

namespace s2l {

static_assert(offsetof(ConvHArgs, inA) == 0 && offsetof(ConvHArgs, inB) == 8 && offsetof(ConvHArgs, w16) == 16 &&
              offsetof(ConvHArgs, bias) == 24 && offsetof(ConvHArgs, out) == 32 && offsetof(ConvHArgs, gate) == 40 &&
              offsetof(ConvHArgs, CA) == 48 && offsetof(ConvHArgs, CB) == 52 && offsetof(ConvHArgs, cout) == 56 &&
              offsetof(ConvHArgs, H) == 60 && offsetof(ConvHArgs, W) == 64 && offsetof(ConvHArgs, tiles_x) == 68 &&
              offsetof(ConvHArgs, tiles_y) == 72 && offsetof(ConvHArgs, n_ct) == 76 && offsetof(ConvHArgs, relu) == 80 &&
              offsetof(ConvHArgs, stat) == 88 && offsetof(ConvHArgs, norm) == 96 && offsetof(ConvHArgs, bz) == 104 && offsetof(ConvHArgs, bst) == 112,
              "gen_convh_body.py (ARG) loads these fields from the kernarg segment by offset");

constexpr int kCHTileH = 32;      // tile = 32 rows x 16 columns (gen_convh_body.py: TILE_H)
constexpr int kCHHalo = (kCHTileH + 2) * 18 * 64, kCHW = 9 * 2 * 2 * 64 * 16, kCHBuf = kCHHalo + kCHW;
constexpr int kCHLds = 2 * kCHBuf + 1024 + 4096 + 2048;      // two buffers + the bias table + the eight waves' per-tile statistics + the normalising form's two [scale | shift] tables (the store staging aliases buffer 1)
static_assert(kCHLds <= 160 * 1024, "LDS budget");

#ifdef S2L_WITH_REFERENCE_KERNELS      // (the four-wave form: libs2l_hip_ref.so only)
__global__ __launch_bounds__(256) void convh_asm_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ch_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();   // the body loads the ConvHArgs fields itself (s_load)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ch_smem);
  // this workgroup's tiles: a contiguous range, tile t = ((frame * n_ct + ct) * tiles_y + ty) * tiles_x + tx
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * a.n_ct * a.n_frames;
  const int tile0 = (int)(total * blockIdx.x / gridDim.x), tile_end = (int)(total * (blockIdx.x + 1) / gridDim.x);
  if (tile0 >= tile_end) return;
  int t = tile0;
  const int tx0 = __builtin_amdgcn_readfirstlane(t % a.tiles_x);
  t /= a.tiles_x;
  const int ty0 = __builtin_amdgcn_readfirstlane(t % a.tiles_y);
  t /= a.tiles_y;
  const int ct0 = __builtin_amdgcn_readfirstlane(t % a.n_ct);
  const int fr0 = __builtin_amdgcn_readfirstlane(t / a.n_ct);
  const int ntl = __builtin_amdgcn_readfirstlane(tile_end - tile0);
  // per-lane constants; they reach the assembly body through LDS ([word 28][thread 256] at the start of buffer 0; the body reads them
  // first).  Halo DMA instruction i of wave w fills slots (10 w + i) * 64 + lane: slot s = 16 bytes at s * 16 of the halo area = pixel
  // s >> 2 (row-major, 18 columns), physical segment s & 3 = logical segment (channels 8 seg .. + 7 of the chunk) ^ ((col >> 2) & 3).
  uint32_t* cst = reinterpret_cast<uint32_t*>(ch_smem);
  constexpr int kSlots = kCHHalo / 16;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int sl = (wave * 10 + i) * 64 + lane;
    const int pi = sl >> 2, row = pi / 18, col = pi % 18, seg = (sl & 3) ^ ((col >> 2) & 3);
    cst[i * 256 + tid] = sl < kSlots ? (uint32_t)(col | (row << 8) | (seg << 16)) : 0x80000000u;
  }
  // Which pixel of its N-block (2 rows x 16 columns) lane n = lane & 31 stands for follows the LDS's lane groups: a ds_read_b128 is
  // serviced in four NON-contiguous groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32; MI355X_MICROARCH.md
  // LDS table) -- the lanes of one group read the sixteen pixels of ONE row, which the swizzle below spreads over all 64 banks for
  // every tap offset (with n -> (row n >> 4, column n & 15) a group straddled both rows: 40 % of the operand reads' LDS cycles were
  // bank conflicts, SQ_LDS_BANK_CONFLICT)
  const int n_ = lane & 31, hh_ = lane >> 5;
  const int in_g0 = (n_ < 4) || (n_ >= 12 && n_ < 16) || (n_ >= 20 && n_ < 28);
  const int prow = in_g0 ? 0 : 1;
  const int pcol = in_g0 ? (n_ < 4 ? n_ : n_ < 16 ? n_ - 8 : n_ - 12) : (n_ < 12 ? n_ - 4 : n_ < 20 ? n_ - 8 : n_ - 16);
  // operand reads: pixel (row 8 wave + prow [+ 2 blk + dy as an immediate], col pcol + dx), 16-byte segment (2 ks + hh) ^ swizzle
  {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int col = pcol + dx, row = 8 * wave + prow;
        cst[(10 + dx * 2 + ks) * 256 + tid] = lds0 + (uint32_t)((row * 18 + col) * 64 + (((2 * ks + hh_) ^ ((col >> 2) & 3)) << 4));
      }
  }
  // store staging (this wave's 4 KiB at the start of buffer 1: [M-block 2][32 pixels][32 channels] bf16, pixel = 16 prow + pcol, a
  // pixel's 16-byte piece index ^ ((pixel >> 1) & 3)): write address of piece pc = 4 mb + rq (channels 8 rq + 4 hh .. of M-block mb:
  // 8 bytes; the sixteen lanes of a ds_write_b64 group land 2-way on the 32 write banks, the minimum for 8-byte pieces of 64-byte
  // pixels), read address of store j = (row j >> 1 of the N-block, M-block j & 1): pixel 16 (j >> 1) + (lane >> 2), piece lane & 3
  {
    const uint32_t stg = lds0 + kCHBuf + wave * 4096;
    const int pix = 16 * prow + pcol;
#pragma unroll
    for (int pc = 0; pc < 8; ++pc)
      cst[(16 + pc) * 256 + tid] = stg + (uint32_t)((pc >> 2) * 2048 + pix * 64 + (((pc & 3) ^ ((pix >> 1) & 3)) << 4) + hh_ * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = 16 * (j >> 1) + (lane >> 2);
      cst[(24 + j) * 256 + tid] = stg + (uint32_t)((j & 1) * 2048 + px * 64 + (((lane & 3) ^ ((px >> 1) & 3)) << 4));
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the constants are in LDS (each lane reads back only its own words)
#include "convh_body.inc"
}

#endif

// The eight-wave form (gen_convh8_body.py): two waves per SIMD, wave w owns rows 4 w .. 4 w + 3 of the tile.  Same arithmetic, same bits.
// Two bodies: linear (the train-mode chain) and max(0, .) before the conversion (the eval-mode chain's folded BatchNorm + ReLU).
struct ConvH8Ctx {
  int tx0, ty0, ct0, fr0, ntl, wave;
  uint32_t lds0;
};
template <int STAGE_OFF = kCHBuf>      // where the eight waves' 4-KiB store-staging blocks start (convh8: buffer 1's halo area)
__device__ __forceinline__ bool convh8_prologue(const ConvHArgs& a, char* smem, ConvH8Ctx& c) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * a.n_ct * a.n_frames;
  const int tile0 = (int)(total * blockIdx.x / gridDim.x), tile_end = (int)(total * (blockIdx.x + 1) / gridDim.x);
  if (tile0 >= tile_end) return false;
  int t = tile0;
  // (tile t = ((frame * tiles_y + ty) * tiles_x + tx) * n_ct + ct: the channel tiles of one position are consecutive)
  c.ct0 = __builtin_amdgcn_readfirstlane(t % a.n_ct);
  t /= a.n_ct;
  c.tx0 = __builtin_amdgcn_readfirstlane(t % a.tiles_x);
  t /= a.tiles_x;
  c.ty0 = __builtin_amdgcn_readfirstlane(t % a.tiles_y);
  c.fr0 = __builtin_amdgcn_readfirstlane(t / a.tiles_y);
  c.ntl = __builtin_amdgcn_readfirstlane(tile_end - tile0);
  c.wave = wave;
  c.lds0 = lds0;
  // per-lane constants ([word 23][thread 512] at the start of buffer 0): halo DMA instruction i of wave w fills slots (5 w + i) * 64 + lane
  uint32_t* cst = reinterpret_cast<uint32_t*>(smem);
  constexpr int kSlots = kCHHalo / 16;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int sl = (wave * 5 + i) * 64 + lane;
    const int pi = sl >> 2, row = pi / 18, col = pi % 18, seg = (sl & 3) ^ ((col >> 2) & 3);
    cst[i * 512 + tid] = sl < kSlots ? (uint32_t)(col | (row << 8) | (seg << 16)) : 0x80000000u;
  }
  const int n_ = lane & 31, hh_ = lane >> 5;      // (lane -> pixel of the N-block by ds_read_b128's lane groups: see convh_asm_kernel)
  const int in_g0 = (n_ < 4) || (n_ >= 12 && n_ < 16) || (n_ >= 20 && n_ < 28);
  const int prow = in_g0 ? 0 : 1;
  const int pcol = in_g0 ? (n_ < 4 ? n_ : n_ < 16 ? n_ - 8 : n_ - 12) : (n_ < 12 ? n_ - 4 : n_ < 20 ? n_ - 8 : n_ - 16);
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int col = pcol + dx, row = 4 * wave + prow;
      cst[(5 + dx * 2 + ks) * 512 + tid] = lds0 + (uint32_t)((row * 18 + col) * 64 + (((2 * ks + hh_) ^ ((col >> 2) & 3)) << 4));
    }
  const uint32_t stg = lds0 + STAGE_OFF + wave * 4096;      // 8 x 4 KiB
  const int pix = 16 * prow + pcol;
#pragma unroll
  for (int pc = 0; pc < 8; ++pc)
    cst[(11 + pc) * 512 + tid] = stg + (uint32_t)((pc >> 2) * 2048 + pix * 64 + (((pc & 3) ^ ((pix >> 1) & 3)) << 4) + hh_ * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int px = 16 * (j >> 1) + (lane >> 2);
    cst[(19 + j) * 512 + tid] = stg + (uint32_t)((j & 1) * 2048 + px * 64 + (((lane & 3) ^ ((px >> 1) & 3)) << 4));
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  return true;
}
__global__ __launch_bounds__(512) void convh8_asm_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ch_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  ConvH8Ctx c;
  if (!convh8_prologue(a, ch_smem, c)) return;
  const int tid = threadIdx.x, wave = c.wave, tx0 = c.tx0, ty0 = c.ty0, ct0 = c.ct0, fr0 = c.fr0, ntl = c.ntl;
  const uint32_t lds0 = c.lds0;
#include "convh8_body.inc"
}
__global__ __launch_bounds__(512) void convh8_relu_asm_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ch_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  ConvH8Ctx c;
  if (!convh8_prologue(a, ch_smem, c)) return;
  const int tid = threadIdx.x, wave = c.wave, tx0 = c.tx0, ty0 = c.ty0, ct0 = c.ct0, fr0 = c.fr0, ntl = c.ntl;
  const uint32_t lds0 = c.lds0;
#include "convh8r_body.inc"
}

// The normalising form (gen_convh8_body.py WITH_NORM): the input is the producing layer's PRE-BatchNorm tensor; the staged halo tile is normalised
// (+ ReLU) in LDS with that layer's per-frame scale / shift before the MFMAs read it -- bn_relu_h_kernel folded into its consumer.
__global__ __launch_bounds__(512) void convh8_norm_asm_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ch_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  ConvH8Ctx c;
  if (!convh8_prologue(a, ch_smem, c)) return;
  const int tid = threadIdx.x, wave = c.wave, tx0 = c.tx0, ty0 = c.ty0, ct0 = c.ct0, fr0 = c.fr0, ntl = c.ntl;
  const uint32_t lds0 = c.lds0;
#include "convh8n_body.inc"
}

#ifdef S2L_WITH_REFERENCE_KERNELS
// The alternating-roles form (gen_convhx_body.py): the same tile and the same per-lane constants as the eight-wave form, the two waves of a
// SIMD taking turns between an MFMA-only segment and a load / request / epilogue segment.  Same arithmetic in the same order: the same bits.
// No gate input (gated launches keep the interleaved kernel).  Store staging: buffer 1's WEIGHT area (32 of its 36 KiB), so that the next
// tile's halo requests into buffer 1 can leave while the partner group is still in its epilogue.
constexpr int kCHStageX = kCHBuf + kCHHalo;
static_assert(8 * 4096 <= kCHW, "the staging blocks fit the weight area");
__global__ __launch_bounds__(512) void convhx_asm_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ch_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  ConvH8Ctx c;
  if (!convh8_prologue<kCHStageX>(a, ch_smem, c)) return;
  const int tid = threadIdx.x, wave = c.wave, tx0 = c.tx0, ty0 = c.ty0, ct0 = c.ct0, fr0 = c.fr0, ntl = c.ntl;
  const uint32_t lds0 = c.lds0;
#include "convhx_body.inc"
}
__global__ __launch_bounds__(512) void convhx_relu_asm_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ch_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  ConvH8Ctx c;
  if (!convh8_prologue<kCHStageX>(a, ch_smem, c)) return;
  const int tid = threadIdx.x, wave = c.wave, tx0 = c.tx0, ty0 = c.ty0, ct0 = c.ct0, fr0 = c.fr0, ntl = c.ntl;
  const uint32_t lds0 = c.lds0;
#include "convhxr_body.inc"
}

#endif

static std::atomic<int> g_convh_kind{0};      // 0: eight waves interleaved, 1: four waves, 2: eight waves in alternating roles (gated launches: 0) (s2l_set_unet_half_kernel)

// 0 if the launch was taken.  Conditions: an even number of 32-channel planes in, whole planes per tensor, cout a multiple of 64
// (<= 256), tensors small enough for 31-bit pixel indices over all their planes.
int launch_convh(const ConvHArgs& a0, hipStream_t st, bool* launched, bool* stats_done, int* stat_blocks) {
  *launched = false;
  if (stats_done) *stats_done = false;
  ConvHArgs a = a0;
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + kCHTileH - 1) / kCHTileH;
  a.n_ct = a.cout / 64;
  const int nch = (a.CA + a.CB) / 32;
  const int cmax = std::max(a.CA, std::max(a.CB, a.cout));
  if (a.CA % 32 != 0 || a.CB % 32 != 0 || a.CA < 32 || nch % 2 != 0 || (a.CB != 0 && !a.inB) || a.cout % 64 != 0 || a.n_ct > 4 || a.n_ct < 1 ||
      a.n_frames <= 0 || (int64_t)a.H * a.W * 64 >= 0x7fffffffLL || (int64_t)a.H * a.W * a.n_frames * (cmax / 32) >= 0x7fffffffLL / 2 ||
      a.H > 255 * 32 || a.W > 255 * 16)
    return S2L_OK;
#ifdef S2L_WITH_REFERENCE_KERNELS
  const bool four = g_convh_kind.load(std::memory_order_relaxed) == 1 && a.relu == 0;      // (the four-wave body exists in the linear form only)
#else
  const bool four = false;
#endif
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * a.n_ct * a.n_frames;
  if (total >= 0x7fffffff) return S2L_OK;
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  static LdsOptIn flag4, flag8, flag8r, flag8n, flagx, flagxr;
  if (a.bz && (a.norm || a.gate || a.relu || !a.bst)) return S2L_E_SIZE;      // (backward statistics: a plain linear launch otherwise)
  if (a.norm) {      // only the default eight-wave form normalises its input; a launch outside its conditions is refused (the caller keeps the two-kernel route)
    if (a.CB != 0 || a.relu || a.gate || a.CA > 128 || misaligned16(a.norm)) return S2L_OK;
    const bool stats_n = a.stat && (int64_t)a.tiles_x * a.tiles_y <= kConvHStatBlocks && !getenv("S2L_NO_CONV_STATS");
    if (!stats_n) a.stat = nullptr;
    if (stats_done) *stats_done = stats_n;
    if (stat_blocks) *stat_blocks = a.tiles_x * a.tiles_y;
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(convh8_norm_asm_kernel), kCHLds, flag8n, dev))) return rc;
    hipLaunchKernelGGL(convh8_norm_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), kCHLds, st, a);
    *launched = true;
    return (int)hipGetLastError();
  }
#ifdef S2L_WITH_REFERENCE_KERNELS
  const bool alternating = g_convh_kind.load(std::memory_order_relaxed) == 2 && a.gate == nullptr;
#else
  const bool alternating = false;
#endif
  // the tile statistics exist in the default form only; a launch that cannot leave them says so and the caller runs its own pass
  static const bool no_conv_stats = getenv("S2L_NO_CONV_STATS") != nullptr;      // (A/B switch of tools/bench_train.py)
  const bool no_bwd_stats = getenv("S2L_NO_CONV_BSTATS") != nullptr;      // (read per launch: tests flip it inside one process)
  const bool stats = a.stat && !alternating && !four && !a.relu && !a.gate && (int64_t)a.tiles_x * a.tiles_y <= kConvHStatBlocks && !no_conv_stats &&
                     !(a.bz && no_bwd_stats);
  if (!stats) a.stat = nullptr;
  if (stats && a.bz) a.gate = a.bz;      // the kernel's backward-statistics mode: stat and gate both set, the "gate" being z (gen_convh8_body.py: bstats_block)
  if (stats_done) *stats_done = stats;
  if (stat_blocks) *stat_blocks = a.tiles_x * a.tiles_y;
#ifdef S2L_WITH_REFERENCE_KERNELS
  if (alternating) {
    const void* fn = a.relu ? reinterpret_cast<const void*>(convhx_relu_asm_kernel) : reinterpret_cast<const void*>(convhx_asm_kernel);
    if ((rc = ensure_dynamic_lds(fn, kCHLds, a.relu ? flagxr : flagx, dev))) return rc;
    if (a.relu)
      hipLaunchKernelGGL(convhx_relu_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), kCHLds, st, a);
    else
      hipLaunchKernelGGL(convhx_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), kCHLds, st, a);
  } else if (four) {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(convh_asm_kernel), kCHLds, flag4, dev))) return rc;
    hipLaunchKernelGGL(convh_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(256), kCHLds, st, a);
  } else
#endif
  if (a.relu) {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(convh8_relu_asm_kernel), kCHLds, flag8r, dev))) return rc;
    hipLaunchKernelGGL(convh8_relu_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), kCHLds, st, a);
  } else {
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(convh8_asm_kernel), kCHLds, flag8, dev))) return rc;
    hipLaunchKernelGGL(convh8_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), kCHLds, st, a);
  }
  *launched = true;
  return (int)hipGetLastError();
}

}  // namespace s2l

// A measurement aid: what does the chip SUSTAIN on v_mfma_f32_32x32x16_bf16 with nothing else in the way?  `waves` waves per CU (4 or 8),
// each issuing `iters` x 8 independent MFMAs on registers; the caller times the launch (tools/ubench_mfma.py).  The dense-MFMA figure of
// MI355X_MICROARCH.md is at the 2.4 GHz peak clock; under this load the clock is lower (power).
namespace s2l {
typedef short bf8v_ __attribute__((ext_vector_type(8)));
typedef float f16v_ __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void mfma_rate_kernel(int64_t iters, float* __restrict__ sink) {
  bf8v_ a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = (short)(0x3f80 + ((threadIdx.x + k) & 7));
    b[k] = (short)(0x3c00 + ((threadIdx.x * 3 + k) & 15));
  }
  f16v_ acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int64_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) t += acc[j][0];
  if (t == 123.456f) sink[0] = t;      // (keeps the chain alive)
}
}  // namespace s2l
extern "C" int s2l_debug_bf16_mfma_rate(int64_t iters, int waves, float* sink, s2l_stream_t stream) {
  if (iters < 1 || (waves != 4 && waves != 8) || !sink) return S2L_E_SIZE;
  int dev = 0, n_cu = 0;
  const int rc = s2l::current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  hipLaunchKernelGGL(s2l::mfma_rate_kernel, dim3(n_cu), dim3(64 * waves), 0, static_cast<hipStream_t>(stream), iters, sink);
  return (int)hipGetLastError();
}

// Which form runs the half-width convolutions: 0 (default) eight waves per workgroup, each interleaving loads and MFMAs, 1 four waves,
// 2 eight waves in alternating roles (gen_convhx_body.py; gated launches run as 0).  Same arithmetic in the same order: the same bits (a test
// aid).  Form 2 was built to test whether the schedule inside a CU bounds this kernel: it does not (same time to +-3 %: LABNOTES §10).
extern "C" int s2l_set_unet_half_kernel(int kind) {
  if (kind < 0 || kind > 2) return S2L_E_SIZE;
#ifndef S2L_WITH_REFERENCE_KERNELS
  if (kind != 0) return S2L_E_UNSUPPORTED;      // (forms 1 and 2 live in libs2l_hip_ref.so)
#endif
  s2l::g_convh_kind.store(kind, std::memory_order_relaxed);
  return S2L_OK;
}
