// The split-bf16 3x3 convolution of the U-Net's inference speed mode as a generated-assembly kernel (csrc/gen_conv16_body.py has
// the design, the register map and the schedule; conv3x3_split_kernel in csrc/unet.hip is the C++ kernel whose arithmetic it
// performs in the same order -- the outputs are the same bits).  Replaces, for `SimpleUnetLight.forward` in eval mode
// (SimpleUnetLight.py:16-111 applied at tf_nerf.py:387), one `nn.Conv2d(3x3, padding=1) + BatchNorm2d (folded) + ReLU`.
#include "s2l_common.h"
#include "conv16.h"

namespace s2l {

static_assert(offsetof(Conv16Args, inA) == 0 && offsetof(Conv16Args, inB) == 8 && offsetof(Conv16Args, w16) == 16 &&
              offsetof(Conv16Args, bias) == 24 && offsetof(Conv16Args, out) == 32 && offsetof(Conv16Args, CA) == 40 &&
              offsetof(Conv16Args, CB) == 44 && offsetof(Conv16Args, cout) == 48 && offsetof(Conv16Args, H) == 52 &&
              offsetof(Conv16Args, W) == 56 && offsetof(Conv16Args, tiles_x) == 60 && offsetof(Conv16Args, tiles_y) == 64 &&
              offsetof(Conv16Args, n_ct) == 68 && offsetof(Conv16Args, relu) == 72,
              "gen_conv16_body.py (ARG) loads these fields from the kernarg segment by offset");

constexpr int kC16TileH = 32;      // tile = 32 rows x 16 columns (gen_conv16_body.py: TILE_H)
constexpr int kC16Halo = (kC16TileH + 2) * 18 * 64, kC16W = 9 * 2 * 2 * 64 * 16, kC16Buf = kC16Halo + kC16W;
constexpr int kC16Lds = 2 * kC16Buf + 1024;      // two buffers + the bias table (the store staging aliases buffer 1's halo area)
static_assert(kC16Lds <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(256) void conv16_asm_kernel(Conv16Args a) {
  extern __shared__ __attribute__((aligned(16))) char c16_smem[];
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();   // the body loads the Conv16Args fields itself (s_load)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)c16_smem);
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
  // per-lane constants.  Staging: quad qi = tid + 256 i of the 34 x 18 x 4 halo quads = pixel qi / 4 (row-major), channels
  // 4 (qi % 4) .. + 3 of the chunk: 8 bytes of hi at 16-byte segment (c4 >> 1) ^ swizzle, 8 bytes of lo at that address ^ 32.
  // They reach the assembly body through LDS ([word 34][thread 256] at the start of buffer 0; the body reads them first).
  uint32_t* cst = reinterpret_cast<uint32_t*>(c16_smem);
  constexpr int kQuads = (kC16TileH + 2) * 18 * 4;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int qi = tid + 256 * i, pi = (qi < kQuads ? qi : 0) / 4, c4 = qi & 3;
    const int row = pi / 18, col = pi % 18;
    cst[i * 256 + tid] = lds0 + (uint32_t)((row * 18 + col) * 64 + ((((c4 >> 1) ^ ((col >> 2) & 3))) << 4) + (c4 & 1) * 8);
    cst[(10 + i) * 256 + tid] = (uint32_t)((row << 8) | col);
  }
  // operand reads: lane (n = lane & 31, hh = lane >> 5): pixel (row 8 wave + (n >> 4) [+ 2 blk + dy as an immediate], col (n & 15) + dx),
  // 16-byte segment (2 part + hh) ^ swizzle
  // (which pixel of its N-block lane n stands for follows ds_read_b128's NON-contiguous lane groups {0-3, 12-15, 20-27} / {4-11, 16-19,
  // 28-31}: the sixteen lanes of a group read one row -- csrc/convh.hip has the measurement)
  const int n_ = lane & 31;
  const int in_g0 = (n_ < 4) || (n_ >= 12 && n_ < 16) || (n_ >= 20 && n_ < 28);
  const int prow = in_g0 ? 0 : 1;
  const int pcol = in_g0 ? (n_ < 4 ? n_ : n_ < 16 ? n_ - 8 : n_ - 12) : (n_ < 12 ? n_ - 4 : n_ < 20 ? n_ - 8 : n_ - 16);
  {
    const int hh = lane >> 5;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int col = pcol + dx, row = 8 * wave + prow;
        cst[(20 + dx * 2 + pt) * 256 + tid] = lds0 + (uint32_t)((row * 18 + col) * 64 + (((2 * pt + hh) ^ ((col >> 2) & 3)) << 4));
      }
  }
  // store staging (this wave's 4 KiB at the start of buffer 1: [32 pixels][32 channels] fp32, 16-byte quad index ^ ((pixel >> 1) & 7)):
  // write address of the lane's register quad rq (pixel n = lane & 31, channels 8 rq + 4 hh ..), read address of store j
  // (pixel 8 j + (lane >> 3), quad lane & 7)
  {
    const uint32_t stg = lds0 + kC16Buf + wave * 4096;
    const int n = 16 * prow + pcol, hh = lane >> 5;      // the lane's pixel of the block
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) cst[(26 + rq) * 256 + tid] = stg + (uint32_t)(n * 128 + (((2 * rq + hh) ^ ((n >> 1) & 7)) << 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = 8 * j + (lane >> 3);
      cst[(30 + j) * 256 + tid] = stg + (uint32_t)(px * 128 + (((lane & 7) ^ ((px >> 1) & 7)) << 4));
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the constants are in LDS (each lane reads back only its own words)
#include "conv16_body.inc"
}

// MaxPool2d(2) of the NHWC activation, one thread per (pooled pixel, channel quad)
__global__ __launch_bounds__(256) void pool16_kernel(const float* __restrict__ a, float* __restrict__ p, int H, int W, int C, int64_t n_quads) {
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

}  // namespace s2l

namespace s2l {

// 0 if the launch was taken.  Conditions: an even number of 16-channel chunks, CA a multiple of 16, cout a multiple of 64, frames
// small enough for 32-bit in-frame offsets.
int launch_conv16_asm(const Conv16Args& a0, hipStream_t st, bool* launched) {
  *launched = false;
  Conv16Args a = a0;
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + kC16TileH - 1) / kC16TileH;
  const int nch = (a.CA + a.CB) / 16;
  if ((a.CA + a.CB) % 16 != 0 || nch % 2 != 0 || a.CA % 16 != 0 || a.cout % 64 != 0 || a.n_ct != a.cout / 64 || a.n_ct > 4 ||
      (int64_t)a.H * a.W * 128 * 4 >= 0x7fffffffLL || (int64_t)a.H * a.W * a.n_frames >= 0x7fffffffLL || a.n_frames <= 0)
    return S2L_OK;
  const int64_t total = (int64_t)a.tiles_x * a.tiles_y * a.n_ct * a.n_frames;
  if (total >= 0x7fffffff) return S2L_OK;
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  static LdsOptIn flag;
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv16_asm_kernel), kC16Lds, flag, dev))) return rc;
  hipLaunchKernelGGL(conv16_asm_kernel, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(256), kC16Lds, st, a);
  if (a.pool) {
    const int64_t nq = (int64_t)a.n_frames * (a.H / 2) * (a.W / 2) * (a.cout / 4);
    hipLaunchKernelGGL(pool16_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, a.out, a.pool, a.H, a.W, a.cout, nq);
  }
  *launched = true;
  return (int)hipGetLastError();
}

}  // namespace s2l
