// Split-half speed mode of the clip renderer: s2l_render_lip_split (OPT-IN; the exact fp32 kernel of render.hip stays the default).
//
// Same function as s2l_render_lip -- TalkingFace.rgb_forward (tf_nerf.py:225-285) over every pixel of every frame of a clip,
// inference.py:140-159 -- with the seven 256x256 layers and the output layer evaluated on v_mfma_f32_16x16x32_f16:
// every fp32 operand x travels as hi = f16(x), lo = f16(x - hi) and a product is W_lo a_hi + W_hi a_lo + W_hi a_hi with
// fp32 accumulation (csrc/gen_render16_body.py has the arithmetic, the register map and the schedule).  Tile shapes, the
// LDS-DMA ring, the p / q tables and the output store are those of render.hip; only the weight slabs differ (half hi | lo
// A operands, s2l_pack_render16 derives them from the fp32 slabs of the packed blob).
// Accuracy against the CPU oracle: RMSE 1.4e-6 / 117 dB (the exact kernel: 6.5e-7); the north-star tolerance is RMSE <= 1e-4 /
// PSNR >= 50 dB.  Range: |pre-activation| < 65504 (values beyond saturate); the exact kernel has no such condition.
#include "s2l_common.h"

namespace s2l {

#ifndef S2L_RENDER16_BF16   // IEEE half parts (default); -DS2L_RENDER16_BF16 + S2L_RENDER16_HALF=bf16 for the generator: bf16 parts (A/B)
__device__ __forceinline__ uint16_t r16_half(float x) {      // (clamped to the half range first: a part is never inf)
  return __builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f));
}
__device__ __forceinline__ float r16_float(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
#else
__device__ __forceinline__ uint16_t r16_half(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }   // round to nearest even
__device__ __forceinline__ float r16_float(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
#endif

struct Render16Args {
  const float* packed;        // fp32 blob: biases
  const uint16_t* packed16;   // [113 slabs][8 k-steps][hi | lo][64 lanes][8] halves
  const float* p0t;
  const float* p5t;
  const float* q0;
  const float* q5;
  float* out;
  int hw, nframes;
  int npg, ntiles;
  int nfg;
};

enum Render16Shape { k16Long = 0, k16Wide = 1, k16Single = 2 };
struct Shape16Dims { int g, pgt, ft; };
__host__ __device__ constexpr Shape16Dims shape16_dims(int shape) {
  return shape == k16Long ? Shape16Dims{3, 1, 12} : shape == k16Wide ? Shape16Dims{3, 12, 1} : Shape16Dims{1, 4, 1};
}

constexpr int k16Ring = 9;
constexpr int k16SlabBytes = kSlab * 4;                  // 16384: 8 k-steps x 2 parts x 64 lanes x 16 B
constexpr int k16Slabs = kHidden * 16 + 1;               // 113
constexpr int k16BiasFloats = kHidden * kW + 4;
constexpr int k16LdsBytes = k16Ring * k16SlabBytes + (k16BiasFloats * 4 + 64 + 15) / 16 * 16;
static_assert(k16LdsBytes <= 160 * 1024, "LDS budget");

// fp32 slab element [j4][lane][jj] = W[16 mb + (lane & 15)][16 j4 + 4 q + jj]  (s2l_layout.h)
//   -> bf16 slab element [s][part][lane][e], e = 0..7: the fp32 element [2 s + e / 4][lane][e % 4], as hi (part 0) or lo (part 1).
// One thread per (slab, k-step, lane): reads two 16-byte quads, writes two 16-byte operands.
__global__ void pack_render16_kernel(const float* __restrict__ packed, uint16_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k16Slabs * 8 * 64) return;
  const int lane = i & 63, s = (i >> 6) & 7, slab = i >> 9;
  const float* src = packed + OFF_WMLP + (int64_t)slab * kSlab;
  uint16_t hi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float w = src[((2 * s + e / 4) * 64 + lane) * 4 + (e & 3)];
    const uint16_t h = r16_half(w);
    hi[e] = h;
    lo[e] = r16_half(w - r16_float(h));
  }
  uint16_t* dst = out + ((int64_t)slab * 8 + s) * 2 * 64 * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dst[lane * 8 + e] = hi[e];
    dst[64 * 8 + lane * 8 + e] = lo[e];
  }
}

template <int SHAPE>
__global__ __launch_bounds__(256) void render16_tiles_kernel(Render16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < k16BiasFloats; i += 256)
    reinterpret_cast<float*>(smem + k16Ring * k16SlabBytes)[i] = a.packed[OFF_BIAS + i];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, px = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
  const int nfg = __builtin_amdgcn_readfirstlane(a.nfg);
  const int npgm1 = __builtin_amdgcn_readfirstlane(a.npg - 1);
  (void)npgm1;
  const int tile0 = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * blockIdx.x / gridDim.x));
  const int tile_end = __builtin_amdgcn_readfirstlane((int)((int64_t)a.ntiles * (blockIdx.x + 1) / gridDim.x));
  const int fg0 = __builtin_amdgcn_readfirstlane(tile0 % nfg), pg0 = __builtin_amdgcn_readfirstlane(tile0 / nfg);
  const uint16_t* wsrc = a.packed16;
  const uint32_t lane16 = lds0 + lane * 16;
  const uint32_t dmaoff = wave * 4096 + lane * 16;
  const uint32_t biasaddr = lds0 + k16Ring * k16SlabBytes + 16 * q;
  const uint32_t boutaddr = lds0 + k16Ring * k16SlabBytes + kHidden * kW * 4;
  const uint32_t qaddr = lds0 + (SHAPE == k16Long ? wave * 3072 : 0) + q * 16;
  if constexpr (SHAPE == k16Long) {
#include "render16_body_long.inc"
  } else if constexpr (SHAPE == k16Wide) {
#include "render16_body_wide.inc"
  } else {
#include "render16_body_single.inc"
  }
}

}  // namespace s2l

extern "C" int s2l_render_shape_mode(void);        // render.hip: 0 = choose per call, 1 + shape = forced (s2l_set_render_shape)
extern "C" int s2l_render_cu_limit(int dev);       // render.hip: s2l_set_render_cus of this device (0 = one workgroup per CU)

extern "C" int64_t s2l_render16_packed_halves(void) { return (int64_t)s2l::k16Slabs * s2l::kSlab * 2; }

extern "C" int s2l_pack_render16(const float* packed, void* packed16, s2l_stream_t stream) {
  if (!packed || !packed16) return S2L_E_NULL;
  if (s2l::misaligned16(packed) || s2l::misaligned16(packed16)) return S2L_E_ALIGN;
  const int n = s2l::k16Slabs * 8 * 64;
  hipLaunchKernelGGL(s2l::pack_render16_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), packed,
                     static_cast<uint16_t*>(packed16));
  return (int)hipGetLastError();
}

extern "C" int s2l_render_lip_split(const float* packed, const void* packed16, const float* p0, const float* p5, const float* q0,
                                    const float* q5, float* out, int64_t hw, int64_t n_frames, s2l_stream_t stream) {
  using namespace s2l;
  if (hw <= 0 || hw > (1 << 24) || n_frames < 0 || n_frames > (1 << 24)) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!packed || !packed16 || !p0 || !p5 || !q0 || !q5 || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(packed16) || misaligned16(p0) || misaligned16(p5) || misaligned16(q0) || misaligned16(q5))
    return S2L_E_ALIGN;
  Render16Args a;
  a.packed = packed; a.packed16 = static_cast<const uint16_t*>(packed16); a.p0t = p0; a.p5t = p5; a.q0 = q0; a.q5 = q5; a.out = out;
  a.hw = (int)hw; a.nframes = (int)n_frames;
  a.npg = (int)((hw + 15) / 16);
  static LdsOptIn lds_flags[3];
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  const int limit = s2l_render_cu_limit(dev);
  if (limit > 0 && limit < n_cu) n_cu = limit;
  // the shape with the fewest rounds of tiles over the persistent grid (tile costs as in render.hip: 1 / 1.03 / 0.36)
  int shape = k16Long;
  double best = 0;
  const double cost[3] = {1.0, 1.03, 0.36};
  for (int shp = 0; shp < 3; ++shp) {
    const Shape16Dims d = shape16_dims(shp);
    const int64_t tiles = (((int64_t)a.npg + d.pgt - 1) / d.pgt) * ((n_frames + d.ft - 1) / d.ft);
    const int64_t grid = tiles < n_cu ? tiles : n_cu;
    const double t = (double)((tiles + grid - 1) / grid) * cost[shp];
    if (shp == 0 || t < best * 0.97) best = t, shape = shp;
  }
  if (const int forced = s2l_render_shape_mode(); forced >= 1 && forced <= 3) shape = forced - 1;      // (4 = the exact renderer's feature-split tile: not a shape of this kernel)
  const Shape16Dims d = shape16_dims(shape);
  a.nfg = (int)((n_frames + d.ft - 1) / d.ft);
  const int64_t ntiles = (int64_t)((a.npg + d.pgt - 1) / d.pgt) * a.nfg;
  if (ntiles > 0x7fffffff) return S2L_E_SIZE;
  a.ntiles = (int)ntiles;
  void (*const kern[3])(Render16Args) = {render16_tiles_kernel<k16Long>, render16_tiles_kernel<k16Wide>, render16_tiles_kernel<k16Single>};
  if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern[shape]), k16LdsBytes, lds_flags[shape], dev))) return rc;
  const int grid = a.ntiles < n_cu ? a.ntiles : n_cu;
  hipLaunchKernelGGL(kern[shape], dim3(grid), dim3(256), k16LdsBytes, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
