// The implicit-GEMM convolution shared by the frozen conv nets of the training losses (SyncNet_color: csrc/syncnet.hip;
// the AlexNet trunk of LPIPS: csrc/lpips.hip).  Kernels 1..11, strides (1..4) x (1..4), any padding, 1..512 channels, optional
// folded BatchNorm, optional residual add, ReLU -- ONE kernel on v_mfma_f32_16x16x4_f32 (exact fp32):
//   D[row][col] = sum_k A[row][k] B[k][col]
//   forward : rows = output channels, cols = output pixels (b,oy,ox), k = (ky,kx,ci), B gathered from the NHWC input
//   dgrad   : rows = input channels,  cols = input pixels  (b,iy,ix), k = (ky,kx,co), B gathered from the masked output
//             gradient at oy = (iy + pad - ky) / stride where that division is exact
// A workgroup (4 waves) owns a 64x64 tile (32x128 / 16x256 for layers of <= 32 / <= 16 rows); K advances in chunks of 16 through
// LDS with the next chunk's global loads in flight during the MFMAs.  Deep layers have few pixels and K up to 4608, so K is also split across gridDim.z into partial
// tiles that a second kernel reduces in a fixed order (deterministic) before the epilogue.
#pragma once
#include "s2l_common.h"

namespace s2l {

struct LayerSpec {
  int cin, cout, kh, kw, sy, sx, py, px, res;
};


inline int ceil_to(int a, int m) { return (a + m - 1) / m * m; }

struct Shape {
  int h, w;
};
inline Shape out_shape(const LayerSpec& s, Shape in) {
  return Shape{(in.h + 2 * s.py - s.kh) / s.sy + 1, (in.w + 2 * s.px - s.kw) / s.sx + 1};
}


constexpr int64_t kPartialFloats = 1 << 23;   // split-K scratch a caller provides per launch (32 MiB)

// ---- packing: fold BatchNorm (eval; gamma == nullptr: a plain convolution), lay out for the implicit GEMM ----------------------------------------------------
// forward:  dst[(tap*cinp + ci)*RP + co] = w[co][ci][ky][kx] * g[co]/sqrt(var[co]+eps)
// dgrad:    dst[(tap*coutp + co)*RP + ci] = the same number, K and row roles swapped
__global__ static void conv_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                    float eps, float* __restrict__ dst, int cin, int cout, int kh, int kw, int kcp, int RP,
                                    int dgrad, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = (int)(i % RP);
  const int64_t kidx = i / RP;
  const int kc = (int)(kidx % kcp), tap = (int)(kidx / kcp);
  const int ky = tap / kw, kx = tap % kw;
  const int co = dgrad ? kc : row, ci = dgrad ? row : kc;
  float v = 0.f;
  if (co < cout && ci < cin) v = w[(((int64_t)co * cin + ci) * kh + ky) * kw + kx] * (gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f);
  dst[i] = v;
}

__global__ static void conv_pack_bias_kernel(const float* __restrict__ b, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, const float* __restrict__ mean,
                                         const float* __restrict__ var, float eps, float* __restrict__ dst, int cout, int RP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RP) return;
  dst[i] = i >= cout ? 0.f : gamma ? (b[i] - mean[i]) * (gamma[i] / sqrtf(var[i] + eps)) + beta[i] : b[i];
}

// ---- the implicit-GEMM convolution --------------------------------------------------------------------------------------
struct ConvArgs {
  const float* in;    // fwd: a_{L-1} [B,hin,win,cin];  dgrad: g_L [B,hout,wout,cout] (gradient w.r.t. the pre-ReLU sum)
  const float* w;     // packed A operand [K/16][16][RP]
  const float* bias;  // fwd: folded bias [RP]
  const float* res;   // fwd: residual source (a_{L-1}) or null;  dgrad: g_L when the layer is residual (pass-through) or null
  const float* mask;  // dgrad: a_{L-1}; the result is multiplied by (a_{L-1} > 0) (ReLU of the layer below) -- or null
  float* out;         // fwd: a_L [B,hout,wout,cout];  dgrad: g_{L-1} [B,hin,win,cin]
  float* partial;     // split-K: [splits][ncols][RP] partial sums, else null
  int hin, win, cin, hout, wout, cout, kh, kw, sy, sx, py, px;
  int rows, RP, kc, kcp, ncols, nchunks, chunks_per_split;
  int PR;             // row stride of the split-K partial tiles: rows rounded up to the tile height (16 / 32 / 64), <= RP
  const uint16_t* w16;  // or null: the A operand as hi + lo bf16 parts in MFMA lane order (conv_pack16_kernel) -- the split form below
  int nsteps, steps_per_split;      // split form: K in steps of 32
};

// one output value: bias / residual / ReLU (forward), pass-through / ReLU mask of the layer below (dgrad)
template <bool DGRAD>
__device__ static __forceinline__ void conv_store(const ConvArgs& a, int c, int row, float v) {
  const int64_t o = (int64_t)c * a.rows + row;
  if (!DGRAD) {
    v += a.bias[row];
    if (a.res) v += a.res[o];
    v = fmaxf(v, 0.f);
  } else {
    if (a.res) v += a.res[o];
    if (a.mask) v = a.mask[o] > 0.f ? v : 0.f;
  }
  a.out[o] = v;
}

__device__ static __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// K advances in GROUPS of up to four 16-wide chunks per barrier pair (round 3): with one chunk per pair a wave issued 16 MFMAs
// (512 cycles) between two barriers and a global-load wait, and the kernel sat at 0.20 of the fp32-MFMA peak on the lip-sync
// expert; the accumulation order (chunk by chunk, k ascending) is unchanged, so the results are the same bits.
constexpr int kGroup = 4;

// Tile shapes.  TM rows x TN columns per workgroup of four waves, TM * TN = 4096 accumulators in every shape:
//   TM = 64: waves 2 x 2, each 32 x 32 (2 x 2 MFMA sub-tiles)            -- the general case
//   TM = 32: waves 1 x 4, each 32 x 32 (2 x 2)         tile 32 x 128    -- <= 32 rows: the 15 -> 32 first layer of the face
//   TM = 16: waves 1 x 4, each 16 x 64 (1 x 4)         tile 16 x 256       encoder, the data gradients of its first two layers
//                                                                          (32 and 15 rows), AlexNet's 3-row input gradient
// A 64-row tile on those layers spent 1/2 .. 3/4 of its MFMAs on zero rows (the two largest launches of the sync-loss gradient).
// The accumulation order per output element does not depend on the shape.
template <int TM> struct TileShape {
  static constexpr int TN = 4096 / TM;
  static constexpr int WGM = TM == 64 ? 2 : 1, WGN = 4 / WGM;
  static constexpr int SM = TM / (16 * WGM), SN = TN / (16 * WGN);
  static constexpr int NB = TN / 64;                       // B pixels fetched per thread and chunk
  static constexpr int LdA = TM == 64 ? 80 : TM == 32 ? 48 : 16;      // LDS row strides in floats, stride % 32 == 16: the four
  static constexpr int LdB = TN + 16;                                 //   k-rows of an operand read land on distinct banks
  static constexpr int G = TM == 16 ? 2 : kGroup;          // chunks per barrier pair (LDS: 40 / 48 / 37 KiB)
};

template <bool DGRAD, int TM>
__global__ static __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
  using S = TileShape<TM>;
  constexpr int TN = S::TN, SM = S::SM, SN = S::SN, NB = S::NB, LdA = S::LdA, LdB = S::LdB, G = S::G;
  __shared__ float As[16 * G * LdA];
  __shared__ float Bs[16 * G * LdB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % S::WGM, wn = wave / S::WGM, q = lane >> 4, l16 = lane & 15;
  const int col0 = blockIdx.x * TN, row0 = blockIdx.y * TM;
  const int chunk_lo = blockIdx.z * a.chunks_per_split;
  const int chunk_hi = min(a.nchunks, chunk_lo + a.chunks_per_split);

  // B-load role: pixels pl + 64 j of the tile, channel quad cq of the chunk
  const int pl = t >> 2, cq = t & 3;
  const int cw = DGRAD ? a.win : a.wout, chw = DGRAD ? a.hin * a.win : a.hout * a.wout;
  bool col_ok[NB];
  int pn[NB], cy[NB], cx[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = col0 + pl + 64 * j;
    col_ok[j] = col < a.ncols;
    const int cc = col_ok[j] ? col : 0;
    pn[j] = cc / chw;
    const int rem = cc - pn[j] * chw;
    cy[j] = rem / cw, cx[j] = rem - cy[j] * cw;
  }
  // A-load role: 16 k-rows x TM / 4 row quads
  const bool a_ok = t < 4 * TM;
  const int ak = t / (TM / 4), ar4 = t % (TM / 4);
  const bool vec = (a.kc & 3) == 0;

  f4 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  auto fetch1 = [&](int chunk, f4& av, f4 (&bv)[NB]) {
    if (a_ok) av = *reinterpret_cast<const f4*>(a.w + ((int64_t)chunk * 16 + ak) * a.RP + row0 + ar4 * 4);
    const int kidx0 = chunk * 16;
    const int tap = kidx0 / a.kcp, c0 = kidx0 - tap * a.kcp + 4 * cq;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bool ok = col_ok[j] && c0 < a.kc;
      const float* src;
      if (!DGRAD) {
        const int iy = cy[j] * a.sy - a.py + ky, ix = cx[j] * a.sx - a.px + kx;
        ok = ok && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
        src = a.in + (((int64_t)pn[j] * a.hin + iy) * a.win + ix) * a.cin + c0;
      } else {
        const int ty = cy[j] + a.py - ky, tx = cx[j] + a.px - kx;
        const int oy = ty / a.sy, ox = tx / a.sx;
        ok = ok && ty >= 0 && tx >= 0 && oy * a.sy == ty && ox * a.sx == tx && oy < a.hout && ox < a.wout;
        src = a.in + (((int64_t)pn[j] * a.hout + oy) * a.wout + ox) * a.cout + c0;
      }
      bv[j] = f4{0.f, 0.f, 0.f, 0.f};
      if (ok) {
        if (vec) {
          bv[j] = *reinterpret_cast<const f4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < a.kc) bv[j][e] = src[e];
        }
      }
    }
  };
  f4 av[G], bv[G][NB];
  auto fetch = [&](int chunk) {      // chunks chunk .. chunk + G - 1 that exist (the MFMA loop stops at the last one)
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (chunk + g < chunk_hi) fetch1(chunk + g, av[g], bv[g]);
  };

  if (chunk_lo < chunk_hi) fetch(chunk_lo);
  for (int chunk = chunk_lo; chunk < chunk_hi; chunk += G) {
    const int ng = min(G, chunk_hi - chunk);      // (uniform over the workgroup)
    __syncthreads();  // the previous group's operand reads are done
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {
        if (a_ok) *reinterpret_cast<f4*>(&As[(16 * g + ak) * LdA + ar4 * 4]) = av[g];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) Bs[(16 * g + 4 * cq + e) * LdB + pl + 64 * j] = bv[g][j][e];
      }
    }
    __syncthreads();
    if (chunk + G < chunk_hi) fetch(chunk + G);
    // the operand values of chunk g + 1 are read from LDS while the MFMAs of chunk g issue (two register sets): with
    // "reads, wait, MFMAs" per k-step a chunk took ~1 400 cycles for 512 cycles of matrix work (one wave per SIMD per workgroup)
    float fa[2][4][SM], fb[2][4][SN];
    auto lds_operands = [&](int g, int set) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < SM; ++i) fa[set][kk][i] = As[(16 * g + 4 * kk + q) * LdA + 16 * SM * wm + 16 * i + l16];
#pragma unroll
        for (int j = 0; j < SN; ++j) fb[set][kk][j] = Bs[(16 * g + 4 * kk + q) * LdB + 16 * SN * wn + 16 * j + l16];
      }
    };
    lds_operands(0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {
        if (g + 1 < ng) lds_operands(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int j = 0; j < SN; ++j) acc[i][j] = mfma16(fa[g & 1][kk][i], fb[g & 1][kk][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // D[row = 4q + r][col = l16] of sub-tile (i, j)
#pragma unroll
  for (int j = 0; j < SN; ++j) {
    const int c = col0 + 16 * SN * wn + 16 * j + l16;
    if (c >= a.ncols) continue;
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const int r0 = row0 + 16 * SM * wm + 16 * i + 4 * q;
      if (a.partial) {
        *reinterpret_cast<f4*>(a.partial + ((int64_t)blockIdx.z * a.ncols + c) * a.PR + r0) = acc[i][j];
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r0 + r < a.rows) conv_store<DGRAD>(a, c, r0 + r, acc[i][j][r]);
    }
  }
}

// ---- the SPLIT form (round 5): the same GEMM with both operands as hi + lo bf16 parts, three v_mfma_f32_16x16x32_bf16 per product
// block (lo x hi, hi x lo, hi x hi; lo x lo, <= 2^-18 of the product, is dropped), fp32 accumulation -- 16 significant bits per
// operand, range of fp32 (so it serves the gradients too, which IEEE halves would flush).  An opt-in speed mode for the bf16-precision
// training steps only: the exact fp32 kernel above stays the default and the one the parity tests pin.  Every layer with
// >= 8 K channels takes it (64 x 64 tiles, 32 x 128 for <= 32 rows); the one-channel mel window keeps the exact kernel.
//   A: packed ONCE in MFMA lane order -- [k-step of 32][16-row block][hi | lo][lane 64][8 bf16]: a wave loads its operand with one
//      coalesced 16-byte load per lane straight from global memory (1 KiB per instruction), no LDS;
//   B: gathered as in the exact kernel, 8 channels (two 16-byte loads) per thread and k-step, split into parts on the way to LDS;
//      LDS rows are K-contiguous (64 B per pixel and part), 16-byte pieces XOR-swizzled by the pixel so that both the commit and the
//      operand reads are conflict-free.
typedef short cg_bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t cg_u4 __attribute__((ext_vector_type(4)));
typedef __bf16 cg_bf2 __attribute__((ext_vector_type(2)));

__device__ static __forceinline__ uint32_t cg_pack_bf2(float x, float y) {
  cg_bf2 v;
  v[0] = (__bf16)x;
  v[1] = (__bf16)y;
  return __builtin_bit_cast(uint32_t, v);
}
// (x, y) -> hi parts (round to nearest even) and the parts of what they leave
__device__ static __forceinline__ void cg_split2(float x, float y, uint32_t& hi, uint32_t& lo) {
  hi = cg_pack_bf2(x, y);
  lo = cg_pack_bf2(x - __uint_as_float(hi << 16), y - __uint_as_float(hi & 0xffff0000u));
}

// one thread per element of [k-step][row block][part][lane][8]
__global__ static void conv_pack16_kernel(const float* __restrict__ w, uint16_t* __restrict__ dst, int K16, int RP, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = (int)(i & 7), lane = (int)(i >> 3) & 63, part = (int)(i >> 9) & 1;
  const int64_t blk = i >> 10;
  const int RB = RP / 16, rb = (int)(blk % RB), ks = (int)(blk / RB);
  const int k = ks * 32 + (lane >> 4) * 8 + e, row = rb * 16 + (lane & 15);
  const float v = k < K16 ? w[(int64_t)k * RP + row] : 0.f;
  uint32_t hi, lo;
  cg_split2(v, 0.f, hi, lo);
  dst[i] = (uint16_t)(part ? lo : hi);
}

// Tile shapes of the split form: 64 x 64 (waves 2 x 2) or, for layers of <= 32 rows, 32 x 128 (waves 1 x 4); a wave owns 32 x 32.
template <int TM> struct SplitShape {
  static constexpr int WGM = TM == 64 ? 2 : 1, WGN = 4 / WGM, TN = 32 * WGN;
  static constexpr int NB = TN / 64;                 // B pixels fetched per thread and k-step
  static constexpr int G = TM == 64 ? 4 : 2;         // k-steps of 32 per barrier pair (LDS 32 KiB in both shapes)
};

template <bool DGRAD, int TM>
__global__ static __launch_bounds__(256) void conv_gemm_split_kernel(ConvArgs a) {
  using S = SplitShape<TM>;
  constexpr int G = S::G, TN = S::TN, NB = S::NB;
  __shared__ cg_u4 Bs[G * 2 * TN * 4];      // [g][part][pixel][piece ^ ((pixel >> 2) & 3)]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % S::WGM, wn = wave / S::WGM, q = lane >> 4, l16 = lane & 15;
  const int col0 = blockIdx.x * TN, row0 = blockIdx.y * TM;
  const int step_lo = blockIdx.z * a.steps_per_split;
  const int step_hi = min(a.nsteps, step_lo + a.steps_per_split);
  const int RB = a.RP / 16, K16 = a.nchunks * 16;

  // B-load role: pixels pl + 64 j of the tile, channel octet oct of the k-step
  const int pl = t >> 2, oct = t & 3;
  const int cw = DGRAD ? a.win : a.wout, chw = DGRAD ? a.hin * a.win : a.hout * a.wout;
  bool col_ok[NB];
  int pn[NB], cy[NB], cx[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = col0 + pl + 64 * j;
    col_ok[j] = col < a.ncols;
    const int cc = col_ok[j] ? col : 0;
    pn[j] = cc / chw;
    const int rem = cc - pn[j] * chw;
    cy[j] = rem / cw, cx[j] = rem - cy[j] * cw;
  }
  const bool vec = (a.kc & 7) == 0;      // else (the 15-channel face window): element loads, the tail of the last octet is zero

  f4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  auto fetch_b = [&](int step, f4 (&bv)[NB][2]) {
    const int kidx0 = step * 32 + oct * 8;
    const int tap = kidx0 / a.kcp, c0 = kidx0 - tap * a.kcp;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bool ok = col_ok[j] && c0 < a.kc && kidx0 < K16;
      const float* src;
      if (!DGRAD) {
        const int iy = cy[j] * a.sy - a.py + ky, ix = cx[j] * a.sx - a.px + kx;
        ok = ok && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
        src = a.in + (((int64_t)pn[j] * a.hin + iy) * a.win + ix) * a.cin + c0;
      } else {
        const int ty = cy[j] + a.py - ky, tx = cx[j] + a.px - kx;
        const int oy = ty / a.sy, ox = tx / a.sx;
        ok = ok && ty >= 0 && tx >= 0 && oy * a.sy == ty && ox * a.sx == tx && oy < a.hout && ox < a.wout;
        src = a.in + (((int64_t)pn[j] * a.hout + oy) * a.wout + ox) * a.cout + c0;
      }
      bv[j][0] = bv[j][1] = f4{0.f, 0.f, 0.f, 0.f};
      if (ok) {
        if (vec) {
          bv[j][0] = *reinterpret_cast<const f4*>(src);
          bv[j][1] = *reinterpret_cast<const f4*>(src + 4);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < a.kc) bv[j][e >> 2][e & 3] = src[e];
        }
      }
    }
  };
  // A operand of one k-step: [sub-tile i][part]
  const uint16_t* abase = a.w16 + ((int64_t)(row0 / 16 + 2 * wm) * 2) * 512 + lane * 8;
  auto fetch_a = [&](int step, cg_u4 (&av)[2][2]) {
    const uint16_t* p = abase + (int64_t)step * RB * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int part = 0; part < 2; ++part) av[i][part] = *reinterpret_cast<const cg_u4*>(p + (i * 2 + part) * 512);
  };

  f4 bv[G][NB][2];
  cg_u4 av[2][2][2];
  auto fetch_group_b = [&](int step) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (step + g < step_hi) fetch_b(step + g, bv[g]);
  };
  if (step_lo < step_hi) {
    fetch_group_b(step_lo);
    fetch_a(step_lo, av[0]);
  }
  for (int step = step_lo; step < step_hi; step += G) {
    const int ng = min(G, step_hi - step);      // (uniform over the workgroup)
    __syncthreads();  // the previous group's operand reads are done
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          cg_u4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t h, l;
            cg_split2(bv[g][j][e >> 1][2 * (e & 1)], bv[g][j][e >> 1][2 * (e & 1) + 1], h, l);
            hi[e] = h, lo[e] = l;
          }
          const int px = pl + 64 * j, piece = oct ^ ((px >> 2) & 3);
          Bs[((g * 2 + 0) * TN + px) * 4 + piece] = hi;
          Bs[((g * 2 + 1) * TN + px) * 4 + piece] = lo;
        }
      }
    }
    __syncthreads();
    if (step + G < step_hi) fetch_group_b(step + G);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {
        if (step + g + 1 < step_hi) fetch_a(step + g + 1, av[(g + 1) & 1]);      // (G is even: a group starts on set 0)
        cg_u4 bo[2][2];      // [sub-tile j][part]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int px = 32 * wn + 16 * j + l16;
#pragma unroll
          for (int part = 0; part < 2; ++part) bo[j][part] = Bs[((g * 2 + part) * TN + px) * 4 + (q ^ ((px >> 2) & 3))];
        }
#pragma unroll
        for (int term = 0; term < 3; ++term) {      // lo x hi, hi x lo, hi x hi
          const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cg_bf8, av[g & 1][i][pa]),
                                                                  __builtin_bit_cast(cg_bf8, bo[j][pb]), acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // D[row = 4q + r][col = l16] of sub-tile (i, j): the exact kernel's epilogue
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = col0 + 32 * wn + 16 * j + l16;
    if (c >= a.ncols) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r0 = row0 + 32 * wm + 16 * i + 4 * q;
      if (a.partial) {
        *reinterpret_cast<f4*>(a.partial + ((int64_t)blockIdx.z * a.ncols + c) * a.PR + r0) = acc[i][j];
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r0 + r < a.rows) conv_store<DGRAD>(a, c, r0 + r, acc[i][j][r]);
    }
  }
}

// split-K: sum the partial tiles in split order, then the same epilogue.  thread = (col, row quad)
// (Tried in round 5 and dropped: the tile's last-arriving workgroup adding the partial sums itself.  Its workgroups sit on different
//  XCDs; with agent-scope fences every workgroup wrote back / invalidated an L2 (SyncNet forward at batch 1: 0.49 -> 0.99 ms), with
//  write-through stores + system-scope loads + an arrival counter it was correct and deterministic but still slower than this second
//  launch: 0.49 -> 0.55 ms forward, 0.84 -> 1.01 ms loss + gradient.)
template <bool DGRAD>
__global__ static __launch_bounds__(256) void conv_reduce_kernel(ConvArgs a, int splits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int rq = a.PR / 4;
  if (i >= (int64_t)a.ncols * rq) return;
  const int c = (int)(i / rq), r0 = (int)(i % rq) * 4;
  f4 s = *reinterpret_cast<const f4*>(a.partial + (int64_t)c * a.PR + r0);
  for (int k = 1; k < splits; ++k) {
    const f4 p = *reinterpret_cast<const f4*>(a.partial + ((int64_t)k * a.ncols + c) * a.PR + r0);
    s += p;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r0 + r < a.rows) conv_store<DGRAD>(a, c, r0 + r, s[r]);
}

template <bool DGRAD>
static int launch_conv(ConvArgs a, int64_t B, hipStream_t st) {
  a.ncols = (int)(B * (DGRAD ? a.hin * a.win : a.hout * a.wout));
  a.rows = DGRAD ? a.cin : a.cout;
  a.RP = ceil_to(a.rows, 64);
  a.kc = DGRAD ? a.cout : a.cin;
  a.kcp = ceil_to(a.kc, 16);
  a.nchunks = a.kh * a.kw * a.kcp / 16;
  // the split form (a.w16): 64 x 64 or 32 x 128 tiles; layers whose K is a single channel (the mel window) stay exact
  const bool split_form = a.w16 && a.kc >= 8;
  const int TM = split_form ? (a.rows <= 32 ? 32 : 64) : a.rows <= 16 ? 16 : a.rows <= 32 ? 32 : 64, TN = 4096 / TM;
  a.PR = ceil_to(a.rows, TM);      // (narrow layers: a quarter / half of the 64-row padding, so their split-K fits the scratch)
  const int tiles = ((a.ncols + TN - 1) / TN) * ((a.rows + TM - 1) / TM);
  // One tile walks its K range alone, one workgroup of four waves: a CU that holds a single such workgroup streams its
  // operands at ~12 GB/s (dependent fetch -> commit -> MFMA rounds), so a launch of ~150-300 tiles ran at a quarter of the MFMA
  // rate however its inner loop was scheduled (round 3 measurements: grouping chunks, pipelining the LDS reads: +-0).  What it lacks
  // is workgroups in flight: K is split until ~4 workgroups per CU exist (the LDS and register budget of 4), each with >= 8 chunks.
  int splits = 1;
  if (tiles < 1024 && a.nchunks >= 16) {
    // few tiles (deep layers, small batches): fill the chip once, >= 8 chunks each; 128..1023 tiles: towards 4 per CU, but only
    // while a workgroup keeps >= 32 chunks (below that the partial sums cost more than the parallelism returns)
    splits = tiles < 128 ? min(min(256 / tiles, a.nchunks / 8), 64) : min((1024 + tiles - 1) / tiles, a.nchunks / 32);
    splits = max(splits, 1);
    while (splits > 1 && (int64_t)splits * a.ncols * a.PR > kPartialFloats) --splits;
  }
  a.chunks_per_split = (a.nchunks + splits - 1) / splits;
  splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
  if (split_form) {
    a.nsteps = (a.nchunks + 1) / 2;
    a.steps_per_split = (a.nsteps + splits - 1) / splits;
    splits = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
  }
  float* partial = a.partial;
  a.partial = splits > 1 ? partial : nullptr;
  const dim3 grid((a.ncols + TN - 1) / TN, (a.rows + TM - 1) / TM, splits);
  if (split_form && TM == 64) hipLaunchKernelGGL((conv_gemm_split_kernel<DGRAD, 64>), grid, dim3(256), 0, st, a);
  else if (split_form) hipLaunchKernelGGL((conv_gemm_split_kernel<DGRAD, 32>), grid, dim3(256), 0, st, a);
  else if (TM == 16) hipLaunchKernelGGL((conv_gemm_kernel<DGRAD, 16>), grid, dim3(256), 0, st, a);
  else if (TM == 32) hipLaunchKernelGGL((conv_gemm_kernel<DGRAD, 32>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_gemm_kernel<DGRAD, 64>), grid, dim3(256), 0, st, a);
  if (splits > 1) {
    const int64_t n = (int64_t)a.ncols * (a.PR / 4);
    hipLaunchKernelGGL(conv_reduce_kernel<DGRAD>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, splits);
  }
  return (int)hipGetLastError();
}

// floats the split form's A operand takes (hi + lo bf16 parts of the [K16 -> ceil 32][RP] operand)
inline int64_t packed16_floats(int taps, int kc, int rows) { return (int64_t)((taps * ceil_to(kc, 16) + 31) / 32) * 32 * ceil_to(rows, 64); }
inline void launch_pack16(const float* w, float* dst, int taps, int kc, int rows, hipStream_t st) {
  const int K16 = taps * ceil_to(kc, 16), RP = ceil_to(rows, 64);
  const int64_t n = packed16_floats(taps, kc, rows) * 2;
  hipLaunchKernelGGL(conv_pack16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<uint16_t*>(dst), K16, RP, n);
}

inline ConvArgs base_args(const LayerSpec& s, Shape in, Shape out) {
  ConvArgs a{};
  a.hin = in.h, a.win = in.w, a.cin = s.cin, a.hout = out.h, a.wout = out.w, a.cout = s.cout;
  a.kh = s.kh, a.kw = s.kw, a.sy = s.sy, a.sx = s.sx, a.py = s.py, a.px = s.px;
  return a;
}

}  // namespace s2l
