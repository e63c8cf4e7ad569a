// bf16 mode of the lip-MLP training step (BASELINE config 5 names bf16): forward with saved
// activations, backward dz chain and weight-gradient GEMMs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32
// master weights and fp32 gradients.  Same mathematics as rows.hip / train.hip (the fp32 parity mode); operands, saved
// activations and saved gradients are bf16, which makes every one of these kernels HBM-bound instead of MFMA-bound
// (9.7 GB of saved state per direction for 64 frames at 96x96 instead of 38 GB).
//
// Layouts: s2l_bf16.h.  A workgroup (8 waves, two per SIMD: one wave's epilogue runs next to the other's MFMAs) owns 256 rows
// of the batch, a wave 32.  A layer's weights pass through LDS in four 2-block stages; the next stage's global loads are in
// flight in registers while the current one computes (one barrier per stage).  Activations stay in registers between layers
// (kfeat16 trick); per 32-feature block the epilogue (bias = the accumulators' initial value) converts to bf16, applies ReLU
// and extracts its mask on the packed pairs (relu_pk / nz01_pk below: one mask dword per lane and stage, 32 B per row and
// layer); the eight dwords a lane then holds ARE its next-layer B operands, and they are stored as they are (two 16-byte
// stores per lane, 2 KiB contiguous per wave and block): the weight-gradient GEMM reads these row-major images back with the
// LDS transpose read (ds_read_b64_tr_b16).
#include "s2l_common.h"
#include "s2l_bf16.h"

#ifndef S2L_KDEPTH
#define S2L_KDEPTH 3
#endif
#ifndef S2L_ALWAYS_X
#define S2L_ALWAYS_X 0   // experiment: 1 copies both parts of every stage unconditionally (straight-line code, exact vmcnt scoreboard): measured neutral
#endif
#ifndef S2L_EXP
#define S2L_EXP 0   // tools/ubench experiments only: 1 no tile store, 2 no masks, 4 no MFMA k-loops, 16 no dxa stores in the backward (results wrong)
#endif

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

// ReLU and its mask on PACKED bf16 pairs -- no compare / ballot / lane broadcast, i.e. no VALU -> SGPR -> VALU round trip (the
// v_cmp + two v_writelane per value of the first version cost ~60 cycles per value, 2 000 per stage):
//   relu_pk:  max as signed 16-bit integers with 0 (a negative bf16, -0 included, is a negative int16): bf16(relu(x)) exactly,
//             since rounding keeps the sign;
//   nz01_pk:  min as unsigned 16-bit integers with 1: 1 where the half is non-zero (= the ReLU passed), else 0;
//   mul01_pk: halves times 0 / 1 as unsigned 16-bit integers: keeps or clears a bf16 bit pattern.
// A lane's 32 values of a stage (two 32-feature blocks) are 16 dwords d = 8 which + 2 a + p (halves c = 2p, 2p + 1); their mask
// is ONE dword per lane: bit 15 - d for the low half of dword d, bit 31 - d for the high half (built by 16 x "shift left, or").
__device__ __forceinline__ uint32_t relu_pk(uint32_t x) {
  uint32_t o;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(o) : "v"(x));
  return o;
}
__device__ __forceinline__ uint32_t nz01_pk(uint32_t x) {
  uint32_t o;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(o) : "v"(x), "s"(0x00010001u));
  return o;
}
__device__ __forceinline__ uint32_t mul01_pk(uint32_t x, uint32_t m01) {
  uint32_t o;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(x), "v"(m01));
  return o;
}
__device__ __forceinline__ uint32_t mask01_of(uint32_t mword, int d) { return (mword >> (15 - d)) & 0x00010001u; }

// K loop of TWO 32-row output blocks (independent accumulators) against one B operand set: the A quads are read from LDS
// S2L_KDEPTH - 1 k-steps ahead into rotating registers; the sched_barrier pins that order.
template <int T>
__device__ __forceinline__ void kloop2(const u4* ap0, const u4* ap1, const u4 (&b)[T], f16v& acc0, f16v& acc1) {
  constexpr int D = S2L_KDEPTH;
  u4 a0[D], a1[D];
#pragma unroll
  for (int t = 0; t < D - 1; ++t) a0[t] = ap0[64 * t], a1[t] = ap1[64 * t];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + D - 1 < T) a0[(t + D - 1) % D] = ap0[64 * (t + D - 1)], a1[(t + D - 1) % D] = ap1[64 * (t + D - 1)];
    __builtin_amdgcn_sched_barrier(0);
    acc0 = mfma32(a0[t % D], b[t], acc0);
    acc1 = mfma32(a1[t % D], b[t], acc1);
  }
}

// kloop2 with a per-k-step callback issued right after the step's two MFMAs: the callback's VALU work (one slice of the
// PREVIOUS stage's epilogue) executes in their shadow.
template <int T, typename F>
__device__ __forceinline__ void kloop2e(const u4* ap0, const u4* ap1, const u4 (&b)[T], f16v& acc0, f16v& acc1, F&& slice) {
  constexpr int D = S2L_KDEPTH;
  u4 a0[D], a1[D];
#pragma unroll
  for (int t = 0; t < D - 1; ++t) a0[t] = ap0[64 * t], a1[t] = ap1[64 * t];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + D - 1 < T) a0[(t + D - 1) % D] = ap0[64 * (t + D - 1)], a1[(t + D - 1) % D] = ap1[64 * (t + D - 1)];
    __builtin_amdgcn_sched_barrier(0);
    acc0 = mfma32(a0[t % D], b[t], acc0);
    acc1 = mfma32(a1[t % D], b[t], acc1);
    slice(t);
  }
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
        v = pf[(L == 0 ? OFF_G0 : OFF_G5) + (int64_t)(32 * R + (lane & 31)) * kGenK + kfeat16(t, lane >> 5, j)];
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
// Eight waves per workgroup, two per SIMD (<= 256 registers each): while one wave of a SIMD runs the VALU epilogue of a
// block the other one issues MFMAs.  Wave w owns rows 32 w .. 32 w + 31 of the workgroup's 256; waves 2i, 2i+1 share the
// 64-row tile 4 tile + i (each writes its 64-byte half of every 128-byte feature row).
constexpr int kLdsW = 2 * kStageF * 2;          // two stage buffers, bytes
constexpr int kLdsFwd = kLdsW + (8 * 256 + 4) * 4;

struct Stage {
  u4 x[2], h[4];   // 16 B pieces of the X part (16 KiB) and the H part (32 KiB), piece = tid + 512 k
};
__device__ __forceinline__ void stage_gload(Stage& st, const uint16_t* src, int tid, bool ld_x, bool ld_h) {
  const u4* p = reinterpret_cast<const u4*>(src) + tid;
  if (ld_x) {
#pragma unroll
    for (int k = 0; k < 2; ++k) st.x[k] = p[512 * k];
  }
  if (ld_h) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st.h[k] = p[1024 + 512 * k];
  }
}
__device__ __forceinline__ void stage_lstore(const Stage& st, uint16_t* dst, int tid, bool ld_x, bool ld_h) {
  u4* p = reinterpret_cast<u4*>(dst) + tid;
  if (ld_x) {
#pragma unroll
    for (int k = 0; k < 2; ++k) p[512 * k] = st.x[k];
  }
  if (ld_h) {
#pragma unroll
    for (int k = 0; k < 4; ++k) p[1024 + 512 * k] = st.h[k];
  }
}

// Store one finished 32-feature block of this wave's 32 rows: the lane's eight bf16 pairs, as held (s2l_bf16.h image).
__device__ __forceinline__ void image_store(const uint32_t (&vals)[4][2], uint16_t* gdst, int lane) {
  u4* g = reinterpret_cast<u4*>(gdst) + lane;      // two contiguous 1-KiB planes per block (s2l_bf16.h)
  g[0] = u4{vals[0][0], vals[0][1], vals[1][0], vals[1][1]};
  g[64] = u4{vals[2][0], vals[2][1], vals[3][0], vals[3][1]};
}

#ifdef S2L_TRACE16   // experiment build: per-stage phase timestamps of waves 0 and 4 of workgroup 0 (s_memtime)
__device__ unsigned long long* g_trace16 = nullptr;
#define T16(slot)                                                                                               \
  do {                                                                                                          \
    if (g_trace16 && blockIdx.x == 0 && (threadIdx.x & 255) == 0 && tile == 0)                                  \
      g_trace16[((threadIdx.x >> 8) * 32 + s) * 8 + (slot)] = __builtin_readcyclecounter();                     \
  } while (0)
extern "C" void s2l_trace16_set(unsigned long long* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_trace16), &p, sizeof p); }
#else
#define T16(slot) do { } while (0)
#endif

struct FwdArgs {
  const uint16_t* wb;
  const float* pf;
  const uint16_t* xT;   // embedded rows as a bf16 image (4 blocks), padded rows zero
  uint16_t* hT;
  uint64_t* masks;
  float* rgb;
  int64_t n_rows, layer_stride, mask_layer_stride;
  int n_tiles;
};

__global__ __launch_bounds__(512, 1) void fwd_bf16_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* wbuf = reinterpret_cast<uint16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5, g = wave & 1;
  float* bias = reinterpret_cast<float*>(smem + kLdsW);
  for (int i = tid; i < 8 * 256; i += 512) {
    const int L = i >> 8, f = i & 255;
    bias[i] = L == 0 ? a.pf[OFF_BG0 + f] : L == 5 ? a.pf[OFF_BG5 + f] : a.pf[OFF_BIAS + (L - 1) * 256 + f];
  }
  if (tid < 4) bias[2048 + tid] = a.pf[OFF_BOUT + tid];

  Stage st;
  stage_gload(st, a.wb + OFF_FWD, tid, true, false);
  stage_lstore(st, wbuf, tid, true, false);
  __syncthreads();
  u4 bx[8], bcur[16], bnext[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bcur[t] = u4{0u, 0u, 0u, 0u};

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t row = (int64_t)tile * kWgRows + 32 * wave + n;
    const int64_t tile64 = (int64_t)tile * 4 + (wave >> 1), group = (int64_t)tile * 8 + wave;
    {   // B operands of the embedded rows: the image's two 16-byte halves of block R are k-steps 2R and 2R + 1
      const u4* xi = reinterpret_cast<const u4*>(a.xT + image_off(group, 4, 0)) + lane;
#pragma unroll
      for (int R = 0; R < 4; ++R) {
        bx[2 * R] = xi[R * 128];
        bx[2 * R + 1] = xi[R * 128 + 64];
      }
    }
    for (int L = 0; L < 8; ++L) {
      const bool use_x = L == 0 || L == 5, use_h = L != 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = 4 * L + q, nxt = (s + 1) & 31;
        const int nL = nxt >> 2;
        const bool nx = nL == 0 || nL == 5 || nxt == 31, nh = nL != 0;
        T16(0);
        stage_gload(st, a.wb + OFF_FWD + (int64_t)nxt * kStageF, tid, S2L_ALWAYS_X || nx, S2L_ALWAYS_X || nh);
        const uint16_t* wl = wbuf + (q & 1) * kStageF;
        f16v acc[2];   // blocks R = 2q, 2q + 1, initialised with the bias (feature 32R + 8a + 4hh + c <-> register 4a + c)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const float* bl = bias + L * 256 + 32 * (2 * q + w2) + 4 * hh;
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            const f4 bv = *reinterpret_cast<const f4*>(bl + 8 * a4);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[w2][4 * a4 + c] = bv[c];
          }
        }
        if (use_x && !(S2L_EXP & 4))
          kloop2<8>(reinterpret_cast<const u4*>(wl) + lane, reinterpret_cast<const u4*>(wl + kSlabX) + lane, bx, acc[0], acc[1]);
        if (use_h && !(S2L_EXP & 4))
          kloop2<16>(reinterpret_cast<const u4*>(wl + 2 * kSlabX) + lane, reinterpret_cast<const u4*>(wl + 2 * kSlabX + kSlabH) + lane,
                     bcur, acc[0], acc[1]);
        T16(1);
        // epilogue: bf16, ReLU and the mask dword on packed pairs (relu_pk / nz01_pk above)
        uint32_t mword = 0;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const int R = 2 * q + which;
          uint32_t vals[4][2];
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              const uint32_t v = relu_pk(pk2(acc[which][4 * a4 + 2 * p2], acc[which][4 * a4 + 2 * p2 + 1]));
              if (!(S2L_EXP & 2)) mword = (mword << 1) | nz01_pk(v);
              vals[a4][p2] = v;
            }
          }
          bnext[2 * R] = u4{vals[0][0], vals[0][1], vals[1][0], vals[1][1]};
          bnext[2 * R + 1] = u4{vals[2][0], vals[2][1], vals[3][0], vals[3][1]};
          if (!(S2L_EXP & 1))
            image_store(vals, a.hT + L * a.layer_stride + image_off(group, 8, R), lane);
        }
        if (!(S2L_EXP & 2))   // dword (layer, 32-row group, stage, lane)
          reinterpret_cast<uint32_t*>(a.masks + L * a.mask_layer_stride)[group * 256 + q * 64 + lane] = mword;
        if (s == 31) {  // output layer on h7 (= bnext), weights in the X part of this stage; rows 0..2 of block 0
          f16v ao;
#pragma unroll
          for (int r = 0; r < 16; ++r) ao[r] = 0.f;
          const u4* ap = reinterpret_cast<const u4*>(wl) + lane;
#pragma unroll
          for (int t = 0; t < 16; ++t) ao = mfma32(ap[64 * t], bnext[t], ao);
          if (hh == 0 && row < a.n_rows) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rgb[row * 3 + c] = ao[c] + bias[2048 + c];
          }
        }
        T16(2);
        stage_lstore(st, wbuf + ((q + 1) & 1) * kStageF, tid, S2L_ALWAYS_X || nx, S2L_ALWAYS_X || nh);
        T16(3);
        __syncthreads();
        T16(4);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) bcur[t] = bnext[t];
    }
  }
}

// ---- the forward kernel with its body as one fixed-register assembly text (csrc/gen_fwd16_body.py): 64 rows per wave, one wave
// per SIMD, stages by LDS-DMA, every memory instruction behind an MFMA.  Same arithmetic in the same order as fwd_bf16_kernel:
// images, mask dwords and rgb are bit-identical (tests/test_gpu_parity.py switches between the two with
// s2l_set_bf16_forward_kernel).  3.04 ms against the C++ kernel's 3.3-3.4 ms for 64 frames of 96x96 (tools/bench_bf16_kernels.py);
// tools/trace_fwd16.py (experiment build) gives the phases of a stage: k-loop 2 250 cycles (64 MFMAs = 2 048), wait 110, barrier
// 120, epilogue 1 180 (its eight image stores keep the texture path busy for 1 280), 110 to the next k-loop.
#ifndef S2L_FWD_ASM
#define S2L_FWD_ASM 1
#endif
#ifdef S2L_EXP_TRACE   // experiment builds (generator run with S2L_FWD_TRACE=1, tools/trace_fwd16.py): per-stage timestamps
__device__ unsigned long long* g_ftrace = nullptr;
extern "C" int s2l_debug_set_fwd_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ftrace), &p, sizeof(p)); }
#endif
__global__ __launch_bounds__(256, 1) void fwd_asm_bf16_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias = reinterpret_cast<float*>(smem + kLdsW);
  for (int i = threadIdx.x; i < 8 * 256; i += 256) {
    const int L = i >> 8, f = i & 255;
    bias[i] = L == 0 ? a.pf[OFF_BG0 + f] : L == 5 ? a.pf[OFF_BG5 + f] : a.pf[OFF_BIAS + (L - 1) * 256 + f];
  }
  if (threadIdx.x < 4) bias[2048 + threadIdx.x] = a.pf[OFF_BOUT + threadIdx.x];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ldsbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const int tile0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x), grid = __builtin_amdgcn_readfirstlane((int)gridDim.x);
#if defined(__HIP_DEVICE_COMPILE__)
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();   // the body loads FwdArgs fields itself (s_load)
#else
  const void* karg = nullptr;   // (host pass of the compiler: never executed)
#endif
#include "fwd16_body.inc"
}

// ---- backward dz chain ------------------------------------------------------------------------------------------------------
// g_7 = (Wout^T drgb) . m_7;  g_{l-1} = (W_l^T g_l) . m_{l-1} for l = 7..1 (l = 5: the h_4 half of pts_linears[5]);
// d audio = G5[:, audio]^T g_5 + G0[:, audio]^T g_0.  Every g_l is stored as a [feature][64 rows] bf16 tile (dzT) for the
// weight-gradient GEMMs; the masks are the forward's per-lane mask dwords, applied to the packed bf16 pairs (mul01_pk).
constexpr int kLdsBwdW = 2 * kStageB * 2;
constexpr int kLdsBwd = kLdsBwdW + 8 * kSlabU0 * 2;

struct BwdArgs {
  const uint16_t* wb;
  const float* drgb;
  const uint64_t* masks;
  uint16_t* dzT;
  float* dxa;
  int64_t n_rows, layer_stride, mask_layer_stride;
  int n_tiles;
};

// masked gradient block -> bf16 pairs; mword = this lane's mask dword of the stage the block belongs to (block `which` of it)
__device__ __forceinline__ void mask_block(const f16v& acc, uint32_t mword, int which, uint32_t (&vals)[4][2]) {
#pragma unroll
  for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2)
      vals[a4][p2] = mul01_pk(pk2(acc[4 * a4 + 2 * p2], acc[4 * a4 + 2 * p2 + 1]), mask01_of(mword, 8 * which + 2 * a4 + p2));
}

__global__ __launch_bounds__(512, 1) void bwd_bf16_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* wbuf = reinterpret_cast<uint16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5, g = wave & 1;
  u4* u0 = reinterpret_cast<u4*>(smem + kLdsBwdW);
  if (tid < 8 * 64) u0[tid] = reinterpret_cast<const u4*>(a.wb + OFF_BWD_U0)[tid];

  u4 st[4];
  auto gload = [&](int u) {
    const u4* p = reinterpret_cast<const u4*>(a.wb + OFF_BWD_H + (int64_t)u * kStageB) + tid;
#pragma unroll
    for (int k = 0; k < 4; ++k) st[k] = p[512 * k];
  };
  auto lstore = [&](int u) {
    u4* p = reinterpret_cast<u4*>(wbuf + (u & 1) * kStageB) + tid;
#pragma unroll
    for (int k = 0; k < 4; ++k) p[512 * k] = st[k];
  };
  gload(0);
  lstore(0);
  __syncthreads();

  u4 bcur[16], bnext[16];
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t row = (int64_t)tile * kWgRows + 32 * wave + n;
    const int64_t tile64 = (int64_t)tile * 4 + (wave >> 1), group = (int64_t)tile * 8 + wave;
    // drgb as a K = 16 B operand: k = 8 hh + j, k < 3 used
    u4 b0;
    {
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
      if (hh == 0 && row < a.n_rows) d0 = a.drgb[row * 3], d1 = a.drgb[row * 3 + 1], d2 = a.drgb[row * 3 + 2];
      b0 = u4{pk2(d0, d1), pk2(d2, 0.f), 0u, 0u};
    }
    // U0: g_7
    uint32_t m7 = 0;
#pragma unroll
    for (int R = 0; R < 8; ++R) {
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = mfma32(u0[R * 64 + lane], b0, acc);
      uint32_t vals[4][2];
      if (!(R & 1)) m7 = reinterpret_cast<const uint32_t*>(a.masks + 7 * a.mask_layer_stride)[group * 256 + (R >> 1) * 64 + lane];
      mask_block(acc, m7, R & 1, vals);
      bcur[2 * R] = u4{vals[0][0], vals[0][1], vals[1][0], vals[1][1]};
      bcur[2 * R + 1] = u4{vals[2][0], vals[2][1], vals[3][0], vals[3][1]};
      image_store(vals, a.dzT + 7 * a.layer_stride + image_off(group, 8, R), lane);
    }

    int u = 0;
    auto next_of = [](int v) { return v + 1 == kBwdStages ? 0 : v + 1; };
    // one stage: two 32-dim blocks of G[:, audio]^T against bcur; the result goes to (first = g_5) or is added to (g_0) dxa
    // [row][64]: lane holds dims 32R + 8a + 4hh + c of row n
    auto audio_stage = [&](bool first) {
      const int nxt = next_of(u);
      gload(nxt);
      const uint16_t* wl = wbuf + (u & 1) * kStageB;
      f16v acc_a[2];
#pragma unroll
      for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_a[R][r] = 0.f;
      kloop2<16>(reinterpret_cast<const u4*>(wl) + lane, reinterpret_cast<const u4*>(wl + kSlabH) + lane, bcur, acc_a[0], acc_a[1]);
      if (row < a.n_rows && !(S2L_EXP & 16)) {
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            f4* dst = reinterpret_cast<f4*>(a.dxa + row * kAud + 32 * R + 8 * a4 + 4 * hh);
            f4 v = f4{acc_a[R][4 * a4], acc_a[R][4 * a4 + 1], acc_a[R][4 * a4 + 2], acc_a[R][4 * a4 + 3]};
            if (!first) v += *dst;
            *dst = v;
          }
      }
      lstore(nxt);
      __syncthreads();
      u = nxt;
    };

    // Software pipeline inside a layer: the masked-ReLU / bf16 / store epilogue of stage q-1 is sliced over the 16 k-steps
    // of stage q (two accumulator sets ping-pong), so its VALU work runs in the shadow of that stage's MFMAs; only the first
    // stage's k-loops and the last stage's epilogue of a layer run alone (the next layer needs all 256 outputs).
    for (int l = 7; l >= 1; --l) {   // W_l^T g_l -> g_{l-1}
      f16v accs[2][2];
      const uint32_t* mbase = reinterpret_cast<const uint32_t*>(a.masks + (l - 1) * a.mask_layer_stride) + group * 256 + lane;
      uint16_t* dbase = a.dzT + (l - 1) * a.layer_stride + image_off(group, 8, 0);
      uint32_t pv[2][4][2];          // bf16 pairs of the stage whose epilogue is in flight
      uint32_t pm = 0;               // this lane's mask dword of that stage
      // one pair (registers 2 j, 2 j + 1 of block `which`) of the pending epilogue
      auto slice_of = [&](const f16v (&accp)[2], int which, int j) {
        pv[which][j >> 1][j & 1] = mul01_pk(pk2(accp[which][2 * j], accp[which][2 * j + 1]), mask01_of(pm, 8 * which + j));
      };
      auto finish = [&](int qp) {    // bnext entries and image stores of stage qp
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const int R = 2 * qp + which;
          bnext[2 * R] = u4{pv[which][0][0], pv[which][0][1], pv[which][1][0], pv[which][1][1]};
          bnext[2 * R + 1] = u4{pv[which][2][0], pv[which][2][1], pv[which][3][0], pv[which][3][1]};
          image_store(pv[which], dbase + image_off(0, 8, R), lane);
        }
      };
      auto load_masks = [&](int qp) { pm = mbase[qp * 64]; };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nxt = next_of(u);
        gload(nxt);
        const uint16_t* wl = wbuf + (u & 1) * kStageB;
        f16v (&acc)[2] = accs[q & 1];
        const f16v (&accp)[2] = accs[(q & 1) ^ 1];
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[w2][r] = 0.f;
        if (q > 0) load_masks(q - 1);
        kloop2e<16>(reinterpret_cast<const u4*>(wl) + lane, reinterpret_cast<const u4*>(wl + kSlabH) + lane, bcur, acc[0], acc[1],
                    [&](int t) {      // 16 pairs over the 16 k-steps
                      if (q > 0) slice_of(accp, t >> 3, t & 7);
                    });
        if (q > 0) finish(q - 1);
        if (q == 3) {                // this layer's last stage: its own epilogue now
          load_masks(3);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            slice_of(acc, 0, j);
            slice_of(acc, 1, j);
          }
          finish(3);
        }
        lstore(nxt);
        __syncthreads();
        u = nxt;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) bcur[t] = bnext[t];
      if (l == 6) audio_stage(true);   // bcur = g_5
    }
    audio_stage(false);              // bcur = g_0; u wraps to 0 for the next tile
  }
}

// ---- the backward kernel with its body as one fixed-register assembly text (csrc/gen_bwd16_body.py): 64 rows per wave, one wave
// per SIMD, stages by LDS-DMA, every memory instruction behind an MFMA -- the twin of fwd_asm_bf16_kernel.  The dz images are
// bit-identical to bwd_bf16_kernel's.  The audio gradient leaves this kernel as the column sums of each 256-row tile
// (BwdArgs::dxa = dxa_tiles [n_tiles][64]) instead of per row: the caller uses it when a tile never straddles two frames.
constexpr int kLdsBwdAsm = 2 * kStageB * 2 + 8 * kSlabU0 * 2 + 1024;      // two stage buffers, Wout^T, the four waves' column sums
__global__ __launch_bounds__(256, 1) void bwd_asm_bf16_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ldsbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const int tile0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x), grid = __builtin_amdgcn_readfirstlane((int)gridDim.x);
#if defined(__HIP_DEVICE_COMPILE__)
  const void* karg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();   // the body loads BwdArgs fields itself (s_load)
#else
  const void* karg = nullptr;   // (host pass of the compiler: never executed)
#endif
#include "bwd16_body.inc"
}

// ---- weight gradients ---------------------------------------------------------------------------------------------------------
// dW[m][k] = sum_rows dz[row][m] in[row][k]: both MFMA operands want "8 consecutive rows of one feature" per lane.  The images
// are row-major per (block, hh): [32 rows][16 features], 32-byte rows; ds_read_b64_tr_b16 reads a [4 rows][16 features] block
// per 16-lane group (lane i passes the address of row i>>2, feature group i&3 and receives column i: that feature's 4 rows),
// so two of them give a lane its 8 rows.  A 16-lane group G = lane>>4 serves operand rows mu = 16 (G&1) + i of k-half G>>1;
// it reads the hh = G&1 half of the block, hence operand row mu is feature fperm(mu) of the block -- a fixed permutation that
// is undone when the result is stored.  A workgroup streams 64 rows (two row groups: dz 32 KiB, in 32 or 16 KiB) per step
// through LDS, double-buffered through registers (halves padded to 1152 B so the two groups of an LDS cycle hit different
// banks); its 4 waves own 128 x (K/2) of the 256 x K result each.  The kernel is HBM-bound by construction; per-workgroup
// partial sums are reduced in a fixed order by wgrad_reduce_kernel (deterministic).  The bias gradient (column sums of dz)
// rides along on the A operands.
constexpr int kWgParts = 256;                      // max workgroups = partial results
constexpr int kHalfBytes = 1024 + 128;             // one (block, hh) half in LDS: 32 rows x 32 B + pad
__host__ __device__ constexpr int fperm(int mu) { return 8 * ((mu & 15) >> 2) + 4 * (mu >> 4) + (mu & 3); }
template <int KB>
struct WgCfg {
  static constexpr int kNB = KB / 64;              // 32-column blocks per wave
  static constexpr int kBBlocks = KB / 32;         // blocks of the B image
  static constexpr int kBPieces = KB * 8 / 256;    // 16-byte pieces of the B chunk per thread
  static constexpr int kBufBytes = 2 * (8 + kBBlocks) * 2 * kHalfBytes;   // two row groups of A and B
  static constexpr int kLds = 2 * kBufBytes;
};

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

typedef uint32_t u2 __attribute__((ext_vector_type(2)));
// 8 rows (n0..n0+7) of this lane's operand feature: two transpose reads 4 rows apart (128 bytes).  The compiler does not
// track inline-asm LDS reads: the results may only be touched after tr_wait(), which names them as in/out operands so that
// no use can be scheduled above the s_waitcnt.
struct TrPair {
  u2 lo, hi;
};
__device__ __forceinline__ void tr_read8(TrPair& t, uint32_t lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:128" : "=&v"(t.lo), "=&v"(t.hi) : "v"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait(TrPair (&t)[N]) {
  if constexpr (N == 8)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0].lo), "+v"(t[0].hi), "+v"(t[1].lo), "+v"(t[1].hi), "+v"(t[2].lo), "+v"(t[2].hi), "+v"(t[3].lo), "+v"(t[3].hi),
                   "+v"(t[4].lo), "+v"(t[4].hi), "+v"(t[5].lo), "+v"(t[5].hi), "+v"(t[6].lo), "+v"(t[6].hi), "+v"(t[7].lo), "+v"(t[7].hi)
                 :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0].lo), "+v"(t[0].hi), "+v"(t[1].lo), "+v"(t[1].hi), "+v"(t[2].lo), "+v"(t[2].hi), "+v"(t[3].lo), "+v"(t[3].hi),
                   "+v"(t[4].lo), "+v"(t[4].hi), "+v"(t[5].lo), "+v"(t[5].hi)
                 :: "memory");
}
__device__ __forceinline__ u4 tr_u4(const TrPair& t) { return u4{t.lo[0], t.lo[1], t.hi[0], t.hi[1]}; }

template <int KB>
__global__ __launch_bounds__(256, 1) void wgrad_bf16_kernel(const uint16_t* __restrict__ dzT, const uint16_t* __restrict__ inT,
                                                            float* __restrict__ part, float* __restrict__ bpart, int n_tiles) {
  using C = WgCfg<KB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1, n = lane & 31, hh = lane >> 5;
  u4 sa[8], sb[C::kBPieces];
  // a 64-row chunk of an image with NBLK blocks: thread-piece p = ((rg * NBLK + R) * 64 + L) * 2 + h16 is the 16-byte piece h16
  // of lane L of block rR; in memory that piece sits at index mem_of(p) (the two pieces of a lane are 1 KiB apart, s2l_bf16.h),
  // in LDS the two are adjacent again (32-byte rows) -- consecutive threads write consecutive 16 bytes of LDS and read two
  // 512-byte runs of memory per wave instruction
  auto mem_of = [](int p) { return (p & ~127) | ((p & 1) << 6) | ((p >> 1) & 63); };
  auto lds_of = [](int p, int nblk, int base_half) {
    const int h16 = p & 1, L = (p >> 1) & 63, rR = p >> 7;   // rR = rg * nblk + R
    return (base_half + rR * 2 + (L >> 5)) * kHalfBytes + (L & 31) * 32 + h16 * 16;
  };
  auto gload = [&](int tile) {
    const u4* pa = reinterpret_cast<const u4*>(dzT + (int64_t)tile * 64 * 256);
    const u4* pb = reinterpret_cast<const u4*>(inT + (int64_t)tile * 64 * KB);
    // both images are streamed exactly once per launch: non-temporal loads keep them out of the caches' retention policy
#pragma unroll
    for (int k = 0; k < 8; ++k) sa[k] = __builtin_nontemporal_load(pa + mem_of(tid + 256 * k));
#pragma unroll
    for (int k = 0; k < C::kBPieces; ++k) sb[k] = __builtin_nontemporal_load(pb + mem_of(tid + 256 * k));
  };
  auto lstore = [&](int buf) {
    char* base = smem + buf * C::kBufBytes;
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<u4*>(base + lds_of(tid + 256 * k, 8, 0)) = sa[k];
#pragma unroll
    for (int k = 0; k < C::kBPieces; ++k) *reinterpret_cast<u4*>(base + lds_of(tid + 256 * k, C::kBBlocks, 2 * 8 * 2)) = sb[k];
  };
  f16v acc[4][C::kNB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < C::kNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // per-lane byte offset inside a (row group, block) of the LDS image: 16-lane group G reads half G&1, rows 8 (G>>1) + (i>>2)
  const int i16 = lane & 15, G = lane >> 4;
  const uint32_t lane_off = (uint32_t)((G & 1) * kHalfBytes + (8 * (G >> 1) + (i16 >> 2)) * 32 + (i16 & 3) * 8);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  int tile = blockIdx.x;
  if (tile < n_tiles) {
    gload(tile);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < n_tiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < n_tiles) gload(nxt);
    const uint32_t bb = lds0 + buf * C::kBufBytes + lane_off;
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // k-step: row group s>>1, rows 16 (s&1) .. +15
      const int rg = s >> 1;
      const uint32_t row_off = (uint32_t)(16 * (s & 1) * 32);
      TrPair tp[4 + C::kNB];
#pragma unroll
      for (int i = 0; i < 4; ++i) tr_read8(tp[i], bb + ((rg * 8 + 4 * wm + i) * 2) * kHalfBytes + row_off);
#pragma unroll
      for (int j = 0; j < C::kNB; ++j)
        tr_read8(tp[4 + j], bb + (2 * 8 * 2 + (rg * C::kBBlocks + C::kNB * wn + j) * 2) * kHalfBytes + row_off);
      tr_wait(tp);
      u4 av[4], bv[C::kNB];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = tr_u4(tp[i]);
#pragma unroll
      for (int j = 0; j < C::kNB; ++j) bv[j] = tr_u4(tp[4 + j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < C::kNB; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
      if (wn == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int d = 0; d < 4; ++d) bsum[i] += bf_lo(av[i][d]) + bf_hi(av[i][d]);
      }
    }
    if (nxt < n_tiles) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial result of this workgroup: D[mu = 8(r>>2) + 4hh + (r&3)][nu = n] of block (i, j); operand rows are permuted features
  float* po = part + (int64_t)blockIdx.x * 256 * KB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < C::kNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 128 * wm + 32 * i + fperm(8 * (r >> 2) + 4 * hh + (r & 3));
        po[m * KB + (KB / 2) * wn + 32 * j + fperm(n)] = acc[i][j][r];
      }
  if (wn == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = bsum[i] + __shfl_xor(bsum[i], 32);
      if (hh == 0) bpart[blockIdx.x * 256 + 128 * wm + 32 * i + fperm(n)] = t;
    }
  }
}

// out[i] = sum over the parts in a fixed order: 16 float4 columns per block, the parts dealt round-robin to 16 threads per column
// (thread p takes parts p, p + 16, ...), whose sums are then added in thread order.  A second, small tensor (the bias gradient's
// partials) rides in the same launch: the blocks behind the first tensor's take it.  (With 64 columns x 4 part-quarters per
// block a thread summed 64 parts one after the other: 19 us per GEMM, ten times per step.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int parts,
                                                          const float* __restrict__ part2, float* __restrict__ out2, int n2) {
  __shared__ f4 red[16][17];
  const int col = threadIdx.x & 15, w = threadIdx.x >> 4;
  const int nb1 = (n / 4 + 15) / 16;
  int blk = blockIdx.x;
  if (blk >= nb1) {     // (block-uniform)
    blk -= nb1;
    part = part2; out = out2; n = n2;
  }
  const int i = (blk * 16 + col) * 4;
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
  if (i < n)
    for (int p = w; p < parts; p += 16) s += *reinterpret_cast<const f4*>(part + (int64_t)p * n + i);
  red[w][col] = s;
  __syncthreads();
  if (w == 0 && i < n) {
    f4 acc = red[0][col];
#pragma unroll
    for (int k = 1; k < 16; ++k) acc += red[k][col];
    *reinterpret_cast<f4*>(out + i) = acc;
  }
}

// x fp32 [N,K] -> bf16 image (K/32 blocks; K = 128: the embedded rows, for dG0 / dG5).  thread = (row, 4 features).
__global__ __launch_bounds__(256) void rows_to_image_kernel(const float* __restrict__ x, uint16_t* __restrict__ xT, int K,
                                                            int64_t n_rows, int64_t n_padded) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kq = K / 4;
  const int64_t row = t / kq;
  if (row >= n_padded) return;
  const int f = (int)(t - row * kq) * 4;
  f4 v = f4{0.f, 0.f, 0.f, 0.f};
  if (row < n_rows) v = *reinterpret_cast<const f4*>(x + row * K + f);
  const int R = f >> 5, a4 = (f & 31) >> 3, hh = (f >> 2) & 1;
  uint16_t* dst = xT + image_off(row >> 5, K / 32, R) + image_quad(a4, (int)(row & 31) + 32 * hh);
  *reinterpret_cast<u2*>(dst) = u2{pk2(v[0], v[1]), pk2(v[2], v[3])};
}

// 4-tap ensemble rows of a whole batch of frames, straight to the bf16 image (no fp32 x, no second pass): the same values as
// ensemble_rows_kernel (ensemble.hip; training.py:198-210, 240-241) rounded to bf16.  Row of (frame b, tap t, pixel p) =
// (4 b + t) * n_pixels + p.  thread = (row, 4 features).
struct EnsBatch {
  float dx[2], dy[2], ry;   // (float)(-rx), (float)(+rx), (float)(-ry), (float)(+ry), (float)ry: the host's python-double shifts
};
__device__ __forceinline__ float embed16(float u, float v, int i) {
  if (i < 2) return i == 0 ? u : v;
  const int blk = (i - 2) >> 1;
  const float x = ((i & 1) ? v : u) * (float)(1 << (blk >> 1));
  return (blk & 1) ? cosf(x) : sinf(x);
}
__global__ __launch_bounds__(256) void ensemble_rows_bf16_kernel(const float* __restrict__ packed, const float* __restrict__ coords,
                                                                const float* __restrict__ feat, const int64_t* __restrict__ tidx,
                                                                const float* __restrict__ u01, EnsBatch sh, uint16_t* __restrict__ xT,
                                                                float* __restrict__ areas, int64_t n_pix, int64_t n_rows,
                                                                int64_t n_padded) {
  // thread = one 16-byte piece of the image: (row group, block R, half, lane = row + 32 hh) holds features
  // 32 R + 16 half + 4 hh + {0..3} and the same + 8; consecutive threads write consecutive 16 bytes
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t grp = t >> 9;
  const int p = (int)(t & 511), R = p >> 7, half = (p >> 6) & 1, lane = p & 63, hh = lane >> 5;
  const int64_t row = grp * kGroupRows + (lane & 31);
  if (row >= n_padded) return;
  const int k0 = 32 * R + 16 * half + 4 * hh;
  float val[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < n_rows) {
    const int64_t bt = row / n_pix, px = row - bt * n_pix;
    const int tap = (int)(bt & 3);
    const int64_t b = bt >> 2;
    const float eps = sh.ry * u01[b] / 2.0f;
    const float u0 = coords[2 * px], v0 = coords[2 * px + 1];
    const float cu = fminf(fmaxf(u0 + (sh.dx[tap >> 1] + eps), 0.f), 1.f);
    const float cv = fminf(fmaxf(v0 + (sh.dy[tap & 1] + eps), 0.f), 1.f);
    const float time_pos = (float)tidx[b];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + (j & 3) + 8 * (j >> 2);
      float v;
      if (k < kEmb) v = embed16(cu, cv, k);
      else if (k < kEmb + kAud) v = feat[b * kAud + (k - kEmb)];
      else if (k < kEmb + kAud + kTime) {
        const int i = k - kEmb - kAud;
        const float arg = time_pos * packed[OFF_DIV + (i >> 1)];
        v = (i & 1) ? cosf(arg) : sinf(arg);
      } else v = 0.f;
      val[j] = v;
    }
    if (k0 == 0) areas[row] = fabsf((cu - u0) * (cv - v0)) + 1e-9f;
  }
  *reinterpret_cast<u4*>(xT + image_off(grp, 4, R) + image_piece(half, lane)) =
      u4{pk2(val[0], val[1]), pk2(val[2], val[3]), pk2(val[4], val[5]), pk2(val[6], val[7])};
}

// output layer: dWout[c][f] = sum_rows drgb[row][c] h7[row][f], dbout[c] = sum_rows drgb[row][c].  A row group's h7 image is
// 16 KiB = 1024 pieces of 16 bytes; thread t reads pieces t + 256 k (1 KiB of contiguous memory per wave instruction): the same
// (half, lane = row + 32 hh) of blocks R = (t >> 7) + 2 k -- 8 features x 4 blocks of ONE row, 96 running sums per thread over
// all the workgroup's row groups; the 32 rows are combined through LDS once, at the end, in row order.  (The first version read
// 8 bytes per thread and row, 64 scattered sectors per wave instruction: 0.46 ms for 1.2 GB.)
__global__ __launch_bounds__(256) void out_grad_kernel(const float* __restrict__ drgb, const uint16_t* __restrict__ h7T,
                                                       float* __restrict__ part, int n_groups, int64_t n_rows) {
  __shared__ float red[32][33];      // [row][feature of a 32-feature slice] (+1: conflict-free column reads)
  const int t = threadIdx.x, lane = t & 63, half = (t >> 6) & 1, r0 = t >> 7;
  const int row = lane & 31, hh = lane >> 5;
  float s[4][8][3] = {};             // [k: block r0 + 2 k][value 4 a1 + c of the piece][rgb]
  float sb[3] = {0.f, 0.f, 0.f};     // (threads t < 32 only: one per row)
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {      // no barrier in here: the loads of successive groups overlap
    const int64_t idx = ((int64_t)grp * kGroupRows + row) * 3;
    const bool live = idx < n_rows * 3;
    const float d0 = live ? drgb[idx] : 0.f, d1 = live ? drgb[idx + 1] : 0.f, d2 = live ? drgb[idx + 2] : 0.f;
    const u4* src = reinterpret_cast<const u4*>(h7T + image_off(grp, 8, 0)) + t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u4 w = src[256 * k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = bf_lo(w[j]), hi = bf_hi(w[j]);
        s[k][2 * j][0] = fmaf(d0, lo, s[k][2 * j][0]);
        s[k][2 * j][1] = fmaf(d1, lo, s[k][2 * j][1]);
        s[k][2 * j][2] = fmaf(d2, lo, s[k][2 * j][2]);
        s[k][2 * j + 1][0] = fmaf(d0, hi, s[k][2 * j + 1][0]);
        s[k][2 * j + 1][1] = fmaf(d1, hi, s[k][2 * j + 1][1]);
        s[k][2 * j + 1][2] = fmaf(d2, hi, s[k][2 * j + 1][2]);
      }
    }
    if (t < 32) sb[0] += d0, sb[1] += d1, sb[2] += d2;
  }
  // value v = 4 a1 + c of the piece (half, lane) of block R is feature 32 R + 8 (2 half + a1) + 4 hh + c of row `row`:
  // per (k, rgb) the workgroup holds a [32 rows][32 features] slice per r0 -- reduce it over the rows in row order
  float* po = part + (int64_t)blockIdx.x * (3 * 256 + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {           // blocks R = rr + 2 k: threads with r0 == rr publish
        __syncthreads();
        if (r0 == rr) {
#pragma unroll
          for (int v = 0; v < 8; ++v) red[row][8 * (2 * half + (v >> 2)) + 4 * hh + (v & 3)] = s[k][v][c];
        }
        __syncthreads();
        if (t < 32) {
          float acc = 0.f;
          for (int r = 0; r < 32; ++r) acc += red[r][t];
          po[c * 256 + 32 * (rr + 2 * k) + t] = acc;
        }
      }
  // bias gradient: the 32 rows' sums, in row order
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
    if (t < 32) red[t][0] = sb[c];
    __syncthreads();
    if (t == 0) {
      float acc = 0.f;
      for (int r = 0; r < 32; ++r) acc += red[r][0];
      po[768 + c] = acc;
    }
  }
  if (t == 0) po[771] = 0.f;
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

// grid size of the persistent kernels + the >64 KiB dynamic-LDS opt-in, per device and thread-safe (s2l_common.h)
static int persistent_grid(const void* kernel, int lds_bytes, LdsOptIn& flags, int n_tiles, int* grid) {
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  if ((rc = ensure_dynamic_lds(kernel, lds_bytes, flags, dev))) return rc;
  *grid = n_tiles < n_cu ? n_tiles : n_cu;
  return 0;
}

// 0 = the assembly forward kernel (64 rows per wave; default), 1 = the C++ kernel: same bits
static std::atomic<int> g_fwd_kernel_kind{0};
extern "C" int s2l_set_bf16_forward_kernel(int kind) {
  if (kind != 0 && kind != 1) return S2L_E_SIZE;
  g_fwd_kernel_kind.store(kind, std::memory_order_relaxed);
  return S2L_OK;
}

extern "C" int s2l_train_forward_bf16(const uint16_t* packed_bf16, const float* packed_f32, const uint16_t* xT, uint16_t* hT,
                                      uint64_t* masks, float* rgb, int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed_bf16 || !packed_f32 || !xT || !hT || !masks || !rgb) return S2L_E_NULL;
  if (misaligned16(packed_bf16) || misaligned16(xT) || misaligned16(hT) || misaligned16(masks)) return S2L_E_ALIGN;
  FwdArgs a;
  const int64_t np = s2l_bf16_rows_padded(n_rows);
  a.wb = packed_bf16, a.pf = packed_f32, a.xT = xT, a.hT = hT, a.masks = masks, a.rgb = rgb;
  a.n_rows = n_rows, a.layer_stride = np * 256, a.mask_layer_stride = np / 64 * 256;
  a.n_tiles = (int)(np / kWgRows);
  static LdsOptIn flags;
  int grid = 0;
  if (S2L_FWD_ASM && g_fwd_kernel_kind.load(std::memory_order_relaxed) == 0 && n_rows < 0x7fffffff) {
    static LdsOptIn flags_asm;
    const int rc = persistent_grid(reinterpret_cast<const void*>(fwd_asm_bf16_kernel), kLdsFwd, flags_asm, a.n_tiles, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(fwd_asm_bf16_kernel, dim3(grid), dim3(256), kLdsFwd, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
  }
  const int rc = persistent_grid(reinterpret_cast<const void*>(fwd_bf16_kernel), kLdsFwd, flags, a.n_tiles, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL(fwd_bf16_kernel, dim3(grid), dim3(512), kLdsFwd, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

extern "C" int s2l_train_backward_bf16(const uint16_t* packed_bf16, const float* drgb, const uint64_t* masks, uint16_t* dzT,
                                       float* dxa, int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed_bf16 || !drgb || !masks || !dzT || !dxa) return S2L_E_NULL;
  if (misaligned16(packed_bf16) || misaligned16(dzT) || misaligned16(dxa) || misaligned16(masks)) return S2L_E_ALIGN;
  BwdArgs a;
  const int64_t np = s2l_bf16_rows_padded(n_rows);
  a.wb = packed_bf16, a.drgb = drgb, a.masks = masks, a.dzT = dzT, a.dxa = dxa;
  a.n_rows = n_rows, a.layer_stride = np * 256, a.mask_layer_stride = np / 64 * 256;
  a.n_tiles = (int)(np / kWgRows);
  static LdsOptIn flags;
  int grid = 0;
  const int rc = persistent_grid(reinterpret_cast<const void*>(bwd_bf16_kernel), kLdsBwd, flags, a.n_tiles, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL(bwd_bf16_kernel, dim3(grid), dim3(512), kLdsBwd, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

// The assembly backward: dzT as s2l_train_backward_bf16 (bit-identical); the audio gradient as per-TILE column sums
// dxa_tiles [n_rows_padded / 256][64] (tile t = rows [256 t, 256 t + 256)) instead of per row.
extern "C" int s2l_train_backward_bf16_tiles(const uint16_t* packed_bf16, const float* drgb, const uint64_t* masks, uint16_t* dzT,
                                             float* dxa_tiles, int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0 || n_rows >= (int64_t(1) << 24) * 16) return S2L_E_SIZE;      // 12 * row must fit 32 bits
  if (n_rows == 0) return S2L_OK;
  if (!packed_bf16 || !drgb || !masks || !dzT || !dxa_tiles) return S2L_E_NULL;
  if (misaligned16(packed_bf16) || misaligned16(dzT) || misaligned16(dxa_tiles) || misaligned16(masks)) return S2L_E_ALIGN;
  BwdArgs a;
  const int64_t np = s2l_bf16_rows_padded(n_rows);
  a.wb = packed_bf16, a.drgb = drgb, a.masks = masks, a.dzT = dzT, a.dxa = dxa_tiles;
  a.n_rows = n_rows, a.layer_stride = np * 256, a.mask_layer_stride = np / 64 * 256;
  a.n_tiles = (int)(np / kWgRows);
  static LdsOptIn flags;
  int grid = 0;
  const int rc = persistent_grid(reinterpret_cast<const void*>(bwd_asm_bf16_kernel), kLdsBwdAsm, flags, a.n_tiles, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL(bwd_asm_bf16_kernel, dim3(grid), dim3(256), kLdsBwdAsm, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

extern "C" int64_t s2l_wgrad_bf16_work_floats(void) { return (int64_t)kWgParts * (256 * 256 + 256); }

extern "C" int s2l_wgrad_bf16(const uint16_t* dzT, const uint16_t* inT, int k_in, float* work, float* dw, float* db,
                              int64_t n_rows, s2l_stream_t stream) {
  if (n_rows <= 0 || (k_in != 256 && k_in != 128)) return S2L_E_SIZE;
  if (!dzT || !inT || !work || !dw) return S2L_E_NULL;
  if (misaligned16(dzT) || misaligned16(inT) || misaligned16(work) || misaligned16(dw)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n_tiles = (int)(s2l_bf16_rows_padded(n_rows) / 64);
  const int parts = n_tiles < kWgParts ? n_tiles : kWgParts;
  float* bpart = work + (int64_t)kWgParts * 256 * 256;
  static LdsOptIn flags256, flags128;
  int dev = 0, n_cu = 0;
  int rc = current_device_cus(&dev, &n_cu);
  if (rc) return rc;
  rc = k_in == 256 ? ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad_bf16_kernel<256>), WgCfg<256>::kLds, flags256, dev)
                   : ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad_bf16_kernel<128>), WgCfg<128>::kLds, flags128, dev);
  if (rc) return rc;
  if (k_in == 256)
    hipLaunchKernelGGL(wgrad_bf16_kernel<256>, dim3(parts), dim3(256), WgCfg<256>::kLds, st, dzT, inT, work, bpart, n_tiles);
  else
    hipLaunchKernelGGL(wgrad_bf16_kernel<128>, dim3(parts), dim3(256), WgCfg<128>::kLds, st, dzT, inT, work, bpart, n_tiles);
  const int n = 256 * k_in;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n / 4 + 15) / 16 + (db ? 4 : 0)), dim3(256), 0, st, work, dw, n, parts,
                     (const float*)bpart, db, 256);
  return (int)hipGetLastError();
}

extern "C" int s2l_rows_to_tiles_bf16(const float* x, int k, uint16_t* xT, int64_t n_rows, s2l_stream_t stream) {
  if (n_rows <= 0 || k < 32 || k > 256 || (k & 31)) return S2L_E_SIZE;
  if (!x || !xT) return S2L_E_NULL;
  if (misaligned16(x) || misaligned16(xT)) return S2L_E_ALIGN;
  const int64_t np = s2l_bf16_rows_padded(n_rows);
  const int64_t threads = np * (k / 4);
  hipLaunchKernelGGL(rows_to_image_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, xT,
                     k, n_rows, np);
  return (int)hipGetLastError();
}

extern "C" int s2l_out_grad_bf16(const float* drgb, const uint16_t* h7T, float* work, float* dwout, float* dbout, int64_t n_rows,
                                 s2l_stream_t stream) {
  if (n_rows <= 0) return S2L_E_SIZE;
  if (!drgb || !h7T || !work || !dwout || !dbout) return S2L_E_NULL;
  if (misaligned16(h7T) || misaligned16(work) || misaligned16(dwout)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n_groups = (int)(s2l_bf16_rows_padded(n_rows) / kGroupRows);
  constexpr int kOutParts = 1024;      // four workgroups per CU: a streaming kernel (1.2 GB of h7 for 64 frames)
  static_assert((int64_t)kOutParts * 772 + 772 <= (int64_t)kWgParts * (256 * 256 + 256), "partials fit the weight-gradient work buffer");
  const int parts = n_groups < kOutParts ? n_groups : kOutParts;
  hipLaunchKernelGGL(out_grad_kernel, dim3(parts), dim3(256), 0, st, drgb, h7T, work, n_groups, n_rows);
  // parts x [772] -> [768] + [4]: reduce into a scratch row behind the partials, then split
  float* sum = work + (int64_t)kOutParts * 772;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((772 / 4 + 15) / 16), dim3(256), 0, st, work, sum, 772, parts, (const float*)nullptr,
                     (float*)nullptr, 0);
  (void)hipMemcpyAsync(dwout, sum, 768 * sizeof(float), hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(dbout, sum + 768, 3 * sizeof(float), hipMemcpyDeviceToDevice, st);
  return (int)hipGetLastError();
}

extern "C" int s2l_ensemble_rows_bf16(const float* packed, const float* coords, const float* feat, const int64_t* time_index,
                                      const float* u01, int width, int height, uint16_t* xT, float* areas, int64_t n_pixels,
                                      int64_t n_frames, s2l_stream_t stream) {
  if (n_pixels <= 0 || n_frames <= 0 || width <= 0 || height <= 0) return S2L_E_SIZE;
  if (!packed || !coords || !feat || !time_index || !u01 || !xT || !areas) return S2L_E_NULL;
  if (misaligned16(xT)) return S2L_E_ALIGN;
  const double rx = 0.5 / width, ry = 0.5 / height;   // as s2l_ensemble_rows forms them (python doubles rounded to fp32)
  EnsBatch sh;
  sh.dx[0] = (float)(-rx), sh.dx[1] = (float)rx, sh.dy[0] = (float)(-ry), sh.dy[1] = (float)ry, sh.ry = (float)ry;
  const int64_t n_rows = 4 * n_pixels * n_frames, np = s2l_bf16_rows_padded(n_rows);
  hipLaunchKernelGGL(ensemble_rows_bf16_kernel, dim3((unsigned)((np * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     packed, coords, feat, time_index, u01, sh, xT, areas, n_pixels, n_rows, np);
  return (int)hipGetLastError();
}
