// General-row implicit-function MLP: forward (s2l_rgb_forward, the exact drop-in for
// TalkingFace.rgb_forward, tf_nerf.py:225-285, on arbitrary [N,66] rows) and the matching
// backward chain for the training step (BASELINE config 5).  The clip renderer
// (s2l_render_lip, the benchmarked kernel) lives in render.hip and shares the MFMA scheme:
//
//   * one wave owns G groups of 16 rows and ALL 256 features of them;
//   * every 256x256 layer is D[feature][row] = W[feature][k] * H[k][row] on
//     v_mfma_f32_16x16x4_f32 (exact fp32): A = weights, B = activations;
//   * weights are packed (pack.hip) in the k-order in which the PREVIOUS layer's accumulator
//     registers hold the features, so a layer's D registers are the next layer's B operands;
//   * rows arrive as x = [E(uv) (42) | audio (64) | PE(t) (20) | 0 0] (frontend.hip) and the first
//     layer / skip half of pts_linears[5] are K=128 products with the folded matrices
//     G0 = W0 [Wuv|Wa|Wt], G5 = W5a [Wuv'|Wa'|Wt'] built at pack time;
//   * the backward runs the same chain with the transposed slabs: dh_{k-1} = W^T dz_k, masked by
//     the saved activations (ReLU'), and emits every dz_k for the weight-gradient GEMMs.
// Weights are read straight from L2 here (compiler-scheduled loads); the LDS-DMA ring of
// render.hip is the next step for this file.
#include "s2l_common.h"

namespace s2l {

struct MlpArgs {
  const float* packed;
  const float* x;     // [N,128]
  float* out;         // [N,3]
  float* hsave;       // optional [8][N][256]: post-ReLU h0..h7 (training), or null
  int64_t total;      // rows
};

struct BwdArgs {
  const float* packed;
  const float* drgb;    // [N,3]
  const float* hsave;   // [8][N][256]
  float* dzsave;        // [8][N][256]: gradient w.r.t. the pre-activation of h_k
  float* dxa;           // [N,64]: gradient w.r.t. the audio columns of x
  int64_t total;
};

__device__ inline f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[g][mb] += W[mb-block] . in[g]  for NMB M-blocks, K = NJ4*16
template <int G, int NJ4, int NMB, int NACC>
__device__ __forceinline__ void gemm_layer(const f4* __restrict__ wl, int lane, f4 (&acc)[G][NACC],
                                           const float (&in)[G][NJ4 * 4]) {
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) {
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

template <int G>
__device__ __forceinline__ void load_x(const float* x, const int64_t (&row)[G], int q, float (&xin)[G][32]) {
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const f4* xr = reinterpret_cast<const f4*>(x + row[g] * kGenK + 32 * q);   // kin(j, q) = 32*q + j
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f4 v = xr[i];
#pragma unroll
      for (int r = 0; r < 4; ++r) xin[g][i * 4 + r] = v[r];
    }
  }
}

template <int G, int NW>
__global__ __launch_bounds__(NW * 64) void mlp_fwd_kernel(MlpArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = lane >> 4, px = lane & 15;
  const int64_t nbase = ((int64_t)blockIdx.x * NW + wave) * (G * 16);
  if (nbase >= a.total) return;
  const float* __restrict__ packed = a.packed;

  float in[G][64];
  f4 acc[G][16];
  int64_t row[G];
  bool live[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int64_t n = nbase + g * 16 + px;
    live[g] = n < a.total;
    row[g] = live[g] ? n : a.total - 1;
  }
  auto relu_to_in = [&](int k) {   // in = relu(acc) = h_k; optionally saved for the backward
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
  auto init_bias = [&](const float* b) {
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const f4 v = *reinterpret_cast<const f4*>(b + mb * 16 + 4 * q);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g][mb] = v;
    }
  };

  // h0 = relu(G0 x + c0)
  init_bias(packed + OFF_BG0);
  {
    float xin[G][32];
    load_x<G>(a.x, row, q, xin);
    gemm_layer<G, 8, 16, 16>(reinterpret_cast<const f4*>(packed + OFF_WG0), lane, acc, xin);
  }
  relu_to_in(0);

  for (int layer = 0; layer < kHidden; ++layer) {
    if (layer == 4) {
      // pts_linears[5] on cat([skip, h4]): the skip half G5 x + c5 initialises the accumulator
      init_bias(packed + OFF_BG5);
      float xin[G][32];
      load_x<G>(a.x, row, q, xin);
      gemm_layer<G, 8, 16, 16>(reinterpret_cast<const f4*>(packed + OFF_WG5), lane, acc, xin);
    } else {
      init_bias(packed + OFF_BIAS + layer * kW);
    }
    gemm_layer<G, 16, 16, 16>(reinterpret_cast<const f4*>(packed + OFF_WMLP) + (int64_t)layer * 16 * 16 * 64, lane, acc, in);
    relu_to_in(layer + 1);
  }

  // output_linear (3 rows, zero-padded to one 16-row M-block); no activation (tf_nerf.py:283)
  f4 rgb[G][1];
  {
    const f4 b = *reinterpret_cast<const f4*>(packed + OFF_BOUT);
#pragma unroll
    for (int g = 0; g < G; ++g) rgb[g][0] = b;
    gemm_layer<G, 16, 1, 1>(reinterpret_cast<const f4*>(packed + OFF_WOUT), lane, rgb, in);
  }
  if (q == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (live[g]) {
        float* o = a.out + row[g] * 3;
        o[0] = rgb[g][0][0];
        o[1] = rgb[g][0][1];
        o[2] = rgb[g][0][2];
      }
  }
}

// Backward chain: drgb -> dz7 -> ... -> dz0 (all stored), plus the audio columns of dx.
template <int G, int NW>
__global__ __launch_bounds__(NW * 64) void mlp_bwd_kernel(BwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = lane >> 4, px = lane & 15;
  const int64_t nbase = ((int64_t)blockIdx.x * NW + wave) * (G * 16);
  if (nbase >= a.total) return;
  const float* __restrict__ packed = a.packed;

  float in[G][64];   // dz_k in accumulator layout = B operand of the next product
  f4 acc[G][16];
  f4 dxa[G][4];
  int64_t row[G];
  bool live[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int64_t n = nbase + g * 16 + px;
    live[g] = n < a.total;
    row[g] = live[g] ? n : a.total - 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) dxa[g][m] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  // dz_k = dh_k (in acc) masked by h_k > 0; stored, and moved to `in`
  auto mask_store = [&](int k) {
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
  auto zero_acc = [&]() {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) acc[g][mb] = (f4){0.f, 0.f, 0.f, 0.f};
  };

  // dh7 = Wout^T drgb: K = 4 (3 used), one MFMA per M-block
  {
    float b[G];
#pragma unroll
    for (int g = 0; g < G; ++g) b[g] = q < 3 ? a.drgb[row[g] * 3 + q] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const float w = packed[OFF_WOUTT + mb * 64 + lane];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g][mb] = mfma16(w, b[g], (f4){0.f, 0.f, 0.f, 0.f});
    }
  }
  mask_store(7);
  for (int k = 7; k >= 1; --k) {
    if (k == 5)   // pts_linears[5] also feeds the skip projection of x: d x_audio += G5[:, audio]^T dz5
      gemm_layer<G, 16, 4, 4>(reinterpret_cast<const f4*>(packed + OFF_G5AT), lane, dxa, in);
    zero_acc();
    gemm_layer<G, 16, 16, 16>(reinterpret_cast<const f4*>(packed + OFF_WMLPT) + (int64_t)(k - 1) * 16 * 16 * 64, lane, acc, in);
    mask_store(k - 1);
  }
  gemm_layer<G, 16, 4, 4>(reinterpret_cast<const f4*>(packed + OFF_G0AT), lane, dxa, in);
#pragma unroll
  for (int g = 0; g < G; ++g)
    if (live[g])
#pragma unroll
      for (int m = 0; m < 4; ++m) *reinterpret_cast<f4*>(a.dxa + row[g] * 64 + m * 16 + 4 * q) = dxa[g][m];
}

int launch_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                      hipStream_t st);

int launch_general_mlp(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st) {
  constexpr int G = 2, NW = 4;
  MlpArgs a{packed, x, out, hsave, n_rows};
  const int64_t blocks = (n_rows + G * 16 * NW - 1) / (G * 16 * NW);
  if (blocks > 0x7fffffff) return S2L_E_SIZE;
  hipLaunchKernelGGL((mlp_fwd_kernel<G, NW>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, a);
  return (int)hipGetLastError();
}

int launch_general_mlp_bwd(const float* packed, const float* drgb, const float* hsave, float* dzsave, float* dxa,
                           int64_t n_rows, hipStream_t st) {
  constexpr int G = 2, NW = 4;
  BwdArgs a{packed, drgb, hsave, dzsave, dxa, n_rows};
  const int64_t blocks = (n_rows + G * 16 * NW - 1) / (G * 16 * NW);
  if (blocks > 0x7fffffff) return S2L_E_SIZE;
  hipLaunchKernelGGL((mlp_bwd_kernel<G, NW>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, a);
  return (int)hipGetLastError();
}

}  // namespace s2l

extern "C" int s2l_rgb_forward(const float* packed, const float* uv_audio, int64_t time_index, float* xbuf, float* out,
                               int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed || !uv_audio || !xbuf || !out) return S2L_E_NULL;
  if (s2l::misaligned16(packed) || s2l::misaligned16(xbuf)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = s2l::launch_embed_rows(packed, uv_audio, time_index, xbuf, n_rows, st);
  if (rc) return rc;
  return s2l::launch_general_mlp(packed, xbuf, out, nullptr, n_rows, st);
}
