// General-row implicit-function MLP: s2l_rgb_forward, the exact drop-in for
// TalkingFace.rgb_forward (tf_nerf.py:225-285) on arbitrary [N,66] rows.  The clip renderer
// (s2l_render_lip, the benchmarked kernel) lives in render.hip and shares the MFMA scheme:
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

int launch_general_mlp(const float* packed, const float* x, float* out, int64_t n_rows, hipStream_t st) {
  MlpArgs a{packed, x, nullptr, nullptr, nullptr, out, n_rows, 1};
  return launch_mlp<2, 4, true>(a, st);
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
  return s2l::launch_general_mlp(packed, xbuf, out, n_rows, st);
}
