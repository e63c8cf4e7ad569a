// Load-time layout work: state-dict tensors -> packed blob (s2l_layout.h).
// No counterpart in the reference (it keeps torch nn.Parameters, tf_nerf.py:91-172).
#include <hip/hip_runtime.h>
#include "s2l_common.h"

namespace s2l {

struct TensorTable {
  const float* t[S2L_NUM_TENSORS];
};

// One thread per (layer, mb, j4, lane): writes the 4 consecutive k-steps a lane reads as one
// 16-byte A-operand quad.  layer 0..6 = pts_linears 1..7 (pts5 uses columns 256..511), 7 = output.
__global__ void pack_mlp_slabs(TensorTable tab, float* __restrict__ packed) {
  const int lane = threadIdx.x & 63;
  const int j4 = (threadIdx.x >> 6) + 4 * (blockIdx.x & 3);   // 256 threads = 4 j4 per block
  const int mb = (blockIdx.x >> 2) & 15;
  const int layer = blockIdx.x >> 6;
  const int row = mb * 16 + (lane & 15);
  const int q = lane >> 4;
  float v[4];
  if (layer < kHidden) {
    const int pts = layer + 1;
    const float* w = tab.t[S2L_T_PTS0_W + 2 * pts];
    const int ld = pts == 5 ? 512 : 256;
    const int c0 = pts == 5 ? 256 : 0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) v[jj] = w[(int64_t)row * ld + c0 + kfeat(j4 * 4 + jj, q)];
    float* dst = packed + OFF_WMLP + ((int64_t)(layer * 16 + mb) * 16 + j4) * 256 + lane * 4;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dst[jj] = v[jj];
  } else if (mb == 0) {
    const float* w = tab.t[S2L_T_OUT_W];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) v[jj] = row < 3 ? w[row * 256 + kfeat(j4 * 4 + jj, q)] : 0.f;
    float* dst = packed + OFF_WOUT + (int64_t)j4 * 256 + lane * 4;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dst[jj] = v[jj];
  }
}

// dst[c][r] = src[r][c]
__device__ inline void transpose_into(float* dst, const float* src, int rows, int cols, int tid, int nthreads) {
  for (int i = tid; i < rows * cols; i += nthreads) {
    const int c = i / rows, r = i - c * rows;
    dst[i] = src[(int64_t)r * cols + c];
  }
}

struct DivTerm {
  float v[10];
};

// Transposed copies for the per-pixel / per-frame products and the audio encoder.
__global__ void pack_small(TensorTable tab, DivTerm div, float* __restrict__ packed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt = gridDim.x * blockDim.x;
  // pts0^T and pts5[:, :256]^T : dst[k][n] = W[n][k]
  for (int i = tid; i < kW * kW; i += nt) {
    const int k = i >> 8, n = i & 255;
    packed[OFF_W0T + i] = tab.t[S2L_T_PTS0_W][n * 256 + k];
    packed[OFF_W5AT + i] = tab.t[S2L_T_PTS5_W][n * 512 + k];
  }
  transpose_into(packed + OFF_WUVT, tab.t[S2L_T_FC_UV_W], kW, kEmb, tid, nt);
  transpose_into(packed + OFF_WUVST, tab.t[S2L_T_FC_UV_SKIP_W], kW, kEmb, tid, nt);
  transpose_into(packed + OFF_WAT, tab.t[S2L_T_FC_AUDIO_W], kW, kAud, tid, nt);
  transpose_into(packed + OFF_WAST, tab.t[S2L_T_FC_AUDIO_SKIP_W], kW, kAud, tid, nt);
  transpose_into(packed + OFF_WTT, tab.t[S2L_T_FC_TIME_W], kW, kTime, tid, nt);
  transpose_into(packed + OFF_WTST, tab.t[S2L_T_FC_TIME_SKIP_W], kW, kTime, tid, nt);
  for (int n = tid; n < kW; n += nt) {
    packed[OFF_BSUM0 + n] = tab.t[S2L_T_FC_UV_B][n] + tab.t[S2L_T_FC_AUDIO_B][n] + tab.t[S2L_T_FC_TIME_B][n];
    packed[OFF_BSUM5 + n] =
        tab.t[S2L_T_FC_UV_SKIP_B][n] + tab.t[S2L_T_FC_AUDIO_SKIP_B][n] + tab.t[S2L_T_FC_TIME_SKIP_B][n];
    packed[OFF_B0 + n] = tab.t[S2L_T_PTS0_B][n];
    packed[OFF_B5 + n] = tab.t[S2L_T_PTS5_B][n];
    // row 4 (pts_linears[5]) is zero: its bias travels inside q5 (table mode) / BG5 (general mode)
    for (int l = 0; l < kHidden; ++l) packed[OFF_BIAS + l * kW + n] = l == 4 ? 0.f : tab.t[S2L_T_PTS0_B + 2 * (l + 1)][n];
  }
  for (int i = tid; i < 16; i += nt) packed[OFF_DIV + i] = i < 10 ? div.v[i] : 0.f;
  for (int i = tid; i < 4; i += nt) packed[OFF_BOUT + i] = i < 3 ? tab.t[S2L_T_OUT_B][i] : 0.f;
  // audio encoder: conv [cout][cin][3] -> [cin][3][cout]; fc [out][in] -> [in][out]
  auto conv_t = [&](int64_t off, const float* w, int cout, int cin) {
    for (int i = tid; i < cout * cin * 3; i += nt) {
      const int o = i % cout, ck = i / cout;
      packed[off + i] = w[(int64_t)o * cin * 3 + ck];
    }
  };
  conv_t(OFF_C0W, tab.t[S2L_T_CONV0_W], 32, 29);
  conv_t(OFF_C2W, tab.t[S2L_T_CONV2_W], 32, 32);
  conv_t(OFF_C4W, tab.t[S2L_T_CONV4_W], 64, 32);
  conv_t(OFF_C6W, tab.t[S2L_T_CONV6_W], 64, 64);
  transpose_into(packed + OFF_F0W, tab.t[S2L_T_FC1_0_W], 64, 64, tid, nt);
  transpose_into(packed + OFF_F2W, tab.t[S2L_T_FC1_2_W], 64, 64, tid, nt);
  for (int i = tid; i < 64; i += nt) {
    if (i < 32) {
      packed[OFF_C0B + i] = tab.t[S2L_T_CONV0_B][i];
      packed[OFF_C2B + i] = tab.t[S2L_T_CONV2_B][i];
    }
    packed[OFF_C4B + i] = tab.t[S2L_T_CONV4_B][i];
    packed[OFF_C6B + i] = tab.t[S2L_T_CONV6_B][i];
    packed[OFF_F0B + i] = tab.t[S2L_T_FC1_0_B][i];
    packed[OFF_F2B + i] = tab.t[S2L_T_FC1_2_B][i];
  }
}

// General-mode folds:  M0 = W0 [Wuv|Wa|Wt|0]  (256x128), c0 = W0 (buv+ba+bt) + b0, and the same
// with W5[:, :256] and the skip projections.  One block per output row n; thread = input column.
__global__ void pack_fold_general(TensorTable tab, float* __restrict__ packed) {
  const int n = blockIdx.x & 255;
  const bool skip = blockIdx.x >= 256;
  const int kk = threadIdx.x;  // 0..127 columns, 128 = bias column
  const float* wrow = skip ? tab.t[S2L_T_PTS5_W] + (int64_t)n * 512 : tab.t[S2L_T_PTS0_W] + (int64_t)n * 256;
  const float* wuv = tab.t[skip ? S2L_T_FC_UV_SKIP_W : S2L_T_FC_UV_W];
  const float* wa = tab.t[skip ? S2L_T_FC_AUDIO_SKIP_W : S2L_T_FC_AUDIO_W];
  const float* wt = tab.t[skip ? S2L_T_FC_TIME_SKIP_W : S2L_T_FC_TIME_W];
  const float* buv = tab.t[skip ? S2L_T_FC_UV_SKIP_B : S2L_T_FC_UV_B];
  const float* ba = tab.t[skip ? S2L_T_FC_AUDIO_SKIP_B : S2L_T_FC_AUDIO_B];
  const float* bt = tab.t[skip ? S2L_T_FC_TIME_SKIP_B : S2L_T_FC_TIME_B];
  float acc = 0.f;
  if (kk < 126) {
    for (int m = 0; m < 256; ++m) {
      float c;
      if (kk < kEmb) c = wuv[m * kEmb + kk];
      else if (kk < kEmb + kAud) c = wa[m * kAud + (kk - kEmb)];
      else c = wt[m * kTime + (kk - kEmb - kAud)];
      acc = fmaf(wrow[m], c, acc);
    }
  } else if (kk == 128) {
    for (int m = 0; m < 256; ++m) acc = fmaf(wrow[m], (buv[m] + ba[m]) + bt[m], acc);
    acc += tab.t[skip ? S2L_T_PTS5_B : S2L_T_PTS0_B][n];
    packed[(skip ? OFF_BG5 : OFF_BG0) + n] = acc;
    return;
  }
  if (kk >= 128) return;
  packed[(skip ? OFF_G5 : OFF_G0) + n * kGenK + kk] = acc;   // plain copy (transposed-packed below for the backward)
  // A layout, K=128: slab(mb)[(j4*64 + lane)*4 + jj], lane = q*16 + (n&15), kin(j, q) = 32*q + j
  const int j = kk & 31, q = kk >> 5;
  const int mb = n >> 4, lane = q * 16 + (n & 15);
  packed[(skip ? OFF_WG5 : OFF_WG0) + (int64_t)mb * (kSlab / 2) + ((j >> 2) * 64 + lane) * 4 + (j & 3)] = acc;
}

// Backward-pass operand copies: transposed hidden layers / output layer (dgrad), audio columns of
// the folded matrices (gradient of the per-frame audio feature).  Runs after pack_fold_general.
__global__ void pack_backward(TensorTable tab, float* __restrict__ packed) {
  const int lane = threadIdx.x & 63;
  const int j4 = (threadIdx.x >> 6) + 4 * (blockIdx.x & 3);
  const int mb = (blockIdx.x >> 2) & 15;
  const int layer = blockIdx.x >> 6;      // 0..6 hidden, 7: output + audio columns
  const int i = lane & 15, q = lane >> 4;
  if (layer < kHidden) {
    const int pts = layer + 1;
    const float* w = tab.t[S2L_T_PTS0_W + 2 * pts];
    const int ld = pts == 5 ? 512 : 256;
    const int c0 = pts == 5 ? 256 : 0;
    float* dst = packed + OFF_WMLPT + ((int64_t)(layer * 16 + mb) * 16 + j4) * 256 + lane * 4;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dst[jj] = w[(int64_t)kfeat(j4 * 4 + jj, q) * ld + c0 + mb * 16 + i];
  } else {
    if (j4 == 0) packed[OFF_WOUTT + mb * 64 + lane] = q < 3 ? tab.t[S2L_T_OUT_W][q * 256 + mb * 16 + i] : 0.f;
    if (mb < 4) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int krow = kfeat(j4 * 4 + jj, q);
        packed[OFF_G0AT + (int64_t)(mb * 16 + j4) * 256 + lane * 4 + jj] = packed[OFF_G0 + krow * kGenK + kEmb + mb * 16 + i];
        packed[OFF_G5AT + (int64_t)(mb * 16 + j4) * 256 + lane * 4 + jj] = packed[OFF_G5 + krow * kGenK + kEmb + mb * 16 + i];
      }
    }
  }
}

}  // namespace s2l

extern "C" int64_t s2l_packed_floats(void) { return s2l::PACKED_FLOATS; }

extern "C" const char* s2l_version(void) { return "s2l_hip 0.1 gfx950"; }

extern "C" int s2l_pack_weights(const float* const* tensors_host, const float* div_term_host, float* packed,
                                s2l_stream_t stream) {
  if (!tensors_host || !div_term_host || !packed) return S2L_E_NULL;
  if (reinterpret_cast<uintptr_t>(packed) & 15) return S2L_E_ALIGN;
  s2l::TensorTable tab;
  for (int i = 0; i < S2L_NUM_TENSORS; ++i) {
    if (!tensors_host[i]) return S2L_E_NULL;
    tab.t[i] = tensors_host[i];
  }
  s2l::DivTerm div;
  for (int i = 0; i < 10; ++i) div.v[i] = div_term_host[i];
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(s2l::pack_mlp_slabs, dim3(8 * 16 * 4), dim3(256), 0, st, tab, packed);
  hipLaunchKernelGGL(s2l::pack_small, dim3(64), dim3(256), 0, st, tab, div, packed);
  hipLaunchKernelGGL(s2l::pack_fold_general, dim3(512), dim3(192), 0, st, tab, packed);
  hipLaunchKernelGGL(s2l::pack_backward, dim3(8 * 16 * 4), dim3(256), 0, st, tab, packed);
  return (int)hipGetLastError();
}
