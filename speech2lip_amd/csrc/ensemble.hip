// 4-tap local-ensemble forward of training: s2l_predict_lip_image.
// Replaces Trainer.predict_lip_image (src/face_simple/training.py:158-251) for one frame:
//   four evaluations of the MLP at clamp(coords + (vx*0.5/W + eps, vy*0.5/H + eps), 0, 1),
//   vx, vy in {-1, 1} (vx outer), eps = (0.5/H) * U / 2 (:198-200); areas |du*dv| + 1e-9 with the
//   diagonal swap 0<->3, 1<->2 (:240-245); area-weighted sum (:246-249).
// Three launches: build the 4N embedded rows (+ areas), the general-row MLP (mlp.hip), reduce.
#include "s2l_common.h"

namespace s2l {

__device__ float embed_feature_f(float u, float v, int i) {
  if (i < 2) return i == 0 ? u : v;
  const int blk = (i - 2) >> 1;
  const float x = ((i & 1) ? v : u) * (float)(1 << (blk >> 1));
  return (blk & 1) ? cosf(x) : sinf(x);
}

struct TapShifts {
  float dx[4], dy[4];   // tap t = 2*ix + iy: (vx, vy) = ((-1,-1), (-1,1), (1,-1), (1,1))
};

// 32 threads per (tap, pixel) row, one 16-byte store each: x[t*N + n] = [E(c) | feat | PE(time) | 0 0]; the thread of
// columns 0..3 also writes the area.
__global__ __launch_bounds__(256) void ensemble_rows_kernel(const float* __restrict__ packed,
                                                           const float* __restrict__ coords,
                                                           const float* __restrict__ feat, float time_pos, TapShifts sh,
                                                           float* __restrict__ x, float* __restrict__ areas, int64_t n) {
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int k0 = (threadIdx.x & 31) * 4;
  if (row >= 4 * n) return;
  const int t = (int)(row / n);
  const int64_t p = row - (int64_t)t * n;
  const float u0 = coords[2 * p], v0 = coords[2 * p + 1];
  // coord_ = clamp(coords + shift, 0, 1): one fp32 add, then clamp (training.py:207-210)
  const float cu = fminf(fmaxf(u0 + sh.dx[t], 0.f), 1.f);
  const float cv = fminf(fmaxf(v0 + sh.dy[t], 0.f), 1.f);
  f4 val;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    float v;
    if (k < kEmb) v = embed_feature_f(cu, cv, k);
    else if (k < kEmb + kAud) v = feat[k - kEmb];
    else if (k < kEmb + kAud + kTime) {
      const int i = k - kEmb - kAud;
      const float arg = time_pos * packed[OFF_DIV + (i >> 1)];
      v = (i & 1) ? cosf(arg) : sinf(arg);
    } else v = 0.f;
    val[j] = v;
  }
  *reinterpret_cast<f4*>(x + row * kGenK + k0) = val;
  if (k0 == 0) areas[row] = fabsf((cu - u0) * (cv - v0)) + 1e-9f;   // training.py:240-241
}

// The same rows for a whole batch of frames in one launch (blockIdx.y = frame): frame b owns rows [4 b n, 4 (b+1) n) of x /
// areas; its audio feature, frame index and U(0,1) draw come from device arrays (no host round trip per frame).
__global__ __launch_bounds__(256) void ensemble_rows_batch_kernel(const float* __restrict__ packed, const float* __restrict__ coords,
                                                                 const float* __restrict__ feat, const int64_t* __restrict__ time_index,
                                                                 const float* __restrict__ u01, float rx, float ry,
                                                                 float* __restrict__ x, float* __restrict__ areas, int64_t n) {
  const int64_t b = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int k0 = (threadIdx.x & 31) * 4;
  if (row >= 4 * n) return;
  const int t = (int)(row / n);
  const int64_t p = row - (int64_t)t * n;
  const float eps = ry * u01[b] / 2.0f;                       // (0.5/H) * U / 2 in fp32, as tap_shifts()
  const float dx = ((t >> 1) ? rx : -rx) + eps, dy = ((t & 1) ? ry : -ry) + eps;
  const float time_pos = (float)time_index[b];
  const float u0 = coords[2 * p], v0 = coords[2 * p + 1];
  const float cu = fminf(fmaxf(u0 + dx, 0.f), 1.f);
  const float cv = fminf(fmaxf(v0 + dy, 0.f), 1.f);
  f4 val;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    float v;
    if (k < kEmb) v = embed_feature_f(cu, cv, k);
    else if (k < kEmb + kAud) v = feat[b * kAud + k - kEmb];
    else if (k < kEmb + kAud + kTime) {
      const int i = k - kEmb - kAud;
      const float arg = time_pos * packed[OFF_DIV + (i >> 1)];
      v = (i & 1) ? cosf(arg) : sinf(arg);
    } else v = 0.f;
    val[j] = v;
  }
  const int64_t grow = b * 4 * n + row;
  *reinterpret_cast<f4*>(x + grow * kGenK + k0) = val;
  if (k0 == 0) areas[grow] = fabsf((cu - u0) * (cv - v0)) + 1e-9f;
}

__global__ __launch_bounds__(256) void ensemble_reduce_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ areas, float* __restrict__ out,
                                                             int64_t n) {
  // blockIdx.y = frame: frame f owns rows [4 f n, 4 (f+1) n) of pred / areas and [f n, (f+1) n) of out
  pred += (int64_t)blockIdx.y * n * 12;
  areas += (int64_t)blockIdx.y * n * 4;
  out += (int64_t)blockIdx.y * n * 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 3) return;
  const int64_t p = i / 3;
  const float a0 = areas[p], a1 = areas[n + p], a2 = areas[2 * n + p], a3 = areas[3 * n + p];
  const float tot = ((a0 + a1) + a2) + a3;                       // torch.stack(areas).sum(0)
  // preds paired with swapped areas (0<->3, 1<->2); ret = 0 + p0*w0 + p1*w1 + p2*w2 + p3*w3
  float r = 0.f;
  r = r + pred[i] * (a3 / tot);
  r = r + pred[n * 3 + i] * (a2 / tot);
  r = r + pred[2 * n * 3 + i] * (a1 / tot);
  r = r + pred[3 * n * 3 + i] * (a0 / tot);
  out[i] = r;
}

int launch_rows_fwd(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st);

}  // namespace s2l

// The two halves of s2l_predict_lip_image on their own (the training step runs the MLP in between
// over many frames at once and keeps x / areas for the backward).
static s2l::TapShifts tap_shifts(int width, int height, float u01) {
  // shifts exactly as the reference forms them: python doubles rounded to fp32, plus the fp32 eps
  const double rx = 0.5 / width, ry = 0.5 / height;
  const float eps = (float)ry * u01 / 2.0f;
  s2l::TapShifts sh;
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy) {
      sh.dx[2 * ix + iy] = (float)((ix ? 1 : -1) * rx) + eps;
      sh.dy[2 * ix + iy] = (float)((iy ? 1 : -1) * ry) + eps;
    }
  return sh;
}

extern "C" int s2l_ensemble_rows(const float* packed, const float* coords, const float* feat, int64_t time_index, int width,
                                 int height, float u01, float* x, float* areas, int64_t n_pixels, s2l_stream_t stream) {
  if (n_pixels < 0 || width <= 0 || height <= 0) return S2L_E_SIZE;
  if (n_pixels == 0) return S2L_OK;
  if (!packed || !coords || !feat || !x || !areas) return S2L_E_NULL;
  hipLaunchKernelGGL(s2l::ensemble_rows_kernel, dim3((unsigned)((4 * n_pixels + 7) / 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), packed, coords, feat, (float)time_index,
                     tap_shifts(width, height, u01), x, areas, n_pixels);
  return (int)hipGetLastError();
}

extern "C" int s2l_ensemble_rows_batch(const float* packed, const float* coords, const float* feat, const int64_t* time_index,
                                       const float* u01, int width, int height, float* x, float* areas, int64_t n_pixels,
                                       int64_t n_frames, s2l_stream_t stream) {
  if (n_pixels < 0 || n_frames < 0 || n_frames > 65535 || width <= 0 || height <= 0) return S2L_E_SIZE;
  if (n_pixels == 0 || n_frames == 0) return S2L_OK;
  if (!packed || !coords || !feat || !time_index || !u01 || !x || !areas) return S2L_E_NULL;
  if (s2l::misaligned16(x)) return S2L_E_ALIGN;
  const double rx = 0.5 / width, ry = 0.5 / height;   // python doubles rounded to fp32, as tap_shifts()
  hipLaunchKernelGGL(s2l::ensemble_rows_batch_kernel, dim3((unsigned)((4 * n_pixels + 7) / 8), (unsigned)n_frames), dim3(256), 0,
                     static_cast<hipStream_t>(stream), packed, coords, feat, time_index, u01, (float)rx, (float)ry, x, areas,
                     n_pixels);
  return (int)hipGetLastError();
}

extern "C" int s2l_ensemble_reduce(const float* pred, const float* areas, float* out, int64_t n_pixels, s2l_stream_t stream) {
  if (n_pixels < 0) return S2L_E_SIZE;
  if (n_pixels == 0) return S2L_OK;
  if (!pred || !areas || !out) return S2L_E_NULL;
  hipLaunchKernelGGL(s2l::ensemble_reduce_kernel, dim3((unsigned)((n_pixels * 3 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pred, areas, out, n_pixels);
  return (int)hipGetLastError();
}

extern "C" int s2l_ensemble_reduce_batch(const float* pred, const float* areas, float* out, int64_t n_pixels, int64_t n_frames,
                                        s2l_stream_t stream) {
  if (n_pixels < 0 || n_frames < 0 || n_frames > 65535) return S2L_E_SIZE;
  if (n_pixels == 0 || n_frames == 0) return S2L_OK;
  if (!pred || !areas || !out) return S2L_E_NULL;
  hipLaunchKernelGGL(s2l::ensemble_reduce_kernel, dim3((unsigned)((n_pixels * 3 + 255) / 256), (unsigned)n_frames), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pred, areas, out, n_pixels);
  return (int)hipGetLastError();
}

extern "C" int64_t s2l_predict_lip_image_work_floats(int64_t n_pixels) {
  return n_pixels < 0 ? 0 : n_pixels * (4 * s2l::kGenK + 4 * 3 + 4);
}

extern "C" int s2l_predict_lip_image(const float* packed, const float* coords, const float* feat, int64_t time_index,
                                     int width, int height, float u01, float* work, float* out, int64_t n_pixels,
                                     s2l_stream_t stream) {
  using namespace s2l;
  if (n_pixels < 0 || width <= 0 || height <= 0) return S2L_E_SIZE;
  if (n_pixels == 0) return S2L_OK;
  if (!packed || !coords || !feat || !work || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(work)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const TapShifts sh = tap_shifts(width, height, u01);
  float* x = work;
  float* pred = x + 4 * n_pixels * kGenK;
  float* areas = pred + 4 * n_pixels * 3;
  hipLaunchKernelGGL(ensemble_rows_kernel, dim3((unsigned)((4 * n_pixels + 7) / 8)), dim3(256), 0, st, packed, coords, feat,
                     (float)time_index, sh, x, areas, n_pixels);
  int rc = launch_rows_fwd(packed, x, pred, nullptr, 4 * n_pixels, st);
  if (rc) return rc;
  hipLaunchKernelGGL(ensemble_reduce_kernel, dim3((unsigned)((n_pixels * 3 + 255) / 256)), dim3(256), 0, st, pred, areas,
                     out, n_pixels);
  return (int)hipGetLastError();
}
