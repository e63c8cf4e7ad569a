// Crop + bilinear resize between the post-fusion U-Net and the lip-sync expert, and its adjoint
// (src/face_simple/training.py:541-544 of the reference):
//     rgb_merged = rgb_merged[:, y:y2, x:x2, :];  rgb_merged = transforms.Resize([96, 96])(rgb_merged.permute(0,3,1,2))
// torchvision 0.9.0 (the reference's pin) resizes tensors with torch.nn.functional.interpolate(mode='bilinear',
// align_corners=False), no antialiasing: per output pixel
//     src = max(scale * (dst + 0.5) - 0.5, 0),  scale = in / out (fp32),  i0 = floor(src), i1 = min(i0 + 1, in - 1),
//     l = src - i0,  out = (1-ly) * ((1-lx) * v00 + lx * v01) + ly * ((1-lx) * v10 + lx * v11).
// ATen's CPU kernel evaluates `scale * (dst + 0.5) - 0.5` with ONE rounding (the compiler contracts it into an fma;
// pinned numerically against F.interpolate in tools/make_goldens.py): fmaf below, not mul + sub.
// HBM-trivial (5 frames of 96x96 per sample); written for exactness and determinism, not speed.
#include "s2l_common.h"

namespace s2l {

struct ResizeArgs {
  int src_h, src_w;      // full frame
  int x, y, cw, ch;      // crop box origin and size
  int out_h, out_w;
  int T;                 // 0: dst [F,out_h,out_w,3] NHWC; > 0: dst [F/T,3,T,out_h,out_w] with frame f = s*T + t (the
                         // rgb_window layout of training.py:547-548)
  float sy, sx;          // in / out
};

__device__ __forceinline__ void resize_tap(float scale, int dst, int in, int* i0, int* i1, float* l) {
  const float src = fmaxf(fmaf(scale, (float)dst + 0.5f, -0.5f), 0.f);
  const int a = min((int)src, in - 1);
  *i0 = a;
  *i1 = min(a + 1, in - 1);
  *l = src - (float)a;
}

__device__ __forceinline__ int64_t window_index(const ResizeArgs& a, int64_t f, int c, int oy, int ox) {
  if (a.T == 0) return ((f * a.out_h + oy) * a.out_w + ox) * 3 + c;
  const int64_t s = f / a.T, t = f - s * a.T;
  return (((s * 3 + c) * a.T + t) * a.out_h + oy) * (int64_t)a.out_w + ox;
}

__global__ __launch_bounds__(256) void crop_resize_kernel(ResizeArgs a, const float* __restrict__ src, float* __restrict__ dst,
                                                         int64_t n_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const int ox = (int)(i % a.out_w);
  const int oy = (int)((i / a.out_w) % a.out_h);
  const int64_t f = i / ((int64_t)a.out_w * a.out_h);
  int y0, y1, x0, x1;
  float ly, lx;
  resize_tap(a.sy, oy, a.ch, &y0, &y1, &ly);
  resize_tap(a.sx, ox, a.cw, &x0, &x1, &lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* s = src + f * (int64_t)a.src_h * a.src_w * 3;
  auto at = [&](int yy, int xx, int c) { return s[((int64_t)(a.y + yy) * a.src_w + (a.x + xx)) * 3 + c]; };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float top = __fadd_rn(__fmul_rn(hx, at(y0, x0, c)), __fmul_rn(lx, at(y0, x1, c)));
    const float bot = __fadd_rn(__fmul_rn(hx, at(y1, x0, c)), __fmul_rn(lx, at(y1, x1, c)));
    dst[window_index(a, f, c, oy, ox)] = __fadd_rn(__fmul_rn(hy, top), __fmul_rn(ly, bot));
  }
}

// Adjoint as a GATHER (deterministic, no atomics): one thread per pixel of the full source frame.  A source pixel inside
// the crop box collects hy/ly * hx/lx * d_out from every output pixel whose two taps along each axis include it; the
// candidate outputs along an axis are a short contiguous range around (p + 0.5) / scale.  Pixels outside the box get 0
// (the crop's adjoint), so the result needs no separate clear.
__device__ __forceinline__ void adjoint_range(float scale, int p, int out, int* lo, int* hi) {
  // outputs whose source index lies in (p - 1, p + 1): dst in ((p - 0.5) / scale - 0.5, (p + 1.5) / scale - 0.5); one of slack
  const float inv = 1.f / scale;
  *lo = max((int)floorf(((float)p - 0.5f) * inv - 0.5f) - 1, 0);
  *hi = min((int)ceilf(((float)p + 1.5f) * inv - 0.5f) + 1, out - 1);
}

__global__ __launch_bounds__(256) void crop_resize_bwd_kernel(ResizeArgs a, const float* __restrict__ d_dst,
                                                             float* __restrict__ d_src, int64_t n_src) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_src) return;
  const int gx = (int)(i % a.src_w);
  const int gy = (int)((i / a.src_w) % a.src_h);
  const int64_t f = i / ((int64_t)a.src_w * a.src_h);
  float acc[3] = {0.f, 0.f, 0.f};
  const int px = gx - a.x, py = gy - a.y;
  if ((unsigned)px < (unsigned)a.cw && (unsigned)py < (unsigned)a.ch) {
    int oy_lo, oy_hi, ox_lo, ox_hi;
    adjoint_range(a.sy, py, a.out_h, &oy_lo, &oy_hi);
    adjoint_range(a.sx, px, a.out_w, &ox_lo, &ox_hi);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      resize_tap(a.sy, oy, a.ch, &y0, &y1, &ly);
      const float wy = (y0 == py ? 1.f - ly : 0.f) + (y1 == py ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        resize_tap(a.sx, ox, a.cw, &x0, &x1, &lx);
        const float wx = (x0 == px ? 1.f - lx : 0.f) + (x1 == px ? lx : 0.f);
        if (wx == 0.f) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fmaf(wy * wx, d_dst[window_index(a, f, c, oy, ox)], acc[c]);
      }
    }
  }
  float* o = d_src + i * 3;
  o[0] = acc[0];
  o[1] = acc[1];
  o[2] = acc[2];
}

static int resize_args(ResizeArgs& a, int src_h, int src_w, int x, int y, int x2, int y2, int out_h, int out_w, int t,
                       int64_t n_frames) {
  if (src_h <= 0 || src_w <= 0 || out_h <= 0 || out_w <= 0 || t < 0 || n_frames < 0) return S2L_E_SIZE;
  // rgb_merged[:, y:y2, x:x2, :] (training.py:541-543): python slicing silently clips an end beyond the frame (face-detector boxes
  // on face-cropped clips often do), and the resize scale follows the CLIPPED crop.  A negative start would wrap around in
  // python; that and an empty box are errors here.
  x2 = x2 < src_w ? x2 : src_w;
  y2 = y2 < src_h ? y2 : src_h;
  if (x < 0 || y < 0 || x2 <= x || y2 <= y) return S2L_E_GEOMETRY;
  if (t > 0 && n_frames % t != 0) return S2L_E_SIZE;
  a.src_h = src_h; a.src_w = src_w; a.x = x; a.y = y; a.cw = x2 - x; a.ch = y2 - y;
  a.out_h = out_h; a.out_w = out_w; a.T = t;
  a.sy = (float)a.ch / (float)out_h;
  a.sx = (float)a.cw / (float)out_w;
  return S2L_OK;
}

}  // namespace s2l

extern "C" int s2l_crop_resize(const float* src, int src_h, int src_w, int x, int y, int x2, int y2, float* dst, int out_h,
                               int out_w, int window_t, int64_t n_frames, s2l_stream_t stream) {
  s2l::ResizeArgs a;
  const int rc = s2l::resize_args(a, src_h, src_w, x, y, x2, y2, out_h, out_w, window_t, n_frames);
  if (rc || n_frames == 0) return rc;
  if (!src || !dst) return S2L_E_NULL;
  const int64_t n = n_frames * out_h * out_w;
  hipLaunchKernelGGL(s2l::crop_resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     src, dst, n);
  return (int)hipGetLastError();
}

extern "C" int s2l_crop_resize_backward(const float* d_dst, int src_h, int src_w, int x, int y, int x2, int y2, float* d_src,
                                        int out_h, int out_w, int window_t, int64_t n_frames, s2l_stream_t stream) {
  s2l::ResizeArgs a;
  const int rc = s2l::resize_args(a, src_h, src_w, x, y, x2, y2, out_h, out_w, window_t, n_frames);
  if (rc || n_frames == 0) return rc;
  if (!d_dst || !d_src) return S2L_E_NULL;
  const int64_t n = n_frames * src_h * src_w;
  hipLaunchKernelGGL(s2l::crop_resize_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, d_dst, d_src, n);
  return (int)hipGetLastError();
}
