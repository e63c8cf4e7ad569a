// Paste + head-pose-warp composite (HBM-bound).
//
// Replaces TalkingFace.post_fusion2_onlylip_light up to the U-Net (tf_nerf.py:320-386).  The
// reference materialises six full-frame temporaries (padded lip, merged canonical image, two
// masks, two grid_sample outputs); here one thread produces one output pixel and samples a
// VIRTUAL merged canonical image: for each bilinear tap it evaluates
//     merged_c(y,x) = mask*lip_pad + (1-mask)*face_canon                     (:352)
// on the fly (the lip is only read inside its box), and the expanded lip mask (:354-364) is an
// axis-aligned rectangle, i.e. two integer range tests instead of an image.
// Unique HBM traffic per frame: coord 8 B + rgb_gt 12 B + out 12 B per face pixel (+ the lip);
// the canonical face and mask are per-clip constants that live in L2 / Infinity Cache.
#include "s2l_common.h"

namespace s2l {

struct CompArgs {
  const float* lip;    // [F,h,w,3]
  const float* face;   // [FH,FW,3] (stride 0) or [F,FH,FW,3]
  const float* mask;   // same
  const float* gt;     // [F,FH,FW,3]
  const float* coord;  // [F,FH,FW,2]
  float* out_new;      // [F,FH,FW,3]
  float* out_can;      // [F,FH,FW,3] or null
  int64_t face_stride, mask_stride, total;
  int h, w, FH, FW;
  int ox, oy;          // paste origin of the lip in the face frame
  int ry0, ry1, rx0, rx1;  // expanded-mask rectangle [ry0,ry1) x [rx0,rx1); ry0 < 0 => use `mask`
};

struct Px {
  float c[3];
};

// merged canonical image at integer (yy, xx) of frame f.  Separate roundings on purpose: the
// reference evaluates mul, rsub, mul, add as four ATen ops (tf_nerf.py:352), so no FMA here.
__device__ inline Px merged_c(const CompArgs& a, const float* face, const float* mask, const float* lip, int yy, int xx,
                              Px* mask_out) {
  const int64_t o = ((int64_t)yy * a.FW + xx) * 3;
  const int ly = yy - a.oy, lx = xx - a.ox;
  const bool in_lip = (unsigned)ly < (unsigned)a.h && (unsigned)lx < (unsigned)a.w;
  const float* lp = lip + ((int64_t)ly * a.w + lx) * 3;
  Px r;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float m = mask[o + c];
    const float l = in_lip ? lp[c] : 0.f;
    r.c[c] = __fadd_rn(__fmul_rn(m, l), __fmul_rn(__fsub_rn(1.f, m), face[o + c]));
    if (mask_out) mask_out->c[c] = m;
  }
  return r;
}

__global__ __launch_bounds__(256) void composite_kernel(CompArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.total) return;
  const int64_t per = (int64_t)a.FH * a.FW;
  const int64_t f = idx / per;
  const float* face = a.face + f * a.face_stride;
  const float* mask = a.mask + f * a.mask_stride;
  const float* lip = a.lip + f * (int64_t)a.h * a.w * 3;
  const bool rect = a.ry0 >= 0;

  const float2 g = reinterpret_cast<const float2*>(a.coord)[idx];
  // grid_sample(align_corners=False): unnormalise as (x+1)*(size/2) - 0.5, bilinear weights from
  // the distances to the four neighbours, zero padding outside [0,size-1].
  const float ix = __fsub_rn(__fmul_rn(__fadd_rn(g.x, 1.f), 0.5f * (float)a.FW), 0.5f);
  const float iy = __fsub_rn(__fmul_rn(__fadd_rn(g.y, 1.f), 0.5f * (float)a.FH), 0.5f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float wx = ix - xw, ex = 1.f - wx, ny = iy - yn, sy = 1.f - ny;
  const float wgt[4] = {__fmul_rn(sy, ex), __fmul_rn(sy, wx), __fmul_rn(ny, ex), __fmul_rn(ny, wx)};

  float acc[3] = {0.f, 0.f, 0.f}, macc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float fx = xw + (float)(t & 1), fy = yn + (float)(t >> 1);
    const bool ok = fx >= 0.f && fx <= (float)(a.FW - 1) && fy >= 0.f && fy <= (float)(a.FH - 1);
    if (ok) {
      const int xx = (int)fx, yy = (int)fy;
      Px m;
      const Px v = merged_c(a, face, mask, lip, yy, xx, rect ? nullptr : &m);
      const float mr = (rect && yy >= a.ry0 && yy < a.ry1 && xx >= a.rx0 && xx < a.rx1) ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[c] = fmaf(v.c[c], wgt[t], acc[c]);
        macc[c] = fmaf(rect ? mr : m.c[c], wgt[t], macc[c]);
      }
    }
  }
  const float* gt = a.gt + idx * 3;
  float* o = a.out_new + idx * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = macc[c] != 0.f ? acc[c] : gt[c];

  if (a.out_can) {
    const int64_t r = idx - f * per;
    const int y = (int)(r / a.FW), x = (int)(r - (int64_t)y * a.FW);
    const Px v = merged_c(a, face, mask, lip, y, x, nullptr);
    float* oc = a.out_can + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) oc[c] = v.c[c];
  }
}

}  // namespace s2l

extern "C" int s2l_composite(const float* lip, const float* face_canon, int64_t face_stride, const float* mask,
                             int64_t mask_stride, const float* rgb_gt, const float* coord, float* out_new,
                             float* out_canonical, int lip_h, int lip_w, int face_h, int face_w, int x0, int y0,
                             int pad_mode, int expand_pad, int64_t n_frames, s2l_stream_t stream) {
  if (n_frames > 0 && (!lip || !face_canon || !mask || !rgb_gt || !coord || !out_new)) return S2L_E_NULL;
  if (lip_h <= 0 || lip_w <= 0 || face_h <= 0 || face_w <= 0 || n_frames < 0) return S2L_E_SIZE;
  const int64_t per = (int64_t)face_h * face_w;
  if ((face_stride != 0 && face_stride != per * 3) || (mask_stride != 0 && mask_stride != per * 3)) return S2L_E_SIZE;
  if (pad_mode != S2L_PAD_MAY && pad_mode != S2L_PAD_DEFAULT) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (reinterpret_cast<uintptr_t>(coord) & 7) return S2L_E_ALIGN;
  s2l::CompArgs a;
  a.lip = lip; a.face = face_canon; a.mask = mask; a.gt = rgb_gt; a.coord = coord;
  a.out_new = out_new; a.out_can = out_canonical;
  a.face_stride = face_stride; a.mask_stride = mask_stride; a.total = per * n_frames;
  a.h = lip_h; a.w = lip_w; a.FH = face_h; a.FW = face_w;
  a.ox = pad_mode == S2L_PAD_MAY ? x0 : x0 - 1;
  a.oy = pad_mode == S2L_PAD_MAY ? y0 : y0 - 1;
  // F.pad with a negative amount would crop: the reference assumes the lip box lies inside the face frame
  if (a.ox < 0 || a.oy < 0 || a.ox + lip_w > face_w || a.oy + lip_h > face_h) return S2L_E_GEOMETRY;
  if (expand_pad >= 0) {
    a.ry0 = y0 - expand_pad; a.rx0 = x0 - expand_pad;
    if (a.ry0 < 0 || a.rx0 < 0) return S2L_E_GEOMETRY;  // python slicing would wrap around
    a.ry1 = y0 + lip_h + 2 * expand_pad; a.rx1 = x0 + lip_w + expand_pad;
    if (a.ry1 > face_h) a.ry1 = face_h;
    if (a.rx1 > face_w) a.rx1 = face_w;
  } else {
    a.ry0 = a.ry1 = a.rx0 = a.rx1 = -1;
  }
  const int64_t blocks = (a.total + 255) / 256;
  if (blocks > 0x7fffffff) return S2L_E_SIZE;
  hipLaunchKernelGGL(s2l::composite_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
