// Pose -> warp grid on the device: the 2 MB/frame `coords/%05d.npy` grid the composite
// consumes is a pure function of a depth map and a 4x4 relative pose, so it is regenerated in HBM from
// 24 B/frame of pose instead of being read from disk and pushed over PCIe.
//
//   s2l_rel_pose     euler/trans -> T [B,4,4]   (utils.py:8-77, face_tracker.py:583-584, training.py:263-275)
//   s2l_warp_grid    depth, T -> grid [B,H,W,2] (+ z)   (BackprojectDepth + Project3D, utils.py:115-169)
//   s2l_grid_sample  bilinear NHWC gather with zero / border padding (F.grid_sample as training.py:312 calls it)
//
// All three are HBM-bound streaming kernels: the grid kernel writes 8 B and reads 4 B (0 when the depth map is
// a per-clip constant and stays in L2) per pixel, one pixel per lane so that a wave's stores are one contiguous
// 512-byte run.
#include "s2l_common.h"

namespace s2l {

// ---- rel pose: B threads, fp64 inside (the composition cancels two ~10-unit translations) ------
struct M34 {
  double r[3][3];
  double t[3];
};

__device__ M34 transform_of(const float* e, const float* tr) {
  // prepare_transform_matrix (utils.py:36-52): euler (e0,-e1,-e2), trans (t0,-t1,-t2); euler2rot = Rx Ry Rz with
  // Rx = [1 0 0; 0 c -s; 0 s c], Ry = [c 0 s; 0 1 0; -s 0 c], Rz = [c s 0; -s c 0; 0 0 1]   (utils.py:19-33)
  const double th = (double)e[0], ph = -(double)e[1], ps = -(double)e[2];
  const double ct = cos(th), st = sin(th), cp = cos(ph), sp = sin(ph), cs = cos(ps), ss = sin(ps);
  const double ry_rz[3][3] = {{cp * cs, cp * ss, sp}, {-ss, cs, 0.0}, {-sp * cs, -sp * ss, cp}};
  M34 m;
  for (int j = 0; j < 3; ++j) {
    m.r[0][j] = ry_rz[0][j];
    m.r[1][j] = ct * ry_rz[1][j] - st * ry_rz[2][j];
    m.r[2][j] = st * ry_rz[1][j] + ct * ry_rz[2][j];
  }
  m.t[0] = (double)tr[0];
  m.t[1] = -(double)tr[1];
  m.t[2] = -(double)tr[2];
  return m;
}

__device__ M34 rigid_inverse(const M34& a) {
  M34 o;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o.r[i][j] = a.r[j][i];
    o.t[i] = -(a.r[0][i] * a.t[0] + a.r[1][i] * a.t[1] + a.r[2][i] * a.t[2]);
  }
  return o;
}

__device__ M34 compose(const M34& a, const M34& b) {  // a . b
  M34 o;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o.r[i][j] = a.r[i][0] * b.r[0][j] + a.r[i][1] * b.r[1][j] + a.r[i][2] * b.r[2][j];
    o.t[i] = a.r[i][0] * b.t[0] + a.r[i][1] * b.t[1] + a.r[i][2] * b.t[2] + a.t[i];
  }
  return o;
}

__global__ void rel_pose_kernel(const float* __restrict__ euler, const float* __restrict__ trans,
                                const float* __restrict__ canon_euler, const float* __restrict__ canon_trans, int mode,
                                float* __restrict__ T, int64_t n) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const M34 tc = transform_of(canon_euler, canon_trans);
  const M34 tf = transform_of(euler + 3 * f, trans + 3 * f);
  // obs->can: Tc . inv(T); can->obs: T . inv(Tc); its inverse is Tc . inv(T) again (training.py:270-275)
  const M34 o = mode == S2L_POSE_CAN2OBS ? compose(tf, rigid_inverse(tc)) : compose(tc, rigid_inverse(tf));
  float* out = T + 16 * f;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out[4 * i + j] = (float)o.r[i][j];
    out[4 * i + 3] = (float)o.t[i];
  }
  out[12] = 0.f, out[13] = 0.f, out[14] = 0.f, out[15] = 1.f;
}

// ---- warp grid --------------------------------------------------------------------------------------------------
constexpr int kGridThreads = 256;
constexpr int kGridPix = 4;  // pixels per lane, strided by the block so every store instruction stays contiguous

__global__ __launch_bounds__(kGridThreads) void warp_grid_kernel(const float* __restrict__ depth, int64_t depth_stride,
                                                                 const float* __restrict__ T, float focal, float cx, float cy,
                                                                 int clamp, float eps, float* __restrict__ grid,
                                                                 float* __restrict__ zout, int H, int W) {
  const int f = blockIdx.y;
  const int hw = H * W;
  const float* t = T + 16 * (int64_t)f;  // wave-uniform: scalar loads
  // P = (K T)[:3]  (utils.py:157): row0 = f*T0 + cx*T2, row1 = f*T1 + cy*T2, row2 = T2
  float p0[4], p1[4], p2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p2[j] = t[8 + j];
    p0[j] = fmaf(focal, t[j], cx * p2[j]);
    p1[j] = fmaf(focal, t[4 + j], cy * p2[j]);
  }
  // pinv(K)[:3,:3] = [1/f 0 -cx/f; 0 1/f -cy/f; 0 0 1]   (utils.py:136)
  const float inv_f = 1.f / focal, ncx = -cx / focal, ncy = -cy / focal;
  const float isx = 1.f / (float)(W - 1), isy = 1.f / (float)(H - 1), inv_w = 1.f / (float)W;
  const float* dsrc = depth + depth_stride * f;
  float2* gdst = reinterpret_cast<float2*>(grid) + (int64_t)f * hw;
  const int base = blockIdx.x * (kGridThreads * kGridPix) + threadIdx.x;
  float d[kGridPix];
#pragma unroll
  for (int k = 0; k < kGridPix; ++k) {
    const int i = base + k * kGridThreads;
    d[k] = i < hw ? dsrc[i] : 1.f;
  }
#pragma unroll
  for (int k = 0; k < kGridPix; ++k) {
    const int i = base + k * kGridThreads;
    if (i >= hw) break;
    // (y, x) = divmod(i, W) by a float reciprocal and one fix-up step (hw <= 2^24, so (float)i is exact)
    int y = (int)(((float)i + 0.5f) * inv_w);
    int x = i - y * W;
    if (x < 0) x += W, --y;
    if (x >= W) x -= W, ++y;
    const float X = d[k] * fmaf((float)x, inv_f, ncx), Y = d[k] * fmaf((float)y, inv_f, ncy), Z = d[k];
    const float px = fmaf(p0[0], X, fmaf(p0[1], Y, fmaf(p0[2], Z, p0[3])));
    const float py = fmaf(p1[0], X, fmaf(p1[1], Y, fmaf(p1[2], Z, p1[3])));
    const float pz = fmaf(p2[0], X, fmaf(p2[1], Y, fmaf(p2[2], Z, p2[3])));
    // utils.py:158-163: pix = p.xy / (p.z + eps); pix /= (size - 1); (pix - 0.5) * 2.  One correctly rounded
    // reciprocal replaces the four divisions (<= 2 ulp on a result in [-1,1]; the kernel was VALU-bound on them).
    const float inv = 1.f / (pz + eps);
    float gx = fmaf(px * inv * isx, 2.f, -1.f);
    float gy = fmaf(py * inv * isy, 2.f, -1.f);
    if (clamp) {  // face_tracker.py:606
      gx = fminf(fmaxf(gx, -1.f), 1.f);
      gy = fminf(fmaxf(gy, -1.f), 1.f);
    }
    gdst[i] = make_float2(gx, gy);
    if (zout) zout[(int64_t)f * hw + i] = pz;
  }
}

// ---- grid sample ------------------------------------------------------------------------------------------------
template <int BORDER>
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ img, int64_t img_stride,
                                                          const float* __restrict__ grid, float* __restrict__ out, int IH,
                                                          int IW, int64_t opix) {
  const int f = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= opix) return;
  const float2 g = reinterpret_cast<const float2*>(grid)[(int64_t)f * opix + i];
  // grid_sampler_unnormalize(align_corners=False): ((g + 1) * size - 1) / 2
  float ix = ((g.x + 1.f) * (float)IW - 1.f) / 2.f;
  float iy = ((g.y + 1.f) * (float)IH - 1.f) / 2.f;
  if (BORDER) {  // clip_coordinates
    ix = fminf((float)(IW - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(IH - 1), fmaxf(iy, 0.f));
  }
  const float xw = floorf(ix), yn = floorf(iy);
  const float wx = ix - xw, ex = 1.f - wx, ny = iy - yn, sy = 1.f - ny;
  const float wraw[4] = {sy * ex, sy * wx, ny * ex, ny * wx};
  const int x0 = (int)fminf(fmaxf(xw, -2.f), (float)IW + 1.f);
  const int y0 = (int)fminf(fmaxf(yn, -2.f), (float)IH + 1.f);
  const float* src = img + img_stride * f;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const bool ok = (unsigned)xx < (unsigned)IW && (unsigned)yy < (unsigned)IH && xw + (float)(t & 1) == (float)xx &&
                    yn + (float)(t >> 1) == (float)yy;
    const float w = ok ? wraw[t] : 0.f;
    const int xc = min(max(xx, 0), IW - 1), yc = min(max(yy, 0), IH - 1);
    const float* p = src + ((int64_t)yc * IW + xc) * 3;
    // ATen accumulates nw, ne, sw, se in this order (GridSampler.cpp bilinear branch)
    acc[0] = acc[0] + p[0] * w;
    acc[1] = acc[1] + p[1] * w;
    acc[2] = acc[2] + p[2] * w;
  }
  float* o = out + ((int64_t)f * opix + i) * 3;
  o[0] = acc[0], o[1] = acc[1], o[2] = acc[2];
}


// ---- canonical-depth photometric loss (training.py:462-477, 621-634) ---------------------------------------------------------
//   pred = grid_sample(rgb_face_gt, Project3D(BackprojectDepth(canonical_depth_head), K, rel_pose), padding_mode='border')
//   loss = weights * sum((pred - rgb_face_canonical)^2 * mask) / (sum(mask) + 1e-6)
// and its gradient with respect to the depth map -- the only parameter this term trains.  The whole chain is LOCAL: output pixel
// p depends on depth[p] alone (the depth moves the sampling position of pixel p in the observed frame), so one thread evaluates
// pixel p of every frame forward and backward and d_depth needs no scatter.  Forward arithmetic follows warp_grid_kernel and
// grid_sample_kernel<1>; the backward follows ATen's grid_sampler_2d_backward (border: zero gradient where the coordinate was
// clipped; bilinear: the four tap values times the opposite-corner distances).
struct DepthLossArgs {
  const float* depth;    // [H,W]
  const float* T;        // [F,16]
  const float* src;      // [F,H,W,3]  rgb_face_gt
  const float* target;   // [H,W,3] (stride 0) or [F,H,W,3]  rgb_face_canonical
  const float* mask;     // NULL, or like target
  float* d_raw;          // [H,W]: sum over frames of d numerator / d depth
  float* part;           // [blocks][2]: numerator, denominator partial sums
  int64_t target_stride, mask_stride;
  float focal, cx, cy, eps;
  int H, W, F;
};

__global__ __launch_bounds__(256) void depth_photo_kernel(DepthLossArgs a) {
  __shared__ float red[2][256];
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float num = 0.f, den = 0.f, graw = 0.f;
  if (i < hw) {
    const int y = i / a.W, x = i - y * a.W;
    const float inv_f = 1.f / a.focal, ncx = -a.cx / a.focal, ncy = -a.cy / a.focal;
    const float isx = 1.f / (float)(a.W - 1), isy = 1.f / (float)(a.H - 1);
    const float d = a.depth[i];
    const float rx = fmaf((float)x, inv_f, ncx), ry = fmaf((float)y, inv_f, ncy);
    for (int f = 0; f < a.F; ++f) {
      const float* t = a.T + 16 * (int64_t)f;
      float p0[4], p1[4], p2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p2[j] = t[8 + j];
        p0[j] = fmaf(a.focal, t[j], a.cx * p2[j]);
        p1[j] = fmaf(a.focal, t[4 + j], a.cy * p2[j]);
      }
      const float X = d * rx, Y = d * ry, Z = d;
      const float px = fmaf(p0[0], X, fmaf(p0[1], Y, fmaf(p0[2], Z, p0[3])));
      const float py = fmaf(p1[0], X, fmaf(p1[1], Y, fmaf(p1[2], Z, p1[3])));
      const float pz = fmaf(p2[0], X, fmaf(p2[1], Y, fmaf(p2[2], Z, p2[3])));
      const float inv = 1.f / (pz + a.eps);
      const float gx = fmaf(px * inv * isx, 2.f, -1.f), gy = fmaf(py * inv * isy, 2.f, -1.f);
      // d (px, py, pz) / d depth
      const float dpx = fmaf(p0[0], rx, fmaf(p0[1], ry, p0[2])), dpy = fmaf(p1[0], rx, fmaf(p1[1], ry, p1[2]));
      const float dpz = fmaf(p2[0], rx, fmaf(p2[1], ry, p2[2]));
      const float dgx = 2.f * isx * (dpx - px * inv * dpz) * inv, dgy = 2.f * isy * (dpy - py * inv * dpz) * inv;
      // grid_sample(align_corners=False, border)
      float ix = ((gx + 1.f) * (float)a.W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)a.H - 1.f) / 2.f;
      float mx = 0.5f * (float)a.W, my = 0.5f * (float)a.H;                      // d ix / d gx, zero where clipped
      if (!(ix > 0.f)) ix = 0.f, mx = 0.f;
      else if (!(ix < (float)(a.W - 1))) ix = (float)(a.W - 1), mx = 0.f;
      if (!(iy > 0.f)) iy = 0.f, my = 0.f;
      else if (!(iy < (float)(a.H - 1))) iy = (float)(a.H - 1), my = 0.f;
      const float xw = floorf(ix), yn = floorf(iy);
      const float wx = ix - xw, ex = 1.f - wx, ny = iy - yn, sy = 1.f - ny;
      const int x0 = (int)xw, y0 = (int)yn;
      const float* sf = a.src + (int64_t)f * hw * 3;
      float v[4][3];
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int xx = x0 + (tq & 1), yy = y0 + (tq >> 1);
        const bool ok = xx < a.W && yy < a.H;                                    // x0, y0 >= 0 after the clip
        const float* p = sf + ((int64_t)min(yy, a.H - 1) * a.W + min(xx, a.W - 1)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[tq][c] = ok ? p[c] : 0.f;
      }
      const float* tg = a.target + a.target_stride * f + (int64_t)i * 3;
      const float* mk = a.mask ? a.mask + a.mask_stride * f + (int64_t)i * 3 : nullptr;
      float gix = 0.f, giy = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // ATen accumulates nw, ne, sw, se in this order (as grid_sample_kernel)
        float pred = 0.f;
        pred = pred + v[0][c] * (sy * ex);
        pred = pred + v[1][c] * (sy * wx);
        pred = pred + v[2][c] * (ny * ex);
        pred = pred + v[3][c] * (ny * wx);
        const float m = mk ? mk[c] : 1.f;
        const float diff = pred - tg[c];
        num = fmaf(diff * diff, m, num);
        den += m;
        const float go = 2.f * diff * m;
        gix += go * (sy * (v[1][c] - v[0][c]) + ny * (v[3][c] - v[2][c]));
        giy += go * (ex * (v[2][c] - v[0][c]) + wx * (v[3][c] - v[1][c]));
      }
      graw += gix * mx * dgx + giy * my * dgy;
    }
    a.d_raw[i] = graw;
  }
  red[0][threadIdx.x] = num;
  red[1][threadIdx.x] = den;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      red[0][threadIdx.x] += red[0][threadIdx.x + k];
      red[1][threadIdx.x] += red[1][threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.part[2 * blockIdx.x] = red[0][0];
    a.part[2 * blockIdx.x + 1] = red[1][0];
  }
}

// one block: totals of the partials (fixed order), loss = weights * num / (den + 1e-6); scale = weights / (den + 1e-6)
__global__ __launch_bounds__(256) void depth_photo_final_kernel(const float* __restrict__ part, int n, float weights, int use_mask,
                                                               float n_elems, float* __restrict__ loss_scale) {
  __shared__ float red[2][256];
  float num = 0.f, den = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    num += part[2 * i];
    den += part[2 * i + 1];
  }
  red[0][threadIdx.x] = num;
  red[1][threadIdx.x] = den;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      red[0][threadIdx.x] += red[0][threadIdx.x + k];
      red[1][threadIdx.x] += red[1][threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float denom = use_mask ? red[1][0] + 1e-6f : n_elems;        // masked mean (training.py:628-629) or torch.mean (:631)
    loss_scale[0] = red[0][0] / denom * weights;
    loss_scale[1] = weights / denom;
  }
}

__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(float* __restrict__ x, const float* __restrict__ s, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= s[1];
}

}  // namespace s2l

extern "C" int s2l_rel_pose(const float* euler, const float* trans, const float* canon_euler, const float* canon_trans,
                            int mode, float* T, int64_t n_frames, s2l_stream_t stream) {
  if (n_frames < 0) return S2L_E_SIZE;
  if (mode != S2L_POSE_OBS2CAN && mode != S2L_POSE_CAN2OBS && mode != S2L_POSE_CAN2OBS_INV) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!euler || !trans || !canon_euler || !canon_trans || !T) return S2L_E_NULL;
  hipLaunchKernelGGL(s2l::rel_pose_kernel, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), euler, trans, canon_euler, canon_trans, mode, T, n_frames);
  return (int)hipGetLastError();
}

extern "C" int s2l_warp_grid(const float* depth, int64_t depth_stride, const float* T, float focal, int clamp, float* grid,
                             float* z, int height, int width, int64_t n_frames, s2l_stream_t stream) {
  if (n_frames < 0 || height < 2 || width < 2 || (int64_t)height * width > (1 << 24) || n_frames > 65535) return S2L_E_SIZE;
  if (!(focal > 0.f) || (depth_stride != 0 && depth_stride != (int64_t)height * width)) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!depth || !T || !grid) return S2L_E_NULL;
  if (reinterpret_cast<uintptr_t>(grid) & 7) return S2L_E_ALIGN;
  const int hw = height * width;
  const int per_block = s2l::kGridThreads * s2l::kGridPix;
  hipLaunchKernelGGL(s2l::warp_grid_kernel, dim3((unsigned)((hw + per_block - 1) / per_block), (unsigned)n_frames),
                     dim3(s2l::kGridThreads), 0, static_cast<hipStream_t>(stream), depth, depth_stride, T, focal,
                     0.5f * (float)width, 0.5f * (float)height, clamp, 1e-7f, grid, z, height, width);
  return (int)hipGetLastError();
}

extern "C" int s2l_grid_sample(const float* img, int64_t img_stride, const float* grid, float* out, int img_h, int img_w,
                               int out_h, int out_w, int padding, int64_t n_frames, s2l_stream_t stream) {
  if (n_frames < 0 || img_h < 1 || img_w < 1 || out_h < 1 || out_w < 1 || n_frames > 65535) return S2L_E_SIZE;
  if (padding != S2L_SAMPLE_ZEROS && padding != S2L_SAMPLE_BORDER) return S2L_E_SIZE;
  if (img_stride != 0 && img_stride != (int64_t)img_h * img_w * 3) return S2L_E_SIZE;
  if (n_frames == 0) return S2L_OK;
  if (!img || !grid || !out) return S2L_E_NULL;
  if (reinterpret_cast<uintptr_t>(grid) & 7) return S2L_E_ALIGN;
  const int64_t opix = (int64_t)out_h * out_w;
  const dim3 g((unsigned)((opix + 255) / 256), (unsigned)n_frames);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (padding == S2L_SAMPLE_BORDER)
    hipLaunchKernelGGL(s2l::grid_sample_kernel<1>, g, dim3(256), 0, st, img, img_stride, grid, out, img_h, img_w, opix);
  else
    hipLaunchKernelGGL(s2l::grid_sample_kernel<0>, g, dim3(256), 0, st, img, img_stride, grid, out, img_h, img_w, opix);
  return (int)hipGetLastError();
}

// Canonical-depth photometric loss and d loss / d depth (see depth_photo_kernel).  depth [H,W]; T [F,16] row-major 4x4 relative
// poses (compute_rel_pose_inverse, training.py:270-275); src [F,H,W,3]; target, mask: [H,W,3] when *_stride == 0 or [F,H,W,3]
// when stride == H*W*3; mask may be NULL (plain mean).  loss: 2 floats (loss, internal scale); d_depth [H,W] (may be NULL);
// work: s2l_depth_photo_work_floats(H, W) floats.
extern "C" int64_t s2l_depth_photo_work_floats(int height, int width) {
  if (height < 2 || width < 2) return 0;
  const int64_t hw = (int64_t)height * width;
  return hw + 2 * ((hw + 255) / 256);
}

extern "C" int s2l_depth_photo_loss(const float* depth, const float* T, float focal, const float* src, const float* target,
                                    int64_t target_stride, const float* mask, int64_t mask_stride, float weights, float* work,
                                    float* loss, float* d_depth, int height, int width, int64_t n_frames, s2l_stream_t stream) {
  const int64_t hw = (int64_t)height * width;
  if (n_frames <= 0 || height < 2 || width < 2 || hw > (1 << 24) || n_frames > 65535 || !(focal > 0.f)) return S2L_E_SIZE;
  if ((target_stride != 0 && target_stride != hw * 3) || (mask_stride != 0 && mask_stride != hw * 3)) return S2L_E_SIZE;
  if (!depth || !T || !src || !target || !work || !loss) return S2L_E_NULL;
  s2l::DepthLossArgs a;
  a.depth = depth; a.T = T; a.src = src; a.target = target; a.mask = mask;
  a.target_stride = target_stride; a.mask_stride = mask_stride;
  a.d_raw = d_depth ? d_depth : work;
  a.part = work + hw;
  a.focal = focal; a.cx = 0.5f * (float)width; a.cy = 0.5f * (float)height; a.eps = 1e-7f;
  a.H = height; a.W = width; a.F = (int)n_frames;
  const int blocks = (int)((hw + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(s2l::depth_photo_kernel, dim3(blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(s2l::depth_photo_final_kernel, dim3(1), dim3(256), 0, st, a.part, blocks, weights, mask ? 1 : 0,
                     (float)(hw * 3 * n_frames), loss);
  if (d_depth) hipLaunchKernelGGL(s2l::scale_by_device_scalar_kernel, dim3(blocks), dim3(256), 0, st, d_depth, loss, (int)hw);
  return (int)hipGetLastError();
}
