// The perceptual term of the stage-1 loss: Trainer.add_perceptual_loss (src/face_simple/training.py:655-674)
// calls lpips.LPIPS(net='alex', version='0.1') (training.py:76) on the lip image (:420-421) and on the fused face (:453-456).
// `lpips` is a third-party package (requirement.txt:11 pins lpips==0.1.4) that is not part of the reference repository; what
// follows restates its published forward pass (lpips/lpips.py LPIPS.forward, lpips/pretrained_networks.py alexnet,
// lpips/__init__.py normalize_tensor / spatial_average):
//     x  = (in - shift) / scale                      per channel, shift = (-.030, -.088, -.188), scale = (.458, .448, .450)
//     a1 = relu(conv(x, 3->64, k11 s4 p2))           a2 = relu(conv(maxpool3s2(a1), 64->192, k5 p2))
//     a3 = relu(conv(maxpool3s2(a2), 192->384, k3 p1))   a4 = relu(conv(a3, 384->256, k3 p1))   a5 = relu(conv(a4, 256->256, k3 p1))
//     u_l = a_l / (sqrt(sum_c a_l^2) + 1e-10)        for both images
//     d   = sum_l mean_pixels( sum_c w_l[c] (u_l(in0) - u_l(in1))^2 )          -> [N,1,1,1]       (dropout is inactive in eval)
// and adds the gradient with respect to in0 (the net and the linear heads are frozen: requires_grad=False in the package).
// Both images go through the trunk as ONE batch of 2N; the convolutions are the implicit-GEMM kernel of csrc/conv_gemm.h
// (fp32 MFMA), except the 64 -> 3 input gradient of conv1, which is a small VALU kernel (a 64-row GEMM tile would be 95 %
// padding).  ~20 GFLOP per 500x500 image pair forward + backward against 315 for the U-Net next to it: written for
// exactness and determinism (no atomics), not for speed.
// Parity: weights of the real package are not available here (they come from torchvision + the package's own alex.pth):
// tests compare with the CPU restatement of the same published algorithm on seeded weights -- structural parity, as for SyncNet.
#include "conv_gemm.h"

namespace s2l {

static const LayerSpec kAlex[5] = {{3, 64, 11, 11, 4, 4, 2, 2, 0}, {64, 192, 5, 5, 1, 1, 2, 2, 0}, {192, 384, 3, 3, 1, 1, 1, 1, 0},
                                   {384, 256, 3, 3, 1, 1, 1, 1, 0}, {256, 256, 3, 3, 1, 1, 1, 1, 0}};
constexpr int kAlexLayers = 5;
constexpr float kLpipsEps = 1e-10f;

struct LpipsPacked {
  int64_t w[kAlexLayers], b[kAlexLayers], wt[kAlexLayers], lin[kAlexLayers], w1raw, shift, scale, total;
};
inline LpipsPacked lpips_packed() {
  LpipsPacked p;
  int64_t o = 0;
  for (int l = 0; l < kAlexLayers; ++l) {
    const LayerSpec& s = kAlex[l];
    p.w[l] = o;
    o += (int64_t)s.kh * s.kw * ceil_to(s.cin, 16) * ceil_to(s.cout, 64);
    p.b[l] = o;
    o += ceil_to(s.cout, 64);
    p.wt[l] = o;                       // dgrad operand (layer 0 uses the raw weights instead)
    if (l > 0) o += (int64_t)s.kh * s.kw * ceil_to(s.cout, 16) * ceil_to(s.cin, 64);
    p.lin[l] = o;
    o += ceil_to(s.cout, 4);
  }
  p.w1raw = o;
  o += 64 * 3 * 11 * 11;               // conv1.weight as it is, [co][ci][ky][kx]
  p.shift = o;
  o += 4;
  p.scale = o;
  o += 4;
  p.total = (o + 3) / 4 * 4;
  return p;
}

inline Shape pool_shape(Shape in) { return Shape{(in.h - 3) / 2 + 1, (in.w - 3) / 2 + 1}; }

// work buffer (floats): scaled inputs, activations and pooled maps of the 2N batch, per-pixel sums, and for the backward pass
// the tap gradients of the N generated images, two ping-pong gradient buffers, the split-K scratch
struct LpipsWork {
  Shape a[kAlexLayers], p[2];
  int64_t xs, act[kAlexLayers], pool[2], pix, tap[kAlexLayers], grad[2], partial, total;
};
inline LpipsWork lpips_work(int H, int W, int64_t N) {
  LpipsWork wl;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
  wl.xs = take(2 * N * H * W * 3);
  Shape sh{H, W};
  int64_t max_pix = 0, max_act = (int64_t)H * W * 3;
  for (int l = 0; l < kAlexLayers; ++l) {
    sh = out_shape(kAlex[l], sh);
    wl.a[l] = sh;
    wl.act[l] = take(2 * N * sh.h * sh.w * kAlex[l].cout);
    max_pix = std::max<int64_t>(max_pix, (int64_t)sh.h * sh.w);
    max_act = std::max<int64_t>(max_act, (int64_t)sh.h * sh.w * kAlex[l].cout);
    if (l < 2) {
      sh = pool_shape(sh);
      wl.p[l] = sh;
      wl.pool[l] = take(2 * N * sh.h * sh.w * kAlex[l].cout);
    }
  }
  wl.pix = take(N * max_pix);
  for (int l = 0; l < kAlexLayers; ++l) wl.tap[l] = take(N * wl.a[l].h * wl.a[l].w * kAlex[l].cout);
  for (int k = 0; k < 2; ++k) wl.grad[k] = take(N * max_act);
  wl.partial = take(kPartialFloats);
  wl.total = o;
  return wl;
}

// ---- elementwise ends ------------------------------------------------------------------------------------------------------
// ScalingLayer (lpips.py): (inp - shift) / scale; in0 -> images [0,N), in1 -> images [N,2N) of the trunk's batch
__global__ __launch_bounds__(256) void lpips_scale_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                         const float* __restrict__ shift, const float* __restrict__ scale,
                                                         float* __restrict__ xs, int64_t n_half, int from01) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * n_half) return;
  const int c = (int)(i % 3);
  float v = i < n_half ? in0[i] : in1[i - n_half];
  if (from01) v = __fmul_rn(__fsub_rn(v, 0.5f), 2.f);     // add_perceptual_loss: (x - 0.5) * 2, training.py:669-670
  xs[i] = (v - shift[c]) / scale[c];
}

__global__ __launch_bounds__(256) void lpips_unscale_grad_kernel(const float* __restrict__ dxs, const float* __restrict__ scale,
                                                                float* __restrict__ d_in0, int64_t n, float chain, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d_in0[i] = (accumulate ? d_in0[i] : 0.f) + dxs[i] / scale[i % 3] * chain;
}

// MaxPool2d(kernel_size=3, stride=2), no padding, floor mode (torchvision alexnet.features[2], [5]); NHWC, thread per f4
__global__ __launch_bounds__(256) void lpips_pool_kernel(const float* __restrict__ a, float* __restrict__ p, int hin, int win,
                                                        int hout, int wout, int C, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c4 = C / 4;
  const int cq = (int)(i % c4);
  int64_t r = i / c4;
  const int ox = (int)(r % wout);
  r /= wout;
  const int oy = (int)(r % hout);
  const int64_t b = r / hout;
  f4 m = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f4 v = *reinterpret_cast<const f4*>(a + ((b * hin + 2 * oy + ky) * (int64_t)win + 2 * ox + kx) * C + 4 * cq);
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
    }
  *reinterpret_cast<f4*>(p + i * 4) = m;
}

// adjoint of the pooling as a gather, fused with what follows on the way down:
//     g_a = (sum over the windows whose FIRST maximum (scan order, as ATen) is this element of g_p  +  tap) * (a > 0)
__global__ __launch_bounds__(256) void lpips_pool_bwd_kernel(const float* __restrict__ a, const float* __restrict__ gp,
                                                            const float* __restrict__ tap, float* __restrict__ ga, int hin, int win,
                                                            int hout, int wout, int C, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  int64_t r = i / C;
  const int x = (int)(r % win);
  r /= win;
  const int y = (int)(r % hin);
  const int64_t b = r / hin;
  const float self = a[i];
  float g = 0.f;
  if (self > 0.f) {    // (a == 0 is masked below anyway)
    const int oy_lo = max(0, (y - 1) / 2), oy_hi = min(hout - 1, y / 2);     // windows with 2 oy <= y <= 2 oy + 2
    const int ox_lo = max(0, (x - 1) / 2), ox_hi = min(wout - 1, x / 2);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        float m = -INFINITY;
        int arg = -1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v = a[((b * hin + 2 * oy + ky) * (int64_t)win + 2 * ox + kx) * C + c];
            if (v > m) { m = v; arg = ky * 3 + kx; }
          }
        if (arg == (y - 2 * oy) * 3 + (x - 2 * ox)) g += gp[((b * hout + oy) * (int64_t)wout + ox) * C + c];
      }
  }
  const float t = g + tap[i];
  ga[i] = self > 0.f ? t : 0.f;
}

// ---- the LPIPS head of one tap: one wave per (image, pixel) ---------------------------------------------------------------
// f0 = a[n], f1 = a[N + n];  s = sum_c w[c] (f0/(|f0| + eps) - f1/(|f1| + eps))^2            -> pix[n][pixel]
// backward (d_scale = d_out[n] / pixels):  g_c = 2 w_c (u_c - v_c) d_scale;  tap_c = (g_c - (g . f0) f0_c / ((|f0| + eps) |f0|)) / (|f0| + eps),
// times (f0_c > 0): the ReLU under the tap.  A pixel whose 64..384 features are all zero gives 0/0 = NaN there, as torch's
// autograd does for x / (sqrt(sum x^2) + eps) at x = 0.
template <bool BACKWARD>
__global__ __launch_bounds__(64) void lpips_head_kernel(const float* __restrict__ a, const float* __restrict__ w, int C, int64_t npix,
                                                       int64_t N, float* __restrict__ pix, const float* __restrict__ d_out,
                                                       float* __restrict__ tap) {
  const int64_t id = blockIdx.x;              // n * npix + pixel
  const int64_t n = id / npix;
  const int l = threadIdx.x;
  const float* f0 = a + id * C;
  const float* f1 = a + (N * npix + id) * C;
  float s0 = 0.f, s1 = 0.f;
  for (int c = l; c < C; c += 64) {
    s0 = fmaf(f0[c], f0[c], s0);
    s1 = fmaf(f1[c], f1[c], s1);
  }
  for (int o = 32; o; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
  }
  const float r0 = sqrtf(s0), n0 = r0 + kLpipsEps, n1 = sqrtf(s1) + kLpipsEps;
  if (!BACKWARD) {
    float s = 0.f;
    for (int c = l; c < C; c += 64) {
      const float e = f0[c] / n0 - f1[c] / n1;
      s = fmaf(w[c], e * e, s);
    }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0) pix[id] = s;
  } else {
    const float ds = d_out[n] / (float)npix;
    float dot = 0.f;
    for (int c = l; c < C; c += 64) {
      const float g = 2.f * w[c] * (f0[c] / n0 - f1[c] / n1) * ds;
      dot = fmaf(g, f0[c], dot);
    }
    for (int o = 32; o; o >>= 1) dot += __shfl_xor(dot, o);
    const float k = dot / (n0 * r0);
    for (int c = l; c < C; c += 64) {
      const float g = 2.f * w[c] * (f0[c] / n0 - f1[c] / n1) * ds;
      const float t = (g - k * f0[c]) / n0;
      tap[id * C + c] = f0[c] > 0.f ? t : (t != t ? t : 0.f);      // (NaN stays NaN)
    }
  }
}

// out[n] (+)= mean over the pixels, summed in a fixed order (one workgroup per image)
__global__ __launch_bounds__(256) void lpips_mean_kernel(const float* __restrict__ pix, int64_t npix, float* __restrict__ out,
                                                        int accumulate) {
  __shared__ float part[256];
  const int64_t n = blockIdx.x;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < npix; i += 256) s += pix[n * npix + i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = (accumulate ? out[n] : 0.f) + part[0] / (float)npix;
}

// input gradient of conv1 (3 <- 64 channels, k11 s4 p2): thread per input pixel, all three channels
__global__ __launch_bounds__(256) void lpips_conv1_dgrad_kernel(const float* __restrict__ g1, const float* __restrict__ w,
                                                               float* __restrict__ dxs, int H, int W, int h1, int w1, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  int64_t r = i / W;
  const int y = (int)(r % H);
  const int64_t b = r / H;
  float acc[3] = {0.f, 0.f, 0.f};
  // outputs with ky = y + 2 - 4 oy in [0, 11)
  const int oy_hi = min(h1 - 1, (y + 2) / 4), ox_hi = min(w1 - 1, (x + 2) / 4);
  for (int oy = max(0, (y + 2 - 10 + 3) / 4); oy <= oy_hi; ++oy) {
    const int ky = y + 2 - 4 * oy;
    if (ky < 0 || ky > 10) continue;
    for (int ox = max(0, (x + 2 - 10 + 3) / 4); ox <= ox_hi; ++ox) {
      const int kx = x + 2 - 4 * ox;
      if (kx < 0 || kx > 10) continue;
      const float* g = g1 + ((b * h1 + oy) * (int64_t)w1 + ox) * 64;
      for (int co = 0; co < 64; ++co) {
        const float gv = g[co];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) acc[ci] = fmaf(gv, w[((co * 3 + ci) * 11 + ky) * 11 + kx], acc[ci]);
      }
    }
  }
  dxs[i * 3 + 0] = acc[0];
  dxs[i * 3 + 1] = acc[1];
  dxs[i * 3 + 2] = acc[2];
}

__global__ void lpips_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

static int lpips_check(int H, int W, int64_t N) {
  if (N < 0 || N > 4096 || H < 1 || W < 1 || H > 4096 || W > 4096) return S2L_E_SIZE;
  // the second pooling needs a 3x3 window: conv1 -> (H-7)/4+1, pool -> (.-3)/2+1, pool again
  Shape sh = pool_shape(out_shape(kAlex[0], Shape{H, W}));
  if (H < 7 || W < 7 || sh.h < 3 || sh.w < 3) return S2L_E_GEOMETRY;
  return S2L_OK;
}

}  // namespace s2l

using namespace s2l;

extern "C" int64_t s2l_lpips_packed_floats(void) { return lpips_packed().total; }
extern "C" int64_t s2l_lpips_work_floats(int height, int width, int64_t batch) {
  return lpips_check(height, width, batch) || batch == 0 ? 0 : lpips_work(height, width, batch).total;
}

// tensors_host: 17 DEVICE pointers in a host array -- conv1..conv5 {weight [co,ci,kh,kw], bias [co]}, lin0..lin4 weight
// [1,C,1,1], scaling_layer shift [3], scale [3]
extern "C" int s2l_lpips_pack(const float* const* tensors_host, float* packed, s2l_stream_t stream) {
  if (!tensors_host || !packed) return S2L_E_NULL;
  for (int i = 0; i < 17; ++i)
    if (!tensors_host[i]) return S2L_E_NULL;
  if (misaligned16(packed)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const LpipsPacked pl = lpips_packed();
  const float* none = nullptr;
  for (int l = 0; l < kAlexLayers; ++l) {
    const LayerSpec& s = kAlex[l];
    const int RP = ceil_to(s.cout, 64), kcp = ceil_to(s.cin, 16);
    const int64_t n = (int64_t)s.kh * s.kw * kcp * RP;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tensors_host[2 * l], none, none, 0.f,
                       packed + pl.w[l], s.cin, s.cout, s.kh, s.kw, kcp, RP, 0, n);
    hipLaunchKernelGGL(conv_pack_bias_kernel, dim3((RP + 255) / 256), dim3(256), 0, st, tensors_host[2 * l + 1], none, none, none, none,
                       0.f, packed + pl.b[l], s.cout, RP);
    if (l > 0) {
      const int RPt = ceil_to(s.cin, 64), kcpt = ceil_to(s.cout, 16);
      const int64_t nt = (int64_t)s.kh * s.kw * kcpt * RPt;
      hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, tensors_host[2 * l], none, none, 0.f,
                         packed + pl.wt[l], s.cin, s.cout, s.kh, s.kw, kcpt, RPt, 1, nt);
    }
    hipLaunchKernelGGL(lpips_copy_kernel, dim3((s.cout + 255) / 256), dim3(256), 0, st, tensors_host[10 + l], packed + pl.lin[l],
                       (int64_t)s.cout);
  }
  hipLaunchKernelGGL(lpips_copy_kernel, dim3((64 * 363 + 255) / 256), dim3(256), 0, st, tensors_host[0], packed + pl.w1raw,
                     (int64_t)64 * 363);
  hipLaunchKernelGGL(lpips_copy_kernel, dim3(1), dim3(64), 0, st, tensors_host[15], packed + pl.shift, (int64_t)3);
  hipLaunchKernelGGL(lpips_copy_kernel, dim3(1), dim3(64), 0, st, tensors_host[16], packed + pl.scale, (int64_t)3);
  return (int)hipGetLastError();
}

extern "C" int s2l_lpips_forward(const float* packed, const float* in0, const float* in1, int from01, float* work, float* out,
                                 int height, int width, int64_t batch, s2l_stream_t stream) {
  int rc = lpips_check(height, width, batch);
  if (rc || batch == 0) return rc;
  if (!packed || !in0 || !in1 || !work || !out) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(work)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const LpipsPacked pl = lpips_packed();
  const LpipsWork wl = lpips_work(height, width, batch);
  const int64_t N = batch, nh = N * height * width * 3;
  hipLaunchKernelGGL(lpips_scale_kernel, dim3((unsigned)((2 * nh + 255) / 256)), dim3(256), 0, st, in0, in1, packed + pl.shift,
                     packed + pl.scale, work + wl.xs, nh, from01);
  Shape sh{height, width};
  const float* cur = work + wl.xs;
  for (int l = 0; l < kAlexLayers; ++l) {
    const LayerSpec& s = kAlex[l];
    ConvArgs a = base_args(s, sh, wl.a[l]);
    a.in = cur;
    a.w = packed + pl.w[l];
    a.bias = packed + pl.b[l];
    a.out = work + wl.act[l];
    a.partial = work + wl.partial;
    if ((rc = launch_conv<false>(a, 2 * N, st))) return rc;
    sh = wl.a[l];
    cur = work + wl.act[l];
    const int64_t npix = (int64_t)sh.h * sh.w;
    hipLaunchKernelGGL(lpips_head_kernel<false>, dim3((unsigned)(N * npix)), dim3(64), 0, st, cur, packed + pl.lin[l], s.cout, npix, N,
                       work + wl.pix, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(lpips_mean_kernel, dim3((unsigned)N), dim3(256), 0, st, work + wl.pix, npix, out, l > 0);
    if (l < 2) {
      const Shape ps = wl.p[l];
      const int64_t n = 2 * N * ps.h * ps.w * (s.cout / 4);
      hipLaunchKernelGGL(lpips_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cur, work + wl.pool[l], sh.h, sh.w,
                         ps.h, ps.w, s.cout, n);
      sh = ps;
      cur = work + wl.pool[l];
    }
  }
  return (int)hipGetLastError();
}

// `work` as s2l_lpips_forward left it; d_out [N] = d loss / d out[n]; d_in0 [N,H,W,3]
extern "C" int s2l_lpips_backward(const float* packed, float* work, const float* d_out, int from01, int accumulate, float* d_in0,
                                  int height, int width, int64_t batch, s2l_stream_t stream) {
  int rc = lpips_check(height, width, batch);
  if (rc || batch == 0) return rc;
  if (!packed || !work || !d_out || !d_in0) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(work)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const LpipsPacked pl = lpips_packed();
  const LpipsWork wl = lpips_work(height, width, batch);
  const int64_t N = batch;
  for (int l = 0; l < kAlexLayers; ++l) {      // every tap's own gradient (already times the ReLU mask of its layer)
    const int64_t npix = (int64_t)wl.a[l].h * wl.a[l].w;
    hipLaunchKernelGGL(lpips_head_kernel<true>, dim3((unsigned)(N * npix)), dim3(64), 0, st, work + wl.act[l], packed + pl.lin[l],
                       kAlex[l].cout, npix, N, (float*)nullptr, d_out, work + wl.tap[l]);
  }
  // g_l = d loss / d z_l (pre-ReLU) of the N generated images.  conv5 -> conv4 -> conv3: (dgrad + tap) * relu mask in the epilogue
  const float* g = work + wl.tap[4];
  float* bufs[2] = {work + wl.grad[0], work + wl.grad[1]};
  int k = 0;
  for (int l = 4; l >= 3; --l) {
    ConvArgs a = base_args(kAlex[l], wl.a[l - 1], wl.a[l]);
    a.in = g;
    a.w = packed + pl.wt[l];
    a.res = work + wl.tap[l - 1];
    a.mask = work + wl.act[l - 1];
    a.out = bufs[k];
    a.partial = work + wl.partial;
    if ((rc = launch_conv<true>(a, N, st))) return rc;
    g = bufs[k];
    k ^= 1;
  }
  // conv3 -> pool2 -> a2, conv2 -> pool1 -> a1: dgrad to the pooled map, then the pooling adjoint + tap + relu mask
  for (int l = 2; l >= 1; --l) {
    ConvArgs a = base_args(kAlex[l], wl.p[l - 1], wl.a[l]);
    a.in = g;
    a.w = packed + pl.wt[l];
    a.out = bufs[k];
    a.partial = work + wl.partial;
    if ((rc = launch_conv<true>(a, N, st))) return rc;
    const Shape as = wl.a[l - 1], ps = wl.p[l - 1];
    const int C = kAlex[l - 1].cout;
    const int64_t n = N * as.h * as.w * C;
    hipLaunchKernelGGL(lpips_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work + wl.act[l - 1], bufs[k],
                       work + wl.tap[l - 1], bufs[k ^ 1], as.h, as.w, ps.h, ps.w, C, n);
    g = bufs[k ^ 1];
  }
  const int64_t npx = N * height * width;
  float* dxs = bufs[g == bufs[0] ? 1 : 0];
  hipLaunchKernelGGL(lpips_conv1_dgrad_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, g, packed + pl.w1raw, dxs, height,
                     width, wl.a[0].h, wl.a[0].w, npx);
  hipLaunchKernelGGL(lpips_unscale_grad_kernel, dim3((unsigned)((npx * 3 + 255) / 256)), dim3(256), 0, st, dxs, packed + pl.scale, d_in0,
                     npx * 3, from01 ? 2.f : 1.f, accumulate);
  return (int)hipGetLastError();
}
