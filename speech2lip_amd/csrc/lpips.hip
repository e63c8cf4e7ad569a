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
// Both images go through the trunk as ONE batch of 2N; conv2..conv5 are the implicit-GEMM kernel of csrc/conv_gemm.h (fp32 MFMA);
// conv1 (k11 s4 on 3 channels, the largest launch at 500x500) and its 64 -> 3 input gradient have their own fp32-MFMA kernels
// below (round 4: 1.0 -> ~0.1 ms and 0.64 -> ~0.15 ms for 16 / 8 images).  ~20 GFLOP per 500x500 image pair forward + backward
// against 315 for the U-Net next to it; exact fp32 products, deterministic (no atomics).
// Parity: weights of the real package are not available here (they come from torchvision + the package's own alex.pth):
// tests compare with the CPU restatement of the same published algorithm on seeded weights -- structural parity, as for SyncNet.
#include "conv_gemm.h"

namespace s2l {

static const LayerSpec kAlex[5] = {{3, 64, 11, 11, 4, 4, 2, 2, 0}, {64, 192, 5, 5, 1, 1, 2, 2, 0}, {192, 384, 3, 3, 1, 1, 1, 1, 0},
                                   {384, 256, 3, 3, 1, 1, 1, 1, 0}, {256, 256, 3, 3, 1, 1, 1, 1, 0}};
constexpr int kAlexLayers = 5;
constexpr float kLpipsEps = 1e-10f;

struct LpipsPacked {
  int64_t w[kAlexLayers], b[kAlexLayers], wt[kAlexLayers], lin[kAlexLayers], w1raw, wd1, shift, scale, w16[kAlexLayers], wt16[kAlexLayers], total;
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
  o = (o + 3) / 4 * 4;
  p.wd1 = o;
  o += 4 * 12288;                      // conv1's input-gradient operand (lpips_conv1_dgrad_pack_kernel)
  p.shift = o;
  o += 4;
  p.scale = o;
  o += 4;
  o = (o + 3) / 4 * 4;
  for (int l = 1; l < kAlexLayers; ++l) {      // the split form's operands (conv_gemm.h) of the layers the implicit GEMM runs
    const LayerSpec& s = kAlex[l];
    p.w16[l] = o;
    o += packed16_floats(s.kh * s.kw, s.cin, s.cout);
    p.wt16[l] = o;
    o += packed16_floats(s.kh * s.kw, s.cout, s.cin);
  }
  p.w16[0] = p.wt16[0] = 0;
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

// ---- conv1 (3 -> 64, k11 s4 p2) and its input gradient as their own fp32-MFMA kernels ------------------------------------------------
// The generic implicit-GEMM kernel gathers conv1's K = 363 operand element by element (3 channels per pixel, stride 4): 11 TFLOP/s, the
// largest launch of the perceptual term at 500 x 500 (1.0 ms of 2.4 for 16 images).  Here a workgroup stages the input patch of a 16 x 16
// output tile in LDS once (71 rows x 72 pixels x 3), wave w keeps the weights of output channels 16 w .. 16 w + 15 in REGISTERS for
// the whole launch (the A operand of v_mfma_f32_16x16x4_f32: a kernel row = 33 contiguous floats of the patch, padded to 36 so that a
// k-step never straddles two rows -- the three pad weights are zeros and their B values the next pixel's, finite), and every MFMA needs
// one 4-byte LDS read: lane (n, q) reads patch[(4 oy + ky) * pitch + 12 ox_n + j0 + q].  K order (ky, kx, c), one accumulator per
// output element, k ascending.
constexpr int kC1T = 16;                            // output tile edge
constexpr int kC1Rows = (kC1T - 1) * 4 + 11;        // 71 patch rows
constexpr int kC1Pitch = (kC1Rows + 1) * 3;         // 72 pixels x 3 floats per row
constexpr int kC1Steps = 11 * 9;                    // k-steps: 11 kernel rows x 36 / 4

__global__ __launch_bounds__(256, 2) void lpips_conv1_kernel(const float* __restrict__ xs, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out, int H, int W, int h1,
                                                            int w1, int tiles_x, int tiles_y, int n_tiles) {
  __shared__ float patch[kC1Rows * kC1Pitch];       // 61 344 B: two workgroups per CU
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
  float a[kC1Steps];
#pragma unroll
  for (int s = 0; s < kC1Steps; ++s) {
    const int ky = s / 9, j = 4 * (s % 9) + q;      // position in the padded kernel row: kx * 3 + c
    a[s] = j < 33 ? w[((16 * wave + l16) * 3 + j % 3) * 121 + ky * 11 + j / 3] : 0.f;
  }
  const f4 bv = *reinterpret_cast<const f4*>(bias + 16 * wave + 4 * q);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
    const int64_t img = tile / (tiles_x * tiles_y);
    const int y0 = ty * kC1T * 4 - 2, c0 = (tx * kC1T * 4 - 2) * 3;      // patch origin: row, float column
    __syncthreads();                                 // (the previous tile's reads)
    for (int base = 0; base < kC1Rows * kC1Pitch; base += 256 * 12) {      // (twelve loads in flight per thread, then their stores)
      float tv[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int idx = base + threadIdx.x + 256 * k;
        const int r = idx / kC1Pitch, cc = idx % kC1Pitch;
        const int gy = y0 + r, gc = c0 + cc;
        const bool in = idx < kC1Rows * kC1Pitch && (unsigned)gy < (unsigned)H && (unsigned)gc < (unsigned)(3 * W);
        const float v = xs[in ? (img * H + gy) * (int64_t)W * 3 + gc : 0];
        tv[k] = in ? v : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int idx = base + threadIdx.x + 256 * k;
        if (idx < kC1Rows * kC1Pitch) patch[idx] = tv[k];
      }
    }
    __syncthreads();
    for (int g = 0; g < 4; ++g) {                    // four output rows at a time
      const float* pb = patch + g * 16 * kC1Pitch + 12 * l16 + q;
      f4 acc[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rr] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < kC1Steps; ++s) {
        const int off = (s / 9) * kC1Pitch + 4 * (s % 9);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[rr] = mfma16(a[s], pb[rr * 4 * kC1Pitch + off], acc[rr]);
      }
      const int ox = tx * kC1T + l16;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int oy = ty * kC1T + 4 * g + rr;
        if (oy < h1 && ox < w1) {
          f4 v = acc[rr] + bv;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          *reinterpret_cast<f4*>(out + ((img * h1 + oy) * (int64_t)w1 + ox) * 64 + 16 * wave + 4 * q) = v;
        }
      }
    }
  }
}

// The input gradient of conv1 as a GEMM per 4 x 4 block of input pixels (stride 4: a block's 48 values (ry, rx, c) depend on the 4 x 4
// neighbourhood of output pixels (by + 1 - r, bx + 1 - cc), r, cc = 0..3, through taps ky = 4 r + ry - 2, kx = 4 cc + rx - 2 where
// those are kernel positions, zero otherwise: 47 % of the 48 x 1024 operand is non-zero, the price of one dense shape):
//     d_in[4 by + ry][4 bx + rx][c] = sum_{r, cc, co} Wd[(ry, rx, c)][(r, cc, co)] g[by + 1 - r][bx + 1 - cc][co].
// A workgroup owns 8 x 16 blocks (32 x 64 input pixels): the gradient patch (11 x 19 output pixels x 64 channels, pixel stride 66
// floats: the sixteen pixels x two k of a half-wave read land on 32 distinct banks) stays in LDS, Wd streams through two 48-KiB LDS
// buffers in four passes (one per r, descending: r' = 3 - r, so that patch offsets grow), wave w computes block rows 2 w, 2 w + 1
// for all 48 rows: six accumulator tiles, five LDS reads per six MFMAs.  The VALU kernel this replaces ran at 9 TFLOP/s.
constexpr int kD1TH = 8, kD1TW = 16, kD1Rows = kD1TH + 3, kD1Cols = kD1TW + 3, kD1Pix = 66;
constexpr int kD1B = 13824, kD1A = 12288;                       // floats: patch (11 x 19 x 66 = 13 794, rounded), one pass of Wd
constexpr int kD1Lds = (kD1B + 2 * kD1A) * 4;                   // 153 600 B

// wd[r'][k-step s][q][48]: Wd[m][k = 4 s + q of pass r'] with k = cc' * 64 + co, r = 3 - r', cc = 3 - cc'
__global__ __launch_bounds__(256) void lpips_conv1_dgrad_pack_kernel(const float* __restrict__ w, float* __restrict__ wd) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 4 * kD1A) return;
  const int rp = e / kD1A, rem = e % kD1A, k = rem / 48, m = rem % 48;
  const int r = 3 - rp, cc = 3 - k / 64, co = k % 64;
  const int ry = m / 12, rx = (m / 3) % 4, c = m % 3;
  const int ky = 4 * r + ry - 2, kx = 4 * cc + rx - 2;
  wd[e] = ((unsigned)ky < 11u && (unsigned)kx < 11u) ? w[((co * 3 + c) * 11 + ky) * 11 + kx] : 0.f;
}

__global__ __launch_bounds__(256) void lpips_conv1_dgrad_mfma_kernel(const float* __restrict__ g1, const float* __restrict__ wd,
                                                                    float* __restrict__ dxs, int H, int W, int h1, int w1, int tiles_x,
                                                                    int tiles_y, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float d1_smem[];
  float* Bt = d1_smem;
  float* As = d1_smem + kD1B;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
    const int64_t img = tile / (tiles_x * tiles_y);
    const int by0 = ty * kD1TH, bx0 = tx * kD1TW;
    __syncthreads();                                   // (the previous tile's reads)
    for (int base = 0; base < kD1Rows * kD1Cols * 32; base += 256 * 9) {      // (nine loads in flight per thread, then their stores)
      float2 tv[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int idx = base + threadIdx.x + 256 * k;
        const int pos = idx >> 5, h2 = idx & 31;
        const int oy = by0 - 2 + pos / kD1Cols, ox = bx0 - 2 + pos % kD1Cols;
        const bool in = idx < kD1Rows * kD1Cols * 32 && (unsigned)oy < (unsigned)h1 && (unsigned)ox < (unsigned)w1;
        const float2 v = *reinterpret_cast<const float2*>(g1 + (in ? ((img * h1 + oy) * (int64_t)w1 + ox) * 64 + 2 * h2 : 0));
        tv[k] = in ? v : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int idx = base + threadIdx.x + 256 * k;
        if (idx < kD1Rows * kD1Cols * 32) *reinterpret_cast<float2*>(Bt + (idx >> 5) * kD1Pix + 2 * (idx & 31)) = tv[k];
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i)
      *reinterpret_cast<f4*>(As + 4 * (threadIdx.x + 256 * i)) = *reinterpret_cast<const f4*>(wd + 4 * (threadIdx.x + 256 * i));
    __syncthreads();
    f4 acc[2][3];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) acc[jn][mb] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int rp = 0; rp < 4; ++rp) {
      f4 nxt[12];
      if (rp < 3) {
#pragma unroll
        for (int i = 0; i < 12; ++i) nxt[i] = *reinterpret_cast<const f4*>(wd + (rp + 1) * kD1A + 4 * (threadIdx.x + 256 * i));
      }
      const float* ab = As + (rp & 1) * kD1A + q * 48 + l16;
      const float* bb = Bt + ((2 * wave + rp) * kD1Cols + l16) * kD1Pix + q;
#pragma unroll
      for (int s_ = 0; s_ < 64; ++s_) {
        const int boff = (s_ / 16) * kD1Pix + 4 * (s_ % 16);
        const float b0 = bb[boff], b1 = bb[kD1Cols * kD1Pix + boff];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          const float av = ab[s_ * 192 + 16 * mb];
          acc[0][mb] = mfma16(av, b0, acc[0][mb]);
          acc[1][mb] = mfma16(av, b1, acc[1][mb]);
        }
      }
      if (rp < 3) {
        float* an = As + ((rp + 1) & 1) * kD1A;
#pragma unroll
        for (int i = 0; i < 12; ++i) *reinterpret_cast<f4*>(an + 4 * (threadIdx.x + 256 * i)) = nxt[i];
      }
      __syncthreads();
    }
    // D[m = 16 mb + 4 q + r][block column l16]
    const int bx = bx0 + l16;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int by = by0 + 2 * wave + jn;
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) {
        const int m0 = 16 * mb + 4 * q, y = 4 * by + m0 / 12;
        if (y >= H) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = (m0 + r) % 12, x = 4 * bx + mm / 3;
          if (x < W) dxs[((img * H + y) * (int64_t)W + x) * 3 + mm % 3] = acc[jn][mb][r];
        }
      }
    }
  }
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
      launch_pack16(packed + pl.w[l], packed + pl.w16[l], s.kh * s.kw, s.cin, s.cout, st);
      launch_pack16(packed + pl.wt[l], packed + pl.wt16[l], s.kh * s.kw, s.cout, s.cin, st);
    }
    hipLaunchKernelGGL(lpips_copy_kernel, dim3((s.cout + 255) / 256), dim3(256), 0, st, tensors_host[10 + l], packed + pl.lin[l],
                       (int64_t)s.cout);
  }
  hipLaunchKernelGGL(lpips_copy_kernel, dim3((64 * 363 + 255) / 256), dim3(256), 0, st, tensors_host[0], packed + pl.w1raw,
                     (int64_t)64 * 363);
  hipLaunchKernelGGL(lpips_conv1_dgrad_pack_kernel, dim3(4 * kD1A / 256), dim3(256), 0, st, tensors_host[0], packed + pl.wd1);
  hipLaunchKernelGGL(lpips_copy_kernel, dim3(1), dim3(64), 0, st, tensors_host[15], packed + pl.shift, (int64_t)3);
  hipLaunchKernelGGL(lpips_copy_kernel, dim3(1), dim3(64), 0, st, tensors_host[16], packed + pl.scale, (int64_t)3);
  return (int)hipGetLastError();
}

static int lpips_forward_impl(const float* packed, const float* in0, const float* in1, int from01, float* work, float* out,
                              int height, int width, int64_t batch, bool split, s2l_stream_t stream) {
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
    a.w16 = split && l > 0 ? reinterpret_cast<const uint16_t*>(packed + pl.w16[l]) : nullptr;
    a.bias = packed + pl.b[l];
    a.out = work + wl.act[l];
    a.partial = work + wl.partial;
    if (l == 0) {
      const int tiles_x = (wl.a[0].w + kC1T - 1) / kC1T, tiles_y = (wl.a[0].h + kC1T - 1) / kC1T;
      const int64_t n_tiles = 2 * N * tiles_x * tiles_y;
      hipLaunchKernelGGL(lpips_conv1_kernel, dim3((unsigned)std::min<int64_t>(n_tiles, 512)), dim3(256), 0, st, cur, packed + pl.w1raw,
                         packed + pl.b[0], work + wl.act[0], height, width, wl.a[0].h, wl.a[0].w, tiles_x, tiles_y, (int)n_tiles);
    } else if ((rc = launch_conv<false>(a, 2 * N, st))) {
      return rc;
    }
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

extern "C" int s2l_lpips_forward(const float* packed, const float* in0, const float* in1, int from01, float* work, float* out,
                                 int height, int width, int64_t batch, s2l_stream_t stream) {
  return lpips_forward_impl(packed, in0, in1, from01, work, out, height, width, batch, false, stream);
}
extern "C" int s2l_lpips_forward_split(const float* packed, const float* in0, const float* in1, int from01, float* work, float* out,
                                       int height, int width, int64_t batch, s2l_stream_t stream) {
  return lpips_forward_impl(packed, in0, in1, from01, work, out, height, width, batch, true, stream);
}

// `work` as s2l_lpips_forward left it; d_out [N] = d loss / d out[n]; d_in0 [N,H,W,3]
static int lpips_backward_impl(const float* packed, float* work, const float* d_out, int from01, int accumulate, float* d_in0,
                               int height, int width, int64_t batch, bool split, s2l_stream_t stream) {
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
    a.w16 = split ? reinterpret_cast<const uint16_t*>(packed + pl.wt16[l]) : nullptr;
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
    a.w16 = split ? reinterpret_cast<const uint16_t*>(packed + pl.wt16[l]) : nullptr;
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
  {
    int dev = 0, n_cu = 0;
    if ((rc = current_device_cus(&dev, &n_cu))) return rc;
    static LdsOptIn flags;
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(lpips_conv1_dgrad_mfma_kernel), kD1Lds, flags, dev))) return rc;
    const int tiles_x = ((width + 3) / 4 + kD1TW - 1) / kD1TW, tiles_y = ((height + 3) / 4 + kD1TH - 1) / kD1TH;
    const int64_t n_tiles = N * tiles_x * tiles_y;
    hipLaunchKernelGGL(lpips_conv1_dgrad_mfma_kernel, dim3((unsigned)std::min<int64_t>(n_tiles, n_cu)), dim3(256), kD1Lds, st, g,
                       packed + pl.wd1, dxs, height, width, wl.a[0].h, wl.a[0].w, tiles_x, tiles_y, (int)n_tiles);
  }
  hipLaunchKernelGGL(lpips_unscale_grad_kernel, dim3((unsigned)((npx * 3 + 255) / 256)), dim3(256), 0, st, dxs, packed + pl.scale, d_in0,
                     npx * 3, from01 ? 2.f : 1.f, accumulate);
  return (int)hipGetLastError();
}
extern "C" int s2l_lpips_backward(const float* packed, float* work, const float* d_out, int from01, int accumulate, float* d_in0,
                                  int height, int width, int64_t batch, s2l_stream_t stream) {
  return lpips_backward_impl(packed, work, d_out, from01, accumulate, d_in0, height, width, batch, false, stream);
}
extern "C" int s2l_lpips_backward_split(const float* packed, float* work, const float* d_out, int from01, int accumulate, float* d_in0,
                                        int height, int width, int64_t batch, s2l_stream_t stream) {
  return lpips_backward_impl(packed, work, d_out, from01, accumulate, d_in0, height, width, batch, true, stream);
}
