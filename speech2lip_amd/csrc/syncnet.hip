// T3: the "lipsync_expert" loss -- SyncNet_color (src/face_simple/models/syncnet.py:7-67, conv.py:5-19)
// in eval mode, cosine similarity + BCE (training.py:576-603), and the gradient of that loss with respect to the
// generated face window (SyncNet itself is frozen, training.py:85-90, so no weight gradients exist).
//
// Every convolution of both encoders -- kernels 1/3/5/7, strides (1|2|3) x (1|2|3), padding 0/1/3, 1..512 channels, with
// the eval-mode BatchNorm folded into the packed weights, optional residual add and ReLU -- is the implicit-GEMM kernel of
// csrc/conv_gemm.h.  The whole net is ~1.2 GMAC per window: launch- and weight-read-bound (65 MB of fp32 weights per pass).
#include "conv_gemm.h"

namespace s2l {

// syncnet.py:11-33
static const LayerSpec kFace[] = {
    {15, 32, 7, 7, 1, 1, 3, 3, 0},    {32, 64, 5, 5, 1, 2, 1, 1, 0},    {64, 64, 3, 3, 1, 1, 1, 1, 1},
    {64, 64, 3, 3, 1, 1, 1, 1, 1},    {64, 128, 3, 3, 2, 2, 1, 1, 0},   {128, 128, 3, 3, 1, 1, 1, 1, 1},
    {128, 128, 3, 3, 1, 1, 1, 1, 1},  {128, 128, 3, 3, 1, 1, 1, 1, 1},  {128, 256, 3, 3, 2, 2, 1, 1, 0},
    {256, 256, 3, 3, 1, 1, 1, 1, 1},  {256, 256, 3, 3, 1, 1, 1, 1, 1},  {256, 512, 3, 3, 2, 2, 1, 1, 0},
    {512, 512, 3, 3, 1, 1, 1, 1, 1},  {512, 512, 3, 3, 1, 1, 1, 1, 1},  {512, 512, 3, 3, 2, 2, 1, 1, 0},
    {512, 512, 3, 3, 1, 1, 0, 0, 0},  {512, 512, 1, 1, 1, 1, 0, 0, 0}};
// syncnet.py:35-54
static const LayerSpec kAudio[] = {
    {1, 32, 3, 3, 1, 1, 1, 1, 0},     {32, 32, 3, 3, 1, 1, 1, 1, 1},    {32, 32, 3, 3, 1, 1, 1, 1, 1},
    {32, 64, 3, 3, 3, 1, 1, 1, 0},    {64, 64, 3, 3, 1, 1, 1, 1, 1},    {64, 64, 3, 3, 1, 1, 1, 1, 1},
    {64, 128, 3, 3, 3, 3, 1, 1, 0},   {128, 128, 3, 3, 1, 1, 1, 1, 1},  {128, 128, 3, 3, 1, 1, 1, 1, 1},
    {128, 256, 3, 3, 3, 2, 1, 1, 0},  {256, 256, 3, 3, 1, 1, 1, 1, 1},  {256, 256, 3, 3, 1, 1, 1, 1, 1},
    {256, 512, 3, 3, 1, 1, 0, 0, 0},  {512, 512, 1, 1, 1, 1, 0, 0, 0}};
constexpr int kNumFace = 17, kNumAudio = 14, kNumLayers = kNumFace + kNumAudio, kSyncEmb = 512;
constexpr int kFaceH = 48, kFaceW = 96, kMelH = 80, kMelW = 16;

inline const LayerSpec& spec_of(int l) { return l < kNumFace ? kFace[l] : kAudio[l - kNumFace]; }

// packed blob: per layer {forward weights [kh*kw*ceil16(cin)][ceil64(cout)], bias [ceil64(cout)]}, then for the face
// encoder the dgrad weights [kh*kw*ceil16(cout)][ceil64(cin)].
struct PackedLayout {
  int64_t w[kNumLayers], b[kNumLayers], wt[kNumFace], w16[kNumLayers], wt16[kNumFace], total;      // w16 / wt16: the split form's operands
};
inline PackedLayout packed_layout() {
  PackedLayout p;
  int64_t o = 0;
  for (int l = 0; l < kNumLayers; ++l) {
    const LayerSpec& s = spec_of(l);
    p.w[l] = o;
    o += (int64_t)s.kh * s.kw * ceil_to(s.cin, 16) * ceil_to(s.cout, 64);
    p.b[l] = o;
    o += ceil_to(s.cout, 64);
  }
  for (int l = 0; l < kNumFace; ++l) {
    const LayerSpec& s = kFace[l];
    p.wt[l] = o;
    o += (int64_t)s.kh * s.kw * ceil_to(s.cout, 16) * ceil_to(s.cin, 64);
  }
  for (int l = 0; l < kNumLayers; ++l) {
    const LayerSpec& s = spec_of(l);
    p.w16[l] = o;
    o += packed16_floats(s.kh * s.kw, s.cin, s.cout);
    if (l < kNumFace) {
      p.wt16[l] = o;
      o += packed16_floats(s.kh * s.kw, s.cout, s.cin);
    }
  }
  p.total = o;
  return p;
}

// ---- embeddings, loss ---------------------------------------------------------------------------------------------------
// F.normalize(x, p=2, dim=1): x / max(||x||, 1e-12)   (syncnet.py:63-64).  One wave per row.
__global__ __launch_bounds__(64) void normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int dim) {
  const int b = blockIdx.x, l = threadIdx.x;
  float s = 0.f;
  for (int i = l; i < dim; i += 64) s = fmaf(x[(int64_t)b * dim + i], x[(int64_t)b * dim + i], s);
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int i = l; i < dim; i += 64) y[(int64_t)b * dim + i] = x[(int64_t)b * dim + i] * inv;
}

// cosine_loss (training.py:576-579): d = cosine_similarity(a, v) (eps 1e-8), loss = BCELoss(d, y) (mean over the batch, log
// clamped at -100), times `weight`; optional gradient with respect to v.  ATen: d = a.v / sqrt(max(|a|^2 |v|^2, eps^2));
// binary_cross_entropy_backward = (d - y) / max((1 - d) d, 1e-12).  One wave per row; losses[b] summed by the host kernel below.
__global__ __launch_bounds__(64) void cosine_bce_kernel(const float* __restrict__ a, const float* __restrict__ v,
                                                        const float* __restrict__ y, float scale, float* __restrict__ losses,
                                                        float* __restrict__ dv, int dim) {
  const int b = blockIdx.x, l = threadIdx.x;
  const float* ab = a + (int64_t)b * dim;
  const float* vb = v + (int64_t)b * dim;
  float saa = 0.f, svv = 0.f, sav = 0.f;
  for (int i = l; i < dim; i += 64) {
    saa = fmaf(ab[i], ab[i], saa);
    svv = fmaf(vb[i], vb[i], svv);
    sav = fmaf(ab[i], vb[i], sav);
  }
  for (int o = 32; o; o >>= 1) {
    saa += __shfl_xor(saa, o);
    svv += __shfl_xor(svv, o);
    sav += __shfl_xor(sav, o);
  }
  const float den2 = fmaxf(saa * svv, 1e-16f);
  const float inv_den = 1.f / sqrtf(den2);
  const float d = sav * inv_den;
  const float yy = y[b];
  // nn.BCELoss rejects inputs outside [0,1] (ATen: "all elements of input should be between 0 and 1" on the CPU, a device-side
  // assert on CUDA): a negative cosine -- embeddings that are not post-ReLU, a corrupted checkpoint -- must not train silently
  // against a clamped target.  A kernel cannot raise, so the loss AND its gradient become NaN, which the trainer's NaN scan
  // (check_weights, training.py:560) turns into the same hard stop.  (fmaxf would drop the NaN of logf(d<0): test first.)
  const bool in_range = d >= 0.f && d <= 1.f;
  const float nan = __builtin_nanf("");
  const float loss = in_range ? -(yy * fmaxf(logf(d), -100.f) + (1.f - yy) * fmaxf(logf(1.f - d), -100.f)) : nan;
  if (l == 0) losses[b] = loss * scale;
  if (dv) {
    const float gd = in_range ? (d - yy) / fmaxf((1.f - d) * d, 1e-12f) * scale : nan;
    // dd/dv = a / den - d * v / |v|^2   (the eps clamp is inactive for unit vectors)
    const float c1 = gd * inv_den, c2 = gd * d / fmaxf(svv, 1e-30f);
    for (int i = l; i < dim; i += 64) dv[(int64_t)b * dim + i] = c1 * ab[i] - c2 * vb[i];
  }
}

__global__ void sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int accumulate) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += x[i];
    out[0] = accumulate ? out[0] + s : s;
  }
}

// backward of F.normalize followed by the ReLU of the last layer: g_top = ((dvn - vn (vn . dvn)) / max(|v|, 1e-12)) * (v > 0)
__global__ __launch_bounds__(64) void normalize_bwd_kernel(const float* __restrict__ vraw, const float* __restrict__ dvn,
                                                           float* __restrict__ g, int dim) {
  const int b = blockIdx.x, l = threadIdx.x;
  const float* vb = vraw + (int64_t)b * dim;
  const float* db = dvn + (int64_t)b * dim;
  float svv = 0.f, svd = 0.f;
  for (int i = l; i < dim; i += 64) {
    svv = fmaf(vb[i], vb[i], svv);
    svd = fmaf(vb[i], db[i], svd);
  }
  for (int o = 32; o; o >>= 1) {
    svv += __shfl_xor(svv, o);
    svd += __shfl_xor(svd, o);
  }
  const float inv = 1.f / fmaxf(sqrtf(svv), 1e-12f);
  const float dot = svd * inv * inv;  // (vn . dvn) / |v| with vn = v * inv
  for (int i = l; i < dim; i += 64) {
    const float v = vb[i];
    g[(int64_t)b * dim + i] = v > 0.f ? (db[i] - v * dot) * inv : 0.f;
  }
}

// ---- window assembly (training.py:588-590): g [B,3,T,H,W] RGB -> face [B,H-H/2,W,3T], channel 3t+c = BGR channel c of
// frame t, rows H/2..H-1; and its adjoint (the upper rows of the gradient are zero).
__global__ __launch_bounds__(256) void sync_window_kernel(const float* __restrict__ g, float* __restrict__ face, int T, int H,
                                                          int W, int64_t n, int adjoint, float* __restrict__ dg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int top = H / 2, hh = H - top;
  if (!adjoint) {  // i indexes face [B,hh,W,3T]
    const int ch = (int)(i % (3 * T));
    int64_t r = i / (3 * T);
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % hh);
    const int b = (int)(r / hh);
    const int t = ch / 3, c = ch % 3;
    face[i] = g[((((int64_t)b * 3 + (2 - c)) * T + t) * H + (top + y)) * W + x];
  } else {  // i indexes dg [B,3,T,H,W]
    const int x = (int)(i % W);
    int64_t r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int t = (int)(r % T);
    r /= T;
    const int c_rgb = (int)(r % 3);
    const int b = (int)(r / 3);
    const int y2 = y - top;
    dg[i] = y2 >= 0 ? face[(((int64_t)b * hh + y2) * W + x) * (3 * T) + 3 * t + (2 - c_rgb)] : 0.f;
  }
}

// ---- host-side plan -------------------------------------------------------------------------------------------------------
// work buffer: face activations a_0..a_16, audio activations a_0..a_13, two gradient ping-pong buffers, split-K partials
struct WorkLayout {
  int64_t act[kNumLayers];
  Shape in_shape[kNumLayers], out_shape_[kNumLayers];
  int64_t grad[2], partial, total;
};
inline WorkLayout work_layout(int64_t B) {
  WorkLayout wl;
  int64_t o = 0, max_act = (int64_t)kFaceH * kFaceW * 16;
  Shape sh{kFaceH, kFaceW};
  for (int l = 0; l < kNumLayers; ++l) {
    if (l == kNumFace) sh = Shape{kMelH, kMelW};
    const LayerSpec& s = spec_of(l);
    wl.in_shape[l] = sh;
    sh = out_shape(s, sh);
    wl.out_shape_[l] = sh;
    wl.act[l] = o;
    const int64_t n = B * sh.h * sh.w * s.cout;
    o += (n + 3) / 4 * 4;
    if (l < kNumFace && n / B > max_act) max_act = n / B;
  }
  for (int k = 0; k < 2; ++k) {
    wl.grad[k] = o;
    o += (B * max_act + 3) / 4 * 4;
  }
  wl.partial = o;
  o += kPartialFloats;
  wl.total = o;
  return wl;
}

}  // namespace s2l

using namespace s2l;

extern "C" int64_t s2l_syncnet_packed_floats(void) { return packed_layout().total; }
extern "C" int64_t s2l_syncnet_work_floats(int64_t batch) { return batch < 1 ? 0 : work_layout(batch).total; }

extern "C" int s2l_syncnet_pack(const float* const* tensors_host, float bn_eps, float* packed, s2l_stream_t stream) {
  if (!tensors_host || !packed) return S2L_E_NULL;
  for (int i = 0; i < kNumLayers * 6; ++i)
    if (!tensors_host[i]) return S2L_E_NULL;
  if (misaligned16(packed)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PackedLayout pl = packed_layout();
  for (int l = 0; l < kNumLayers; ++l) {
    const LayerSpec& s = spec_of(l);
    const float* const* t = tensors_host + 6 * l;  // conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var
    const int RP = ceil_to(s.cout, 64), kcp = ceil_to(s.cin, 16);
    const int64_t n = (int64_t)s.kh * s.kw * kcp * RP;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t[0], t[2], t[5], bn_eps,
                       packed + pl.w[l], s.cin, s.cout, s.kh, s.kw, kcp, RP, 0, n);
    hipLaunchKernelGGL(conv_pack_bias_kernel, dim3((RP + 255) / 256), dim3(256), 0, st, t[1], t[2], t[3], t[4], t[5], bn_eps,
                       packed + pl.b[l], s.cout, RP);
    if (l < kNumFace) {
      const int RPt = ceil_to(s.cin, 64), kcpt = ceil_to(s.cout, 16);
      const int64_t nt = (int64_t)s.kh * s.kw * kcpt * RPt;
      hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, t[0], t[2], t[5], bn_eps,
                         packed + pl.wt[l], s.cin, s.cout, s.kh, s.kw, kcpt, RPt, 1, nt);
      launch_pack16(packed + pl.wt[l], packed + pl.wt16[l], s.kh * s.kw, s.cout, s.cin, st);
    }
    launch_pack16(packed + pl.w[l], packed + pl.w16[l], s.kh * s.kw, s.cin, s.cout, st);
  }
  return (int)hipGetLastError();
}

// audio_batch mel windows and face_batch >= audio_batch face windows in ONE pass per encoder (the contrastive loss embeds the generated
// and the negative windows of the same audio, training.py:592-601: twice the columns per weight read for the face encoder, the
// audio encoder once instead of twice).  The activations are laid out for face_batch.
static int syncnet_forward_impl(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                                float* face_emb, int64_t audio_batch, int64_t face_batch, bool split, s2l_stream_t stream) {
  if (face_batch < 0 || face_batch > 4096 || audio_batch < 0 || audio_batch > face_batch) return S2L_E_SIZE;
  if (face_batch == 0) return S2L_OK;
  if (!packed || !face || !work || !face_emb || (audio_batch && (!mel || !audio_emb))) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(work) || misaligned16(face) || misaligned16(mel)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PackedLayout pl = packed_layout();
  const WorkLayout wl = work_layout(face_batch);
  for (int l = 0; l < kNumLayers; ++l) {
    const int64_t batch = l < kNumFace ? face_batch : audio_batch;
    if (batch == 0) continue;
    const LayerSpec& s = spec_of(l);
    ConvArgs a = base_args(s, wl.in_shape[l], wl.out_shape_[l]);
    a.in = l == 0 ? face : l == kNumFace ? mel : work + wl.act[l - 1];
    a.w = packed + pl.w[l];
    a.w16 = split ? reinterpret_cast<const uint16_t*>(packed + pl.w16[l]) : nullptr;
    a.bias = packed + pl.b[l];
    a.res = s.res ? a.in : nullptr;
    a.out = work + wl.act[l];
    a.partial = work + wl.partial;
    const int rc = launch_conv<false>(a, batch, st);
    if (rc) return rc;
  }
  // both encoders end 1x1x512 (syncnet.py:59-60 flattens them): normalise
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)face_batch), dim3(64), 0, st, work + wl.act[kNumFace - 1], face_emb, kSyncEmb);
  if (audio_batch)
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)audio_batch), dim3(64), 0, st, work + wl.act[kNumLayers - 1], audio_emb,
                       kSyncEmb);
  return (int)hipGetLastError();
}
extern "C" int s2l_syncnet_forward(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                                   float* face_emb, int64_t batch, s2l_stream_t stream) {
  if (batch > 0 && (!mel || !audio_emb)) return S2L_E_NULL;
  return syncnet_forward_impl(packed, mel, face, work, audio_emb, face_emb, batch, batch, false, stream);
}
extern "C" int s2l_syncnet_forward_pair(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                                        float* face_emb, int64_t audio_batch, int64_t face_batch, s2l_stream_t stream) {
  return syncnet_forward_impl(packed, mel, face, work, audio_emb, face_emb, audio_batch, face_batch, false, stream);
}
extern "C" int s2l_syncnet_forward_pair_split(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                                              float* face_emb, int64_t audio_batch, int64_t face_batch, s2l_stream_t stream) {
  return syncnet_forward_impl(packed, mel, face, work, audio_emb, face_emb, audio_batch, face_batch, true, stream);
}

extern "C" int s2l_sync_loss(const float* audio_emb, const float* face_emb, const float* y, float weight, float* scratch,
                             float* loss, int accumulate, float* d_face_emb, int64_t batch, s2l_stream_t stream) {
  if (batch < 0 || batch > 4096) return S2L_E_SIZE;
  if (batch == 0) return S2L_OK;
  if (!audio_emb || !face_emb || !y || !scratch || !loss) return S2L_E_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cosine_bce_kernel, dim3((unsigned)batch), dim3(64), 0, st, audio_emb, face_emb, y, weight / (float)batch,
                     scratch, d_face_emb, kSyncEmb);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(64), 0, st, scratch, loss, (int)batch, accumulate);
  return (int)hipGetLastError();
}

// the data gradient for the FIRST `batch` windows of a forward over work_batch windows (NHWC, batch outermost: a prefix of every
// activation)
static int syncnet_face_backward_impl(const float* packed, const float* face, float* work, const float* d_face_emb, float* d_face,
                                      int64_t batch, int64_t work_batch, bool split, s2l_stream_t stream) {
  if (batch < 0 || work_batch > 4096 || batch > work_batch) return S2L_E_SIZE;
  if (batch == 0) return S2L_OK;
  if (!packed || !face || !work || !d_face_emb || !d_face) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(work)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PackedLayout pl = packed_layout();
  const WorkLayout wl = work_layout(work_batch);
  float* g_cur = work + wl.grad[0];
  float* g_nxt = work + wl.grad[1];
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3((unsigned)batch), dim3(64), 0, st, work + wl.act[kNumFace - 1], d_face_emb, g_cur,
                     kSyncEmb);
  for (int l = kNumFace - 1; l >= 0; --l) {
    const LayerSpec& s = kFace[l];
    ConvArgs a = base_args(s, wl.in_shape[l], wl.out_shape_[l]);
    a.in = g_cur;
    a.w = packed + pl.wt[l];
    a.w16 = split ? reinterpret_cast<const uint16_t*>(packed + pl.wt16[l]) : nullptr;
    a.res = s.res ? g_cur : nullptr;
    a.mask = l > 0 ? work + wl.act[l - 1] : nullptr;
    a.out = l > 0 ? g_nxt : d_face;
    a.partial = work + wl.partial;
    const int rc = launch_conv<true>(a, batch, st);
    if (rc) return rc;
    float* tmp = g_cur;
    g_cur = g_nxt;
    g_nxt = tmp;
  }
  return (int)hipGetLastError();
}
extern "C" int s2l_syncnet_face_backward(const float* packed, const float* face, float* work, const float* d_face_emb,
                                         float* d_face, int64_t batch, s2l_stream_t stream) {
  return syncnet_face_backward_impl(packed, face, work, d_face_emb, d_face, batch, batch, false, stream);
}
extern "C" int s2l_syncnet_face_backward_prefix(const float* packed, const float* face, float* work, const float* d_face_emb,
                                                float* d_face, int64_t batch, int64_t work_batch, s2l_stream_t stream) {
  return syncnet_face_backward_impl(packed, face, work, d_face_emb, d_face, batch, work_batch, false, stream);
}
extern "C" int s2l_syncnet_face_backward_prefix_split(const float* packed, const float* face, float* work, const float* d_face_emb,
                                                      float* d_face, int64_t batch, int64_t work_batch, s2l_stream_t stream) {
  return syncnet_face_backward_impl(packed, face, work, d_face_emb, d_face, batch, work_batch, true, stream);
}

extern "C" int s2l_sync_window(const float* g_rgb, float* face, int n_frames_t, int height, int width, int64_t batch,
                               s2l_stream_t stream) {
  if (batch < 0 || n_frames_t < 1 || height < 2 || width < 1) return S2L_E_SIZE;
  if (batch == 0) return S2L_OK;
  if (!g_rgb || !face) return S2L_E_NULL;
  const int64_t n = batch * (height - height / 2) * width * 3 * n_frames_t;
  hipLaunchKernelGGL(sync_window_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g_rgb,
                     face, n_frames_t, height, width, n, 0, (float*)nullptr);
  return (int)hipGetLastError();
}

extern "C" int s2l_sync_window_backward(const float* d_face, float* d_g_rgb, int n_frames_t, int height, int width, int64_t batch,
                                        s2l_stream_t stream) {
  if (batch < 0 || n_frames_t < 1 || height < 2 || width < 1) return S2L_E_SIZE;
  if (batch == 0) return S2L_OK;
  if (!d_face || !d_g_rgb) return S2L_E_NULL;
  const int64_t n = batch * 3 * n_frames_t * height * width;
  hipLaunchKernelGGL(sync_window_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)nullptr, const_cast<float*>(d_face), n_frames_t, height, width, n, 1, d_g_rgb);
  return (int)hipGetLastError();
}
