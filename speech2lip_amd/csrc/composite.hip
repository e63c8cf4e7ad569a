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
  const float* bgm;    // optional per-clip table [FH,FW,4] = ((1-mask)*face, face>0 bits) from s2l_composite_tables, or null
  const float* hole1;  // training-time black-hole augmentation (tf_nerf.py:371-384): the two N(0,1) fields [F,FH,FW] that
  const float* hole2;  //   add_black_hole draws; both null = inference branch
  int64_t face_stride, mask_stride, total;
  int h, w, FH, FW;
  int ox, oy;          // paste origin of the lip in the face frame
  int ry0, ry1, rx0, rx1;  // expanded-mask rectangle [ry0,ry1) x [rx0,rx1); ry0 < 0 => use `mask`
  int rsize, chunks;       // pixels per XCD region, 256-pixel chunks per region
};

struct Px {
  float c[3];
};

// 12-byte pixel accessed as one dwordx3 (pixels are only 4-byte aligned: offset = 12 * index)
struct __attribute__((packed, aligned(4))) F3 {
  float x, y, z;
};
__device__ __forceinline__ Px load_px(const float* p) {
  const F3 v = *reinterpret_cast<const F3*>(p);
  return Px{{v.x, v.y, v.z}};
}
__device__ __forceinline__ void store_px(float* p, const float (&c)[3]) {
  *reinterpret_cast<F3*>(p) = F3{c[0], c[1], c[2]};
}
// streaming (touched once) variants: keep the per-clip tables in L2, not the frame streams
__device__ __forceinline__ Px load_px_stream(const float* p) {
  return Px{{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2)}};
}
__device__ __forceinline__ void store_px_stream(float* p, const float (&c)[3]) {
  __builtin_nontemporal_store(c[0], p);
  __builtin_nontemporal_store(c[1], p + 1);
  __builtin_nontemporal_store(c[2], p + 2);
}

// merged canonical image at in-range integer (yy, xx).  Separate roundings on purpose: the
// reference evaluates mul, rsub, mul, add as four ATen ops (tf_nerf.py:352), so no FMA here.
__device__ __forceinline__ Px blend_px(const Px& m, const Px& l, const Px& fv) {
  Px r;
#pragma unroll
  for (int c = 0; c < 3; ++c) r.c[c] = __fadd_rn(__fmul_rn(m.c[c], l.c[c]), __fmul_rn(__fsub_rn(1.f, m.c[c]), fv.c[c]));
  return r;
}

// Black-hole augmentation (tf_nerf.py:306-318, 371-384).  mask_face_observed = grid_sample(face_canon > 0) == 1 per channel
// (the same taps and weights as the image; a 0/1 image makes every product exact, so the fma chain below is the plain
// left-to-right sum ATen forms); a hole is punched where the pixel's N(0,1) draw is < 1e-6 inside that mask.  With
// keep_k = !(inside && draw_k < 1e-6):   merged' = keep_1 ? merged : gt,   gt' = keep_2 ? gt : merged,
// out = M ? merged' : gt'  =>  the warped image shows where (M && keep_1) || (!M && !keep_2).
struct HoleSel {
  bool hole1[3], hole2[3];
};
__device__ __forceinline__ HoleSel hole_select(const CompArgs& a, int64_t idx, const float (&wgt)[4], const int (&fbits)[4]) {
  const bool h1 = __builtin_nontemporal_load(a.hole1 + idx) < 0.000001f;
  const bool h2 = __builtin_nontemporal_load(a.hole2 + idx) < 0.000001f;
  HoleSel hs;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float fo = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) fo = fmaf((fbits[t] >> c) & 1 ? 1.f : 0.f, wgt[t], fo);
    const bool inside = fo == 1.f;
    hs.hole1[c] = inside && h1;
    hs.hole2[c] = inside && h2;
  }
  return hs;
}
__device__ __forceinline__ bool warped_shows(bool M, const HoleSel& hs, int c) { return M ? !hs.hole1[c] : hs.hole2[c]; }

// One thread per output pixel; grid.y = frame.  All eight table gathers (mask + face at the four
// bilinear taps) are issued up front with clamped addresses and zeroed weights for out-of-range
// taps -- no per-tap branches, so the loads overlap instead of serialising on L2 latency.  The
// lip is fetched only by waves that touch its box (wave-uniform test).
constexpr int kPPT = 1;   // pixels per thread (256 apart): the second pixel's streaming loads overlap the first's gathers

// value of one output pixel (res); `idx` is only read by the black-hole branch
// PLAIN: the caller guarantees the expanded-rectangle mask and no black holes (the inference fast path): those branches and
// the registers they keep alive are compiled out.
template <bool BGM, bool PLAIN = false>
__device__ __forceinline__ void composite_value(const CompArgs& a, const float* face, const float* mask, const float* lip,
                                                int64_t idx, float2 g, const Px& gt, float (&res)[3]) {
  const bool rect = PLAIN || a.ry0 >= 0;
  // grid_sample(align_corners=False): unnormalise as (x+1)*(size/2) - 0.5, bilinear weights from
  // the distances to the four neighbours, zero padding outside [0,size-1].
  const float ix = __fsub_rn(__fmul_rn(__fadd_rn(g.x, 1.f), 0.5f * (float)a.FW), 0.5f);
  const float iy = __fsub_rn(__fmul_rn(__fadd_rn(g.y, 1.f), 0.5f * (float)a.FH), 0.5f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float wx = ix - xw, ex = 1.f - wx, ny = iy - yn, sy = 1.f - ny;
  const float wraw[4] = {__fmul_rn(sy, ex), __fmul_rn(sy, wx), __fmul_rn(ny, ex), __fmul_rn(ny, wx)};
  // integer tap origin; the float clamp keeps the conversion defined for wild / non-finite coords
  const int x0 = (int)fminf(fmaxf(xw, -2.f), (float)a.FW + 1.f);
  const int y0 = (int)fminf(fmaxf(yn, -2.f), (float)a.FH + 1.f);

  Px m[4], fv[4], l[4];
  float wgt[4];
  int lipoff[4], moff[4], fbits[4];
  bool inlip[4];
  bool anylip = false;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    // bitwise & on purpose: short-circuit && turns into per-tap branches that keep the four gathers from being issued together
    const bool ok = ((unsigned)xx < (unsigned)a.FW) & ((unsigned)yy < (unsigned)a.FH) & (xw + (float)(t & 1) == (float)xx) &
                    (yn + (float)(t >> 1) == (float)yy);
    wgt[t] = ok ? wraw[t] : 0.f;
    const int xc = min(max(xx, 0), a.FW - 1), yc = min(max(yy, 0), a.FH - 1);
    const int o = (yc * a.FW + xc) * 3;
    const int ly = yc - a.oy, lx = xc - a.ox;
    inlip[t] = ok & ((unsigned)ly < (unsigned)a.h) & ((unsigned)lx < (unsigned)a.w);
    lipoff[t] = (ly * a.w + lx) * 3;
    anylip |= inlip[t];
    l[t] = Px{{0.f, 0.f, 0.f}};
    if (BGM) {
      const f4 b = *reinterpret_cast<const f4*>(a.bgm + (yc * a.FW + xc) * 4);   // one aligned 16-byte gather
      fv[t] = Px{{b[0], b[1], b[2]}};           // (1-mask)*face, already rounded as the reference rounds it
      fbits[t] = PLAIN ? 0 : (int)b[3];         // bit c: face_canon[c] > 0 (only the black-hole augmentation reads it)
      moff[t] = o;
      m[t] = Px{{0.f, 0.f, 0.f}};
    } else {
      m[t] = load_px(mask + o);
      fv[t] = load_px(face + o);
      fbits[t] = (fv[t].c[0] > 0.f ? 1 : 0) | (fv[t].c[1] > 0.f ? 2 : 0) | (fv[t].c[2] > 0.f ? 4 : 0);
    }
  }
  if (__any(anylip || (BGM && !rect))) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (BGM && (inlip[t] || !rect)) m[t] = load_px(mask + moff[t]);   // mask only matters where the lip is (or as warp mask)
      if (inlip[t]) l[t] = load_px(lip + lipoff[t]);
    }
  }

  float acc[3] = {0.f, 0.f, 0.f}, macc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    Px v;
    if (BGM) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v.c[c] = __fadd_rn(__fmul_rn(m[t].c[c], l[t].c[c]), fv[t].c[c]);
    } else {
      v = blend_px(m[t], l[t], fv[t]);
    }
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const float mr = ((yy >= a.ry0) & (yy < a.ry1) & (xx >= a.rx0) & (xx < a.rx1)) ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[c] = fmaf(v.c[c], wgt[t], acc[c]);
      macc[c] = fmaf(rect ? mr : m[t].c[c], wgt[t], macc[c]);
    }
  }
  if (!PLAIN && a.hole1) {   // training branch (wave-uniform): black holes inside the warped canonical face
    const HoleSel hs = hole_select(a, idx, wgt, fbits);
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = warped_shows(macc[c] != 0.f, hs, c) ? acc[c] : gt.c[c];
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = macc[c] != 0.f ? acc[c] : gt.c[c];
  }
}

template <bool BGM>
__device__ __forceinline__ void composite_pixel(const CompArgs& a, const float* face, const float* mask, const float* lip,
                                                int64_t idx, int pix, float2 g, const Px& gt) {
  float res[3];
  composite_value<BGM>(a, face, mask, lip, idx, g, gt, res);
  store_px_stream(a.out_new + idx * 3, res);

  if (a.out_can) {
    const int y = pix / a.FW, x = pix - y * a.FW;
    const int o = pix * 3;
    const int ly = y - a.oy, lx = x - a.ox;
    Px lc = Px{{0.f, 0.f, 0.f}};
    if ((unsigned)ly < (unsigned)a.h && (unsigned)lx < (unsigned)a.w) lc = load_px(lip + (ly * a.w + lx) * 3);
    const Px v = blend_px(load_px(mask + o), lc, load_px(face + o));
    store_px(a.out_can + idx * 3, v.c);
  }
}

// One thread per kPPT output pixels.  All table gathers of a pixel (mask + face, or the fused
// per-clip table, at the four bilinear taps) are issued up front with clamped addresses and
// zeroed weights for out-of-range taps -- no per-tap branches, so the loads overlap instead of
// serialising on L2 latency.  The lip is fetched only by waves that touch its box.
template <bool BGM>
__global__ __launch_bounds__(256) void composite_kernel(CompArgs a) {
  // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed dispatch order; speed only, not
  // correctness).  XCD r owns the r-th eighth of the face for EVERY frame, so its slice of the
  // per-clip constants stays resident in its 4 MiB L2 across frames.
  const int per = a.FH * a.FW;
  const int region = blockIdx.x & 7;
  const int k = blockIdx.x >> 3;
  const int chunk = k % a.chunks;
  const int64_t f = k / a.chunks;
  const float* face = a.face + f * a.face_stride;
  const float* mask = a.mask + f * a.mask_stride;
  const float* lip = a.lip + f * (int64_t)a.h * a.w * 3;
  int pix[kPPT];
  bool live[kPPT];
  float2 g[kPPT];
  Px gt[kPPT];
#pragma unroll
  for (int i = 0; i < kPPT; ++i) {   // streaming loads of every pixel first
    const int in_region = (chunk * kPPT + i) * 256 + threadIdx.x;
    pix[i] = region * a.rsize + in_region;
    live[i] = in_region < a.rsize && pix[i] < per;
    const int64_t idx = f * per + (live[i] ? pix[i] : 0);
    g[i] = float2{__builtin_nontemporal_load(a.coord + 2 * idx), __builtin_nontemporal_load(a.coord + 2 * idx + 1)};
    gt[i] = load_px_stream(a.gt + idx * 3);
  }
#pragma unroll
  for (int i = 0; i < kPPT; ++i)
    if (live[i]) composite_pixel<BGM>(a, face, mask, lip, f * per + pix[i], pix[i], g[i], gt[i]);
}

// ---- the inference fast path: a wave per span of 256 consecutive pixels ------------------------------------------------------
// Requirements (checked by the launcher): per-clip constants with their fused table, expanded-rectangle mask, no black holes,
// no canonical output, FH*FW a multiple of 4, 16-byte-aligned streams.
// The warped image only shows where a bilinear tap falls inside the expanded rectangle; everywhere else out = rgb_gt.
//   lip_merge_kernel: per frame, the merged canonical image INSIDE the lip box as 16-byte pixels, mask*lip + bgm (0.26 MB per
//     128x128 lip).  With it every bilinear tap is ONE aligned 16-byte gather -- from this table inside the box, from the per-clip
//     bgm table outside -- instead of a table gather plus two unaligned 12-byte gathers (mask, lip).
//   span_kernel: coord = 2 KiB = two 1-KiB vector loads, rgb_gt = 3 KiB = three, out = three vector stores; lane l moves bytes
//     [16 l, 16 l + 16) of each KiB, so every byte of the frame streams is requested exactly once, perfectly coalesced (the
//     one-pixel kernel needs 8 scalar stream instructions per 64 pixels with 8- and 12-byte lane strides).  A span none of whose
//     pixels can touch the rectangle (64 % of the spans of a 500x500 frame with a 128x128 lip; 80 % of its 64-pixel quarters) is a
//     pure vector copy and the 12-byte pixels are never un-packed.  Otherwise the wave stages its 3 KiB of rgb_gt in LDS
//     (wave-private, no barrier) and walks the quarters that can touch the rectangle, one pixel per lane: coordinates by
//     cross-lane shuffle, four aligned gathers, the result written over the staged pixel; then the same three vector stores.
//     Results are bit-identical to composite_kernel's (same operations on the same values in the same order).
struct SpanArgs {
  CompArgs c;
  float* merged;          // [F][h][w][4]: the merged canonical image inside the lip box, mask*lip + (1-mask)*face (tf_nerf.py:352)
  int nspans;             // spans per frame = ceil(FH*FW / 256)
  int64_t n_cand;         // F * nspans
};

__global__ __launch_bounds__(256) void lip_merge_kernel(SpanArgs sa) {
  const CompArgs& a = sa.c;
  const int n = a.h * a.w;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int64_t f = blockIdx.y;
  if (p >= n) return;
  const int ly = p / a.w, lx = p - ly * a.w;
  // a lip pixel the (possibly cropped) paste puts outside the frame is never looked up: taps are clamped to the frame first
  if ((unsigned)(a.oy + ly) >= (unsigned)a.FH || (unsigned)(a.ox + lx) >= (unsigned)a.FW) return;
  const int o = (a.oy + ly) * a.FW + a.ox + lx;
  const Px m = load_px(a.mask + o * 3);
  const Px l = load_px(a.lip + (f * n + p) * 3);
  const f4 b = *reinterpret_cast<const f4*>(a.bgm + o * 4);
  f4 v;
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = __fadd_rn(__fmul_rn(m.c[c], l.c[c]), b[c]);   // the reference's roundings (tf_nerf.py:352)
  v[3] = 0.f;
  *reinterpret_cast<f4*>(sa.merged + (f * n + p) * 4) = v;
}

// composite_value<true, true> with the lip box read from the merged table
__device__ __forceinline__ void composite_value_merged(const CompArgs& a, const float* merged_f, float gx, float gy, const Px& gt,
                                                       float (&res)[3]) {
  const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f * (float)a.FW), 0.5f);
  const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f * (float)a.FH), 0.5f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float wx = ix - xw, ex = 1.f - wx, ny = iy - yn, sy = 1.f - ny;
  const float wraw[4] = {__fmul_rn(sy, ex), __fmul_rn(sy, wx), __fmul_rn(ny, ex), __fmul_rn(ny, wx)};
  const int x0 = (int)fminf(fmaxf(xw, -2.f), (float)a.FW + 1.f);
  const int y0 = (int)fminf(fmaxf(yn, -2.f), (float)a.FH + 1.f);
  f4 v[4];
  float wgt[4], mr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const bool ok = ((unsigned)xx < (unsigned)a.FW) & ((unsigned)yy < (unsigned)a.FH) & (xw + (float)(t & 1) == (float)xx) &
                    (yn + (float)(t >> 1) == (float)yy);
    wgt[t] = ok ? wraw[t] : 0.f;
    mr[t] = ((yy >= a.ry0) & (yy < a.ry1) & (xx >= a.rx0) & (xx < a.rx1)) ? 1.f : 0.f;
    const int xc = min(max(xx, 0), a.FW - 1), yc = min(max(yy, 0), a.FH - 1);
    const int ly = yc - a.oy, lx = xc - a.ox;
    const bool inlip = ((unsigned)ly < (unsigned)a.h) & ((unsigned)lx < (unsigned)a.w);
    const float* src = inlip ? merged_f + (ly * a.w + lx) * 4 : a.bgm + (yc * a.FW + xc) * 4;
    v[t] = *reinterpret_cast<const f4*>(src);
  }
  float acc[3] = {0.f, 0.f, 0.f}, macc = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = fmaf(v[t][c], wgt[t], acc[c]);
    macc = fmaf(mr[t], wgt[t], macc);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) res[c] = macc != 0.f ? acc[c] : gt.c[c];
}

__global__ __launch_bounds__(256) void span_kernel(SpanArgs sa) {
  __shared__ __attribute__((aligned(16))) float stage[4][768];
  const CompArgs& a = sa.c;
  const int per = a.FH * a.FW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 4 + wave;                 // span, frame-major
  if (c >= sa.n_cand) return;                                       // wave-uniform
  const int64_t f = c / sa.nspans;
  const int pix0 = (int)(c - f * sa.nspans) * 256;
  const int nvalid = min(256, per - pix0);                          // a multiple of 4
  const int64_t idx0 = f * per + pix0;
  const float* cbase = a.coord + 2 * idx0;
  const float* gbase = a.gt + 3 * idx0;
  f4 cv[2], gv[3];
#pragma unroll
  for (int s = 0; s < 2; ++s)       // lane holds the coords of pixels 128 s + 2 lane, + 1: quarter 2 s + (lane >> 5)
    cv[s] = 128 * s + 2 * lane < nvalid ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(cbase + 256 * s + 4 * lane))
                                        : (f4){-9.f, -9.f, -9.f, -9.f};
#pragma unroll
  for (int s = 0; s < 3; ++s)       // lane moves floats [256 s + 4 lane, + 4) of the span's 768: pixel-unaligned on purpose
    gv[s] = 256 * s + 4 * lane < 3 * nvalid ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(gbase + 256 * s + 4 * lane))
                                            : (f4){0.f, 0.f, 0.f, 0.f};
  // conservative rectangle test on the integer tap origins of the lane's four pixels
  bool hit[2] = {false, false};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float gx = cv[i >> 1][2 * (i & 1)], gy = cv[i >> 1][2 * (i & 1) + 1];
    const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f * (float)a.FW), 0.5f);
    const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f * (float)a.FH), 0.5f);
    const int x0 = (int)fminf(fmaxf(floorf(ix), -2.f), (float)a.FW + 1.f);
    const int y0 = (int)fminf(fmaxf(floorf(iy), -2.f), (float)a.FH + 1.f);
    hit[i >> 1] |= (128 * (i >> 1) + 2 * lane < nvalid) & (x0 + 1 >= a.rx0) & (x0 < a.rx1) & (y0 + 1 >= a.ry0) & (y0 < a.ry1);
  }
  const unsigned long long b0 = __ballot(hit[0]), b1 = __ballot(hit[1]);
  unsigned qmask = ((unsigned)b0 != 0 ? 1u : 0u) | ((b0 >> 32) != 0 ? 2u : 0u) | ((unsigned)b1 != 0 ? 4u : 0u) |
                   ((b1 >> 32) != 0 ? 8u : 0u);
#ifdef S2L_EXP_COPYONLY
  qmask = 0;
#endif
  if (qmask) {                      // wave-uniform
    float* sg = stage[wave];
#pragma unroll
    for (int s = 0; s < 3; ++s) *reinterpret_cast<f4*>(sg + 256 * s + 4 * lane) = gv[s];
    const float* merged_f = sa.merged + f * (int64_t)a.h * a.w * 4;
    while (qmask) {                 // the quarters that can touch the rectangle, one pixel per lane
      const int q = __builtin_ctz(qmask);
      qmask &= qmask - 1;
      // pixel 64 q + lane: its coordinates sit in lane 32 (q & 1) + lane / 2 of cv[q >> 1], components 2 (lane & 1), + 1
      const f4 cq = (q >> 1) ? cv[1] : cv[0];
      const int src = 32 * (q & 1) + (lane >> 1);
      const float e0 = __shfl(cq[0], src), e1 = __shfl(cq[1], src), e2 = __shfl(cq[2], src), e3 = __shfl(cq[3], src);
      const float gx = (lane & 1) ? e2 : e0, gy = (lane & 1) ? e3 : e1;
      const int p = 64 * q + lane;
      if (p < nvalid) {
        const Px gt = Px{{sg[3 * p], sg[3 * p + 1], sg[3 * p + 2]}};
        float res[3];
        composite_value_merged(a, merged_f, gx, gy, gt, res);
        sg[3 * p] = res[0];
        sg[3 * p + 1] = res[1];
        sg[3 * p + 2] = res[2];
      }
    }
    // every pixel is owned by exactly one lane and a wave's LDS operations complete in program order: the vector reads below
    // see all the per-pixel results without a barrier (other waves of the block never touch this wave's staging area)
#pragma unroll
    for (int s = 0; s < 3; ++s) gv[s] = *reinterpret_cast<const f4*>(sg + 256 * s + 4 * lane);
  }
  float* obase = a.out_new + 3 * idx0;
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (256 * s + 4 * lane < 3 * nvalid) __builtin_nontemporal_store(gv[s], reinterpret_cast<f4*>(obase + 256 * s + 4 * lane));
}

// Per-clip table for the fast path: bgm[p] = ((1-mask[p]) * face[p] (3 floats), bits of face[p] > 0): 16-byte pixels.
__global__ void composite_tables_kernel(const float* __restrict__ face, const float* __restrict__ mask,
                                        float* __restrict__ bgm, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Px m = load_px(mask + 3 * i), f = load_px(face + 3 * i);
  f4 bg;
#pragma unroll
  for (int c = 0; c < 3; ++c) bg[c] = __fmul_rn(__fsub_rn(1.f, m.c[c]), f.c[c]);
  bg[3] = (float)((f.c[0] > 0.f ? 1 : 0) | (f.c[1] > 0.f ? 2 : 0) | (f.c[2] > 0.f ? 4 : 0));   // for the black-hole mask
  *reinterpret_cast<f4*>(bgm + 4 * i) = bg;
}

// ---- gradient with respect to the lip (training: the autograd of tf_nerf.py:339-386 for rgb_lip_warped) -----------------
// out = sel ? sum_t wgt[t] * (mask[t] * lip[t] + (1-mask[t]) * face[t]) : gt   =>   d lip[tap t] += mask[t] * wgt[t] * d out
// for the taps inside the lip box, where sel is the forward's choice (expanded-rectangle / warped-mask test, black holes).
// One thread per output pixel; waves that touch no lip pixel leave after the coordinate load (most of the frame).  The
// scatter is a hardware float atomic (global_atomic_add_f32): the summation order, and so the last bit, is not fixed --
// as for ATen's grid_sample backward.
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs a, const float* __restrict__ d_new, float* __restrict__ d_lip) {
  const int per = a.FH * a.FW;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int64_t f = blockIdx.y;
  if (pix >= per) return;
  const int64_t idx = f * per + pix;
  const float2 g = *reinterpret_cast<const float2*>(a.coord + 2 * idx);
  const bool rect = a.ry0 >= 0;
  const float ix = __fsub_rn(__fmul_rn(__fadd_rn(g.x, 1.f), 0.5f * (float)a.FW), 0.5f);
  const float iy = __fsub_rn(__fmul_rn(__fadd_rn(g.y, 1.f), 0.5f * (float)a.FH), 0.5f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float wx = ix - xw, ex = 1.f - wx, ny = iy - yn, sy = 1.f - ny;
  const float wraw[4] = {__fmul_rn(sy, ex), __fmul_rn(sy, wx), __fmul_rn(ny, ex), __fmul_rn(ny, wx)};
  const int x0 = (int)fminf(fmaxf(xw, -2.f), (float)a.FW + 1.f);
  const int y0 = (int)fminf(fmaxf(yn, -2.f), (float)a.FH + 1.f);
  float wgt[4];
  int lipoff[4], toff[4];
  bool inlip[4], anylip = false;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const bool ok = (unsigned)xx < (unsigned)a.FW && (unsigned)yy < (unsigned)a.FH && xw + (float)(t & 1) == (float)xx &&
                    yn + (float)(t >> 1) == (float)yy;
    wgt[t] = ok ? wraw[t] : 0.f;
    const int xc = min(max(xx, 0), a.FW - 1), yc = min(max(yy, 0), a.FH - 1);
    toff[t] = yc * a.FW + xc;
    const int ly = yc - a.oy, lx = xc - a.ox;
    inlip[t] = ok && (unsigned)ly < (unsigned)a.h && (unsigned)lx < (unsigned)a.w;
    lipoff[t] = (ly * a.w + lx) * 3;
    anylip |= inlip[t];
  }
  if (!anylip) return;
  const float* face = a.face + f * a.face_stride;
  const float* mask = a.mask + f * a.mask_stride;
  Px m[4];
  int fbits[4];
  float macc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    m[t] = load_px(mask + toff[t] * 3);
    fbits[t] = 0;
    if (a.hole1) {
      const Px fv = load_px(face + toff[t] * 3);
      fbits[t] = (fv.c[0] > 0.f ? 1 : 0) | (fv.c[1] > 0.f ? 2 : 0) | (fv.c[2] > 0.f ? 4 : 0);
    }
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const float mr = (yy >= a.ry0 && yy < a.ry1 && xx >= a.rx0 && xx < a.rx1) ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) macc[c] = fmaf(rect ? mr : m[t].c[c], wgt[t], macc[c]);
  }
  const Px d = load_px(d_new + idx * 3);
  float dw[3];
  if (a.hole1) {
    const HoleSel hs = hole_select(a, idx, wgt, fbits);
#pragma unroll
    for (int c = 0; c < 3; ++c) dw[c] = warped_shows(macc[c] != 0.f, hs, c) ? d.c[c] : 0.f;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) dw[c] = macc[c] != 0.f ? d.c[c] : 0.f;
  }
  float* dl = d_lip + f * (int64_t)a.h * a.w * 3;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!inlip[t]) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = m[t].c[c] * wgt[t] * dw[c];
      if (v != 0.f) unsafeAtomicAdd(dl + lipoff[t] + c, v);
    }
  }
}

}  // namespace s2l

extern "C" int s2l_composite_tables(const float* face_canon, const float* mask, float* bgm, int face_h, int face_w,
                                    s2l_stream_t stream) {
  if (face_h <= 0 || face_w <= 0 || (int64_t)face_h * face_w * 6 > 0x7fffffff) return S2L_E_SIZE;
  if (!face_canon || !mask || !bgm) return S2L_E_NULL;
  const int n = face_h * face_w;
  hipLaunchKernelGGL(s2l::composite_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     face_canon, mask, bgm, n);
  return (int)hipGetLastError();
}

// shared argument checks + geometry of the forward and backward entry points
static int composite_args(s2l::CompArgs& a, const float* face_canon, int64_t face_stride, const float* mask, int64_t mask_stride,
                          const float* coord, const float* hole1, const float* hole2, int lip_h, int lip_w, int face_h,
                          int face_w, int x0, int y0, int pad_mode, int expand_pad, int64_t n_frames) {
  if (lip_h <= 0 || lip_w <= 0 || face_h <= 0 || face_w <= 0 || n_frames < 0) return S2L_E_SIZE;
  const int64_t per = (int64_t)face_h * face_w;
  if ((face_stride != 0 && face_stride != per * 3) || (mask_stride != 0 && mask_stride != per * 3)) return S2L_E_SIZE;
  if (pad_mode != S2L_PAD_MAY && pad_mode != S2L_PAD_DEFAULT) return S2L_E_SIZE;
  if ((hole1 == nullptr) != (hole2 == nullptr)) return S2L_E_NULL;
  if (n_frames == 0) return S2L_OK;
  if (!face_canon || !mask || !coord) return S2L_E_NULL;
  if (reinterpret_cast<uintptr_t>(coord) & 7) return S2L_E_ALIGN;
  a.face = face_canon; a.mask = mask; a.coord = coord; a.hole1 = hole1; a.hole2 = hole2;
  a.face_stride = face_stride; a.mask_stride = mask_stride; a.total = per * n_frames;
  a.h = lip_h; a.w = lip_w; a.FH = face_h; a.FW = face_w;
  a.ox = pad_mode == S2L_PAD_MAY ? x0 : x0 - 1;
  a.oy = pad_mode == S2L_PAD_MAY ? y0 : y0 - 1;
  // F.pad with a negative amount CROPS (tf_nerf.py:343-350): a lip box that leaves the face frame is pasted with its outside
  // part cut off -- every kernel tests taps against the box AND the frame, so any origin works (a box that only touches the
  // frame from outside is cropped to nothing: the lip contributes zeros).  A box BEYOND that would need a crop larger than the
  // lip: F.pad raises in the reference ("narrow(): length must be non-negative"): S2L_E_GEOMETRY here (golden G17).
  if (a.ox + lip_w < 0 || a.oy + lip_h < 0 || a.ox > face_w || a.oy > face_h) return S2L_E_GEOMETRY;
  if ((int64_t)a.ox + lip_w > 0x3fffffff || (int64_t)a.oy + lip_h > 0x3fffffff || a.ox < -0x3fffffff || a.oy < -0x3fffffff) return S2L_E_SIZE;
  if (expand_pad >= 0) {
    // the rectangle is a PYTHON SLICE (tf_nerf.py:362): a negative bound has the axis length added once, then both bounds are
    // clamped to [0, length]; start >= stop is an empty slice (no warped pixel shows) -- e.g. x0 < padding wraps the start to the
    // far side.  Reproduced as the reference behaves, not repaired.
    const auto bound = [](int64_t v, int len) { if (v < 0) v += len; return (int)(v < 0 ? 0 : v > len ? len : v); };
    a.ry0 = bound((int64_t)y0 - expand_pad, face_h); a.ry1 = bound((int64_t)y0 + lip_h + 2 * (int64_t)expand_pad, face_h);
    a.rx0 = bound((int64_t)x0 - expand_pad, face_w); a.rx1 = bound((int64_t)x0 + lip_w + expand_pad, face_w);
  } else {
    a.ry0 = a.ry1 = a.rx0 = a.rx1 = -1;
  }
  if (per * 6 > 0x7fffffff) return S2L_E_SIZE;   // 32-bit in-frame offsets
  return S2L_OK;
}

extern "C" int s2l_composite_train(const float* lip, const float* face_canon, int64_t face_stride, const float* mask,
                                   int64_t mask_stride, const float* rgb_gt, const float* coord, const float* hole1,
                                   const float* hole2, float* out_new, float* out_canonical, const float* bgm, int lip_h,
                                   int lip_w, int face_h, int face_w, int x0, int y0, int pad_mode, int expand_pad,
                                   int64_t n_frames, s2l_stream_t stream) {
  if (n_frames > 0 && (!lip || !rgb_gt || !out_new)) return S2L_E_NULL;
  s2l::CompArgs a;
  const int rc = composite_args(a, face_canon, face_stride, mask, mask_stride, coord, hole1, hole2, lip_h, lip_w, face_h, face_w,
                                x0, y0, pad_mode, expand_pad, n_frames);
  if (rc || n_frames == 0) return rc;
  if (s2l::misaligned16(bgm)) return S2L_E_ALIGN;
  a.lip = lip; a.gt = rgb_gt; a.out_new = out_new; a.out_can = out_canonical;
  a.bgm = (face_stride == 0 && mask_stride == 0) ? bgm : nullptr;   // the table is per clip
  const int64_t per = (int64_t)face_h * face_w;
  a.rsize = (int)((per + 7) / 8);
  a.chunks = (a.rsize + 256 * s2l::kPPT - 1) / (256 * s2l::kPPT);
  const int64_t blocks = 8 * (int64_t)a.chunks * n_frames;
  if (blocks > 0x7fffffff) return S2L_E_SIZE;
  if (a.bgm)
    hipLaunchKernelGGL(s2l::composite_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(s2l::composite_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

extern "C" int s2l_composite(const float* lip, const float* face_canon, int64_t face_stride, const float* mask,
                             int64_t mask_stride, const float* rgb_gt, const float* coord, float* out_new,
                             float* out_canonical, const float* bgm, int lip_h, int lip_w, int face_h, int face_w, int x0,
                             int y0, int pad_mode, int expand_pad, int64_t n_frames, s2l_stream_t stream) {
  return s2l_composite_train(lip, face_canon, face_stride, mask, mask_stride, rgb_gt, coord, nullptr, nullptr, out_new,
                             out_canonical, bgm, lip_h, lip_w, face_h, face_w, x0, y0, pad_mode, expand_pad, n_frames, stream);
}

extern "C" int s2l_composite_backward_lip(const float* d_new, const float* face_canon, int64_t face_stride, const float* mask,
                                          int64_t mask_stride, const float* coord, const float* hole1, const float* hole2,
                                          float* d_lip, int lip_h, int lip_w, int face_h, int face_w, int x0, int y0,
                                          int pad_mode, int expand_pad, int64_t n_frames, s2l_stream_t stream) {
  if (n_frames > 0 && (!d_new || !d_lip)) return S2L_E_NULL;
  s2l::CompArgs a;
  const int rc = composite_args(a, face_canon, face_stride, mask, mask_stride, coord, hole1, hole2, lip_h, lip_w, face_h, face_w,
                                x0, y0, pad_mode, expand_pad, n_frames);
  if (rc || n_frames == 0) return rc;
  if (n_frames > 65535) return S2L_E_SIZE;
  a.lip = nullptr; a.gt = nullptr; a.out_new = nullptr; a.out_can = nullptr; a.bgm = nullptr; a.rsize = a.chunks = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(d_lip, 0, sizeof(float) * 3 * (size_t)lip_h * lip_w * n_frames, st);
  if (e != hipSuccess) return (int)e;
  const int64_t per = (int64_t)face_h * face_w;
  hipLaunchKernelGGL(s2l::composite_bwd_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, a, d_new,
                     d_lip);
  return (int)hipGetLastError();
}

// ---- inference fast path (lip_merge_kernel + span_kernel) ---------------------------------------------------------------------
extern "C" int64_t s2l_composite_stream_work_bytes(int lip_h, int lip_w, int face_h, int face_w, int64_t n_frames) {
  if (lip_h <= 0 || lip_w <= 0 || face_h <= 0 || face_w <= 0 || n_frames < 0) return 0;
  return n_frames * lip_h * lip_w * 16;      // merged lip boxes
}

extern "C" int s2l_composite_stream(const float* lip, const float* mask, const float* bgm, const float* rgb_gt, const float* coord,
                                    float* out_new, void* work, int lip_h, int lip_w, int face_h, int face_w, int x0, int y0,
                                    int pad_mode, int expand_pad, int64_t n_frames, s2l_stream_t stream) {
  if (n_frames > 0 && (!lip || !bgm || !rgb_gt || !out_new || !work)) return S2L_E_NULL;
  if (expand_pad < 0) return S2L_E_SIZE;                       // the rectangle mask is what makes most spans pure copies
  s2l::SpanArgs sa;
  s2l::CompArgs& a = sa.c;
  // `mask` doubles as face_canon for the shared checks: the fused table replaces the face, the mask is read at lip taps only
  const int rc = composite_args(a, mask, 0, mask, 0, coord, nullptr, nullptr, lip_h, lip_w, face_h, face_w, x0, y0, pad_mode,
                                expand_pad, n_frames);
  if (rc || n_frames == 0) return rc;
  const int64_t per = (int64_t)face_h * face_w;
  if (per % 4 != 0) return S2L_E_SIZE;
  if (s2l::misaligned16(bgm) || s2l::misaligned16(coord) || s2l::misaligned16(rgb_gt) || s2l::misaligned16(out_new)) return S2L_E_ALIGN;
  a.lip = lip; a.gt = rgb_gt; a.out_new = out_new; a.out_can = nullptr; a.bgm = bgm;
  a.rsize = a.chunks = 0;
  sa.nspans = (int)((per + 255) / 256);
  if (s2l::misaligned16(work)) return S2L_E_ALIGN;
  sa.merged = static_cast<float*>(work);
  sa.n_cand = n_frames * sa.nspans;
  if (n_frames > 65535) return S2L_E_SIZE;
  const int64_t blocks1 = (sa.n_cand + 3) / 4;
  if (blocks1 > 0x7fffffff) return S2L_E_SIZE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(s2l::lip_merge_kernel, dim3((unsigned)((lip_h * lip_w + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, sa);
  hipLaunchKernelGGL(s2l::span_kernel, dim3((unsigned)blocks1), dim3(256), 0, st, sa);
  return (int)hipGetLastError();
}
