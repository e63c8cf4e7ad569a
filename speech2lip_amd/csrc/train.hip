// Training-step kernels (BASELINE config 5: MLP forward + backward).
//
//   s2l_train_forward   general-row MLP forward that also saves h0..h7      (tf_nerf.py:225-285)
//   s2l_train_backward  dz7..dz0 chain + audio-column gradient              (autograd of the above)
//   s2l_wgrad           dW[256,K] = dz^T in, K in {128, 256}: fp32 MFMA, split over rows, two-stage
//                       deterministic reduction
//   s2l_colsum          bias gradients / small outer products (output layer)
//   s2l_ensemble_*      4-tap ensemble rows, area-weighted reduce and its backward (training.py:158-251)
//   s2l_mse             photometric loss + its gradient                     (training.py:605-619)
// All fp32 (exact-parity mode).  The reference reaches these through torch autograd
// (training.py:559 loss.backward()); there is no reference source to mirror line by line.
#include "s2l_common.h"

namespace s2l {

int launch_rows_fwd(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st);
int launch_rows_bwd(const float* packed, const float* drgb, const float* hsave, float* dzsave, float* dxa,
                           int64_t n_rows, hipStream_t st);

__device__ inline f4 mfma16w(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- weight gradient: partial[blk][out 256][in K] = sum over the block's rows of dz[row][out] * in[row][k] --------
// MFMA 16x16x4 with k = row: A[i = out feature][k = row] = dz[row][out], B[k = row][j = in feature] = in[row][in].
// Wave w of 4 owns out-feature blocks 4w..4w+3 x all KB in-feature blocks: 4*KB accumulators of 4 registers.
template <int KB>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ in,
                                                    int ldin, float* __restrict__ partial, float* __restrict__ bias_partial,
                                                    int64_t n_rows, int64_t rows_per_block) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = lane >> 4, i = lane & 15;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n_rows) r1 = n_rows;
  f4 acc[4][KB];
  float asum[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of dz (bias gradient) ride along for free
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < KB; ++n) acc[m][n] = (f4){0.f, 0.f, 0.f, 0.f};
  // branch-free operand loads: rows past the block's end read a valid (clamped) row and are
  // cancelled by zeroing the A operand only (inputs are finite)
  auto load = [&](int64_t r, float (&a)[4], float (&b)[KB]) {
    const int64_t row = r + q;
    const float okf = row < r1 ? 1.f : 0.f;
    const int64_t rc = row < n_rows ? row : n_rows - 1;
    const float* dzr = dz + rc * ldz + wave * 64 + i;
    const float* inr = in + rc * ldin + i;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = dzr[m * 16] * okf;
#pragma unroll
    for (int n = 0; n < KB; ++n) b[n] = inr[n * 16];
  };
  float a0[4], b0[KB], a1[4], b1[KB];
  load(r0, a0, b0);
  for (int64_t r = r0; r < r1; r += 8) {   // two 4-row k-steps per iteration, operands loaded one step ahead
    load(r + 4, a1, b1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      asum[m] += a0[m];
#pragma unroll
      for (int n = 0; n < KB; ++n) acc[m][n] = mfma16w(a0[m], b0[n], acc[m][n]);
    }
    load(r + 8, a0, b0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      asum[m] += a1[m];
#pragma unroll
      for (int n = 0; n < KB; ++n) acc[m][n] = mfma16w(a1[m], b1[n], acc[m][n]);
    }
  }
  // D[row = 4q + r -> out feature][col = i -> in feature]
  // the accumulators live in AGPRs: move them out and store them one M-block at a time (a scheduling barrier between the
  // blocks keeps the compiler from reading all 256 of them into VGPRs first, which spilled 36 B per lane)
  float* p = partial + (int64_t)blockIdx.x * 256 * (KB * 16) + (int64_t)(wave * 64 + 4 * q) * (KB * 16) + i;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int n = 0; n < KB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[(m * 16 + r) * (KB * 16) + n * 16] = acc[m][n][r];
    __builtin_amdgcn_sched_barrier(0);
  }
  if (bias_partial) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float v = asum[m];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (q == 0) bias_partial[(int64_t)blockIdx.x * 256 + wave * 64 + m * 16 + i] = v;
    }
  }
}

// out[e] = sum over the blocks' partials in a fixed order: 64 elements per workgroup, four threads per element take every fourth
// partial in order, then the four sums are added in order.  A second small tensor (the bias gradient's partials) rides in the
// same launch: the workgroups behind the first tensor's take it.  (One thread per element and a workgroup of its own for the
// bias cost 155 us each per GEMM: 512 dependent loads.)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int n_blocks, int64_t n_elems, const float* __restrict__ partial2,
                                                             float* __restrict__ out2, int64_t n_elems2) {
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int64_t nb1 = (n_elems + 63) / 64;
  int64_t blk = blockIdx.x;
  if (blk >= nb1) {     // (block-uniform)
    blk -= nb1;
    partial = partial2; out = out2; n_elems = n_elems2;
  }
  const int64_t e = blk * 64 + el;
  float s = 0.f;
  if (e < n_elems)
    for (int b = pl; b < n_blocks; b += 4) s += partial[(int64_t)b * n_elems + e];
  red[pl][el] = s;
  __syncthreads();
  if (pl == 0 && e < n_elems) out[e] = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
}

// partial[blk][m][c] = sum over the block's rows of a[row][m] * b[row][c], m < M <= 4, c < C <= 256 (thread = c).
// M = 1 with a = nullptr is a plain column sum (bias gradient).
__global__ __launch_bounds__(256) void small_outer_kernel(const float* __restrict__ a, int lda, int M,
                                                         const float* __restrict__ b, int ldb, int C,
                                                         float* __restrict__ partial, int64_t n_rows,
                                                         int64_t rows_per_block) {
  const int c = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n_rows) r1 = n_rows;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (int64_t r = r0; r < r1; ++r) {
      const float v = b[r * ldb + c];
      if (a) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < M) acc[m] = fmaf(a[r * lda + m], v, acc[m]);
      } else {
        acc[0] += v;
      }
    }
    for (int m = 0; m < M; ++m) partial[((int64_t)blockIdx.x * M + m) * C + c] = acc[m];
  }
}

// ---- ensemble / loss elementwise ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ensemble_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ areas,
                                                          float* __restrict__ drgb, int64_t n) {
  // blockIdx.y = frame: frame f owns rows [4 f n, 4 (f+1) n) of drgb / areas and [f n, (f+1) n) of dpred
  dpred += (int64_t)blockIdx.y * n * 3;
  areas += (int64_t)blockIdx.y * n * 4;
  drgb += (int64_t)blockIdx.y * n * 12;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 3) return;
  const int64_t p = i / 3;
  const float a0 = areas[p], a1 = areas[n + p], a2 = areas[2 * n + p], a3 = areas[3 * n + p];
  const float tot = ((a0 + a1) + a2) + a3;
  const float g = dpred[i];
  drgb[i] = g * (a3 / tot);               // tap t is weighted by the diagonally opposite area (training.py:244-245)
  drgb[n * 3 + i] = g * (a2 / tot);
  drgb[2 * n * 3 + i] = g * (a1 / tot);
  drgb[3 * n * 3 + i] = g * (a0 / tot);
}

// dpred = scale * (pred - target); block partial of sum (pred - target)^2
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                 float scale, float* __restrict__ dpred, float* __restrict__ partial,
                                                 int64_t n) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    s = fmaf(d, d, s);
    if (dpred) dpred[i] = scale * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ loss) {
  // n <= 1024 block partials: four per thread in index order, then a fixed-shape tree (deterministic)
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < n && i < threadIdx.x * 4 + 4; ++i) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] * scale;
}

// ---- small reductions / un-folding that used to run through ATen -----------------------------------------------------
// partial[s][k][c] = sum of rows [k*rpc, (k+1)*rpc) of segment s (rows_per_seg rows of src, row stride ld), c < C <= 256;
// blockDim 256 = (256 / C) row lanes x C columns, combined in lane order through LDS.
constexpr int kSegChunks = 32;
__global__ __launch_bounds__(256) void segment_colsum_kernel(const float* __restrict__ src, int ld, int C, int64_t rows_per_seg,
                                                            int64_t rpc, float* __restrict__ partial) {
  __shared__ float red[256];
  const int lanes = 256 / C;
  const int c = threadIdx.x % C, lane = threadIdx.x / C;
  const int64_t seg = blockIdx.y, k = blockIdx.x;
  const int64_t r0 = k * rpc;
  int64_t r1 = r0 + rpc;
  if (r1 > rows_per_seg) r1 = rows_per_seg;
  float acc = 0.f;
  if (lane < lanes)
    for (int64_t r = r0 + lane; r < r1; r += lanes) acc += src[(seg * rows_per_seg + r) * ld + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * C + threadIdx.x];
    partial[(seg * kSegChunks + k) * C + threadIdx.x] = s;
  }
}
__global__ __launch_bounds__(256) void segment_colsum_final_kernel(const float* __restrict__ partial, int C, float* __restrict__ out,
                                                                  int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t seg = i / C;
  const int c = (int)(i - seg * C);
  float s = 0.f;
  for (int k = 0; k < kSegChunks; ++k) s += partial[(seg * kSegChunks + k) * C + c];
  out[i] = s;
}

// Un-fold of the pack-time fold G = Wf [Wuv|Wa|Wt] (126 columns), c = Wf (buv + ba + bt) + bf  (s2l_layout.h; Wf = W0 or
// W5[:, :256]):   dWf = dG C^T + dc bsum^T,   dC = Wf^T dG,   d(buv) = d(ba) = d(bt) = Wf^T dc.
// Block j (256 threads): thread i forms dWf[i][j]; threads k < 126 form dC[j][k]; thread 126 forms the bias gradient j.
struct UnfoldArgs {
  const float* dG;       // [256,128], columns 0..125 used
  const float* dc;       // [256]
  const float* wf;       // [256, ldwf]
  int ldwf;
  const float *wuv, *wa, *wt, *buv, *ba, *bt;      // [256,42] [256,64] [256,20] [256] x3
  float* dwf;            // [256, ldo] (columns 0..255 written)
  int ldo;
  const float* right;    // NULL, or [256,256] copied to columns 256..511 of dwf (the W5[:, 256:] gradient)
  float *dwuv, *dwa, *dwt, *db;                    // db [256]: the common gradient of the three biases
};
__global__ __launch_bounds__(256) void unfold_kernel(UnfoldArgs a) {
  __shared__ float crow[128], wcol[256], dcs[256];
  const int j = blockIdx.x, t = threadIdx.x;
  if (t < 42) crow[t] = a.wuv[j * 42 + t];
  else if (t < 106) crow[t] = a.wa[j * 64 + t - 42];
  else if (t < 126) crow[t] = a.wt[j * 20 + t - 106];
  wcol[t] = a.wf[(int64_t)t * a.ldwf + j];
  dcs[t] = a.dc[t];
  __syncthreads();
  const float bsum = (a.buv[j] + a.ba[j]) + a.bt[j];
  float acc = 0.f;
  for (int k = 0; k < 126; ++k) acc = fmaf(a.dG[t * 128 + k], crow[k], acc);
  a.dwf[(int64_t)t * a.ldo + j] = fmaf(dcs[t], bsum, acc);
  if (a.right) a.dwf[(int64_t)t * a.ldo + 256 + j] = a.right[t * 256 + j];
  if (t < 126) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s = fmaf(wcol[i], a.dG[i * 128 + t], s);
    if (t < 42) a.dwuv[j * 42 + t] = s;
    else if (t < 106) a.dwa[j * 64 + t - 42] = s;
    else a.dwt[j * 20 + t - 106] = s;
  } else if (t == 126) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s = fmaf(wcol[i], dcs[i], s);
    a.db[j] = s;
  }
}

constexpr int kSplitBlocks = 512;   // row blocks of the split reductions (2 per CU)

}  // namespace s2l

using namespace s2l;

extern "C" int s2l_train_forward(const float* packed, const float* x, float* hsave, float* rgb, int64_t n_rows,
                                 s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed || !x || !hsave || !rgb) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(x) || misaligned16(hsave)) return S2L_E_ALIGN;
  return launch_rows_fwd(packed, x, rgb, hsave, n_rows, static_cast<hipStream_t>(stream));
}

extern "C" int s2l_train_backward(const float* packed, const float* drgb, const float* hsave, float* dzsave, float* dxa,
                                  int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed || !drgb || !hsave || !dzsave || !dxa) return S2L_E_NULL;
  if (misaligned16(packed) || misaligned16(hsave) || misaligned16(dzsave) || misaligned16(dxa)) return S2L_E_ALIGN;
  return launch_rows_bwd(packed, drgb, hsave, dzsave, dxa, n_rows, static_cast<hipStream_t>(stream));
}

extern "C" int64_t s2l_split_work_floats(int64_t n_elems) { return n_elems < 0 ? 0 : (n_elems + 256) * kSplitBlocks; }

extern "C" int s2l_wgrad(const float* dz, int ldz, const float* in, int ldin, int k_in, float* work, float* dw, float* db,
                         int64_t n_rows, s2l_stream_t stream) {
  if (n_rows <= 0 || (k_in != 128 && k_in != 256) || ldz < 256 || ldin < k_in) return S2L_E_SIZE;
  if (!dz || !in || !work || !dw) return S2L_E_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t rpb = (n_rows + kSplitBlocks - 1) / kSplitBlocks;
  rpb = (rpb + 7) / 8 * 8;
  const int nblk = (int)((n_rows + rpb - 1) / rpb);
  const int64_t ne = 256 * (int64_t)k_in;
  float* bwork = db ? work + (int64_t)kSplitBlocks * ne : nullptr;   // bias partials behind the dW partials
  if (k_in == 256)
    hipLaunchKernelGGL(wgrad_kernel<16>, dim3(nblk), dim3(256), 0, st, dz, ldz, in, ldin, work, bwork, n_rows, rpb);
  else
    hipLaunchKernelGGL(wgrad_kernel<8>, dim3(nblk), dim3(256), 0, st, dz, ldz, in, ldin, work, bwork, n_rows, rpb);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((ne + 63) / 64 + (db ? 4 : 0))), dim3(256), 0, st, work, dw, nblk, ne,
                     (const float*)bwork, db, (int64_t)256);
  return (int)hipGetLastError();
}

// out[m][c] = sum_rows a[row][m] * b[row][c]  (a == NULL, M == 1: column sums of b).  M <= 4, C <= 256.
extern "C" int s2l_small_outer(const float* a, int lda, int m, const float* b, int ldb, int c, float* work, float* out,
                               int64_t n_rows, s2l_stream_t stream) {
  if (n_rows <= 0 || m < 1 || m > 4 || c < 1 || c > 256 || (!a && m != 1)) return S2L_E_SIZE;
  if (!b || !work || !out) return S2L_E_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rpb = (n_rows + kSplitBlocks - 1) / kSplitBlocks;
  const int nblk = (int)((n_rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(small_outer_kernel, dim3(nblk), dim3(256), 0, st, a, lda, m, b, ldb, c, work, n_rows, rpb);
  const int64_t ne = (int64_t)m * c;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((ne + 63) / 64)), dim3(256), 0, st, work, out, nblk, ne, (const float*)nullptr,
                     (float*)nullptr, (int64_t)0);
  return (int)hipGetLastError();
}

extern "C" int s2l_ensemble_backward(const float* dpred, const float* areas, float* drgb, int64_t n_pixels,
                                     s2l_stream_t stream) {
  if (n_pixels < 0) return S2L_E_SIZE;
  if (n_pixels == 0) return S2L_OK;
  if (!dpred || !areas || !drgb) return S2L_E_NULL;
  hipLaunchKernelGGL(ensemble_bwd_kernel, dim3((unsigned)((n_pixels * 3 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dpred, areas, drgb, n_pixels);
  return (int)hipGetLastError();
}

extern "C" int s2l_ensemble_backward_batch(const float* dpred, const float* areas, float* drgb, int64_t n_pixels,
                                           int64_t n_frames, s2l_stream_t stream) {
  if (n_pixels < 0 || n_frames < 0 || n_frames > 65535) return S2L_E_SIZE;
  if (n_pixels == 0 || n_frames == 0) return S2L_OK;
  if (!dpred || !areas || !drgb) return S2L_E_NULL;
  hipLaunchKernelGGL(ensemble_bwd_kernel, dim3((unsigned)((n_pixels * 3 + 255) / 256), (unsigned)n_frames), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dpred, areas, drgb, n_pixels);
  return (int)hipGetLastError();
}

// loss = weight * mean((pred - target)^2) over n elements; dpred (optional) = d loss / d pred.  work: 1024 floats.
extern "C" int s2l_mse(const float* pred, const float* target, float weight, float* dpred, float* work, float* loss,
                       int64_t n_elems, s2l_stream_t stream) {
  if (n_elems <= 0) return S2L_E_SIZE;
  if (!pred || !target || !work || !loss) return S2L_E_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = (int)((n_elems + 255) / 256 < 1024 ? (n_elems + 255) / 256 : 1024);
  hipLaunchKernelGGL(mse_kernel, dim3(nblk), dim3(256), 0, st, pred, target, 2.f * weight / (float)n_elems, dpred, work,
                     n_elems);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, work, nblk, weight / (float)n_elems, loss);
  return (int)hipGetLastError();
}

// ---- one launch of Adam over every tensor of a parameter group (train.py:173-199's optimizer.step(); torch/optim/adam.py _single_tensor_adam) ----
// table: n_tensors records of {param, grad, exp_avg, exp_avg_sq} device pointers; blocks: one int2 per workgroup = {tensor, first element};
// counts[t] = elements of tensor t.  The arithmetic is torch's, operation for operation in fp32 (built with -ffp-contract=off):
//   g' = g + wd * p;  m = m + (g' - m) * (1 - b1)  [Tensor.lerp_];  v = v * b2 + ((1 - b2) * g') * g'  [mul_ + addcmul_];
//   p = p + (-(lr / bc1)) * (m / (sqrt(v) / sqrt(bc2) + eps))  [addcdiv_]
// nan_flags (optional, one int per tensor): set to 1 when the parameter AS READ holds a NaN -- check_weights (src/common.py:56-64), which
// the reference calls right before optimizer.step() (training.py:572), folded into the pass that reads every parameter anyway.
struct AdamRec { float* p; const float* g; float* m; float* v; };
constexpr int kAdamChunk = 4096;      // elements per workgroup of 256 threads
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamRec* __restrict__ table, const int2* __restrict__ blocks,
                                                        const int64_t* __restrict__ counts, float neg_step, float omb1, float b2, float omb2,
                                                        float eps, float wd, float bc2_sqrt, int* __restrict__ nan_flags) {
  const int2 blk = blocks[blockIdx.x];
  const AdamRec r = table[blk.x];
  const int64_t n = counts[blk.x];
  bool bad = false;
  for (int k = 0; k < kAdamChunk / 256; ++k) {
    const int64_t i = (int64_t)blk.y + k * 256 + threadIdx.x;
    if (i >= n) break;
    const float p = r.p[i];
    bad |= (p != p);
    float g = r.g[i];
    if (wd != 0.f) g = g + wd * p;
    float m = r.m[i], v = r.v[i];
    m = m + (g - m) * omb1;
    v = v * b2 + (omb2 * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    r.m[i] = m;
    r.v[i] = v;
    r.p[i] = p + neg_step * (m / denom);
  }
  if (nan_flags && bad) nan_flags[blk.x] = 1;      // (benign race: every writer stores the same value)
}

extern "C" int64_t s2l_adam_chunk(void) { return kAdamChunk; }

extern "C" int s2l_adam_step(const void* table, const void* blocks, const int64_t* counts, int64_t n_tensors, int64_t n_blocks, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int64_t step, int* nan_flags, s2l_stream_t stream) {
  if (n_tensors < 0 || n_blocks < 0 || n_blocks > 0x7fffffff || step < 1) return S2L_E_SIZE;
  if (n_tensors == 0 || n_blocks == 0) return S2L_OK;
  if (!table || !blocks || !counts) return S2L_E_NULL;
  // the scalars as torch forms them: python doubles (1 - beta, 1 - beta ** step, -(lr / bc1), bc2 ** 0.5) that reach the kernels as floats
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const AdamRec*>(table), static_cast<const int2*>(blocks), counts, (float)(-(lr / bc1)), (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2), nan_flags);
  return (int)hipGetLastError();
}

// out [S,c] = per-segment column sums of src [S*rows_per_segment, ld] (c <= 256 and a divisor of 256): the per-frame gradient
// of the audio feature from dxa (frame b owns 4*HW consecutive rows).  work: S * 32 * c floats.  Fixed summation order.
extern "C" int s2l_segment_colsums(const float* src, int ld, int c, int64_t rows_per_segment, int64_t n_segments, float* work,
                                   float* out, s2l_stream_t stream) {
  if (c < 1 || c > 256 || 256 % c != 0 || ld < c || rows_per_segment <= 0 || n_segments < 0 || n_segments > 65535) return S2L_E_SIZE;
  if (n_segments == 0) return S2L_OK;
  if (!src || !work || !out) return S2L_E_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rpc = (rows_per_segment + kSegChunks - 1) / kSegChunks;
  hipLaunchKernelGGL(segment_colsum_kernel, dim3(kSegChunks, (unsigned)n_segments), dim3(256), 0, st, src, ld, c, rows_per_segment,
                     rpc, work);
  const int64_t n = n_segments * c;
  hipLaunchKernelGGL(segment_colsum_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, c, out, n);
  return (int)hipGetLastError();
}

// Gradients of the tensors behind a folded first/skip layer (see unfold_kernel).  first_w [256, ld_first] = pts_linears.0.weight
// (ld 256) or pts_linears.5.weight (ld 512, its left half is the folded factor); d_first [256, ld_first] receives the gradient of
// that factor in columns 0..255 and, when right != NULL, `right` [256,256] in columns 256..511.
extern "C" int s2l_unfold_first_layer(const float* dG, const float* dc, const float* first_w, int ld_first, const float* w_uv,
                                      const float* w_audio, const float* w_time, const float* b_uv, const float* b_audio,
                                      const float* b_time, const float* right, float* d_first, float* d_w_uv, float* d_w_audio,
                                      float* d_w_time, float* d_bias, s2l_stream_t stream) {
  if (ld_first != 256 && ld_first != 512) return S2L_E_SIZE;
  if (right && ld_first != 512) return S2L_E_SIZE;
  if (!dG || !dc || !first_w || !w_uv || !w_audio || !w_time || !b_uv || !b_audio || !b_time || !d_first || !d_w_uv || !d_w_audio ||
      !d_w_time || !d_bias)
    return S2L_E_NULL;
  UnfoldArgs a{dG, dc, first_w, ld_first, w_uv, w_audio, w_time, b_uv, b_audio, b_time, d_first, ld_first, right,
               d_w_uv, d_w_audio, d_w_time, d_bias};
  hipLaunchKernelGGL(unfold_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
