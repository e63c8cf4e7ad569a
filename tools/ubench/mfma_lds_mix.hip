// How fast can v_mfma_f32_32x32x16_bf16 issue when every MFMA needs its own 1-KiB A operand from LDS (one 32-row group per
// wave, the layout of the bf16 training kernels)?  CHAINS = independent accumulators per wave; blockDim 256 = one wave per
// SIMD, 512 = two.  A quads are prefetched D-1 steps ahead with static register rotation (fully unrolled 16-step body).
// Prints cycles per MFMA per SIMD (32 = the pipe's limit).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int CHAINS, int D>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u4* l = reinterpret_cast<u4*>(smem);
  for (int i = threadIdx.x; i < 6144; i += blockDim.x) l[i] = u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f16v a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
  const u4 b0 = u4{0x3c003c00u, 0, 0, 0};
  const u4* p = l + lane;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u4 q0[D], q1[D];
#pragma unroll
    for (int t = 0; t < D - 1; ++t) q0[t] = p[t * 128], q1[t] = p[t * 128 + 64];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t + D - 1 < 16) q0[(t + D - 1) % D] = p[(t + D - 1) * 128], q1[(t + D - 1) % D] = p[(t + D - 1) * 128 + 64];
      __builtin_amdgcn_sched_barrier(0);
      const int c0 = CHAINS == 4 ? 2 * (t & 1) : 0;
      a[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, q0[t % D]), __builtin_bit_cast(bf8, b0), a[c0], 0, 0, 0);
      a[c0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, q1[t % D]), __builtin_bit_cast(bf8, b0), a[c0 + 1], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float sacc = 0;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) sacc += a[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS, int D> void run(int threads) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 512;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<CHAINS, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<CHAINS, D>), dim3(256), dim3(threads), 98304, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
  printf("chains %d, prefetch depth %d, waves/SIMD %d: %.1f cycles per MFMA per SIMD\n", CHAINS, D, threads / 256,
         avg / (32.0 * iters * (threads / 256)));
}
int main() {
  run<2, 3>(256); run<2, 3>(512); run<2, 5>(256); run<2, 5>(512);
  run<4, 3>(256); run<4, 3>(512); run<4, 5>(256); run<4, 5>(512);
  return 0;
}
