// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate on this box (1 wave/SIMD, NACC accumulators).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters) {
  f4 acc[NACC];
  float a = threadIdx.x * 1e-3f, b[8];
  for (int i = 0; i < 8; ++i) b[i] = threadIdx.x * 1e-4f + i;
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[u & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Replica of the render kernel's register pattern: 192 accumulator registers (AGPRs), 192 B-operand
// VGPRs, A operand reused by 3 consecutive MFMAs, accumulators rotating with period 3.
__device__ inline float rnd(unsigned x) {   // hash -> [-1, 1)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int)x * (1.0f / 2147483648.0f);
}

template <int G, bool RANDOM>
__global__ __launch_bounds__(256) void replica(float* out, int iters, long long* clk) {
  float in[G][64];
  f4 acc[G][16];
  f4 w = (f4){threadIdx.x * 1e-3f, 1.f, 2.f, 3.f};
  if (RANDOM) w = (f4){rnd(threadIdx.x * 4 + 1) / 16, rnd(threadIdx.x * 4 + 2) / 16, rnd(threadIdx.x * 4 + 3) / 16, rnd(threadIdx.x * 4 + 4) / 16};
  f4 ws[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) ws[i][r] = RANDOM ? rnd(threadIdx.x * 64 + i * 4 + r + 99) / 16 * ((i + r) & 1 ? 1.f : -1.f) : w[r];
  long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int g = 0; g < G; ++g)
    for (int i = 0; i < 64; ++i) in[g][i] = RANDOM ? rnd(threadIdx.x * 1000 + blockIdx.x * 7919 + g * 64 + i) : threadIdx.x * 1e-4f + i + g;
  for (int g = 0; g < G; ++g)
    for (int i = 0; i < 16; ++i) acc[g][i] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
      for (int j4 = 0; j4 < 16; ++j4) {
        w = ws[j4 & 7];
        asm volatile("" : "+v"(ws[j4 & 7]));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int g = 0; g < G; ++g)
            acc[g][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[jj], in[g][j4 * 4 + jj], acc[g][mb], 0, 0, 0);
      }
  }
  float s = 0;
  for (int g = 0; g < G; ++g)
    for (int i = 0; i < 16; ++i) s += acc[g][i][0] + acc[g][i][1] + acc[g][i][2] + acc[g][i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - r0; }
}

// Same FLOPs through v_mfma_f32_32x32x2_f32: 2 groups of 32 samples, 8 M-blocks of 32 features.
typedef float f16v __attribute__((ext_vector_type(16)));
template <bool RANDOM>
__global__ __launch_bounds__(256) void replica32(float* out, int iters, long long* clk) {
  float in[128];
  f16v acc[8];
  f4 w = (f4){threadIdx.x * 1e-3f, 1.f, 2.f, 3.f};
  if (RANDOM) w = (f4){rnd(threadIdx.x * 4 + 1) / 16, rnd(threadIdx.x * 4 + 2) / 16, rnd(threadIdx.x * 4 + 3) / 16, rnd(threadIdx.x * 4 + 4) / 16};
  long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < 128; ++i) in[i] = RANDOM ? rnd(threadIdx.x * 1000 + blockIdx.x * 7919 + i) : threadIdx.x * 1e-4f + i;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int j4 = 0; j4 < 32; ++j4) {
        if (RANDOM) { w = -w; float t = w[0]; w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = t; }
        asm volatile("" : "+v"(w));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[jj], in[j4 * 4 + jj], acc[mb], 0, 0, 0);
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - r0; }
}

template <bool RANDOM>
void run_replica32(const char* name) {
  float* out; long long* clk;
  hipMalloc(&clk, 16); hipMalloc(&out, sizeof(float) * 256 * 256);
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  replica32<RANDOM><<<256, 256>>>(out, 2, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  replica32<RANDOM><<<256, 256>>>(out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = 1024.0 * iters * 8 * 128;
  double tf = nm * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-28s %8.3f ms  %7.2f TFLOP/s | s_memtime %.2f ticks/MFMA (ideal 64), memtime rate %.1f MHz\n", name, ms, tf,
         (double)h[0] / (iters * 8.0 * 128), 100.0 * h[0] / h[1]);
}

template <int G, bool RANDOM>
void run_replica(const char* name) {
  float* out;
  long long* clk;
  hipMalloc(&clk, 16);
  hipMalloc(&out, sizeof(float) * 256 * 256);
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  replica<G, RANDOM><<<256, 256>>>(out, 2, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  replica<G, RANDOM><<<256, 256>>>(out, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = 1024.0 * iters * 256.0 * 4 * G;
  double tf = nm * 2 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  float chk; hipMemcpy(&chk, out, 4, hipMemcpyDeviceToHost);
  printf("%-28s %8.3f ms  %7.2f TFLOP/s   %.2f cyc/MFMA @2.4GHz | s_memtime %.2f ticks/MFMA, memtime rate %.1f MHz (chk %g)\n", name, ms, tf,
         ms * 1e-3 * 2.4e9 / (iters * 256.0 * 4 * G), (double)h[0] / (iters * 256.0 * 4 * G), 100.0 * h[0] / h[1], chk);
  hipFree(out);
}

template <int NACC, int WAVES>
void run(const char* name, int blocks) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * WAVES * 64);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, WAVES><<<blocks, WAVES * 64>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, WAVES><<<blocks, WAVES * 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)blocks * WAVES * iters * 16.0 * NACC;
  double tf = nm * 2 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
  double cyc_per = (ms * 1e-3 * 2.4e9) / (iters * 16.0 * NACC * ((blocks * WAVES + 1023) / 1024));
  printf("%-28s blocks %5d  %8.3f ms  %7.2f TFLOP/s   %.2f cyc/MFMA/SIMD @2.4GHz\n", name, blocks, ms, tf, cyc_per);
  hipFree(out);
}

int main() {
  run<1, 4>("nacc1 4waves", 256);
  run<2, 4>("nacc2 4waves", 256);
  run<3, 4>("nacc3 4waves", 256);
  run<4, 4>("nacc4 4waves", 256);
  run<12, 4>("nacc12 4waves", 256);
  run<3, 4>("nacc3 4waves x2 blocks/CU", 512);
  run<1, 12>("nacc1 12waves", 256);
  run<2, 8>("nacc2 8waves", 256);
  run_replica<3, false>("replica G=3 trivial data");
  run_replica<3, true>("replica G=3 RANDOM data");
  run_replica32<false>("replica 32x32x2 trivial");
  run_replica32<true>("replica 32x32x2 RANDOM");
  run_replica<3, true>("replica G=3 RANDOM data");
  return 0;
}
