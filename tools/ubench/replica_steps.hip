// Microbenchmark: add the render kernel's ingredients one at a time to the bare MFMA stream and
// report cycles per MFMA (s_memtime) -- finds which ingredient costs what.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/replica_steps.hip -o ubin/replica_steps
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ inline float rnd(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int)x * (1.0f / 2147483648.0f);
}
enum { DSREAD = 1, BIAS = 2, BARRIER = 4, EPILOGUE = 8, DYNBUF = 16, LGKM0 = 32, DMA = 64, TWOSET = 128 };

__device__ __forceinline__ void dma_4k(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int layers, long long* clk, const float* wsrc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 3;
  const int lane = threadIdx.x & 63;
  float in[G][64];
  f4 acc[G][16];
  f4* ring = reinterpret_cast<f4*>(smem);
  for (int i = threadIdx.x; i < 9 * 1024 + 512; i += 256)
    ring[i] = (f4){rnd(i * 4 + 1) / 16, rnd(i * 4 + 2) / 16, rnd(i * 4 + 3) / 16, rnd(i * 4 + 4) / 16};
  __syncthreads();
  for (int g = 0; g < G; ++g)
    for (int i = 0; i < 64; ++i) in[g][i] = rnd(threadIdx.x * 1000 + blockIdx.x * 7919 + g * 64 + i);
  for (int g = 0; g < G; ++g)
    for (int i = 0; i < 16; ++i) acc[g][i] = (f4){0.f, 0.f, 0.f, 0.f};
  const f4* bias = ring + 9 * 1024 + (lane >> 4);
  int cur = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* gq = reinterpret_cast<const char*>(wsrc) + wave * 4096 + lane * 16;
  const unsigned lq = (unsigned)(unsigned long long)smem + wave * 4096;
  int islab = 0;
  if (MODE & DMA) {
    for (int s2 = 0; s2 < 8; ++s2) { dma_4k(gq + (islab % 113) * 16384, lq + (islab % 9) * 16384); ++islab; }
  }
  long long t0 = __builtin_readcyclecounter();
  f4 w0 = ring[lane], w1 = ring[64 + lane];
  for (int layer = 0; layer < layers; ++layer) {
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      const f4* sl = ring + ((MODE & DYNBUF) ? cur : (mb % 8)) * 1024 + lane;
      if (MODE & BIAS) {
        const f4 b = bias[mb * 4];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][mb] = b;
      }
#pragma unroll
      for (int j4 = 0; j4 < 16; ++j4) {
        if (j4 == 14) {
          if (MODE & LGKM0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (MODE & DMA) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
          if (MODE & BARRIER) asm volatile("s_barrier" ::: "memory");
          if (MODE & DMA) { dma_4k(gq + (islab % 113) * 16384, lq + (islab % 9) * 16384); ++islab; }
          if (MODE & DYNBUF) { cur = cur + 1 == 9 ? 0 : cur + 1; sl = ring + cur * 1024 + lane - 16 * 64; }
          else sl = ring + ((mb + 1) % 8) * 1024 + lane - 16 * 64;
        }
        f4 w2;
        if (MODE & DSREAD) w2 = sl[(j4 + ((MODE & TWOSET) ? 1 : 2)) * 64];
        else { w2 = w0; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[jj], in[g][j4 * 4 + jj], acc[g][mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & TWOSET) { w0 = w2; }
        else if (MODE & DSREAD) { w0 = w1; w1 = w2; }
      }
    }
    if (MODE & EPILOGUE) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mb = 0; mb < 16; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) in[g][mb * 4 + r] = fmaxf(acc[g][mb][r], 0.f) * 0.05f;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int g = 0; g < G; ++g)
    for (int i = 0; i < 16; ++i) s += acc[g][i][0] + acc[g][i][1] + acc[g][i][2] + acc[g][i][3] + in[g][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
  float* out; long long* clk; float* wsrc;
  hipMalloc(&clk, 16); hipMalloc(&out, sizeof(float) * 256 * 256); hipMalloc(&wsrc, 113 * 16384); hipMemset(wsrc, 0, 113 * 16384);
  const int lds = (9 * 1024 + 512) * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int layers = 70;
  k<MODE><<<256, 256, lds>>>(out, 7, clk, wsrc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<256, 256, lds>>>(out, layers, clk, wsrc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  double nm = layers * 3072.0;
  printf("%-44s %8.3f ms %7.2f TFLOP/s  %.2f ticks/MFMA  (%.0f ticks/layer)\n", name, ms,
         1024.0 * nm * 2048 / (ms * 1e-3) / 1e12, h / nm, (double)h / layers);
}

int main() {
  run<0>("bare MFMA stream");
  run<DSREAD>("+ ds_read w (3-set rotation)");
  run<DSREAD | BIAS>("+ bias init from LDS");
  run<DSREAD | BIAS | LGKM0>("+ lgkmcnt(0) per slab");
  run<DSREAD | BIAS | LGKM0 | BARRIER>("+ s_barrier per slab");
  run<DSREAD | BIAS | LGKM0 | BARRIER | EPILOGUE>("+ relu epilogue per layer");
  run<DSREAD | BIAS | LGKM0 | BARRIER | EPILOGUE | DYNBUF>("+ dynamic ring index");
  run<DSREAD | EPILOGUE>("ds_read + epilogue only");
  run<DSREAD | BIAS | LGKM0 | BARRIER | EPILOGUE | DYNBUF | DMA>("+ LDS-DMA ring (all ingredients)");
  run<DSREAD | BIAS | LGKM0 | BARRIER | DYNBUF | DMA>("all but epilogue");
  run<DSREAD | TWOSET>("ds_read w, 2-set rotation");
  run<DSREAD | BIAS | LGKM0 | BARRIER | EPILOGUE | DYNBUF | DMA>("+ LDS-DMA ring (all ingredients) again");
  return 0;
}
