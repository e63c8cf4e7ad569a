// Streaming ceiling for the composite's access mix: per pixel read 8 B (coord) + 12 B (observed frame), write 12 B,
// nothing else.  256 frames of 500x500.  Prints GB/s for a few per-thread widths.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PPT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ coord, const float* __restrict__ gt, float* __restrict__ out, long n) {
  long base = ((long)blockIdx.x * 256 * PPT) + threadIdx.x;
  float2 g[PPT]; float a[PPT][3];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    long p = base + i * 256; if (p >= n) p = n - 1;
    g[i] = *reinterpret_cast<const float2*>(coord + 2 * p);
    a[i][0] = __builtin_nontemporal_load(gt + 3 * p); a[i][1] = __builtin_nontemporal_load(gt + 3 * p + 1); a[i][2] = __builtin_nontemporal_load(gt + 3 * p + 2);
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    long p = base + i * 256; if (p >= n) continue;
    __builtin_nontemporal_store(a[i][0] + g[i].x, out + 3 * p); __builtin_nontemporal_store(a[i][1] + g[i].y, out + 3 * p + 1); __builtin_nontemporal_store(a[i][2], out + 3 * p + 2);
  }
}
template <int PPT> void run(const float* c, const float* g, float* o, long n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = (int)((n + 256 * PPT - 1) / (256 * PPT));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<PPT>, dim3(blocks), dim3(256), 0, 0, c, g, o, n);
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<PPT>, dim3(blocks), dim3(256), 0, 0, c, g, o, n); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("PPT %d: %.3f ms  %.0f GB/s\n", PPT, ms, 32.0 * n / ms / 1e6);
}
// the composite's block -> pixel mapping: block b works in region b & 7 (an eighth of the frame), chunk (b >> 3) % chunks of frame (b >> 3) / chunks
__global__ __launch_bounds__(256) void kreg(const float* __restrict__ coord, const float* __restrict__ gt, float* __restrict__ out, int per, int rsize, int chunks, int mode) {
  const int region = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int chunk = kk % chunks; const long f = kk / chunks;
  const int in_region = chunk * 256 + threadIdx.x;
  const int pix = region * rsize + in_region;
  if (in_region >= rsize || pix >= per) return;
  const long p = f * per + pix;
  float2 g; float a0, a1, a2;
  if (mode == 0) { g.x = __builtin_nontemporal_load(coord + 2 * p); g.y = __builtin_nontemporal_load(coord + 2 * p + 1); }
  else g = *reinterpret_cast<const float2*>(coord + 2 * p);
  a0 = __builtin_nontemporal_load(gt + 3 * p); a1 = __builtin_nontemporal_load(gt + 3 * p + 1); a2 = __builtin_nontemporal_load(gt + 3 * p + 2);
  __builtin_nontemporal_store(a0 + g.x, out + 3 * p); __builtin_nontemporal_store(a1 + g.y, out + 3 * p + 1); __builtin_nontemporal_store(a2, out + 3 * p + 2);
}
void runreg(const float* c, const float* g, float* o, int frames, int mode, int rsize_align) {
  const int per = 250000; int rsize = (per + 7) / 8; if (rsize_align) rsize = (rsize + rsize_align - 1) / rsize_align * rsize_align;
  const int chunks = (rsize + 255) / 256; const long blocks = 8L * chunks * frames;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kreg, dim3(blocks), dim3(256), 0, 0, c, g, o, per, rsize, chunks, mode);
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kreg, dim3(blocks), dim3(256), 0, 0, c, g, o, per, rsize, chunks, mode); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("region mapping, coord load mode %d, rsize %d: %.3f ms  %.0f GB/s\n", mode, rsize, ms, 32.0 * per * frames / ms / 1e6);
}
int main() {
  long n = 256L * 500 * 500; float *c, *g, *o;
  hipMalloc(&c, n * 8); hipMalloc(&g, n * 12); hipMalloc(&o, n * 12);
  hipMemset(c, 0, n * 8); hipMemset(g, 0, n * 12);
  run<1>(c, g, o, n); run<2>(c, g, o, n); run<4>(c, g, o, n);
  runreg(c, g, o, 256, 0, 0); runreg(c, g, o, 256, 1, 0); runreg(c, g, o, 256, 1, 256); runreg(c, g, o, 256, 1, 32);
  return 0;
}
