// Probe ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = element index; every lane passes its own byte address; prints what
// the lanes receive for three address patterns (element indices; a [4][16] row-major block has row stride 16 elements).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t l[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) l[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const int i = lane & 15, g = lane >> 4;
  uint32_t addr;
  if (mode == 0) addr = 0;
  else if (mode == 1) addr = (uint32_t)(g * 128);
  else addr = (uint32_t)(g * 128 + (i >> 2) * 32 + (i & 3) * 8);
  addr += (uint32_t)(uintptr_t)l;
  u2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = (uint16_t)(v[0] & 0xffff);
  out[lane * 4 + 1] = (uint16_t)(v[0] >> 16);
  out[lane * 4 + 2] = (uint16_t)(v[1] & 0xffff);
  out[lane * 4 + 3] = (uint16_t)(v[1] >> 16);
}
int main() {
  uint16_t* d; hipMalloc(&d, 512);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int lane = 0; lane < 64; ++lane) if (lane < 20 || lane == 32 || lane == 48)
      printf("  lane %2d: %4d %4d %4d %4d\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
  }
  return 0;
}
