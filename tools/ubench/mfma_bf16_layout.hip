// Checks the operand/result lane maps assumed for v_mfma_f32_32x32x16_bf16 on gfx950:
//   A: lane l holds A[i = l&31][k = 8*(l>>5) + j], j = 0..7;  B: lane l holds B[k = 8*(l>>5) + j][n = l&31]
//   D: lane l, reg r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// with asymmetric integer-valued matrices (exact in bf16).  Prints "layout OK" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ short bf(float x) { return (short)(__float_as_uint(x) >> 16); }
__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  bf8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = bf(A[(l & 31) * 16 + 8 * (l >> 5) + j]);
    b[j] = bf(B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
  }
  f16v c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) hA[i * 16 + kk] = (float)((i * 3 + kk * 5) % 7 - 3);
  for (int kk = 0; kk < 16; ++kk) for (int n = 0; n < 32; ++n) hB[kk * 32 + n] = (float)((kk * 2 + n * 7) % 5 - 2);
  for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + n]; ref[i * 32 + n] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  for (int i = 0; i < 1024; ++i) if (hD[i] != ref[i]) { printf("MISMATCH at %d,%d: %f vs %f\n", i / 32, i % 32, hD[i], ref[i]); return 1; }
  printf("layout OK\n");
  return 0;
}
