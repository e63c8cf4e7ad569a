// 8-bit output frames as the reference writes them: cv2.imwrite(path, rgb * 255) (inference.py:177) converts the float
// image with saturate_cast<uchar>, i.e. round-to-nearest-even then clamp to [0,255].  Quantising on the device cuts the
// D2H copy and the optional uint8 all-gather of a clip 4x.  HBM-bound streaming kernel: 16 B in, 4 B out per thread.
#include "s2l_common.h"

namespace s2l {

__device__ __forceinline__ uint32_t q8(float x) {
  const float r = rintf(x * 255.f);                 // cvRound
  return (uint32_t)fminf(fmaxf(r, 0.f), 255.f);     // saturate; NaN -> 0 like the integer conversion of a clamped value
}

// head: elements before the first 16-byte-aligned float of x (0..3); they are converted one by one by the first threads, the
// aligned body four at a time.  `out` only needs byte alignment for the head; the body stores bytes when out+head is not
// 4-byte aligned (ALIGNED_OUT false).
template <bool ALIGNED_OUT>
__global__ __launch_bounds__(256) void to8b_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t n, int head) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < head) out[t] = (uint8_t)q8(x[t]);
  const int64_t i = head + t * 4;
  if (i + 3 < n) {
    const f4 v = *reinterpret_cast<const f4*>(x + i);
    if (ALIGNED_OUT) {
      *reinterpret_cast<uint32_t*>(out + i) = q8(v[0]) | (q8(v[1]) << 8) | (q8(v[2]) << 16) | (q8(v[3]) << 24);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) out[i + k] = (uint8_t)q8(v[k]);
    }
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = (uint8_t)q8(x[j]);
  }
}

// The reader's conversion the other way (someones_lip_dataset.py:196-217 get_color: imageio array / 255. in float64, then
// torch.Tensor(...) -> float32): out = (float)((double)u8 / 255.0), the same two roundings.  Lets decoded frames cross PCIe as
// bytes (a quarter of the traffic) and land as the floats the reference's reader would have produced, bit for bit.
__global__ __launch_bounds__(256) void from8b_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(in + i);
    f4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (float)((double)((w >> (8 * k)) & 255u) / 255.0);
    *reinterpret_cast<f4*>(out + i) = v;
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = (float)((double)in[j] / 255.0);
  }
}

}  // namespace s2l

extern "C" int s2l_from8b(const uint8_t* in, float* out, int64_t n, s2l_stream_t stream) {
  if (n < 0) return S2L_E_SIZE;
  if (n == 0) return S2L_OK;
  if (!in || !out) return S2L_E_NULL;
  if ((reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 15)) return S2L_E_ALIGN;
  const unsigned blocks = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(s2l::from8b_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, n);
  return (int)hipGetLastError();
}

extern "C" int s2l_to8b(const float* rgb, uint8_t* out, int64_t n, s2l_stream_t stream) {
  if (n < 0) return S2L_E_SIZE;
  if (n == 0) return S2L_OK;
  if (!rgb || !out) return S2L_E_NULL;
  if (reinterpret_cast<uintptr_t>(rgb) & 3) return S2L_E_ALIGN;   // floats
  // a frame slice of a clip starts at base + k*H*W*3 floats: any 4-byte phase of the 16-byte vector loads is legal
  int head = (int)((16 - (reinterpret_cast<uintptr_t>(rgb) & 15)) & 15) / 4;
  if (head > n) head = (int)n;
  const int64_t body = n - head;
  const unsigned blocks = (unsigned)((body + 1023) / 1024 > 0 ? (body + 1023) / 1024 : 1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (((reinterpret_cast<uintptr_t>(out) + head) & 3) == 0)
    hipLaunchKernelGGL(s2l::to8b_kernel<true>, dim3(blocks), dim3(256), 0, st, rgb, out, n, head);
  else
    hipLaunchKernelGGL(s2l::to8b_kernel<false>, dim3(blocks), dim3(256), 0, st, rgb, out, n, head);
  return (int)hipGetLastError();
}
