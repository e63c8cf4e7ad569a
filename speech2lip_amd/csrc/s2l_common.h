// Shared by every translation unit of libs2l_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

#include "../../include/s2l_hip.h"
#include "s2l_layout.h"

namespace s2l {

typedef float f4 __attribute__((ext_vector_type(4)));

inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

// ---- per-device launch setup, safe to call from several host threads (one thread per GPU is a normal host) ----------
// Every cache below is an array of atomics indexed by the HIP device ordinal: two threads that race on the first call both
// query / set the same value (hipFuncSetAttribute is idempotent), so the worst case is one redundant runtime call.
constexpr int kMaxDevices = 64;

// Current device ordinal and its CU count.  Returns 0, S2L_E_SIZE for an ordinal beyond kMaxDevices, or a hipError_t.
inline int current_device_cus(int* dev_out, int* n_cu_out) {
  static std::atomic<int> n_cu_of[kMaxDevices];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return S2L_E_SIZE;
  int n = n_cu_of[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (n <= 0) return S2L_E_SIZE;
    n_cu_of[dev].store(n, std::memory_order_relaxed);
  }
  *dev_out = dev;
  *n_cu_out = n;
  return 0;
}

// One-time opt-in to more than 64 KiB of dynamic LDS for `kernel` on device `dev`; `done` is that kernel's own flag array.
struct LdsOptIn {
  std::atomic<unsigned char> done[kMaxDevices];
};
inline int ensure_dynamic_lds(const void* kernel, int bytes, LdsOptIn& flags, int dev) {
  if (flags.done[dev].load(std::memory_order_acquire)) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  flags.done[dev].store(1, std::memory_order_release);
  return 0;
}

}  // namespace s2l
