// Shared by every translation unit of libs2l_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/s2l_hip.h"
#include "s2l_layout.h"

namespace s2l {

typedef float f4 __attribute__((ext_vector_type(4)));

inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace s2l
