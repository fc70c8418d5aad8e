// Shared host-side helpers of libmvkpconv.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mvkpconv.h"

extern "C" void mvk_set_error(const char* fmt, ...);

#define MVK_CHECK_HIP(expr)                                                           \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      mvk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                    __LINE__);                                                        \
      return -2;                                                                      \
    }                                                                                 \
  } while (0)

#define MVK_REQUIRE(cond, ...)   \
  do {                           \
    if (!(cond)) {               \
      mvk_set_error(__VA_ARGS__); \
      return -1;                 \
    }                            \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A pointer the compiler cannot trace to a kernel argument (read from a table in memory) is "generic": its loads
// become flat_load, which also counts on lgkmcnt. Device allocations are global memory: reading the table slot AS a
// global-address-space pointer lets the compiler infer that for every access made through the returned pointer.
template <typename T>
__device__ __forceinline__ T* load_global_ptr(T* const* slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(1))) T* gp;
  return (T*)(*static_cast<const gp*>(static_cast<const void*>(slot)));
#else
  return *slot;
#endif
}

// neighbour index load: int32 or int64 storage, -1 for shadow / out of range
template <bool IDX64>
__device__ __forceinline__ int load_idx(const void* idx, int64_t pos, int64_t Ns) {
  int64_t j = IDX64 ? ((const int64_t*)idx)[pos] : (int64_t)((const int32_t*)idx)[pos];
  return (j >= 0 && j < Ns) ? (int)j : -1;
}
