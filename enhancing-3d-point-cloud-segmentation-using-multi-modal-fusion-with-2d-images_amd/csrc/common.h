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

// Slices of the arena of ordered reductions (mvk_gemm_split_arena, csrc/gemm.hip) for kernels outside the GEMM that sum
// per-workgroup partials in a fixed order: `floats` of parking space, `counters` zeroed int32 words that the kernel must
// leave at zero. False: no arena is set (the caller keeps its atomic path) or the request does not fit.
bool mvk_internal_arena_take(int64_t floats, int64_t counters, float** ws, int** cnt);

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

// Parking space of the ordered reductions: AGENT-SCOPE loads / stores (gfx950: the sc1 bit -- served at the memory side,
// past the XCD-private L2s) written as inline assembly so that the compiler neither serialises them (it waits for every
// relaxed ATOMIC load before issuing the next: one memory round trip per element) nor needs a fence around them (an
// agent-scope fence writes back / invalidates the whole L2 of the XCD). The compiler does not count these operations:
// park_wait() before the first use of loaded values (and before the barrier that publishes stores), park_pin() on every
// loaded register after the wait so that no use is scheduled above it. Untracked operations only make the compiler's
// own counted waits conservative (memory operations of one kind retire in order).
typedef float mvk_f32x4 __attribute__((ext_vector_type(4)));
// (s_nop: a VALU instruction must not overwrite the data registers of a store wider than 8 bytes within two wait states
// of its issue -- the store reads them late. The compiler's hazard recogniser pads its own stores; it cannot see into
// inline assembly, and without the padding the first two components of lanes 12-15 of every 16 left as the NEXT
// instruction's result: found with tools/park_debug.py)
__device__ __forceinline__ void park_store4(float* p, mvk_f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void park_store1(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void park_load4(mvk_f32x4& v, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void park_load1(float& v, const float* p) {
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void park_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void park_pin(mvk_f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void park_pin(float& v) { asm volatile("" : "+v"(v)); }

// neighbour index load: int32 or int64 storage, -1 for shadow / out of range
template <bool IDX64>
__device__ __forceinline__ int load_idx(const void* idx, int64_t pos, int64_t Ns) {
  int64_t j = IDX64 ? ((const int64_t*)idx)[pos] : (int64_t)((const int32_t*)idx)[pos];
  return (j >= 0 && j < Ns) ? (int)j : -1;
}
