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
// leave at zero, for a launch on `stream` (a capturing stream keeps its slices for good, gemm.hip). False: no arena is
// set (the caller keeps its atomic path) or the request cannot be served.
bool mvk_internal_arena_take(void* stream, int64_t floats, int64_t counters, float** ws, int** cnt);

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
// past the XCD-private L2s), so that neither a fence is needed around them (an agent-scope fence writes back / invalidates
// the whole L2 of the XCD) nor relaxed ATOMIC loads (the compiler waits for each before issuing the next: one memory round
// trip per element). Round 5: these are raw BUFFER accesses with the sc1 cache-policy bit (aux bit 4 on gfx94x / gfx950)
// through the compiler's own intrinsics instead of inline assembly -- the compiler counts them (exact vmcnt waits in front
// of every use, its hazard recogniser pads the wide stores), so no register written by a load can be copied, spilled or
// merged at a join before the data has arrived (the hand-placed `s_nop 2` and the pin statements are gone).
// A buffer resource = a wave-uniform base (SGPRs) + a 32-bit byte offset per lane: callers pass the base of their slice.
// park_wait() is still required before the barrier that PUBLISHES stores (the compiler does not know that another
// workgroup will read them): s_waitcnt vmcnt(0) -- stores count on vmcnt on gfx9.
typedef float mvk_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned mvk_u32x4 __attribute__((ext_vector_type(4)));
constexpr int MVK_AUX_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t park_rsrc(const void* base) {
  // raw buffer (stride 0), no range limit, gfx9 data format word
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ void park_store4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, mvk_f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mvk_u32x4, v), r, byte_off, 0, MVK_AUX_SC1);
}
__device__ __forceinline__ void park_store1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, MVK_AUX_SC1);
}
__device__ __forceinline__ mvk_f32x4 park_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(mvk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, MVK_AUX_SC1));
}
__device__ __forceinline__ float park_load1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, MVK_AUX_SC1));
}
__device__ __forceinline__ int park_load1i(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, MVK_AUX_SC1);
}
__device__ __forceinline__ void park_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// neighbour index load: int32 or int64 storage, -1 for shadow / out of range
template <bool IDX64>
__device__ __forceinline__ int load_idx(const void* idx, int64_t pos, int64_t Ns) {
  int64_t j = IDX64 ? ((const int64_t*)idx)[pos] : (int64_t)((const int32_t*)idx)[pos];
  return (j >= 0 && j < Ns) ? (int)j : -1;
}
