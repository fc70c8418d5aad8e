// Block-wide scans used by the one-workgroup-per-cloud geometry kernels (TPB threads).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

constexpr int TPB = 1024;

// Loads of words that other lanes of the workgroup update with (L2-side) atomics: bypass L1.
__device__ __forceinline__ int ld_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same loads for arrays in GLOBAL memory with N of them in flight. The compiler waits for every relaxed ATOMIC load
// before it issues the next (one L2 round trip per element: the one-workgroup-per-cloud kernels are chains of such trips);
// these are raw buffer loads with the sc1 cache-policy bit (common.h: park_*), which the compiler counts like any other
// load -- N back to back, one counted wait in front of each use (round 5: no inline assembly). `base` must be
// wave-uniform (an array of the workgroup's cloud), `idx` are element indices into it. NOT for LDS pointers.
template <int N>
__device__ __forceinline__ void ldg_agent(int (&v)[N], const int* base, const int (&idx)[N]) {
  const __amdgpu_buffer_rsrc_t r = park_rsrc(base);
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = park_load1i(r, (uint32_t)idx[u] * 4u);
}
template <int N>
__device__ __forceinline__ void ldg_agent(unsigned long long (&v)[N], const unsigned long long* base, const int (&idx)[N]) {
  const __amdgpu_buffer_rsrc_t r = park_rsrc(base);
#pragma unroll
  for (int u = 0; u < N; ++u) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)idx[u] * 8u, 0, MVK_AUX_SC1);
    v[u] = ((unsigned long long)w[1] << 32) | w[0];
  }
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* sh /* TPB/64 + 1 */) {
  // wave scan + cross-wave
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[w] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < TPB / 64; ++i) {
      int t = sh[i];
      sh[i] = acc;
      acc += t;
    }
    sh[TPB / 64] = acc;
  }
  __syncthreads();
  int res = x - v + sh[w];
  *total = sh[TPB / 64];
  __syncthreads();
  return res;
}

// Exclusive scan of arr[0..n) in place (int), returns total. All threads of the block call it.
__device__ int block_scan_array(int* arr, int n, int* sh, bool suffix) {
  // each thread owns a contiguous chunk
  const int per = (n + TPB - 1) / TPB;
  const int beg = min(n, (int)threadIdx.x * per), end = min(n, beg + per);
  int s = 0;
  if (!suffix) {
    for (int i = beg; i < end; ++i) s += ld_agent(&arr[i]);
  } else {
    for (int i = beg; i < end; ++i) s += ld_agent(&arr[n - 1 - i]);
  }
  int total;
  int base = block_exclusive_scan(s, &total, sh);
  if (!suffix) {
    for (int i = beg; i < end; ++i) {
      int t = ld_agent(&arr[i]);
      arr[i] = base;
      base += t;
    }
  } else {  // arr[j] <- sum of arr[j'] for j' > j
    for (int i = beg; i < end; ++i) {
      int t = ld_agent(&arr[n - 1 - i]);
      arr[n - 1 - i] = base;
      base += t;
    }
  }
  __syncthreads();
  return total;
}



// block_scan_array for an array in GLOBAL memory: the same result, the loads of a thread's chunk eight at a time.
__device__ int block_scan_array_g(int* arr, int n, int* sh, bool suffix) {
  const int per = (n + TPB - 1) / TPB;
  const int beg = min(n, (int)threadIdx.x * per), end = min(n, beg + per);
  int s = 0;
  for (int i0 = beg; i0 < end; i0 += 8) {
    int ai[8];
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u < end ? i0 + u : end - 1;
      ai[u] = suffix ? n - 1 - i : i;
    }
    ldg_agent<8>(v, arr, ai);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += i0 + u < end ? v[u] : 0;
  }
  int total;
  int base = block_exclusive_scan(s, &total, sh);
  for (int i0 = beg; i0 < end; i0 += 8) {
    int ai[8];
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u < end ? i0 + u : end - 1;
      ai[u] = suffix ? n - 1 - i : i;
    }
    ldg_agent<8>(v, arr, ai);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < end) {
        arr[suffix ? n - 1 - (i0 + u) : i0 + u] = base;
        base += v[u];
      }
  }
  __syncthreads();
  return total;
}


// block_scan_array for an array in LDS (the workgroup's own memory: plain loads are coherent after a barrier, and the
// compiler may keep several in flight).
__device__ int block_scan_array_lds(int* arr, int n, int* sh, bool suffix) {
  const int per = (n + TPB - 1) / TPB;
  const int beg = min(n, (int)threadIdx.x * per), end = min(n, beg + per);
  int s = 0;
  for (int i = beg; i < end; ++i) s += arr[suffix ? n - 1 - i : i];
  int total;
  int base = block_exclusive_scan(s, &total, sh);
  for (int i = beg; i < end; ++i) {
    const int j = suffix ? n - 1 - i : i;
    const int t = arr[j];
    arr[j] = base;
    base += t;
  }
  __syncthreads();
  return total;
}
