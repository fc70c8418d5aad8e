// Block-wide scans used by the one-workgroup-per-cloud geometry kernels (TPB threads).
#pragma once
#include <hip/hip_runtime.h>

constexpr int TPB = 1024;

// Loads of words that other lanes of the workgroup update with (L2-side) atomics: bypass L1.
__device__ __forceinline__ int ld_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

