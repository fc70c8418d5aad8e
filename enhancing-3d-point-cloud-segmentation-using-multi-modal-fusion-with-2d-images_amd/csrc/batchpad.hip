// Capacity padding of one pyramid level for hipGraph replay (no reference counterpart: the reference
// re-allocates every batch; a captured graph needs fixed shapes). One launch per matrix replaces the
// fill / compare / where / slice-copy chain of the tensor library:
//   points   dst[r,:] = r < n ? src[r,:] : fill                     (+ the level's row count word)
//   indices  dst[r,c] = (r < n && c < w_src) ? remap(src[r,c]) : shadow_dst,
//            remap(j) = j == shadow_src ? shadow_dst : j            (shadow = "no neighbour" index)
#include "common.h"

namespace {

__global__ void pad_points_kernel(const float* __restrict__ src, int64_t n, float* __restrict__ dst, int64_t cap,
                                  int w, float fill, int32_t* __restrict__ count_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && count_out) *count_out = (int32_t)n;
  if (t >= cap * w) return;
  dst[t] = t < n * w ? src[t] : fill;
}

template <typename I>
__global__ void pad_index_kernel(const I* __restrict__ src, int64_t n, int w_src, int64_t shadow_src,
                                 I* __restrict__ dst, int64_t cap, int w_dst, int64_t shadow_dst) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cap * w_dst) return;
  const int64_t r = t / w_dst;
  const int c = (int)(t - r * w_dst);
  int64_t v = shadow_dst;
  if (r < n && c < w_src) {
    v = (int64_t)src[r * w_src + c];
    if (v == shadow_src) v = shadow_dst;
  }
  dst[t] = (I)v;
}

}  // namespace

extern "C" int mvk_pad_points(const float* src, int64_t n, float* dst, int64_t cap, int w, float fill,
                              int32_t* count_out, void* stream) {
  MVK_REQUIRE(n >= 0 && cap >= n && w >= 1, "pad_points: %lld rows do not fit the capacity %lld", (long long)n,
              (long long)cap);
  if (cap == 0) return 0;
  hipLaunchKernelGGL(pad_points_kernel, dim3((unsigned)cdiv64(cap * w, 256)), dim3(256), 0, (hipStream_t)stream, src, n,
                     dst, cap, w, fill, count_out);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_pad_index_rows(const void* src, int idx64, int64_t n, int w_src, int64_t shadow_src, void* dst,
                                  int64_t cap, int w_dst, int64_t shadow_dst, void* stream) {
  MVK_REQUIRE(n >= 0 && cap >= n && w_src >= 0 && w_dst >= w_src,
              "pad_index_rows: a %lld x %d matrix does not fit the capacity %lld x %d", (long long)n, w_src,
              (long long)cap, w_dst);
  if (cap == 0 || w_dst == 0) return 0;
  const dim3 grid((unsigned)cdiv64(cap * w_dst, 256));
  if (idx64)
    hipLaunchKernelGGL((pad_index_kernel<int64_t>), grid, dim3(256), 0, (hipStream_t)stream, (const int64_t*)src, n,
                       w_src, shadow_src, (int64_t*)dst, cap, w_dst, shadow_dst);
  else
    hipLaunchKernelGGL((pad_index_kernel<int32_t>), grid, dim3(256), 0, (hipStream_t)stream, (const int32_t*)src, n,
                       w_src, shadow_src, (int32_t*)dst, cap, w_dst, shadow_dst);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
