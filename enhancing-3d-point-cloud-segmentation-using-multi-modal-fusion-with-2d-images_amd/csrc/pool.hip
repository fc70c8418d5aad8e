// Row gathers of the KPConv blocks: max_pool / closest_pool (reference
// KPConv-PyTorch/models/blocks.py:35-66 gather, :79-91 closest_pool, :94-110 max_pool)
// fused so that the [N2,H,D] gathered tensor is never materialised.
// One lane per (row n, channel c), c fastest => coalesced row reads and writes.
#include "common.h"

namespace {

template <bool IDX64>
__global__ void max_pool_fwd_k(const float* __restrict__ x, int64_t Ns, int C, const void* idx,
                               int64_t Nq, int H, float* __restrict__ out, int32_t* __restrict__ arg) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Nq * C) return;
  const int64_t n = t / C;
  const int c = (int)(t % C);
  float best = -INFINITY;
  int bh = 0;
  constexpr int UB = 8;      // eight neighbours per trip: indices first, then the eight row loads, then the compares
  int h = 0;                 // (the index -> row chain of one neighbour at a time was latency bound)
  for (; h + UB <= H; h += UB) {
    int j[UB];
    float v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) j[u] = load_idx<IDX64>(idx, n * H + h + u, Ns);
#pragma unroll
    for (int u = 0; u < UB; ++u) v[u] = j[u] >= 0 ? x[(int64_t)j[u] * C + c] : 0.f;  // zero shadow row takes part (blocks.py:103)
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (v[u] > best) {
        best = v[u];
        bh = h + u;
      }
  }
  for (; h < H; ++h) {
    const int j = load_idx<IDX64>(idx, n * H + h, Ns);
    const float v = j >= 0 ? x[(int64_t)j * C + c] : 0.f;
    if (v > best) {
      best = v;
      bh = h;
    }
  }
  out[t] = H > 0 ? best : 0.f;
  if (arg) arg[t] = bh;
}

template <bool IDX64>
__global__ void max_pool_bwd_k(const float* __restrict__ g, const int32_t* __restrict__ arg,
                               const void* idx, int64_t Nq, int H, int64_t Ns, int C,
                               float* __restrict__ dx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Nq * C) return;
  const int64_t n = t / C;
  const int c = (int)(t % C);
  const int j = load_idx<IDX64>(idx, n * H + arg[t], Ns);
  if (j >= 0) atomicAdd(dx + (int64_t)j * C + c, g[t]);
}

template <bool IDX64>
__global__ void gather_rows_fwd_k(const float* __restrict__ x, int64_t Ns, int C, const void* idx,
                                  int64_t Nq, int64_t stride, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Nq * C) return;
  const int64_t n = t / C;
  const int c = (int)(t % C);
  const int j = load_idx<IDX64>(idx, n * stride, Ns);
  out[t] = j >= 0 ? x[(int64_t)j * C + c] : 0.f;
}

// out[n] = [x[idx[n,0]] | skip[n]]: nearest upsampling of the coarse features and the concatenation with the encoder's
// skip features (KPFCNN decoder, architectures.py:334-335) in one launch; rows as float4 where the widths allow
template <bool IDX64, int V>
__global__ void gather_rows_cat_k(const float* __restrict__ x, int64_t Ns, int C1, const void* idx, int64_t Nq,
                                  int64_t stride, const float* __restrict__ skip, int C2, float* __restrict__ out) {
  const int W = (C1 + C2) / V;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Nq * W) return;
  const int64_t n = t / W;
  const int c = (int)(t % W) * V;
  float v[V];
#pragma unroll
  for (int e = 0; e < V; ++e) v[e] = 0.f;
  if (c < C1) {
    const int j = load_idx<IDX64>(idx, n * stride, Ns);
    if (j >= 0) {
      const float* p = x + (int64_t)j * C1 + c;
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = p[e];
    }
  } else {
    const float* p = skip + n * C2 + (c - C1);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = p[e];
  }
  float* o = out + n * (int64_t)(C1 + C2) + c;
#pragma unroll
  for (int e = 0; e < V; ++e) o[e] = v[e];
}

template <bool IDX64>
__global__ void gather_rows_bwd_k(const float* __restrict__ g, int64_t g_ld, const void* idx, int64_t Nq,
                                  int64_t stride, int64_t Ns, int C, float* __restrict__ dx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Nq * C) return;
  const int64_t n = t / C;
  const int c = (int)(t % C);
  const int j = load_idx<IDX64>(idx, n * stride, Ns);
  if (j >= 0) atomicAdd(dx + (int64_t)j * C + c, g[n * g_ld + c]);
}

// backward of gather_rows_cat_k: the upsampled half of g [Nq, C1+C2] scattered into dx [Ns,C1] (f32 atomics, dx zeroed
// by the caller), the skip half copied out as a dense [Nq,C2] tensor (it becomes a gradient others accumulate onto)
template <bool IDX64>
__global__ void gather_rows_cat_bwd_k(const float* __restrict__ g, const void* idx, int64_t Nq, int64_t stride, int64_t Ns,
                                      int C1, int C2, float* __restrict__ dx, float* __restrict__ d_skip) {
  const int W = C1 + C2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Nq * W) return;
  const int64_t n = t / W;
  const int c = (int)(t % W);
  if (c < C1) {
    if (dx == nullptr) return;
    const int j = load_idx<IDX64>(idx, n * stride, Ns);
    if (j >= 0) atomicAdd(dx + (int64_t)j * C1 + c, g[t]);
  } else if (d_skip != nullptr) {
    d_skip[n * C2 + (c - C1)] = g[t];
  }
}

inline dim3 grid1d(int64_t total) { return dim3((unsigned)cdiv64(total, 256)); }

}  // namespace

#define DISPATCH_IDX(kern, ...)                                                   \
  if (idx64)                                                                      \
    hipLaunchKernelGGL((kern<true>), grid1d(total), dim3(256), 0, st, __VA_ARGS__); \
  else                                                                            \
    hipLaunchKernelGGL((kern<false>), grid1d(total), dim3(256), 0, st, __VA_ARGS__)

extern "C" int mvk_max_pool_fwd(const float* x, int64_t Ns, int C, const void* idx, int idx64,
                                int64_t Nq, int H, float* out, int32_t* arg, void* stream) {
  MVK_REQUIRE(C > 0 && Nq >= 0 && H >= 0, "max_pool: bad sizes");
  const int64_t total = Nq * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_IDX(max_pool_fwd_k, x, Ns, C, idx, Nq, H, out, arg);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_max_pool_bwd(const float* g, const int32_t* arg, const void* idx, int idx64,
                                int64_t Nq, int H, int64_t Ns, int C, float* dx, void* stream) {
  const int64_t total = Nq * C;
  if (total == 0 || H == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_IDX(max_pool_bwd_k, g, arg, idx, Nq, H, Ns, C, dx);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_gather_rows_fwd(const float* x, int64_t Ns, int C, const void* idx, int idx64,
                                   int64_t Nq, int64_t idx_stride, float* out, void* stream) {
  const int64_t total = Nq * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_IDX(gather_rows_fwd_k, x, Ns, C, idx, Nq, idx_stride, out);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_gather_rows_cat_fwd(const float* x, int64_t Ns, int C1, const void* idx, int idx64, int64_t Nq,
                                       int64_t idx_stride, const float* skip, int C2, float* out, void* stream) {
  MVK_REQUIRE(C1 > 0 && C2 > 0 && Nq >= 0 && Ns >= 0, "gather_rows_cat: bad sizes");
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = C1 % 4 == 0 && C2 % 4 == 0 &&
                  (((uintptr_t)x | (uintptr_t)skip | (uintptr_t)out) & 15) == 0;
  const int64_t total = Nq * (int64_t)((C1 + C2) / (v4 ? 4 : 1));
  if (v4) {
    if (idx64) hipLaunchKernelGGL((gather_rows_cat_k<true, 4>), grid1d(total), dim3(256), 0, st, x, Ns, C1, idx, Nq, idx_stride, skip, C2, out);
    else hipLaunchKernelGGL((gather_rows_cat_k<false, 4>), grid1d(total), dim3(256), 0, st, x, Ns, C1, idx, Nq, idx_stride, skip, C2, out);
  } else {
    if (idx64) hipLaunchKernelGGL((gather_rows_cat_k<true, 1>), grid1d(total), dim3(256), 0, st, x, Ns, C1, idx, Nq, idx_stride, skip, C2, out);
    else hipLaunchKernelGGL((gather_rows_cat_k<false, 1>), grid1d(total), dim3(256), 0, st, x, Ns, C1, idx, Nq, idx_stride, skip, C2, out);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_gather_rows_cat_bwd(const float* g, const void* idx, int idx64, int64_t Nq, int64_t idx_stride,
                                       int64_t Ns, int C1, int C2, float* dx, float* d_skip, void* stream) {
  MVK_REQUIRE(C1 > 0 && C2 > 0 && Nq >= 0 && Ns >= 0, "gather_rows_cat_bwd: bad sizes");
  const int64_t total = Nq * (int64_t)(C1 + C2);
  if (total == 0 || (dx == nullptr && d_skip == nullptr)) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_IDX(gather_rows_cat_bwd_k, g, idx, Nq, idx_stride, Ns, C1, C2, dx, d_skip);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_gather_rows_bwd_ld(const float* g, int64_t g_ld, const void* idx, int idx64, int64_t Nq,
                                      int64_t idx_stride, int64_t Ns, int C, float* dx, void* stream) {
  MVK_REQUIRE(g_ld >= C, "gather_rows_bwd: row stride of g smaller than its row length");
  const int64_t total = Nq * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_IDX(gather_rows_bwd_k, g, g_ld, idx, Nq, idx_stride, Ns, C, dx);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_gather_rows_bwd(const float* g, const void* idx, int idx64, int64_t Nq,
                                   int64_t idx_stride, int64_t Ns, int C, float* dx, void* stream) {
  return mvk_gather_rows_bwd_ld(g, C, idx, idx64, Nq, idx_stride, Ns, C, dx, stream);
}

// ---- pointwise epilogue of the frozen 2D encoder's convolutions (mvpnet/models/unet_resnet34.py: conv -> BatchNorm
// (eval) -> [+ identity] -> ReLU): with the BatchNorm folded into the convolution weights what is left is
// y = act(x + bias[c] (+ res (+ bias2[c]))) over a channels-last tensor -- one launch instead of three or four.
namespace {
__global__ void bias_act_nhwc_k(const float4* __restrict__ x, const float4* __restrict__ bias, const float4* __restrict__ res,
                                const float4* __restrict__ bias2, float4* __restrict__ y, int64_t n4, int c4, int relu) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n4) return;
  const int c = (int)(t % c4);
  float4 v = x[t];
  const float4 b = bias[c];
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  if (res) {
    const float4 r = res[t];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    if (bias2) {
      const float4 b2 = bias2[c];
      v.x += b2.x; v.y += b2.y; v.z += b2.z; v.w += b2.w;
    }
  }
  if (relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  y[t] = v;
}
}  // namespace

extern "C" int mvk_bias_act_nhwc(const float* x, const float* bias, const float* res, const float* bias2, float* y,
                                 int64_t n_elems, int channels, int relu, void* stream) {
  MVK_REQUIRE(n_elems >= 0 && channels > 0 && channels % 4 == 0 && n_elems % channels == 0,
              "bias_act: the channel count must be a multiple of 4 and divide the element count");
  MVK_REQUIRE(!(bias2 && !res), "bias_act: a second bias belongs to a residual");
  if (n_elems == 0) return 0;
  const int64_t n4 = n_elems / 4;
  hipLaunchKernelGGL(bias_act_nhwc_k, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (const float4*)bias, (const float4*)res, (const float4*)bias2, (float4*)y, n4, channels / 4, relu);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
