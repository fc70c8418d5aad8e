// Segmentation loss of KPFCNN in three launches, forward and backward together (reference: KPConv-PyTorch/models/architectures.py:345-372 --
// labels outside `valid_labels` become -1, then torch.nn.CrossEntropyLoss(weight=class_w, ignore_index=-1) on the
// logits transposed to (1, C, N); mean over the kept points weighted by the class weights):
//     loss = sum_i w[t_i] * (logsumexp(x_i) - x_i[t_i]) / sum_i w[t_i]            (i over points with t_i >= 0)
//     dx_i = g * w[t_i] / sum_j w[t_j] * (softmax(x_i) - onehot(t_i))             (0 for ignored points)
// As library calls the same is ~12 launches at the turn-around of every step (label lookup x3, transpose copy, 2-D
// log-softmax, nll_loss2d + its size-average kernel, and their backward kernels with a zero fill).
// Forward: one thread per point, workgroup partial sums, then one small workgroup adds the partials in order
// (deterministic; a kernel boundary instead of an in-kernel agent-scope fence, which on the 8 XCDs means L2
// write-backs under the other branches of the step). Backward recomputes the softmax (C is ~20).
#include "common.h"

namespace {

constexpr int XT = 256;        // points per workgroup
constexpr int XC_MAX = 256;    // classes

template <bool L64>
__device__ __forceinline__ int class_of(const void* labels, int64_t i, const int32_t* __restrict__ lut, int lut_n) {
  // architectures.py:352-355: any value that is not a valid label -> -1 (lut[0] serves negatives, lut[lut_n-1] the
  // values above the table)
  int64_t l = L64 ? ((const int64_t*)labels)[i] : (int64_t)((const int32_t*)labels)[i];
  l = l < -1 ? -1 : (l > lut_n - 2 ? lut_n - 2 : l);
  return lut[l + 1];
}

template <bool L64>
__global__ __launch_bounds__(XT) void xent_fwd_k(const float* __restrict__ x, int64_t N, int C, const void* labels,
                                                 const int32_t* __restrict__ lut, int lut_n,
                                                 const float* __restrict__ weight, float* __restrict__ partials) {
  __shared__ float sl[XT / 64], sw[XT / 64];
  const int64_t i = (int64_t)blockIdx.x * XT + threadIdx.x;
  float li = 0.f, wi = 0.f;
  if (i < N) {
    const int t = class_of<L64>(labels, i, lut, lut_n);
    if (t >= 0 && t < C) {
      const float* r = x + i * C;
      float m = r[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, r[c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(r[c] - m);
      wi = weight ? weight[t] : 1.f;
      li = wi * (logf(s) + m - r[t]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    li += __shfl_xor(li, o);
    wi += __shfl_xor(wi, o);
  }
  if ((threadIdx.x & 63) == 0) {
    sl[threadIdx.x >> 6] = li;
    sw[threadIdx.x >> 6] = wi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < XT / 64; ++w) {
      a += sl[w];
      b += sw[w];
    }
    partials[2 * blockIdx.x] = a;
    partials[2 * blockIdx.x + 1] = b;
  }
}

// one wave: lane l adds partials l, l + 64, ... in order, then the 64 lane sums are added in lane order
__global__ __launch_bounds__(64) void xent_finish_k(const float* __restrict__ partials, int nblk, float* __restrict__ out2) {
  __shared__ float sa[64], sb[64];
  float a = 0.f, b = 0.f;
  for (int g = threadIdx.x; g < nblk; g += 64) {
    a += partials[2 * g];
    b += partials[2 * g + 1];
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0.f;
    b = 0.f;
    for (int l = 0; l < 64; ++l) {
      a += sa[l];
      b += sb[l];
    }
    out2[0] = a / b;                     // no kept point: 0 / 0 = NaN, like torch's mean over an empty set
    out2[1] = b;
  }
}

template <bool L64>
__global__ __launch_bounds__(XT) void xent_bwd_k(const float* __restrict__ x, int64_t N, int C, const void* labels,
                                                 const int32_t* __restrict__ lut, int lut_n,
                                                 const float* __restrict__ weight, const float* __restrict__ out2,
                                                 const float* __restrict__ g, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * XT + threadIdx.x;
  if (i >= N) return;
  const int t = class_of<L64>(labels, i, lut, lut_n);
  float* d = dx + i * C;
  if (t < 0 || t >= C) {
    for (int c = 0; c < C; ++c) d[c] = 0.f;
    return;
  }
  const float* r = x + i * C;
  float m = r[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, r[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(r[c] - m);
  const float k = g[0] * (weight ? weight[t] : 1.f) / out2[1], inv = 1.f / s;
  for (int c = 0; c < C; ++c) d[c] = k * (expf(r[c] - m) * inv - (c == t ? 1.f : 0.f));
}

}  // namespace

extern "C" int64_t mvk_xent_workspace_floats(int64_t N) { return 2 * cdiv64(N > 0 ? N : 1, XT); }

extern "C" int mvk_xent_fwd(const float* logits, int64_t N, int C, const void* labels, int labels64,
                            const int32_t* lut, int lut_n, const float* class_weight, float* partials, float* out2,
                            void* stream) {
  MVK_REQUIRE(N >= 0 && C > 0 && C <= XC_MAX && lut_n >= 3, "xent: bad sizes N=%lld C=%d lut=%d", (long long)N, C, lut_n);
  MVK_REQUIRE(lut && partials && out2 && (N == 0 || (logits && labels)), "xent: null operand");
  hipStream_t st = (hipStream_t)stream;
  const unsigned gx = (unsigned)cdiv64(N > 0 ? N : 1, XT);
  if (labels64)
    hipLaunchKernelGGL(xent_fwd_k<true>, dim3(gx), dim3(XT), 0, st, logits, N, C, labels, lut, lut_n, class_weight, partials);
  else
    hipLaunchKernelGGL(xent_fwd_k<false>, dim3(gx), dim3(XT), 0, st, logits, N, C, labels, lut, lut_n, class_weight, partials);
  hipLaunchKernelGGL(xent_finish_k, dim3(1), dim3(64), 0, st, partials, (int)gx, out2);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_xent_bwd(const float* logits, int64_t N, int C, const void* labels, int labels64, const int32_t* lut,
                            int lut_n, const float* class_weight, const float* out2, const float* grad_loss,
                            float* dlogits, void* stream) {
  MVK_REQUIRE(N >= 0 && C > 0 && C <= XC_MAX && lut_n >= 3, "xent: bad sizes");
  if (N == 0) return 0;
  MVK_REQUIRE(logits && labels && lut && out2 && grad_loss && dlogits, "xent: null operand");
  hipStream_t st = (hipStream_t)stream;
  const unsigned gx = (unsigned)cdiv64(N, XT);
  if (labels64)
    hipLaunchKernelGGL(xent_bwd_k<true>, dim3(gx), dim3(XT), 0, st, logits, N, C, labels, lut, lut_n, class_weight, out2,
                       grad_loss, dlogits);
  else
    hipLaunchKernelGGL(xent_bwd_k<false>, dim3(gx), dim3(XT), 0, st, logits, N, C, labels, lut, lut_n, class_weight, out2,
                       grad_loss, dlogits);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
