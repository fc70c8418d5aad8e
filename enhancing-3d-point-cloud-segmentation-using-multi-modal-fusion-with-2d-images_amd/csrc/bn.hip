// Masked training-mode BatchNorm over the stacked point axis, fused with LeakyReLU.
//
// The reference normalises every KPConv / unary output with BatchNorm1d over all points of the
// stacked batch followed by LeakyReLU(0.1) (KPConv-PyTorch/models/blocks.py:430-467, :549-561,
// :621-649). For hipGraph replay the per-level tensors are padded to a fixed row capacity; the
// statistics must then run over the first n_valid rows only, with n_valid read from DEVICE memory
// (it changes from batch to batch while the launch geometry stays fixed). Padded rows are written
// as zeros. Statistics use per-channel shifted sums (shift = first row) so that the variance does
// not suffer the E[x^2] - E[x]^2 cancellation.
//
//   x [R, D] row-major f32, n_valid int32[1] (device), gamma/beta [D]
//   stats  : mean[D], invstd[D]  (+ running_mean / running_var update, unbiased variance)
//   apply  : y = leaky( (x - mean) * invstd * gamma + beta ), rows >= n_valid -> 0
//   bwd    : dgamma, dbeta, dx   (g masked through the LeakyReLU by recomputing the sign)
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BN_ROWS = 64;   // rows per workgroup in the reduction kernels
constexpr int BN_T = 256;

// grid (ceil(D/64), ceil(R/BN_ROWS)); block 256 = 64 channels x 4 row lanes
__global__ __launch_bounds__(BN_T) void bn_stats_partial(const float* __restrict__ x, const int* __restrict__ n_valid,
                                                         int R, int D, float* __restrict__ part /* [nblk,2,D] */) {
  __shared__ float s1[4][64], s2[4][64];
  const int n = min(*n_valid, R);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * BN_ROWS;
  float a = 0.f, b = 0.f;
  if (c < D && r0 < n) {
    const float k = x[c];  // shift: row 0 (n >= 1 here)
    const int r1 = min(r0 + BN_ROWS, n);
    int r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      float xv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) xv[t] = x[(int64_t)(r + 4 * t) * D + c];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v = xv[t] - k;
        a += v;
        b += v * v;
      }
    }
    for (; r < r1; r += 4) {
      const float v = x[(int64_t)r * D + c] - k;
      a += v;
      b += v * v;
    }
  }
  s1[rl][threadIdx.x & 63] = a;
  s2[rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && c < D) {  // every row block writes its slot (zeros past n_valid): no memset, no atomics
    a = (s1[0][threadIdx.x] + s1[1][threadIdx.x]) + (s1[2][threadIdx.x] + s1[3][threadIdx.x]);
    b = (s2[0][threadIdx.x] + s2[1][threadIdx.x]) + (s2[2][threadIdx.x] + s2[3][threadIdx.x]);
    part[((int64_t)blockIdx.y * 2) * D + c] = a;
    part[((int64_t)blockIdx.y * 2 + 1) * D + c] = b;
  }
}

// Sums the per-row-block partials of 64 channels with a 1024-thread block (64 channels x 16 row
// parts; fixed summation tree -> bit-reproducible). Returns the two totals to the lanes with part == 0.
__device__ __forceinline__ bool bn_sum_partials(const float* __restrict__ part, int nblk, int D, float* a, float* b) {
  __shared__ float r1[16][64], r2[16][64];
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float x = 0.f, y = 0.f;
  if (c < D) {
    // four blocks per trip, all eight loads issued before the first add: the partials come from L2 and a chain of
    // dependent loads was most of this kernel's time (same summation order as one block per trip)
    int i = pr;
    for (; i + 48 < nblk; i += 64) {
      float u[4], v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        u[t] = part[((int64_t)(i + 16 * t) * 2) * D + c];
        v[t] = part[((int64_t)(i + 16 * t) * 2 + 1) * D + c];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        x += u[t];
        y += v[t];
      }
    }
    for (; i < nblk; i += 16) {
      x += part[((int64_t)i * 2) * D + c];
      y += part[((int64_t)i * 2 + 1) * D + c];
    }
  }
  r1[pr][cl] = x;
  r2[pr][cl] = y;
  __syncthreads();
  if (pr != 0 || c >= D) return false;
  x = 0.f;
  y = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x += r1[i][cl];
    y += r2[i][cl];
  }
  *a = x;
  *b = y;
  return true;
}

// The same for partials written by the GEMM epilogue (csrc/gemm.hip): per row block of `blk` rows the plain
// column sum and the sum of squares about the block's own mean. ONE fixed-order pass with batched loads: about the
// mean c of block 0 (every block's mean is close to it, so nothing cancels),
//   S = sum_b sum_b,   Q = sum_b [ M2_b + n_b (mean_b - c)^2 ],   mean = S / n,   M2 = Q - n (mean - c)^2.
// Returns (mean, M2) to the lanes with part == 0.
__device__ __forceinline__ bool bn_sum_partials_m2(const float* __restrict__ part, int R, int blk, int n, int D, float* mean,
                                                   float* m2) {
  __shared__ float q1[16][64], q2[16][64];
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int nb = n > 0 ? (n + blk - 1) / blk : 0;          // blocks that hold valid rows
  float x = 0.f, y = 0.f, ref = 0.f;
  if (c < D && nb > 0) {
    const int n0 = min(blk, n);
    ref = part[c] / (float)n0;
    int i = pr;
    for (; i + 48 < nb; i += 64) {
      float u[4], v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        u[t] = part[((int64_t)(i + 16 * t) * 2) * D + c];
        v[t] = part[((int64_t)(i + 16 * t) * 2 + 1) * D + c];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ni = min(blk, n - (i + 16 * t) * blk);
        const float d = u[t] / (float)ni - ref;
        x += u[t];
        y += v[t] + (float)ni * d * d;
      }
    }
    for (; i < nb; i += 16) {
      const int ni = min(blk, n - i * blk);
      const float u = part[((int64_t)i * 2) * D + c], v = part[((int64_t)i * 2 + 1) * D + c];
      const float d = u / (float)ni - ref;
      x += u;
      y += v + (float)ni * d * d;
    }
  }
  q1[pr][cl] = x;
  q2[pr][cl] = y;
  __syncthreads();
  if (pr != 0 || c >= D) return false;
  x = 0.f;
  y = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x += q1[i][cl];
    y += q2[i][cl];
  }
  const float mu = n > 0 ? x / (float)n : 0.f;
  const float dm = mu - ref;
  float M2 = y - (float)n * dm * dm;
  *mean = mu;
  *m2 = M2 > 0.f ? M2 : 0.f;
  return true;
}

// grid ceil(D/64), block 1024
__global__ __launch_bounds__(1024) void bn_stats_finish(const float* __restrict__ x, const int* __restrict__ n_valid,
                                                        int R, int D, const float* __restrict__ part, float eps,
                                                        float momentum, float* __restrict__ mean,
                                                        float* __restrict__ invstd, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, long long* __restrict__ nbt) {
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;   // num_batches_tracked (nn.BatchNorm bookkeeping)
  float p1, p2;
  if (!bn_sum_partials(part, (R + BN_ROWS - 1) / BN_ROWS, D, &p1, &p2)) return;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int n = min(*n_valid, R);
  if (n < 1) {
    mean[c] = 0.f;
    invstd[c] = 0.f;
    return;
  }
  const float k = x[c];
  const float m1 = p1 / (float)n, m2 = p2 / (float)n;
  const float mu = k + m1;
  float var = m2 - m1 * m1;  // biased
  var = var > 0.f ? var : 0.f;
  mean[c] = mu;
  invstd[c] = rsqrtf(var + eps);
  if (running_mean) {
    const float unbiased = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

__global__ void bn_apply(const float* __restrict__ x, const int* __restrict__ n_valid, int R, int D,
                         const float* __restrict__ mean, const float* __restrict__ invstd,
                         const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                         const float* __restrict__ addend, float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)R * D) return;
  const int n = min(*n_valid, R);
  const int c = (int)(t % D);
  const int64_t r = t / D;
  float v = 0.f;
  if (r < n) {
    v = (x[t] - mean[c]) * invstd[c] * gamma[c] + beta[c];
    if (addend) v += addend[t];   // residual join (blocks.py:649) fused with its LeakyReLU
    v = v > 0.f ? v : v * slope;
  }
  y[t] = v;
}

// dbeta[c] = sum g', dgamma[c] = sum g' * xhat   (g' = g through the LeakyReLU)
__device__ __forceinline__ void bn_bwd_reduce_body(const float* __restrict__ x, const float* __restrict__ g,
                                                      const int* __restrict__ n_valid, int R, int D,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float slope, const float* __restrict__ yout,
                                                      float* __restrict__ part /* [nblk,2,D] */) {
  // yout != NULL: the forward added a residual before the LeakyReLU, so its sign comes from the saved output
  __shared__ float s1[4][64], s2[4][64];
  const int n = min(*n_valid, R);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * BN_ROWS;
  float a = 0.f, b = 0.f;
  if (c < D && r0 < n) {
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    const int r1 = min(r0 + BN_ROWS, n);
    int r = r0 + rl;
    for (; r + 12 < r1; r += 16) {        // four rows per trip, loads first (same summation order)
      float xv[4], gv4[4], yv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        xv[t] = x[(int64_t)(r + 4 * t) * D + c];
        gv4[t] = g[(int64_t)(r + 4 * t) * D + c];
        yv[t] = yout ? yout[(int64_t)(r + 4 * t) * D + c] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float xh = (xv[t] - mu) * is;
        float gv = gv4[t];
        if (yout ? yv[t] <= 0.f : xh * ga + be <= 0.f) gv *= slope;
        a += gv;
        b += gv * xh;
      }
    }
    for (; r < r1; r += 4) {
      const float xh = (x[(int64_t)r * D + c] - mu) * is;
      float gv = g[(int64_t)r * D + c];
      if (yout ? yout[(int64_t)r * D + c] <= 0.f : xh * ga + be <= 0.f) gv *= slope;
      a += gv;
      b += gv * xh;
    }
  }
  s1[rl][threadIdx.x & 63] = a;
  s2[rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && c < D) {
    a = (s1[0][threadIdx.x] + s1[1][threadIdx.x]) + (s1[2][threadIdx.x] + s1[3][threadIdx.x]);
    b = (s2[0][threadIdx.x] + s2[1][threadIdx.x]) + (s2[2][threadIdx.x] + s2[3][threadIdx.x]);
    part[((int64_t)blockIdx.y * 2) * D + c] = a;
    part[((int64_t)blockIdx.y * 2 + 1) * D + c] = b;
  }
}

__global__ __launch_bounds__(BN_T) void bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ g,
                                                      const int* __restrict__ n_valid, int R, int D,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float slope, const float* __restrict__ yout,
                                                      float* __restrict__ part /* [nblk,2,D] */) {
  bn_bwd_reduce_body(x, g, n_valid, R, D, mean, invstd, gamma, beta, slope, yout, part);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_bwd_reduce_args {
  const float* x;
  const float* g;
  const int* n_valid;
  int R;
  int D;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float slope;
  const float* yout;
  float* part;
};
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_pair(bn_bwd_reduce_args a0, bn_bwd_reduce_args a1) {
  const bn_bwd_reduce_args& a = blockIdx.z ? a1 : a0;
  bn_bwd_reduce_body(a.x, a.g, a.n_valid, a.R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.yout, a.part);
}

__global__ __launch_bounds__(1024) void bn_bwd_finish(const float* __restrict__ part, int R, int D,
                                                      float* __restrict__ dgb /* [2,D] */) {
  float a, b;
  if (!bn_sum_partials(part, (R + BN_ROWS - 1) / BN_ROWS, D, &a, &b)) return;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  dgb[c] = a;
  dgb[D + c] = b;
}

__global__ void bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ g,
                             const int* __restrict__ n_valid, int R, int D, const float* __restrict__ mean,
                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float slope, const float* __restrict__ part,
                             const float* __restrict__ yout, float* __restrict__ d_addend, float* __restrict__ dx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)R * D) return;
  const int n = min(*n_valid, R);
  const int c = (int)(t % D);
  const int64_t r = t / D;
  float v = 0.f, ga_out = 0.f;
  if (r < n) {
    const float xh = (x[t] - mean[c]) * invstd[c];
    float gv = g[t];
    if (yout ? yout[t] <= 0.f : xh * gamma[c] + beta[c] <= 0.f) gv *= slope;
    const float inv_n = 1.f / (float)n;
    v = gamma[c] * invstd[c] * (gv - part[c] * inv_n - xh * part[D + c] * inv_n);
    ga_out = gv;
  }
  dx[t] = v;
  if (d_addend) d_addend[t] = ga_out;
}

// ---- finish + apply in one launch: every workgroup of the apply grid first reduces the per-row-block
// partials of its 64 channels itself (same fixed order everywhere -> every workgroup holds the same
// mean / invstd bit for bit; nblk x 512 bytes from L2, at most 64 workgroups per channel group do it), then
// normalises its share of the rows. 3 graph nodes -> 2 per BatchNorm pass; the first row-group publishes
// mean / invstd (saved for the backward), the running statistics and the batch counter.
//   grid (ceil(D/64), gy <= 64), block 1024 = 64 channels x 16 row lanes; rows are dealt round-robin.
constexpr int BN_FUSED_GY_DEFAULT = 128;   // measured: 5.62 ms per step at 64 row groups, 5.53 at 128, 5.51 at 256
int bn_fused_gy() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MVK_BN_FUSED_GY");
    v = e ? atoi(e) : BN_FUSED_GY_DEFAULT;
    if (v < 1) v = 1;
  }
  return v;
}

__device__ __forceinline__ void bn_finish_apply_body(const float* __restrict__ x, const int* __restrict__ n_valid, int R,
                                                        int D, const float* __restrict__ part, float eps, float momentum,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float slope, float* __restrict__ mean, float* __restrict__ invstd,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        long long* __restrict__ nbt, const float* __restrict__ addend,
                                                        float* __restrict__ y, int ext_rows) {
  // ext_rows == 0: `part` holds the shifted sums of bn_stats_partial; > 0: the (sum, centred M2) partials of the
  // GEMM epilogue, one slot per ext_rows rows
  __shared__ float smu[64], sis[64];
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int n = min(*n_valid, R);
  if (ext_rows < 0) {
    // statistics FINISHED by the producing GEMM (mvk_bn_finish, csrc/gemm.hip): mean / invstd are inputs, the running
    // statistics and the batch counter have been updated there -- nothing to reduce, nothing to publish
    if (pr == 0 && c < D) {
      smu[cl] = mean[c];
      sis[cl] = invstd[c];
    }
  }
  float p1, p2;
  const bool owner = ext_rows < 0 ? false
                     : ext_rows > 0 ? bn_sum_partials_m2(part, R, ext_rows, n, D, &p1, &p2)
                                    : bn_sum_partials(part, (R + BN_ROWS - 1) / BN_ROWS, D, &p1, &p2);   // lanes pr == 0, c < D
  if (owner) {
    float mu = 0.f, is = 0.f, var = 0.f;
    if (n > 0 && ext_rows > 0) {
      mu = p1;
      var = p2 / (float)n;  // biased
      is = rsqrtf(var + eps);
    } else if (n > 0) {
      const float k = x[c];
      const float m1 = p1 / (float)n, m2 = p2 / (float)n;
      mu = k + m1;
      var = m2 - m1 * m1;  // biased
      var = var > 0.f ? var : 0.f;
      is = rsqrtf(var + eps);
    }
    smu[cl] = mu;
    sis[cl] = is;
    if (blockIdx.y == 0) {
      mean[c] = mu;
      invstd[c] = is;
      if (running_mean && n > 0) {
        const float unbiased = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  }
  if (nbt && ext_rows >= 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *nbt += 1;
  __syncthreads();
  if (c >= D) return;
  const float mu = smu[cl], is = sis[cl], ga = gamma[c], be = beta[c];
  const int step = gridDim.y * 16;
  int r = blockIdx.y * 16 + pr;
  for (; r + 3 * step < R; r += 4 * step) {      // four rows per trip, loads first
    float xv[4], av[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rr = r + t * step;
      xv[t] = rr < n ? x[(int64_t)rr * D + c] : 0.f;
      av[t] = (addend && rr < n) ? addend[(int64_t)rr * D + c] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rr = r + t * step;
      float v = 0.f;
      if (rr < n) {
        v = (xv[t] - mu) * is * ga + be + av[t];
        v = v > 0.f ? v : v * slope;
      }
      y[(int64_t)rr * D + c] = v;
    }
  }
  for (; r < R; r += step) {
    float v = 0.f;
    if (r < n) {
      v = (x[(int64_t)r * D + c] - mu) * is * ga + be;
      if (addend) v += addend[(int64_t)r * D + c];
      v = v > 0.f ? v : v * slope;
    }
    y[(int64_t)r * D + c] = v;
  }
}

__global__ __launch_bounds__(1024) void bn_finish_apply(const float* __restrict__ x, const int* __restrict__ n_valid, int R,
                                                        int D, const float* __restrict__ part, float eps, float momentum,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float slope, float* __restrict__ mean, float* __restrict__ invstd,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        long long* __restrict__ nbt, const float* __restrict__ addend,
                                                        float* __restrict__ y, int ext_rows) {
  bn_finish_apply_body(x, n_valid, R, D, part, eps, momentum, gamma, beta, slope, mean, invstd, running_mean, running_var, nbt, addend, y, ext_rows);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_finish_apply_args {
  const float* x;
  const int* n_valid;
  int R;
  int D;
  const float* part;
  float eps;
  float momentum;
  const float* gamma;
  const float* beta;
  float slope;
  float* mean;
  float* invstd;
  float* running_mean;
  float* running_var;
  long long* nbt;
  const float* addend;
  float* y;
  int ext_rows;
};
__global__ __launch_bounds__(1024) void bn_finish_apply_pair(bn_finish_apply_args a0, bn_finish_apply_args a1) {
  const bn_finish_apply_args& a = blockIdx.z ? a1 : a0;
  bn_finish_apply_body(a.x, a.n_valid, a.R, a.D, a.part, a.eps, a.momentum, a.gamma, a.beta, a.slope, a.mean, a.invstd, a.running_mean, a.running_var, a.nbt, a.addend, a.y, a.ext_rows);
}

__device__ __forceinline__ void bn_bwd_finish_apply_body(const float* __restrict__ x, const float* __restrict__ g,
                                                            const int* __restrict__ n_valid, int R, int D,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float slope, const float* __restrict__ part,
                                                            const float* __restrict__ yout, float* __restrict__ d_addend,
                                                            float* __restrict__ dgb, float* __restrict__ dx) {
  __shared__ float sa[64], sb[64];
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int n = min(*n_valid, R);
  float a, b;
  if (bn_sum_partials(part, (R + BN_ROWS - 1) / BN_ROWS, D, &a, &b)) {
    sa[cl] = a;
    sb[cl] = b;
    if (blockIdx.y == 0) {
      dgb[c] = a;
      dgb[D + c] = b;
    }
  }
  __syncthreads();
  if (c >= D) return;
  const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c], s1 = sa[cl], s2 = sb[cl];
  const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
  const int step = gridDim.y * 16;
  int r = blockIdx.y * 16 + pr;
  for (; r + 3 * step < R; r += 4 * step) {      // four rows per trip, loads first
    float xv[4], gv4[4], yv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rr = r + t * step;
      xv[t] = rr < n ? x[(int64_t)rr * D + c] : 0.f;
      gv4[t] = rr < n ? g[(int64_t)rr * D + c] : 0.f;
      yv[t] = (yout && rr < n) ? yout[(int64_t)rr * D + c] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rr = r + t * step;
      float v = 0.f, ga_out = 0.f;
      if (rr < n) {
        const float xh = (xv[t] - mu) * is;
        float gv = gv4[t];
        if (yout ? yv[t] <= 0.f : xh * ga + be <= 0.f) gv *= slope;
        v = ga * is * (gv - s1 * inv_n - xh * s2 * inv_n);
        ga_out = gv;
      }
      dx[(int64_t)rr * D + c] = v;
      if (d_addend) d_addend[(int64_t)rr * D + c] = ga_out;
    }
  }
  for (; r < R; r += step) {
    float v = 0.f, ga_out = 0.f;
    if (r < n) {
      const float xh = (x[(int64_t)r * D + c] - mu) * is;
      float gv = g[(int64_t)r * D + c];
      if (yout ? yout[(int64_t)r * D + c] <= 0.f : xh * ga + be <= 0.f) gv *= slope;
      v = ga * is * (gv - s1 * inv_n - xh * s2 * inv_n);
      ga_out = gv;
    }
    dx[(int64_t)r * D + c] = v;
    if (d_addend) d_addend[(int64_t)r * D + c] = ga_out;
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_finish_apply(const float* __restrict__ x, const float* __restrict__ g,
                                                            const int* __restrict__ n_valid, int R, int D,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float slope, const float* __restrict__ part,
                                                            const float* __restrict__ yout, float* __restrict__ d_addend,
                                                            float* __restrict__ dgb, float* __restrict__ dx) {
  bn_bwd_finish_apply_body(x, g, n_valid, R, D, mean, invstd, gamma, beta, slope, part, yout, d_addend, dgb, dx);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_bwd_finish_apply_args {
  const float* x;
  const float* g;
  const int* n_valid;
  int R;
  int D;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float slope;
  const float* part;
  const float* yout;
  float* d_addend;
  float* dgb;
  float* dx;
};
__global__ __launch_bounds__(1024) void bn_bwd_finish_apply_pair(bn_bwd_finish_apply_args a0, bn_bwd_finish_apply_args a1) {
  const bn_bwd_finish_apply_args& a = blockIdx.z ? a1 : a0;
  bn_bwd_finish_apply_body(a.x, a.g, a.n_valid, a.R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.part, a.yout, a.d_addend, a.dgb, a.dx);
}

bool bn_fused_finish() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MVK_BN_FUSED_FINISH");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// ---- small tensors (coarse pyramid levels): one workgroup per 64 channels does statistics AND
// normalisation in a single launch (the rows are re-read from L2), 3 kernel nodes -> 1.
constexpr int BN_SMALL_ROWS_DEFAULT = 128;   // measured: above ~128 rows three parallel launches beat one workgroup per 64 channels

__device__ __forceinline__ void block_sum2(float& a, float& b, float (*r1)[64], float (*r2)[64]) {
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  r1[pr][cl] = a;
  r2[pr][cl] = b;
  __syncthreads();
  float x = 0.f, y = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x += r1[i][cl];
    y += r2[i][cl];
  }
  a = x;
  b = y;
  __syncthreads();
}

// grid ceil(D/64), block 1024 = 64 channels x 16 row lanes
__device__ __forceinline__ void bn_small_fwd_body(const float* __restrict__ x, const int* __restrict__ n_valid, int R,
                                                     int D, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float momentum,
                                                     float slope, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float* __restrict__ mean,
                                                     float* __restrict__ invstd, float* __restrict__ y,
                                                     long long* __restrict__ nbt, const float* __restrict__ addend) {
  __shared__ float r1[16][64], r2[16][64];
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  const int n = min(*n_valid, R);
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool on = c < D;
  const float k = (on && n > 0) ? x[c] : 0.f;
  float a = 0.f, b = 0.f;
  if (on)
    for (int r = pr; r < n; r += 16) {
      const float v = x[(int64_t)r * D + c] - k;
      a += v;
      b += v * v;
    }
  block_sum2(a, b, r1, r2);
  float mu = 0.f, is = 0.f;
  if (n > 0) {
    const float m1 = a / (float)n, m2 = b / (float)n;
    mu = k + m1;
    float var = m2 - m1 * m1;
    var = var > 0.f ? var : 0.f;
    is = rsqrtf(var + eps);
    if (on && pr == 0) {
      mean[c] = mu;
      invstd[c] = is;
      if (running_mean) {
        const float unbiased = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  } else if (on && pr == 0) {
    mean[c] = 0.f;
    invstd[c] = 0.f;
  }
  if (on) {
    const float ga = gamma[c], be = beta[c];
    for (int r = pr; r < R; r += 16) {
      float v = 0.f;
      if (r < n) {
        v = (x[(int64_t)r * D + c] - mu) * is * ga + be;
        if (addend) v += addend[(int64_t)r * D + c];
        v = v > 0.f ? v : v * slope;
      }
      y[(int64_t)r * D + c] = v;
    }
  }
}

__global__ __launch_bounds__(1024) void bn_small_fwd(const float* __restrict__ x, const int* __restrict__ n_valid, int R,
                                                     int D, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float momentum,
                                                     float slope, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float* __restrict__ mean,
                                                     float* __restrict__ invstd, float* __restrict__ y,
                                                     long long* __restrict__ nbt, const float* __restrict__ addend) {
  bn_small_fwd_body(x, n_valid, R, D, gamma, beta, eps, momentum, slope, running_mean, running_var, mean, invstd, y, nbt, addend);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_small_fwd_args {
  const float* x;
  const int* n_valid;
  int R;
  int D;
  const float* gamma;
  const float* beta;
  float eps;
  float momentum;
  float slope;
  float* running_mean;
  float* running_var;
  float* mean;
  float* invstd;
  float* y;
  long long* nbt;
  const float* addend;
};
__global__ __launch_bounds__(1024) void bn_small_fwd_pair(bn_small_fwd_args a0, bn_small_fwd_args a1) {
  const bn_small_fwd_args& a = blockIdx.z ? a1 : a0;
  bn_small_fwd_body(a.x, a.n_valid, a.R, a.D, a.gamma, a.beta, a.eps, a.momentum, a.slope, a.running_mean, a.running_var, a.mean, a.invstd, a.y, a.nbt, a.addend);
}

__device__ __forceinline__ void bn_small_bwd_body(const float* __restrict__ x, const float* __restrict__ g,
                                                     const int* __restrict__ n_valid, int R, int D,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float slope, const float* __restrict__ yout,
                                                     float* __restrict__ d_addend, float* __restrict__ dgb,
                                                     float* __restrict__ dx) {
  __shared__ float r1[16][64], r2[16][64];
  const int n = min(*n_valid, R);
  const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool on = c < D;
  const float mu = on ? mean[c] : 0.f, is = on ? invstd[c] : 0.f, ga = on ? gamma[c] : 0.f, be = on ? beta[c] : 0.f;
  float a = 0.f, b = 0.f;
  if (on)
    for (int r = pr; r < n; r += 16) {
      const float xh = (x[(int64_t)r * D + c] - mu) * is;
      float gv = g[(int64_t)r * D + c];
      if (yout ? yout[(int64_t)r * D + c] <= 0.f : xh * ga + be <= 0.f) gv *= slope;
      a += gv;
      b += gv * xh;
    }
  block_sum2(a, b, r1, r2);
  if (on) {
    if (pr == 0) {
      dgb[c] = a;
      dgb[D + c] = b;
    }
    const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
    for (int r = pr; r < R; r += 16) {
      float v = 0.f, ga_out = 0.f;
      if (r < n) {
        const float xh = (x[(int64_t)r * D + c] - mu) * is;
        float gv = g[(int64_t)r * D + c];
        if (yout ? yout[(int64_t)r * D + c] <= 0.f : xh * ga + be <= 0.f) gv *= slope;
        v = ga * is * (gv - a * inv_n - xh * b * inv_n);
        ga_out = gv;
      }
      dx[(int64_t)r * D + c] = v;
      if (d_addend) d_addend[(int64_t)r * D + c] = ga_out;
    }
  }
}

__global__ __launch_bounds__(1024) void bn_small_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                     const int* __restrict__ n_valid, int R, int D,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float slope, const float* __restrict__ yout,
                                                     float* __restrict__ d_addend, float* __restrict__ dgb,
                                                     float* __restrict__ dx) {
  bn_small_bwd_body(x, g, n_valid, R, D, mean, invstd, gamma, beta, slope, yout, d_addend, dgb, dx);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_small_bwd_args {
  const float* x;
  const float* g;
  const int* n_valid;
  int R;
  int D;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float slope;
  const float* yout;
  float* d_addend;
  float* dgb;
  float* dx;
};
__global__ __launch_bounds__(1024) void bn_small_bwd_pair(bn_small_bwd_args a0, bn_small_bwd_args a1) {
  const bn_small_bwd_args& a = blockIdx.z ? a1 : a0;
  bn_small_bwd_body(a.x, a.g, a.n_valid, a.R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.yout, a.d_addend, a.dgb, a.dx);
}

// ---- the same single-launch BatchNorm for 129..1024 rows and D % 4 == 0, latency-shaped: a workgroup owns 16
// channels (4 quads x 256 row lanes), a thread holds ALL its rows (at most 4) of x (and g, y) as float4 in
// registers -- one round of loads, a reduction over the 256 row lanes (wave shuffles, then 16 partials through LDS),
// one round of stores. The two-launch path costs a second kernel node (~8 us in a graph) and a re-read of x.
constexpr int BN_MID_ROWS_DEFAULT = 1024;
constexpr int BN_MID_MAX = 1024;      // 4 rows per thread x 256 row lanes

// dflt unless cond, then the 16 bytes at p (written as a branch: `cond ? *p : dflt` on a struct becomes a select of
// POINTERS -- the default spilled to scratch and a flat load)
__device__ __forceinline__ float4 ld4_if(bool cond, const float* p, float4 dflt) {
  float4 v = dflt;
  if (cond) v = *reinterpret_cast<const float4*>(p);
  return v;
}

__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// sum of a and of b over the 256 row lanes of the thread's channel quad (tid & 3), returned to every thread
__device__ __forceinline__ void mid_sum(float4& a, float4& b, float4 (*ra)[4], float4 (*rb)[4]) {
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) {
    a.x += __shfl_xor(a.x, m); a.y += __shfl_xor(a.y, m); a.z += __shfl_xor(a.z, m); a.w += __shfl_xor(a.w, m);
    b.x += __shfl_xor(b.x, m); b.y += __shfl_xor(b.y, m); b.z += __shfl_xor(b.z, m); b.w += __shfl_xor(b.w, m);
  }
  const int cq = threadIdx.x & 3, wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 4) {
    ra[wv][cq] = a;
    rb[wv][cq] = b;
  }
  __syncthreads();
  float4 x = ra[0][cq], y = rb[0][cq];
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    add4(x, ra[i][cq]);
    add4(y, rb[i][cq]);
  }
  a = x;
  b = y;
}

// grid ceil(D/16), block 1024 = 4 channel quads x 256 row lanes; R <= BN_MID_MAX
__device__ __forceinline__ void bn_mid_fwd_body(const float* __restrict__ x, const int* __restrict__ n_valid, int R,
                                                   int D, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, float momentum,
                                                   float slope, float* __restrict__ running_mean,
                                                   float* __restrict__ running_var, float* __restrict__ mean,
                                                   float* __restrict__ invstd, float* __restrict__ y,
                                                   long long* __restrict__ nbt, const float* __restrict__ addend) {
  __shared__ float4 ra[16][4], rb[16][4];
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  const int n = min(*n_valid, R);
  const int cq = threadIdx.x & 3, pr = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + cq * 4;
  const bool on = c < D;                                        // D % 4 == 0: the whole quad is in range
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 k = ld4_if(on && n > 0, x + c, z4);      // shift: row 0
  float4 v[4], ad[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = pr + 256 * u;
    v[u] = ld4_if(on && r < n, x + (int64_t)r * D + c, k);
    ad[u] = ld4_if(on && addend && r < n, addend + (int64_t)r * D + c, z4);
  }
  float4 a = z4, b = z4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float dx = v[u].x - k.x, dy = v[u].y - k.y, dz = v[u].z - k.z, dw = v[u].w - k.w;   // rows >= n hold k: 0
    a.x += dx; a.y += dy; a.z += dz; a.w += dw;
    b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
  }
  mid_sum(a, b, ra, rb);
  float4 mu = z4, is = z4, var = z4;
  if (n > 0) {
    const float inv_n = 1.0f / (float)n;
    const float4 m1 = make_float4(a.x / (float)n, a.y / (float)n, a.z / (float)n, a.w / (float)n);
    const float4 m2 = make_float4(b.x / (float)n, b.y / (float)n, b.z / (float)n, b.w / (float)n);
    (void)inv_n;
    mu = make_float4(k.x + m1.x, k.y + m1.y, k.z + m1.z, k.w + m1.w);
    var = make_float4(fmaxf(m2.x - m1.x * m1.x, 0.f), fmaxf(m2.y - m1.y * m1.y, 0.f), fmaxf(m2.z - m1.z * m1.z, 0.f),
                      fmaxf(m2.w - m1.w * m1.w, 0.f));
    is = make_float4(rsqrtf(var.x + eps), rsqrtf(var.y + eps), rsqrtf(var.z + eps), rsqrtf(var.w + eps));
  }
  if (on && pr == 0) {
    *reinterpret_cast<float4*>(mean + c) = mu;
    *reinterpret_cast<float4*>(invstd + c) = is;
    if (running_mean && n > 0) {
      const float ub = n > 1 ? (float)n / (float)(n - 1) : 1.f;
      float4 rm = *reinterpret_cast<float4*>(running_mean + c), rv = *reinterpret_cast<float4*>(running_var + c);
      rm.x = (1.f - momentum) * rm.x + momentum * mu.x;
      rm.y = (1.f - momentum) * rm.y + momentum * mu.y;
      rm.z = (1.f - momentum) * rm.z + momentum * mu.z;
      rm.w = (1.f - momentum) * rm.w + momentum * mu.w;
      rv.x = (1.f - momentum) * rv.x + momentum * (n > 1 ? var.x * ub : var.x);
      rv.y = (1.f - momentum) * rv.y + momentum * (n > 1 ? var.y * ub : var.y);
      rv.z = (1.f - momentum) * rv.z + momentum * (n > 1 ? var.z * ub : var.z);
      rv.w = (1.f - momentum) * rv.w + momentum * (n > 1 ? var.w * ub : var.w);
      *reinterpret_cast<float4*>(running_mean + c) = rm;
      *reinterpret_cast<float4*>(running_var + c) = rv;
    }
  }
  if (on) {
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = pr + 256 * u;
      if (r >= R) continue;
      float4 o = z4;
      if (r < n) {
        o.x = (v[u].x - mu.x) * is.x * ga.x + be.x;
        o.y = (v[u].y - mu.y) * is.y * ga.y + be.y;
        o.z = (v[u].z - mu.z) * is.z * ga.z + be.z;
        o.w = (v[u].w - mu.w) * is.w * ga.w + be.w;
        add4(o, ad[u]);
        o.x = o.x > 0.f ? o.x : o.x * slope;
        o.y = o.y > 0.f ? o.y : o.y * slope;
        o.z = o.z > 0.f ? o.z : o.z * slope;
        o.w = o.w > 0.f ? o.w : o.w * slope;
      }
      *reinterpret_cast<float4*>(y + (int64_t)r * D + c) = o;
    }
  }
}

__global__ __launch_bounds__(1024) void bn_mid_fwd(const float* __restrict__ x, const int* __restrict__ n_valid, int R,
                                                   int D, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, float momentum,
                                                   float slope, float* __restrict__ running_mean,
                                                   float* __restrict__ running_var, float* __restrict__ mean,
                                                   float* __restrict__ invstd, float* __restrict__ y,
                                                   long long* __restrict__ nbt, const float* __restrict__ addend) {
  bn_mid_fwd_body(x, n_valid, R, D, gamma, beta, eps, momentum, slope, running_mean, running_var, mean, invstd, y, nbt, addend);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_mid_fwd_args {
  const float* x;
  const int* n_valid;
  int R;
  int D;
  const float* gamma;
  const float* beta;
  float eps;
  float momentum;
  float slope;
  float* running_mean;
  float* running_var;
  float* mean;
  float* invstd;
  float* y;
  long long* nbt;
  const float* addend;
};
__global__ __launch_bounds__(1024) void bn_mid_fwd_pair(bn_mid_fwd_args a0, bn_mid_fwd_args a1) {
  const bn_mid_fwd_args& a = blockIdx.z ? a1 : a0;
  bn_mid_fwd_body(a.x, a.n_valid, a.R, a.D, a.gamma, a.beta, a.eps, a.momentum, a.slope, a.running_mean, a.running_var, a.mean, a.invstd, a.y, a.nbt, a.addend);
}

__device__ __forceinline__ void bn_mid_bwd_body(const float* __restrict__ x, const float* __restrict__ g,
                                                   const int* __restrict__ n_valid, int R, int D,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float slope, const float* __restrict__ yout,
                                                   float* __restrict__ d_addend, float* __restrict__ dgb,
                                                   float* __restrict__ dx) {
  __shared__ float4 ra[16][4], rb[16][4];
  const int n = min(*n_valid, R);
  const int cq = threadIdx.x & 3, pr = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + cq * 4;
  const bool on = c < D;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 mu = ld4_if(on, mean + c, z4), is = ld4_if(on, invstd + c, z4);
  const float4 ga = ld4_if(on, gamma + c, z4), be = ld4_if(on, beta + c, z4);
  float4 xh[4], gv[4];           // normalised input, gradient behind the LeakyReLU (zero on rows >= n)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = pr + 256 * u;
    const bool in = on && r < n;
    const float4 xv = ld4_if(in, x + (int64_t)r * D + c, mu);
    float4 gg = ld4_if(in, g + (int64_t)r * D + c, z4);
    const float4 yv = ld4_if(in && yout, yout + (int64_t)r * D + c, z4);
    xh[u] = make_float4((xv.x - mu.x) * is.x, (xv.y - mu.y) * is.y, (xv.z - mu.z) * is.z, (xv.w - mu.w) * is.w);
    if (yout ? yv.x <= 0.f : xh[u].x * ga.x + be.x <= 0.f) gg.x *= slope;
    if (yout ? yv.y <= 0.f : xh[u].y * ga.y + be.y <= 0.f) gg.y *= slope;
    if (yout ? yv.z <= 0.f : xh[u].z * ga.z + be.z <= 0.f) gg.z *= slope;
    if (yout ? yv.w <= 0.f : xh[u].w * ga.w + be.w <= 0.f) gg.w *= slope;
    gv[u] = gg;
  }
  float4 a = z4, b = z4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    add4(a, gv[u]);
    b.x += gv[u].x * xh[u].x; b.y += gv[u].y * xh[u].y; b.z += gv[u].z * xh[u].z; b.w += gv[u].w * xh[u].w;
  }
  mid_sum(a, b, ra, rb);
  if (on) {
    if (pr == 0) {
      *reinterpret_cast<float4*>(dgb + c) = a;
      *reinterpret_cast<float4*>(dgb + D + c) = b;
    }
    const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = pr + 256 * u;
      if (r >= R) continue;
      float4 o = z4, go = z4;
      if (r < n) {
        go = gv[u];
        o.x = ga.x * is.x * (go.x - a.x * inv_n - xh[u].x * b.x * inv_n);
        o.y = ga.y * is.y * (go.y - a.y * inv_n - xh[u].y * b.y * inv_n);
        o.z = ga.z * is.z * (go.z - a.z * inv_n - xh[u].z * b.z * inv_n);
        o.w = ga.w * is.w * (go.w - a.w * inv_n - xh[u].w * b.w * inv_n);
      }
      *reinterpret_cast<float4*>(dx + (int64_t)r * D + c) = o;
      if (d_addend) *reinterpret_cast<float4*>(d_addend + (int64_t)r * D + c) = go;
    }
  }
}

__global__ __launch_bounds__(1024) void bn_mid_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                   const int* __restrict__ n_valid, int R, int D,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float slope, const float* __restrict__ yout,
                                                   float* __restrict__ d_addend, float* __restrict__ dgb,
                                                   float* __restrict__ dx) {
  bn_mid_bwd_body(x, g, n_valid, R, D, mean, invstd, gamma, beta, slope, yout, d_addend, dgb, dx);
}

// two independent problems of the same row count in one launch (blockIdx.z): the BatchNorm of a bottleneck block's
// convolution and the one of its shortcut (blocks.py:596-649) -- a launch of this chain costs more than its work
struct bn_mid_bwd_args {
  const float* x;
  const float* g;
  const int* n_valid;
  int R;
  int D;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float slope;
  const float* yout;
  float* d_addend;
  float* dgb;
  float* dx;
};
__global__ __launch_bounds__(1024) void bn_mid_bwd_pair(bn_mid_bwd_args a0, bn_mid_bwd_args a1) {
  const bn_mid_bwd_args& a = blockIdx.z ? a1 : a0;
  bn_mid_bwd_body(a.x, a.g, a.n_valid, a.R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.yout, a.d_addend, a.dgb, a.dx);
}

// ---- y = LeakyReLU(a + b) (the residual join of ResnetBottleneckBlock, blocks.py:649) in one launch
__global__ void add_lrelu_fwd_k(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float slope,
                                float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float v = a[t] + b[t];
  y[t] = v > 0.f ? v : v * slope;
}

__global__ void add_lrelu_bwd_k(const float* __restrict__ y, const float* __restrict__ g, int64_t n, float slope,
                                float* __restrict__ d) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  d[t] = y[t] > 0.f ? g[t] : g[t] * slope;   // slope > 0: sign(y) == sign(a + b)
}

// rows up to which the single-launch kernels are used (development override: MVK_BN_SMALL_ROWS)
int bn_small_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MVK_BN_SMALL_ROWS");
    v = e ? atoi(e) : BN_SMALL_ROWS_DEFAULT;
  }
  return v;
}

// every non-null pointer 16-byte aligned (the vectorised kernels load float4)
bool aligned16(const void* a, const void* b, const void* c, const void* d, const void* e, const void* f, const void* g,
               const void* h) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f | (uintptr_t)g |
           (uintptr_t)h) & 15) == 0;
}

// rows up to which the vectorised single-launch kernels are used when D % 4 == 0 (MVK_BN_MID_ROWS)
int bn_mid_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MVK_BN_MID_ROWS");
    v = e ? atoi(e) : BN_MID_ROWS_DEFAULT;
    if (v > BN_MID_MAX) v = BN_MID_MAX;
  }
  return v;
}

}  // namespace

// rows up to which mvk_bn_lrelu_fwd / _bwd are ONE launch for a D-channel input (a producer need not emit statistics)
extern "C" int mvk_bn_single_launch_rows(int D) {
  static const bool keep_stats = getenv("MVK_BN_MID_KEEP_STATS") != nullptr;     // development: producers still emit partials
  if (keep_stats) return bn_small_rows();
  return (D % 4 == 0 && bn_mid_rows() > bn_small_rows()) ? bn_mid_rows() : bn_small_rows();
}

extern "C" int mvk_bn_lrelu_fwd(const float* x, const int32_t* n_valid, int64_t R, int D, const float* gamma,
                                const float* beta, float eps, float momentum, float slope, float* running_mean,
                                float* running_var, float* mean, float* invstd, float* scratch2D /* [ceil(R/64),2,D] */, float* y,
                                int64_t* num_batches_tracked, const float* addend, const float* ext_part, int ext_rows,
                                void* stream) {
  MVK_REQUIRE(R >= 0 && D > 0 && R < (1ll << 31), "bn: bad sizes");
  MVK_REQUIRE((ext_part == nullptr) == (ext_rows <= 0) && ext_rows >= -1, "bn: ext_part and ext_rows go together (-1: finished statistics)");
  if (R == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (ext_rows < 0) {      // mean / invstd finished by the producing GEMM: the apply pass alone, whatever the row count
    MVK_REQUIRE(bn_fused_finish(), "bn: finished statistics need the fused apply kernel (MVK_BN_FUSED_FINISH=1)");
    const unsigned gyr = (unsigned)cdiv64(R, BN_ROWS);
    const unsigned gy = gyr < (unsigned)bn_fused_gy() ? gyr : (unsigned)bn_fused_gy();
    hipLaunchKernelGGL(bn_finish_apply, dim3((unsigned)cdiv64(D, 64), gy), dim3(1024), 0, st, x, n_valid, (int)R, D, nullptr,
                       eps, momentum, gamma, beta, slope, mean, invstd, nullptr, nullptr, nullptr, addend, y, -1);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (R > bn_small_rows() && R <= bn_mid_rows() && D % 4 == 0 &&
      aligned16(x, y, addend, gamma, beta, mean, invstd, running_mean) &&
      aligned16(running_var, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) {
    hipLaunchKernelGGL(bn_mid_fwd, dim3((unsigned)cdiv64(D, 16)), dim3(1024), 0, st, x, n_valid, (int)R, D, gamma, beta,
                       eps, momentum, slope, running_mean, running_var, mean, invstd, y, (long long*)num_batches_tracked, addend);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (R <= bn_small_rows()) {
    hipLaunchKernelGGL(bn_small_fwd, dim3((unsigned)cdiv64(D, 64)), dim3(1024), 0, st, x, n_valid, (int)R, D, gamma, beta,
                       eps, momentum, slope, running_mean, running_var, mean, invstd, y, (long long*)num_batches_tracked, addend);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  dim3 g1((unsigned)cdiv64(D, 64), (unsigned)cdiv64(R, BN_ROWS));
  if (ext_part && bn_fused_finish()) {       // statistics already produced by the GEMM epilogue: one launch
    const unsigned gy = g1.y < (unsigned)bn_fused_gy() ? g1.y : (unsigned)bn_fused_gy();
    hipLaunchKernelGGL(bn_finish_apply, dim3(g1.x, gy), dim3(1024), 0, st, x, n_valid, (int)R, D, ext_part, eps, momentum,
                       gamma, beta, slope, mean, invstd, running_mean, running_var, (long long*)num_batches_tracked, addend, y,
                       ext_rows);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(bn_stats_partial, g1, dim3(BN_T), 0, st, x, n_valid, (int)R, D, scratch2D);
  if (bn_fused_finish()) {
    const unsigned gy = g1.y < (unsigned)bn_fused_gy() ? g1.y : (unsigned)bn_fused_gy();
    hipLaunchKernelGGL(bn_finish_apply, dim3(g1.x, gy), dim3(1024), 0, st, x, n_valid, (int)R, D, scratch2D, eps, momentum,
                       gamma, beta, slope, mean, invstd, running_mean, running_var, (long long*)num_batches_tracked, addend, y, 0);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(bn_stats_finish, dim3((unsigned)cdiv64(D, 64)), dim3(1024), 0, st, x, n_valid, (int)R, D, scratch2D,
                     eps, momentum, mean, invstd, running_mean, running_var, (long long*)num_batches_tracked);
  hipLaunchKernelGGL(bn_apply, dim3((unsigned)cdiv64(R * D, 256)), dim3(256), 0, st, x, n_valid, (int)R, D, mean, invstd,
                     gamma, beta, slope, addend, y);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_bn_lrelu_bwd(const float* x, const float* g, const int32_t* n_valid, int64_t R, int D,
                                const float* gamma, const float* beta, const float* mean, const float* invstd,
                                float slope, float* scratch /* [ceil(R/64),2,D] */,
                                float* dgamma_dbeta /* [2,D]: dbeta then dgamma */, float* dx, const float* y_out,
                                float* d_addend, void* stream) {
  MVK_REQUIRE(R >= 0 && D > 0 && R < (1ll << 31), "bn: bad sizes");
  MVK_REQUIRE((y_out == nullptr) == (d_addend == nullptr), "bn: y_out and d_addend go together (residual-join mode)");
  hipStream_t st = (hipStream_t)stream;
  if (R == 0) {
    MVK_CHECK_HIP(hipMemsetAsync(dgamma_dbeta, 0, sizeof(float) * 2 * D, st));
    return 0;
  }
  if (R > bn_small_rows() && R <= bn_mid_rows() && D % 4 == 0 &&
      aligned16(x, g, y_out, d_addend, dgamma_dbeta, dx, mean, invstd) && aligned16(gamma, beta, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) {
    hipLaunchKernelGGL(bn_mid_bwd, dim3((unsigned)cdiv64(D, 16)), dim3(1024), 0, st, x, g, n_valid, (int)R, D, mean,
                       invstd, gamma, beta, slope, y_out, d_addend, dgamma_dbeta, dx);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (R <= bn_small_rows()) {
    hipLaunchKernelGGL(bn_small_bwd, dim3((unsigned)cdiv64(D, 64)), dim3(1024), 0, st, x, g, n_valid, (int)R, D, mean,
                       invstd, gamma, beta, slope, y_out, d_addend, dgamma_dbeta, dx);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  dim3 g1((unsigned)cdiv64(D, 64), (unsigned)cdiv64(R, BN_ROWS));
  hipLaunchKernelGGL(bn_bwd_reduce, g1, dim3(BN_T), 0, st, x, g, n_valid, (int)R, D, mean, invstd, gamma, beta, slope,
                     y_out, scratch);
  if (bn_fused_finish()) {
    const unsigned gy = g1.y < (unsigned)bn_fused_gy() ? g1.y : (unsigned)bn_fused_gy();
    hipLaunchKernelGGL(bn_bwd_finish_apply, dim3(g1.x, gy), dim3(1024), 0, st, x, g, n_valid, (int)R, D, mean, invstd, gamma,
                       beta, slope, scratch, y_out, d_addend, dgamma_dbeta, dx);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(bn_bwd_finish, dim3((unsigned)cdiv64(D, 64)), dim3(1024), 0, st, scratch, (int)R, D, dgamma_dbeta);
  hipLaunchKernelGGL(bn_bwd_apply, dim3((unsigned)cdiv64(R * D, 256)), dim3(256), 0, st, x, g, n_valid, (int)R, D, mean,
                     invstd, gamma, beta, slope, dgamma_dbeta, y_out, d_addend, dx);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- two BatchNorm problems of the same row count per launch (include/mvkpconv.h: mvk_bn_fwd_problem / _bwd_problem).
// A bottleneck block normalises its convolution output and its shortcut (blocks.py:596-649) independently; issued as a
// pair they cost one launch each way instead of two (three instead of five... for the two-launch backward of the big
// levels). Problems that fall into different kernel families (row count, alignment) run one after the other.
namespace {

int bn_family_fwd(const mvk_bn_fwd_problem& p) {
  if (p.ext_rows < 0) return 2;          // finished statistics: the apply kernel alone
  if (p.R <= bn_small_rows()) return 0;
  if (p.R <= bn_mid_rows() && p.D % 4 == 0 &&
      aligned16(p.x, p.y, p.addend, p.gamma, p.beta, p.mean, p.invstd, p.running_mean) &&
      aligned16(p.running_var, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr))
    return 1;
  return 2;
}

int bn_family_bwd(const mvk_bn_bwd_problem& p) {
  if (p.R <= bn_small_rows()) return 0;
  if (p.R <= bn_mid_rows() && p.D % 4 == 0 &&
      aligned16(p.x, p.g, p.y_out, p.d_addend, p.dgamma_dbeta, p.dx, p.mean, p.invstd) &&
      aligned16(p.gamma, p.beta, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr))
    return 1;
  return 2;
}

int bn_fwd_single(const mvk_bn_fwd_problem& p, void* stream) {
  return mvk_bn_lrelu_fwd(p.x, p.n_valid, p.R, p.D, p.gamma, p.beta, p.eps, p.momentum, p.slope, p.running_mean,
                          p.running_var, p.mean, p.invstd, p.scratch2D, p.y, p.num_batches_tracked, p.addend, p.ext_part,
                          p.ext_rows, stream);
}

int bn_bwd_single(const mvk_bn_bwd_problem& p, void* stream) {
  return mvk_bn_lrelu_bwd(p.x, p.g, p.n_valid, p.R, p.D, p.gamma, p.beta, p.mean, p.invstd, p.slope, p.scratch,
                          p.dgamma_dbeta, p.dx, p.y_out, p.d_addend, stream);
}

}  // namespace

extern "C" int mvk_bn_lrelu_fwd_pair(const mvk_bn_fwd_problem* pa, const mvk_bn_fwd_problem* pb, void* stream) {
  MVK_REQUIRE(pa && pb, "bn pair: null problem");
  const mvk_bn_fwd_problem &a = *pa, &b = *pb;
  const int fam = bn_family_fwd(a);
  if (a.R != b.R || a.R <= 0 || fam != bn_family_fwd(b) || (fam == 2 && !bn_fused_finish())) {
    if (int e = bn_fwd_single(a, stream)) return e;
    return bn_fwd_single(b, stream);
  }
  MVK_REQUIRE(a.D > 0 && b.D > 0 && a.R < (1ll << 31), "bn: bad sizes");
  MVK_REQUIRE((a.ext_part == nullptr) == (a.ext_rows <= 0) && (b.ext_part == nullptr) == (b.ext_rows <= 0) &&
                  a.ext_rows >= -1 && b.ext_rows >= -1,
              "bn: ext_part and ext_rows go together");
  hipStream_t st = (hipStream_t)stream;
  const int R = (int)a.R, Dm = a.D > b.D ? a.D : b.D;
  if (fam == 0) {
    bn_small_fwd_args k0{a.x, a.n_valid, R, a.D, a.gamma, a.beta, a.eps, a.momentum, a.slope, a.running_mean, a.running_var,
                         a.mean, a.invstd, a.y, (long long*)a.num_batches_tracked, a.addend};
    bn_small_fwd_args k1{b.x, b.n_valid, R, b.D, b.gamma, b.beta, b.eps, b.momentum, b.slope, b.running_mean, b.running_var,
                         b.mean, b.invstd, b.y, (long long*)b.num_batches_tracked, b.addend};
    hipLaunchKernelGGL(bn_small_fwd_pair, dim3((unsigned)cdiv64(Dm, 64), 1, 2), dim3(1024), 0, st, k0, k1);
  } else if (fam == 1) {
    bn_mid_fwd_args k0{a.x, a.n_valid, R, a.D, a.gamma, a.beta, a.eps, a.momentum, a.slope, a.running_mean, a.running_var,
                       a.mean, a.invstd, a.y, (long long*)a.num_batches_tracked, a.addend};
    bn_mid_fwd_args k1{b.x, b.n_valid, R, b.D, b.gamma, b.beta, b.eps, b.momentum, b.slope, b.running_mean, b.running_var,
                       b.mean, b.invstd, b.y, (long long*)b.num_batches_tracked, b.addend};
    hipLaunchKernelGGL(bn_mid_fwd_pair, dim3((unsigned)cdiv64(Dm, 16), 1, 2), dim3(1024), 0, st, k0, k1);
  } else {
    const unsigned gyr = (unsigned)cdiv64(R, BN_ROWS);
    for (const mvk_bn_fwd_problem* p : {pa, pb})
      if (p->ext_part == nullptr && p->ext_rows == 0)       // no statistics from the producing GEMM (a split reduction): its own pass
        hipLaunchKernelGGL(bn_stats_partial, dim3((unsigned)cdiv64(p->D, 64), gyr), dim3(BN_T), 0, st, p->x, p->n_valid, R,
                           p->D, p->scratch2D);
    const unsigned gy = gyr < (unsigned)bn_fused_gy() ? gyr : (unsigned)bn_fused_gy();
    bn_finish_apply_args k0{a.x, a.n_valid, R, a.D, a.ext_part ? a.ext_part : a.scratch2D, a.eps, a.momentum, a.gamma, a.beta,
                            a.slope, a.mean, a.invstd, a.running_mean, a.running_var, (long long*)a.num_batches_tracked,
                            a.addend, a.y, a.ext_rows};
    bn_finish_apply_args k1{b.x, b.n_valid, R, b.D, b.ext_part ? b.ext_part : b.scratch2D, b.eps, b.momentum, b.gamma, b.beta,
                            b.slope, b.mean, b.invstd, b.running_mean, b.running_var, (long long*)b.num_batches_tracked,
                            b.addend, b.y, b.ext_rows};
    hipLaunchKernelGGL(bn_finish_apply_pair, dim3((unsigned)cdiv64(Dm, 64), gy, 2), dim3(1024), 0, st, k0, k1);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_bn_lrelu_bwd_pair(const mvk_bn_bwd_problem* pa, const mvk_bn_bwd_problem* pb, void* stream) {
  MVK_REQUIRE(pa && pb, "bn pair: null problem");
  const mvk_bn_bwd_problem &a = *pa, &b = *pb;
  const int fam = bn_family_bwd(a);
  if (a.R != b.R || a.R <= 0 || fam != bn_family_bwd(b) || (fam == 2 && !bn_fused_finish())) {
    if (int e = bn_bwd_single(a, stream)) return e;
    return bn_bwd_single(b, stream);
  }
  MVK_REQUIRE(a.D > 0 && b.D > 0 && a.R < (1ll << 31), "bn: bad sizes");
  MVK_REQUIRE((a.y_out == nullptr) == (a.d_addend == nullptr) && (b.y_out == nullptr) == (b.d_addend == nullptr),
              "bn: y_out and d_addend go together (residual-join mode)");
  hipStream_t st = (hipStream_t)stream;
  const int R = (int)a.R, Dm = a.D > b.D ? a.D : b.D;
  if (fam == 0) {
    bn_small_bwd_args k0{a.x, a.g, a.n_valid, R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.y_out, a.d_addend,
                         a.dgamma_dbeta, a.dx};
    bn_small_bwd_args k1{b.x, b.g, b.n_valid, R, b.D, b.mean, b.invstd, b.gamma, b.beta, b.slope, b.y_out, b.d_addend,
                         b.dgamma_dbeta, b.dx};
    hipLaunchKernelGGL(bn_small_bwd_pair, dim3((unsigned)cdiv64(Dm, 64), 1, 2), dim3(1024), 0, st, k0, k1);
  } else if (fam == 1) {
    bn_mid_bwd_args k0{a.x, a.g, a.n_valid, R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.y_out, a.d_addend,
                       a.dgamma_dbeta, a.dx};
    bn_mid_bwd_args k1{b.x, b.g, b.n_valid, R, b.D, b.mean, b.invstd, b.gamma, b.beta, b.slope, b.y_out, b.d_addend,
                       b.dgamma_dbeta, b.dx};
    hipLaunchKernelGGL(bn_mid_bwd_pair, dim3((unsigned)cdiv64(Dm, 16), 1, 2), dim3(1024), 0, st, k0, k1);
  } else {
    const unsigned gyr = (unsigned)cdiv64(R, BN_ROWS);
    bn_bwd_reduce_args r0{a.x, a.g, a.n_valid, R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.y_out, a.scratch};
    bn_bwd_reduce_args r1{b.x, b.g, b.n_valid, R, b.D, b.mean, b.invstd, b.gamma, b.beta, b.slope, b.y_out, b.scratch};
    hipLaunchKernelGGL(bn_bwd_reduce_pair, dim3((unsigned)cdiv64(Dm, 64), gyr, 2), dim3(BN_T), 0, st, r0, r1);
    const unsigned gy = gyr < (unsigned)bn_fused_gy() ? gyr : (unsigned)bn_fused_gy();
    bn_bwd_finish_apply_args k0{a.x, a.g, a.n_valid, R, a.D, a.mean, a.invstd, a.gamma, a.beta, a.slope, a.scratch, a.y_out,
                                a.d_addend, a.dgamma_dbeta, a.dx};
    bn_bwd_finish_apply_args k1{b.x, b.g, b.n_valid, R, b.D, b.mean, b.invstd, b.gamma, b.beta, b.slope, b.scratch, b.y_out,
                                b.d_addend, b.dgamma_dbeta, b.dx};
    hipLaunchKernelGGL(bn_bwd_finish_apply_pair, dim3((unsigned)cdiv64(Dm, 64), gy, 2), dim3(1024), 0, st, k0, k1);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- bias + LeakyReLU of the layers without BatchNorm (blocks.py:462-463 `x + self.bias`, then the block's
// LeakyReLU: the two head layers of every network). As tensor ops that is 2 launches forward and 4 backward (activation
// backward, two library reductions for the bias gradient at 11-17 us each, a fill); here one launch each way.
// Row-major [R, C]; a workgroup owns 64 rows, thread t the column t % CP of the row lanes t / CP (CP = C rounded up to a
// power of two <= 256), the backward's column sums meet in LDS and leave as one atomic per workgroup and column.
__global__ __launch_bounds__(256) void bias_lrelu_fwd_k(const float* __restrict__ x, const float* __restrict__ bias, int64_t R,
                                                       int C, int CP, float slope, float* __restrict__ y) {
  const int col = threadIdx.x % CP, rl = threadIdx.x / CP, RL = 256 / CP;
  if (col >= C) return;
  const float b = bias[col];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int64_t r = r0 + rl; r < r0 + 64 && r < R; r += RL) {
    const float v = x[r * C + col] + b;
    y[r * C + col] = v > 0.f ? v : v * slope;
  }
}

// ordered form (park != null): every workgroup parks its column sums in park [workgroup][C] (agent-scope relaxed stores,
// see the split reduction of csrc/gemm.hip); the one that arrives last at *counter adds them in workgroup order and
// writes dbias with plain stores -- bit-identical from run to run, no zero-initialised dbias, *counter back at zero.
__global__ __launch_bounds__(256) void bias_lrelu_bwd_k(const float* __restrict__ y, const float* __restrict__ g, int64_t R, int C,
                                                       int CP, float slope, float* __restrict__ dx, float* __restrict__ dbias,
                                                       float* __restrict__ park, int* __restrict__ counter) {
  __shared__ float part[256];
  __shared__ int last_flag;
  const int col = threadIdx.x % CP, rl = threadIdx.x / CP, RL = 256 / CP;
  float s = 0.f;
  if (col < C) {
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int64_t r = r0 + rl; r < r0 + 64 && r < R; r += RL) {
      // the sign of the output is the sign of the pre-activation (slope > 0); slope == 1: identity
      const float d = g[r * C + col] * (y[r * C + col] > 0.f ? 1.f : slope);
      dx[r * C + col] = d;
      s += d;
    }
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && col < C) {
    for (int i = 1; i < RL; ++i) s += part[i * CP + col];
    if (park)
      park_store1(park_rsrc(park + (int64_t)blockIdx.x * C), (uint32_t)col * 4u, s);
    else
      atomicAdd(dbias + col, s);
  }
  if (!park) return;
  park_wait();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = old == (int)gridDim.x - 1;
    if (last_flag) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_flag) return;
  // the last workgroup: row lane rl adds the parked sums of workgroups rl, rl + RL, ... in that order, sixteen loads per
  // round trip; the RL partial results meet in LDS and are added in lane order -- a fixed order throughout
  float t = 0.f;
  if (col < C) {
    const int nb = (int)gridDim.x;
    for (int b0 = rl; b0 < nb; b0 += 16 * RL) {
      float v[16];
      // (all sixteen loads issued before the first add: blocks past the end are clamped, then skipped)
      const __amdgpu_buffer_rsrc_t pr = park_rsrc(park);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = b0 + u * RL < nb ? b0 + u * RL : nb - 1;
        v[u] = park_load1(pr, ((uint32_t)b * (uint32_t)C + (uint32_t)col) * 4u);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (b0 + u * RL < nb) t += v[u];
    }
  }
  __syncthreads();
  part[threadIdx.x] = t;
  __syncthreads();
  if (rl == 0 && col < C) {
    for (int i = 1; i < RL; ++i) t += part[i * CP + col];
    dbias[col] = t;
  }
}

extern "C" int mvk_add_lrelu_fwd(const float* a, const float* b, int64_t n, float slope, float* y, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_lrelu_fwd_k, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, slope, y);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_add_lrelu_bwd(const float* y, const float* g, int64_t n, float slope, float* d, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_lrelu_bwd_k, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, y, g, n, slope, d);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

static int pow2_ge(int c) {
  int p = 1;
  while (p < c) p <<= 1;
  return p;
}

extern "C" int mvk_bias_lrelu_fwd(const float* x, const float* bias, int64_t R, int C, float slope, float* y, void* stream) {
  MVK_REQUIRE(R >= 0 && C >= 1 && C <= 256, "bias_lrelu: C=%d unsupported (1..256)", C);
  if (R == 0) return 0;
  hipLaunchKernelGGL(bias_lrelu_fwd_k, dim3((unsigned)cdiv64(R, 64)), dim3(256), 0, (hipStream_t)stream, x, bias, R, C,
                     pow2_ge(C), slope, y);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

/* dbias [C]: with the arena of ordered reductions set (mvk_gemm_split_ordered() == 1) it is WRITTEN (the workgroups'
 * column sums are added in workgroup order: bit-identical from run to run); otherwise it must be zero-initialised (one
 * atomic per 64-row block and column is added to it) */
extern "C" int mvk_bias_lrelu_bwd(const float* y, const float* g, int64_t R, int C, float slope, float* dx, float* dbias,
                                  void* stream) {
  MVK_REQUIRE(R >= 0 && C >= 1 && C <= 256, "bias_lrelu: C=%d unsupported (1..256)", C);
  if (R == 0) {
    if (mvk_gemm_split_ordered()) MVK_CHECK_HIP(hipMemsetAsync(dbias, 0, sizeof(float) * C, (hipStream_t)stream));
    return 0;
  }
  float* park = nullptr;
  int* counter = nullptr;
  const int64_t blocks = cdiv64(R, 64);
  if (mvk_gemm_split_ordered())
    MVK_REQUIRE(mvk_internal_arena_take(stream, blocks * C, 1, &park, &counter), "bias_lrelu: the arena of ordered reductions is too small");
  hipLaunchKernelGGL(bias_lrelu_bwd_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, g, R, C,
                     pow2_ge(C), slope, dx, dbias, park, counter);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
