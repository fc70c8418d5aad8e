// fp32-in / fp32-accumulate GEMM on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf chain, 64 FLOP/clk/SIMD).
//
// This is the dense K x Cin x Cout contraction of KPConv (reference
// KPConv-PyTorch/models/blocks.py:370-374: permute + matmul + sum over K ==
// one [N, K*Cin] x [K*Cin, Cout] GEMM) and its two backward
// products. bf16/fp16 MFMA would miss the 1e-4 parity bar, so
// the f32 MFMA forms are used.
//
// Shapes are tall-skinny (20 000 x 990 x 64) or short-and-deep (65 x 7680 x 512), never square:
//   * 64 x 64 output tile per 256-thread workgroup (128 x 32 for outputs of at most 32 columns), BK = 32;
//     four waves in a 2 x 2 (4 x 1) arrangement, each owning one 32 x 32 accumulator (16 VGPRs) -> 16 MFMAs
//     (>= 1024 cycles) per wave and k-tile;
//   * operands staged k-major in LDS ([BK][rows + 4]): the MFMA operand reads (lane l -> row/col l & 31,
//     k = l >> 5) are bank-conflict free, m/n-contiguous sources are written with ds_write_b128;
//   * two LDS buffers + register prefetch: the global loads of k-tile t+1 are issued before the MFMAs
//     of tile t and written to the other buffer after them -> one barrier per k-tile, loads hidden;
//   * split-K (grid.z) with f32 atomics fills the chip when M*N is small (coarse layers, dW).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

// One operand tile = ROWS (i) x 32 (k) elements, ROWS / 32 float4 per thread (ROWS = 32, 64 or 128).
//   CONTIG_K == true : source element (i,k) at src[i*ld + k]  (contiguous along k)
//   CONTIG_K == false: source element (i,k) at src[k*ld + i]  (contiguous along i)
template <bool CONTIG_K, int ROWS>
struct TileLoader {
  static constexpr int NV = ROWS / 32;          // float4 per thread
  static constexpr int IPT = ROWS / 4;          // threads along i (i-contiguous sources)
  static constexpr int KSTEP = 256 / IPT;       // k rows covered per pass
  float4 v[NV];

  // vec: 4 = rows 16-byte aligned, 2 = 8-byte aligned (e.g. K*Cin = 990), 1 = scalar
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int64_t i0, int64_t imax,
                                       int64_t k0, int64_t kmax, int vec, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      if (CONTIG_K) {
        const int i = (tid >> 3) + 32 * u, kq = (tid & 7) * 4;
        const int64_t gi = i0 + i, gk = k0 + kq;
        if (gi < imax) {
          const float* p = src + gi * ld + gk;
          if (vec == 4 && gk + 3 < kmax) {
            const float4 q = *reinterpret_cast<const float4*>(p);
            t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
          } else if (vec == 2 && gk + 3 < kmax) {
            const float2 q0 = *reinterpret_cast<const float2*>(p), q1 = *reinterpret_cast<const float2*>(p + 2);
            t[0] = q0.x; t[1] = q0.y; t[2] = q1.x; t[3] = q1.y;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gk + e < kmax) t[e] = p[e];
          }
        }
      } else {
        const int k = tid / IPT + KSTEP * u, iq = (tid % IPT) * 4;
        const int64_t gk = k0 + k, gi = i0 + iq;
        if (gk < kmax) {
          const float* p = src + gk * ld + gi;
          if (vec == 4 && gi + 3 < imax) {
            const float4 q = *reinterpret_cast<const float4*>(p);
            t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
          } else if (vec == 2 && gi + 3 < imax) {
            const float2 q0 = *reinterpret_cast<const float2*>(p), q1 = *reinterpret_cast<const float2*>(p + 2);
            t[0] = q0.x; t[1] = q0.y; t[2] = q1.x; t[3] = q1.y;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gi + e < imax) t[e] = p[e];
          }
        }
      }
      v[u] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }

  __device__ __forceinline__ void store(float (*T)[ROWS + 4], int tid) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (CONTIG_K) {
        const int i = (tid >> 3) + 32 * u, kq = (tid & 7) * 4;
        T[kq + 0][i] = v[u].x;
        T[kq + 1][i] = v[u].y;
        T[kq + 2][i] = v[u].z;
        T[kq + 3][i] = v[u].w;
      } else {
        const int k = tid / IPT + KSTEP * u, iq = (tid % IPT) * 4;
        *reinterpret_cast<float4*>(&T[k][iq]) = v[u];
      }
    }
  }
};

// TM x TN output tile per 256-thread workgroup: 64 x 64 (waves 2 x 2) or, for outputs of at most 32
// columns (Cout = 32 layers, logits), 128 x 32 (waves 4 x 1) so that no MFMA work is spent on padding.
template <bool TA, bool TB, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const float* __restrict__ A,
                                                     const float* __restrict__ B,
                                                     float* __restrict__ C, int64_t M, int64_t N,
                                                     int64_t Kd, int64_t lda, int64_t ldb,
                                                     int64_t k_per_split, int atomic_out,
                                                     int accumulate, int vecA, int vecB) {
  static_assert((TM / 32) * (TN / 32) == 4, "four 32 x 32 wave tiles per workgroup");
  __shared__ __attribute__((aligned(16))) float As[2][BK][TM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][TN + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / (TN / 32), wn = wave % (TN / 32);
  const int64_t m0 = (int64_t)blockIdx.y * TM, n0 = (int64_t)blockIdx.x * TN;
  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = kbeg + k_per_split < Kd ? kbeg + k_per_split : Kd;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // A tile: element (m,k); TA == false -> A[m*lda + k] (k-contiguous). B tile: element (n,k);
  // TB == false -> B[k*ldb + n] (n-contiguous).
  TileLoader<!TA, TM> la;
  TileLoader<TB, TN> lb;
  la.load(A, lda, m0, M, kbeg, kend, vecA, tid);
  lb.load(B, ldb, n0, N, kbeg, kend, vecB, tid);
  la.store(As[0], tid);
  lb.store(Bs[0], tid);
  __syncthreads();

  const int i = wm * 32 + (lane & 31), j = wn * 32 + (lane & 31), kh = lane >> 5;
  int buf = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {  // prefetch the next k-tile into registers while this one is multiplied
      la.load(A, lda, m0, M, k0 + BK, kend, vecA, tid);
      lb.load(B, ldb, n0, N, k0 + BK, kend, vecB, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[buf][kk + kh][i];
      const float b = Bs[buf][kk + kh][j];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (more) {
      la.store(As[buf ^ 1], tid);
      lb.store(Bs[buf ^ 1], tid);
    }
    __syncthreads();
    buf ^= 1;
  }

  const int64_t col = n0 + wn * 32 + (lane & 31);
  if (col < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float* c = C + row * N + col;
        if (atomic_out)
          atomicAdd(c, acc[r]);
        else if (accumulate)
          *c += acc[r];
        else
          *c = acc[r];
      }
    }
  }
}

}  // namespace

extern "C" int mvk_gemm_f32(const float* A, const float* B, float* C, int64_t M, int64_t N,
                            int64_t Kd, int transA, int transB, int accumulate, int split_k,
                            void* stream) {
  MVK_REQUIRE(M >= 0 && N >= 0 && Kd >= 0, "gemm: negative size");
  if (M == 0 || N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (Kd == 0) {
    if (!accumulate && split_k <= 1) MVK_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(float) * M * N, st));
    return 0;
  }
  if (split_k < 1) split_k = 1;
  int64_t ksteps = cdiv64(Kd, BK);
  if (split_k > ksteps) split_k = (int)ksteps;
  int64_t k_per_split = cdiv64(ksteps, split_k) * BK;
  split_k = (int)cdiv64(Kd, k_per_split);
  const int64_t lda = transA ? M : Kd, ldb = transB ? Kd : N;
  const int vecA = ((lda % 4 == 0) && ((uintptr_t)A % 16 == 0)) ? 4 : ((lda % 2 == 0) && ((uintptr_t)A % 8 == 0)) ? 2 : 1;
  const int vecB = ((ldb % 4 == 0) && ((uintptr_t)B % 16 == 0)) ? 4 : ((ldb % 2 == 0) && ((uintptr_t)B % 8 == 0)) ? 2 : 1;
  const bool narrow = N <= 32;   // 128 x 32 tiles: no MFMA work on padded columns
  const int TMh = narrow ? 128 : 64, TNh = narrow ? 32 : 64;
  MVK_REQUIRE(cdiv64(M, TMh) < 65536 && split_k < 65536, "gemm: grid too large");
  dim3 grid((unsigned)cdiv64(N, TNh), (unsigned)cdiv64(M, TMh), (unsigned)split_k), block(256);
  const int atomic_out = split_k > 1;
#define LAUNCH(TA, TB)                                                                                     \
  do {                                                                                                     \
    if (narrow)                                                                                            \
      hipLaunchKernelGGL((gemm_f32_mfma<TA, TB, 128, 32>), grid, block, 0, st, A, B, C, M, N, Kd, lda, ldb, \
                         k_per_split, atomic_out, accumulate, vecA, vecB);                                 \
    else                                                                                                   \
      hipLaunchKernelGGL((gemm_f32_mfma<TA, TB, 64, 64>), grid, block, 0, st, A, B, C, M, N, Kd, lda, ldb,  \
                         k_per_split, atomic_out, accumulate, vecA, vecB);                                 \
  } while (0)
  if (!transA && !transB) LAUNCH(false, false);
  else if (!transA && transB) LAUNCH(false, true);
  else if (transA && !transB) LAUNCH(true, false);
  else LAUNCH(true, true);
#undef LAUNCH
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
