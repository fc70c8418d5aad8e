// fp16-input / fp32-accumulate GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x8_f16) for the
// fp16-feature mode of KPConv (BASELINE config 5: "fp16 features with MFMA on the KP contraction"):
// the K x Cin x Cout contraction y = A . W (reference KPConv-PyTorch/models/blocks.py:370-374) and its
// two backward products dA = g . W^T, dW = A^T . g, with A / W stored in fp16 and g arriving in f32.
//
// Same blocking as gemm.hip: 64 x 64 output tile per 256-thread workgroup, BK = 32, four waves in a
// 2 x 2 arrangement with one 32 x 32 f32 accumulator each, two LDS buffers + register prefetch, split-K
// (grid.z) with f32 atomics. Differences:
//   * either operand may live in memory as f32 or f16; the loaders convert to f16 (round to nearest
//     even) while staging, so the MFMA always sees f16 -- numerically "round the operand to fp16, multiply
//     exactly, accumulate in f32";
//   * operands are staged i-major ([64][BK + 4] halfs, row = 72 bytes): an MFMA operand read is one
//     ds_read_b64 per lane (4 consecutive k), bank-conflict free (18 i mod 64 hits 32 distinct even banks);
//   * one k-tile = 4 MFMAs per wave (k = 8 each) instead of 16; f16 sources are staged bit for bit.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int BM = 64, BN = 64, BK = 32, LDH = BK + 4;   // LDS row: 36 halfs = 72 bytes

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

// n (4 or 8) consecutive elements starting at p into out[]; `al` = bytes of alignment every such run is
// known to have (16, 8, 4, or the element size): the widest vector load that alignment allows is used
// (K*Cin = 990 rows of f16 are only 4-byte aligned, of f32 8-byte aligned).
template <int NEL>
__device__ __forceinline__ void ld_run(const float* p, int al, _Float16* out) {
  if (al >= 16) {
#pragma unroll
    for (int e = 0; e < NEL; e += 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + e);
      out[e] = (_Float16)a.x; out[e + 1] = (_Float16)a.y; out[e + 2] = (_Float16)a.z; out[e + 3] = (_Float16)a.w;
    }
  } else if (al >= 8) {
#pragma unroll
    for (int e = 0; e < NEL; e += 2) {
      const float2 a = *reinterpret_cast<const float2*>(p + e);
      out[e] = (_Float16)a.x; out[e + 1] = (_Float16)a.y;
    }
  } else {
#pragma unroll
    for (int e = 0; e < NEL; ++e) out[e] = (_Float16)p[e];
  }
}
template <int NEL>
__device__ __forceinline__ void ld_run(const _Float16* p, int al, _Float16* out) {
  if (al >= 2 * NEL) {                     // one load for the whole run (16 B for 8 halfs, 8 B for 4)
    if (NEL == 8) {
      const uint4 q = *reinterpret_cast<const uint4*>(p);
      const _Float16* h = reinterpret_cast<const _Float16*>(&q);
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = h[e];
    } else {
      const h4 q = *reinterpret_cast<const h4*>(p);
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = q[e];
    }
  } else if (al >= 8) {
#pragma unroll
    for (int e = 0; e < NEL; e += 4) {
      const h4 q = *reinterpret_cast<const h4*>(p + e);
      out[e] = q[0]; out[e + 1] = q[1]; out[e + 2] = q[2]; out[e + 3] = q[3];
    }
  } else if (al >= 4) {
#pragma unroll
    for (int e = 0; e < NEL; e += 2) {
      const h2 a = *reinterpret_cast<const h2*>(p + e);
      out[e] = a[0]; out[e + 1] = a[1];
    }
  } else {
#pragma unroll
    for (int e = 0; e < NEL; ++e) out[e] = p[e];
  }
}

// One operand tile = 64 (i) x 32 (k) elements, 8 per thread, held as floats between load and store.
//   CONTIG_K: source element (i,k) at src[i*ld + k]: thread -> row i = tid / 4, k = (tid % 4) * 8 .. +7
//   else    : source element (i,k) at src[k*ld + i]: thread -> i = (tid % 16) * 4 .. +3, k = (tid / 16) * 2, +1
template <bool CONTIG_K, typename T>
struct TileLoader16 {
  _Float16 v[8];   // already rounded to fp16 (f16 sources are copied bit for bit)

  __device__ __forceinline__ void load(const T* __restrict__ src, int64_t ld, int64_t i0, int64_t imax, int64_t k0,
                                       int64_t kmax, int al, int tid) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
    if (CONTIG_K) {
      const int64_t gi = i0 + (tid >> 2), gk = k0 + (tid & 3) * 8;
      if (gi >= imax) return;
      const T* p = src + gi * ld + gk;
      if (gk + 7 < kmax) {
        ld_run<8>(p, al, v);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (gk + e < kmax) v[e] = (_Float16)ldf(p + e);
      }
    } else {
      const int64_t gi = i0 + (tid & 15) * 4;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int64_t gk = k0 + (tid >> 4) * 2 + kk;
        if (gk >= kmax) continue;
        const T* p = src + gk * ld + gi;
        if (gi + 3 < imax) {
          ld_run<4>(p, al, v + kk * 4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gi + e < imax) v[kk * 4 + e] = (_Float16)ldf(p + e);
        }
      }
    }
  }

  __device__ __forceinline__ void store(_Float16 (*Tl)[LDH], int tid) const {
    if (CONTIG_K) {
      const int i = tid >> 2, k = (tid & 3) * 8;
      h4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = v[e];
        b[e] = v[4 + e];
      }
      *reinterpret_cast<h4*>(&Tl[i][k]) = a;        // 72 i + 2 k: 8-byte aligned
      *reinterpret_cast<h4*>(&Tl[i][k + 4]) = b;
    } else {
      const int i = (tid & 15) * 4, k = (tid >> 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h2 w;
        w[0] = v[e];
        w[1] = v[4 + e];
        *reinterpret_cast<h2*>(&Tl[i + e][k]) = w;  // 4-byte aligned (k even)
      }
    }
  }
};

template <bool TA, bool TB, typename AT, typename BT, bool OUT16>
__global__ __launch_bounds__(256) void gemm_f16_mfma(const AT* __restrict__ A, const BT* __restrict__ B,
                                                     void* __restrict__ Cv, int64_t M, int64_t N, int64_t Kd,
                                                     int64_t lda, int64_t ldb, int64_t k_per_split, int atomic_out,
                                                     int alignedA, int alignedB) {
  __shared__ __attribute__((aligned(16))) _Float16 As[2][BM][LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][BN][LDH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = kbeg + k_per_split < Kd ? kbeg + k_per_split : Kd;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // A tile: element (m,k); TA == false -> A[m*lda + k] (k-contiguous). B tile: element (n,k);
  // TB == false -> B[k*ldb + n] (n-contiguous).
  TileLoader16<!TA, AT> la;
  TileLoader16<TB, BT> lb;
  la.load(A, lda, m0, M, kbeg, kend, alignedA, tid);
  lb.load(B, ldb, n0, N, kbeg, kend, alignedB, tid);
  la.store(As[0], tid);
  lb.store(Bs[0], tid);
  __syncthreads();

  const int i = wm * 32 + (lane & 31), j = wn * 32 + (lane & 31), kq = (lane >> 5) * 4;
  int buf = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      la.load(A, lda, m0, M, k0 + BK, kend, alignedA, tid);
      lb.load(B, ldb, n0, N, k0 + BK, kend, alignedB, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      const h4 a = *reinterpret_cast<const h4*>(&As[buf][i][kk + kq]);
      const h4 b = *reinterpret_cast<const h4*>(&Bs[buf][j][kk + kq]);
      acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc, 0, 0, 0);
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
        if (OUT16) {
          reinterpret_cast<_Float16*>(Cv)[row * N + col] = (_Float16)acc[r];
        } else {
          float* c = reinterpret_cast<float*>(Cv) + row * N + col;
          if (atomic_out)
            atomicAdd(c, acc[r]);
          else
            *c = acc[r];
        }
      }
    }
  }
}

template <bool TA, bool TB, typename AT, typename BT>
void launch16(const void* A, const void* B, void* C, int c_f16, int64_t M, int64_t N, int64_t Kd, int64_t lda,
              int64_t ldb, int64_t k_per_split, int split_k, int alA, int alB, hipStream_t st) {
  dim3 grid((unsigned)cdiv64(N, BN), (unsigned)cdiv64(M, BM), (unsigned)split_k), block(256);
  if (c_f16)
    hipLaunchKernelGGL((gemm_f16_mfma<TA, TB, AT, BT, true>), grid, block, 0, st, (const AT*)A, (const BT*)B, C, M, N, Kd,
                       lda, ldb, k_per_split, 0, alA, alB);
  else
    hipLaunchKernelGGL((gemm_f16_mfma<TA, TB, AT, BT, false>), grid, block, 0, st, (const AT*)A, (const BT*)B, C, M, N, Kd,
                       lda, ldb, k_per_split, split_k > 1, alA, alB);
}

template <typename AT, typename BT>
void launch16_t(int transA, int transB, const void* A, const void* B, void* C, int c_f16, int64_t M, int64_t N,
                int64_t Kd, int64_t lda, int64_t ldb, int64_t kps, int split_k, int alA, int alB, hipStream_t st) {
  if (!transA && !transB) launch16<false, false, AT, BT>(A, B, C, c_f16, M, N, Kd, lda, ldb, kps, split_k, alA, alB, st);
  else if (!transA && transB) launch16<false, true, AT, BT>(A, B, C, c_f16, M, N, Kd, lda, ldb, kps, split_k, alA, alB, st);
  else if (transA && !transB) launch16<true, false, AT, BT>(A, B, C, c_f16, M, N, Kd, lda, ldb, kps, split_k, alA, alB, st);
  else launch16<true, true, AT, BT>(A, B, C, c_f16, M, N, Kd, lda, ldb, kps, split_k, alA, alB, st);
}

}  // namespace

extern "C" int mvk_gemm_f16(const void* A, int a_f16, const void* B, int b_f16, void* C, int c_f16, int64_t M,
                            int64_t N, int64_t Kd, int transA, int transB, int split_k, void* stream) {
  MVK_REQUIRE(M >= 0 && N >= 0 && Kd >= 0, "gemm16: negative size");
  if (M == 0 || N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (Kd == 0) {
    if (split_k <= 1) MVK_CHECK_HIP(hipMemsetAsync(C, 0, (c_f16 ? 2 : 4) * M * N, st));
    return 0;
  }
  if (split_k < 1) split_k = 1;
  MVK_REQUIRE(!(c_f16 && split_k > 1), "gemm16: a split reduction needs the f32 output (atomics)");
  int64_t ksteps = cdiv64(Kd, BK);
  if (split_k > ksteps) split_k = (int)ksteps;
  const int64_t k_per_split = cdiv64(ksteps, split_k) * BK;
  split_k = (int)cdiv64(Kd, k_per_split);
  const int64_t lda = transA ? M : Kd, ldb = transB ? Kd : N;
    const int ea = a_f16 ? 2 : 4, eb = b_f16 ? 2 : 4;
  auto align_of = [](int64_t ld_bytes, const void* base) {      // bytes every row start (and 16-byte column step) shares
    int al = 16;
    while (al > 2 && ((ld_bytes % al) != 0 || ((uintptr_t)base % al) != 0)) al >>= 1;
    return al;
  };
  const int alA = align_of(lda * ea, A), alB = align_of(ldb * eb, B);
  MVK_REQUIRE(cdiv64(M, BM) < 65536 && split_k < 65536, "gemm16: grid too large");
  if (a_f16 && b_f16) launch16_t<_Float16, _Float16>(transA, transB, A, B, C, c_f16, M, N, Kd, lda, ldb, k_per_split, split_k, alA, alB, st);
  else if (a_f16) launch16_t<_Float16, float>(transA, transB, A, B, C, c_f16, M, N, Kd, lda, ldb, k_per_split, split_k, alA, alB, st);
  else if (b_f16) launch16_t<float, _Float16>(transA, transB, A, B, C, c_f16, M, N, Kd, lda, ldb, k_per_split, split_k, alA, alB, st);
  else launch16_t<float, float>(transA, transB, A, B, C, c_f16, M, N, Kd, lda, ldb, k_per_split, split_k, alA, alB, st);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
