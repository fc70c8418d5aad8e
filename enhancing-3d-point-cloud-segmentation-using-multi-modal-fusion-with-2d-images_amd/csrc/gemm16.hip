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
#include <stdlib.h>

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


// ----------------------------------------------------------------------------------------------------------------
// Streaming contraction for the rigid fp16 layers (gfx950: v_mfma_f32_16x16x32_f16, 8 halfs per lane and operand).
//
// y [M, N] = A16 [M, Kp] . Wt16 [N, Kp]^T with M in the tens of thousands, N = 32 or 64 and Kp <= 1024: 60 flop per
// byte of A, i.e. the product is bound by the stream of A (38.5 MB for 19 464 x 992), and the 64 x 64 x 32 LDS-staged
// kernel above moves that stream through LDS behind a barrier per 32-deep step (51 us). Here
//   * both operands are k-contiguous in memory with 64-byte aligned rows (the gather kernel writes the aggregate with
//     a row stride padded to 32 halfs, mvk_round_weights_f16 writes the weights transposed and zero padded), so an
//     MFMA operand fragment IS one 16-byte global load: no LDS staging, no conversion, no barrier in the k loop;
//   * the small operand is STATIONARY IN REGISTERS: wave w of a workgroup keeps the weight fragments of its quarter of
//     the reduction (SW steps of 32 x all N columns = SW*CT*4 VGPRs <= 128) for the whole launch;
//   * the rows stream past in 16-row tiles: per tile a wave issues SW 16-byte loads per lane (one tile ahead, second
//     register set), runs SW*CT MFMAs, and the four partial 16 x N blocks of the workgroup's waves meet in LDS (fixed
//     order: deterministic; no atomics), from where they are stored as whole 256-byte rows;
//   * BatchNorm statistics of the layer's output (blocks.py:456-460) for the workgroup's rows in the same pass
//     (shifted sums per thread, merged with the parallel-variance formula): same partials format as gemm.hip.
struct S16Args {
  const _Float16* A;
  const _Float16* Bt;
  float* C;
  int64_t M, lda, ldb;
  int N, steps, tiles_per_wg;
  const int* n_valid;
  float* bn_part;
};

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A 16-byte store the compiler does not see: global stores share the vmcnt counter with the loads and may retire out of
// order with them, so a store the compiler knows about between a load and its use turns the counted wait in front of
// that use into a wait for EVERYTHING in flight (measured in the first version of the streaming kernel: one drained
// pipeline per store phase). Hidden from its bookkeeping the waits stay counted; they are still sufficient: a wait
// for "at most k outstanding" then returns later than needed (the counter also holds the stores), never earlier --
// loads return in order among themselves, so with s hidden stores and j older loads, <= k outstanding means at least
// j + 1 loads have returned.
__device__ __forceinline__ void store_f4_hidden(float* p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v q = {v.x, v.y, v.z, v.w};
  // s_nop 2: the store reads its data registers late; a VALU write to them within two wait states would be stored
  // instead (the compiler pads its own wide stores, it cannot see into this one -- common.h, round 4)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" : : "v"(p), "v"(q) : "memory");
}

template <int CT, int SW, int NS>
__global__ __launch_bounds__(256, 1) void gemm_f16_stream(const S16Args a) {
  constexpr int N = 16 * CT, LDR = N + 4;       // + 4: the four row groups of an accumulator land in distinct banks
  constexpr int G = NS;                         // tiles per group = register sets: one set per tile of the group
  __shared__ __attribute__((aligned(16))) float red[G][4][16][LDR];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int s0 = w * SW;

  const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t ntiles = (a.M + 15) / 16;
  const int T = (int)(ntiles - tile0 < a.tiles_per_wg ? ntiles - tile0 : a.tiles_per_wg);
  const int64_t last = tile0 + (T > 0 ? T - 1 : 0);

  auto load_tile = [&](int64_t tile, h8 (&af)[SW]) {
    tile = tile < last ? tile : last;                      // tiles beyond the workgroup's last: loaded again, never computed
    int64_t row = tile * 16 + r;
    row = row < a.M ? row : a.M - 1;                       // clamped rows are never stored nor counted
    const _Float16* p = a.A + row * a.lda + g * 8;
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      int st = s0 + s;
      st = st < a.steps ? st : a.steps - 1;                // beyond the reduction: a real address, zero weights
      af[s] = *reinterpret_cast<const h8*>(p + st * 32);
    }
  };

  auto compute = [&](int slot, const h8 (&af)[SW], const h8 (&b)[SW][CT]) {
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], b[s][ct], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[slot][w][g * 4 + i][ct * 16 + r] = acc[ct][i];
  };

  int64_t nv = a.n_valid ? (int64_t)*a.n_valid : a.M;
  nv = nv < a.M ? nv : a.M;
  // statistics of this thread's (row in tile, 4 columns) over the tiles: count, shift, shifted sums
  const bool red_thread = tid < 64 * CT;
  const int rr = tid / (4 * CT), cq = (tid % (4 * CT)) * 4;
  float cnt = 0.f;
  float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), s1 = sh, s2 = sh;

  // NS register sets = NS tiles in flight per wave (one workgroup per CU: the registers of a second resident workgroup
  // are spent on depth instead -- what bounds the stream is bytes in flight per CU, 2 sets x 2 workgroups were 126 KB,
  // 6 sets are 190 KB). Phase 1 of a group of NS tiles: MFMAs of tile i on set i, then the loads of tile i + NS into
  // the set just freed: the queue of outstanding loads is [sets i .. NS-1, sets 0 .. i-1] on every path, the
  // compiler's counted waits are exact. Phase 2: the four waves' partial blocks are added in a fixed order and stored
  // as whole rows (hidden stores, see store_f4_hidden); statistics.
  // Prologue: the first NS tiles of A and the stationary weights go out back to back, no wait in between (the weight
  // rows are allocated 4 * SW * 32 halfs long and zero beyond the reduction, mvk_gemm_f16_stream_plan out[3]: every
  // wave loads SW whole steps unconditionally -- zeroing the surplus step in registers had put a wait for the weights
  // in front of the first load of A: one memory latency of a ~4-latency kernel).
  h8 af[NS][SW];
#pragma unroll
  for (int i = 0; i < NS; ++i) load_tile(tile0 + i, af[i]);
  h8 b[SW][CT];
#pragma unroll
  for (int s = 0; s < SW; ++s)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      b[s][ct] = *reinterpret_cast<const h8*>(a.Bt + (int64_t)(ct * 16 + r) * a.ldb + (s0 + s) * 32 + g * 8);
  for (int t0 = 0; t0 < T; t0 += G) {
    const int ng = T - t0 < G ? T - t0 : G;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i < ng) compute(i, af[i], b);
      load_tile(tile0 + t0 + i + NS, af[i]);
    }
    __syncthreads();
    if (red_thread) {
      for (int i = 0; i < ng; ++i) {
        const float4 p0 = *reinterpret_cast<const float4*>(&red[i][0][rr][cq]);
        const float4 p1 = *reinterpret_cast<const float4*>(&red[i][1][rr][cq]);
        const float4 p2 = *reinterpret_cast<const float4*>(&red[i][2][rr][cq]);
        const float4 p3 = *reinterpret_cast<const float4*>(&red[i][3][rr][cq]);
        float4 v;
        v.x = ((p0.x + p1.x) + p2.x) + p3.x;
        v.y = ((p0.y + p1.y) + p2.y) + p3.y;
        v.z = ((p0.z + p1.z) + p2.z) + p3.z;
        v.w = ((p0.w + p1.w) + p2.w) + p3.w;
        const int64_t row = (tile0 + t0 + i) * 16 + rr;
        if (row < a.M) store_f4_hidden(a.C + row * N + cq, v);
        if (a.bn_part != nullptr && row < nv) {
          if (cnt == 0.f) sh = v;
          cnt += 1.f;
          const float dx = v.x - sh.x, dy = v.y - sh.y, dz = v.z - sh.z, dw = v.w - sh.w;
          s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
          s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
      }
    }
    if (t0 + G < T) __syncthreads();                       // the next group overwrites the partial blocks
  }

  if (a.bn_part != nullptr) {
    // merge the 16 row positions of every column in a fixed order (parallel-variance formula, no cancellation)
    __syncthreads();
    float* L = &red[0][0][0][0];                       // [3][16][N]: count, mean, M2
    if (red_thread) {
      const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
      const float m[4] = {sh.x + s1.x * inv, sh.y + s1.y * inv, sh.z + s1.z * inv, sh.w + s1.w * inv};
      const float q[4] = {s2.x - s1.x * s1.x * inv, s2.y - s1.y * s1.y * inv, s2.z - s1.z * s1.z * inv, s2.w - s1.w * s1.w * inv};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        L[(0 * 16 + rr) * N + cq + e] = cnt;
        L[(1 * 16 + rr) * N + cq + e] = m[e];
        L[(2 * 16 + rr) * N + cq + e] = q[e] > 0.f ? q[e] : 0.f;
      }
    }
    __syncthreads();
    if (tid < N && tile0 < ntiles) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
      for (int i = 0; i < 16; ++i) {
        const float ni = L[(0 * 16 + i) * N + tid], mi = L[(1 * 16 + i) * N + tid], qi = L[(2 * 16 + i) * N + tid];
        if (ni > 0.f) {
          const float nn = n + ni, d = mi - mean;
          mean += d * (ni / nn);
          m2 += qi + d * d * (n * ni / nn);
          n = nn;
        }
      }
      a.bn_part[((int64_t)blockIdx.x * 2) * N + tid] = mean * n;      // the block's sum
      a.bn_part[((int64_t)blockIdx.x * 2 + 1) * N + tid] = m2;        // squares about the block's own mean
    }
  }
}

// Weights of a KPConv layer for the fp16 mode in one launch: W [Kd, N] f32 -> Wt16 [N, Kp] fp16 (transposed, k
// contiguous, zero padded to Kp) and, optionally, the rounded values back in f32 [Kd, N] for the f32 backward product
// dA = g . W16^T. One thread per (k, n): reads coalesced over n.
__global__ __launch_bounds__(256) void round_weights_f16_kernel(const float* __restrict__ W, int64_t Kd, int N, int64_t Kp,
                                                               _Float16* __restrict__ Wt, float* __restrict__ Wr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= Kp * N) return;
  const int64_t k = i / N;
  const int n = (int)(i - k * N);
  _Float16 h = (_Float16)0.f;
  if (k < Kd) {
    h = (_Float16)W[k * N + n];
    if (Wr) Wr[k * N + n] = (float)h;
  }
  Wt[(int64_t)n * Kp + k] = h;
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

// ---- streaming contraction (rigid fp16 layers) -----------------------------------------------------------------

// Which launch the streaming kernel would use for y [M,N] = A16 [M,Kp] . Wt16 [N,Kp]^T: out[0] = 1 when supported
// (N = 32 or 64, Kp a multiple of 32 with Kp / 32 <= 32), out[1] = 16-row tiles per workgroup (= rows per statistics
// block / 16), out[2] = workgroups, out[3] = halfs per row the weight operand must be allocated with (zeros beyond
// the reduction; >= Kp). MVK_GEMM16_TILES overrides the tiles per workgroup (development).
static int stream_sw(int64_t steps) { return steps <= 4 ? 1 : (steps <= 8 ? 2 : (steps <= 16 ? 4 : 8)); }

extern "C" int mvk_gemm_f16_stream_plan(int64_t M, int N, int64_t Kp, int64_t* out) {
  MVK_REQUIRE(out != nullptr, "gemm16 stream plan: null output");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (M <= 0 || (N != 32 && N != 64) || Kp <= 0 || Kp % 32 != 0 || Kp / 32 > 32) return 0;
  const int64_t ntiles = cdiv64(M, 16);
  int64_t T = cdiv64(ntiles, 256);                 // one workgroup per CU (the kernel spends its registers on tiles in flight), one round
  if (T < 2) T = ntiles >= 2 ? 2 : 1;              // amortise the load of the stationary weights
  if (const char* e = getenv("MVK_GEMM16_TILES")) {
    const long v = atol(e);
    if (v > 0) T = v;
  }
  out[0] = 1;
  out[1] = T;
  out[2] = cdiv64(ntiles, T);
  out[3] = 4 * 32 * stream_sw(Kp / 32);
  return 0;
}

extern "C" int mvk_gemm_f16_stream(const void* A16, int64_t lda, const void* Wt16, int64_t ldb, float* C, int64_t M,
                                   int N, int64_t Kp, const int* n_valid, float* bn_part, void* stream) {
  int64_t plan[4];
  if (int e = mvk_gemm_f16_stream_plan(M, N, Kp, plan)) return e;
  MVK_REQUIRE(plan[0] == 1, "gemm16 stream: unsupported shape M=%lld N=%d Kp=%lld", (long long)M, N, (long long)Kp);
  MVK_REQUIRE(lda >= Kp && ldb >= plan[3] && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A16 % 16) == 0 && ((uintptr_t)Wt16 % 16) == 0,
              "gemm16 stream: operands must be 16-byte aligned, row strides multiples of 8 halfs, weight rows >= %lld halfs (zero padded)",
              (long long)plan[3]);
  S16Args a;
  a.A = (const _Float16*)A16; a.Bt = (const _Float16*)Wt16; a.C = C; a.M = M; a.lda = lda; a.ldb = ldb; a.N = N;
  a.steps = (int)(Kp / 32); a.tiles_per_wg = (int)plan[1]; a.n_valid = n_valid; a.bn_part = bn_part;
  const int sw = stream_sw(a.steps);
  const dim3 grid((unsigned)plan[2]), block(256);
  hipStream_t st = (hipStream_t)stream;
  int ns = 6;                                      // register sets = tiles in flight per wave (development: MVK_GEMM16_SETS=4)
  if (const char* e = getenv("MVK_GEMM16_SETS")) ns = atoi(e) == 4 ? 4 : 6;
#define MVK_S16(CT, SW)                                                                         \
  do {                                                                                          \
    if (ns == 4) hipLaunchKernelGGL((gemm_f16_stream<CT, SW, 4>), grid, block, 0, st, a);       \
    else hipLaunchKernelGGL((gemm_f16_stream<CT, SW, 6>), grid, block, 0, st, a);               \
  } while (0)
  if (N == 64) {
    if (sw == 1) MVK_S16(4, 1); else if (sw == 2) MVK_S16(4, 2); else if (sw == 4) MVK_S16(4, 4); else MVK_S16(4, 8);
  } else {
    if (sw == 1) MVK_S16(2, 1); else if (sw == 2) MVK_S16(2, 2); else if (sw == 4) MVK_S16(2, 4); else MVK_S16(2, 8);
  }
#undef MVK_S16
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_round_weights_f16(const float* W, int64_t Kd, int N, int64_t Kp, void* Wt16, float* W_rounded,
                                     void* stream) {
  MVK_REQUIRE(Kd >= 0 && N > 0 && Kp >= Kd, "round_weights_f16: bad sizes");
  if (Kp == 0) return 0;
  MVK_REQUIRE(W && Wt16, "round_weights_f16: null pointer");
  hipLaunchKernelGGL(round_weights_f16_kernel, dim3((unsigned)cdiv64(Kp * N, 256)), dim3(256), 0, (hipStream_t)stream, W, Kd, N,
                     Kp, (_Float16*)Wt16, W_rounded);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
