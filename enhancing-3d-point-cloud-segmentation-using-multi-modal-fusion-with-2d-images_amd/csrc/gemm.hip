// fp32-in / fp32-accumulate GEMM on the gfx950 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fmaf chain, 64 FLOP/clk/SIMD = the f32 MFMA peak).
//
// This is the dense K x Cin x Cout contraction of KPConv (reference
// KPConv-PyTorch/models/blocks.py:370-374: permute + matmul + sum over K ==
// one [N, K*Cin] x [K*Cin, Cout] GEMM), its two backward
// products and the unary (nn.Linear) layers of the blocks (blocks.py:470-504).
// bf16/fp16 MFMA would miss the 1e-4 parity bar, so the f32 MFMA forms are used.
//
// Shapes are tall-skinny (19 464 x 990 x 64) or short-and-deep (85 x 7680 x 512), never square, and the
// chip has 1024 SIMDs to balance, so the ROW granularity of a workgroup tile is a launch choice:
//   * 256-thread workgroup = 4 waves; the 16 x 16 x 4 MFMA gives a row granularity of 16: the tile is
//     (16*PM*WM) x (16*QN*WN), every wave owns PM x QN accumulator blocks (4 VGPRs each). For outputs of >= 48
//     columns the waves split the columns (WM, WN = 1, 4; TN = 64 or 128) and every wave sweeps all 16*PM rows;
//     for <= 32 columns they split the rows (4, 1). Measured (plan_gemm below): the products are latency bound,
//     small row tiles with 3-5 workgroups per CU win everywhere, so the plan uses PM = 2 (32 rows) / 1 and only
//     chooses the split of the reduction; the larger instantiations stay selectable (MVK_GEMM_FORCE);
//   * operands staged through LDS in the layout their SOURCE is contiguous in: a k-contiguous source
//     ([rows][k], A of NN / NT, B of NT) is stored [row][40] and read with one ds_read_b128 per lane and four
//     k-steps (the k order inside a 16-deep chunk is permuted: step s of lane group g uses k = 4g + s -- both
//     operands agree, and a sum does not care); a row-contiguous source ([k][rows]) is stored [k][rows + 4]
//     and read with ds_read_b32. Both reads and both ds_write_b128 patterns are bank-conflict free;
//   * two LDS buffers + TWO register sets: the global loads of k-tile t+2 are issued before the MFMAs of
//     tile t and written to LDS after those of tile t+1 -> one barrier per k-tile, two tiles in flight;
//   * split-K (grid.z) with f32 atomics only where the output is small (coarse layers, dW);
//   * optional epilogue for the BatchNorm that follows (blocks.py:456-460): per (wave row block, column) the
//     sum and the centred sum of squares over the rows below a DEVICE-side row count, straight from the
//     accumulators -- the separate statistics pass over the GEMM output disappears.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

// One operand tile = ROWS (i) x 32 (k) elements staged by 256 threads.
//   CONTIG_K == true : source element (i,k) at src[i*ld + k]; LDS image [i][40]
//   CONTIG_K == false: source element (i,k) at src[k*ld + i]; LDS image [k][ROWS + 4]
template <bool CONTIG_K, int ROWS>
struct Tile {
  static constexpr int LD = CONTIG_K ? 40 : ROWS + 4;
  static constexpr int FLOATS = CONTIG_K ? ROWS * 40 : BK * (ROWS + 4);
  static constexpr int NF4 = ROWS * BK / 4;        // float4 of one tile
  static constexpr int NV = (NF4 + 255) / 256;     // float4 per thread
  static constexpr int IPT = ROWS / 4;             // threads along i (row-contiguous sources)
  float4 v[NV];

  // A-operand transform of a k-contiguous tile held in registers (GemmArgs (b)): xs = LDS table [k - kbase0][4] =
  // {mean, invstd * gamma, beta, -} of this workgroup's k-range, k0 = the tile's first k, rows >= nvalid -> 0. `out`
  // (column-tile-0 workgroups, else null): the transformed values also go to out[row * ld + k] for rows < imax, k < kmax.
  __device__ __forceinline__ void transform(const float* __restrict__ xs, int64_t k0, int64_t kbase0, int64_t i0, int64_t imax,
                                            int64_t kmax, int64_t nvalid, float slope, float* __restrict__ out, int64_t ld,
                                            int tid) {
    static_assert(CONTIG_K, "the transform is written for k-contiguous A tiles");
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = tid + 256 * u;
      if (NF4 % 256 == 0 || f < NF4) {
        const int i = f >> 3, kq = (f & 7) * 4;
        const int64_t gi = i0 + i, gk = k0 + kq;
        const float* t = xs + (gk - kbase0) * 4;
        float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 q = *reinterpret_cast<const float4*>(t + 4 * c);       // mean, invstd * gamma, beta
          float z = (e[c] - q.x) * q.y + q.z;
          z = z > 0.f ? z : z * slope;
          e[c] = gi < nvalid ? z : 0.f;
        }
        v[u] = make_float4(e[0], e[1], e[2], e[3]);
        if (out && gi < imax) {
          float* o = out + gi * ld + gk;
          if (gk + 3 < kmax && (ld & 3) == 0)
            *reinterpret_cast<float4*>(o) = v[u];
          else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (gk + c < kmax) o[c] = e[c];
          }
        }
      }
    }
  }

  // vec: 4 = rows 16-byte aligned, 2 = 8-byte aligned (e.g. K*Cin = 990), 1 = scalar
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int64_t i0, int64_t imax,
                                       int64_t k0, int64_t kmax, int vec, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      const int f = tid + 256 * u;
      if (NF4 % 256 == 0 || f < NF4) {
        if (CONTIG_K) {
          const int i = f >> 3, kq = (f & 7) * 4;
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
          const int k = f / IPT, iq = (f % IPT) * 4;
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
      }
      v[u] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }

  // Branch-free load of a k-tile that lies fully inside [k0, kmax): out-of-range rows are CLAMPED to the last valid
  // row / float4 (they only feed output rows / columns that are never stored) and threads beyond the tile
  // re-load its last float4 (their store is predicated), so no load result is merged under a predicate and the
  // loads stay asynchronous until the LDS write. Requires fast_ok().
  template <int VEC>
  __device__ __forceinline__ void load_fast(const float* __restrict__ src, int64_t ld, int64_t i0, int64_t imax,
                                            int64_t k0, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int f = tid + 256 * u;
      if (NF4 % 256 != 0) f = f < NF4 ? f : NF4 - 1;
      const float* p;
      if (CONTIG_K) {
        const int i = f >> 3, kq = (f & 7) * 4;
        int64_t gi = i0 + i;
        gi = gi < imax ? gi : imax - 1;
        p = src + gi * ld + (k0 + kq);
      } else {
        const int k = f / IPT, iq = (f % IPT) * 4;
        int64_t gi = i0 + iq;
        gi = gi + 3 < imax ? gi : imax - 4;
        p = src + (k0 + k) * ld + gi;
      }
      if (VEC == 4) {
        // (a typed vector load, not a float4 struct copy: a struct copy is a memcpy in the IR, which keeps register sets
        // held in an ARRAY of tiles in scratch -- found with the whole-range prefetch experiment of round 5)
        const f32x4 q = *reinterpret_cast<const f32x4*>(p);
        v[u] = make_float4(q[0], q[1], q[2], q[3]);
      } else {
        const float2 q0 = *reinterpret_cast<const float2*>(p), q1 = *reinterpret_cast<const float2*>(p + 2);
        v[u] = make_float4(q0.x, q0.y, q1.x, q1.y);
      }
    }
  }

  // whether load_fast may be used for full k-tiles of this operand (wave-uniform)
  static __device__ __forceinline__ bool fast_ok(int64_t imax, int vec) {
    return CONTIG_K ? (vec >= 2 && imax >= 1) : (vec == 4 && imax >= 4 && (imax & 3) == 0);
  }

  // VEC = 4 / 2: load_fast (full k-tiles only); VEC = 0: the general predicated load
  template <int VEC>
  __device__ __forceinline__ void fetch(const float* __restrict__ src, int64_t ld, int64_t i0, int64_t imax, int64_t k0,
                                        int64_t kmax, int vec, int tid) {
    if (VEC == 0)
      load(src, ld, i0, imax, k0, kmax, vec, tid);
    else
      load_fast<VEC>(src, ld, i0, imax, k0, tid);
  }

  __device__ __forceinline__ void store(float* __restrict__ T, int tid) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = tid + 256 * u;
      if (NF4 % 256 == 0 || f < NF4) {
        const f32x4 q = {v[u].x, v[u].y, v[u].z, v[u].w};
        if (CONTIG_K) {
          const int i = f >> 3, kq = (f & 7) * 4;
          *reinterpret_cast<f32x4*>(&T[i * LD + kq]) = q;
        } else {
          const int k = f / IPT, iq = (f % IPT) * 4;
          *reinterpret_cast<f32x4*>(&T[k * LD + iq]) = q;
        }
      }
    }
  }

  // operand fragment of the 16-row block at r0 for the 16-deep k chunk j: element s feeds k-step s
  // (lane group g = lane >> 4 holds k = 16 j + 4 g + s)
  static __device__ __forceinline__ f32x4 frag(const float* __restrict__ T, int r0, int j, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if (CONTIG_K) {
      return *reinterpret_cast<const f32x4*>(&T[(r0 + i) * LD + 16 * j + 4 * g]);
    } else {
      f32x4 r;
      const float* p = &T[(16 * j + 4 * g) * LD + r0 + i];
      r[0] = p[0];
      r[1] = p[LD];
      r[2] = p[2 * LD];
      r[3] = p[3 * LD];
      return r;
    }
  }
};

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int64_t M, N, Kd, lda, ldb, k_per_split;
  int atomic_out, accumulate, vecA, vecB;
  float* bn_part;          // [ceil(M / TM), 2, N]: column sum and centred sum of squares per workgroup row block
  const int32_t* n_valid;  // device row count for the statistics (null: M)
  const float* bias;       // [N] or null: C = LeakyReLU_slope(A.op(B) + bias) -- the BatchNorm-less layers (blocks.py:462-463);
  float act_slope;         // plain stores only (no split, no accumulate)
  // DUAL instantiations: C = A.op(B) + A2.op(B2), the second product's reduction appended to the first's (Kd a multiple
  // of the k-tile): the gradient of a tensor that feeds two linear layers (unary1 and the shortcut of a bottleneck block)
  const float* A2;
  const float* B2;
  int64_t Kd2, lda2, ldb2;
  int vecA2, vecB2;
  // scatter epilogue (null sc_idx: off): C is the gradient of [x[idx[m,0]] | skip[m]] (the decoder's upsampling +
  // concatenation, architectures.py:334-335) and is never stored as such -- its first sc_c1 columns are added onto row
  // idx[m,0] of sc_dst [sc_ns, sc_c1] (f32 atomics, zero-initialised by the caller), the rest goes to sc_rest [M, N - sc_c1]
  const void* sc_idx;
  int sc_idx64, sc_c1;
  int64_t sc_stride, sc_ns;
  float* sc_dst;
  float* sc_rest;
  // ordered split reduction (null det_ws: a split adds its partial sums with f32 atomics): every workgroup of a split
  // product parks its accumulators in its slot of det_ws [tile][split][TM * TN]; the workgroup that arrives LAST at the
  // tile's counter det_cnt[tile] adds the slots in the order 0 .. nsplit-1 and runs the ordinary epilogue (plain store,
  // accumulate, bias, scatter, BatchNorm statistics) -- the sum no longer depends on the order the workgroups ran in,
  // and C needs no zero fill. The counter is back at zero when the launch ends.
  float* det_ws;
  int* det_cnt;
  int det_gx, det_nsplit;
  // segmented B (seg_shift < 0: off; NT products only): the reduction is a sequence of segments of 2^seg_shift elements
  // (a multiple of the k-tile) and B's row pointer advances by seg_extra extra floats from one segment to the next --
  // B[(k, o), c] = W[k, c, o] read in place from a KPConv weight tensor [K, Cin, Cout] (segment = one kernel point:
  // 2^seg_shift = Cout, ldb = Cout, seg_extra = Cin * Cout - Cout): the per-kernel-point transposed weights of the
  // gather-form feature gradient, without a transposed copy of the weights.
  int seg_shift;
  int64_t seg_extra;
  // Round 5: the BatchNorm around a product, folded into it.
  // (a) FINISHED statistics (fin_cnt != null; needs bn_part): after writing its partials every workgroup of a column tile
  //     bumps fin_cnt[column tile]; the one that arrives last merges the fin_gy row-block partials of its columns in a
  //     fixed order (parallel-variance formula, as bn_sum_partials_m2 of csrc/bn.hip) and writes mean / invstd, the
  //     running statistics (blocks.py:456-460: nn.BatchNorm1d(momentum)) and, for column tile 0, the batch counter --
  //     the normalising launch (or the consumer's operand load, (b)) no longer reduces anything. Counters return to zero.
  int* fin_cnt;
  int fin_gy;
  float fin_eps, fin_momentum;
  float* fin_mean;
  float* fin_invstd;
  float* fin_rmean;
  float* fin_rvar;
  long long* fin_nbt;
  // (b) A-operand transform (XF instantiations; NT / NN products, ax_mean != null): the product reads
  //     A'(m, k) = m < *ax_nvalid ? LeakyReLU_slope((A(m,k) - mean[k]) * invstd[k] * gamma[k] + beta[k]) : 0
  //     i.e. the BatchNorm + activation between the producer of A and this layer (blocks.py:456-460, :639-644) applied
  //     while the tile is staged; the column-tile-0 workgroups also write A' to ax_out [M, Kd] (the tensor the backward's
  //     weight-gradient product reads), so nothing downstream changes.
  const float* ax_mean;
  const float* ax_invstd;
  const float* ax_gamma;
  const float* ax_beta;
  float ax_slope;
  const int32_t* ax_nvalid;
  float* ax_out;
};

constexpr int XF_KMAX = 512;     // longest k-range of one workgroup with an A-operand transform (LDS table of 4 floats per k)

template <bool TA, bool TB, int PM, int QN, int WM, int WN, bool DUAL = false, bool XF = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& a, const int bx, const int by, const int bz) {
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int TM = 16 * PM * WM, TN = 16 * QN * WN;
  typedef Tile<!TA, TM> TileA;   // A tile element (m,k): TA == false -> A[m*lda + k] (k-contiguous)
  typedef Tile<TB, TN> TileB;    // B tile element (n,k): TB == false -> B[k*ldb + n] (n-contiguous)
  __shared__ __attribute__((aligned(16))) float lds[2 * (TileA::FLOATS + TileB::FLOATS)];
  __shared__ __attribute__((aligned(16))) float xs[XF ? 4 * (XF_KMAX + BK) : 4];
  float* const As = lds;                          // two buffers of each operand
  float* const Bs = lds + 2 * TileA::FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t m0 = (int64_t)by * TM, n0 = (int64_t)bx * TN;
  const int64_t ktot = DUAL ? a.Kd + a.Kd2 : a.Kd;
  const int64_t kbeg_all = (int64_t)bz * a.k_per_split;
  const int64_t kend_all = kbeg_all + a.k_per_split < ktot ? kbeg_all + a.k_per_split : ktot;

  f32x4 acc[PM][QN];
#pragma unroll
  for (int p = 0; p < PM; ++p)
#pragma unroll
    for (int q = 0; q < QN; ++q) acc[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int64_t x_nvalid = 0;
  if (XF) {
    // this workgroup's slice of the transform table (host: k_per_split <= XF_KMAX); entries past Kd are never used by a
    // product term that is not multiplied by a zero-filled B row, but they are read: zeros
    for (int k = tid; k < XF_KMAX + BK; k += 256) {
      const int64_t gk = kbeg_all + k;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < a.Kd && k < a.k_per_split + BK) q = make_float4(a.ax_mean[gk], a.ax_invstd[gk] * a.ax_gamma[gk], a.ax_beta[gk], 0.f);
      *reinterpret_cast<float4*>(&xs[4 * k]) = q;
    }
    x_nvalid = a.ax_nvalid ? (int64_t)*a.ax_nvalid : a.M;
    __syncthreads();
  }
  float* const x_out = (XF && bx == 0) ? a.ax_out : nullptr;

  // Two register sets: the loads of k-tile t+2 are issued before the MFMAs of tile t and written to LDS after
  // the MFMAs of tile t+1 -- two k-tiles of global-memory latency budget, two tiles of bytes in flight per
  // workgroup (the A operand streams from HBM: 77 MB for the first layer). The loop is unrolled by two so
  // that the sets are compile-time registers.
  TileA la0, la1;
  TileB lb0, lb1;
  const int ra = wm * 16 * PM, rb = wn * 16 * QN;
  auto compute = [&](const float* __restrict__ Ta, const float* __restrict__ Tb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 fa[PM], fb[QN];
#pragma unroll
      for (int p = 0; p < PM; ++p) fa[p] = TileA::frag(Ta, ra + 16 * p, j, lane);
#pragma unroll
      for (int q = 0; q < QN; ++q) fb[q] = TileB::frag(Tb, rb + 16 * q, j, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int p = 0; p < PM; ++p)
#pragma unroll
          for (int q = 0; q < QN; ++q)
            acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][s], fb[q][s], acc[p][q], 0, 0, 0);
    }
  };
  float* const A0 = As;
  float* const A1 = As + TileA::FLOATS;
  float* const B0 = Bs;
  float* const B1 = Bs + TileB::FLOATS;
  // The pipeline over [kbeg, klim) with loaders fixed at compile time (VA, VB = 4 / 2: branch-free loads of full
  // k-tiles; 0: the general loader, which also masks a ragged k-tail). Selected ONCE per workgroup below: a loader
  // chosen inside the loop would merge the loaded registers of its alternatives at every join, and each such
  // copy is a wait for the load it copies -- the loads would be synchronous again.
  // one operand pair over its own k-range [kbeg, kend): the pipelined full tiles, then the ragged tail
  auto run_range = [&](const float* __restrict__ pA, const int64_t lda, const int vecA, const float* __restrict__ pB,
                       const int64_t ldb, const int vecB, int64_t kbeg, const int64_t kend) {
  // B's base for the k-tile at k (segmented B: every segment starts seg_extra floats further than its position implies)
  auto Bat = [&](const int64_t k) -> const float* {
    return (TB && !DUAL && a.seg_shift >= 0) ? pB + (k >> a.seg_shift) * a.seg_extra : pB;
  };
  // publishes an A tile (first k: kt) in LDS, through the operand transform in the XF instantiations
  auto putA = [&](TileA& la, float* __restrict__ T, const int64_t kt) {
    if constexpr (XF) la.transform(xs, kt, kbeg_all, m0, a.M, a.Kd, x_nvalid, a.ax_slope, x_out, lda, tid);
    la.store(T, tid);
  };
  auto pipeline = [&](auto va, auto vb, const int64_t klim) {
    constexpr int VA = decltype(va)::value, VB = decltype(vb)::value;
    if (kbeg >= klim) return;
    la0.template fetch<VA>(pA, lda, m0, a.M, kbeg, klim, vecA, tid);
    lb0.template fetch<VB>(Bat(kbeg), ldb, n0, a.N, kbeg, klim, vecB, tid);
    if (kbeg + BK < klim) {
      la1.template fetch<VA>(pA, lda, m0, a.M, kbeg + BK, klim, vecA, tid);
      lb1.template fetch<VB>(Bat(kbeg + BK), ldb, n0, a.N, kbeg + BK, klim, vecB, tid);
    }
    putA(la0, A0, kbeg);
    lb0.store(B0, tid);
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < klim; k0 += 2 * BK) {
      // even k-tile (buffer 0): fetch tile +2 into set 0, multiply, publish tile +1 (set 1) in buffer 1
      if (k0 + 2 * BK < klim) {
        la0.template fetch<VA>(pA, lda, m0, a.M, k0 + 2 * BK, klim, vecA, tid);
        lb0.template fetch<VB>(Bat(k0 + 2 * BK), ldb, n0, a.N, k0 + 2 * BK, klim, vecB, tid);
      }
      compute(A0, B0);
      if (k0 + BK >= klim) break;
      putA(la1, A1, k0 + BK);
      lb1.store(B1, tid);
      __syncthreads();
      // odd k-tile (buffer 1): fetch tile +3 into set 1, multiply, publish tile +2 (set 0) in buffer 0
      if (k0 + 3 * BK < klim) {
        la1.template fetch<VA>(pA, lda, m0, a.M, k0 + 3 * BK, klim, vecA, tid);
        lb1.template fetch<VB>(Bat(k0 + 3 * BK), ldb, n0, a.N, k0 + 3 * BK, klim, vecB, tid);
      }
      compute(A1, B1);
      if (k0 + 2 * BK >= klim) break;
      putA(la0, A0, k0 + 2 * BK);
      lb0.store(B0, tid);
      __syncthreads();
    }
  };
  const bool fastA = TileA::fast_ok(a.M, vecA), fastB = TileB::fast_ok(a.N, vecB);
  const int64_t kfull = kbeg + (kend - kbeg) / BK * BK;     // end of the full k-tiles of this split
  typedef std::integral_constant<int, 4> I4;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 0> I0;
  if (fastA && fastB && vecA == 4 && vecB == 4) {
    pipeline(I4(), I4(), kfull);
  } else if (fastA && fastB && vecA == 2 && vecB == 4) {      // K*Cin = 990: rows of A are only 8-byte aligned
    pipeline(I2(), I4(), kfull);
  } else {
    pipeline(I0(), I0(), kend);
    kbeg = kend;
  }
  if (kfull < kend && kbeg < kend) {     // ragged k-tail of a fast pipeline: one predicated tile, unpipelined
    __syncthreads();
    la0.load(pA, lda, m0, a.M, kfull, kend, vecA, tid);
    lb0.load(Bat(kfull), ldb, n0, a.N, kfull, kend, vecB, tid);
    putA(la0, A0, kfull);
    lb0.store(B0, tid);
    __syncthreads();
    compute(A0, B0);
  }
  };
  if (!DUAL) {
    run_range(a.A, a.lda, a.vecA, a.B, a.ldb, a.vecB, kbeg_all, kend_all);
  } else {
    // this workgroup's share [kbeg_all, kend_all) of the concatenated reduction: its part in the first product, then its
    // part in the second (a.Kd is a multiple of the k-tile, so is k_per_split: tiles never straddle the seam)
    if (kbeg_all < a.Kd) run_range(a.A, a.lda, a.vecA, a.B, a.ldb, a.vecB, kbeg_all, kend_all < a.Kd ? kend_all : a.Kd);
    if (kend_all > a.Kd) {
      if (kbeg_all < a.Kd) __syncthreads();       // the first range's last tile is still being read from LDS
      run_range(a.A2, a.lda2, a.vecA2, a.B2, a.ldb2, a.vecB2, kbeg_all > a.Kd ? kbeg_all - a.Kd : 0, kend_all - a.Kd);
    }
  }

  if (a.det_ws) {
    // Slots are written and read with agent-scope (sc1) buffer accesses (park_* of common.h), the counter with a relaxed
    // agent-scope atomic: no fence, no L2 write-back. Order: a workgroup's slot stores are complete (park_wait + barrier)
    // before thread 0 bumps the tile's counter; the last arriver reads the slots only after its own bump returned the
    // full count (its loads are issued after the barrier that follows the bump: program order, and the compiler does not
    // move memory operations across the barrier). Slot layout [accumulator block][thread][4]: one 16-byte access per lane.
    constexpr int NB = PM * QN, SLOT = NB * 4 * 256;
    const int64_t tile = (int64_t)by * a.det_gx + bx;
    // the tile's slots as one buffer resource (uniform base in SGPRs; offsets inside it stay far below 4 GB: <= 64 slots
    // of <= 32 KB); the accesses are counted by the compiler (common.h)
    const __amdgpu_buffer_rsrc_t slots = park_rsrc(a.det_ws + tile * a.det_nsplit * SLOT);
    const uint32_t lane_off = (uint32_t)tid * 16u;
#pragma unroll
    for (int p = 0; p < PM; ++p)
#pragma unroll
      for (int q = 0; q < QN; ++q)
        park_store4(slots, (uint32_t)bz * (SLOT * 4u) + (uint32_t)(p * QN + q) * 4096u + lane_off, acc[p][q]);
    park_wait();
    __syncthreads();              // every thread's stores have arrived; the operand tiles in LDS are dead
    int* const flag = reinterpret_cast<int*>(lds);
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(a.det_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == a.det_nsplit - 1;
      if (last) __hip_atomic_store(a.det_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();              // (the statistics epilogue reuses lds)
    if (!last) return;
#pragma unroll
    for (int p = 0; p < PM; ++p)
#pragma unroll
      for (int q = 0; q < QN; ++q) acc[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int ZC = NB >= 8 ? 2 : 4;      // slots in flight: up to 16 loads per thread and round trip
    for (int z0 = 0; z0 < a.det_nsplit; z0 += ZC) {
      f32x4 t[ZC][NB];
      // all loads of a round issued before the first add (slots beyond the split are clamped to the last one and skipped
      // when adding: no branch around a load keeps them back to back)
#pragma unroll
      for (int zz = 0; zz < ZC; ++zz) {
        const int z = z0 + zz < a.det_nsplit ? z0 + zz : a.det_nsplit - 1;
#pragma unroll
        for (int b = 0; b < NB; ++b) t[zz][b] = park_load4(slots, (uint32_t)z * (SLOT * 4u) + (uint32_t)b * 4096u + lane_off);
      }
#pragma unroll
      for (int zz = 0; zz < ZC; ++zz)
        if (z0 + zz < a.det_nsplit) {
#pragma unroll
          for (int p = 0; p < PM; ++p)
#pragma unroll
            for (int q = 0; q < QN; ++q) acc[p][q] += t[zz][p * QN + q];          // split order 0, 1, 2, ...: fixed
        }
    }
  }

  // C/D map of the 16 x 16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
  const int ci = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int64_t col = n0 + rb + 16 * q + ci;
    if (col < a.N) {
      const float bq = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int p = 0; p < PM; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + ra + 16 * p + g4 + r;
          if (row < a.M) {
            if (a.sc_idx) {
              const float v = acc[p][q][r];
              if (col < a.sc_c1) {
                const int j = a.sc_idx64 ? load_idx<true>(a.sc_idx, row * a.sc_stride, a.sc_ns)
                                         : load_idx<false>(a.sc_idx, row * a.sc_stride, a.sc_ns);
                if (j >= 0) atomicAdd(a.sc_dst + (int64_t)j * a.sc_c1 + col, v);
              } else {
                float* c2 = a.sc_rest + row * (a.N - a.sc_c1) + (col - a.sc_c1);
                if (a.atomic_out) atomicAdd(c2, v);
                else *c2 = v;
              }
              continue;
            }
            float* c = a.C + row * a.N + col;
            if (a.atomic_out)
              atomicAdd(c, acc[p][q][r]);
            else if (a.accumulate)
              *c += acc[p][q][r];
            else if (a.bias) {
              const float v = acc[p][q][r] + bq;
              *c = v > 0.f ? v : v * a.act_slope;
            } else
              *c = acc[p][q][r];
          }
        }
    }
  }

  if (a.bn_part) {
    // statistics of this workgroup's TM rows for the BatchNorm that consumes C (host guarantees one split and no
    // accumulate): rows below n_valid only; the sum, then the sum of squares about the block's OWN mean (combined
    // across blocks with the parallel-variance formula -- no E[x^2] - E[x]^2 cancellation). Waves that share
    // columns (narrow tiles: WM = 4) combine through LDS in a fixed order.
    int64_t nv = a.n_valid ? (int64_t)*a.n_valid : a.M;
    nv = nv < a.M ? nv : a.M;
    const int64_t wcnt = nv - (m0 + ra) < 0 ? 0 : (nv - (m0 + ra) > 16 * PM ? 16 * PM : nv - (m0 + ra));   // this wave's rows
    const int64_t bcnt = nv - m0 < 0 ? 0 : (nv - m0 > TM ? TM : nv - m0);                                  // the block's
    const int64_t blk = m0 / TM;
    float* red = lds;                      // [WM][TN] sums, then [WM][TN] centred squares (operand tiles are dead)
    // finished statistics: the partials cross to another workgroup inside this launch -> agent-scope stores (common.h)
    const __amdgpu_buffer_rsrc_t part_rs = park_rsrc(a.bn_part);
    auto put_part = [&](const int64_t col, const float sum, const float m2) {
      if (a.fin_cnt) {
        park_store1(part_rs, (uint32_t)(((blk * 2) * a.N + col) * 4), sum);
        park_store1(part_rs, (uint32_t)(((blk * 2 + 1) * a.N + col) * 4), m2);
      } else {
        a.bn_part[(blk * 2) * a.N + col] = sum;
        a.bn_part[(blk * 2 + 1) * a.N + col] = m2;
      }
    };
    if (WM > 1) __syncthreads();
    float s[QN], mu[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      s[q] = 0.f;
#pragma unroll
      for (int p = 0; p < PM; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * p + g4 + r < wcnt) s[q] += acc[p][q][r];
      s[q] += __shfl_xor(s[q], 16);
      s[q] += __shfl_xor(s[q], 32);
      if (WM > 1 && lane < 16) red[wm * TN + rb + 16 * q + ci] = s[q];
    }
    if (WM > 1) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        s[q] = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s[q] += red[w * TN + rb + 16 * q + ci];
      }
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      mu[q] = bcnt > 0 ? s[q] / (float)bcnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int p = 0; p < PM; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * p + g4 + r < wcnt) {
            const float d = acc[p][q][r] - mu[q];
            m2 += d * d;
          }
      m2 += __shfl_xor(m2, 16);
      m2 += __shfl_xor(m2, 32);
      if (WM > 1) {
        if (lane < 16) red[(WM + wm) * TN + rb + 16 * q + ci] = m2;
      } else {
        const int64_t col = n0 + rb + 16 * q + ci;
        if (lane < 16 && col < a.N && m0 < a.M) put_part(col, s[q], m2);
      }
    }
    if (WM > 1) {
      __syncthreads();
      if (wm == 0 && lane < 16) {
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          float m2 = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) m2 += red[(WM + w) * TN + rb + 16 * q + ci];
          const int64_t col = n0 + rb + 16 * q + ci;
          if (col < a.N && m0 < a.M) put_part(col, s[q], m2);
        }
      }
    }
    if (a.fin_cnt) {
      // ---- the last workgroup of this column tile finishes the statistics (GemmArgs (a)). Hand-off as in the ordered
      // split: stores acknowledged (park_wait) + barrier, then one relaxed agent-scope bump; the bump's return value
      // orders the last arriver's loads behind every other workgroup's stores.
      park_wait();
      __syncthreads();
      int* const flag = reinterpret_cast<int*>(lds);
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(a.fin_cnt + bx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == a.fin_gy - 1;
        if (last) __hip_atomic_store(a.fin_cnt + bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
      }
      __syncthreads();
      const int last = *flag;
      __syncthreads();
      if (!last) return;
      // thread -> (four columns c4, row-block lane bl): its blocks bl, bl + BL, ... in that order, FP (sum, M2) pairs of
      // float4 in flight (more would raise the register count of the small-tile instantiations above their main loop's); the BL lanes of a column meet in LDS and are added in lane order: one fixed order throughout.
      // Merge about the mean of block 0 (bn_sum_partials_m2 of csrc/bn.hip): S = sum_b sum_b, Q = sum_b [M2_b + n_b
      // (mean_b - ref)^2], mean = S / n, M2 = Q - n (mean - ref)^2. (host: N % 4 == 0)
      constexpr int C4 = TN / 4, BL = 256 / C4;
      const int c4 = tid % C4, bl = tid / C4;
      const int64_t col0 = n0 + 4 * c4;
      const int nb = nv > 0 ? (int)((nv + TM - 1) / TM) : 0;
      f32x4 X = {0.f, 0.f, 0.f, 0.f}, Y = {0.f, 0.f, 0.f, 0.f}, ref = {0.f, 0.f, 0.f, 0.f};
      if (col0 < a.N && nb > 0) {
        const float n0f = (float)(nv < TM ? nv : TM);
        ref = park_load4(part_rs, (uint32_t)(col0 * 4));
        ref = ref * (1.f / n0f);
        constexpr int FP = 4;
        for (int i0 = bl; i0 < nb; i0 += FP * BL) {
          f32x4 u[FP], w[FP];
#pragma unroll
          for (int t = 0; t < FP; ++t) {
            const int i = i0 + t * BL < nb ? i0 + t * BL : nb - 1;        // clamped, skipped when adding
            u[t] = park_load4(part_rs, (uint32_t)((((int64_t)i * 2) * a.N + col0) * 4));
            w[t] = park_load4(part_rs, (uint32_t)((((int64_t)i * 2 + 1) * a.N + col0) * 4));
          }
#pragma unroll
          for (int t = 0; t < FP; ++t)
            if (i0 + t * BL < nb) {
              const int64_t left = nv - (int64_t)(i0 + t * BL) * TM;
              const float ni = (float)(left < TM ? left : TM);
              const f32x4 d = u[t] * (1.f / ni) - ref;
              X += u[t];
              Y += w[t] + ni * d * d;
            }
        }
      }
      float* const q1 = lds;                   // [BL][TN]
      float* const q2 = lds + 256 * 4;         // [BL][TN]   (2 x 4 KB: every instantiation has more LDS than that)
      *reinterpret_cast<f32x4*>(&q1[bl * TN + 4 * c4]) = X;
      *reinterpret_cast<f32x4*>(&q2[bl * TN + 4 * c4]) = Y;
      // ref of a column: lane 0's value, through LDS as well
      float* const qr = lds + 2 * 256 * 4;     // [TN]
      if (bl == 0) *reinterpret_cast<f32x4*>(&qr[4 * c4]) = ref;
      __syncthreads();
      const int64_t col = n0 + tid;
      if (tid < TN && col < a.N) {
        float x = 0.f, y = 0.f;
#pragma unroll 4
        for (int l = 0; l < BL; ++l) {
          x += q1[l * TN + tid];
          y += q2[l * TN + tid];
        }
        float mu = 0.f, var = 0.f, is = 0.f;
        if (nv > 0) {
          mu = x / (float)nv;
          const float dm = mu - qr[tid];
          float M2 = y - (float)nv * dm * dm;
          M2 = M2 > 0.f ? M2 : 0.f;
          var = M2 / (float)nv;                 // biased
          is = rsqrtf(var + a.fin_eps);
        }
        a.fin_mean[col] = mu;
        a.fin_invstd[col] = is;
        if (a.fin_rmean && nv > 0) {
          const float unbiased = nv > 1 ? var * ((float)nv / (float)(nv - 1)) : var;
          a.fin_rmean[col] = (1.f - a.fin_momentum) * a.fin_rmean[col] + a.fin_momentum * mu;
          a.fin_rvar[col] = (1.f - a.fin_momentum) * a.fin_rvar[col] + a.fin_momentum * unbiased;
        }
      }
      if (a.fin_nbt && bx == 0 && tid == 0) *a.fin_nbt += 1;
    }
  }
}

template <bool TA, bool TB, int PM, int QN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmArgs a) {
  gemm_body<TA, TB, PM, QN, WM, WN>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// the same with the A-operand transform (GemmArgs (b)): wide tiles, A not transposed
template <bool TB, int PM>
__global__ __launch_bounds__(256) void gemm_f32_mfma_xf(const GemmArgs a) {
  gemm_body<false, TB, PM, 1, 1, 4, false, true>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Two independent products of the same operand layout, tile shape and row count in one launch: their column tiles side
// by side along blockIdx.x, each with its own split of the reduction (workgroups beyond a product's split leave at once).
// unary1 and the shortcut layer of a bottleneck block read the same input (blocks.py:596-649).
template <bool TA, bool TB, int PM, int QN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_mfma_pair(const GemmArgs a0, const GemmArgs a1, int gx0, int gz0, int gz1) {
  if ((int)blockIdx.x < gx0) {
    if ((int)blockIdx.z >= gz0) return;
    gemm_body<TA, TB, PM, QN, WM, WN>(a0, blockIdx.x, blockIdx.y, blockIdx.z);
  } else {
    if ((int)blockIdx.z >= gz1) return;
    gemm_body<TA, TB, PM, QN, WM, WN>(a1, blockIdx.x - gx0, blockIdx.y, blockIdx.z);
  }
}

// C = A . op(B) + A2 . op(B2): two products of the same output shape with their reductions laid end to end (gemm_body's
// DUAL form). dx of a tensor that feeds two linear layers.
template <bool TA, bool TB, int PM, int QN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_mfma_dual(const GemmArgs a) {
  gemm_body<TA, TB, PM, QN, WM, WN, true>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped launch: ONE grid walks a table of independent products of the same operand layout and tile shape (the
// weight gradients of a whole backward pass: ~100 TN products dW = A^T g, each a latency-bound launch of ~10 us on
// its own). Workgroup b finds its problem by a search over the prefix sums of the problems' workgroup counts.
struct GroupEntry {
  GemmArgs args;
  int gx, gy, gz;       // grid of this problem
  int wg_begin;         // first workgroup of this problem in the grouped grid
};

template <bool TA, bool TB, int PM, int QN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_mfma_grouped(const GroupEntry* __restrict__ tab, int n_prob) {
  const int b = blockIdx.x;
  int lo = 0, hi = n_prob - 1;          // last problem with wg_begin <= b (wave uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].wg_begin <= b)
      lo = mid;
    else
      hi = mid - 1;
  }
  const GroupEntry& e = tab[lo];
  const int local = b - e.wg_begin;
  const int bx = local % e.gx, by = (local / e.gx) % e.gy, bz = local / (e.gx * e.gy);
  GemmArgs a = e.args;
  // pointers read from a table in memory are generic to the compiler (flat_load / flat_atomic, which also count on
  // lgkmcnt and so serialise against the LDS pipeline): they are device allocations
  a.A = load_global_ptr(&e.args.A);
  a.B = load_global_ptr(&e.args.B);
  a.C = load_global_ptr(&e.args.C);
  a.bn_part = load_global_ptr(&e.args.bn_part);
  a.n_valid = load_global_ptr(&e.args.n_valid);
  a.det_ws = load_global_ptr(&e.args.det_ws);
  a.det_cnt = load_global_ptr(&e.args.det_cnt);
  gemm_body<TA, TB, PM, QN, WM, WN>(a, bx, by, bz);
}

// ---------------------------------------------------------------------------------------------- host side

struct Plan {
  int pm, qn, narrow, split;
};

// Arena of the ordered split reductions (mvk_gemm_split_arena): the host layer hands over one large slot buffer and one
// zero-initialised counter buffer; every split product takes a slice of both. A slice is in use only while its launch
// runs and the counters return to zero by themselves, so nothing is ever cleared -- what must never happen is two
// launches that may run CONCURRENTLY sharing a slice (wrong sums, a wrong "last arriver", a counter left non-zero for
// good). Round 5 makes that a checked property instead of an assumption about sizes:
//   * a launch on a CAPTURING stream takes its slices from the TOP of the arena, downwards, and keeps them for good: the
//     nodes of a graph replay again and again, and branches of one graph run side by side;
//   * an eager launch takes its slices from the BOTTOM, upwards. When the bottom region is used up (it would run into
//     the reserved top) the library waits for the device (hipDeviceSynchronize: every earlier owner has finished) and
//     starts again at offset 0 -- once per arena's worth of split products;
//   * a request that cannot be served without breaking these rules fails loudly ("arena too small ... MVK_GEMM_ARENA_MB").
// Host state behind a mutex: ctypes calls drop the GIL.
struct SplitArena {
  float* ws = nullptr;
  int* cnt = nullptr;
  int64_t ws_floats = 0, n_cnt = 0;
  int64_t ws_off = 0, cnt_off = 0;        // next eager slice (bottom region, grows upwards)
  int64_t ws_top = 0, cnt_top = 0;        // start of the region reserved by captured launches (grows downwards)
};
SplitArena g_arena;
std::mutex g_arena_mu;

bool ordered_splits() { return g_arena.ws != nullptr; }

// slices for `tiles` output tiles of `slot` floats each, split `split` ways, for a launch on `st`.
// 0: ok; 1: the request does not fit the arena at all; 2: the captured launches have used the arena up; -2: HIP error
int arena_take(hipStream_t st, int64_t tiles, int split, int64_t slot, float** ws, int** cnt) {
  const int64_t need = (tiles * split * slot + 63) / 64 * 64;
  std::lock_guard<std::mutex> lock(g_arena_mu);
  if (need > g_arena.ws_floats || tiles > g_arena.n_cnt) return 1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
    (void)hipGetLastError();
    cap = hipStreamCaptureStatusNone;
  }
  if (cap != hipStreamCaptureStatusNone) {
    // reserved for good; must not reach into what eager launches may still be using
    if (g_arena.ws_top - need < g_arena.ws_off || g_arena.cnt_top - tiles < g_arena.cnt_off) return 2;
    g_arena.ws_top -= need;
    g_arena.cnt_top -= tiles;
    *ws = g_arena.ws + g_arena.ws_top;
    *cnt = g_arena.cnt + g_arena.cnt_top;
    return 0;
  }
  // eager launches recycle the FIRST QUARTER of the arena (or what the captured slices left of it): a process that runs
  // eager steps before it captures its graphs (bench.py's eager figure, a warm-up) must not leave its high-water mark
  // where the captures' slices go
  const int64_t ws_cap = g_arena.ws_top < g_arena.ws_floats / 4 ? g_arena.ws_top : (need > g_arena.ws_floats / 4 ? need : g_arena.ws_floats / 4);
  const int64_t cnt_cap = g_arena.cnt_top < g_arena.n_cnt / 4 ? g_arena.cnt_top : (tiles > g_arena.n_cnt / 4 ? tiles : g_arena.n_cnt / 4);
  if (g_arena.ws_off + need > ws_cap || g_arena.cnt_off + tiles > cnt_cap) {
    if (need > g_arena.ws_top || tiles > g_arena.cnt_top) return 2;
    if (hipDeviceSynchronize() != hipSuccess) return -2;     // every earlier owner of the bottom region has finished
    g_arena.ws_off = 0;
    g_arena.cnt_off = 0;
  }
  *ws = g_arena.ws + g_arena.ws_off;
  *cnt = g_arena.cnt + g_arena.cnt_off;
  g_arena.ws_off += need;
  g_arena.cnt_off += tiles;
  return 0;
}

const char* arena_error(int rc) {
  return rc == 1 ? "the split-reduction arena is smaller than one product's parking space (MVK_GEMM_ARENA_MB)"
         : rc == 2 ? "the split-reduction arena is used up by the slices captured graphs hold for good: too small for the "
                     "captured steps of this process (MVK_GEMM_ARENA_MB)"
                   : "hipDeviceSynchronize failed while recycling the split-reduction arena";
}

// gives a split product its slices (or leaves it on the atomic path when no arena is set); != 0: arena_error(rc)
int make_ordered(hipStream_t st, GemmArgs& a, int split, int gx, int64_t gy, int64_t tm, int64_t tn) {
  a.det_ws = nullptr; a.det_cnt = nullptr; a.det_gx = gx; a.det_nsplit = split;
  if (split <= 1 || !ordered_splits()) return 0;
  const int rc = arena_take(st, (int64_t)gx * gy, split, tm * tn, &a.det_ws, &a.det_cnt);
  if (rc != 0) return rc;
  a.atomic_out = 0;
  return 0;
}

#define MVK_ORDERED(call)                                      \
  do {                                                         \
    const int _rc = (call);                                    \
    MVK_REQUIRE(_rc == 0, "gemm: %s", arena_error(_rc));       \
  } while (0)

// Measured on MI355X (tools/gemm_bench.py --sweep, device time of graph-captured launches): for every product
// shape of the networks the fastest tiles are the SMALL ones -- 32 rows per workgroup for wide outputs (PM = 2),
// 64 for narrow ones (PM = 1) -- with the reduction split until ~800-1200 workgroups are in flight (3-5 per CU):
// the products are bound by the latency of their operand streams, which only concurrency hides; larger row tiles
// (PM = 4, 5, 8) won nowhere by more than 2 %. The plan therefore fixes the tile and models only the split.
Plan plan_gemm(int64_t M, int64_t N, int64_t Kd, int split_req, bool want_stats) {
  const int narrow = N <= 32 ? 1 : 0;
  // with statistics: 64-row tiles (one partial per 64 rows keeps the BatchNorm's reduction of the partials short)
  const int pm = narrow ? 1 : (want_stats ? 4 : 2), qn = narrow ? (N <= 16 ? 1 : 2) : 1;
  const int64_t tm = narrow ? 64 * pm : 16 * pm, tn = narrow ? 16 * qn : 64 * qn;
  const int64_t tiles = cdiv64(M, tm) * cdiv64(N, tn), ksteps = cdiv64(Kd, BK);
  Plan best{pm, qn, narrow, 1};
  double best_t = 1e300, t1 = 1e300;
  for (int split = 1; split <= 64; ++split) {
    if (split_req > 0 && split != split_req) continue;
    if (split > 1 && ksteps / split < 4) break;               // keep >= 4 k-tiles per workgroup
    const int64_t kt = cdiv64(ksteps, split), wgs = tiles * split;
    // shader cycles (fitted to tools/gemm_bench.py --sweep, device times): latency chain of one workgroup (k-tiles x
    // ~1500 at ~5 resident workgroups per CU) per round of 1280 resident workgroups + MFMA work spread over the 256
    // CUs + the atomic epilogue of a split (bytes, plus a per-split term: small outputs contend on few addresses)
    double t = (double)cdiv64(wgs, 1280) * (kt * 1500.0 + 3000.0) + (double)wgs * kt * 256.0 * pm * qn / 256.0;
    if (split > 1) t += (double)M * N * split * 4.0 / 2160.0 + 500.0 * split;
    if (split == 1) t1 = t;
    if (t < best_t) {
      best_t = t;
      best.split = split;
    }
  }
  // statistics in the epilogue need the whole reduction in one workgroup: worth it unless a split saves more
  // than the separate statistics launch costs (~6 us)
  static double bonus = -1.0;
  if (bonus < 0.0) {
    const char* eb = getenv("MVK_GEMM_STATS_BONUS");
    bonus = eb ? atof(eb) : 14000.0;
  }
  if (want_stats && !ordered_splits() && t1 <= best_t + bonus) best.split = 1;   // (an ordered split keeps its statistics)
  const char* e = getenv("MVK_GEMM_FORCE");       // development override: "pm,qn,split"
  if (e) {
    int pm = 0, qn = 0, sp = 0;
    if (sscanf(e, "%d,%d,%d", &pm, &qn, &sp) == 3) {
      if (pm > 0) best.pm = pm;
      if (qn > 0 && !best.narrow) best.qn = qn;
      if (sp > 0 && split_req <= 0) best.split = sp;
    }
  }
  return best;
}

template <bool TA, bool TB, int PM, int QN, int WM, int WN>
void launch_one(dim3 grid, hipStream_t st, const GemmArgs& a) {
  hipLaunchKernelGGL((gemm_f32_mfma<TA, TB, PM, QN, WM, WN>), grid, dim3(256), 0, st, a);
}

template <bool TA, bool TB>
bool launch_cfg(const Plan& p, dim3 grid, hipStream_t st, const GemmArgs& a) {
  if (p.narrow) {
#define NARROW(PMv, QNv) \
  if (p.pm == PMv && p.qn == QNv) return launch_one<TA, TB, PMv, QNv, 4, 1>(grid, st, a), true
    NARROW(1, 1); NARROW(2, 1); NARROW(1, 2); NARROW(2, 2);
#undef NARROW
    return false;
  }
#define WIDE(PMv, QNv) \
  if (p.pm == PMv && p.qn == QNv) return launch_one<TA, TB, PMv, QNv, 1, 4>(grid, st, a), true
  WIDE(1, 1); WIDE(2, 1); WIDE(4, 1); WIDE(5, 1);
  WIDE(1, 2); WIDE(2, 2); WIDE(4, 2);
#undef WIDE
  return false;
}

template <bool TA, bool TB, int PM>
void launch_pair_one(dim3 grid, hipStream_t st, const GemmArgs& a0, const GemmArgs& a1, int gx0, int gz0, int gz1) {
  hipLaunchKernelGGL((gemm_f32_mfma_pair<TA, TB, PM, 1, 1, 4>), grid, dim3(256), 0, st, a0, a1, gx0, gz0, gz1);
}

template <bool TA, bool TB>
bool launch_pair_cfg(const Plan& p, dim3 grid, hipStream_t st, const GemmArgs& a0, const GemmArgs& a1, int gx0, int gz0, int gz1) {
  if (p.qn != 1) return false;
  if (p.pm == 2) return launch_pair_one<TA, TB, 2>(grid, st, a0, a1, gx0, gz0, gz1), true;
  if (p.pm == 4) return launch_pair_one<TA, TB, 4>(grid, st, a0, a1, gx0, gz0, gz1), true;
  return false;
}

// a product of <= 32 columns (the narrow tile class) beside a wide one with a short reduction: it rides on the wide
// tiles of its partner, half of its one column tile empty -- level 0's unary1 (64 -> 32) beside its shortcut (64 -> 128)
void pair_widen(Plan& a, const Plan& b, int64_t Kd) {
  if (a.narrow && !b.narrow && Kd <= 512) {
    a.narrow = 0;
    a.pm = b.pm;
    a.qn = 1;
    a.split = 1;
  }
}

void no_bn_fold(GemmArgs& a) {
  a.fin_cnt = nullptr; a.fin_gy = 0; a.fin_eps = 0.f; a.fin_momentum = 0.f; a.fin_mean = nullptr; a.fin_invstd = nullptr;
  a.fin_rmean = nullptr; a.fin_rvar = nullptr; a.fin_nbt = nullptr;
  a.ax_mean = nullptr; a.ax_invstd = nullptr; a.ax_gamma = nullptr; a.ax_beta = nullptr; a.ax_slope = 1.f;
  a.ax_nvalid = nullptr; a.ax_out = nullptr;
}

// the producer's half of a folded BatchNorm (GemmArgs (a)); gy = row tiles of the launch
int set_finish(GemmArgs& a, const mvk_bn_finish* fin, int64_t gy) {
  if (!fin) return 0;
  MVK_REQUIRE(a.bn_part != nullptr, "gemm: finished BatchNorm statistics need the statistics epilogue (plan: stat rows 0)");
  MVK_REQUIRE(fin->counters && fin->mean && fin->invstd && a.N % 4 == 0 && (uintptr_t)a.bn_part % 16 == 0,
              "gemm: finished statistics need counters, mean, invstd and a channel count that is a multiple of 4");
  MVK_REQUIRE((fin->running_mean == nullptr) == (fin->running_var == nullptr), "gemm: running_mean and running_var go together");
  a.fin_cnt = fin->counters; a.fin_gy = (int)gy; a.fin_eps = fin->eps; a.fin_momentum = fin->momentum;
  a.fin_mean = fin->mean; a.fin_invstd = fin->invstd; a.fin_rmean = fin->running_mean; a.fin_rvar = fin->running_var;
  a.fin_nbt = (long long*)fin->num_batches_tracked;
  return 0;
}

// fills the arguments of one product like mvk_gemm_f32_ex does; returns its split
int fill_args(GemmArgs& a, const Plan& p, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd,
              int transA, int transB, float* bn_part, const int32_t* n_valid) {
  int split = p.split;
  const int64_t ksteps = cdiv64(Kd, BK);
  if (split > ksteps) split = (int)ksteps;
  const int64_t k_per_split = cdiv64(ksteps, split) * BK;
  split = (int)cdiv64(Kd, k_per_split);
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.Kd = Kd;
  a.lda = transA ? M : Kd;
  a.ldb = transB ? Kd : N;
  a.k_per_split = k_per_split;
  a.atomic_out = split > 1;
  a.accumulate = 0;
  a.vecA = ((a.lda % 4 == 0) && ((uintptr_t)A % 16 == 0)) ? 4 : ((a.lda % 2 == 0) && ((uintptr_t)A % 8 == 0)) ? 2 : 1;
  a.vecB = ((a.ldb % 4 == 0) && ((uintptr_t)B % 16 == 0)) ? 4 : ((a.ldb % 2 == 0) && ((uintptr_t)B % 8 == 0)) ? 2 : 1;
  a.bn_part = bn_part;
  a.n_valid = n_valid;
  a.bias = nullptr;
  a.act_slope = 1.f;
  a.A2 = nullptr; a.B2 = nullptr; a.Kd2 = 0; a.lda2 = 0; a.ldb2 = 0; a.vecA2 = 1; a.vecB2 = 1;
  a.sc_idx = nullptr; a.sc_idx64 = 0; a.sc_c1 = 0; a.sc_stride = 0; a.sc_ns = 0; a.sc_dst = nullptr; a.sc_rest = nullptr;
  a.det_ws = nullptr; a.det_cnt = nullptr; a.det_gx = 0; a.det_nsplit = split;
  a.seg_shift = -1; a.seg_extra = 0;
  no_bn_fold(a);
  return split;
}

}  // namespace

bool mvk_internal_arena_take(void* stream, int64_t floats, int64_t counters, float** ws, int** cnt) {
  if (!ordered_splits() || floats < 0 || counters < 0) return false;
  return arena_take((hipStream_t)stream, counters > 0 ? counters : 1, 1, cdiv64(floats, counters > 0 ? counters : 1), ws, cnt) == 0;
}

// Hands the library the arena of its ordered split reductions: `ws` (bytes, HBM) for the parked partial tiles, `counters`
// (n_counters int32, HBM, ZERO on entry and never touched by the caller again). ws == null: back to f32 atomics onto a
// zero-initialised output. With an arena set, a split product writes every element of C itself (no zero fill needed),
// sums its partial tiles in a fixed order (run-to-run bit-identical) and may carry the BatchNorm statistics epilogue.
extern "C" int mvk_gemm_split_arena(void* ws, int64_t ws_bytes, void* counters, int64_t n_counters) {
  MVK_REQUIRE((ws == nullptr) == (counters == nullptr) && ws_bytes >= 0 && n_counters >= 0, "gemm arena: bad arguments");
  MVK_REQUIRE(ws == nullptr || (ws_bytes >= (1 << 20) && n_counters >= 4096 && (uintptr_t)ws % 256 == 0),
              "gemm arena: at least 1 MB of 256-byte aligned slots and 4096 counters");
  std::lock_guard<std::mutex> lock(g_arena_mu);
  g_arena.ws = (float*)ws;
  g_arena.cnt = (int*)counters;
  g_arena.ws_floats = ws ? ws_bytes / 4 : 0;
  g_arena.n_cnt = ws ? n_counters : 0;
  g_arena.ws_off = 0;
  g_arena.cnt_off = 0;
  g_arena.ws_top = g_arena.ws_floats;
  g_arena.cnt_top = g_arena.n_cnt;
  return 0;
}

// 1 when split reductions are ordered (an arena is set), 0 when they use atomics (outputs must then be zero-initialised)
extern "C" int mvk_gemm_split_ordered(void) { return ordered_splits() ? 1 : 0; }

// Split of mvk_gemm_f32_dual's concatenated reduction (the caller zeroes C when > 1); 0: the shape is not supported (the
// first reduction must be a whole number of k-tiles, the output wider than 32 columns).
extern "C" int mvk_gemm_f32_dual_plan(int64_t M, int64_t N, int64_t Kd, int64_t Kd2, int* out_split) {
  MVK_REQUIRE(M >= 0 && N >= 0 && Kd >= 0 && Kd2 >= 0 && out_split, "gemm dual plan: bad arguments");
  *out_split = 0;
  if (M == 0 || N == 0 || Kd == 0 || Kd2 == 0 || Kd % BK != 0) return 0;
  const Plan p = plan_gemm(M, N, Kd + Kd2, 0, false);
  if (p.narrow || p.qn != 1 || p.pm != 2) return 0;
  int split = p.split;
  const int64_t ksteps = cdiv64(Kd + Kd2, BK);
  if (split > ksteps) split = (int)ksteps;
  const int64_t k_per_split = cdiv64(ksteps, split) * BK;
  *out_split = (int)cdiv64(Kd + Kd2, k_per_split);
  return 0;
}

// C [M,N] = A [M,Kd] . B [Kd,N] + A2 [M,Kd2] . B2 [Kd2,N] (all row-major, nothing transposed) in ONE launch: the two
// reductions end to end, split as mvk_gemm_f32_dual_plan says (atomics into a zero-initialised C when > 1).
extern "C" int mvk_gemm_f32_dual(const float* A, const float* B, const float* A2, const float* B2, float* C, int64_t M,
                                 int64_t N, int64_t Kd, int64_t Kd2, void* stream) {
  MVK_REQUIRE(M > 0 && N > 0 && Kd > 0 && Kd2 > 0 && Kd % BK == 0, "gemm dual: unsupported shape (ask mvk_gemm_f32_dual_plan)");
  const Plan p = plan_gemm(M, N, Kd + Kd2, 0, false);
  MVK_REQUIRE(!p.narrow && p.qn == 1 && p.pm == 2, "gemm dual: unsupported shape (ask mvk_gemm_f32_dual_plan)");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a;
  const int split = fill_args(a, p, A, B, C, M, N, Kd + Kd2, 0, 0, nullptr, nullptr);
  a.Kd = Kd;
  a.lda = Kd;
  a.vecA = ((a.lda % 4 == 0) && ((uintptr_t)A % 16 == 0)) ? 4 : ((a.lda % 2 == 0) && ((uintptr_t)A % 8 == 0)) ? 2 : 1;
  a.A2 = A2; a.B2 = B2; a.Kd2 = Kd2; a.lda2 = Kd2; a.ldb2 = N;
  a.vecA2 = ((a.lda2 % 4 == 0) && ((uintptr_t)A2 % 16 == 0)) ? 4 : ((a.lda2 % 2 == 0) && ((uintptr_t)A2 % 8 == 0)) ? 2 : 1;
  a.vecB2 = ((a.ldb2 % 4 == 0) && ((uintptr_t)B2 % 16 == 0)) ? 4 : ((a.ldb2 % 2 == 0) && ((uintptr_t)B2 % 8 == 0)) ? 2 : 1;
  MVK_REQUIRE(cdiv64(M, 32) < 65536 && split < 65536, "gemm: grid too large");
  dim3 grid((unsigned)cdiv64(N, 64), (unsigned)cdiv64(M, 32), (unsigned)split);
  MVK_ORDERED(make_ordered(st, a, split, (int)grid.x, grid.y, 32, 64));
  hipLaunchKernelGGL((gemm_f32_mfma_dual<false, false, 2, 1, 1, 4>), grid, dim3(256), 0, st, a);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// Plan of a pair (mvk_gemm_f32_pair): out[0] = 1 when C0 = A.op(B0), C1 = A.op(B1) can share one launch (both on the wide
// tile class with the same row tile), out[1..2] = the splits of the two reductions (their outputs must be zero-initialised
// when > 1), out[3..4] = the row-block sizes of their BatchNorm partials (0: none).
extern "C" int mvk_gemm_f32_pair_plan(int64_t M, int64_t N0, int64_t N1, int64_t Kd, int want_stats, int* out /* [5] */) {
  MVK_REQUIRE(M >= 0 && N0 >= 0 && N1 >= 0 && Kd >= 0 && out, "gemm pair plan: bad arguments");
  for (int i = 0; i < 5; ++i) out[i] = 0;
  if (M == 0 || N0 == 0 || N1 == 0 || Kd == 0) return 0;
  Plan p0 = plan_gemm(M, N0, Kd, 0, want_stats != 0), p1 = plan_gemm(M, N1, Kd, 0, want_stats != 0);
  pair_widen(p0, p1, Kd);
  if (p0.narrow || p1.narrow || p0.pm != p1.pm || p0.qn != 1 || p1.qn != 1 || (p0.pm != 2 && p0.pm != 4)) return 0;
  out[0] = 1;
  out[1] = p0.split;
  out[2] = p1.split;
  out[3] = (want_stats && (p0.split == 1 || ordered_splits())) ? 16 * p0.pm : 0;
  out[4] = (want_stats && (p1.split == 1 || ordered_splits())) ? 16 * p1.pm : 0;
  return 0;
}

// C0 [M,N0] = A . op(B0) and C1 [M,N1] = A . op(B1) in ONE launch (A [M,Kd] row-major, not transposed; transB as in
// mvk_gemm_f32_ex); plain stores (split reductions: atomics into zero-initialised outputs), statistics partials as in
// mvk_gemm_f32_ex where the plan gives a row-block size. Fails when mvk_gemm_f32_pair_plan does not allow the pair.
namespace {
int gemm_pair_run(const float* A, const float* B0, const float* B1, float* C0, float* C1, int64_t M, int64_t N0, int64_t N1,
                  int64_t Kd, int transB, int want_stats, float* bn_part0, float* bn_part1, const int32_t* n_valid,
                  const mvk_bn_finish* fin0, const mvk_bn_finish* fin1, void* stream);
}
extern "C" int mvk_gemm_f32_pair(const float* A, const float* B0, const float* B1, float* C0, float* C1, int64_t M,
                                 int64_t N0, int64_t N1, int64_t Kd, int transB, int want_stats, float* bn_part0,
                                 float* bn_part1, const int32_t* n_valid, void* stream) {
  return gemm_pair_run(A, B0, B1, C0, C1, M, N0, N1, Kd, transB, want_stats, bn_part0, bn_part1, n_valid, nullptr, nullptr, stream);
}
// mvk_gemm_f32_pair with the statistics of either output finished inside the launch (fin0 / fin1, null = not)
extern "C" int mvk_gemm_f32_pair_bn(const float* A, const float* B0, const float* B1, float* C0, float* C1, int64_t M,
                                    int64_t N0, int64_t N1, int64_t Kd, int transB, float* bn_part0, float* bn_part1,
                                    const int32_t* n_valid, const mvk_bn_finish* fin0, const mvk_bn_finish* fin1, void* stream) {
  return gemm_pair_run(A, B0, B1, C0, C1, M, N0, N1, Kd, transB, 1, bn_part0, bn_part1, n_valid, fin0, fin1, stream);
}
namespace {
int gemm_pair_run(const float* A, const float* B0, const float* B1, float* C0, float* C1, int64_t M, int64_t N0, int64_t N1,
                  int64_t Kd, int transB, int want_stats, float* bn_part0, float* bn_part1, const int32_t* n_valid,
                  const mvk_bn_finish* fin0, const mvk_bn_finish* fin1, void* stream) {
  MVK_REQUIRE(M > 0 && N0 > 0 && N1 > 0 && Kd > 0, "gemm pair: empty product");
  Plan p0 = plan_gemm(M, N0, Kd, 0, want_stats != 0), p1 = plan_gemm(M, N1, Kd, 0, want_stats != 0);
  pair_widen(p0, p1, Kd);
  MVK_REQUIRE(!p0.narrow && !p1.narrow && p0.pm == p1.pm && p0.qn == 1 && p1.qn == 1 && (p0.pm == 2 || p0.pm == 4),
              "gemm pair: the two products do not share a tile shape (ask mvk_gemm_f32_pair_plan first)");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a0, a1;
  const bool ord = ordered_splits();
  const int s0 = fill_args(a0, p0, A, B0, C0, M, N0, Kd, 0, transB, (want_stats && (p0.split == 1 || ord)) ? bn_part0 : nullptr, n_valid);
  const int s1 = fill_args(a1, p1, A, B1, C1, M, N1, Kd, 0, transB, (want_stats && (p1.split == 1 || ord)) ? bn_part1 : nullptr, n_valid);
  const int64_t tm = 16 * p0.pm, tn = 64;
  MVK_REQUIRE(cdiv64(M, tm) < 65536 && s0 < 65536 && s1 < 65536, "gemm: grid too large");
  const int gx0 = (int)cdiv64(N0, tn), gx1 = (int)cdiv64(N1, tn);
  MVK_ORDERED(make_ordered(st, a0, s0, gx0, cdiv64(M, tm), tm, tn));
  MVK_ORDERED(make_ordered(st, a1, s1, gx1, cdiv64(M, tm), tm, tn));
  MVK_REQUIRE(ord || ((s0 == 1 || !a0.bn_part) && (s1 == 1 || !a1.bn_part)), "gemm pair: statistics of a split product");
  if (int e = set_finish(a0, fin0, cdiv64(M, tm))) return e;
  if (int e = set_finish(a1, fin1, cdiv64(M, tm))) return e;
  dim3 grid((unsigned)(gx0 + gx1), (unsigned)cdiv64(M, tm), (unsigned)(s0 > s1 ? s0 : s1));
  const bool ok = transB ? launch_pair_cfg<false, true>(p0, grid, st, a0, a1, gx0, s0, s1)
                         : launch_pair_cfg<false, false>(p0, grid, st, a0, a1, gx0, s0, s1);
  MVK_REQUIRE(ok, "gemm pair: no kernel for plan pm=%d", p0.pm);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// Plan of mvk_gemm_f32_ex for a shape: the split of the reduction it will use (the caller zeroes C when > 1)
// and the row-block size of the BatchNorm partials (0: no statistics are produced for this shape).
extern "C" int mvk_gemm_f32_plan(int64_t M, int64_t N, int64_t Kd, int split_k, int want_stats, int* out_split,
                                 int* out_stat_rows) {
  MVK_REQUIRE(M >= 0 && N >= 0 && Kd >= 0 && out_split && out_stat_rows, "gemm plan: bad arguments");
  if (M == 0 || N == 0 || Kd == 0) {
    *out_split = 1;
    *out_stat_rows = 0;
    return 0;
  }
  const Plan p = plan_gemm(M, N, Kd, split_k > 0 ? split_k : 0, want_stats != 0);
  *out_split = p.split;
  *out_stat_rows = (want_stats && (p.split == 1 || ordered_splits())) ? (p.narrow ? 64 * p.pm : 16 * p.pm) : 0;
  return 0;
}

namespace {
struct ScatterOut {
  const void* idx;
  int idx64, c1;
  int64_t stride, ns;
  float* dst;
  float* rest;
};
int gemm_run(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transA, int transB,
             int accumulate, int split_k, float* bn_part, const int32_t* n_valid, const float* bias, float act_slope,
             void* stream, const ScatterOut* scatter = nullptr, int seg_shift = -1, int64_t seg_extra = 0, int64_t ldb = 0,
             const mvk_bn_finish* fin = nullptr, const mvk_a_transform* ax = nullptr);
}

extern "C" int mvk_gemm_f32_ex(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd,
                               int transA, int transB, int accumulate, int split_k, float* bn_part,
                               const int32_t* n_valid, void* stream) {
  return gemm_run(A, B, C, M, N, Kd, transA, transB, accumulate, split_k, bn_part, n_valid, nullptr, 1.f, stream);
}

// mvk_gemm_f32_ex (A not transposed, plain store) with the BatchNorm on either side of the product folded into it
// (include/mvkpconv.h): fin = the statistics of C finished inside this launch (needs bn_part: ask mvk_gemm_f32_plan),
// ax = the BatchNorm + LeakyReLU between A's producer and this layer applied while A is staged.
extern "C" int mvk_gemm_f32_bn(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transB,
                               float* bn_part, const int32_t* n_valid, const mvk_bn_finish* fin, const mvk_a_transform* ax,
                               void* stream) {
  MVK_REQUIRE(M > 0 && N > 0 && Kd > 0, "gemm bn: empty product");
  return gemm_run(A, B, C, M, N, Kd, 0, transB, 0, 0, bn_part, n_valid, nullptr, 1.f, stream, nullptr, -1, 0, 0, fin, ax);
}

// C = LeakyReLU_slope(A . op(B) + bias[col]) (slope 1: the bias alone): a BatchNorm-less layer (blocks.py:462-463, the two
// head layers of every network) in one launch. The reduction is never split (a plain store is required).
extern "C" int mvk_gemm_f32_bias_act(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transB,
                                     const float* bias, float slope, void* stream) {
  MVK_REQUIRE(bias != nullptr && slope > 0.f && Kd > 0, "gemm bias_act: bias, a positive slope and a non-empty product");
  return gemm_run(A, B, C, M, N, Kd, 0, transB, 0, 1, nullptr, nullptr, bias, slope, stream);
}

// dx [M, Cin] = sum_k A[:, k, :] [M, Cout] . W[k]^T [Cout, Cin] with A [M, K, Cout] row-major and W [K, Cin, Cout] the
// KPConv weight tensor read in place (segmented B, see GemmArgs): the contraction of the gather-form feature gradient
// (csrc/revlist.hip). Cout a power of two >= 32.
extern "C" int mvk_gemm_f32_kp_transposed(const float* A, const float* W, float* dx, int64_t M, int K, int Cin, int Cout,
                                          void* stream) {
  MVK_REQUIRE(K >= 1 && Cin >= 1 && Cout >= BK && (Cout & (Cout - 1)) == 0, "gemm kp_transposed: Cout must be a power of two >= 32");
  int shift = 0;
  while ((1 << shift) < Cout) ++shift;
  return gemm_run(A, W, dx, M, Cin, (int64_t)K * Cout, 0, 1, 0, 0, nullptr, nullptr, nullptr, 1.f, stream, nullptr, shift,
                  (int64_t)Cin * Cout - Cout);
}

// C [M,N] = A [M,Kd] . B with B [Kd, N] a column block of a wider row-major matrix (rows ldb >= N floats apart)
extern "C" int mvk_gemm_f32_ldb(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int64_t ldb,
                                void* stream) {
  MVK_REQUIRE(ldb >= N, "gemm ldb: ldb < N");
  return gemm_run(A, B, C, M, N, Kd, 0, 0, 0, 0, nullptr, nullptr, nullptr, 1.f, stream, nullptr, -1, 0, ldb);
}

// [d_x | d_skip] = A [M,Kd] . B [Kd,N] where the product is the gradient of cat([x[idx[m,0]], skip[m]]) (decoder:
// closest_pool + torch.cat + nn.Linear, architectures.py:334-335): the first c1 columns are scattered onto d_x [Ns,c1]
// (f32 atomics, zero-initialised by the caller), the others written to d_skip [M, N-c1] (atomics into a zero-initialised
// d_skip when the plan of (M, N, Kd) splits the reduction: mvk_gemm_f32_plan). The product itself is never stored.
extern "C" int mvk_gemm_f32_scatter_cat(const float* A, const float* B, int64_t M, int64_t N, int64_t Kd, const void* idx,
                                        int idx64, int64_t idx_stride, int64_t Ns, int c1, float* d_x, float* d_skip,
                                        void* stream) {
  MVK_REQUIRE(idx && d_x && d_skip && c1 > 0 && c1 < N && Kd > 0 && Ns >= 0 && idx_stride >= 1, "gemm scatter_cat: bad arguments");
  const ScatterOut sc{idx, idx64, c1, idx_stride, Ns, d_x, d_skip};
  return gemm_run(A, B, d_skip /* unused */, M, N, Kd, 0, 0, 0, 0, nullptr, nullptr, nullptr, 1.f, stream, &sc);
}

namespace {
int gemm_run(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transA, int transB,
             int accumulate, int split_k, float* bn_part, const int32_t* n_valid, const float* bias, float act_slope,
             void* stream, const ScatterOut* scatter, int seg_shift, int64_t seg_extra, int64_t ldb,
             const mvk_bn_finish* fin, const mvk_a_transform* ax) {
  MVK_REQUIRE(M >= 0 && N >= 0 && Kd >= 0, "gemm: negative size");
  if (M == 0 || N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  MVK_REQUIRE(!(bn_part && (accumulate || Kd == 0)), "gemm: BatchNorm statistics need a plain store of a non-empty product");
  if (Kd == 0) {
    if (!accumulate && split_k <= 1) MVK_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(float) * M * N, st));
    return 0;
  }
  MVK_REQUIRE(!(bn_part && split_k > 1 && !ordered_splits()), "gemm: BatchNorm statistics need an unsplit (or ordered) reduction");
  Plan p = plan_gemm(M, N, Kd, (bn_part && !ordered_splits()) ? 1 : (split_k > 0 ? split_k : 0), bn_part != nullptr);
  int split = p.split;
  const int64_t ksteps = cdiv64(Kd, BK);
  if (split > ksteps) split = (int)ksteps;
  const int64_t k_per_split = cdiv64(ksteps, split) * BK;
  split = (int)cdiv64(Kd, k_per_split);
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.Kd = Kd;
  a.lda = transA ? M : Kd;
  a.ldb = transB ? Kd : N;
  a.k_per_split = k_per_split;
  a.atomic_out = split > 1;
  a.accumulate = accumulate;
  a.vecA = ((a.lda % 4 == 0) && ((uintptr_t)A % 16 == 0)) ? 4 : ((a.lda % 2 == 0) && ((uintptr_t)A % 8 == 0)) ? 2 : 1;
  a.vecB = ((a.ldb % 4 == 0) && ((uintptr_t)B % 16 == 0)) ? 4 : ((a.ldb % 2 == 0) && ((uintptr_t)B % 8 == 0)) ? 2 : 1;
  if (ldb > 0) {                              // B = a column block of a wider matrix
    a.ldb = ldb;
    a.vecB = ((a.ldb % 4 == 0) && ((uintptr_t)B % 16 == 0)) ? 4 : ((a.ldb % 2 == 0) && ((uintptr_t)B % 8 == 0)) ? 2 : 1;
  }
  if (seg_shift >= 0) {
    a.ldb = (int64_t)1 << seg_shift;          // rows of W[k]: Cout floats
    a.vecB = ((uintptr_t)B % 16 == 0 && seg_extra % 4 == 0) ? 4 : 1;
  }
  a.bn_part = bn_part;
  a.n_valid = n_valid;
  a.bias = bias;
  a.act_slope = act_slope;
  a.A2 = nullptr; a.B2 = nullptr; a.Kd2 = 0; a.lda2 = 0; a.ldb2 = 0; a.vecA2 = 1; a.vecB2 = 1;
  a.sc_idx = nullptr; a.sc_idx64 = 0; a.sc_c1 = 0; a.sc_stride = 0; a.sc_ns = 0; a.sc_dst = nullptr; a.sc_rest = nullptr;
  a.seg_shift = seg_shift;
  a.seg_extra = seg_extra;
  no_bn_fold(a);
  if (scatter) {
    a.sc_idx = scatter->idx; a.sc_idx64 = scatter->idx64; a.sc_c1 = scatter->c1; a.sc_stride = scatter->stride;
    a.sc_ns = scatter->ns; a.sc_dst = scatter->dst; a.sc_rest = scatter->rest;
  }
  const int64_t tm = p.narrow ? 64 * p.pm : 16 * p.pm, tn = p.narrow ? 16 * p.qn : 64 * p.qn;
  MVK_REQUIRE(cdiv64(M, tm) < 65536 && split < 65536, "gemm: grid too large");
  dim3 grid((unsigned)cdiv64(N, tn), (unsigned)cdiv64(M, tm), (unsigned)split);
  MVK_ORDERED(make_ordered(st, a, split, (int)grid.x, grid.y, tm, tn));
  MVK_REQUIRE(!(bn_part && split > 1 && !a.det_ws), "gemm: BatchNorm statistics of a split product need the ordered reduction");
  if (int e = set_finish(a, fin, grid.y)) return e;
  if (ax) {
    // the consumer's half of a folded BatchNorm (GemmArgs (b)): wide NT tiles, the workgroup's k-range inside the table
    MVK_REQUIRE(!transA && transB && !p.narrow && p.qn == 1 && (p.pm == 2 || p.pm == 4) && k_per_split <= XF_KMAX && !scatter &&
                    seg_shift < 0 && ax->mean && ax->invstd && ax->gamma && ax->beta && ax->slope > 0.f,
                "gemm: the operand transform needs an NT product of more than 32 columns with k-ranges <= %d", XF_KMAX);
    a.ax_mean = ax->mean; a.ax_invstd = ax->invstd; a.ax_gamma = ax->gamma; a.ax_beta = ax->beta; a.ax_slope = ax->slope;
    a.ax_nvalid = ax->n_valid; a.ax_out = ax->out;
    if (p.pm == 2) hipLaunchKernelGGL((gemm_f32_mfma_xf<true, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_mfma_xf<true, 4>), grid, dim3(256), 0, st, a);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  bool ok;
  if (!transA && !transB) ok = launch_cfg<false, false>(p, grid, st, a);
  else if (!transA && transB) ok = launch_cfg<false, true>(p, grid, st, a);
  else if (transA && !transB) ok = launch_cfg<true, false>(p, grid, st, a);
  else ok = launch_cfg<true, true>(p, grid, st, a);
  MVK_REQUIRE(ok, "gemm: no kernel for plan pm=%d qn=%d narrow=%d", p.pm, p.qn, p.narrow);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// One record of mvk_gemm_f32_tn_grouped (host side, 48 bytes): C [M,N] (+)= A^T B with A [Kd,M], B [Kd,N] row-major.
struct MvkGemmProblem {
  const float* A;
  const float* B;
  float* C;
  int64_t M, N, Kd;
};

extern "C" int64_t mvk_gemm_group_entry_bytes(void) { return (int64_t)sizeof(GroupEntry); }

// Split of the reduction of ONE product inside a grouped launch. A product launched alone splits until the chip is
// full (plan_gemm); in a grouped launch the other products fill it, and a split only has to bound the longest
// workgroup: at most `ktiles` k-tiles of 32 per workgroup. Every split costs M x N x 4 bytes of f32 atomics and a zero
// fill of the output. Measured (tools/dw_group_bench.py, the 51 products of the early-fusion net): one sphere,
// 16.5 GFLOP: 305 us with the stand-alone splits (81 MB of atomics), 297 us at 16 k-tiles (54 MB), 309 / 352 us at
// 40 / 80; eight spheres, 140 GFLOP: 1773 us stand-alone (323 MB), 1708 us at 84 k-tiles (87 MB), 1767 us at 170.
// The launch is bound by its MFMA pipeline (54 -> 82 TFLOP/s), not by the atomics: the policy is worth 2-4 %; putting
// the products with the longest workgroups first in the table (autograd records the first layer last) changes nothing.
// MVK_DW_GROUP_KTILES: 0 = the stand-alone plan, > 0 = fixed; default = by the size of the whole group, 16..96.
constexpr int GROUP_KTILES_MIN = 16, GROUP_KTILES_MAX = 96;

// Row tile of the wide class of the grouped launch in units of 16 rows (development: MVK_DW_WIDE_PM = 2 / 4 / 8).
int group_wide_pm() {
  static const int v = getenv("MVK_DW_WIDE_PM") ? atoi(getenv("MVK_DW_WIDE_PM")) : 4;   // 64-row tiles: -5 % on the product list of a step (tools/dw_group_bench.py), the long reductions re-read g half as often
  return v == 2 || v == 8 ? v : 4;
}

int group_ktiles_env() {
  static const int v = getenv("MVK_DW_GROUP_KTILES") ? atoi(getenv("MVK_DW_GROUP_KTILES")) : -1;
  return v;
}

int grouped_split_at(int64_t M, int64_t N, int64_t Kd, int ktiles) {
  if (ktiles <= 0) return plan_gemm(M, N, Kd, 0, false).split;
  int64_t split = cdiv64(cdiv64(Kd, BK), ktiles);
  return (int)(split > 64 ? 64 : (split < 1 ? 1 : split));
}

// Upper bound of the split the grouped launch may give this product (the caller zero-initialises the output when > 1).
extern "C" int mvk_gemm_f32_tn_grouped_split(int64_t M, int64_t N, int64_t Kd) {
  const int e = group_ktiles_env();
  return grouped_split_at(M, N, Kd, e >= 0 ? e : GROUP_KTILES_MIN);
}

// Fills `table_host` (n * mvk_gemm_group_entry_bytes() bytes, narrow problems first) for the device-side grouped launch
// and reports how many of the problems are narrow (N <= 32) and the workgroup counts of the two launches.
extern "C" int mvk_gemm_f32_tn_grouped_plan(const void* problems, int n, void* table_host, int* n_narrow,
                                            int64_t* wgs_narrow, int64_t* wgs_wide, int32_t* splits /* [n] out */,
                                            void* stream /* the stream mvk_gemm_f32_tn_grouped will launch on */) {
  MVK_REQUIRE(n >= 0 && problems && table_host && n_narrow && wgs_narrow && wgs_wide && splits, "grouped gemm: bad arguments");
  const MvkGemmProblem* pr = (const MvkGemmProblem*)problems;
  GroupEntry* tab = (GroupEntry*)table_host;
  int nn = 0;
  for (int i = 0; i < n; ++i) nn += pr[i].N <= 32 ? 1 : 0;
  int64_t wn = 0, ww = 0;
  int in = 0, iw = nn;
  int ktiles = group_ktiles_env();
  if (ktiles < 0) {      // by the size of the group: ~8 000 workgroup-sized pieces of the longest allowed length
    double units = 0.0;
    for (int i = 0; i < n; ++i)
      units += (double)cdiv64(pr[i].M, pr[i].N <= 32 ? 64 : 16 * group_wide_pm()) * cdiv64(pr[i].N, pr[i].N <= 32 ? 32 : 64) * cdiv64(pr[i].Kd, BK);
    const double t = units / 8000.0;
    ktiles = t < GROUP_KTILES_MIN ? GROUP_KTILES_MIN : (t > GROUP_KTILES_MAX ? GROUP_KTILES_MAX : (int)t);
  }
  for (int i = 0; i < n; ++i) {
    const MvkGemmProblem& q = pr[i];
    MVK_REQUIRE(q.M > 0 && q.N > 0 && q.Kd > 0 && q.A && q.B && q.C, "grouped gemm: empty problem %d", i);
    const bool narrow = q.N <= 32;
    MVK_REQUIRE(!narrow || q.N > 16, "grouped gemm: outputs of <= 16 columns are not grouped");
    int split = grouped_split_at(q.M, q.N, q.Kd, ktiles);
    const int64_t ksteps = cdiv64(q.Kd, BK);
    if (split > ksteps) split = (int)ksteps;
    const int64_t k_per_split = cdiv64(ksteps, split) * BK;
    split = (int)cdiv64(q.Kd, k_per_split);
    GroupEntry e;
    e.args.A = q.A; e.args.B = q.B; e.args.C = q.C; e.args.M = q.M; e.args.N = q.N; e.args.Kd = q.Kd;
    e.args.lda = q.M; e.args.ldb = q.N;      // TN: A [Kd,M], B [Kd,N]
    e.args.k_per_split = k_per_split;
    e.args.atomic_out = split > 1;
    e.args.accumulate = 0;
    e.args.vecA = ((q.M % 4 == 0) && ((uintptr_t)q.A % 16 == 0)) ? 4 : ((q.M % 2 == 0) && ((uintptr_t)q.A % 8 == 0)) ? 2 : 1;
    e.args.vecB = ((q.N % 4 == 0) && ((uintptr_t)q.B % 16 == 0)) ? 4 : ((q.N % 2 == 0) && ((uintptr_t)q.B % 8 == 0)) ? 2 : 1;
    e.args.bn_part = nullptr;
    e.args.n_valid = nullptr;
    e.args.bias = nullptr;
    e.args.act_slope = 1.f;
    e.args.A2 = nullptr; e.args.B2 = nullptr; e.args.Kd2 = 0; e.args.lda2 = 0; e.args.ldb2 = 0;
    e.args.vecA2 = 1; e.args.vecB2 = 1;
    e.args.sc_idx = nullptr; e.args.sc_idx64 = 0; e.args.sc_c1 = 0; e.args.sc_stride = 0; e.args.sc_ns = 0;
    e.args.sc_dst = nullptr; e.args.sc_rest = nullptr;
    e.args.seg_shift = -1; e.args.seg_extra = 0;
    no_bn_fold(e.args);
    const int64_t tm = narrow ? 64 : 16 * group_wide_pm(), tn = narrow ? 32 : 64;      // plan tiles: narrow (pm 1, qn 2), wide (pm 2, qn 1)
    e.gx = (int)cdiv64(q.N, tn);
    e.gy = (int)cdiv64(q.M, tm);
    e.gz = split;
    MVK_ORDERED(make_ordered((hipStream_t)stream, e.args, split, e.gx, e.gy, tm, tn));
    const int64_t wgs = (int64_t)e.gx * e.gy * e.gz;
    if (narrow) {
      e.wg_begin = (int)wn;
      wn += wgs;
      tab[in++] = e;
    } else {
      e.wg_begin = (int)ww;
      ww += wgs;
      tab[iw++] = e;
    }
    splits[i] = split;
  }
  MVK_REQUIRE(wn < (1ll << 31) && ww < (1ll << 31), "grouped gemm: grid too large");
  *n_narrow = nn;
  *wgs_narrow = wn;
  *wgs_wide = ww;
  return 0;
}

// Launches the table prepared by mvk_gemm_f32_tn_grouped_plan (now in DEVICE memory): at most two launches.
// Every C with a split > 1 (plan: splits[i]) must be zero-initialised by the caller.
extern "C" int mvk_gemm_f32_tn_grouped(const void* table_dev, int n, int n_narrow, int64_t wgs_narrow, int64_t wgs_wide,
                                       void* stream) {
  MVK_REQUIRE(n >= 0 && n_narrow >= 0 && n_narrow <= n, "grouped gemm: bad counts");
  hipStream_t st = (hipStream_t)stream;
  const GroupEntry* tab = (const GroupEntry*)table_dev;
  if (n_narrow > 0 && wgs_narrow > 0)
    hipLaunchKernelGGL((gemm_f32_mfma_grouped<true, false, 1, 2, 4, 1>), dim3((unsigned)wgs_narrow), dim3(256), 0, st, tab,
                       n_narrow);
  if (n - n_narrow > 0 && wgs_wide > 0) {
    if (group_wide_pm() == 4)
      hipLaunchKernelGGL((gemm_f32_mfma_grouped<true, false, 4, 1, 1, 4>), dim3((unsigned)wgs_wide), dim3(256), 0, st,
                         tab + n_narrow, n - n_narrow);
    else if (group_wide_pm() == 8)
      hipLaunchKernelGGL((gemm_f32_mfma_grouped<true, false, 8, 1, 1, 4>), dim3((unsigned)wgs_wide), dim3(256), 0, st,
                         tab + n_narrow, n - n_narrow);
    else
      hipLaunchKernelGGL((gemm_f32_mfma_grouped<true, false, 2, 1, 1, 4>), dim3((unsigned)wgs_wide), dim3(256), 0, st,
                         tab + n_narrow, n - n_narrow);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_gemm_f32(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transA,
                            int transB, int accumulate, int split_k, void* stream) {
  return mvk_gemm_f32_ex(A, B, C, M, N, Kd, transA, transB, accumulate, split_k, nullptr, nullptr, stream);
}
