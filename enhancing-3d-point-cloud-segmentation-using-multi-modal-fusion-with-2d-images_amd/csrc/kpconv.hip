// KPConv gather / correlate / aggregate kernels for gfx950 (CDNA4, wave64).
//
// Replaces the ~15 ATen ops of KPConv.forward (reference
// KPConv-PyTorch/models/blocks.py:277-367) and their autograd backward by two
// fused kernels that never materialise [N,H,3], [N,H,K,3], [N,H,K] or [N,H,Cin]:
//
//   gather  (forward)  A[n,k,c]  = sum_h w[n,h,k] * x+[idx[n,h], c]
//   scatter (backward) dx[idx[n,h],c] += sum_k w[n,h,k] * dA[n,k,c]
//
// w[n,h,k] is a pure function of geometry (q, s, kernel points, extent), so the
// backward recomputes it instead of storing [N,H,K].
//
// Work decomposition (one 64-lane wavefront per workgroup): see the comment on
// kpconv_gather_vec (forward, rigid) and kpconv_lane_channel (backward scatter dx, deformable forward:
// lane = channel, one point per wave, wave-uniform neighbour loop). The offset gradient of the deformable
// layers is csrc/deform.hip (lane = neighbour).
#include <type_traits>

#include "common.h"

#define KMAX 16

namespace {

struct KPParams {
  const float* q;
  const float* s;
  const void* idx;
  const float* x;
  const float* kp;
  const float* offsets;  // [Nq,K,3] or null
  float* min_d2;         // [Nq,K] or null
  int32_t* min_arg;      // [Nq,K] or null: column h of the first entry attaining min_d2 (for the backward)
  float* A;              // fwd: out [Nq,K,Cin]; bwd: in dA
  float* dx;             // bwd out [Ns,Cin]
  const float* g_min_d2; // bwd deformable
  float* d_offsets;      // bwd deformable out [Nq,K,3]
  int64_t Nq, Ns;
  int H, Cin, K;
  float extent;
  int influence, aggregation;
  int zero_skip;         // gather: neighbours without any influence become shadow entries (development switch MVK_GATHER_ZEROSKIP)
  const int32_t* order;  // vector gather (rigid): work list -- the wave working on slots w .. w+PPW-1 takes the points order[w ..] (a
                         // spatially sorted permutation of 0 .. Nq-1); results land in the points' own rows. null: slot = point
  int xcd_blocks;        // with `order`: the first xcd_blocks workgroups are dealt to the 8 XCDs as 8 contiguous runs of the work list
  // MFMA gather, KPM = 2 (gather-form feature gradient of a deformable layer): the kernel points belong to the NEIGHBOUR
  // rows -- nb_offsets [rows of s, K, 3] are added to kp (and the sum is used negated: the relation is transposed) -- and
  // every weight is multiplied by nb_mod [rows of s, K] (modulated layers; null: 1)
  const float* nb_offsets;
  const float* nb_mod;
};

__device__ __forceinline__ float influence_w(float d2, float extent, int influence) {
  // blocks.py:329-344
  if (influence == MVK_INFL_LINEAR) return fmaxf(1.0f - sqrtf(d2) / extent, 0.0f);
  if (influence == MVK_INFL_GAUSSIAN) {
    float sig = extent * 0.3f;
    return expf(-d2 / (2.0f * sig * sig + 1e-9f));
  }
  return 1.0f;
}

__device__ __forceinline__ void wave_sync_lds() {   // LDS hand-off inside ONE wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// Phase A (shared): fills rel[64] (xyz + neighbour row or -1) and w[64][16]. One WAVE works on its own rel / wl
// (the hand-offs are wave-level: a workgroup may hold several independent waves).
// Lane mapping: A1 lane = ph = p*HC + h ; A2 item = t*64 + lane, k = lane & 15, ph = item >> 4.
// WPAD = extra floats between consecutive points' weight blocks (bank spreading).
// Returns the lane's neighbour row j (A1 mapping) so callers can shuffle it.
// ---------------------------------------------------------------------------
template <int PPW, int HC, int WPAD, bool IDX64, bool DEFORM>
__device__ __forceinline__ int phase_a(const KPParams& P, int64_t n0, int h0, int lane,
                                       float4* rel, float* wl, float qx, float qy, float qz,
                                       bool nvalid, float kx, float ky, float kz,
                                       float* run_min /* DEFORM only, PPW==1 */, int* run_arg = nullptr) {
  const int p = lane / HC, h = lane % HC;
  const int64_t n = n0 + p;
  int j = -2;  // -2: no entry, -1: shadow entry
  float rx = 0.f, ry = 0.f, rz = 0.f;
  if (nvalid && h0 + h < P.H) {
    j = load_idx<IDX64>(P.idx, n * P.H + h0 + h, P.Ns);
    if (j >= 0) {
      const float* sp = P.s + (int64_t)j * 3;
      rx = sp[0] - qx;
      ry = sp[1] - qy;
      rz = sp[2] - qz;
    } else {  // shadow support point (1e6,1e6,1e6), blocks.py:277
      rx = 1e6f - qx;
      ry = 1e6f - qy;
      rz = 1e6f - qz;
    }
  }
  rel[lane] = make_float4(rx, ry, rz, __int_as_float(j));
  wave_sync_lds();

  const int k = lane & 15;
  const float ext2 = P.extent * P.extent;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int ph = t * 4 + (lane >> 4);
    const float4 r = rel[ph];
    const int jj = __float_as_int(r.w);
    float dx = r.x - kx, dy = r.y - ky, dz = r.z - kz;
    float d2 = dx * dx + dy * dy + dz * dz;  // blocks.py:294-297
    float w = 0.f;
    if (jj >= 0 && k < P.K) w = influence_w(d2, P.extent, P.influence);
    if (P.aggregation == MVK_AGG_CLOSEST) {
      // one-hot of argmin_k d2 (first minimum), blocks.py:349-351
      float bd = (k < P.K) ? d2 : INFINITY;
      int bk = k;
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        float od = __shfl_xor(bd, m, 16);
        int ok = __shfl_xor(bk, m, 16);
        if (od < bd || (od == bd && ok < bk)) {
          bd = od;
          bk = ok;
        }
      }
      if (bk != k) w = 0.f;
    }
    if (!DEFORM) {
      // a neighbour on which no kernel point has any influence (beyond the extent of all of them: a third of the real
      // neighbours at conv_radius 2.5 / KP_extent 1.2) contributes exactly nothing: its entry becomes a shadow entry, so
      // the scatter spends no atomics (its bound) and the lane = channel gather no row loads on it
      const unsigned long long mz = __ballot(w != 0.f);
      if (((mz >> (lane & 48)) & 0xFFFFull) == 0ull && k == 0 && jj >= 0) rel[ph].w = __int_as_float(-1);
    }
    if (DEFORM) {
      // neighbours out of range of every deformed kernel point are dropped (blocks.py:306-325)
      unsigned long long m = __ballot(jj >= 0 && k < P.K && d2 < ext2);
      bool inrange = ((m >> (lane & 48)) & 0xFFFFull) != 0ull;
      if (!inrange) {
        w = 0.f;
        if (k == 0 && jj >= 0) rel[ph].w = __int_as_float(-1);
      }
      // running min_h d2 over real + shadow entries (blocks.py:303)
      if (jj >= -1 && k < P.K && d2 < *run_min) {     // strict: this lane visits its entries in ascending h
        *run_min = d2;
        if (run_arg) *run_arg = h0 + ph;
      }
    }
    wl[(ph / HC) * (HC * 16 + WPAD) + (ph % HC) * 16 + k] = w;
  }
  wave_sync_lds();
  return j;
}

// ---------------------------------------------------------------------------
// MFMA gather kernel (round 5; rigid layers, linear influence + sum aggregation, f32): the aggregation
//   A[n, k, c] = sum_h w[n, h, k] x[idx[n, h], c]
// of ONE query point is a [K x H] . [H x Cin] product, and with K = 15 it fits the 16 x 16 x 4 f32 matrix instruction
// exactly: v_mfma_f32_16x16x4_f32 takes A[i][kk] from lane (i = lane & 15, kk = lane >> 4) and B[kk][j] from lane
// (kk = lane >> 4, j = lane & 15) -- so lane (m, kq) computes ONE correlation weight, that of kernel point m for the kq-th
// neighbour of the current group of four, and loads ONE feature value per 16-channel tile, channel m of that neighbour's
// row. No LDS, no phase A / phase B hand-off, no per-neighbour loop of 15 x 4 FMAs per lane: per group of four
// neighbours a wave issues 5 cross-lane reads (the neighbour's row and relative position, read once per 64 columns by the
// lane of that column), ~12 VALU instructions for its weight, T dword loads (16 lanes = 64 contiguous bytes of a row) and
// T MFMAs. The vector kernel below is bound by its own instruction stream (15 x 4 v_pk_fma_f32 per neighbour and lane:
// 57 us for 19 464 x 66 against a 28 us floor of pure FMA issue); here the products run on the matrix
// pipe (exact f32 products, f32 accumulation -- four neighbours per step instead of one: another summation order,
// ~1e-7), 56 MFMAs of 32 cycles per point of that layer = 14 us chip-wide.
// One wave per query point (waves of a workgroup are independent), T accumulator tiles of 16 channels (4 VGPRs each);
// rows wider than 16 T channels run as gridDim.y channel blocks. The work list / XCD runs are those of the vector kernel.
// ---------------------------------------------------------------------------
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

// MODE: which channel each lane feeds to tile t -- any assignment works as long as the stores use the same one (column
// j of D depends on column j of B alone), and the texture addresser handles a wave's load at ~16 cycles per instruction
// whatever its width (one CU's addresser serves four matrix pipes), so the loads must be WIDE:
//   1  rows of a multiple of 16 T channels: lane m takes the T consecutive channels c0 + T m .. of its neighbour as
//      8- / 16-byte loads (tile t = its t-th channel): T / 4 load instructions per group instead of T (measured with
//      one dword load per tile: 51 us for 19 464 x 66, 40 us for x 64 -- the addresser, not the matrix pipe, set the pace);
//   2  even rows of 66 .. 80 channels (the early-fusion net's first layer: 64 + 2): channels 4 m .. 4 m + 3 as two
//      8-byte loads (rows are 8-byte aligned) for tiles 0-3, tile 4 = channel 64 + m by a dword load;
//   0  any other row length: tile t = channel c0 + 16 t + m, one dword load per tile.
// KPM: whose kernel points a weight is measured from.
//   0  rigid layer: kernel point m of the layer (P.kp);
//   1  deformable layer, forward (blocks.py:286-327): kernel point m of THIS query point, kp[m] + offsets[n, m] -- one
//      per-lane load per point --, and the layer's second output, min over ALL columns (shadow entries at 1e6 included,
//      blocks.py:303) of the squared distance to each deformed kernel point with the first arg-min column: lane (m, kq)
//      sees columns 4 g + kq in ascending order, the four kq lanes of a kernel point meet at the end. The in-range
//      filter of blocks.py:306-325 needs no code: with the linear influence a neighbour out of range of every kernel
//      point has weight 0 for all of them;
//   2  deformable layer, gather-form feature gradient (the transposed relation: `s` holds the layer's QUERY rows): kernel
//      point m of the NEIGHBOUR row, -(kp[m] + nb_offsets[j, m]), times the modulation nb_mod[j, m] -- two more gathered
//      loads per group, issued with the feature loads.
//
// SW = 4 (launches of few points with long rows: the coarse levels, and every deformable layer -- hundreds of points with
// hundreds of columns at the deform radius): the four waves of a workgroup SHARE one point, wave w takes the group pairs
// w, w + 4, ...; waves 1-3 hand their accumulators (and running minima) to wave 0 through LDS, which adds them in wave
// order (a fixed order) and stores. One wave per point would be a serial chain of H / 4 groups on a fraction of the SIMDs.
template <int T, int MODE, bool IDX64, int KPM = 0, bool SHARE = false>
__global__ __launch_bounds__(256) void kpconv_gather_mfma(KPParams P) {
  static_assert(MODE != 2 || T == 5, "mode 2 is the 64 + tail layout");
  static_assert(MODE != 1 || T == 2 || T % 4 == 0, "mode 1: 8- or 16-byte vectors");
  constexpr int SW = SHARE ? 4 : 1;        // (a template parameter: the one-wave-per-point instantiations keep their registers)
  __shared__ float red_acc[SHARE ? 3 : 1][SHARE ? T * 4 : 1][SHARE ? 64 : 1];
  __shared__ float red_min[SHARE ? 3 : 1][SHARE ? 64 : 1];
  __shared__ int red_arg[SHARE ? 3 : 1][SHARE ? 64 : 1];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  int64_t blk = blockIdx.x;
  if ((int)blockIdx.x < P.xcd_blocks) {      // XCD x works on one contiguous run of the work list (see kpconv_gather_vec)
    const int xq = P.xcd_blocks >> 3, xr = P.xcd_blocks & 7, xc = blockIdx.x & 7;
    blk = (int64_t)xc * xq + min(xc, xr) + (blockIdx.x >> 3);
  }
  const int64_t slot = SW > 1 ? blk : blk * 4 + wid;
  if (slot >= P.Nq) return;                  // (SW 1: the waves of a workgroup never meet; SW 4: the whole workgroup leaves)
  const int gfirst = SW > 1 ? 2 * wid : 0, gstep = 2 * SW;      // this wave's group pairs
  const int n = __builtin_amdgcn_readfirstlane(P.order ? P.order[slot] : (int)slot);
  const int c0 = blockIdx.y * (16 * T);      // first channel of this wave's block
  const float* __restrict__ X = P.x;
  const float qx = P.q[(int64_t)n * 3], qy = P.q[(int64_t)n * 3 + 1], qz = P.q[(int64_t)n * 3 + 2];
  // this lane's kernel point (rows 15.. of the 16-row tile: a point no neighbour is near -> weight 0)
  const int mk = m < P.K ? m : P.K - 1;
  float kx = 1e9f, ky = 1e9f, kz = 1e9f;
  if (m < P.K) {
    kx = P.kp[m * 3];
    ky = P.kp[m * 3 + 1];
    kz = P.kp[m * 3 + 2];
    if (KPM == 1) {
      const float* o = P.offsets + ((int64_t)n * P.K + m) * 3;
      kx += o[0];
      ky += o[1];
      kz += o[2];
    }
  }
  const float inv_ext = 1.0f / P.extent;
  // MODE 0: channel of tile t in this lane, clamped into the row: a lane beyond the row's end loads the last channel again
  // and feeds an output column that is never stored -- no branch, no mask. MODE 1 / 2: first channel of the lane's vector.
  uint32_t chan[MODE == 0 ? T : 1];
  if (MODE == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) chan[t] = (uint32_t)min(c0 + 16 * t + m, P.Cin - 1);
  } else {
    chan[0] = MODE == 1 ? (uint32_t)(c0 + T * m) : (uint32_t)(4 * m);
  }
  const uint32_t tailc = MODE == 2 ? (uint32_t)min(64 + m, P.Cin - 1) : 0u;
  mfma_f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = (mfma_f32x4){0.f, 0.f, 0.f, 0.f};
  float run_min = INFINITY;                  // KPM 1: min over this lane's columns of d2 to kernel point m, first arg-min
  int run_arg = 0x7fffffff;

  for (int h0 = 0; h0 < P.H; h0 += 64) {
    // column h0 + lane of the row: neighbour index (-1: shadow entry, -2: beyond the row), its position relative to the
    // query, the offset of its feature row
    int jl = -2;
    if (h0 + lane < P.H) jl = load_idx<IDX64>(P.idx, (int64_t)n * P.H + h0 + lane, P.Ns);
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (jl >= 0) {
      const float* sp = P.s + (int64_t)jl * 3;
      rx = sp[0] - qx;
      ry = sp[1] - qy;
      rz = sp[2] - qz;
    } else if (KPM == 1) {                  // shadow support point (1e6, 1e6, 1e6), blocks.py:277: it takes part in min_d2
      rx = 1e6f - qx;
      ry = 1e6f - qy;
      rz = 1e6f - qz;
    }
    const unsigned long long live = __ballot(jl >= 0);
    if (KPM != 1 && live == 0ull) continue;
    // KPM 1 walks every column of the row (the shadow entries count for min_d2), the others stop at the last real entry
    const int ncol = min(64, P.H - h0);
    const int groups = KPM == 1 ? (ncol + 3) >> 2 : (64 - __builtin_clzll(live) + 3) >> 2;        // wave-uniform
    const uint32_t roff = (uint32_t)(jl >= 0 ? jl : 0) * (uint32_t)P.Cin;      // (host: Ns * Cin < 2^32)
    // Group g: lane (m, kq) works on column 4 g + kq. Two register sets, the loop unrolled by two: the cross-lane reads
    // and the feature loads of the NEXT group are issued before the MFMAs of the current one, nothing in the body is
    // conditional (a branch around a load makes the compiler wait for everything in flight). A group beyond `groups`
    // (odd counts; columns >= 64 wrap around in the cross-lane read) has weight 0: its columns are shadow entries.
    auto fetch = [&](const int g, float& w, float (&xv)[T]) {
      const int src = 4 * g + kq;
      const int j = __shfl(jl, src);
      float gx = __shfl(rx, src), gy = __shfl(ry, src), gz = __shfl(rz, src);
      const uint32_t off = (uint32_t)__shfl((int)roff, src);
      float wmod = 1.0f;
      if (KPM == 2) {
        // the neighbour row's own kernel point m (clamped indices: the loads are unconditional), used negated
        const int jr = j >= 0 ? j : 0;
        const float* o = P.nb_offsets + ((int64_t)jr * P.K + mk) * 3;
        gx += kx + o[0];
        gy += ky + o[1];
        gz += kz + o[2];
        if (P.nb_mod) wmod = P.nb_mod[(int64_t)jr * P.K + mk];
      } else {
        gx -= kx;
        gy -= ky;
        gz -= kz;
      }
      const float d2 = gx * gx + gy * gy + gz * gz;                                  // blocks.py:294-297
      const float d = __builtin_amdgcn_sqrtf(d2);                                    // :333-335
      w = (j >= 0 && src < 64) ? fmaxf(1.0f - d * inv_ext, 0.0f) * wmod : 0.0f;
      if (KPM == 1 && j >= -1 && src < 64 && m < P.K && d2 < run_min) {             // ascending columns: strict <
        run_min = d2;
        run_arg = h0 + src;
      }
      if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) xv[t] = X[off + chan[t]];
      } else if (MODE == 1 && T == 2) {
        const float2 v = *reinterpret_cast<const float2*>(X + off + chan[0]);
        xv[0] = v.x;
        xv[1] = v.y;
      } else if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < T / 4; ++u) {
          const float4 v = *reinterpret_cast<const float4*>(X + off + chan[0] + 4 * u);
          xv[4 * u] = v.x;
          xv[4 * u + 1] = v.y;
          xv[4 * u + 2] = v.z;
          xv[4 * u + 3] = v.w;
        }
      } else {
        const float2 v0 = *reinterpret_cast<const float2*>(X + off + chan[0]);
        const float2 v1 = *reinterpret_cast<const float2*>(X + off + chan[0] + 2);
        xv[0] = v0.x;
        xv[1] = v0.y;
        xv[2] = v1.x;
        xv[3] = v1.y;
        xv[T - 1] = X[off + tailc];
      }
    };
    float wa, wb;
    float xa[T], xb[T];
    // (KPM 1: a group is fetched exactly once -- fetch also updates the running minimum -- so the look-ahead stops at
    // the last group; the other modes may fetch one group past the end, whose weights are zero)
    if (gfirst >= groups) continue;
    fetch(gfirst, wa, xa);
    for (int g = gfirst; g < groups; g += gstep) {
      if (KPM != 1 || g + 1 < groups) fetch(g + 1, wb, xb);
      else wb = 0.0f;
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xa[t], acc[t], 0, 0, 0);
      if (KPM != 1 || g + gstep < groups) fetch(g + gstep, wa, xa);
      if (KPM != 1 || g + 1 < groups) {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, xb[t], acc[t], 0, 0, 0);
      }
    }
  }
  if (SHARE) {
    // waves 1 .. 3 park their partial results, wave 0 adds them in wave order
    if (wid > 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red_acc[wid - 1][t * 4 + r][lane] = acc[t][r];
      red_min[wid - 1][lane] = run_min;
      red_arg[wid - 1][lane] = run_arg;
    }
    __syncthreads();
    if (wid > 0) return;
    for (int w = 0; w < SW - 1; ++w) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red_acc[w][t * 4 + r][lane];
      const float od = red_min[w][lane];
      const int oa = red_arg[w][lane];
      if (od < run_min || (od == run_min && oa < run_arg)) {
        run_min = od;
        run_arg = oa;
      }
    }
  }
  if (KPM == 1 && P.min_d2 != nullptr && blockIdx.y == 0) {
    // the four column residues of kernel point m: smaller distance wins, equal distances the smaller column (the FIRST
    // arg-min over all columns, like the sequential scan of the vector kernel)
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float od = __shfl_xor(run_min, o);
      const int oa = __shfl_xor(run_arg, o);
      if (od < run_min || (od == run_min && oa < run_arg)) {
        run_min = od;
        run_arg = oa;
      }
    }
    if (kq == 0 && m < P.K) {
      P.min_d2[(int64_t)n * P.K + m] = run_min;
      if (P.min_arg) P.min_arg[(int64_t)n * P.K + m] = run_arg;
    }
  }
  // D[i][j]: lane (j = m, i = 4 kq + r) holds kernel point 4 kq + r in register r of tile t, i.e. of the channel the lane
  // fed to tile t
  float* __restrict__ out = P.A + (int64_t)n * P.K * P.Cin;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kpt = 4 * kq + r;
    if (kpt >= P.K) continue;
    float* __restrict__ o = out + (int64_t)kpt * P.Cin;
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (c0 + 16 * t + m < P.Cin) o[c0 + 16 * t + m] = acc[t][r];
    } else if (MODE == 1 && T == 2) {
      *reinterpret_cast<float2*>(o + c0 + 2 * m) = make_float2(acc[0][r], acc[1][r]);
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < T / 4; ++u)
        *reinterpret_cast<float4*>(o + c0 + T * m + 4 * u) =
            make_float4(acc[4 * u][r], acc[4 * u + 1][r], acc[4 * u + 2][r], acc[4 * u + 3][r]);
    } else {
      *reinterpret_cast<float2*>(o + 4 * m) = make_float2(acc[0][r], acc[1][r]);
      *reinterpret_cast<float2*>(o + 4 * m + 2) = make_float2(acc[2][r], acc[3][r]);
      if (64 + m < P.Cin) o[64 + m] = acc[T - 1][r];
    }
  }
}

// ---------------------------------------------------------------------------
// Vector gather kernel (rigid KPConv, any Cin <= 4*64*NCH).
//   LPP lanes own one query point (4 channels per lane and channel chunk), PPW = 64/LPP points per
//   wave, neighbours in chunks of HC = 64/PPW per point. LPP/PPW/HC are launch parameters so that
//   e.g. Cin = 66 runs with LPP = 17, PPW = 3 (80 % of the lanes busy) instead of a padded power of 2.
//   Phase A: lane (p,h) gathers its neighbour's xyz and evaluates all K kernel-point correlations
//            itself (kernel points are wave-uniform -> scalar registers), then writes one 64-byte LDS
//            row {w[0..14], neighbour row index}.
//   Phase B: per neighbour, the point group reads that row (4 x ds_read_b128, broadcast inside the
//            group, rows of different groups are skewed by 16 B to spread banks), loads the
//            16-byte slice of the feature row and does K x 4 FMAs (v_pk_fma_f32).
//   FAST = linear influence + sum aggregation (the network default): sqrt via v_sqrt_f32 and a
//   multiplication by 1/extent (1 ulp class differences, far inside the 1e-4 parity bar).
// ---------------------------------------------------------------------------
// GWPB independent waves per workgroup (nothing is shared between them): one-wave workgroups are
// dispatched too slowly to keep the SIMDs' wave slots filled (measured 2.3 resident waves per SIMD of 4)
constexpr int GWPB = 4;


// 4 consecutive channels of a feature row / of the aggregate
typedef float f4v __attribute__((ext_vector_type(4)));        // accumulator quad (a native vector: usable as an asm operand)
__device__ __forceinline__ float4 as_float4(f4v v) { return make_float4(v.x, v.y, v.z, v.w); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// DEFORM (deformable layers, blocks.py:286-327; LPP >= 4 so that PPW <= 16): every point's deformed kernel points
// (kernel point + its offset) live in LDS, phase A reads them instead of the wave-uniform rigid ones, drops the
// neighbours that no deformed kernel point reaches (their row index becomes the shadow -> phase B skips them) and
// writes a second LDS row with the 15 squared distances, from which the lanes (point, kernel point, column
// residue) keep the running (min, first arg-min column) over ALL entries, shadow included (blocks.py:303).
// The deformable levels are the coarse ones (hundreds of points, hundreds of neighbour columns), so the GWPB
// waves of a workgroup share the SAME points and take every GWPB-th neighbour chunk; wave 0 adds the partial
// aggregates (LDS, fixed order -> deterministic) and writes A / min_d2 / min_arg.
constexpr int DPPW = 16;      // most points per wave of the deformable variant

//
// FUB > 0 (rows of >= 4 elements, feature table < 4 GB, FUB = rows per batch): phase B carries no branches at
// all, so that a whole batch of feature rows is in flight behind counted waits: the neighbour rows of a batch
// are read from LDS together, shadow entries load row 0 and multiply it by their zero weights, the weight rows
// of a point are padded to a multiple of FUB with zero rows, addresses are 32-bit offsets from the table base,
// and the lane holding the ragged last quad of a row works on the row's LAST four channels instead (it
// recomputes up to three channels of its left neighbour, bit for bit, and stores them again).
//
// TAIL > 0 (with FUB; rows of 4 LPP + TAIL channels, TAIL = 1..3, LPP >= 15: Cin = 66 of the early-fusion net's
// first layer): the LPP lanes of a point take the 4 LPP leading channels as whole quads; the TAIL trailing channels
// x 15 kernel points are spread over the same lanes -- lane l < 15 keeps kernel point l's TAIL sums and adds, per
// neighbour, ITS weight (one ds_read_b32 of the row) times the row's TAIL trailing values (the same address for the
// whole group). A 17th lane per point for two channels would cost a quarter of the launch (3 points per wave, not 4).
template <int NCH, bool IDX64, bool FAST, bool DEFORM = false, int FUB = 0, int TAIL = 0>
__global__ __launch_bounds__(64 * GWPB, NCH == 1 ? 4 : 2) void kpconv_gather_vec(KPParams P, int LPP, int PPW, int HC,
                                                                                  int SW, int B1) {
  constexpr bool FASTLD = FUB > 0;
  static_assert(TAIL == 0 || (FASTLD && NCH == 1 && !DEFORM), "trailing channels: branch-free rigid variant only");
  constexpr int TL = TAIL > 0 ? TAIL : 1;
  // Sharing workgroups (all of them when DEFORM; with SW > 1 those from block B1 on): the waves of the workgroup
  // share the same PPW points and take every nwv-th neighbour chunk; wave 0 adds the partial aggregates through
  // LDS in a fixed order and stores them. A wave lives for tens of microseconds, so a launch of 1.2 rounds of
  // independent waves takes two rounds; its last 0.2 round runs as sharing workgroups instead, four times as many
  // waves of a quarter of the length, which fill the chip.
  const int nwv = blockDim.x >> 6;
  const bool share = DEFORM || (FASTLD && SW > 1 && (int)blockIdx.x >= B1);
  const float* __restrict__ X = P.x;      // features [Ns,Cin]
  float* __restrict__ Aout = P.A;         // aggregate [Nq,K,Cin]
  constexpr int UB = FASTLD ? FUB : (NCH == 1 ? 6 : 4);  // feature rows in flight per lane
  __shared__ __align__(16) float wl_all[GWPB][64 * 16 + 64 * 4];
  __shared__ float d2_all[DEFORM ? GWPB : 1][DEFORM ? 64 * 16 + DPPW * 16 : 1];  // squared distances of the chunk (rows skewed by point)
  __shared__ float4 kd[DEFORM ? DPPW * 16 : 1];                                 // deformed kernel points per point
  const int wid = threadIdx.x >> 6;
  float* wl = wl_all[wid];
  float* d2l = d2_all[DEFORM ? wid : 0];
  const int lane = threadIdx.x & 63;
  const int32_t* __restrict__ ord = DEFORM ? nullptr : P.order;
  int64_t blk = blockIdx.x;
  if (!DEFORM && (int)blockIdx.x < P.xcd_blocks) {
    // workgroups go to the XCDs round robin (b % 8): XCD x then works on ONE contiguous run of the sorted work list,
    // so that the rows its L2 pulls are those of one region of the cloud
    const int xq = P.xcd_blocks >> 3, xr = P.xcd_blocks & 7, xc = blockIdx.x & 7;
    blk = (int64_t)xc * xq + min(xc, xr) + (blockIdx.x >> 3);
  }
  const int64_t n0 = share ? ((int64_t)B1 * GWPB + ((int64_t)blockIdx.x - B1)) * PPW : (blk * GWPB + wid) * PPW;
  const int hbeg = share ? wid * HC : 0, hstep = share ? nwv * HC : HC;        // this wave's neighbour chunks
  // phase-A identity: (point pa, neighbour slot ha)
  const int pa = lane / HC, ha = lane - pa * HC;
  const bool a_on = pa < PPW && n0 + pa < P.Nq;
  const int64_t na = (a_on && ord) ? (int64_t)ord[n0 + pa] : n0 + pa;      // the point of this lane's phase-A identity
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (a_on) {
    const float* qp = P.q + na * 3;
    qx = qp[0];
    qy = qp[1];
    qz = qp[2];
  }
  const int HCP = FASTLD ? (HC + UB - 1) / UB * UB : HC;        // weight rows per point in LDS
  float* wrow_a = wl + (pa * HCP + ha) * 16 + pa * 4;
  // phase-B identity: (point pb, channel quad cl)
  const int pb = lane / LPP, cl = lane - pb * LPP;
  const int64_t nslot = n0 + pb;
  const bool b_on = pb < PPW && nslot < P.Nq;
  const float* wblk_b = wl + (FASTLD ? min(pb, PPW - 1) : pb) * (HCP * 16 + 4);
  if (FASTLD && HCP > HC) {      // padding rows: zero weights, shadow index
    const int npad = HCP - HC;
    if (lane < PPW * npad) {
      const int pp = lane / npad, r = HC + lane % npad;
      float4* dst = reinterpret_cast<float4*>(wl + (pp * HCP + r) * 16 + pp * 4);
      dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
      dst[3] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    }
  }
  uint32_t c4e[NCH];      // FASTLD: first channel of the lane's quad (the ragged quad moved left to end at Cin)
  bool c_on[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int c4 = (cl + c * LPP) * 4;
    c_on[c] = c4 < P.Cin;
    c4e[c] = c_on[c] ? (uint32_t)min(c4, P.Cin - 4) : 0u;
  }
  const uint32_t row_bytes = (uint32_t)P.Cin * 4u;
  uint64_t batch_mask0 = 0;     // phase-A lanes of the columns 0 .. UB-1 of every point (shifted by the batch's first column)
  for (int pp = 0; pp < PPW; ++pp) batch_mask0 |= ((1ull << UB) - 1ull) << (pp * HC);
  const uint32_t tail_bytes = (uint32_t)(P.Cin - TAIL) * 4u;   // TAIL: offset of the trailing channels
  const bool t_on = cl < KMAX - 1;                                               // TAIL: this lane keeps kernel point cl
  float acc_t[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) acc_t[t] = 0.f;
  const float inv_ext = 1.0f / P.extent;
  const float* __restrict__ kp = P.kp;
  const float ext2 = P.extent * P.extent;
  // DEFORM scan identity: pair (point, k) = pk, of which there are NPK = 16 PPW. NPK >= 64: pairs lane + 64 t, every
  // column. NPK < 64: pair lane % NPK, columns congruent to lane / NPK modulo NSUB = 64 / NPK.
  const int NPK = PPW * 16;
  const int NSUB = NPK >= 64 ? 1 : 64 / NPK;
  const int sub = NPK >= 64 ? 0 : lane / NPK;
  float run_min[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
  int run_arg[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  if (DEFORM) {
    for (int e = threadIdx.x; e < DPPW * 16; e += blockDim.x) {
      const int pp = e >> 4, kk = e & 15;
      float4 v = make_float4(1e9f, 1e9f, 1e9f, 0.f);        // slots beyond K / beyond the last point: out of every range
      if (kk < P.K && n0 + pp < P.Nq) {
        const float* o = P.offsets + ((n0 + pp) * P.K + kk) * 3;
        v = make_float4(kp[kk * 3] + o[0], kp[kk * 3 + 1] + o[1], kp[kk * 3 + 2] + o[2], 0.f);   // blocks.py:287
      }
      kd[e] = v;
    }
    __syncthreads();
  }

  f4v acc[NCH][KMAX - 1];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int kk = 0; kk < KMAX - 1; ++kk) acc[c][kk] = f4v{0.f, 0.f, 0.f, 0.f};

  // software pipeline over neighbour chunks: index two chunks ahead, support xyz one chunk ahead
  auto ld_j = [&](int h) -> int {
    return (a_on && h < P.H) ? load_idx<IDX64>(P.idx, na * P.H + h, P.Ns) : -2;
  };
  int jA = ld_j(hbeg + ha), jB = ld_j(hbeg + hstep + ha);
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (jA >= 0) {
    const float* sp = P.s + (int64_t)jA * 3;
    sx = sp[0];
    sy = sp[1];
    sz = sp[2];
  }
  for (int h0 = hbeg; h0 < P.H; h0 += hstep) {
    const int jC = ld_j(h0 + 2 * hstep + ha);
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (jB >= 0) {
      const float* sp = P.s + (int64_t)jB * 3;
      tx = sp[0];
      ty = sp[1];
      tz = sp[2];
    }
    // ---------------- phase A: this lane's neighbour against every kernel point
    float wv[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) wv[kk] = 0.f;
    int jrow = jA;                 // row index phase B sees (DEFORM: shadow for neighbours out of every range)
    if (DEFORM) {
      // entry present (real or shadow, blocks.py:277): distances to the point's deformed kernel points
      float dv[16];
      bool inrange = false;
      float bd = INFINITY;
      int bk = 0;
      const float rx = (jA >= 0 ? sx : 1e6f) - qx, ry = (jA >= 0 ? sy : 1e6f) - qy, rz = (jA >= 0 ? sz : 1e6f) - qz;
#pragma unroll
      for (int kk = 0; kk < KMAX - 1; ++kk) {
        const float4 kq = kd[(pa < PPW ? pa : 0) * 16 + kk];
        const float dx = rx - kq.x, dy = ry - kq.y, dz = rz - kq.z;
        const float d2 = dx * dx + dy * dy + dz * dz;  // blocks.py:294-297
        dv[kk] = jA >= -1 ? d2 : INFINITY;
        if (jA >= 0) {
          inrange = inrange || d2 < ext2;
          wv[kk] = FAST ? fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_ext, 0.0f) : influence_w(d2, P.extent, P.influence);
          if (!FAST && kk < P.K && d2 < bd) {
            bd = d2;
            bk = kk;
          }
        }
      }
      dv[15] = INFINITY;
      if (!FAST && P.aggregation == MVK_AGG_CLOSEST) {  // one-hot of the first arg-min (blocks.py:349-351)
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk)
          if (kk != bk) wv[kk] = 0.f;
      }
      if (!inrange) {                      // dropped by the in-range filter (blocks.py:306-325)
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk) wv[kk] = 0.f;
        if (jA >= 0) jrow = -1;
      }
      float4* dd = reinterpret_cast<float4*>(d2l + lane * 16 + pa * 16);
      dd[0] = make_float4(dv[0], dv[1], dv[2], dv[3]);
      dd[1] = make_float4(dv[4], dv[5], dv[6], dv[7]);
      dd[2] = make_float4(dv[8], dv[9], dv[10], dv[11]);
      dd[3] = make_float4(dv[12], dv[13], dv[14], dv[15]);
    } else if (jA >= 0) {
      const float rx = sx - qx, ry = sy - qy, rz = sz - qz;
      float bd = INFINITY;
      int bk = 0;
#pragma unroll
      for (int kk = 0; kk < KMAX - 1; ++kk) {
        if (kk < P.K) {
          const float dx = rx - kp[kk * 3], dy = ry - kp[kk * 3 + 1], dz = rz - kp[kk * 3 + 2];
          const float d2 = dx * dx + dy * dy + dz * dz;  // blocks.py:294-297
          if (FAST) {
            wv[kk] = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_ext, 0.0f);  // blocks.py:335-338
          } else {
            wv[kk] = influence_w(d2, P.extent, P.influence);
            if (d2 < bd) {
              bd = d2;
              bk = kk;
            }
          }
        }
      }
      if (!FAST && P.aggregation == MVK_AGG_CLOSEST) {  // one-hot of the first arg-min (blocks.py:349-351)
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk)
          if (kk != bk) wv[kk] = 0.f;
      }
      if (FASTLD && P.zero_skip) {     // no kernel point has any influence on this neighbour: a shadow entry (no row of its own is loaded,
                        // and batches of such entries are skipped as a whole)
        float wany = 0.f;
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk) wany = fmaxf(wany, fabsf(wv[kk]));
        if (wany == 0.f) jrow = -1;
      }
    }
    wv[15] = __int_as_float(jrow);
    {
      float4* dst = reinterpret_cast<float4*>(wrow_a);
      dst[0] = make_float4(wv[0], wv[1], wv[2], wv[3]);
      dst[1] = make_float4(wv[4], wv[5], wv[6], wv[7]);
      dst[2] = make_float4(wv[8], wv[9], wv[10], wv[11]);
      dst[3] = make_float4(wv[12], wv[13], wv[14], wv[15]);
    }
    wave_sync_lds();
    if (DEFORM) {      // running (min, first column) of d2 per (point, kernel point)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = lane + 64 * t;
        const int pk = NPK >= 64 ? e : lane - sub * NPK;
        if (pk < NPK && sub < NSUB && e < (NPK >= 64 ? NPK : 64)) {
          const int pp = pk >> 4, kk = pk & 15;
          for (int h = sub; h < HC; h += NSUB) {
            const float d = d2l[(pp * HC + h) * 16 + pp * 16 + kk];
            if (d < run_min[t]) {           // strict: this lane visits its columns in ascending order
              run_min[t] = d;
              run_arg[t] = h0 + h;
            }
          }
        }
      }
    }
    // ---------------- phase B: UB feature rows in flight, then their FMAs
    const uint64_t live = __ballot(jrow >= 0);        // bit = phase-A lane (point pa, column ha) holds a real neighbour
    if (live != 0ull) {
      for (int hb = 0; hb < HCP; hb += UB) {
        // FASTLD: a batch whose columns are shadow entries for every point of the wave (the tail of the sorted
        // neighbour rows: H is the 90th-percentile width, the mean row holds 3/4 of it) is skipped as a whole
        // (only in the instantiations where the extra branch leaves the register allocation alone: with 5-row
        // batches or three trailing channels it costs 30 spilled registers and up to a third of the speed)
        constexpr bool SKIP = FASTLD && NCH == 1 &&
                              (FUB == 6 || FUB == 7 || FUB == 8 || (FUB == 4 && (TAIL == 1 || TAIL == 2)));
        if (SKIP && (live & (batch_mask0 << hb)) == 0ull) continue;
        int jj[UB];
        float4 xv[UB][NCH];
        float xt[UB][TL];
        if (FASTLD) {
#pragma unroll
          for (int u = 0; u < UB; ++u) jj[u] = __float_as_int(wblk_b[(hb + u) * 16 + 15]);
#pragma unroll
          for (int u = 0; u < UB; ++u) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {   // byte offset < 2^32, row index and row bytes < 2^24 (checked by the host)
              const uint32_t off = __umul24((uint32_t)max(jj[u], 0), row_bytes) + c4e[c] * 4u;
              xv[u][c] = ld4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(X) + off));
            }
            if (TAIL > 0) {
              const float* pt = reinterpret_cast<const float*>(reinterpret_cast<const char*>(X) +
                                                         (__umul24((uint32_t)max(jj[u], 0), row_bytes) + tail_bytes));
#pragma unroll
              for (int t = 0; t < TL; ++t) xt[u][t] = (float)pt[t];
            }
            asm volatile("" ::: "memory");    // loads leave in entry order, so that entry u waits for u + 1 loads only
          }
        } else
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          jj[u] = (hb + u < HC && b_on) ? __float_as_int(wblk_b[(hb + u) * 16 + 15]) : -2;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int c4 = (cl + c * LPP) * 4;
            xv[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jj[u] >= 0 && c4 < P.Cin) {
              const float* xr = X + (int64_t)jj[u] * P.Cin + c4;
              if (c4 + 3 < P.Cin) {
                xv[u][c] = ld4(xr);
              } else {  // ragged tail of a row whose length is not a multiple of 4
                xv[u][c].x = (float)xr[0];
                if (c4 + 1 < P.Cin) xv[u][c].y = (float)xr[1];
                if (c4 + 2 < P.Cin) xv[u][c].z = (float)xr[2];
                if (c4 + 3 < P.Cin) xv[u][c].w = (float)xr[3];
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if (FASTLD || jj[u] >= 0) {
            const float4* w4 = reinterpret_cast<const float4*>(wblk_b + (hb + u) * 16);
            const float4 wa = w4[0], wb = w4[1], wc = w4[2], wd = w4[3];
            const float wk[15] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w,
                                  wc.x, wc.y, wc.z, wc.w, wd.x, wd.y, wd.z};
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
              for (int kk = 0; kk < KMAX - 1; ++kk) {
                acc[c][kk].x += wk[kk] * xv[u][c].x;
                acc[c][kk].y += wk[kk] * xv[u][c].y;
                acc[c][kk].z += wk[kk] * xv[u][c].z;
                acc[c][kk].w += wk[kk] * xv[u][c].w;
              }
            }
            if (TAIL > 0) {
              const float wraw = wblk_b[(hb + u) * 16 + (t_on ? cl : 0)];
              const float wt = t_on ? wraw : 0.f;
#pragma unroll
              for (int t = 0; t < TL; ++t) acc_t[t] += wt * xt[u][t];
            }
            if (FASTLD) {   // entry u's products stay together: otherwise every entry's weight row is live at once
#pragma unroll
              for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int kk = 0; kk < KMAX - 1; ++kk) asm volatile("" : "+v"(acc[c][kk]));
              if (TAIL > 0) {
#pragma unroll
                for (int t = 0; t < TL; ++t) asm volatile("" : "+v"(acc_t[t]));
              }
            }
          }
        }
      }
    }
    wave_sync_lds();
    jA = jB;
    jB = jC;
    sx = tx;
    sy = ty;
    sz = tz;
  }
  if (share) {
    // partial aggregates of waves 1..nwv-1 -> wave 0, five kernel points per round through the waves' weight rows
    __syncthreads();
    float4* mine = reinterpret_cast<float4*>(wl_all[wid]);        // [5][64] float4 = the wave's 1280 floats
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (wid > 0) {
#pragma unroll
          for (int q5 = 0; q5 < 5; ++q5) mine[q5 * 64 + lane] = as_float4(acc[c][r * 5 + q5]);
        }
        __syncthreads();
        if (wid == 0) {
          for (int w = 1; w < nwv; ++w) {
            const float4* theirs = reinterpret_cast<const float4*>(wl_all[w]);
#pragma unroll
            for (int q5 = 0; q5 < 5; ++q5) {
              const float4 v = theirs[q5 * 64 + lane];
              acc[c][r * 5 + q5].x += v.x;
              acc[c][r * 5 + q5].y += v.y;
              acc[c][r * 5 + q5].z += v.z;
              acc[c][r * 5 + q5].w += v.w;
            }
          }
        }
        __syncthreads();
      }
    }
    if (TAIL > 0) {       // the trailing channels' sums, one more round
      if (wid > 0) mine[lane] = make_float4(acc_t[0], TL > 1 ? acc_t[TL > 1 ? 1 : 0] : 0.f, TL > 2 ? acc_t[TL > 2 ? 2 : 0] : 0.f, 0.f);
      __syncthreads();
      if (wid == 0) {
        for (int w = 1; w < nwv; ++w) {
          const float4 v = reinterpret_cast<const float4*>(wl_all[w])[lane];
          acc_t[0] += v.x;
          if (TL > 1) acc_t[TL > 1 ? 1 : 0] += v.y;
          if (TL > 2) acc_t[TL > 2 ? 2 : 0] += v.z;
        }
      }
      __syncthreads();
    }
  }
  if (DEFORM) {
    // (min, column) candidates of every wave and column residue
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wl[t * 64 + lane] = run_min[t];
      wl[256 + t * 64 + lane] = __int_as_float(run_arg[t]);
    }
    __syncthreads();
    if (wid == 0 && P.min_d2 != nullptr) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pk = lane + 64 * t;
        if (pk < NPK && (NPK >= 64 || t == 0)) {
          float m = INFINITY;
          int a = 0x7fffffff;
          for (int w = 0; w < nwv; ++w)
            for (int sb = 0; sb < NSUB; ++sb) {
              const int slot = NPK >= 64 ? pk : sb * NPK + pk;
              const float om = wl_all[w][slot];
              const int oa = __float_as_int(wl_all[w][256 + slot]);
              if (om < m || (om == m && oa < a)) {
                m = om;
                a = oa;
              }
            }
          const int pp = pk >> 4, kk = pk & 15;
          if (kk < P.K && n0 + pp < P.Nq) {
            P.min_d2[(n0 + pp) * P.K + kk] = m;
            if (P.min_arg) P.min_arg[(n0 + pp) * P.K + kk] = a == 0x7fffffff ? 0 : a;
          }
        }
      }
    }
  }
  if (b_on && (!share || wid == 0)) {
    const int64_t n = ord ? (int64_t)ord[nslot] : nslot;      // the row of this lane's phase-B identity
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int c4 = FASTLD ? (int)c4e[c] : (cl + c * LPP) * 4;
      if (FASTLD ? c_on[c] : c4 < P.Cin) {
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk) {
          if (kk < P.K) {
            // (the f32 instantiations keep the dense [Nq,K,Cin] addressing: they sit at their register limit)
            float* o = Aout + (n * P.K + kk) * P.Cin + c4;
            if (FASTLD || c4 + 3 < P.Cin) {
              st4(o, as_float4(acc[c][kk]));
            } else {
              o[0] = acc[c][kk].x;
              if (c4 + 1 < P.Cin) o[1] = acc[c][kk].y;
              if (c4 + 2 < P.Cin) o[2] = acc[c][kk].z;
              if (c4 + 3 < P.Cin) o[3] = acc[c][kk].w;
            }
          }
        }
      }
    }
    if (TAIL > 0 && cl < P.K && t_on) {      // lane cl holds kernel point cl's sums over the trailing channels
      float* o = Aout + (n * P.K + cl) * P.Cin + (P.Cin - TAIL);
#pragma unroll
      for (int t = 0; t < TL; ++t) o[t] = acc_t[t];
    }
  }
}

// ---------------------------------------------------------------------------
// Generic lane = channel kernel (one point per wave). MODE 0: forward gather
// (any Cin, rigid or deformable). MODE 1: backward scatter (+ deformable grads).
// Channels handled: c = c0 + lane + 64*slot, slot < NSLOT.
// ---------------------------------------------------------------------------
// WPB > 1 (backward scatter of the layers searched at the deform radius: hundreds of neighbour columns for a few
// hundred points): the WPB waves of a workgroup work on the SAME point, wave w on the 64-neighbour chunks w, w + WPB,
// ...; they meet only in the atomics on dx (one wave per point was a 26 us serial chain on a third of the SIMDs).
// HS > 1 (scatter of the coarse levels, a few hundred points or fewer: one wave per point leaves most SIMDs idle behind
// a serial chain of H neighbours x NSLOT atomics): the workgroup's HS waves all evaluate the chunk's weights and take
// every HS-th neighbour each, and the 64-channel slots of a point are spread over blockIdx.y (NSLOT = 1) -- level 4
// (85 points, 512 channels): 85 waves -> 2 720.
template <int NSLOT, bool IDX64, int MODE, bool DEFORM, int WPB = 1, int HS = 1>
__global__ __launch_bounds__(64 * WPB * HS) void kpconv_lane_channel(KPParams P, int c0_base) {
  constexpr int HC = 64, WPAD = 0;
  static_assert(WPB == 1 || MODE == 1, "only the scatter splits a point over waves");
  static_assert(HS == 1 || (MODE == 1 && WPB == 1 && !DEFORM), "neighbour interleave: rigid scatter only");
  const int c0 = c0_base + (int)blockIdx.y * 64 * NSLOT;
  __shared__ float4 rel_all[WPB * HS][64];
  __shared__ float wl_all[WPB * HS][64 * 16];
  const int lane = threadIdx.x & 63, wid_all = threadIdx.x >> 6;
  const int wid = HS > 1 ? 0 : wid_all;          // HS: every wave walks all chunks
  float4* rel = rel_all[wid_all];
  float* wl = wl_all[wid_all];
  const int64_t n = blockIdx.x;
  const float* qp = P.q + n * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const int k = lane & 15;
  float kx = 0.f, ky = 0.f, kz = 0.f;
  if (k < P.K) {
    kx = P.kp[k * 3];
    ky = P.kp[k * 3 + 1];
    kz = P.kp[k * 3 + 2];
    if (DEFORM) {  // deformed kernel point (blocks.py:287)
      const float* o = P.offsets + (n * P.K + k) * 3;
      kx += o[0];
      ky += o[1];
      kz += o[2];
    }
  }
  float run_min = INFINITY;
  int run_arg = 0;

  float acc[NSLOT][KMAX - 1];  // MODE 0: A accumulators; MODE 1: dA values
#pragma unroll
  for (int sidx = 0; sidx < NSLOT; ++sidx) {
    const int c = c0 + lane + 64 * sidx;
#pragma unroll
    for (int kk = 0; kk < KMAX - 1; ++kk) {
      if (MODE == 0)
        acc[sidx][kk] = 0.f;
      else
        acc[sidx][kk] = (c < P.Cin && kk < P.K) ? P.A[(n * P.K + kk) * P.Cin + c] : 0.f;
    }
  }

  for (int h0 = wid * HC; h0 < P.H; h0 += HC * WPB) {
    int j = phase_a<1, HC, WPAD, IDX64, DEFORM>(P, n, h0, lane, rel, wl, qx, qy, qz, true, kx, ky,
                                                 kz, &run_min, &run_arg);
    if (__ballot(j >= 0) != 0ull) {
      const int hend = min(HC, P.H - h0);
      for (int hh = HS > 1 ? wid_all : 0; hh < hend; hh += HS) {
        const int jj = __builtin_amdgcn_readfirstlane(__float_as_int(rel[hh].w));
        if (jj < 0) continue;  // wave-uniform
        const float4* w4 = reinterpret_cast<const float4*>(wl + hh * 16);
        const float4 wa = w4[0], wb = w4[1], wc = w4[2], wd = w4[3];
        const float wv[16] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w,
                              wc.x, wc.y, wc.z, wc.w, wd.x, wd.y, wd.z, wd.w};
#pragma unroll
        for (int sidx = 0; sidx < NSLOT; ++sidx) {
          const int c = c0 + lane + 64 * sidx;
          if (c < P.Cin) {
            if (MODE == 0) {
              const float xv = P.x[(int64_t)jj * P.Cin + c];
#pragma unroll
              for (int kk = 0; kk < KMAX - 1; ++kk) acc[sidx][kk] += wv[kk] * xv;
            } else {
              float contrib = 0.f;
#pragma unroll
              for (int kk = 0; kk < KMAX - 1; ++kk) contrib += wv[kk] * acc[sidx][kk];
              atomicAdd(P.dx + (int64_t)jj * P.Cin + c, contrib);
            }
          }
        }
      }
    }
    wave_sync_lds();
  }

  if (MODE == 0) {
#pragma unroll
    for (int sidx = 0; sidx < NSLOT; ++sidx) {
      const int c = c0 + lane + 64 * sidx;
      if (c < P.Cin) {
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk)
          if (kk < P.K) P.A[(n * P.K + kk) * P.Cin + c] = acc[sidx][kk];
      }
    }
    if (DEFORM && P.min_d2 != nullptr && c0 == 0) {
      float m = run_min;      // lanes l, l^16, l^32, l^48 hold kernel point k = l & 15: (value, column) minimum, first column on ties
      int a = run_arg;
#pragma unroll
      for (int sh = 16; sh <= 32; sh <<= 1) {
        const float om = __shfl_xor(m, sh);
        const int oa = __shfl_xor(a, sh);
        if (om < m || (om == m && oa < a)) {
          m = om;
          a = oa;
        }
      }
      if (lane < 16 && lane < P.K) {
        P.min_d2[n * P.K + lane] = m;
        if (P.min_arg) P.min_arg[n * P.K + lane] = a;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Rigid gather for rows of <= 4 channels (the first layer of the baseline / middle- / late-fusion nets: features
// (1, z) or (1, r, g, b)). The vector kernel would run with one lane per point AND one neighbour per chunk --
// an LDS round trip and a barrier per neighbour (119 us for 19 464 points). Here SLP = 4 lanes own a point: lane
// `sub` takes the neighbours sub, sub + 4, ... in batches of four (indices, then xyz + feature rows, then the
// products), all accumulators (15 x <= 4) in registers, no LDS; the four partial aggregates are added by two
// butterfly steps at the end (one lane per point alone leaves 304 waves on 1 024 SIMDs, each a 50 us chain).
// ---------------------------------------------------------------------------
template <bool IDX64, bool FAST>
__global__ __launch_bounds__(256) void kpconv_gather_small(KPParams P) {
  constexpr int SLP = 4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(gid & (SLP - 1));
  const bool live = gid / SLP < P.Nq;                      // (whole waves take part in the shuffles below)
  const int64_t n = live ? gid / SLP : P.Nq - 1;
  const float qx = P.q[n * 3], qy = P.q[n * 3 + 1], qz = P.q[n * 3 + 2];
  const float inv_ext = 1.0f / P.extent;
  const float* __restrict__ kp = P.kp;
  const int Cin = P.Cin;
  float acc[KMAX - 1][4];
#pragma unroll
  for (int kk = 0; kk < KMAX - 1; ++kk)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[kk][c] = 0.f;
  constexpr int UBS = 4;
  for (int h0 = sub; h0 < P.H; h0 += UBS * SLP) {
    int j[UBS];
#pragma unroll
    for (int u = 0; u < UBS; ++u)
      j[u] = h0 + u * SLP < P.H ? load_idx<IDX64>(P.idx, n * P.H + h0 + u * SLP, P.Ns) : -1;
    float sx[UBS], sy[UBS], sz[UBS], xv[UBS][4];
#pragma unroll
    for (int u = 0; u < UBS; ++u) {
      const int64_t jj = j[u] >= 0 ? j[u] : 0;                  // shadow entries read row 0 and are not applied
      sx[u] = P.s[jj * 3];
      sy[u] = P.s[jj * 3 + 1];
      sz[u] = P.s[jj * 3 + 2];
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[u][c] = c < Cin ? P.x[jj * Cin + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UBS; ++u) {
      if (j[u] < 0) continue;
      const float rx = sx[u] - qx, ry = sy[u] - qy, rz = sz[u] - qz;
      float wv[KMAX - 1];
      float bd = INFINITY;
      int bk = 0;
#pragma unroll
      for (int kk = 0; kk < KMAX - 1; ++kk) {
        wv[kk] = 0.f;
        if (kk < P.K) {
          const float dx = rx - kp[kk * 3], dy = ry - kp[kk * 3 + 1], dz = rz - kp[kk * 3 + 2];
          const float d2 = dx * dx + dy * dy + dz * dz;  // blocks.py:294-297
          if (FAST) {
            wv[kk] = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_ext, 0.0f);  // blocks.py:335-338
          } else {
            wv[kk] = influence_w(d2, P.extent, P.influence);
            if (d2 < bd) {
              bd = d2;
              bk = kk;
            }
          }
        }
      }
      if (!FAST && P.aggregation == MVK_AGG_CLOSEST) {  // one-hot of the first arg-min (blocks.py:349-351)
#pragma unroll
        for (int kk = 0; kk < KMAX - 1; ++kk)
          if (kk != bk) wv[kk] = 0.f;
      }
#pragma unroll
      for (int kk = 0; kk < KMAX - 1; ++kk)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[kk][c] += wv[kk] * xv[u][c];
    }
  }
#pragma unroll
  for (int kk = 0; kk < KMAX - 1; ++kk)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[kk][c] += __shfl_xor(acc[kk][c], 1);
      acc[kk][c] += __shfl_xor(acc[kk][c], 2);
    }
  if (!live) return;
#pragma unroll
  for (int kk = 0; kk < KMAX - 1; ++kk) {
    if (kk < P.K && (kk & (SLP - 1)) == sub) {             // the four lanes share the stores
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < Cin) P.A[(n * P.K + kk) * Cin + c] = acc[kk][c];
    }
  }
}

// Launch geometry of the vector gather kernel for one layer (also exported: mvk_kpconv_gather_plan).
struct VecPlan {
  int LPP, PPW, HC;   // lanes per point, points per wave, neighbours per chunk and point
  int fub;            // rows per batch of the branch-free variant, 0 = the general variant
  int tail;           // trailing channels handled beside the quads (TAIL of the kernel), 0 = none
  int SW, B1, nw;     // sharing: waves per sharing workgroup (1 = none), first sharing workgroup, waves per workgroup
  int64_t wgs;        // workgroups
};

VecPlan plan_vec(int64_t Nq, int64_t Ns, int H, int Cin, bool fast, bool deform) {
  VecPlan v{};
  const int NCH = Cin <= 256 ? 1 : 2;
  static const bool fastld_on = getenv("MVK_GATHER_FASTLD") == nullptr || atoi(getenv("MVK_GATHER_FASTLD")) != 0;
  static const bool tail_on = getenv("MVK_GATHER_TAIL") == nullptr || atoi(getenv("MVK_GATHER_TAIL")) != 0;
  const bool branch_free = fastld_on && fast && Ns < (1 << 24) && (uint64_t)Ns * (uint64_t)Cin * 4ull < (1ull << 32) - 64;
  // first choice for rows of 4 m + t channels (m >= 15, t = 1..3): m lanes per point and the t trailing channels
  // beside the quads -- if the resulting chunk length is a multiple of the 4-row batch; else ceil(Cin / 4) lanes
  // (measured: 19 464 points x 66 channels 72 -> 67 us; 171 k points 484 -> 495 us -- the 4-row batches hide less
  // latency once the feature table no longer sits in the caches: hence the row limit)
  for (int with_tail = (tail_on && branch_free && !deform && NCH == 1 && Cin >= 60 && Cin % 4 != 0 &&
                        Ns <= 65536) ? 1 : 0;
       with_tail >= 0; --with_tail) {
    const int c4 = with_tail ? Cin / 4 : (Cin + 3) / 4;
    v.LPP = c4 < 64 ? c4 : 64;
    v.PPW = 64 / v.LPP;
    v.HC = 64 / v.PPW;
    v.fub = 0;
    v.tail = 0;
    // rows per batch of the branch-free variant: the divisor-like batch size with the least padding of HC
    if (branch_free && v.LPP >= 5) {
      if (NCH == 2) {
        v.fub = 4;
      } else {
        int best_pad = 1 << 30;
        for (int ub = 8; ub >= 5; --ub) {
          const int pad = (v.HC + ub - 1) / ub * ub - v.HC;
          if (pad < best_pad) {
            best_pad = pad;
            v.fub = ub;
          }
        }
      }
    }
    if (with_tail && v.fub > 0 && v.HC % 4 == 0) {      // batches of 4 rows: the trailing values take registers too
      v.tail = Cin % 4;
      v.fub = 4;
      break;
    }
  }
  // sharing workgroups (kernel comment): all of them for the deformable variant; for the branch-free rigid one
  // the workgroups beyond the last full round of independent waves, when that remainder is well below a round
  static const int split_env = getenv("MVK_GATHER_SPLIT") ? atoi(getenv("MVK_GATHER_SPLIT")) : -1;
  const int chunks = (H + v.HC - 1) / v.HC;
  const int64_t groups = cdiv64(Nq, v.PPW);               // point groups = independent waves
  const int64_t plain_wgs = cdiv64(groups, GWPB);
  const int64_t slots = NCH == 1 ? 1024 : 512;           // resident workgroups: 256 CUs x (4 | 2)
  int64_t b1 = plain_wgs;                                 // first sharing workgroup
  if (v.fub > 0 && chunks > 1 && split_env != 0) {
    const int64_t full = plain_wgs / slots * slots;
    const int64_t rem = groups - full * GWPB;             // groups left after the full rounds
    if (split_env == 1) b1 = 0;
    else if (rem > 0 && rem * 10 <= slots * GWPB * 7) b1 = full;
  }
  if (deform) b1 = 0;
  const bool any_share = b1 < plain_wgs || deform;
  v.nw = (b1 == 0 && any_share) ? (chunks < GWPB ? (chunks < 1 ? 1 : chunks) : GWPB) : GWPB;
  v.SW = any_share ? GWPB : 1;
  v.B1 = (int)b1;
  v.wgs = any_share ? b1 + (groups - b1 * GWPB) : plain_wgs;
  return v;
}

template <int NCH, bool DEFORM = false>
int launch_vec(KPParams P, int idx64, hipStream_t st) {
  const bool fast = P.influence == MVK_INFL_LINEAR && P.aggregation == MVK_AGG_SUM;
  const VecPlan v = plan_vec(P.Nq, P.Ns, P.H, P.Cin, fast, DEFORM);
  static const bool xcd_runs = getenv("MVK_GATHER_XCD_RUNS") == nullptr || atoi(getenv("MVK_GATHER_XCD_RUNS")) != 0;
  if (DEFORM) P.order = nullptr;
  P.xcd_blocks = (P.order != nullptr && xcd_runs) ? (int)(v.SW > 1 ? v.B1 : v.wgs) : 0;
  const int LPP = v.LPP, PPW = v.PPW, HC = v.HC, fub = v.fub, SW = v.SW, B1 = v.B1, tail = v.tail;
  dim3 grid((unsigned)v.wgs), block(64 * v.nw);
#define LV(I64, F, L) \
  hipLaunchKernelGGL((kpconv_gather_vec<NCH, I64, F, DEFORM, L>), grid, block, 0, st, P, LPP, PPW, HC, SW, B1)
#define LVT(I64, T)                                                                                                   \
  hipLaunchKernelGGL((kpconv_gather_vec<1, I64, true, false, 4, (TAIL_OK ? T : 0)>), grid, block, 0, st, P, LPP, \
                     PPW, HC, SW, B1)
  constexpr bool TAIL_OK = NCH == 1 && !DEFORM;
#define LVF(I64)                                                                            \
  if (TAIL_OK && tail == 1 && fub == 4) LVT(I64, 1);                                        \
  else if (TAIL_OK && tail == 2 && fub == 4) LVT(I64, 2);                                   \
  else if (TAIL_OK && tail == 3 && fub == 4) LVT(I64, 3);                                   \
  else if (NCH == 2 && fub == 4) LV(I64, true, (NCH == 2 ? 4 : 0));                  \
  else if (fub == 8) LV(I64, true, (NCH == 1 ? 8 : 0));                              \
  else if (fub == 7) LV(I64, true, (NCH == 1 ? 7 : 0));                              \
  else if (fub == 6) LV(I64, true, (NCH == 1 ? 6 : 0));                              \
  else if (fub == 5) LV(I64, true, (NCH == 1 ? 5 : 0));                              \
  else if (fast) LV(I64, true, 0);                                                          \
  else LV(I64, false, 0);
  if (idx64) {
    LVF(true)
  } else {
    LVF(false)
  }
#undef LVF
#undef LVT
#undef LV
  return 0;
}

// channel tiles per wave of the MFMA gather for rows of Cin channels (0: the layer stays on the vector kernels)
int mfma_tiles(int64_t Ns, int Cin, int K, int influence, int aggregation) {
  static const bool on = getenv("MVK_GATHER_MFMA") == nullptr || atoi(getenv("MVK_GATHER_MFMA")) != 0;
  static const int min_cin = getenv("MVK_GATHER_MFMA_MIN_CIN") ? atoi(getenv("MVK_GATHER_MFMA_MIN_CIN")) : 1;      // (5: rows of <= 4 channels stay on kpconv_gather_small)
  if (!on || influence != MVK_INFL_LINEAR || aggregation != MVK_AGG_SUM || K > 16 || Cin < min_cin) return 0;
  if ((uint64_t)Ns * (uint64_t)Cin >= (1ull << 32) - 4096) return 0;
  if (Cin <= 32) return 2;
  if (Cin <= 64) return 4;
  if (Cin <= 80) return 5;            // 66: the early-fusion net's first layer
  if (Cin <= 128) return 8;
  return 16;                          // wider rows: gridDim.y blocks of 256 channels
}

// launch shape of the MFMA gather (shared by the launch and by mvk_kpconv_gather_plan): T = 0: not on this kernel
struct MfmaPlan {
  int T, SW;               // accumulator tiles per wave; waves sharing one point (1 or 4)
  int64_t wgs, blocks_y;   // grid
};
MfmaPlan plan_mfma(int64_t Nq, int64_t Ns, int H, int Cin, int K, int influence, int aggregation, bool deform) {
  MfmaPlan m{};
  static const bool mfma_deform = getenv("MVK_DEFORM_MFMA") == nullptr || atoi(getenv("MVK_DEFORM_MFMA")) != 0;
  if (deform && !mfma_deform) return m;
  m.T = mfma_tiles(Ns, Cin, K, influence, aggregation);
  if (m.T == 0) return m;
  // few points with long rows: the four waves of a workgroup share one point (kernel comment)
  static const int sw_env = getenv("MVK_GATHER_MFMA_SHARE") ? atoi(getenv("MVK_GATHER_MFMA_SHARE")) : -1;
  m.blocks_y = cdiv64(Cin, 16 * m.T);
  const bool can_share = m.T >= 4 && m.T != 5;          // (instantiated for the tile counts the coarse levels use)
  m.SW = !can_share ? 1 : (sw_env >= 0 ? (sw_env > 1 ? 4 : 1) : ((H >= 128 || (Nq * m.blocks_y <= 512 && H >= 32)) ? 4 : 1));
  m.wgs = m.SW > 1 ? Nq : cdiv64(Nq, 4);
  return m;
}

template <int KPM>
bool launch_mfma(KPParams P, int idx64, hipStream_t st) {
  const MfmaPlan mp = plan_mfma(P.Nq, P.Ns, P.H, P.Cin, P.K, P.influence, P.aggregation, KPM == 1);
  const int T = mp.T;
  if (T == 0) return false;
  static const bool xcd_runs = getenv("MVK_GATHER_XCD_RUNS") == nullptr || atoi(getenv("MVK_GATHER_XCD_RUNS")) != 0;
  const int SW = mp.SW;
  const int64_t wgs = mp.wgs;
  P.xcd_blocks = (P.order != nullptr && xcd_runs) ? (int)wgs : 0;
  dim3 grid((unsigned)wgs, (unsigned)mp.blocks_y), block(256);
#define LM1(TT, MD, SH)                                                                              \
  if (idx64) hipLaunchKernelGGL((kpconv_gather_mfma<TT, MD, true, KPM, SH>), grid, block, 0, st, P);     \
  else hipLaunchKernelGGL((kpconv_gather_mfma<TT, MD, false, KPM, SH>), grid, block, 0, st, P)
#define LM(TT, MD)                                          \
  if (SW > 1 && TT >= 4 && TT != 5) { LM1(TT, MD, (TT >= 4 && TT != 5)); } else { LM1(TT, MD, false); }
  // wide loads where the row length allows them (kernel comment): rows of a multiple of 16 T channels from a 16-byte
  // aligned table; the 64 + tail layout for even rows of 66 .. 80 channels; one dword per tile otherwise
  const bool wide = P.Cin % (16 * T) == 0 && ((uintptr_t)P.x & 15) == 0 && ((uintptr_t)P.A & 15) == 0;
  const bool tail5 = T == 5 && P.Cin > 64 && P.Cin % 2 == 0 && ((uintptr_t)P.x & 7) == 0 && ((uintptr_t)P.A & 7) == 0;
  if (T == 2) { if (wide) { LM(2, 1); } else { LM(2, 0); } }
  else if (T == 4) { if (wide) { LM(4, 1); } else { LM(4, 0); } }
  else if (T == 5) { if (tail5 && KPM == 0) { LM(5, 2); } else { LM(5, 0); } }
  else if (T == 8) { if (wide) { LM(8, 1); } else { LM(8, 0); } }
  else { if (wide) { LM(16, 1); } else { LM(16, 0); } }
#undef LM1
#undef LM
  return true;
}

template <int MODE, bool DEFORM>
int launch_lane_channel(const KPParams& P, int idx64, hipStream_t st) {
  // scatter of a layer with more than one 64-neighbour chunk (searched at the deform radius): four waves per point
  static const bool split_on = getenv("MVK_SCATTER_SPLIT") == nullptr || atoi(getenv("MVK_SCATTER_SPLIT")) != 0;
  static const bool spread_on = getenv("MVK_SCATTER_SPREAD") == nullptr || atoi(getenv("MVK_SCATTER_SPREAD")) != 0;
  const bool split = MODE == 1 && split_on && P.H > 64;
  if (MODE == 1 && !DEFORM && !split && spread_on && P.Nq < 2048) {
    // few points (coarse levels): one wave per (point, 64-channel slot), and below ~4 000 such waves four waves per
    // slot that take every fourth neighbour (see the kernel's HS parameter)
    const int slots = (P.Cin + 63) / 64;
    const bool hs = (int64_t)P.Nq * slots < 4096;
    dim3 grid((unsigned)P.Nq, (unsigned)slots), block(hs ? 256 : 64);
    if (hs) {
      if (idx64) hipLaunchKernelGGL((kpconv_lane_channel<1, true, 1, false, 1, 4>), grid, block, 0, st, P, 0);
      else hipLaunchKernelGGL((kpconv_lane_channel<1, false, 1, false, 1, 4>), grid, block, 0, st, P, 0);
    } else {
      if (idx64) hipLaunchKernelGGL((kpconv_lane_channel<1, true, 1, false, 1, 1>), grid, block, 0, st, P, 0);
      else hipLaunchKernelGGL((kpconv_lane_channel<1, false, 1, false, 1, 1>), grid, block, 0, st, P, 0);
    }
    return 0;
  }
  dim3 grid((unsigned)P.Nq), block(split ? 256 : 64);
  for (int c0 = 0; c0 < P.Cin; c0 += 512) {
    int cw = P.Cin - c0 < 512 ? P.Cin - c0 : 512;
    int ns = (cw + 63) / 64;
#define LC(NS)                                                                                       \
  if (split) {                                                                                       \
    if (idx64)                                                                                       \
      hipLaunchKernelGGL((kpconv_lane_channel<NS, true, MODE, DEFORM, (MODE == 1 ? 4 : 1)>), grid, block, 0, st, P, c0);  \
    else                                                                                             \
      hipLaunchKernelGGL((kpconv_lane_channel<NS, false, MODE, DEFORM, (MODE == 1 ? 4 : 1)>), grid, block, 0, st, P, c0); \
  } else if (idx64)                                                                                  \
    hipLaunchKernelGGL((kpconv_lane_channel<NS, true, MODE, DEFORM>), grid, block, 0, st, P, c0);    \
  else                                                                                               \
    hipLaunchKernelGGL((kpconv_lane_channel<NS, false, MODE, DEFORM>), grid, block, 0, st, P, c0);
    if (ns <= 1) {
      LC(1)
    } else if (ns <= 2) {
      LC(2)
    } else if (ns <= 4) {
      LC(4)
    } else {
      LC(8)
    }
#undef LC
  }
  return 0;
}

int check_common(int64_t Nq, int64_t Ns, int H, int Cin, int K, int influence, int aggregation) {
  MVK_REQUIRE(Nq >= 0 && Ns >= 0 && H >= 0 && Cin > 0, "kpconv: bad sizes Nq=%lld Ns=%lld H=%d Cin=%d",
              (long long)Nq, (long long)Ns, H, Cin);
  MVK_REQUIRE(K >= 1 && K < KMAX, "kpconv: kernel size K=%d unsupported (1..%d)", K, KMAX - 1);
  MVK_REQUIRE(influence >= 0 && influence <= 2, "Unknown influence function type (config.KP_influence)");
  MVK_REQUIRE(aggregation == 0 || aggregation == 1, "Unknown convolution mode. Should be 'closest' or 'sum'");
  MVK_REQUIRE(Nq < (1ll << 31), "kpconv: Nq too large for one launch");
  return 0;
}

}  // namespace

extern "C" int mvk_kpconv_gather_fwd(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                     const void* idx, int idx64, int H, const float* x, int Cin,
                                     const float* kp, int K, float extent, int influence,
                                     int aggregation, const float* offsets, float* min_d2,
                                     int32_t* min_arg, float* A_out, void* stream) {
  return mvk_kpconv_gather_fwd_ordered(q, Nq, s, Ns, idx, idx64, H, x, Cin, kp, K, extent, influence, aggregation, offsets,
                                       min_d2, min_arg, A_out, nullptr, stream);
}

extern "C" int mvk_kpconv_gather_fwd_ordered(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                             const void* idx, int idx64, int H, const float* x, int Cin,
                                             const float* kp, int K, float extent, int influence,
                                             int aggregation, const float* offsets, float* min_d2,
                                             int32_t* min_arg, float* A_out, const int32_t* order, void* stream) {
  if (int e = check_common(Nq, Ns, H, Cin, K, influence, aggregation)) return e;
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  KPParams P{};
  P.order = order;        // used by the vector kernels of rigid layers (5 <= Cin <= 512); the other kernels work in row order
  P.q = q; P.s = s; P.idx = idx; P.x = x; P.kp = kp; P.offsets = offsets; P.min_d2 = min_d2; P.min_arg = min_arg;
  P.A = A_out; P.Nq = Nq; P.Ns = Ns; P.H = H; P.Cin = Cin; P.K = K; P.extent = extent;
  P.influence = influence; P.aggregation = aggregation;
  static const int zero_skip = getenv("MVK_GATHER_ZEROSKIP") ? atoi(getenv("MVK_GATHER_ZEROSKIP")) : 1;
  P.zero_skip = zero_skip;
  if (H == 0) {
    MVK_CHECK_HIP(hipMemsetAsync(A_out, 0, sizeof(float) * Nq * K * Cin, st));
    if (offsets == nullptr) return 0;
  }
  if (offsets != nullptr) {
    static const bool vec_deform = getenv("MVK_DEFORM_VEC") == nullptr || atoi(getenv("MVK_DEFORM_VEC")) != 0;
    KPParams Pm = P;
    Pm.order = nullptr;           // (the deformable levels are the coarse ones: no work list)
    if (launch_mfma<1>(Pm, idx64, st)) {
      // (round 5: the deformable forward on the matrix pipe as well, KPM 1 of kpconv_gather_mfma)
    } else if (vec_deform && Cin >= 13 && Cin <= 256) {          // 64 / ceil(Cin/4) <= DPPW points per wave
      launch_vec<1, true>(P, idx64, st);
    } else if (vec_deform && Cin > 256 && Cin <= 512) {
      launch_vec<2, true>(P, idx64, st);
    } else {
      launch_lane_channel<0, true>(P, idx64, st);
    }
  } else if (Cin <= 4 && launch_mfma<0>(P, idx64, st)) {
    // (narrow rows -- the first layer of the baseline / middle / late nets -- on the matrix pipe too: 21.7 against 28.8 us
    // for 19 464 points x 4 channels; most of the 16 x 16 tile is idle, but the kernel's cost there is the geometry)
  } else if (Cin <= 4) {
    const bool fast = influence == MVK_INFL_LINEAR && aggregation == MVK_AGG_SUM;
    dim3 grid((unsigned)cdiv64(Nq * 4, 256)), block(256);       // four lanes per point
    if (idx64) {
      if (fast) hipLaunchKernelGGL((kpconv_gather_small<true, true>), grid, block, 0, st, P);
      else hipLaunchKernelGGL((kpconv_gather_small<true, false>), grid, block, 0, st, P);
    } else {
      if (fast) hipLaunchKernelGGL((kpconv_gather_small<false, true>), grid, block, 0, st, P);
      else hipLaunchKernelGGL((kpconv_gather_small<false, false>), grid, block, 0, st, P);
    }
  } else if (launch_mfma<0>(P, idx64, st)) {
    // (the aggregation on the matrix pipe: linear influence, sum aggregation, feature table < 2^32 elements)
  } else if (Cin <= 256) {
    launch_vec<1>(P, idx64, st);
  } else if (Cin <= 512) {
    launch_vec<2>(P, idx64, st);
  } else {
    launch_lane_channel<0, false>(P, idx64, st);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// Gather-form feature gradient of a DEFORMABLE KPConv (round 5; blocks.py:286-327, :360 through autograd). With the
// transposed neighbourhood relation rev [Ns, Hr] (row j = the query rows n whose list holds support j; entries outside
// [0, Nq) are shadow entries), g [Nq, C] the gradient of the layer's output and the layer's own deformed kernel points
// kp [K, 3] + offsets [Nq, K, 3] and modulations mod [Nq, K] (null: not modulated):
//   A2 [Ns, K, C] [j, k, :] = sum over n in rev[j] of max(0, 1 - |s_j - q_n - kp_k - offsets[n, k]| / extent) mod[n, k] g[n, :]
// i.e. the forward aggregation over the transposed relation with the kernel points of the NEIGHBOUR rows; the feature
// gradient is then dx = sum_k A2[:, k, :] . W[k]^T (mvk_gemm_f32_kp_transposed) -- no atomics, a fixed summation order per
// row of rev. Linear influence, sum aggregation, 5 <= C, Nq * C < 2^32 (returns an error otherwise: the caller keeps
// mvk_kpconv_scatter_bwd). order: a work list over the Ns rows or NULL.
extern "C" int mvk_kpconv_gather_rev_deform(const float* s, int64_t Ns, const float* q, int64_t Nq, const void* rev, int rev64,
                                            int Hr, const float* g, int C, const float* kp, int K, float extent,
                                            const float* offsets, const float* mod, float* A2, const int32_t* order,
                                            void* stream) {
  if (int e = check_common(Ns, Nq, Hr, C, K, MVK_INFL_LINEAR, MVK_AGG_SUM)) return e;
  MVK_REQUIRE(offsets != nullptr && A2 != nullptr, "kpconv rev deform: offsets and an output are required");
  if (Ns == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (Hr == 0 || Nq == 0) {
    MVK_CHECK_HIP(hipMemsetAsync(A2, 0, sizeof(float) * Ns * K * C, st));
    return 0;
  }
  KPParams P{};
  // roles transposed: the "queries" of the launch are the layer's supports, its "supports" the layer's query rows
  P.q = s; P.s = q; P.idx = rev; P.x = g; P.kp = kp; P.A = A2; P.Nq = Ns; P.Ns = Nq; P.H = Hr; P.Cin = C; P.K = K;
  P.extent = extent; P.influence = MVK_INFL_LINEAR; P.aggregation = MVK_AGG_SUM;
  P.order = order; P.nb_offsets = offsets; P.nb_mod = mod;
  MVK_REQUIRE(launch_mfma<2>(P, rev64, st), "kpconv rev deform: shape not supported by the MFMA gather (C >= 5, Nq * C < 2^32)");
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_kpconv_scatter_bwd(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                      const void* idx, int idx64, int H, int Cin, const float* kp,
                                      int K, float extent, int influence, int aggregation,
                                      const float* dA, float* dx, const float* x,
                                      const float* offsets, const float* g_min_d2, const int32_t* min_arg,
                                      float* d_offsets, void* stream) {
  if (int e = check_common(Nq, Ns, H, Cin, K, influence, aggregation)) return e;
  if (Nq == 0 || H == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  KPParams P{};
  P.q = q; P.s = s; P.idx = idx; P.x = x; P.kp = kp; P.offsets = offsets;
  P.A = const_cast<float*>(dA); P.dx = dx; P.g_min_d2 = g_min_d2; P.d_offsets = d_offsets;
  P.Nq = Nq; P.Ns = Ns; P.H = H; P.Cin = Cin; P.K = K; P.extent = extent;
  P.influence = influence; P.aggregation = aggregation;
  if (offsets != nullptr) {
    MVK_REQUIRE(x != nullptr && d_offsets != nullptr, "kpconv bwd: deformable needs x and d_offsets");
    MVK_REQUIRE(aggregation == MVK_AGG_SUM, "kpconv bwd: deformable + 'closest' aggregation has no offset gradient path");
    launch_lane_channel<1, true>(P, idx64, st);      // dx: same scatter as the rigid layers, deformed weights + in-range filter
    MVK_CHECK_HIP(hipGetLastError());
    // d_offsets (A path and min_d2 path): one wave per point, lane = neighbour (csrc/deform.hip)
    return mvk_kpconv_deform_doff(q, Nq, s, Ns, idx, idx64, H, x, Cin, kp, K, extent, influence, offsets, dA, g_min_d2,
                                  min_arg, d_offsets, stream);
  } else {
    launch_lane_channel<1, false>(P, idx64, st);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// Launch geometry mvk_kpconv_gather_fwd uses for a layer (elem_bytes: 4, the f32 rows -- the only row type)
// with linear influence and sum aggregation: out[0..6] = lanes per point, points per wave, rows per batch of the
// branch-free variant (0 = general variant), first sharing workgroup, waves per workgroup, workgroups, grid
// threads (what a kernel trace reports). out[5] = 0: the layer runs on another kernel (one point per wave, or one
// point per lane for rows of <= 4 channels).
extern "C" int mvk_kpconv_gather_plan(int64_t Nq, int64_t Ns, int H, int Cin, int elem_bytes, int deformable,
                                      int64_t* out) {
  MVK_REQUIRE(out != nullptr && elem_bytes == 4, "kpconv plan: bad arguments (feature rows are f32: elem_bytes 4)");
  for (int i = 0; i < 8; ++i) out[i] = 0;
  const MfmaPlan mp = plan_mfma(Nq, Ns, H, Cin, 15, MVK_INFL_LINEAR, MVK_AGG_SUM, deformable != 0);
  if (Nq > 0 && mp.T > 0) {        // the MFMA gather: four waves per workgroup (one point each, or sharing one), channel blocks in gridDim.y
    out[0] = 64; out[1] = 1; out[2] = mp.T; out[3] = mp.SW > 1 ? 0 : -1; out[4] = 4; out[5] = mp.wgs * mp.blocks_y;
    out[6] = out[5] * 256; out[7] = 1;
    return 0;
  }
  if (Nq <= 0 || Cin <= 0 || Cin > 512 || (deformable && Cin < 13) || (!deformable && Cin <= 4)) return 0;      // (one point per 4 lanes: kpconv_gather_small)
  const VecPlan v = plan_vec(Nq, Ns, H, Cin, true, deformable != 0);
  out[0] = v.LPP; out[1] = v.PPW; out[2] = v.fub; out[3] = v.B1; out[4] = v.nw; out[5] = v.wgs;
  out[6] = v.wgs * 64 * v.nw;
  return 0;
}
