// Reverse neighbour lists: rev[j] = the query rows n whose neighbour row idx[n, :] holds support j, ascending.
//
// The backward of KPConv.forward's feature gather (KPConv-PyTorch/models/blocks.py:52-64 `gather`, :360 `neighb_x =
// gather(x, new_neighb_inds)`) is dx[idx[n,h]] += sum_k w[n,h,k] dA[n,k,:] -- on the reference's CPU path
// an ordered index-add, on a GPU a scatter with float atomics (order of the additions = order of arrival). With the
// reverse lists the same sums run as a GATHER: dx[j] = (sum over n in rev[j] of w_k(q_n - s_j) g[n]) . W_k^T, i.e. the
// FORWARD kernel of csrc/kpconv.hip over the transposed neighbourhood relation with the kernel points negated --
// every sum in a fixed order (ascending n): bit-identical from run to run, and no atomic traffic.
//
// Two launches per neighbour matrix (input side of a step, beside the neighbour searches):
//   rev_fill_kernel   one thread per entry (n, h): slot = atomicAdd(count[j], 1) (integer: the COUNT is order
//                     independent, the slot is not), rev[j][slot] = n; entries beyond the row capacity raise the
//                     overflow word;
//   rev_sort_kernel   one wavefront per support row: the row's entries ranked by counting (they are distinct), written
//                     back ascending, the tail padded with the shadow value, count[j] reset to zero (the counters are
//                     a persistent, self-cleaning buffer) and the longest row reported.
#include "common.h"

namespace {

template <bool IDX64>
__global__ __launch_bounds__(256) void rev_fill_kernel(const void* __restrict__ idx, int64_t Nq, int H, int64_t stride,
                                                       int64_t Ns, int32_t* __restrict__ rev, int Hr,
                                                       int32_t* __restrict__ count, int32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= Nq * H) return;
  const int j = load_idx<IDX64>(idx, (e / H) * stride + e % H, Ns);
  if (j < 0) return;
  const int slot = atomicAdd(count + j, 1);
  if (slot < Hr)
    rev[(int64_t)j * Hr + slot] = (int32_t)(e / H);
  else if (status)
    atomicOr(status + 1, 1);
}

// RW: waves per workgroup. A row of up to 64 * EPL entries: lane l keeps entries l, l + 64, ...
template <int EPL, bool SORT>
__global__ __launch_bounds__(256) void rev_sort_kernel(int64_t Ns, int32_t* __restrict__ rev, int Hr, int32_t shadow,
                                                       int32_t* __restrict__ count, int32_t* __restrict__ status) {
  __shared__ int32_t buf[4][64 * EPL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int longest = 0;
  for (int64_t j = (int64_t)blockIdx.x * 4 + wv; j < Ns; j += (int64_t)gridDim.x * 4) {
    const int c_all = count[j];
    const int c = c_all < Hr ? c_all : Hr;
    longest = c_all > longest ? c_all : longest;
    int32_t* row = rev + j * Hr;
    if (!SORT) {        // arrival order is good enough when run-to-run identical sums are not asked for
      for (int i = c + lane; i < Hr; i += 64) row[i] = shadow;
      if (lane == 0 && c_all != 0) count[j] = 0;
      continue;
    }
    int32_t v[EPL];
#pragma unroll
    for (int u = 0; u < EPL; ++u) {
      const int i = lane + 64 * u;
      v[u] = i < c ? row[i] : 0x7fffffff;
      buf[wv][i] = v[u];
    }
    __builtin_amdgcn_wave_barrier();
    // rank of each own entry = number of smaller entries (entries are distinct query rows); LDS broadcast reads
    int rank[EPL];
#pragma unroll
    for (int u = 0; u < EPL; ++u) rank[u] = 0;
    for (int i = 0; i < c; ++i) {
      const int32_t w = buf[wv][i];
#pragma unroll
      for (int u = 0; u < EPL; ++u) rank[u] += w < v[u] ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < EPL; ++u) {
      const int i = lane + 64 * u;
      if (i < c) row[rank[u]] = v[u];
    }
    for (int i = c + lane; i < Hr; i += 64) row[i] = shadow;
    if (lane == 0 && c_all != 0) count[j] = 0;
  }
  if (status) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int t = __shfl_xor(longest, o);
      longest = t > longest ? t : longest;
    }
    if (lane == 0 && longest > 0 && longest > status[0]) atomicMax(status, longest);
  }
}

// Sorted rows wider than 512 entries (the relations of the deformable layers, searched at the deform radius, in the
// deterministic mode): one WORKGROUP per row, the row staged in LDS, every thread ranks its entries tid, tid + 256, ...
// by counting and writes them to their places. Same contract as rev_sort_kernel<.., true>.
__global__ __launch_bounds__(256) void rev_sort_wide_kernel(int64_t Ns, int32_t* __restrict__ rev, int Hr, int32_t shadow,
                                                            int32_t* __restrict__ count, int32_t* __restrict__ status) {
  __shared__ int32_t buf[MVK_REV_MAX_WIDTH];
  int longest = 0;
  for (int64_t j = blockIdx.x; j < Ns; j += gridDim.x) {
    const int c_all = count[j];
    const int c = c_all < Hr ? c_all : Hr;
    longest = c_all > longest ? c_all : longest;
    int32_t* row = rev + j * Hr;
    for (int i = threadIdx.x; i < c; i += 256) buf[i] = row[i];
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) {
      const int32_t v = buf[i];
      int rank = 0;
      for (int t = 0; t < c; ++t) rank += buf[t] < v ? 1 : 0;          // broadcast reads; the entries are distinct
      row[rank] = v;
    }
    for (int i = c + threadIdx.x; i < Hr; i += 256) row[i] = shadow;
    __syncthreads();                                                     // the row's counter was read by everyone; buf is free
    if (threadIdx.x == 0 && c_all != 0) count[j] = 0;
  }
  if (status && threadIdx.x == 0 && longest > 0 && longest > status[0]) atomicMax(status, longest);
}

// The tails of up to MVK_REV_MANY lists in ONE launch (round 5): the lists' rows were filled by the neighbour searches
// themselves (mvk_radius_neighbors_dev_rev), their counters hold the row lengths. One wavefront per row of the
// concatenated row range: pads the tail with the list's shadow value, returns the counter to zero, reports the longest row.
struct RevManyArgs {
  mvk_rev_list l[MVK_REV_MANY];
  int64_t row_end[MVK_REV_MANY];      // prefix sums of the lists' row counts
  int n;
};

__global__ __launch_bounds__(256) void rev_finish_many_kernel(const RevManyArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t total = A.row_end[A.n - 1];
  int longest[MVK_REV_MANY];
#pragma unroll
  for (int k = 0; k < MVK_REV_MANY; ++k) longest[k] = 0;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < total; r += (int64_t)gridDim.x * 4) {
    int k = 0;
    while (k + 1 < A.n && r >= A.row_end[k]) ++k;                 // wave-uniform: a handful of lists
    const int64_t j = r - (k > 0 ? A.row_end[k - 1] : 0);
    const mvk_rev_list& L = A.l[k];
    const int c_all = L.counts[j];
    const int c = c_all < L.width ? c_all : L.width;
    int32_t* row = L.rev + j * L.width;
    for (int i = c + lane; i < L.width; i += 64) row[i] = L.shadow;
    if (lane == 0 && c_all != 0) L.counts[j] = 0;
#pragma unroll
    for (int kk = 0; kk < MVK_REV_MANY; ++kk)
      if (kk == k) longest[kk] = c_all > longest[kk] ? c_all : longest[kk];
  }
#pragma unroll
  for (int k = 0; k < MVK_REV_MANY; ++k) {
    if (k >= A.n || !A.l[k].status) continue;
    int v = longest[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int t = __shfl_xor(v, o);
      v = t > v ? t : v;
    }
    if (lane == 0 && v > 0 && v > A.l[k].status[0]) atomicMax(A.l[k].status, v);
  }
}

}  // namespace

// Finishes n <= MVK_REV_MANY reverse lists whose rows were filled by mvk_radius_neighbors_dev_rev (or by any producer
// that left the row lengths in `counts`): tails padded with `shadow`, counters back at zero, status[0] = max(status[0],
// longest row). One launch.
extern "C" int mvk_reverse_finish_many(const mvk_rev_list* lists, int n, void* stream) {
  MVK_REQUIRE(lists && n >= 1 && n <= MVK_REV_MANY, "reverse finish: 1..%d lists", MVK_REV_MANY);
  RevManyArgs A;
  int64_t rows = 0;
  for (int k = 0; k < MVK_REV_MANY; ++k) {
    if (k < n) {
      MVK_REQUIRE(lists[k].rev && lists[k].counts && lists[k].rows >= 0 && lists[k].width >= 1 && lists[k].width <= MVK_REV_MAX_WIDTH,
                  "reverse finish: bad list %d", k);
      A.l[k] = lists[k];
      rows += lists[k].rows;
    } else {
      A.l[k] = mvk_rev_list{nullptr, nullptr, nullptr, 0, 0, 0};
    }
    A.row_end[k] = rows;
  }
  A.n = n;
  if (rows == 0) return 0;
  const int64_t want = cdiv64(rows, 4);
  const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(rev_finish_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// rev [Ns, Hr] int32 <- the transposed relation of idx (Nq rows of H entries, row n at idx + n * idx_stride; int32 /
// int64; entries outside [0, Ns) are shadow entries and are skipped): row j lists the rows n of idx that hold j --
// ASCENDING when sort != 0, in order of arrival otherwise -- and its tail is `shadow`.
// counts [Ns] int32 must be ZERO on entry and is zero again when the launches have run (a persistent buffer).
// status (int32 [2], may be null): [0] = max over calls of the longest row (atomicMax), [1] |= 1 when a row is longer
// than Hr (its surplus entries are dropped: the caller must treat that as an error). Hr <= MVK_REV_MAX_WIDTH.
// Two launches.
extern "C" int mvk_reverse_neighbors(const void* idx, int idx64, int64_t Nq, int H, int64_t idx_stride, int64_t Ns, int32_t* rev,
                                     int Hr, int32_t shadow, int sort, int32_t* counts, int32_t* status, void* stream) {
  // (sorted rows of <= 512 entries are ranked by counting in registers, one wave per row; wider ones -- the relations of
  // the deformable layers in the deterministic mode -- by a workgroup per row)
  MVK_REQUIRE(Nq >= 0 && H >= 0 && Ns >= 0 && Hr >= 1 && Hr <= MVK_REV_MAX_WIDTH && rev && counts && idx_stride >= H,
              "reverse neighbours: bad arguments (rows of at most %d entries)", MVK_REV_MAX_WIDTH);
  MVK_REQUIRE(Nq * (int64_t)H < (1ll << 40) && Ns < (1ll << 31), "reverse neighbours: too large");
  if (Ns == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = Nq * H;
  if (n > 0) {
    const unsigned blocks = (unsigned)cdiv64(n, 256);
    if (idx64) hipLaunchKernelGGL(rev_fill_kernel<true>, dim3(blocks), dim3(256), 0, st, idx, Nq, H, idx_stride, Ns, rev, Hr, counts, status);
    else hipLaunchKernelGGL(rev_fill_kernel<false>, dim3(blocks), dim3(256), 0, st, idx, Nq, H, idx_stride, Ns, rev, Hr, counts, status);
  }
  const int64_t want = cdiv64(Ns, 4);
  const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
#define REV_SORT(E, S) hipLaunchKernelGGL((rev_sort_kernel<E, S>), dim3(blocks), dim3(256), 0, st, Ns, rev, Hr, shadow, counts, status)
  if (!sort) REV_SORT(1, false);
  else if (Hr <= 64) REV_SORT(1, true);
  else if (Hr <= 128) REV_SORT(2, true);
  else if (Hr <= 256) REV_SORT(4, true);
  else if (Hr <= 512) REV_SORT(8, true);
  else hipLaunchKernelGGL(rev_sort_wide_kernel, dim3((unsigned)(Ns < 65536 ? Ns : 65536)), dim3(256), 0, st, Ns, rev, Hr, shadow, counts, status);
#undef REV_SORT
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace {

// out[j, c] (+ base[j, c]) = sum over the entries n of rev[j] of g[n, c] -- the backward of a row gather x[idx[n]]
// (closest_pool / nearest upsampling, blocks.py:79-91: the reference's index_select backward) as a gather over the
// transposed relation: fixed summation order, no atomics, every element written. One thread per (row, 4 channels).
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const float* __restrict__ g, int64_t ldg, int64_t Nq,
                                                              const int32_t* __restrict__ rev, int Hr, int64_t Ns, int C,
                                                              const float* __restrict__ base, float* __restrict__ out) {
  const int cq = (C + 3) / 4;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= Ns * cq) return;
  const int64_t j = t / cq;
  const int c = (int)(t % cq) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (base)
    for (int e = 0; e < 4; ++e)
      if (c + e < C) acc[e] = base[j * C + c + e];
  const int32_t* row = rev + j * Hr;
  for (int i = 0; i < Hr; ++i) {
    const int32_t n = row[i];
    if (n < 0 || n >= Nq) break;        // rows are packed: the first shadow entry ends them
    const float* p = g + (int64_t)n * ldg + c;
    if (c + 3 < C && ((ldg | c) & 3) == 0 && ((uintptr_t)g & 15) == 0) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
    } else {
      for (int e = 0; e < 4; ++e)
        if (c + e < C) acc[e] += p[e];
    }
  }
  for (int e = 0; e < 4; ++e)
    if (c + e < C) out[j * C + c + e] = acc[e];
}

// out[j, c] (+ base[j, c]) = sum over n in rev[j] of (idx[n, arg[n, c]] == j ? g[n, c] : 0): the backward of max_pool
// (blocks.py:94-110: torch.max over the gathered neighbourhood routes each output's gradient to its arg-max row;
// arg = the winning COLUMN of the pooling matrix, as mvk_max_pool_fwd records it).
template <bool IDX64>
__global__ __launch_bounds__(256) void max_pool_bwd_gather_kernel(const float* __restrict__ g, const int32_t* __restrict__ arg,
                                                                  const void* __restrict__ idx, int H, int64_t Nq,
                                                                  const int32_t* __restrict__ rev, int Hr, int64_t Ns, int C,
                                                                  const float* __restrict__ base, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= Ns * C) return;
  const int64_t j = t / C;
  const int c = (int)(t % C);
  float acc = base ? base[t] : 0.f;
  const int32_t* row = rev + j * Hr;
  for (int i = 0; i < Hr; ++i) {
    const int32_t n = row[i];
    if (n < 0 || n >= Nq) break;
    const int h = arg[(int64_t)n * C + c];
    if (load_idx<IDX64>(idx, (int64_t)n * H + h, Ns) == (int)j) acc += g[(int64_t)n * C + c];
  }
  out[t] = acc;
}

}  // namespace

// out [Ns, C] = (base [Ns, C] or 0) + sum_{n in rev[j]} g[n, 0:C] with g rows ldg floats apart (a column block of a wider
// gradient is read in place). rev [Ns, Hr] from mvk_reverse_neighbors (packed rows, shadow tail).
extern "C" int mvk_gather_sum_rows(const float* g, int64_t ldg, int64_t Nq, const int32_t* rev, int Hr, int64_t Ns, int C,
                                   const float* base, float* out, void* stream) {
  MVK_REQUIRE(g && rev && out && Nq >= 0 && Ns >= 0 && Hr >= 1 && C >= 1 && ldg >= C, "gather_sum_rows: bad arguments");
  if (Ns == 0) return 0;
  const int64_t n = Ns * ((C + 3) / 4);
  hipLaunchKernelGGL(gather_sum_rows_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, g, ldg, Nq, rev, Hr,
                     Ns, C, base, out);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// dx [Ns, C] = (base or 0) + the max_pool gradient: arg [Nq, C] int32 = the winning column of idx [Nq, H] for every pooled
// element (mvk_max_pool_fwd), rev [Ns, Hr] the transposed pooling matrix (mvk_reverse_neighbors of idx).
extern "C" int mvk_max_pool_bwd_gather(const float* g, const int32_t* arg, const void* idx, int idx64, int H, int64_t Nq,
                                       const int32_t* rev, int Hr, int64_t Ns, int C, const float* base, float* dx,
                                       void* stream) {
  MVK_REQUIRE(g && arg && idx && rev && dx && Nq >= 0 && Ns >= 0 && Hr >= 1 && C >= 1 && H >= 1, "max_pool_bwd_gather: bad arguments");
  if (Ns == 0) return 0;
  const dim3 grid((unsigned)cdiv64(Ns * C, 256));
  if (idx64) hipLaunchKernelGGL(max_pool_bwd_gather_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, g, arg, idx, H, Nq, rev, Hr, Ns, C, base, dx);
  else hipLaunchKernelGGL(max_pool_bwd_gather_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, g, arg, idx, H, Nq, rev, Hr, Ns, C, base, dx);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
