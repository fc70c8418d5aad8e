// Fixed-radius neighbour search on gfx950 with the reference's result contract
// (KPConv-PyTorch/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332 + nanoflann
// radiusSearch): for every query, all supports OF THE SAME CLOUD with float32
//   d2 = ((dx*dx) + dy*dy) + dz*dz < radius*radius          (nanoflann.hpp:433-441, :249-251)
// sorted by ascending d2 (ties: ascending index), indices into the stacked support array,
// rows padded with Ns. Compiled with -ffp-contract=off so d2 is bit-identical to x86-64.
//
// The KD-tree is an implementation detail of the reference; here:
//   build  (one 1024-thread workgroup per cloud): support bounding box -> uniform grid with
//          cell >= 1.001 r (cells grow if the box would need too many) -> counting sort of the
//          supports by x-fastest linear cell id (histogram atomics, block scan, scatter of
//          {x,y,z,index} records). Rows of 3 x-adjacent cells are contiguous in that order.
//   query  (one wavefront per query): the 27-cell neighbourhood = 9 contiguous record ranges;
//          lanes stride over the flattened candidates (16-byte coalesced record loads), test
//          d2 < r2, wave-ballot compaction into an LDS list, rank-by-counting sort in LDS,
//          coalesced row write + padding.
// Phase 1 (count) and phase 2 (fill) of the C ABI run the same candidate scan.
#include "common.h"
#include "blockscan.h"

namespace {

constexpr int LIST_CAP = 1024;  // in-range neighbours per query held in LDS (8 KB / wave)

struct CloudGrid {
  float gmin[3];
  float cell;
  int dims[3];
  int cell_base;  // offset of this cloud's cell_start array
};

struct NbWs {
  CloudGrid* grids;   // B
  int* cell_start;    // sum_b (cap_b + 1)
  int* cell_fill;     // same size
  float4* recs;       // Ns records sorted by cell
  int* qoffs;         // B + 1
  int* soffs;         // B + 1
  const int* q_lens;  // device-lens entry: the kernels derive the offsets from the lengths themselves (B is a handful
  const int* s_lens;  // of clouds) -- one launch less per search; null: qoffs / soffs hold them
  int* counts;        // Nq
  int* maxcount;      // 1
  int* overflow;      // 1
};

__host__ __device__ inline int64_t cell_cap(int64_t ns) { return 4 * ns + 4096; }
// a cloud's cell array: cap + 1 counters, padded so that every cloud's array starts on a 16-byte boundary
__host__ __device__ inline int64_t cell_stride(int64_t ns) { return cell_cap(ns) + 4; }

__global__ __launch_bounds__(TPB) void nb_build_kernel(const float* __restrict__ s, NbWs W, float radius, int grid_only) {
  __shared__ float red[6][TPB / 64];
  __shared__ CloudGrid G;
  __shared__ int sh[TPB / 64 + 2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int off, n;
  int64_t base = 0;
  if (W.s_lens) {
    off = 0;
    for (int i = 0; i < b; ++i) {
      const int li = max(W.s_lens[i], 0);
      off += li;
      base += cell_stride(li);
    }
    n = max(W.s_lens[b], 0);
  } else {
    off = W.soffs[b];
    n = W.soffs[b + 1] - off;
    for (int i = 0; i < b; ++i) base += cell_stride(W.soffs[i + 1] - W.soffs[i]);
  }
  const float* P = s + (int64_t)off * 3;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += TPB)
    for (int c = 0; c < 3; ++c) {
      float v = P[i * 3 + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  for (int c = 0; c < 3; ++c) {
    float a = mn[c], z = mx[c];
    for (int o = 32; o >= 1; o >>= 1) {
      a = fminf(a, __shfl_xor(a, o));
      z = fmaxf(z, __shfl_xor(z, o));
    }
    if (lane == 0) {
      red[c][wv] = a;
      red[3 + c][wv] = z;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
      lo[c] = red[c][0];
      hi[c] = red[3 + c][0];
      for (int i = 1; i < TPB / 64; ++i) {
        lo[c] = fminf(lo[c], red[c][i]);
        hi[c] = fmaxf(hi[c], red[3 + c][i]);
      }
      if (n == 0) lo[c] = hi[c] = 0.f;
    }
    // cell slightly larger than r: a pair closer than r can then never be two cells apart,
    // whatever the rounding of (p - gmin) / cell (candidate filter only; the test is exact)
    float cell = radius * 1.001f;
    if (!(cell > 0.f)) cell = 1e-6f;
    const int64_t cap = cell_cap(n);
    int d[3];
    while (true) {
      int64_t prod = 1;
      bool ok = true;
      for (int c = 0; c < 3; ++c) {
        float e = floorf((hi[c] - lo[c]) / cell);
        if (!(e < 2.0e6f)) ok = false;
        d[c] = ok ? (int)e + 1 : 1;
        prod *= d[c];
        if (prod > cap) ok = false;
      }
      if (ok) break;
      cell *= 1.3f;
    }
    for (int c = 0; c < 3; ++c) {
      G.gmin[c] = lo[c];
      G.dims[c] = d[c];
    }
    G.cell = cell;
    G.cell_base = (int)base;
    W.grids[b] = G;
  }
  __syncthreads();
  const int ncell = G.dims[0] * G.dims[1] * G.dims[2];
  int* cstart = W.cell_start + base;
  int* cfill = W.cell_fill + base;
  for (int i = tid; i <= ncell; i += TPB) {
    cstart[i] = 0;
    cfill[i] = 0;
  }
  if (grid_only) return;          // multi-workgroup build: histogram, scan and scatter are launches of their own
  __syncthreads();
  const float gx = G.gmin[0], gy = G.gmin[1], gz = G.gmin[2], cell = G.cell;
  const int dx = G.dims[0], dy = G.dims[1];
  for (int i = tid; i < n; i += TPB) {
    int cx = (int)floorf((P[i * 3] - gx) / cell), cy = (int)floorf((P[i * 3 + 1] - gy) / cell),
        cz = (int)floorf((P[i * 3 + 2] - gz) / cell);
    atomicAdd(&cstart[(cz * dy + cy) * dx + cx], 1);
  }
  __syncthreads();
  block_scan_array(cstart, ncell + 1, sh, false);  // cstart[c] = first record of cell c; [ncell] = n
  for (int i = tid; i < n; i += TPB) {
    const float x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
    int cx = (int)floorf((x - gx) / cell), cy = (int)floorf((y - gy) / cell), cz = (int)floorf((z - gz) / cell);
    const int c = (cz * dy + cy) * dx + cx;
    const int pos = cstart[c] + atomicAdd(&cfill[c], 1);
    W.recs[off + pos] = make_float4(x, y, z, __int_as_float(off + i));  // stacked support index
  }
}

// ------------------------------------------------------------------------------------------------------------
// Multi-workgroup build (round 4). The counting sort of one workgroup per cloud is bound by what ONE compute unit's
// memory pipeline retires (a 19 464-point cloud: two rounds of 19 k scattered atomics and a scan over 64 k cells, ~100 us);
// for large clouds the same grid is built by four launches: the cloud's workgroup computes the grid and clears the
// cells, the histogram and the scatter run over 256-point chunks of all clouds, the scan in between is a kernel of its
// own (plain loads after the kernel boundary, a thread's cells as 16-byte accesses). Same cells, same record ranges;
// the order inside a cell is arrival order in both builds (the queries sort by (d2, index)).

__device__ __forceinline__ void support_range(const NbWs& W, int b, int& off, int& n, int64_t& base) {
  off = 0;
  base = 0;
  if (W.s_lens) {
    for (int i = 0; i < b; ++i) {
      const int li = max(W.s_lens[i], 0);
      off += li;
      base += cell_stride(li);
    }
    n = max(W.s_lens[b], 0);
  } else {
    off = W.soffs[b];
    n = W.soffs[b + 1] - off;
    for (int i = 0; i < b; ++i) base += cell_stride(W.soffs[i + 1] - W.soffs[i]);
  }
}

__device__ __forceinline__ int cell_of(const CloudGrid& G, float x, float y, float z) {
  const int cx = (int)floorf((x - G.gmin[0]) / G.cell), cy = (int)floorf((y - G.gmin[1]) / G.cell),
            cz = (int)floorf((z - G.gmin[2]) / G.cell);
  return (cz * G.dims[1] + cy) * G.dims[0] + cx;
}

__global__ __launch_bounds__(256) void nb_hist_kernel(const float* __restrict__ s, NbWs W) {
  const int b = blockIdx.y;
  int off, n;
  int64_t base;
  support_range(W, b, off, n, base);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const CloudGrid G = W.grids[b];
  const float* P = s + ((int64_t)off + i) * 3;
  atomicAdd(&W.cell_start[base + cell_of(G, P[0], P[1], P[2])], 1);
}

// exclusive prefix over the cloud's ncell + 1 counters, in place: a thread owns a run of 4 * Q consecutive cells
__global__ __launch_bounds__(TPB) void nb_scan_kernel(NbWs W) {
  __shared__ int sh[TPB / 64 + 2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const CloudGrid G = W.grids[b];
  const int n = G.dims[0] * G.dims[1] * G.dims[2] + 1;
  int* arr = W.cell_start + G.cell_base;
  constexpr int Q = 16;                       // quads per thread and pass: 64 cells, 65 536 per pass of the workgroup
  int carry = 0;
  if ((((uintptr_t)arr) & 15) == 0) {
    for (int s0 = 0; s0 < n; s0 += TPB * 4 * Q) {
      const int cnt = min(n - s0, TPB * 4 * Q);
      const int quads = ((cnt + TPB - 1) / TPB + 3) >> 2;          // per thread, <= Q
      const int beg = s0 + tid * quads * 4, lim = s0 + cnt;
      int4 v[Q];
      int sum = 0;
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const int q = beg + 4 * u;
        v[u] = (u < quads && q < lim) ? *(const int4*)(arr + q) : make_int4(0, 0, 0, 0);   // (the cell array has >= 4 words of slack)
        if (q + 1 >= lim) v[u].y = 0;
        if (q + 2 >= lim) v[u].z = 0;
        if (q + 3 >= lim) v[u].w = 0;
        sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      }
      int total;
      int run = carry + block_exclusive_scan(sum, &total, sh);
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const int q = beg + 4 * u;
        if (u < quads && q < lim) {
          int4 o;
          o.x = run;
          o.y = o.x + v[u].x;
          o.z = o.y + v[u].y;
          o.w = o.z + v[u].z;
          run = o.w + v[u].w;
          if (q + 3 < lim) *(int4*)(arr + q) = o;
          else {
            arr[q] = o.x;
            if (q + 1 < lim) arr[q + 1] = o.y;
            if (q + 2 < lim) arr[q + 2] = o.z;
          }
        }
      }
      carry += total;
    }
  } else {
    block_scan_array(arr, n, sh, false);
  }
}

__global__ __launch_bounds__(256) void nb_scatter_kernel(const float* __restrict__ s, NbWs W) {
  const int b = blockIdx.y;
  int off, n;
  int64_t base;
  support_range(W, b, off, n, base);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const CloudGrid G = W.grids[b];
  const float* P = s + ((int64_t)off + i) * 3;
  const float x = P[0], y = P[1], z = P[2];
  const int c = cell_of(G, x, y, z);
  const int pos = W.cell_start[base + c] + atomicAdd(&W.cell_fill[base + c], 1);
  W.recs[off + pos] = make_float4(x, y, z, __int_as_float(off + i));
}

// Work list for the KPConv gather out of a built grid: the stacked support rows sorted by (cloud, cell, row) -- the
// record order of the counting sort with the rows of a cell put in ascending order (the scatter's atomics leave them in
// arrival order; a cell holds a dozen rows), so the list is the same on every run. Up to 256 workgroups of 256 threads
// per cloud, one thread per cell; the last cloud's first workgroup appends the identity for the rows up to order_cap
// (capacity padding).
__global__ __launch_bounds__(TPB) void nb_cell_order_kernel(NbWs W, int B, int* __restrict__ order, int64_t order_cap) {
  const int b = blockIdx.x, tid = threadIdx.x;
  int off = 0, n;
  if (W.s_lens) {
    for (int i = 0; i < b; ++i) off += max(W.s_lens[i], 0);
    n = max(W.s_lens[b], 0);
  } else {
    off = W.soffs[b];
    n = W.soffs[b + 1] - off;
  }
  const CloudGrid G = W.grids[b];
  const int ncell = G.dims[0] * G.dims[1] * G.dims[2];
  const int* cstart = W.cell_start + G.cell_base;
  constexpr int CO_REG = 16;      // rows of a cell sorted in registers (a cell of the conv grid holds about a dozen)
  const int lane = tid & 63;
  // cells are dealt to the wavefronts of the cloud's workgroups round-robin (cell c -> wavefront c mod NW): the big cells
  // of a scan lie along its walls, i.e. in runs of the x-fastest cell order, and a wavefront works them off one after the
  // other -- with 64 consecutive cells per wavefront a few wavefronts held most of them (37 us; 7 us without big cells)
  const int NW = gridDim.y * (blockDim.x >> 6), gw = blockIdx.y * (blockDim.x >> 6) + (tid >> 6);
  for (int c0 = 0; c0 < ncell; c0 += 64 * NW) {
    const int c = c0 + lane * NW + gw;
    int beg = 0, m = 0;
    if (c < ncell) {
      beg = cstart[c];
      m = cstart[c + 1] - beg;
    }
    if (m <= CO_REG) {
      // all loads, then the rank of every row among the cell's rows, then all stores: no dependent round trips
      int v[CO_REG];
#pragma unroll
      for (int i = 0; i < CO_REG; ++i) v[i] = i < m ? __float_as_int(W.recs[off + beg + i].w) : 0x7fffffff;
#pragma unroll
      for (int i = 0; i < CO_REG; ++i) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < CO_REG; ++j) rank += v[j] < v[i] ? 1 : 0;      // rows are distinct; the fillers are largest
        if (i < m) order[off + beg + rank] = v[i];
      }
    }
    // larger cells (a sixth of the occupied cells of a room scan hold 17-33 rows), by the whole wavefront: a lane per row,
    // the rank of a row = how many rows of the cell are smaller, the others read lane by lane (v_readlane: the lane
    // number is wave-uniform). Four cells at a time so that their loads are in flight together -- one cell after the
    // other, each waiting for its own load and ranking through ds_bpermute, was 51 of the kernel's 59 us at level 0.
    unsigned long long big = __ballot(m > CO_REG);
    while (big) {
      int cb[4], cm[4], mine[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cb[k] = 0;
        cm[k] = 0;
        if (big) {
          const int src = __ffsll((long long)big) - 1;
          big &= big - 1;
          cb[k] = __builtin_amdgcn_readfirstlane(__shfl(beg, src));
          cm[k] = __builtin_amdgcn_readfirstlane(__shfl(m, src));
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        mine[k] = lane < cm[k] ? __float_as_int(W.recs[off + cb[k] + lane].w) : 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (cm[k] == 0) continue;
        if (cm[k] <= 64) {
          int rank = 0;
          for (int j = 0; j < cm[k]; ++j) rank += __builtin_amdgcn_readlane(mine[k], j) < mine[k] ? 1 : 0;
          if (lane < cm[k]) order[off + cb[k] + rank] = mine[k];
        } else {                 // more than 64 rows in a cell: 64 at a time against all
          for (int r0 = 0; r0 < cm[k]; r0 += 64) {
            const int me = r0 + lane < cm[k] ? __float_as_int(W.recs[off + cb[k] + r0 + lane].w) : 0x7fffffff;
            int rank = 0;
            for (int t0 = 0; t0 < cm[k]; t0 += 64) {
              const int other = t0 + lane < cm[k] ? __float_as_int(W.recs[off + cb[k] + t0 + lane].w) : 0x7fffffff;
              const int lim = min(64, cm[k] - t0);
              for (int j = 0; j < lim; ++j) rank += __builtin_amdgcn_readlane(other, j) < me ? 1 : 0;
            }
            if (r0 + lane < cm[k]) order[off + cb[k] + rank] = me;
          }
        }
      }
    }
  }
  if (b == B - 1 && blockIdx.y == 0)
    for (int64_t i = (int64_t)off + n + tid; i < order_cap; i += blockDim.x) order[i] = (int)i;
}

// max-reduction into one word shared by every query wave: read first, most waves then skip the atomic
__device__ __forceinline__ void note_count(int* maxcount, int n) {
  if (n > __hip_atomic_load(maxcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxcount, n);
}

// CAP = in-range neighbours a query can hold in LDS (8 bytes each): 1024 in general, 256 when the caller
// keeps at most 64 columns (the conv-radius searches: more waves per CU)
// Round 5: the transposed relation out of the search itself (REV; csrc/revlist.hip has the stand-alone form). The lanes
// that write row i of `out` hold every kept pair (i, id) in registers: each takes its slot in rev[id] with the same
// INTEGER atomic rev_fill_kernel uses (the count does not depend on the order, the slot does -- arrival order, which is
// all the default mode asks for), so the second pass over the finished matrix (as long as the search that produced it:
// 47.6 against 42.7 us at level 0) disappears. Tails and counters are finished for all lists of a pyramid by ONE
// launch afterwards (mvk_reverse_finish_many).
struct RevOut {
  int32_t* rev;      // [Ns_cap, Hr]
  int32_t* count;    // [Ns_cap], zero on entry
  int32_t* status;   // [2]: longest row, overflow
  int Hr;
};

template <bool FILL, int CAP, bool REV = false>
__global__ __launch_bounds__(64) void nb_query_kernel(const float* __restrict__ q, NbWs W, int B,
                                                      float radius, int64_t Ns, int* __restrict__ out,
                                                      int width, RevOut R) {
  __shared__ float ld2[FILL ? CAP : 1];
  __shared__ int lidx[FILL ? CAP : 1];
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  int b = 0, soff = 0;
  bool padding;
  if (W.q_lens) {          // offsets from the device lengths (wave-uniform scalar loop over a handful of clouds)
    int qend = max(W.q_lens[0], 0);
    while (b + 1 < B && i >= qend) {
      soff += max(W.s_lens[b], 0);
      ++b;
      qend += max(W.q_lens[b], 0);
    }
    padding = i >= qend;
  } else {
    padding = i >= W.qoffs[B];
    while (b + 1 < B && i >= W.qoffs[b + 1]) ++b;
    soff = W.soffs[b];
  }
  if (padding) {   // fixed-capacity launch (device-lens variant): a padding row has no neighbours
    if (FILL)
      for (int c = lane; c < width; c += 64) out[i * width + c] = (int)Ns;
    return;
  }
  const CloudGrid G = W.grids[b];
  const int* cstart = W.cell_start + G.cell_base;
  const float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
  const float r2 = radius * radius;  // neighbors.cpp:226
  // query cell (may lie outside the support grid)
  const float fx = floorf((qx - G.gmin[0]) / G.cell), fy = floorf((qy - G.gmin[1]) / G.cell),
              fz = floorf((qz - G.gmin[2]) / G.cell);
  int rs = 0, re = 0;  // record range of row `lane` (9 rows: dy,dz in -1..1)
  if (lane < 9 && fabsf(fx) < 4.0e6f && fabsf(fy) < 4.0e6f && fabsf(fz) < 4.0e6f) {
    const int cx = (int)fx, cy = (int)fy + (lane % 3) - 1, cz = (int)fz + (lane / 3) - 1;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.dims[0] - 1);
    if (cy >= 0 && cy < G.dims[1] && cz >= 0 && cz < G.dims[2] && x0 <= x1) {
      const int row = (cz * G.dims[1] + cy) * G.dims[0];
      rs = cstart[row + x0];
      re = cstart[row + x1 + 1];
    }
  }
  // inclusive scan of row lengths over lanes 0..8
  int len = re - rs, pre = len;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    int y = __shfl_up(pre, o);
    if (lane >= o) pre += y;
  }
  const int total = __shfl(pre, 8);
  int n = 0;  // in-range count (wave-uniform)
  for (int c0 = 0; c0 < total; c0 += 64) {
    const int c = c0 + lane;
    bool hit = false;
    float d2 = 0.f;
    int idx = 0;
    // locate the row of candidate c (shuffles stay outside divergent control flow)
    int rec = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int pend = __shfl(pre, r), plen = __shfl(len, r), pstart = __shfl(rs, r);
      if (c < pend && c >= pend - plen) rec = pstart + (c - (pend - plen));
    }
    if (c < total) {
      const float4 p = W.recs[soff + rec];
      const float ddx = qx - p.x, ddy = qy - p.y, ddz = qz - p.z;
      d2 = 0.0f;
      d2 += ddx * ddx;
      d2 += ddy * ddy;
      d2 += ddz * ddz;
      hit = d2 < r2;  // nanoflann.hpp:249-251
      idx = __float_as_int(p.w);
    }
    const unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < CAP) {
        ld2[pos] = d2;
        lidx[pos] = idx;
      }
    }
    n += __popcll(m);
  }
  if (!FILL) {
    if (lane == 0) {
      W.counts[i] = n;
      note_count(W.maxcount, n);
    }
    return;
  }
  if (lane == 0) note_count(W.maxcount, n);
  if (n > CAP) {
    if (lane == 0) atomicExch(W.overflow, 1);
    n = CAP;
  }
  __syncthreads();
  // rank-by-counting sort on (d2, idx)
  for (int e = lane; e < n; e += 64) {
    const float d = ld2[e];
    const int id = lidx[e];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const float du = ld2[u];
      const int iu = lidx[u];
      rank += (du < d) || (du == d && iu < id);
    }
    if (rank < width) {
      out[i * width + rank] = id;
      if (REV) {
        const int slot = atomicAdd(R.count + id, 1);
        if (slot < R.Hr)
          R.rev[(int64_t)id * R.Hr + slot] = (int32_t)i;
        else
          atomicOr(R.status + 1, 1);
      }
    }
  }
  for (int c = n + lane; c < width; c += 64) out[i * width + c] = (int)Ns;  // neighbors.cpp:324
}

// The same search with WV wavefronts per query, for the wide lists of the deformable levels (hundreds of in-range supports
// per query, the whole LDS list of 1024 at the coarsest ones). One wavefront ranks its list by counting in
// n * n / 64 steps: with n = 1000 that is 300 us for ONE query, and a level of 128 queries took as long as one of 2 304
// -- the launch lasted as long as its slowest wavefront. Here the candidates and the ranking are dealt over 64 * WV
// lanes (list positions from an LDS counter, one update per wavefront and round); rows come out identical: the order
// of the list before the ranking is the only thing that differs, and (d2, index) is a total order.
template <int WV>
__global__ __launch_bounds__(64 * WV) void nb_query_wide_kernel(const float* __restrict__ q, NbWs W, int B, float radius,
                                                                int64_t Ns, int* __restrict__ out, int width) {
  constexpr int CAP = LIST_CAP, T = 64 * WV;
  __shared__ float ld2[CAP];
  __shared__ int lidx[CAP];
  __shared__ int ln;
  const int64_t i = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int b = 0, soff = 0;
  bool padding;
  if (W.q_lens) {
    int qend = max(W.q_lens[0], 0);
    while (b + 1 < B && i >= qend) {
      soff += max(W.s_lens[b], 0);
      ++b;
      qend += max(W.q_lens[b], 0);
    }
    padding = i >= qend;
  } else {
    padding = i >= W.qoffs[B];
    while (b + 1 < B && i >= W.qoffs[b + 1]) ++b;
    soff = W.soffs[b];
  }
  if (padding) {
    for (int c = tid; c < width; c += T) out[i * width + c] = (int)Ns;
    return;
  }
  if (tid == 0) ln = 0;
  const CloudGrid G = W.grids[b];
  const int* cstart = W.cell_start + G.cell_base;
  const float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
  const float r2 = radius * radius;
  const float fx = floorf((qx - G.gmin[0]) / G.cell), fy = floorf((qy - G.gmin[1]) / G.cell),
              fz = floorf((qz - G.gmin[2]) / G.cell);
  int rs = 0, re = 0;
  if (lane < 9 && fabsf(fx) < 4.0e6f && fabsf(fy) < 4.0e6f && fabsf(fz) < 4.0e6f) {
    const int cx = (int)fx, cy = (int)fy + (lane % 3) - 1, cz = (int)fz + (lane / 3) - 1;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.dims[0] - 1);
    if (cy >= 0 && cy < G.dims[1] && cz >= 0 && cz < G.dims[2] && x0 <= x1) {
      const int row = (cz * G.dims[1] + cy) * G.dims[0];
      rs = cstart[row + x0];
      re = cstart[row + x1 + 1];
    }
  }
  int len = re - rs, pre = len;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    int y = __shfl_up(pre, o);
    if (lane >= o) pre += y;
  }
  const int total = __shfl(pre, 8);
  __syncthreads();                                   // ln = 0 is visible
  for (int c0 = wv * 64; c0 < total; c0 += T) {      // (wave-uniform trip count: the shuffles below stay convergent)
    const int c = c0 + lane;
    bool hit = false;
    float d2 = 0.f;
    int idx = 0, rec = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int pend = __shfl(pre, r), plen = __shfl(len, r), pstart = __shfl(rs, r);
      if (c < pend && c >= pend - plen) rec = pstart + (c - (pend - plen));
    }
    if (c < total) {
      const float4 p = W.recs[soff + rec];
      const float ddx = qx - p.x, ddy = qy - p.y, ddz = qz - p.z;
      d2 = 0.0f;
      d2 += ddx * ddx;
      d2 += ddy * ddy;
      d2 += ddz * ddz;
      hit = d2 < r2;
      idx = __float_as_int(p.w);
    }
    const unsigned long long m = __ballot(hit);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&ln, __popcll(m));
      base = __shfl(base, 0);
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < CAP) {
        ld2[pos] = d2;
        lidx[pos] = idx;
      }
    }
  }
  __syncthreads();
  int n = ln;
  if (tid == 0) {
    note_count(W.maxcount, n);
    if (n > CAP) atomicExch(W.overflow, 1);
  }
  n = min(n, CAP);
  for (int e = tid; e < n; e += T) {
    const float d = ld2[e];
    const int id = lidx[e];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const float du = ld2[u];
      const int iu = lidx[u];
      rank += (du < d) || (du == d && iu < id);
    }
    if (rank < width) out[i * width + rank] = id;
  }
  for (int c = n + tid; c < width; c += T) out[i * width + c] = (int)Ns;
}

struct Carver {
  char* p;
  template <typename T>
  T* take(int64_t count) {
    uintptr_t a = ((uintptr_t)p + 15) & ~(uintptr_t)15;
    p = (char*)a + sizeof(T) * count;
    return (T*)a;
  }
};

int64_t ws_bytes(int64_t Nq, int64_t Ns, int B) {
  int64_t cells = cell_cap(Ns) + (int64_t)B * (4096 + 1 + 4);
  return (int64_t)B * sizeof(CloudGrid) + 2 * cells * 4 + (Ns + 1) * 16 + 2 * (int64_t)(B + 1) * 4 +
         (Nq + 1) * 4 + 64 + 16 * 12 + 16;
}

}  // namespace

extern "C" int64_t mvk_radius_neighbors_workspace(int64_t Nq, int64_t Ns, int B) { return ws_bytes(Nq, Ns, B); }

namespace {

// 0 = one workgroup per cloud always; otherwise the multi-workgroup build from this many supports (longest cloud, or the
// capacity of the device-lens entry) on
int64_t multi_build_min() {
  static const int64_t v = [] {
    const char* e = getenv("MVK_NB_MULTI_MIN");
    return e ? (int64_t)atoll(e) : (int64_t)4096;
  }();
  return v;
}

// rows of more than 64 columns: eight wavefronts per query (MVK_NB_WIDE_WAVES=1: one, as for the narrow rows)
int wide_waves() {
  static const int v = [] {
    const char* e = getenv("MVK_NB_WIDE_WAVES");
    return e ? atoi(e) : 8;
  }();
  return v;
}

void launch_build(const float* s, const NbWs& W, float radius, int B, int64_t maxn, hipStream_t st) {
  const int64_t mm = multi_build_min();
  if (mm <= 0 || maxn < mm) {
    hipLaunchKernelGGL(nb_build_kernel, dim3(B), dim3(TPB), 0, st, s, W, radius, 0);
    return;
  }
  const dim3 g((unsigned)((maxn + 255) / 256), B);
  hipLaunchKernelGGL(nb_build_kernel, dim3(B), dim3(TPB), 0, st, s, W, radius, 1);
  hipLaunchKernelGGL(nb_hist_kernel, g, dim3(256), 0, st, s, W);
  hipLaunchKernelGGL(nb_scan_kernel, dim3(B), dim3(TPB), 0, st, W);
  hipLaunchKernelGGL(nb_scatter_kernel, g, dim3(256), 0, st, s, W);
}

// Common body. status_dev == nullptr: classic two-phase contract (synchronises, reports through
// width_host). status_dev != nullptr: enqueue only -- the max row count / overflow flag are folded into
// status_dev[0] / status_dev[1] on the device. reuse_grid: the workspace still holds the grid built by
// the previous call for the SAME supports, batch lengths and radius (conv / pool / upsample searches of
// one pyramid level share it), so only the query pass is launched.
int nb_run(const float* q, int64_t Nq, const float* s, int64_t Ns, const int32_t* q_lens_host,
           const int32_t* s_lens_host, int B, float radius, int32_t* out, int width, int* width_host,
           int32_t* status_dev, int reuse_grid, void* workspace, int64_t workspace_bytes, void* stream) {
  MVK_REQUIRE(B >= 1 && Nq >= 0 && Ns >= 0 && Nq < (1ll << 31) && Ns < (1ll << 29), "neighbors: bad sizes");
  MVK_REQUIRE(radius > 0.f, "neighbors: radius must be positive");
  MVK_REQUIRE(workspace && workspace_bytes >= ws_bytes(Nq, Ns, B), "neighbors: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // header = {max count, overflow, query offsets [B+1], support offsets [B+1]}: ONE host-to-device copy
  static thread_local int hdr_h[4 + 2 * 4097];
  MVK_REQUIRE(B <= 4096, "neighbors: more than 4096 clouds in a batch");
  int* qo = hdr_h + 4;
  int* so = hdr_h + 4 + B + 1;
  int64_t tq = 0, ts = 0;
  for (int b = 0; b < B; ++b) {
    MVK_REQUIRE(q_lens_host[b] >= 0 && s_lens_host[b] >= 0, "neighbors: negative batch length");
    qo[b] = (int)tq;
    so[b] = (int)ts;
    tq += q_lens_host[b];
    ts += s_lens_host[b];
  }
  qo[B] = (int)tq;
  so[B] = (int)ts;
  hdr_h[0] = hdr_h[1] = hdr_h[2] = hdr_h[3] = 0;
  MVK_REQUIRE(tq == Nq && ts == Ns, "neighbors: batch lengths do not sum to the point counts");

  Carver cv{(char*)workspace};
  NbWs W{};
  int64_t cells = cell_cap(Ns) + (int64_t)B * (4096 + 1 + 4);
  W.grids = cv.take<CloudGrid>(B);              // grid part first: its layout depends on (Ns, B) only
  W.cell_start = cv.take<int>(cells);
  W.cell_fill = cv.take<int>(cells);
  W.recs = cv.take<float4>(Ns + 1);
  int* hdr = cv.take<int>(4 + 2 * (B + 1));
  W.maxcount = hdr;
  W.overflow = hdr + 1;
  W.qoffs = hdr + 4;
  W.soffs = hdr + 4 + B + 1;
  W.counts = cv.take<int>(Nq + 1);
  MVK_REQUIRE(cv.p <= (char*)workspace + workspace_bytes, "neighbors: workspace carve overflow");
  if (status_dev) {
    W.maxcount = status_dev;
    W.overflow = status_dev + 1;
  }

  MVK_CHECK_HIP(hipMemcpyAsync(hdr, hdr_h, sizeof(int) * (4 + 2 * (B + 1)), hipMemcpyHostToDevice, st));
  if (!reuse_grid) {
    int64_t maxn = 0;
    for (int b = 0; b < B; ++b) maxn = s_lens_host[b] > maxn ? s_lens_host[b] : maxn;
    launch_build(s, W, radius, B, maxn, st);
  }
  if (out == nullptr) {
    MVK_REQUIRE(width_host != nullptr && status_dev == nullptr, "neighbors: phase 1 needs width_host");
    if (Nq > 0)
      hipLaunchKernelGGL((nb_query_kernel<false, LIST_CAP>), dim3((unsigned)Nq), dim3(64), 0, st, q, W, B, radius, Ns,
                         (int*)nullptr, 0, RevOut{});
    MVK_CHECK_HIP(hipGetLastError());
    MVK_CHECK_HIP(hipMemcpyAsync(width_host, W.maxcount, sizeof(int), hipMemcpyDeviceToHost, st));
    MVK_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
  }
  MVK_REQUIRE(width >= 0, "neighbors: negative width");
  if (Nq > 0 && width > 0) {
    // enqueue-only mode with a column limit: a row holding more than 4x the calibrated limit is reported
    // through the overflow flag like a row beyond LIST_CAP
    if (status_dev && width <= 64)
      hipLaunchKernelGGL((nb_query_kernel<true, 256>), dim3((unsigned)Nq), dim3(64), 0, st, q, W, B, radius, Ns, out, width, RevOut{});
    else if (wide_waves() > 1 && width > 64)
      hipLaunchKernelGGL((nb_query_wide_kernel<8>), dim3((unsigned)Nq), dim3(512), 0, st, q, W, B, radius, Ns, out, width);
    else
      hipLaunchKernelGGL((nb_query_kernel<true, LIST_CAP>), dim3((unsigned)Nq), dim3(64), 0, st, q, W, B, radius, Ns, out, width, RevOut{});
  }
  MVK_CHECK_HIP(hipGetLastError());
  if (status_dev) return 0;
  int res[2] = {0, 0};
  MVK_CHECK_HIP(hipMemcpyAsync(res, hdr, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
  MVK_CHECK_HIP(hipStreamSynchronize(st));
  if (width_host) *width_host = res[0];
  MVK_REQUIRE(res[1] == 0, "neighbors: a query has more than %d in-range supports (LDS list capacity)", LIST_CAP);
  return 0;
}

}  // namespace

extern "C" int mvk_radius_neighbors_batch(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                          const int32_t* q_lens_host, const int32_t* s_lens_host,
                                          int B, float radius, int32_t* out, int width,
                                          int* width_host, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
  return nb_run(q, Nq, s, Ns, q_lens_host, s_lens_host, B, radius, out, width, width_host, nullptr, 0, workspace,
                workspace_bytes, stream);
}

extern "C" int mvk_radius_neighbors_enqueue(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                            const int32_t* q_lens_host, const int32_t* s_lens_host,
                                            int B, float radius, int32_t* out, int width,
                                            int32_t* status_dev, int reuse_grid, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
  MVK_REQUIRE(out != nullptr && status_dev != nullptr, "neighbors: enqueue needs an output matrix and a status word");
  return nb_run(q, Nq, s, Ns, q_lens_host, s_lens_host, B, radius, out, width, nullptr, status_dev, reuse_grid,
                workspace, workspace_bytes, stream);
}

namespace {
int nb_dev_run(const float* q, int64_t Nq_cap, const float* s, int64_t Ns_cap, const int32_t* q_lens_dev,
               const int32_t* s_lens_dev, int B, float radius, int32_t* out, int width, int32_t shadow, int32_t* status_dev,
               int reuse_grid, void* workspace, int64_t workspace_bytes, const RevOut* rev, void* stream);
}

extern "C" int mvk_radius_neighbors_dev(const float* q, int64_t Nq_cap, const float* s, int64_t Ns_cap,
                                        const int32_t* q_lens_dev, const int32_t* s_lens_dev, int B,
                                        float radius, int32_t* out, int width, int32_t shadow,
                                        int32_t* status_dev, int reuse_grid, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
  return nb_dev_run(q, Nq_cap, s, Ns_cap, q_lens_dev, s_lens_dev, B, radius, out, width, shadow, status_dev, reuse_grid,
                    workspace, workspace_bytes, nullptr, stream);
}

// mvk_radius_neighbors_dev that also fills the TRANSPOSED relation while it writes the rows (width <= 64): rev
// [Ns_cap, rev_width] int32 receives, in row j, the query rows whose list holds support j, in order of arrival;
// rev_counts [Ns_cap] int32 (ZERO on entry) end up holding the row lengths; rev_status [2] as in mvk_reverse_neighbors.
// The tails of the rows are NOT written and the counters NOT reset here: mvk_reverse_finish_many does both for all the
// lists of a pyramid in one launch.
extern "C" int mvk_radius_neighbors_dev_rev(const float* q, int64_t Nq_cap, const float* s, int64_t Ns_cap,
                                            const int32_t* q_lens_dev, const int32_t* s_lens_dev, int B, float radius,
                                            int32_t* out, int width, int32_t shadow, int32_t* status_dev, int reuse_grid,
                                            void* workspace, int64_t workspace_bytes, int32_t* rev, int rev_width,
                                            int32_t* rev_counts, int32_t* rev_status, void* stream) {
  MVK_REQUIRE(rev && rev_counts && rev_status && rev_width >= 1 && rev_width <= MVK_REV_MAX_WIDTH && width <= 64,
              "neighbors: the fused reverse list needs rev, counters, a status word and at most 64 columns");
  const RevOut R{rev, rev_counts, rev_status, rev_width};
  return nb_dev_run(q, Nq_cap, s, Ns_cap, q_lens_dev, s_lens_dev, B, radius, out, width, shadow, status_dev, reuse_grid,
                    workspace, workspace_bytes, &R, stream);
}

namespace {
int nb_dev_run(const float* q, int64_t Nq_cap, const float* s, int64_t Ns_cap, const int32_t* q_lens_dev,
               const int32_t* s_lens_dev, int B, float radius, int32_t* out, int width, int32_t shadow, int32_t* status_dev,
               int reuse_grid, void* workspace, int64_t workspace_bytes, const RevOut* rev, void* stream) {
  MVK_REQUIRE(B >= 1 && B <= 4096 && Nq_cap >= 1 && Ns_cap >= 1 && Nq_cap < (1ll << 31) && Ns_cap < (1ll << 29),
              "neighbors: bad sizes");
  MVK_REQUIRE(radius > 0.f && width >= 1 && out && status_dev, "neighbors: bad arguments");
  MVK_REQUIRE(workspace && workspace_bytes >= ws_bytes(Nq_cap, Ns_cap, B), "neighbors: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Carver cv{(char*)workspace};
  NbWs W{};
  int64_t cells = cell_cap(Ns_cap) + (int64_t)B * (4096 + 1 + 4);
  W.grids = cv.take<CloudGrid>(B);
  W.cell_start = cv.take<int>(cells);
  W.cell_fill = cv.take<int>(cells);
  W.recs = cv.take<float4>(Ns_cap + 1);
  int* hdr = cv.take<int>(4 + 2 * (B + 1));
  W.qoffs = hdr + 4;
  W.soffs = hdr + 4 + B + 1;
  W.counts = cv.take<int>(Nq_cap + 1);
  MVK_REQUIRE(cv.p <= (char*)workspace + workspace_bytes, "neighbors: workspace carve overflow");
  W.maxcount = status_dev;
  W.overflow = status_dev + 1;
  W.q_lens = q_lens_dev;       // the kernels derive the offsets themselves (was: a one-thread launch per search)
  W.s_lens = s_lens_dev;
  if (!reuse_grid) launch_build(s, W, radius, B, Ns_cap, st);
  if (width <= 64 && rev)
    hipLaunchKernelGGL((nb_query_kernel<true, 256, true>), dim3((unsigned)Nq_cap), dim3(64), 0, st, q, W, B, radius,
                       (int64_t)shadow, out, width, *rev);
  else if (width <= 64)
    hipLaunchKernelGGL((nb_query_kernel<true, 256>), dim3((unsigned)Nq_cap), dim3(64), 0, st, q, W, B, radius,
                       (int64_t)shadow, out, width, RevOut{});
  else if (wide_waves() > 1)
    hipLaunchKernelGGL((nb_query_wide_kernel<8>), dim3((unsigned)Nq_cap), dim3(512), 0, st, q, W, B, radius,
                       (int64_t)shadow, out, width);
  else
    hipLaunchKernelGGL((nb_query_kernel<true, LIST_CAP>), dim3((unsigned)Nq_cap), dim3(64), 0, st, q, W, B, radius,
                       (int64_t)shadow, out, width, RevOut{});
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// The supports of the grid the workspace holds (built by the last search on it: same Ns, B and lengths), as a work
// list for mvk_kpconv_gather_fwd_ordered: order_out[0 .. total) = the stacked rows sorted by cloud, grid cell
// (x fastest) and row; order_out[total .. order_cap) = identity. s_lens_dev: the device lengths of a
// mvk_radius_neighbors_dev search, NULL after the host-length entry points (their offsets are in the workspace).
extern "C" int mvk_neighbors_cell_order(int64_t Ns, int B, const int32_t* s_lens_dev, int32_t* order_out,
                                        int64_t order_cap, void* workspace, int64_t workspace_bytes, void* stream) {
  MVK_REQUIRE(B >= 1 && B <= 4096 && Ns >= 0 && Ns < (1ll << 29) && order_out && order_cap >= Ns, "cell order: bad arguments");
  MVK_REQUIRE(workspace && workspace_bytes >= ws_bytes(0, Ns, B), "cell order: workspace too small");
  Carver cv{(char*)workspace};
  NbWs W{};
  int64_t cells = cell_cap(Ns) + (int64_t)B * (4096 + 1 + 4);
  W.grids = cv.take<CloudGrid>(B);
  W.cell_start = cv.take<int>(cells);
  W.cell_fill = cv.take<int>(cells);
  W.recs = cv.take<float4>(Ns + 1);
  int* hdr = cv.take<int>(4 + 2 * (B + 1));
  W.soffs = hdr + 4 + B + 1;
  W.s_lens = s_lens_dev;
  // a thread's cells are chains of dependent loads (cell range -> records -> stores): one workgroup per cloud took
  // 150 us on the level-0 grid of a 19 464-point sphere (15 cells per thread, one after the other)
  // (256-thread workgroups: the cells of a 19 464-point cloud are 10 k, i.e. ten compute units with 1024 threads each)
  const int64_t want = (cell_cap(Ns) / B + 255) / 256;
  const unsigned chunks = (unsigned)(want < 256 ? want : 256);
  hipLaunchKernelGGL(nb_cell_order_kernel, dim3(B, chunks ? chunks : 1), dim3(256), 0, (hipStream_t)stream, W, B, order_out,
                     order_cap);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
