// Multi-view 2D -> 3D fusion inputs on gfx950:
//   * depth back-projection   (reference KPConv-PyTorch/datasets/ScanNet_sphere_color.py:66-72, :409-417)
//   * exact k-NN of sphere points among the unprojected pixels, float64, brute force with the keys
//     tiled through LDS (reference: scikit-learn NearestNeighbors(ball_tree), :448-451)
//   * group_points gather / scatter-add (reference mvpnet/ops/cuda/group_points_kernel.cu:25-145)
// Compiled with -ffp-contract=off: the float64 products and sums are evaluated exactly in the
// written order, like the NumPy / scikit-learn code they replace.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// xyz_cam = (Kinv . [u, v, 1]) * depth ;  valid = z_cam > 0 ;  xyz_world = xyz_cam . R^T + t
// ---------------------------------------------------------------------------------------------
struct Cam {
  double kinv[9];
};

__global__ void unproject_kernel(const uint16_t* __restrict__ depth, int nv, int h, int w, Cam cam,
                                 const float* __restrict__ poses, double* __restrict__ xyz,
                                 uint8_t* __restrict__ valid) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t hw = (int64_t)h * w;
  if (t >= nv * hw) return;
  const int view = (int)(t / hw);
  const int pix = (int)(t % hw);
  const double u = (double)(pix % w), v = (double)(pix / w);
  // depth: uint16 mm -> float32 / 1000.f (ScanNet_sphere_color.py:410), then promoted to float64
  const double d = (double)((float)depth[t] / 1000.f);
  double c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) c[r] = ((cam.kinv[r * 3] * u + cam.kinv[r * 3 + 1] * v) + cam.kinv[r * 3 + 2]) * d;
  valid[t] = c[2] > 0.0;
  const float* P = poses + view * 16;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    xyz[t * 3 + r] = ((c[0] * (double)P[r * 4] + c[1] * (double)P[r * 4 + 1]) + c[2] * (double)P[r * 4 + 2]) +
                     (double)P[r * 4 + 3];
}

// ---------------------------------------------------------------------------------------------
// brute-force k-NN, float64. One lane = one query; a 256-lane workgroup stages 256 keys at a time
// in LDS (coalesced loads, broadcast reads); top-k kept sorted in registers (k <= 8).
// Keys are split into `nsplit` contiguous ranges (grid.y) so that small query sets still fill the
// chip; a merge kernel combines the per-split candidates.
// ---------------------------------------------------------------------------------------------
constexpr int KNN_T = 256;
constexpr int KMAXNN = 8;

template <int K>
__device__ __forceinline__ void topk_insert(double (&bd)[K], int64_t (&bi)[K], double d2, int64_t j) {
  if (d2 < bd[K - 1] || (d2 == bd[K - 1] && j < bi[K - 1])) {
    bd[K - 1] = d2;
    bi[K - 1] = j;
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
      if (bd[p] < bd[p - 1] || (bd[p] == bd[p - 1] && bi[p] < bi[p - 1])) {
        double td = bd[p]; bd[p] = bd[p - 1]; bd[p - 1] = td;
        int64_t ti = bi[p]; bi[p] = bi[p - 1]; bi[p - 1] = ti;
      }
    }
  }
}

// QPT queries per lane: every LDS key read (3 broadcast ds_read_b64) is reused QPT times, which moves the
// loop from LDS-issue bound (1 query per lane: ~6 LDS cycles per 28 fp64 VALU cycles per wave, four waves
// sharing one LDS) to fp64-VALU bound.
constexpr int QPT = 4;

template <int K>
__global__ __launch_bounds__(KNN_T) void knn_partial_kernel(const float* __restrict__ q, int64_t nq,
                                                            const double* __restrict__ keys,
                                                            const uint8_t* __restrict__ kvalid,
                                                            int64_t nk, int64_t per_split,
                                                            double* __restrict__ pd,
                                                            int64_t* __restrict__ pi) {
  __shared__ double kx[KNN_T], ky[KNN_T], kz[KNN_T];
  __shared__ int kok[KNN_T];
  const int64_t i0 = ((int64_t)blockIdx.x * KNN_T + threadIdx.x) * QPT;
  const int64_t kbeg = (int64_t)blockIdx.y * per_split, kend = min(nk, kbeg + per_split);
  double qx[QPT], qy[QPT], qz[QPT];
  double bd[QPT][K];
  int64_t bi[QPT][K];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const int64_t i = i0 + u;
    qx[u] = i < nq ? (double)q[i * 3] : 0.0;
    qy[u] = i < nq ? (double)q[i * 3 + 1] : 0.0;
    qz[u] = i < nq ? (double)q[i * 3 + 2] : 0.0;
#pragma unroll
    for (int c = 0; c < K; ++c) {
      bd[u][c] = INFINITY;
      bi[u][c] = INT64_MAX;
    }
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += KNN_T) {
    const int64_t j = k0 + threadIdx.x;
    if (j < kend) {
      kx[threadIdx.x] = keys[j * 3];
      ky[threadIdx.x] = keys[j * 3 + 1];
      kz[threadIdx.x] = keys[j * 3 + 2];
      kok[threadIdx.x] = kvalid ? (int)kvalid[j] : 1;
    } else {
      kok[threadIdx.x] = 0;
    }
    __syncthreads();
    const int cnt = (int)min((int64_t)KNN_T, kend - k0);
    for (int t = 0; t < cnt; ++t) {
      if (!kok[t]) continue;  // uniform across the workgroup
      const double x = kx[t], y = ky[t], z = kz[t];
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const double dx = qx[u] - x, dy = qy[u] - y, dz = qz[u] - z;
        double d2 = 0.0;
        d2 += dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        topk_insert<K>(bd[u], bi[u], d2, k0 + t);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const int64_t i = i0 + u;
    if (i < nq) {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        pd[((int64_t)blockIdx.y * nq + i) * K + c] = bd[u][c];
        pi[((int64_t)blockIdx.y * nq + i) * K + c] = bi[u][c];
      }
    }
  }
}

template <int K>
__global__ void knn_merge_kernel(const double* __restrict__ pd, const int64_t* __restrict__ pi, int nsplit,
                                 int64_t nq, int kout, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  double bd[K];
  int64_t bi[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    bd[c] = INFINITY;
    bi[c] = INT64_MAX;
  }
  for (int s = 0; s < nsplit; ++s)
#pragma unroll
    for (int c = 0; c < K; ++c) {
      const double d = pd[((int64_t)s * nq + i) * K + c];
      const int64_t j = pi[((int64_t)s * nq + i) * K + c];
      if (j != INT64_MAX) topk_insert<K>(bd, bi, d, j);
    }
#pragma unroll
  for (int c = 0; c < K; ++c)
    if (c < kout) out[i * kout + c] = bi[c] == INT64_MAX ? -1 : bi[c];
}

// ---------------------------------------------------------------------------------------------
// Pruned exact k-NN (same results as the brute-force kernels above, bit for bit):
//   1. queries and valid keys are binned into a 32^3 Morton-ordered torus grid (cell 0.1 m; the
//      binning only decides the ORDER things are visited in, never the result), counting sort;
//   2. every run of 64 sorted keys is a tile with an axis-aligned box (float, rounded outwards);
//   3. a workgroup owns 64 consecutive sorted queries (a compact patch). Per query an upper bound of
//      the k-th distance follows from the boxes alone (ub = min over tiles holding >= k keys of the
//      farthest box corner); a tile is visited only if some query of the wave has
//      min-dist(box) <= min(ub, its current k-th best). Every key with d2 <= the true k-th distance
//      sits in such a tile, ties included, so the (d2, index)-ordered top-k is the brute-force one.
//      The waves of the workgroup share out the candidate tiles and merge their results in LDS.
// Distances are the same float64 expression as above.
// ---------------------------------------------------------------------------------------------
constexpr int PG = 32, PNC = PG * PG * PG;        // grid cells
constexpr float PCELL_INV = 10.f;                  // 1 / 0.1 m
constexpr int PT = 64;                             // keys per tile
constexpr int PCH = 1024;                          // tiles considered at a time (candidate list <= 36 KB of LDS)

__device__ __forceinline__ unsigned spread3(unsigned v) {  // 5 bits -> every third bit
  v &= 31u;
  v = (v | (v << 8)) & 0x0000100Fu;
  v = (v | (v << 4)) & 0x000010C3u;
  v = (v | (v << 2)) & 0x00001249u;
  return v;
}
__device__ __forceinline__ int morton_cell(float x, float y, float z) {
  const int ix = (int)floorf(x * PCELL_INV), iy = (int)floorf(y * PCELL_INV), iz = (int)floorf(z * PCELL_INV);
  return (int)(spread3((unsigned)ix) | (spread3((unsigned)iy) << 1) | (spread3((unsigned)iz) << 2));
}

struct PrunedWs {
  int* cnt;        // [2][PNC]   (0 = queries, 1 = keys), zeroed by the host
  int* start;      // [2][PNC]
  int* cell;       // [nq + nk]
  int* rank;       // [nq + nk]
  int* qorder;     // [nq]
  double* skey;    // [nk][3] sorted valid keys
  int* sidx;       // [nk]    their original index
  float* box;      // [ntiles_max][8]  lo xyz, hi xyz, count, pad
  int* nvalid;     // [1]
};

// One counter update per wavefront and distinct cell, not per element: neighbouring pixels unproject into the same 10 cm
// cell (a cell close to a camera holds thousands of keys), and returning atomics on one address retire one after the other
// -- with an atomic per key the hottest cell set the kernel's time (177 us for 96 000 keys of five views, 25 us for three
// views). The lanes of a wavefront are grouped by cell with ballots first (no memory traffic), then every group's first
// lane adds the group's size once and the others take their rank from its return value.
__global__ __launch_bounds__(256) void pk_count_kernel(const float* __restrict__ q, int64_t nq, const double* __restrict__ keys,
                                                       const uint8_t* __restrict__ kvalid, int64_t nk, PrunedWs ws) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool valid = e < nq + nk;
  int key = -1;                                  // set * PNC + cell
  if (valid) {
    if (e < nq) {
      key = morton_cell(q[e * 3], q[e * 3 + 1], q[e * 3 + 2]);
    } else {
      const int64_t j = e - nq;
      if (kvalid && !kvalid[j]) {
        ws.cell[e] = -1;
        valid = false;
      } else {
        key = PNC + morton_cell((float)keys[j * 3], (float)keys[j * 3 + 1], (float)keys[j * 3 + 2]);
      }
    }
  }
  int leader = lane, before = 0, group = 1;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const int k0 = __shfl(key, first);
    const bool mine = valid && key == k0;
    const unsigned long long same = __ballot(mine);
    if (mine) {
      leader = first;
      before = __popcll(same & ((1ull << lane) - 1ull));
      group = __popcll(same);
    }
    todo &= ~same;
  }
  int base = 0;
  if (valid && leader == lane) base = atomicAdd(ws.cnt + key, group);
  base = __shfl(base, leader);
  if (valid) {
    ws.cell[e] = key >= PNC ? key - PNC : key;
    ws.rank[e] = base + before;
  }
}

// exclusive scan of the PNC cell counts of one set (blockIdx.x), 1024 threads x 32 cells
__global__ __launch_bounds__(1024) void pk_scan_kernel(PrunedWs ws) {
  __shared__ int part[1024];
  const int set = blockIdx.x, tid = threadIdx.x;
  const int* cnt = ws.cnt + set * PNC;
  int* start = ws.start + set * PNC;
  constexpr int PER = PNC / 1024;
  int loc[PER], s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    loc[i] = s;
    s += cnt[tid * PER + i];
  }
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const int base = part[tid] - s;
#pragma unroll
  for (int i = 0; i < PER; ++i) start[tid * PER + i] = base + loc[i];
  if (set == 1 && tid == 1023) *ws.nvalid = part[1023];
}

__global__ void pk_scatter_kernel(int64_t nq, const double* __restrict__ keys, int64_t nk, PrunedWs ws) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nq + nk) return;
  const int c = ws.cell[e];
  if (c < 0) return;
  if (e < nq) {
    ws.qorder[ws.start[c] + ws.rank[e]] = (int)e;
  } else {
    const int64_t j = e - nq;
    const int p = ws.start[PNC + c] + ws.rank[e];
    ws.skey[(int64_t)p * 3] = keys[j * 3];
    ws.skey[(int64_t)p * 3 + 1] = keys[j * 3 + 1];
    ws.skey[(int64_t)p * 3 + 2] = keys[j * 3 + 2];
    ws.sidx[p] = (int)j;
  }
}

// one wave per tile: outward-rounded float box of its keys
__global__ __launch_bounds__(64) void pk_box_kernel(PrunedWs ws) {
  const int nv = *ws.nvalid;
  const int t = blockIdx.x, p = t * PT + threadIdx.x;
  float* b = ws.box + (int64_t)t * 8;
  if (t * PT >= nv) {
    if (threadIdx.x == 0) b[6] = 0.f;
    return;
  }
  const bool ok = p < nv;
  float lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double v = ok ? ws.skey[(int64_t)p * 3 + a] : 0.0;
    lo[a] = ok ? __double2float_rd(v) : INFINITY;
    hi[a] = ok ? __double2float_ru(v) : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  }
  if (threadIdx.x == 0) {
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2];
    b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
    b[6] = (float)min(PT, nv - t * PT);
  }
}

template <int K>
__device__ __forceinline__ void topk_insert32(double (&bd)[K], int (&bi)[K], double d2, int id) {
  if (d2 < bd[K - 1] || (d2 == bd[K - 1] && id < bi[K - 1])) {
    bd[K - 1] = d2;
    bi[K - 1] = id;
#pragma unroll
    for (int c = K - 1; c > 0; --c) {
      if (bd[c] < bd[c - 1] || (bd[c] == bd[c - 1] && bi[c] < bi[c - 1])) {
        const double td = bd[c]; bd[c] = bd[c - 1]; bd[c - 1] = td;
        const int ti = bi[c]; bi[c] = bi[c - 1]; bi[c - 1] = ti;
      }
    }
  }
}

template <int K, int PW>   // PW waves per workgroup share one set of 64 queries
__global__ __launch_bounds__(64 * PW) void knn_pruned_kernel(const float* __restrict__ q, int64_t nq, PrunedWs ws,
                                                             int ntiles_max, int kout,
                                                             int64_t* __restrict__ out) {
  __shared__ double sk[PW][3][PT];
  __shared__ int si[PW][PT];
  __shared__ float sub[PW][64];
  // candidate list during the sweeps, merge buffers afterwards (same LDS bytes)
  constexpr int CAND_BYTES = PCH * 36, MERGE_BYTES = PW * 64 * K * 12;
  __shared__ __attribute__((aligned(16))) char un[CAND_BYTES > MERGE_BYTES ? CAND_BYTES : MERGE_BYTES];
  float4 (*cbox)[2] = reinterpret_cast<float4 (*)[2]>(un);          // [PCH][2] boxes of the candidate tiles
  int* cand = reinterpret_cast<int*>(un + PCH * 32);                 // [PCH] their tile ids
  double (*md)[64][K] = reinterpret_cast<double (*)[64][K]>(un);      // [PW][64][K]
  int (*mi)[64][K] = reinterpret_cast<int (*)[64][K]>(un + PW * 64 * K * 8);
  __shared__ int ncand;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t slot = (int64_t)blockIdx.x * 64 + lane;
  const bool live = slot < nq;
  const int qi = live ? ws.qorder[slot] : 0;
  const float fx = live ? q[(int64_t)qi * 3] : 0.f, fy = live ? q[(int64_t)qi * 3 + 1] : 0.f,
              fz = live ? q[(int64_t)qi * 3 + 2] : 0.f;
  const double qx = fx, qy = fy, qz = fz;
  const int nv = *ws.nvalid;
  const int ntiles = min(ntiles_max, (nv + PT - 1) / PT);

  // box of the workgroup's 64 queries (every wave holds the same queries)
  float ql[3] = {live ? fx : INFINITY, live ? fy : INFINITY, live ? fz : INFINITY};
  float qh[3] = {live ? fx : -INFINITY, live ? fy : -INFINITY, live ? fz : -INFINITY};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ql[a] = fminf(ql[a], __shfl_xor(ql[a], off));
      qh[a] = fmaxf(qh[a], __shfl_xor(qh[a], off));
    }

  // A. workgroup bound: the farthest pair (query box, tile box) of the best tile holding >= K keys
  float ubw = INFINITY;
  for (int t = tid; t < ntiles; t += 64 * PW) {
    const float4 b0 = reinterpret_cast<const float4*>(ws.box)[t * 2], b1 = reinterpret_cast<const float4*>(ws.box)[t * 2 + 1];
    if (b1.z >= (float)K) {
      const float dx = fmaxf(b0.w - ql[0], qh[0] - b0.x), dy = fmaxf(b1.x - ql[1], qh[1] - b0.y),
                  dz = fmaxf(b1.y - ql[2], qh[2] - b0.z);
      ubw = fminf(ubw, (dx * dx + dy * dy + dz * dz) * 1.0001f + 1e-30f);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ubw = fminf(ubw, __shfl_xor(ubw, off));
  if (lane == 0) sub[wv][0] = ubw;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < PW; ++w) ubw = fminf(ubw, sub[w][0]);
  __syncthreads();

  double bd[K];
  int bi[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    bd[c] = INFINITY;
    bi[c] = INT_MAX;
  }
  float ub = INFINITY;
  // Sweep 0 tightens the per-query bound over the candidate tiles, sweep 1 scans them. Tiles are taken
  // PCH at a time so that the candidate list always fits in LDS; with one chunk (the usual case) the
  // list of sweep 0 is reused.
  const bool one_chunk = ntiles <= PCH;
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int c0 = 0; c0 < ntiles; c0 += PCH) {
      if (sweep == 0 || !one_chunk) {
        // B. candidates of this chunk: tile boxes within ubw of the query box
        __syncthreads();
        if (tid == 0) ncand = 0;
        __syncthreads();
        for (int t = c0 + tid; t < min(ntiles, c0 + PCH); t += 64 * PW) {
          const float4 b0 = reinterpret_cast<const float4*>(ws.box)[t * 2], b1 = reinterpret_cast<const float4*>(ws.box)[t * 2 + 1];
          const float dx = fmaxf(fmaxf(b0.x - qh[0], ql[0] - b0.w), 0.f);
          const float dy = fmaxf(fmaxf(b0.y - qh[1], ql[1] - b1.x), 0.f);
          const float dz = fmaxf(fmaxf(b0.z - qh[2], ql[2] - b1.y), 0.f);
          if ((dx * dx + dy * dy + dz * dz) * 0.9999f <= ubw) {
            const int s = atomicAdd(&ncand, 1);
            cand[s] = t;
            cbox[s][0] = b0;
            cbox[s][1] = b1;
          }
        }
        __syncthreads();
      }
      const int nc = ncand;
      if (sweep == 0) {
        for (int c = wv; c < nc; c += PW) {
          const float4 b0 = cbox[c][0], b1 = cbox[c][1];
          if (b1.z >= (float)K) {
            const float dx = fmaxf(fabsf(fx - b0.x), fabsf(b0.w - fx));
            const float dy = fmaxf(fabsf(fy - b0.y), fabsf(b1.x - fy));
            const float dz = fmaxf(fabsf(fz - b0.z), fabsf(b1.y - fz));
            ub = fminf(ub, (dx * dx + dy * dy + dz * dz) * 1.0001f + 1e-30f);
          }
        }
        continue;
      }
      for (int c = wv; c < nc; c += PW) {
        const float4 b0 = cbox[c][0], b1 = cbox[c][1];
        const float dx = fmaxf(fmaxf(b0.x - fx, fx - b0.w), 0.f);
        const float dy = fmaxf(fmaxf(b0.y - fy, fy - b1.x), 0.f);
        const float dz = fmaxf(fmaxf(b0.z - fz, fz - b1.y), 0.f);
        const float mind = (dx * dx + dy * dy + dz * dz) * 0.9999f;
        const float thr = fminf(ub, (float)bd[K - 1] * 1.0001f);      // (float)inf stays inf
        if (!__any(mind <= thr)) continue;
        const int cntk = (int)b1.z;
        const int p = cand[c] * PT + lane;
        const bool has = lane < cntk;                                 // padding keys can never be inserted
        sk[wv][0][lane] = has ? ws.skey[(int64_t)p * 3] : INFINITY;
        sk[wv][1][lane] = has ? ws.skey[(int64_t)p * 3 + 1] : INFINITY;
        sk[wv][2][lane] = has ? ws.skey[(int64_t)p * 3 + 2] : INFINITY;
        si[wv][lane] = has ? ws.sidx[p] : INT_MAX;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int U = 8;                                          // independent distance chains in flight
        for (int j0 = 0; j0 < PT; j0 += U) {
          double d2[U];
          int id[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double ddx = qx - sk[wv][0][j0 + u], ddy = qy - sk[wv][1][j0 + u], ddz = qz - sk[wv][2][j0 + u];
            double s = 0.0;
            s += ddx * ddx;
            s += ddy * ddy;
            s += ddz * ddz;
            d2[u] = s;
            id[u] = si[wv][j0 + u];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) topk_insert32<K>(bd, bi, d2[u], id[u]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (sweep == 0) {   // combine the per-wave partial bounds
      sub[wv][lane] = ub;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < PW; ++w) ub = fminf(ub, sub[w][lane]);
      if (!live) ub = -1.f;   // idle lanes never ask for a tile
    }
  }
  __syncthreads();   // the candidate list is dead: its bytes become the merge buffers
#pragma unroll
  for (int c = 0; c < K; ++c) {
    md[wv][lane][c] = bd[c];
    mi[wv][lane][c] = bi[c];
  }
  __syncthreads();
  if (wv == 0 && live) {
    for (int w = 1; w < PW; ++w)
#pragma unroll
      for (int c = 0; c < K; ++c) {
        const int id = mi[w][lane][c];
        if (id != INT_MAX) topk_insert32<K>(bd, bi, md[w][lane][c], id);
      }
#pragma unroll
    for (int c = 0; c < K; ++c)
      if (c < kout) out[(int64_t)qi * kout + c] = bi[c] == INT_MAX ? -1 : (int64_t)bi[c];
  }
}

// ---------------------------------------------------------------------------------------------
// group_points: out[b,c,n,k] = in[b,c,idx[b,n,k]]; one lane per output element, (n,k) fastest so
// index reads and output writes are coalesced; the gather itself is element granular by nature
// (channel-major feature maps), served by L2 / Infinity Cache.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void group_points_fwd_kernel(const T* __restrict__ in, const int64_t* __restrict__ idx, int B,
                                        int C, int64_t N1, int64_t NK, T* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)B * C * NK) return;
  const int64_t e = t % NK;
  const int64_t bc = t / NK;
  const int64_t b = bc / C;
  const int64_t j = idx[b * NK + e];
  out[t] = (j >= 0 && j < N1) ? in[bc * N1 + j] : (T)0;
}

template <typename T>
__global__ void group_points_bwd_kernel(const T* __restrict__ go, const int64_t* __restrict__ idx, int B,
                                        int C, int64_t N1, int64_t NK, T* __restrict__ gi) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)B * C * NK) return;
  const int64_t e = t % NK;
  const int64_t bc = t / NK;
  const int64_t b = bc / C;
  const int64_t j = idx[b * NK + e];
  if (j >= 0 && j < N1) atomicAdd(gi + bc * N1 + j, go[t]);  // group_points_kernel.cu:85-87
}

// ---------------------------------------------------------------------------------------------
// Fused input of FeatureAggregation (reference mvpnet/models/mvpnet_3d.py:54-58 fed by the two
// group_points calls of architectures_sphere.py:266-274): for every (point n, neighbour kk) pair
//   X[c, n*k+kk]   = feature_2d[view(p), c, pix(p)]            c < C,  p = knn[n,kk]
//   X[C+0..2, .]   = image_xyz[p] - point[n]                   (diff_xyz)
//   X[C+3, .]      = |diff|^2                                   (sum of squares in x,y,z order)
// X is channel-major [C+4, np*k] (the layout the 1x1 convolution wants as a transposed GEMM operand);
// lanes run over pairs, so every channel row is written coalesced and the gather stays inside one
// h*w channel plane (L2 resident). Reads the 2D feature map in its native (nv, C, h, w) layout: the
// (b, C, nv*h*w) transpose copy of the reference is not needed.
// ---------------------------------------------------------------------------------------------
__global__ void fa_gather_kernel(const float* __restrict__ feat /* [nv,C,hw], or [nv,hw,C] (channels last) */,
                                 const float* __restrict__ xyz /* [nv*hw,3] */,
                                 const int64_t* __restrict__ knn /* [np*k] */, const float* __restrict__ pts /* [np,3] */,
                                 int C, int nv, int64_t hw, int64_t npk, int k, float* __restrict__ X /* [C+4, npk] */,
                                 int64_t ch_stride, int64_t px_stride) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npk) return;
  const int64_t p = knn[e];
  const bool ok = p >= 0 && p < nv * hw;
  const int64_t view = ok ? p / hw : 0, pix = ok ? p % hw : 0;
  const float* f = feat + view * C * hw + pix * px_stride;
  for (int c = blockIdx.y; c < C; c += gridDim.y) X[(int64_t)c * npk + e] = ok ? f[(int64_t)c * ch_stride] : 0.f;
  if (blockIdx.y == 0) {
    const int64_t n = e / k;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (ok) {
      dx = xyz[p * 3] - pts[n * 3];
      dy = xyz[p * 3 + 1] - pts[n * 3 + 1];
      dz = xyz[p * 3 + 2] - pts[n * 3 + 2];
    }
    X[(int64_t)C * npk + e] = dx;
    X[(int64_t)(C + 1) * npk + e] = dy;
    X[(int64_t)(C + 2) * npk + e] = dz;
    X[(int64_t)(C + 3) * npk + e] = (dx * dx + dy * dy) + dz * dz;
  }
}

// ---------------------------------------------------------------------------------------------
// Sphere extraction around a centre (reference ScanNetDataset.potential_item,
// KPConv-PyTorch/datasets/ScanNet_sphere_color.py:556-597: sklearn KDTree.query_radius on float64
// data, membership rdist = dx^2+dy^2+dz^2 <= r^2) and the Tukey update of the sampling potentials
// (:576-584). Points are float32 promoted to float64 like the tree's data; evaluation order as
// sklearn's euclidean rdist. Ordered stream compaction: per-block ballot counts -> block scan ->
// scatter, so the indices come out ascending (the reference's order is KD-tree traversal order; the
// contract here is the SET).
// ---------------------------------------------------------------------------------------------
struct Ball {
  double cx, cy, cz, r2;
};

__device__ __forceinline__ double ball_rdist(const float* __restrict__ pts, int64_t i, const Ball& b) {
  const double dx = (double)pts[i * 3] - b.cx, dy = (double)pts[i * 3 + 1] - b.cy, dz = (double)pts[i * 3 + 2] - b.cz;
  double d2 = 0.0;
  d2 += dx * dx;
  d2 += dy * dy;
  d2 += dz * dz;
  return d2;
}

// grid ceil(N/1024), block 1024: block_count[b] = members in the block
__global__ __launch_bounds__(1024) void ball_count_kernel(const float* __restrict__ pts, int64_t N, Ball b,
                                                          int* __restrict__ block_count) {
  __shared__ int wc[16];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const bool in = i < N && ball_rdist(pts, i, b) <= b.r2;
  const unsigned long long m = __ballot(in);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wc[w];
    block_count[blockIdx.x] = t;
  }
}

// single block: exclusive scan of block counts in place, total -> *count
__global__ __launch_bounds__(1024) void ball_scan_kernel(int* __restrict__ block_count, int nblk, int64_t* __restrict__ count) {
  __shared__ int carry;
  __shared__ int ws[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? block_count[i] : 0;
    int x = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) ws[w] = x;
    __syncthreads();
    int off = carry;
    for (int k = 0; k < w; ++k) off += ws[k];
    if (i < nblk) block_count[i] = off + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(1024) void ball_scatter_kernel(const float* __restrict__ pts, int64_t N, Ball b,
                                                            const int* __restrict__ block_off,
                                                            int64_t* __restrict__ out_idx, double* __restrict__ out_d2) {
  __shared__ int wc[16];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  double d2 = 0.0;
  bool in = false;
  if (i < N) {
    d2 = ball_rdist(pts, i, b);
    in = d2 <= b.r2;
  }
  const unsigned long long m = __ballot(in);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wc[w] = __popcll(m);
  __syncthreads();
  if (in) {
    int off = block_off[blockIdx.x];
    for (int k = 0; k < w; ++k) off += wc[k];
    off += __popcll(m & ((1ull << lane) - 1ull));
    out_idx[off] = i;
    if (out_d2) out_d2[off] = d2;
  }
}

// potentials[i] += tukey(d2) for the members (ScanNet_sphere_color.py:576-582)
__global__ void tukey_update_kernel(const float* __restrict__ pts, int64_t N, Ball b, double* __restrict__ pot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double rd = ball_rdist(pts, i, b);
  if (rd <= b.r2) {
    const double d = sqrt(rd);          // query_radius returns dist = sqrt(rdist) ...
    const double d2s = d * d;           // ... and the reference squares it again (:576)
    double t = 1.0 - d2s / b.r2;
    t = t * t;
    if (d2s > b.r2) t = 0.0;            // :579
    pot[i] += t;
  }
}

}  // namespace

extern "C" int64_t mvk_ball_query_workspace(int64_t N) { return (cdiv64(N > 0 ? N : 1, 1024) + 4) * 4 + 64; }

extern "C" int mvk_ball_query(const float* pts, int64_t N, const double* center_host, double radius,
                              int64_t* out_idx, double* out_d2, int64_t* count_dev, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  MVK_REQUIRE(N >= 0 && radius >= 0.0, "ball_query: bad arguments");
  MVK_REQUIRE(workspace && workspace_bytes >= mvk_ball_query_workspace(N), "ball_query: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)cdiv64(N > 0 ? N : 1, 1024);
  Ball b{center_host[0], center_host[1], center_host[2], radius * radius};
  int* bc = (int*)workspace;
  hipLaunchKernelGGL(ball_count_kernel, dim3(nblk), dim3(1024), 0, st, pts, N, b, bc);
  hipLaunchKernelGGL(ball_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nblk, count_dev);
  hipLaunchKernelGGL(ball_scatter_kernel, dim3(nblk), dim3(1024), 0, st, pts, N, b, bc, out_idx, out_d2);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_tukey_update(const float* pts, int64_t N, const double* center_host, double radius,
                                double* potentials, void* stream) {
  MVK_REQUIRE(N >= 0 && radius > 0.0, "tukey_update: bad arguments");
  if (N == 0) return 0;
  Ball b{center_host[0], center_host[1], center_host[2], radius * radius};
  hipLaunchKernelGGL(tukey_update_kernel, dim3((unsigned)cdiv64(N, 256)), dim3(256), 0, (hipStream_t)stream, pts, N, b,
                     potentials);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace {

}  // namespace

extern "C" int mvk_fa_gather_fwd_ex(const float* feature_2d, int channels_last, const float* image_xyz, const int64_t* knn,
                                    const float* points, int C, int nv, int64_t hw, int64_t np, int k, float* X,
                                    void* stream) {
  MVK_REQUIRE(C > 0 && nv > 0 && hw > 0 && np >= 0 && k > 0, "fa_gather: bad sizes");
  const int64_t npk = np * k;
  if (npk == 0) return 0;
  // channels last: a pixel's C channels are one contiguous run, the 8 channel slices of a point (grid.y) share its lines
  dim3 grid((unsigned)cdiv64(npk, 256), 8);
  hipLaunchKernelGGL(fa_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, feature_2d, image_xyz, knn, points, C, nv,
                     hw, npk, k, X, channels_last ? (int64_t)1 : hw, channels_last ? (int64_t)C : (int64_t)1);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_fa_gather_fwd(const float* feature_2d, const float* image_xyz, const int64_t* knn,
                                 const float* points, int C, int nv, int64_t hw, int64_t np, int k, float* X,
                                 void* stream) {
  return mvk_fa_gather_fwd_ex(feature_2d, 0, image_xyz, knn, points, C, nv, hw, np, k, X, stream);
}

extern "C" int mvk_unproject_depth(const uint16_t* depth, int nv, int h, int w, const double* cam_inv,
                                   const float* poses, double* xyz, uint8_t* valid, void* stream) {
  MVK_REQUIRE(nv >= 0 && h > 0 && w > 0, "unproject: bad sizes");
  if (nv == 0) return 0;
  Cam cam;
  for (int i = 0; i < 9; ++i) cam.kinv[i] = cam_inv[i];  // HOST pointer: 9 doubles
  const int64_t tot = (int64_t)nv * h * w;
  hipLaunchKernelGGL(unproject_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream, depth,
                     nv, h, w, cam, poses, xyz, valid);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace {
// key splits: enough (query block x key range) workgroups to give every CU ~4 of them
void knn_plan(int64_t nq, int64_t nk, int* nsplit_out, int64_t* per_split_out) {
  const int64_t qblocks = cdiv64(nq > 0 ? nq : 1, KNN_T * QPT);
  const int64_t keys = nk > 0 ? nk : 1;
  int64_t nsplit = cdiv64(1024, qblocks);
  const int64_t max_split = cdiv64(keys, 4 * KNN_T);
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 64) nsplit = 64;
  const int64_t per_split = cdiv64(cdiv64(keys, nsplit), KNN_T) * KNN_T;
  *nsplit_out = (int)cdiv64(keys, per_split);
  *per_split_out = per_split;
}
}  // namespace

namespace {
int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

// the pruned path pays five small launches: only worth it when the brute-force product is large
bool knn_use_pruned(int64_t nq, int64_t nk) {
  const char* e = getenv("MVK_KNN_BRUTE");
  if (e && e[0] == '1') return false;
  return nq >= 1024 && nk >= 4096 && nq + nk < (int64_t)1 << 30;
}

int64_t pruned_layout(int64_t nq, int64_t nk, char* base, PrunedWs* ws) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const int64_t ntiles = cdiv64(nk, PT);
  char* cnt = take(2 * PNC * 4);
  char* start = take(2 * PNC * 4);
  char* cell = take((nq + nk) * 4);
  char* rank = take((nq + nk) * 4);
  char* qorder = take(nq * 4);
  char* skey = take(nk * 24);
  char* sidx = take(nk * 4);
  char* box = take(ntiles * 32);
  char* nvalid = take(4);
  if (ws) {
    ws->cnt = (int*)cnt; ws->start = (int*)start; ws->cell = (int*)cell; ws->rank = (int*)rank;
    ws->qorder = (int*)qorder; ws->skey = (double*)skey; ws->sidx = (int*)sidx; ws->box = (float*)box;
    ws->nvalid = (int*)nvalid;
  }
  return off;
}
}  // namespace

extern "C" int64_t mvk_knn_workspace(int64_t nq, int64_t nk, int k) {
  int nsplit;
  int64_t per;
  knn_plan(nq, nk, &nsplit, &per);
  const int64_t brute = (int64_t)nsplit * (nq > 0 ? nq : 1) * (k <= 3 ? 3 : 8) * 16 + 64;
  const int64_t pruned = pruned_layout(nq > 0 ? nq : 1, nk > 0 ? nk : 1, nullptr, nullptr) + 256;
  return brute > pruned ? brute : pruned;
}

extern "C" int mvk_knn_f64(const float* queries, int64_t nq, const double* keys, const uint8_t* key_valid,
                           int64_t nk, int k, int64_t* out_idx, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  MVK_REQUIRE(k >= 1 && k <= KMAXNN, "knn: k=%d unsupported (1..%d)", k, KMAXNN);
  MVK_REQUIRE(nq >= 0 && nk >= 0, "knn: bad sizes");
  if (nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int nsplit;
  int64_t per_split;
  knn_plan(nq, nk, &nsplit, &per_split);
  const int KK = k <= 3 ? 3 : 8;
  MVK_REQUIRE(workspace && workspace_bytes >= mvk_knn_workspace(nq, nk, k), "knn: workspace too small");
  if (knn_use_pruned(nq, nk)) {
    PrunedWs ws;
    char* base = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
    pruned_layout(nq, nk, base, &ws);
    const int ntiles = (int)cdiv64(nk, PT);
    MVK_CHECK_HIP(hipMemsetAsync(ws.cnt, 0, 2 * PNC * 4, st));
    const unsigned eb = (unsigned)cdiv64(nq + nk, 256);
    hipLaunchKernelGGL(pk_count_kernel, dim3(eb), dim3(256), 0, st, queries, nq, keys, key_valid, nk, ws);
    hipLaunchKernelGGL(pk_scan_kernel, dim3(2), dim3(1024), 0, st, ws);
    hipLaunchKernelGGL(pk_scatter_kernel, dim3(eb), dim3(256), 0, st, nq, keys, nk, ws);
    hipLaunchKernelGGL(pk_box_kernel, dim3((unsigned)ntiles), dim3(64), 0, st, ws);
    const dim3 grid((unsigned)cdiv64(nq, 64));
    if (KK == 3)
      hipLaunchKernelGGL((knn_pruned_kernel<3, 8>), grid, dim3(64 * 8), 0, st, queries, nq, ws, ntiles, k, out_idx);
    else
      hipLaunchKernelGGL((knn_pruned_kernel<8, 4>), grid, dim3(64 * 4), 0, st, queries, nq, ws, ntiles, k, out_idx);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  double* pd = (double*)workspace;
  int64_t* pi = (int64_t*)((char*)workspace + (int64_t)nsplit * nq * KK * 8);
  dim3 grid((unsigned)cdiv64(nq, KNN_T * QPT), (unsigned)nsplit);
  if (KK == 3) {
    hipLaunchKernelGGL((knn_partial_kernel<3>), grid, dim3(KNN_T), 0, st, queries, nq, keys, key_valid, nk,
                       per_split, pd, pi);
    hipLaunchKernelGGL((knn_merge_kernel<3>), dim3((unsigned)cdiv64(nq, 256)), dim3(256), 0, st, pd, pi, nsplit,
                       nq, k, out_idx);
  } else {
    hipLaunchKernelGGL((knn_partial_kernel<8>), grid, dim3(KNN_T), 0, st, queries, nq, keys, key_valid, nk,
                       per_split, pd, pi);
    hipLaunchKernelGGL((knn_merge_kernel<8>), dim3((unsigned)cdiv64(nq, 256)), dim3(256), 0, st, pd, pi, nsplit,
                       nq, k, out_idx);
  }
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_group_points_fwd(const float* points, const int64_t* index, int B, int C, int64_t N1,
                                    int64_t N2, int K, float* out, void* stream) {
  MVK_REQUIRE(B >= 0 && C >= 0 && N1 >= 0 && N2 >= 0 && K >= 0, "group_points: bad sizes");
  const int64_t tot = (int64_t)B * C * N2 * K;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(group_points_fwd_kernel<float>, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     points, index, B, C, N1, N2 * K, out);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// float64 twins (the reference's extension dispatches float and double, group_points_kernel.cu:60,130)
extern "C" int mvk_group_points_fwd_f64(const double* points, const int64_t* index, int B, int C, int64_t N1,
                                        int64_t N2, int K, double* out, void* stream) {
  MVK_REQUIRE(B >= 0 && C >= 0 && N1 >= 0 && N2 >= 0 && K >= 0, "group_points: bad sizes");
  const int64_t tot = (int64_t)B * C * N2 * K;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(group_points_fwd_kernel<double>, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     points, index, B, C, N1, N2 * K, out);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_group_points_bwd_f64(const double* grad_out, const int64_t* index, int B, int C, int64_t N1,
                                        int64_t N2, int K, double* grad_in, void* stream) {
  const int64_t tot = (int64_t)B * C * N2 * K;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(group_points_bwd_kernel<double>, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out, index, B, C, N1, N2 * K, grad_in);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_group_points_bwd(const float* grad_out, const int64_t* index, int B, int C, int64_t N1,
                                    int64_t N2, int K, float* grad_in, void* stream) {
  const int64_t tot = (int64_t)B * C * N2 * K;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(group_points_bwd_kernel<float>, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out, index, B, C, N1, N2 * K, grad_in);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
