// Voxel-grid barycentre subsampling on gfx950, bit-identical (values AND output order) to the
// reference CPU extension (KPConv-PyTorch/cpp_wrappers/cpp_subsampling/grid_subsampling/
// grid_subsampling.cpp:5-211, grid_subsampling.h:10-80, cpp_utils/cloud/cloud.cpp:27-67).
//
// Compiled with -ffp-contract=off: the reference is x86-64 without FMA, and its float32
// arithmetic (true division, floor, sequential in-order sums) is reproduced exactly.
//
// One 1024-thread workgroup per cloud runs every phase of that cloud back to back (phases are
// separated by __syncthreads(); all scratch lives in an HBM workspace that stays L2 resident):
//   P1  min / max corner, origin, grid dims                         (cloud.cpp:27-67, :25-31)
//   P2  voxel key per point (:53-56) -> open-addressing hash table insert (64-bit CAS);
//       first[slot] = min point index (= first occurrence of the voxel)
//   P3  flag first occurrences, block scan -> voxel id in FIRST-OCCURRENCE order, which is the
//       insertion order of the reference's unordered_map (:59-60)
//   P4-6 member counts, segment offsets (scan), scatter of point indices into segments
//   P7  one lane per voxel: sort its members by point index (segments are tiny) and accumulate
//       the float32 sums sequentially in input order (grid_subsampling.h:74-79); barycentre =
//       sum * (float)(1.0/count) (:87), feature mean = fsum / (float)count (:90-94)
//   P8  emulate libstdc++'s unordered_map<size_t,...> iteration order: one
//       "epoch" per bucket-count of the growth schedule 13, 29, 59, 127, ... ; in each epoch the
//       new list position of every element is
//           #elements in buckets first touched later  +  #same-bucket elements processed later
//       computed with atomics + a block scan, no sequential list walk
//   P9  write barycentres in that order to the cloud's staging rows
// A second tiny kernel compacts the clouds (prefix over B) and applies max_p.
#include "common.h"
#include "blockscan.h"
#include "../../include/mvk_prime_list.h"

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;

struct SubWs {
  // per point (N)
  int* slot;      // hash slot of the point's voxel
  int* member;    // point indices grouped by voxel
  int* scan;      // scratch for scans / flags
  // hash table (4N)
  unsigned long long* hkey;
  int* hfirst;
  int* hvox;
  // per voxel (N)
  unsigned long long* vkey;
  int* vcount;
  int* vseg;
  int* vcursor;
  float* vbary;   // 3N
  float* vfeat;   // N*fdim
  int* vlab;      // N*ldim
  int* lab_key;   // N (first-seen distinct labels of a voxel, at its segment offset)
  int* lab_cnt;   // N
  int* errflag;   // 1
  int* tau;
  int* posnew;
  int* nextb;
  int* tarr;
  // per bucket (3N + 32B)
  int* ft;
  int* bcnt;
  int* bhead;
  // per cloud
  int* out_count;  // B
  // device-lens entry (mvk_grid_subsample_batch_dev): per-cloud lengths instead of host offsets, optional grid
  // orientation applied by the cloud's own workgroup before phase 1 (all null / unused otherwise)
  const int* lens_dev;
  const float* rot_dev;   // B x 9
  float* rot_pts;         // N x 3 scratch for the oriented cloud
  // staging (N rows)
  float* stage_pts;
  float* stage_feat;
  int* stage_lab;
  // development (MVK_SUB_TIMING=1): [cloud][16] wall-clock stamps (100 MHz) at the phase boundaries, or null
  long long* dbg;
  // multi-workgroup front end (round 4): phases 0-7 of a cloud run as launches over all its points (sub_*_kernel below);
  // the per-cloud kernel then starts at phase 8 with the voxel count from mdev
  int multi;
  int chunks;            // 256-point chunks per cloud in those launches
  unsigned int* bbox;    // B x 6: order-preserving integer images of min xyz / max xyz (atomicMin / atomicMax)
  int* wgcnt;            // B x chunks: first occurrences per chunk
  int* mdev;             // B: voxels per cloud
};

#define SUB_STAMP(k)                                                         \
  do {                                                                       \
    if (W.dbg && threadIdx.x == 0) W.dbg[blockIdx.x * 16 + (k)] = wall_clock64(); \
  } while (0)

// Bucket-count schedule of libstdc++'s unordered_map (13, 29, 59, 127, ...): nb[e+1] =
// next_bkt(2 * nb[e]); identical for every cloud, computed once on the host.
struct Schedule {
  unsigned long long nb[48];
};

__device__ __forceinline__ unsigned int hash64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  return (unsigned int)k;
}

// Majority label of one voxel: first maximum in the iteration order of the reference's
// unordered_map<int,int> histogram (grid_subsampling.cpp:100-101). Distinct labels arrive in
// first-seen (= insertion) order; the order emulation is the per-epoch rank rule of P8 evaluated by
// one lane with O(d^2) loops (d = distinct labels in the voxel, <= MAXLAB).
constexpr int MAXLAB = 64;
__device__ int label_vote(const int* L, const int* Cn, int nd, const Schedule& sched) {
  if (nd == 1) return L[0];
  int tau[MAXLAB], ftv[MAXLAB], posn[MAXLAB];
  int start = 0, epoch = 0;
  unsigned long long nb = sched.nb[0];
  while (start < nd) {
    const int end = (unsigned long long)nd < nb ? nd : (int)nb;
    for (int v = start; v < end; ++v) tau[v] = v;
    for (int v = 0; v < end; ++v) {
      const unsigned long long bv = (unsigned long long)(long long)L[v] % nb;   // hash<int> = sign-extending cast
      int f = tau[v];
      for (int u = 0; u < end; ++u)
        if ((unsigned long long)(long long)L[u] % nb == bv && tau[u] < f) f = tau[u];
      ftv[v] = f;
    }
    for (int v = 0; v < end; ++v) {
      const unsigned long long bv = (unsigned long long)(long long)L[v] % nb;
      int r = 0;
      for (int u = 0; u < end; ++u) {
        const bool same = (unsigned long long)(long long)L[u] % nb == bv;
        r += same ? (tau[u] > tau[v]) : (ftv[u] > ftv[v]);
      }
      posn[v] = r;
    }
    for (int v = 0; v < end; ++v) tau[v] = posn[v];
    start = end;
    nb = sched.nb[++epoch];
  }
  int best = L[0], bestc = -1;
  for (int r = 0; r < nd; ++r)
    for (int v = 0; v < nd; ++v)
      if (tau[v] == r && Cn[v] > bestc) {
        bestc = Cn[v];
        best = L[v];
      }
  return best;
}

// Phase 7 for ONE voxel: its members sorted by point index, the float32 sums in input order (grid_subsampling.h:74-79),
// barycentre, feature means, label vote. m = the voxel's segment of `member`, c = its point count.
__device__ void voxel_accumulate(int v, int* m, int c, const float* __restrict__ P, const float* __restrict__ F, int fdim,
                                 const int* __restrict__ labels, int ldim, int off, const int* vseg, float* vbary,
                                 float* vfeat, const SubWs& W, const Schedule& sched) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (c <= 8) {
      // the common case (a voxel holds ~4 points): members into registers in one round of loads, sorted there by an
      // odd-even transposition network, the eight point loads issued together, the sums added in input order as before
      int mm[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) mm[a] = m[a < c ? a : 0];
#pragma unroll
      for (int a = 0; a < 8; ++a) mm[a] = a < c ? mm[a] : 0x7fffffff;
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int a = r & 1; a + 1 < 8; a += 2) {
          const int lo = min(mm[a], mm[a + 1]), hi = max(mm[a], mm[a + 1]);
          mm[a] = lo;
          mm[a + 1] = hi;
        }
      float px[8], py[8], pz[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const float* pp = P + (int64_t)(a < c ? mm[a] : mm[0]) * 3;
        px[a] = pp[0];
        py[a] = pp[1];
        pz[a] = pp[2];
      }
#pragma unroll
      for (int a = 0; a < 8; ++a)
        if (a < c) {
          m[a] = mm[a];          // (the feature / label sums below read the sorted members from memory)
          sx += px[a];
          sy += py[a];
          sz += pz[a];
        }
    } else {
      for (int a = 1; a < c; ++a) {  // insertion sort by point index
        int key = m[a], z = a - 1;
        while (z >= 0 && m[z] > key) {
          m[z + 1] = m[z];
          --z;
        }
        m[z + 1] = key;
      }
      for (int a = 0; a < c; ++a) {
        const float* p = P + (int64_t)m[a] * 3;
        sx += p[0];
        sy += p[1];
        sz += p[2];
      }
    }
    const float r = (float)(1.0 / (double)c);  // :87, cloud.h:120
    vbary[v * 3] = sx * r;
    vbary[v * 3 + 1] = sy * r;
    vbary[v * 3 + 2] = sz * r;
    if (fdim > 0) {
      const float cf = (float)c;
      for (int d = 0; d < fdim; ++d) {
        float s = 0.f;
        for (int a = 0; a < c; ++a) s += F[(int64_t)m[a] * fdim + d];
        vfeat[(int64_t)v * fdim + d] = s / cf;  // :90-94
      }
    }
    for (int d = 0; d < ldim; ++d) {  // label histograms in first-seen order (grid_subsampling.h:44-52)
      int* Lk = W.lab_key + off + vseg[v];
      int* Lc = W.lab_cnt + off + vseg[v];
      int nd = 0;
      for (int a = 0; a < c; ++a) {
        const int l = labels[((int64_t)off + m[a]) * ldim + d];
        int z = 0;
        while (z < nd && Lk[z] != l) ++z;
        if (z == nd) {
          Lk[nd] = l;
          Lc[nd] = 0;
          ++nd;
        }
        Lc[z] += 1;
      }
      if (nd > MAXLAB) {
        atomicExch(W.errflag, 1);
        nd = MAXLAB;
      }
      W.vlab[((int64_t)off + v) * ldim + d] = label_vote(Lk, Lc, nd, sched);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Multi-workgroup front end (round 4). One workgroup per cloud is bound by what ONE compute unit's memory pipeline
// retires (phases 2-7 of a 19 464-point cloud: ~100 k scattered accesses each, 35-80 us per phase whatever is in flight);
// here phases 0-7 run as launches over ALL points of all clouds (grid = 256-point chunks x clouds), the same data
// structures in the same workspace, and the per-cloud kernel starts at phase 8 (the unordered_map order is sequential
// by nature and lives in LDS). Results are bit-identical: the voxel keys, first occurrences, member lists and the
// per-voxel arithmetic are the same; only WHO computes them changes.

__device__ __forceinline__ void cloud_range(const SubWs& W, const int* __restrict__ offs, int b, int& off, int& n) {
  if (W.lens_dev) {
    off = 0;
    for (int i = 0; i < b; ++i) off += max(W.lens_dev[i], 0);
    n = max(W.lens_dev[b], 0);
  } else {
    off = offs[b];
    n = offs[b + 1] - off;
  }
}

// order-preserving integer image of a float (atomicMin / atomicMax on unsigned words)
__device__ __forceinline__ unsigned int f2ord(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int k) { return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ unsigned int table_size(int n) {
  unsigned int T = 2;
  while (T < 2u * (unsigned)n) T <<= 1;
  return T;
}

// phase "-1": empty hash table, zero voxel counters, neutral bounding box
__global__ __launch_bounds__(256) void sub_prep_kernel(const int* __restrict__ offs, SubWs W) {
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const unsigned int i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) {
    for (int c = 0; c < 3; ++c) {
      W.bbox[b * 6 + c] = 0xffffffffu;
      W.bbox[b * 6 + 3 + c] = 0u;
    }
    W.mdev[b] = 0;
  }
  if (n == 0) return;
  if (i < table_size(n)) {
    W.hkey[(int64_t)off * 4 + i] = EMPTY_KEY;
    W.hfirst[(int64_t)off * 4 + i] = 0x7fffffff;
    W.hvox[(int64_t)off * 4 + i] = 0;          // until phase 3b: the number of points of the slot's voxel
  }
  if (i < (unsigned)n) W.vcursor[off + i] = 0;
}

// phases 0-1: grid orientation (device-lens entry), min / max corner
__global__ __launch_bounds__(256) void sub_minmax_kernel(const float* __restrict__ pts, const int* __restrict__ offs, SubWs W) {
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  float p[3] = {INFINITY, INFINITY, INFINITY}, q[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < n) {
    const float* P = pts + ((int64_t)off + i) * 3;
    float v[3] = {P[0], P[1], P[2]};
    if (W.rot_dev) {
      const float* M = W.rot_dev + b * 9;
      float* Q = W.rot_pts + ((int64_t)off + i) * 3;
      const float p0 = v[0], p1 = v[1], p2 = v[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] = (p0 * M[0 * 3 + c] + p1 * M[1 * 3 + c]) + p2 * M[2 * 3 + c];
        Q[c] = v[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = q[c] = v[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (int o = 32; o >= 1; o >>= 1) {
      p[c] = fminf(p[c], __shfl_xor(p[c], o));
      q[c] = fmaxf(q[c], __shfl_xor(q[c], o));
    }
  }
  // one atomic per corner component and WORKGROUP (same-address atomics retire one after the other in L2: with one per
  // wavefront the 1 800 of a 19 464-point cloud were the kernel's 17 us)
  __shared__ float red[6][4];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      red[c][threadIdx.x >> 6] = p[c];
      red[3 + c][threadIdx.x >> 6] = q[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6 && blockIdx.x * 256 < n) {
    const int c = threadIdx.x;
    const float a = red[c][0], b1 = red[c][1], c2 = red[c][2], d3 = red[c][3];
    if (c < 3) atomicMin(&W.bbox[b * 6 + c], f2ord(fminf(fminf(a, b1), fminf(c2, d3))));
    else atomicMax(&W.bbox[b * 6 + c], f2ord(fmaxf(fmaxf(a, b1), fmaxf(c2, d3))));
  }
}

// phase 2: voxel key, hash insert, first occurrence
__global__ __launch_bounds__(256) void sub_insert_kernel(const float* __restrict__ pts, const int* __restrict__ offs, float dl,
                                                         SubWs W) {
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* P = (W.rot_dev ? W.rot_pts : pts) + (int64_t)off * 3;
  const float inv = 1 / dl;  // grid_subsampling.cpp:27
  float o3[3], m3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o3[c] = floorf(ord2f(W.bbox[b * 6 + c]) * inv) * dl;
    m3[c] = ord2f(W.bbox[b * 6 + 3 + c]);
  }
  const unsigned long long NX = (unsigned long long)floorf((m3[0] - o3[0]) / dl) + 1;  // :30
  const unsigned long long NY = (unsigned long long)floorf((m3[1] - o3[1]) / dl) + 1;  // :31
  const unsigned long long iX = (unsigned long long)floorf((P[i * 3] - o3[0]) / dl);      // :53
  const unsigned long long iY = (unsigned long long)floorf((P[i * 3 + 1] - o3[1]) / dl);  // :54
  const unsigned long long iZ = (unsigned long long)floorf((P[i * 3 + 2] - o3[2]) / dl);  // :55
  const unsigned long long key = iX + NX * iY + NX * NY * iZ;                             // :56
  const unsigned int T = table_size(n);
  unsigned long long* hkey = W.hkey + (int64_t)off * 4;
  unsigned int h = hash64(key) & (T - 1);
  unsigned long long prev = atomicCAS(&hkey[h], EMPTY_KEY, key);
  while (!(prev == EMPTY_KEY || prev == key)) {       // occupied by another voxel: linear probing
    h = (h + 1) & (T - 1);
    prev = atomicCAS(&hkey[h], EMPTY_KEY, key);
  }
  W.slot[off + i] = (int)h;
  atomicMin(&W.hfirst[(int64_t)off * 4 + h], i);
  atomicAdd(&W.hvox[(int64_t)off * 4 + h], 1);   // phase 4 (points per voxel) rides along: counted per slot
}

// phase 3a: first-occurrence flags and their count per chunk
__global__ __launch_bounds__(256) void sub_flag_kernel(const int* __restrict__ offs, SubWs W) {
  __shared__ int wsum[4];
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  int f = 0;
  if (i < n) {
    f = W.hfirst[(int64_t)off * 4 + W.slot[off + i]] == i ? 1 : 0;
    W.scan[off + i] = f;
  }
  const int cnt = __popcll(__ballot(f));
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) W.wgcnt[(int64_t)b * W.chunks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// phases 3b, 4: voxel ids in first-occurrence order (exclusive prefix of the flags: chunk base + scan inside the chunk),
// the voxel's point count from its slot
__global__ __launch_bounds__(256) void sub_vox_kernel(const int* __restrict__ offs, SubWs W) {
  __shared__ int red[256];
  __shared__ int wsum[4];
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int* cnts = W.wgcnt + (int64_t)b * W.chunks;
  int part = 0, all = 0;
  for (int c = threadIdx.x; c < W.chunks; c += 256) {
    const int v = cnts[c];
    all += v;
    part += c < (int)blockIdx.x ? v : 0;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const int base = red[0];
  __syncthreads();
  if (blockIdx.x == 0) {                      // the cloud's voxel count, once
    red[threadIdx.x] = all;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) W.mdev[b] = red[0];
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int f = i < n ? W.scan[off + i] : 0;
  const unsigned long long bal = __ballot(f);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wv; ++w) wbase += wsum[w];
  if (f) {
    const int vid = base + wbase + before;
    const int sl = W.slot[off + i];
    W.vcount[off + vid] = W.hvox[(int64_t)off * 4 + sl];
    W.hvox[(int64_t)off * 4 + sl] = vid;
    W.vkey[off + vid] = W.hkey[(int64_t)off * 4 + sl];
  }
}

// phase 5: segment offsets (one workgroup per cloud: a scan over the cloud's voxels)
__global__ __launch_bounds__(TPB) void sub_offsets_kernel(const int* __restrict__ offs, SubWs W) {
  __shared__ int sh[TPB / 64 + 2];
  const int b = blockIdx.x;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int M = W.mdev[b];
  int* vseg = W.vseg + off;
  const int* vcount = W.vcount + off;
  for (int v = threadIdx.x; v < M; v += TPB) vseg[v] = vcount[v];
  __syncthreads();
  block_scan_array_g(vseg, M, sh, false);
}

// phase 6: member lists
__global__ __launch_bounds__(256) void sub_scatter_kernel(const int* __restrict__ offs, SubWs W) {
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = W.hvox[(int64_t)off * 4 + W.slot[off + i]];
  const int pos = atomicAdd(&W.vcursor[off + v], 1);
  W.member[off + W.vseg[off + v] + pos] = i;
}

// phase 7: one thread per voxel
// (one wavefront per workgroup: a voxel is a chain of dependent scattered loads, and 5 000 voxels in 256-thread
// workgroups sit on 20 compute units)
__global__ __launch_bounds__(64) void sub_sums_kernel(const float* __restrict__ pts, const float* __restrict__ feats, int fdim,
                                                       const int* __restrict__ labels, int ldim,
                                                       const int* __restrict__ offs, SubWs W, Schedule sched) {
  const int b = blockIdx.y;
  int off, n;
  cloud_range(W, offs, b, off, n);
  const int v = blockIdx.x * 64 + threadIdx.x;
  if (v >= W.mdev[b]) return;
  const float* P = (W.rot_dev ? W.rot_pts : pts) + (int64_t)off * 3;
  const float* F = fdim > 0 ? feats + (int64_t)off * fdim : nullptr;
  const int* vseg = W.vseg + off;
  voxel_accumulate(v, W.member + off + vseg[v], W.vcount[off + v], P, F, fdim, labels, ldim, off, vseg,
                   W.vbary + (int64_t)off * 3, fdim > 0 ? W.vfeat + (int64_t)off * fdim : nullptr, W, sched);
}

__global__ __launch_bounds__(TPB) void subsample_cloud_kernel(const float* __restrict__ pts,
                                                               const float* __restrict__ feats,
                                                               int fdim, const int* __restrict__ labels,
                                                               int ldim, const int* __restrict__ offs,
                                                               float dl, SubWs W, int B,
                                                               Schedule sched) {
  __shared__ float red[6][TPB / 64];
  __shared__ float corner[8];  // org xyz, NX, NY as bits
  __shared__ unsigned long long dims[2];
  __shared__ int sh[TPB / 64 + 2];
  const int b = blockIdx.x;
  int off, n;
  if (W.lens_dev) {          // offsets from the device lengths (was: a one-thread launch per call)
    off = 0;
    for (int i = 0; i < b; ++i) off += max(W.lens_dev[i], 0);
    n = max(W.lens_dev[b], 0);
  } else {
    off = offs[b];
    n = offs[b + 1] - off;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (n == 0) {
    if (tid == 0) W.out_count[b] = 0;
    return;
  }
  const float* P = pts + (int64_t)off * 3;
  int M = 0;
  unsigned long long* vkey = W.vkey + off;
  float* vbary = W.vbary + (int64_t)off * 3;
  float* vfeat = fdim > 0 ? W.vfeat + (int64_t)off * fdim : nullptr;
  if (W.multi) {            // phases 0-7 ran as launches over all points (sub_*_kernel): start at the iteration order
    M = W.mdev[b];
    if (W.dbg && threadIdx.x == 0) {          // (the front end's phases are launches of their own: no time in this kernel)
      const unsigned long long t = wall_clock64();
      for (int k = 0; k < 7; ++k) W.dbg[blockIdx.x * 16 + k] = t;
    }
  } else {
  if (W.rot_dev) {
    // ---- P0: the cloud in the random grid orientation (datasets/common.py:118), same arithmetic as
    // rotate_cloud_kernel -- (p0 * R[0][i] + p1 * R[1][i]) + p2 * R[2][i] -- by the cloud's own workgroup
    // (was: a launch of its own in front of this kernel)
    const float* M = W.rot_dev + b * 9;
    float* Q = W.rot_pts + (int64_t)off * 3;
    for (int i = tid; i < n; i += TPB) {
      const float p0 = P[i * 3], p1 = P[i * 3 + 1], p2 = P[i * 3 + 2];
#pragma unroll
      for (int c = 0; c < 3; ++c) Q[i * 3 + c] = (p0 * M[0 * 3 + c] + p1 * M[1 * 3 + c]) + p2 * M[2 * 3 + c];
    }
    __syncthreads();
    P = Q;
  }

  SUB_STAMP(0);
  // ---- P1: min / max
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
    float inv = 1 / dl;  // grid_subsampling.cpp:27
    float o3[3], m3[3];
    for (int c = 0; c < 3; ++c) {
      float a = red[c][0], z = red[3 + c][0];
      for (int i = 1; i < TPB / 64; ++i) {
        a = fminf(a, red[c][i]);
        z = fmaxf(z, red[3 + c][i]);
      }
      o3[c] = floorf(a * inv) * dl;
      m3[c] = z;
      corner[c] = o3[c];
    }
    dims[0] = (unsigned long long)floorf((m3[0] - o3[0]) / dl) + 1;  // :30
    dims[1] = (unsigned long long)floorf((m3[1] - o3[1]) / dl) + 1;  // :31
  }
  __syncthreads();
  const float ox = corner[0], oy = corner[1], oz = corner[2];
  const unsigned long long NX = dims[0], NY = dims[1];

  SUB_STAMP(1);
  // ---- P2: hash insert
  unsigned int T = 2;
  while (T < 2u * (unsigned)n) T <<= 1;
  unsigned long long* hkey = W.hkey + (int64_t)off * 4;
  int* hfirst = W.hfirst + (int64_t)off * 4;
  int* hvox = W.hvox + (int64_t)off * 4;
  for (unsigned int i = tid; i < T; i += TPB) {
    hkey[i] = EMPTY_KEY;
    hfirst[i] = 0x7fffffff;
  }
  __syncthreads();
  int* slot = W.slot + off;
  // Eight points of a thread at a time (four until round 4): their first compare-and-swap attempts (and then their atomicMin) are issued
  // together, so a thread's chain of dependent L2 round trips is a quarter as long (the kernel is one workgroup per
  // cloud: 19 points per thread at 19 k points, every atomic a ~1.5 us round trip; 151 -> see DESIGN 4.3). The table
  // ends up with the same keys and the same minima whatever the order of the attempts.
  for (int i0 = tid; i0 < n; i0 += 8 * TPB) {
    unsigned long long key[8], prev[8];
    unsigned int h[8];
    bool live[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * TPB;
      live[u] = i < n;
      const int ii = live[u] ? i : i0;
      unsigned long long iX = (unsigned long long)floorf((P[ii * 3] - ox) / dl);      // :53
      unsigned long long iY = (unsigned long long)floorf((P[ii * 3 + 1] - oy) / dl);  // :54
      unsigned long long iZ = (unsigned long long)floorf((P[ii * 3 + 2] - oz) / dl);  // :55
      key[u] = iX + NX * iY + NX * NY * iZ;                                           // :56
      h[u] = hash64(key[u]) & (T - 1);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) prev[u] = live[u] ? atomicCAS(&hkey[h[u]], EMPTY_KEY, key[u]) : key[u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!live[u]) continue;
      unsigned long long p = prev[u];
      while (!(p == EMPTY_KEY || p == key[u])) {       // occupied by another voxel: linear probing
        h[u] = (h[u] + 1) & (T - 1);
        p = atomicCAS(&hkey[h[u]], EMPTY_KEY, key[u]);
      }
      slot[i0 + u * TPB] = (int)h[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (live[u]) atomicMin(&hfirst[h[u]], i0 + u * TPB);
  }
  __syncthreads();

  SUB_STAMP(2);
  // ---- P3: first-occurrence flags -> voxel ids in insertion order
  int* scan = W.scan + off;
  for (int i0 = tid; i0 < n; i0 += 8 * TPB) {          // eight dependent slot -> first chains of a thread in flight together
    int ai[8];
    int f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ai[u] = slot[i0 + u * TPB < n ? i0 + u * TPB : i0];
    ldg_agent<8>(f, hfirst, ai);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * TPB < n) scan[i0 + u * TPB] = (f[u] == i0 + u * TPB) ? 1 : 0;
  }
  __syncthreads();
  M = block_scan_array_g(scan, n, sh, false);
  int* vcount = W.vcount + off;
  int* vseg = W.vseg + off;
  int* vcursor = W.vcursor + off;
  for (int i0 = tid; i0 < n; i0 += 4 * TPB) {
    int f[4], sl[4];
    unsigned long long kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) sl[u] = slot[i0 + u * TPB < n ? i0 + u * TPB : i0];
    ldg_agent<4>(f, hfirst, sl);
    ldg_agent<4>(kv, hkey, sl);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TPB;
      if (i < n && f[u] == i) {
        const int vid = scan[i];
        hvox[sl[u]] = vid;
        vkey[vid] = kv[u];
      }
    }
  }
  for (int v = tid; v < M; v += TPB) {
    vcount[v] = 0;
    vcursor[v] = 0;
  }
  __syncthreads();
  SUB_STAMP(3);
  // ---- P4: counts
  for (int i0 = tid; i0 < n; i0 += 8 * TPB) {          // slot -> voxel id chains eight at a time, then the atomics
    int sl[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sl[u] = slot[i0 + u * TPB < n ? i0 + u * TPB : i0];
#pragma unroll
    for (int u = 0; u < 8; ++u) vv[u] = hvox[sl[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * TPB < n) atomicAdd(&vcount[vv[u]], 1);
  }
  __syncthreads();
  SUB_STAMP(4);
  // ---- P5: segment offsets
  for (int v0 = tid; v0 < M; v0 += 8 * TPB) {
    int ai[8];
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ai[u] = v0 + u * TPB < M ? v0 + u * TPB : v0;
    ldg_agent<8>(c, vcount, ai);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v0 + u * TPB < M) vseg[v0 + u * TPB] = c[u];
  }
  __syncthreads();
  block_scan_array_g(vseg, M, sh, false);
  SUB_STAMP(5);
  // ---- P6: scatter members
  int* member = W.member + off;
  for (int i0 = tid; i0 < n; i0 += 8 * TPB) {       // eight cursor increments of a thread in flight together (see P2)
    int sl[8], v[8], pos[8], sg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sl[u] = slot[i0 + u * TPB < n ? i0 + u * TPB : i0];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = hvox[sl[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) sg[u] = vseg[v[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) pos[u] = i0 + u * TPB < n ? atomicAdd(&vcursor[v[u]], 1) : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * TPB < n) member[sg[u] + pos[u]] = i0 + u * TPB;
  }
  __syncthreads();
  SUB_STAMP(6);
  // ---- P7: ordered sums
  const float* F = fdim > 0 ? feats + (int64_t)off * fdim : nullptr;
  for (int v = tid; v < M; v += TPB)
    voxel_accumulate(v, member + vseg[v], ld_agent(&vcount[v]), P, F, fdim, labels, ldim, off, vseg, vbary, vfeat, W, sched);
  __syncthreads();
  }   // (!W.multi)

  SUB_STAMP(7);
  // ---- P8: unordered_map iteration order
  // The epochs are inherently sequential (the final order depends on every rehash), and each one is a handful of
  // block-wide phases of atomics and scans over its live elements: with the per-epoch arrays in HBM every phase was a
  // round of L2 atomics / agent-scope loads (~1.5 us each way), ~8 us per epoch whatever its size, and a cloud of
  // 5 000 voxels runs nine of them. The epochs whose bucket count fits (nb <= L_NB: 13 ... 5087, i.e. all nine of a
  // 19 464-point level-0 cloud and everything below) keep the arrays in LDS (7 x 5 200 words = 146 of the CU's 160 KB: the
  // kernel is one workgroup per cloud anyway; with 2 400 the last epoch of level 0 ran on HBM arrays and was 28 of the
  // phase's 64 us); tau moves to HBM once the table outgrows it (clouds of more than 5 087 voxels).
  constexpr int L_NB = 5200;
  __shared__ int l_tau[L_NB], l_posnew[L_NB], l_nextb[L_NB], l_tarr[L_NB], l_ft[L_NB], l_bcnt[L_NB], l_bhead[L_NB];
  int* tau_g = W.tau + off;
  int* posnew_g = W.posnew + off;
  int* nextb_g = W.nextb + off;
  int* tarr_g = W.tarr + off;
  int* ft_g = W.ft + (int64_t)off * 3 + 32 * b;
  int* bcnt_g = W.bcnt + (int64_t)off * 3 + 32 * b;
  int* bhead_g = W.bhead + (int64_t)off * 3 + 32 * b;
  int epoch = 0;
  unsigned long long nb = sched.nb[0];
  int start = 0;
  bool in_lds = true;
  while (start < M) {
    const int end = (unsigned long long)M < nb ? M : (int)nb;  // elements [0,end) live in this epoch
    const bool use_lds = nb <= (unsigned long long)L_NB;
    if (in_lds && !use_lds) {                                   // the table outgrew LDS: tau of the older elements moves
      for (int v = tid; v < start; v += TPB) tau_g[v] = l_tau[v];
      in_lds = false;
      __syncthreads();
    }
    int* tau = use_lds ? l_tau : tau_g;
    int* posnew = use_lds ? l_posnew : posnew_g;
    int* nextb = use_lds ? l_nextb : nextb_g;
    int* tarr = use_lds ? l_tarr : tarr_g;
    int* ft = use_lds ? l_ft : ft_g;
    int* bcnt = use_lds ? l_bcnt : bcnt_g;
    int* bhead = use_lds ? l_bhead : bhead_g;
    for (int v = start + tid; v < end; v += TPB) tau[v] = v;   // new elements: processed in insertion order
    for (unsigned long long i = tid; i < nb; i += TPB) {
      ft[i] = 0x7fffffff;
      bcnt[i] = 0;
      bhead[i] = -1;
    }
    for (int t = tid; t < end; t += TPB) tarr[t] = 0;
    __syncthreads();
    for (int v = tid; v < end; v += TPB) {
      const int bk = (int)(vkey[v] % nb);
      atomicMin(&ft[bk], tau[v]);
      atomicAdd(&bcnt[bk], 1);
      nextb[v] = atomicExch(&bhead[bk], v);
    }
    __syncthreads();
    // (LDS epochs: the arrays are the workgroup's own memory, plain loads are coherent after the barrier and several can be
    // in flight; HBM epochs: agent-scope loads of words the L2-side atomics updated)
    auto LD = [&](const int* q) -> int { return use_lds ? *q : ld_agent(q); };
    for (int v = tid; v < end; v += TPB) {
      const int bk = (int)(vkey[v] % nb);
      if (LD(&ft[bk]) == tau[v]) tarr[tau[v]] = LD(&bcnt[bk]);
    }
    __syncthreads();
    if (use_lds) block_scan_array_lds(tarr, end, sh, true);  // tarr[t] = #elements in buckets first touched after t
    else block_scan_array_g(tarr, end, sh, true);
    for (int v = tid; v < end; v += TPB) {
      const int bk = (int)(vkey[v] % nb);
      const int tv = tau[v];
      int later = 0;
      for (int u = LD(&bhead[bk]); u >= 0; u = nextb[u]) later += (tau[u] > tv);
      posnew[v] = tarr[LD(&ft[bk])] + later;
    }
    __syncthreads();
    for (int v = tid; v < end; v += TPB) tau[v] = posnew[v];
    __syncthreads();
    start = end;
    nb = sched.nb[++epoch];
  }
  const int* tau = in_lds ? l_tau : tau_g;
  SUB_STAMP(8);
  // ---- P9: staging rows in iteration order
  float* sp = W.stage_pts + (int64_t)off * 3;
  float* sf = fdim > 0 ? W.stage_feat + (int64_t)off * fdim : nullptr;
  for (int v = tid; v < M; v += TPB) {
    const int o = tau[v];
    sp[o * 3] = vbary[v * 3];
    sp[o * 3 + 1] = vbary[v * 3 + 1];
    sp[o * 3 + 2] = vbary[v * 3 + 2];
    for (int d = 0; d < fdim; ++d) sf[(int64_t)o * fdim + d] = vfeat[(int64_t)v * fdim + d];
    for (int d = 0; d < ldim; ++d) W.stage_lab[((int64_t)off + o) * ldim + d] = W.vlab[((int64_t)off + v) * ldim + d];
  }
  if (tid == 0) W.out_count[b] = M;
  SUB_STAMP(9);
}

// Compacts the per-cloud staging rows into the stacked output and applies max_p
// (grid_subsampling.cpp:181-204).
__global__ void subsample_compact_kernel(const int* __restrict__ offs, SubWs W, int B, int fdim, int ldim,
                                         int max_p, float* __restrict__ out_pts,
                                         float* __restrict__ out_feats, int* __restrict__ out_labels,
                                         int* __restrict__ out_lens, int64_t out_cap, int* __restrict__ overflow) {
  const int b = blockIdx.y;
  int base = 0;
  for (int i = 0; i < b; ++i) base += min(W.out_count[i], max_p);
  int m = min(W.out_count[b], max_p);
  if (blockIdx.x == 0 && threadIdx.x == 0) out_lens[b] = m;
  if (base + m > out_cap) {   // device-lens variant with a fixed output capacity: flag, never write past it
    if (blockIdx.x == 0 && threadIdx.x == 0 && overflow) atomicExch(overflow, 1);
    m = base < out_cap ? (int)(out_cap - base) : 0;
  }
  const int off = offs[b];
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < m; o += gridDim.x * blockDim.x) {
    for (int c = 0; c < 3; ++c) out_pts[(int64_t)(base + o) * 3 + c] = W.stage_pts[(int64_t)(off + o) * 3 + c];
    for (int d = 0; d < fdim; ++d)
      out_feats[(int64_t)(base + o) * fdim + d] = W.stage_feat[(int64_t)(off + o) * fdim + d];
    for (int d = 0; d < ldim; ++d)
      out_labels[(int64_t)(base + o) * ldim + d] = W.stage_lab[(int64_t)(off + o) * ldim + d];
  }
}

// ---- device-lens variant (capturable in a hipGraph: no host reads, fixed launch geometry) ----------------

// Device-lens finish in ONE launch (was four: compaction, rotation back, padding, and the offsets launch in front):
// row t of the fixed-capacity output = barycentre o of cloud b from its staging rows, rotated back into the world frame
// (datasets/common.py:134: R transposed, same arithmetic as rotate_cloud_kernel), rows beyond the total <- pad;
// thread 0 publishes the per-cloud counts, their total and the overflow flag (a level that outgrew its capacity).
__global__ void subsample_finish_dev_kernel(SubWs W, int B, float* __restrict__ out_pts, int* __restrict__ out_lens,
                                            int64_t out_cap, float pad, int* __restrict__ total_out,
                                            int* __restrict__ overflow) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int tot = 0;
  for (int i = 0; i < B; ++i) tot += W.out_count[i];
  if (t == 0) {
    for (int i = 0; i < B; ++i) out_lens[i] = W.out_count[i];
    if (tot > out_cap && overflow) atomicExch(overflow, 1);
    if (total_out) *total_out = tot > out_cap ? (int)out_cap : tot;
  }
  if (t >= out_cap) return;
  if (t >= tot) {
    out_pts[t * 3] = pad;
    out_pts[t * 3 + 1] = pad;
    out_pts[t * 3 + 2] = pad;
    return;
  }
  int b = 0, base = 0, off = 0;           // cloud of output row t, first output row / first input row of that cloud
  while (b + 1 < B && t >= base + W.out_count[b]) {
    base += W.out_count[b];
    off += max(W.lens_dev[b], 0);
    ++b;
  }
  const float* p = W.stage_pts + ((int64_t)off + (t - base)) * 3;
  const float p0 = p[0], p1 = p[1], p2 = p[2];
  if (W.rot_dev) {
    const float* M = W.rot_dev + b * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) out_pts[t * 3 + i] = (p0 * M[i * 3 + 0] + p1 * M[i * 3 + 1]) + p2 * M[i * 3 + 2];
  } else {
    out_pts[t * 3] = p0;
    out_pts[t * 3 + 1] = p1;
    out_pts[t * 3 + 2] = p2;
  }
}

struct Carver {
  char* p;
  char* end;
  template <typename T>
  T* take(int64_t count) {
    uintptr_t a = ((uintptr_t)p + 15) & ~(uintptr_t)15;
    T* r = (T*)a;
    p = (char*)a + sizeof(T) * count;
    return r;
  }
};

int64_t ws_bytes(int64_t N, int B, int fdim, int ldim) {
  int64_t n = N > 0 ? N : 1;
  int64_t bytes = 0;
  bytes += 3 * n * 4;                       // slot member scan
  bytes += 4 * n * (8 + 4 + 4);             // hash
  bytes += n * (8 + 4 * 3 + 12 + 4 * 4);    // vkey vcount vseg vcursor vbary tau posnew nextb tarr
  bytes += n * (int64_t)fdim * 4 * 2;       // vfeat + stage_feat
  bytes += n * (int64_t)ldim * 4 * 2 + n * 8 + 64;  // vlab + stage_lab, lab_key, lab_cnt, errflag
  bytes += 3 * (3 * n + 32 * (int64_t)B) * 4;  // ft bcnt bhead
  bytes += (int64_t)(B + 1) * 4 * 2;        // out_count, offs
  bytes += n * 12;                          // stage_pts
  bytes += n * 12 + 16;                     // rot_pts (oriented variant)
  bytes += (int64_t)B * (6 + 1) * 4 + ((n + 255) / 256 + 1) * (int64_t)B * 4 + 64;   // bbox, mdev, wgcnt (multi-workgroup front end)
  return bytes + 64 * 32;
}

}  // namespace

namespace {

struct Rot3 {
  float m[9];
};

// out[n,i] = (p0 * R[0][i] + p1 * R[1][i]) + p2 * R[2][i]  (or R transposed): the float32 products and
// NumPy's axis-1 summation order of datasets/common.py:118 / :134. `count_dev` (optional) = device row
// counts per cloud; the rows of cloud `b` then start at sum(count_dev[:b]) (the compacted output).
__global__ void rotate_cloud_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t first,
                                    int64_t count, const int32_t* __restrict__ count_dev, int b, Rot3 R,
                                    int transpose) {
  if (count_dev) {
    first = 0;
    for (int i = 0; i < b; ++i) first += count_dev[i];
    count = count_dev[b];
  }
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const float* p = in + (first + t) * 3;
  const float p0 = p[0], p1 = p[1], p2 = p[2];
  float o[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r0 = transpose ? R.m[i * 3 + 0] : R.m[0 * 3 + i];
    const float r1 = transpose ? R.m[i * 3 + 1] : R.m[1 * 3 + i];
    const float r2 = transpose ? R.m[i * 3 + 2] : R.m[2 * 3 + i];
    o[i] = (p0 * r0 + p1 * r1) + p2 * r2;
  }
  float* q = out + (first + t) * 3;
  q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
}

// Clouds of at least this many points (the largest of a call, or the capacity of a device-lens call) take the
// multi-workgroup front end; below it one workgroup per cloud runs every phase itself (MVK_SUB_MULTI_MIN, 0 = never).
int64_t multi_min_points() {
  static const int64_t v = getenv("MVK_SUB_MULTI_MIN") ? atoll(getenv("MVK_SUB_MULTI_MIN")) : 8192;
  return v;
}

// phases 0-7 as launches over all points of all clouds; maxn = the longest cloud (or the capacity)
void launch_front_end(const float* pts, const float* feats, int fdim, const int32_t* labels, int ldim, const int* offs_d,
                      float dl, SubWs& W, int B, int64_t maxn, const Schedule& sched, hipStream_t st) {
  W.multi = 1;
  W.chunks = (int)cdiv64(maxn, 256);
  const dim3 g((unsigned)W.chunks, (unsigned)B), g4((unsigned)cdiv64(4 * maxn, 256), (unsigned)B), blk(256);
  hipLaunchKernelGGL(sub_prep_kernel, g4, blk, 0, st, offs_d, W);
  hipLaunchKernelGGL(sub_minmax_kernel, g, blk, 0, st, pts, offs_d, W);
  hipLaunchKernelGGL(sub_insert_kernel, g, blk, 0, st, pts, offs_d, dl, W);
  hipLaunchKernelGGL(sub_flag_kernel, g, blk, 0, st, offs_d, W);
  hipLaunchKernelGGL(sub_vox_kernel, g, blk, 0, st, offs_d, W);
  hipLaunchKernelGGL(sub_offsets_kernel, dim3(B), dim3(TPB), 0, st, offs_d, W);
  hipLaunchKernelGGL(sub_scatter_kernel, g, blk, 0, st, offs_d, W);
  hipLaunchKernelGGL(sub_sums_kernel, dim3((unsigned)((maxn + 63) / 64), g.y), dim3(64), 0, st, pts, feats, fdim, labels, ldim, offs_d, W, sched);
}

}  // namespace

extern "C" int64_t mvk_grid_subsample_workspace(int64_t N, int B, int fdim, int ldim) {
  return ws_bytes(N, B, fdim, ldim);
}

namespace {
int subsample_run(const float* pts, int64_t N, const int32_t* lens_host, int B, const float* rot_host,
                  const float* feats, int fdim, const int32_t* labels, int ldim,
                  float dl, int max_p, float* out_pts, float* out_feats,
                  int32_t* out_labels, int32_t* out_lens,
                  int32_t* out_lens_host, void* workspace,
                  int64_t workspace_bytes, void* stream) {
  MVK_REQUIRE(B >= 1 && N >= 0 && N < (1ll << 29), "subsample: bad sizes N=%lld B=%d", (long long)N, B);
  MVK_REQUIRE(dl > 0.f, "subsample: sampleDl must be positive");
  MVK_REQUIRE(fdim >= 0 && (fdim == 0 || (feats && out_feats)), "subsample: features pointer missing");
  MVK_REQUIRE(ldim >= 0 && (ldim == 0 || (labels && out_labels)), "subsample: labels pointer missing");
  MVK_REQUIRE(workspace != nullptr && workspace_bytes >= ws_bytes(N, B, fdim, ldim),
              "subsample: workspace too small (%lld < %lld)", (long long)workspace_bytes,
              (long long)ws_bytes(N, B, fdim, ldim));
  hipStream_t st = (hipStream_t)stream;
  int64_t tot = 0;
  int* offs_h = (int*)alloca(sizeof(int) * (B + 1));
  for (int b = 0; b < B; ++b) {
    MVK_REQUIRE(lens_host[b] >= 0, "subsample: negative batch length");
    offs_h[b] = (int)tot;
    tot += lens_host[b];
  }
  offs_h[B] = (int)tot;
  MVK_REQUIRE(tot == N, "subsample: batch lengths sum to %lld, not N=%lld", (long long)tot, (long long)N);
  if (max_p < 1) max_p = (int)(N > 0 ? N : 1);  // grid_subsampling.cpp:134-135

  const int64_t n = N > 0 ? N : 1;
  Carver cv{(char*)workspace, (char*)workspace + workspace_bytes};
  SubWs W{};
  W.slot = cv.take<int>(n); W.member = cv.take<int>(n); W.scan = cv.take<int>(n);
  W.hkey = cv.take<unsigned long long>(4 * n); W.hfirst = cv.take<int>(4 * n); W.hvox = cv.take<int>(4 * n);
  W.vkey = cv.take<unsigned long long>(n); W.vcount = cv.take<int>(n); W.vseg = cv.take<int>(n);
  W.vcursor = cv.take<int>(n); W.vbary = cv.take<float>(3 * n);
  W.vfeat = cv.take<float>(n * (fdim > 0 ? fdim : 0) + 1);
  W.vlab = cv.take<int>(n * (ldim > 0 ? ldim : 0) + 1);
  W.lab_key = cv.take<int>(n); W.lab_cnt = cv.take<int>(n); W.errflag = cv.take<int>(1);
  W.tau = cv.take<int>(n); W.posnew = cv.take<int>(n); W.nextb = cv.take<int>(n); W.tarr = cv.take<int>(n);
  W.ft = cv.take<int>(3 * n + 32 * B); W.bcnt = cv.take<int>(3 * n + 32 * B); W.bhead = cv.take<int>(3 * n + 32 * B);
  W.out_count = cv.take<int>(B + 1);
  int* offs_d = cv.take<int>(B + 1);
  W.stage_pts = cv.take<float>(3 * n);
  W.stage_feat = cv.take<float>(n * (fdim > 0 ? fdim : 0) + 1);
  W.stage_lab = cv.take<int>(n * (ldim > 0 ? ldim : 0) + 1);
  float* rot_pts = cv.take<float>(3 * n);
  W.bbox = cv.take<unsigned int>(6 * B); W.mdev = cv.take<int>(B); W.wgcnt = cv.take<int>(((n + 255) / 256 + 1) * B);
  MVK_REQUIRE(cv.p <= cv.end, "subsample: workspace carve overflow");
  if (rot_host) {   // random grid orientation (datasets/common.py:89-118): rotate every cloud by its matrix
    for (int b = 0; b < B; ++b) {
      if (lens_host[b] == 0) continue;
      Rot3 R;
      for (int i = 0; i < 9; ++i) R.m[i] = rot_host[b * 9 + i];
      hipLaunchKernelGGL(rotate_cloud_kernel, dim3((unsigned)cdiv64(lens_host[b], 256)), dim3(256), 0, st, pts, rot_pts,
                         (int64_t)offs_h[b], (int64_t)lens_host[b], (const int32_t*)nullptr, b, R, 0);
    }
    pts = rot_pts;
  }

  // _Prime_rehash_policy::_M_need_rehash: first allocation 13 buckets, then
  // next_bkt(max(count + 2, 2 * nb)) = next_bkt(2 * nb) each time count reaches nb.
  Schedule sched;
  sched.nb[0] = 13;
  for (int e = 1; e < 48; ++e) sched.nb[e] = mvk_next_bkt(2 * sched.nb[e - 1]);
  MVK_CHECK_HIP(hipMemcpyAsync(offs_d, offs_h, sizeof(int) * (B + 1), hipMemcpyHostToDevice, st));
  MVK_CHECK_HIP(hipMemsetAsync(W.errflag, 0, sizeof(int), st));
  static const bool timing = getenv("MVK_SUB_TIMING") != nullptr;
  static long long* dbg_dev = nullptr;
  if (timing && B <= 64) {
    if (!dbg_dev) MVK_CHECK_HIP(hipMalloc(&dbg_dev, sizeof(long long) * 64 * 16));
    W.dbg = dbg_dev;
  }
  int64_t longest = 0;
  for (int b = 0; b < B; ++b) longest = lens_host[b] > longest ? lens_host[b] : longest;
  if (multi_min_points() > 0 && longest >= multi_min_points())
    launch_front_end(pts, feats, fdim, labels, ldim, offs_d, dl, W, B, longest, sched, st);
  hipLaunchKernelGGL(subsample_cloud_kernel, dim3(B), dim3(TPB), 0, st, pts, feats, fdim, labels, ldim, offs_d, dl, W,
                     B, sched);
  if (W.dbg) {
    long long h[16];
    MVK_CHECK_HIP(hipMemcpyAsync(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost, st));
    MVK_CHECK_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "subsample timing (cloud 0, n = %d, us):", lens_host[0]);
    static const char* names[9] = {"P1 minmax", "P2 hash", "P3 flags+scan", "P4 counts", "P5 offsets", "P6 members", "P7 sums",
                                   "P8 order", "P9 write"};
    for (int k = 0; k < 9; ++k) fprintf(stderr, " %s %.1f |", names[k], (h[k + 1] - h[k]) / 100.0);
    fprintf(stderr, " total %.1f\n", (h[9] - h[0]) / 100.0);
  }
  int gx = (int)cdiv64(n, 256 * (int64_t)B);
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(subsample_compact_kernel, dim3(gx, B), dim3(256), 0, st, offs_d, W, B, fdim, ldim, max_p,
                     out_pts, out_feats, out_labels, out_lens, (int64_t)n, (int*)nullptr);
  if (rot_host) {   // ... and the barycentres back by its transpose (:134), in place on the compacted output
    for (int b = 0; b < B; ++b) {
      if (lens_host[b] == 0) continue;
      Rot3 R;
      for (int i = 0; i < 9; ++i) R.m[i] = rot_host[b * 9 + i];
      hipLaunchKernelGGL(rotate_cloud_kernel, dim3((unsigned)cdiv64(lens_host[b], 256)), dim3(256), 0, st, out_pts, out_pts,
                         (int64_t)0, (int64_t)0, (const int32_t*)out_lens, b, R, 1);
    }
  }
  MVK_CHECK_HIP(hipGetLastError());
  if (out_lens_host)
    MVK_CHECK_HIP(hipMemcpyAsync(out_lens_host, out_lens, sizeof(int) * B, hipMemcpyDeviceToHost, st));
  int err = 0;
  MVK_CHECK_HIP(hipMemcpyAsync(&err, W.errflag, sizeof(int), hipMemcpyDeviceToHost, st));
  MVK_CHECK_HIP(hipStreamSynchronize(st));  // offs_h (stack) and out_lens_host must be settled on return
  MVK_REQUIRE(err == 0, "subsample: a voxel holds more than %d distinct labels", MAXLAB);
  return 0;
}
}  // namespace

extern "C" int mvk_grid_subsample_batch(const float* pts, int64_t N, const int32_t* lens_host, int B,
                                        const float* feats, int fdim, const int32_t* labels, int ldim,
                                        float dl, int max_p, float* out_pts, float* out_feats,
                                        int32_t* out_labels, int32_t* out_lens,
                                        int32_t* out_lens_host, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
  return subsample_run(pts, N, lens_host, B, nullptr, feats, fdim, labels, ldim, dl, max_p, out_pts, out_feats,
                       out_labels, out_lens, out_lens_host, workspace, workspace_bytes, stream);
}

extern "C" int mvk_grid_subsample_batch_oriented(const float* pts, int64_t N, const int32_t* lens_host, int B,
                                                 const float* rot_host, const float* feats, int fdim,
                                                 const int32_t* labels, int ldim, float dl, int max_p,
                                                 float* out_pts, float* out_feats, int32_t* out_labels,
                                                 int32_t* out_lens, int32_t* out_lens_host, void* workspace,
                                                 int64_t workspace_bytes, void* stream) {
  MVK_REQUIRE(rot_host != nullptr, "subsample: the oriented variant needs one rotation per cloud");
  return subsample_run(pts, N, lens_host, B, rot_host, feats, fdim, labels, ldim, dl, max_p, out_pts, out_feats,
                       out_labels, out_lens, out_lens_host, workspace, workspace_bytes, stream);
}

extern "C" int mvk_grid_subsample_batch_dev(const float* pts, int64_t cap_in, const int32_t* lens_dev, int B,
                                            const float* rot_dev, float dl, float* out_pts, int64_t out_cap,
                                            float pad_value, int32_t* out_lens_dev, int32_t* total_out_dev,
                                            int32_t* status_dev, void* workspace, int64_t workspace_bytes,
                                            void* stream) {
  MVK_REQUIRE(B >= 1 && cap_in >= 1 && cap_in < (1ll << 29) && out_cap >= 1, "subsample: bad sizes");
  MVK_REQUIRE(dl > 0.f, "subsample: sampleDl must be positive");
  MVK_REQUIRE(status_dev != nullptr && out_lens_dev != nullptr, "subsample: the device-lens variant needs a status word");
  MVK_REQUIRE(workspace != nullptr && workspace_bytes >= ws_bytes(cap_in, B, 0, 0), "subsample: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = cap_in;
  Carver cv{(char*)workspace, (char*)workspace + workspace_bytes};
  SubWs W{};
  W.slot = cv.take<int>(n); W.member = cv.take<int>(n); W.scan = cv.take<int>(n);
  W.hkey = cv.take<unsigned long long>(4 * n); W.hfirst = cv.take<int>(4 * n); W.hvox = cv.take<int>(4 * n);
  W.vkey = cv.take<unsigned long long>(n); W.vcount = cv.take<int>(n); W.vseg = cv.take<int>(n);
  W.vcursor = cv.take<int>(n); W.vbary = cv.take<float>(3 * n);
  W.vfeat = cv.take<float>(1);
  W.vlab = cv.take<int>(1);
  W.lab_key = cv.take<int>(n); W.lab_cnt = cv.take<int>(n); W.errflag = cv.take<int>(1);
  W.tau = cv.take<int>(n); W.posnew = cv.take<int>(n); W.nextb = cv.take<int>(n); W.tarr = cv.take<int>(n);
  W.ft = cv.take<int>(3 * n + 32 * B); W.bcnt = cv.take<int>(3 * n + 32 * B); W.bhead = cv.take<int>(3 * n + 32 * B);
  W.out_count = cv.take<int>(B + 1);
  cv.take<int>(B + 1);
  W.stage_pts = cv.take<float>(3 * n);
  W.stage_feat = cv.take<float>(1);
  W.stage_lab = cv.take<int>(1);
  float* rot_pts = cv.take<float>(3 * n);
  W.bbox = cv.take<unsigned int>(6 * B); W.mdev = cv.take<int>(B); W.wgcnt = cv.take<int>(((n + 255) / 256 + 1) * B);
  MVK_REQUIRE(cv.p <= cv.end, "subsample: workspace carve overflow");
  W.errflag = status_dev + 1;   // label overflow cannot happen without labels; shares the overflow word

  Schedule sched;
  sched.nb[0] = 13;
  for (int e = 1; e < 48; ++e) sched.nb[e] = mvk_next_bkt(2 * sched.nb[e - 1]);
  // two launches: the per-cloud kernel (offsets from the device lengths, orientation as its phase 0) and the finish
  W.lens_dev = lens_dev;
  W.rot_dev = rot_dev;
  W.rot_pts = rot_pts;
  if (multi_min_points() > 0 && cap_in >= multi_min_points())       // (fixed launch geometry: the capacity decides)
    launch_front_end(pts, nullptr, 0, nullptr, 0, nullptr, dl, W, B, cap_in, sched, st);
  hipLaunchKernelGGL(subsample_cloud_kernel, dim3(B), dim3(TPB), 0, st, pts, (const float*)nullptr, 0,
                     (const int32_t*)nullptr, 0, (const int*)nullptr, dl, W, B, sched);
  hipLaunchKernelGGL(subsample_finish_dev_kernel, dim3((unsigned)cdiv64(out_cap, 256)), dim3(256), 0, st, W, B, out_pts,
                     out_lens_dev, out_cap, pad_value, total_out_dev, status_dev + 1);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
