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
// group_points: out[b,c,n,k] = in[b,c,idx[b,n,k]]; one lane per output element, (n,k) fastest so
// index reads and output writes are coalesced; the gather itself is element granular by nature
// (channel-major feature maps), served by L2 / Infinity Cache.
// ---------------------------------------------------------------------------------------------
__global__ void group_points_fwd_kernel(const float* __restrict__ in, const int64_t* __restrict__ idx, int B,
                                        int C, int64_t N1, int64_t NK, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)B * C * NK) return;
  const int64_t e = t % NK;
  const int64_t bc = t / NK;
  const int64_t b = bc / C;
  const int64_t j = idx[b * NK + e];
  out[t] = (j >= 0 && j < N1) ? in[bc * N1 + j] : 0.f;
}

__global__ void group_points_bwd_kernel(const float* __restrict__ go, const int64_t* __restrict__ idx, int B,
                                        int C, int64_t N1, int64_t NK, float* __restrict__ gi) {
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
__global__ void fa_gather_kernel(const float* __restrict__ feat /* [nv,C,hw] */, const float* __restrict__ xyz /* [nv*hw,3] */,
                                 const int64_t* __restrict__ knn /* [np*k] */, const float* __restrict__ pts /* [np,3] */,
                                 int C, int nv, int64_t hw, int64_t npk, int k, float* __restrict__ X /* [C+4, npk] */) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npk) return;
  const int64_t p = knn[e];
  const bool ok = p >= 0 && p < nv * hw;
  const int64_t view = ok ? p / hw : 0, pix = ok ? p % hw : 0;
  const float* f = feat + view * C * hw + pix;
  for (int c = blockIdx.y; c < C; c += gridDim.y) X[(int64_t)c * npk + e] = ok ? f[(int64_t)c * hw] : 0.f;
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

extern "C" int mvk_fa_gather_fwd(const float* feature_2d, const float* image_xyz, const int64_t* knn,
                                 const float* points, int C, int nv, int64_t hw, int64_t np, int k, float* X,
                                 void* stream) {
  MVK_REQUIRE(C > 0 && nv > 0 && hw > 0 && np >= 0 && k > 0, "fa_gather: bad sizes");
  const int64_t npk = np * k;
  if (npk == 0) return 0;
  dim3 grid((unsigned)cdiv64(npk, 256), 8);
  hipLaunchKernelGGL(fa_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, feature_2d, image_xyz, knn, points, C, nv,
                     hw, npk, k, X);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
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

extern "C" int64_t mvk_knn_workspace(int64_t nq, int64_t nk, int k) {
  int nsplit;
  int64_t per;
  knn_plan(nq, nk, &nsplit, &per);
  return (int64_t)nsplit * (nq > 0 ? nq : 1) * (k <= 3 ? 3 : 8) * 16 + 64;
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
  hipLaunchKernelGGL(group_points_fwd_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     points, index, B, C, N1, N2 * K, out);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_group_points_bwd(const float* grad_out, const int64_t* index, int B, int C, int64_t N1,
                                    int64_t N2, int K, float* grad_in, void* stream) {
  const int64_t tot = (int64_t)B * C * N2 * K;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(group_points_bwd_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out, index, B, C, N1, N2 * K, grad_in);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
