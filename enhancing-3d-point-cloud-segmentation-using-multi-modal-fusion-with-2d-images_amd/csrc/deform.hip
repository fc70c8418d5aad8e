// Deformable KPConv: gradient of the kernel-point offsets, and the offset regulariser.
//
// (1) d_offsets. With A[n,k,c] = sum_h w[n,h,k] x[j_h,c] and w a function of (s[j_h] - q[n]) - (kp[k] + off[n,k])
//     (reference KPConv-PyTorch/models/blocks.py:283-344, deformable branch :286-327), autograd gives
//         d_off[n,k,:] = sum_h (dw[n,h,k] / d off[n,k,:]) * B[n,h,k],   B[n,h,k] = sum_c x[j_h,c] dA[n,k,c]
//     plus, through min_d2 (blocks.py:303, read by the regulariser), -2 (rel[h*] - kpdef[k]) g_min[n,k] at the arg-min
//     neighbour h*. Round 1 accumulated sum_h x[j_h,c] dw/doff per CHANNEL lane (45 registers per 64 channels, one
//     launch per 64 channels, atomics): the slow lane of the deformable networks (4.3 of 24 ms per step).
//     Here: one wave per query point, lane = NEIGHBOUR. Pass 1 walks all H entries (in-range filter blocks.py:306-325)
//     and compacts the kept neighbours into an LDS list (the arg-min column of the min_d2 path comes from the forward); pass 2 takes 64 kept neighbours at a
//     time: every lane streams ITS neighbour's feature row (64 rows in flight per wave) against the point's dA block
//     staged in LDS (broadcast reads) -> B[15] per lane, then 45 offset-gradient accumulators; one wave reduction per
//     point at the end, plain stores (the wave owns d_off[n]). 3x fewer flops than the per-channel form, no atomics,
//     one launch whatever Cin is.
//
// (2) p2p_fitting_regularizer (models/architectures.py:20-58): per deformable layer
//         power * ( 2 * mean_{n,k} min_d2 / ext^2  +  sum_i mean_n sum_{j != i} clamp_max(|loc_i - sg(loc_j)| - R, 0)^2 / K )
//     with loc = deformed_KP / ext. One lane per point evaluates the 15 x 14 pairs, the launch also writes both
//     gradients (they do not depend on anything upstream), so the ~50 tensor ops per layer become one launch.
#include <stdlib.h>

#include "common.h"

#define DKMAX 16

namespace {

constexpr int DOFF_LIST = 2048;   // most neighbour columns one point can keep (rows of the neighbour matrix are narrower)

struct DoffParams {
  const float* q;
  const float* s;
  const void* idx;
  const float* x;
  const float* kp;
  const float* offsets;   // [Nq,K,3]
  const float* dA;        // [Nq,K,Cin]
  const float* g_min_d2;  // [Nq,K] or null
  const int32_t* min_arg; // [Nq,K] column of the arg-min entry (forward), with g_min_d2
  float* d_offsets;       // [Nq,K,3] out
  int64_t Nq, Ns;
  int H, Cin, K;
  float extent;
  int influence;
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WPB waves per point (4 when the neighbour rows are wider than one 64-column chunk, i.e. at the deform radius):
// wave w walks the chunks w, w + WPB, ... of pass 1 into ITS list; the lists are then read as ONE sequence (region
// offsets through LDS) of which wave w takes the 64-entry pieces w, w + WPB, ... in pass 2 (full lanes: a point that
// keeps 60 neighbours is one piece, not four quarter-filled ones), and the WPB x 45 partial gradients meet in LDS
// (one wave per point was a serial chain over 400+ columns on a third of the SIMDs).
template <bool IDX64, int WPB>
__global__ __launch_bounds__(64 * WPB) void kpconv_deform_doff(const DoffParams P, int list_cap) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // [K][Cin4] dA block of this point, the lists, the partials
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t n = blockIdx.x;
  const int Cin4 = (P.Cin + 3) & ~3;
  float* dA_l = dsm;
  int* list = reinterpret_cast<int*>(dsm + (size_t)DKMAX * Cin4) + (size_t)wid * list_cap;
  float* red = dsm + (size_t)DKMAX * Cin4 + (size_t)WPB * list_cap;          // [WPB][48]
  constexpr int NT = 64 * WPB;
  const int tid = threadIdx.x;

  // stage dA[n] (zero padded to a multiple of 4 channels)
  if ((P.Cin & 3) == 0) {
    const float4* src = reinterpret_cast<const float4*>(P.dA + n * P.K * P.Cin);     // K*Cin contiguous floats, 16-byte aligned
    float4* dst = reinterpret_cast<float4*>(dA_l);
    const int n4 = P.K * P.Cin / 4;
#pragma unroll 4
    for (int e = tid; e < n4; e += NT) dst[e] = src[e];
    for (int e = n4 + tid; e < (DKMAX - 1) * Cin4 / 4; e += NT) dst[e] = make_float4(0.f, 0.f, 0.f, 0.f);   // rows k >= K
  } else {
    for (int e = P.K * Cin4 + tid; e < (DKMAX - 1) * Cin4; e += NT) dA_l[e] = 0.f;
    for (int e = tid; e < P.K * Cin4; e += NT) {
      const int kk = e / Cin4, c = e - kk * Cin4;
      dA_l[e] = c < P.Cin ? P.dA[(n * P.K + kk) * P.Cin + c] : 0.f;
    }
  }
  const float qx = P.q[n * 3], qy = P.q[n * 3 + 1], qz = P.q[n * 3 + 2];
  float kx[DKMAX - 1], ky[DKMAX - 1], kz[DKMAX - 1];   // deformed kernel points (wave uniform), blocks.py:287
#pragma unroll
  for (int kk = 0; kk < DKMAX - 1; ++kk) {
    kx[kk] = ky[kk] = kz[kk] = 1e9f;      // kernel points beyond K: out of every range, zero weight, zero dA row
    if (kk < P.K) {
      const float* o = P.offsets + (n * P.K + kk) * 3;
      kx[kk] = P.kp[kk * 3] + o[0];
      ky[kk] = P.kp[kk * 3 + 1] + o[1];
      kz[kk] = P.kp[kk * 3 + 2] + o[2];
    }
  }
  const float ext2 = P.extent * P.extent;
  float dOff[(DKMAX - 1) * 3];
#pragma unroll
  for (int e = 0; e < (DKMAX - 1) * 3; ++e) dOff[e] = 0.f;

  // ---- pass 1: all H entries -> list of the kept neighbours (real neighbours within the extent of some deformed
  //      kernel point, blocks.py:306-325). The index of the next chunk is loaded one chunk ahead.
  int nkept = 0;
  int jn = wid * 64 + lane < P.H ? load_idx<IDX64>(P.idx, n * P.H + wid * 64 + lane, P.Ns) : -2;
  for (int h0 = wid * 64; h0 < P.H; h0 += 64 * WPB) {
    const int j = jn;
    jn = h0 + 64 * WPB + lane < P.H ? load_idx<IDX64>(P.idx, n * P.H + h0 + 64 * WPB + lane, P.Ns) : -2;
    bool keep = false;
    if (j >= 0) {
      const float* sp = P.s + (int64_t)j * 3;
      const float rx = sp[0] - qx, ry = sp[1] - qy, rz = sp[2] - qz;
#pragma unroll
      for (int kk = 0; kk < DKMAX - 1; ++kk) {
        const float dx = rx - kx[kk], dy = ry - ky[kk], dz = rz - kz[kk];
        keep = keep || (dx * dx + dy * dy + dz * dz < ext2);
      }
    }
    const unsigned long long bal = __ballot(keep);
    if (keep) {
      const int pos = nkept + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      if (pos < list_cap) list[pos] = j;
    }
    nkept += __builtin_popcountll(bal);
  }
  nkept = nkept < list_cap ? nkept : list_cap;
  // ---- min_d2 path: d min_d2[n,k] / d off[n,k,:] = -2 (rel[h*] - kpdef[k]) at the forward's arg-min column h*
  //      (shadow entries included, like torch.min over dim 1); lane k carries kernel point k's term
  if (P.g_min_d2 != nullptr && wid == 0 && lane < P.K && P.H > 0) {
    const int hs = P.min_arg[n * P.K + lane];
    const int j = load_idx<IDX64>(P.idx, n * P.H + hs, P.Ns);
    float rx = 1e6f - qx, ry = 1e6f - qy, rz = 1e6f - qz;
    if (j >= 0) {
      const float* sp = P.s + (int64_t)j * 3;
      rx = sp[0] - qx; ry = sp[1] - qy; rz = sp[2] - qz;
    }
    const float g = -2.f * P.g_min_d2[n * P.K + lane];
#pragma unroll
    for (int kk = 0; kk < DKMAX - 1; ++kk) {
      if (kk == lane) {
        dOff[kk * 3 + 0] += g * (rx - kx[kk]);
        dOff[kk * 3 + 1] += g * (ry - ky[kk]);
        dOff[kk * 3 + 2] += g * (rz - kz[kk]);
      }
    }
  }
  int pre[WPB + 1];        // kept entries before wave w's list in the common sequence
  pre[0] = 0;
  if (WPB > 1) {
    if (lane == 0) red[wid] = __int_as_float(nkept);
    __syncthreads();      // the dA block was staged by all waves; every list and count is complete
#pragma unroll
    for (int w = 0; w < WPB; ++w) pre[w + 1] = pre[w] + __float_as_int(red[w]);
    __syncthreads();      // (red is reused for the partial gradients)
  } else {
    pre[WPB] = nkept;
    wave_lds_sync();
  }
  const int total = pre[WPB];
  const int* lists = reinterpret_cast<const int*>(dsm + (size_t)DKMAX * Cin4);

  // ---- pass 2: 64 kept neighbours at a time, lane = neighbour
  for (int t0 = wid * 64; t0 < total; t0 += 64 * WPB) {
    const bool on = t0 + lane < total;
    int j = 0;
    if (on) {
      const int gi = t0 + lane;
      int r = 0;
#pragma unroll
      for (int w = 1; w < WPB; ++w) r += gi >= pre[w] ? 1 : 0;
      int base = 0;
#pragma unroll
      for (int w = 1; w < WPB; ++w) base = r >= w ? pre[w] : base;
      j = lists[(size_t)r * list_cap + (gi - base)];
    }
    float B[DKMAX - 1];
#pragma unroll
    for (int kk = 0; kk < DKMAX - 1; ++kk) B[kk] = 0.f;
    const float* xr = P.x + (int64_t)j * P.Cin;
    if ((P.Cin & 3) == 0) {
      // 8 float4 of the lane's feature row in flight (the row loads are the long-latency part: 64 different rows
      // per wave instruction), then their 15 x 4 FMAs against the broadcast dA block
      constexpr int UB = 8;
      for (int c0 = 0; c0 < P.Cin; c0 += 4 * UB) {
        float4 xv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u)
          xv[u] = (on && c0 + 4 * u < P.Cin) ? *reinterpret_cast<const float4*>(xr + c0 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int c = c0 + 4 * u;
          if (c < P.Cin) {          // wave uniform
#pragma unroll
            for (int kk = 0; kk < DKMAX - 1; ++kk) {
              const float4 d = *reinterpret_cast<const float4*>(dA_l + kk * Cin4 + c);   // broadcast read
              B[kk] = fmaf(xv[u].x, d.x, fmaf(xv[u].y, d.y, fmaf(xv[u].z, d.z, fmaf(xv[u].w, d.w, B[kk]))));
            }
          }
        }
      }
    } else {
      for (int c = 0; c < P.Cin; ++c) {
        const float xv = on ? xr[c] : 0.f;
#pragma unroll
        for (int kk = 0; kk < DKMAX - 1; ++kk) B[kk] = fmaf(xv, dA_l[kk * Cin4 + c], B[kk]);
      }
    }
    if (on) {
      const float* sp = P.s + (int64_t)j * 3;
      const float rx = sp[0] - qx, ry = sp[1] - qy, rz = sp[2] - qz;
#pragma unroll
      for (int kk = 0; kk < DKMAX - 1; ++kk) {
        const float dx = rx - kx[kk], dy = ry - ky[kk], dz = rz - kz[kk];
        const float d2 = dx * dx + dy * dy + dz * dz;
        // d w / d off[k,:] = sc * (rel - kpdef):  linear  w = 1 - sqrt(d2)/ext (w > 0): sc = 1 / (ext sqrt(d2));
        //                                         gaussian w = exp(-d2/den):            sc = 2 w / den;  constant: 0
        float sc = 0.f;
        if (P.influence == MVK_INFL_LINEAR) {
          const float dist = sqrtf(d2);
          sc = (1.0f - dist / P.extent > 0.f && d2 > 0.f) ? 1.0f / (P.extent * dist) : 0.f;
        } else if (P.influence == MVK_INFL_GAUSSIAN) {
          const float sig = P.extent * 0.3f, den = 2.0f * sig * sig + 1e-9f;
          sc = 2.0f * expf(-d2 / den) / den;
        }
        const float f = sc * B[kk];
        dOff[kk * 3 + 0] = fmaf(f, dx, dOff[kk * 3 + 0]);
        dOff[kk * 3 + 1] = fmaf(f, dy, dOff[kk * 3 + 1]);
        dOff[kk * 3 + 2] = fmaf(f, dz, dOff[kk * 3 + 2]);
      }
    }
  }

  // ---- one reduction over the 64 lanes per component (then over the waves, in wave order), plain stores
#pragma unroll
  for (int e = 0; e < (DKMAX - 1) * 3; ++e) {
    float v = dOff[e];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (WPB == 1) {
      if (lane == 0 && e < P.K * 3) P.d_offsets[n * P.K * 3 + e] = v;
    } else if (lane == 0) {
      red[wid * 48 + e] = v;
    }
  }
  if (WPB > 1) {
    __syncthreads();
    if (wid == 0 && lane < P.K * 3) {
      float v = red[lane];
#pragma unroll
      for (int w = 1; w < WPB; ++w) v += red[w * 48 + lane];
      P.d_offsets[n * P.K * 3 + lane] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same gradient with B on the matrix pipe (round 5; Cin % 4 == 0). One workgroup of four waves per query point.
// Pass 1 is the column walk above (lane = neighbour column, kept neighbours compacted into per-wave LDS lists read as
// one sequence). Pass 2 takes the kept neighbours 16 at a time, wave w the tiles w, w + 4, ...:
//     B[16 neighbours x 16 kernel points] = X[16 x Cin] . dA[n]^T[Cin x 16]
// as Cin / 4 v_mfma_f32_16x16x4_f32: lane (i = lane & 15, kq = lane >> 4) loads the float4 x[j_i, c0 + 4 kq ..] (the A
// operand of four MFMAs: their contraction index stands for the channels c0 + 4 kq + r) and the float4
// dA[n, m = lane & 15, c0 + 4 kq ..] (the B operand of the same four; row m = 15 is zero): 16 channels of 16 feature
// rows per wave load, no LDS staging of dA (the block is 15 x Cin floats read by <= 4 waves: L1 / L2 hits), no
// per-lane accumulator rows. The result lands as D[neighbour 4 kq + r, kernel point m]: the lane holds ITS kernel
// point's three offset-gradient sums over the neighbours 4 kq + r of every tile (3 accumulators instead of 45), the
// neighbours' relative positions come from the lanes that loaded them (ds_bpermute), two xor-shuffles fold kq at the
// end, the four waves meet in LDS in wave order (a fixed summation order: results do not depend on timing).
typedef float doff_f32x4 __attribute__((ext_vector_type(4)));

template <bool IDX64>
__global__ __launch_bounds__(256) void kpconv_deform_doff_mfma(const DoffParams P, int list_cap) {
  constexpr int WPB = 4;
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // [WPB][list_cap] lists, [WPB][48] partials, [16] float4 kernel points
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int64_t n = blockIdx.x;
  int* lists = reinterpret_cast<int*>(dsm);
  int* list = lists + (size_t)wid * list_cap;
  float* red = dsm + (size_t)WPB * list_cap;                     // [WPB][48]
  float4* kd = reinterpret_cast<float4*>(red + WPB * 48);        // deformed kernel points (blocks.py:287), 1e9 beyond K
  const float qx = P.q[n * 3], qy = P.q[n * 3 + 1], qz = P.q[n * 3 + 2];
  float mx = 1e9f, my = 1e9f, mz = 1e9f;                         // this lane's kernel point m
  if (m < P.K) {
    const float* o = P.offsets + (n * P.K + m) * 3;
    mx = P.kp[m * 3] + o[0];
    my = P.kp[m * 3 + 1] + o[1];
    mz = P.kp[m * 3 + 2] + o[2];
  }
  if (threadIdx.x < 16) kd[threadIdx.x] = make_float4(mx, my, mz, 0.f);
  __syncthreads();
  const float ext2 = P.extent * P.extent;

  // ---- pass 1 (as in kpconv_deform_doff): all H entries -> lists of the kept neighbours
  int nkept = 0;
  int jn = wid * 64 + lane < P.H ? load_idx<IDX64>(P.idx, n * P.H + wid * 64 + lane, P.Ns) : -2;
  for (int h0 = wid * 64; h0 < P.H; h0 += 64 * WPB) {
    const int j = jn;
    jn = h0 + 64 * WPB + lane < P.H ? load_idx<IDX64>(P.idx, n * P.H + h0 + 64 * WPB + lane, P.Ns) : -2;
    bool keep = false;
    if (j >= 0) {
      const float* sp = P.s + (int64_t)j * 3;
      const float rx = sp[0] - qx, ry = sp[1] - qy, rz = sp[2] - qz;
#pragma unroll
      for (int kk = 0; kk < DKMAX - 1; ++kk) {
        const float4 k4 = kd[kk];                                 // broadcast read
        const float dx = rx - k4.x, dy = ry - k4.y, dz = rz - k4.z;
        keep = keep || (dx * dx + dy * dy + dz * dz < ext2);
      }
    }
    const unsigned long long bal = __ballot(keep);
    if (keep) {
      const int pos = nkept + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      if (pos < list_cap) list[pos] = j;
    }
    nkept += __builtin_popcountll(bal);
  }
  nkept = nkept < list_cap ? nkept : list_cap;
  int pre[WPB + 1];
  pre[0] = 0;
  if (lane == 0) red[wid] = __int_as_float(nkept);
  __syncthreads();
#pragma unroll
  for (int w = 0; w < WPB; ++w) pre[w + 1] = pre[w] + __float_as_int(red[w]);
  __syncthreads();
  const int total = pre[WPB];

  // ---- pass 2: tiles of 16 kept neighbours
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const float* dArow = P.dA + (n * P.K + (m < P.K ? m : 0)) * P.Cin + 4 * kq;
  const bool m_on = m < P.K;
  for (int t0 = wid * 16; t0 < total; t0 += 16 * WPB) {
    const int gi = t0 + m;
    const bool on = gi < total;
    int j = 0;
    if (on) {
      int r = 0;
#pragma unroll
      for (int w = 1; w < WPB; ++w) r += gi >= pre[w] ? 1 : 0;
      int base = 0;
#pragma unroll
      for (int w = 1; w < WPB; ++w) base = r >= w ? pre[w] : base;
      j = lists[(size_t)r * list_cap + (gi - base)];
    }
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (on) {
      const float* sp = P.s + (int64_t)j * 3;
      rx = sp[0] - qx; ry = sp[1] - qy; rz = sp[2] - qz;
    }
    const float* xr = P.x + (int64_t)j * P.Cin + 4 * kq;
    doff_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int UB = 4;                                         // 4 + 4 float4 loads in flight
    for (int c0 = 0; c0 < P.Cin; c0 += 16 * UB) {
      float4 xv[UB], dv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const bool cin = c0 + 16 * u + 4 * kq < P.Cin;
        xv[u] = (on && cin) ? *reinterpret_cast<const float4*>(xr + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        dv[u] = (m_on && cin) ? *reinterpret_cast<const float4*>(dArow + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (c0 + 16 * u < P.Cin) {        // wave uniform
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].x, dv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].y, dv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].z, dv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].w, dv[u].w, acc, 0, 0, 0);
        }
      }
    }
    // D[row 4 kq + r, col m]: the neighbours 4 kq + r of this tile against kernel point m
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int src = 4 * kq + r;
      const float nx = __shfl(rx, src), ny = __shfl(ry, src), nz = __shfl(rz, src);
      const bool von = t0 + src < total;
      const float dx = nx - mx, dy = ny - my, dz = nz - mz;
      const float d2 = dx * dx + dy * dy + dz * dz;
      // d w / d off[k,:] = sc * (rel - kpdef):  linear  w = 1 - sqrt(d2)/ext (w > 0): sc = 1 / (ext sqrt(d2));
      //                                         gaussian w = exp(-d2/den):            sc = 2 w / den;  constant: 0
      float sc = 0.f;
      if (P.influence == MVK_INFL_LINEAR) {
        const float dist = sqrtf(d2);
        sc = (1.0f - dist / P.extent > 0.f && d2 > 0.f) ? 1.0f / (P.extent * dist) : 0.f;
      } else if (P.influence == MVK_INFL_GAUSSIAN) {
        const float sig = P.extent * 0.3f, den = 2.0f * sig * sig + 1e-9f;
        sc = 2.0f * expf(-d2 / den) / den;
      }
      const float f = (von && m_on) ? sc * acc[r] : 0.f;
      gx = fmaf(f, dx, gx);
      gy = fmaf(f, dy, gy);
      gz = fmaf(f, dz, gz);
    }
  }
  // fold kq (lanes m, m + 16, m + 32, m + 48), then the waves in wave order
  gx += __shfl_xor(gx, 16); gy += __shfl_xor(gy, 16); gz += __shfl_xor(gz, 16);
  gx += __shfl_xor(gx, 32); gy += __shfl_xor(gy, 32); gz += __shfl_xor(gz, 32);
  if (lane < 16) {
    red[wid * 48 + lane * 3 + 0] = gx;
    red[wid * 48 + lane * 3 + 1] = gy;
    red[wid * 48 + lane * 3 + 2] = gz;
  }
  __syncthreads();
  if (wid == 0 && lane < P.K * 3) {
    float v = red[lane];
#pragma unroll
    for (int w = 1; w < WPB; ++w) v += red[w * 48 + lane];
    // ---- min_d2 path: d min_d2[n,k] / d off[n,k,:] = -2 (rel[h*] - kpdef[k]) at the forward's arg-min column h*
    //      (shadow entries included, like torch.min over dim 1)
    if (P.g_min_d2 != nullptr && P.H > 0) {
      const int k = lane / 3, e = lane - 3 * k;
      const int hs = P.min_arg[n * P.K + k];
      const int j = load_idx<IDX64>(P.idx, n * P.H + hs, P.Ns);
      const float qe = e == 0 ? qx : (e == 1 ? qy : qz);
      const float rel = j >= 0 ? P.s[(int64_t)j * 3 + e] - qe : 1e6f - qe;
      const float4 k4 = kd[k];
      const float ke = e == 0 ? k4.x : (e == 1 ? k4.y : k4.z);
      v += -2.f * P.g_min_d2[n * P.K + k] * (rel - ke);
    }
    P.d_offsets[n * P.K * 3 + lane] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// regulariser: one lane per (point, kernel point): 16-lane groups hold one point's deformed kernel points and
// exchange them by shuffles
template <bool ATOMIC = true>
__device__ __forceinline__ float deform_regularizer_body(const int64_t block, const float* __restrict__ min_d2,
                                                        const float* __restrict__ dkp, const int32_t* __restrict__ n_valid,
                                                        int64_t N, int K, float extent, float repulse, float power,
                                                        float* __restrict__ loss /* [1], += ; or null */,
                                                        const float* __restrict__ gscale /* [1] or null: 1 */,
                                                        float* __restrict__ d_min_d2 /* [N,K] or null */,
                                                        float* __restrict__ d_dkp /* [N,K,3] or null */) {
  __shared__ float red[4];
  const int64_t t = block * blockDim.x + threadIdx.x;
  const int64_t n = t >> 4;
  const int i = (int)(t & 15);
  int64_t nv = n_valid ? (int64_t)*n_valid : N;
  nv = nv < N ? nv : N;
  const bool on = n < N && i < K;
  const bool valid = on && n < nv;
  const float inv_e = 1.0f / extent;
  const float cf = valid ? power * 2.0f / (extent * extent * (float)nv * (float)K) : 0.f;   // d loss / d min_d2
  const float cr = valid ? power / ((float)nv * (float)K) : 0.f;                                // weight of one (point, i) pair sum
  const float gsc = gscale ? gscale[0] : 1.f;     // upstream gradient of the loss term (backward launch)
  float lx = 0.f, ly = 0.f, lz = 0.f, part = 0.f;
  if (on) {
    const float* p = dkp + (n * K + i) * 3;
    lx = p[0] * inv_e; ly = p[1] * inv_e; lz = p[2] * inv_e;             // KP_locs (architectures.py:44)
    const float m = min_d2[n * K + i];
    if (d_min_d2) d_min_d2[n * K + i] = gsc * (m > 0.f ? cf : (m < 0.f ? -cf : 0.f));   // L1 to zero: sign(m)
    part = cf * fabsf(m);
  }
  float gx = 0.f, gy = 0.f, gz = 0.f, acc = 0.f;
#pragma unroll
  for (int j = 0; j < DKMAX - 1; ++j) {
    const float ox = __shfl(lx, j, 16), oy = __shfl(ly, j, 16), oz = __shfl(lz, j, 16);
    if (j < K && j != i) {
      const float dx = lx - ox, dy = ly - oy, dz = lz - oz;
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      const float c = fminf(d - repulse, 0.f);           // clamp_max(dist - repulse_extent, 0)  (:52)
      acc += c * c;
      const float s = 2.f * c / d;                        // d c^2 / d loc_i  (the other point is detached, :49)
      gx += s * dx; gy += s * dy; gz += s * dz;
    }
  }
  if (on) {
    part += cr * acc;
    if (d_dkp) {
      float* o = d_dkp + (n * K + i) * 3;
      const float f = gsc * cr * inv_e;
      o[0] = f * gx; o[1] = f * gy; o[2] = f * gz;
    }
  }
  if (loss == nullptr) return 0.f;          // (uniform)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const float sum = (red[0] + red[1]) + (red[2] + red[3]);
  if (ATOMIC && threadIdx.x == 0) atomicAdd(loss, sum);
  return sum;           // (!ATOMIC: the caller adds the workgroup sums itself and keeps `red` apart between two calls)
}

__global__ __launch_bounds__(256) void deform_regularizer_kernel(const float* __restrict__ min_d2, const float* __restrict__ dkp,
                                                                const int32_t* __restrict__ n_valid, int64_t N, int K,
                                                                float extent, float repulse, float power,
                                                                float* __restrict__ loss, const float* __restrict__ gscale,
                                                                float* __restrict__ d_min_d2, float* __restrict__ d_dkp) {
  deform_regularizer_body(blockIdx.x, min_d2, dkp, n_valid, N, K, extent, repulse, power, loss, gscale, d_min_d2, d_dkp);
}

// every deformable layer of a network in ONE launch (ten launches of a few hundred rows each were ten launch
// boundaries on the step's chain, twice per step): blockIdx.y = layer, the blocks beyond a layer's rows leave
struct RegManyArgs {
  mvk_reg_layer l[MVK_REG_MANY];
};
// the ordered mode (mvk_gemm_split_ordered): ONE workgroup walks every layer and every 16-point block in order and adds
// the block sums itself -- the loss has the same bits from run to run (float atomics add in order of arrival)
__global__ __launch_bounds__(256) void deform_regularizer_serial_kernel(const RegManyArgs a, int n, int K, float* __restrict__ loss,
                                                                       const float* __restrict__ gscale) {
  float total = 0.f;
  for (int k = 0; k < n; ++k) {
    const mvk_reg_layer& L = a.l[k];
    for (int64_t b = 0; b * 16 < L.N; ++b) {
      total += deform_regularizer_body<false>(b, L.min_d2, L.deformed_kp, L.n_valid, L.N, K, L.extent, L.repulse_extent, L.power,
                                              loss, gscale, L.d_min_d2, L.d_deformed_kp);
      __syncthreads();
    }
  }
  if (loss != nullptr && threadIdx.x == 0) loss[0] += total;
}

__global__ __launch_bounds__(256) void deform_regularizer_many_kernel(const RegManyArgs a, int K, float* __restrict__ loss,
                                                                     const float* __restrict__ gscale) {
  const mvk_reg_layer& L = a.l[blockIdx.y];
  if ((int64_t)blockIdx.x * 16 >= L.N) return;          // (uniform)
  deform_regularizer_body(blockIdx.x, L.min_d2, L.deformed_kp, L.n_valid, L.N, K, L.extent, L.repulse_extent, L.power, loss,
                          gscale, L.d_min_d2, L.d_deformed_kp);
}

}  // namespace

extern "C" int mvk_deform_regularizer_many(const mvk_reg_layer* layers, int n, int K, float* loss_accum,
                                           const float* grad_scale, void* stream) {
  MVK_REQUIRE(n >= 0 && n <= MVK_REG_MANY && K >= 1 && K < DKMAX, "regulariser: at most %d layers per call", MVK_REG_MANY);
  RegManyArgs a{};
  int64_t most = 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const mvk_reg_layer& L = layers[i];
    MVK_REQUIRE(L.N >= 0 && L.extent > 0.f, "regulariser: bad sizes");
    if (L.N == 0) continue;          // an empty level contributes nothing (its tensors have no storage)
    MVK_REQUIRE(L.min_d2 && L.deformed_kp && (loss_accum || L.d_min_d2 || L.d_deformed_kp), "regulariser: null operand");
    a.l[m++] = L;
    most = L.N > most ? L.N : most;
  }
  if (m == 0) return 0;
  if (loss_accum != nullptr && mvk_gemm_split_ordered()) {
    hipLaunchKernelGGL(deform_regularizer_serial_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, m, K, loss_accum, grad_scale);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(deform_regularizer_many_kernel, dim3((unsigned)cdiv64(most * 16, 256), (unsigned)m), dim3(256), 0,
                     (hipStream_t)stream, a, K, loss_accum, grad_scale);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_kpconv_deform_doff(const float* q, int64_t Nq, const float* s, int64_t Ns, const void* idx,
                                      int idx64, int H, const float* x, int Cin, const float* kp, int K, float extent,
                                      int influence, const float* offsets, const float* dA,
                                      const float* g_min_d2, const int32_t* min_arg, float* d_offsets, void* stream) {
  MVK_REQUIRE(Nq >= 0 && Ns >= 0 && H >= 0 && Cin > 0 && K >= 1 && K < DKMAX, "deform d_offsets: bad sizes");
  MVK_REQUIRE(influence >= 0 && influence <= 2, "Unknown influence function type (config.KP_influence)");
  if (Nq == 0) return 0;
  MVK_REQUIRE(offsets && dA && d_offsets && x, "deform d_offsets: null operand");
  MVK_REQUIRE(!g_min_d2 || min_arg, "deform d_offsets: the min_d2 gradient needs the forward's arg-min columns");
  MVK_REQUIRE(H <= DOFF_LIST, "deform d_offsets: neighbour rows wider than %d columns", DOFF_LIST);
  MVK_REQUIRE(Nq < (1ll << 31), "deform d_offsets: Nq too large for one launch");
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DoffParams P{};
  P.q = q; P.s = s; P.idx = idx; P.x = x; P.kp = kp; P.offsets = offsets; P.dA = dA;
  P.g_min_d2 = g_min_d2; P.min_arg = min_arg; P.d_offsets = d_offsets; P.Nq = Nq; P.Ns = Ns; P.H = H; P.Cin = Cin; P.K = K;
  P.extent = extent; P.influence = influence;
  const int Cin4 = (Cin + 3) & ~3;
  static const bool mfma_on = getenv("MVK_DOFF_MFMA") == nullptr || atoi(getenv("MVK_DOFF_MFMA")) != 0;
  if (mfma_on && (Cin & 3) == 0) {
    const int chunks = (H + 63) / 64;
    const int list_cap = ((chunks + 3) / 4) * 64 > 64 ? ((chunks + 3) / 4) * 64 : 64;      // columns one wave walks
    const size_t lds = sizeof(int) * 4 * (size_t)list_cap + sizeof(float) * 4 * 48 + sizeof(float4) * 16;
    if (idx64) hipLaunchKernelGGL((kpconv_deform_doff_mfma<true>), dim3((unsigned)Nq), dim3(256), lds, st, P, list_cap);
    else hipLaunchKernelGGL((kpconv_deform_doff_mfma<false>), dim3((unsigned)Nq), dim3(256), lds, st, P, list_cap);
    MVK_CHECK_HIP(hipGetLastError());
    return 0;
  }
  static const bool split_on = getenv("MVK_DOFF_SPLIT") == nullptr || atoi(getenv("MVK_DOFF_SPLIT")) != 0;
  // four waves per point only where points are few (the coarse levels searched at the deform radius: 750 / 160 / 36
  // points x 420 / 349 / 124 columns): with thousands of points the chip is full anyway and the per-wave overhead
  // (kernel points, the 45-value reductions) costs more than the shorter chains save (4 000 x 200: 59 -> 183 us)
  const int wpb = (split_on && H > 64 && Nq <= 1024) ? 4 : 1;
  const int chunks = (H + 63) / 64;
  const int list_cap = ((chunks + wpb - 1) / wpb) * 64 > 64 ? ((chunks + wpb - 1) / wpb) * 64 : 64;   // columns one wave walks
  const size_t lds = sizeof(float) * (size_t)DKMAX * Cin4 + sizeof(int) * (size_t)wpb * list_cap + sizeof(float) * wpb * 48;
  MVK_REQUIRE(lds <= 160 * 1024, "deform d_offsets: Cin=%d does not fit the LDS staging", Cin);
#define DOFF_LAUNCH(I64, W)                                                                                              \
  {                                                                                                                      \
    if (lds > 64 * 1024)                                                                                                 \
      MVK_CHECK_HIP(hipFuncSetAttribute((const void*)kpconv_deform_doff<I64, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((kpconv_deform_doff<I64, W>), dim3((unsigned)Nq), dim3(64 * W), lds, st, P, list_cap);           \
  }
  if (idx64) {
    if (wpb == 4) DOFF_LAUNCH(true, 4) else DOFF_LAUNCH(true, 1)
  } else {
    if (wpb == 4) DOFF_LAUNCH(false, 4) else DOFF_LAUNCH(false, 1)
  }
#undef DOFF_LAUNCH
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// (3) From the inner convolution's output to the operands of the deformable one (blocks.py:243-266, :287) in one
// launch: feat = raw + bias; offsets[n,k,:] = feat[n, 3k..3k+2] * extent; deformed_KP = offsets + kernel_points;
// modulated: modulations[n,k] = 2 sigmoid(feat[n, 3K + k]). As tensor ops this is 3 (6) launches forward and as many
// backward per deformable layer (ten layers in the middle-fusion net).
namespace {
__global__ void deform_operands_fwd_k(const float* __restrict__ raw, const float* __restrict__ bias,
                                      const float* __restrict__ kp, int64_t N, int K, int D, float extent,
                                      float* __restrict__ feat, float* __restrict__ offsets, float* __restrict__ dkp,
                                      float* __restrict__ mod) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const int64_t n = t / D;
  const int d = (int)(t - n * D);
  const float f = raw[t] + bias[d];
  feat[t] = f;
  if (d < 3 * K) {
#pragma clang fp contract(off)                        // rounded product, then rounded sum: the reference's two tensor ops
    const float o = f * extent;
    offsets[n * 3 * K + d] = o;
    dkp[n * 3 * K + d] = o + kp[d];
  } else {
    mod[n * K + (d - 3 * K)] = 2.0f / (1.0f + expf(-f));
  }
}

// d_raw[n,d] = (g_off + g_dkp)[n,d] * extent (d < 3K) | g_mod[n,k] * mod (1 - mod / 2) (d >= 3K: d/df of 2 sigmoid);
// d_bias[d] += column sums (f32 atomics onto a zero-initialised vector: one atomic per (workgroup, column))
__global__ __launch_bounds__(256) void deform_operands_bwd_k(const float* __restrict__ g_off, const float* __restrict__ g_dkp,
                                                            const float* __restrict__ g_mod, const float* __restrict__ mod,
                                                            const float* __restrict__ g_feat, int64_t N, int K, int D,
                                                            float extent, float* __restrict__ d_raw, float* __restrict__ d_bias) {
  // a workgroup owns 256 / DP rows x DP (= D rounded up to a power of two <= 64) columns
  __shared__ float part[256];
  int DP = 1;
  while (DP < D) DP <<= 1;
  const int rows_per = 256 / DP;
  const int d = threadIdx.x % DP, rl = threadIdx.x / DP;
  float acc = 0.f;
  for (int64_t n = (int64_t)blockIdx.x * rows_per + rl; n < N; n += (int64_t)gridDim.x * rows_per) {
    if (d < D) {
      float v;
      if (d < 3 * K) {
        v = 0.f;
        if (g_off) v += g_off[n * 3 * K + d];
        if (g_dkp) v += g_dkp[n * 3 * K + d];
        v *= extent;
      } else {
        const float m = mod[n * K + (d - 3 * K)];
        v = g_mod ? g_mod[n * K + (d - 3 * K)] * m * (1.0f - 0.5f * m) : 0.f;
      }
      if (g_feat) v += g_feat[n * D + d];
      d_raw[n * D + d] = v;
      acc += v;
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && d < D) {
    float s = 0.f;
    for (int r = 0; r < rows_per; ++r) s += part[r * DP + d];
    atomicAdd(d_bias + d, s);
  }
}
}  // namespace

extern "C" int mvk_deform_operands_fwd(const float* raw, const float* bias, const float* kernel_points, int64_t N, int K,
                                       int modulated, float extent, float* feat, float* offsets, float* deformed_kp,
                                       float* modulations, void* stream) {
  MVK_REQUIRE(N >= 0 && K >= 1 && K < DKMAX, "deform operands: bad sizes");
  if (N == 0) return 0;
  const int D = (modulated ? 4 : 3) * K;
  MVK_REQUIRE(raw && bias && kernel_points && feat && offsets && deformed_kp && (!modulated || modulations),
              "deform operands: null operand");
  hipLaunchKernelGGL(deform_operands_fwd_k, dim3((unsigned)cdiv64(N * D, 256)), dim3(256), 0, (hipStream_t)stream, raw, bias,
                     kernel_points, N, K, D, extent, feat, offsets, deformed_kp, modulations);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_deform_operands_bwd(const float* g_offsets, const float* g_deformed_kp, const float* g_modulations,
                                       const float* modulations, const float* g_feat, int64_t N, int K, int modulated,
                                       float extent, float* d_raw, float* d_bias /* [D], zero-initialised */, void* stream) {
  MVK_REQUIRE(N >= 0 && K >= 1 && K < DKMAX, "deform operands: bad sizes");
  if (N == 0) return 0;
  const int D = (modulated ? 4 : 3) * K;
  MVK_REQUIRE(d_raw && d_bias && (!modulated || modulations), "deform operands: null operand");
  int DP = 1;
  while (DP < D) DP <<= 1;
  const int64_t rows_per = 256 / DP;
  int64_t g = cdiv64(N, rows_per * 4);          // ~4 rows per thread
  g = g < 1 ? 1 : (g > 256 ? 256 : g);
  if (mvk_gemm_split_ordered()) g = 1;          // ordered mode: one workgroup, one (ordered) sum per column of d_bias
  hipLaunchKernelGGL(deform_operands_bwd_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, g_offsets, g_deformed_kp,
                     g_modulations, modulations, g_feat, N, K, D, extent, d_raw, d_bias);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_deform_regularizer_ex(const float* min_d2, const float* deformed_kp, const int32_t* n_valid, int64_t N,
                                         int K, float extent, float repulse_extent, float power, float* loss_accum,
                                         const float* grad_scale, float* d_min_d2, float* d_deformed_kp, void* stream) {
  MVK_REQUIRE(N >= 0 && K >= 1 && K < DKMAX && extent > 0.f, "regulariser: bad sizes");
  if (N == 0) return 0;          // an empty level contributes nothing (its tensors have no storage)
  MVK_REQUIRE(min_d2 && deformed_kp && (loss_accum || d_min_d2 || d_deformed_kp), "regulariser: null operand");
  hipLaunchKernelGGL(deform_regularizer_kernel, dim3((unsigned)cdiv64(N * 16, 256)), dim3(256), 0, (hipStream_t)stream, min_d2,
                     deformed_kp, n_valid, N, K, extent, repulse_extent, power, loss_accum, grad_scale, d_min_d2, d_deformed_kp);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int mvk_deform_regularizer(const float* min_d2, const float* deformed_kp, const int32_t* n_valid, int64_t N,
                                      int K, float extent, float repulse_extent, float power, float* loss_accum,
                                      float* d_min_d2, float* d_deformed_kp, void* stream) {
  if (N > 0) MVK_REQUIRE(loss_accum && d_min_d2 && d_deformed_kp, "regulariser: null operand");
  return mvk_deform_regularizer_ex(min_d2, deformed_kp, n_valid, N, K, extent, repulse_extent, power, loss_accum, nullptr,
                                   d_min_d2, d_deformed_kp, stream);
}
