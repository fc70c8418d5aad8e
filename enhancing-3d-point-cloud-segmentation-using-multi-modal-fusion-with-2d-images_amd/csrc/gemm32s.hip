// f32 streaming contraction for the big rigid layers (v_mfma_f32_16x16x4_f32, exact f32)
// for y [M, N] = A [M, Kd] . W [Kd, N] with M in the tens of thousands, N = 32 or 64 and
// Kd <= 1024 (reference shape contract models/blocks.py:370-374) -- the first layers of the network, where the tiled
// kernel of gemm.hip (every workgroup re-stages its slice of W through LDS behind a barrier per 32-deep step) reaches
// 0.36-0.47 of the f32 MFMA peak.
//   * one workgroup per CU of NW = 4 or 8 waves; wave w keeps the weight fragments of its 1 / NW of the reduction for
//     all N columns in registers for the whole launch (slices of at most 128 columns: 128 registers at N = 64 -- four
//     waves with 256-column slices need 256 registers of weights plus 128 of rows in flight, and the compiler spills);
//   * the rows of A stream past in 16-row tiles as 8-byte fragment loads straight from HBM (K*Cin = 990 floats: rows
//     are only 8-byte aligned; lane (row r, group g) takes k = 8 j + 2 g + {0, 1}: two MFMA steps per load, the same k
//     permutation on both operands), two tiles in flight per wave through two register sets;
//   * lane c owns the columns CT c .. CT c + CT - 1 (the "column tiles" of its accumulators), so a weight fragment for
//     all of them is one 16-byte load and the partial blocks go to LDS as 16-byte rows; the eight waves' partial blocks
//     are added in a fixed order (deterministic, no atomics) and stored as whole rows (hidden stores, see store_f4_hidden32);
//   * BatchNorm statistics of the output for the workgroup's rows in the same pass (gemm.hip's partials format).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct S32Args {
  const float* A;
  const float* B;
  float* C;
  int64_t M, lda;
  int N, Kd, ks, tiles_per_wg;
  const int* n_valid;
  float* bn_part;
};

// A 16-byte store the compiler does not see: global stores share the vmcnt counter with the loads and may retire out of
// order with them, so a store the compiler knows about between a load and its use turns the counted wait in front of
// that use into a wait for EVERYTHING in flight (measured: one drained pipeline per store phase). Hidden from its
// bookkeeping the waits stay counted; they are still sufficient: a wait for "at most k outstanding" then returns later
// than needed (the counter also holds the stores), never earlier -- loads return in order among themselves, so with s
// hidden stores and j older loads, <= k outstanding means at least j + 1 loads have returned.
__device__ __forceinline__ void store_f4_hidden32(float* p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v q = {v.x, v.y, v.z, v.w};
  // s_nop 2: the store reads its data registers late; a VALU write to them within two wait states would be stored
  // instead (the compiler pads its own wide stores, it cannot see into this one -- common.h, round 4)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" : : "v"(p), "v"(q) : "memory");
}

__device__ __forceinline__ void wave_lds_handoff() {   // LDS hand-off inside ONE wave (its own region)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CT, int KJ, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gemm_f32_stream(const S32Args a) {
  constexpr int N = 16 * CT, LDR = N + 4, G = 2;
  constexpr int KS = 8 * KJ;                    // columns of a wave's slice
  constexpr int LDA = KS + 12;                  // LDS row of the slice image: 140 floats at KS = 128 -- row r starts at bank
                                                // 12 r mod 64, sixteen distinct multiples of 4: the 8-byte fragment reads of
                                                // a half wave (16 rows x 2 lane groups) hit 64 distinct banks
  constexpr int LPR = KS / 2;                   // lanes that carry one row of the slice (8 bytes each): 64 at KS = 128
  constexpr int RPI = 64 / LPR;                 // rows per load instruction
  constexpr int NLD = 16 / RPI;                 // load instructions per 16-row tile
  __shared__ __attribute__((aligned(16))) float red[G][NW][16][LDR];
  __shared__ __attribute__((aligned(16))) float abuf[NW][16][LDA];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int k0 = w * KS;

  const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wg;
  const int64_t ntiles = (a.M + 15) / 16;
  const int T = (int)(ntiles - tile0 < a.tiles_per_wg ? ntiles - tile0 : a.tiles_per_wg);
  const int64_t last = tile0 + (T > 0 ? T - 1 : 0);

  // Staging identity: lane -> (row within the instruction's RPI rows, 8-byte column of the slice). The rows of A are only
  // 8-byte aligned (K*Cin = 990 floats), so a row slice travels as LPR x 8 bytes -- contiguous over the lanes, unlike a
  // fragment load (lane = row: 64 rows per instruction; the first version of this kernel loaded fragments directly
  // and was bound by the address unit: 39 us against 34 us for the tiled kernel on 19 464 x 990 x 64).
  const int srow = lane / LPR, scol = 2 * (lane % LPR);
  const bool s_in = k0 + scol + 1 < a.Kd;                   // columns beyond the reduction (last wave): zeros in the image
  const int64_t s_off = (int64_t)(s_in ? k0 + scol : (a.Kd >= 2 ? a.Kd - 2 : 0));     // a real address either way

  auto load_tile = [&](int64_t tile, float2 (&st)[NLD]) {
    tile = tile < last ? tile : last;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int64_t row = tile * 16 + i * RPI + srow;
      row = row < a.M ? row : a.M - 1;                     // clamped rows are never stored nor counted
      st[i] = *reinterpret_cast<const float2*>(a.A + row * a.lda + s_off);
    }
  };
  auto stage_tile = [&](const float2 (&st)[NLD]) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      *reinterpret_cast<float2*>(&abuf[w][i * RPI + srow][scol]) = s_in ? st[i] : make_float2(0.f, 0.f);
  };

  float2 st[NLD];
  load_tile(tile0, st);

  // stationary weights: b[2 j + t][ct] = W[k0 + 8 j + 2 g + t][CT c + ct], zero beyond the reduction
  float b[2 * KJ][CT];
#pragma unroll
  for (int j = 0; j < KJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = k0 + 8 * j + 2 * g + t;
      const int kc = k < a.Kd ? k : a.Kd - 1;
      const float* p = a.B + (int64_t)kc * N + c * CT;
      if (CT == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        b[2 * j + t][0] = v.x; b[2 * j + t][1] = v.y; b[2 * j + t][CT > 2 ? 2 : 0] = v.z; b[2 * j + t][CT > 2 ? 3 : 0] = v.w;
      } else {
        const float2 v = *reinterpret_cast<const float2*>(p);
        b[2 * j + t][0] = v.x; b[2 * j + t][1] = v.y;
      }
    }
#pragma unroll
  for (int j = 0; j < KJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (k0 + 8 * j + 2 * g + t >= a.Kd) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[2 * j + t][ct] = 0.f;
      }

  auto compute = [&](int slot) {
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* arow = &abuf[w][c][2 * g];               // A operand: the lane's row = lane & 15, its group's column pair
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const float2 x = *reinterpret_cast<const float2*>(arow + 8 * j);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b[2 * j][ct], acc[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b[2 * j + 1][ct], acc[ct], 0, 0, 0);
    }
    // C/D map: column = lane & 15 (-> the lane's columns CT c + ct), row = 4 (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* dst = &red[slot][w][g * 4 + i][c * CT];
      if (CT == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0][i], acc[1][i], acc[CT > 2 ? 2 : 0][i], acc[CT > 2 ? 3 : 0][i]);
      } else {
        *reinterpret_cast<float2*>(dst) = make_float2(acc[0][i], acc[1][i]);
      }
    }
  };

  int64_t nv = a.n_valid ? (int64_t)*a.n_valid : a.M;
  nv = nv < a.M ? nv : a.M;
  const bool red_thread = tid < 64 * CT;
  const int rr = tid / (4 * CT), cq = (tid % (4 * CT)) * 4;
  float cnt = 0.f;
  float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), s1 = sh, s2 = sh;

  // Per tile: the slice of the tile (in registers since the previous iteration) goes to the wave's LDS image, the loads
  // of the NEXT tile are issued, and the MFMAs of this tile read their fragments from the image -- one register set, the
  // next tile's HBM latency runs behind this tile's MFMAs (4 096 cycles at KS = 128, N = 64) and behind the other wave
  // of the SIMD. The image is the wave's own: wave-level hand-offs only, no workgroup barrier in the stream.
  for (int t0 = 0; t0 < T; t0 += G) {
    const int ng = T - t0 < G ? T - t0 : G;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      if (i < ng) {
        stage_tile(st);
        load_tile(tile0 + t0 + i + 1, st);
        wave_lds_handoff();
        compute(i);
        wave_lds_handoff();
      }
    }
    __syncthreads();
    if (red_thread) {
      for (int i = 0; i < ng; ++i) {
        float4 v = *reinterpret_cast<const float4*>(&red[i][0][rr][cq]);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
          const float4 p = *reinterpret_cast<const float4*>(&red[i][ww][rr][cq]);
          v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int64_t row = (tile0 + t0 + i) * 16 + rr;
        if (row < a.M) store_f4_hidden32(a.C + row * N + cq, v);
        if (a.bn_part != nullptr && row < nv) {
          if (cnt == 0.f) sh = v;
          cnt += 1.f;
          const float dx = v.x - sh.x, dy = v.y - sh.y, dz = v.z - sh.z, dw = v.w - sh.w;
          s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
          s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
      }
    }
    if (t0 + G < T) __syncthreads();
  }

  if (a.bn_part != nullptr) {
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
      a.bn_part[((int64_t)blockIdx.x * 2) * N + tid] = mean * n;
      a.bn_part[((int64_t)blockIdx.x * 2 + 1) * N + tid] = m2;
    }
  }
}

}  // namespace

// out[0] = 1 when C [M,N] = A [M,Kd] . B [Kd,N] runs on the streaming kernel, out[1] = 16-row tiles per workgroup
// (= statistics block rows / 16), out[2] = workgroups, out[3] = 0.
// Measured (round 3, profiles/r03_gemm_stream_bench.txt): the kernel reaches 0.44 of the f32 MFMA peak
// on 19 464 x 990 x 64 (35.4 us) -- no better than the tiled kernel of gemm.hip (33.8 us, 0.47) -- and 0.60 on
// 171 000 x 990 x 64 (229 against 252 us); with 32 output columns or reductions of <= 512 it loses (15 against 13 us,
// 44 against 33 us). Both kernels sit at what the matrix pipe sustains here with f32 operands (v_mfma_f32_16x16x4_f32:
// 128 MFMAs per tile and wave; ~7 us per tile round of two waves per SIMD), so the plan takes the streaming kernel only
// where it measured faster: N = 64, 512 < Kd <= 1024 (even), M >= 98 304 -- the first layer of batches of five spheres
// and more. MVK_GEMM32_STREAM=0 disables it, =2 takes it for every supported shape (M >= 4096; development). MVK_GEMM32_STREAM=0 disables it, MVK_GEMM32_TILES overrides out[1] (development).
static int stream32_nw(int64_t Kd) { return Kd > 512 ? 8 : 4; }          // waves per workgroup: slices of <= 128 columns

static int stream32_kj(int64_t Kd) {
  const int per = (int)cdiv64(Kd, stream32_nw(Kd));
  const int kj = ((per + 7) / 8 * 8) / 8;
  return kj <= 4 ? 4 : (kj <= 8 ? 8 : 16);
}

extern "C" int mvk_gemm_f32_stream_plan(int64_t M, int N, int64_t Kd, int64_t* out) {
  MVK_REQUIRE(out != nullptr, "gemm32 stream plan: null output");
  out[0] = out[1] = out[2] = out[3] = 0;
  const char* env = getenv("MVK_GEMM32_STREAM");          // read per call: tests switch it for single cases
  const int mode = env == nullptr ? 1 : atoi(env);
  if (mode == 0 || (N != 32 && N != 64) || Kd < 16 || Kd > 1024 || (Kd & 1)) return 0;
  if (mode == 2 ? M < 4096 : (M < 98304 || N != 64 || Kd <= 512)) return 0;
  const int64_t ntiles = cdiv64(M, 16);
  int64_t T = cdiv64(ntiles, 256);
  if (T < 2) T = 2;
  if (const char* e = getenv("MVK_GEMM32_TILES")) {
    const long v = atol(e);
    if (v > 0) T = v;
  }
  out[0] = 1;
  out[1] = T;
  out[2] = cdiv64(ntiles, T);
  out[3] = 0;                  // floats needed behind A: none (the loader clamps its columns)
  return 0;
}

// a_slack_floats: how many floats behind the last element of A are readable (any content): the kernel needs
// mvk_gemm_f32_stream_plan's out[3] of them.
extern "C" int mvk_gemm_f32_stream(const float* A, int64_t a_slack_floats, const float* B, float* C, int64_t M, int N,
                                   int64_t Kd, const int* n_valid, float* bn_part, void* stream) {
  int64_t plan[4];
  if (int e = mvk_gemm_f32_stream_plan(M, N, Kd, plan)) return e;
  MVK_REQUIRE(plan[0] == 1, "gemm32 stream: unsupported shape M=%lld N=%d Kd=%lld", (long long)M, N, (long long)Kd);
  (void)a_slack_floats;        // (kept in the signature: the staged loader never reads past a row)
  MVK_REQUIRE(((uintptr_t)A % 8) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0, "gemm32 stream: operand alignment");
  S32Args a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.lda = Kd; a.N = N; a.Kd = (int)Kd;
  a.tiles_per_wg = (int)plan[1]; a.n_valid = n_valid; a.bn_part = bn_part;
  const int nw = stream32_nw(Kd);
  const dim3 grid((unsigned)plan[2]), block(64 * nw);
  hipStream_t st = (hipStream_t)stream;
#define MVK_S32(CT, KJ)                                                                         \
  do {                                                                                          \
    if (nw == 8) hipLaunchKernelGGL((gemm_f32_stream<CT, KJ, 8>), grid, block, 0, st, a);       \
    else hipLaunchKernelGGL((gemm_f32_stream<CT, KJ, 4>), grid, block, 0, st, a);               \
  } while (0)
  const int kjr = stream32_kj(Kd);                      // 8-byte loads per lane and tile
  a.ks = 8 * kjr;                                       // slice length of a wave (surplus columns: zero weights)
  if (N == 64) {
    if (kjr == 4) MVK_S32(4, 4); else if (kjr == 8) MVK_S32(4, 8); else MVK_S32(4, 16);
  } else {
    if (kjr == 4) MVK_S32(2, 4); else if (kjr == 8) MVK_S32(2, 8); else MVK_S32(2, 16);
  }
#undef MVK_S32
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
