// Development probe (GPU box): which store / wait / load combination makes per-workgroup partial results parked in HBM
// visible to the workgroup that arrives last at a counter -- the hand-off of the ordered split reductions (csrc/gemm.hip).
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/park_probe.hip -o /tmp/park_probe && /tmp/park_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float tag(int tile, int z, int tid, int b, int it) {
  return (float)(((tile * 131 + z * 17 + b * 5 + it * 7919) & 0xffff)) + tid * (1.0f / 1024.0f);
}

// STORE: 0 = asm x4 sc1, 1 = builtin atomic dword (agent, relaxed), 2 = plain x4 store, 3 = asm x4 sc0 sc1
// REL:   0 = s_waitcnt vmcnt(0) only, 1 = buffer_wbl2 sc1 + wait, 2 = __threadfence()
// LOAD:  0 = asm x4 sc1, 1 = builtin atomic dword, 2 = plain x4 load, 3 = asm x4 sc0 sc1
// ACQ:   0 = nothing, 1 = buffer_inv sc1, 2 = __threadfence()
template <int STORE, int REL, int LOAD, int ACQ>
__global__ __launch_bounds__(256) void probe(float* ws, int* cnt, int* err, int S, int NB, int it, int spin) {
  __shared__ int flag;
  const int tile = blockIdx.x, z = blockIdx.z, tid = threadIdx.x;
  // uneven arrival: some workgroups dawdle
  if (spin > 0) {
    const long long t0 = clock64();
    const long long d = (long long)((tile * 7 + z * 13) % 5) * spin;
    while (clock64() - t0 < d) {}
  }
  float* mine = ws + ((size_t)(tile * S + z) * NB) * 1024 + tid * 4;
  for (int b = 0; b < NB; ++b) {
    f32x4 v;
    for (int e = 0; e < 4; ++e) v[e] = tag(tile, z, tid, b, it) + e * 0.125f;
    float* p = mine + b * 1024;
    if (STORE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (STORE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (STORE == 1) { for (int e = 0; e < 4; ++e) __hip_atomic_store(p + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else *reinterpret_cast<f32x4*>(p) = v;
  }
  if (REL == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (REL == 1) asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)" ::: "memory");
  else __threadfence();
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag = old == S - 1;
    if (flag) __hip_atomic_store(cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!flag) return;
  if (ACQ == 1) asm volatile("buffer_inv sc1" ::: "memory");
  else if (ACQ == 2) __threadfence();
  int bad = 0;
  for (int zz = 0; zz < S; ++zz)
    for (int b = 0; b < NB; ++b) {
      const float* p = ws + ((size_t)(tile * S + zz) * NB + b) * 1024 + tid * 4;
      f32x4 v;
      if (LOAD == 0) { asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
      else if (LOAD == 3) { asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
      else if (LOAD == 1) { for (int e = 0; e < 4; ++e) v[e] = __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else v = *reinterpret_cast<const f32x4*>(p);
      for (int e = 0; e < 4; ++e) bad += v[e] != tag(tile, zz, tid, b, it) + e * 0.125f;
    }
  if (bad) atomicAdd(err, bad);
}

template <int STORE, int REL, int LOAD, int ACQ>
int run(const char* name, float* ws, int* cnt, int* err, int tiles, int S, int NB, int iters, int spin) {
  CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it)
    hipLaunchKernelGGL((probe<STORE, REL, LOAD, ACQ>), dim3(tiles, 1, S), dim3(256), 0, 0, ws, cnt, err, S, NB, it, spin);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  int h = 0;
  CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
  std::vector<int> c(tiles);
  CK(hipMemcpy(c.data(), cnt, 4 * tiles, hipMemcpyDeviceToHost));
  int nz = 0;
  for (int v : c) nz += v != 0;
  printf("%-44s tiles %5d S %2d NB %d spin %5d: %9d bad values, %d counters not zero, %.1f us per launch\n", name, tiles, S, NB, spin, h, nz,
         1000.f * ms / iters);
  return 0;
}

int main() {
  float* ws; int* cnt; int* err;
  const size_t floats = (size_t)2048 * 24 * 4 * 1024;
  CK(hipMalloc(&ws, floats * 4));
  CK(hipMalloc(&cnt, 4 * 4096));
  CK(hipMalloc(&err, 4));
  CK(hipMemset(cnt, 0, 4 * 4096));
  CK(hipMemset(ws, 0xff, floats * 4));
  for (int spin = 0; spin <= 2000; spin += 2000)
    for (int cfg = 0; cfg < 3; ++cfg) {
      const int tiles = cfg == 0 ? 609 : (cfg == 1 ? 60 : 2048), S = cfg == 0 ? 4 : (cfg == 1 ? 20 : 2), NB = 2;
      run<0, 0, 0, 0>("asm sc1 store | wait | asm sc1 load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<1, 0, 1, 0>("atomic store | wait | atomic load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<1, 0, 0, 0>("atomic store | wait | asm sc1 load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<0, 0, 1, 0>("asm sc1 store | wait | atomic load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<3, 0, 3, 0>("asm sc0 sc1 store | wait | asm sc0 sc1 load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<0, 1, 0, 0>("asm sc1 store | wbl2 + wait | asm sc1 load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<2, 1, 0, 0>("plain store | wbl2 + wait | asm sc1 load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<2, 1, 2, 1>("plain store | wbl2 + wait | inv + plain load", ws, cnt, err, tiles, S, NB, 200, spin);
      run<2, 2, 2, 2>("plain store | threadfence | threadfence + plain", ws, cnt, err, tiles, S, NB, 200, spin);
      run<2, 0, 2, 0>("plain store | wait | plain load (broken)", ws, cnt, err, tiles, S, NB, 200, spin);
    }
  return 0;
}
