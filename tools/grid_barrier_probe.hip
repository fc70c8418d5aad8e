// Development probe (GPU box): what a grid-wide barrier costs inside ONE persistent launch when the data that crosses it
// moves with agent-scope (sc1) accesses -- no L2 write-back / invalidate -- i.e. what a "phase boundary" of a persistent
// level kernel would cost against the ~4.6-5 us floor of a launch boundary in the captured chain.
// Every workgroup: write its chunk (sc1 stores), arrive (relaxed agent atomic), spin on the counter (sc1 loads), read the
// chunk of workgroup (b + 7) % G written in this phase and check it. A spin that lasts longer than ~50 ms sets the error
// word and leaves (every wave reaches the end whatever happens).
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SC1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000);
}

// CHUNK: u32x4 per thread and phase (the payload that crosses the barrier); WORK: dependent FMAs per phase (stand-in)
template <int CHUNK>
__global__ __launch_bounds__(256) void persistent(unsigned* data, int* counter, int* err, int phases, int work) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rd = rsrc(data), rc = rsrc(counter);
  float acc = (float)tid;
  int bad = 0;
  for (int p = 0; p < phases; ++p) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
      const unsigned tag = (unsigned)(p * 65536 + b * 64 + c);
      const u32x4 v = {tag, tag + 1u, (unsigned)tid, __float_as_uint(acc)};
      __builtin_amdgcn_raw_buffer_store_b128(v, rd, (unsigned)(((b * CHUNK + c) * 256 + tid) * 16), 0, SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = G * (p + 1);
      const long long t0 = clock64();
      while ((int)__builtin_amdgcn_raw_buffer_load_b32(rc, 0, 0, SC1) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (clock64() - t0 > 100000000ll || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          atomicOr(err, 1);      // ~50 ms at 2 GHz: something is wrong, do not hang the box (sticky: later barriers fall through)
          break;
        }
      }
    }
    __syncthreads();
    const int src = (b + 7) % G;
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rd, (unsigned)(((src * CHUNK + c) * 256 + tid) * 16), 0, SC1);
      const unsigned tag = (unsigned)(p * 65536 + src * 64 + c);
      bad += (v[0] != tag) + (v[1] != tag + 1u) + (v[2] != (unsigned)tid);
    }
    // the chunk is overwritten in the next phase: a second barrier would be needed before that in real use (WAR); here the
    // reader of (b + 7) % G checks tags, and a torn value counts as bad -- so run with two barriers per phase
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter + 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = G * (p + 1);
      const long long t0 = clock64();
      while ((int)__builtin_amdgcn_raw_buffer_load_b32(rc, 256, 0, SC1) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (clock64() - t0 > 100000000ll || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          atomicOr(err, 1);
          break;
        }
      }
    }
    __syncthreads();
  }
  if (bad) atomicAdd(err + 1, bad);
  if (acc == 12345.678f) data[0] = 1;      // keep the work loop
}

// the same work as `phases` separate launches (what the captured chain pays today)
template <int CHUNK>
__global__ __launch_bounds__(256) void one_phase(unsigned* data, int p, int work) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float acc = (float)tid;
  for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
  u32x4* d = reinterpret_cast<u32x4*>(data);
#pragma unroll
  for (int c = 0; c < CHUNK; ++c) {
    const unsigned tag = (unsigned)(p * 65536 + b * 64 + c);
    d[(b * CHUNK + c) * 256 + tid] = (u32x4){tag, tag + 1u, (unsigned)tid, __float_as_uint(acc)};
  }
}

template <int CHUNK>
int run(unsigned* data, int* counter, int* err, int G, int phases, int work) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  int herr[2] = {0, 0};
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(counter, 0, 4 * 128));
    CK(hipMemset(err, 0, 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((persistent<CHUNK>), dim3(G), dim3(256), 0, 0, data, counter, err, phases, work);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    int h[2];
    CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
    herr[0] |= h[0]; herr[1] += h[1];
  }
  // the launch-per-phase form, captured in a graph (as the step is)
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int p = 0; p < phases; ++p) hipLaunchKernelGGL((one_phase<CHUNK>), dim3(G), dim3(256), 0, st, data, p, work);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float bestg = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    bestg = ms < bestg ? ms : bestg;
  }
  printf("G %5d chunk %d x 4 KB work %5d: persistent %7.2f us per phase (2 barriers each; timeout %d, bad %d) | graph of launches %6.2f us per launch\n",
         G, CHUNK, work, 1000.f * best / phases, herr[0], herr[1], 1000.f * bestg / phases);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
  return 0;
}

int main() {
  unsigned* data; int* counter; int* err;
  CK(hipMalloc(&data, (size_t)2048 * 4 * 256 * 16));
  CK(hipMalloc(&counter, 4 * 128));
  CK(hipMalloc(&err, 8));
  const int phases = 200;
  for (int G : {64, 256, 512, 1024}) {      // <= 4 workgroups per CU: all resident
    if (run<1>(data, counter, err, G, phases, 0)) return 1;
    if (run<1>(data, counter, err, G, phases, 2000)) return 1;
    if (run<4>(data, counter, err, G, phases, 0)) return 1;
  }
  return 0;
}
