// Tail of a training step in ONE launch: gradient value clipping + SGD with momentum and weight decay over
// every parameter tensor of the network (reference: utils/trainer.py:190-195 -- torch.nn.utils.clip_grad_value_
// followed by torch.optim.SGD.step, built at trainer.py:72-79 with two parameter groups; torch semantics with
// dampening 0, no Nesterov):
//     g' = clamp(g, -clip, +clip);  d = g' + wd * p;  m = momentum * m + d;  p = p - lr * m
// (a zero-initialised momentum buffer makes the first step m = d, torch's "buf = clone(d)").
// The step is pure streaming (16 B per parameter read, 8 B written: ~0.6 GB for the 24.4 M-parameter KPFCNN), but
// as library calls it is ~50 launches (a clamp per tensor + multi-tensor chunks); here a device table of tensor
// records and a chunk list let one grid walk all tensors.
#include "common.h"

namespace {

struct SgdTensor {          // mirrored by ops.FusedClipSGD (ctypes: 3 pointers, int64, 2 floats = 40 bytes)
  float* p;
  const float* g;
  float* m;
  int64_t n;
  float lr, wd;
};

constexpr int SGD_CHUNK = 4096;   // elements per workgroup: 256 threads x 4 float4

__global__ __launch_bounds__(256) void sgd_clip_kernel(const SgdTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                      float clip, float momentum, int clip_in_place) {
  const int2 c = chunks[blockIdx.x];
  SgdTensor t = tab[c.x];
  t.p = load_global_ptr(&tab[c.x].p);     // table pointers are device allocations (global_load, not flat_load)
  t.g = load_global_ptr(&tab[c.x].g);
  t.m = load_global_ptr(&tab[c.x].m);
  const int64_t base = (int64_t)c.y * SGD_CHUNK;
  const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
    if (e >= t.n) break;
    if (vec && e + 3 < t.n) {
      float4 p = *reinterpret_cast<const float4*>(t.p + e);
      float4 g = *reinterpret_cast<const float4*>(t.g + e);
      float4 m = *reinterpret_cast<const float4*>(t.m + e);
      g.x = fminf(fmaxf(g.x, -clip), clip); g.y = fminf(fmaxf(g.y, -clip), clip);
      g.z = fminf(fmaxf(g.z, -clip), clip); g.w = fminf(fmaxf(g.w, -clip), clip);
      if (clip_in_place) *reinterpret_cast<float4*>(const_cast<float*>(t.g) + e) = g;
      m.x = momentum * m.x + (g.x + t.wd * p.x); m.y = momentum * m.y + (g.y + t.wd * p.y);
      m.z = momentum * m.z + (g.z + t.wd * p.z); m.w = momentum * m.w + (g.w + t.wd * p.w);
      p.x -= t.lr * m.x; p.y -= t.lr * m.y; p.z -= t.lr * m.z; p.w -= t.lr * m.w;
      *reinterpret_cast<float4*>(t.m + e) = m;
      *reinterpret_cast<float4*>(t.p + e) = p;
    } else {
      for (int64_t j = e; j < e + 4 && j < t.n; ++j) {
        const float g = fminf(fmaxf(t.g[j], -clip), clip);
        if (clip_in_place) const_cast<float*>(t.g)[j] = g;
        const float m = momentum * t.m[j] + (g + t.wd * t.p[j]);
        t.m[j] = m;
        t.p[j] -= t.lr * m;
      }
    }
  }
}

}  // namespace

extern "C" int mvk_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" int mvk_sgd_clip_step(const void* table, const int32_t* chunks, int64_t n_chunks, float clip,
                                 float momentum, int clip_in_place, void* stream) {
  MVK_REQUIRE(n_chunks >= 0 && n_chunks < (1ll << 31) && clip >= 0.f, "sgd: bad arguments");
  if (n_chunks == 0) return 0;
  MVK_REQUIRE(table && chunks, "sgd: null table");
  hipLaunchKernelGGL(sgd_clip_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                     (const SgdTensor*)table, (const int2*)chunks, clip, momentum, clip_in_place);
  MVK_CHECK_HIP(hipGetLastError());
  return 0;
}
