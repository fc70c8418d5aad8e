// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// extern "C" shim over the *unmodified* reference C++ core, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile). Output:
// oracle/_ref/libref.so (git-ignored). It is used to (a) pin the C restatement
// in oracle/mvk_oracle.c bit-for-bit and (b) generate tests/golden/*.npz.
//
// Reference entry points wrapped here:
//   batch_grid_subsampling   KPConv-PyTorch/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109
//   grid_subsampling         .../grid_subsampling.cpp:5
//   batch_nanoflann_neighbors KPConv-PyTorch/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211
// The marshalling mirrors what the reference CPython wrappers do
// (cpp_subsampling/wrapper.cpp:236-262, cpp_neighbors/wrapper.cpp:184-198):
// reinterpret (N,3) float32 as PointXYZ, copy into std::vector, call, memcpy out.

#include <cstring>
#include <vector>

#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"
#include "cpp_neighbors/neighbors/neighbors.h"

extern "C" {

// returns M (number of subsampled points); out buffers must hold N rows.
long ref_subsample_batch(const float* pts, long N, const float* feats, int fdim,
                         const int* labels, int ldim, const int* lens, int B,
                         float dl, int max_p, float* out_pts, float* out_feats,
                         int* out_labels, int* out_lens) {
  std::vector<PointXYZ> op((const PointXYZ*)pts, (const PointXYZ*)pts + N);
  std::vector<float> of;
  if (feats && fdim > 0) of.assign(feats, feats + N * (long)fdim);
  std::vector<int> oc;
  if (labels && ldim > 0) oc.assign(labels, labels + N * (long)ldim);
  std::vector<int> ob(lens, lens + B);
  std::vector<PointXYZ> sp;
  std::vector<float> sf;
  std::vector<int> sc, sb;
  batch_grid_subsampling(op, sp, of, sf, oc, sc, ob, sb, dl, max_p);
  std::memcpy(out_pts, sp.data(), sp.size() * sizeof(PointXYZ));
  if (out_feats && !sf.empty()) std::memcpy(out_feats, sf.data(), sf.size() * sizeof(float));
  if (out_labels && !sc.empty()) std::memcpy(out_labels, sc.data(), sc.size() * sizeof(int));
  std::memcpy(out_lens, sb.data(), sb.size() * sizeof(int));
  return (long)sp.size();
}

long ref_subsample(const float* pts, long N, const float* feats, int fdim,
                   const int* labels, int ldim, float dl, float* out_pts,
                   float* out_feats, int* out_labels) {
  std::vector<PointXYZ> op((const PointXYZ*)pts, (const PointXYZ*)pts + N);
  std::vector<float> of;
  if (feats && fdim > 0) of.assign(feats, feats + N * (long)fdim);
  std::vector<int> oc;
  if (labels && ldim > 0) oc.assign(labels, labels + N * (long)ldim);
  std::vector<PointXYZ> sp;
  std::vector<float> sf;
  std::vector<int> sc;
  grid_subsampling(op, sp, of, sf, oc, sc, dl, 0);
  std::memcpy(out_pts, sp.data(), sp.size() * sizeof(PointXYZ));
  if (out_feats && !sf.empty()) std::memcpy(out_feats, sf.data(), sf.size() * sizeof(float));
  if (out_labels && !sc.empty()) std::memcpy(out_labels, sc.data(), sc.size() * sizeof(int));
  return (long)sp.size();
}

// Two-phase: call with out == NULL to get the width, then with a buffer of
// Nq*width ints. (The search is simply run twice.)
int ref_radius_neighbors_batch(const float* q, long Nq, const float* s, long Ns,
                               const int* ql, const int* sl, int B, float radius,
                               int* out) {
  std::vector<PointXYZ> vq((const PointXYZ*)q, (const PointXYZ*)q + Nq);
  std::vector<PointXYZ> vs((const PointXYZ*)s, (const PointXYZ*)s + Ns);
  std::vector<int> vql(ql, ql + B), vsl(sl, sl + B);
  std::vector<int> res;
  batch_nanoflann_neighbors(vq, vs, vql, vsl, res, radius);
  int width = Nq > 0 ? (int)(res.size() / (size_t)Nq) : 0;
  if (out) std::memcpy(out, res.data(), res.size() * sizeof(int));
  return width;
}

}  // extern "C"
