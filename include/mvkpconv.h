/* mvkpconv.h -- C ABI of the MI355X-native MV-KPConv hot path (libmvkpconv.so).
 *
 * Drop-in boundary. Every entry point takes plain
 * pointers and sizes; all data pointers are DEVICE pointers (HBM) unless the
 * parameter name ends in _host. `stream` is a hipStream_t passed as void*
 * (NULL = the default stream). Functions only enqueue work unless documented
 * as synchronising. Return value: 0 on success, negative on error
 * (mvk_last_error() holds the message; Python maps it to RuntimeError with the
 * reference's message strings where the reference defines one).
 *
 * Reference interfaces replaced (paths relative to /root/reference/):
 *   mvk_grid_subsample_*      KPConv-PyTorch/cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333 (subsample_batch),
 *                             :338-566 (subsample) -> grid_subsampling/grid_subsampling.cpp:5-211
 *   mvk_radius_neighbors_*    KPConv-PyTorch/cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238 (batch_query)
 *                             -> neighbors/neighbors.cpp:211-332
 *   mvk_kpconv_*              KPConv-PyTorch/models/blocks.py:237-374 (KPConv.forward) and its autograd backward
 *   mvk_max_pool_*, mvk_gather_rows_*  KPConv-PyTorch/models/blocks.py:35-66,79-110
 *   mvk_group_points_*        mvpnet/ops/cuda/group_points.cpp:16-19 (group_points_forward/backward),
 *                             mvpnet/ops/cuda/group_points_kernel.cu:25-145
 *   mvk_knn_f64               KPConv-PyTorch/datasets/ScanNet_sphere_color.py:448-451 (sklearn ball_tree k-NN)
 *   mvk_unproject_depth       KPConv-PyTorch/datasets/ScanNet_sphere_color.py:66-72,409-417
 */
#ifndef MVKPCONV_H
#define MVKPCONV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVK_ABI_VERSION 8   /* 8: the fp16-feature entry points (mvk_kpconv_gather_fwd_f16 / _ld, mvk_gemm_f16, mvk_gemm_f16_stream / _plan, mvk_round_weights_f16) are gone, mvk_deform_regularizer_many, sorted reverse lists of any width; 7: BatchNorm folded into the GEMMs around it (statistics finished by the producer, apply in the consumer's operand load), reverse lists out of the neighbour search, grouped plan takes the stream; 6: ordered split reductions (mvk_gemm_split_arena), reverse neighbour lists; 5: gather with a work list (mvk_kpconv_gather_fwd_ordered); 4: fp16 mode on gfx950 forms: padded fp16 aggregate rows, streaming contraction (v_mfma_f32_16x16x32_f16) with statistics epilogue, one-launch weight rounding; 3: gemm plan / BatchNorm-statistics epilogue, BatchNorm takes epilogue partials, fused clip + SGD, offset gradient + regulariser, segmentation loss, gather launch plan, strided gather-rows backward, channels-last fusion gather; 2: masked BatchNorm takes the batch counter and a residual addend; enqueue-only / device-lens pyramid entry points; fp16-feature mode; capacity padding */

/* influence / aggregation codes (blocks.py:329-354) */
#define MVK_INFL_CONSTANT 0
#define MVK_INFL_LINEAR 1
#define MVK_INFL_GAUSSIAN 2
#define MVK_AGG_SUM 0
#define MVK_AGG_CLOSEST 1

int mvk_abi_version(void);
const char* mvk_last_error(void);

/* ---------------- KPConv: gather + correlation + aggregation ------------- */

/* A[n,k,c] = sum_h w[n,h,k] * x+[idx[n,h], c]        
 *   q [Nq,3] f32, s [Ns,3] f32, idx [Nq,H] int32 or int64 (idx64 != 0), values in [0,Ns]
 *   (Ns = shadow neighbour), x [Ns,Cin] f32, kp [K,3] f32 (K <= 16).
 *   offsets: NULL (rigid) or [Nq,K,3] f32 = deformed kernel point displacements already scaled
 *   by KP_extent (blocks.py:266,287); when non-NULL neighbours out of range of every deformed
 *   kernel point are dropped (blocks.py:300-325) and min_d2 [Nq,K] (may be NULL) receives
 *   min_h d2 (blocks.py:303).
 *   A_out [Nq,K,Cin] f32. */
int mvk_kpconv_gather_fwd(const float* q, int64_t Nq, const float* s, int64_t Ns,
                          const void* idx, int idx64, int H, const float* x, int Cin,
                          const float* kp, int K, float extent, int influence, int aggregation,
                          const float* offsets, float* min_d2,
                          int32_t* min_arg /* [Nq,K] or NULL: neighbour column of the first entry attaining min_d2 */,
                          float* A_out, void* stream);

/* The same with a work list: order [Nq] int32 = a permutation of 0 .. Nq-1 (NULL = row order). The vector kernels of
 * rigid layers (5 <= Cin <= 512) process the points in that order -- consecutive entries share a wave, consecutive
 * waves an XCD -- and write every result in the point's own row: A_out is what mvk_kpconv_gather_fwd writes, whatever
 * the permutation (a spatially sorted one keeps the neighbour rows a workgroup pulls inside its XCD's L2). Entries
 * outside [0, Nq) or repeated entries are the caller's error (rows written twice / not at all). */
int mvk_kpconv_gather_fwd_ordered(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                  const void* idx, int idx64, int H, const float* x, int Cin,
                                  const float* kp, int K, float extent, int influence, int aggregation,
                                  const float* offsets, float* min_d2, int32_t* min_arg, float* A_out,
                                  const int32_t* order, void* stream);

/* Launch geometry mvk_kpconv_gather_fwd uses for a layer with linear influence and sum aggregation
 * (elem_bytes: 4 -- feature rows are f32; the fp16-feature mode of rounds 2-4 left the tree in round 5, DESIGN.md 4.7;
 * host only, no GPU call):
 * out[0..6] = lanes per point, points per wave, feature rows per batch of the branch-free kernel variant
 * (0 = general variant), first workgroup whose waves share their points, waves per workgroup, workgroups,
 * grid threads (the figure a kernel trace reports). out[5] = 0: the layer runs on another kernel (one point per wave;
 * one point per lane for rows of <= 4 channels). out[7] (ABI 7) = 1: the MFMA gather (kpconv_gather_mfma, round 5: layers
 * with linear influence and sum aggregation -- one wave per point, out[2] = its 16-channel accumulator tiles), 0: the vector kernel. */
int mvk_kpconv_gather_plan(int64_t Nq, int64_t Ns, int H, int Cin, int elem_bytes, int deformable, int64_t* out /* [8] */);

/* Gather-form feature gradient of a DEFORMABLE KPConv (round 5): A2 [Ns, K, C] = the forward aggregation over the
 * transposed neighbourhood relation rev [Ns, Hr] (mvk_reverse_neighbors of the layer's neighbour matrix) of g [Nq, C], the
 * gradient of the layer's output, with the kernel points of the NEIGHBOUR (query) rows: kp [K,3] + offsets [Nq,K,3], times
 * mod [Nq,K] (NULL: not modulated); dx = sum_k A2[:,k,:] . W[k]^T follows (mvk_gemm_f32_kp_transposed). Replaces the atomic
 * scatter of mvk_kpconv_scatter_bwd for deformable layers (KPConv-PyTorch/models/blocks.py:286-327, :360 through autograd);
 * linear influence, sum aggregation, C >= 5. order: work list over the Ns rows or NULL. */
int mvk_kpconv_gather_rev_deform(const float* s, int64_t Ns, const float* q, int64_t Nq, const void* rev, int rev64, int Hr,
                                 const float* g, int C, const float* kp, int K, float extent, const float* offsets,
                                 const float* mod, float* A2, const int32_t* order, void* stream);

/* dx[idx[n,h], c] += sum_k w[n,h,k] * dA[n,k,c]   (shadow rows discarded).
 * dx [Ns,Cin] must be zero-initialised by the caller (accumulated with f32 atomics).
 * Deformable extras (all NULL for rigid):
 *   x, offsets as in the forward; d_offsets [Nq,K,3] receives
 *   sum_h <x+[idx[n,h]], dA[n,k,:]> * dw/d(offset)  plus the min_d2 path g_min_d2 [Nq,K] (may be NULL). */
int mvk_kpconv_scatter_bwd(const float* q, int64_t Nq, const float* s, int64_t Ns,
                           const void* idx, int idx64, int H, int Cin,
                           const float* kp, int K, float extent, int influence, int aggregation,
                           const float* dA, float* dx,
                           const float* x, const float* offsets, const float* g_min_d2,
                           const int32_t* min_arg /* from the forward; needed with g_min_d2 */,
                           float* d_offsets, void* stream);

/* ---------------- fp32 MFMA GEMM (the K x Cin x Cout contraction) -------- */

/* C[M,N] (+)= op(A)[M,Kd] @ op(B)[Kd,N], fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
 *   transA == 0: A is [M,Kd] row-major (lda = Kd);  transA != 0: A is stored [Kd,M] (lda = M).
 *   transB == 0: B is [Kd,N] row-major (ldb = N);   transB != 0: B is stored [N,Kd] (ldb = Kd).
 *   accumulate != 0: C += (C must be initialised); split_k > 1 partitions Kd over workgroups and
 *   accumulates with f32 atomics (C must then be zero- or value-initialised by the caller and
 *   accumulate is implied).
 *   KPConv uses: y = A @ W (NN), dA = g @ W^T (NT), dW = A^T @ g (TN, split over points). */
int mvk_gemm_f32(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd,
                 int transA, int transB, int accumulate, int split_k, void* stream);

/* The same product with split_k <= 0 = "choose" and an optional epilogue for the BatchNorm that consumes C
 * (BatchNormBlock after every KPConv / unary layer, blocks.py:456-460): bn_part [ceil(M/rows), 2, N] receives,
 * per block of `rows` rows and column, the sum and the sum of squares about the block's own mean over the rows
 * below *n_valid (DEVICE int32; NULL = M); `rows` and the split the call will use come from mvk_gemm_f32_plan
 * for the same (M, N, Kd, split_k, want_stats = bn_part != NULL). The caller zeroes C when the plan's split
 * is > 1; statistics are only produced by unsplit plans (plan: *out_stat_rows > 0). */
int mvk_gemm_f32_plan(int64_t M, int64_t N, int64_t Kd, int split_k, int want_stats, int* out_split,
                      int* out_stat_rows);
int mvk_gemm_f32_ex(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transA,
                    int transB, int accumulate, int split_k, float* bn_part, const int32_t* n_valid, void* stream);

/* BatchNorm folded into the products around it (round 5; the reference normalises every KPConv / unary output with
 * nn.BatchNorm1d over the stacked point axis followed by LeakyReLU, KPConv-PyTorch/models/blocks.py:430-467, :549-561,
 * :621-649 -- as separate launches that is one more read + write of every activation and ~40 % of the launches of a step).
 *   mvk_bn_finish: the PRODUCER's half. A product that writes the statistics partials (bn_part, mvk_gemm_f32_plan:
 *     *out_stat_rows > 0; N % 4 == 0) also finishes them: the workgroup that arrives last at counters[column tile] (int32,
 *     >= ceil(N / 16) words, ZERO on entry, returned to zero by the kernel: a persistent buffer per BatchNorm) merges the
 *     partials in a fixed order and writes mean / invstd [N] (biased variance, eps), updates running_mean / running_var
 *     (nullable pair; unbiased variance, momentum) and adds 1 to *num_batches_tracked (nullable) -- exactly what
 *     mvk_bn_lrelu_fwd does in its own statistics pass. The normalising launch then only applies (mvk_bn_lrelu_fwd with
 *     ext_rows = -1) or disappears:
 *   mvk_a_transform: the CONSUMER's half. The product reads A'(m,k) = m < *n_valid ? LeakyReLU_slope((A(m,k) - mean[k]) *
 *     invstd[k] * gamma[k] + beta[k]) : 0 instead of A (NT products of more than 32 columns, Kd <= 512 per split) and also
 *     writes A' to `out` [M, Kd] (nullable) -- the activation tensor the backward's weight-gradient product reads.
 * mvk_gemm_f32_bn = mvk_gemm_f32_ex (A not transposed, plain store, the library's own split) with either half;
 * mvk_gemm_f32_pair_bn = mvk_gemm_f32_pair with want_stats = 1 and the producer's half per output. */
typedef struct mvk_bn_finish {
  int32_t* counters;
  float eps, momentum;
  float* mean;
  float* invstd;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
} mvk_bn_finish;
typedef struct mvk_a_transform {
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float slope;
  const int32_t* n_valid;
  float* out;
} mvk_a_transform;
int mvk_gemm_f32_bn(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transB, float* bn_part,
                    const int32_t* n_valid, const mvk_bn_finish* fin, const mvk_a_transform* ax, void* stream);
int mvk_gemm_f32_pair_bn(const float* A, const float* B0, const float* B1, float* C0, float* C1, int64_t M, int64_t N0,
                         int64_t N1, int64_t Kd, int transB, float* bn_part0, float* bn_part1, const int32_t* n_valid,
                         const mvk_bn_finish* fin0, const mvk_bn_finish* fin1, void* stream);

/* Ordered split reductions (round 4). By default a split product adds its partial sums with f32 atomics onto a
 * zero-initialised C: the value depends on the order the workgroups ran in (rounding only, but a LeakyReLU input within
 * an ulp of zero then flips, and two runs of the same network differ). mvk_gemm_split_arena hands the library `ws`
 * (ws_bytes of HBM, 256-byte aligned, >= 1 MB) and `counters` (n_counters >= 4096 int32 in HBM, ZERO on entry; the
 * library returns every counter to zero itself, the caller never touches the buffer again): from then on EVERY split
 * product of this library (mvk_gemm_f32 / _ex / _pair / _dual / _scatter_cat's dense half / _tn_grouped) parks its
 * partial tiles in a slice of `ws` and the workgroup that arrives last at a tile adds them in the fixed order of the
 * splits -- bit-identical from run to run, C needs no zero fill (every element is written), `accumulate` adds onto C
 * once, and the BatchNorm-statistics epilogue works on split plans too (mvk_gemm_f32_plan then reports rows > 0 with
 * split > 1). Slices (round 5): a launch on a CAPTURING stream takes its slices from the top of the arena and keeps
 * them for good (graph nodes replay, graph branches run side by side); an eager launch takes the next slice of the
 * bottom region (the first quarter of the arena, or what the captured slices left of it), and when that region is used
 * up the library waits for the device once (hipDeviceSynchronize) and starts it again from offset 0. A request that cannot be served under these rules -- a product larger than the arena,
 * or an arena used up by captured slices -- fails with an error; two launches that may run concurrently never share a
 * slice. The host state is mutex-protected. ws == NULL restores the atomic path. mvk_gemm_split_ordered() = 1 while an arena
 * is set. (The reference's products are single ATen matmuls: deterministic. This makes ours so.) */
int mvk_gemm_split_arena(void* ws, int64_t ws_bytes, void* counters, int64_t n_counters);
int mvk_gemm_split_ordered(void);

/* Gather form of KPConv's feature gradient (round 4). The reference's backward of `gather` (models/blocks.py:52-64, used
 * at :360) is an index-add dx[idx[n,h]] += sum_k w[n,h,k] dA[n,k,:]; mvk_kpconv_scatter_bwd runs it with float atomics
 * (sums in order of arrival). With the TRANSPOSED neighbourhood relation the same sums are a gather:
 *   dx[j, :] = sum_k ( sum_{n in rev[j]} w_k(q_n - s_j) g[n, :] ) . W[k]^T
 * = mvk_kpconv_gather_fwd over (queries = the layer's supports, supports = its queries, idx = rev, features = g,
 * kernel points NEGATED) followed by mvk_gemm_f32_kp_transposed -- fixed summation order, no atomics.
 *
 * mvk_reverse_neighbors: rev [Ns, Hr] int32 <- idx [Nq, H] (int32 / int64; entries outside [0, Ns) are shadow entries):
 *   row j = the rows n of idx that contain j (ascending with sort != 0), padded with `shadow` (normally Nq). counts [Ns] int32: ZERO on
 *   entry, zero again afterwards (keep one persistent buffer). status (int32 [2] or NULL): [0] = running maximum of the
 *   row lengths, [1] |= 1 when a row is longer than Hr (entries dropped: treat as an error). Hr <= MVK_REV_MAX_WIDTH
 *   (sorted rows wider than 512 entries take a workgroup per row). Two launches.
 * mvk_gemm_f32_kp_transposed: dx [M, Cin] = sum_k A[:, k, :] . W[k]^T, A [M, K, Cout], W [K, Cin, Cout] (the layer's
 *   weights, read in place), Cout a power of two >= 32. */
int mvk_reverse_neighbors(const void* idx, int idx64, int64_t Nq, int H, int64_t idx_stride, int64_t Ns, int32_t* rev, int Hr,
                          int32_t shadow, int sort, int32_t* counts, int32_t* status, void* stream);

/* Round 5: the transposed relation out of the neighbour search itself. mvk_radius_neighbors_dev_rev is
 * mvk_radius_neighbors_dev (width <= 64) whose lanes also take their slot in rev [Ns_cap, rev_width] for every kept pair
 * (query row, support) -- rows in order of arrival, row lengths left in rev_counts [Ns_cap] (ZERO on entry),
 * rev_status [2] = [longest row, overflow] as in mvk_reverse_neighbors; tails are not written and counters not reset.
 * mvk_reverse_finish_many then pads the tails with each list's shadow value, returns the counters to zero and reports
 * the longest rows of up to MVK_REV_MANY lists in ONE launch (the nine lists of a five-level pyramid: 18 launches -> 1
 * on top of the searches). The ascending order deterministic mode needs stays with mvk_reverse_neighbors(sort = 1). */
#define MVK_REV_MANY 12
#define MVK_REV_MAX_WIDTH 8192   /* longest row of a reverse list */
typedef struct mvk_rev_list {
  int32_t* rev;        /* [rows, width] */
  int32_t* counts;     /* [rows] */
  int32_t* status;     /* [2] or NULL */
  int64_t rows;
  int32_t width;
  int32_t shadow;
} mvk_rev_list;
int mvk_radius_neighbors_dev_rev(const float* q, int64_t Nq_cap, const float* s, int64_t Ns_cap, const int32_t* q_lens_dev,
                                 const int32_t* s_lens_dev, int B, float radius, int32_t* out, int width, int32_t shadow,
                                 int32_t* status_dev, int reuse_grid, void* workspace, int64_t workspace_bytes, int32_t* rev,
                                 int rev_width, int32_t* rev_counts, int32_t* rev_status, void* stream);
int mvk_reverse_finish_many(const mvk_rev_list* lists, int n, void* stream);
int mvk_gemm_f32_kp_transposed(const float* A, const float* W, float* dx, int64_t M, int K, int Cin, int Cout, void* stream);

/* The other scatter backwards of the network as gathers over a reverse list (deterministic mode; every output element is
 * written, `base` [Ns, C] or NULL is added): mvk_gather_sum_rows -- out[j, :] = base[j, :] + sum_{n in rev[j]} g[n, 0:C]
 * (g rows ldg floats apart): backward of closest_pool / nearest upsampling (models/blocks.py:79-91), rev = the transposed
 * FIRST column of the upsampling matrix (mvk_reverse_neighbors with H = 1, idx_stride = its width);
 * mvk_max_pool_bwd_gather -- backward of max_pool (blocks.py:94-110) over the transposed pooling matrix, arg [Nq, C] the
 * winning columns recorded by mvk_max_pool_fwd. mvk_gemm_f32_ldb: C [M,N] = A [M,Kd] . B where B is a column block of a
 * wider row-major matrix (rows ldb floats apart) -- the two halves of the decoder's concatenated gradient as two
 * products. (In mvk_reverse_neighbors: idx row n starts at idx + n * idx_stride; sort != 0: rows ascending -- what a
 * fixed summation order needs --, 0: in order of arrival, which saves the ranking pass.) */
int mvk_gather_sum_rows(const float* g, int64_t ldg, int64_t Nq, const int32_t* rev, int Hr, int64_t Ns, int C,
                        const float* base, float* out, void* stream);
int mvk_max_pool_bwd_gather(const float* g, const int32_t* arg, const void* idx, int idx64, int H, int64_t Nq,
                            const int32_t* rev, int Hr, int64_t Ns, int C, const float* base, float* dx, void* stream);
int mvk_gemm_f32_ldb(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int64_t ldb, void* stream);

/* C [M,N] = LeakyReLU_slope(A . op(B) + bias[col]) (slope = 1: the bias alone; A [M,Kd] row-major): a BatchNorm-less
 * layer -- `x W^T + self.bias` (blocks.py:462-463) and the block's activation, the two head layers of every network -- in
 * one launch. The reduction is not split. */
int mvk_gemm_f32_bias_act(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kd, int transB,
                          const float* bias, float slope, void* stream);

/* Backward of closest_pool + torch.cat + nn.Linear (KPFCNN decoder, architectures.py:334-335 + the unary block behind):
 * the product A [M,Kd] . B [Kd,N] is the gradient of cat([x[idx[m,0]], skip[m]]) and is never stored -- its first c1
 * columns are added onto row idx[m,0] of d_x [Ns,c1] (f32 atomics; zero-initialised by the caller; shadow indices
 * dropped), the other N - c1 columns are written to d_skip [M, N-c1] (atomics into a zero-initialised d_skip when
 * mvk_gemm_f32_plan(M, N, Kd) splits the reduction). */
int mvk_gemm_f32_scatter_cat(const float* A, const float* B, int64_t M, int64_t N, int64_t Kd, const void* idx, int idx64,
                             int64_t idx_stride, int64_t Ns, int c1, float* d_x, float* d_skip, void* stream);

/* C [M,N] = A [M,Kd] . B [Kd,N] + A2 [M,Kd2] . B2 [Kd2,N] (row-major, nothing transposed) in ONE launch, the two
 * reductions laid end to end: the gradient of a tensor that feeds two linear layers (dx = g0 W0 + g1 W1; unary1 and the
 * shortcut of a bottleneck block, blocks.py:596-649). mvk_gemm_f32_dual_plan: the split of the concatenated reduction
 * (zero-initialise C when > 1), or 0 when the shape is not supported (Kd must be a multiple of 32, N > 32). */
int mvk_gemm_f32_dual_plan(int64_t M, int64_t N, int64_t Kd, int64_t Kd2, int* out_split);
int mvk_gemm_f32_dual(const float* A, const float* B, const float* A2, const float* B2, float* C, int64_t M, int64_t N,
                      int64_t Kd, int64_t Kd2, void* stream);

/* Two products that share the left operand in ONE launch: C0 [M,N0] = A . op(B0), C1 [M,N1] = A . op(B1) (A [M,Kd]
 * row-major; transB as above) -- unary1 and the shortcut layer of a bottleneck block read the same input
 * (blocks.py:596-649). mvk_gemm_f32_pair_plan: out[0] = 1 when the pair can share a launch (both on the wide tile
 * class with the same row tile), out[1..2] = the splits of the two reductions (zero-initialise an output whose split
 * is > 1), out[3..4] = row-block sizes of the BatchNorm partials bn_part0 / bn_part1 (0: none produced). */
int mvk_gemm_f32_pair_plan(int64_t M, int64_t N0, int64_t N1, int64_t Kd, int want_stats, int* out /* [5] */);
int mvk_gemm_f32_pair(const float* A, const float* B0, const float* B1, float* C0, float* C1, int64_t M, int64_t N0,
                      int64_t N1, int64_t Kd, int transB, int want_stats, float* bn_part0, float* bn_part1,
                      const int32_t* n_valid, void* stream);

/* Upper bound of the split of the reduction the grouped launch gives one product (the plan may choose fewer pieces for
 * a large group); the caller zero-initialises the outputs for which it is > 1. */
int mvk_gemm_f32_tn_grouped_split(int64_t M, int64_t N, int64_t Kd);
/* Grouped weight-gradient products: n independent TN products C_i [M_i,N_i] = A_i^T B_i (A_i [Kd_i,M_i], B_i [Kd_i,N_i],
 * row-major: dW = A^T g of a KPConv layer, dW^T = g^T x of a unary layer, blocks.py / SURVEY.md A.6) in at most two
 * launches. `problems` = HOST array of records {const float* A; const float* B; float* C; int64 M, N, Kd;} (48 bytes,
 * N > 16). _plan fills `table_host` (n * mvk_gemm_group_entry_bytes() bytes) and reports per problem the split of its
 * reduction (C_i must be zero-initialised when splits[i] > 1); the caller copies the table to the device and passes
 * that copy to mvk_gemm_f32_tn_grouped together with the three counts the plan returned. */
int64_t mvk_gemm_group_entry_bytes(void);
int mvk_gemm_f32_tn_grouped_plan(const void* problems, int n, void* table_host, int* n_narrow, int64_t* wgs_narrow,
                                 int64_t* wgs_wide, int32_t* splits, void* stream /* the launch's stream (ABI 7) */);
int mvk_gemm_f32_tn_grouped(const void* table_dev, int n, int n_narrow, int64_t wgs_narrow, int64_t wgs_wide,
                            void* stream);

/* The K x Cin x Cout contraction of the big rigid layers as a stream (models/blocks.py:370-374; exact f32 on
 * v_mfma_f32_16x16x4_f32): C [M,N] = A [M,Kd] . B [Kd,N], N = 32 or 64, Kd even and <= 1024, M >= 4096. The weights stay in
 * registers (wave w of the workgroup: its 1/4 or 1/8 of the reduction), the rows of A stream through as 8-byte fragment
 * loads without LDS staging, the waves' partial blocks are added in a fixed order (deterministic, no atomics).
 * bn_part != NULL: BatchNorm partials of C over the rows below *n_valid (NULL: all M), format of mvk_gemm_f32_ex with
 * blocks of plan[1] * 16 rows: [plan[2], 2, N]. The kernel reads up to plan[3] floats behind the last element of A
 * (content irrelevant, it is masked): a_slack_floats states how many are readable.
 * _plan (host only): out[0] = 1 when the shape runs on this kernel, out[1] = 16-row tiles per workgroup, out[2] =
 * workgroups, out[3] = floats needed behind A. */
int mvk_gemm_f32_stream_plan(int64_t M, int N, int64_t Kd, int64_t* out /* [4] */);
int mvk_gemm_f32_stream(const float* A, int64_t a_slack_floats, const float* B, float* C, int64_t M, int N, int64_t Kd,
                        const int* n_valid, float* bn_part, void* stream);

/* ---------------- deformable KPConv: offset gradient, offset regulariser ---- */

/* d_offsets [Nq,K,3] of a deformable KPConv (blocks.py:286-327, 366-374 through autograd):
 *   sum_h dw[n,h,k]/d off[n,k,:] * (sum_c x[j_h,c] dA[n,k,c])  -  2 (rel[h*] - kpdef[k]) g_min_d2[n,k]
 * over the neighbours kept by the in-range filter (h* = min_arg[n,k], the first arg-min of d2 over ALL entries,
 * shadow included, recorded by mvk_kpconv_gather_fwd).
 * dA is the gradient of the aggregate (already multiplied by the modulations when modulated); g_min_d2 may be NULL.
 * Plain stores (no zero-initialisation needed). Called by mvk_kpconv_scatter_bwd for deformable layers. */
int mvk_kpconv_deform_doff(const float* q, int64_t Nq, const float* s, int64_t Ns, const void* idx, int idx64, int H,
                           const float* x, int Cin, const float* kp, int K, float extent, int influence,
                           const float* offsets, const float* dA, const float* g_min_d2,
                           const int32_t* min_arg, float* d_offsets, void* stream);

/* From the inner (offset) convolution's output raw [N,D] (D = 3K, or 4K when modulated) to the operands of the
 * deformable KPConv, one launch (blocks.py:243-266, :287): feat = raw + bias; offsets[n,k,:] = feat[n,3k..3k+2] * extent;
 * deformed_kp = offsets + kernel_points; modulations[n,k] = 2 sigmoid(feat[n, 3K + k]). */
int mvk_deform_operands_fwd(const float* raw, const float* bias, const float* kernel_points, int64_t N, int K,
                            int modulated, float extent, float* feat, float* offsets, float* deformed_kp,
                            float* modulations /* NULL unless modulated */, void* stream);
/* Its backward: d_raw [N,D] and d_bias [D] (+=, zero-initialised by the caller; f32 atomics) from the gradients of
 * offsets, deformed_kp, modulations and feat (each may be NULL). */
int mvk_deform_operands_bwd(const float* g_offsets, const float* g_deformed_kp, const float* g_modulations,
                            const float* modulations, const float* g_feat, int64_t N, int K, int modulated,
                            float extent, float* d_raw, float* d_bias, void* stream);

/* p2p_fitting_regularizer of ONE deformable layer (models/architectures.py:20-58) and both its gradients:
 *   *loss_accum += power * ( 2 * mean_{n<nv,k} |min_d2| / ext^2 + sum_i mean_{n<nv} sum_{j!=i} clamp_max(|loc_i - loc_j| - R, 0)^2 / K ),
 *   loc = deformed_kp / ext, the other point of a pair detached; d_min_d2 [N,K] and d_deformed_kp [N,K,3] receive
 *   d loss / d input (for an upstream gradient of 1). n_valid: DEVICE int32 row count or NULL (= N). */
int mvk_deform_regularizer(const float* min_d2, const float* deformed_kp, const int32_t* n_valid, int64_t N, int K,
                           float extent, float repulse_extent, float power, float* loss_accum, float* d_min_d2,
                           float* d_deformed_kp, void* stream);
/* The same with every output optional and the gradients scaled by grad_scale[0] (DEVICE scalar, NULL = 1): a forward
 * launch takes loss_accum only (all layers of a network accumulate into ONE scalar), the backward launch takes the
 * upstream gradient and writes d_min_d2 / d_deformed_kp already scaled. */
int mvk_deform_regularizer_ex(const float* min_d2, const float* deformed_kp, const int32_t* n_valid, int64_t N, int K,
                              float extent, float repulse_extent, float power, float* loss_accum, const float* grad_scale,
                              float* d_min_d2, float* d_deformed_kp, void* stream);

/* Every deformable layer of a network in one launch (a forward call: loss_accum only; a backward call: grad_scale and
 * the gradient outputs of each layer). At most MVK_REG_MANY layers per call; all layers share K. Layers with N == 0 are
 * skipped. */
#define MVK_REG_MANY 16
typedef struct mvk_reg_layer {
  const float* min_d2;        /* [N,K] */
  const float* deformed_kp;   /* [N,K,3] */
  const int32_t* n_valid;     /* DEVICE row count or NULL (= N) */
  float* d_min_d2;            /* [N,K] or NULL */
  float* d_deformed_kp;       /* [N,K,3] or NULL */
  int64_t N;
  float extent, repulse_extent, power;
} mvk_reg_layer;
int mvk_deform_regularizer_many(const mvk_reg_layer* layers, int n, int K, float* loss_accum, const float* grad_scale,
                                void* stream);

/* ---------------- frozen 2D encoder: pointwise epilogue of a convolution --- */

/* y = act(x + bias[c] (+ res (+ bias2[c]))) over a channels-last (N,H,W,C) f32 tensor, C % 4 == 0: what is left of
 * conv -> BatchNorm(eval) -> [+ identity] -> ReLU of the frozen UNet-ResNet34 (mvpnet/models/unet_resnet34.py:9-125,
 * frozen in architectures_sphere.py:232-237) once the BatchNorm is folded into the convolution weights.
 * res / bias2 may be NULL; y may alias x. */
int mvk_bias_act_nhwc(const float* x, const float* bias, const float* res, const float* bias2, float* y,
                      int64_t n_elems, int channels, int relu, void* stream);

/* ---------------- optimiser tail of a training step ------------------------ */

/* Segmentation loss of KPFCNN (architectures.py:345-372): labels that are not in `valid_labels` are ignored, the rest
 * are renumbered 0..C-1 (lut[l + 1] = class of label l or -1; lut[0] serves negative labels, lut[lut_n - 1] the labels
 * above the table), then weighted cross entropy, mean over the kept points:
 *   out2[0] = sum_i w[t_i] (logsumexp(x_i) - x_i[t_i]) / sum_i w[t_i],  out2[1] = sum_i w[t_i]   (class_weight NULL: 1)
 * partials: mvk_xent_workspace_floats(N) floats. Two launches (partial sums, ordered finish): deterministic. */
int64_t mvk_xent_workspace_floats(int64_t N);
int mvk_xent_fwd(const float* logits /* [N,C] */, int64_t N, int C, const void* labels, int labels64, const int32_t* lut,
                 int lut_n, const float* class_weight, float* partials, float* out2, void* stream);
/* dlogits[i,:] = grad_loss[0] * w[t_i] / out2[1] * (softmax(x_i) - onehot(t_i)), 0 for ignored points. */
int mvk_xent_bwd(const float* logits, int64_t N, int C, const void* labels, int labels64, const int32_t* lut, int lut_n,
                 const float* class_weight, const float* out2, const float* grad_loss, float* dlogits, void* stream);

/* Gradient value clipping + SGD (momentum, weight decay; torch.optim.SGD semantics, dampening 0, no Nesterov) over
 * ALL parameter tensors in one launch (utils/trainer.py:190-195: clip_grad_value_ + optimizer.step; groups with
 * their own learning rate as built at trainer.py:72-79):
 *     g' = clamp(g, -clip, clip); m = momentum * m + (g' + wd * p); p -= lr * m.
 *   table : DEVICE array of records {float* p; const float* g; float* m; int64 n; float lr; float wd;} (40 bytes)
 *   chunks: DEVICE int32 [n_chunks][2] = (record index, chunk index within the tensor); a chunk is
 *           mvk_sgd_chunk_elems() consecutive elements; every tensor is covered by ceil(n / chunk) chunks.
 *   clip_in_place != 0 also stores the clamped gradient (what clip_grad_value_ leaves in .grad). */
int mvk_sgd_chunk_elems(void);
int mvk_sgd_clip_step(const void* table, const int32_t* chunks, int64_t n_chunks, float clip, float momentum,
                      int clip_in_place, void* stream);

/* ---------------- masked BatchNorm + LeakyReLU (capacity-padded levels) --- */

/* Rows up to which mvk_bn_lrelu_fwd / mvk_bn_lrelu_bwd run as ONE launch for a D-channel input (statistics and
 * normalisation together): a producing GEMM need not emit BatchNorm partials for such an output. */
int mvk_bn_single_launch_rows(int D);

/* Training-mode BatchNorm over the first *n_valid rows of x [R,D] (n_valid: DEVICE int32, so the
 * launch geometry is fixed while the row count varies -- hipGraph replay), fused with
 * LeakyReLU(slope) (slope = 1 -> no activation); rows >= n_valid of y are zero. Replaces
 * BatchNormBlock + LeakyReLU (KPConv-PyTorch/models/blocks.py:430-467,:549-561). running_mean /
 * running_var (may be NULL) are updated with `momentum` (unbiased variance). mean / invstd [D] are
 * saved for the backward; scratch is a [ceil(R/64),2,D] f32 scratch (per-row-block partial sums,
 * reduced in block order: no atomics, bit-reproducible). */
int mvk_bn_lrelu_fwd(const float* x, const int32_t* n_valid, int64_t R, int D, const float* gamma,
                     const float* beta, float eps, float momentum, float slope, float* running_mean,
                     float* running_var, float* mean, float* invstd, float* scratch, float* y,
                     int64_t* num_batches_tracked /* DEVICE counter += 1, may be NULL */,
                     const float* addend /* [R,D] or NULL: y = LeakyReLU(BN(x) + addend), the residual join of
                                            ResnetBottleneckBlock (blocks.py:649) */,
                     const float* ext_part /* NULL, or the bn_part of the mvk_gemm_f32_ex call that produced x */,
                     int ext_rows /* its row-block size (0 with NULL) */, void* stream);
/* dgamma_dbeta [2,D] receives dbeta (row 0) and dgamma (row 1); dx [R,D] (rows >= n_valid zero). */
int mvk_bn_lrelu_bwd(const float* x, const float* g, const int32_t* n_valid, int64_t R, int D,
                     const float* gamma, const float* beta, const float* mean, const float* invstd,
                     float slope, float* scratch, float* dgamma_dbeta, float* dx,
                     const float* y_out /* forward output, needed (with d_addend) when the forward had an addend */,
                     float* d_addend /* [R,D] gradient of the addend, or NULL */, void* stream);

/* Two BatchNorm problems of the SAME row count in one launch each way (same kernels, the problem picked by blockIdx.z):
 * a bottleneck block's convolution output and its shortcut (blocks.py:596-649) are normalised independently. The
 * fields are the arguments of mvk_bn_lrelu_fwd / _bwd; problems that do not share a kernel family (row count,
 * alignment) run one after the other. */
typedef struct mvk_bn_fwd_problem {
  const float* x; const int32_t* n_valid; int64_t R; int32_t D; const float* gamma; const float* beta;
  float eps, momentum, slope; float* running_mean; float* running_var; float* mean; float* invstd; float* scratch2D;
  float* y; int64_t* num_batches_tracked; const float* addend; const float* ext_part; int32_t ext_rows;
} mvk_bn_fwd_problem;
typedef struct mvk_bn_bwd_problem {
  const float* x; const float* g; const int32_t* n_valid; int64_t R; int32_t D; const float* gamma; const float* beta;
  const float* mean; const float* invstd; float slope; float* scratch; float* dgamma_dbeta; float* dx;
  const float* y_out; float* d_addend;
} mvk_bn_bwd_problem;
int mvk_bn_lrelu_fwd_pair(const mvk_bn_fwd_problem* a, const mvk_bn_fwd_problem* b, void* stream);
int mvk_bn_lrelu_bwd_pair(const mvk_bn_bwd_problem* a, const mvk_bn_bwd_problem* b, void* stream);

/* y = LeakyReLU_slope(x + bias) over rows of C <= 256 channels: the BatchNorm-less form of BatchNormBlock
 * (models/blocks.py:462-463) fused with the LeakyReLU that follows it in UnaryBlock (:493-498) -- the two head layers.
 * _bwd: dx = g * (y > 0 ? 1 : slope) and dbias = column sums of dx: WRITTEN, the 64-row blocks' sums added in block
 * order, while mvk_gemm_split_ordered() == 1 (the arena of ordered reductions); otherwise ADDED with one float atomic
 * per block and column onto a dbias the caller zero-initialised. */
int mvk_bias_lrelu_fwd(const float* x, const float* bias, int64_t R, int C, float slope, float* y, void* stream);
int mvk_bias_lrelu_bwd(const float* y, const float* g, int64_t R, int C, float slope, float* dx, float* dbias, void* stream);

/* y = LeakyReLU_slope(a + b) over n elements and its backward d = g * (y > 0 ? 1 : slope) (both
 * addends receive d): the residual join of ResnetBottleneckBlock (blocks.py:649), slope > 0. */
int mvk_add_lrelu_fwd(const float* a, const float* b, int64_t n, float slope, float* y, void* stream);
int mvk_add_lrelu_bwd(const float* y, const float* g, int64_t n, float slope, float* d, void* stream);

/* ---------------- row gathers of blocks.py ------------------------------- */

/* out[n,:] = max_h x+[idx[n,h],:]  (zero shadow row participates, blocks.py:94-110);
 * arg [Nq,C] int32 (may be NULL) receives the winning h for the backward. */
int mvk_max_pool_fwd(const float* x, int64_t Ns, int C, const void* idx, int idx64, int64_t Nq,
                     int H, float* out, int32_t* arg, void* stream);
/* dx[idx[n,arg[n,c]], c] += g[n,c] (dx zero-initialised by the caller). */
int mvk_max_pool_bwd(const float* g, const int32_t* arg, const void* idx, int idx64, int64_t Nq,
                     int H, int64_t Ns, int C, float* dx, void* stream);
/* out[n,:] = x+[idx[n*stride],:]  (closest_pool = first column, blocks.py:79-91). */
int mvk_gather_rows_fwd(const float* x, int64_t Ns, int C, const void* idx, int idx64, int64_t Nq,
                        int64_t idx_stride, float* out, void* stream);
int mvk_gather_rows_bwd(const float* g, const void* idx, int idx64, int64_t Nq, int64_t idx_stride,
                        int64_t Ns, int C, float* dx, void* stream);
/* The same with the rows of g `g_ld` floats apart (g_ld >= C): the gradient of one half of a concatenation is a
 * column slice of the concatenated gradient (KPFCNN decoder, architectures.py:334), read in place. */
int mvk_gather_rows_bwd_ld(const float* g, int64_t g_ld, const void* idx, int idx64, int64_t Nq,
                           int64_t idx_stride, int64_t Ns, int C, float* dx, void* stream);

/* out [Nq, C1+C2] = [x[idx[n,0]] | skip[n]] : closest_pool of the coarse features (a row of zeros for the shadow index)
 * and torch.cat with the encoder's skip features [Nq,C2] (KPFCNN decoder, architectures.py:334-335) in one launch. */
int mvk_gather_rows_cat_fwd(const float* x, int64_t Ns, int C1, const void* idx, int idx64, int64_t Nq,
                            int64_t idx_stride, const float* skip, int C2, float* out, void* stream);
/* Its backward: g [Nq, C1+C2] -> dx [Ns,C1] += the upsampled half (f32 atomics; zero-initialised by the caller; NULL =
 * not needed) and d_skip [Nq,C2] = the skip half as a dense tensor (NULL = not needed). */
int mvk_gather_rows_cat_bwd(const float* g, const void* idx, int idx64, int64_t Nq, int64_t idx_stride, int64_t Ns,
                            int C1, int C2, float* dx, float* d_skip, void* stream);

/* ---------------- input pyramid ------------------------------------------ */

/* Scratch size in bytes needed by mvk_grid_subsample_batch for N points in B clouds. */
int64_t mvk_grid_subsample_workspace(int64_t N, int B, int fdim, int ldim);

/* Voxel-grid barycentre subsampling of a stacked batch, bit-identical (values and ORDER) to the
 * reference. pts [N,3] f32, lens_host [B] int32 (HOST), optional feats [N,fdim]
 * f32 (barycentre of features) and labels [N,ldim] int32 (majority vote, ties broken like the
 * reference's unordered_map<int,int> iteration). Outputs (device): out_pts [N,3] (capacity N rows),
 * out_feats [N,fdim] / out_labels [N,ldim] or NULL, out_lens [B] int32. The per-cloud counts are
 * also copied to out_lens_host [B] (HOST) -- this call SYNCHRONISES the stream once to return them.
 * max_p as in the reference (0 = no cap). */
int mvk_grid_subsample_batch(const float* pts, int64_t N, const int32_t* lens_host, int B,
                             const float* feats, int fdim, const int32_t* labels, int ldim,
                             float dl, int max_p, float* out_pts, float* out_feats,
                             int32_t* out_labels, int32_t* out_lens,
                             int32_t* out_lens_host, void* workspace, int64_t workspace_bytes,
                             void* stream);

/* batch_grid_subsampling with random_grid_orient (datasets/common.py:77-182): cloud b is rotated by
 * rot_host[b] (HOST, [B,3,3] f32, row-major; p' = sum_j p_j R[j,:] in float32, :118), subsampled, and its
 * barycentres are rotated back by the transpose (:134) -- same results as rotating with NumPy around
 * mvk_grid_subsample_batch, without the intermediate tensors. Features / labels are untouched. */
int mvk_grid_subsample_batch_oriented(const float* pts, int64_t N, const int32_t* lens_host, int B,
                                      const float* rot_host, const float* feats, int fdim,
                                      const int32_t* labels, int ldim, float dl, int max_p,
                                      float* out_pts, float* out_feats, int32_t* out_labels,
                                      int32_t* out_lens, int32_t* out_lens_host, void* workspace,
                                      int64_t workspace_bytes, void* stream);

/* Device-lens variant for a hipGraph-capturable input chain: nothing is read back, every launch has a
 * fixed geometry. pts [cap_in,3] holds the clouds back to back, cloud b = lens_dev[b] rows (DEVICE int32
 * [B]); rows past sum(lens) are ignored. rot_dev (DEVICE [B,3,3] f32 or NULL) as rot_host above. out_pts
 * [out_cap,3]: the subsampled clouds back to back, rows past their total = pad_value; out_lens_dev [B];
 * *total_out_dev (DEVICE, may be NULL) = sum(out_lens). status_dev (DEVICE int32 [2]): [1] is set when the
 * result does not fit out_cap (rows beyond it are dropped, never written). Workspace as for N = cap_in. */
int mvk_grid_subsample_batch_dev(const float* pts, int64_t cap_in, const int32_t* lens_dev, int B,
                                 const float* rot_dev, float dl, float* out_pts, int64_t out_cap,
                                 float pad_value, int32_t* out_lens_dev, int32_t* total_out_dev,
                                 int32_t* status_dev, void* workspace, int64_t workspace_bytes, void* stream);

int64_t mvk_radius_neighbors_workspace(int64_t Nq, int64_t Ns, int B);

/* Fixed-radius neighbours of a stacked batch: out [Nq,width] int32, row = indices (into the stacked
 * support array) of all supports of the same cloud with float32 d2 < r*r, ascending d2 (ties:
 * ascending index), padded with Ns.
 *   Phase 1 (out == NULL): counts every row, returns max count in *width_host (SYNCHRONISES).
 *   Phase 2 (out != NULL): fills out [Nq,width] keeping the `width` nearest of every row
 *   (width may be smaller than the max count = the reference's neighborhood_limits crop,
 *   datasets/common.py:411-421); *width_host (if not NULL) receives the max count seen, so a caller
 *   that passed width = limit can narrow the matrix to min(limit, max count) like the reference.
 *   SYNCHRONISES (overflow flag + max count). q_lens_host / s_lens_host are HOST arrays. */
int mvk_radius_neighbors_batch(const float* q, int64_t Nq, const float* s, int64_t Ns,
                               const int32_t* q_lens_host, const int32_t* s_lens_host, int B,
                               float radius, int32_t* out, int width, int* width_host,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* Enqueue-only variant of phase 2 (never synchronises): out [Nq,width] is always written at full
 * `width`; status_dev (DEVICE, int32 [2], zeroed by the caller once per batch) accumulates
 * [0] = max(row count seen) and [1] = 1 if a row overflowed the in-kernel list capacity -- the caller
 * reads it back once after the last search of a pyramid. reuse_grid != 0: the workspace still holds
 * the cell grid of the previous call for the SAME supports, s_lens and radius (the conv / pool /
 * upsample searches of one level of datasets/common.py:832-857 share supports and radius), so only
 * the query pass runs. */
int mvk_radius_neighbors_enqueue(const float* q, int64_t Nq, const float* s, int64_t Ns,
                                 const int32_t* q_lens_host, const int32_t* s_lens_host, int B,
                                 float radius, int32_t* out, int width, int32_t* status_dev,
                                 int reuse_grid, void* workspace, int64_t workspace_bytes, void* stream);

/* Device-lens variant of mvk_radius_neighbors_enqueue (hipGraph capturable): q [Nq_cap,3] / s [Ns_cap,3]
 * hold the clouds back to back with DEVICE lengths; out [Nq_cap,width] is written for EVERY row: real rows
 * as above with pad value `shadow`, rows past sum(q_lens) all `shadow`. Workspace as for (Nq_cap, Ns_cap). */
int mvk_radius_neighbors_dev(const float* q, int64_t Nq_cap, const float* s, int64_t Ns_cap,
                             const int32_t* q_lens_dev, const int32_t* s_lens_dev, int B, float radius,
                             int32_t* out, int width, int32_t shadow, int32_t* status_dev, int reuse_grid,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* No reference counterpart. Work list for mvk_kpconv_gather_fwd_ordered out of the cell grid the workspace holds (built
 * by the last search on it with the same Ns -- Ns_cap for the device-lens entry point --, B and lengths; the conv
 * search of a level has supports = the level's points): order_out[0 .. total) = the stacked support rows sorted by
 * cloud, grid cell (x fastest) and row, a permutation of 0 .. total-1 that is the same on every run;
 * order_out[total .. order_cap) = identity (capacity-padded rows). s_lens_dev: the DEVICE lengths given to
 * mvk_radius_neighbors_dev, NULL after the host-length entry points. One launch, never synchronises. */
int mvk_neighbors_cell_order(int64_t Ns, int B, const int32_t* s_lens_dev, int32_t* order_out, int64_t order_cap,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------- capacity padding (hipGraph replay over fixed shapes) ----- */

/* No reference counterpart (the reference re-allocates every batch). dst [cap,w] f32: rows < n copied
 * from src [n,w], the rest = fill; *count_out (DEVICE, may be NULL) = n, the row count word the masked
 * BatchNorm kernels read. */
int mvk_pad_points(const float* src, int64_t n, float* dst, int64_t cap, int w, float fill,
                   int32_t* count_out, void* stream);
/* dst [cap,w_dst] (int32, or int64 when idx64): dst[r,c] = src[r,c] for r < n, c < w_src with the
 * "no neighbour" index shadow_src rewritten to shadow_dst; every other element = shadow_dst. */
int mvk_pad_index_rows(const void* src, int idx64, int64_t n, int w_src, int64_t shadow_src, void* dst,
                       int64_t cap, int w_dst, int64_t shadow_dst, void* stream);

/* ---------------- 2D -> 3D fusion ----------------------------------------- */

/* depth (nv,h,w) uint16 millimetres, cam_inv_host [3,3] f64 (HOST; inverse intrinsics), poses [nv,4,4] f32
 * -> xyz (nv,h,w,3) f64 world coordinates, valid (nv,h,w) uint8 (camera-frame z > 0). */
int mvk_unproject_depth(const uint16_t* depth, int nv, int h, int w, const double* cam_inv_host,
                        const float* poses, double* xyz, uint8_t* valid, void* stream);

/* Exact brute-force k-NN in float64 (LDS-tiled): queries [nq,3] f32 (promoted to f64 like
 * sklearn does), keys [nk,3] f64 with key_valid [nk] uint8 (NULL = all valid);
 * out_idx [nq,k] int64 = key indices ascending by distance (ties: ascending index), k <= 8. */
int64_t mvk_knn_workspace(int64_t nq, int64_t nk, int k);
int mvk_knn_f64(const float* queries, int64_t nq, const double* keys, const uint8_t* key_valid,
                int64_t nk, int k, int64_t* out_idx, void* workspace, int64_t workspace_bytes,
                void* stream);

/* Fused input of FeatureAggregation for one sphere: X [C+4, np*k] channel-major =
 * {2D features of the k nearest pixels | xyz(pixel) - xyz(point) | squared distance}
 * (mvpnet/models/mvpnet_3d.py:54-58 + the two group_points calls of architectures_sphere.py:266-274).
 * feature_2d [nv,C,h*w] f32 (the encoder's native layout), image_xyz [nv*h*w,3] f32, knn [np,k] int64
 * flat pixel indices, points [np,3] f32. */
int mvk_fa_gather_fwd(const float* feature_2d, const float* image_xyz, const int64_t* knn,
                      const float* points, int C, int nv, int64_t hw, int64_t np, int k, float* X,
                      void* stream);
/* The same reading a channels-last feature map ([nv,h,w,C] in memory: what the frozen encoder's convolutions
 * produce) in place when channels_last != 0. */
int mvk_fa_gather_fwd_ex(const float* feature_2d, int channels_last, const float* image_xyz, const int64_t* knn,
                         const float* points, int C, int nv, int64_t hw, int64_t np, int k, float* X, void* stream);

/* ---------------- sphere extraction + sampling potentials ---- */

/* All points of pts [N,3] f32 within `radius` of center_host (HOST double[3]): float64 rdist <= r^2
 * like sklearn KDTree.query_radius (KPConv-PyTorch/datasets/ScanNet_sphere_color.py:571-573,592-597).
 * out_idx [cap N] int64 ascending, out_d2 [cap N] f64 or NULL, *count_dev (DEVICE int64). */
int64_t mvk_ball_query_workspace(int64_t N);
int mvk_ball_query(const float* pts, int64_t N, const double* center_host, double radius,
                   int64_t* out_idx, double* out_d2, int64_t* count_dev, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* potentials[i] += (1 - d2/r^2)^2 for the points of the ball (Tukey weights, :576-582). */
int mvk_tukey_update(const float* pts, int64_t N, const double* center_host, double radius,
                     double* potentials, void* stream);

/* group_points forward: points [B,C,N1] f32, index [B,N2,K] int64 -> out [B,C,N2,K]. */
int mvk_group_points_fwd(const float* points, const int64_t* index, int B, int C, int64_t N1,
                         int64_t N2, int K, float* out, void* stream);
/* backward: grad_in [B,C,N1] (zero-initialised by the caller) += scatter of grad_out. */
int mvk_group_points_bwd(const float* grad_out, const int64_t* index, int B, int C, int64_t N1,
                         int64_t N2, int K, float* grad_in, void* stream);
/* float64 twins of the two entry points above (group_points_kernel.cu:60,130 dispatch float and double). */
int mvk_group_points_fwd_f64(const double* points, const int64_t* index, int B, int C, int64_t N1, int64_t N2, int K,
                             double* out, void* stream);
int mvk_group_points_bwd_f64(const double* grad_out, const int64_t* index, int B, int C, int64_t N1, int64_t N2, int K,
                             double* grad_in, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVKPCONV_H */
