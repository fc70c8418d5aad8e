/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread, IEEE float32 without FMA
 * contraction -- build with -ffp-contract=off) of the integer / index part of
 * the MV-KPConv hot path. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Parity status: PINNED -- tests/test_oracle_vs_golden.py checks every function
 * against golden vectors captured from the compiled reference core
 * (oracle/_ref/libref.so, built by oracle/Makefile from /root/reference) and,
 * for k-NN, from scikit-learn's BallTree (the reference's third-party
 * dependency, requirements.txt:108 pins 0.24.1; sklearn 1.7.2 is what is
 * installed here -- exact k-NN is version independent except on exact ties).
 *
 * Reference lines each function follows (paths relative to
 * /root/reference/KPConv-PyTorch/):
 *   orc_grid_subsample        cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
 *                             + grid_subsampling.h:10-80 (SampledData)
 *                             + cpp_utils/cloud/cloud.cpp:27-67 (min_point/max_point)
 *                             + libstdc++ unordered_map iteration order
 *   orc_grid_subsample_batch  .../grid_subsampling.cpp:109-211
 *   orc_radius_neighbors_batch cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332
 *                             + cpp_utils/nanoflann/nanoflann.hpp:423-444 (L2_Simple metric),
 *                               :220-255 (RadiusResultSet: dist < radius), :1279-1290 (sorted)
 *   orc_knn_f64               datasets/ScanNet_sphere_color.py:448-449 (sklearn NearestNeighbors,
 *                             ball_tree, float64, ascending distance)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mvk_prime_list.h"

/* ------------------------------------------------------------------ */
/* libstdc++ _Hashtable<size_t/int, ...> order emulation                */
/* (identity hash, max_load_factor 1, unique keys, one singly linked     */
/*  list shared by all buckets; bits/hashtable.h _M_insert_bucket_begin, */
/*  _M_rehash_aux(unique), hashtable_c++0x.cc _M_need_rehash)            */
/* ------------------------------------------------------------------ */

#define HM_EMPTY (-2)
#define HM_BEFORE_BEGIN (-1)

typedef struct {
  uint64_t* key;   /* per node */
  int64_t* next;   /* per node, -1 = end of list */
  int64_t* bucket; /* per bucket: node BEFORE the bucket's first node, HM_BEFORE_BEGIN, or HM_EMPTY */
  uint64_t nb;
  int64_t head;    /* before_begin.next */
  uint64_t count, cap;
  uint64_t next_resize;
} hmap;

static void hm_init(hmap* m) {
  memset(m, 0, sizeof(*m));
  m->nb = 1;
  m->bucket = (int64_t*)malloc(sizeof(int64_t));
  m->bucket[0] = HM_EMPTY;
  m->head = -1;
  m->cap = 16;
  m->key = (uint64_t*)malloc(m->cap * sizeof(uint64_t));
  m->next = (int64_t*)malloc(m->cap * sizeof(int64_t));
}

static void hm_free(hmap* m) {
  free(m->key);
  free(m->next);
  free(m->bucket);
}

static int64_t hm_find(const hmap* m, uint64_t k) {
  uint64_t b = k % m->nb;
  int64_t prev = m->bucket[b];
  if (prev == HM_EMPTY) return -1;
  int64_t p = (prev == HM_BEFORE_BEGIN) ? m->head : m->next[prev];
  while (p >= 0) {
    if (m->key[p] == k) return p;
    if (m->key[p] % m->nb != b) break;
    p = m->next[p];
  }
  return -1;
}

static void hm_insert_bucket_begin(hmap* m, int64_t* bucket, uint64_t nb, uint64_t b, int64_t node) {
  if (bucket[b] != HM_EMPTY) {
    int64_t prev = bucket[b];
    if (prev == HM_BEFORE_BEGIN) {
      m->next[node] = m->head;
      m->head = node;
    } else {
      m->next[node] = m->next[prev];
      m->next[prev] = node;
    }
  } else {
    m->next[node] = m->head;
    m->head = node;
    if (m->next[node] >= 0) bucket[m->key[m->next[node]] % nb] = node;
    bucket[b] = HM_BEFORE_BEGIN;
  }
}

static void hm_rehash(hmap* m, uint64_t nb2) {
  int64_t* nbk = (int64_t*)malloc(nb2 * sizeof(int64_t));
  for (uint64_t i = 0; i < nb2; i++) nbk[i] = HM_EMPTY;
  int64_t p = m->head;
  m->head = -1;
  uint64_t bbegin_bkt = 0;
  while (p >= 0) {
    int64_t nxt = m->next[p];
    uint64_t b = m->key[p] % nb2;
    if (nbk[b] == HM_EMPTY) {
      m->next[p] = m->head;
      m->head = p;
      nbk[b] = HM_BEFORE_BEGIN;
      if (m->next[p] >= 0) nbk[bbegin_bkt] = p;
      bbegin_bkt = b;
    } else {
      int64_t prev = nbk[b];
      if (prev == HM_BEFORE_BEGIN) {
        m->next[p] = m->head;
        m->head = p;
      } else {
        m->next[p] = m->next[prev];
        m->next[prev] = p;
      }
    }
    p = nxt;
  }
  free(m->bucket);
  m->bucket = nbk;
  m->nb = nb2;
}

/* insert a key known to be absent; returns node id (= insertion rank) */
static int64_t hm_insert_new(hmap* m, uint64_t k) {
  /* _Prime_rehash_policy::_M_need_rehash(n_bkt, n_elt, 1) */
  if (m->count + 1 > m->next_resize) {
    uint64_t want = m->count + 1;
    if (m->next_resize == 0 && want < 11) want = 11;
    double min_bkts = (double)want / 1.0;
    if (min_bkts >= (double)m->nb) {
      uint64_t n = (uint64_t)floor(min_bkts) + 1;
      if (n < m->nb * 2) n = m->nb * 2;
      /* n >= 12 always here; the fast table gives 13 for n in {12,13} */
      uint64_t nb2 = (n <= 13) ? 13 : mvk_next_bkt(n);
      m->next_resize = (uint64_t)floor((double)nb2 * 1.0);
      hm_rehash(m, nb2);
    } else {
      m->next_resize = (uint64_t)floor((double)m->nb * 1.0);
    }
  }
  if (m->count == m->cap) {
    m->cap *= 2;
    m->key = (uint64_t*)realloc(m->key, m->cap * sizeof(uint64_t));
    m->next = (int64_t*)realloc(m->next, m->cap * sizeof(int64_t));
  }
  int64_t node = (int64_t)m->count++;
  m->key[node] = k;
  hm_insert_bucket_begin(m, m->bucket, m->nb, k % m->nb, node);
  return node;
}

/* ------------------------------------------------------------------ */
/* grid subsampling                                                     */
/* ------------------------------------------------------------------ */

typedef struct {
  int n;
  int cap;
  int* lab;
  int* cnt;
} labhist; /* first-seen ordered (label,count) list for one voxel, one label column */

static void lh_add(labhist* h, int l) {
  for (int i = 0; i < h->n; i++)
    if (h->lab[i] == l) {
      h->cnt[i]++;
      return;
    }
  if (h->n == h->cap) {
    h->cap = h->cap ? h->cap * 2 : 4;
    h->lab = (int*)realloc(h->lab, h->cap * sizeof(int));
    h->cnt = (int*)realloc(h->cnt, h->cap * sizeof(int));
  }
  h->lab[h->n] = l;
  h->cnt[h->n] = 1;
  h->n++;
}

/* arg-max label: first maximum in the iteration order of unordered_map<int,int>
 * (grid_subsampling.cpp:100-101). hash<int>(v) = (size_t)v. */
static int lh_vote(const labhist* h) {
  hmap m;
  hm_init(&m);
  for (int i = 0; i < h->n; i++) hm_insert_new(&m, (uint64_t)(size_t)(long)h->lab[i]);
  int best = -1, bestc = -1;
  for (int64_t p = m.head; p >= 0; p = m.next[p]) {
    int c = h->cnt[p];
    if (c > bestc) { /* max_element keeps the first max */
      bestc = c;
      best = h->lab[p];
    }
  }
  hm_free(&m);
  return best;
}

long orc_grid_subsample(const float* pts, long N, const float* feats, int fdim,
                        const int* labels, int ldim, float dl, float* out_pts,
                        float* out_feats, int* out_labels) {
  if (N <= 0) return 0;
  int use_f = feats != NULL && fdim > 0, use_l = labels != NULL && ldim > 0;
  /* cloud.cpp:27-67 */
  float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
  for (long i = 0; i < N; i++)
    for (int c = 0; c < 3; c++) {
      float v = pts[3 * i + c];
      if (v < mn[c]) mn[c] = v;
      if (v > mx[c]) mx[c] = v;
    }
  /* :27  originCorner = floor(minCorner * (1/sampleDl)) * sampleDl */
  float inv = 1 / dl;
  float org[3];
  for (int c = 0; c < 3; c++) {
    float t = mn[c] * inv;
    org[c] = floorf(t) * dl;
  }
  /* :30-31 */
  size_t NX = (size_t)floorf((mx[0] - org[0]) / dl) + 1;
  size_t NY = (size_t)floorf((mx[1] - org[1]) / dl) + 1;

  hmap m;
  hm_init(&m);
  long cap = 1024, M = 0;
  int* count = (int*)malloc(cap * sizeof(int));
  float* sum = (float*)malloc(cap * 3 * sizeof(float));
  float* fsum = use_f ? (float*)calloc(cap * fdim, sizeof(float)) : NULL;
  labhist* lh = use_l ? (labhist*)calloc(cap * ldim, sizeof(labhist)) : NULL;

  for (long i = 0; i < N; i++) {
    const float* p = pts + 3 * i;
    size_t iX = (size_t)floorf((p[0] - org[0]) / dl);
    size_t iY = (size_t)floorf((p[1] - org[1]) / dl);
    size_t iZ = (size_t)floorf((p[2] - org[2]) / dl);
    size_t key = iX + NX * iY + NX * NY * iZ;
    int64_t v = hm_find(&m, key);
    if (v < 0) {
      v = hm_insert_new(&m, key);
      if (v >= cap) {
        long ncap = cap * 2;
        count = (int*)realloc(count, ncap * sizeof(int));
        sum = (float*)realloc(sum, ncap * 3 * sizeof(float));
        if (use_f) {
          fsum = (float*)realloc(fsum, ncap * fdim * sizeof(float));
          memset(fsum + cap * fdim, 0, (ncap - cap) * fdim * sizeof(float));
        }
        if (use_l) {
          lh = (labhist*)realloc(lh, ncap * ldim * sizeof(labhist));
          memset(lh + cap * ldim, 0, (ncap - cap) * ldim * sizeof(labhist));
        }
        cap = ncap;
      }
      count[v] = 0;
      sum[3 * v] = sum[3 * v + 1] = sum[3 * v + 2] = 0.0f;
      M++;
    }
    /* grid_subsampling.h:74-79 (and :38-71 for features / labels) */
    count[v] += 1;
    sum[3 * v] += p[0];
    sum[3 * v + 1] += p[1];
    sum[3 * v + 2] += p[2];
    if (use_f)
      for (int c = 0; c < fdim; c++) fsum[v * fdim + c] += feats[i * fdim + c];
    if (use_l)
      for (int c = 0; c < ldim; c++) lh_add(&lh[v * ldim + c], labels[i * ldim + c]);
  }

  /* :81-104: iterate the map */
  long o = 0;
  for (int64_t v = m.head; v >= 0; v = m.next[v], o++) {
    float a = (float)(1.0 / (double)count[v]); /* :87 + cloud.h:120 */
    out_pts[3 * o] = sum[3 * v] * a;
    out_pts[3 * o + 1] = sum[3 * v + 1] * a;
    out_pts[3 * o + 2] = sum[3 * v + 2] * a;
    if (use_f) {
      float cf = (float)count[v];
      for (int c = 0; c < fdim; c++) out_feats[o * fdim + c] = fsum[v * fdim + c] / cf; /* :90-94 */
    }
    if (use_l)
      for (int c = 0; c < ldim; c++) out_labels[o * ldim + c] = lh_vote(&lh[v * ldim + c]);
  }
  if (use_l) {
    for (long i = 0; i < M * ldim; i++) {
      free(lh[i].lab);
      free(lh[i].cnt);
    }
    free(lh);
  }
  free(fsum);
  free(sum);
  free(count);
  hm_free(&m);
  return M;
}

/* grid_subsampling.cpp:109-211. The reference's class-slice end iterator
 * (:157-158) is only right for ldim == 1; this restatement slices correctly
 * for every ldim (documented deviation, unreachable from the network path). */
long orc_grid_subsample_batch(const float* pts, long N, const float* feats, int fdim,
                              const int* labels, int ldim, const int* lens, int B,
                              float dl, int max_p, float* out_pts, float* out_feats,
                              int* out_labels, int* out_lens) {
  if (max_p < 1) max_p = (int)N;
  long sum_b = 0, M = 0;
  for (int b = 0; b < B; b++) {
    long n = lens[b];
    float* tp = (float*)malloc((n > 0 ? n : 1) * 3 * sizeof(float));
    float* tf = (feats && fdim > 0) ? (float*)malloc((n > 0 ? n : 1) * fdim * sizeof(float)) : NULL;
    int* tl = (labels && ldim > 0) ? (int*)malloc((n > 0 ? n : 1) * ldim * sizeof(int)) : NULL;
    long m = orc_grid_subsample(pts + 3 * sum_b, n, feats ? feats + sum_b * fdim : NULL, fdim,
                                labels ? labels + sum_b * ldim : NULL, ldim, dl, tp, tf, tl);
    if (m > max_p) m = max_p;
    memcpy(out_pts + 3 * M, tp, m * 3 * sizeof(float));
    if (tf) memcpy(out_feats + M * fdim, tf, m * fdim * sizeof(float));
    if (tl) memcpy(out_labels + M * ldim, tl, m * ldim * sizeof(int));
    out_lens[b] = (int)m;
    M += m;
    sum_b += n;
    free(tp);
    free(tf);
    free(tl);
  }
  return M;
}

/* ------------------------------------------------------------------ */
/* radius neighbours (brute force restatement of the result contract)   */
/* ------------------------------------------------------------------ */

typedef struct {
  float d2;
  int idx;
} cand;

static int cand_cmp(const void* a, const void* b) {
  const cand* x = (const cand*)a;
  const cand* y = (const cand*)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx); /* ties: ascending index (documented contract) */
}

/* Returns the output width W = max_i |row_i|. If out != NULL it must hold
 * Nq*W ints (call once with NULL to learn W). Pad value = Ns (neighbors.cpp:324). */
int orc_radius_neighbors_batch(const float* q, long Nq, const float* s, long Ns,
                               const int* ql, const int* sl, int B, float radius, int* out) {
  float r2 = radius * radius; /* neighbors.cpp:226 */
  int W = 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1 && out == NULL) break;
    long q0 = 0, s0 = 0;
    cand* buf = NULL;
    long bcap = 0;
    for (int b = 0; b < B; b++) {
      long nq = ql[b], ns = sl[b];
      if (ns > bcap) {
        bcap = ns;
        buf = (cand*)realloc(buf, bcap * sizeof(cand));
      }
      for (long i = q0; i < q0 + nq; i++) {
        float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
        int n = 0;
        for (long j = 0; j < ns; j++) {
          const float* p = s + 3 * (s0 + j);
          /* nanoflann.hpp:433-441: result = 0; result += diff*diff (x, y, z in order) */
          float dx = qx - p[0], dy = qy - p[1], dz = qz - p[2];
          float d2 = 0.0f;
          d2 += dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) { /* nanoflann.hpp:249-251 */
            buf[n].d2 = d2;
            buf[n].idx = (int)(s0 + j); /* neighbors.cpp:322 */
            n++;
          }
        }
        if (pass == 0) {
          if (n > W) W = n;
        } else {
          qsort(buf, n, sizeof(cand), cand_cmp);
          for (int c = 0; c < W; c++) out[i * W + c] = c < n ? buf[c].idx : (int)Ns;
        }
      }
      q0 += nq;
      s0 += ns;
    }
    free(buf);
  }
  return W;
}

/* ------------------------------------------------------------------ */
/* exact k-NN in float64 (brute force), ascending distance              */
/* ------------------------------------------------------------------ */

/* queries (nq,3) float64, keys (nk,3) float64 -> idx (nq,k) int64, d2 (nq,k) float64 (optional).
 * Squared distance accumulated x,y,z in order like sklearn's euclidean rdist. */
void orc_knn_f64(const double* q, long nq, const double* keys, long nk, int k,
                 int64_t* out_idx, double* out_d2) {
  for (long i = 0; i < nq; i++) {
    double bd[16];
    int64_t bi[16];
    int n = 0;
    for (long j = 0; j < nk; j++) {
      double dx = q[3 * i] - keys[3 * j], dy = q[3 * i + 1] - keys[3 * j + 1],
             dz = q[3 * i + 2] - keys[3 * j + 2];
      double d2 = 0.0;
      d2 += dx * dx;
      d2 += dy * dy;
      d2 += dz * dz;
      if (n < k || d2 < bd[n - 1]) {
        int p = n < k ? n : k - 1;
        while (p > 0 && bd[p - 1] > d2) {
          bd[p] = bd[p - 1];
          bi[p] = bi[p - 1];
          p--;
        }
        bd[p] = d2;
        bi[p] = j;
        if (n < k) n++;
      }
    }
    for (int c = 0; c < k; c++) {
      out_idx[i * k + c] = c < n ? bi[c] : -1;
      if (out_d2) out_d2[i * k + c] = c < n ? bd[c] : INFINITY;
    }
  }
}
