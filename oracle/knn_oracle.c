/*
 * oracle/knn_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * CPU restatement, in plain C, of the k-NN arithmetic of the reference:
 *   dgcnn/ops.py:8-19   k_nn(points, k)
 *       :12-13  inner_prod = matmul(M, transpose(M))          -> p_ij
 *       :14-15  squared    = reduce_sum(square(M), -1)         -> s_i
 *       :16     nn_dist    = squared + squared^T - 2*inner     -> D_ij = (s_i + s_j) - 2 p_ij
 *       :18     _, idx     = top_k(-nn_dist, k)                -> k smallest D per row, ascending,
 *                                                                  ties -> lower index first, self included
 * and of the row gather of dgcnn/ops.py:21-40 (edges).
 *
 * PARITY UNPINNED: TensorFlow 1.x (where the arithmetic actually lives) is not in
 * /root/reference and cannot be installed here, and the reference has no tests or golden
 * vectors.  The arithmetic order below is therefore *normative by
 * choice*:  s_i   = sequential sum of fl(x*x), no FMA (square and reduce_sum are separate TF ops)
 *           p_ij  = fmaf chain over c ascending starting from +0 (an FMA GEMM micro-kernel)
 *           D_ij  = fl( fl(s_i + s_j) - fl(2 p_ij) )
 * Build with -ffp-contract=off so the compiler never fuses or reassociates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float sq_norm(const float *x, int C) {
  float s = 0.0f;
  for (int c = 0; c < C; ++c) {
    float q = x[c] * x[c];
    s = s + q;
  }
  return s;
}

/* -DORACLE_NO_FMA (liboracle_nofma.so): the other chain SURVEY A.1 leaves open -- a GEMM micro-kernel WITHOUT fused multiply-add
 * (an SSE2-only TensorFlow build): p = fl(p + fl(a*b)), same order.  Not normative; tests/test_oracle.py measures how many rows'
 * neighbour lists (and how far the logits) move between the two chains -- what "parity unpinned" could cost. */
static inline float inner(const float *a, const float *b, int C) {
  float p = 0.0f;
#ifdef ORACLE_NO_FMA
  for (int c = 0; c < C; ++c) {
    float q = a[c] * b[c];
    p = p + q;
  }
#else
  for (int c = 0; c < C; ++c) p = fmaf(a[c], b[c], p);
#endif
  return p;
}

/* Full distance matrix of one cloud, D[N][N] (for tests that want to look at the values). */
void oracle_dist_f32(const float *x, int N, int C, long ldx, float *D) {
  float *s = (float *)malloc(sizeof(float) * (size_t)N);
  for (int i = 0; i < N; ++i) s[i] = sq_norm(x + (size_t)i * ldx, C);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      float p = inner(x + (size_t)i * ldx, x + (size_t)j * ldx, C);
      float t = s[i] + s[j];
      float tp = 2.0f * p;
      D[(size_t)i * N + j] = t - tp;
    }
  free(s);
}

/* One query row i of one cloud: the k best (D_ij, j) over all j, into bd/bj (caller scratch of k). */
static void knn_row(const float *xb, const float *s, int N, int C, long ldx, int k, int i,
                    float *bd, int32_t *bj) {
  int filled = 0;
  const float *xi = xb + (size_t)i * ldx;
  for (int j = 0; j < N; ++j) {
    float p = inner(xi, xb + (size_t)j * ldx, C);
    float t = s[i] + s[j];
    float tp = 2.0f * p;
    float d = t - tp;
    /* sorted insert, strict '<' so that an equal distance with larger j goes after */
    if (filled < k) {
      int t2 = filled++;
      while (t2 > 0 && d < bd[t2 - 1]) { bd[t2] = bd[t2 - 1]; bj[t2] = bj[t2 - 1]; --t2; }
      bd[t2] = d; bj[t2] = j;
    } else if (d < bd[k - 1]) {
      int t2 = k - 1;
      while (t2 > 0 && d < bd[t2 - 1]) { bd[t2] = bd[t2 - 1]; bj[t2] = bj[t2 - 1]; --t2; }
      bd[t2] = d; bj[t2] = j;
    }
  }
}

/* idx[B][N][k] int32, batch-local.  Returns 0, or -1 when N < k (TF top_k raises).
 * Rows are independent: OpenMP over i only spreads them over cores, every row's arithmetic order is unchanged. */
int oracle_knn_f32(const float *x, int B, int N, int C, long ldx, int k, int32_t *idx) {
  if (k > N || k <= 0) return -1;
  float *s = (float *)malloc(sizeof(float) * (size_t)N);
  for (int b = 0; b < B; ++b) {
    const float *xb = x + (size_t)b * N * ldx;
    for (int i = 0; i < N; ++i) s[i] = sq_norm(xb + (size_t)i * ldx, C);
#pragma omp parallel
    {
      float *bd = (float *)malloc(sizeof(float) * (size_t)k);
      int32_t *bj = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
#pragma omp for schedule(dynamic, 16)
      for (int i = 0; i < N; ++i) {
        knn_row(xb, s, N, C, ldx, k, i, bd, bj);
        memcpy(idx + ((size_t)b * N + i) * k, bj, sizeof(int32_t) * (size_t)k);
      }
      free(bd); free(bj);
    }
  }
  free(s);
  return 0;
}

/* The same for a subset of the query rows of ONE cloud (x: (N,C)): idx[nrows][k] for rows[0..nrows).
 * Lets the large-N tests (N = 65536: 4.3e9 pairs per cloud) check a seeded sample of rows in seconds. */
int oracle_knn_rows_f32(const float *x, int N, int C, long ldx, int k, const int32_t *rows, int nrows,
                        int32_t *idx) {
  if (k > N || k <= 0) return -1;
  float *s = (float *)malloc(sizeof(float) * (size_t)N);
  for (int i = 0; i < N; ++i) s[i] = sq_norm(x + (size_t)i * ldx, C);
#pragma omp parallel
  {
    float *bd = (float *)malloc(sizeof(float) * (size_t)k);
    int32_t *bj = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
#pragma omp for schedule(dynamic, 16)
    for (int r = 0; r < nrows; ++r) {
      knn_row(x, s, N, C, ldx, k, rows[r], bd, bj);
      memcpy(idx + (size_t)r * k, bj, sizeof(int32_t) * (size_t)k);
    }
    free(bd); free(bj);
  }
  free(s);
  return 0;
}

/* Distances of selected (row, candidate) pairs of one cloud, normative arithmetic: D[nrows][k]. */
void oracle_dist_pairs_f32(const float *x, int N, int C, long ldx, const int32_t *rows, int nrows,
                           const int32_t *cand, int k, float *D) {
  for (int r = 0; r < nrows; ++r) {
    const float *xi = x + (size_t)rows[r] * ldx;
    float si = sq_norm(xi, C);
    for (int m = 0; m < k; ++m) {
      const float *xj = x + (size_t)cand[(size_t)r * k + m] * ldx;
      float p = inner(xi, xj, C);
      float t = si + sq_norm(xj, C);
      float tp = 2.0f * p;
      D[(size_t)r * k + m] = t - tp;
    }
  }
}

/* E[B][N][k][2C] = concat(x_i, x_j - x_i)  (dgcnn/ops.py:30-39); exact: one subtract. */
void oracle_edges_f32(const float *x, const int32_t *idx, int B, int N, int C, int k, float *E) {
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < N; ++i) {
      const float *xi = x + ((size_t)b * N + i) * C;
      for (int m = 0; m < k; ++m) {
        const float *xj = x + ((size_t)b * N + idx[((size_t)b * N + i) * k + m]) * C;
        float *e = E + (((size_t)b * N + i) * k + m) * 2 * C;
        for (int c = 0; c < C; ++c) { e[c] = xi[c]; e[C + c] = xj[c] - xi[c]; }
      }
    }
}
