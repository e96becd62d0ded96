/* Oracle: exact brute-force squared-L2 kNN.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference has no kNN implementation (its
 * only neighbour-graph consumers are umap.UMAP / sklearn TSNE at reference
 * src/intercepts/plot_intercepts.py:92,176,183), so this file DEFINES the
 * semantics the HIP path must reproduce bit-for-bit:
 *
 *   dist(i,j) = sum_{t=0..d-1} ((double)x[i][t] - (double)x[j][t])^2
 *               accumulated left to right in float64, every operation rounded
 *               separately (no fused multiply-add: build with -ffp-contract=off)
 *   neighbours of i = the k rows j with the smallest (dist(i,j), j), in that
 *               order; j == i is skipped unless include_self.
 *   returned distance = (float)dist.
 */
#include <stdint.h>
#include <stdlib.h>

static double dist64(const float *a, const float *b, int d) {
  double s = 0.0;
  for (int t = 0; t < d; ++t) {
    double diff = (double)a[t] - (double)b[t];
    double sq = diff * diff;
    s = s + sq;
  }
  return s;
}

/* returns 0 on success, 1 when k exceeds the number of admissible candidates */
int knn_oracle(const float *x, int64_t n, int d, int64_t q_begin, int64_t n_q, int k,
               int include_self, int32_t *idx, float *dist32, double *dist64_out) {
  if ((int64_t)k > n - (include_self ? 0 : 1) || k < 1) return 1;
  int rc = 0;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t ql = 0; ql < n_q; ++ql) {
    const int64_t q = q_begin + ql;
    double *bd = (double *)malloc(sizeof(double) * (size_t)k);
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    int cnt = 0;
    for (int64_t j = 0; j < n; ++j) {
      if (!include_self && j == q) continue;
      const double dj = dist64(x + q * d, x + j * d, d);
      /* j increases, so on equal distance the earlier (smaller) index stays ahead */
      if (cnt == k && !(dj < bd[k - 1])) continue;
      int pos = cnt < k ? cnt : k - 1;
      while (pos > 0 && dj < bd[pos - 1]) {
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        --pos;
      }
      bd[pos] = dj;
      bi[pos] = (int32_t)j;
      if (cnt < k) ++cnt;
    }
    for (int i = 0; i < k; ++i) {
      idx[ql * k + i] = bi[i];
      dist32[ql * k + i] = (float)bd[i];
      if (dist64_out) dist64_out[ql * k + i] = bd[i];
    }
    free(bd);
    free(bi);
  }
  return rc;
}
