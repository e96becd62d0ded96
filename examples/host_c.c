/* A host program in plain C on the C ABI of libcellcomm_hip.so: no Python, no PyTorch.
 *
 *   gcc -O2 -Iinclude examples/host_c.c -o host_c -Lcellcomm_amd -lcellcomm_hip -Wl,-rpath,$PWD/cellcomm_amd -lm
 *   ./host_c [n_cells] [n_genes] [latent] [k]
 *
 * Builds a seeded sparse count matrix and seeded encoder weights (a tiny LCG, so that
 * tests/test_gpu_host_c.py can rebuild exactly the same inputs in NumPy), runs
 * CSR -> encoder -> kNN on the GPU and prints checksums of the encodings and the neighbour graph.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "cellcomm_hip.h"

#define CK(call)                                                           \
  do {                                                                     \
    if ((call) != 0) {                                                     \
      fprintf(stderr, "%s failed: %s\n", #call, cc_last_error());          \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static uint64_t lcg_state;
static uint32_t lcg(void) {   /* Knuth MMIX constants; the upper 32 bits */
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(lcg_state >> 32);
}
static float lcg_unit(void) { return (float)(lcg() >> 8) * (1.0f / 16777216.0f); }   /* [0, 1) exactly in fp32 */

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2000;
  const int G = argc > 2 ? atoi(argv[2]) : 3000;
  const int E = argc > 3 ? atoi(argv[3]) : 32;
  const int k = argc > 4 ? atoi(argv[4]) : 15;

  if (cc_device_count() == 0) { fprintf(stderr, "no HIP device\n"); return 2; }
  CK(cc_init(0));

  /* sparse counts: every 16th-ish column of a row is non-zero, value 1..4 */
  int64_t *indptr = malloc(sizeof(int64_t) * (size_t)(n + 1));
  int32_t *indices = malloc(sizeof(int32_t) * (size_t)n * (size_t)(G / 8 + 1));
  float *vals = malloc(sizeof(float) * (size_t)n * (size_t)(G / 8 + 1));
  lcg_state = 12345;
  int64_t nnz = 0;
  for (int64_t i = 0; i < n; ++i) {
    indptr[i] = nnz;
    int c = (int)(lcg() % 16);
    while (c < G) {
      indices[nnz] = c;
      vals[nnz] = (float)(1 + lcg() % 4);
      ++nnz;
      c += 1 + (int)(lcg() % 31);
    }
  }
  indptr[n] = nnz;

  void *enc = NULL, *csr = NULL;
  CK(cc_encoder_create(CC_ARCH_CONT, G, E, CC_DTYPE_F32, &enc));
  const int nw = cc_encoder_num_weights(enc);
  lcg_state = 777;
  for (int w = 0; w < nw; ++w) {
    int64_t rows, cols;
    CK(cc_encoder_weight_shape(enc, w, &rows, &cols));
    const int64_t count = cols < 0 ? rows : rows * cols;
    float *buf = malloc(sizeof(float) * (size_t)(count > 0 ? count : 1));
    /* kernels: uniform in +-sqrt(6/(in+out)); vectors: uniform in [0.5, 1.5) (keeps BN variances positive) */
    const float lim = cols < 0 ? 0.f : sqrtf(6.0f / (float)(rows + cols));
    for (int64_t j = 0; j < count; ++j) buf[j] = cols < 0 ? 0.5f + lcg_unit() : (2.0f * lcg_unit() - 1.0f) * lim;
    CK(cc_encoder_set_weights(enc, w, buf, rows, cols));
    free(buf);
  }
  CK(cc_csr_upload(indptr, indices, vals, n, nnz, G, &csr));

  float *z = malloc(sizeof(float) * (size_t)n * (size_t)E);
  int32_t *idx = malloc(sizeof(int32_t) * (size_t)n * (size_t)k);
  float *dist = malloc(sizeof(float) * (size_t)n * (size_t)k);
  CK(cc_encode(enc, csr, NULL, 0, n, z));
  CK(cc_knn_l2(z, n, E, k, 0, idx, dist));

  double zsum = 0.0, dsum = 0.0;
  uint64_t ihash = 1469598103934665603ull;   /* FNV-1a over the neighbour ids */
  for (int64_t i = 0; i < n * E; ++i) zsum += (double)z[i];
  for (int64_t i = 0; i < n * k; ++i) {
    dsum += (double)dist[i];
    ihash = (ihash ^ (uint64_t)(uint32_t)idx[i]) * 1099511628211ull;
  }
  printf("cells %lld genes %d latent %d k %d nnz %lld\n", (long long)n, G, E, k, (long long)nnz);
  printf("encodings_sum %.9e\nknn_dist_sum %.9e\nknn_idx_fnv %016llx\nfallback_queries %lld\n", zsum, dsum,
         (unsigned long long)ihash, (long long)cc_knn_last_fallback_count());

  cc_csr_free(csr);
  cc_encoder_free(enc);
  free(indptr); free(indices); free(vals); free(z); free(idx); free(dist);
  return 0;
}
