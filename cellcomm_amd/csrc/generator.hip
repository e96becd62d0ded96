// Generator forward pass (the mirror-image inference path of the BiGAN: generate_cells,
// reference src/bigan_basic.py:39-44) on the same fp32 MFMA GEMM as the encoder.
//
// Graphs at inference (Dropout = identity, BatchNormalization folded in float64 into the rows of
// the following Dense that read it):
//   cont (reference src/bigan_cont.py:7-25)       in = [encoding, random] (2E)
//     s(Dense 50)(in) -> [., in] -> s(Dense 256) -> s(Dense 256) -> BN -> [., in]
//     -> s(Dense int(0.1 G)) -> relu(Dense int(0.2 G)) -> BN -> relu(Dense G)
//   cls  (reference src/bigan_classify.py:10-25)
//     s(Dense 50)(in) -> [., in] -> s(Dense 256) -> BN -> [., in] -> s(Dense 256)
//     -> relu(Dense 1024) -> relu(Dense G)
// Weights arrive in Keras get_weights() order (20 arrays cont, 14 cls).
#include "common.h"

namespace cc {

struct GShape { int64_t rows, cols; };

struct GLayer {
  int k_idx, b_idx;     // weight slots of kernel and bias
  int bn_idx;           // first slot of a BatchNormalization folded into segment 1 (-1: none)
  int in1, in2;         // buffer ids of the two K-segments (in2 = -1: single input)
  int out;              // output buffer id
  int act;
};

struct Generator {
  int arch = 0, G = 0, E = 0;
  int64_t batch = 2048;
  std::vector<GShape> shapes;
  std::vector<std::vector<float>> w;
  std::vector<bool> have;
  std::vector<GLayer> layers;
  std::vector<int64_t> width;          // logical width of every buffer (0 = network input, last = output)
  std::vector<PackedWeight> packed;
  bool dirty = true;
  int64_t ws_rows = 0;
  std::vector<float *> buf;
  std::vector<int64_t> ld;
};

static int64_t padk(int64_t v) { return v == 0 ? GEMM_BK : round_up(v, GEMM_BK); }

static void generator_plan(Generator *g) {
  const int64_t G = g->G, in = 2 * (int64_t)g->E;
  if (g->arch == CC_ARCH_CONT) {
    const int64_t w0 = (int64_t)((double)G * 0.2), w1 = (int64_t)((double)G * 0.1);   // bigan_cont.py:8
    //            0   1   2    3    4   5   6
    g->width = {in, 50, 256, 256, w1, w0, G};
    g->shapes = {{in, 50}, {50, -1}, {50 + in, 256}, {256, -1}, {256, 256}, {256, -1},
                 {256, -1}, {256, -1}, {256, -1}, {256, -1},                       // BN(256)
                 {256 + in, w1}, {w1, -1}, {w1, w0}, {w0, -1},
                 {w0, -1}, {w0, -1}, {w0, -1}, {w0, -1},                           // BN(w0)
                 {w0, G}, {G, -1}};
    g->layers = {{0, 1, -1, 0, -1, 1, ACT_SIGMOID},  {2, 3, -1, 1, 0, 2, ACT_SIGMOID}, {4, 5, -1, 2, -1, 3, ACT_SIGMOID},
                 {10, 11, 6, 3, 0, 4, ACT_SIGMOID},  {12, 13, -1, 4, -1, 5, ACT_RELU},  {18, 19, 14, 5, -1, 6, ACT_RELU}};
  } else {
    g->width = {in, 50, 256, 256, 1024, G};
    g->shapes = {{in, 50}, {50, -1}, {50 + in, 256}, {256, -1},
                 {256, -1}, {256, -1}, {256, -1}, {256, -1},                       // BN(256)
                 {256 + in, 256}, {256, -1}, {256, 1024}, {1024, -1}, {1024, G}, {G, -1}};
    g->layers = {{0, 1, -1, 0, -1, 1, ACT_SIGMOID}, {2, 3, -1, 1, 0, 2, ACT_SIGMOID}, {8, 9, 4, 2, 0, 3, ACT_SIGMOID},
                 {10, 11, -1, 3, -1, 4, ACT_RELU},  {12, 13, -1, 4, -1, 5, ACT_RELU}};
  }
  g->w.assign(g->shapes.size(), {});
  g->have.assign(g->shapes.size(), false);
  g->packed.assign(g->layers.size(), PackedWeight());
}

static int generator_pack(Generator *g) {
  for (size_t i = 0; i < g->have.size(); ++i)
    if (!g->have[i]) { set_error("generator weight %zu has not been set", i); return CC_ESTATE; }
  for (size_t li = 0; li < g->layers.size(); ++li) {
    const GLayer &L = g->layers[li];
    const int64_t K1 = g->width[L.in1], K2 = L.in2 >= 0 ? g->width[L.in2] : 0, N = g->width[L.out];
    const std::vector<float> &K = g->w[L.k_idx], &B = g->w[L.b_idx];
    int rc;
    if (L.bn_idx < 0) {
      if ((rc = pack_weight(K.data(), B.data(), K1, K2, N, &g->packed[li]))) return rc;
      continue;
    }
    // y = BN(h)·K[0:K1] + other·K[K1:] + b  with BN(h) = h·inv + shift (keras epsilon 1e-3)
    const float *gamma = g->w[L.bn_idx].data(), *beta = g->w[L.bn_idx + 1].data();
    const float *mean = g->w[L.bn_idx + 2].data(), *var = g->w[L.bn_idx + 3].data();
    std::vector<float> k2(K), b2(N);
    std::vector<double> acc(N);
    for (int64_t j = 0; j < N; ++j) acc[j] = (double)B[j];
    for (int64_t i = 0; i < K1; ++i) {
      const double inv = (double)gamma[i] / std::sqrt((double)var[i] + 1e-3);
      const double shift = (double)beta[i] - (double)mean[i] * inv;
      for (int64_t j = 0; j < N; ++j) {
        acc[j] += shift * (double)K[i * N + j];
        k2[i * N + j] = (float)(inv * (double)K[i * N + j]);
      }
    }
    for (int64_t j = 0; j < N; ++j) b2[j] = (float)acc[j];
    if ((rc = pack_weight(k2.data(), b2.data(), K1, K2, N, &g->packed[li]))) return rc;
  }
  g->dirty = false;
  return CC_OK;
}

static void generator_free_ws(Generator *g) {
  for (float *p : g->buf)
    if (p) (void)hipFree(p);
  g->buf.clear();
  g->ld.clear();
  g->ws_rows = 0;
}

static int generator_ensure_ws(Generator *g, int64_t rows) {
  if (rows <= g->ws_rows) return CC_OK;
  generator_free_ws(g);
  g->buf.assign(g->width.size(), nullptr);
  g->ld.assign(g->width.size(), 0);
  for (size_t i = 0; i < g->width.size(); ++i) {
    g->ld[i] = (i + 1 == g->width.size()) ? round_up(g->width[i] ? g->width[i] : 1, 4) : padk(g->width[i]);
    const size_t bytes = sizeof(float) * (size_t)(rows * g->ld[i]);
    CC_HIP(hipMalloc(&g->buf[i], bytes));
    CC_HIP(hipMemsetAsync(g->buf[i], 0, bytes, stream()));
  }
  g->ws_rows = rows;
  return CC_OK;
}

}  // namespace cc

using namespace cc;

extern "C" {

int cc_generator_create(int arch, int G, int E, void **gen) {
  CC_REQUIRE(gen != nullptr, "gen is NULL");
  CC_REQUIRE(arch == CC_ARCH_CONT || arch == CC_ARCH_CLS, "unknown generator arch %d", arch);
  CC_REQUIRE(G >= 1 && E >= 1, "gene_size=%d and encoding_size=%d must be >= 1", G, E);
  Generator *g = new Generator();
  g->arch = arch; g->G = G; g->E = E;
  generator_plan(g);
  *gen = g;
  return CC_OK;
}

void cc_generator_free(void *gen) {
  if (!gen) return;
  Generator *g = static_cast<Generator *>(gen);
  generator_free_ws(g);
  for (auto &p : g->packed) free_weight(&p);
  delete g;
}

int cc_generator_num_weights(void *gen) {
  if (!gen) { set_error("generator handle is NULL"); return -1; }
  return (int)static_cast<Generator *>(gen)->shapes.size();
}

int cc_generator_weight_shape(void *gen, int idx, int64_t *rows, int64_t *cols) {
  CC_REQUIRE(gen != nullptr, "generator handle is NULL");
  CC_REQUIRE(rows && cols, "rows / cols is NULL");
  Generator *g = static_cast<Generator *>(gen);
  CC_REQUIRE(idx >= 0 && idx < (int)g->shapes.size(), "weight index %d out of range", idx);
  *rows = g->shapes[idx].rows;
  *cols = g->shapes[idx].cols;
  return CC_OK;
}

int cc_generator_set_weights(void *gen, int idx, const float *host, int64_t rows, int64_t cols) {
  CC_REQUIRE(gen != nullptr, "generator handle is NULL");
  Generator *g = static_cast<Generator *>(gen);
  CC_REQUIRE(idx >= 0 && idx < (int)g->shapes.size(), "weight index %d out of range", idx);
  const GShape s = g->shapes[idx];
  CC_REQUIRE(rows == s.rows && cols == s.cols, "weight %d has shape (%lld,%lld), expected (%lld,%lld)", idx,
             (long long)rows, (long long)cols, (long long)s.rows, (long long)s.cols);
  CC_REQUIRE(host != nullptr || s.rows * (s.cols < 0 ? 1 : s.cols) == 0, "weight %d: host pointer is NULL", idx);
  g->w[idx].assign(host, host + s.rows * (s.cols < 0 ? 1 : s.cols));
  g->have[idx] = true;
  g->dirty = true;
  return CC_OK;
}

int cc_generate_dense(void *gen, const float *host_in, int64_t n, float *host_out) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_REQUIRE(gen != nullptr, "generator handle is NULL");
  Generator *g = static_cast<Generator *>(gen);
  CC_REQUIRE(n >= 0, "negative row count");
  CC_REQUIRE((host_in != nullptr && host_out != nullptr) || n == 0, "host_in / host_out is NULL");
  int rc;
  if (g->dirty && (rc = generator_pack(g))) return rc;
  if (n == 0) return CC_OK;
  const int64_t B = n < g->batch ? n : g->batch;
  if ((rc = generator_ensure_ws(g, B))) return rc;
  const int64_t in_w = g->width[0], out_id = (int64_t)g->width.size() - 1;
  for (int64_t off = 0; off < n; off += B) {
    const int64_t m = (n - off < B) ? n - off : B;
    CC_HIP(hipMemcpy2DAsync(g->buf[0], sizeof(float) * (size_t)g->ld[0], host_in + off * in_w,
                            sizeof(float) * (size_t)in_w, sizeof(float) * (size_t)in_w, (size_t)m,
                            hipMemcpyHostToDevice, stream()));
    for (size_t li = 0; li < g->layers.size(); ++li) {
      const GLayer &L = g->layers[li];
      const float *a2 = L.in2 >= 0 ? g->buf[L.in2] : nullptr;
      const int64_t lda2 = L.in2 >= 0 ? g->ld[L.in2] : 0;
      if ((rc = launch_gemm(g->buf[L.in1], g->ld[L.in1], a2, lda2, m, g->packed[li], g->buf[L.out], g->ld[L.out],
                            L.act, li + 1 == g->layers.size() ? CC_PROF_GEMM_L1 : CC_PROF_GEMM_TAIL)))
        return rc;
    }
    CC_HIP(hipMemcpy2DAsync(host_out + off * g->G, sizeof(float) * (size_t)g->G, g->buf[out_id],
                            sizeof(float) * (size_t)g->ld[out_id], sizeof(float) * (size_t)g->G, (size_t)m,
                            hipMemcpyDeviceToHost, stream()));
    CC_HIP(hipStreamSynchronize(stream()));
  }
  return CC_OK;
}

}  // extern "C"
