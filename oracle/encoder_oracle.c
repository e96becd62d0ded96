/* Oracle: scalar float32 port of the encoder forward pass.  TEST INFRASTRUCTURE ONLY.
 *
 * A second, independent restatement (plain loops, float32 accumulate) of the
 * graphs at reference src/bigan_cont.py:28-41 and src/bigan_classify.py:28-40,
 * used to cross-check oracle/encoder_oracle.py.  PARITY UNPINNED vs TF (see
 * oracle/__init__.py).  Weights arrive in Keras get_weights() order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

/* y[n_out] = act(concat(a[na], b[nb]) @ K + bias); act: 0 sigmoid, 1 linear */
static void dense(const float *a, int64_t na, const float *b, int64_t nb, const float *K,
                  const float *bias, int64_t n_out, int act, float *y) {
  for (int64_t j = 0; j < n_out; ++j) y[j] = bias[j];
  for (int64_t i = 0; i < na; ++i) {
    const float v = a[i];
    if (v == 0.0f) continue;
    const float *row = K + i * n_out;
    for (int64_t j = 0; j < n_out; ++j) y[j] += v * row[j];
  }
  for (int64_t i = 0; i < nb; ++i) {
    const float v = b[i];
    if (v == 0.0f) continue;
    const float *row = K + (na + i) * n_out;
    for (int64_t j = 0; j < n_out; ++j) y[j] += v * row[j];
  }
  if (act == 0)
    for (int64_t j = 0; j < n_out; ++j) y[j] = sigmoidf_(y[j]);
}

/* arch 0 = cont, 1 = cls.  w = pointers to the weight arrays in Keras order. */
int encoder_oracle_forward(int arch, int64_t G, int64_t E, const float *const *w, const float *x,
                           int64_t n, float *out) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t r = 0; r < n; ++r) {
    const float *xr = x + r * G;
    float *o = out + r * E;
    if (arch == 0) {
      const int64_t w0 = (int64_t)((double)G * 0.1), w1 = (int64_t)((double)G * 0.05);
      float *h1 = (float *)malloc(sizeof(float) * (size_t)(w0 + 1));
      float *h2 = (float *)malloc(sizeof(float) * (size_t)(w1 + 1));
      float h3[150], h4[150];
      dense(xr, G, NULL, 0, w[0], w[1], w0, 0, h1);
      dense(h1, w0, xr, G, w[2], w[3], w1, 0, h2);
      for (int64_t i = 0; i < w1; ++i)
        h2[i] = (h2[i] - w[6][i]) / sqrtf(w[7][i] + 1e-3f) * w[4][i] + w[5][i];
      dense(h2, w1, NULL, 0, w[8], w[9], 150, 0, h3);
      dense(h3, 150, NULL, 0, w[10], w[11], 150, 0, h4);
      dense(h4, 150, NULL, 0, w[12], w[13], E, 0, o);
      free(h1);
      free(h2);
    } else {
      float p[1000], h[300], h3[150];
      dense(xr, G, NULL, 0, w[0], w[1], 1000, 0, p);
      dense(p, 1000, NULL, 0, w[2], w[3], 300, 0, h);
      dense(h, 300, p, 1000, w[4], w[5], 150, 0, h3);
      dense(h3, 150, NULL, 0, w[6], w[7], E, 1, o);
      float mx = o[0];
      for (int64_t j = 1; j < E; ++j) mx = o[j] > mx ? o[j] : mx;
      float s = 0.f;
      for (int64_t j = 0; j < E; ++j) { o[j] = expf(o[j] - mx); s += o[j]; }
      for (int64_t j = 0; j < E; ++j) o[j] /= s;
    }
  }
  return 0;
}
