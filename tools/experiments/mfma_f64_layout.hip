// Probe: which D[i][j] does lane l, register r of v_mfma_f64_16x16x4_f64 hold?  (gfx950)
// A[i][k] supplied by lane i + 16k, B[k][j] by lane j + 16k (assumed; the probe checks that too).
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;
__global__ void probe(double *out) {
  const int l = threadIdx.x, i = l & 15, k = l >> 4;
  // A[i][k] = (i + 1) * (k == 0), B[k][j] = 100 * (j + 1) * (k == 0) -> D[i][j] = 100 (i+1)(j+1)
  const double a = k == 0 ? (double)(i + 1) : 0.0, b = k == 0 ? 100.0 * (i + 1) : 0.0;
  f64x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
  // second product: only k = 2 populated, to check the k <-> lane-group pairing of A and B
  const double a2 = k == 2 ? (double)(i + 1) : 0.0, b2 = k == 2 ? 100.0 * (i + 1) : 0.0;
  f64x4 acc2 = {0, 0, 0, 0};
  acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc2, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = acc2[r];
}
int main() {
  double *d, h[512];
  hipMalloc(&d, sizeof h);
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      const int v = (int)(h[l * 4 + r] / 100.0 + 0.5);
      int fi = -1, fj = -1;
      for (int i = 1; i <= 16 && fi < 0; ++i)
        if (v % i == 0 && v / i >= 1 && v / i <= 16 && (l & 15) + 1 == v / i) { fi = i - 1; fj = v / i - 1; }
      printf("  r%d: %6.0f (i=%d,j=%d)", r, h[l * 4 + r], fi, fj);
    }
    printf("  | k2: %6.0f\n", h[256 + l * 4]);
  }
  return 0;
}
