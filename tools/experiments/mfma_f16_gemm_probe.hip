// Probe (round 3): bare v_mfma_f32_16x16x32_f16 loops shaped like the fp16 GEMM's inner block — 32 independent
// accumulators (8 x 4 blocks of 16 x 16), A fragments reused over 4 B fragments, operands in registers, no memory
// traffic — at 1 and 2 waves per SIMD.  What do the matrix cores sustain on this instruction with nothing else going on?
//   hipcc -O3 --offload-arch=gfx950 mfma_f16_gemm_probe.hip -o bin/mfma_f16_gemm_probe && bin/mfma_f16_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

__global__ __launch_bounds__(512) void probe(const float *__restrict__ in, float *__restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  h8 a[8], b[4];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[t][i] = (_Float16)in[(tid * 128 + t * 8 + i) & 0xfffff];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) b[t][i] = (_Float16)in[(tid * 128 + 64 + t * 8 + i) & 0xfffff];
  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
  }
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][3];
  out[tid] = sum;
}

int main() {
  const int n = 1 << 20;
  std::vector<float> h(n);
  for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *din, *dout;
  hipMalloc(&din, n * 4);
  hipMalloc(&dout, 1024 * 512 * 4 * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep2 = 0; rep2 < 2; ++rep2)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 128 * wps, iters = 4000;   // 512 threads = 8 waves = 2 per SIMD per block
      const double flop = (double)blocks * 8 * iters * 32 * 16384.0;
      for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe, dim3(blocks * 2), dim3(512), 0, 0, din, dout, iters);
      (void)hipEventRecord(e0);
      for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(probe, dim3(blocks * 2), dim3(512), 0, 0, din, dout, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("16x16x32 f16, 32 independent accumulators, %d waves/SIMD: %7.1f TFLOP/s (%.0f %% of 2500)\n", wps * 2 / 2 * 1,
             2 * flop * 10 / (ms * 1e-3) / 1e12, 2 * flop * 10 / (ms * 1e-3) / 1e12 / 25.0);
    }
  return 0;
}
