// Bare fp32 MFMA loops on random operands in registers: does the chip hold a different clock (and rate) with the
// 16x16x4 shape than with the 32x32x2 shape the product GEMM uses?  (For fp16 the 16x16x32 shape ran 13-15 % faster
// than 32x32x16 at equal cycles, see gemm16_dma.hip.)
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_shape_probe.hip -o bin/mfma_f32_shape_probe && bin/mfma_f32_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ in, float *__restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xfffff]; b[i] = in[(tid * 16 + 8 + i) & 0xfffff]; }
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 7], b[u], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[j][e];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 7], b[(u + (j >> 2)) & 7], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += acc[j][e];
  }
  out[tid] = sum;
}

int main() {
  const int n = 1 << 20;
  std::vector<float> h(n);
  for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *din, *dout;
  hipMalloc(&din, n * 4);
  hipMalloc(&dout, 1024 * 256 * 8 * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd;   // 4 waves per block: one (or two) per SIMD
    for (int shape : {32, 16, 32, 16}) {
      const int iters = 20000;
      // per iteration and wave: shape 32: 32 MFMAs x 4096 flop; shape 16: 64 MFMAs x 2048 flop
      const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
      for (int rep = 0; rep < 40; ++rep) {   // ~2 s of back-to-back launches before the timed ones
        if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
      }
      hipEventRecord(e0);
      for (int rep = 0; rep < 10; ++rep) {
        if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("waves/SIMD %d shape %dx%d: %.1f TFLOP/s (%.2f ms per launch)\n", waves_per_simd, shape, shape,
             flop * 10 / (ms * 1e-3) / 1e12, ms / 10);
    }
  }
  return 0;
}
