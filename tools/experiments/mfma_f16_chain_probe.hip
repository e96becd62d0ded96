// Probe (round 3): bare v_mfma_f32_32x32x16_f16 loops shaped like the kNN collect's tile — chains of NDEP dependent
// MFMAs into one accumulator, NCH chains interleaved, the accumulator re-zeroed and its result consumed by a vector
// compare after every chain (as the sign tests do) — at 1, 2 and 3 waves per SIMD.  Question: do 5-long dependent
// chains, two at a time, keep the matrix pipes of a SIMD busy with three waves resident?
//   hipcc -O3 --offload-arch=gfx950 mfma_f16_chain_probe.hip -o bin/mfma_f16_chain_probe && bin/mfma_f16_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

template <int NCH, int NDEP, int CONSUME>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ in, float *__restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  h8 a[NDEP], b[NCH][NDEP];
#pragma unroll
  for (int t = 0; t < NDEP; ++t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[t][i] = (_Float16)in[(tid * 64 + t * 8 + i) & 0xfffff];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) b[c][t][i] = (_Float16)in[(tid * 64 + 40 + c * 8 + t + i) & 0xfffff];
  }
  unsigned hits = 0;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
#pragma unroll
    for (int t = 0; t < NDEP; ++t)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[c][t], acc[c], 0, 0, 0);
    if (CONSUME) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {   // an OR tree over the signs + one ballot, like the collect's whole-tile rejection
        unsigned o = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) o |= __float_as_uint(acc[c][e]);
        if (__any((int)o < 0 && acc[c][3] == 12345.f)) hits += 1;
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) a[0][0] += (_Float16)(acc[c][0] * 1e-30f);   // keeps the chain alive, no wave-level test
    }
  }
  out[tid] = (float)hits + (float)a[0][0];
}

template <int NCH, int NDEP, int CONSUME>
void run(const float *din, float *dout, hipEvent_t e0, hipEvent_t e1) {
  for (int wps = 1; wps <= 3; ++wps) {
    const int blocks = 256 * wps, iters = 20000 / NCH;
    const double flop = (double)blocks * 4 * iters * NCH * NDEP * 32768.0;
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((probe<NCH, NDEP, CONSUME>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((probe<NCH, NDEP, CONSUME>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("chains %d x %d dependent, consume %d, waves/SIMD %d: %7.1f TFLOP/s (%.0f %% of 2500)\n", NCH, NDEP, CONSUME, wps,
           flop * 10 / (ms * 1e-3) / 1e12, flop * 10 / (ms * 1e-3) / 1e12 / 25.0);
  }
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
  run<2, 5, 1>(din, dout, e0, e1);   // the collect's shape: two interleaved chains of 5, sign test after each pair
  run<4, 5, 1>(din, dout, e0, e1);   // four chains in flight (two candidate blocks at once)
  run<2, 4, 1>(din, dout, e0, e1);   // 4 MFMAs per tile (norms through the C operand)
  run<2, 5, 0>(din, dout, e0, e1);   // no wave-level test on the result
  run<8, 5, 1>(din, dout, e0, e1);
  return 0;
}
