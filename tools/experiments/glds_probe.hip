#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// 256 threads copy 16 KB global -> LDS with global_load_lds (16 B per lane), then LDS -> global
__global__ void k(const float4* __restrict__ src, float4* __restrict__ dst) {
  __shared__ __attribute__((aligned(16))) float4 lds[1024];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = 0; i < 4; ++i) {
    // LDS destination: wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 256 + tid),
                                     (void __attribute__((address_space(3)))*)(lds + i * 256 + wave * 64), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) dst[i * 256 + tid] = lds[i * 256 + tid];
}
int main() {
  std::vector<float> h(4096), o(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  float4 *s, *d; hipMalloc(&s, 16384); hipMalloc(&d, 16384);
  hipMemcpy(s, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d);
  hipMemcpy(o.data(), d, 16384, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 4096; ++i) bad += o[i] != h[i];
  printf("bad=%d err=%s\n", bad, hipGetErrorString(hipGetLastError()));
}
