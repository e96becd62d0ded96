// What else shares the issue time of a SIMD with its MFMAs on gfx950?  (round 4, last session: the collect's cycles per tile
// fit 18 x MFMA + 4 x (every other instruction), not 18 x MFMA + 4 x VALU — does a scalar / LDS / branch instruction of one
// wave really cost the SIMD issue time?)
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_salu_overlap tools/experiments/mfma_salu_overlap.hip
// WAVES waves per SIMD, each a loop of 8 MFMAs (v_mfma_f32_32x32x16_f16, two accumulators alternating) with, behind each
// MFMA, KV vector + KS scalar + KL LDS-read + KB (taken, wave-uniform) branch instructions.  ns per MFMA and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KV, int KS, int KL, int KB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(unsigned long long *out, int iters, unsigned seed) {
  __shared__ f32x4 lds[64 * 12];
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (seed + i)); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  lds[threadIdx.x % (64 * 12)] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  unsigned x0 = threadIdx.x * 77u + seed, x1 = seed ^ 0x55u, m0 = 0xffffffffu, t0 = 0;
  unsigned s0 = seed, s1 = seed * 3u + 1u, kmask = 0xffffe000u;
  const f32x4 *lp = lds + (threadIdx.x & 63);
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 lv[KL ? KL : 1];
  for (int j = 0; j < (KL ? KL : 1); ++j) lv[j] = acc4;
  const unsigned long long start = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define STEP(C)                                                                                                        \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));                                   \
    _Pragma("unroll") for (int j = 0; j < KV; ++j) asm volatile("v_and_or_b32 %0, %1, %3, %2" : "+v"(t0) : "v"(x0), "v"(x1), "s"(kmask)); \
    _Pragma("unroll") for (int j = 0; j < KS; ++j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");            \
    _Pragma("unroll") for (int j = 0; j < KL; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[j]) : "v"((unsigned)(size_t)(lp + 64 * j))); \
    _Pragma("unroll") for (int j = 0; j < KB; ++j) asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc0 1f\n 1:" :: "s"(s1) : "scc");
    STEP(c0) STEP(c1) STEP(c0) STEP(c1) STEP(c0) STEP(c1) STEP(c0) STEP(c1)
    // the asynchronous LDS results stay live registers until they have landed (a dead "=v" output of an inline-asm ds_read
    // is a register the compiler re-uses while the read is in flight: tools/check_async_lds.py)
    if (KL) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); _Pragma("unroll") for (int j = 0; j < (KL ? KL : 1); ++j) asm volatile("" : "+v"(lv[j])); }
  }
  const unsigned long long stop = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = stop - start; for (int j = 0; j < (KL ? KL : 1); ++j) acc4 += lv[j]; out[1] = (unsigned long long)(s + m0 + t0 + s0 + acc4[0]); }
}

template <int KV, int KS, int KL, int KB, int WAVES>
static void run(unsigned long long *d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KV, KS, KL, KB, WAVES><<<1024, 64 * WAVES>>>(d, 100, 1);
  hipEventRecord(e0);
  probe<KV, KS, KL, KB, WAVES><<<1024, 64 * WAVES>>>(d, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // 1024 workgroups of WAVES waves on 256 CUs = WAVES waves per SIMD; a SIMD runs iters * 8 * WAVES MFMAs
  printf("%d wave(s)/SIMD, per MFMA: %2d VALU %2d SALU %2d ds_read_b128 %2d cmp+branch: %7.3f ms  %6.1f ns per MFMA and SIMD\n", WAVES, KV, KS, KL,
         KB, ms, ms * 1e6 / (iters * 8.0 * WAVES));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned long long *d;
  hipMalloc(&d, 64);
  run<0, 0, 0, 0, 3>(d);
  run<4, 0, 0, 0, 3>(d);
  run<8, 0, 0, 0, 3>(d);
  run<4, 3, 0, 0, 3>(d);
  run<4, 6, 0, 0, 3>(d);
  run<4, 12, 0, 0, 3>(d);
  run<4, 0, 1, 0, 3>(d);
  run<4, 0, 2, 0, 3>(d);
  run<4, 0, 0, 1, 3>(d);
  run<4, 0, 0, 2, 3>(d);
  run<4, 3, 1, 1, 3>(d);
  run<0, 12, 0, 0, 3>(d);
  run<4, 6, 0, 0, 2>(d);
  run<4, 6, 0, 0, 4>(d);
  return 0;
}
