// Does a wave's vector issue overlap its own MFMAs on gfx950?  (round 4: the model behind DESIGN.md §3 K5's "issue time")
//   hipcc -O3 --offload-arch=gfx950 -o tools/experiments/bin/mfma_valu_overlap tools/experiments/mfma_valu_overlap.hip
// One wave per SIMD (1024 one-wave workgroups), a loop of 8 independent-pair MFMAs (v_mfma_f32_32x32x16_f16, two
// accumulators alternating) with K vector instructions (v_and_or_b32 / v_min_u32 on registers nobody else touches)
// behind each MFMA; cycles per MFMA from s_memtime.  ACC = "v" (accumulators in VGPRs) or "a" (AGPRs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VAL2 "v_and_or_b32 %[t0], %[x0], %[k], %[x1]\n v_min_u32 %[m0], %[t0], %[m0]\n"
#define VAL1 "v_and_or_b32 %[t1], %[x1], %[k], %[x0]\n"

template <int K, bool AG, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(unsigned long long *out, int iters, unsigned seed) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (seed + i)); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  unsigned x0 = threadIdx.x * 77u + seed, x1 = seed ^ 0x55u, m0 = 0xffffffffu, t0 = 0, t1 = 0, k = 0xffffe000u;
  const unsigned long long start = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define STEP(C)                                                                                              \
    if (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(C) : "v"(a), "v"(b));               \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));                  \
    if (K >= 2) asm volatile(VAL2 : [t0] "+v"(t0), [m0] "+v"(m0) : [x0] "v"(x0), [x1] "v"(x1), [k] "s"(k));  \
    if (K >= 4) asm volatile(VAL2 : [t0] "+v"(t0), [m0] "+v"(m0) : [x0] "v"(x0), [x1] "v"(x1), [k] "s"(k));  \
    if (K >= 6) asm volatile(VAL2 : [t0] "+v"(t0), [m0] "+v"(m0) : [x0] "v"(x0), [x1] "v"(x1), [k] "s"(k));  \
    if (K >= 7) asm volatile(VAL1 : [t1] "+v"(t1) : [x0] "v"(x0), [x1] "v"(x1), [k] "s"(k));                  \
    if (K >= 14) { asm volatile(VAL2 VAL2 VAL2 VAL1 : [t0] "+v"(t0), [m0] "+v"(m0), [t1] "+v"(t1) : [x0] "v"(x0), [x1] "v"(x1), [k] "s"(k)); }
    STEP(c0) STEP(c1) STEP(c0) STEP(c1) STEP(c0) STEP(c1) STEP(c0) STEP(c1)
  }
  const unsigned long long stop = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = stop - start; out[1] = (unsigned long long)(s + m0 + t0 + t1); }
}

template <int K, bool AG, int WAVES>
static void run(const char *name, unsigned long long *d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<K, AG, WAVES><<<256 * 4 / 1, 64 * WAVES>>>(d, 100, 1);
  hipEventRecord(e0);
  probe<K, AG, WAVES><<<256 * 4 / 1, 64 * WAVES>>>(d, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // 1024 workgroups of WAVES waves on 256 CUs: 4 workgroups per CU = WAVES waves per SIMD
  printf("%-34s %7.3f ms  %6.1f ns per MFMA and wave   (%d MFMA + %d VALU per step, %d wave(s) per SIMD)\n", name, ms,
         ms * 1e6 / (iters * 8.0), 1, K, WAVES);
}

int main() {
  unsigned long long *d;
  hipMalloc(&d, 64);
  run<0, false, 1>("VGPR acc, MFMA only", d);
  run<2, false, 1>("VGPR acc, +2 VALU", d);
  run<4, false, 1>("VGPR acc, +4 VALU", d);
  run<7, false, 1>("VGPR acc, +7 VALU", d);
  run<14, false, 1>("VGPR acc, +14 VALU", d);
  run<0, true, 1>("AGPR acc, MFMA only", d);
  run<4, true, 1>("AGPR acc, +4 VALU", d);
  run<7, true, 1>("AGPR acc, +7 VALU", d);
  run<14, true, 1>("AGPR acc, +14 VALU", d);
  run<0, false, 2>("VGPR acc, MFMA only, 2 waves", d);
  run<7, false, 2>("VGPR acc, +7 VALU, 2 waves", d);
  run<14, false, 2>("VGPR acc, +14 VALU, 2 waves", d);
  run<7, true, 2>("AGPR acc, +7 VALU, 2 waves", d);
  return 0;
}
