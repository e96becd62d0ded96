// Prototype (not part of the product): fp16 GEMM with LDS-DMA staging (global_load_lds), two LDS stages and
// ONE barrier per K-step, 8 waves per workgroup on a 256 x 256 tile — the structure csrc/gemm16.hip would move
// to if the fp16 path became the headline.  Standalone: builds its own inputs, checks a sample of outputs against
// a float64 evaluation of the same fp16 operands, prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/gemm16_dma.hip -o /tmp/gemm16_dma && /tmp/gemm16_dma
//
// C[M x N] (fp32) = A[M x K] (fp16, row-major, lda) x W (fp16, packed per (256-col tile, 64-k tile) as
// [kc 4][h 2][n 256][8 halfs] = the linear LDS image a B fragment read wants).
// A lands in LDS as rows of 8 16-byte chunks with the chunk index XOR-ed by (row & 7): the DMA keeps 128-byte
// contiguous global reads per 8 lanes and ds_read_b128 of one chunk column over 32 rows is conflict-free.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TM = 256, TN = 256, THREADS = 512;
// TERMS = 1: one fp16 product, K-step 64 (stage = A 32 KiB + B 32 KiB).
// TERMS = 2: W = W_hi + W_lo (the f16x2 scheme), both planes against the same A fragments, K-step 32
//            (stage = A 16 KiB + B_hi 16 KiB + B_lo 16 KiB); same MFMAs per barrier, half the LDS reads per MFMA.
#ifndef G16_TK1
#define G16_TK1 64     // K-step of the one-term kernel (experiment knob)
#endif
#ifndef G16_NST
#define G16_NST 0      // LDS stages; 0 = 3 if they fit, else 2
#endif
template <int TERMS> struct Cfg {
  static constexpr int TK = TERMS == 1 ? G16_TK1 : 32;
  static constexpr int KC = TK / 16;                    // 16-wide MFMA k-chunks per K-step
  static constexpr int CH = TK / 8;                     // 16-byte chunks per A row
  static constexpr int STAGE_A = TM * TK * 2;
  static constexpr int PLANE_B = TK * TN * 2;
  static constexpr int STAGE = STAGE_A + TERMS * PLANE_B;
  static constexpr int A_INSTR = STAGE_A / 16 / THREADS;   // DMA instructions per thread for A
  static constexpr int B_INSTR = PLANE_B / 16 / THREADS;   // ... per plane of B
  static constexpr int NSTAGE = (TERMS == 1 && G16_NST) ? G16_NST : (3 * STAGE <= 160 * 1024 ? 3 : 2);   // LDS stages (DMA runs NSTAGE-1 K-steps ahead)
  static_assert(NSTAGE * STAGE <= 160 * 1024, "stages exceed the LDS");
  static constexpr int DMA_PER_STEP = A_INSTR + TERMS * B_INSTR;
};

// chunk swizzle of an A row in LDS: 8 consecutive rows must land in 8 different 16-byte slots mod 256 B
template <int CH> __device__ __forceinline__ int swz(int row) { return CH == 8 ? (row & 7) : ((row ^ (row >> 2)) & 3); }
// the same for the 16x16x32 shape: a ds_read_b128 lane group there holds rows {0-3, 12-15} of one k-quarter and rows
// {4-11} of the next, which must land on 16 different 16-byte slots of a 256-byte bank row
template <int CH> __device__ __forceinline__ int swz16(int row) {
  if (CH == 4) return (0x78 >> (2 * ((row >> 2) & 3))) & 3;          // {0, 2, 3, 1}[(row >> 2) & 3]
  const int g = (row >> 1) & 7;
  return g ^ (((g >> 1) ^ (g >> 2)) & 1);                             // {0, 1, 3, 2, 5, 4, 6, 7}[(row >> 1) & 7]
}

#define GLDS(src, dst) \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src), \
                                   (void __attribute__((address_space(3))) *)(dst), 16, 0, 0)

template <int TERMS, bool PIPE, bool SHAPE16 = false>
__global__ __launch_bounds__(THREADS, 1) void gemm16_dma_kernel(const _Float16 *__restrict__ A, int64_t lda,
                                                                const _Float16 *__restrict__ Wp, int KT, int NT,
                                                                float *__restrict__ C, int64_t ldc) {
  using C_ = Cfg<TERMS>;
  constexpr int TK = C_::TK, KC = C_::KC, CH = C_::CH, STAGE_A = C_::STAGE_A, PLANE_B = C_::PLANE_B, STAGE = C_::STAGE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware grouped order as in the product kernels
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
  const int wgid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  const int per_group = 8 * NT;
  const int MT = nwg / NT;
  const int first_m = (wgid / per_group) * 8;
  const int gsz = (MT - first_m < 8) ? MT - first_m : 8;
  const int m_tile = first_m + (wgid % per_group) % gsz;
  const int n_tile = (wgid % per_group) / gsz;
  const int64_t m0 = (int64_t)m_tile * TM;

  // per-thread DMA sources: 4 A chunks and 4 B chunks per K-step
  const _Float16 *a_src[C_::A_INSTR];
#pragma unroll
  for (int i = 0; i < C_::A_INSTR; ++i) {
    const int L = i * THREADS + tid, row = L / CH, s = L % CH;
    a_src[i] = A + (m0 + row) * lda + ((s ^ (SHAPE16 ? swz16<CH>(row) : swz<CH>(row))) << 3);
  }
  const _Float16 *b_src = Wp + (int64_t)n_tile * KT * (TERMS * TK * TN) + tid * 8;

  auto issue = [&](int kt, int stage) {
    char *sa = smem + stage * STAGE, *sb = sa + STAGE_A;
#pragma unroll
    for (int i = 0; i < C_::A_INSTR; ++i) GLDS(a_src[i] + (int64_t)kt * TK, sa + (i * THREADS + wave * 64) * 16);
    const _Float16 *bt = b_src + (int64_t)kt * (TERMS * TK * TN);
#pragma unroll
    for (int i = 0; i < TERMS * C_::B_INSTR; ++i) GLDS(bt + i * THREADS * 8, sb + (i * THREADS + wave * 64) * 16);
  };

  if constexpr (SHAPE16) {
    // v_mfma_f32_16x16x32_f16: wave tile 128 x 64 = 8 x 4 blocks of 16 x 16, K = 32 per instruction (16 cycles).
    // Lane (r = lane & 15, q = lane >> 4) feeds A[row r][k = 8q..8q+7] and B[k = 8q..8q+7][col r]; D[4q + j][r].
    constexpr int KQ = TK / 32;   // 32-deep k-chunks per K-step
    const int r16 = lane & 15, q = lane >> 4;
    f32x4 acc16[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NSTAGE = C_::NSTAGE;
    // the DMA runs NSTAGE - 1 K-steps ahead; at a barrier the youngest NSTAGE - 2 steps may still be in flight
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i)
      if (i < KT) issue(i, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * C_::DMA_PER_STEP) : "memory");
    __syncthreads();
    int a_off[8][KQ];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int row = wm * 128 + mt * 16 + r16;
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq) a_off[mt][kq] = (row * CH + ((4 * kq + q) ^ swz16<CH>(row))) * 16;
    }
    // B image of a plane: [k / 8][256 columns][8 halfs]
    const int b_off = STAGE_A + (q * 256 + wn * 64 + r16) * 16;   // + plane * PLANE_B + kq * 4 * 256 * 16 + nt * 16 * 16
    int stage = 0;
    for (int kt = 0; kt < KT; ++kt) {
      if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1, (stage + NSTAGE - 1) % NSTAGE);
      const char *st = smem + stage * STAGE;
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq) {
        h8 a[8], b[TERMS][4];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a[mt] = *reinterpret_cast<const h8 *>(st + a_off[mt][kq]);
#pragma unroll
        for (int p = 0; p < TERMS; ++p)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            b[p][nt] = *reinterpret_cast<const h8 *>(st + b_off + p * PLANE_B + kq * 16384 + nt * 256);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int p = 0; p < TERMS; ++p)
#pragma unroll
          for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              acc16[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], b[p][nt], acc16[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      if (kt + NSTAGE - 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * C_::DMA_PER_STEP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail: fewer steps in flight than the count assumes
      __syncthreads();
      stage = (stage + 1) % NSTAGE;
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int col = n_tile * TN + wn * 64 + nt * 16 + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t row = m0 + wm * 128 + mt * 16 + 4 * q + j;
          C[row * ldc + col] = acc16[mt][nt][j];
        }
      }
    return;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr int NSTAGE = C_::NSTAGE;
  issue(0, 0);
  if (NSTAGE == 3 && KT > 1) issue(1, 1);
  if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_::DMA_PER_STEP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // fragment addresses inside a stage
  int a_off[4][KC];   // [mt][kc] byte offsets
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = wm * 128 + mt * 32 + r;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) a_off[mt][kc] = (row * CH + ((2 * kc + h) ^ swz<CH>(row))) * 16;
  }
  const int b_off = STAGE_A + (h * 256 + wn * 64 + r) * 16;   // + plane * PLANE_B + kc * 2 * 256 * 16 + nt * 32 * 16

  if constexpr (PIPE && NSTAGE == 3 && KC == 2) {
    // Pipelined form: the barrier sits in the MIDDLE of a K-step's MFMAs, and the fragment reads of the next
    // chunk are always issued 16 MFMAs before they are used, so neither LDS latency nor the barrier is exposed
    // even though all 8 waves of the (only) workgroup on the CU move in step.  Stages: kt being read,
    // kt+1 landed, kt+2 in flight; the DMA for kt+3 reuses kt's stage right after the barrier.
    auto load_frags = [&](const char *st, int kc, h8 (&fa)[4], h8 (&fb)[TERMS][2]) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const h8 *>(st + a_off[mt][kc]);
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) fb[p][nt] = *reinterpret_cast<const h8 *>(st + b_off + p * PLANE_B + kc * 8192 + nt * 512);
    };
    auto mfmas = [&](const h8 (&fa)[4], const h8 (&fb)[TERMS][2]) {
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt], fb[p][nt], acc[mt][nt], 0, 0, 0);
    };
    h8 a0[4], b0[TERMS][2], a1[4], b1[TERMS][2];
    if (KT > 2) issue(2, 2);
    load_frags(smem, 0, a0, b0);
    int stage = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const char *st = smem + stage * STAGE;
      const int nstage = stage == 2 ? 0 : stage + 1;
      load_frags(st, 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mfmas(a0, b0);
      __builtin_amdgcn_s_setprio(0);
      // stage kt+1 landed for every wave, and every wave has finished reading stage kt
      if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C_::DMA_PER_STEP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 3 < KT) issue(kt + 3, stage);
      if (kt + 1 < KT) load_frags(smem + nstage * STAGE, 0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mfmas(a1, b1);
      __builtin_amdgcn_s_setprio(0);
      stage = nstage;
    }
  } else {
    int stage = 0;
    for (int kt = 0; kt < KT; ++kt) {
      // the stage written now was last read NSTAGE-1... one K-step ago at the latest, and every wave is past that barrier
      if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1, (stage + NSTAGE - 1) % NSTAGE);
      const char *st = smem + stage * STAGE;
      // fragments of k-chunk kc+1 are read from LDS while the MFMAs of chunk kc run
      h8 a[2][4], b[2][TERMS][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[0][mt] = *reinterpret_cast<const h8 *>(st + a_off[mt][0]);
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[0][p][nt] = *reinterpret_cast<const h8 *>(st + b_off + p * PLANE_B + nt * 512);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int c = kc & 1, n = c ^ 1;
        if (kc + 1 < KC) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a[n][mt] = *reinterpret_cast<const h8 *>(st + a_off[mt][kc + 1]);
#pragma unroll
          for (int p = 0; p < TERMS; ++p)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              b[n][p][nt] = *reinterpret_cast<const h8 *>(st + b_off + p * PLANE_B + (kc + 1) * 8192 + nt * 512);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads ahead of the MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int p = 0; p < TERMS; ++p)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c][mt], b[c][p][nt], acc[mt][nt], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      // next K-step's data must have landed; with three stages the DMA issued above may stay in flight
      if (NSTAGE == 3 && kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_::DMA_PER_STEP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stage = (stage + 1) % NSTAGE;
    }

  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n_tile * TN + wn * 64 + nt * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 128 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        C[row * ldc + col] = acc[mt][nt][e];
      }
    }
}

template <int TERMS, bool PIPE, bool SHAPE16 = false>
static int run(int M, int N, int K, const std::vector<_Float16> &hA, const std::vector<float> &hWf) {
  using C_ = Cfg<TERMS>;
  constexpr int TK = C_::TK, KC = C_::KC;
  const int MT = M / TM, NT = N / TN, KT = K / TK;
  // planes: TERMS = 1 -> fp16(W); TERMS = 2 -> hi = fp16(W), lo = fp16(W - hi)
  std::vector<_Float16> hWp((size_t)TERMS * K * N);
  for (int nt = 0; nt < NT; ++nt)
    for (int kt = 0; kt < KT; ++kt)
      for (int p = 0; p < TERMS; ++p)
        for (int kc = 0; kc < KC; ++kc)
          for (int h = 0; h < 2; ++h)
            for (int n = 0; n < 256; ++n)
              for (int j = 0; j < 8; ++j) {
                const int k = kt * TK + kc * 16 + h * 8 + j;
                const float w = hWf[(size_t)k * N + nt * 256 + n];
                const _Float16 hi = (_Float16)w;
                const _Float16 v = p == 0 ? hi : (_Float16)(w - (float)hi);
                hWp[((((size_t)nt * KT + kt) * TERMS + p) * KC + kc) * 4096 + (size_t)h * 2048 + n * 8 + j] = v;
              }
  _Float16 *dA, *dW;
  float *dC;
  hipMalloc(&dA, hA.size() * 2);
  hipMalloc(&dW, hWp.size() * 2);
  hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dW, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void *>(gemm16_dma_kernel<TERMS, PIPE, SHAPE16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      C_::NSTAGE * C_::STAGE);
  auto launch = [&]() {
    hipLaunchKernelGGL((gemm16_dma_kernel<TERMS, PIPE, SHAPE16>), dim3(MT * NT), dim3(THREADS), C_::NSTAGE * C_::STAGE, 0, dA, (int64_t)K, dW, KT, NT, dC,
                       (int64_t)N);
  };
  launch();
  hipDeviceSynchronize();
  printf("terms=%d pipe=%d shape16=%d launch: %s\n", TERMS, (int)PIPE, (int)SHAPE16, hipGetErrorString(hipGetLastError()));
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0.0;
  unsigned s = 999;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0xffff) / 65536.0f; };
  for (int t = 0; t < 64; ++t) {
    const int i = (int)(rnd() * M) % M, j = (int)(rnd() * N) % N;
    double ref = 0.0;
    for (int k = 0; k < K; ++k) {
      const float w = hWf[(size_t)k * N + j];
      const double wv = TERMS == 2 ? (double)(float)(_Float16)w + (double)(float)(_Float16)(w - (float)(_Float16)w) : (double)(float)(_Float16)w;
      ref += (double)(float)hA[(size_t)i * K + k] * wv;
    }
    worst = fmax(worst, fabs(ref - hC[(size_t)i * N + j]) / (fabs(ref) + 1e-3));
  }
  printf("terms=%d max rel err over 64 samples (vs float64 on the same operands): %.3e\n", TERMS, worst);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("terms=%d pipe=%d shape16=%d M=%d N=%d K=%d: %.3f ms per launch, %.1f TFLOP/s executed fp16 (%.1f algorithmic)\n", TERMS, (int)PIPE, (int)SHAPE16, M, N, K, ms,
         TERMS * 2.0 * M * N * K / (ms * 1e-3) / 1e12, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  hipFree(dA); hipFree(dW); hipFree(dC);
  return worst < 1e-2 ? 0 : 1;
}

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, N = 3072, K = 30016;
  std::vector<_Float16> hA((size_t)M * K);
  std::vector<float> hWf((size_t)K * N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0xffff) / 65536.0f; };
  for (auto &v : hA) v = (_Float16)(float)(int)(rnd() * 4.0f);                 // small counts, like the encoder input
  for (auto &v : hWf) v = (rnd() - 0.5f) * 0.03f;
  int rc = 0;
#ifndef G16_ONLY16
  rc |= run<1, false>(M, N, K, hA, hWf);
  rc |= run<2, false>(M, N, K, hA, hWf);
  rc |= run<2, true>(M, N, K, hA, hWf);
#endif
  rc |= run<1, false, true>(M, N, K, hA, hWf);
  rc |= run<2, false, true>(M, N, K, hA, hWf);
  return rc;
}
