// K2' — encoder layers 1 and 2 on the fp16 matrix cores (v_mfma_f32_16x16x32_f16, fp32 accumulate).
//
// Two uses of one kernel:
//   CC_DTYPE_F16    operands rounded to fp16 once (BASELINE config 5 "fp16 MFMA"): one term, one weight plane.
//   CC_DTYPE_F16X2  fp32-equivalent result from fp16 operands: every fp32 weight is split as
//                   W·s = W_hi + W_lo (two fp16 values, s a power of two so that max|W·s| is in
//                   [2^13, 2^14)), 22 significant bits; an activation that is not exactly an fp16
//                   value (the sigmoid outputs of layer 1) is split the same way, A = A_hi + A_lo.
//                   A·W ≈ A_hi·W_hi + A_hi·W_lo + A_lo·W_hi  (A_lo·W_lo ~ 2^-24 relative, dropped).
//                   Count data (integers <= 2048) is exact in fp16, so for the raw-input segments
//                   two terms suffice; the gather kernel verifies exactness and the host falls
//                   back to the fp32 kernel for a pass that is not exact.
// The terms are expressed as K-segments of one accumulation:  {A, nterms} with nterms = 2 meaning
// "multiply this A tile with W_hi and with W_lo", nterms = 1 "with W_hi only".  A 3-term input is
// the two segments {A_hi, 2} and {A_lo, 1} over the same weight rows.
//
// Structure.  256 x 256 tile per 512-thread workgroup (one per CU), 8 waves as 2 x 4, 128 x 64 per wave = 8 x 4 blocks of
// 16 x 16 (128 accumulator registers).  Operands reach LDS by LDS-DMA (global_load_lds, 16 bytes per lane, no staging
// registers) with counted vmcnt and bare s_barrier; A rows sit in LDS as 16-byte chunks whose position is XOR-swizzled by
// the row so that the fragment reads of the 16x16x32 shape are bank-conflict free, in planes of 32 k ([k / 32][256
// rows][4 chunks]); a weight plane is packed as the linear LDS image [k / 8][256 columns][8 halfs].  BOTH operands are
// stored in HBM as their LDS images — the weights since round 2, the activations since round 4 (A16T, common.h: the
// gather, the dense-host converter and layer 1's epilogue write 256-row x 32-column tiles with the swizzle applied) —
// so every LDS-DMA instruction copies one contiguous KiB (round 3 swizzled the per-lane source address of row-major
// rows instead: eight 128-byte pieces 60 KB apart per instruction, +8-9 % on the one-plane layers when they went).
// With two planes the A fragments of a K-step are used against W_hi and W_lo.
// THREE K-loop forms (all issue the same MFMAs in the same order per accumulator: results do not depend on which runs):
//   gemm16_segment_pp    whole rounds of tiles, one plane: ping-pong of the two wave groups, K-step 64, 2 x 64 KiB
//   gemm16_segment_pp2   whole rounds, two planes, two-term segments: the same over K-step 32, 3 x 48 KiB {A, W_hi, W_lo}
//   gemm16_segment       everything else — the half-height (128-row) tiles of a last round that is at most half full
//                        and of every call of a few thousand rows (gemm16_kernel<.., HALVES>: two workgroups per tile,
//                        8 waves as 2 x 4 of 64 x 64), and the one-term segments (A_lo x W_hi: 4 % of layer 2's MFMAs):
//                        one barrier per K-step, the DMA NSTAGE - 1 steps ahead, fragment waits placed by hipcc
// (Round 3 also carried a hand-waited one-plane form for the half-height tiles, +1.7 % on those tiles only, and a set
// of result-destroying ablation switches inside this file; round 4 moved all of that to tools/experiments/
// g16_ablations.patch — this translation unit has no build-time switches.)
// Measured on layer 1 at C3's width: one plane 1408-1437 TFLOP/s (round 1: 926 with a register-staged 32x32x16 kernel;
// round 2: 1223-1251; round 3 before the ping-pong 1281-1346), two planes 1603-1650 executed; DESIGN.md §3 K2'.
#include "common.h"

namespace cc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int H_TM = GEMM16_TILE, H_TN = GEMM16_TILE, H_THREADS = 512;
template <int PLANES> struct Cfg16 {
  static constexpr int TK = PLANES == 1 ? 64 : 32;      // K-step
  static constexpr int KQ = TK / 32;                    // MFMA k-chunks per K-step
  static constexpr int CH = TK / 8;                     // 16-byte chunks per A row
  static constexpr int STAGE_A = H_TM * TK * 2;
  static constexpr int PLANE_B = TK * H_TN * 2;
  static constexpr int STAGE = STAGE_A + PLANES * PLANE_B;
  static constexpr int A_INSTR = STAGE_A / 16 / H_THREADS;   // DMA instructions per thread and K-step for A
  static constexpr int B_INSTR = PLANE_B / 16 / H_THREADS;   // ... per weight plane
  static constexpr int NSTAGE = PLANES == 1 ? 2 : 3;
  static constexpr int LDS = NSTAGE * STAGE;
  static_assert(LDS <= 160 * 1024, "stages exceed the LDS of a CU");
};

struct Seg16 {
  const _Float16 *A;   // fp16, A16T layout (common.h): 256-row x 32-column tiles of 16 KiB
  int64_t lda;         // columns of the A16T matrix (a multiple of 32)
  int KT;              // K-steps of this segment
  int wkt0;            // first K-step of the packed weight this segment multiplies
  int nterms;          // 1: W_hi only, 2: W_hi and W_lo
};
struct Segs16 { Seg16 s[4]; int n; };

// What a wave does in the K loop: its 128 x 64 share of a 256-row tile; nothing but DMA requests and barriers (all its
// rows lie beyond M); or 64 x 64 of a half-height (128-row) tile, where all 8 waves stay busy.
enum { W_FULL = 0, W_IDLE = 1, W_HALF = 2 };

__device__ __forceinline__ float sigmoid16_(float z) { return 1.0f / (1.0f + expf(-z)); }

// Chunk swizzle of an A row in LDS.  A ds_read_b128 is served in groups of 16 lanes; with the 16x16x32 fragment
// (lane = 16 q + r reads chunk q of row r) a group holds rows {0-3, 12-15} of one k-quarter and rows {4-11} of the
// next, which must fall on 16 different 16-byte slots of the 256-byte bank row.
template <int CH> __device__ __forceinline__ int swz16(int row) {
  if (CH == 4) return (0x78 >> (2 * ((row >> 2) & 3))) & 3;          // {0, 2, 3, 1}[(row >> 2) & 3]
  const int g = (row >> 1) & 7;
  return g ^ (((g >> 1) ^ (g >> 2)) & 1);                             // {0, 1, 3, 2, 5, 4, 6, 7}[(row >> 1) & 7]
}

#define CC_GLDS16(src, dst)                                                                 \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src),   \
                                   (void __attribute__((address_space(3))) *)(dst), 16, 0, 0)

// The K-step barrier.  With more than one K-step of LDS-DMA in flight the barrier must not drain them: __syncthreads()
// carries a fence that waits vmcnt(0) for pending LDS-DMA writes (the guide's "Pipelining across barriers"), so the
// counted vmcnt of the 3-stage ring only works with the bare instruction.  Every wave's own LDS reads of the K-step
// have been consumed by its MFMAs (or explicitly waited for, staggered waves) when it gets here.
__device__ __forceinline__ void step_sync16() {
  asm volatile("" ::: "memory");   // no LDS access of the compiler's may cross the barrier in either direction
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// DMA sources of a K-segment: wave-uniform base pointers (SGPR pairs, advanced by scalar adds) plus per-lane byte
// offsets that stay constant over the whole segment, so the K loop spends no vector instruction on addresses.
template <int PLANES> struct Src16 {
  const char *a_base[Cfg16<PLANES>::A_INSTR];   // piece i (8 KiB) of the A16T tiles of the current K-step
  unsigned a_voff[Cfg16<PLANES>::A_INSTR];      // tid * 16: the tiles are copied as they are
  const char *b_base;                           // packed weight tile of K-step 0
  unsigned b_voff;
};

// One K-segment: the DMA runs NSTAGE - 1 K-steps ahead, one barrier per K-step, everything drained at its end.
// TWO: the segment multiplies W_lo as well as W_hi.  (A function template with the accumulators passed by reference:
// as a lambda capturing them hipcc kept part of the state in scratch.)
template <int PLANES, bool TWO, bool HALF>
__device__ __forceinline__ void gemm16_segment(f32x4 (&acc)[8][4], Src16<PLANES> src, int KT, char *smem,
                                               const int (&a_off)[8][Cfg16<PLANES>::KQ], int b_off, int wave, int mode) {
  using C_ = Cfg16<PLANES>;
  constexpr int TK = C_::TK, KQ = C_::KQ, STAGE_A = C_::STAGE_A, PLANE_B = C_::PLANE_B, STAGE = C_::STAGE;
  constexpr int NSTAGE = C_::NSTAGE;
  constexpr int NPL = TWO ? 2 : 1;
  constexpr int DPS = C_::A_INSTR + NPL * C_::B_INSTR;     // DMA instructions per thread and K-step
  auto issue = [&](int stage) {   // the next K-step not yet requested; `wave` is wave-uniform, so every LDS destination is scalar
    char *sa = smem + stage * STAGE, *sb = sa + STAGE_A;
#pragma unroll
    for (int i = 0; i < C_::A_INSTR; ++i) {
      CC_GLDS16(src.a_base[i] + src.a_voff[i], sa + (i * H_THREADS + wave * 64) * 16);
      src.a_base[i] += STAGE_A;   // the A16T tiles of the next K-step follow
    }
#pragma unroll
    for (int i = 0; i < NPL * C_::B_INSTR; ++i)
      CC_GLDS16(src.b_base + i * (H_THREADS * 16) + src.b_voff, sb + (i * H_THREADS + wave * 64) * 16);
    src.b_base += PLANES * TK * H_TN * 2;
  };
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i)
    if (i < KT) issue(i);
  if (KT >= NSTAGE - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DPS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  step_sync16();
  int stage = 0;
  // A K-step is consumed in half-chunks (k-chunk kq, row half hf: 4 of the wave's 8 row blocks = 16 or 32 MFMAs).  The
  // fragments of half-chunk i+1 are read from LDS BEFORE the MFMAs of half-chunk i are issued; the B fragments of a
  // k-chunk serve both halves.
  constexpr int NH = 2 * KQ;
  auto mfma_half = [&](int hf, const h8 (&af)[4], const h8 (&bf)[NPL][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[4 * hf + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[p][nt], acc[4 * hf + mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto load_a = [&](const char *st, int kq, int hf, h8 (&dst)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) dst[mt] = *reinterpret_cast<const h8 *>(st + a_off[4 * hf + mt][kq]);
  };
  auto load_b = [&](const char *st, int kq, h8 (&dst)[NPL][4]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        dst[p][nt] = *reinterpret_cast<const h8 *>(st + b_off + p * PLANE_B + kq * 16384 + nt * 256);
  };
  auto step_barrier = [&](int kt) {
    // the next K-step's data must have landed; with three stages one more step may stay in flight
    if (kt + NSTAGE - 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail: fewer steps in flight than the count assumes
    step_sync16();
  };
  for (int kt = 0; kt < KT; ++kt) {
    const char *st = smem + stage * STAGE;
    h8 a[2][4], b[KQ > 1 ? 2 : 1][NPL][4];
    // the first fragment reads go out right behind the barrier, the DMA requests of the K-step NSTAGE - 1 ahead
    // follow (the stage they overwrite was read one K-step ago at the latest, and every wave is past that barrier)
    if (mode != W_IDLE) {
      load_b(st, 0, b[0]);
      load_a(st, 0, 0, a[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NSTAGE - 1 < KT) issue((stage + NSTAGE - 1) % NSTAGE);
    if (HALF && mode == W_HALF) {   // 64 rows of a half-height tile: row blocks 0-3 only (a_off already points at them)
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq) {
        if (kq + 1 < KQ) {
          load_b(st, kq + 1, b[(kq + 1) & 1]);
          load_a(st, kq + 1, 0, a[(kq + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(0, a[kq & 1], b[KQ > 1 ? (kq & 1) : 0]);
      }
    } else if (!HALF && mode == W_FULL) {   // (W_IDLE: no rows of the tile — the wave only feeds the DMA and keeps the barriers)
#pragma unroll
      for (int hc = 0; hc < NH; ++hc) {
        const int kq = hc >> 1, hf = hc & 1;
        if (hc + 1 < NH) {
          const int nkq = (hc + 1) >> 1, nhf = (hc + 1) & 1;
          if (nhf == 0) load_b(st, nkq, b[nkq & 1]);
          load_a(st, nkq, nhf, a[(hc + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
        mfma_half(hf, a[hc & 1], b[KQ > 1 ? (kq & 1) : 0]);
      }
    }
    step_barrier(kt);
    stage = (stage + 1) % NSTAGE;
  }
}

// Inline-asm LDS reads of the ping-pong loops (the guide's §5.7 form iii: "=v" loads, a wait-only statement,
// sched_barrier(0) in front of the consumers).  hipcc neither counts nor waits for them, and it believes the output is
// written AT the statement while the hardware writes it when the LDS answers: every output MUST stay live until a
// consumer behind the wait (a dead output's registers are reused at once — e.g. as the address pair of the next
// global_load_lds — and the late data lands on top: round 3's `-DCC_PP_NOMFMA` ablation faulted the GPU exactly so,
// tools/experiments/README.md).  tools/check_async_lds.py checks the built assembly for it (tests/test_cabi_symbols.py).
#define CC_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define CC_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

// ---- the one-plane K-segment as a ping-pong of the two wave groups (the guide's "8-phase" structure) ----
// A K-step (64) is four phases of 16 MFMAs: (rows 0-63, k 0-31), (rows 64-127, k 0-31), (rows 0-63, k 32-63),
// (rows 64-127, k 32-63) of the wave's 128 x 64 — the same MFMAs in the same order per accumulator as the other
// segment forms.  Every phase is  [fragment reads + one 16 KiB DMA piece] s_barrier [lgkmcnt(0), 16 MFMAs] s_barrier,
// and waves 4-7 run ONE barrier behind waves 0-3: while one group multiplies the other reads and requests, so each
// SIMD (one wave of either group) always has exactly one wave in its MFMAs.
// LDS stage (64 KiB, two of them): A as two k-half planes [k / 32][256 rows][4 chunks] (chunk swizzle swz16<4>) + the
// weight image [k / 8][256][8] as packed.  The DMA pieces are the k-halves of A and of W: a piece of K-step t + 2 is
// requested two phases after the last read of the region it overwrites, and is waited for (counted: three younger
// pieces stay in flight) one phase before its first read:
//   P1(t): request W k0 of t+1 | P2(t): A k1 of t+1, wait k1 of t | P3(t): W k1 of t+1 | P4(t): A k0 of t+2, wait k0 of t+1
struct SrcPP {
  const char *a_cur[2];   // the two 8 KiB halves of the first A16T tile (k-half 0) of K-step t (wave-uniform)
  unsigned a_voff[2];     // tid * 16
  const char *b_cur;      // packed weight tile of K-step t
  unsigned b_voff;
};

__device__ __forceinline__ void gemm16_segment_pp(f32x4 (&acc)[8][4], SrcPP src, int KT, char *smem, int a_lane, int b_lane,
                                                  int wave, int wm) {
  constexpr int STAGE = 65536, A_PLANE = 16384, B_OFF = 32768, B_HALF = 16384, A_STEP = 32768, B_STEP = 32768;
  // one DMA piece = 16 KiB = two instructions per thread
  auto issue_a = [&](int step, int kq, int stage) {   // A k-half kq of K-step t + step
    char *dst = smem + stage * STAGE + kq * A_PLANE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      CC_GLDS16(src.a_cur[i] + (step * A_STEP + kq * A_PLANE) + src.a_voff[i], dst + (i * H_THREADS + wave * 64) * 16);
  };
  auto issue_b = [&](int step, int kq, int stage) {   // weight k-half kq of K-step t + step
    char *dst = smem + stage * STAGE + B_OFF + kq * B_HALF;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      CC_GLDS16(src.b_cur + (step * B_STEP + kq * B_HALF + i * (H_THREADS * 16)) + src.b_voff,
                dst + (i * H_THREADS + wave * 64) * 16);
  };
  auto mfma16 = [&](int hf, const h8 (&af)[4], const h8 (&bf)[4]) {
    CC_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[4 * hf + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[4 * hf + mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // prologue: K-step 0 whole, A k0 of K-step 1
  if (KT > 0) { issue_a(0, 0, 0); issue_b(0, 0, 0); issue_a(0, 1, 0); issue_b(0, 1, 0); }
  if (KT > 1) {
    issue_a(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // k0 of K-step 0 has landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  bar();
  if (wm) bar();   // waves 4-7 run one barrier behind
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  h8 a0[4], a1[4], b[4];
#define CC_PP_A(dst, base, kq, hf)                                                                           \
  CC_DSR128(dst[0], base, (kq) * 16384 + (4 * (hf) + 0) * 1024); CC_DSR128(dst[1], base, (kq) * 16384 + (4 * (hf) + 1) * 1024); \
  CC_DSR128(dst[2], base, (kq) * 16384 + (4 * (hf) + 2) * 1024); CC_DSR128(dst[3], base, (kq) * 16384 + (4 * (hf) + 3) * 1024)
#define CC_PP_B(dst, base, kq)                                                                               \
  CC_DSR128(dst[0], base, (kq) * 16384 + 0); CC_DSR128(dst[1], base, (kq) * 16384 + 256);                    \
  CC_DSR128(dst[2], base, (kq) * 16384 + 512); CC_DSR128(dst[3], base, (kq) * 16384 + 768)
  int stage = 0;
  for (int t = 0; t < KT; ++t) {
    const unsigned so = lds0 + stage * STAGE;
    const unsigned va = so + a_lane, vb = so + b_lane;
    const bool n1 = t + 1 < KT, n2 = t + 2 < KT;   // wave-uniform
    // P1
    CC_PP_B(b, vb, 0);
    CC_PP_A(a0, va, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n1) issue_b(1, 0, stage ^ 1);
    bar();
    mfma16(0, a0, b);
    bar();
    // P2
    CC_PP_A(a1, va, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (n1) {
      issue_a(1, 1, stage ^ 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // k1 of this K-step has landed (read in P3)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    mfma16(1, a1, b);
    bar();
    // P3
    CC_PP_B(b, vb, 1);
    CC_PP_A(a0, va, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n1) issue_b(1, 1, stage ^ 1);
    bar();
    mfma16(0, a0, b);
    bar();
    // P4
    CC_PP_A(a1, va, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (n2) {
      issue_a(2, 0, stage);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // k0 of the next K-step has landed (read in its P1)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    mfma16(1, a1, b);
    bar();
    src.a_cur[0] += A_STEP; src.a_cur[1] += A_STEP; src.b_cur += B_STEP;
    stage ^= 1;
  }
  if (!wm) bar();   // waves 0-3 wait for the other group's last phase: every read of the stages is done
#undef CC_PP_A
#undef CC_PP_B
}

// The two-plane form of the ping-pong (f16x2, segments that multiply W_hi and W_lo): K-step 32 through a ring of three
// 48 KiB stages {A, W_hi, W_lo} (the layout of gemm16_segment), four phases per K-step — (rows 0-63, W_hi),
// (rows 64-127, W_hi), (rows 0-63, W_lo), (rows 64-127, W_lo): per accumulator hi then lo of a K-step, as before.
// Three DMA pieces per K-step, each requested two phases after the last read of what it overwrites, six younger
// pieces in flight at either wait:
//   P1(u): read W_hi, A rows 0-63 | request W_lo(u+2)        P2(u): read A rows 64-127 | wait: W_lo(u) has landed
//   P3(u): read W_lo | request W_hi(u+3)                     P4(u): request A(u+3) | wait: W_hi, A of u+1 have landed
// (One-term segments — the A_lo x W_hi product of layer 2, 4 % of its MFMAs — keep gemm16_segment.)
__device__ __forceinline__ void gemm16_segment_pp2(f32x4 (&acc)[8][4], Src16<2> src, int NU, char *smem, int a_lane,
                                                   int b_lane, int wave, int wm) {
  using C_ = Cfg16<2>;
  constexpr int NS = C_::NSTAGE, STAGE = C_::STAGE, STAGE_A = C_::STAGE_A, PLANE_B = C_::PLANE_B;
  static_assert(NS == 3 && C_::TK == 32 && C_::A_INSTR == 2 && C_::B_INSTR == 2, "ring of three 48 KiB stages");
  constexpr int A_STEP = STAGE_A, B_STEP = 2 * PLANE_B;   // one A16T tile (16 KiB) per K-step of 32
  auto issue_a = [&](int x, int slot) {
    char *dst = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      CC_GLDS16(src.a_base[i] + (int64_t)x * A_STEP + src.a_voff[i], dst + (i * H_THREADS + wave * 64) * 16);
  };
  auto issue_w = [&](int x, int plane, int slot) {
    char *dst = smem + slot * STAGE + STAGE_A + plane * PLANE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      CC_GLDS16(src.b_base + ((int64_t)x * B_STEP + plane * PLANE_B + i * (H_THREADS * 16)) + src.b_voff,
                dst + (i * H_THREADS + wave * 64) * 16);
  };
  auto mfma16 = [&](int hf, const h8 (&af)[4], const h8 (&bf)[4]) {
    CC_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[4 * hf + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[4 * hf + mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // prologue, in the order the counted waits assume: hi0 A0 lo0 hi1 A1 lo1 hi2 A2
  issue_w(0, 0, 0); issue_a(0, 0); issue_w(0, 1, 0);
  if (NU > 1) { issue_w(1, 0, 1); issue_a(1, 1); issue_w(1, 1, 1); }
  if (NU > 2) { issue_w(2, 0, 2); issue_a(2, 2); }
  if (NU > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // W_hi, A of K-step 0 have landed
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar();
  if (wm) bar();   // waves 4-7 run one barrier behind
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  h8 a0[4], a1[4], bh[4], bl[4];
#define CC_P2_A(dst, base, hf)                                                                   \
  CC_DSR128(dst[0], base, (4 * (hf) + 0) * 1024); CC_DSR128(dst[1], base, (4 * (hf) + 1) * 1024); \
  CC_DSR128(dst[2], base, (4 * (hf) + 2) * 1024); CC_DSR128(dst[3], base, (4 * (hf) + 3) * 1024)
#define CC_P2_B(dst, base, plane)                                                                              \
  CC_DSR128(dst[0], base, (plane) * 16384 + 0); CC_DSR128(dst[1], base, (plane) * 16384 + 256);                \
  CC_DSR128(dst[2], base, (plane) * 16384 + 512); CC_DSR128(dst[3], base, (plane) * 16384 + 768)
  static_assert(PLANE_B == 16384, "plane offset in the read macros");
  int slot = 0, slot2 = 2;   // u % 3, (u + 2) % 3
  for (int u = 0; u < NU; ++u) {
    const unsigned so = lds0 + slot * STAGE;
    const unsigned va = so + a_lane, vb = so + b_lane;
    // P1
    CC_P2_B(bh, vb, 0);
    CC_P2_A(a0, va, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < NU) issue_w(u + 2, 1, slot2);
    bar();
    mfma16(0, a0, bh);
    bar();
    // P2
    CC_P2_A(a1, va, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < NU) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // W_lo of this K-step has landed (read in P3)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    mfma16(1, a1, bh);
    bar();
    // P3
    CC_P2_B(bl, vb, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < NU) issue_w(u + 3, 0, slot);
    bar();
    mfma16(0, a0, bl);
    bar();
    // P4
    if (u + 3 < NU) {
      issue_a(u + 3, slot);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // W_hi, A of the next K-step have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    mfma16(1, a1, bl);
    bar();
    slot = slot == NS - 1 ? 0 : slot + 1;
    slot2 = slot2 == NS - 1 ? 0 : slot2 + 1;
  }
  if (!wm) bar();   // waves 0-3 wait for the other group's last phase: every read of the stages is done
#undef CC_P2_A
#undef CC_P2_B
}

// PLANES: weight planes in the packed matrix (1: CC_DTYPE_F16, 2: CC_DTYPE_F16X2).
// OUT_MODE 0: C fp32 [M x ldc] row-major;  1: C_hi (and C_lo when split) fp16 planes, A16T layout of ldc columns
template <int PLANES, int OUT_MODE, bool HALVES>
__global__ __launch_bounds__(H_THREADS, 1) void gemm16_kernel(Segs16 segs, int64_t M, const _Float16 *__restrict__ Wp,
                                                              const float *__restrict__ bias, float inv_scale,
                                                              int KTW, int N, int MT, int NT, void *__restrict__ C,
                                                              void *__restrict__ C_lo, int64_t ldc, int act, int tile0) {
  constexpr bool halves = HALVES;
  using C_ = Cfg16<PLANES>;
  constexpr int TK = C_::TK, KQ = C_::KQ, CH = C_::CH, STAGE_A = C_::STAGE_A;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r16 = lane & 15, q = lane >> 4;

  // XCD-aware grouped tile order: blockIdx % 8 = XCD under round-robin dispatch gets a contiguous slab of tiles,
  // 8 m-tiles share each weight tile consecutively
  // A launch covers the tiles tile0 .. of that order.  `halves`: every tile is worked by TWO workgroups, each on 128
  // of its 256 rows (8 waves as 2 x 4 of 64 x 64): the last, partly filled round of a product runs this way on twice
  // the CUs (launch_gemm16).  A row's sums are the same MFMAs in the same order either way.
  const int nwg = halves ? (int)(gridDim.x >> 1) : (int)gridDim.x, bid = halves ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int wgid = tile0 + (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int per_group = 8 * NT;
  const int first_m = (wgid / per_group) * 8;
  const int gsz = (MT - first_m < 8) ? MT - first_m : 8;
  const int m_tile = first_m + (wgid % per_group) % gsz;
  const int n_tile = (wgid % per_group) / gsz;
  const int64_t m0 = (int64_t)m_tile * H_TM + (halves ? (int64_t)(blockIdx.x & 1) * (H_TM / 2) : 0);
  if (m0 >= M) return;                                         // the lower half of a tile that has <= 128 rows
  const int64_t m_lim = (halves && m0 + H_TM / 2 < M) ? m0 + H_TM / 2 : M;   // rows this workgroup computes: [m0, m_lim)
  const int wrow = halves ? wm * 64 : wm * 128;                // first row of the wave inside the workgroup's rows
  const int mode = m0 + wrow >= m_lim ? W_IDLE : (halves ? W_HALF : W_FULL);   // wave-uniform
  const int n0 = n_tile * H_TN;

  // DMA geometry: the A operand is stored as A16T tiles (common.h) = the LDS image itself, so a K-step's A is STAGE_A
  // contiguous bytes of the 256-row block this tile lies in, copied 8 KiB per instruction (thread tid: bytes tid * 16).
  // A half-height workgroup copies the whole block's tiles and reads its own 128 rows of them.  Rows beyond M hold
  // whatever the buffer holds; their results are not stored.
  const int row_in_blk = halves ? (int)(blockIdx.x & 1) * (H_TM / 2) : 0;   // first row of this workgroup inside the 256-row block
  (void)CH;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage
  int a_off[8][KQ];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int row = row_in_blk + wrow + mt * 16 + r16;   // (half-height tiles use row blocks 0-3 only)
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) a_off[mt][kq] = kq * 16384 + (row * 4 + (q ^ swz16<4>(row))) * 16;   // k-half plane kq of the stage
  }
  const int b_off = STAGE_A + (q * 256 + wn * 64 + r16) * 16;   // + plane * PLANE_B + kq * 16384 + nt * 256

  for (int si = 0; si < segs.n; ++si) {
    const Seg16 sg = segs.s[si];
    const int KT = sg.KT;
    if (KT == 0) continue;
    Src16<PLANES> src;
    // tile (m_tile, 0) of this segment's A16T matrix; K-step t = the STAGE_A bytes at + t * STAGE_A
    const char *a_tiles = reinterpret_cast<const char *>(sg.A) + (int64_t)m_tile * (sg.lda >> 5) * 16384;
#pragma unroll
    for (int i = 0; i < C_::A_INSTR; ++i) {
      src.a_base[i] = a_tiles + i * (H_THREADS * 16);
      src.a_voff[i] = (unsigned)tid * 16u;
    }
    src.b_base = reinterpret_cast<const char *>(Wp + ((int64_t)n_tile * KTW + sg.wkt0) * (PLANES * TK * H_TN));
    src.b_voff = (unsigned)tid * 16u;

    if constexpr (PLANES == 1 && !HALVES) {
      static_assert(C_::TK == 64 && C_::LDS == 131072, "ping-pong: K-step 64, two 64 KiB stages");
      SrcPP sp;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        sp.a_cur[i] = a_tiles + i * (H_THREADS * 16);
        sp.a_voff[i] = (unsigned)tid * 16u;
      }
      sp.b_cur = src.b_base;
      sp.b_voff = src.b_voff;
      const int prow_w = wm * 128 + r16;
      const int a_lane_pp = (prow_w * 4 + (q ^ swz16<4>(prow_w))) * 16;
      gemm16_segment_pp(acc, sp, KT, smem, a_lane_pp, b_off, wave, wm);
    } else if constexpr (PLANES == 2 && !HALVES) {
      if (sg.nterms == 2) gemm16_segment_pp2(acc, src, KT, smem, a_off[0][0], b_off, wave, wm);
      else gemm16_segment<PLANES, false, HALVES>(acc, src, KT, smem, a_off, b_off, wave, mode);   // A_lo x W_hi
    } else {   // half-height tiles
      if (PLANES == 2 && sg.nterms == 2) gemm16_segment<PLANES, true, HALVES>(acc, src, KT, smem, a_off, b_off, wave, mode);
      else gemm16_segment<PLANES, false, HALVES>(acc, src, KT, smem, a_off, b_off, wave, mode);
    }
  }

  // epilogue: undo the weight scale, bias, activation.  Lane (r16, q) holds D[4q + j][r16] of every 16 x 16 block.
  // (The lane id is laundered through an empty asm so that hipcc does not compute the epilogue's addresses and bias
  // loads ahead of the K loop, where they cost registers the loop needs: it spilled 22 of them.)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int r16e = lane_e & 15, qe = lane_e >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = n0 + wn * 64 + nt * 16 + r16e;
    const bool col_ok = col < N;
    const float bcol = col_ok ? bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t row = m0 + wrow + mt * 16 + 4 * qe + j;
        if (col_ok && row < m_lim && (mt < 4 || !halves)) {
          float z = acc[mt][nt][j] * inv_scale + bcol;
          z = (act == ACT_SIGMOID) ? sigmoid16_(z) : z;
          if (OUT_MODE == 0) {
            static_cast<float *>(C)[row * ldc + col] = z;
          } else {
            const _Float16 zh = (_Float16)z;   // fp16 planes feed the next fp16 GEMM: A16T layout, ldc = its column count
            const int64_t at = a16t_offset(row, col, ldc);
            static_cast<_Float16 *>(C)[at] = zh;
            if (C_lo) static_cast<_Float16 *>(C_lo)[at] = (_Float16)(z - (float)zh);
          }
        }
      }
    }
  }
}

// ---- weight packing: [NT][KT][plane][TK / 8][256][8] fp16 = per (n-tile, K-step, plane) the linear LDS image ------
// 16-byte unit u of a plane tile = (j = u / 256, i = u % 256) holds k = 8j .. 8j+7 of column i.
template <int PLANES>
__global__ void pack_weight16_kernel(const float *__restrict__ w, int64_t K1, int64_t K2, int64_t N, int64_t KT1,
                                     int64_t KT, int64_t NT, float scale, _Float16 *__restrict__ dst) {
  constexpr int TK = Cfg16<PLANES>::TK, UNITS = TK / 8 * H_TN;   // 16-byte units per plane tile
  const int64_t total = NT * KT * (int64_t)UNITS;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
    const int within = (int)(u % UNITS);
    const int i = within % H_TN, j = within / H_TN;
    const int64_t tile = u / UNITS;
    const int64_t kt = tile % KT, nt = tile / KT;
    const int64_t n = nt * H_TN + i;
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int64_t src_row;
      bool ok;
      if (kt < KT1) { const int64_t k = kt * TK + 8 * j + c; ok = k < K1; src_row = k; }
      else { const int64_t k = (kt - KT1) * TK + 8 * j + c; ok = k < K2; src_row = K1 + k; }
      const float v = (ok && n < N) ? w[src_row * N + n] * scale : 0.f;
      hi[c] = (_Float16)v;
      lo[c] = (_Float16)(v - (float)hi[c]);
    }
    _Float16 *d_hi = dst + (tile * PLANES + 0) * (int64_t)(UNITS * 8) + (int64_t)within * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) d_hi[c] = hi[c];
    if (PLANES == 2) {
      _Float16 *d_lo = dst + (tile * PLANES + 1) * (int64_t)(UNITS * 8) + (int64_t)within * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) d_lo[c] = lo[c];
    }
  }
}

void free_weight16(PackedWeight16 *w) {
  if (w->d_w) (void)hipFree(w->d_w);
  if (w->d_bias) (void)hipFree(w->d_bias);
  *w = PackedWeight16();
}

int pack_weight16(const float *w, const float *bias, int64_t K1, int64_t K2, int64_t N, int planes, PackedWeight16 *out) {
  free_weight16(out);
  CC_REQUIRE(planes == 1 || planes == 2, "pack_weight16: 1 or 2 planes");
  const int TK = planes == 1 ? Cfg16<1>::TK : Cfg16<2>::TK;
  out->K1 = K1; out->K2 = K2; out->N = N; out->planes = planes; out->TK = TK;
  out->KT1 = ceil_div(K1, TK); out->KT2 = ceil_div(K2, TK);
  out->NT = ceil_div(N, H_TN);
  const int64_t KT = out->KT1 + out->KT2;
  if (N == 0) return CC_OK;
  // power-of-two scale bringing max|W| into [2^13, 2^14): keeps W_lo out of the fp16 subnormals
  float mx = 0.f;
  for (int64_t i = 0; i < (K1 + K2) * N; ++i) { const float a = fabsf(w[i]); mx = a > mx ? a : mx; }
  int e = 0;
  float scale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { (void)frexpf(mx, &e); scale = ldexpf(1.f, 14 - e); }
  out->scale = scale;
  const int64_t units = out->NT * KT * (int64_t)(TK / 8 * H_TN);
  DevPtr raw;
  if ((K1 + K2) > 0) {
    CC_HIP(raw.alloc(sizeof(float) * (size_t)((K1 + K2) * N)));
    CC_HIP(hipMemcpyAsync(raw.p, w, sizeof(float) * (size_t)((K1 + K2) * N), hipMemcpyHostToDevice, stream()));
  }
  if (units > 0) {
    CC_HIP(hipMalloc(&out->d_w, 16 * (size_t)planes * (size_t)units));
    int64_t blocks = ceil_div(units, 256);
    if (blocks > 65536) blocks = 65536;
    if (planes == 1)
      hipLaunchKernelGGL(pack_weight16_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream(), raw.as<float>(), K1, K2, N,
                         out->KT1, KT, out->NT, scale, static_cast<_Float16 *>(out->d_w));
    else
      hipLaunchKernelGGL(pack_weight16_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream(), raw.as<float>(), K1, K2, N,
                         out->KT1, KT, out->NT, scale, static_cast<_Float16 *>(out->d_w));
    CC_HIP(hipGetLastError());
  }
  CC_HIP(hipMalloc(&out->d_bias, sizeof(float) * (size_t)(out->NT * H_TN)));
  CC_HIP(hipMemsetAsync(out->d_bias, 0, sizeof(float) * (size_t)(out->NT * H_TN), stream()));
  CC_HIP(hipMemcpyAsync(out->d_bias, bias, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, stream()));
  CC_HIP(hipStreamSynchronize(stream()));
  return CC_OK;
}

int launch_gemm16(const GemmSeg16 *segs, int nsegs, int64_t M, const PackedWeight16 &W, void *C, void *C_lo,
                  int64_t ldc, int out_fp16, int act, int prof_class) {
  if (M == 0 || W.N == 0) return CC_OK;
  CC_REQUIRE(W.d_bias != nullptr, "gemm16: weight not packed");
  CC_REQUIRE(nsegs >= 1 && nsegs <= 4, "gemm16: 1..4 segments");
  Segs16 sg;
  sg.n = nsegs;
  for (int i = 0; i < nsegs; ++i) {
    const GemmSeg16 &g = segs[i];
    const int64_t KT = g.which == 0 ? W.KT1 : W.KT2;
    CC_REQUIRE(g.nterms == 1 || (g.nterms == 2 && W.planes == 2), "gemm16: segment %d asks for %d terms of a %d-plane weight",
               i, g.nterms, W.planes);
    CC_REQUIRE(KT == 0 || (g.A && g.lda >= KT * W.TK && g.lda % W.TK == 0),
               "gemm16: segment %d needs an A16T matrix of a multiple of %d columns, >= %lld (got %lld)", i, (int)W.TK,
               (long long)(KT * W.TK), (long long)g.lda);
    sg.s[i].A = static_cast<const _Float16 *>(g.A);
    sg.s[i].lda = g.lda;
    sg.s[i].KT = (int)KT;
    sg.s[i].wkt0 = g.which == 0 ? 0 : (int)W.KT1;
    sg.s[i].nterms = g.nterms;
  }
  const int64_t MT = ceil_div(M, H_TM);
  const int64_t nwg = MT * W.NT;
  CC_REQUIRE(nwg < (int64_t)2147483647, "gemm16: grid too large");
  static int attr_gen = -1;   // per library context: redone after cc_shutdown + cc_init (maybe another device)
  if (attr_gen != init_generation()) {
#define CC_ATTR16(PL, OM, HV) \
  CC_HIP(hipFuncSetAttribute((const void *)gemm16_kernel<PL, OM, HV>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg16<PL>::LDS))
    CC_ATTR16(1, 0, false); CC_ATTR16(1, 1, false); CC_ATTR16(2, 0, false); CC_ATTR16(2, 1, false);
    CC_ATTR16(1, 0, true); CC_ATTR16(1, 1, true); CC_ATTR16(2, 0, true); CC_ATTR16(2, 1, true);
#undef CC_ATTR16
    attr_gen = init_generation();
  }
  ProfScope prof(prof_class, 2.0 * (double)M * (double)(W.K1 + W.K2) * (double)W.N);
  const _Float16 *wp = static_cast<const _Float16 *>(W.d_w);
  const int KTW = (int)(W.KT1 + W.KT2);
  // One workgroup per CU: the product runs in rounds of g_num_cus tiles and its last round costs a whole tile time
  // however few tiles it holds.  A last round that is at most half full runs as half-height tiles instead — two
  // workgroups per tile, 128 rows each, on twice the CUs (0.6 of a tile time) — in a second launch behind the whole
  // rounds.  (configs[4]'s N = 3300 is 13 column tiles, which no pass size turns into whole rounds.)
  const int64_t cus = num_cus();
  const int64_t tail = nwg % cus;
  // (not while a call alternates its passes between the two library streams: there the other stream's tiles fill the
  // partial round, and a second launch per product only delays them — a 12.5k-row shard step 4.3 -> 4.8 ms)
  const int64_t whole = (2 * tail <= cus && !two_streams_active()) ? nwg - tail : nwg;   // tiles of the full-height launch
#define CC_LAUNCH16(PL, OM, HV, GRID, TILE0)                                                                         \
  hipLaunchKernelGGL((gemm16_kernel<PL, OM, HV>), dim3((unsigned)(GRID)), dim3(H_THREADS), Cfg16<PL>::LDS, stream(), \
                     sg, M, wp, W.d_bias, 1.0f / W.scale, KTW, (int)W.N, (int)MT, (int)W.NT, C, C_lo, ldc, act,      \
                     (int)(TILE0))
#define CC_LAUNCH16_ALL(HV, GRID, TILE0)                                                                       \
  do {                                                                                                         \
    if (W.planes == 1) { if (out_fp16) CC_LAUNCH16(1, 1, HV, GRID, TILE0); else CC_LAUNCH16(1, 0, HV, GRID, TILE0); } \
    else { if (out_fp16) CC_LAUNCH16(2, 1, HV, GRID, TILE0); else CC_LAUNCH16(2, 0, HV, GRID, TILE0); }               \
  } while (0)
  if (whole > 0) CC_LAUNCH16_ALL(false, whole, 0);
  if (whole < nwg) CC_LAUNCH16_ALL(true, 2 * (nwg - whole), whole);
#undef CC_LAUNCH16_ALL
#undef CC_LAUNCH16
  CC_HIP(hipGetLastError());
  return CC_OK;
}

}  // namespace cc
