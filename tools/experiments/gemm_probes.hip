// K2/K4 — Dense layers of the BiGAN encoder as fp32 MFMA GEMMs with a fused
// bias + activation epilogue, plus the row-softmax of the classify encoder.
//
// Replaces keras `layers.Dense(..., activation=tf.nn.sigmoid)` and
// `layers.Concatenate()` of the encoder graphs (reference
// src/bigan_cont.py:31-41, src/bigan_classify.py:29-40) at inference.
//
//   C[M x N] = act( A1[M x K1] @ W[0:K1] + A2[M x K2] @ W[K1:K1+K2] + bias )
//
// The Concatenate is never materialised: the concatenated input is expressed
// as two K-segments (A1 then A2) of one accumulation.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 products, fp32 accumulate
// (gfx950 has no xf32/TF32; dense fp32 matrix peak 157.3 TFLOP/s).
//
// Tiling: 128 x 128 output tile per 256-thread workgroup, 4 waves as 2 x 2,
// each wave 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs).  K advances in
// steps of 32 through LDS:
//   A tile  [128][32+4] fp32, +4 pad → ds_read_b128 of 4 consecutive k per
//           lane is bank-conflict free (36·r mod 64 distinct for r mod 16).
//   W tile  [8][128][4] fp32: weights are pre-packed on upload so that the 4
//           k-values a lane feeds to 4 consecutive MFMAs are one 16-byte read
//           and a whole 16 KiB tile is contiguous in HBM.
// Lane (r = lane&31, h = lane>>5) of MFMA step (t,u) supplies k = 8t + 4h + u.
// Global → register prefetch of K-step t+1 is issued before the MFMAs of step
// t and written to LDS after them (2 barriers per K-step; 3 workgroups per CU
// cover each other's barrier gaps — the CU's arbiter serves its oldest one first).
//
// Workgroup → tile map: XCD-aware (blockIdx%8 = XCD under round-robin dispatch
// gets a contiguous slab of tiles) with grouped ordering (8 M-tiles share each
// W tile consecutively) so that concurrently resident tiles on one XCD re-use
// A and W panels out of that XCD's 4 MiB L2.
#include "common.h"

namespace cc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// native vector (not HIP's float4 struct): arrays of it stay in registers after unrolling
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Tuned on MI355X (tools/gemm_sweep.sh): 3 workgroups per CU (<=168 VGPRs) and a raised
// priority over the MFMA cluster gave 127 -> 137 TFLOP/s on layer 1 (139 with the K loop as it is now).
#ifndef CC_GEMM_MINWAVES
#define CC_GEMM_MINWAVES 3
#endif
#ifndef CC_GEMM_SETPRIO
#define CC_GEMM_SETPRIO 1
#endif

// CC_GEMM_PACE=1 (experiment, off): workgroups that share panels are kept within a few K-steps of each
// other (chunks of CC_GEMM_PACE_EVERY K-steps, at most CC_GEMM_PACE_LEAD + 1 chunks ahead of the slowest
// member).  Measured on layer 1 (tools/experiments/round1/pace_sweep.sh, pmc_collect.sh): L2 hit rate 29 % -> 62 % (EVERY=8)
// / 81 % (EVERY=2), L2-miss traffic 33 -> 17.9 / 8.9 GB per launch, but 137.4 -> 134.7 / 128.5 TFLOP/s:
// the misses are served on-die and are not on the critical path, while lock-step issue is slower.
#ifndef CC_GEMM_PACE
#define CC_GEMM_PACE 0
#endif
#ifndef CC_GEMM_PACE_EVERY
#define CC_GEMM_PACE_EVERY 4
#endif
#ifndef CC_GEMM_PACE_LEAD
#define CC_GEMM_PACE_LEAD 2
#endif
#ifndef CC_GEMM_PACE_BUDGET
#define CC_GEMM_PACE_BUDGET 8192      // polls (~0.25 us each) a workgroup may spend waiting, in total
#endif
bool g_gemm_pace = false;             // set by the encoder around single-stream calls

constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = GEMM_BK;
constexpr int LDS_A_STRIDE = BK + 4;            // floats; (BK+4)·r mod 64 distinct for r mod 16
constexpr int GEMM_THREADS = 256;
constexpr int TILE_W_FLOAT4 = (BK / 4) * BN;    // float4 per packed W tile
constexpr int A_F4_PER_ROW = BK / 4;            // float4 per A-tile row
constexpr int A_ROWS_PER_PASS = GEMM_THREADS / A_F4_PER_ROW;
constexpr int NA = BM / A_ROWS_PER_PASS;        // A float4 staged per thread
constexpr int NB = TILE_W_FLOAT4 / GEMM_THREADS;  // W float4 staged per thread
#ifndef CC_GEMM_GROUP_M
#define CC_GEMM_GROUP_M 8
#endif
constexpr int GROUP_M = CC_GEMM_GROUP_M;
static_assert(BN == 128 && (BK == 32 || BK == 64), "tile shape");

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

#ifdef CC_GEMM_TRACE
// diagnostic build only (tools/experiments/round1/gemm_trace.py): per-workgroup wall-clock stamps of the layer-1 kernel
__device__ unsigned long long g_gemm_trace[16384 * 4];
#endif

// State of the (optional) pacing experiment carried through the K loop.
struct PaceState {
  int *pace_grp = nullptr;
  int pace_size = 0, pace_budget = 0, pace_seen = 0, wave_u = 0;
};

// The K loop of one wave (wave tile = MTV x NTV blocks of 32 x 32; only 2 x 2 is instantiated).
// Tried and dropped: edge variants that skip the blocks outside the matrix (N = 3000 leaves the second wave
// column of the last n-tile all padding, 1/48 of layer 1's MFMAs).  A second instantiation of this loop in
// the kernel made the allocator spill (336 B scratch), and a staging-only path for all-padding waves,
// which compiles cleanly, gains nothing: the workgroup still runs at the pace of its three full waves.
template <int KCLASS, int MTV, int NTV>
__device__ __forceinline__ void gemm_k_loop(f32x16 (&acc)[2][2], f32x4 (&ra)[NA], f32x4 (&rb)[NB], const float *(&pa)[NA],
                                            const f32x4 *&wt, const int (&a_lr)[NA], const float *__restrict__ A2,
                                            int64_t m0, int64_t lda2, int KT1, int KT, float *As, f32x4 *Bs4,
                                            const float *a_lds, const f32x4 *b_lds, int tid, int a_row, int a_k4,
                                            PaceState &ps, int bid) {
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;  // wave-uniform
    if (more) {
      if (kt + 1 == KT1) {  // first tile of segment 2: re-base the A pointers (once per kernel)
        const float *base = A2 + m0 * lda2;
#pragma unroll
        for (int p = 0; p < NA; ++p) pa[p] = base + a_lr[p] * (int)lda2 + 4 * a_k4;
      }
#pragma unroll
      for (int p = 0; p < NA; ++p) { ra[p] = *reinterpret_cast<const f32x4 *>(pa[p]); pa[p] += BK; }
#pragma unroll
      for (int p = 0; p < NB; ++p) rb[p] = wt[p * GEMM_THREADS];
      wt += TILE_W_FLOAT4;
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
#if CC_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(CC_GEMM_SETPRIO);
#endif
    if (MTV > 0 && NTV > 0) {
#pragma unroll
      for (int t = 0; t < BK / 8; ++t) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int mt = 0; mt < MTV; ++mt)
          a[mt] = *reinterpret_cast<const f32x4 *>(a_lds + mt * 32 * LDS_A_STRIDE + 8 * t);
#pragma unroll
        for (int nt = 0; nt < NTV; ++nt) b[nt] = b_lds[2 * t * BN + nt * 32];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int mt = 0; mt < MTV; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTV; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][u], b[nt][u], acc[mt][nt], 0, 0, 0);
      }
    }
#if CC_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifndef CC_GEMM_FAKE_NOSYNC   // probe: no barriers, no LDS refill (results are wrong): the pure MFMA + fragment-read rate
    __syncthreads();
#endif
    if (more) {
      float *aw = As + a_row * LDS_A_STRIDE + 4 * a_k4;
#pragma unroll
      for (int p = 0; p < NA; ++p) *reinterpret_cast<f32x4 *>(aw + p * A_ROWS_PER_PASS * LDS_A_STRIDE) = ra[p];
#pragma unroll
      for (int p = 0; p < NB; ++p) Bs4[tid + p * GEMM_THREADS] = rb[p];
    }
#if CC_GEMM_PACE
    // wave 0 as a whole (wave-uniform control flow, scalar state: nothing per-lane stays live across
    // the K loop): its 64 lanes' atomic adds are merged into one +64 by the compiler
    if (ps.pace_grp && ps.wave_u == 0 && (kt + 1) % CC_GEMM_PACE_EVERY == 0) {
      const int c = (kt + 1) / CC_GEMM_PACE_EVERY - 1;
      // the counter this boundary depends on was read one boundary ago (ps.pace_seen: a whole chunk for the
      // load to return), so a group that is in step never waits for memory here; nothing of this wave
      // is in flight at this point, so consuming ps.pace_seen does not wait for the atomic below either
      if (c >= CC_GEMM_PACE_LEAD && __builtin_amdgcn_readfirstlane(ps.pace_seen) < ps.pace_size) {
        const int *w = ps.pace_grp + c - CC_GEMM_PACE_LEAD;
        while (ps.pace_budget > 0 &&
               __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < ps.pace_size) {
          __builtin_amdgcn_s_sleep(8);
          --ps.pace_budget;
        }
      }
      __hip_atomic_fetch_add(ps.pace_grp + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c + 1 >= CC_GEMM_PACE_LEAD)
        ps.pace_seen = __hip_atomic_load(ps.pace_grp + c + 1 - CC_GEMM_PACE_LEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
#ifndef CC_GEMM_FAKE_NOSYNC
    __syncthreads();
#endif
#ifdef CC_GEMM_TRACE
    if (KCLASS == 1 && tid == 0 && bid < 16384 && kt == KT / 2) g_gemm_trace[4 * bid + 1] = wall_clock64();
#endif
  }
}

// KCLASS only names the instantiation (so profilers tell layer 1 / layer 2 / tail apart).
template <int KCLASS>
__global__
#if CC_GEMM_MINWAVES > 0
__launch_bounds__(GEMM_THREADS, CC_GEMM_MINWAVES)
#else
__launch_bounds__(GEMM_THREADS)
#endif
void gemm_bias_act_kernel(const float *__restrict__ A1, int64_t lda1, const float *__restrict__ A2,
                          int64_t lda2, int64_t M, const f32x4 *__restrict__ Wp,
                          const float *__restrict__ bias, int KT1, int KT, int N, int MT, int NT,
                          float *__restrict__ C, int64_t ldc, int act, int *__restrict__ pace, int pace_nch) {
  constexpr int STAGE_FLOATS = BM * LDS_A_STRIDE + TILE_W_FLOAT4 * 4;
  __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS];
  float *As = lds;
  f32x4 *Bs4 = reinterpret_cast<f32x4 *>(lds + BM * LDS_A_STRIDE);

  // ---- workgroup → tile ---------------------------------------------------
  const int nwg = MT * NT;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
  const int wgid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  const int per_group = GROUP_M * NT;
  const int first_m = (wgid / per_group) * GROUP_M;
  const int gsz = (MT - first_m < GROUP_M) ? MT - first_m : GROUP_M;
  const int m_tile = first_m + (wgid % per_group) % gsz;
  const int n_tile = (wgid % per_group) / gsz;
  const int64_t m0 = (int64_t)m_tile * BM;
  const int n0 = n_tile * BN;
#if CC_GEMM_PACE
  // Pacing group = the 32 workgroups dispatched back to back to this XCD (one per CU): the CU's wave
  // arbiter serves its oldest workgroup first, so these advance through K at the same rate while the
  // three workgroups sharing a CU do not (measured: K/2 reached at 1.95 / 3.08 / 3.29 ms).  Keeping a
  // group within a few K-steps keeps its A/W panels (8 m-tiles x 4 n-tiles) in the XCD's L2.  Waiting is
  // bounded (poll budget), so a group member that is not resident costs time, never progress.
  int *pace_grp = nullptr;
  int pace_size = 0, pace_budget = CC_GEMM_PACE_BUDGET, pace_seen = 0;
  if (pace) {
    const int n_local = q + (xcd < rem ? 1 : 0), j = bid >> 3;
    const int grp = j >> 5;
    pace_size = 64 * (n_local - 32 * grp < 32 ? n_local - 32 * grp : 32);   // every member adds 64 per chunk
    pace_grp = pace + ((int64_t)xcd * ((q + 32) >> 5) + grp) * pace_nch;
  }
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
#if CC_GEMM_PACE
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#endif
#ifdef CC_GEMM_TRACE
  if (KCLASS == 1 && tid == 0 && bid < 16384) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_gemm_trace[4 * bid + 0] = wall_clock64();
    g_gemm_trace[4 * bid + 3] = ((unsigned long long)xcc << 32) | hw;
  }
#endif

  // ---- staging: per-thread running pointers into A (re-based once at the
  //      segment switch) and a scalar running pointer into the packed W tiles ----
  const int a_row = tid / A_F4_PER_ROW;  // + A_ROWS_PER_PASS · p
  const int a_k4 = tid % A_F4_PER_ROW;
  int a_lr[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int64_t gr = m0 + a_row + A_ROWS_PER_PASS * p;
    a_lr[p] = (int)((gr < M ? gr : M - 1) - m0);  // clamp the M tail onto a valid row
  }
  const float *pa[NA];
  {
#ifdef CC_GEMM_FAKE_A   // sensitivity probe: every workgroup reads the first m-tile's rows (results are wrong)
    const int64_t m0 = 0;
#endif
    const float *base = KT1 > 0 ? A1 + m0 * lda1 : A2 + m0 * lda2;
    const int lda = (int)(KT1 > 0 ? lda1 : lda2);
#pragma unroll
    for (int p = 0; p < NA; ++p) pa[p] = base + a_lr[p] * lda + 4 * a_k4;
  }
#ifdef CC_GEMM_FAKE_W   // sensitivity probe: every workgroup reads the first n-tile's weights (results are wrong)
  const f32x4 *wt = Wp + tid;
#else
  const f32x4 *wt = Wp + (int64_t)n_tile * KT * TILE_W_FLOAT4 + tid;
#endif

  // The staging registers must stay in VGPRs (hipcc demotes them to scratch when
  // they are touched through lambdas or selected between arrays) so that the
  // loads stay in flight behind the MFMAs of the current K-step.  Register assignment
  // matters too: with `tid` and one more per-lane value alive across the K loop the
  // allocator moved the accumulators from v[0:15].. to v[2:17].. (and the A/B operands
  // with them) and the same instruction stream ran 11 % slower (137 -> 123 TFLOP/s), so
  // the accumulators are pinned below (any 16-aligned VGPR placement measured the same; AGPRs were
  // slower: the 84/84 register split spills).  The fp16 and kNN kernels did not react to the same pin
  // (tools/pin_sweep.sh).
  f32x4 ra[NA], rb[NB];

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // pin the four accumulator tuples where the fast build has them (see the note above)
  asm volatile("" : "+{v[0:15]}"(acc[1][1]));
  asm volatile("" : "+{v[16:31]}"(acc[0][1]));
  asm volatile("" : "+{v[32:47]}"(acc[1][0]));
  asm volatile("" : "+{v[48:63]}"(acc[0][0]));

  if (KT > 0) {
#pragma unroll
    for (int p = 0; p < NA; ++p) { ra[p] = *reinterpret_cast<const f32x4 *>(pa[p]); pa[p] += BK; }
#pragma unroll
    for (int p = 0; p < NB; ++p) rb[p] = wt[p * GEMM_THREADS];
    wt += TILE_W_FLOAT4;
    float *aw = As + a_row * LDS_A_STRIDE + 4 * a_k4;
#pragma unroll
    for (int p = 0; p < NA; ++p) *reinterpret_cast<f32x4 *>(aw + p * A_ROWS_PER_PASS * LDS_A_STRIDE) = ra[p];
#pragma unroll
    for (int p = 0; p < NB; ++p) Bs4[tid + p * GEMM_THREADS] = rb[p];
  }
  __syncthreads();

  const float *a_lds = As + (wm * 64 + r) * LDS_A_STRIDE + 4 * h;
  const f32x4 *b_lds = Bs4 + h * BN + wn * 64 + r;

  PaceState ps;
#if CC_GEMM_PACE
  ps.pace_grp = pace_grp; ps.pace_size = pace_size; ps.pace_budget = pace_budget; ps.wave_u = wave_u;
#endif
#define CC_KLOOP(MV, NV) gemm_k_loop<KCLASS, MV, NV>(acc, ra, rb, pa, wt, a_lr, A2, m0, lda2, KT1, KT, As, Bs4, a_lds, b_lds, tid, a_row, a_k4, ps, bid)
  CC_KLOOP(2, 2);
#undef CC_KLOOP
#ifdef CC_GEMM_TRACE
  if (KCLASS == 1 && tid == 0 && bid < 16384) g_gemm_trace[4 * bid + 2] = wall_clock64();
#endif

  // ---- epilogue: bias + activation, 128-byte row segments per half-wave ------
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = n0 + wn * 64 + nt * 32 + r;
    const bool col_ok = col < N;
    const float bcol = col_ok ? bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (col_ok && row < M) {
          float z = acc[mt][nt][e] + bcol;
          C[row * ldc + col] = (act == ACT_SIGMOID) ? sigmoidf_(z) : (act == ACT_RELU ? fmaxf(z, 0.f) : z);
        }
      }
    }
  }
}

// ---- weight packing ----------------------------------------------------------
// dst float index f → (n_tile, kt, j, i, c): k = 4j + c within the k-tile,
// n = n_tile·128 + i.
__global__ void pack_weight_kernel(const float *__restrict__ w, int64_t K1, int64_t K2, int64_t N,
                                   int64_t KT1, int64_t KT, int64_t NT, float *__restrict__ dst) {
  constexpr int TILE_FLOATS = TILE_W_FLOAT4 * 4;
  const int64_t total = NT * KT * (int64_t)TILE_FLOATS;
  for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total;
       f += (int64_t)gridDim.x * blockDim.x) {
    const int within = (int)(f % TILE_FLOATS);
    const int c = within & 3;
    const int i = (within >> 2) & (BN - 1);
    const int j = within / (4 * BN);
    const int64_t tile = f / TILE_FLOATS;
    const int64_t kt = tile % KT, nt = tile / KT;
    const int64_t n = nt * BN + i;
    int64_t src_row;
    bool ok;
    if (kt < KT1) {
      int64_t k = kt * BK + 4 * j + c;
      ok = k < K1; src_row = k;
    } else {
      int64_t k = (kt - KT1) * BK + 4 * j + c;
      ok = k < K2; src_row = K1 + k;
    }
    dst[f] = (ok && n < N) ? w[src_row * N + n] : 0.f;
  }
}

void free_weight(PackedWeight *w) {
  if (w->d_w) (void)hipFree(w->d_w);
  if (w->d_bias) (void)hipFree(w->d_bias);
  *w = PackedWeight();
}

int pack_weight(const float *w, const float *bias, int64_t K1, int64_t K2, int64_t N,
                PackedWeight *out) {
  free_weight(out);
  out->K1 = K1; out->K2 = K2; out->N = N;
  out->KT1 = ceil_div(K1, BK); out->KT2 = ceil_div(K2, BK);
  out->NT = ceil_div(N, BN);
  const int64_t KT = out->KT1 + out->KT2;
  if (N == 0) return CC_OK;
  const int64_t packed = out->NT * KT * (int64_t)TILE_W_FLOAT4 * 4;
  float *d_raw = nullptr;
  if ((K1 + K2) > 0) {
    CC_HIP(hipMalloc(&d_raw, sizeof(float) * (size_t)((K1 + K2) * N)));
    CC_HIP(hipMemcpyAsync(d_raw, w, sizeof(float) * (size_t)((K1 + K2) * N), hipMemcpyHostToDevice, stream()));
  }
  if (packed > 0) {
    CC_HIP(hipMalloc(&out->d_w, sizeof(float) * (size_t)packed));
    int64_t blocks = ceil_div(packed, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), d_raw, K1, K2, N,
                       out->KT1, KT, out->NT, out->d_w);
    CC_HIP(hipGetLastError());
  }
  CC_HIP(hipMalloc(&out->d_bias, sizeof(float) * (size_t)(out->NT * BN)));
  CC_HIP(hipMemsetAsync(out->d_bias, 0, sizeof(float) * (size_t)(out->NT * BN), stream()));
  CC_HIP(hipMemcpyAsync(out->d_bias, bias, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, stream()));
  CC_HIP(hipStreamSynchronize(stream()));
  if (d_raw) CC_HIP(hipFree(d_raw));
  return CC_OK;
}

int launch_gemm(const float *A1, int64_t lda1, const float *A2, int64_t lda2, int64_t M,
                const PackedWeight &W, float *C, int64_t ldc, int act, int prof_class) {
  if (M == 0 || W.N == 0) return CC_OK;
  CC_REQUIRE(W.d_bias != nullptr, "gemm: weight not packed");
  CC_REQUIRE(W.KT1 == 0 || (A1 && lda1 >= W.KT1 * BK && lda1 % 4 == 0),
             "gemm: segment 1 needs lda >= %lld (got %lld)", (long long)(W.KT1 * BK), (long long)lda1);
  CC_REQUIRE(W.KT2 == 0 || (A2 && lda2 >= W.KT2 * BK && lda2 % 4 == 0),
             "gemm: segment 2 needs lda >= %lld (got %lld)", (long long)(W.KT2 * BK), (long long)lda2);
  CC_REQUIRE(lda1 * BM < (int64_t)2147483647 && lda2 * BM < (int64_t)2147483647, "gemm: lda too large");
  const int64_t MT = ceil_div(M, BM);
  const int64_t nwg = MT * W.NT;
  CC_REQUIRE(nwg < (int64_t)2147483647, "gemm: grid too large");
  // pacing counters: [8 XCDs][groups of 32 workgroups][chunks of K-steps], zeroed per launch
  int *pace = nullptr;
  int pace_nch = 0;
#if CC_GEMM_PACE
  static int *pace_buf = nullptr;
  static size_t pace_cap = 0;
  if (g_gemm_pace && prof_class != CC_PROF_GEMM_TAIL && nwg >= 512) {
    pace_nch = (int)ceil_div(W.KT1 + W.KT2, CC_GEMM_PACE_EVERY);
    const size_t need = sizeof(int) * 8 * (size_t)(((nwg >> 3) + 32) >> 5) * (size_t)pace_nch;
    if (need > pace_cap) {
      if (pace_buf) (void)hipFree(pace_buf);
      pace_buf = nullptr;
      pace_cap = 0;
      CC_HIP(hipMalloc(&pace_buf, need));
      pace_cap = need;
    }
    CC_HIP(hipMemsetAsync(pace_buf, 0, need, stream()));
    pace = pace_buf;
  }
#endif
  ProfScope prof(prof_class, 2.0 * (double)M * (double)(W.K1 + W.K2) * (double)W.N);
#define CC_LAUNCH_GEMM(KC)                                                                                  \
  hipLaunchKernelGGL(gemm_bias_act_kernel<KC>, dim3((unsigned)nwg), dim3(GEMM_THREADS), 0, stream(), A1,    \
                     lda1, A2, lda2, M, reinterpret_cast<const f32x4 *>(W.d_w), W.d_bias, (int)W.KT1,     \
                     (int)(W.KT1 + W.KT2), (int)W.N, (int)MT, (int)W.NT, C, ldc, act, pace, pace_nch)
  if (prof_class == CC_PROF_GEMM_L1) CC_LAUNCH_GEMM(1);
  else if (prof_class == CC_PROF_GEMM_L2) CC_LAUNCH_GEMM(2);
  else CC_LAUNCH_GEMM(3);
#undef CC_LAUNCH_GEMM
  CC_HIP(hipGetLastError());
  return CC_OK;
}

#ifdef CC_GEMM_TRACE
}  // namespace cc
extern "C" int cc_debug_gemm_trace(unsigned long long *out, int n_wg) {
  CC_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(cc::g_gemm_trace), sizeof(unsigned long long) * 4 * (size_t)n_wg));
  return CC_OK;
}
namespace cc {
#endif

// ---- row softmax (classify encoder's last activation,
//      reference src/bigan_classify.py:39) ------------------------------------
__global__ __launch_bounds__(256) void row_softmax_kernel(float *__restrict__ X, int64_t M, int E,
                                                           int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float *x = X + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < E; c += 64) mx = fmaxf(mx, x[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float s = 0.f;
  for (int c = lane; c < E; c += 64) s += expf(x[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  for (int c = lane; c < E; c += 64) x[c] = expf(x[c] - mx) / s;
}

int launch_row_softmax(float *X, int64_t M, int E, int64_t ld) {
  if (M == 0 || E == 0) return CC_OK;
  ProfScope prof(CC_PROF_SOFTMAX, 4.0 * (double)M * E);
  hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, stream(), X, M, E, ld);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

}  // namespace cc
