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

constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = GEMM_BK;
constexpr int LDS_A_STRIDE = BK + 4;            // floats; (BK+4)·r mod 64 distinct for r mod 16
constexpr int GEMM_THREADS = 256;
constexpr int TILE_W_FLOAT4 = (BK / 4) * BN;    // float4 per packed W tile
constexpr int A_F4_PER_ROW = BK / 4;            // float4 per A-tile row
constexpr int A_ROWS_PER_PASS = GEMM_THREADS / A_F4_PER_ROW;
constexpr int NA = BM / A_ROWS_PER_PASS;        // A float4 staged per thread (128-row tile)
constexpr int NA_HALF = NA / 2;                 // ... of a half-height (64-row) tile
constexpr int NB = TILE_W_FLOAT4 / GEMM_THREADS;  // W float4 staged per thread
#ifndef CC_GEMM_GROUP_M
#define CC_GEMM_GROUP_M 8
#endif
constexpr int GROUP_M = CC_GEMM_GROUP_M;
static_assert(BN == 128 && (BK == 32 || BK == 64), "tile shape");

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// The K loop of one wave (wave tile = MTV x NTV blocks of 32 x 32; only 2 x 2 is instantiated).
// Tried and dropped: edge variants that skip the blocks outside the matrix (N = 3000 leaves the second wave
// column of the last n-tile all padding, 1/48 of layer 1's MFMAs).  A second instantiation of this loop in
// the kernel made the allocator spill (336 B scratch), and a staging-only path for all-padding waves,
// which compiles cleanly, gains nothing: the workgroup still runs at the pace of its three full waves.
template <int KCLASS, int MTV, int NTV, int NAV>
__device__ __forceinline__ void gemm_k_loop(f32x16 (&acc)[2][2], f32x4 (&ra)[NAV], f32x4 (&rb)[NB], const float *(&pa)[NAV],
                                            const f32x4 *&wt, const int (&a_lr)[NAV], const float *__restrict__ A2,
                                            int64_t m0, int64_t lda2, int KT1, int KT, float *As, f32x4 *Bs4,
                                            const float *a_lds, const f32x4 *b_lds, int tid, int a_row, int a_k4) {
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;  // wave-uniform
    if (more) {
      if (kt + 1 == KT1) {  // first tile of segment 2: re-base the A pointers (once per kernel)
        const float *base = A2 + m0 * lda2;
#pragma unroll
        for (int p = 0; p < NAV; ++p) pa[p] = base + a_lr[p] * (int)lda2 + 4 * a_k4;
      }
#pragma unroll
      for (int p = 0; p < NAV; ++p) { ra[p] = *reinterpret_cast<const f32x4 *>(pa[p]); pa[p] += BK; }
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
    __syncthreads();
    if (more) {
      float *aw = As + a_row * LDS_A_STRIDE + 4 * a_k4;
#pragma unroll
      for (int p = 0; p < NAV; ++p) *reinterpret_cast<f32x4 *>(aw + p * A_ROWS_PER_PASS * LDS_A_STRIDE) = ra[p];
#pragma unroll
      for (int p = 0; p < NB; ++p) Bs4[tid + p * GEMM_THREADS] = rb[p];
    }
    __syncthreads();
  }
}

// (Round 4 also gave the half-height tiles a K loop of their own with the global -> register prefetch TWO K-steps ahead
// (two register sets, peeled so that hipcc's counted vmcnt left the younger set in flight; 122 registers, four resident
// per CU instead of five), on the theory that a lone tile of a partial round exposes the memory round trip at every
// K-step: no gain — a 1696-row pass, configs[1], a batch of 32 all within 2 % — and configs[1] lost 6 % to the lower
// occupancy.  A lone half-height tile needs 1 us of MFMA time per K-step (32 x 64 cycles on its one wave per SIMD) and takes
// 3: the rest is the two barriers, the LDS round trip and the fragment reads of a K-step with nobody to overlap them,
// not the global loads.  Dropped.)
// KCLASS only names the instantiation (so profilers tell layer 1 / layer 2 / tail apart).
// HALF: half-height tiles.  Every 128 x 128 tile of the launch is worked by TWO workgroups, each on 64 of its rows
// (4 waves as 2 x 2 of 32 x 64: one row of MFMA blocks per wave instead of two; the W tile and the K loop unchanged),
// which is how a last round of tiles that is at most half full runs on twice the slots (launch_gemm): an output
// element's sum is the same sequence of v_mfma_f32_32x32x2_f32 steps either way, so results are bit-identical.
template <int KCLASS, bool HALF>
__global__
__launch_bounds__(GEMM_THREADS, HALF ? 5 : CC_GEMM_MINWAVES)
void gemm_bias_act_kernel(const float *__restrict__ A1, int64_t lda1, const float *__restrict__ A2,
                          int64_t lda2, int64_t M, const f32x4 *__restrict__ Wp,
                          const float *__restrict__ bias, int KT1, int KT, int N, int MT, int NT,
                          float *__restrict__ C, int64_t ldc, int act, int tile0) {
  constexpr int TM = HALF ? BM / 2 : BM;            // rows of this workgroup
  constexpr int NAV = HALF ? NA_HALF : NA, MTV = HALF ? 1 : 2;
  constexpr int STAGE_FLOATS = TM * LDS_A_STRIDE + TILE_W_FLOAT4 * 4;
  __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS];
  float *As = lds;
  f32x4 *Bs4 = reinterpret_cast<f32x4 *>(lds + TM * LDS_A_STRIDE);

  // ---- workgroup → tile: the launch covers tiles tile0 .. tile0 + nwg - 1 of the grouped order ----------------
  const int nwg = HALF ? (int)(gridDim.x >> 1) : (int)gridDim.x;
  const int bid = HALF ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
  const int wgid = tile0 + (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  const int per_group = GROUP_M * NT;
  const int first_m = (wgid / per_group) * GROUP_M;
  const int gsz = (MT - first_m < GROUP_M) ? MT - first_m : GROUP_M;
  const int m_tile = first_m + (wgid % per_group) % gsz;
  const int n_tile = (wgid % per_group) / gsz;
  const int64_t m0 = (int64_t)m_tile * BM + (HALF ? (int64_t)(blockIdx.x & 1) * TM : 0);
  if (HALF && m0 >= M) return;                      // the lower half of a tile that has <= 64 rows
  const int n0 = n_tile * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- staging: per-thread running pointers into A (re-based once at the
  //      segment switch) and a scalar running pointer into the packed W tiles ----
  const int a_row = tid / A_F4_PER_ROW;  // + A_ROWS_PER_PASS · p
  const int a_k4 = tid % A_F4_PER_ROW;
  int a_lr[NAV];
#pragma unroll
  for (int p = 0; p < NAV; ++p) {
    const int64_t gr = m0 + a_row + A_ROWS_PER_PASS * p;
    a_lr[p] = (int)((gr < M ? gr : M - 1) - m0);  // clamp the M tail onto a valid row
  }
  const float *pa[NAV];
  {
    const float *base = KT1 > 0 ? A1 + m0 * lda1 : A2 + m0 * lda2;
    const int lda = (int)(KT1 > 0 ? lda1 : lda2);
#pragma unroll
    for (int p = 0; p < NAV; ++p) pa[p] = base + a_lr[p] * lda + 4 * a_k4;
  }
  const f32x4 *wt = Wp + (int64_t)n_tile * KT * TILE_W_FLOAT4 + tid;

  // The staging registers must stay in VGPRs (hipcc demotes them to scratch when
  // they are touched through lambdas or selected between arrays) so that the
  // loads stay in flight behind the MFMAs of the current K-step.  Register assignment
  // matters too: with `tid` and one more per-lane value alive across the K loop the
  // allocator moved the accumulators from v[0:15].. to v[2:17].. (and the A/B operands
  // with them) and the same instruction stream ran 11 % slower (137 -> 123 TFLOP/s), so
  // the accumulators are pinned below (any 16-aligned VGPR placement measured the same; AGPRs were
  // slower: the 84/84 register split spills).  The fp16 and kNN kernels did not react to the same pin
  // (tools/pin_sweep.sh).
  f32x4 ra[NAV], rb[NB];

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // pin the four accumulator tuples where the fast build has them (see the note above)
  if (!HALF) {
    asm volatile("" : "+{v[0:15]}"(acc[1][1]));
    asm volatile("" : "+{v[16:31]}"(acc[0][1]));
    asm volatile("" : "+{v[32:47]}"(acc[1][0]));
    asm volatile("" : "+{v[48:63]}"(acc[0][0]));
  } else {
    asm volatile("" : "+{v[0:15]}"(acc[0][1]));
    asm volatile("" : "+{v[16:31]}"(acc[0][0]));
  }

  if (KT > 0) {
#pragma unroll
    for (int p = 0; p < NAV; ++p) { ra[p] = *reinterpret_cast<const f32x4 *>(pa[p]); pa[p] += BK; }
#pragma unroll
    for (int p = 0; p < NB; ++p) rb[p] = wt[p * GEMM_THREADS];
    wt += TILE_W_FLOAT4;
    float *aw = As + a_row * LDS_A_STRIDE + 4 * a_k4;
#pragma unroll
    for (int p = 0; p < NAV; ++p) *reinterpret_cast<f32x4 *>(aw + p * A_ROWS_PER_PASS * LDS_A_STRIDE) = ra[p];
#pragma unroll
    for (int p = 0; p < NB; ++p) Bs4[tid + p * GEMM_THREADS] = rb[p];
  }
  __syncthreads();

  constexpr int WROWS = 32 * MTV;                   // rows of a wave: 64 (two MFMA block rows) or 32
  const float *a_lds = As + (wm * WROWS + r) * LDS_A_STRIDE + 4 * h;
  const f32x4 *b_lds = Bs4 + h * BN + wn * 64 + r;

  gemm_k_loop<KCLASS, MTV, 2, NAV>(acc, ra, rb, pa, wt, a_lr, A2, m0, lda2, KT1, KT, As, Bs4, a_lds, b_lds, tid, a_row, a_k4);

  // ---- epilogue: bias + activation, 128-byte row segments per half-wave ------
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = n0 + wn * 64 + nt * 32 + r;
    const bool col_ok = col < N;
    const float bcol = col_ok ? bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MTV; ++mt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * WROWS + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
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
  DevPtr raw;   // staging copy of the host weights, released on every return path
  if ((K1 + K2) > 0) {
    CC_HIP(raw.alloc(sizeof(float) * (size_t)((K1 + K2) * N)));
    CC_HIP(hipMemcpyAsync(raw.p, w, sizeof(float) * (size_t)((K1 + K2) * N), hipMemcpyHostToDevice, stream()));
  }
  const float *d_raw = raw.as<float>();
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
  return CC_OK;
}

int launch_gemm(const float *A1, int64_t lda1, const float *A2, int64_t lda2, int64_t M,
                const PackedWeight &W, float *C, int64_t ldc, int act, int prof_class, bool allow_split_k) {
  if (M == 0 || W.N == 0) return CC_OK;
  CC_REQUIRE(W.d_bias != nullptr, "gemm: weight not packed");
  CC_REQUIRE(W.KT1 == 0 || (A1 && lda1 >= W.KT1 * BK && lda1 % 4 == 0),
             "gemm: segment 1 needs lda >= %lld (got %lld)", (long long)(W.KT1 * BK), (long long)lda1);
  CC_REQUIRE(W.KT2 == 0 || (A2 && lda2 >= W.KT2 * BK && lda2 % 4 == 0),
             "gemm: segment 2 needs lda >= %lld (got %lld)", (long long)(W.KT2 * BK), (long long)lda2);
  CC_REQUIRE(lda1 * BM < (int64_t)2147483647 && lda2 * BM < (int64_t)2147483647, "gemm: lda too large");
  // one m-tile over a long K (batches of 32 / 128, the reference's call shapes): split-K form, gemm_small.hip
  if (allow_split_k && gemm_small_applies(M, W)) return launch_gemm_small(A1, lda1, A2, lda2, M, W, C, ldc, act, prof_class);
  const int64_t MT = ceil_div(M, BM);
  const int64_t nwg = MT * W.NT;
  CC_REQUIRE(nwg < (int64_t)2147483647, "gemm: grid too large");
  ProfScope prof(prof_class, 2.0 * (double)M * (double)(W.K1 + W.K2) * (double)W.N);
  // Three workgroups per CU: the product runs in rounds of 3 x CUs tiles and its last round costs a whole tile time
  // however few tiles it holds.  A partial last round runs as half-height tiles instead — two workgroups per tile, 64
  // rows each, five resident per CU (90 registers, 25 KiB of LDS) — in a second launch behind the whole rounds (round 4,
  // as the fp16 kernels since round 3).  C3's 8192-row passes are whole rounds (24 / 12 column tiles x 64 = 2 / 1
  // rounds) and never take it; configs[1] (1264 + 632 tiles = 1.65 + 0.82 rounds), a 12.5k-row shard (3.06 + 1.53) and
  // every call of a few rows do.  A row's sums are the same MFMAs in the same order either way: results do not change.
  const int64_t slots = (CC_GEMM_MINWAVES > 0 ? CC_GEMM_MINWAVES : 3) * (int64_t)num_cus();
  const int64_t tail = nwg % slots;
  // (a full round of half-height tiles — 5 x CUs x 2 of them — takes 0.94 of a full-height round: halves pay for any
  // tail below ~0.89 of a round; 0.85 used)
  const int64_t whole = (tail > 0 && 20 * tail <= 17 * slots) ? nwg - tail : nwg;   // tiles of the full-height launch
#define CC_LAUNCH_GEMM(KC, HV, GRID, TILE0)                                                                          \
  hipLaunchKernelGGL((gemm_bias_act_kernel<KC, HV>), dim3((unsigned)(GRID)), dim3(GEMM_THREADS), 0, stream(), A1,    \
                     lda1, A2, lda2, M, reinterpret_cast<const f32x4 *>(W.d_w), W.d_bias, (int)W.KT1,              \
                     (int)(W.KT1 + W.KT2), (int)W.N, (int)MT, (int)W.NT, C, ldc, act, (int)(TILE0))
#define CC_LAUNCH_GEMM_CLASS(HV, GRID, TILE0)                                \
  do {                                                                      \
    if (prof_class == CC_PROF_GEMM_L1) CC_LAUNCH_GEMM(1, HV, GRID, TILE0);  \
    else if (prof_class == CC_PROF_GEMM_L2) CC_LAUNCH_GEMM(2, HV, GRID, TILE0); \
    else CC_LAUNCH_GEMM(3, HV, GRID, TILE0);                                \
  } while (0)
  if (whole > 0) CC_LAUNCH_GEMM_CLASS(false, whole, 0);
  if (whole < nwg) CC_LAUNCH_GEMM_CLASS(true, 2 * (nwg - whole), whole);
#undef CC_LAUNCH_GEMM_CLASS
#undef CC_LAUNCH_GEMM
  CC_HIP(hipGetLastError());
  return CC_OK;
}

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
