// K5 — exact brute-force squared-L2 kNN graph on latent encodings.
//
// The reference has no kNN code: neighbour graphs are built
// inside UMAP/TSNE (reference src/intercepts/plot_intercepts.py:92,176,183).
// Semantics are those of oracle/knn_oracle.c: for query i the k rows j with the
// smallest (dist64(i,j), j), dist64 = sequential float64 sum of squared float64
// differences; self excluded by index unless include_self.
//
// Three stages, bit-exact by construction (stage 1 has two routes, see knn_run):
//  1. select  — approximate distances ‖c‖² + ‖q‖² − 2 q·c for a 32-candidate x
//     32-query tile come straight out of one chain of v_mfma_f32_32x32x2_f32
//     (the norms ride along as two extra K rows), one wave per 32 queries.
//     Points are first translated by the fp32 centroid μ (distances are translation
//     invariant; ‖x−μ‖² ≪ ‖x‖² for clustered encodings, which is what makes the fp32
//     error bound ε small enough to certify).
//     Candidates are pre-packed so each lane's operands for 4 MFMA steps are one
//     coalesced 16-byte load; queries stay in registers for the whole scan.
//     Per query a running threshold τ (its K'-th best so far, K' = k + slack)
//     filters the accumulator registers; survivors are appended to a per-query
//     LDS list with a wavefront-local atomic, and a list that could overflow is
//     compacted by the wave (rank-by-counting on 64-bit (dist,idx) keys).
//     Inputs beyond a few hundred candidates take route 1b (the product route): a SAMPLE pass over every 3rd-8th
//     candidate block yields a per-query threshold T under which at least m >= k candidates lie (per-slot minima of
//     a coarse fp16 tile + direct fp32 re-evaluation of the m best: knn_sample16_kernel / knn_slot_threshold_kernel);
//     the COLLECT pass then streams ALL candidates through one fp16 product per pair with a rigorous error bound,
//     re-evaluates the few pairs at or below T + bound directly in fp32 and appends those <= T to a per-query
//     buffer (knn_collect16c_kernel) — no lists, no compaction, freely split over candidate ranges.  The fp32 tiles
//     described above remain for d > 64, for data whose scale does not fit fp16, and as cross-checks behind options.
//  2. rerank  — the K' survivors are re-scored in float64 exactly as the oracle
//     does and ordered by (dist64, idx).  The result is CERTIFIED when the k-th
//     exact distance is below τ_approx(K') − ε, ε a rigorous bound on the fp32
//     error of stage 1: every point outside the list is then provably farther.
//  3. fallback — uncertified queries (dense duplicates / near ties) are rescanned
//     against all candidates in float64, bounded by their stage-2 k-th key.
#include "common.h"
#include <sys/mman.h>

namespace cc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;  // native vector: arrays of it stay in VGPRs
using u64 = unsigned long long;

#ifndef CC_KNN_SELECT16
#define CC_KNN_SELECT16 1
#endif

__device__ __forceinline__ unsigned f32_to_ordered(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(unsigned o) {
  unsigned b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(b);
}

// ---- centroid (fixed summation order: deterministic) ------------------------------
constexpr int MEAN_BLOCKS = 2048;   // fixed partition: the centroid does not depend on the launch shape
__global__ void knn_colsum_kernel(const float *__restrict__ x, int64_t N, int d, double *__restrict__ partial) {
  const int c = threadIdx.x;
  if (c >= d) return;
  const int64_t per = (N + MEAN_BLOCKS - 1) / MEAN_BLOCKS;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < N ? lo + per : N;
  double s = 0.0;
  int64_t i = lo;
  for (; i + 4 <= hi; i += 4) {   // four independent loads in flight, summed in row order
    const float v0 = x[i * d + c], v1 = x[(i + 1) * d + c], v2 = x[(i + 2) * d + c], v3 = x[(i + 3) * d + c];
    s += (double)v0;
    s += (double)v1;
    s += (double)v2;
    s += (double)v3;
  }
  for (; i < hi; ++i) s += (double)x[i * d + c];
  partial[(int64_t)blockIdx.x * d + c] = s;
}
// One workgroup: MEAN_PARTS groups of threads each add a fixed quarter of the partial sums of their column (four loads in
// flight), the first group adds the four in order.  (Round 4: one dependent chain of 2048 loads per column took 76 µs,
// 2.5 % of a 100k-query search.)  The order of the sum is fixed by the constants, not by the launch.
constexpr int MEAN_PARTS = 4;   // x round_up(d, 64) <= 192 threads: one workgroup
static_assert(MEAN_BLOCKS % (MEAN_PARTS * 4) == 0, "whole groups of four per part");
__global__ void knn_mean_kernel(const double *__restrict__ partial, int64_t N, int d, float *__restrict__ mu) {
  __shared__ double part_sum[MEAN_PARTS][132];
  const int c = threadIdx.x, part = threadIdx.y;
  if (c < d) {
    constexpr int PER = MEAN_BLOCKS / MEAN_PARTS;
    const double *p = partial + (int64_t)part * PER * d + c;
    double s = 0.0;
    for (int b = 0; b < PER; b += 4) {
      const double v0 = p[(int64_t)b * d], v1 = p[(int64_t)(b + 1) * d], v2 = p[(int64_t)(b + 2) * d], v3 = p[(int64_t)(b + 3) * d];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    part_sum[part][c] = s;
  }
  __syncthreads();
  if (part == 0 && c < d) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < MEAN_PARTS; ++q) s += part_sum[q][c];
    mu[c] = (float)(s / (double)N);
  }
}

// the one expression used everywhere for a centred coordinate (so duplicate rows stay duplicates)
__device__ __forceinline__ float centred(float x, float mu) { return __fsub_rn(x, mu); }

// ---- norms of the centred points -----------------------------------------------------
// One row per thread, float64 accumulation in coordinate order (what every later stage assumes of norms[]).  Round 4,
// last session: the kernel took 0.93 ms for 1M x 64 (15x its HBM time, 4 % of a rank's search of configs[3]) — one
// 4-byte load per coordinate and lane, and two same-address atomics per WAVE (31 000 of them at 1M rows).  Now 16-byte
// loads where the rows allow them, a grid-stride loop over at most NORM_BLOCKS blocks, and one pair of atomics per BLOCK.
constexpr int NORM_BLOCKS = 2048;
__global__ __launch_bounds__(256) void knn_norm_kernel(const float *__restrict__ x, const float *__restrict__ mu, int64_t N, int d,
                                                       float *__restrict__ norms, unsigned *__restrict__ maxnorm_bits) {
  __shared__ unsigned red[2][4];
  const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(mu) & 15) == 0;
  unsigned fb = 0u, ab = 0u;   // non-negative floats order as uints (a NaN's bits win: the scale check rejects it)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    float amax = 0.f;
    const float *row = x + i * d;
    if (vec) {
      for (int t = 0; t < d; t += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(row + t), m = *reinterpret_cast<const float4 *>(mu + t);
        float y = centred(v.x, m.x); amax = fmaxf(amax, fabsf(y)); s += (double)y * (double)y;
        y = centred(v.y, m.y); amax = fmaxf(amax, fabsf(y)); s += (double)y * (double)y;
        y = centred(v.z, m.z); amax = fmaxf(amax, fabsf(y)); s += (double)y * (double)y;
        y = centred(v.w, m.w); amax = fmaxf(amax, fabsf(y)); s += (double)y * (double)y;
      }
    } else {
      for (int t = 0; t < d; ++t) {
        const float y = centred(row[t], mu[t]);
        amax = fmaxf(amax, fabsf(y));
        s += (double)y * (double)y;
      }
    }
    const float sf = (float)s;
    norms[i] = sf;
    const unsigned f1 = __float_as_uint(sf), a1 = __float_as_uint(amax);
    fb = f1 > fb ? f1 : fb;
    ab = a1 > ab ? a1 : ab;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned f2 = __shfl_xor(fb, o), a2 = __shfl_xor(ab, o);
    fb = f2 > fb ? f2 : fb;
    ab = a2 > ab ? a2 : ab;
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = fb; red[1][threadIdx.x >> 6] = ab; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      fb = red[0][w] > fb ? red[0][w] : fb;
      ab = red[1][w] > ab ? red[1][w] : ab;
    }
    atomicMax(maxnorm_bits, fb);
    atomicMax(maxnorm_bits + 1, ab);      // largest |centred coordinate| (scale of the fp16 panels)
  }
}

// ---- candidate packing ---------------------------------------------------------
// P[((b·NG + g)·64 + lane)·4 + u] = Aaug[32b + (lane&31)][8g + 4(lane>>5) + u]
// Aaug[c] = [y_c[0..d), ‖y_c‖², 1, 0 …], y = x − μ; rows beyond N: [0…, +inf, 1, 0 …].
__global__ void knn_pack_kernel(const float *__restrict__ x, const float *__restrict__ mu,
                                const float *__restrict__ norms, int64_t N, int d, int NG, int64_t NB,
                                float4 *__restrict__ P) {
  const int64_t total = NB * NG * 64;
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= total) return;
  const int lane = f & 63;
  const int g = (int)((f >> 6) % NG);
  const int64_t b = (f >> 6) / NG;
  const int64_t cand = b * 32 + (lane & 31);
  const int kbase = 8 * g + 4 * (lane >> 5);
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = kbase + u;
    float val = 0.f;
    if (cand < N) {
      if (k < d) val = centred(x[cand * d + k], mu[k]);
      else if (k == d) val = norms[cand];
      else if (k == d + 1) val = 1.f;
    } else {
      if (k == d) val = INFINITY;
      else if (k == d + 1) val = 1.f;
    }
    v[u] = val;
  }
  P[f] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- stage 1: select -------------------------------------------------------------
constexpr int SEL_MAXJ = 4;  // list capacity <= 256 entries

// Keep the KP smallest keys of list L[0..c) (sorted ascending) — rank by counting.  The
// list is spread over the wave's registers and every key is broadcast with
// v_readlane (scalar operand) instead of being re-read from LDS.
__device__ __forceinline__ void compact_list(u64 *L, int *cnt_p, float *tau_p, int KP, int lane) {
  const int c = *cnt_p;
  u64 e[SEL_MAXJ];
  int rk[SEL_MAXJ];
#pragma unroll
  for (int j = 0; j < SEL_MAXJ; ++j) {
    const int idx = lane + 64 * j;
    e[j] = (idx < c) ? L[idx] : ~0ull;
    rk[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < SEL_MAXJ; ++j) {
    int cj = c - 64 * j;
    if (cj > 0) {  // wave-uniform
      cj = cj < 64 ? cj : 64;
      const int lo = (int)(unsigned)(e[j] & 0xffffffffull), hi = (int)(unsigned)(e[j] >> 32);
      for (int i = 0; i < cj; ++i) {
        const u64 v = ((u64)(unsigned)__builtin_amdgcn_readlane(hi, i) << 32) |
                      (u64)(unsigned)__builtin_amdgcn_readlane(lo, i);
#pragma unroll
        for (int jj = 0; jj < SEL_MAXJ; ++jj)
          if (64 * jj < c) rk[jj] += (v < e[jj]) ? 1 : 0;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < SEL_MAXJ; ++j) {
    const int idx = lane + 64 * j;
    if (idx < c && rk[j] < KP) {
      L[rk[j]] = e[j];
      if (rk[j] == KP - 1) *tau_p = ordered_to_f32((unsigned)(e[j] >> 32));
    }
  }
  if (lane == 0) *cnt_p = c < KP ? c : KP;
  __builtin_amdgcn_wave_barrier();
}

// Append the passing accumulator registers [E0, E1) of this lane to its query's list.
template <int E0, int E1>
__device__ __forceinline__ int append_regs(const f32x16 &acc, float tau, u64 *L, int pos, unsigned cbase) {
#pragma unroll
  for (int e = E0; e < E1; ++e) {
    const bool pass = acc[e] <= tau;
    if (__ballot(pass)) {  // wave-uniform skip of registers nobody appends from
      if (pass) {
        L[pos] = ((u64)f32_to_ordered(acc[e]) << 32) | (u64)(cbase + (e & 3) + 8 * (e >> 2));
        ++pos;
      }
    }
  }
  return pos;
}
template <int E0, int E1>
__device__ __forceinline__ int count_regs(const f32x16 &acc, float tau) {
  int c = 0;
#pragma unroll
  for (int e = E0; e < E1; ++e) c += (acc[e] <= tau) ? 1 : 0;
  return c;
}

__device__ __forceinline__ void select_filter(f32x16 &acc, int64_t b, int64_t NB, int64_t self_blk, int self_e,
                                              unsigned tail_vm, float &tau, u64 *list, float *tau_s, int *cnt,
                                              int LS, int KP, int CAP, int lane);

// One wave owns 32 queries (MFMA columns) and their LDS lists exclusively, so appends
// need no atomics: lanes r and r+32 (the two row-halves of query r) split the slots
// [cnt, cnt + c0 + c1) by exchanging their pass counts.  Lists hold CAP = K' + 16 keys
// (12.5 KiB per wave at K' = 32 → 12 waves per CU, which is what covers the VALU
// filter/compaction phases of one wave with the MFMAs of the others).  A step that
// would overflow a list is replayed in two halves of 8 registers (≤ 16 appends per
// query each), compacting to K' in front of each half.
template <int NG>
__device__ __forceinline__ void select_step(const float4 (&a)[NG], const float4 (&bq)[NG], int64_t b, int64_t NB,
                                            int64_t self_blk, int self_e, unsigned tail_vm, float &tau, u64 *list,
                                            float *tau_s, int *cnt, int LS, int KP, int CAP, int lane) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, bq[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, bq[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, bq[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, bq[g].w, acc, 0, 0, 0);
  }
  select_filter(acc, b, NB, self_blk, self_e, tail_vm, tau, list, tau_s, cnt, LS, KP, CAP, lane);
}

// The list side of a select step: filter the 32 x 32 tile `acc` against the running thresholds and
// append the survivors (shared by the fp32 and the fp16-split tiles; any monotone scaling of the
// distances works as long as tau and the keys live in the same scale).
__device__ __forceinline__ void select_filter(f32x16 &acc, int64_t b, int64_t NB, int64_t self_blk, int self_e,
                                              unsigned tail_vm, float &tau, u64 *list, float *tau_s, int *cnt,
                                              int LS, int KP, int CAP, int lane) {
  const int r = lane & 31, h = lane >> 5;
  // inadmissible candidates (the query itself, rows beyond N) become NaN: NaN <= tau is false
  if (__any(b == self_blk) || b == NB - 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const bool kill = (b == self_blk && e == self_e) || (b == NB - 1 && !((tail_vm >> e) & 1u));
      acc[e] = kill ? __builtin_nanf("") : acc[e];
    }
  }
  int c_l = count_regs<0, 16>(acc, tau);
  if (!__any(c_l)) return;
  u64 *L = list + r * LS;
  const unsigned cbase = (unsigned)(b * 32) + 4 * h;
  int c_o = __shfl_xor(c_l, 32);
  int base = cnt[r];
  if (!__any(base + c_l + c_o > CAP)) {
    append_regs<0, 16>(acc, tau, L, base + (h ? c_o : 0), cbase);
    __builtin_amdgcn_wave_barrier();
    if (h == 0) cnt[r] = base + c_l + c_o;
    __builtin_amdgcn_wave_barrier();
    return;
  }
  // careful path: two halves of 8 registers, room for 16 appends made in front of each
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u64 need = __ballot(lane < 32 && cnt[r] > CAP - 16);
    if (need) {
      while (need) {
        const int qi = __builtin_ctzll(need);
        need &= need - 1;
        compact_list(list + qi * LS, cnt + qi, tau_s + qi, KP, lane);
      }
      tau = tau_s[r];
    }
    c_l = half == 0 ? count_regs<0, 8>(acc, tau) : count_regs<8, 16>(acc, tau);
    c_o = __shfl_xor(c_l, 32);
    base = cnt[r];
    if (half == 0) append_regs<0, 8>(acc, tau, L, base + (h ? c_o : 0), cbase);
    else append_regs<8, 16>(acc, tau, L, base + (h ? c_o : 0), cbase);
    __builtin_amdgcn_wave_barrier();
    if (h == 0) cnt[r] = base + c_l + c_o;
    __builtin_amdgcn_wave_barrier();
  }
}

#ifndef CC_KNN_SEL_MINW
#define CC_KNN_SEL_MINW 3
#endif
template <int NG>
__global__ __launch_bounds__(64, CC_KNN_SEL_MINW) void knn_select_kernel(
    const float4 *__restrict__ P, const float *__restrict__ norms, const float *__restrict__ x,
    const float *__restrict__ mu, int64_t N, int d, int64_t NB, int64_t q_begin, int64_t n_q, int include_self,
    int KP, int CAP, int S, int64_t NBs, int bstride, u64 *__restrict__ out_keys, int *__restrict__ out_cnt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LS = CAP + 1;                               // odd stride: conflict-free 8-byte appends
  u64 *list = reinterpret_cast<u64 *>(smem);            // [32][LS]
  float *tau_s = reinterpret_cast<float *>(list + 32 * LS);  // [32]
  int *cnt = reinterpret_cast<int *>(tau_s + 32);       // [32]

  // workgroup = (query tile, candidate split); split = blockIdx % S, so with S = 8 every XCD
  // (round-robin dispatch) streams one eighth of the candidate panels out of its own L2
  const int64_t tile = blockIdx.x / S;
  const int split = blockIdx.x % S;
  // this wave visits blocks b_lo, b_lo + bstride, ... < b_hi (bstride > 1: a sample of the blocks)
  const int64_t b_lo = split * NBs;
  const int64_t b_hi = (b_lo + NBs < NB) ? b_lo + NBs : NB;

  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const int64_t ql = tile * 32 + r;
  const bool qvalid = ql < n_q;
  const int64_t q = q_begin + ql;

  float4 bq[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 8 * g + 4 * h + u;
      float val = 0.f;
      if (qvalid) {
        if (k < d) val = -2.0f * centred(x[q * d + k], mu[k]);
        else if (k == d) val = 1.f;
        else if (k == d + 1) val = norms[q];
      }
      v[u] = val;
    }
    bq[g] = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (lane < 32) {
    cnt[r] = 0;
    tau_s[r] = qvalid ? INFINITY : -INFINITY;
  }
  __builtin_amdgcn_wave_barrier();
  float tau = tau_s[r];

  // accumulator register e of this lane is candidate row crow(e) = (e&3) + 8(e>>2) + 4h of the block
  int64_t self_blk = -1;
  int self_e = -1;
  if (qvalid && !include_self) {
    const int row = (int)(q & 31);
    if (((row >> 2) & 1) == h) {
      self_blk = q >> 5;
      self_e = (row & 3) | ((row >> 3) << 2);
    }
  }
  unsigned tail_vm = 0xffffu;  // bit e clear = row beyond N in the last block
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if ((NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= N) tail_vm &= ~(1u << e);

  // two register sets of candidate operands, ping-pong (no copies): block b+1 loads while b computes
  float4 a0[NG], a1[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) a0[g] = P[(b_lo * NG + g) * 64 + lane];
  for (int64_t b = b_lo; b < b_hi; b += 2 * bstride) {
    const int64_t b1 = b + bstride, b2 = b + 2 * bstride;
    {
      const int64_t bn = (b1 < b_hi) ? b1 : b;
#pragma unroll
      for (int g = 0; g < NG; ++g) a1[g] = P[(bn * NG + g) * 64 + lane];
    }
    select_step<NG>(a0, bq, b, NB, self_blk, self_e, tail_vm, tau, list, tau_s, cnt, LS, KP, CAP, lane);
    if (b1 < b_hi) {
      {
        const int64_t bn = (b2 < b_hi) ? b2 : b1;
#pragma unroll
        for (int g = 0; g < NG; ++g) a0[g] = P[(bn * NG + g) * 64 + lane];
      }
      select_step<NG>(a1, bq, b1, NB, self_blk, self_e, tail_vm, tau, list, tau_s, cnt, LS, KP, CAP, lane);
    }
  }

  // final: sort every list, emit the best KP keys per query
  for (int qi = 0; qi < 32; ++qi) {
    compact_list(list + qi * LS, cnt + qi, tau_s + qi, KP, lane);
    const int64_t qo = tile * 32 + qi;
    if (qo < n_q) {
      const int c = cnt[qi];
      const int64_t slot = qo * S + split;
      for (int i = lane; i < c; i += 64) out_keys[slot * KP + i] = list[qi * LS + i];
      if (lane == 0) out_cnt[slot] = c;
    }
  }
}

// ---- stage 1b: collect everything below a fixed per-query threshold -----------------------
// Same MFMA tile as select, but the threshold T[q] is fixed (from the sampled select), so there
// is no list state.  Passing (approx, idx) keys are parked in a small per-wave LDS buffer
// (positions from a wave prefix sum, no atomics) and flushed to buf[q][0..cap) through the
// global per-query cursors 64 entries at a time — one atomic round trip per flush instead of
// one per appending step.
constexpr int COL_WB = 128;  // per-wave parking buffer entries

struct WaveBuf {
  u64 *key;    // [COL_WB]
  int *qr;     // [COL_WB] query column 0..31
  int n;       // entries parked (wave-uniform)
};

__device__ __forceinline__ void wavebuf_flush(WaveBuf &wb, u64 *__restrict__ buf, int *__restrict__ cursor,
                                              int64_t q0, int64_t n_q, int cap, int lane) {
  for (int i = lane; i < wb.n; i += 64) {
    const int64_t ql = q0 + wb.qr[i];
    if (ql < n_q) {
      const int pos = atomicAdd(cursor + ql, 1);
      if (pos < cap) buf[ql * cap + pos] = wb.key[i];
    }
  }
  __builtin_amdgcn_wave_barrier();
  wb.n = 0;
}

// Park the accumulator entries at or below T: per register one compare whose lane mask is the ballot
// (registers nobody passes in cost a compare and a scalar branch); passing lanes take consecutive slots by
// mbcnt.  ~1.3 of a tile's 1024 pairs pass, so this replaces a 64-lane prefix scan per tile.
__device__ __forceinline__ void park_passes(const f32x16 &acc, float T, float key_scale, unsigned cbase, int r,
                                            WaveBuf &wb, u64 *__restrict__ buf, int *__restrict__ cursor, int64_t q0,
                                            int64_t n_q, int cap, int lane) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const bool pass = acc[e] <= T;
    const unsigned long long mask = __ballot(pass);
    if (mask) {   // wave-uniform
      const int cnt = __builtin_popcountll(mask);
      if (wb.n + cnt > COL_WB) wavebuf_flush(wb, buf, cursor, q0, n_q, cap, lane);
      if (pass) {
        const int pos = wb.n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        wb.key[pos] = ((u64)f32_to_ordered(acc[e] * key_scale) << 32) | (u64)(cbase + (e & 3) + 8 * (e >> 2));
        wb.qr[pos] = r;
      }
      wb.n += cnt;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int NG>
__device__ __forceinline__ void collect_step(const float4 (&a)[NG], const float4 (&bq)[NG], int64_t b, int64_t NB,
                                             int64_t self_blk, int self_e, unsigned tail_vm, float T, WaveBuf &wb,
                                             u64 *__restrict__ buf, int *__restrict__ cursor, int64_t q0,
                                             int64_t n_q, int cap, int lane) {
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, bq[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, bq[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, bq[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, bq[g].w, acc, 0, 0, 0);
  }
  if (__any(b == self_blk) || b == NB - 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const bool kill = (b == self_blk && e == self_e) || (b == NB - 1 && !((tail_vm >> e) & 1u));
      acc[e] = kill ? __builtin_nanf("") : acc[e];
    }
  }
  park_passes(acc, T, 1.0f, (unsigned)(b * 32) + 4 * h, r, wb, buf, cursor, q0, n_q, cap, lane);
}

// Workgroup = 4 waves = 128 queries x one candidate split.  Candidate panels are staged through
// LDS (4 blocks = 128 candidates per stage, double buffered, one barrier per stage) so that one
// global read of a panel feeds all four waves: at one wave per panel read the kernel needs
// 16 B/clk/CU of panel traffic and is fabric-bound once the candidate set outgrows the L2s.
#ifndef CC_KNN_COL_CB
#define CC_KNN_COL_CB 2
#endif
#ifndef CC_KNN_COL_MINW
#define CC_KNN_COL_MINW 2
#endif
constexpr int COL_CB = CC_KNN_COL_CB;  // candidate blocks per LDS stage

template <int NG>
__global__ __launch_bounds__(256, CC_KNN_COL_MINW) void knn_collect_kernel(
    const float4 *__restrict__ P, const float *__restrict__ norms, const float *__restrict__ x,
    const float *__restrict__ mu, int64_t N, int d, int64_t NB, int64_t q_begin, int64_t n_q, int include_self,
    const float *__restrict__ T, int S, int64_t NBs, int cap, u64 *__restrict__ buf, int *__restrict__ cursor) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *stage = reinterpret_cast<float4 *>(smem);   // [2][COL_CB·NG·64], then the 4 parking buffers
  constexpr int STAGE_F4 = COL_CB * NG * 64;

  const int64_t tile = blockIdx.x / S;
  const int split = blockIdx.x % S;
  const int64_t b_lo = split * NBs;
  const int64_t b_hi = (b_lo + NBs < NB) ? b_lo + NBs : NB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t ql = tile * 128 + wave * 32 + r;
  const bool qvalid = ql < n_q;
  const int64_t q = q_begin + ql;

  float4 bq[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 8 * g + 4 * h + u;
      float val = 0.f;
      if (qvalid) {
        if (k < d) val = -2.0f * centred(x[q * d + k], mu[k]);
        else if (k == d) val = 1.f;
        else if (k == d + 1) val = norms[q];
      }
      v[u] = val;
    }
    bq[g] = make_float4(v[0], v[1], v[2], v[3]);
  }
  const float Tq = qvalid ? T[ql] : -INFINITY;
  const int64_t q0 = tile * 128 + wave * 32;   // first query of this wave
  WaveBuf wb;
  wb.key = reinterpret_cast<u64 *>(smem + sizeof(float4) * 2 * STAGE_F4) + wave * COL_WB;
  wb.qr = reinterpret_cast<int *>(smem + sizeof(float4) * 2 * STAGE_F4 + sizeof(u64) * 4 * COL_WB) + wave * COL_WB;
  wb.n = 0;

  int64_t self_blk = -1;
  int self_e = -1;
  if (qvalid && !include_self) {
    const int row = (int)(q & 31);
    if (((row >> 2) & 1) == h) {
      self_blk = q >> 5;
      self_e = (row & 3) | ((row >> 3) << 2);
    }
  }
  unsigned tail_vm = 0xffffu;
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if ((NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= N) tail_vm &= ~(1u << e);

  // stage s covers blocks b_lo + s·COL_CB .. +COL_CB-1 (clamped to the last block of the split:
  // the clamped copies are loaded but never used)
  const int64_t nstages = (b_hi - b_lo + COL_CB - 1) / COL_CB;
  const f32x4 *Pv = reinterpret_cast<const f32x4 *>(P);
  f32x4 *stage_v = reinterpret_cast<f32x4 *>(smem);
  // staged float4 number f = tid + 256·i of a stage lies in block cb = f / (NG·64) of that stage
  constexpr int NS = (STAGE_F4 + 255) / 256;   // float4 staged per thread
  int s_cb[NS], s_off[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    int f = i * 256 + tid;
    f = f < STAGE_F4 ? f : STAGE_F4 - 1;       // surplus threads of the last round duplicate the last float4
    s_cb[i] = f / (NG * 64);
    s_off[i] = f - s_cb[i] * (NG * 64);
  }
  f32x4 rs[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    int64_t b = b_lo + s_cb[i];
    b = b < b_hi ? b : b_hi - 1;
    rs[i] = Pv[b * (NG * 64) + s_off[i]];
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) stage_v[s_cb[i] * (NG * 64) + s_off[i]] = rs[i];
  __syncthreads();

  int cur = 0;
  for (int64_t st = 0; st < nstages; ++st) {
    const bool more = st + 1 < nstages;
    if (more) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        int64_t b = b_lo + (st + 1) * COL_CB + s_cb[i];
        b = b < b_hi ? b : b_hi - 1;
        rs[i] = Pv[b * (NG * 64) + s_off[i]];
      }
    }
    const float4 *sbuf = stage + cur * STAGE_F4;
#pragma unroll
    for (int cb = 0; cb < COL_CB; ++cb) {
      const int64_t b = b_lo + st * COL_CB + cb;
      if (b < b_hi) {
        float4 a[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) a[g] = sbuf[(cb * NG + g) * 64 + lane];
        collect_step<NG>(a, bq, b, NB, self_blk, self_e, tail_vm, Tq, wb, buf, cursor, q0, n_q, cap, lane);
      }
    }
    if (more) {
      f32x4 *dbuf = stage_v + (cur ^ 1) * STAGE_F4;  // last read in stage st-1: every wave is past that barrier
#pragma unroll
      for (int i = 0; i < NS; ++i) dbuf[s_cb[i] * (NG * 64) + s_off[i]] = rs[i];
    }
    __syncthreads();
    cur ^= 1;
  }
  wavebuf_flush(wb, buf, cursor, q0, n_q, cap, lane);
}

// ---- stage 1b on the fp16 matrix cores ------------------------------------------------------------
// Both passes of stage 1b only need approximate distances with a KNOWN error bound (the fine re-check and the
// certificate of stage 2 absorb it), so they run on v_mfma_f32_32x32x16_f16 at 16x the fp32 MFMA rate on the centred
// coordinates rounded to fp16, y·s -> fp16 (s a power of two putting max|y·s| in [2^6, 2^7)):  −2 q·c ≈ qh·ch
// (ND MFMAs, ND = ceil(d/16)); the two norms ride in one extra K-step as three fp16 pieces each (33 bits).
// 5 MFMAs x 32 cycles per 32 x 32 tile at d = 64 instead of 36 x 64.  (Rounds 1-2 also kept a three-product form,
// qh·ch + qh·cl + ql·ch, 13 MFMAs, ~2^-22 accurate, first as the collect and then as a cross-check of the coarse
// one; round 3 retired it: the fp32 list kernels are the one cross-check route.)
using h8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ void split3(float v, _Float16 *a, _Float16 *b, _Float16 *c) {
  *a = (_Float16)v;
  const float r1 = v - (float)*a;
  *b = (_Float16)r1;
  *c = (_Float16)(r1 - (float)*b);
}

// query-side norm unit of lane half h for the value v (already divided by 256): {256, 256, 256, v_a, v_b, v_c, 0, 0}
__device__ __forceinline__ h8 query_norm_unit(float v, int h) {
  h8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.f;
  if (h == 0) {
    _Float16 a, b, c;
    split3(v, &a, &b, &c);
    const _Float16 k256 = (_Float16)256.f;
    o[0] = k256; o[1] = k256; o[2] = k256; o[3] = a; o[4] = b; o[5] = c;
  }
  return o;
}

// one 16-byte unit (8 halfs) of operand data for row `i` of x, K-step t, lane half h:
//   t < ND          hi parts of (fac · y[16t + 8h + j]),  j = 0..7
//   t == ND         norm step: h == 0 -> {n_a, n_b, n_c, 256, 256, 256, 0, 0} (candidate side) or
//                                        {256, 256, 256, n_a, n_b, n_c, 0, 0} (query side); h == 1 -> 0
__device__ __forceinline__ h8 panel16_unit(const float *__restrict__ x, const float *__restrict__ mu,
                                           const float *__restrict__ norms, int64_t i, bool valid, int d, int ND, int t,
                                           int h, float fac, float s2_over_256, bool query_side) {
  h8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.f;
  if (!valid) return o;
  if (t == ND) {
    if (h == 0) {
      _Float16 a, b, c;
      split3(norms[i] * s2_over_256, &a, &b, &c);
      const _Float16 k256 = (_Float16)256.f;
      if (query_side) { o[0] = k256; o[1] = k256; o[2] = k256; o[3] = a; o[4] = b; o[5] = c; }
      else { o[0] = a; o[1] = b; o[2] = c; o[3] = k256; o[4] = k256; o[5] = k256; }
    }
    return o;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * t + 8 * h + j;
    if (k < d) o[j] = (_Float16)(fac * centred(x[i * d + k], mu[k]));
  }
  return o;
}

// P16[((b·(ND+1) + t)·64 + lane)] (16-byte units): candidate 32b + (lane&31), lane half = lane>>5
// ncs[i] = s²·‖c_i − μ‖² for i < N, 0 behind (n_ncs floats): the candidates' norms as the accumulator-form collect reads them
__global__ void knn_pack16_kernel(const float *__restrict__ x, const float *__restrict__ mu,
                                  const float *__restrict__ norms, int64_t N, int d, int ND, int64_t NB, float s,
                                  h8 *__restrict__ P16, float *__restrict__ ncs, int64_t n_ncs) {
  const int NT = ND + 1;
  const int64_t total = NB * NT * 64;
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n_ncs) ncs[f] = f < N ? norms[f] * (s * s) : 0.f;
  if (f >= total) return;
  const int lane = f & 63;
  const int t = (int)((f >> 6) % NT);
  const int64_t b = (f >> 6) / NT;
  const int64_t cand = b * 32 + (lane & 31);
  P16[f] = panel16_unit(x, mu, norms, cand, cand < N, d, ND, t, lane >> 5, s, s * s * (1.0f / 256.0f), false);
}

// direct fp32 evaluation of one squared distance from the original coordinates (error ~d·2^-24 of the distance itself)
__device__ __forceinline__ float fine_dist(const float *__restrict__ xq, const float *__restrict__ xc, int d, int vec) {
  float sum = 0.f;
  if (vec) {
    for (int t = 0; t < d; t += 4) {
      const float4 a = *reinterpret_cast<const float4 *>(xq + t), b = *reinterpret_cast<const float4 *>(xc + t);
      float df = a.x - b.x; sum = fmaf(df, df, sum);
      df = a.y - b.y; sum = fmaf(df, df, sum);
      df = a.z - b.z; sum = fmaf(df, df, sum);
      df = a.w - b.w; sum = fmaf(df, df, sum);
    }
  } else {
    for (int t = 0; t < d; ++t) {
      const float df = xq[t] - xc[t];
      sum = fmaf(df, df, sum);
    }
  }
  return sum;
}

// the same sum in the same order as fine_dist (bit-identical result), with the loads of 32 coordinates in flight at
// once instead of 4: for callers that evaluate few pairs and wait on each
__device__ __forceinline__ float fine_dist_batched(const float *__restrict__ xq, const float *__restrict__ xc, int d, int vec) {
  if (!vec) return fine_dist(xq, xc, d, 0);
  float sum = 0.f;
  for (int t0 = 0; t0 < d; t0 += 32) {
    float4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + 4 * i < d ? t0 + 4 * i : 0;   // past the end: any valid address, not used
      a[i] = *reinterpret_cast<const float4 *>(xq + t);
      b[i] = *reinterpret_cast<const float4 *>(xc + t);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (t0 + 4 * i < d) {
        float df = a[i].x - b[i].x; sum = fmaf(df, df, sum);
        df = a[i].y - b[i].y; sum = fmaf(df, df, sum);
        df = a[i].z - b[i].z; sum = fmaf(df, df, sum);
        df = a[i].w - b[i].w; sum = fmaf(df, df, sum);
      }
    }
  }
  return sum;
}

// ---- sample pass on the fp16 matrix cores: slot minima instead of lists -------------------------
// The sample only has to produce a threshold T[q] under which at least m candidates lie.  A 32 x 32 tile leaves
// each lane with 16 distances of ITS query to 16 candidates; the lane keeps the running minimum of every such
// register slot over the sampled tiles, NSET interleaved sets of them (tile j feeds set j % NSET).  A query thus
// ends with G = 32·NSET minima (x the candidate ranges the pass is split over), each belonging to a DIFFERENT
// candidate.  The tile is the coarse one of the collect (hi·hi product + norm step: ND + 1 MFMAs), made positive by
// adding the coarse bound through the query's norm slot, and the low SKEY_BITS of each value are replaced by the
// index of the sampled tile, so that one v_and_or + one v_min_u32 per distance keep the minimum AND where it came
// from.  knn_slot_threshold_kernel then takes the m smallest keys of a query, re-evaluates those m candidates
// directly in fp32 (the arithmetic of the collect's final test) and returns the largest of the m distances: m
// distinct candidates lie at or below T whatever the coarse values were worth.  T is a little looser than the m-th
// best of the whole sample (two of the best m may share a slot: the expected rank is G·ln(G/(G-m)), 18.4 instead of
// 16 at G = 64) but the pass costs 5 MFMAs + 32 vector instructions per tile where round 1's list select cost
// 13 + 511, and is cheap enough to sample every 4th block instead of every 8th.
constexpr int SKEY_BITS = 13;                       // sampled tiles per candidate range <= 8192
constexpr unsigned SKEY_MASK = (1u << SKEY_BITS) - 1u;

#define CC_KNN_GLDS16(src, dst)                                                             \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src),   \
                                   (void __attribute__((address_space(3))) *)(dst), 16, 0, 0)

constexpr int SMP_WAVES = 4;   // waves of a sample workgroup = candidate tiles per round of its LDS ring

// One sampled tile for one wave: ND + 1 fragment reads from the stage, (ND + 1)·QW MFMAs, then v_and_or_b32 +
// v_min_u32 per distance.  MASKED = the tile may hold a query of this wave itself or the padding behind candidate N - 1.
template <int ND, int QW, bool MASKED>
__device__ __forceinline__ void sample16_tile(const h8 *__restrict__ frag, const h8 (&bqh)[QW][ND + 1],
                                              __attribute__((ext_vector_type(16))) unsigned (&mn)[QW], unsigned keep,
                                              unsigned j, int b, int NB, const int (&self_blk)[QW],
                                              const unsigned (&self_bit)[QW], unsigned tail_vm) {
  h8 ah[ND + 1];
#pragma unroll
  for (int t = 0; t <= ND; ++t) ah[t] = frag[t * 64];
  f32x16 acc[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
#pragma unroll
    for (int t = 0; t <= ND; ++t) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bqh[g][t], acc[g], 0, 0, 0);
  }
  unsigned jv;   // the tile index in a VECTOR register: see `keep`
  asm volatile("v_mov_b32 %0, %1" : "=v"(jv) : "s"(j));
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    if (MASKED) {
      if (__any(b == self_blk[g]) || b == NB - 1) {
        const unsigned km = (b == self_blk[g] ? self_bit[g] : 0u) | (b == NB - 1 ? ~tail_vm & 0xffffu : 0u);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = ((km >> e) & 1u) ? INFINITY : acc[g][e];
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const unsigned key = (__float_as_uint(acc[g][e]) & keep) | jv;
      mn[g][e] = key < mn[g][e] ? key : mn[g][e];
    }
  }
}

// Round 4: a workgroup of four waves (4·QW·32 queries) shares every sampled tile through LDS.  As one-wave workgroups
// (rounds 2-3) each wave fetched its own copy of every tile, (ND + 1) KiB per (ND + 1)·QW MFMAs: 76 GB of L2 -> register
// traffic at 125k x 1M, and hipcc spent 4 vector instructions per distance where 2 were meant (see `keep`).  Now wave w
// brings tile 4r + w of round r into the stage of the round (LDS-DMA, issued one round ahead), and every wave works the
// four tiles from there: a quarter of the fetches, one barrier per four tiles.  Keys, slots and therefore T[q] are the
// same bits as before.  125k x 1M: 3.37 -> 2.45 ms, C3: 0.75 -> 0.55 ms, 62.5k x 500k (k = 50, QW = 1): 1.04 -> 0.90 ms.
// Measured and NOT kept (tools/experiments/README.md): the MFMAs of a tile laid over the vector instructions of the
// previous one by hand (sched_group_barrier: 7 % slower — an MFMA of this shape keeps the issue port for ~18 of its
// 32 cycles, tools/experiments/mfma_valu_overlap.hip, and the other wave of the SIMD already fills the rest), a higher
// priority for the wave in the odd slot of each SIMD (nothing).
template <int ND, int NSET, int QW>
__global__ __launch_bounds__(64 * SMP_WAVES, 2) void knn_sample16_kernel(
    const h8 *__restrict__ P16, const float *__restrict__ norms, const unsigned *__restrict__ maxnorm_bits,
    const float *__restrict__ x, const float *__restrict__ mu, int64_t N, int d, int64_t NB, int64_t q_begin,
    int64_t n_q, int include_self, float s, int S, int64_t NBs, int bstride, unsigned *__restrict__ out_key) {
  constexpr int NT = ND + 1;
  static_assert(SMP_WAVES % NSET == 0, "tile t of a round feeds set t % NSET");
  __shared__ h8 stage[2][SMP_WAVES][NT * 64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)(blockIdx.x / S) * SMP_WAVES + wave;   // this wave's QW groups of 32 queries
  const int split = blockIdx.x % S;
  const int64_t b_lo = split * NBs;
  const int64_t b_hi = (b_lo + NBs < NB) ? b_lo + NBs : NB;
  const int r = lane & 31, h = lane >> 5;
  const float s2 = s * s;

  // QW query groups of 32 share every candidate fragment
  h8 bqh[QW][ND + 1];
  int self_blk[QW];        // block indices fit 32 bits: N < 2^31 (knn_run)
  unsigned self_bit[QW];
  int first_blk[QW];   // wave-uniform: the blocks that can hold a query of group g are first_blk[g] and the next
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int64_t ql = (tile * QW + g) * 32 + r;
    const bool qvalid = ql < n_q;
    const int64_t q = q_begin + ql;
    first_blk[g] = (int)((q_begin + (tile * QW + g) * 32) >> 5);
#pragma unroll
    for (int t = 0; t < ND; ++t)
      bqh[g][t] = panel16_unit(x, mu, norms, q, qvalid, d, ND, t, h, -2.0f * s, s2 * (1.0f / 256.0f), true);
#pragma unroll
    for (int j = 0; j < 8; ++j) bqh[g][ND][j] = (_Float16)0.f;
    if (qvalid) {
      // norm slot = s²‖q−μ‖² + the bound of the coarse tile (see knn_collect16c_kernel): every value is positive
      const float nq = norms[q];
      const float bias = 1.953125e-3f * (nq + __uint_as_float(maxnorm_bits[0])) * s2 + (float)d * 3.0517578125e-5f;
      bqh[g][ND] = query_norm_unit((nq * s2 + bias) * (1.0f / 256.0f), h);
    }
    self_blk[g] = -1;
    self_bit[g] = 0;
    if (!include_self) {   // (rows of invalid queries are never read back: nothing to mask)
      const int row = (int)(q & 31);
      if (qvalid && ((row >> 2) & 1) == h) {
        self_blk[g] = (int)(q >> 5);
        self_bit[g] = 1u << ((row & 3) | ((row >> 3) << 2));
      }
    }
  }
  unsigned tail_vm = 0xffffu;
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if ((NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= N) tail_vm &= ~(1u << e);

  using u32x16 = __attribute__((ext_vector_type(16))) unsigned;
  u32x16 mn[NSET][QW];
#pragma unroll
  for (int u = 0; u < NSET; ++u)
#pragma unroll
    for (int g = 0; g < QW; ++g)
#pragma unroll
      for (int e = 0; e < 16; ++e) mn[u][g][e] = 0xffffffffu;
  // One v_and_or_b32 per distance needs the mask in a SCALAR register and the tile index in a VECTOR one (a gfx9 VOP3
  // takes one scalar source and no literal).  Left to itself hipcc folded the mask into v_and_b32 + v_or_b32 with a
  // literal, and with one wave per SIMD allowed it kept the tiles in AGPRs behind 16 v_accvgpr_read each: 4 vector
  // instructions per distance where 2 were meant (now 1.5: hipcc fuses the updates of two tiles of a set into
  // v_min3_u32); the launch bound asks for two waves per SIMD, i.e. MFMAs that write VGPRs.
  unsigned keep;
  asm volatile("s_mov_b32 %0, 0xffffe000" : "=s"(keep));
  static_assert(SKEY_BITS == 13, "the literal above is ~SKEY_MASK");

  // sampled tiles of this range: i = 0 .. nt-1 at block b_lo + i·bstride; tile i feeds set i % NSET and carries i in its keys
  const int64_t nt = b_hi > b_lo ? (b_hi - b_lo + bstride - 1) / bstride : 0;
  const int64_t rounds = (nt + SMP_WAVES - 1) / SMP_WAVES;
  // wave w brings tile 4·round + w into its quarter of the stage by LDS-DMA (no registers; a unit of 64 lanes x 16 bytes
  // is contiguous in the panel and in LDS)
  auto fetch = [&](int64_t round, int buf) {
    int64_t i = round * SMP_WAVES + wave;
    if (i >= nt) i = nt - 1;   // past the end: any valid tile, never worked
    const h8 *pb = P16 + (b_lo + i * bstride) * (NT * 64) + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) CC_KNN_GLDS16(pb + t * 64, &stage[buf][wave][t * 64]);
  };
  if (rounds > 0) fetch(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int64_t rd = 0; rd < rounds; ++rd) {
    const int buf = (int)(rd & 1);
    // the other half of the stage was read in round rd - 1 and every wave is past the barrier that ended it; the
    // requests have this whole round to land
    if (rd + 1 < rounds) fetch(rd + 1, buf ^ 1);
    const int64_t i0 = rd * SMP_WAVES;
    const int bf = (int)(b_lo + i0 * bstride), bl = bf + (SMP_WAVES - 1) * bstride;   // blocks of the round: bf .. bl
    bool special = bl >= (int)NB - 1 || i0 + SMP_WAVES > nt;
#pragma unroll
    for (int g = 0; g < QW; ++g) special = special || (first_blk[g] + 1 >= bf && first_blk[g] <= bl);
    if (!special) {   // wave-uniform: four whole tiles, none with a self pair or the padding: no branch inside
#pragma unroll
      for (int t4 = 0; t4 < SMP_WAVES; ++t4) {
        sample16_tile<ND, QW, false>(&stage[buf][t4][lane], bqh, mn[t4 % NSET], keep, (unsigned)(i0 + t4),
                                     bf + t4 * bstride, (int)NB, self_blk, self_bit, tail_vm);
        // two tiles in flight at most (fragments + accumulators of a third do not fit 256 registers: hipcc spilled)
        if (t4 % 2 == 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t4 = 0; t4 < SMP_WAVES; ++t4)
        if (i0 + t4 < nt)
          sample16_tile<ND, QW, true>(&stage[buf][t4][lane], bqh, mn[t4 % NSET], keep, (unsigned)(i0 + t4),
                                      bf + t4 * bstride, (int)NB, self_blk, self_bit, tail_vm);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's tile of the next round is in LDS
    __syncthreads();
  }
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int64_t ql = (tile * QW + g) * 32 + r;
    if (ql < n_q) {
      // slot s = u·32 + h·16 + e of (query, range): the consumer decodes the candidate from s and the key's tile index
      using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
      u32x4 *dst = reinterpret_cast<u32x4 *>(out_key + ((ql * S + split) * NSET * 32 + h * 16));
#pragma unroll
      for (int u = 0; u < NSET; ++u)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          u32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = mn[u][g][4 * jj + i];
          dst[u * 8 + jj] = v;
        }
    }
  }
}

// T[q] from the query's n = 32·NSET·S slot keys (a multiple of 64; layout and key format: knn_sample16_kernel).  One
// wave per query: the m-th smallest key by bisection over its bits, counting with ballots; the lanes holding keys at or
// below it (m of them, more on ties) evaluate their candidate's distance directly in fp32, T = the largest of those.
// Fewer than m real keys: T = +inf (the collect lists everything it can hold; stage 3 takes over if that overflows).
// MAXR = key registers per lane the instance is unrolled for (>= n / 64).  The key loops are unrolled MAXR times with a
// `j < nreg` test per step, and hipcc keeps most of a skipped step's vector work (selects instead of branches): the
// single 16-register instance issued 903 vector + 576 scalar instructions per query at C3, whose queries hold n = 192 keys
// (3 registers per lane) — `tools/experiments/README.md`, the round's re-entry session.
template <int MAXR>
__global__ __launch_bounds__(256) void knn_slot_threshold_kernel(const unsigned *__restrict__ skey, const float *__restrict__ x,
                                                                 int d, int64_t q_begin, int64_t n_q, int n, int m, int NSET,
                                                                 int64_t NBs, int bstride, float *__restrict__ T) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ql = (int64_t)blockIdx.x * 4 + wave;
  if (ql >= n_q) return;
  unsigned key[MAXR];
  const int nreg = n / 64;
#pragma unroll
  for (int j = 0; j < MAXR; ++j) key[j] = j < nreg ? skey[ql * n + j * 64 + lane] : 0xffffffffu;
  // only the distance bits order the keys (bit 31 is clear, the low SKEY_BITS are the tile index): 18 steps; keys
  // that agree with the m-th smallest in those bits are all taken
  // (round 4) The bisection starts at the highest bit in which two real keys differ: the bits above it are the same in
  // every real key, hence in the m-th smallest — the steps it would spend on them can only copy them (a common 1: no
  // key lies under the trial value; a common 0: all do).  Slot minima of one query lie within a few binades, so a third
  // of the 18 steps goes.  Fewer than m real keys: the m-th smallest is not a distance, T = +inf as before.
  unsigned k_or = 0u, k_and = 0x7fffffffu;
  int n_real = 0;
#pragma unroll
  for (int j = 0; j < MAXR; ++j)
    if (j < nreg) {
      const bool real = key[j] < 0x7f800000u;
      k_or |= real ? key[j] : 0u;
      k_and &= real ? key[j] : 0x7fffffffu;
      n_real += __builtin_popcountll(__ballot(real));
    }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    k_or |= __shfl_xor(k_or, o);
    k_and &= __shfl_xor(k_and, o);
  }
  const unsigned differ = (k_or ^ k_and) & ~SKEY_MASK;
  const int top = differ ? 31 - __builtin_clz(differ) : SKEY_BITS - 1;   // wave-uniform
  unsigned prefix = n_real >= m ? (k_and & ~((2u << top) - 1u)) : 0x7f800000u;
  for (int bit = n_real >= m ? top : -1; bit >= SKEY_BITS; --bit) {
    const unsigned trial = prefix | ((1u << bit) - 1u);
    int c = 0;
#pragma unroll
    for (int j = 0; j < MAXR; ++j)
      if (j < nreg) c += __builtin_popcountll(__ballot(key[j] <= trial));
    if (c < m) prefix |= 1u << bit;
  }
  prefix |= SKEY_MASK;
  float worst = 0.f;
  if (prefix < 0x7f800000u) {   // the m-th smallest key is a real distance
    const int vec = (d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
    const float *xq = x + (q_begin + ql) * d;
    const int per_range = NSET * 32;
    // The chosen keys sit anywhere in the wave's registers (a lane may hold several, most hold none): they are first
    // compacted into a wave-private list, then ONE lane per candidate evaluates its distance — all of them at once
    // instead of one dependent gather per register row (round 3: 0.30 -> 0.23 ms per 100k queries; round 4: 0.17 with the key loops unrolled for the registers a query really has).  The distance of a
    // candidate is the same sequential fp32 sum as before (the collect's final test compares against exactly that).
    __shared__ int64_t chosen[4][64];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0;; base += 64) {   // (more than 64 keys only when many agree in the compared bits)
      int total = 0;                    // wave-uniform
#pragma unroll
      for (int j = 0; j < MAXR; ++j) {
        if (j < nreg) {
          const bool sel = key[j] <= prefix;
          const unsigned long long bal = __ballot(sel);
          const int pos = total + __builtin_popcountll(bal & below) - base;
          if (sel && pos >= 0 && pos < 64) {
            const int slot = j * 64 + lane;
            const int split = slot / per_range, sl = slot - split * per_range;
            const int e = sl & 15, h = (sl >> 4) & 1;
            const int64_t b = split * NBs + (int64_t)(key[j] & SKEY_MASK) * bstride;
            chosen[wave][pos] = b * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          }
          total += __builtin_popcountll(bal);
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the list is wave-private: its writes are done, no block barrier
      const int here = total - base < 64 ? total - base : 64;
      if (lane < here) worst = fmaxf(worst, fine_dist_batched(xq, x + chosen[wave][lane] * d, d, vec));
      if (total - base <= 64) break;
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    worst = INFINITY;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
  if (lane == 0) T[ql] = worst;
}

// ---- stage 1b, coarse form of the collect (default) ------------------------------------------------------------
// The 3-product tile above evaluates every (query, candidate) pair to ~2^-22, but only ~1.3 of a tile's 1024 pairs
// pass the threshold.  Here the tile is the hi·hi product alone (ND + 1 MFMAs instead of 3·ND + 1: 5 instead of 13
// at d = 64), accurate to 2^-10 of (‖q−μ‖² + ‖c−μ‖²); pairs at or below T + that bound are parked, and when a wave's
// parking buffer is flushed each lane re-evaluates ONE parked pair directly, Σ (q_j − c_j)² in fp32 from the
// original coordinates (error ~d·2^-24 of the distance itself), and only pairs with that distance <= T are appended.
// What stage 2 needs holds a fortiori: an unlisted candidate either failed the coarse test (true distance > T) or the
// fine one (true distance > T − ε_fine, ε_fine far below the ε of the certificate).
//
// Bound of the coarse tile (scaled by s²):  |acc − s²·dist| <= 2·Σ|qh·ch − s²·q·c| + fp32 accumulation.  Each operand
// is y·s rounded to fp16: relative 2^-11 above the subnormal range (|y·s| >= 2^-14), absolute 2^-25 below it, so
// |qh_i·ch_i − s²q_i c_i| <= (2^-10 + 2^-22) s²|q_i c_i| + 2^-25·2^7·2 ; summed (Cauchy-Schwarz, 2|q||c| <= |q|²+|c|²)
// and doubled for the −2q·c factor: <= 2^-10·1.001·s²(‖q‖² + ‖c‖²) + d·2^-16, plus 16(ND+1) fp32 roundings on terms that
// sum to <= 3 s²(‖q‖² + ‖c‖²) (2^-18 relative).  Used: 2^-9 · s²(‖q−μ‖² + max‖c−μ‖²) + d·2^-15  (>= 1.9x the bound).
//
// The test itself costs one instruction per pair: the query's norm slot of the operand carries s²‖q−μ‖² − Ts (Ts = the
// scaled threshold + bound, capped at 1.5·2^23, above any scaled distance at d <= 128 — the three-piece fp16 split holds
// it to 2^-33), so the accumulator IS (coarse distance − Ts) and a pair passes when its sign bit is set.  The 16 sign
// bits of a lane's accumulators are shifted into one mask register (v_alignbit), lanes with a non-zero mask park their
// pairs in rounds (one pair per lane and round), and a tile none of whose 1024 signs is set is rejected by an OR tree
// when hits are rare.  The fp32 rounding of the shifted accumulation (a few units at 2^-2) disappears in the spare
// half of the bound (2^-10 · s²·norms: thousands of units).  Masked pairs (self, padding) are set to +0.

struct WaveBufC {
  unsigned *cand;   // [COL_WB] candidate index
  int *qr;          // [COL_WB] query column 0..31
  int n;
};

__device__ __forceinline__ void wavebufc_flush(WaveBufC &wb, const float *__restrict__ x, int d, int vec, int64_t q_begin,
                                               const float *__restrict__ T, u64 *__restrict__ buf,
                                               int *__restrict__ cursor, int64_t q0, int64_t n_q, int cap, int lane) {
  for (int i = lane; i < wb.n; i += 64) {
    const int64_t ql = q0 + wb.qr[i];
    if (ql < n_q) {
      const unsigned cand = wb.cand[i];
      // (fine_dist, not its batched form: 64 registers of loads in flight here were the register peak of the whole kernel —
      // 168 -> 155 / 159 -> 132 without them — and the flush is rare enough that the one-quad-at-a-time form costs nothing:
      // measured 1-3 % FASTER at every shape, tools/experiments/knn_w4b_ab.sh)
      const float fine = fine_dist(x + (q_begin + ql) * d, x + (int64_t)cand * d, d, vec);
      if (fine <= T[ql]) {
        const int pos = atomicAdd(cursor + ql, 1);
        if (pos < cap) buf[ql * cap + pos] = ((u64)f32_to_ordered(fine) << 32) | (u64)cand;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  wb.n = 0;
}

// the passing pairs of one 32 x 32 tile (acc = coarse distance − threshold: sign bit set = pass) as a lane mask:
// bit (15 − e) = pair (this lane's query, candidate entry e) passes; 0 for the whole wave when the tile is rejected
__device__ __forceinline__ unsigned collect16c_mask(f32x16 &acc, int b, int NB, int self_blk, unsigned self_bit,
                                                    unsigned tail_vm, int min_first) {
  if (__any(b == self_blk) || b == NB - 1) {
    // masked pairs (the query itself, padding rows of the last block) become +0; plain bit arithmetic on a per-lane
    // mask, so that no per-entry lane masks live in scalar registers across the loop
    const unsigned km = (b == self_blk ? self_bit : 0u) | (b == NB - 1 ? ~tail_vm & 0xffffu : 0u);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = __uint_as_float(__float_as_uint(acc[e]) & (((km >> e) & 1u) - 1u));
  }
  if (min_first) {
    // few pairs pass per tile when the candidate set is large (1024 * 8m / N expected: 0.13 at N = 1M): an OR tree
    // over the 16 accumulators + one ballot of its sign rejects the whole tile
    unsigned o = __float_as_uint(acc[0]) | __float_as_uint(acc[1]) | __float_as_uint(acc[2]);
    o |= __float_as_uint(acc[3]) | __float_as_uint(acc[4]);
    o |= __float_as_uint(acc[5]) | __float_as_uint(acc[6]);
    o |= __float_as_uint(acc[7]) | __float_as_uint(acc[8]);
    o |= __float_as_uint(acc[9]) | __float_as_uint(acc[10]);
    o |= __float_as_uint(acc[11]) | __float_as_uint(acc[12]);
    o |= __float_as_uint(acc[13]) | __float_as_uint(acc[14]);
    o |= __float_as_uint(acc[15]);
    if (!__any((int)o < 0)) return 0u;
  }
  unsigned hm = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) hm = __builtin_amdgcn_alignbit(hm, __float_as_uint(acc[e]), 31);
  return hm;
}

// parks the pairs of up to four tiles at once: m64 holds their lane masks, 16 bits per tile slot, slot = (block − b0)·QW
// + query group.  One pair per lane and round, wave-uniform trip count (the rounds cost a ballot, a popcount and a
// branch each: sharing them between the tiles of two blocks is what this grouping is for).
template <int QW>
__device__ __forceinline__ void collect16c_park(unsigned long long m64, int b0, WaveBufC &wb,
                                                const float *__restrict__ x, int d, int vec, int64_t q_begin,
                                                const float *__restrict__ T, u64 *__restrict__ buf,
                                                int *__restrict__ cursor, int64_t q0, int64_t n_q, int cap, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const unsigned cbase = (unsigned)b0 * 32u + 4 * h;
  while (true) {
    const unsigned long long lm = __ballot(m64 != 0);
    if (!lm) break;
    const int cnt = __builtin_popcountll(lm);
    if (wb.n + cnt > COL_WB) wavebufc_flush(wb, x, d, vec, q_begin, T, buf, cursor, q0, n_q, cap, lane);
    if (m64 != 0) {
      const int bit = __builtin_ffsll((long long)m64) - 1;
      const int slot = bit >> 4, e = 15 - (bit & 15);
      const int pos = wb.n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
      wb.cand[pos] = cbase + 32 * (slot / QW) + (e & 3) + 8 * (e >> 2);
      wb.qr[pos] = (slot % QW) * 32 + r;
      m64 &= m64 - 1;
    }
    wb.n += cnt;
  }
}

// QW = 32-query groups per wave: 2 makes a workgroup own 256 queries, so a candidate panel crosses the fabric and LDS
// once per 256 queries instead of 128 and every fragment read feeds two MFMA chains; 1 keeps more workgroups for
// short query ranges.  The hi + norm units of CB candidate blocks form a stage (8-12 KiB); stages reach LDS by
// LDS-DMA (global_load_lds: a unit = 64 lanes x 16 bytes is contiguous in the panel and in LDS) through a ring of
// C16C_NSTAGE buffers, the DMA running NSTAGE - 1 stages ahead of the MFMAs with counted vmcnt waits (at N = 1M
// a quarter of the panel reads miss the L2 and take > 1 us: one stage ahead left the workgroup waiting at every
// barrier), one barrier per stage.
//
// Round 4 (the counters of round 3 showed 5.7 scalar + 5.5 vector instructions per MFMA: control, not arithmetic):
//  * a stage is CONTIGUOUS in the panel (blocks b .. b + CB - 1 = NU unit rows of 1 KiB back to back), so the DMA source
//    is one wave-uniform running pointer advanced once per stage + a constant per-lane offset, and the NU rows are dealt
//    to the four waves without duplicates (round 3 computed a clamped 64-bit block address per request — ~15 scalar
//    instructions each — and issued 12 requests for 10 rows); the panel carries C16C_PAD_BLOCKS blocks of padding so a
//    stage that straddles the end of the range needs no clamp (blocks beyond the range are loaded and not used);
//  * block indices are 32-bit (N < 2^31: at most 2^26 blocks);
//  * a stage is SPECIAL when it holds a block with one of the wave's own queries (self pairs), the partly filled last
//    block, or the end of the range: only then the per-tile masking code runs (collect16c_mask); every other stage takes
//    a straight path — MFMAs of all its tiles, the sign bits of each tile folded into a lane mask (16 v_alignbit), one
//    parking call per pair of blocks — with no per-tile compare, ballot or branch;
//  * when hits are rare (min_first: N large against the list size) the straight path first ORs ALL accumulators of a
//    block pair (v_or3 tree) and rejects the pair with ONE ballot; only a pair with a hit looks at its tiles;
//  * (the parked pairs were re-evaluated with 32 coordinates in flight for a while — fine_dist_batched, still what the
//    threshold kernel uses; the round's last session went back to fine_dist here: see wavebufc_flush.)
constexpr int C16C_NSTAGE = 4;
constexpr int C16C_PAD_BLOCKS = 4;   // >= the largest CB: padding blocks behind the panel (knn_pack16_kernel fills them as "no candidate")
// The stage barrier of the ring as the bare instruction: __syncthreads() carries a fence that waits vmcnt(0) while
// LDS-DMA writes are pending, which drained the ring at every stage (round 3: the counted vmcnt in front of it was
// followed by the fence's vmcnt(0) in the assembly).  Each wave's own fragment reads of the stage are consumed by its
// MFMAs before it gets here, its own DMA pieces of the next stage are covered by the counted wait in front.
__device__ __forceinline__ void knn_stage_barrier() {
  asm volatile("" ::: "memory");   // the compiler moves no LDS access across the barrier
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// CN (round 4, late): the candidates' norms enter through the accumulator instead of a K-step of their own — the
// first MFMA of a tile takes C = s²‖c−μ‖² (16 fp32 per lane, read from a 1 KiB norm row of the stage: all lanes of a
// half read the same 16 bytes), the tile is ND MFMAs instead of ND + 1 (4 instead of 5 at d = 64), and a pair passes
// when accumulator < Ts − s²‖q−μ‖² (one value per lane).  That comparison has no one-instruction sign form, so CN is
// the kernel of the rare-hit shapes only: a v_min3 tree over the tiles of a block pair and ONE compare + ballot reject
// it (the same count as the v_or3 tree of the K-step form); a tile with a hit pays 16 subtractions first.
// NW = waves per workgroup, CBM = stage length multiplier: the product runs NW = 4 (three workgroups per CU, each with
// its own stage ring), CBM = 1.  ONE 12-wave workgroup per CU (768 queries share every candidate stage: a third of the
// LDS-DMA requests and panel bytes per MFMA), with stages of 2 or 4 blocks, was built on these parameters, is bit-identical
// and measured null to slower at every shape (tools/experiments/knn_collect_12waves.patch has its launcher and options).
template <int ND, bool CN = false, int NW = 4, int CBM = 1>
struct C16C {
  static constexpr int CB = (ND == 1 ? 4 : ND == 2 ? 4 : ND == 4 ? 2 : 1) * CBM;   // candidate blocks per stage
  static constexpr int TU = CN ? ND : ND + 1;                               // unit rows of a block in the stage: hi (+ norm K-step)
  static constexpr int HU = TU * 64;                                        // their 16-byte units
  static constexpr int NU = CB * TU + (CN ? 1 : 0);                         // 64-lane unit rows (1 KiB) per stage; CN: the last one = fp32 norms
  static constexpr int STAGE_U = NU * 64;
  static constexpr int DPS = (NU + NW - 1) / NW;                            // most DMA instructions a wave issues per stage
  static_assert(CB <= C16C_PAD_BLOCKS, "panel padding");
  static constexpr size_t lds_bytes() {
    return sizeof(h8) * C16C_NSTAGE * STAGE_U + (sizeof(unsigned) + sizeof(int)) * NW * COL_WB;
  }
};

// lane mask of one tile without any test: bit (15 - e) = sign of accumulator e
__device__ __forceinline__ unsigned collect16c_signs(const f32x16 &acc) {
  unsigned hm = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) hm = __builtin_amdgcn_alignbit(hm, __float_as_uint(acc[e]), 31);
  return hm;
}
// OR of the 16 accumulators of a tile (sign bit set = some pair of this lane passes)
__device__ __forceinline__ unsigned collect16c_or(const f32x16 &acc) {
  unsigned o = __float_as_uint(acc[0]) | __float_as_uint(acc[1]) | __float_as_uint(acc[2]);
  o |= __float_as_uint(acc[3]) | __float_as_uint(acc[4]);
  o |= __float_as_uint(acc[5]) | __float_as_uint(acc[6]);
  o |= __float_as_uint(acc[7]) | __float_as_uint(acc[8]);
  o |= __float_as_uint(acc[9]) | __float_as_uint(acc[10]);
  o |= __float_as_uint(acc[11]) | __float_as_uint(acc[12]);
  o |= __float_as_uint(acc[13]) | __float_as_uint(acc[14]);
  o |= __float_as_uint(acc[15]);
  return o;
}

// lane minimum of the 16 accumulators of a tile
__device__ __forceinline__ float collect16c_min(const f32x16 &acc) {
  float m = fminf(fminf(acc[0], acc[1]), acc[2]);
#pragma unroll
  for (int e = 3; e < 15; e += 2) m = fminf(fminf(m, acc[e]), acc[e + 1]);
  return fminf(m, acc[15]);
}

template <int ND, int QW, bool CN, int NW, int CBM>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 1) void knn_collect16c_kernel(
    const h8 *__restrict__ P16, const float *__restrict__ norms, const unsigned *__restrict__ maxnorm_bits,
    const float *__restrict__ x, const float *__restrict__ mu, int64_t N, int d, int NB, int64_t q_begin,
    int64_t n_q, int include_self, const float *__restrict__ T, float s, int S, int NBs, int cap,
    u64 *__restrict__ buf, int *__restrict__ cursor, int min_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C_ = C16C<ND, CN, NW, CBM>;
  constexpr int NT = ND + 1;                 // units per candidate block in the panel
  constexpr int TU = C_::TU;                 // unit rows of a block that reach the stage (CN: without the norm K-step)
  constexpr int CB = C_::CB, HU = C_::HU, STAGE_U = C_::STAGE_U, NU = C_::NU, DPS = C_::DPS, NSTAGE = C16C_NSTAGE;
  constexpr int QPW = 32 * QW;               // queries per wave
  constexpr int CBH = CB / CBM;              // blocks worked off together (= CB unless the stage is stretched)
  constexpr int PB = CBH >= 2 ? 2 : 1;       // blocks handled (and parked) together: PB * QW <= 4 tiles = the 64 mask bits of a lane
  h8 *stage = reinterpret_cast<h8 *>(smem);  // [NSTAGE][STAGE_U], then the 4 parking buffers

  const int tile = (int)(blockIdx.x / (unsigned)S);
  const int split = (int)(blockIdx.x % (unsigned)S);
  const int b_lo = split * NBs;
  const int b_hi = (b_lo + NBs < NB) ? b_lo + NBs : NB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < NW);   // the request loop's "does this wave have a row i" tests fold for all but the last i
  const int r = lane & 31, h = lane >> 5;
  const float s2 = s * s;
  const int vec = (d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
  const int64_t q0 = (int64_t)tile * (NW * QPW) + wave * QPW;
  const int nstages = (b_hi - b_lo + CB - 1) / CB;

  // DMA: unit rows u = wave, wave + 4, .. < NU of a stage; source = running wave-uniform pointer + lane * 16.
  // CN: stage row u < CB·ND is panel row (u / ND)·NT + u % ND of the stage's first block (the norm K-steps stay
  // behind), the last row = 1 KiB of ncs from the stage's first candidate on (8 blocks' worth, CB of them used)
  const int my_dps = (NU - wave + NW - 1) / NW;   // wave-uniform, DPS or DPS - 1
  const char *gsrc = reinterpret_cast<const char *>(P16 + (int64_t)b_lo * (NT * 64)) + (CN ? 0 : wave * 1024);
  // (the fp32 norms sit behind the panel and its padding blocks in the same allocation: knn_pack16_kernel)
  const char *nsrc = CN ? reinterpret_cast<const char *>(P16 + (int64_t)(NB + C16C_PAD_BLOCKS) * (NT * 64)) + (int64_t)b_lo * 128
                        : nullptr;
  const unsigned voff = (unsigned)lane * 16u;
  int roff[CN ? DPS : 1];                    // CN: wave-uniform byte offsets of this wave's rows inside the stage's panel slice
  if constexpr (CN) {
#pragma unroll
    for (int i = 0; i < DPS; ++i) {
      const int u = wave + NW * i;
      roff[i] = ((u / ND) * NT + u % ND) * 1024;
    }
  }
  auto issue = [&](int slot_u) {   // slot_u = slot · STAGE_U (running offsets below: no multiplication per stage)
#pragma unroll
    for (int i = 0; i < DPS; ++i)
      if (wave + NW * i < NU) {   // wave-uniform
        if constexpr (CN) {
          if (wave + NW * i == NU - 1) CC_KNN_GLDS16(nsrc + voff, stage + (slot_u + (NU - 1) * 64));
          else CC_KNN_GLDS16(gsrc + roff[i] + voff, stage + (slot_u + (wave + NW * i) * 64));
        } else {
          CC_KNN_GLDS16(gsrc + i * (NW * 1024) + voff, stage + (slot_u + (wave + NW * i) * 64));
        }
      }
    if constexpr (CN) { gsrc += CB * NT * 1024; nsrc += CB * 128; }
    else gsrc += NU * 1024;
  };
  // wait until at most `keep` of this wave's stages are still in flight (a stage = my_dps instructions)
  auto wait_stages = [&](bool counted) {
    if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (my_dps == DPS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (DPS - 1)) : "memory");
  };
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p)
    if (p < nstages) issue(p * STAGE_U);   // under the set-up below

  h8 bqh[QW][TU];
  float thr[QW];   // CN: Ts − s²‖q−μ‖², a pair passes when its accumulator is below
  int self_blk[QW];
  unsigned self_bit[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int64_t ql = q0 + g * 32 + r;
    const bool qvalid = ql < n_q;
    const int64_t q = q_begin + ql;
#pragma unroll
    for (int t = 0; t < ND; ++t)
      bqh[g][t] = panel16_unit(x, mu, norms, q, qvalid, d, ND, t, h, -2.0f * s, s2 * (1.0f / 256.0f), true);
    // norm slot = s²‖q−μ‖² − Ts: the accumulator becomes (coarse distance − threshold), see the bound above.
    // CN: the same value, negated, is the lane's threshold (an invalid query has zero operands: its accumulators are the
    // candidates' norms >= 0 and never fall below −1)
    if constexpr (!CN) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bqh[g][ND][j] = (_Float16)0.f;
    }
    if constexpr (CN) thr[g] = -1.0f;
    if (qvalid) {
      const float nq = norms[q];
      const float Ts = fminf((T[ql] + 1.953125e-3f * (nq + __uint_as_float(maxnorm_bits[0]))) * s2 + (float)d * 3.0517578125e-5f,
                             12582912.0f);
      if constexpr (CN) thr[g] = Ts - nq * s2;
      else bqh[g][ND] = query_norm_unit((nq * s2 - Ts) * (1.0f / 256.0f), h);
    }
    self_blk[g] = -1;
    self_bit[g] = 0;
    if (qvalid && !include_self) {
      const int row = (int)(q & 31);
      if (((row >> 2) & 1) == h) {
        self_blk[g] = (int)(q >> 5);
        self_bit[g] = 1u << ((row & 3) | ((row >> 3) << 2));
      }
    }
  }
  // the candidate blocks that hold one of this wave's own queries (wave-uniform; empty when self pairs are admitted)
  const int sp_lo = include_self ? 0x7fffffff : (int)((q_begin + q0) >> 5);
  const int sp_hi = include_self ? -1 : (int)((q_begin + q0 + QPW - 1) >> 5);
  const int tail_blk = (N & 31) ? NB - 1 : 0x7fffffff;   // the partly filled last block, if any
  // The same three conditions as stage indices, once per workgroup instead of ~20 scalar instructions per stage (the
  // collect's time follows its instruction count, tools/experiments/README.md): stage st (blocks b_lo + st·CB ..) is
  // plain when st < st_end (all its blocks inside the range and in front of a partly filled last block) and it holds
  // none of this wave's own queries, st outside [sp_first, sp_first + sp_span].
  auto floor_div = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
  const int lim = (b_hi < tail_blk ? b_hi : tail_blk) - b_lo;
  const int st_end = lim > 0 ? lim / CB : 0;
  const int sp_a = floor_div(sp_lo - b_lo, CB), sp_b = floor_div(sp_hi - b_lo, CB);
  const int sp_first = sp_b >= sp_a ? sp_a : 0x7fffffff;          // empty (self pairs admitted): no stage matches
  const unsigned sp_span = sp_b >= sp_a ? (unsigned)(sp_b - sp_a) : 0u;
  WaveBufC wb;
  wb.cand = reinterpret_cast<unsigned *>(smem + sizeof(h8) * NSTAGE * STAGE_U) + wave * COL_WB;
  wb.qr = reinterpret_cast<int *>(smem + sizeof(h8) * NSTAGE * STAGE_U + sizeof(unsigned) * NW * COL_WB) + wave * COL_WB;
  wb.n = 0;

  unsigned tail_vm = 0xffffu;
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if ((int64_t)(NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= N) tail_vm &= ~(1u << e);

  // stage 0 must have landed; NSTAGE - 2 later ones may stay in flight
  wait_stages(nstages >= NSTAGE - 1);
  knn_stage_barrier();

  int rd_u = 0, wr_u = (NSTAGE - 1) * STAGE_U;   // unit offsets of the slot read now and of the slot refilled now = (slot + NSTAGE - 1) % NSTAGE
  for (int st = 0; st < nstages; ++st) {
    // the slot written now was read one stage ago, and every wave is past that barrier
    const bool ahead = st + NSTAGE - 1 < nstages;
    if (ahead) issue(wr_u);
    // (CBM > 1: the stage is worked off in CBM parts of CBH blocks, so that only a part's fragments are in registers)
#pragma unroll
    for (int hf = 0; hf < CBM; ++hf) {
    const h8 *sbuf = stage + rd_u + hf * CBH * HU + lane;
    h8 ah[CBH][TU];
#pragma unroll
    for (int cb = 0; cb < CBH; ++cb)
#pragma unroll
      for (int t = 0; t < TU; ++t) ah[cb][t] = sbuf[cb * HU + t * 64];
    // CN: what the accumulators of block cb start from — entry e of lane (r, h) is candidate (e & 3) + 8 (e >> 2) + 4 h:
    // four runs of four consecutive floats of the block's 32 (every lane of a half reads the same 16 bytes)
    const f32x4 *nbuf = reinterpret_cast<const f32x4 *>(stage + rd_u + (NU - 1) * 64) + hf * CBH * 8 + h;
    // (read inside the path that uses them: hoisted beside the fragment reads they stay live across the masked path
    // and the kernel spills)
    auto cinit = [&](int cb) {
      f32x16 c;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 v = nbuf[cb * 8 + 2 * gq];
        c[4 * gq] = v[0]; c[4 * gq + 1] = v[1]; c[4 * gq + 2] = v[2]; c[4 * gq + 3] = v[3];
      }
      return c;
    };
    const int b0 = b_lo + st * CB + hf * CBH;
    // wave-uniform: no block of this stage needs a mask (own queries, padding rows) and all of them lie in the range
    const bool plain = CBM == 1 ? (st < st_end && (unsigned)(st - sp_first) > sp_span)
                                : (b0 + CBH <= b_hi && (b0 > sp_hi || b0 + CBH <= sp_lo) && b0 + CBH <= tail_blk);
    if (plain) {
#pragma unroll
      for (int pb = 0; pb < CBH; pb += PB) {
        f32x16 acc[PB][QW];
#pragma unroll
        for (int c = 0; c < PB; ++c) {
          f32x16 c0;
          if constexpr (CN) c0 = cinit(pb + c);
#pragma unroll
          for (int g = 0; g < QW; ++g) {
            if constexpr (CN) {
              acc[c][g] = c0;
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[c][g][e] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < TU; ++t)
              acc[c][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[pb + c][t], bqh[g][t], acc[c][g], 0, 0, 0);
          }
        }
        unsigned long long m64 = 0;
        if constexpr (CN) {
          float mn[PB][QW];
          bool hit = false;
#pragma unroll
          for (int g = 0; g < QW; ++g) {
            float mg = 3.0e38f;
#pragma unroll
            for (int c = 0; c < PB; ++c) { mn[c][g] = collect16c_min(acc[c][g]); mg = fminf(mg, mn[c][g]); }
            hit |= mg < thr[g];
          }
          if (__any(hit)) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
#pragma unroll
              for (int g = 0; g < QW; ++g)
                if (__any(mn[c][g] < thr[g])) {
                  f32x16 df;
#pragma unroll
                  for (int e = 0; e < 16; ++e) df[e] = acc[c][g][e] - thr[g];
                  m64 |= (unsigned long long)collect16c_signs(df) << (16 * (c * QW + g));
                }
          }
        } else if (min_first) {
          unsigned o[PB][QW], oall = 0;
#pragma unroll
          for (int c = 0; c < PB; ++c)
#pragma unroll
            for (int g = 0; g < QW; ++g) { o[c][g] = collect16c_or(acc[c][g]); oall |= o[c][g]; }
          if (__any((int)oall < 0)) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
#pragma unroll
              for (int g = 0; g < QW; ++g)
                if (__any((int)o[c][g] < 0)) m64 |= (unsigned long long)collect16c_signs(acc[c][g]) << (16 * (c * QW + g));
          }
        } else {
#pragma unroll
          for (int c = 0; c < PB; ++c)
#pragma unroll
            for (int g = 0; g < QW; ++g) m64 |= (unsigned long long)collect16c_signs(acc[c][g]) << (16 * (c * QW + g));
        }
        collect16c_park<QW>(m64, b0 + pb, wb, x, d, vec, q_begin, T, buf, cursor, q0, n_q, cap, lane);
      }
    } else {
      unsigned long long m64 = 0;
#pragma unroll
      for (int cb = 0; cb < CBH; ++cb) {
        const int b = b0 + cb;
        if (b < b_hi) {   // workgroup-uniform
          f32x16 acc[QW];
          f32x16 c0;
          if constexpr (CN) c0 = cinit(cb);
#pragma unroll
          for (int g = 0; g < QW; ++g) {
            if constexpr (CN) {
              acc[g] = c0;
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < TU; ++t)
              acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][t], bqh[g][t], acc[g], 0, 0, 0);
            if constexpr (CN) {   // the sign form of the masked path (rare: own queries, the last block, the end of the range)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[g][e] -= thr[g];
            }
          }
#pragma unroll
          for (int g = 0; g < QW; ++g)
            m64 |= (unsigned long long)collect16c_mask(acc[g], b, NB, self_blk[g], self_bit[g], tail_vm, min_first)
                   << (16 * ((PB == 2 ? (cb & 1) : 0) * QW + g));
        }
        if (PB == 1 || (cb & 1) || cb == CBH - 1) {   // the tiles of PB candidate blocks are parked together
          collect16c_park<QW>(m64, b - (PB == 2 ? (cb & 1) : 0), wb, x, d, vec, q_begin, T, buf, cursor, q0, n_q, cap, lane);
          m64 = 0;
        }
      }
    }
    }   // hf
    // the next stage's data must have landed (in the tail fewer stages are in flight than the count assumes)
    wait_stages(ahead);
    knn_stage_barrier();
    rd_u = rd_u == (NSTAGE - 1) * STAGE_U ? 0 : rd_u + STAGE_U;
    wr_u = wr_u == (NSTAGE - 1) * STAGE_U ? 0 : wr_u + STAGE_U;
  }
  wavebufc_flush(wb, x, d, vec, q_begin, T, buf, cursor, q0, n_q, cap, lane);
}

__global__ void knn_fill_kernel(float *__restrict__ p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// sample threshold: T[q] = approximate distance of the m-th best sampled candidate (+inf if fewer);
// the sample pass may itself be split over S candidate ranges (S sorted lists per query: m-step merge)
__global__ void knn_threshold_kernel(const u64 *__restrict__ skeys, const int *__restrict__ scnt, int64_t n_q,
                                     int S, int KPS, int m, float *__restrict__ T) {
  const int64_t ql = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ql >= n_q) return;
  int head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 last = 0;
  int taken = 0;
  for (; taken < m; ++taken) {
    int best = -1;
    u64 bk = ~0ull;
    for (int sp = 0; sp < S; ++sp) {
      if (head[sp] < scnt[ql * S + sp]) {
        const u64 kx = skeys[(ql * S + sp) * KPS + head[sp]];
        if (best < 0 || kx < bk) { bk = kx; best = sp; }
      }
    }
    if (best < 0) break;
    head[best] += 1;
    last = bk;
  }
  T[ql] = taken == m ? ordered_to_f32((unsigned)(last >> 32)) : INFINITY;
}

// ---- merge: the K' smallest keys of a query's candidate lists -------------------------------------
// One wave per query.  Input: S lists of at most `lcap` keys each (route 1a: the per-split select
// lists; route 1b: S = 1, the collect buffer, whose cursor may exceed lcap = overflow).  Output:
// keys[q][0..c) sorted, cnts[q] = c = min(total, KP) and bound[q] = the approximate distance
// below which EVERY candidate is in keys[q] (+inf: all admissible candidates are listed).
// A query that holds fewer than k keys goes straight to the fallback.
__global__ __launch_bounds__(256) void knn_merge_kernel(const u64 *__restrict__ skeys, const int *__restrict__ scnt,
                                                        const float *__restrict__ T, int64_t n_q, int S, int lcap,
                                                        int KP, int k, u64 *__restrict__ keys, int *__restrict__ cnts,
                                                        float *__restrict__ bound, int *__restrict__ nflag,
                                                        int *__restrict__ flagged, double *__restrict__ bound_d,
                                                        int *__restrict__ bound_i) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ql = (int64_t)blockIdx.x * 4 + wave;
  if (ql >= n_q) return;
  u64 *buf = reinterpret_cast<u64 *>(smem) + (size_t)wave * S * lcap;
  int c = 0;
  bool overflow = false, all_listed = true;
  for (int sp = 0; sp < S; ++sp) {
    int cs = scnt[ql * S + sp];
    if (cs > lcap) { overflow = true; cs = lcap; }
    if (cs >= KP) all_listed = false;   // a full select list may have dropped candidates
    for (int i = lane; i < cs; i += 64) buf[c + i] = skeys[(ql * S + sp) * (int64_t)lcap + i];
    c += cs;
  }
  const float Tq = T ? T[ql] : INFINITY;   // route 1b: nothing above T was collected
  if (T && Tq < INFINITY) all_listed = false;
  // too few keys: nothing to bound the search with, the query takes the full float64 scan.  An overflowed
  // buffer still yields K' real candidates: stage 2 turns them into an exact (if loose) bound for stage 3,
  // and bound = -inf keeps it from certifying them.
  if (c < k) {
    if (lane == 0) {
      const int pos = atomicAdd(nflag, 1);
      flagged[pos] = (int)ql;
      bound_d[ql] = INFINITY;
      bound_i[ql] = 0x7fffffff;
      cnts[ql] = 0;
    }
    return;
  }
  __builtin_amdgcn_wave_barrier();
  // rank by counting; a lane carries up to four of the wave's entries through ONE sweep over the list (a broadcast LDS
  // read per list entry serves all of them: at k = 50 a query holds ~180 entries, three per lane, and the kernel took
  // 0.30 ms per 62.5k queries with one sweep per entry)
  auto sweep = [&](auto ne, int j0) {
    constexpr int NE = decltype(ne)::value;
    u64 e[NE];
    int rk[NE];
#pragma unroll
    for (int m = 0; m < NE; ++m) {
      const int j = j0 + m * 64 + lane;
      e[m] = j < c ? buf[j] : ~0ull;
      rk[m] = 0;
    }
#pragma unroll 2
    for (int i = 0; i < c; ++i) {
      const u64 b = buf[i];
#pragma unroll
      for (int m = 0; m < NE; ++m) rk[m] += (b < e[m]) ? 1 : 0;
    }
#pragma unroll
    for (int m = 0; m < NE; ++m) {
      if (j0 + m * 64 + lane < c) {
        if (rk[m] < KP) keys[ql * KP + rk[m]] = e[m];
        if (rk[m] == KP - 1) bound[ql] = overflow ? -INFINITY : fminf(ordered_to_f32((unsigned)(e[m] >> 32)), Tq);
      }
    }
  };
  for (int j0 = 0; j0 < c; j0 += 256) {
    const int left = c - j0;   // wave-uniform
    if (left > 192) sweep(std::integral_constant<int, 4>{}, j0);
    else if (left > 128) sweep(std::integral_constant<int, 3>{}, j0);
    else if (left > 64) sweep(std::integral_constant<int, 2>{}, j0);
    else sweep(std::integral_constant<int, 1>{}, j0);
  }
  if (lane == 0) {
    cnts[ql] = c < KP ? c : KP;
    if (c < KP) bound[ql] = overflow ? -INFINITY : (all_listed ? INFINITY : Tq);
  }
}

// ---- exact float64 distance, bit-for-bit the oracle's expression ------------------
__device__ __forceinline__ double exact_dist(const double *__restrict__ xq, const float *__restrict__ xc, int d) {
  double s = 0.0;
  for (int t = 0; t < d; ++t) {
    const double diff = __dsub_rn(xq[t], (double)xc[t]);
    s = __dadd_rn(s, __dmul_rn(diff, diff));
  }
  return s;
}

// the same sum in the same order with the candidate's row read 16 bytes at a time (d % 4 == 0, x 16-byte aligned): a lane
// walks its own row, so every load instruction of the scalar form touches 64 different cache lines for 4 bytes each
__device__ __forceinline__ double exact_dist_vec4(const double *__restrict__ xq, const float *__restrict__ xc, int d) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  double s = 0.0;
  for (int t = 0; t < d; t += 4) {
    const f32x4 c = *reinterpret_cast<const f32x4 *>(xc + t);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double diff = __dsub_rn(xq[t + u], (double)c[u]);
      s = __dadd_rn(s, __dmul_rn(diff, diff));
    }
  }
  return s;
}

__device__ __forceinline__ bool key_less(double da, int ia, double db, int ib) {
  return da < db || (da == db && ia < ib);
}

// ---- stage 2: rerank + certificate ----------------------------------------------------
// one wave per query; dynamic LDS per query: double xq[d], double dd[KP], int ii[KP].  HALF (KP <= 32, i.e. k <= 18: every
// BASELINE config but configs[4]): a query has at most 32 survivors, one per lane, so a wave serves TWO queries with 32
// lanes each instead of leaving half of itself idle through both phases; a query's arithmetic does not change.
template <bool HALF>
__global__ __launch_bounds__(256) void knn_rerank_kernel(
    const float *__restrict__ x, const float *__restrict__ norms, const unsigned *__restrict__ maxnorm_bits,
    int d, int64_t q_begin, int64_t n_q, const u64 *__restrict__ keys, const int *__restrict__ cnts,
    const float *__restrict__ bound, int KP, int k, double eps_coef, int force_flag,
    int32_t *__restrict__ out_idx, float *__restrict__ out_dist,
    int *__restrict__ nflag, int *__restrict__ flagged, double *__restrict__ bound_d,
    int *__restrict__ bound_i) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int W = HALF ? 32 : 64;   // lanes per query
  const int slot = threadIdx.x / W, lane = threadIdx.x % W;
  const int64_t ql = (int64_t)blockIdx.x * (256 / W) + slot;
  if (ql >= n_q) return;
  const size_t per_wave = sizeof(double) * (size_t)(d + KP) + sizeof(int) * (size_t)KP;
  char *base = smem + (size_t)slot * ((per_wave + 15) / 16 * 16);
  double *xq = reinterpret_cast<double *>(base);
  double *dd = xq + d;
  int *ii = reinterpret_cast<int *>(dd + KP);
  const int64_t q = q_begin + ql;
  for (int t = lane; t < d; t += W) xq[t] = (double)x[q * d + t];
  __builtin_amdgcn_wave_barrier();   // (a query's lanes sit in one wave, whose LDS accesses complete in order)
  const int c = cnts[ql];
  const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  for (int j = lane; j < c; j += W) {
    const int cand = (int)(unsigned)(keys[ql * KP + j] & 0xffffffffull);
    dd[j] = vec ? exact_dist_vec4(xq, x + (int64_t)cand * d, d) : exact_dist(xq, x + (int64_t)cand * d, d);
    ii[j] = cand;
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < c; j += W) {
    const double dj = dd[j];
    const int ij = ii[j];
    int rk = 0;
    for (int i = 0; i < c; ++i) rk += key_less(dd[i], ii[i], dj, ij) ? 1 : 0;
    if (rk < k) {
      out_idx[ql * k + rk] = ij;
      out_dist[ql * k + rk] = (float)dj;
    }
    if (rk == k - 1) {
      // every candidate outside keys[ql] has approximate distance >= bound[ql]
      const float tau = bound[ql];
      const double eps = eps_coef * ((double)norms[q] + (double)__uint_as_float(*maxnorm_bits));
      const bool certified = tau == INFINITY || dj < (double)tau - eps;
      if (!certified || force_flag) {
        const int pos = atomicAdd(nflag, 1);
        flagged[pos] = (int)ql;
        bound_d[ql] = force_flag ? INFINITY : dj;
        bound_i[ql] = force_flag ? 0x7fffffff : ij;
      }
    }
  }
}

// ---- stage 3: float64 fallback scan ------------------------------------------------------
constexpr int FB_THREADS = 256;
constexpr int FB_CAP = 2048;
constexpr int FB_PER_THREAD = FB_CAP / FB_THREADS;

// MODE 0: one workgroup per flagged query, scanning its prefilter list; a query WITHOUT a usable list (no prefilter,
//         or more than list_cap candidates under its bound) is appended to full_list and left to mode 1.
// MODE 1: queries of full_list scan ALL candidates, each split over SP workgroups (candidate ranges of `chunk`):
//         SP == 1 writes the final neighbours, SP > 1 the k best of the range to part_d / part_i.
// MODE 2: merges the SP partial lists of a full-scan query.  (One workgroup walking 500k candidates alone took
//         6.4 ms for a single uncertified query of a k = 50 search; split 40 ways it is 0.2 ms.)
template <int MODE>
__global__ __launch_bounds__(FB_THREADS) void knn_exact_scan_kernel(
    const float *__restrict__ x, int64_t N, int d, int64_t q_begin, const int *__restrict__ flagged,
    const double *__restrict__ bound_d, const int *__restrict__ bound_i, int k, int include_self,
    int32_t *__restrict__ out_idx, float *__restrict__ out_dist, const int *__restrict__ list,
    const int *__restrict__ list_cnt, int list_cap, int *__restrict__ n_full, int *__restrict__ full_list, int SP,
    int64_t chunk, double *__restrict__ part_d, int *__restrict__ part_i) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *bd = reinterpret_cast<double *>(smem);         // [FB_CAP]
  double *xq = bd + FB_CAP;                              // [d]
  double &s_bound_d = xq[d];
  int *bi = reinterpret_cast<int *>(xq + d + 1);         // [FB_CAP]
  int &s_cnt = bi[FB_CAP];
  int &s_bound_i = bi[FB_CAP + 1];

  const int tid = threadIdx.x;
  const int fslot = MODE == 0 ? (int)blockIdx.x : full_list[MODE == 1 ? blockIdx.x / SP : blockIdx.x];   // index into flagged[]
  const int sp = MODE == 1 ? (int)(blockIdx.x % SP) : 0;
  const int64_t ql = flagged[fslot];
  const int64_t q = q_begin + ql;

  const int *my_list = nullptr;
  int64_t c_lo = 0, c_hi = N;
  if (MODE == 0) {
    const int lc = list ? list_cnt[fslot] : list_cap + 1;
    if (lc > list_cap) {                                 // uniform: no usable list -> full scan, mode 1
      if (tid == 0) full_list[atomicAdd(n_full, 1)] = fslot;
      return;
    }
    my_list = list + (int64_t)fslot * list_cap;
    c_hi = lc;
  } else if (MODE == 1) {
    c_lo = sp * chunk;
    c_hi = c_lo + chunk < N ? c_lo + chunk : N;
  }

  for (int t = tid; t < d; t += FB_THREADS) xq[t] = (double)x[q * d + t];
  if (tid == 0) {
    s_cnt = 0;
    s_bound_d = MODE == 2 ? (double)INFINITY : bound_d[ql];
    s_bound_i = MODE == 2 ? 0x7fffffff : bound_i[ql];
  }
  __syncthreads();

  auto compact = [&](bool final_pass) {
    // rank-by-counting over the buffer; keep the k smallest keys
    const int c = s_cnt;
    double ed[FB_PER_THREAD];
    int ei[FB_PER_THREAD], rk[FB_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FB_PER_THREAD; ++j) {
      const int idx = tid + FB_THREADS * j;
      ed[j] = idx < c ? bd[idx] : INFINITY;
      ei[j] = idx < c ? bi[idx] : 0x7fffffff;
      rk[j] = 0;
    }
    for (int i = 0; i < c; ++i) {
      const double di = bd[i];
      const int ix = bi[i];
#pragma unroll
      for (int j = 0; j < FB_PER_THREAD; ++j)
        if (FB_THREADS * j < c) rk[j] += key_less(di, ix, ed[j], ei[j]) ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FB_PER_THREAD; ++j) {
      const int idx = tid + FB_THREADS * j;
      if (idx < c && rk[j] < k) {
        if (final_pass) {
          if (MODE == 1 && SP > 1) {
            part_d[((int64_t)blockIdx.x) * k + rk[j]] = ed[j];
            part_i[((int64_t)blockIdx.x) * k + rk[j]] = ei[j];
          } else {
            out_idx[ql * k + rk[j]] = ei[j];
            out_dist[ql * k + rk[j]] = (float)ed[j];
          }
        } else {
          bd[rk[j]] = ed[j];
          bi[rk[j]] = ei[j];
          if (rk[j] == k - 1) { s_bound_d = ed[j]; s_bound_i = ei[j]; }
        }
      }
    }
    if (final_pass && MODE == 1 && SP > 1) {
      // ranks the range could not fill (fewer than k candidates at or below the bound) are marked empty
      for (int rnk = (c < k ? c : k) + tid; rnk < k; rnk += FB_THREADS) {
        part_d[((int64_t)blockIdx.x) * k + rnk] = INFINITY;
        part_i[((int64_t)blockIdx.x) * k + rnk] = 0x7fffffff;
      }
    }
    if (tid == 0 && !final_pass) s_cnt = c < k ? c : k;
    __syncthreads();
  };

  if (MODE == 2) {
    // SP * k <= FB_CAP partial entries, already exact distances: one rank-by-counting pass orders them
    const int total = SP * k;
    const int64_t base = (int64_t)blockIdx.x * total;
    for (int t = tid; t < total; t += FB_THREADS) {
      bd[t] = part_d[base + t];
      bi[t] = part_i[base + t];
    }
    if (tid == 0) s_cnt = total;
    __syncthreads();
    compact(true);
    return;
  }

  // with a prefilter list (stage 3a) only the listed candidates are evaluated: the list holds every
  // candidate whose float64 distance can be <= the bound, so the outcome is that of the full scan
  for (int64_t base = c_lo; base < c_hi; base += FB_THREADS) {
    const int64_t pos = base + tid;
    const int64_t cand = pos < c_hi ? (my_list ? (int64_t)my_list[pos] : pos) : N;
    if (cand < N && (include_self || cand != q)) {
      const double dist = exact_dist(xq, x + cand * d, d);
      const double bdv = s_bound_d;
      const int biv = s_bound_i;
      if (dist < bdv || (dist == bdv && (int)cand <= biv)) {
        const int slot = atomicAdd(&s_cnt, 1);
        bd[slot] = dist;
        bi[slot] = (int)cand;
      }
    }
    __syncthreads();
    if (s_cnt > FB_CAP - FB_THREADS) compact(false);  // uniform: s_cnt read after the barrier
  }
  compact(true);
}

// ---- stage 3a: float64 matrix-core prefilter for the queries stage 2 could not certify -------------
// Tight clusters in many dimensions (all neighbours nearly equidistant, far from the centroid) defeat
// the fp32/fp16 certificate for EVERY query, and the exact scan re-reads all candidates per query
// (100k x 100k: 410 ms, bound by 2.5 TB of L2/Infinity-Cache reads).  Here 16 flagged queries share
// each candidate block through v_mfma_f64_16x16x4_f64: D[cand][query] = |c|^2 + |q|^2 - 2 q.c in
// float64 (error ~1e-14 of the norms), and every candidate at or below the query's bound (the exact
// k-th distance stage 2 found, so an upper bound of the true one) plus that error goes to a short
// list; the exact scan then evaluates the oracle's expression on the list only.
// K-slot map (any permutation works as long as A and B agree): step 4u + t, lane group g = lane / 16
// <-> dimension 16u + 4g + t, so a lane's 4 steps of superchunk u are one 16-byte load.
using f64x4 = __attribute__((ext_vector_type(4))) double;
using f32x4v = __attribute__((ext_vector_type(4))) float;

__global__ void knn_norm64_kernel(const float *__restrict__ x, int64_t N, int d, double *__restrict__ norm64,
                                  unsigned long long *__restrict__ max_bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  if (i < N)
    for (int t = 0; t < d; ++t) { const double v = (double)x[i * d + t]; s += v * v; }
  if (i < N) norm64[i] = s;
  // non-negative doubles order like their bit patterns
  unsigned long long b = (i < N && s == s) ? (unsigned long long)__double_as_longlong(s) : 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long ob = __shfl_xor(b, o); b = ob > b ? ob : b; }
  if ((threadIdx.x & 63) == 0) atomicMax(max_bits, b);
}

__device__ __forceinline__ f32x4v ld4_dims(const float *__restrict__ row, int k0, int d, bool vec) {
  f32x4v v = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    if (k0 < d) v = *reinterpret_cast<const f32x4v *>(row + k0);   // d % 4 == 0: the chunk is inside or outside
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (k0 + t < d) v[t] = row[k0 + t];
  }
  return v;
}

template <int ND>
__global__ __launch_bounds__(64) void knn_prefilter64_kernel(
    const float *__restrict__ x, const double *__restrict__ norm64, const unsigned long long *__restrict__ max_bits,
    int64_t N, int d, int vec, int64_t q_begin, const int *__restrict__ flagged, int nflag,
    const double *__restrict__ bound_d, double eps_coef, int include_self, int S, int64_t NBs, int cap,
    int *__restrict__ list, int *__restrict__ list_cnt) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  const int64_t tile = blockIdx.x / S;
  const int split = blockIdx.x % S;
  const int64_t NB = (N + 15) >> 4;
  const int64_t b_lo = split * NBs, b_hi = (b_lo + NBs < NB) ? b_lo + NBs : NB;
  const int64_t f = tile * 16 + j;
  const bool fvalid = f < nflag;
  const int64_t ql = fvalid ? flagged[f] : 0;
  const int64_t q = q_begin + ql;
  // threshold: the bound, widened by what the two float64 evaluations can differ
  double T = -INFINITY;
  if (fvalid) {
    const double b = bound_d[ql];
    if (b < INFINITY) {
      T = b * (1.0 + (double)(d + 4) * 2.220446049250313e-16) +
          eps_coef * (norm64[q] + __longlong_as_double((long long)*max_bits));
    } else if (split == 0 && g == 0) {
      list_cnt[f] = cap + 1;   // no finite bound: this query takes the full scan
    }
  }
  double bq[4 * ND];
#pragma unroll
  for (int u = 0; u < ND; ++u) {
    const f32x4v v = ld4_dims(x + q * d, 16 * u + 4 * g, d, vec != 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) bq[4 * u + t] = fvalid ? -2.0 * (double)v[t] : 0.0;
  }
  const double bn = g == 0 ? 1.0 : (g == 1 ? (fvalid ? norm64[q] : 0.0) : 0.0);

  f32x4v cur[ND], nxt[ND];
  {
    int64_t c = b_lo * 16 + j;
    c = c < N ? c : N - 1;
#pragma unroll
    for (int u = 0; u < ND; ++u) cur[u] = ld4_dims(x + c * d, 16 * u + 4 * g, d, vec != 0);
  }
  for (int64_t b = b_lo; b < b_hi; ++b) {
    int64_t cl = b * 16 + j;
    cl = cl < N ? cl : N - 1;
    {
      int64_t cn = (b + 1 < b_hi ? b + 1 : b) * 16 + j;
      cn = cn < N ? cn : N - 1;
#pragma unroll
      for (int u = 0; u < ND; ++u) nxt[u] = ld4_dims(x + cn * d, 16 * u + 4 * g, d, vec != 0);
    }
    const double an = g == 0 ? norm64[cl] : (g == 1 ? 1.0 : 0.0);
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < ND; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)cur[u][t], bq[4 * u + t], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(an, bn, acc, 0, 0, 0);
    // f64 16x16x4 result layout (probed: tools/experiments/mfma_f64_layout.hip): register r of lane
    // (j, g) is D[4r + g][j] - NOT the 4g + r of the f32 16x16 shapes - i.e. candidate 16b + 4r + g
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t cand = b * 16 + 4 * r + g;
      if (acc[r] <= T && cand < N && (include_self || cand != q)) {
        const int pos = atomicAdd(&list_cnt[f], 1);
        if (pos < cap) list[f * cap + pos] = (int)cand;
      }
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) cur[u] = nxt[u];
  }
}

// ---- host orchestration ----------------------------------------------------------------------
// All state of a search lives in a workspace handle: grow-only device scratch kept across calls
// (hipMalloc/hipFree per call cost ~10 ms of the 100k-cell step), the route options and the statistics of
// the last call.  cc_knn_create / cc_knn_free own one; the convenience entry points (cc_knn_l2, cc_knn_l2_dev,
// cc_knn_l2_sharded) use the library's default workspace, which cc_shutdown releases.
struct KnnWorkspace {
  DevBuf xin, P, P16, norms, maxn, keys, cnts, bound, skeys, scnts, smin, Tq, cbuf, ccur, oidx, odist, nflag, flagged,
      bnd_d, bnd_i, partial, mu, norm64, max64, plist, pcnt, nfull, full_list, part_d, part_i;
  bool force_fallback = false;             // test hook: every query through the float64 stage
  bool force_route_a = false;              // test hook: always take the full-select route
  bool use_fp16_select = CC_KNN_SELECT16 != 0;   // sample pass of stage 1b on the fp16 matrix cores too
  bool use_fp16_collect = true;            // stage 1b collect on the fp16 matrix cores when the data's scale allows
  bool use_fp64_prefilter = true;          // stage 3: float64 matrix-core prefilter in front of the exact scan
  bool use_coarse_collect = true;          // stage 1b collect on the fp16 tile + direct fp32 re-evaluation of the parked pairs; off: fp32 list collect
  int collect_qw = 0;                      // 0 = chosen per call; 1 / 2 = 32-query groups per wave of the coarse collect
  int collect_norm_form = 0;               // 0 = chosen per call; 1 = norm K-step (sign test); 2 = norms through the accumulator
  int collect_splits = 0;                  // 0 = chosen per call; > 0 = forced (tools/knn_split_sweep.sh)
  int sample_splits = 0;                   // same for the candidate ranges of the fp16 sample pass
  int sample_stride = 0;                   // 0 = CC_KNN_SAMPLE; > 0 = every n-th candidate block is sampled
  int sample_qw = 0;                       // 0 = chosen per call; 1 / 2 = query groups per wave of the fp16 sample pass
  int64_t last_fallback = 0;
  int64_t last_full_scan = 0;              // fallback queries that had to scan every candidate in float64 VALU
  int64_t last_sample_stride = 0;          // sampling stride of the last search that took route 1b with a sample (else 0)
  int64_t last_norm_form = 0;              // tile of the last coarse collect: 1 = norm K-step, 2 = norms through the accumulator (0: no coarse collect ran)
  // pinned bounce buffer for results that go to PAGEABLE caller memory (copy_results)
  void *pin = nullptr;
  size_t pin_cap = 0;
  hipEvent_t pin_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  void release() {
    if (pin) (void)hipHostFree(pin);
    pin = nullptr;
    pin_cap = 0;
    for (hipEvent_t &ev : pin_ev) {
      if (ev) (void)hipEventDestroy(ev);
      ev = nullptr;
    }
    DevBuf *all[] = {&xin, &P, &P16, &norms, &maxn, &keys, &cnts, &bound, &skeys, &scnts, &smin, &Tq, &cbuf, &ccur, &oidx, &odist,
                     &nflag, &flagged, &bnd_d, &bnd_i, &partial, &mu, &norm64, &max64, &plist, &pcnt, &nfull, &full_list, &part_d,
                     &part_i};
    for (DevBuf *b : all) b->release();
  }
};

static KnnWorkspace *g_default_ws = nullptr;
KnnWorkspace *knn_default_ws() {
  if (!g_default_ws) g_default_ws = new KnnWorkspace();
  return g_default_ws;
}
// cc_shutdown: the device scratch goes, the options set through cc_knn_set_option(NULL, ..) stay
void knn_default_ws_release() {
  if (g_default_ws) g_default_ws->release();
}

struct SelectArgs {
  const float4 *P; const float *norms, *x, *mu;
  int64_t N; int d; int64_t NB, q_begin, n_q; int include_self;
};

static int run_select(int NG, const SelectArgs &a, int KP, int CAP, int S, int64_t NBs, int bstride, u64 *keys,
                      int *cnts) {
  const int64_t nwaves = ceil_div(a.n_q, 32) * S;
  const size_t lds = sizeof(u64) * 32 * (size_t)(CAP + 1) + 32 * 8;
#define CC_SEL(NGV)                                                                                          \
  case NGV:                                                                                                  \
    hipLaunchKernelGGL(knn_select_kernel<NGV>, dim3((unsigned)nwaves), dim3(64), lds, stream(), a.P, a.norms, \
                       a.x, a.mu, a.N, a.d, a.NB, a.q_begin, a.n_q, a.include_self, KP, CAP, S, NBs, bstride, \
                       keys, cnts);                                                                          \
    break;
  switch (NG) {
    CC_SEL(1) CC_SEL(2) CC_SEL(3) CC_SEL(5) CC_SEL(9) CC_SEL(17)
    default: CC_REQUIRE(false, "knn: no kernel for NG=%d", NG);
  }
#undef CC_SEL
  CC_HIP(hipGetLastError());
  return CC_OK;
}

static int run_sample16(int ND, int NSET, int QW, const h8 *P16, const SelectArgs &a, const unsigned *maxnorm_bits, float s,
                        int S, int64_t NBs, int bstride, unsigned *out_key) {
  const int64_t nwaves = ceil_div(a.n_q, 32 * QW * SMP_WAVES) * S;   // workgroups of SMP_WAVES waves
#define CC_SMP16(NDV, NS, QWV)                                                                                    \
  hipLaunchKernelGGL((knn_sample16_kernel<NDV, NS, QWV>), dim3((unsigned)nwaves), dim3(64 * SMP_WAVES), 0, stream(), P16, a.norms, \
                     maxnorm_bits, a.x, a.mu, a.N, a.d, a.NB, a.q_begin, a.n_q, a.include_self, s, S, NBs, bstride, out_key)
#define CC_SMP16_ND(NS, QWV)                                                                                      \
  switch (ND) {                                                                                                   \
    case 1: CC_SMP16(1, NS, QWV); break;                                                                          \
    case 2: CC_SMP16(2, NS, QWV); break;                                                                          \
    case 4: CC_SMP16(4, NS, QWV); break;                                                                          \
    default: CC_REQUIRE(false, "knn: no fp16 sample kernel for ND=%d", ND);                                       \
  }
  if (NSET == 2 && QW == 2) { CC_SMP16_ND(2, 2) }
  else if (NSET == 2) { CC_SMP16_ND(2, 1) }
  else { CC_SMP16_ND(4, 1) }
#undef CC_SMP16_ND
#undef CC_SMP16
  CC_HIP(hipGetLastError());
  return CC_OK;
}

static int run_collect(int NG, const SelectArgs &a, const float *T, int S, int64_t NBs, int cap, u64 *buf,
                       int *cursor) {
  const int64_t nwg = ceil_div(a.n_q, 128) * S;
#define CC_COL(NGV)                                                                                           \
  case NGV: {                                                                                                 \
    const size_t lds = sizeof(float4) * 2 * COL_CB * NGV * 64 + (sizeof(u64) + sizeof(int)) * 4 * COL_WB;     \
    if (lds > 64 * 1024)                                                                                      \
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_collect_kernel<NGV>),                    \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
    hipLaunchKernelGGL(knn_collect_kernel<NGV>, dim3((unsigned)nwg), dim3(256), lds, stream(), a.P, a.norms,  \
                       a.x, a.mu, a.N, a.d, a.NB, a.q_begin, a.n_q, a.include_self, T, S, NBs, cap, buf,      \
                       cursor);                                                                               \
    break;                                                                                                    \
  }
  switch (NG) {
    CC_COL(1) CC_COL(2) CC_COL(3) CC_COL(5) CC_COL(9) CC_COL(17)
    default: CC_REQUIRE(false, "knn: no kernel for NG=%d", NG);
  }
#undef CC_COL
  CC_HIP(hipGetLastError());
  return CC_OK;
}

// expected pairs of a 32 x 32 tile under the threshold: ~cap / 4 collected per query out of N candidates
static bool collect16c_hits_are_rare(int cap, int64_t N) { return (1024.0 * (double)(cap / 4) / (double)N) < 0.5; }

// norm_form: 0 = the norms ride in a K-step of their own (sign test), 1 = through the accumulator (C16C CN)
static int run_collect16c(int ND, int QW, int norm_form, const h8 *P16, const SelectArgs &a,
                          const unsigned *maxnorm_bits, const float *T, float s, int S, int64_t NBs, int cap, u64 *buf,
                          int *cursor) {
  const int64_t nwg = ceil_div(a.n_q, 128 * QW) * S;
  const int min_first = collect16c_hits_are_rare(cap, a.N) ? 1 : 0;
#define CC_COL16C_L(NDV, QWV, CNV, NWV, CBMV, MF)                                                                    \
  do {                                                                                                               \
    const size_t lds = C16C<NDV, CNV, NWV, CBMV>::lds_bytes();                                                       \
    if (lds > 64 * 1024)                                                                                             \
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_collect16c_kernel<NDV, QWV, CNV, NWV, CBMV>),    \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    hipLaunchKernelGGL((knn_collect16c_kernel<NDV, QWV, CNV, NWV, CBMV>), dim3((unsigned)nwg), dim3(64 * NWV), lds,   \
                       stream(), P16, a.norms, maxnorm_bits, a.x, a.mu, a.N, a.d, (int)a.NB, a.q_begin, a.n_q,       \
                       a.include_self, T, s, S, (int)NBs, cap, buf, cursor, MF);                                     \
  } while (0)
#define CC_COL16C(NDV, QWV)                                      \
  if (norm_form) CC_COL16C_L(NDV, QWV, true, 4, 1, 1);           \
  else CC_COL16C_L(NDV, QWV, false, 4, 1, min_first)
  if (QW == 2) {
    switch (ND) {
      case 1: CC_COL16C(1, 2); break;
      case 2: CC_COL16C(2, 2); break;
      case 4: CC_COL16C(4, 2); break;
      default: CC_REQUIRE(false, "knn: no two-group coarse collect kernel for ND=%d", ND);
    }
  } else {
    switch (ND) {
      case 1: CC_COL16C(1, 1); break;
      case 2: CC_COL16C(2, 1); break;
      case 4: CC_COL16C(4, 1); break;
      case 8: CC_COL16C(8, 1); break;
      default: CC_REQUIRE(false, "knn: no coarse fp16 collect kernel for ND=%d", ND);
    }
  }
#undef CC_COL16C_L
#undef CC_COL16C
  CC_HIP(hipGetLastError());
  return CC_OK;
}

static void make_splits(int64_t NB, int want, int *S, int64_t *NBs) {
  int s = want < 1 ? 1 : want;
  while (s > 1 && NB / s < 16) --s;
  int64_t nbs = ceil_div(NB, s);
  while (s > 1 && (int64_t)(s - 1) * nbs >= NB) { --s; nbs = ceil_div(NB, s); }
  *S = s;
  *NBs = nbs;
}

// Results to the caller's host arrays.  Pinned destinations (the pipeline's buffers) take the DMA directly.  A PAGEABLE
// destination made hipMemcpyAsync stage the copy itself at ~5 GB/s — 2-3 ms per 100k x 15 search outside any kernel
// (knn_bench wall 5.6 ms against 3.6 ms of kernels).  Here such results cross the link into a pinned bounce
// buffer of the workspace in four pieces, and the host copies piece i to the caller while pieces i+1.. are in flight.
static bool host_ptr_is_pinned(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();   // unregistered host memory: clear the sticky error
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}
// plain pageable host memory, i.e. nothing the HIP runtime knows (managed or device allocations are left alone)
static bool host_ptr_is_plain(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return true;
  }
  return attr.type == hipMemoryTypeUnregistered;
}

// Fresh pageable result arrays fault their pages in on the first store — ~0.25 us per 4 KiB page, a large part of the
// host copy below when the caller allocates new arrays per call (numpy does).  knn_run populates them while the search
// kernels run: one madvise(MADV_POPULATE_WRITE) per array (contents untouched), or a same-value store per page where
// the kernel predates that advice.
static void host_populate(void *p, size_t bytes) {
  if (!p || !bytes) return;
  const uintptr_t page = 4096, lo = reinterpret_cast<uintptr_t>(p) & ~(page - 1),
                  hi = (reinterpret_cast<uintptr_t>(p) + bytes + page - 1) & ~(page - 1);
#ifdef MADV_POPULATE_WRITE
  if (madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_WRITE) == 0) return;
#endif
  volatile char *c = static_cast<volatile char *>(p);
  for (size_t off = 0; off < bytes; off += page) c[off] = c[off];
  c[bytes - 1] = c[bytes - 1];
}

static int copy_results(KnnWorkspace &ws, const void *d_idx, const void *d_dist, size_t bytes_each, void *h_idx, void *h_dist) {
  if (host_ptr_is_pinned(h_idx) && host_ptr_is_pinned(h_dist)) {
    CC_HIP(hipMemcpyAsync(h_idx, d_idx, bytes_each, hipMemcpyDeviceToHost, stream()));
    CC_HIP(hipMemcpyAsync(h_dist, d_dist, bytes_each, hipMemcpyDeviceToHost, stream()));
    CC_HIP(hipStreamSynchronize(stream()));
    return CC_OK;
  }
  if (ws.pin_cap < 2 * bytes_each) {
    if (ws.pin) (void)hipHostFree(ws.pin);
    ws.pin = nullptr;
    ws.pin_cap = 0;
    CC_HIP(hipHostMalloc(&ws.pin, 2 * bytes_each, hipHostMallocDefault));
    ws.pin_cap = 2 * bytes_each;
  }
  for (hipEvent_t &ev : ws.pin_ev)
    if (!ev) CC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  char *pin_idx = static_cast<char *>(ws.pin), *pin_dist = pin_idx + bytes_each;
  const size_t half = (bytes_each / 2 + 63) / 64 * 64 < bytes_each ? (bytes_each / 2 + 63) / 64 * 64 : bytes_each;
  struct Piece { char *pin; const char *dev; char *dst; size_t off, len; };
  const Piece pieces[4] = {{pin_idx, static_cast<const char *>(d_idx), static_cast<char *>(h_idx), 0, half},
                           {pin_dist, static_cast<const char *>(d_dist), static_cast<char *>(h_dist), 0, half},
                           {pin_idx, static_cast<const char *>(d_idx), static_cast<char *>(h_idx), half, bytes_each - half},
                           {pin_dist, static_cast<const char *>(d_dist), static_cast<char *>(h_dist), half, bytes_each - half}};
  for (int i = 0; i < 4; ++i) {
    if (pieces[i].len) CC_HIP(hipMemcpyAsync(pieces[i].pin + pieces[i].off, pieces[i].dev + pieces[i].off, pieces[i].len,
                                             hipMemcpyDeviceToHost, stream()));
    CC_HIP(hipEventRecord(ws.pin_ev[i], stream()));
  }
  // (the host copy is what is left: 12 MB per 100k x 15 search, 1.2 ms with fresh destination pages faulting in; four
  // copying threads made it SLOWER — 5.6 against 4.7 ms per call — and were taken out again)
  for (int i = 0; i < 4; ++i) {
    CC_HIP(hipEventSynchronize(ws.pin_ev[i]));
    if (pieces[i].len) memcpy(pieces[i].dst + pieces[i].off, pieces[i].pin + pieces[i].off, pieces[i].len);
  }
  return CC_OK;
}

int knn_run(KnnWorkspace *wsp, const float *d_x, int64_t N, int d, int64_t q_begin, int64_t n_q, int k,
            int include_self, int32_t *h_idx, float *h_dist) {
  KnnWorkspace &ws = wsp ? *wsp : *knn_default_ws();
  ws.last_fallback = 0;
  ws.last_full_scan = 0;
  ws.last_sample_stride = 0;
  ws.last_norm_form = 0;
  CC_REQUIRE(N > 0 && N < (int64_t)2147483000, "knn: n_total=%lld out of range", (long long)N);
  CC_REQUIRE(d >= 1 && d <= 132, "knn: d=%d unsupported (1..132)", d);
  CC_REQUIRE(k >= 1 && k <= 128, "knn: k=%d unsupported (1..128)", k);
  CC_REQUIRE(q_begin >= 0 && n_q >= 0 && q_begin + n_q <= N, "knn: query range [%lld,+%lld) outside n_total=%lld",
             (long long)q_begin, (long long)n_q, (long long)N);
  CC_REQUIRE((int64_t)k <= N - (include_self ? 0 : 1),
             "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", k,
             (long long)(N - (include_self ? 0 : 1)));
  if (n_q == 0) return CC_OK;
  CC_REQUIRE(d_x && h_idx && h_dist, "knn: NULL %s", !d_x ? "points" : (!h_idx ? "index output" : "distance output"));

  static const int ng_opts[] = {1, 2, 3, 5, 9, 17};
  int NG = 0;
  for (int o : ng_opts)
    if (8 * o >= d + 2) { NG = o; break; }
  static const int kp_opts[] = {32, 64, 96, 128, 160};
  int KP = 0;
  for (int o : kp_opts)
    if (o >= k + 14) { KP = o; break; }
  const int64_t NB = ceil_div(N, 32);
  const int64_t tiles = ceil_div(n_q, 32);

  // Route 1b (sampled threshold + collect) is the product route; route 1a (full select with a running
  // threshold) remains for inputs below 8 sampled blocks that do not fit the collect buffer (none with
  // the default knobs) and as the cross-check behind cc_knn_force_full_select.  Sample = every S-th candidate block
  // (S = 3..8, below), T = (about) the m-th best of the sample, m = max(16, 3k/S): the number of candidates below T
  // is ~Gamma(m)·S, i.e. S·m >= 3k on average, >= k and <= 4·S·m except with probability < 1e-4 per query; either
  // miss only sends that query to the float64 fallback.
#ifndef CC_KNN_SAMPLE
#define CC_KNN_SAMPLE 8
#endif
#ifndef CC_KNN_MIN_M
#define CC_KNN_MIN_M 16   // m >= k for k <= 16, so the sampled points alone fill the list
#endif
  // Sampling stride.  The sample costs ~N/stride tiles per query, the collect parks ~1.15·stride·m pairs per query:
  // the sum is smallest near stride ∝ sqrt(N/m) (measured with the slot-minima sample, CC_KNN_OPT_SAMPLE_STRIDE
  // sweeps: 100k candidates, k = 15: 4.13 / 3.64 / 3.43 / 3.36 / 4.67 ms at stride 8 / 6 / 4 / 3 / 2; 1M: 24.1 / 24.5 /
  // 25.7 / 27.7 / 35.2; 500k, k = 50: 8.2 / 8.0 / 8.9 / 9.4 / 11.1).  The list-based fp32 sample keeps every 8th block.
  int SAMPLE = CC_KNN_SAMPLE;
  if (ws.sample_stride > 0) {
    SAMPLE = ws.sample_stride;
  } else if (ws.use_fp16_select && ws.use_fp16_collect && d <= 64) {
    const int m8 = std::max(CC_KNN_MIN_M, (3 * k + 7) / 8);
    const long st = lround(sqrt((double)N / (double)m8) / 22.6);
    // ... but m = 3k / stride has to stay well below the 128 slots a query has per candidate range (m = 100 of 128
    // would put the threshold at rank 195: k = 100 at 100k candidates 7.4 ms with stride 8, 8.4 with 3)
    SAMPLE = (int)std::min(8L, std::max(std::max(3L, st), (long)ceil_div(3 * k, 40)));
  }
  const int m_min = CC_KNN_MIN_M, m_k = (3 * k + SAMPLE - 1) / SAMPLE;   // SAMPLE·m >= 3k expected below T
  const int m = m_k > m_min ? m_k : m_min;
  // Route 1a pays ~50 us per block while its lists fill (every block overflows them), 1-4 ms even at
  // N = 1000; 1b has no such phase, so it takes everything from 8 sampled blocks up, and inputs that
  // fit the collect buffer whole skip the sample (T = +inf: every candidate is listed).
#ifndef CC_KNN_MIN_SBLK
#define CC_KNN_MIN_SBLK 1
#endif
  const int col_cap = 4 * SAMPLE * m;   // collect buffer entries per query
#ifndef CC_KNN_SAMPLED_MAXK
#define CC_KNN_SAMPLED_MAXK 128
#endif
  const bool collect_all = !ws.force_route_a && k <= CC_KNN_SAMPLED_MAXK && N <= col_cap;
  const bool sampled = collect_all || (!ws.force_route_a && NB >= CC_KNN_MIN_SBLK * SAMPLE &&
                                       N >= (int64_t)CC_KNN_MIN_SBLK * SAMPLE * m && k <= CC_KNN_SAMPLED_MAXK);
  if (sampled && !collect_all) ws.last_sample_stride = SAMPLE;

  DevBuf &P = ws.P, &P16 = ws.P16, &norms = ws.norms, &maxn = ws.maxn, &keys = ws.keys, &cnts = ws.cnts,
         &bound = ws.bound, &skeys = ws.skeys, &scnts = ws.scnts, &smin = ws.smin, &Tq = ws.Tq, &cbuf = ws.cbuf, &ccur = ws.ccur,
         &oidx = ws.oidx, &odist = ws.odist, &nflag = ws.nflag, &flagged = ws.flagged, &bnd_d = ws.bnd_d,
         &bnd_i = ws.bnd_i, &partial = ws.partial, &mu = ws.mu;
  int rc;
  if ((rc = norms.alloc(sizeof(float) * (size_t)N))) return rc;
  if ((rc = maxn.alloc(16))) return rc;
  if ((rc = partial.alloc(sizeof(double) * (size_t)MEAN_BLOCKS * (size_t)d))) return rc;
  if ((rc = mu.alloc(sizeof(float) * (size_t)d))) return rc;
  if ((rc = keys.alloc(sizeof(u64) * (size_t)(n_q * KP)))) return rc;
  if ((rc = cnts.alloc(sizeof(int) * (size_t)n_q))) return rc;
  if ((rc = bound.alloc(sizeof(float) * (size_t)n_q))) return rc;
  if ((rc = oidx.alloc(sizeof(int32_t) * (size_t)(n_q * k)))) return rc;
  if ((rc = odist.alloc(sizeof(float) * (size_t)(n_q * k)))) return rc;
  if ((rc = nflag.alloc(16))) return rc;
  if ((rc = flagged.alloc(sizeof(int) * (size_t)n_q))) return rc;
  if ((rc = bnd_d.alloc(sizeof(double) * (size_t)n_q))) return rc;
  if ((rc = bnd_i.alloc(sizeof(int) * (size_t)n_q))) return rc;
  CC_HIP(hipMemsetAsync(maxn.p, 0, 16, stream()));
  CC_HIP(hipMemsetAsync(nflag.p, 0, 16, stream()));
  CC_HIP(hipMemsetAsync(oidx.p, 0xff, sizeof(int32_t) * (size_t)(n_q * k), stream()));

  {
    ProfScope prof(CC_PROF_KNN_PACK, 0.0);
    const int mean_threads = (int)round_up(d, 64);
    hipLaunchKernelGGL(knn_colsum_kernel, dim3(MEAN_BLOCKS), dim3(mean_threads), 0, stream(), d_x, N, d,
                       partial.as<double>());
    hipLaunchKernelGGL(knn_mean_kernel, dim3(1), dim3(mean_threads, MEAN_PARTS), 0, stream(), partial.as<double>(), N, d,
                       mu.as<float>());
    hipLaunchKernelGGL(knn_norm_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(N, 256), NORM_BLOCKS)), dim3(256), 0,
                       stream(), d_x, mu.as<float>(), N, d, norms.as<float>(), maxn.as<unsigned>());
    CC_HIP(hipGetLastError());
  }
  // the fp32 panels are packed only if an fp32 tile kernel runs (the fp16 route has its own panels)
  SelectArgs sa{nullptr, norms.as<float>(), d_x, mu.as<float>(), N, d, NB, q_begin, n_q, include_self};
  auto pack_fp32_panels = [&]() -> int {
    if (sa.P) return CC_OK;
    int prc;
    if ((prc = P.alloc(sizeof(float4) * (size_t)(NB * NG * 64)))) return prc;
    hipLaunchKernelGGL(knn_pack_kernel, dim3((unsigned)ceil_div(NB * NG * 64, 256)), dim3(256), 0, stream(), d_x,
                       mu.as<float>(), norms.as<float>(), N, d, NG, NB, P.as<float4>());
    CC_HIP(hipGetLastError());
    sa.P = P.as<float4>();
    return CC_OK;
  };
  // fp32 error bound of stage 1 relative to (‖q−μ‖² + ‖c−μ‖²), u = 2^-24:
  //   (2(d+2)+1)·u  d+2 fma terms on |terms| summing to <= 2(‖q‖²+‖c‖²), norms rounded to fp32
  // + 4(1+3u)·u     rounding of the centred coordinates fl(x−μ)
  // < (2d+10)·u; twice that is used.
  double eps_coef = (4.0 * (double)(d + 4) + 8.0) * 5.9604644775390625e-08;
  {
    ProfScope prof(CC_PROF_KNN_SELECT, 2.0 * (double)n_q * (double)N * (double)d);
    int S = 1, lcap = KP;
    int64_t NBs = NB;
    const float *Tptr = nullptr;
    if (sampled) {
      if ((rc = Tq.alloc(sizeof(float) * (size_t)n_q))) return rc;
      // both passes of 1b run on the fp16 matrix cores when the centred data's scale keeps s, s² and the
      // scaled norms well inside the fp32/fp16 ranges; otherwise on the fp32 ones
      int ND16 = 0;
      for (int o : {1, 2, 4, 8})
        if (16 * o >= d) { ND16 = o; break; }
      float scale16 = 0.f;
      if (ws.use_fp16_collect && ND16) {
        unsigned h_max[2] = {0, 0};
        CC_HIP(hipMemcpyAsync(h_max, maxn.p, sizeof h_max, hipMemcpyDeviceToHost, stream()));
        CC_HIP(hipStreamSynchronize(stream()));
        float amax;
        memcpy(&amax, &h_max[1], sizeof amax);
        int e = 0;
        if (std::isfinite(amax) && amax > 0.f) {
          (void)frexpf(amax, &e);
          if (e > -30 && e < 30) scale16 = ldexpf(1.f, 7 - e);   // max|y|·s in [2^6, 2^7)
        }
      }
      if (scale16 > 0.f) {
        // (+ C16C_PAD_BLOCKS blocks of "no candidate" behind the panel: the collect's stages are loaded whole)
        // (and, behind them, the scaled norms once more as plain fp32, + 8 blocks: the accumulator form of the collect
        // loads them 1 KiB at a time)
        const int64_t units = (NB + C16C_PAD_BLOCKS) * (ND16 + 1) * 64;
        const int64_t n_ncs = (NB + C16C_PAD_BLOCKS + 8) * 32;
        if ((rc = P16.alloc(sizeof(h8) * (size_t)units + sizeof(float) * (size_t)n_ncs))) return rc;
        hipLaunchKernelGGL(knn_pack16_kernel, dim3((unsigned)ceil_div(units, 256)), dim3(256), 0, stream(), d_x,
                           mu.as<float>(), norms.as<float>(), N, d, ND16, NB + C16C_PAD_BLOCKS, scale16, P16.as<h8>(),
                           reinterpret_cast<float *>(P16.as<h8>() + units), n_ncs);
      }
      // 1b-i: threshold from the sample (every SAMPLE-th candidate block)
      if (collect_all) {
        hipLaunchKernelGGL(knn_fill_kernel, dim3((unsigned)ceil_div(n_q, 256)), dim3(256), 0, stream(), Tq.as<float>(),
                           n_q, INFINITY);
      } else if (scale16 > 0.f && ws.use_fp16_select && ND16 <= 4 &&   // ND = 8 (d > 64): fp32 lists below
                 ceil_div(ceil_div(NB, SAMPLE), (int64_t)(32 / (m <= 20 ? 2 : 4))) <= (int64_t)(SKEY_MASK + 1)) {
        // slot minima (knn_sample16_kernel + knn_slot_threshold_kernel).  One wave per 32 or 64 queries and candidate
        // range; what counts is how evenly the waves load the 1024 SIMDs: the candidate range is split so that the
        // per-SIMD wave count rounds up by little (3125 whole-range waves would put 4 on some SIMDs and 3 on the
        // rest), at a fixed cost per wave of ~8 tiles; a range holds at most 2^SKEY_BITS sampled tiles.
        const int NSET = m <= 20 ? 2 : 4;
        // 64 queries per wave (two groups sharing each candidate fragment: half the L2 -> register traffic per MFMA)
        // unless the query range is short
        const int QWS = (NSET == 2 && ws.sample_qw != 1 && (ws.sample_qw == 2 || n_q >= 32768)) ? 2 : 1;
        const int64_t stiles = ceil_div(n_q, 32 * QWS);
        int SS = 1;
        int64_t NBss = NB;
        {
          int want = 1;
          if (ws.sample_splits > 0) {
            want = ws.sample_splits;
          } else {
            double best = 1e300;
            for (int cand = 1; cand <= 8; ++cand) {
              const double cost = (double)ceil_div(stiles * cand, 1024) * ((double)ceil_div(NB, (int64_t)cand * SAMPLE) + 8.0);
              if (cost < best * 0.97) { best = cost; want = cand; }
            }
          }
          const int need = (int)ceil_div(ceil_div(NB, SAMPLE), (int64_t)(SKEY_MASK + 1));
          if (want < need) want = need;
          if (want * NSET > 32) want = 32 / NSET;
          make_splits(NB, want, &SS, &NBss);
          if (SS == 1) NBss = NB;
          CC_REQUIRE(ceil_div(NBss, SAMPLE) <= (int64_t)(SKEY_MASK + 1), "knn: %lld sampled tiles per range",
                     (long long)ceil_div(NBss, SAMPLE));
        }
        const int nslots = 32 * NSET * SS;
        if ((rc = smin.alloc(sizeof(unsigned) * (size_t)n_q * (size_t)nslots))) return rc;
        if ((rc = run_sample16(ND16, NSET, QWS, P16.as<h8>(), sa, maxn.as<unsigned>(), scale16, SS, NBss, SAMPLE,
                               smin.as<unsigned>())))
          return rc;
        CC_REQUIRE(nslots % 64 == 0 && nslots <= 16 * 64, "knn: %d sample slots per query", nslots);
#define CC_SLOT_THR(R)                                                                                             \
  hipLaunchKernelGGL(knn_slot_threshold_kernel<R>, dim3((unsigned)ceil_div(n_q, 4)), dim3(256), 0, stream(),       \
                     smin.as<unsigned>(), d_x, d, q_begin, n_q, nslots, m, NSET, NBss, SAMPLE, Tq.as<float>())
        const int key_regs = nslots / 64;
        if (key_regs <= 1) CC_SLOT_THR(1);
        else if (key_regs <= 2) CC_SLOT_THR(2);
        else if (key_regs <= 3) CC_SLOT_THR(3);
        else if (key_regs <= 4) CC_SLOT_THR(4);
        else if (key_regs <= 6) CC_SLOT_THR(6);
        else if (key_regs <= 8) CC_SLOT_THR(8);
        else CC_SLOT_THR(16);
#undef CC_SLOT_THR
      } else {
        // fp32 route: select with m-entry LDS lists over the sampled blocks, T = the m-th best of the sample
#ifndef CC_KNN_SAMPLE_SPARE
#define CC_KNN_SAMPLE_SPARE 32
#endif
        const int KPS = (int)round_up(m, 16), CAPS = KPS + CC_KNN_SAMPLE_SPARE;
        // split over candidate ranges too, so that the (long-lived, list-carrying) waves fill the chip
        int SS = 1;
        int64_t NBss = NB;
        {
          const size_t sl = sizeof(u64) * 32 * (size_t)(CAPS + 1) + 32 * 8;
          int64_t wpc = (int64_t)(160 * 1024 / sl);
          if (wpc > 12) wpc = 12;
          const int want = (int)ceil_div(256 * wpc, tiles);
          make_splits(NB, want > 8 ? 8 : want, &SS, &NBss);
          while (SS > 1 && NBss / SAMPLE < 4 * m) { --SS; NBss = ceil_div(NB, SS); }   // keep >= 4m sampled blocks per range
          if (SS == 1) NBss = NB;
        }
        if ((rc = skeys.alloc(sizeof(u64) * (size_t)(n_q * KPS) * (size_t)SS))) return rc;
        if ((rc = scnts.alloc(sizeof(int) * (size_t)n_q * (size_t)SS))) return rc;
        if ((rc = pack_fp32_panels())) return rc;
        if ((rc = run_select(NG, sa, KPS, CAPS, SS, NBss, SAMPLE, skeys.as<u64>(), scnts.as<int>()))) return rc;
        hipLaunchKernelGGL(knn_threshold_kernel, dim3((unsigned)ceil_div(n_q, 256)), dim3(256), 0, stream(),
                           skeys.as<u64>(), scnts.as<int>(), n_q, SS, KPS, m, Tq.as<float>());
      }
      // 1b-ii: collect over all candidates, split into S candidate ranges.  A multiple of 8 keeps one XCD on one
      // slice of the panels (workgroup i runs on XCD i % 8 and takes range i % S), which stays in that XCD's L2.
      // Workgroups all take about the same time, so what counts is whole rounds over the 768 resident slots
      // (3 per CU): 8 splits of a 100k-query search are 4.07 rounds = 5 (the last one nearly empty), 32 shorter
      // ones 16.3 = 17 quarter-length rounds; fewer, longer splits that fit the rounds can win for short shards
      // even though they lose the XCD affinity (x 1.15, measured with the CC_KNN_OPT_COLLECT_SPLITS override,
      // tools/knn_split_sweep.sh).  Each workgroup pays ~4 blocks' worth of set-up.
      const bool coarse = scale16 > 0.f && ws.use_coarse_collect && !collect_all;
      // 256 queries per workgroup once 8 ranges of them still fill a round
      int QW = (coarse && ND16 <= 4 && ceil_div(n_q, 256) * 8 >= 768) ? 2 : 1;
      if (coarse && ws.collect_qw == 1) QW = 1;
      if (coarse && ws.collect_qw == 2 && ND16 <= 4) QW = 2;
      {
        int want = 8;
        const int forced = ws.collect_splits;
        if (forced > 0) {
          want = forced;
        } else {
          const int64_t wgs = ceil_div(n_q, 128 * QW), slots = 768;
          double best = 1e300;
          for (int cand : {8, 7, 6, 5, 16, 24, 32}) {
            if (cand > 8 && (!coarse || NB / cand < 64)) continue;
            const double cost = (double)ceil_div(wgs * cand, slots) * ((double)ceil_div(NB, cand) + 4.0) *
                                (cand % 8 == 0 ? 1.0 : 1.15);
            if (cost < best * 0.97) { best = cost; want = cand; }
          }
        }
        make_splits(NB, want, &S, &NBs);
      }
      if ((rc = cbuf.alloc(sizeof(u64) * (size_t)n_q * (size_t)col_cap))) return rc;
      if ((rc = ccur.alloc(sizeof(int) * (size_t)n_q))) return rc;
      CC_HIP(hipMemsetAsync(ccur.p, 0, sizeof(int) * (size_t)n_q, stream()));
      if (coarse) {
        // the accumulator form is the kernel of the rare-hit shapes (see C16C)
        const int norm_form = ws.collect_norm_form == 0 ? (ND16 <= 4 && collect16c_hits_are_rare(col_cap, N) ? 1 : 0)
                                                         : ws.collect_norm_form - 1;
        ws.last_norm_form = norm_form + 1;
        if ((rc = run_collect16c(ND16, QW, norm_form, P16.as<h8>(), sa, maxn.as<unsigned>(), Tq.as<float>(),
                                 scale16, S, NBs, col_cap, cbuf.as<u64>(), ccur.as<int>())))
          return rc;
        // The list holds exactly the pairs whose DIRECT fp32 distance is <= T (error ~d·2^-24 of the distance, far inside
        // the fp32 bound above); the certificate keeps the wider bound rounds 1-2 derived for the three-product tile,
        // relative to (‖q−μ‖² + ‖c−μ‖²), u = 2^-24: 16(3·ND+1) fp32-accumulated product terms on |terms| summing to
        // <= 3(‖q‖²+‖c‖²), + 12u, + 4u for the centred coordinates; twice the sum.
        eps_coef = 2.0 * (3.0 * 16.0 * (3.0 * ND16 + 1.0) + 16.0) * 5.9604644775390625e-08;
      } else {
        if ((rc = pack_fp32_panels())) return rc;
        if ((rc = run_collect(NG, sa, Tq.as<float>(), S, NBs, col_cap, cbuf.as<u64>(), ccur.as<int>()))) return rc;
      }
      S = 1;
      lcap = col_cap;
      Tptr = Tq.as<float>();
      hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)ceil_div(n_q, 4)), dim3(256),
                         sizeof(u64) * 4 * (size_t)lcap, stream(), cbuf.as<u64>(), ccur.as<int>(), Tptr, n_q, 1, lcap,
                         KP, k, keys.as<u64>(), cnts.as<int>(), bound.as<float>(), nflag.as<int>(),
                         flagged.as<int>(), bnd_d.as<double>(), bnd_i.as<int>());
    } else {
      // 1a: full select; enough (tile, split) waves for ~4 rounds over the resident slots
      const int CAP = KP + 16;
      const size_t sel_lds = sizeof(u64) * 32 * (size_t)(CAP + 1) + 32 * 8;
      int64_t waves_per_cu = (int64_t)(160 * 1024 / sel_lds);
      if (waves_per_cu > 12) waves_per_cu = 12;
      int want = (int)ceil_div(4 * 256 * waves_per_cu, tiles);
      make_splits(NB, want > 8 ? 8 : want, &S, &NBs);
      if ((rc = skeys.alloc(sizeof(u64) * (size_t)(n_q * KP) * (size_t)S))) return rc;
      if ((rc = scnts.alloc(sizeof(int) * (size_t)n_q * (size_t)S))) return rc;
      if ((rc = pack_fp32_panels())) return rc;
      if ((rc = run_select(NG, sa, KP, CAP, S, NBs, 1, skeys.as<u64>(), scnts.as<int>()))) return rc;
      hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)ceil_div(n_q, 4)), dim3(256),
                         sizeof(u64) * 4 * (size_t)S * (size_t)KP, stream(), skeys.as<u64>(), scnts.as<int>(),
                         (const float *)nullptr, n_q, S, KP, KP, k, keys.as<u64>(), cnts.as<int>(),
                         bound.as<float>(), nflag.as<int>(), flagged.as<int>(), bnd_d.as<double>(),
                         bnd_i.as<int>());
    }
    CC_HIP(hipGetLastError());
  }
  {
    const size_t per_wave = (sizeof(double) * (size_t)(d + KP) + sizeof(int) * (size_t)KP + 15) / 16 * 16;
    ProfScope prof(CC_PROF_KNN_RERANK, 3.0 * (double)n_q * (double)KP * (double)d);
#define CC_RERANK(HALFV, QPB)                                                                                          \
  hipLaunchKernelGGL(knn_rerank_kernel<HALFV>, dim3((unsigned)ceil_div(n_q, QPB)), dim3(256), per_wave * QPB, stream(),  \
                     d_x, norms.as<float>(), maxn.as<unsigned>(), d, q_begin, n_q, keys.as<u64>(), cnts.as<int>(),       \
                     bound.as<float>(), KP, k, eps_coef, ws.force_fallback ? 1 : 0, oidx.as<int32_t>(),                  \
                     odist.as<float>(), nflag.as<int>(), flagged.as<int>(), bnd_d.as<double>(), bnd_i.as<int>())
    if (KP <= 32) CC_RERANK(true, 8);   // two queries per wave
    else CC_RERANK(false, 4);
#undef CC_RERANK
    CC_HIP(hipGetLastError());
  }
  // (before the flag readback: a device-to-host copy into pageable memory — the flag below — blocks in the runtime
  // until the stream has drained, so this is the last point where the host is free while the kernels run)
  if (host_ptr_is_plain(h_idx)) host_populate(h_idx, sizeof(int32_t) * (size_t)(n_q * k));
  if (host_ptr_is_plain(h_dist)) host_populate(h_dist, sizeof(float) * (size_t)(n_q * k));
  int h_nflag = 0;
  CC_HIP(hipMemcpyAsync(&h_nflag, nflag.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
  CC_HIP(hipStreamSynchronize(stream()));
  ws.last_fallback = h_nflag;
  ws.last_full_scan = 0;
  if (h_nflag > 0) {
    const size_t lds = sizeof(double) * (size_t)(FB_CAP + d + 1) + sizeof(int) * (size_t)(FB_CAP + 2);
    ProfScope prof(CC_PROF_KNN_FALLBACK, 3.0 * (double)h_nflag * (double)N * (double)d);
    const int *d_list = nullptr, *d_list_cnt = nullptr;
    const int list_cap = 4096;
    DevBuf &norm64 = ws.norm64, &max64 = ws.max64, &plist = ws.plist, &pcnt = ws.pcnt, &nfull = ws.nfull;
    if ((rc = nfull.alloc(16))) return rc;
    CC_HIP(hipMemsetAsync(nfull.p, 0, 16, stream()));
    if (ws.use_fp64_prefilter && N >= 2048) {
      // 3a: float64 matrix-core prefilter, 16 flagged queries per wave
      int ND64 = 0;
      for (int o : {1, 2, 4, 9})
        if (16 * o >= d) { ND64 = o; break; }
      if ((rc = norm64.alloc(sizeof(double) * (size_t)N))) return rc;
      if ((rc = max64.alloc(16))) return rc;
      if ((rc = plist.alloc(sizeof(int) * (size_t)h_nflag * (size_t)list_cap))) return rc;
      if ((rc = pcnt.alloc(sizeof(int) * (size_t)h_nflag))) return rc;
      CC_HIP(hipMemsetAsync(max64.p, 0, 16, stream()));
      CC_HIP(hipMemsetAsync(pcnt.p, 0, sizeof(int) * (size_t)h_nflag, stream()));
      hipLaunchKernelGGL(knn_norm64_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, stream(), d_x, N, d,
                         norm64.as<double>(), max64.as<unsigned long long>());
      const int64_t NB16 = ceil_div(N, 16), qtiles = ceil_div(h_nflag, 16);
      int S64 = (int)ceil_div(8192, qtiles);              // enough waves for ~2 rounds over the chip
      if (S64 > NB16 / 64) S64 = (int)(NB16 / 64);        // but scans of at least 64 blocks each
      if (S64 < 1) S64 = 1;
      const int64_t NBs64 = ceil_div(NB16, S64);
      S64 = (int)ceil_div(NB16, NBs64);
      // the two float64 evaluations differ by <= (16·ND + d + 8) roundings of terms that sum to <= 2(|q|^2 + |c|^2)
      const double eps64 = 4.0 * (16.0 * ND64 + (double)d + 8.0) * 1.1102230246251565e-16;
      const int vec = (d % 4 == 0 && (reinterpret_cast<uintptr_t>(d_x) & 15) == 0) ? 1 : 0;
#define CC_PF64(NDV)                                                                                              \
  case NDV:                                                                                                       \
    hipLaunchKernelGGL(knn_prefilter64_kernel<NDV>, dim3((unsigned)(qtiles * S64)), dim3(64), 0, stream(), d_x,    \
                       norm64.as<double>(), max64.as<unsigned long long>(), N, d, vec, q_begin, flagged.as<int>(), \
                       h_nflag, bnd_d.as<double>(), eps64, include_self, S64, NBs64, list_cap, plist.as<int>(),    \
                       pcnt.as<int>());                                                                            \
    break;
      switch (ND64) {
        CC_PF64(1) CC_PF64(2) CC_PF64(4) CC_PF64(9)
        default: CC_REQUIRE(false, "knn: no float64 prefilter for d=%d", d);
      }
#undef CC_PF64
      CC_HIP(hipGetLastError());
      d_list = plist.as<int>();
      d_list_cnt = pcnt.as<int>();
    }
    // 3b: the oracle's expression on the listed candidates; queries without a usable list are handed to the
    // split full scan below
    DevBuf &full_list = ws.full_list, &part_d = ws.part_d, &part_i = ws.part_i;
    if ((rc = full_list.alloc(sizeof(int) * (size_t)h_nflag))) return rc;
    hipLaunchKernelGGL(knn_exact_scan_kernel<0>, dim3((unsigned)h_nflag), dim3(FB_THREADS), lds, stream(), d_x, N, d,
                       q_begin, flagged.as<int>(), bnd_d.as<double>(), bnd_i.as<int>(), k, include_self,
                       oidx.as<int32_t>(), odist.as<float>(), d_list, d_list_cnt, list_cap, nfull.as<int>(),
                       full_list.as<int>(), 1, (int64_t)0, (double *)nullptr, (int *)nullptr);
    CC_HIP(hipGetLastError());
    int h_full = 0;
    CC_HIP(hipMemcpyAsync(&h_full, nfull.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    CC_HIP(hipStreamSynchronize(stream()));
    if (h_full > 0) {
      // enough workgroups to fill the chip, ranges of at least 4096 candidates, SP * k entries must fit the merge buffer
      int SP = (int)ceil_div(2048, h_full);
      if (SP > FB_CAP / k) SP = FB_CAP / k;
      if ((int64_t)SP > ceil_div(N, 4096)) SP = (int)ceil_div(N, 4096);
      if (SP < 1) SP = 1;
      const int64_t chunk = ceil_div(N, SP);
      SP = (int)ceil_div(N, chunk);
      if (SP > 1) {
        if ((rc = part_d.alloc(sizeof(double) * (size_t)h_full * (size_t)SP * (size_t)k))) return rc;
        if ((rc = part_i.alloc(sizeof(int) * (size_t)h_full * (size_t)SP * (size_t)k))) return rc;
      }
      hipLaunchKernelGGL(knn_exact_scan_kernel<1>, dim3((unsigned)((int64_t)h_full * SP)), dim3(FB_THREADS), lds, stream(),
                         d_x, N, d, q_begin, flagged.as<int>(), bnd_d.as<double>(), bnd_i.as<int>(), k, include_self,
                         oidx.as<int32_t>(), odist.as<float>(), (const int *)nullptr, (const int *)nullptr, list_cap,
                         nfull.as<int>(), full_list.as<int>(), SP, chunk, part_d.as<double>(), part_i.as<int>());
      if (SP > 1)
        hipLaunchKernelGGL(knn_exact_scan_kernel<2>, dim3((unsigned)h_full), dim3(FB_THREADS), lds, stream(), d_x, N, d,
                           q_begin, flagged.as<int>(), bnd_d.as<double>(), bnd_i.as<int>(), k, include_self,
                           oidx.as<int32_t>(), odist.as<float>(), (const int *)nullptr, (const int *)nullptr, list_cap,
                           nfull.as<int>(), full_list.as<int>(), SP, chunk, part_d.as<double>(), part_i.as<int>());
      CC_HIP(hipGetLastError());
      CC_HIP(hipStreamSynchronize(stream()));
    }
    ws.last_full_scan = h_full;
  }
  return copy_results(ws, oidx.p, odist.p, sizeof(float) * (size_t)(n_q * k), h_idx, h_dist);
}

// The same for points in HOST memory (cc_knn_l2): they cross the link into a grow-only buffer of the workspace (a
// device allocation per call was ~0.3 ms of a 100k-point search).
int knn_run_host(KnnWorkspace *wsp, const float *h_x, int64_t N, int d, int k, int include_self, int32_t *h_idx,
                 float *h_dist) {
  KnnWorkspace &ws = wsp ? *wsp : *knn_default_ws();
  CC_REQUIRE(N > 0 && d > 0, "knn: empty input");
  CC_REQUIRE(h_x != nullptr, "host_x is NULL");
  int rc;
  if ((rc = ws.xin.alloc(sizeof(float) * (size_t)(N * d)))) return rc;
  CC_HIP(hipMemcpyAsync(ws.xin.p, h_x, sizeof(float) * (size_t)(N * d), hipMemcpyHostToDevice, stream()));
  return knn_run(&ws, ws.xin.as<float>(), N, d, 0, N, k, include_self, h_idx, h_dist);
}

}  // namespace cc

extern "C" {

int cc_knn_create(void **knn) {
  CC_REQUIRE(knn != nullptr, "knn is NULL");
  *knn = new cc::KnnWorkspace();
  return CC_OK;
}

void cc_knn_free(void *knn) {
  if (!knn) return;
  cc::KnnWorkspace *ws = static_cast<cc::KnnWorkspace *>(knn);
  ws->release();
  delete ws;
}

int cc_knn_set_option(void *knn, int option, int value) {
  cc::KnnWorkspace *ws = knn ? static_cast<cc::KnnWorkspace *>(knn) : cc::knn_default_ws();
  switch (option) {
    case CC_KNN_OPT_FORCE_FALLBACK: ws->force_fallback = value != 0; break;
    case CC_KNN_OPT_FORCE_FULL_SELECT: ws->force_route_a = value != 0; break;
    case CC_KNN_OPT_FP16_COLLECT: ws->use_fp16_collect = value != 0; break;
    case CC_KNN_OPT_FP16_SELECT: ws->use_fp16_select = value != 0; break;
    case CC_KNN_OPT_FP64_PREFILTER: ws->use_fp64_prefilter = value != 0; break;
    case CC_KNN_OPT_COARSE_COLLECT: ws->use_coarse_collect = value != 0; break;
    case CC_KNN_OPT_COLLECT_SPLITS:
      CC_REQUIRE(value >= 0 && value <= 64, "collect splits %d outside [0, 64]", value);
      ws->collect_splits = value;
      break;
    case CC_KNN_OPT_SAMPLE_SPLITS:
      CC_REQUIRE(value >= 0 && value <= 16, "sample splits %d outside [0, 16]", value);
      ws->sample_splits = value;
      break;
    case CC_KNN_OPT_SAMPLE_GROUPS:
      CC_REQUIRE(value >= 0 && value <= 2, "sample groups %d outside [0, 2]", value);
      ws->sample_qw = value;
      break;
    case CC_KNN_OPT_COLLECT_GROUPS:
      CC_REQUIRE(value >= 0 && value <= 2, "collect groups %d outside [0, 2]", value);
      ws->collect_qw = value;
      break;
    case CC_KNN_OPT_COLLECT_NORM_FORM:
      CC_REQUIRE(value >= 0 && value <= 2, "collect norm form %d outside [0, 2]", value);
      ws->collect_norm_form = value;
      break;
    case CC_KNN_OPT_SAMPLE_STRIDE:
      CC_REQUIRE(value >= 0 && value <= 64, "sample stride %d outside [0, 64]", value);
      ws->sample_stride = value;
      break;
    default: CC_REQUIRE(false, "unknown kNN option %d", option);
  }
  return CC_OK;
}

int cc_knn_stat(void *knn, int stat, int64_t *value) {
  CC_REQUIRE(value != nullptr, "value is NULL");
  cc::KnnWorkspace *ws = knn ? static_cast<cc::KnnWorkspace *>(knn) : cc::knn_default_ws();
  switch (stat) {
    case CC_KNN_STAT_FALLBACK_QUERIES: *value = ws->last_fallback; break;
    case CC_KNN_STAT_FULL_SCAN_QUERIES: *value = ws->last_full_scan; break;
    case CC_KNN_STAT_SAMPLE_STRIDE: *value = ws->last_sample_stride; break;
    case CC_KNN_STAT_COLLECT_NORM_FORM: *value = ws->last_norm_form; break;
    default: CC_REQUIRE(false, "unknown kNN statistic %d", stat);
  }
  return CC_OK;
}

int cc_knn_search_dev(void *knn, const float *dev_x, int64_t n_total, int d, int64_t q_begin, int64_t n_q, int k,
                      int include_self, int32_t *host_idx, float *host_dist) {
  CC_REQUIRE(cc::initialised(), "cc_init has not been called");
  return cc::knn_run(static_cast<cc::KnnWorkspace *>(knn), dev_x, n_total, d, q_begin, n_q, k, include_self, host_idx,
                     host_dist);
}

// the hooks of ABI v101, kept as shorthands for options of the default workspace
int cc_knn_force_fallback(int on) { return cc_knn_set_option(nullptr, CC_KNN_OPT_FORCE_FALLBACK, on); }
int cc_knn_use_fp64_prefilter(int on) { return cc_knn_set_option(nullptr, CC_KNN_OPT_FP64_PREFILTER, on); }
int cc_knn_use_fp16_collect(int on) { return cc_knn_set_option(nullptr, CC_KNN_OPT_FP16_COLLECT, on); }
int cc_knn_force_full_select(int on) { return cc_knn_set_option(nullptr, CC_KNN_OPT_FORCE_FULL_SELECT, on); }
int64_t cc_knn_last_fallback_count(void) { return cc::knn_default_ws()->last_fallback; }
int64_t cc_knn_last_full_scan_count(void) { return cc::knn_default_ws()->last_full_scan; }

}  // extern "C"
