// Small-batch form of the Dense layers (M <= 128 rows): the reference's real call shapes are batches of 32 (keras
// predict, reference src/bigan_basic.py:30) and minibatches of 128 (src/__main__.py:44).  With one m-tile the tiled
// kernel of gemm.hip has 24 (layer 1) / 12 (layer 2) workgroups, each walking K = 30000 alone: 1.5-3 ms of latency for
// a layer that only has to stream its weights once (369 MB: ~0.1 ms of HBM time).  Here the K range is split over
// the resident workgroup slots instead (split-K: 500-770 workgroups, k-tiles dealt round-robin), every workgroup streams its share of the SAME packed weight tiles straight
// from HBM into MFMA operands (no LDS: a B fragment is one coalesced 16-byte load per lane, an A fragment one
// 16-byte load from the L2-resident activations), and a second small kernel adds the partial sums in a FIXED
// order (deterministic, bit-reproducible) and applies bias + activation.
//
// Arithmetic: v_mfma_f32_32x32x2_f32, exact fp32 as in gemm.hip; the summation order over K differs from the tiled
// kernel (per split, then over splits), so a row encoded alone and inside a large pass agree to fp32 rounding,
// not bit for bit — the parity tests hold both to the same 1e-4 against the float64 oracle.
#include "common.h"

namespace cc {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SBK = GEMM_BK;                    // k-tile of the packed weights (32)
constexpr int SBN = GEMM_BN;                    // 128 columns per packed n-tile
constexpr int S_THREADS = 256;                  // 4 waves, 32 columns each
constexpr int TILE_W_F4 = (SBK / 4) * SBN;      // float4 per packed W tile

// partial[split][m][n_pad] = A[:, k-tiles of the split] @ W[those k-tiles, :]   (MS = 32-row blocks of A: 1..4)
template <int MS>
__global__ __launch_bounds__(S_THREADS) void gemm_small_partial_kernel(
    const float *__restrict__ A1, int64_t lda1, const float *__restrict__ A2, int64_t lda2, int M,
    const f32x4 *__restrict__ Wp, int KT1, int KT, int NT, int nsplit, float *__restrict__ partial,
    int64_t n_pad) {
  const int n_tile = blockIdx.x % NT, split = blockIdx.x / NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // split s takes k-tiles s, s + nsplit, s + 2 nsplit, ...: the workgroups of an n-tile, which run side by side, read
  // ONE advancing window of the packed weights (nsplit x 16 KB) instead of nsplit far-apart streams that keep
  // evicting each other's DRAM pages
  const int nk = split < KT ? (KT - split + nsplit - 1) / nsplit : 0;

  f32x16 acc[MS];
#pragma unroll
  for (int i = 0; i < MS; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // rows beyond M re-read row M - 1 (their partial sums are never used)
  int arow[MS];
#pragma unroll
  for (int i = 0; i < MS; ++i) arow[i] = (i * 32 + r) < M ? (i * 32 + r) : M - 1;
  const f32x4 *wt = Wp + ((int64_t)n_tile * KT + split) * TILE_W_F4 + h * SBN + wave * 32 + r;

  // The operands of the next BD - 1 k-tiles are in flight while the MFMAs of the current one run (a ring of BD register
  // sets for the weights AND the activations: loads complete in order, so an activation fragment requested after a
  // later weight fragment would make the wave wait for both).  With one set every wave sat out a full HBM round trip
  // per k-tile (layer 1 at 128 rows: 347 us for 369 MB); under this load a round trip takes ~5 us.
  constexpr int BD = MS <= 2 ? 4 : 2;
  f32x4 b[BD][4], a[BD][MS][4];
  auto load = [&](int j, int set) {                                  // j-th k-tile of this split
    const int kt = split + j * nsplit;
    const f32x4 *w = wt + (int64_t)j * nsplit * TILE_W_F4;
#pragma unroll
    for (int t = 0; t < 4; ++t) b[set][t] = __builtin_nontemporal_load(w + 2 * t * SBN);   // k = 8t + 4h .. +3 of this lane's column; read once: no cache allocation
    const bool seg1 = kt < KT1;                                      // wave-uniform
    const float *abase = seg1 ? A1 + (int64_t)kt * SBK : A2 + (int64_t)(kt - KT1) * SBK;
    const int64_t lda = seg1 ? lda1 : lda2;
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) a[set][i][t] = *reinterpret_cast<const f32x4 *>(abase + arow[i] * lda + 8 * t + 4 * h);
  };
  if constexpr (MS <= 2) {
    // Bandwidth-bound forms (<= 64 rows).  Straight-line loop body: every load is issued unconditionally (past the
    // split's last k-tile it re-reads that tile) and a k-tile past the end is multiplied by zero instead of being
    // branched around — with branches in the body the compiler's wait-count insertion fell back to s_waitcnt vmcnt(0)
    // right after issuing the newest loads, i.e. no prefetch at all.
    const int last = nk > 0 ? nk - 1 : 0;
#pragma unroll
    for (int p = 0; p < BD - 1; ++p) load(p < last ? p : last, p);
    __builtin_amdgcn_sched_barrier(0);
    for (int j = 0; j < nk; j += BD) {
#pragma unroll
      for (int u = 0; u < BD; ++u) {
        const int jn = j + u + BD - 1;
        load(jn < last ? jn : last, (u + BD - 1) % BD);
        __builtin_amdgcn_sched_barrier(0);                           // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
        const float live = j + u < nk ? 1.f : 0.f;                   // wave-uniform
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int i = 0; i < MS; ++i)
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][t][v], b[u][t][v] * live, acc[i], 0, 0, 0);
      }
    }
  } else {
    // MFMA-bound forms (65-128 rows: 48-64 MFMAs of 64 cycles per k-tile): two waves per SIMD cover each other's
    // loads; the plain ping-pong measured faster here (0.45 vs 0.54 ms per 128-row call) than the form above
    if (nk > 0) load(0, 0);
    for (int j = 0; j < nk; j += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (j + u < nk) {                                            // wave-uniform
          if (j + u + 1 < nk) load(j + u + 1, u ^ 1);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
              for (int i = 0; i < MS; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][t][v], b[u][t][v], acc[i], 0, 0, 0);
        }
      }
    }
  }

  float *dst = partial + (int64_t)split * (MS * 32) * n_pad + n_tile * SBN + wave * 32 + r;
#pragma unroll
  for (int i = 0; i < MS; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      dst[(int64_t)row * n_pad] = acc[i][e];
    }
}

// C[m][n] = act(bias[n] + sum over splits in order 0, 1, 2, ... of partial[split][m][n])
__global__ void gemm_small_reduce_kernel(const float *__restrict__ partial, int nsplit, int m_pad, int64_t n_pad, int M,
                                         int N, const float *__restrict__ bias, float *__restrict__ C, int64_t ldc,
                                         int act) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)M * N) return;
  const int m = (int)(t / N), n = (int)(t % N);
  // the partials of 16 splits are requested together, then added in split order
  const float *p = partial + (int64_t)m * n_pad + n;
  const int64_t step = (int64_t)m_pad * n_pad;
  float z = 0.f;
  for (int s0 = 0; s0 < nsplit; s0 += 16) {
    float q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = s0 + j < nsplit ? p[(s0 + j) * step] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) z += q[j];   // + 0.f for absent splits changes nothing
  }
  z += bias[n];
  C[(int64_t)m * ldc + n] = act == ACT_SIGMOID ? 1.0f / (1.0f + expf(-z)) : (act == ACT_RELU ? fmaxf(z, 0.f) : z);
}

// The tail of the cont encoder for <= 128 rows in one launch (reference src/bigan_cont.py:38-41): the split-K partial
// sums of the first tail layer are reduced in a fixed order (+ bias, sigmoid) into LDS, then two more Dense + sigmoid
// layers of <= 160 inputs run out of LDS with their packed weights read straight from L2.  One workgroup per 32 rows;
// wave w owns columns [32w, 32w + 32) of every 128-column tile.  As separate tiled launches the two 150-wide layers
// took ~27 us each (5 k-tiles behind two barriers apiece) plus the reduce kernel of the layer before.
constexpr int TS_LD = 164;   // floats per LDS row: 160 inputs + 4 (16-byte reads of 32 rows spread over the banks)

__device__ __forceinline__ float ts_sigmoid(float z) { return 1.0f / (1.0f + expf(-z)); }

constexpr int TS_KT = 5;     // k-tiles of a fused layer (<= 160 inputs)

// the B fragments of one 128-column weight tile for all its k-tiles, in flight together (one L2 round trip per tile
// instead of one per k-tile: a single workgroup has nothing else to hide them behind)
__device__ __forceinline__ void ts_load_w(const f32x4 *__restrict__ Wp, int nt, int KT, f32x4 (&bw)[TS_KT][4], int lane,
                                          int wave) {
  const int r = lane & 31, h = lane >> 5;
  const f32x4 *wt = Wp + (int64_t)nt * KT * TILE_W_F4 + h * SBN + wave * 32 + r;
#pragma unroll
  for (int kt = 0; kt < TS_KT; ++kt)
#pragma unroll
    for (int t = 0; t < 4; ++t) bw[kt][t] = wt[(kt < KT ? kt : 0) * TILE_W_F4 + 2 * t * SBN];
}

// columns [128 nt + 32 wave, +32) of sigmoid(src[32][TS_LD] @ W + bias): into LDS rows (0 beyond N, up to kcols_out
// columns) or, for the last layer, into the output rows
__device__ __forceinline__ void ts_tile(const float *src, const f32x4 (&bw)[TS_KT][4], const float *__restrict__ bias,
                                        int nt, int KT, int N, float *dst_lds, int kcols_out, float *dst_glob,
                                        int64_t ldo, int rows_valid, int lane, int wave) {
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int kt = 0; kt < TS_KT; ++kt) {
    if (kt < KT) {
      f32x4 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const f32x4 *>(src + r * TS_LD + kt * SBK + 8 * t + 4 * h);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][u], bw[kt][t][u], acc, 0, 0, 0);
    }
  }
  const int col = nt * SBN + wave * 32 + r;
  const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    const float v = col < N ? ts_sigmoid(acc[e] + bv) : 0.f;
    if (dst_lds) {
      if (col < kcols_out) dst_lds[row * TS_LD + col] = v;
    } else if (col < N && row < rows_valid) {
      dst_glob[(int64_t)row * ldo + col] = v;
    }
  }
}

__global__ __launch_bounds__(S_THREADS) void tail_small_fused_kernel(
    const float *__restrict__ partial, int nsplit, int m_pad, int64_t n_pad, int M, const float *__restrict__ bias3,
    int N3, const f32x4 *__restrict__ W4, const float *__restrict__ bias4, int KT4, int NT4, int N4,
    const f32x4 *__restrict__ W5, const float *__restrict__ bias5, int KT5, int NT5, int N5, float *__restrict__ out,
    int64_t ldo) {
  __shared__ __attribute__((aligned(16))) float hA[32 * TS_LD];
  __shared__ __attribute__((aligned(16))) float hB[32 * TS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 32;
  // the weights do not depend on anything computed here: all of layer 4's and the first tile of layer 5's are
  // requested before the reduction below starts
  f32x4 w4a[TS_KT][4], w4b[TS_KT][4], w5a[TS_KT][4];
  ts_load_w(W4, 0, KT4, w4a, lane, wave);
  ts_load_w(W4, NT4 > 1 ? 1 : 0, KT4, w4b, lane, wave);
  ts_load_w(W5, 0, KT5, w5a, lane, wave);
  // layer 3's output: fixed-order sum of the split-K partials, bias, sigmoid; zero beyond its N3 columns.  A thread
  // owns one column of several rows; the partials of a row are requested 16 at a time.
  const int kc = KT4 * SBK;
  const int64_t step = (int64_t)m_pad * n_pad;
  for (int i0 = tid; i0 < 32 * kc; i0 += 4 * S_THREADS) {
    float q[4][16];
    int rowv[4], colv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + c * S_THREADS;
      rowv[c] = i / kc;
      colv[c] = i - rowv[c] * kc;
    }
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsplit; s0 += 16) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool live = rowv[c] < 32 && colv[c] < N3;
        const float *p = partial + (int64_t)(row0 + (live ? rowv[c] : 0)) * n_pad + (live ? colv[c] : 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) q[c][j] = live && s0 + j < nsplit ? p[(s0 + j) * step] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) z[c] += q[c][j];   // split order 0, 1, 2, ...; + 0.f for absent splits changes nothing
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (rowv[c] < 32) hA[rowv[c] * TS_LD + colv[c]] = colv[c] < N3 ? ts_sigmoid(z[c] + bias3[colv[c]]) : 0.f;
  }
  __syncthreads();
  ts_tile(hA, w4a, bias4, 0, KT4, N4, hB, KT5 * SBK, nullptr, 0, 0, lane, wave);
  if (NT4 > 1) ts_tile(hA, w4b, bias4, 1, KT4, N4, hB, KT5 * SBK, nullptr, 0, 0, lane, wave);
  __syncthreads();
  ts_tile(hB, w5a, bias5, 0, KT5, N5, nullptr, 0, out + (int64_t)row0 * ldo, ldo, M - row0, lane, wave);
  for (int t = 1; t < NT5; ++t) {   // wide encodings (E > 128): the remaining column tiles of the last layer
    ts_load_w(W5, t, KT5, w4a, lane, wave);
    ts_tile(hB, w4a, bias5, t, KT5, N5, nullptr, 0, out + (int64_t)row0 * ldo, ldo, M - row0, lane, wave);
  }
}

DevBuf g_partial;   // split-K partial sums; released by gemm_small_release (cc_shutdown)

}  // namespace

void gemm_small_release() { g_partial.release(); }

bool gemm_small_applies(int64_t M, const PackedWeight &W) {
  // worth it when the tiled kernel would run a single m-tile over a long K: layers 1-2 and the first tail layer
  // (K = 1650: two workgroups walking 52 k-tiles took 120 us of a 0.43 ms call); the 150-wide layers stay tiled
  return M >= 1 && M <= 128 && W.N > 0 && (W.KT1 + W.KT2) >= 16;
}

// the split-K partial sums of act-less A @ W into g_partial; *nsplit_out / *m_pad_out / *n_pad_out describe its layout
static int launch_partial(const float *A1, int64_t lda1, const float *A2, int64_t lda2, int64_t M, const PackedWeight &W,
                          int *nsplit_out, int *m_pad_out, int64_t *n_pad_out) {
  const int KT = (int)(W.KT1 + W.KT2), NT = (int)W.NT;
  const int MS = (int)ceil_div(M, 32), m_pad = MS * 32;
  const int64_t n_pad = (int64_t)NT * SBN;
  // As many workgroups as are resident at once and no more: by its registers the 32-row kernel runs 3 workgroups per
  // CU, the others 2 (a wave per SIMD each) — 43 x 24 = 1032 workgroups on 1024 slots ran a second round for 8 of them
  // (layer 1 of a 32-row call: 106 us instead of 60).  At least 8 k-tiles each (4 for a short K).
  const int slots = 256 * (MS == 1 ? 3 : 2);
  const int min_kt = KT >= 256 ? 8 : 4;
  int nsplit = slots / NT > 0 ? slots / NT : 1;
  if (nsplit > KT / min_kt) nsplit = KT / min_kt > 0 ? KT / min_kt : 1;
  int rc;
  if ((rc = g_partial.alloc(sizeof(float) * (size_t)nsplit * (size_t)m_pad * (size_t)n_pad))) return rc;
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(W.d_w);
#define CC_SMALL(MSV)                                                                                              \
  hipLaunchKernelGGL(gemm_small_partial_kernel<MSV>, dim3((unsigned)(NT * nsplit)), dim3(S_THREADS), 0, stream(), A1, \
                     lda1, A2, lda2, (int)M, wp, (int)W.KT1, KT, NT, nsplit, g_partial.as<float>(), n_pad)
  switch (MS) {
    case 1: CC_SMALL(1); break;
    case 2: CC_SMALL(2); break;
    case 3: CC_SMALL(3); break;
    default: CC_SMALL(4); break;
  }
#undef CC_SMALL
  CC_HIP(hipGetLastError());
  *nsplit_out = nsplit;
  *m_pad_out = m_pad;
  *n_pad_out = n_pad;
  return CC_OK;
}

int launch_gemm_small(const float *A1, int64_t lda1, const float *A2, int64_t lda2, int64_t M, const PackedWeight &W,
                      float *C, int64_t ldc, int act, int prof_class) {
  ProfScope prof(prof_class, 2.0 * (double)M * (double)(W.K1 + W.K2) * (double)W.N);
  int nsplit, m_pad, rc;
  int64_t n_pad;
  if ((rc = launch_partial(A1, lda1, A2, lda2, M, W, &nsplit, &m_pad, &n_pad))) return rc;
  const int64_t total = M * W.N;
  hipLaunchKernelGGL(gemm_small_reduce_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream(),
                     g_partial.as<float>(), nsplit, m_pad, n_pad, (int)M, (int)W.N, W.d_bias, C, ldc, act);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

bool tail_small_applies(int64_t M, const PackedWeight &W3, const PackedWeight &W4, const PackedWeight &W5) {
  // layer 3 by split-K, layers 4 and 5 out of LDS: single-segment Dense layers of <= 160 inputs
  return gemm_small_applies(M, W3) && W3.N <= 5 * SBK && W4.K2 == 0 && W5.K2 == 0 && W4.K1 == W3.N && W5.K1 == W4.N &&
         W4.N <= 5 * SBK && W4.KT1 <= 5 && W5.KT1 <= 5 && W5.N > 0;
}

int launch_tail_small(const float *A, int64_t lda, int64_t M, const PackedWeight &W3, const PackedWeight &W4,
                      const PackedWeight &W5, float *out, int64_t ldo, int prof_class) {
  ProfScope prof(prof_class, 2.0 * (double)M * ((double)(W3.K1 + W3.K2) * (double)W3.N + (double)W4.K1 * (double)W4.N +
                                                 (double)W5.K1 * (double)W5.N));
  int nsplit, m_pad, rc;
  int64_t n_pad;
  if ((rc = launch_partial(A, lda, nullptr, 0, M, W3, &nsplit, &m_pad, &n_pad))) return rc;
  hipLaunchKernelGGL(tail_small_fused_kernel, dim3((unsigned)(m_pad / 32)), dim3(S_THREADS), 0, stream(),
                     g_partial.as<float>(), nsplit, m_pad, n_pad, (int)M, W3.d_bias, (int)W3.N,
                     reinterpret_cast<const f32x4 *>(W4.d_w), W4.d_bias, (int)W4.KT1, (int)W4.NT, (int)W4.N,
                     reinterpret_cast<const f32x4 *>(W5.d_w), W5.d_bias, (int)W5.KT1, (int)W5.NT, (int)W5.N, out, ldo);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

}  // namespace cc
