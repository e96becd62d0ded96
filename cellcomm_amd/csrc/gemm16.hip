// K2' — encoder layers 1 and 2 on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate).
//
// Two uses of one kernel:
//   CC_DTYPE_F16    operands rounded to fp16 once (BASELINE config 5 "fp16 MFMA"): one term.
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
// Tile: 256 x 128 per 256-thread workgroup, 4 waves as 2 x 2, 128 x 64 per wave (128 accumulator
// registers: the A fragments are re-used by W_hi and W_lo and LDS traffic per MFMA halves against a
// 64 x 64 wave tile, which is what bound the first version), K-step 64.  LDS: A [256][64+8] fp16 (the pad makes the 16-byte fragment reads
// conflict-free), W_hi / W_lo tiles [8][128][8] fp16 — pre-packed so a lane's 8 consecutive k for
// one column are one 16-byte read and a tile is 16 KiB contiguous in HBM.
#include "common.h"

namespace cc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

#ifndef CC_G16_WM
#define CC_G16_WM 4          // 32-row MFMA tiles per wave along M: 2 -> 128 x 128 workgroup tile, 4 -> 256 x 128
#endif
#ifndef CC_G16_MINW
#define CC_G16_MINW 2
#endif
constexpr int WM_T = CC_G16_WM;
constexpr int HB_M = 64 * WM_T, HB_N = 128, HB_K = 64;
constexpr int A_PASSES = HB_M / 32;                  // staging passes of 32 rows
constexpr int H_LDS_A_STRIDE = HB_K + 8;             // halfs
constexpr int H_THREADS = 256;
constexpr int H_TILE_W = (HB_K / 8) * HB_N;          // 16-byte units per W tile (1024 = 16 KiB)

struct Seg16 {
  const _Float16 *A;   // [M x lda] fp16
  int64_t lda;         // halfs
  int KT;              // k-tiles of this segment
  int wkt0;            // first k-tile of the packed weight this segment multiplies
  int nterms;          // 1: W_hi only, 2: W_hi and W_lo
};
struct Segs16 { Seg16 s[4]; int n; };

__device__ __forceinline__ float sigmoid16_(float z) { return 1.0f / (1.0f + expf(-z)); }

// OUT_MODE 0: C fp32 [M x ldc];  1: C_hi (and C_lo when split) fp16 planes [M x ldc]
template <int OUT_MODE>
__global__ __launch_bounds__(H_THREADS, CC_G16_MINW) void gemm16_kernel(Segs16 segs, int64_t M, const f32x4 *__restrict__ Wp,
                                                              const float *__restrict__ bias, float inv_scale,
                                                              int KTW, int N, int MT, int NT, void *__restrict__ C,
                                                              void *__restrict__ C_lo, int64_t ldc, int act) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[HB_M * H_LDS_A_STRIDE + 2 * H_TILE_W * 8];
  _Float16 *As = lds;
  f32x4 *Bh = reinterpret_cast<f32x4 *>(lds + HB_M * H_LDS_A_STRIDE);
  f32x4 *Bl = Bh + H_TILE_W;

  const int nwg = MT * NT;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int wgid = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int per_group = 8 * NT;
  const int first_m = (wgid / per_group) * 8;
  const int gsz = (MT - first_m < 8) ? MT - first_m : 8;
  const int m_tile = first_m + (wgid % per_group) % gsz;
  const int n_tile = (wgid % per_group) / gsz;
  const int64_t m0 = (int64_t)m_tile * HB_M;
  const int n0 = n_tile * HB_N;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // A staging: 128 rows x 8 chunks of 16 B; thread -> (row = tid/8 + 32p, chunk = tid%8)
  const int a_row = tid >> 3, a_ch = tid & 7;
  int a_lr[A_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    const int64_t gr = m0 + a_row + 32 * p;
    a_lr[p] = (int)((gr < M ? gr : M - 1) - m0);
  }

  f32x16 acc[WM_T][2];
#pragma unroll
  for (int i = 0; i < WM_T; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const f32x4 *Wn = Wp + (int64_t)n_tile * KTW * (2 * H_TILE_W) + tid;   // [n_tile][kt][hi|lo][1024]

  f32x4 ra[A_PASSES], rh[4], rl[4];
  const _Float16 *a_lds = As + (wm * 32 * WM_T + r) * H_LDS_A_STRIDE + 8 * h;
  const f32x4 *bh_lds = Bh + h * HB_N + wn * 64 + r;
  const f32x4 *bl_lds = Bl + h * HB_N + wn * 64 + r;

  for (int si = 0; si < segs.n; ++si) {
    const Seg16 sg = segs.s[si];
    if (sg.KT == 0) continue;
    const bool two = sg.nterms == 2;   // uniform
    const _Float16 *pa[A_PASSES];
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) pa[p] = sg.A + (m0 + a_lr[p]) * sg.lda + 8 * a_ch;
    const f32x4 *wt = Wn + (int64_t)sg.wkt0 * (2 * H_TILE_W);

    // prologue of the segment: stage k-tile 0
    __syncthreads();   // previous segment's last tile fully consumed
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) { ra[p] = *reinterpret_cast<const f32x4 *>(pa[p]); pa[p] += HB_K; }
#pragma unroll
    for (int p = 0; p < 4; ++p) rh[p] = wt[p * H_THREADS];
    if (two) {
#pragma unroll
      for (int p = 0; p < 4; ++p) rl[p] = wt[H_TILE_W + p * H_THREADS];
    }
    wt += 2 * H_TILE_W;
    {
      _Float16 *aw = As + a_row * H_LDS_A_STRIDE + 8 * a_ch;
#pragma unroll
      for (int p = 0; p < A_PASSES; ++p) *reinterpret_cast<f32x4 *>(aw + p * 32 * H_LDS_A_STRIDE) = ra[p];
#pragma unroll
      for (int p = 0; p < 4; ++p) Bh[tid + p * H_THREADS] = rh[p];
      if (two) {
#pragma unroll
        for (int p = 0; p < 4; ++p) Bl[tid + p * H_THREADS] = rl[p];
      }
    }
    __syncthreads();

    for (int kt = 0; kt < sg.KT; ++kt) {
      const bool more = kt + 1 < sg.KT;
      if (more) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) { ra[p] = *reinterpret_cast<const f32x4 *>(pa[p]); pa[p] += HB_K; }
#pragma unroll
        for (int p = 0; p < 4; ++p) rh[p] = wt[p * H_THREADS];
        if (two) {
#pragma unroll
          for (int p = 0; p < 4; ++p) rl[p] = wt[H_TILE_W + p * H_THREADS];
        }
        wt += 2 * H_TILE_W;
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < HB_K / 16; ++t) {
        h8 a[WM_T], bh[2];
#pragma unroll
        for (int mt = 0; mt < WM_T; ++mt)
          a[mt] = *reinterpret_cast<const h8 *>(a_lds + mt * 32 * H_LDS_A_STRIDE + 16 * t);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bh[nt] = __builtin_bit_cast(h8, bh_lds[2 * t * HB_N + nt * 32]);
#pragma unroll
        for (int mt = 0; mt < WM_T; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        if (two) {
          h8 bl[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) bl[nt] = __builtin_bit_cast(h8, bl_lds[2 * t * HB_N + nt * 32]);
#pragma unroll
          for (int mt = 0; mt < WM_T; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bl[nt], acc[mt][nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      if (more) {
        _Float16 *aw = As + a_row * H_LDS_A_STRIDE + 8 * a_ch;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) *reinterpret_cast<f32x4 *>(aw + p * 32 * H_LDS_A_STRIDE) = ra[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) Bh[tid + p * H_THREADS] = rh[p];
        if (two) {
#pragma unroll
          for (int p = 0; p < 4; ++p) Bl[tid + p * H_THREADS] = rl[p];
        }
      }
      __syncthreads();
    }
  }

  // epilogue: undo the weight scale, bias, activation
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = n0 + wn * 64 + nt * 32 + r;
    const bool col_ok = col < N;
    const float bcol = col_ok ? bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < WM_T; ++mt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 32 * WM_T + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (col_ok && row < M) {
          float z = acc[mt][nt][e] * inv_scale + bcol;
          z = (act == ACT_SIGMOID) ? sigmoid16_(z) : z;
          if (OUT_MODE == 0) {
            static_cast<float *>(C)[row * ldc + col] = z;
          } else {
            const _Float16 zh = (_Float16)z;
            static_cast<_Float16 *>(C)[row * ldc + col] = zh;
            if (C_lo) static_cast<_Float16 *>(C_lo)[row * ldc + col] = (_Float16)(z - (float)zh);
          }
        }
      }
    }
  }
}

// ---- weight packing: [NT][KT][hi|lo][8][128][8] fp16 ------------------------------------------------
// 16-byte unit u of a tile = (j = u / 128, i = u % 128) holds k = 8j .. 8j+7 of column i.
__global__ void pack_weight16_kernel(const float *__restrict__ w, int64_t K1, int64_t K2, int64_t N, int64_t KT1,
                                     int64_t KT, int64_t NT, float scale, _Float16 *__restrict__ dst) {
  const int64_t total = NT * KT * (int64_t)H_TILE_W;   // 16-byte units per plane
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
    const int within = (int)(u % H_TILE_W);
    const int i = within % HB_N, j = within / HB_N;
    const int64_t tile = u / H_TILE_W;
    const int64_t kt = tile % KT, nt = tile / KT;
    const int64_t n = nt * HB_N + i;
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int64_t src_row;
      bool ok;
      if (kt < KT1) { const int64_t k = kt * HB_K + 8 * j + c; ok = k < K1; src_row = k; }
      else { const int64_t k = (kt - KT1) * HB_K + 8 * j + c; ok = k < K2; src_row = K1 + k; }
      const float v = (ok && n < N) ? w[src_row * N + n] * scale : 0.f;
      hi[c] = (_Float16)v;
      lo[c] = (_Float16)(v - (float)hi[c]);
    }
    _Float16 *d_hi = dst + (tile * 2 + 0) * (int64_t)(H_TILE_W * 8) + (int64_t)within * 8;
    _Float16 *d_lo = dst + (tile * 2 + 1) * (int64_t)(H_TILE_W * 8) + (int64_t)within * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) { d_hi[c] = hi[c]; d_lo[c] = lo[c]; }
  }
}

void free_weight16(PackedWeight16 *w) {
  if (w->d_w) (void)hipFree(w->d_w);
  if (w->d_bias) (void)hipFree(w->d_bias);
  *w = PackedWeight16();
}

int pack_weight16(const float *w, const float *bias, int64_t K1, int64_t K2, int64_t N, PackedWeight16 *out) {
  free_weight16(out);
  out->K1 = K1; out->K2 = K2; out->N = N;
  out->KT1 = ceil_div(K1, HB_K); out->KT2 = ceil_div(K2, HB_K);
  out->NT = ceil_div(N, HB_N);
  const int64_t KT = out->KT1 + out->KT2;
  if (N == 0) return CC_OK;
  // power-of-two scale bringing max|W| into [2^13, 2^14): keeps W_lo out of the fp16 subnormals
  float mx = 0.f;
  for (int64_t i = 0; i < (K1 + K2) * N; ++i) { const float a = fabsf(w[i]); mx = a > mx ? a : mx; }
  int e = 0;
  float scale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { (void)frexpf(mx, &e); scale = ldexpf(1.f, 14 - e); }
  out->scale = scale;
  const int64_t units = out->NT * KT * (int64_t)H_TILE_W;
  float *d_raw = nullptr;
  if ((K1 + K2) > 0) {
    CC_HIP(hipMalloc(&d_raw, sizeof(float) * (size_t)((K1 + K2) * N)));
    CC_HIP(hipMemcpyAsync(d_raw, w, sizeof(float) * (size_t)((K1 + K2) * N), hipMemcpyHostToDevice, stream()));
  }
  if (units > 0) {
    CC_HIP(hipMalloc(&out->d_w, 16 * 2 * (size_t)units));
    int64_t blocks = ceil_div(units, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pack_weight16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), d_raw, K1, K2, N, out->KT1,
                       KT, out->NT, scale, static_cast<_Float16 *>(out->d_w));
    CC_HIP(hipGetLastError());
  }
  CC_HIP(hipMalloc(&out->d_bias, sizeof(float) * (size_t)(out->NT * HB_N)));
  CC_HIP(hipMemsetAsync(out->d_bias, 0, sizeof(float) * (size_t)(out->NT * HB_N), stream()));
  CC_HIP(hipMemcpyAsync(out->d_bias, bias, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, stream()));
  CC_HIP(hipStreamSynchronize(stream()));
  if (d_raw) CC_HIP(hipFree(d_raw));
  return CC_OK;
}

int launch_gemm16(const GemmSeg16 *segs, int nsegs, int64_t M, const PackedWeight16 &W, void *C, void *C_lo,
                  int64_t ldc, int out_fp16, int act, int prof_class) {
  if (M == 0 || W.N == 0) return CC_OK;
  CC_REQUIRE(W.d_bias != nullptr, "gemm16: weight not packed");
  CC_REQUIRE(nsegs >= 1 && nsegs <= 4, "gemm16: 1..4 segments");
  Segs16 sg;
  sg.n = nsegs;
  double kflops = 0;
  for (int i = 0; i < nsegs; ++i) {
    const GemmSeg16 &g = segs[i];
    const int64_t KT = g.which == 0 ? W.KT1 : W.KT2;
    CC_REQUIRE(KT == 0 || (g.A && g.lda >= KT * HB_K && g.lda % 8 == 0), "gemm16: segment %d needs lda >= %lld (got %lld)",
               i, (long long)(KT * HB_K), (long long)g.lda);
    CC_REQUIRE(g.lda * HB_M < (int64_t)2147483647, "gemm16: lda too large");
    sg.s[i].A = static_cast<const _Float16 *>(g.A);
    sg.s[i].lda = g.lda;
    sg.s[i].KT = (int)KT;
    sg.s[i].wkt0 = g.which == 0 ? 0 : (int)W.KT1;
    sg.s[i].nterms = g.nterms;
  }
  kflops = (double)(W.K1 + W.K2);
  const int64_t MT = ceil_div(M, HB_M);
  const int64_t nwg = MT * W.NT;
  CC_REQUIRE(nwg < (int64_t)2147483647, "gemm16: grid too large");
  ProfScope prof(prof_class, 2.0 * (double)M * kflops * (double)W.N);
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(W.d_w);
  const int KTW = (int)(W.KT1 + W.KT2);
  if (out_fp16)
    hipLaunchKernelGGL(gemm16_kernel<1>, dim3((unsigned)nwg), dim3(H_THREADS), 0, stream(), sg, M, wp, W.d_bias,
                       1.0f / W.scale, KTW, (int)W.N, (int)MT, (int)W.NT, C, C_lo, ldc, act);
  else
    hipLaunchKernelGGL(gemm16_kernel<0>, dim3((unsigned)nwg), dim3(H_THREADS), 0, stream(), sg, M, wp, W.d_bias,
                       1.0f / W.scale, KTW, (int)W.N, (int)MT, (int)W.NT, C, C_lo, ldc, act);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

}  // namespace cc
