// Shared host-side plumbing for libcellcomm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cellcomm_hip.h"

namespace cc {

void set_error(const char *fmt, ...);
hipStream_t stream();
bool two_streams_active();   // the running call alternates its passes between the two library streams
bool initialised();
// Counts successful cc_init calls that (re)created the library context: per-device one-time setup (kernel attributes)
// is keyed on it, so it is redone after cc_shutdown + cc_init — possibly on another device.
int init_generation();
int num_cus();   // compute units of the device the library is initialised on

// Kernel-class timers (HIP events on the library stream).
struct ProfScope {
  int klass;
  double work;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(int klass, double work);
  ~ProfScope();
};

#define CC_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      cc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                    __FILE__, __LINE__);                                          \
      return CC_EHIP;                                                             \
    }                                                                             \
  } while (0)

#define CC_REQUIRE(cond, ...)                                                     \
  do {                                                                            \
    if (!(cond)) {                                                                \
      cc::set_error(__VA_ARGS__);                                                 \
      return CC_EINVAL;                                                           \
    }                                                                             \
  } while (0)

// Device allocation released when it goes out of scope (temporaries of one call: no leak on an early return).
struct DevPtr {
  void *p = nullptr;
  DevPtr() = default;
  DevPtr(const DevPtr &) = delete;
  DevPtr &operator=(const DevPtr &) = delete;
  ~DevPtr() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Grow-only device scratch owned by a handle and kept across calls; release() on the owner's free.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int alloc(size_t bytes) {
    if (bytes <= cap && p) return CC_OK;
    release();
    bytes = bytes ? bytes : 16;
    CC_HIP(hipMalloc(&p, bytes));
    cap = bytes;
    return CC_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// ---- kernels' host launchers (defined in the .hip files) -------------------

// gather.hip
int gather_chunk_count(int G);  // entries per row of the chunk index
int build_chunk_index(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, int G,
                      int32_t *d_chunk_rel);
int launch_gather(const int64_t *d_indptr, const int32_t *d_indices, const float *d_vals,
                  const int32_t *d_chunk_rel, const int64_t *d_row_ids, int64_t row_start, int64_t n, int G,
                  int64_t ld, float *d_out, double nnz_hint);

// gemm.hip
#ifndef CC_GEMM_BK
#define CC_GEMM_BK 32
#endif
constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = CC_GEMM_BK;
enum Act { ACT_SIGMOID = 0, ACT_LINEAR = 1, ACT_RELU = 2 };
struct PackedWeight {
  float *d_w = nullptr;     // [NT][KT][BK/4][BN][4]
  float *d_bias = nullptr;  // [NT*BN] zero padded
  int64_t K1 = 0, K2 = 0;   // logical rows of segment 1 / 2
  int64_t KT1 = 0, KT2 = 0; // k-tiles of segment 1 / 2
  int64_t N = 0, NT = 0;
};
// Pack host kernel `w` [K1+K2 x N] row-major (rows [0,K1) multiply segment 1,
// rows [K1,K1+K2) segment 2) and bias[N] into the device layout.
int pack_weight(const float *w, const float *bias, int64_t K1, int64_t K2, int64_t N,
                PackedWeight *out);
void free_weight(PackedWeight *w);
// C[M x N] (ldc) = act(A1[M x K1](lda1) @ W[0:K1] + A2[M x K2](lda2) @ W[K1:] + bias)
// lda* must cover KT*·BK columns and the columns beyond K* must be zero.
// allow_split_k: calls of <= 128 rows over a long K may take the split-K form of gemm_small.hip (lower latency,
// same tolerance, but a different summation order than the tiled kernel: opt-in per encoder, cc_encoder_set_low_latency)
int launch_gemm(const float *A1, int64_t lda1, const float *A2, int64_t lda2, int64_t M,
                const PackedWeight &W, float *C, int64_t ldc, int act, int prof_class, bool allow_split_k = false);
int launch_row_softmax(float *X, int64_t M, int E, int64_t ld);
// gemm_small.hip — the same product for M <= 128 rows as split-K over ~1024 workgroups (launch_gemm dispatches to it)
bool gemm_small_applies(int64_t M, const PackedWeight &W);
int launch_gemm_small(const float *A1, int64_t lda1, const float *A2, int64_t lda2, int64_t M, const PackedWeight &W,
                      float *C, int64_t ldc, int act, int prof_class);
// the whole tail of the cont encoder (three Dense + sigmoid layers, the last two <= 160 wide) for <= 128 rows in two
// launches: split-K partials of the first, then reduce + both remaining layers fused (same opt-in as above)
bool tail_small_applies(int64_t M, const PackedWeight &W3, const PackedWeight &W4, const PackedWeight &W5);
int launch_tail_small(const float *A, int64_t lda, int64_t M, const PackedWeight &W3, const PackedWeight &W4,
                      const PackedWeight &W5, float *out, int64_t ldo, int prof_class);
void gemm_small_release();

// gemm16.hip — layers 1/2 on the fp16 matrix cores (CC_DTYPE_F16 / CC_DTYPE_F16X2)
constexpr int GEMM16_TILE = 256;   // rows and columns of a workgroup's output tile (one workgroup per CU)
struct PackedWeight16 {
  void *d_w = nullptr;      // [NT][KT][plane][TK / 8][256][8] fp16 (plane 0 = hi, plane 1 = lo)
  float *d_bias = nullptr;  // [NT*256]
  float scale = 1.f;        // power of two applied to W before the fp16 split
  int planes = 2;           // 1: CC_DTYPE_F16 (K-step 64), 2: CC_DTYPE_F16X2 (K-step 32)
  int TK = 32;
  int64_t K1 = 0, K2 = 0, KT1 = 0, KT2 = 0, N = 0, NT = 0;
};
// A16T — the layout of every fp16 activation matrix the fp16 GEMMs read (round 4): MFMA-operand tiles instead of rows.
// A matrix of M rows x Kp columns (Kp a multiple of 32) is cut into tiles of 256 rows x 32 columns, stored tile after
// tile, row block major: tile (mb, kb) = rows 256 mb .. 256 mb + 255, columns 32 kb .. 32 kb + 31 = 16 KiB contiguous;
// inside a tile a row is 64 bytes = four 16-byte chunks, chunk c of row r at chunk position c ^ a16t_swz(r).  A tile is
// byte for byte the LDS image the kernels read fragments from (rows as XOR-swizzled chunks: conflict-free
// ds_read_b128 of the 16x16x32 fragments), so the LDS-DMA of a K-step is a straight copy of 16 / 32 contiguous KiB — one
// KiB per wave instruction like the packed weights — where rows of a row-major matrix made every request touch
// eight 128-byte pieces 60 KB apart (layer 1: 1438 -> 1550 TFLOP/s at C3's width, 1418 -> 1570 at configs[4]'s).
// Buffers hold round_up(M, 256) rows; rows beyond M are never stored to a result.
#if defined(__HIPCC__)
#define CC_HD __host__ __device__
#else
#define CC_HD
#endif
CC_HD inline int a16t_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // {0, 2, 3, 1}[(row >> 2) & 3]
CC_HD inline int64_t a16t_offset(int64_t row, int64_t col, int64_t Kp) {               // in halfs
  const int rl = (int)(row & 255);
  return (((row >> 8) * (Kp >> 5) + (col >> 5)) * 256 + rl) * 32 + ((((int)(col >> 3) & 3) ^ a16t_swz(rl)) << 3) + (col & 7);
}

struct GemmSeg16 {
  const void *A;   // fp16, A16T layout (above), M rows x lda columns
  int64_t lda;     // columns Kp of the A16T matrix: a multiple of 32 (64 for one-plane weights), >= KT·TK, zero beyond K
  int which;       // 0: multiplies weight rows [0,K1), 1: rows [K1,K1+K2)
  int nterms;      // 1: W_hi, 2: W_hi + W_lo
};
int pack_weight16(const float *w, const float *bias, int64_t K1, int64_t K2, int64_t N, int planes, PackedWeight16 *out);
void free_weight16(PackedWeight16 *w);
int launch_gemm16(const GemmSeg16 *segs, int nsegs, int64_t M, const PackedWeight16 &W, void *C, void *C_lo,
                  int64_t ldc, int out_fp16, int act, int prof_class);
int launch_convert_f16(const float *d_x, int64_t ldx, int64_t M, int G, void *d_out_half, int64_t ldo, int *d_inexact);
// gather.hip — fp16 output; *d_inexact is OR-ed with 1 when a value is not exactly an fp16 number
int launch_gather_f16(const int64_t *d_indptr, const int32_t *d_indices, const float *d_vals,
                      const int32_t *d_chunk_rel, const int64_t *d_row_ids, int64_t row_start, int64_t n, int G,
                      int64_t ld, void *d_out_half, int *d_inexact, double nnz_hint);

// gather.hip — canonical-form check of device-resident CSR arrays (cc_csr_upload / cc_csr_wrap_device):
// indptr monotone from 0 to nnz, columns of a row strictly increasing and inside [0, G)
int validate_csr(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, int64_t nnz, int G);

// spmm.hip — sparse-aware first layer: CSR rows x fp32 weight panels on the vector ALUs
constexpr int SPMM_ROWS = 256, SPMM_KB = 64, SPMM_NT = 256;
struct SpmmWeight {
  float *d_w = nullptr;      // [npan][K][256] fp32, zero beyond N
  float *d_bias = nullptr;   // [N] or NULL
  int64_t K = 0, N = 0, npan = 0;
};
struct SpmmStream {           // the matrix's non-zeros re-ordered per 16-row group by (64-column stage, row, column)
  void *d_ent = nullptr;      // {code, value} pairs
  int *d_offs = nullptr;      // [ngroups][nstages + 1] stream index of each (group, stage); bit 0 = padded
  int64_t n_rows = 0, n_entries = 0, ngroups = 0;
  int G = 0, nstages = 0;
  bool unusable = false;      // the matrix does not fit the 32-bit stream offsets: dense path only
};
// w_rows: [K x N] row-major host weights (rows that multiply the raw counts), bias: [N] or NULL
int pack_spmm_weight(const float *w_rows, const float *bias, int64_t K, int64_t N, SpmmWeight *out);
void free_spmm_weight(SpmmWeight *w);
int build_spmm_stream(const int64_t *d_indptr, const int32_t *d_indices, const float *d_vals, int64_t n_rows, int G,
                      SpmmStream *st);
void free_spmm_stream(SpmmStream *st);
// C[n x N] (ldc) = act(X[row_start : row_start + n] @ W (+ addend[n x N] (ld_add)) + bias)
int launch_spmm(const SpmmStream &st, int64_t row_start, int64_t n, const SpmmWeight &W, const float *addend,
                int64_t ld_add, float *C, int64_t ldc, int act, int prof_class, double nnz_hint);

// knn.hip — ws == nullptr: the library's default workspace (released by cc_shutdown)
struct KnnWorkspace;
KnnWorkspace *knn_default_ws();
void knn_default_ws_release();
int knn_run_host(KnnWorkspace *ws, const float *h_x, int64_t n_total, int d, int k, int include_self, int32_t *h_idx,
                 float *h_dist);
int knn_run(KnnWorkspace *ws, const float *d_x, int64_t n_total, int d, int64_t q_begin, int64_t n_q, int k,
            int include_self, int32_t *h_idx, float *h_dist);

}  // namespace cc
