// K1 — CSR → dense minibatch gather (HBM-bound).
//
// Replaces the dense pivot of load_matrix plus the row sampling of
// CellTraining.sample_cell_data (reference src/cell_type_training.py:12-14,
// 37-38): rows of the cell x gene matrix are kept as CSR in HBM and only the
// rows of the current pass are densified.
//
// One 256-thread workgroup owns one (row, 4096-column chunk).  The chunk is
// staged in LDS: zero-filled with 16-byte LDS stores, the row's non-zeros that
// fall in the chunk are scattered into it with 4-byte LDS stores, and the chunk
// then leaves as full, coalesced 16-byte-per-lane global stores.  Where a row's
// non-zeros cross each 4096-column boundary is precomputed once per matrix
// (csr_chunk_index_kernel): a binary search per workgroup put ~22 dependent
// global-load round trips in front of every 16 KiB of stores (3.0 TB/s); with
// the index the dependent chain is indptr/index -> (col, val) -> LDS -> store.
// HBM sees one streaming read of (index, value) pairs and one streaming write of
// the dense tile; nothing is read-modify-written.  16 KiB of LDS per workgroup
// keeps 8 workgroups (32 waves) resident per CU.
//
// Algorithmic bytes per cell: 8·nnz + 8 (CSR read) + 4·ld (dense write).
#include "common.h"

namespace cc {

constexpr int GATHER_CH = 4096;
constexpr int GATHER_THREADS = 256;

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t *__restrict__ a, int64_t lo,
                                                   int64_t hi, int32_t key) {
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// chunk_rel[row·(NC+1) + j] = number of non-zeros of `row` with column < j·4096
__global__ void csr_chunk_index_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                       int64_t n_rows, int nc1, int32_t *__restrict__ chunk_rel) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_rows * nc1) return;
  const int64_t row = t / nc1;
  const int j = (int)(t % nc1);
  const int64_t beg = indptr[row], end = indptr[row + 1];
  chunk_rel[t] = (int32_t)(lower_bound_i32(indices, beg, end, j * GATHER_CH) - beg);
}

__global__ __launch_bounds__(GATHER_THREADS) void csr_gather_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ vals, const int32_t *__restrict__ chunk_rel, int nc1,
    const int64_t *__restrict__ row_ids, int64_t row_start, int nchunks, int64_t ld,
    float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[GATHER_CH];
  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x / nchunks;
  const int chunk = blockIdx.x % nchunks;
  const int64_t row = row_ids ? row_ids[r] : row_start + r;
  const int64_t c0 = (int64_t)chunk * GATHER_CH;
  const int64_t c1 = (c0 + GATHER_CH < ld) ? c0 + GATHER_CH : ld;
  const int64_t beg = indptr[row];
  // chunks that lie wholly in the [G, ld) padding have no boundary of their own: clamp
  const int32_t *rel = chunk_rel + row * nc1;
  const int64_t lo = beg + rel[chunk < nc1 - 1 ? chunk : nc1 - 1];
  const int64_t hi = beg + rel[chunk + 1 < nc1 - 1 ? chunk + 1 : nc1 - 1];

  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 *tile4 = reinterpret_cast<float4 *>(tile);
#pragma unroll
  for (int i = tid; i < GATHER_CH / 4; i += GATHER_THREADS) tile4[i] = z;

  __syncthreads();
  for (int64_t p = lo + tid; p < hi; p += GATHER_THREADS) tile[indices[p] - (int32_t)c0] = vals[p];
  __syncthreads();

  float4 *orow = reinterpret_cast<float4 *>(out + r * ld + c0);
  const int n4 = (int)((c1 - c0) >> 2);
#pragma unroll 4
  for (int i = tid; i < n4; i += GATHER_THREADS) orow[i] = tile4[i];
}

__device__ __forceinline__ int fp16_flags(float v) {
  const float back = (float)(_Float16)v;
  return (back != v ? 1 : 0) | ((isinf(back) && !isinf(v)) || isnan(v) ? 2 : 0);
}
__device__ __forceinline__ void raise_fp16_flags(int bad, int *flags) {
  const int any1 = __any(bad & 1), any2 = __any(bad & 2);
  if ((any1 | any2) && (threadIdx.x & 63) == 0) atomicOr(flags, (any1 ? 1 : 0) | (any2 ? 2 : 0));
}

// fp16 output variant (CC_DTYPE_F16 / F16X2 encoders): same staging; the rows leave in the A16T layout the fp16 GEMMs
// copy straight into LDS (common.h: 256-row x 32-column tiles, rows as XOR-swizzled 16-byte chunks).  *inexact gets
// bit 0 when a value does not survive the round trip through fp16 and bit 1 when a finite value leaves the fp16
// range (|v| > 65504 becomes inf: such a pass must run on the fp32 kernels in BOTH fp16 modes — inf·0 in the MFMA
// would turn the row into NaN).
__global__ __launch_bounds__(GATHER_THREADS) void csr_gather_f16_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ vals, const int32_t *__restrict__ chunk_rel, int nc1,
    const int64_t *__restrict__ row_ids, int64_t row_start, int64_t n_rows, int nchunks, int64_t Kp,
    _Float16 *__restrict__ out, int *__restrict__ inexact) {
  // TWO consecutive output rows x one index chunk (4096 columns) per workgroup, staged as fp16: 16 KiB of LDS and of
  // output per workgroup (a workgroup that writes 8 KiB runs at half the byte rate).  Rows r and r + 1 of an A16T tile
  // are 64 bytes apart, so every 32-column block of the pair leaves as one 128-byte line.
  constexpr int CH = GATHER_CH;
  __shared__ __attribute__((aligned(16))) _Float16 tile[2][CH];
  const int tid = threadIdx.x;
  const int64_t r0 = 2 * (int64_t)(blockIdx.x / nchunks);
  const int chunk = (int)(blockIdx.x % nchunks);
  const int64_t c0 = (int64_t)chunk * CH;

  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 *tile4 = reinterpret_cast<float4 *>(&tile[0][0]);   // 8 halfs
#pragma unroll
  for (int i = tid; i < 2 * CH / 8; i += GATHER_THREADS) tile4[i] = z;
  __syncthreads();
  int bad = 0;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int64_t r = r0 + rr;
    if (r < n_rows) {   // workgroup-uniform
      const int64_t row = row_ids ? row_ids[r] : row_start + r;
      const int64_t beg = indptr[row];
      const int32_t *rel = chunk_rel + row * nc1;
      const int64_t lo = beg + rel[chunk < nc1 - 1 ? chunk : nc1 - 1];
      const int64_t hi = beg + rel[chunk + 1 < nc1 - 1 ? chunk + 1 : nc1 - 1];
      for (int64_t p = lo + tid; p < hi; p += GATHER_THREADS) {
        const float v = vals[p];
        bad |= fp16_flags(v);
        tile[rr][indices[p] - (int32_t)c0] = (_Float16)v;
      }
    }
  }
  raise_fp16_flags(bad, inexact);
  __syncthreads();

  // unit u = (32-column block j of the chunk, row rr, chunk position slot): 8 consecutive units = one 128-byte line
  const int64_t kb32 = Kp >> 5;
  for (int u = tid; u < CH / 32 * 8; u += GATHER_THREADS) {
    const int j = u >> 3, rr = (u >> 2) & 1, slot = u & 3;
    const int64_t kb = (c0 >> 5) + j, r = r0 + rr;
    if (kb < kb32 && r < n_rows) {
      const int rl = (int)(r & 255);
      const int c = slot ^ a16t_swz(rl);   // the chunk that sits at this position
      float4 *dst = reinterpret_cast<float4 *>(out + (((r >> 8) * kb32 + kb) * 256 + rl) * 32 + slot * 8);
      *dst = *reinterpret_cast<const float4 *>(&tile[rr][j * 32 + c * 8]);
    }
  }
}

// ---- canonical-form check (replaces a serial host loop over every non-zero) -----------------------
// *err = min over offending rows of ((row + 1) << 2 | code), all ones = clean; code 1: indptr (not monotone, outside
// [0, nnz], not starting at 0 in row 0, not ending at nnz in the last row), 2: column outside [0, G), 3: column order
__global__ __launch_bounds__(256) void csr_validate_kernel(const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int64_t n_rows,
                                                           int64_t nnz, int G, unsigned long long *__restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per row
  if (row >= n_rows) return;
  const int64_t lo = indptr[row], hi = indptr[row + 1];
  int bad = 0;
  if (lo > hi || lo < 0 || hi > nnz || (row == 0 && lo != 0) || (row == n_rows - 1 && hi != nnz)) {
    bad = 1;
  } else {
    for (int64_t p = lo + lane; p < hi; p += 64) {
      const int32_t c = indices[p];
      if (c < 0 || c >= G) bad = 2;
      else if (p > lo && indices[p - 1] >= c) bad = bad ? bad : 3;
    }
  }
  const int code = __any(bad == 1) ? 1 : __any(bad == 2) ? 2 : __any(bad == 3) ? 3 : 0;
  if (code && lane == 0) atomicMin(err, ((unsigned long long)(row + 1) << 2) | (unsigned)code);
}

int validate_csr(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, int64_t nnz, int G) {
  if (n_rows == 0) return CC_OK;
  DevPtr err;
  CC_HIP(err.alloc(sizeof(unsigned long long)));
  CC_HIP(hipMemsetAsync(err.p, 0xff, sizeof(unsigned long long), stream()));
  hipLaunchKernelGGL(csr_validate_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, stream(), d_indptr,
                     d_indices, n_rows, nnz, G, err.as<unsigned long long>());
  CC_HIP(hipGetLastError());
  unsigned long long h = 0;
  CC_HIP(hipMemcpyAsync(&h, err.p, sizeof h, hipMemcpyDeviceToHost, stream()));
  CC_HIP(hipStreamSynchronize(stream()));
  if (h == ~0ull) return CC_OK;
  const long long row = (long long)(h >> 2) - 1;
  switch ((int)(h & 3)) {
    case 1: CC_REQUIRE(false, "indptr not monotone from 0 to nnz at row %lld", row);
    case 2: CC_REQUIRE(false, "a column of row %lld is outside [0,%d)", row, G);
    default: CC_REQUIRE(false, "columns of row %lld not strictly increasing", row);
  }
  return CC_OK;
}

int gather_chunk_count(int G) { return (int)ceil_div(G, GATHER_CH) + 1; }

int build_chunk_index(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, int G,
                      int32_t *d_chunk_rel) {
  const int nc1 = gather_chunk_count(G);
  const int64_t total = n_rows * nc1;
  if (total == 0) return CC_OK;
  hipLaunchKernelGGL(csr_chunk_index_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream(),
                     d_indptr, d_indices, n_rows, nc1, d_chunk_rel);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

int launch_gather(const int64_t *d_indptr, const int32_t *d_indices, const float *d_vals,
                  const int32_t *d_chunk_rel, const int64_t *d_row_ids, int64_t row_start, int64_t n, int G,
                  int64_t ld, float *d_out, double nnz_hint) {
  if (n == 0) return CC_OK;
  CC_REQUIRE(ld % 4 == 0 && ld >= G, "gather: ld=%lld must be a multiple of 4 and >= G=%d",
             (long long)ld, G);
  const int nchunks = (int)ceil_div(ld, GATHER_CH);
  const int nc1 = gather_chunk_count(G);
  const int64_t nblocks = n * nchunks;
  CC_REQUIRE(nblocks < (int64_t)2147483647, "gather: grid too large (%lld blocks)", (long long)nblocks);
  ProfScope prof(CC_PROF_GATHER, 8.0 * nnz_hint + 8.0 * (double)n + 4.0 * (double)ld * (double)n);
  hipLaunchKernelGGL(csr_gather_kernel, dim3((unsigned)nblocks), dim3(GATHER_THREADS), 0, stream(),
                     d_indptr, d_indices, d_vals, d_chunk_rel, nc1, d_row_ids, row_start, nchunks, ld, d_out);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

// dense fp32 rows (cc_encode_dense) -> fp16 pass buffer in the A16T layout, same exactness flags; one 16-byte chunk
// (8 columns) per thread and step
__global__ void convert_f32_f16_kernel(const float *__restrict__ x, int64_t ldx, int64_t M, int G,
                                       _Float16 *__restrict__ out, int64_t Kp, int *__restrict__ inexact) {
  using h8v = __attribute__((ext_vector_type(8))) _Float16;
  const int64_t cpr = Kp >> 3, total = M * cpr;   // chunks per row
  int bad = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / cpr;
    const int col = (int)(t % cpr) * 8;
    h8v o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = col + j < G ? x[row * ldx + col + j] : 0.f;
      bad |= fp16_flags(v);
      o[j] = (_Float16)v;
    }
    *reinterpret_cast<h8v *>(out + a16t_offset(row, col, Kp)) = o;
  }
  raise_fp16_flags(bad, inexact);
}

int launch_convert_f16(const float *d_x, int64_t ldx, int64_t M, int G, void *d_out_half, int64_t Kp, int *d_inexact) {
  if (M == 0) return CC_OK;
  CC_REQUIRE(Kp % 32 == 0 && Kp >= G, "convert: %lld columns for G=%d (a multiple of 32 >= G)", (long long)Kp, G);
  int64_t blocks = ceil_div(M * (Kp / 8), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(convert_f32_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), d_x, ldx, M, G,
                     static_cast<_Float16 *>(d_out_half), Kp, d_inexact);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

int launch_gather_f16(const int64_t *d_indptr, const int32_t *d_indices, const float *d_vals,
                      const int32_t *d_chunk_rel, const int64_t *d_row_ids, int64_t row_start, int64_t n, int G,
                      int64_t Kp, void *d_out_half, int *d_inexact, double nnz_hint) {
  if (n == 0) return CC_OK;
  // Kp = the columns of the A16T matrix the rows are written into (G rounded up to the GEMM's K-step): columns
  // [G, Kp) are written as zeros, the buffer holds round_up(n, 256) rows
  CC_REQUIRE(Kp % 32 == 0 && Kp >= G, "gather: %lld columns for G=%d (a multiple of 32 >= G)", (long long)Kp, G);
  const int nchunks = (int)ceil_div(Kp, GATHER_CH);   // one index chunk (and two rows) per workgroup
  const int nc1 = gather_chunk_count(G);
  const int64_t nblocks = ceil_div(n, 2) * nchunks;
  CC_REQUIRE(nblocks < (int64_t)2147483647, "gather: grid too large (%lld blocks)", (long long)nblocks);
  ProfScope prof(CC_PROF_GATHER, 8.0 * nnz_hint + 8.0 * (double)n + 2.0 * (double)Kp * (double)n);
  hipLaunchKernelGGL(csr_gather_f16_kernel, dim3((unsigned)nblocks), dim3(GATHER_THREADS), 0, stream(), d_indptr,
                     d_indices, d_vals, d_chunk_rel, nc1, d_row_ids, row_start, n, nchunks, Kp,
                     static_cast<_Float16 *>(d_out_half), d_inexact);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

}  // namespace cc
