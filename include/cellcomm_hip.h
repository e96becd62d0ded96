/*
 * cellcomm_hip.h — C ABI of libcellcomm_hip.so (MI355X / gfx950 only).
 *
 * The reference (mikemey/cellcomm) has NO FFI on this path: its seam is
 * constructor injection of an encoder factory whose product is only used
 * through `.predict(cell_data)` (reference src/bigan_basic.py:11-21,28-30),
 * and the kNN graph is built inside third-party UMAP/TSNE
 * (reference src/intercepts/plot_intercepts.py:92,176,183).  The entry points
 * below are what a ctypes stub behind that factory binds; each one names the
 * reference behaviour it replaces.  See INTEGRATION.md for the stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero (CC_E*) on error; the
 *     message of the last error on the calling thread is cc_last_error().
 *   - host buffers are caller-owned, C-contiguous, row-major.
 *   - `dev` pointers are HIP device pointers in the current process
 *     (e.g. torch.Tensor.data_ptr()); whatever produced their contents on another
 *     stream must have completed before the call (the library only orders its own work).
 *   - handles are opaque; free them with the matching *_free.  A NULL handle or a NULL caller buffer is
 *     refused with CC_EINVAL (the *_num_weights / *_fallback_passes counters return -1), *_free(NULL) is a no-op.
 *   - ONE device and ONE host thread at a time per process: cc_init binds the library to a
 *     device and owns its two HIP streams; every call returns synchronised; calls are not
 *     re-entrant (the reference's callers are single-threaded Python,
 *     reference src/intercepts/db_recorder.py:82-108).  All device state hangs off handles
 *     (encoder, generator, CSR, kNN workspace, communicator) or off the library context that
 *     cc_shutdown releases; after cc_shutdown another device may be chosen with cc_init.
 *   - there is NO CPU fallback anywhere in this library.
 */
#ifndef CELLCOMM_HIP_H
#define CELLCOMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CC_OK 0
#define CC_EINVAL 1   /* bad argument / shape mismatch            */
#define CC_EHIP 2     /* HIP runtime error (message has details)  */
#define CC_ESTATE 3   /* call order violated (e.g. weights unset) */
#define CC_ENOMEM 4

#define CC_ARCH_CONT 0 /* reference src/bigan_cont.py:28-41     */
#define CC_ARCH_CLS 1  /* reference src/bigan_classify.py:28-40 */

#define CC_DTYPE_F32 0   /* exact fp32 products (v_mfma_f32_32x32x2_f32) — the default            */
#define CC_DTYPE_F16 1   /* layers 1-2 with operands rounded to fp16, fp32 accumulate (config 5)  */
#define CC_DTYPE_F16X2 2 /* layers 1-2 on fp16 MFMA with hi+lo split operands: fp32-equivalent    */

/* ---- runtime ----------------------------------------------------------- */

/* Select the HIP device used by this process and create the library streams.
 * Fails with CC_EHIP when no gfx950 device is visible, with CC_ESTATE when the library is already
 * initialised on a DIFFERENT device (handles and scratch live there; call cc_shutdown first).
 * Re-initialising on the same device is a no-op. */
int cc_init(int device);
/* Release everything the library context owns: the streams, pooled events, the default kNN workspace.
 * Handles created by the caller must have been freed before (their device memory belongs to them).
 * No-op when not initialised. */
int cc_shutdown(void);
int cc_version(void);
const char *cc_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host); never fails. */
int cc_device_count(void);
int cc_device_name(char *buf, int buflen);
int cc_synchronize(void);

/* ---- encoder: replaces keras Model.predict behind
 *      BasicBiGan.encoding_prediction (reference src/bigan_basic.py:28-30) -- */

/* arch: CC_ARCH_CONT / CC_ARCH_CLS; G = gene_size, E = encoding_size. */
int cc_encoder_create(int arch, int G, int E, int dtype, void **enc);
void cc_encoder_free(void *enc);
/* Number of weight arrays in Keras get_weights() order (14 cont, 8 cls) and
 * the expected shape of array `idx` (cols == -1 for a vector; a kernel may
 * legitimately be [rows x 0] when int(G*0.1) == 0). */
int cc_encoder_num_weights(void *enc);
int cc_encoder_weight_shape(void *enc, int idx, int64_t *rows, int64_t *cols);
/* Upload array `idx` (Keras order: kernels are [in x out] row-major, y = x@K+b;
 * BatchNormalization contributes gamma, beta, moving_mean, moving_variance).
 * reference src/bigan_cont.py:31-41, src/bigan_classify.py:29-40. */
int cc_encoder_set_weights(void *enc, int idx, const float *host, int64_t rows, int64_t cols);
/* Rows processed per pass through the layer stack.  Default: the number of rows whose layer-1 / layer-2 tiles fill whole
 * rounds of the chip best — 8192 for CC_DTYPE_F32 and 32768 for the fp16 dtypes at G = 30000, e.g. 12288 for the
 * classify encoder or at G = 20000, 65536 for fp16 at G = 33000 (between half and twice those defaults).
 * The sparse first layer runs passes of 8 x this many rows (rounded up to whole 256-row blocks; default 65536),
 * calls of more than 32 passes are processed in chunks of 32: device memory per call is bounded by the pass size,
 * and a row's result depends neither on the pass nor on the call it falls into. */
int cc_encoder_set_batch(void *enc, int64_t batch_rows);
/* The rows per pass in force (the default chosen at creation, or the last cc_encoder_set_batch); -1 for a NULL handle. */
int64_t cc_encoder_batch(void *enc);
/* Sparse first layer (CC_DTYPE_F32 encoders only, default off).  When on, cc_encode /
 * cc_encode_dev of a CONTIGUOUS row range (row_ids == NULL) never densify the counts: x @ K1 and the raw-input
 * rows of K2 (reference src/bigan_cont.py:32-35, src/bigan_classify.py:31) run as CSR x dense products in fp32 on
 * the vector ALUs over the non-zeros only; the hidden-activation rows of K2 and the later layers stay on the fp32
 * matrix cores.  Same arithmetic type and tolerance as the dense path (summation order differs); sampled rows
 * and dense host input keep the gather + dense GEMM path.
 * The first sparse call on a CSR handle builds that matrix's non-zero stream ON THE HANDLE: a re-ordered copy of
 * the indices AND values (8 bytes per non-zero more device memory, ~62 ms per 100k cells once), kept until
 * cc_csr_free.  It is a snapshot: see cc_csr_wrap_device.  Matrices the stream cannot hold (more than ~2.1e9
 * non-zeros, more than ~154k genes) silently keep the gather + dense path. */
int cc_encoder_set_sparse_input(void *enc, int on);
/* Low-latency form for small calls (default off).  The reference calls predict on batches of 32 (keras default,
 * reference src/bigan_basic.py:30) and minibatches of 128 (src/__main__.py:44): with one row tile the two big
 * fp32 layers otherwise walk K = 30000 in 24 / 12 workgroups.  When on, calls of <= 128 rows run them (and the first
 * of the narrow layers behind them) split-K over ~1024 workgroups with a fixed-order reduction, and the cont
 * encoder's last two layers in one launch: same fp32 arithmetic and tolerance, reproducible call to call, but a
 * row's result is then no longer bit-identical to the same row inside a larger call (off keeps that invariant). */
int cc_encoder_set_low_latency(void *enc, int on);
/* fp16 dtypes: passes of the last cc_encode* call that ran on the fp32 kernels instead — CC_DTYPE_F16X2:
 * some value of the pass is not exactly an fp16 number; CC_DTYPE_F16: some value lies outside the fp16
 * range (|v| > 65504 would become inf and turn the row into NaN). */
int64_t cc_encoder_fp32_fallback_passes(void *enc);

/* ---- generator: replaces keras Model.predict behind BasicBiGan.generate_cells
 *      (reference src/bigan_basic.py:39-44; graphs src/bigan_cont.py:7-25,
 *      src/bigan_classify.py:10-25) ---------------------------------------------- */

int cc_generator_create(int arch, int G, int E, void **gen);
void cc_generator_free(void *gen);
/* Keras get_weights() order: 20 arrays (cont), 14 (cls); cols == -1 marks a vector. */
int cc_generator_num_weights(void *gen);
int cc_generator_weight_shape(void *gen, int idx, int64_t *rows, int64_t *cols);
int cc_generator_set_weights(void *gen, int idx, const float *host, int64_t rows, int64_t cols);
/* host_in[n x 2E] = [encoding_in, random_in] per row -> host_out[n x G] (before the rounding
 * that generate_cells applies). */
int cc_generate_dense(void *gen, const float *host_in, int64_t n, float *host_out);

/* ---- CSR cell x gene matrix: replaces the dense pivot of
 *      load_matrix (reference src/cell_type_training.py:9-17) ------------- */

/* indptr[n_rows+1] int64, indices[nnz] int32 (sorted, unique within a row,
 * < G), vals[nnz] float32.  Copies to the device and checks the canonical form there (one
 * kernel pass; CC_EINVAL names the first offending row). */
int cc_csr_upload(const int64_t *indptr, const int32_t *indices, const float *vals,
                  int64_t n_rows, int64_t nnz, int G, void **csr);
/* Adopt device-resident CSR arrays (not copied, not freed by cc_csr_free); the same canonical
 * form is checked (indptr monotone from 0 to nnz, columns strictly increasing per row, < G).
 * The arrays are treated as IMMUTABLE for the life of the handle: the chunk index built here and the sparse
 * first layer's stream (cc_encoder_set_sparse_input) are derived once and never refreshed, so a matrix whose
 * values or structure change in place needs a new handle (free + wrap again). */
int cc_csr_wrap_device(const int64_t *dev_indptr, const int32_t *dev_indices, const float *dev_vals,
                       int64_t n_rows, int64_t nnz, int G, void **csr);
void cc_csr_free(void *csr);

/* Densify rows into out[n x ld] float32 (ld >= G; columns [G, ld) are written
 * as zero).  row_ids == NULL means rows row_start .. row_start+n-1, otherwise
 * row_ids[n] (host int64) picks rows in that order — the minibatch of
 * CellTraining.sample_cell_data (reference src/cell_type_training.py:37-38).
 * out is a host pointer (out_is_device = 0) or a device pointer (1). */
int cc_gather_dense(void *csr, const int64_t *row_ids, int64_t row_start, int64_t n,
                    int64_t ld, void *out, int out_is_device);

/* Encoder forward for CSR rows → host_out[n x E] float32. */
int cc_encode(void *enc, void *csr, const int64_t *row_ids, int64_t row_start, int64_t n,
              float *host_out);
/* Same, result left on the device (dev_out[n x E], contiguous). */
int cc_encode_dev(void *enc, void *csr, const int64_t *row_ids, int64_t row_start, int64_t n,
                  float *dev_out);
/* Encoder forward for a dense host matrix x[n x G] (what predict() receives
 * when cell_data is a DataFrame/ndarray/list). */
int cc_encode_dense(void *enc, const float *host_x, int64_t n, float *host_out);

/* ---- matrix ingest on the host: replaces pandas read_csv + pivot_table of
 *      load_matrix (reference src/cell_type_training.py:9-17); needs no GPU ------ */

/* Parse a 10x-style MatrixMarket file (3 lines skipped, then `gene barcode value` triplets)
 * into CSR with pivot semantics: rows = sorted distinct barcodes, columns = sorted distinct
 * genes, duplicate entries averaged.  Sizes via cc_mtx_shape, arrays via cc_mtx_export
 * (indptr[n_rows+1], indices[nnz], vals[nnz], barcodes[n_rows], genes[n_genes] = original ids). */
int cc_mtx_load(const char *path, void **mtx);
int cc_mtx_shape(void *mtx, int64_t *n_rows, int64_t *n_genes, int64_t *nnz);
int cc_mtx_export(void *mtx, int64_t *indptr, int32_t *indices, float *vals, int64_t *barcodes, int64_t *genes);
void cc_mtx_free(void *mtx);

/* ---- kNN graph: no reference implementation; semantics are
 *      defined by oracle/knn_oracle.c -------------------------------------- */

/* Exact k nearest neighbours under squared L2 for queries
 * x[q_begin .. q_begin+n_q) against all n_total rows of x[n_total x d]
 * (float32).  Order: (distance asc, index asc); distance is the float64
 * sequential sum of squared float64 differences, returned rounded to float32.
 * include_self = 0 drops candidate j == query index.  idx holds GLOBAL row
 * indices (int32).  Requires k <= n_total - (include_self ? 0 : 1); supported 1 <= k <= 128,
 * 1 <= d <= 132, n_total < 2^31.  cc_knn_l2 / cc_knn_l2_dev / cc_knn_l2_sharded use the library's
 * default workspace (grow-only scratch, released by cc_shutdown; options set on it with
 * cc_knn_set_option(NULL, ..) outlive a shutdown).  host_idx / host_dist may be pageable or pinned (hipHostMalloc /
 * hipHostRegister / torch pin_memory): pinned arrays take the DMA directly, pageable ones are filled through a pinned
 * bounce buffer of the workspace (ABI v105); their pages are populated (faulted in for writing, contents unchanged)
 * while the search kernels run, so both arrays must be writable for the whole call.  The search needs about 5 KB of device scratch per query at k = 15
 * (n_q = 1 000 000: 5 GB), 12 KB at k = 50. */
int cc_knn_l2(const float *host_x, int64_t n_total, int d, int k, int include_self,
              int32_t *host_idx, float *host_dist);
int cc_knn_l2_dev(const float *dev_x, int64_t n_total, int d, int64_t q_begin, int64_t n_q,
                  int k, int include_self, int32_t *host_idx, float *host_dist);

/* A search workspace as a handle of its own: device scratch (grown on demand, kept across calls),
 * route options and the statistics of the last search.  knn == NULL in the calls below means the
 * default workspace. */
int cc_knn_create(void **knn);
void cc_knn_free(void *knn);
int cc_knn_search_dev(void *knn, const float *dev_x, int64_t n_total, int d, int64_t q_begin, int64_t n_q,
                      int k, int include_self, int32_t *host_idx, float *host_dist);
#define CC_KNN_OPT_FORCE_FALLBACK 0    /* test hook: every query through the float64 stage             */
#define CC_KNN_OPT_FORCE_FULL_SELECT 1 /* test hook: no sampled threshold, full select (route 1a)      */
#define CC_KNN_OPT_FP16_COLLECT 2      /* collect pass on the fp16 matrix cores (default 1)            */
#define CC_KNN_OPT_FP16_SELECT 3       /* sample pass on the fp16 matrix cores (default 1)             */
#define CC_KNN_OPT_FP64_PREFILTER 4    /* float64 matrix-core prefilter in stage 3 (default 1)         */
#define CC_KNN_OPT_COLLECT_SPLITS 5    /* candidate splits of the collect pass; 0 = chosen per call    */
#define CC_KNN_OPT_COARSE_COLLECT 6    /* collect as one fp16 product + fp32 re-check of survivors (1); 0: fp32 list collect */
#define CC_KNN_OPT_SAMPLE_SPLITS 7     /* candidate splits of the fp16 sample pass; 0 = chosen per call */
#define CC_KNN_OPT_SAMPLE_STRIDE 8     /* every n-th candidate block feeds the threshold; 0 = default   */
#define CC_KNN_OPT_SAMPLE_GROUPS 9     /* 32-query groups per wave of the fp16 sample pass; 0 = per call */
#define CC_KNN_OPT_COLLECT_GROUPS 10   /* 32-query groups per wave of the coarse collect (1, 2); 0 = per call    */
#define CC_KNN_OPT_COLLECT_NORM_FORM 11 /* coarse collect tile: 1 = norms in a K-step of their own, 2 = through the accumulator (ABI v106); 0 = per call */
/* None of the options changes a result bit (stage 2 certifies or stage 3 recomputes); only speed. */
int cc_knn_set_option(void *knn, int option, int value);
#define CC_KNN_STAT_FALLBACK_QUERIES 0  /* queries of the last search that stage 2 could not certify   */
#define CC_KNN_STAT_FULL_SCAN_QUERIES 1 /* of those, the ones that scanned ALL candidates in float64   */
#define CC_KNN_STAT_SAMPLE_STRIDE 2     /* every n-th candidate block was sampled for the threshold (0: no sample pass) */
#define CC_KNN_STAT_COLLECT_NORM_FORM 3  /* tile of the last coarse collect: 1 = norm K-step, 2 = accumulator form (0: none ran; ABI v106) */
int cc_knn_stat(void *knn, int stat, int64_t *value);

/* Shorthands (ABI v101) for options / statistics of the default workspace. */
/* Test hook: route every query through the float64 fallback scan. */
int cc_knn_force_fallback(int on);
/* Stage 1b collect pass on the fp16 matrix cores (default on) or the fp32 ones; results are
 * identical either way (stage 2 certifies or falls back), only speed differs. */
int cc_knn_use_fp16_collect(int on);
/* Test hook: disable the sampled-threshold route of stage 1 (always full select). */
int cc_knn_force_full_select(int on);
/* Queries of the last cc_knn_* call that stage 2 could not certify (they are re-done in float64). */
int64_t cc_knn_last_fallback_count(void);
/* Stage 3 first narrows each uncertified query's candidates with a float64 matrix-core pass
 * (default on); off = every such query scans all candidates.  Results are identical. */
int cc_knn_use_fp64_prefilter(int on);
/* Of those, the queries that had to scan ALL candidates in float64 (no finite bound from stage 2,
 * or more than 4096 candidates under it). */
int64_t cc_knn_last_full_scan_count(void);

/* ---- the sharded path's one exchange (nothing in the reference: it is single-process) ----
 * One process per GPU.  RCCL (librccl.so.1) is bound at run time on the first cc_comm_* call, so
 * single-GPU users never load it.  Rank 0 makes the 128-byte id, the host program hands it to every
 * rank (file, env, MPI, torch.distributed store ...), every rank calls cc_comm_init after cc_init. */
int cc_comm_unique_id(void *out_128_bytes);
int cc_comm_init(int nranks, int rank, const void *unique_id_128_bytes, void **comm);
int cc_comm_free(void *comm);
/* recv[nranks x count] <- every rank's send[count]; device pointers; returns when the data is in place. */
int cc_allgather_f32(void *comm, const float *dev_send, int64_t count, float *dev_recv);
/* This rank holds rows [rank*per, min(n_total, (rank+1)*per)) of the encodings, per = ceil(n_total/nranks)
 * (device pointer, row-major [n_local x d]).  All-gathers the blocks over RCCL and writes the k nearest
 * neighbours (GLOBAL row ids, same order and ties as cc_knn_l2) of this rank's rows to host_idx / host_dist
 * [n_local x k].  Collective: every rank of the communicator must call it. */
int cc_knn_l2_sharded(void *comm, const float *dev_local_x, int64_t n_local, int64_t n_total, int d, int k,
                      int include_self, int32_t *host_idx, float *host_dist);

/* ---- duplicate coordinates: the `ds` field of an encits document
 *      (reference src/intercepts/db_recorder.py:113-117, O(N·U) there) ------------- */

/* rep[i] = smallest row index j whose (x, y, z) equals row i's (float equality: -0 == +0,
 * NaN equals nothing, so a NaN row is its own representative).  xyz is host float32 [n x 3]. */
int cc_duplicate_representatives(const float *host_xyz, int64_t n, int32_t *host_rep);

/* ---- profiling ---------------------------------------------------------- */

#define CC_PROF_GATHER 0
#define CC_PROF_GEMM_L1 1
#define CC_PROF_GEMM_L2 2
#define CC_PROF_GEMM_TAIL 3
#define CC_PROF_SOFTMAX 4
#define CC_PROF_KNN_PACK 5
#define CC_PROF_KNN_SELECT 6
#define CC_PROF_KNN_RERANK 7
#define CC_PROF_KNN_FALLBACK 8
#define CC_PROF_SPMM_L1 9   /* sparse first layer: x @ K1 (work = 2 * nnz * N FLOP) */
#define CC_PROF_SPMM_L2 10  /* sparse raw-input segment of layer 2                 */
#define CC_PROF_NUM 11

/* When enabled every kernel launch is bracketed by HIP events on the library
 * stream; totals accumulate until cc_profile_reset(). */
int cc_profile_enable(int on);
int cc_profile_reset(void);
/* Per kernel class: number of launches, summed duration in ms, and the
 * algorithmic work (bytes for GATHER, FLOP otherwise) those launches did. */
int cc_profile_get(int klass, int64_t *launches, double *ms, double *work);

#ifdef __cplusplus
}
#endif
#endif
