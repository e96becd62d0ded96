// C ABI of libcellcomm_hip.so — see include/cellcomm_hip.h for the contract and
// the reference interfaces each entry point stands in for.
#include <cmath>
#include <mutex>

#include <cstdlib>

#include "common.h"

namespace cc {

static thread_local std::string g_err;
static hipStream_t g_stream = nullptr;    // the library stream
static hipStream_t g_stream2 = nullptr;   // second stream: alternate passes of a short encode call run on it
static hipStream_t g_cur = nullptr;       // stream the launchers currently issue on (g_stream unless overridden)
static hipEvent_t g_slot_free[2] = {nullptr, nullptr}, g_slot_full[2] = {nullptr, nullptr};   // cc_encode_dense pipeline
static int g_device = -1;
static int g_generation = 0;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
hipStream_t stream() { return g_cur ? g_cur : g_stream; }
bool initialised() { return g_stream != nullptr; }
static bool g_two_streams = false;   // a call is running its passes on both library streams
bool two_streams_active() { return g_two_streams; }
int init_generation() { return g_generation; }
static int g_num_cus = 256;
int num_cus() { return g_num_cus; }

// ---- profiling ---------------------------------------------------------------
struct ProfRec { int klass; double work; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_pending;
static int64_t g_launches[CC_PROF_NUM];
static double g_ms[CC_PROF_NUM], g_work[CC_PROF_NUM];

// events are recycled: creating/destroying ~170 of them per 100k-cell pass is measurable
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t event_get() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(int klass_, double work_) : klass(klass_), work(work_) {
  if (!g_prof_on) return;
  e0 = event_get();
  e1 = event_get();
  if (!e0 || !e1) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, stream());
}
ProfScope::~ProfScope() {
  if (!e0 || !e1) return;
  (void)hipEventRecord(e1, stream());
  g_pending.push_back({klass, work, e0, e1});
}
static void prof_drain() {
  if (g_pending.empty()) return;
  (void)hipStreamSynchronize(g_stream);
  if (g_stream2) (void)hipStreamSynchronize(g_stream2);
  for (auto &r : g_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      g_launches[r.klass] += 1;
      g_ms[r.klass] += ms;
      g_work[r.klass] += r.work;
    }
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_pending.clear();
}

// ---- CSR handle ------------------------------------------------------------------
struct Csr {
  int64_t *indptr = nullptr;
  int32_t *indices = nullptr;
  float *vals = nullptr;
  int32_t *chunk_rel = nullptr;  // per-row non-zero offsets at every 4096-column boundary (gather.hip)
  int64_t n_rows = 0, nnz = 0;
  int G = 0;
  bool owned = false;
  SpmmStream stream;             // the non-zeros in the order the sparse first layer consumes them (built on first use)
};

static int csr_build_index(Csr *c) {
  const size_t count = (size_t)c->n_rows * (size_t)gather_chunk_count(c->G);
  CC_HIP(hipMalloc(&c->chunk_rel, sizeof(int32_t) * (count ? count : 1)));
  int rc = build_chunk_index(c->indptr, c->indices, c->n_rows, c->G, c->chunk_rel);
  if (rc) return rc;
  CC_HIP(hipStreamSynchronize(stream()));
  return CC_OK;
}

// ---- encoder handle ----------------------------------------------------------------
struct WShape { int64_t rows, cols; };

struct Encoder {
  int arch = 0, G = 0, E = 0;
  int dtype = CC_DTYPE_F32;
  PackedWeight16 L16[2];           // layers 1-2 for the fp16-MFMA modes
  _Float16 *xh[2] = {nullptr, nullptr}, *h1h[2] = {nullptr, nullptr}, *h1l[2] = {nullptr, nullptr};
  int64_t ldxh = 0, ld1h = 0;
  int *d_flag[2] = {nullptr, nullptr};  // gather raises it when a value is not an fp16 number
  // cont arch, calls longer than one pass: layer-2 outputs of ALL rows, so the three small tail
  // layers run once over n rows instead of once per pass (39 -> 3 launches at 100k cells)
  int64_t tail_rows = 0;
  float *h2_all = nullptr, *h3_all = nullptr, *h4_all = nullptr;
  // sparse first layer (cc_encoder_set_sparse_input): x @ K1 and the raw-input rows of K2 as CSR x dense products
  // on the vector ALUs; passes of 8 x `batch` rows, h1_all / h2_all ... hold one pass of it
  bool low_latency = false;        // calls of <= 128 rows: layers 1-3 as split-K (gemm_small.hip)
  bool sparse = false;
  SpmmWeight S1, S2;
  PackedWeight L2a;                // cont: the hidden-activation rows of K2 (+ b2) for the dense MFMA part of layer 2
  float *h1_all = nullptr;
  int64_t h1_rows = 0;
  DevBuf out_buf;                  // [n x E] results of a host-output call before they cross PCIe (kept across calls:
                                   // two device allocations per call were ~20 % of a batch-32 call)
  int64_t fallback_passes = 0;
  int64_t w0 = 0, w1 = 0;  // hidden widths of the first two layers
  int64_t batch = 8192;
  bool batch_user = false;   // cc_encoder_set_batch was called: the default below is not re-derived
  int batch_gen = -1;        // init generation the default was derived for (the CU count is only known after cc_init)
  std::vector<WShape> shapes;
  std::vector<std::vector<float>> w;
  std::vector<bool> have;
  bool dirty = true;
  PackedWeight L[5];
  int nlayers = 0;
  // workspace
  int64_t ws_rows = 0;
  int64_t ldx = 0, ld1 = 0, ld2 = 0, ld3 = 0;
  // two workspace slots: alternate passes of a short call run on two streams (see encode_csr_dev)
  float *x[2] = {nullptr, nullptr}, *h1[2] = {nullptr, nullptr}, *h2[2] = {nullptr, nullptr};
  float *h3[2] = {nullptr, nullptr}, *h4[2] = {nullptr, nullptr};
  int64_t *d_row_ids[2] = {nullptr, nullptr};
  int ws_slots = 0;
};

static int64_t pad32(int64_t v) { return v == 0 ? GEMM_BK : round_up(v, GEMM_BK); }  // pad to the GEMM K-step

static void encoder_shapes(Encoder *e) {
  const int64_t G = e->G, E = e->E;
  if (e->arch == CC_ARCH_CONT) {
    // reference src/bigan_cont.py:29 — int(gene_size * f) for f in [0.1, 0.05]
    e->w0 = (int64_t)((double)G * 0.1);
    e->w1 = (int64_t)((double)G * 0.05);
    e->shapes = {{G, e->w0}, {e->w0, -1}, {e->w0 + G, e->w1}, {e->w1, -1}, {e->w1, -1}, {e->w1, -1},
                 {e->w1, -1}, {e->w1, -1}, {e->w1, 150}, {150, -1}, {150, 150}, {150, -1}, {150, E}, {E, -1}};
    e->nlayers = 5;
  } else {
    // reference src/bigan_classify.py:31-39
    e->w0 = 1000;
    e->w1 = 300;
    e->shapes = {{G, 1000}, {1000, -1}, {1000, 300}, {300, -1}, {1300, 150}, {150, -1}, {150, E}, {E, -1}};
    e->nlayers = 4;
  }
  e->w.assign(e->shapes.size(), {});
  e->have.assign(e->shapes.size(), false);
}

// Rows per pass.  GEMM time is quantised in rounds of resident tiles (fp32: 128 x 128, 3 per CU; fp16: 256 x 256, 1 per
// CU), so the pass is the number of m-tiles whose layer-1 and layer-2 tile counts (weighted by their FLOP) fill whole
// rounds best, searched from half to twice the default.  At G = 30000 the cont encoder keeps the defaults (8192 rows:
// exactly 2 + 1 rounds; 32768 for the fp16 kernels: 6 + 3); the classify encoder (1000 columns = 8 n-tiles: 512 tiles
// = 2/3 of a round at 8192 rows) takes 12288, configs[1]'s width (G = 20000) 12288, configs[4]'s (13 + 7 n-tiles) 65536.
static int64_t default_pass_rows(const Encoder *e) {
  const bool f32 = e->dtype == CC_DTYPE_F32;
  const int64_t tm = f32 ? GEMM_BM : GEMM16_TILE, tn = f32 ? GEMM_BN : GEMM16_TILE;
  const int64_t round = (f32 ? 3 : 1) * (int64_t)num_cus();
  const int64_t def_mt = (f32 ? 8192 : 32768) / tm;
  if (e->w0 <= 0 || e->w1 <= 0) return def_mt * tm;
  const int64_t nt1 = ceil_div(e->w0, tn), nt2 = ceil_div(e->w1, tn);
  const double f1 = (double)e->G * (double)e->w0;
  const double f2 = (e->arch == CC_ARCH_CONT ? (double)(e->w0 + e->G) : (double)e->w0) * (double)e->w1;
  auto fill = [&](int64_t tiles) { return (double)tiles / (double)(ceil_div(tiles, round) * round); };
  auto eff = [&](int64_t mt) { return (f1 * fill(mt * nt1) + f2 * fill(mt * nt2)) / (f1 + f2); };
  int64_t best = def_mt;
  double best_eff = eff(def_mt) + 0.01;   // the default unless another size is worth more than 1 %
  for (int64_t mt = def_mt / 2; mt <= 2 * def_mt; ++mt) {
    const double v = eff(mt);
    if (v > best_eff + 1e-9) { best = mt; best_eff = v; }
  }
  return best * tm;
}

// The default rows per pass depend on the CU count of the device, which cc_encoder_create may not know yet (it does not
// require cc_init) and which changes with cc_shutdown + cc_init on another device: re-derived per init generation.
static void encoder_refresh_batch(Encoder *e) {
  if (e->batch_user || !initialised() || e->batch_gen == init_generation()) return;
  e->batch = default_pass_rows(e);
  e->batch_gen = init_generation();
}

static int encoder_pack(Encoder *e) {
  encoder_refresh_batch(e);
  for (size_t i = 0; i < e->have.size(); ++i)
    if (!e->have[i]) {
      set_error("encoder weight %zu has not been set", i);
      return CC_ESTATE;
    }
  int rc;
  if (e->arch == CC_ARCH_CONT) {
    if ((rc = pack_weight(e->w[0].data(), e->w[1].data(), e->G, 0, e->w0, &e->L[0]))) return rc;
    if ((rc = pack_weight(e->w[2].data(), e->w[3].data(), e->w0, e->G, e->w1, &e->L[1]))) return rc;
    // BatchNormalization at inference (keras defaults, epsilon = 1e-3) folded into
    // the next Dense:  (h·inv + shift) @ K3 + b3 = h @ (diag(inv) K3) + (b3 + shift @ K3)
    const int64_t w1 = e->w1;
    std::vector<float> k3(e->w[8].size()), b3(150);
    std::vector<double> inv(w1), shift(w1);
    for (int64_t i = 0; i < w1; ++i) {
      inv[i] = (double)e->w[4][i] / std::sqrt((double)e->w[7][i] + 1e-3);
      shift[i] = (double)e->w[5][i] - (double)e->w[6][i] * inv[i];
    }
    for (int64_t j = 0; j < 150; ++j) {
      double acc = (double)e->w[9][j];
      for (int64_t i = 0; i < w1; ++i) acc += shift[i] * (double)e->w[8][i * 150 + j];
      b3[j] = (float)acc;
    }
    for (int64_t i = 0; i < w1; ++i)
      for (int64_t j = 0; j < 150; ++j) k3[i * 150 + j] = (float)(inv[i] * (double)e->w[8][i * 150 + j]);
    if ((rc = pack_weight(k3.data(), b3.data(), w1, 0, 150, &e->L[2]))) return rc;
    if ((rc = pack_weight(e->w[10].data(), e->w[11].data(), 150, 0, 150, &e->L[3]))) return rc;
    if ((rc = pack_weight(e->w[12].data(), e->w[13].data(), 150, 0, e->E, &e->L[4]))) return rc;
  } else {
    if ((rc = pack_weight(e->w[0].data(), e->w[1].data(), e->G, 0, 1000, &e->L[0]))) return rc;
    if ((rc = pack_weight(e->w[2].data(), e->w[3].data(), 1000, 0, 300, &e->L[1]))) return rc;
    if ((rc = pack_weight(e->w[4].data(), e->w[5].data(), 300, 1000, 150, &e->L[2]))) return rc;
    if ((rc = pack_weight(e->w[6].data(), e->w[7].data(), 150, 0, e->E, &e->L[3]))) return rc;
  }
  if (e->sparse && e->w0 > 0 && e->w1 > 0) {
    if ((rc = pack_spmm_weight(e->w[0].data(), e->w[1].data(), e->G, e->w0, &e->S1))) return rc;
    if (e->arch == CC_ARCH_CONT) {
      // K2 rows [0, w0) multiply the hidden activation (dense MFMA), rows [w0, w0 + G) the raw counts (sparse)
      if ((rc = pack_weight(e->w[2].data(), e->w[3].data(), e->w0, 0, e->w1, &e->L2a))) return rc;
      if ((rc = pack_spmm_weight(e->w[2].data() + e->w0 * e->w1, nullptr, e->G, e->w1, &e->S2))) return rc;
    }
  }
  if (e->dtype != CC_DTYPE_F32) {
    // layers 1-2 again for the fp16 matrix cores: one plane (f16) or the hi/lo split (f16x2)
    const int planes16 = e->dtype == CC_DTYPE_F16X2 ? 2 : 1;
    if (e->arch == CC_ARCH_CONT) {
      if ((rc = pack_weight16(e->w[0].data(), e->w[1].data(), e->G, 0, e->w0, planes16, &e->L16[0]))) return rc;
      if ((rc = pack_weight16(e->w[2].data(), e->w[3].data(), e->w0, e->G, e->w1, planes16, &e->L16[1]))) return rc;
    } else {
      if ((rc = pack_weight16(e->w[0].data(), e->w[1].data(), e->G, 0, 1000, planes16, &e->L16[0]))) return rc;
    }
  }
  e->dirty = false;
  return CC_OK;
}

static int alloc_zero(float **p, int64_t rows, int64_t ld) {
  const size_t bytes = sizeof(float) * (size_t)(rows * ld);
  CC_HIP(hipMalloc(p, bytes));
  CC_HIP(hipMemsetAsync(*p, 0, bytes, stream()));
  return CC_OK;
}

static void encoder_free_ws(Encoder *e) {
  for (int sl = 0; sl < 2; ++sl) {
    _Float16 **hb[] = {&e->xh[sl], &e->h1h[sl], &e->h1l[sl]};
    for (auto b : hb) {
      if (*b) (void)hipFree(*b);
      *b = nullptr;
    }
    if (e->d_flag[sl]) (void)hipFree(e->d_flag[sl]);
    e->d_flag[sl] = nullptr;
    float **bufs[] = {&e->x[sl], &e->h1[sl], &e->h2[sl], &e->h3[sl], &e->h4[sl]};
    for (auto b : bufs) {
      if (*b) (void)hipFree(*b);
      *b = nullptr;
    }
    if (e->d_row_ids[sl]) (void)hipFree(e->d_row_ids[sl]);
    e->d_row_ids[sl] = nullptr;
  }
  e->ws_slots = 0;
  float **tb[] = {&e->h2_all, &e->h3_all, &e->h4_all, &e->h1_all};
  for (auto b : tb) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  e->tail_rows = 0;
  e->h1_rows = 0;
  e->ws_rows = 0;
}

// [n x ld] buffers for calls that keep an activation of ALL rows (tail layers once per call; the sparse path)
static int encoder_ensure_all_rows(Encoder *e, int64_t n, bool with_h1) {
  int rc;
  if (n > e->tail_rows) {
    float **tb[] = {&e->h2_all, &e->h3_all, &e->h4_all};
    for (auto b : tb) {
      if (*b) (void)hipFree(*b);
      *b = nullptr;
    }
    e->tail_rows = 0;
    if ((rc = alloc_zero(&e->h2_all, n, e->ld2))) return rc;
    if ((rc = alloc_zero(&e->h3_all, n, e->ld3))) return rc;
    if ((rc = alloc_zero(&e->h4_all, n, e->ld3))) return rc;
    e->tail_rows = n;
  }
  if (with_h1 && n > e->h1_rows) {
    if (e->h1_all) (void)hipFree(e->h1_all);
    e->h1_all = nullptr;
    e->h1_rows = 0;
    if ((rc = alloc_zero(&e->h1_all, n, e->ld1))) return rc;
    e->h1_rows = n;
  }
  return CC_OK;
}

static int encoder_tail_cont(Encoder *e, const float *h2, int64_t M, float *h3, float *h4, float *out);

// Sparse first layer: no dense copy of the counts (reference graphs src/bigan_cont.py:32-41,
// src/bigan_classify.py:31-39; see spmm.hip).  The range runs in passes of 8 x the encoder's pass size (65536 rows by
// default: 12 panels x 256 row blocks = 12 whole rounds of layer-1 workgroups), cut at absolute 256-row block
// boundaries so that no workgroup's rows belong to two passes; the activations exist for one pass only.  A row's
// result does not depend on the pass it falls into (its non-zeros are always added in column order).
static int encode_csr_sparse(Encoder *e, Csr *c, int64_t row_start, int64_t n, float *dev_out) {
  int rc;
  e->ldx = pad32(e->G); e->ld1 = pad32(e->w0); e->ld2 = pad32(e->w1); e->ld3 = pad32(150);
  const int64_t P = round_up(e->batch, SPMM_ROWS) * 8;
  if ((rc = encoder_ensure_all_rows(e, n < P ? n : P, true))) return rc;
  const double nnz_per_row = c->n_rows ? (double)c->nnz / (double)c->n_rows : 0.0;
  const int64_t end = row_start + n;
  // rows [r0, r1) of the matrix through the layer stack; `at` = their offset inside the pass buffers
  auto run_rows = [&](int64_t r0, int64_t r1, int64_t at) -> int {
    const int64_t m = r1 - r0;
    const double nnz_hint = nnz_per_row * (double)m;
    float *h1 = e->h1_all + at * e->ld1, *h2 = e->h2_all + at * e->ld2, *h3 = e->h3_all + at * e->ld3, *h4 = e->h4_all + at * e->ld3;
    float *out = dev_out + (r0 - row_start) * e->E;
    int r;
    if ((r = launch_spmm(c->stream, r0, m, e->S1, nullptr, 0, h1, e->ld1, ACT_SIGMOID, CC_PROF_SPMM_L1, nnz_hint))) return r;
    if (e->arch == CC_ARCH_CONT) {
      // layer 2 = sigmoid(h1 @ K2[:w0] + b2  (dense MFMA, linear)  +  x @ K2[w0:]  (sparse, adds the dense part, activates))
      if ((r = launch_gemm(h1, e->ld1, nullptr, 0, m, e->L2a, h2, e->ld2, ACT_LINEAR, CC_PROF_GEMM_L2))) return r;
      if ((r = launch_spmm(c->stream, r0, m, e->S2, h2, e->ld2, h2, e->ld2, ACT_SIGMOID, CC_PROF_SPMM_L2, nnz_hint))) return r;
      return encoder_tail_cont(e, h2, m, h3, h4, out);
    }
    if ((r = launch_gemm(h1, e->ld1, nullptr, 0, m, e->L[1], h2, e->ld2, ACT_SIGMOID, CC_PROF_GEMM_L2))) return r;
    if ((r = launch_gemm(h2, e->ld2, h1, e->ld1, m, e->L[2], h3, e->ld3, ACT_SIGMOID, CC_PROF_GEMM_TAIL, e->low_latency))) return r;
    if ((r = launch_gemm(h3, e->ld3, nullptr, 0, m, e->L[3], out, e->E, ACT_LINEAR, CC_PROF_GEMM_TAIL))) return r;
    return launch_row_softmax(out, m, e->E, e->E);
  };
  // Short ranges (a rank's shard at 4-8 GPUs: 12.5k rows are 49 row blocks x 12 panels = 2.3 rounds of layer-1
  // workgroups at one workgroup per CU, then 1.15 rounds for layer 2): two halves, cut at a 256-row block, on the two
  // library streams, so that the CUs one half leaves idle in its last round run workgroups of the other half.
  const int64_t rb0 = row_start / SPMM_ROWS, nrb = ceil_div(end, SPMM_ROWS) - rb0;
  const double rounds = (double)nrb * (double)e->S1.npan / (double)num_cus();
  if (n <= P && nrb >= 4 && rounds < 6.0) {
    int64_t mid = (rb0 + nrb / 2) * SPMM_ROWS;
    if (mid > row_start && mid < end) {
      CC_HIP(hipStreamSynchronize(g_stream));   // buffer memsets of encoder_ensure_all_rows are on the library stream
      struct StreamReset { ~StreamReset() { g_cur = nullptr; (void)hipStreamSynchronize(g_stream); (void)hipStreamSynchronize(g_stream2); } } guard;
      g_cur = g_stream;
      // (the launches of a half are stream-ordered; the two halves only share the read-only weights and the matrix)
      const int rc_a = run_rows(row_start, mid, 0);
      g_cur = g_stream2;
      const int rc_b = rc_a ? rc_a : run_rows(mid, end, mid - row_start);
      g_cur = nullptr;
      CC_HIP(hipStreamSynchronize(g_stream2));
      CC_HIP(hipStreamSynchronize(g_stream));
      return rc_b;
    }
  }
  for (int64_t r0 = row_start; r0 < end;) {
    const int64_t cut = r0 / SPMM_ROWS * SPMM_ROWS + P;
    const int64_t r1 = cut < end ? cut : end;
    if ((rc = run_rows(r0, r1, 0))) return rc;
    r0 = r1;
  }
  CC_HIP(hipStreamSynchronize(stream()));
  return CC_OK;
}

static int encoder_ensure_ws(Encoder *e, int64_t rows, int slots) {
  if (rows <= e->ws_rows && slots <= e->ws_slots) return CC_OK;
  if (rows < e->ws_rows) rows = e->ws_rows;
  encoder_free_ws(e);
  e->ldx = pad32(e->G);
  e->ld1 = pad32(e->w0);
  e->ld2 = pad32(e->w1);
  e->ld3 = pad32(150);
  int rc;
  for (int sl = 0; sl < slots; ++sl) {
    if ((rc = alloc_zero(&e->x[sl], rows, e->ldx))) return rc;
    if ((rc = alloc_zero(&e->h1[sl], rows, e->ld1))) return rc;
    if ((rc = alloc_zero(&e->h2[sl], rows, e->ld2))) return rc;
    if ((rc = alloc_zero(&e->h3[sl], rows, e->ld3))) return rc;
    if ((rc = alloc_zero(&e->h4[sl], rows, e->ld3))) return rc;
    CC_HIP(hipMalloc(&e->d_row_ids[sl], sizeof(int64_t) * (size_t)rows));
    if (e->dtype != CC_DTYPE_F32) {
      // fp16 pass buffers in the A16T layout (common.h): 256-row x 32-column tiles, so whole row blocks are allocated
      // (the row-stride pad that kept rows of a row-major buffer off the same memory channels went with the rows)
      e->ldxh = round_up(e->G, 64);
      e->ld1h = e->w0 == 0 ? 64 : round_up(e->w0, 64);
      const int64_t rows256 = round_up(rows, 256);
      const size_t bx = sizeof(_Float16) * (size_t)(rows256 * e->ldxh), b1 = sizeof(_Float16) * (size_t)(rows256 * e->ld1h);
      CC_HIP(hipMalloc(&e->xh[sl], bx));
      CC_HIP(hipMalloc(&e->h1h[sl], b1));
      CC_HIP(hipMalloc(&e->h1l[sl], b1));
      CC_HIP(hipMemsetAsync(e->xh[sl], 0, bx, stream()));
      CC_HIP(hipMemsetAsync(e->h1h[sl], 0, b1, stream()));
      CC_HIP(hipMemsetAsync(e->h1l[sl], 0, b1, stream()));
      CC_HIP(hipMalloc(&e->d_flag[sl], sizeof(int)));
    }
  }
  e->ws_slots = slots;
  e->ws_rows = rows;
  return CC_OK;
}

// Tail of the cont encoder (reference src/bigan_cont.py:38-41) on M rows of h2: three Dense + sigmoid layers.  Calls of
// <= 128 rows of a low-latency encoder take the two-launch form of gemm_small.hip.
static int encoder_tail_cont(Encoder *e, const float *h2, int64_t M, float *h3, float *h4, float *out) {
  int rc;
  if (e->low_latency && tail_small_applies(M, e->L[2], e->L[3], e->L[4]))
    return launch_tail_small(h2, e->ld2, M, e->L[2], e->L[3], e->L[4], out, e->E, CC_PROF_GEMM_TAIL);
  if ((rc = launch_gemm(h2, e->ld2, nullptr, 0, M, e->L[2], h3, e->ld3, ACT_SIGMOID, CC_PROF_GEMM_TAIL, e->low_latency))) return rc;
  if ((rc = launch_gemm(h3, e->ld3, nullptr, 0, M, e->L[3], h4, e->ld3, ACT_SIGMOID, CC_PROF_GEMM_TAIL))) return rc;
  return launch_gemm(h4, e->ld3, nullptr, 0, M, e->L[4], out, e->E, ACT_SIGMOID, CC_PROF_GEMM_TAIL);
}

// Layers 1-2 on the fp16 matrix cores from M rows already in e->xh[sl] (fp16); tail as fp32.
static int encoder_forward16(Encoder *e, int sl, int64_t M, float *out, float *h2_dst = nullptr) {
  float *h2 = h2_dst ? h2_dst : e->h2[sl];   // h2_dst given: stop after layer 2 (the caller runs the tail)
  int rc;
  const bool split = e->dtype == CC_DTYPE_F16X2;
  const int nt = split ? 2 : 1;
  if (e->arch == CC_ARCH_CONT) {
    const GemmSeg16 s0[1] = {{e->xh[sl], e->ldxh, 0, nt}};
    if ((rc = launch_gemm16(s0, 1, M, e->L16[0], e->h1h[sl], split ? e->h1l[sl] : nullptr, e->ld1h, 1, ACT_SIGMOID, CC_PROF_GEMM_L1))) return rc;
    GemmSeg16 s1[3];
    int n1 = 0;
    s1[n1++] = {e->h1h[sl], e->ld1h, 0, nt};
    if (split) s1[n1++] = {e->h1l[sl], e->ld1h, 0, 1};   // third product term A_lo·W_hi of the sigmoid outputs
    s1[n1++] = {e->xh[sl], e->ldxh, 1, nt};
    if ((rc = launch_gemm16(s1, n1, M, e->L16[1], h2, nullptr, e->ld2, 0, ACT_SIGMOID, CC_PROF_GEMM_L2))) return rc;
    if (h2_dst) return CC_OK;
    if ((rc = encoder_tail_cont(e, h2, M, e->h3[sl], e->h4[sl], out))) return rc;
  } else {
    const GemmSeg16 s0[1] = {{e->xh[sl], e->ldxh, 0, nt}};
    if ((rc = launch_gemm16(s0, 1, M, e->L16[0], e->h1[sl], nullptr, e->ld1, 0, ACT_SIGMOID, CC_PROF_GEMM_L1))) return rc;
    if ((rc = launch_gemm(e->h1[sl], e->ld1, nullptr, 0, M, e->L[1], e->h2[sl], e->ld2, ACT_SIGMOID, CC_PROF_GEMM_L2))) return rc;
    if ((rc = launch_gemm(e->h2[sl], e->ld2, e->h1[sl], e->ld1, M, e->L[2], e->h3[sl], e->ld3, ACT_SIGMOID, CC_PROF_GEMM_TAIL, e->low_latency))) return rc;
    if ((rc = launch_gemm(e->h3[sl], e->ld3, nullptr, 0, M, e->L[3], out, e->E, ACT_LINEAR, CC_PROF_GEMM_TAIL))) return rc;
    if ((rc = launch_row_softmax(out, M, e->E, e->E))) return rc;
  }
  return CC_OK;
}

// Layer stack on M dense rows already in e->x[sl]; writes out[M x E] (device).
static int encoder_forward(Encoder *e, int sl, int64_t M, float *out, float *h2_dst = nullptr) {
  int rc;
  float *h2 = h2_dst ? h2_dst : e->h2[sl];   // h2_dst given (cont arch only): stop after layer 2
  if (e->arch == CC_ARCH_CONT) {
    // reference src/bigan_cont.py:32-41 (Dropout = identity at inference)
    if ((rc = launch_gemm(e->x[sl], e->ldx, nullptr, 0, M, e->L[0], e->h1[sl], e->ld1, ACT_SIGMOID, CC_PROF_GEMM_L1, e->low_latency))) return rc;
    if ((rc = launch_gemm(e->h1[sl], e->ld1, e->x[sl], e->ldx, M, e->L[1], h2, e->ld2, ACT_SIGMOID, CC_PROF_GEMM_L2, e->low_latency))) return rc;
    if (h2_dst) return CC_OK;
    if ((rc = encoder_tail_cont(e, h2, M, e->h3[sl], e->h4[sl], out))) return rc;
  } else {
    // reference src/bigan_classify.py:31-39: p → h → concat[h, p] → 150 → softmax(E)
    if ((rc = launch_gemm(e->x[sl], e->ldx, nullptr, 0, M, e->L[0], e->h1[sl], e->ld1, ACT_SIGMOID, CC_PROF_GEMM_L1, e->low_latency))) return rc;
    if ((rc = launch_gemm(e->h1[sl], e->ld1, nullptr, 0, M, e->L[1], e->h2[sl], e->ld2, ACT_SIGMOID, CC_PROF_GEMM_L2, e->low_latency))) return rc;
    if ((rc = launch_gemm(e->h2[sl], e->ld2, e->h1[sl], e->ld1, M, e->L[2], e->h3[sl], e->ld3, ACT_SIGMOID, CC_PROF_GEMM_TAIL, e->low_latency))) return rc;
    if ((rc = launch_gemm(e->h3[sl], e->ld3, nullptr, 0, M, e->L[3], out, e->E, ACT_LINEAR, CC_PROF_GEMM_TAIL))) return rc;
    if ((rc = launch_row_softmax(out, M, e->E, e->E))) return rc;
  }
  return CC_OK;
}

// gather.hip / convert: bit 0 = some value is not an fp16 number, bit 1 = a finite value left the fp16 range (or NaN)
static bool fp16_pass_needs_fp32(int dtype, int flag) {
  return dtype == CC_DTYPE_F16X2 ? flag != 0 : (flag & 2) != 0;
}

static int check_rows(const Csr *c, const int64_t *row_ids, int64_t row_start, int64_t n) {
  CC_REQUIRE(n >= 0, "negative row count");
  if (row_ids) {
    for (int64_t i = 0; i < n; ++i)
      CC_REQUIRE(row_ids[i] >= 0 && row_ids[i] < c->n_rows, "row id %lld at position %lld outside [0,%lld)",
                 (long long)row_ids[i], (long long)i, (long long)c->n_rows);
  } else {
    CC_REQUIRE(row_start >= 0 && row_start + n <= c->n_rows, "rows [%lld,+%lld) outside [0,%lld)",
               (long long)row_start, (long long)n, (long long)c->n_rows);
  }
  return CC_OK;
}

static int encode_csr_dev(Encoder *e, Csr *c, const int64_t *row_ids, int64_t row_start, int64_t n,
                          float *dev_out, bool nested = false) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_REQUIRE(c->G == e->G, "matrix has %d genes, encoder expects %d", c->G, e->G);
  int rc;
  if ((rc = check_rows(c, row_ids, row_start, n))) return rc;
  encoder_refresh_batch(e);
  if (e->dirty && (rc = encoder_pack(e))) return rc;
  if (!nested) e->fallback_passes = 0;
  if (n == 0) return CC_OK;
  // contiguous rows of a matrix through the sparse first layer; sampled rows (minibatches) and degenerate widths
  // (int(G*0.1) == 0) keep the gather + dense path
  if (!nested && e->sparse && e->dtype == CC_DTYPE_F32 && !row_ids && e->w0 > 0 && e->w1 > 0) {
    if (!c->stream.d_offs && !c->stream.unusable &&
        (rc = build_spmm_stream(c->indptr, c->indices, c->vals, c->n_rows, c->G, &c->stream)))
      return rc;
    if (!c->stream.unusable) return encode_csr_sparse(e, c, row_start, n, dev_out);
  }
  // very long calls: in chunks of 32 passes, so that the all-rows buffer of the once-per-call tail (6 KB per row at
  // G = 30000) stays bounded; a row's result does not depend on the chunk or pass it falls into
  if (!nested && n > 32 * e->batch) {
    const int64_t chunk = 32 * e->batch;
    for (int64_t off = 0; off < n; off += chunk) {
      const int64_t m = n - off < chunk ? n - off : chunk;
      if ((rc = encode_csr_dev(e, c, row_ids ? row_ids + off : nullptr, row_start + off, m, dev_out + off * e->E, true)))
        return rc;
    }
    return CC_OK;
  }
  // Pass plan.  GEMM time is quantised in rounds of resident tiles (fp32: 3 per CU), and every pass ends in a partial
  // round per layer.  fp32 (round 4): ONE stream; a call of at most two default passes runs as a single pass of all its
  // rows — its whole rounds are as efficient as any, and only ONE partial round per layer is left, which launch_gemm
  // runs as half-height tiles (five resident per CU: 0.94 of a round time for a full round of them, pro rata below
  // that).  Measured against the two-stream plan of rounds 1-3 (passes cut in halves that alternate between the two
  // library streams so that each fills the other's partial rounds; tools/experiments/README.md, gemm_tail_ab): configs[1]
  // (10k x 20k) 890-920k -> 972k cells/s, 10k / 16k rows at G = 30000 22.5 / 34.4 -> 21.7 / 32.7 ms, 25k / 50k rows
  // 53.2 / 104.9 -> 53.0 / 104.0 ms; a 12.5k- or 14k-row call stays within 2 % of the old plan (27.0 vs 26.5, 29.9 vs
  // 29.7 ms: the two equal halves filled a 3.06-round product better); per-kernel timings mean one kernel again.  Longer calls keep the default pass (8192 rows at G = 30000:
  // exactly 2 + 1 rounds for layers 1 and 2) and a short last pass.  The fp16 kernels (one workgroup per CU) keep the
  // two-stream form for one short pass, below.
  struct Pass { int64_t off, m; int slot; };
  std::vector<Pass> plan;
  const int64_t full = e->batch;
  const int64_t npass = ceil_div(n, full);
  int64_t B = n < full ? n : full;
  bool dual = false;
  if (e->dtype == CC_DTYPE_F32 && n <= 2 * full) {
    // ... unless the single pass would end layer 1 in a nearly EMPTY partial round (a 12.5k-row shard at G = 30000:
    // 3.06 rounds; a lone round of 48 tiles still costs ~0.6 of a round time, half-height or not): then the rows of the
    // whole rounds form pass A and the few remaining rows pass B on the second stream, whose kernels fill the partial
    // rounds of A's layer 2.  (Starting B only behind A's layer 1, by an event, measured the same 26.9 ms of encode for
    // 12.5k rows: beside layer 1 B's workgroups push A's exactly filled rounds into a fourth one, behind it B's own
    // layer 1 -> layer 2 chain — two latency-bound tile times — ends later than A's layer 2.  The 12288 rows of pass
    // A alone take 25.0 ms.  Nor did a high-priority second stream change it — 27.0 — while it unbalanced the equal halves
    // of the sparse and fp16 short calls: 7.8 -> 8.9 and 4.1 -> 4.5 ms per 12.5k-row shard step.)
    const int64_t slots = 3 * (int64_t)num_cus(), nt1 = e->L[0].NT > 0 ? e->L[0].NT : 1;
    const int64_t round_rows = slots % nt1 == 0 ? slots / nt1 * GEMM_BM : 0;
    const int64_t A = round_rows > 0 ? n / round_rows * round_rows : 0, rest = n - A;
    if (A > 0 && rest > 0 && (double)(ceil_div(rest, GEMM_BM) * nt1) < 0.35 * (double)slots) {
      dual = true;
      B = A;
      plan.push_back({0, A, 0});
      plan.push_back({A, rest, 1});
    } else {
      B = n;
      plan.push_back({0, n, 0});
    }
  } else if (e->dtype != CC_DTYPE_F32 && npass == 1 && n >= 2 * GEMM16_TILE &&
             (double)ceil_div(n, GEMM16_TILE) * (double)e->L16[0].NT >= (double)num_cus() &&
             (double)ceil_div(n, GEMM16_TILE) * (double)e->L16[0].NT < 4.0 * (double)num_cus()) {
    // (at least one whole round of layer-1 tiles: a 512-4096-row minibatch at a small gene set is a handful of tiles,
    // which the half-height tail kernel of launch_gemm16 serves better than two streams with two gathers each)
    // fp16 kernels, one short pass (a rank's shard at 4-8 GPUs: 12.5k rows are 588 layer-1 tiles = 2.3 rounds at one
    // 256 x 256 tile per CU, then 1.15 rounds of layer 2): two halves on the two streams fill each other's last rounds
    // (x1.15-1.2 on the shard step).  Longer calls run whole rounds per pass and lose with a second stream (round 2:
    // 36.4 -> 38.1 ms at C3), so they stay on one.
    dual = true;
    const int64_t each = round_up(ceil_div(n, 2), GEMM16_TILE);
    plan.push_back({0, each, 0});
    plan.push_back({each, n - each, 1});
    B = each;
  } else {
    for (int64_t off = 0; off < n; off += B) plan.push_back({off, (n - off < B) ? n - off : B, 0});
  }
  if ((rc = encoder_ensure_ws(e, B, dual ? 2 : 1))) return rc;
  const double nnz_per_row = c->n_rows ? (double)c->nnz / (double)c->n_rows : 0.0;
  const bool tail_once = e->arch == CC_ARCH_CONT && n > B;
  if (tail_once && (rc = encoder_ensure_all_rows(e, n, false))) return rc;
  if (dual) CC_HIP(hipStreamSynchronize(g_stream));   // workspace / h2_all memsets are on the library stream
  struct StreamReset { ~StreamReset() { g_cur = nullptr; g_two_streams = false; } } stream_reset;
  g_two_streams = dual;
  for (const Pass &ps : plan) {
    const int sl = ps.slot;
    const int64_t off = ps.off, m = ps.m;
    g_cur = sl ? g_stream2 : g_stream;
    const int64_t *d_ids = nullptr;
    if (row_ids) {
      CC_HIP(hipMemcpyAsync(e->d_row_ids[sl], row_ids + off, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, stream()));
      d_ids = e->d_row_ids[sl];
    }
    bool fp32_pass = e->dtype == CC_DTYPE_F32;
    if (!fp32_pass) {
      CC_HIP(hipMemsetAsync(e->d_flag[sl], 0, sizeof(int), stream()));
      if ((rc = launch_gather_f16(c->indptr, c->indices, c->vals, c->chunk_rel, d_ids, row_start + off, m, e->G,
                                  e->ldxh, e->xh[sl], e->d_flag[sl], nnz_per_row * (double)m)))
        return rc;
      {   // f16x2: the 2-term raw-input product needs fp16-exact values; f16: values must stay inside the fp16 range
        int flag = 0;
        CC_HIP(hipMemcpyAsync(&flag, e->d_flag[sl], sizeof(int), hipMemcpyDeviceToHost, stream()));
        CC_HIP(hipStreamSynchronize(stream()));
        if (fp16_pass_needs_fp32(e->dtype, flag)) { fp32_pass = true; e->fallback_passes += 1; }
      }
      if (!fp32_pass && (rc = encoder_forward16(e, sl, m, dev_out + off * e->E, tail_once ? e->h2_all + off * e->ld2 : nullptr)))
        return rc;
    }
    if (fp32_pass) {
      if ((rc = launch_gather(c->indptr, c->indices, c->vals, c->chunk_rel, d_ids, row_start + off, m, e->G, e->ldx,
                              e->x[sl], nnz_per_row * (double)m)))
        return rc;
      if ((rc = encoder_forward(e, sl, m, dev_out + off * e->E, tail_once ? e->h2_all + off * e->ld2 : nullptr))) return rc;
    }
  }
  g_cur = nullptr;
  if (dual) CC_HIP(hipStreamSynchronize(g_stream2));   // join before the tail / before returning
  if (tail_once) {
    if ((rc = encoder_tail_cont(e, e->h2_all, n, e->h3_all, e->h4_all, dev_out))) return rc;
  }
  CC_HIP(hipStreamSynchronize(stream()));
  return CC_OK;
}

}  // namespace cc

using namespace cc;

extern "C" {

int cc_version(void) { return 106; }   // 106: CC_KNN_OPT_COLLECT_NORM_FORM; 105: CC_KNN_OPT_COLLECT_GROUPS, pageable kNN results through a pinned bounce buffer, fp32 half-height tail tiles; 104: cc_encoder_batch, rows per pass chosen from the layer widths; 103: NULL handles / buffers refused, sparse passes, wrapped indptr checked end to end (102: cc_shutdown, kNN workspace handles)
const char *cc_last_error(void) { return g_err.c_str(); }

int cc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int cc_init(int device) {
  int n = 0;
  CC_HIP(hipGetDeviceCount(&n));
  CC_REQUIRE(device >= 0 && device < n, "device %d not available (%d HIP devices visible)", device, n);
  if (g_stream) {
    if (g_device == device) return CC_OK;
    // handles, packed weights and scratch live on the first device: switching under them would launch on one
    // device with pointers of the other
    set_error("cc_init(%d): the library is initialised on device %d; call cc_shutdown() first", device, g_device);
    return CC_ESTATE;
  }
  CC_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  CC_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return CC_EHIP;
  }
  CC_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  CC_HIP(hipStreamCreateWithFlags(&g_stream2, hipStreamNonBlocking));
  g_cur = nullptr;
  g_device = device;
  g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  ++g_generation;
  return CC_OK;
}

int cc_shutdown(void) {
  if (!g_stream) return CC_OK;
  (void)hipStreamSynchronize(g_stream);
  if (g_stream2) (void)hipStreamSynchronize(g_stream2);
  prof_drain();
  g_prof_on = false;
  for (hipEvent_t ev : g_event_pool) (void)hipEventDestroy(ev);
  g_event_pool.clear();
  hipEvent_t *evs[] = {&g_slot_free[0], &g_slot_free[1], &g_slot_full[0], &g_slot_full[1]};
  for (hipEvent_t *ev : evs) {
    if (*ev) (void)hipEventDestroy(*ev);
    *ev = nullptr;
  }
  knn_default_ws_release();
  gemm_small_release();
  (void)hipStreamDestroy(g_stream);
  if (g_stream2) (void)hipStreamDestroy(g_stream2);
  g_stream = g_stream2 = g_cur = nullptr;
  g_device = -1;
  return CC_OK;
}

int cc_device_name(char *buf, int buflen) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_REQUIRE(buf != nullptr && buflen > 0, "buf is NULL or empty");
  hipDeviceProp_t prop;
  CC_HIP(hipGetDeviceProperties(&prop, g_device));
  snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return CC_OK;
}

int cc_synchronize(void) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_HIP(hipStreamSynchronize(g_stream));
  return CC_OK;
}

// ---- encoder -----------------------------------------------------------------------
int cc_encoder_create(int arch, int G, int E, int dtype, void **enc) {
  CC_REQUIRE(enc != nullptr, "enc is NULL");
  CC_REQUIRE(arch == CC_ARCH_CONT || arch == CC_ARCH_CLS, "unknown encoder arch %d", arch);
  CC_REQUIRE(dtype == CC_DTYPE_F32 || dtype == CC_DTYPE_F16 || dtype == CC_DTYPE_F16X2, "unsupported dtype %d", dtype);
  CC_REQUIRE(G >= 1 && E >= 1, "gene_size=%d and encoding_size=%d must be >= 1", G, E);
  Encoder *e = new Encoder();
  e->arch = arch; e->G = G; e->E = E; e->dtype = dtype;
  encoder_shapes(e);
  e->batch = default_pass_rows(e);
  e->batch_gen = initialised() ? init_generation() : -1;
  *enc = e;
  return CC_OK;
}

void cc_encoder_free(void *enc) {
  if (!enc) return;
  Encoder *e = static_cast<Encoder *>(enc);
  encoder_free_ws(e);
  for (auto &l : e->L) free_weight(&l);
  for (auto &l : e->L16) free_weight16(&l);
  e->out_buf.release();
  free_weight(&e->L2a);
  free_spmm_weight(&e->S1);
  free_spmm_weight(&e->S2);
  delete e;
}

#define CC_HANDLE(h, what) CC_REQUIRE((h) != nullptr, what " handle is NULL")

int cc_encoder_num_weights(void *enc) {
  if (!enc) { set_error("encoder handle is NULL"); return -1; }
  return (int)static_cast<Encoder *>(enc)->shapes.size();
}

int cc_encoder_weight_shape(void *enc, int idx, int64_t *rows, int64_t *cols) {
  CC_HANDLE(enc, "encoder");
  CC_REQUIRE(rows && cols, "rows / cols is NULL");
  Encoder *e = static_cast<Encoder *>(enc);
  CC_REQUIRE(idx >= 0 && idx < (int)e->shapes.size(), "weight index %d out of range", idx);
  *rows = e->shapes[idx].rows;
  *cols = e->shapes[idx].cols;
  return CC_OK;
}

int cc_encoder_set_weights(void *enc, int idx, const float *host, int64_t rows, int64_t cols) {
  CC_HANDLE(enc, "encoder");
  Encoder *e = static_cast<Encoder *>(enc);
  CC_REQUIRE(idx >= 0 && idx < (int)e->shapes.size(), "weight index %d out of range", idx);
  const WShape s = e->shapes[idx];
  CC_REQUIRE(rows == s.rows && cols == s.cols, "weight %d has shape (%lld,%lld), expected (%lld,%lld)", idx,
             (long long)rows, (long long)cols, (long long)s.rows, (long long)s.cols);
  const int64_t count = s.rows * (s.cols < 0 ? 1 : s.cols);
  CC_REQUIRE(host != nullptr || count == 0, "weight %d: host pointer is NULL", idx);
  e->w[idx].assign(host, host + count);
  e->have[idx] = true;
  e->dirty = true;
  return CC_OK;
}

int cc_encoder_set_sparse_input(void *enc, int on) {
  CC_HANDLE(enc, "encoder");
  Encoder *e = static_cast<Encoder *>(enc);
  CC_REQUIRE(!on || e->dtype == CC_DTYPE_F32, "the sparse first layer computes in fp32: create the encoder with CC_DTYPE_F32");
  if (e->sparse != (on != 0)) e->dirty = true;   // the panel-packed weights are made at the next pack
  e->sparse = on != 0;
  return CC_OK;
}

int cc_encoder_set_low_latency(void *enc, int on) {
  CC_HANDLE(enc, "encoder");
  static_cast<Encoder *>(enc)->low_latency = on != 0;
  return CC_OK;
}

int64_t cc_encoder_fp32_fallback_passes(void *enc) {
  if (!enc) { set_error("encoder handle is NULL"); return -1; }
  return static_cast<Encoder *>(enc)->fallback_passes;
}

int64_t cc_encoder_batch(void *enc) {
  if (!enc) { set_error("encoder handle is NULL"); return -1; }
  encoder_refresh_batch(static_cast<Encoder *>(enc));
  return static_cast<Encoder *>(enc)->batch;
}

int cc_encoder_set_batch(void *enc, int64_t batch_rows) {
  CC_HANDLE(enc, "encoder");
  CC_REQUIRE(batch_rows >= 1, "batch_rows must be >= 1");
  static_cast<Encoder *>(enc)->batch = batch_rows;
  static_cast<Encoder *>(enc)->batch_user = true;
  return CC_OK;
}

// ---- CSR -----------------------------------------------------------------------------
int cc_csr_upload(const int64_t *indptr, const int32_t *indices, const float *vals, int64_t n_rows,
                  int64_t nnz, int G, void **csr) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_REQUIRE(csr && indptr && n_rows >= 0 && nnz >= 0 && G >= 1, "bad CSR arguments");
  CC_REQUIRE(nnz == 0 || (indices && vals), "bad CSR arguments: indices / vals is NULL");
  CC_REQUIRE(indptr[0] == 0 && indptr[n_rows] == nnz, "indptr must start at 0 and end at nnz");
  Csr *c = new Csr();
  c->n_rows = n_rows; c->nnz = nnz; c->G = G; c->owned = true;
  auto fail = [&](int rc) { cc_csr_free(c); return rc; };
  if (hipMalloc(&c->indptr, sizeof(int64_t) * (size_t)(n_rows + 1)) != hipSuccess ||
      hipMalloc(&c->indices, sizeof(int32_t) * (size_t)(nnz ? nnz : 1)) != hipSuccess ||
      hipMalloc(&c->vals, sizeof(float) * (size_t)(nnz ? nnz : 1)) != hipSuccess) {
    set_error("hipMalloc failed for CSR (%lld rows, %lld nnz)", (long long)n_rows, (long long)nnz);
    return fail(CC_ENOMEM);
  }
  if (hipMemcpy(c->indptr, indptr, sizeof(int64_t) * (size_t)(n_rows + 1), hipMemcpyHostToDevice) != hipSuccess ||
      (nnz && hipMemcpy(c->indices, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess) ||
      (nnz && hipMemcpy(c->vals, vals, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess)) {
    set_error("hipMemcpy failed for CSR upload");
    return fail(CC_EHIP);
  }
  // canonical form (monotone indptr, columns strictly increasing and < G) is checked where the arrays now are
  if (int rc = validate_csr(c->indptr, c->indices, n_rows, nnz, G)) return fail(rc);
  if (int rc = csr_build_index(c)) return fail(rc);
  *csr = c;
  return CC_OK;
}

int cc_csr_wrap_device(const int64_t *dev_indptr, const int32_t *dev_indices, const float *dev_vals,
                       int64_t n_rows, int64_t nnz, int G, void **csr) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_REQUIRE(csr && dev_indptr && n_rows >= 0 && nnz >= 0 && G >= 1, "bad CSR arguments");
  CC_REQUIRE(nnz == 0 || (dev_indices && dev_vals), "bad CSR arguments: indices / vals is NULL");
  Csr *c = new Csr();
  c->indptr = const_cast<int64_t *>(dev_indptr);
  c->indices = const_cast<int32_t *>(dev_indices);
  c->vals = const_cast<float *>(dev_vals);
  c->n_rows = n_rows; c->nnz = nnz; c->G = G; c->owned = false;
  if (int rc = validate_csr(c->indptr, c->indices, n_rows, nnz, G)) { cc_csr_free(c); return rc; }
  if (int rc = csr_build_index(c)) { cc_csr_free(c); return rc; }
  *csr = c;
  return CC_OK;
}

void cc_csr_free(void *csr) {
  if (!csr) return;
  Csr *c = static_cast<Csr *>(csr);
  if (c->chunk_rel) (void)hipFree(c->chunk_rel);
  free_spmm_stream(&c->stream);
  if (c->owned) {
    if (c->indptr) (void)hipFree(c->indptr);
    if (c->indices) (void)hipFree(c->indices);
    if (c->vals) (void)hipFree(c->vals);
  }
  delete c;
}

int cc_gather_dense(void *csr, const int64_t *row_ids, int64_t row_start, int64_t n, int64_t ld, void *out,
                    int out_is_device) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_HANDLE(csr, "CSR");
  Csr *c = static_cast<Csr *>(csr);
  int rc;
  if ((rc = check_rows(c, row_ids, row_start, n))) return rc;
  CC_REQUIRE(out != nullptr || n == 0, "out is NULL");
  CC_REQUIRE(ld >= c->G && ld % 4 == 0, "ld=%lld must be a multiple of 4 and >= G=%d", (long long)ld, c->G);
  if (n == 0) return CC_OK;
  DevPtr ids_buf, out_buf;   // released on every return path
  const int64_t *d_ids = nullptr;
  float *d_out = static_cast<float *>(out);
  if (row_ids) {
    CC_HIP(ids_buf.alloc(sizeof(int64_t) * (size_t)n));
    CC_HIP(hipMemcpyAsync(ids_buf.p, row_ids, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, stream()));
    d_ids = ids_buf.as<int64_t>();
  }
  if (!out_is_device) {
    CC_HIP(out_buf.alloc(sizeof(float) * (size_t)(n * ld)));
    d_out = out_buf.as<float>();
  }
  const double nnz_hint = c->n_rows ? (double)c->nnz / (double)c->n_rows * (double)n : 0.0;
  rc = launch_gather(c->indptr, c->indices, c->vals, c->chunk_rel, d_ids, row_start, n, c->G, ld, d_out, nnz_hint);
  if (rc == CC_OK && !out_is_device)
    CC_HIP(hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)(n * ld), hipMemcpyDeviceToHost, stream()));
  CC_HIP(hipStreamSynchronize(stream()));   // also when the launch failed: the row-id copy may still be in flight
  return rc;
}

int cc_encode_dev(void *enc, void *csr, const int64_t *row_ids, int64_t row_start, int64_t n, float *dev_out) {
  CC_HANDLE(enc, "encoder");
  CC_HANDLE(csr, "CSR");
  CC_REQUIRE(dev_out != nullptr || n <= 0, "dev_out is NULL");
  return encode_csr_dev(static_cast<Encoder *>(enc), static_cast<Csr *>(csr), row_ids, row_start, n, dev_out);
}

int cc_encode(void *enc, void *csr, const int64_t *row_ids, int64_t row_start, int64_t n, float *host_out) {
  CC_HANDLE(enc, "encoder");
  CC_HANDLE(csr, "CSR");
  CC_REQUIRE(host_out != nullptr || n <= 0, "host_out is NULL");
  Encoder *e = static_cast<Encoder *>(enc);
  if (n <= 0) return encode_csr_dev(e, static_cast<Csr *>(csr), row_ids, row_start, n, nullptr);
  if (int rc = e->out_buf.alloc(sizeof(float) * (size_t)(n * e->E))) return rc;
  if (int rc = encode_csr_dev(e, static_cast<Csr *>(csr), row_ids, row_start, n, e->out_buf.as<float>())) return rc;
  CC_HIP(hipMemcpy(host_out, e->out_buf.p, sizeof(float) * (size_t)(n * e->E), hipMemcpyDeviceToHost));
  return CC_OK;
}

// Dense host rows (what predict() receives for a DataFrame / ndarray: reference src/bigan_basic.py:28-30 with
// Keras' float32 cast).  Two workspace slots form a pipeline: while the layer stack of pass i runs on the library
// stream, the rows of pass i+1 cross PCIe on the second stream into the other slot.  Pageable host memory makes
// the copy itself block the calling thread, which is why the compute of pass i is enqueued BEFORE the copy of
// pass i+1 is started; with pinned host memory both are asynchronous.  The path is bound by the host link
// (120 KB per cell at G = 30000 against ~2 us of GPU time per cell), see DESIGN.md §6.
int cc_encode_dense(void *enc, const float *host_x, int64_t n, float *host_out) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_HANDLE(enc, "encoder");
  Encoder *e = static_cast<Encoder *>(enc);
  CC_REQUIRE(n >= 0, "negative row count");
  CC_REQUIRE((host_x != nullptr && host_out != nullptr) || n == 0, "host_x / host_out is NULL");
  int rc;
  encoder_refresh_batch(e);
  if (e->dirty && (rc = encoder_pack(e))) return rc;
  e->fallback_passes = 0;
  if (n == 0) return CC_OK;
  // passes of at most 4096 rows (or the encoder's own pass size when that is smaller): 4096 x 24 tiles are one whole
  // round of the fp32 layer-1 kernel (768 resident tiles), shorter passes leave it half empty; the first copy
  // (10 ms at G = 30000) is the only one not hidden
  const int64_t cap = e->batch < 4096 ? e->batch : 4096;
  const int64_t B = n < cap ? n : cap;
  // (Round 4 tried short first and last passes — a quarter and three quarters of a pass, then a half and a quarter — to
  // shrink the exposed first copy and last kernels: 373k -> 345k cells/s at 16384 x 30000.  A 1024-row pass is a third
  // of a round of layer-1 tiles whose K loops are latency-bound, half-height or not: it costs 0.6 of a full pass.
  // Uniform passes it is.)
  std::vector<int64_t> plan_off, plan_m;
  for (int64_t off = 0; off < n; off += B) { plan_off.push_back(off); plan_m.push_back(n - off < B ? n - off : B); }
  const int64_t npass = (int64_t)plan_off.size();
  const int slots = npass > 1 ? 2 : 1;
  if ((rc = encoder_ensure_ws(e, B, slots))) return rc;
  for (int sl = 0; sl < 2; ++sl) {
    if (!g_slot_free[sl]) CC_HIP(hipEventCreateWithFlags(&g_slot_free[sl], hipEventDisableTiming));
    if (!g_slot_full[sl]) CC_HIP(hipEventCreateWithFlags(&g_slot_full[sl], hipEventDisableTiming));
  }
  if ((rc = e->out_buf.alloc(sizeof(float) * (size_t)(n * e->E)))) return rc;
  float *const d_out = e->out_buf.as<float>();
  CC_HIP(hipStreamSynchronize(g_stream));   // workspace memsets
  struct Drain {   // an early return must not leave copies or kernels running on buffers about to be freed
    ~Drain() { (void)hipStreamSynchronize(g_stream); (void)hipStreamSynchronize(g_stream2); g_cur = nullptr; }
  } drain;
  auto copy_in = [&](int64_t pass) -> int {
    const int sl = (int)(pass & 1);
    const int64_t off = plan_off[pass], m = plan_m[pass];
    if (pass >= 2) CC_HIP(hipStreamWaitEvent(g_stream2, g_slot_free[sl], 0));   // pass - 2 has left this slot
    CC_HIP(hipMemcpy2DAsync(e->x[sl], sizeof(float) * (size_t)e->ldx, host_x + off * e->G, sizeof(float) * (size_t)e->G,
                            sizeof(float) * (size_t)e->G, (size_t)m, hipMemcpyHostToDevice, g_stream2));
    CC_HIP(hipEventRecord(g_slot_full[sl], g_stream2));
    return CC_OK;
  };
  if ((rc = copy_in(0))) return rc;
  for (int64_t pass = 0; pass < npass; ++pass) {
    const int sl = (int)(pass & 1);
    const int64_t off = plan_off[pass], m = plan_m[pass];
    g_cur = g_stream;
    CC_HIP(hipStreamWaitEvent(g_stream, g_slot_full[sl], 0));
    bool fp32_pass = e->dtype == CC_DTYPE_F32;
    if (!fp32_pass) {
      int flag = 0;
      CC_HIP(hipMemsetAsync(e->d_flag[sl], 0, sizeof(int), g_stream));
      if ((rc = launch_convert_f16(e->x[sl], e->ldx, m, e->G, e->xh[sl], e->ldxh, e->d_flag[sl]))) return rc;
      CC_HIP(hipMemcpyAsync(&flag, e->d_flag[sl], sizeof(int), hipMemcpyDeviceToHost, g_stream));
      CC_HIP(hipStreamSynchronize(g_stream));
      if (fp16_pass_needs_fp32(e->dtype, flag)) { fp32_pass = true; e->fallback_passes += 1; }
      else if ((rc = encoder_forward16(e, sl, m, d_out + off * e->E))) return rc;
    }
    if (fp32_pass && (rc = encoder_forward(e, sl, m, d_out + off * e->E))) return rc;
    CC_HIP(hipEventRecord(g_slot_free[sl], g_stream));
    if (pass + 1 < npass && (rc = copy_in(pass + 1))) return rc;   // crosses PCIe underneath the kernels just enqueued
  }
  CC_HIP(hipStreamSynchronize(g_stream));
  CC_HIP(hipMemcpy(host_out, d_out, sizeof(float) * (size_t)(n * e->E), hipMemcpyDeviceToHost));
  return CC_OK;
}

// ---- kNN -------------------------------------------------------------------------------
int cc_knn_l2_dev(const float *dev_x, int64_t n_total, int d, int64_t q_begin, int64_t n_q, int k,
                  int include_self, int32_t *host_idx, float *host_dist) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  return knn_run(nullptr, dev_x, n_total, d, q_begin, n_q, k, include_self, host_idx, host_dist);
}

int cc_knn_l2(const float *host_x, int64_t n_total, int d, int k, int include_self, int32_t *host_idx,
              float *host_dist) {
  CC_REQUIRE(initialised(), "cc_init has not been called");
  return knn_run_host(nullptr, host_x, n_total, d, k, include_self, host_idx, host_dist);
}

// ---- profiling ----------------------------------------------------------------------------
int cc_profile_enable(int on) {
  prof_drain();
  g_prof_on = on != 0;
  return CC_OK;
}
int cc_profile_reset(void) {
  prof_drain();
  for (int i = 0; i < CC_PROF_NUM; ++i) { g_launches[i] = 0; g_ms[i] = 0; g_work[i] = 0; }
  return CC_OK;
}
int cc_profile_get(int klass, int64_t *launches, double *ms, double *work) {
  CC_REQUIRE(klass >= 0 && klass < CC_PROF_NUM, "profile class %d out of range", klass);
  CC_REQUIRE(launches && ms && work, "output pointer is NULL");
  prof_drain();
  *launches = g_launches[klass];
  *ms = g_ms[klass];
  *work = g_work[klass];
  return CC_OK;
}

}  // extern "C"
