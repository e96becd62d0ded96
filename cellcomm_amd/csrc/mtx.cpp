// Streaming 10x-style MatrixMarket triplets -> CSR on the host (no GPU involved).
//
// Replaces the pandas read_csv + pivot_table of load_matrix (reference
// src/cell_type_training.py:9-17), which is the wall-clock and memory limiter in front of the
// GPU path: the pivot materialises a dense float64 N x G table (24 GB at
// 100k x 30k).  Same semantics:
//   * exactly 3 lines are skipped, whatever they contain; the declared dimensions are ignored;
//   * every following non-empty line is `gene barcode p` separated by blanks/tabs;
//   * rows = sorted distinct barcodes present, columns = sorted distinct genes present;
//   * duplicate (barcode, gene) entries are averaged (pivot_table's default aggfunc='mean'),
//     the mean is formed in float64 and stored as float32; absent entries are 0.
// The file is memory-mapped and parsed in one pass with a hand-rolled number scanner; triplets
// that already arrive ordered by (barcode, gene) — the usual 10x layout — are not sorted again.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

struct Triplet {
  int64_t gene, barcode;
  double p;
};

struct Mtx {
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<float> vals;
  std::vector<int64_t> barcodes, genes;   // original ids of rows / columns
};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// integer fast path, anything else (sign/decimal point/exponent) through strtod
inline bool parse_number(const char *&s, const char *end, double *out, bool *is_int, int64_t *ival) {
  const char *p = s;
  bool neg = false;
  if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
  const char *digits = p;
  int64_t v = 0;
  while (p < end && *p >= '0' && *p <= '9' && p - digits < 18) { v = v * 10 + (*p - '0'); ++p; }
  if (p > digits && (p == end || is_blank(*p) || *p == '\n')) {
    *ival = neg ? -v : v;
    *out = (double)*ival;
    *is_int = true;
    s = p;
    return true;
  }
  char buf[64];
  const char *q = s;
  size_t n = 0;
  while (q < end && !is_blank(*q) && *q != '\n' && n + 1 < sizeof buf) buf[n++] = *q++;
  buf[n] = 0;
  char *e = nullptr;
  const double d = std::strtod(buf, &e);
  if (e == buf || *e != 0) return false;
  *out = d;
  *is_int = false;
  *ival = 0;
  s = q;
  return true;
}

// Distinct ids in ascending order plus a way to look up an id's position.  10x ids are small dense
// non-negative integers: a presence table (O(n), no sort) covers them; anything else is sorted.
struct IdMap {
  std::vector<int64_t> uniq;
  std::vector<int32_t> rank;   // rank[id - lo] = position (dense case only)
  int64_t lo = 0;
  bool dense = false;
  int32_t pos(int64_t id) const {
    if (dense) return rank[(size_t)(id - lo)];
    return (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), id) - uniq.begin());
  }
};

template <class Get>
void build_idmap(size_t n, Get get, IdMap *m) {
  if (n == 0) return;
  int64_t lo = get(0), hi = get(0);
  for (size_t i = 1; i < n; ++i) { const int64_t v = get(i); lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  const unsigned __int128 span = (unsigned __int128)((__int128)hi - (__int128)lo) + 1;
  if (span <= (unsigned __int128)(1u << 28) && span <= (unsigned __int128)n * 64 + 1024) {
    m->dense = true;
    m->lo = lo;
    m->rank.assign((size_t)span, 0);
    for (size_t i = 0; i < n; ++i) m->rank[(size_t)(get(i) - lo)] = 1;
    int32_t next = 0;
    for (size_t v = 0; v < (size_t)span; ++v)
      if (m->rank[v]) { m->uniq.push_back(lo + (int64_t)v); m->rank[v] = next++; }
    return;
  }
  std::vector<int64_t> ids(n);
  for (size_t i = 0; i < n; ++i) ids[i] = get(i);
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  m->uniq.swap(ids);
}

}  // namespace

extern "C" {

int cc_mtx_load(const char *path, void **handle) {
  CC_REQUIRE(path && handle, "cc_mtx_load: NULL argument");
  const int fd = open(path, O_RDONLY);
  CC_REQUIRE(fd >= 0, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); cc::set_error("cannot stat %s", path); return CC_EINVAL; }
  const size_t size = (size_t)st.st_size;
  const char *data = size ? static_cast<const char *>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0)) : "";
  close(fd);
  if (size && data == MAP_FAILED) { cc::set_error("cannot map %s", path); return CC_EINVAL; }
  const char *s = data, *end = data + size;
  for (int skipped = 0; skipped < 3 && s < end; ++skipped) {   // skiprows=3
    const char *nl = static_cast<const char *>(memchr(s, '\n', (size_t)(end - s)));
    s = nl ? nl + 1 : end;
  }
  std::vector<Triplet> t;
  t.reserve((size_t)(end - s) / 12 + 16);
  int64_t line = 3;
  bool ordered = true;
  int rc = CC_OK;
  while (s < end) {
    ++line;
    while (s < end && is_blank(*s)) ++s;
    if (s < end && *s == '\n') { ++s; continue; }   // blank line
    if (s >= end) break;
    Triplet tr;
    double d;
    bool is_int;
    int64_t iv;
    bool ok = parse_number(s, end, &d, &is_int, &iv) && is_int;
    tr.gene = iv;
    while (s < end && is_blank(*s)) ++s;
    ok = ok && parse_number(s, end, &d, &is_int, &iv) && is_int;
    tr.barcode = iv;
    while (s < end && is_blank(*s)) ++s;
    ok = ok && parse_number(s, end, &d, &is_int, &iv);
    tr.p = d;
    while (s < end && is_blank(*s)) ++s;
    if (!ok || (s < end && *s != '\n')) {
      cc::set_error("%s line %lld: expected `gene barcode value`", path, (long long)line);
      rc = CC_EINVAL;
      break;
    }
    if (s < end) ++s;
    if (!t.empty()) {
      const Triplet &pv = t.back();
      if (tr.barcode < pv.barcode || (tr.barcode == pv.barcode && tr.gene <= pv.gene)) ordered = false;
    }
    t.push_back(tr);
  }
  if (size) munmap(const_cast<char *>(data), size);
  if (rc) return rc;

  Mtx *m = new Mtx();
  IdMap gmap, bmap;
  build_idmap(t.size(), [&](size_t i) { return t[i].gene; }, &gmap);
  build_idmap(t.size(), [&](size_t i) { return t[i].barcode; }, &bmap);
  m->genes = gmap.uniq;
  m->barcodes = bmap.uniq;
  if (m->genes.size() > (size_t)2147483647 || m->barcodes.size() > (size_t)2147483647) {
    delete m;
    cc::set_error("%s: too many distinct genes or barcodes", path);
    return CC_EINVAL;
  }
  if (!ordered)
    std::sort(t.begin(), t.end(), [](const Triplet &a, const Triplet &b) {
      return a.barcode < b.barcode || (a.barcode == b.barcode && a.gene < b.gene);
    });
  const size_t n_rows = m->barcodes.size();
  m->indptr.assign(n_rows + 1, 0);
  m->indices.reserve(t.size());
  m->vals.reserve(t.size());
  for (size_t i = 0; i < t.size();) {
    size_t j = i;
    double sum = 0.0;
    while (j < t.size() && t[j].barcode == t[i].barcode && t[j].gene == t[i].gene) sum += t[j++].p;
    const size_t row = (size_t)bmap.pos(t[i].barcode);
    const int32_t col = gmap.pos(t[i].gene);
    m->indices.push_back(col);
    m->vals.push_back((float)(sum / (double)(j - i)));
    m->indptr[row + 1] += 1;
    i = j;
  }
  for (size_t r = 0; r < n_rows; ++r) m->indptr[r + 1] += m->indptr[r];
  *handle = m;
  return CC_OK;
}

int cc_mtx_shape(void *handle, int64_t *n_rows, int64_t *n_genes, int64_t *nnz) {
  CC_REQUIRE(handle, "cc_mtx_shape: NULL handle");
  CC_REQUIRE(n_rows && n_genes && nnz, "cc_mtx_shape: NULL output pointer");
  const Mtx *m = static_cast<const Mtx *>(handle);
  *n_rows = (int64_t)m->barcodes.size();
  *n_genes = (int64_t)m->genes.size();
  *nnz = (int64_t)m->indices.size();
  return CC_OK;
}

int cc_mtx_export(void *handle, int64_t *indptr, int32_t *indices, float *vals, int64_t *barcodes, int64_t *genes) {
  CC_REQUIRE(handle, "cc_mtx_export: NULL handle");
  CC_REQUIRE(indptr && indices && vals && barcodes && genes, "cc_mtx_export: NULL output pointer");
  const Mtx *m = static_cast<const Mtx *>(handle);
  memcpy(indptr, m->indptr.data(), sizeof(int64_t) * m->indptr.size());
  if (!m->indices.empty()) {
    memcpy(indices, m->indices.data(), sizeof(int32_t) * m->indices.size());
    memcpy(vals, m->vals.data(), sizeof(float) * m->vals.size());
  }
  if (!m->barcodes.empty()) memcpy(barcodes, m->barcodes.data(), sizeof(int64_t) * m->barcodes.size());
  if (!m->genes.empty()) memcpy(genes, m->genes.data(), sizeof(int64_t) * m->genes.size());
  return CC_OK;
}

void cc_mtx_free(void *handle) { delete static_cast<Mtx *>(handle); }

}  // extern "C"
