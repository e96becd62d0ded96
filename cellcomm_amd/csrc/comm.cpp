// The one exchange of the sharded path at the C-ABI level: an all-gather of per-rank encodings
// over RCCL (xGMI inside a node), then the kNN of this rank's queries against all candidates
// (the reference itself has no multi-GPU code).
//
// RCCL is bound at run time (dlopen + dlsym): a process that already carries librccl.so.1 (PyTorch
// does) shares that copy, a plain C host gets the ROCm one, and nothing else in this library depends
// on it — single-GPU users never load it.  One process per GPU; the communicator is created from a
// 128-byte unique id that rank 0 makes (cc_comm_unique_id) and the host program distributes
// (file, env, MPI, torch.distributed store ...).
#include <dlfcn.h>
// Types and prototypes only: nothing here links against librccl, the entry points come from dlsym.  With the ROCm
// header present (the normal case; `make RCCL_INC=-I<dir>` for another location) the function-pointer types below are
// the header's own prototypes; without it the library still builds from the five declarations of the stable NCCL 2
// ABI restated here (the static_asserts then only check this file against itself).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
#include <hip/hip_runtime_api.h>
#include <cstddef>
extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include <cstdlib>

#include "common.h"

namespace {

// The C ABI (include/cellcomm_hip.h) promises a 128-byte id; the pointer types below are the header's own
// prototypes, so a change of RCCL's signatures or enums is a compile error here instead of a wrong call at N > 1.
constexpr int kUniqueIdBytes = 128;
static_assert(sizeof(ncclUniqueId) == kUniqueIdBytes && NCCL_UNIQUE_ID_BYTES == kUniqueIdBytes,
              "cc_comm_unique_id / cc_comm_init exchange a 128-byte id");
using UniqueId = ncclUniqueId;
constexpr ncclDataType_t kNcclFloat = ncclFloat32;
static_assert(ncclSuccess == 0, "CC_RCCL treats 0 as success");

using GetUniqueIdFn = decltype(&ncclGetUniqueId);
using CommInitRankFn = decltype(&ncclCommInitRank);   // the id travels by value
using CommDestroyFn = decltype(&ncclCommDestroy);
using AllGatherFn = decltype(&ncclAllGather);
using ErrorStringFn = decltype(&ncclGetErrorString);

struct Rccl {
  void *handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllGatherFn all_gather = nullptr;
  ErrorStringFn error_string = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return CC_OK;
  const char *override_path = std::getenv("CC_RCCL_LIB");
  const char *names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already has, if any
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  CC_REQUIRE(h != nullptr, "cannot load RCCL (librccl.so.1): %s", dlerror());
  Rccl r;
  r.handle = h;
  r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
  r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
  r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(h, "ncclAllGather"));
  r.error_string = reinterpret_cast<ErrorStringFn>(dlsym(h, "ncclGetErrorString"));
  CC_REQUIRE(r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather && r.error_string,
             "RCCL library lacks an expected entry point");
  g_rccl = r;
  return CC_OK;
}

#define CC_RCCL(expr)                                                                         \
  do {                                                                                        \
    const ncclResult_t _r = (expr);                                                           \
    if (_r != ncclSuccess) {                                                                            \
      cc::set_error("%s failed: %s", #expr, g_rccl.error_string ? g_rccl.error_string(_r) : "?"); \
      return CC_EHIP;                                                                         \
    }                                                                                         \
  } while (0)

struct Comm {
  ncclComm_t nccl = nullptr;
  int nranks = 0, rank = 0;
  float *d_send = nullptr, *d_all = nullptr;   // grow-only staging for cc_knn_l2_sharded
  size_t send_cap = 0, all_cap = 0;
};

int grow(float **p, size_t *cap, size_t bytes) {
  if (bytes <= *cap && *p) return CC_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  CC_HIP(hipMalloc(p, bytes ? bytes : 16));
  *cap = bytes;
  return CC_OK;
}

}  // namespace

extern "C" {

int cc_comm_unique_id(void *out_128_bytes) {
  CC_REQUIRE(out_128_bytes, "cc_comm_unique_id: NULL argument");
  int rc;
  if ((rc = load_rccl())) return rc;
  UniqueId id;
  CC_RCCL(g_rccl.get_unique_id(&id));
  memcpy(out_128_bytes, id.internal, kUniqueIdBytes);
  return CC_OK;
}

int cc_comm_init(int nranks, int rank, const void *unique_id_128_bytes, void **comm) {
  CC_REQUIRE(cc::initialised(), "cc_init has not been called");
  CC_REQUIRE(comm && unique_id_128_bytes, "cc_comm_init: NULL argument");
  CC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "cc_comm_init: rank %d of %d", rank, nranks);
  int rc;
  if ((rc = load_rccl())) return rc;
  UniqueId id;
  memcpy(id.internal, unique_id_128_bytes, kUniqueIdBytes);
  Comm *c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  const ncclResult_t r = g_rccl.comm_init_rank(&c->nccl, nranks, id, rank);
  if (r != ncclSuccess) {
    cc::set_error("ncclCommInitRank failed: %s", g_rccl.error_string(r));
    delete c;
    return CC_EHIP;
  }
  *comm = c;
  return CC_OK;
}

int cc_comm_free(void *comm) {
  if (!comm) return CC_OK;
  Comm *c = static_cast<Comm *>(comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_all) (void)hipFree(c->d_all);
  if (c->nccl && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->nccl);
  delete c;
  return CC_OK;
}

int cc_allgather_f32(void *comm, const float *dev_send, int64_t count, float *dev_recv) {
  CC_REQUIRE(comm && dev_send && dev_recv && count >= 0, "cc_allgather_f32: bad argument");
  Comm *c = static_cast<Comm *>(comm);
  if (count == 0) return CC_OK;
  CC_RCCL(g_rccl.all_gather(dev_send, dev_recv, (size_t)count, kNcclFloat, c->nccl, cc::stream()));
  CC_HIP(hipStreamSynchronize(cc::stream()));
  return CC_OK;
}

int cc_knn_l2_sharded(void *comm, const float *dev_local_x, int64_t n_local, int64_t n_total, int d, int k,
                      int include_self, int32_t *host_idx, float *host_dist) {
  CC_REQUIRE(comm, "cc_knn_l2_sharded: NULL communicator");
  Comm *c = static_cast<Comm *>(comm);
  CC_REQUIRE(n_total > 0 && d >= 1, "cc_knn_l2_sharded: n_total=%lld d=%d", (long long)n_total, d);
  // contiguous blocks of ceil(n_total / nranks) rows; tail ranks short or empty (pipeline.shard_bounds)
  const int64_t per = cc::ceil_div(n_total, c->nranks);
  int64_t lo = (int64_t)c->rank * per;
  lo = lo < n_total ? lo : n_total;
  const int64_t hi = lo + per < n_total ? lo + per : n_total;
  CC_REQUIRE(n_local == hi - lo, "cc_knn_l2_sharded: rank %d of %d must hold %lld of %lld rows, got %lld", c->rank,
             c->nranks, (long long)(hi - lo), (long long)n_total, (long long)n_local);
  CC_REQUIRE(n_local == 0 || dev_local_x, "cc_knn_l2_sharded: NULL encodings");
  int rc;
  const size_t shard_bytes = sizeof(float) * (size_t)per * (size_t)d;
  if ((rc = grow(&c->d_all, &c->all_cap, shard_bytes * (size_t)c->nranks))) return rc;
  const float *send = dev_local_x;
  if (n_local < per) {   // every rank contributes `per` rows: pad the short block with zeros
    if ((rc = grow(&c->d_send, &c->send_cap, shard_bytes))) return rc;
    CC_HIP(hipMemsetAsync(c->d_send, 0, shard_bytes, cc::stream()));
    if (n_local)
      CC_HIP(hipMemcpyAsync(c->d_send, dev_local_x, sizeof(float) * (size_t)n_local * (size_t)d,
                            hipMemcpyDeviceToDevice, cc::stream()));
    send = c->d_send;
  }
  CC_RCCL(g_rccl.all_gather(send, c->d_all, (size_t)per * (size_t)d, kNcclFloat, c->nccl, cc::stream()));
  // blocks are back to back for the first n_total rows (only the last non-empty block is short)
  return cc::knn_run(nullptr, c->d_all, n_total, d, lo, n_local, k, include_self, host_idx, host_dist);
}

}  // extern "C"
