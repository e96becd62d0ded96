// Duplicate-coordinate grouping (the `ds` field of an encits document) on the GPU.
//
// Reference: find_duplicate_ids (src/intercepts/db_recorder.py:113-117) compares every coordinate
// tuple with every unique one — O(N·U) in pure Python, the reference's slowest step at N >= 100k
// (SURVEY.md §8f rank 1).  Here every row gets the smallest row index that carries the same (x, y, z):
// an open-addressing hash table of row indices (2N..4N slots) is filled with atomicCAS, ties between
// equal rows are settled with atomicMin, and a second pass reads each row's representative back.
// Equality is float equality as Python's tuple comparison sees it: -0.0 == +0.0, NaN equals nothing.
#include "common.h"

namespace cc {

struct Key3 { unsigned a, b, c; };

__device__ __forceinline__ bool load_key(const float *__restrict__ xyz, int64_t i, Key3 *k) {
  float v[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  unsigned u[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (v[t] != v[t]) return false;              // NaN: never equal to anything
    u[t] = __float_as_uint(v[t] == 0.f ? 0.f : v[t]);   // -0.0 -> +0.0
  }
  k->a = u[0]; k->b = u[1]; k->c = u[2];
  return true;
}
__device__ __forceinline__ unsigned hash_key(const Key3 &k) {
  unsigned h = k.a * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + k.b * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + k.c * 0xC2B2AE3Du;
  h ^= h >> 16;
  return h * 0x27D4EB2Fu;
}
__device__ __forceinline__ bool same_key(const Key3 &x, const Key3 &y) { return x.a == y.a && x.b == y.b && x.c == y.c; }

__global__ void dup_insert_kernel(const float *__restrict__ xyz, int64_t n, int *__restrict__ table, unsigned mask,
                                  int32_t *__restrict__ rep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Key3 k;
  if (!load_key(xyz, i, &k)) { rep[i] = (int32_t)i; return; }
  unsigned h = hash_key(k) & mask;
  for (unsigned probes = 0; probes <= mask; ++probes) {
    int cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur < 0) {
      const int prev = atomicCAS(&table[h], -1, (int)i);
      if (prev < 0) return;          // claimed an empty slot
      cur = prev;
    }
    Key3 o;
    (void)load_key(xyz, cur, &o);    // rows in the table are never NaN rows
    if (same_key(k, o)) { atomicMin(&table[h], (int)i); return; }
    h = (h + 1) & mask;
  }
}

__global__ void dup_lookup_kernel(const float *__restrict__ xyz, int64_t n, const int *__restrict__ table,
                                  unsigned mask, int32_t *__restrict__ rep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Key3 k;
  if (!load_key(xyz, i, &k)) return;
  unsigned h = hash_key(k) & mask;
  for (unsigned probes = 0; probes <= mask; ++probes) {
    const int cur = table[h];
    if (cur < 0) break;              // cannot happen after the insert pass
    Key3 o;
    (void)load_key(xyz, cur, &o);
    if (same_key(k, o)) { rep[i] = cur; return; }
    h = (h + 1) & mask;
  }
  rep[i] = (int32_t)i;
}

}  // namespace cc

extern "C" int cc_duplicate_representatives(const float *host_xyz, int64_t n, int32_t *host_rep) {
  using namespace cc;
  CC_REQUIRE(initialised(), "cc_init has not been called");
  CC_REQUIRE(n >= 0 && n < (int64_t)1 << 30, "n=%lld out of range", (long long)n);
  if (n == 0) return CC_OK;
  uint64_t slots = 1024;
  while (slots < (uint64_t)n * 2) slots <<= 1;
  float *d_xyz = nullptr;
  int *d_table = nullptr;
  int32_t *d_rep = nullptr;
  int rc = CC_OK;
  auto cleanup = [&]() {
    if (d_xyz) (void)hipFree(d_xyz);
    if (d_table) (void)hipFree(d_table);
    if (d_rep) (void)hipFree(d_rep);
  };
#define CC_TRY(expr)                                                                     \
  if ((expr) != hipSuccess) {                                                            \
    set_error("%s failed in cc_duplicate_representatives", #expr);                       \
    cleanup();                                                                           \
    return CC_EHIP;                                                                      \
  }
  CC_TRY(hipMalloc(&d_xyz, sizeof(float) * 3 * (size_t)n));
  CC_TRY(hipMalloc(&d_table, sizeof(int) * (size_t)slots));
  CC_TRY(hipMalloc(&d_rep, sizeof(int32_t) * (size_t)n));
  CC_TRY(hipMemcpyAsync(d_xyz, host_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, stream()));
  CC_TRY(hipMemsetAsync(d_table, 0xff, sizeof(int) * (size_t)slots, stream()));
  const unsigned blocks = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL(dup_insert_kernel, dim3(blocks), dim3(256), 0, stream(), d_xyz, n, d_table, (unsigned)(slots - 1), d_rep);
  hipLaunchKernelGGL(dup_lookup_kernel, dim3(blocks), dim3(256), 0, stream(), d_xyz, n, d_table, (unsigned)(slots - 1), d_rep);
  CC_TRY(hipGetLastError());
  CC_TRY(hipMemcpyAsync(host_rep, d_rep, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, stream()));
  CC_TRY(hipStreamSynchronize(stream()));
#undef CC_TRY
  cleanup();
  return rc;
}
