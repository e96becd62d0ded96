#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <cstdarg>
extern "C" { int cc_mtx_load(const char*, void**); int cc_mtx_shape(void*, int64_t*, int64_t*, int64_t*);
int cc_mtx_export(void*, int64_t*, int32_t*, float*, int64_t*, int64_t*); void cc_mtx_free(void*); }
namespace cc { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);} }
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    void* h = nullptr; int rc = cc_mtx_load(argv[i], &h);
    if (rc) { printf("%s: rc=%d\n", argv[i], rc); continue; }
    int64_t r, g, z; cc_mtx_shape(h, &r, &g, &z);
    std::vector<int64_t> ip(r + 1), bc(r), ge(g); std::vector<int32_t> ix(z); std::vector<float> v(z);
    cc_mtx_export(h, ip.data(), ix.data(), v.data(), bc.data(), ge.data());
    printf("%s: %lld x %lld nnz %lld\n", argv[i], (long long)r, (long long)g, (long long)z);
    cc_mtx_free(h);
  }
}
