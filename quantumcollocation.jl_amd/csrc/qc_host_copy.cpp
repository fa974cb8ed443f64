// Streaming host-side copy used by the replication workers of the host-buffer path (qc_host_eval.cpp): the caller's value
// vector is written once per evaluation and read by the consumer later, so the destination lines are written with
// non-temporal stores (no read-for-ownership of 41 MB per config-3 evaluation).  Plain C++ (compiled by the host compiler,
// not hipcc: x86 intrinsics), run-time dispatch on the CPU's vector extension; falls back to memcpy.
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace {

__attribute__((target("avx512f"))) void copy_nt512(double* d, const double* s, size_t n) {
    size_t i = 0;
    while (i < n && ((uintptr_t)(d + i) & 63)) { d[i] = s[i]; ++i; }          // to the next 64-byte line
    for (; i + 8 <= n; i += 8) _mm512_stream_pd(d + i, _mm512_loadu_pd(s + i));
    for (; i < n; ++i) d[i] = s[i];
}

__attribute__((target("avx2"))) void copy_nt256(double* d, const double* s, size_t n) {
    size_t i = 0;
    while (i < n && ((uintptr_t)(d + i) & 31)) { d[i] = s[i]; ++i; }
    for (; i + 4 <= n; i += 4) _mm256_stream_pd(d + i, _mm256_loadu_pd(s + i));
    for (; i < n; ++i) d[i] = s[i];
}

void copy_plain(double* d, const double* s, size_t n) { memcpy(d, s, n * sizeof(double)); }

typedef void (*copy_fn)(double*, const double*, size_t);

copy_fn pick(int mode) {
    __builtin_cpu_init();
    if (mode == 0) return copy_plain;
    if (mode != 2 && __builtin_cpu_supports("avx512f")) return copy_nt512;
    if (__builtin_cpu_supports("avx2")) return copy_nt256;
    return copy_plain;
}

}  // namespace

// mode: 0 memcpy, 1 widest non-temporal form the CPU has, 2 32-byte non-temporal stores
void qc_host_copy_select(int mode, void (**fn)(double*, const double*, size_t)) { *fn = pick(mode); }
void qc_host_copy_fence() { _mm_sfence(); }
