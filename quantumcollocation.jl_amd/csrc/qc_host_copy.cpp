// Streaming host-side copy used by the replication workers of the host-buffer path (qc_host_eval.cpp): the caller's value
// vector is written once per evaluation and read by the consumer later, so the destination lines are written with
// non-temporal stores (no read-for-ownership of 41 MB per config-3 evaluation).  Plain C++ (compiled by the host compiler,
// not hipcc: x86 intrinsics), run-time dispatch on the CPU's vector extension; falls back to memcpy.
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
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

// ---- landing watch (qc_host_eval.cpp): the pinned staging buffers are pre-filled with a sentinel word and the kernels' stores
// ---- are watched as they land; these are the two primitives, with the same run-time dispatch as the copies above.
namespace {

__attribute__((target("avx512f"))) size_t scan512(const double* p, size_t n, unsigned long long s) {
    const __m512i sv = _mm512_set1_epi64((long long)s);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __mmask8 k = _mm512_cmpeq_epi64_mask(_mm512_loadu_si512((const void*)(p + i)), sv);
        if (k) return i + (size_t)__builtin_ctz((unsigned)k);
    }
    for (; i < n; ++i) if (*(const volatile unsigned long long*)(p + i) == s) return i;
    return n;
}

__attribute__((target("avx2"))) size_t scan256(const double* p, size_t n, unsigned long long s) {
    const __m256i sv = _mm256_set1_epi64x((long long)s);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const int k = _mm256_movemask_pd(_mm256_castsi256_pd(_mm256_cmpeq_epi64(_mm256_loadu_si256((const __m256i*)(p + i)), sv)));
        if (k) return i + (size_t)__builtin_ctz((unsigned)k);
    }
    for (; i < n; ++i) if (*(const volatile unsigned long long*)(p + i) == s) return i;
    return n;
}

size_t scan_plain(const double* p, size_t n, unsigned long long s) {
    const volatile unsigned long long* u = (const volatile unsigned long long*)p;
    for (size_t i = 0; i < n; ++i) if (u[i] == s) return i;
    return n;
}

typedef size_t (*scan_fn)(const double*, size_t, unsigned long long);
scan_fn pick_scan() {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) return scan512;
    if (__builtin_cpu_supports("avx2")) return scan256;
    return scan_plain;
}

}  // namespace

// index of the first word of p[0 .. n) that still holds the sentinel bit pattern, n when there is none
size_t qc_host_scan(const double* p, size_t n, unsigned long long sentinel) {
    static const scan_fn fn = pick_scan();
    return fn(p, n, sentinel);
}

namespace {
__attribute__((target("avx512f"))) void fill_nt512(double* p, size_t n, unsigned long long s) {
    unsigned long long* u = (unsigned long long*)p;
    const __m512i sv = _mm512_set1_epi64((long long)s);
    size_t i = 0;
    while (i < n && ((uintptr_t)(u + i) & 63)) u[i++] = s;
    for (; i + 8 <= n; i += 8) _mm512_stream_si512((__m512i*)(u + i), sv);
    for (; i < n; ++i) u[i] = s;
}
__attribute__((target("avx2"))) void fill_nt256(double* p, size_t n, unsigned long long s) {
    unsigned long long* u = (unsigned long long*)p;
    const __m256i sv = _mm256_set1_epi64x((long long)s);
    size_t i = 0;
    while (i < n && ((uintptr_t)(u + i) & 31)) u[i++] = s;
    for (; i + 4 <= n; i += 4) _mm256_stream_si256((__m256i*)(u + i), sv);
    for (; i < n; ++i) u[i] = s;
}
void fill_plain(double* p, size_t n, unsigned long long s) {
    unsigned long long* u = (unsigned long long*)p;
    for (size_t i = 0; i < n; ++i) u[i] = s;
}
typedef void (*fill_fn)(double*, size_t, unsigned long long);
fill_fn pick_fill(int mode) {
    __builtin_cpu_init();
    if (mode == 0) return fill_plain;
    if (__builtin_cpu_supports("avx512f")) return fill_nt512;
    if (__builtin_cpu_supports("avx2")) return fill_nt256;
    return fill_plain;
}
}  // namespace

// p[0 .. n) = sentinel.  Non-temporal stores (QC_HOST_FILL_NT=0: ordinary ones): the block is written next by the GPU's copy
// engine, and a line left dirty in a CPU cache would have to be probed out of it first.
void qc_host_fill(double* p, size_t n, unsigned long long sentinel) {
    static const fill_fn fn = pick_fill(getenv("QC_HOST_FILL_NT") ? atoi(getenv("QC_HOST_FILL_NT")) : 1);
    fn(p, n, sentinel);
}
