// Host-only microbenchmark (run on the GPU box's CPU): how fast can worker threads expand the compact Jacobian form
// (one copy of -F and B per interval + the drive columns) into the caller's full value vector?  This is the part of
// qc_eval_F_jac that remains after the PCIe transfer (qc_host_eval.cpp, expand_intervals).  Config 3: 999 intervals,
// 9296 B compact -> 40320 B full per interval.
//   g++ -O3 -march=native -pthread tests/hip/host_expand_bw.cpp -o tests/hip/host_expand_bw && tests/hip/host_expand_bw
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static const int n2 = 256, copies = 8, tail = 5040 - 4096, comp_len = 2 * n2 + tail, full_len = 5040;

static inline void copy_memcpy(double* d, const double* s, size_t n) { memcpy(d, s, n * 8); }
static inline void copy_nt(double* d, const double* s, size_t n) {   // 32-byte non-temporal stores (d 32-byte aligned, n % 4 == 0)
    size_t i = 0;
    for (; i + 4 <= n; i += 4) _mm256_stream_pd(d + i, _mm256_loadu_pd(s + i));
    for (; i < n; ++i) d[i] = s[i];
}

template <void (*COPY)(double*, const double*, size_t)>
static void expand(const double* comp, double* vals, int b0, int b1) {
    for (int b = b0; b < b1; ++b) {
        const double* src = comp + (size_t)b * comp_len;
        double* dst = vals + (size_t)b * full_len;
        for (int c = 0; c < copies; ++c) COPY(dst + (size_t)c * n2, src, n2);
        for (int c = 0; c < copies; ++c) COPY(dst + (size_t)(copies + c) * n2, src + n2, n2);
        COPY(dst + 2 * copies * n2, src + 2 * n2, tail);
    }
}
// one source block held in registers-ish: read once, written 8 times with NT stores, 64-byte rows
static void expand_nt_fused(const double* comp, double* vals, int b0, int b1) {
    for (int b = b0; b < b1; ++b) {
        const double* src = comp + (size_t)b * comp_len;
        double* dst = vals + (size_t)b * full_len;
        for (int blk = 0; blk < 2; ++blk) {
            const double* s = src + blk * n2;
            double* d = dst + (size_t)blk * copies * n2;
            for (int i = 0; i < n2; i += 8) {
                const __m256d a = _mm256_loadu_pd(s + i), c = _mm256_loadu_pd(s + i + 4);
                for (int q = 0; q < copies; ++q) { _mm256_stream_pd(d + q * n2 + i, a); _mm256_stream_pd(d + q * n2 + i + 4, c); }
            }
        }
        copy_nt(dst + 2 * copies * n2, src + 2 * n2, tail);
    }
    _mm_sfence();
}

int main(int argc, char** argv) {
    const int n_int = argc > 1 ? atoi(argv[1]) : 999;
    double* comp = (double*)aligned_alloc(4096, (size_t)n_int * comp_len * 8);
    double* vals = (double*)aligned_alloc(4096, (size_t)n_int * full_len * 8);
    for (size_t i = 0; i < (size_t)n_int * comp_len; ++i) comp[i] = (double)i;
    memset(vals, 0, (size_t)n_int * full_len * 8);
    typedef void (*fn_t)(const double*, double*, int, int);
    struct { const char* name; fn_t fn; } variants[] = {{"memcpy", expand<copy_memcpy>}, {"nt-32B", expand<copy_nt>}, {"nt-fused", expand_nt_fused}};
    printf("%d intervals: %.1f MB compact -> %.1f MB full\n", n_int, n_int * comp_len * 8 / 1e6, n_int * full_len * 8 / 1e6);
    for (auto& v : variants)
        for (int nt : {1, 2, 4, 8, 12, 16, 24, 32}) {
            const int reps = 30;
            double best = 1e9, sum = 0;
            for (int r = 0; r < reps; ++r) {
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                const int per = (n_int + nt - 1) / nt;
                for (int t = 0; t < nt; ++t) th.emplace_back(v.fn, comp, vals, std::min(n_int, t * per), std::min(n_int, (t + 1) * per));
                for (auto& t : th) t.join();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (r >= 3) { best = std::min(best, ms); sum += ms; }
            }
            printf("%-9s %2d threads: best %.3f ms, mean %.3f ms  (%.1f GB/s written, thread spawn included)\n", v.name, nt, best, sum / (reps - 3),
                   n_int * full_len * 8 / 1e6 / best);
        }
    return 0;
}
