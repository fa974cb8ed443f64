// Microbenchmark (GPU box): how do the stores of ONE kernel launch LAND in host memory, and can the host watch them land?
// The host-buffer entry points of the library are bound by the PCIe link (config 3: 12.7 MB of compact values per
// evaluation); sixteen chunk launches reach 36 - 47 GB/s where one launch reaches 53.  This probe answers what a one-launch
// design needs to know:
//   1. the link rate of one launch writing `n_int` blocks of `blk` doubles into pinned host memory, by grid size
//      (all workgroups resident at once, or a small persistent grid walking the intervals in order);
//   2. the ORDER in which the blocks land: the buffer is pre-filled with a sentinel, a host thread scans it while the kernel
//      runs and stamps the moment each block is complete (the data is its own completion flag, no in-kernel drain);
//   3. hipHostRegister of ordinary malloc memory: cost of the call, link rate of kernel stores into it, rate of
//      hipMemcpyAsync out of it (no pinned staging copy);
//   4. the delay between the last block landing and hipEventQuery reporting the kernel complete.
//   hipcc -O3 --offload-arch=gfx950 tests/hip/landing_probe.hip -o tests/hip/landing_probe -lpthread && tests/hip/landing_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static const unsigned long long kSentinel = 0x7FF4C0DEC0DE5A5Aull;   // a signalling NaN no arithmetic produces

// block b: `blk` doubles written with 512-byte wave stores (the kernels' pattern), value = b + 0.25
__global__ __launch_bounds__(128) void land_kernel(double* __restrict__ out, int n_int, int blk, int xcd_runs) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int vb = blockIdx.x; vb < n_int; vb += gridDim.x) {
        int b = vb;
        if (xcd_runs) {   // qc_xcd_remap: workgroups b, b + 8, ... (one XCD) take a contiguous run of intervals
            const int q = n_int >> 3, r = n_int & 7, x = vb & 7, i = vb >> 3;
            b = x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
        }
        double* p = out + (size_t)b * blk;
        const double v = (double)b + 0.25;
        for (int o = w * 64 + lane; o < blk; o += 128) __builtin_nontemporal_store(v, p + o);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool block_landed(const unsigned long long* p, int blk) {
    for (int i = 0; i < blk; ++i) if (p[i] == kSentinel) return false;
    return true;
}

int main(int argc, char** argv) {
    const int n_int = argc > 1 ? atoi(argv[1]) : 999, blk = argc > 2 ? atoi(argv[2]) : 1456 + 140;
    const size_t n = (size_t)n_int * blk;
    double* pin = nullptr;
    CK(hipHostMalloc((void**)&pin, n * 8, hipHostMallocDefault));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("landing probe: %d blocks of %d doubles = %.2f MB\n", n_int, blk, n * 8 / 1e6);

    auto run = [&](double* buf, int grid, int xcd, const char* what, int watchers) -> int {
        std::vector<double> land(n_int, 0.0);
        double best_ms = 1e9, t_launch = 0, t_done = 0;
        for (int rep = 0; rep < 4; ++rep) {
            unsigned long long* u = reinterpret_cast<unsigned long long*>(buf);
            for (size_t i = 0; i < n; ++i) u[i] = kSentinel;
            std::fill(land.begin(), land.end(), 0.0);
            std::atomic<int> go{0}, stop{0};
            std::vector<std::thread> th;
            for (int wth = 0; wth < watchers; ++wth)
                th.emplace_back([&, wth] {
                    while (!go.load()) {}
                    // watcher wth scans blocks wth, wth + watchers, ... in landing order (runs interleaved if xcd)
                    int pending = 0;
                    std::vector<int> mine;
                    for (int b = wth; b < n_int; b += watchers) mine.push_back(b);
                    pending = (int)mine.size();
                    while (pending > 0 && !stop.load()) {
                        for (int& b : mine) {
                            if (b < 0) continue;
                            if (block_landed(u + (size_t)b * blk, blk)) { land[b] = now_us(); b = -1; --pending; }
                        }
                    }
                });
            CK(hipStreamSynchronize(st));
            go.store(1);
            const double t0 = now_us();
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(land_kernel, dim3(grid), dim3(128), 0, st, buf, n_int, blk, xcd);
            CK(hipEventRecord(e1, st));
            const double t1 = now_us();
            while (hipEventQuery(e1) == hipErrorNotReady) {}
            const double t2 = now_us();
            // give the watchers 2 ms to see the rest
            while (now_us() - t2 < 2000.0) { bool all = true; for (int b = 0; b < n_int; ++b) if (land[b] == 0.0) { all = false; break; } if (all) break; }
            stop.store(1);
            for (auto& t : th) t.join();
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 0) continue;   // warm-up
            if (ms < best_ms) { best_ms = ms; t_launch = t1 - t0; t_done = t2 - t0; }
            if (rep == 3) {
                std::vector<double> rel;
                int missing = 0, inversions = 0;
                double last = 0;
                for (int b = 0; b < n_int; ++b) { if (land[b] == 0.0) ++missing; else rel.push_back(land[b] - t0); }
                for (int b = 1; b < n_int; ++b) if (land[b] && land[b - 1] && land[b] + 5.0 < land[b - 1]) ++inversions;
                std::vector<double> s = rel;
                std::sort(s.begin(), s.end());
                if (!s.empty()) last = s.back();
                auto pct = [&](double f) { return s.empty() ? 0.0 : s[std::min(s.size() - 1, (size_t)(f * s.size()))]; };
                printf("%-34s grid %4d xcd %d: kernel %.1f us = %.1f GB/s | launch returned +%.0f, event seen +%.0f us | blocks landed (host clock, %d watchers): "
                       "first +%.0f, 10%% +%.0f, 50%% +%.0f, 90%% +%.0f, last +%.0f us; unseen %d, >5us order inversions %d\n",
                       what, grid, xcd, ms * 1e3, n * 8 / (ms * 1e-3) / 1e9, t_launch, t_done, watchers, pct(0.0), pct(0.1), pct(0.5), pct(0.9), last, missing, inversions);
                // landing time by position: 8 samples along the buffer
                printf("    landing time by block index (us):");
                for (int k = 0; k < 16; ++k) { const int b = (int)((long long)k * (n_int - 1) / 15); printf(" %d:%.0f", b, land[b] ? land[b] - t0 : -1.0); }
                printf("\n");
            }
        }
        return 0;
    };

    const int grids[] = {n_int, 512, 256, 128, 64, 32, 16};
    for (int g : grids) if (run(pin, std::min(g, n_int), 1, "pinned (hipHostMalloc)", 6)) return 1;
    for (int g : {n_int, 64, 32}) if (run(pin, std::min(g, n_int), 0, "pinned, linear block map", 6)) return 1;

    // ---- hipHostRegister of ordinary memory -------------------------------------------------------------------------
    double* raw = nullptr;
    if (posix_memalign((void**)&raw, 64, n * 8 + 64)) return 1;
    double* user = raw + 3;   // deliberately not page- or line-aligned (a numpy / Julia array)
    memset(raw, 0, n * 8 + 64);
    double t0 = now_us();
    hipError_t er = hipHostRegister(user, n * 8, hipHostRegisterDefault);
    double t1 = now_us();
    printf("hipHostRegister(%.1f MB, unaligned): %s, %.0f us\n", n * 8 / 1e6, hipGetErrorString(er), t1 - t0);
    if (er == hipSuccess) {
        double* dptr = nullptr;
        CK(hipHostGetDevicePointer((void**)&dptr, user, 0));
        printf("  device pointer %s the host pointer\n", dptr == user ? "==" : "!=");
        if (run(dptr, n_int, 1, "registered malloc memory", 6)) return 1;
        if (run(dptr, 64, 1, "registered malloc memory", 6)) return 1;
        // H2D / D2H copies straight from / into registered memory vs pinned vs pageable
        double* dev = nullptr;
        const size_t zn = 147 * 1000;   // config 3's trajectory vector
        CK(hipMalloc((void**)&dev, n * 8));
        double* pageable = (double*)malloc(n * 8);
        memset(pageable, 1, n * 8);
        auto tcopy = [&](const char* what, void* dst, const void* src, size_t bytes, hipMemcpyKind k) {
            double best = 1e9;
            for (int r = 0; r < 6; ++r) {
                const double a = now_us();
                (void)hipMemcpyAsync(dst, src, bytes, k, st);
                const double b = now_us();
                (void)hipStreamSynchronize(st);
                const double c = now_us();
                if (r && c - a < best) best = c - a;
                if (r == 5) printf("  %-44s %8.2f MB: issue %.0f us, done %.0f us (best %.0f us = %.1f GB/s)\n", what, bytes / 1e6, b - a, c - a, best, bytes / best / 1e3);
            }
        };
        tcopy("H2D from registered", dev, user, zn * 8, hipMemcpyHostToDevice);
        tcopy("H2D from pinned", dev, pin, zn * 8, hipMemcpyHostToDevice);
        tcopy("H2D from pageable", dev, pageable, zn * 8, hipMemcpyHostToDevice);
        tcopy("D2H into registered", user, dev, n * 8, hipMemcpyDeviceToHost);
        tcopy("D2H into pinned", pin, dev, n * 8, hipMemcpyDeviceToHost);
        tcopy("D2H into pageable", pageable, dev, n * 8, hipMemcpyDeviceToHost);
        tcopy("D2H into registered (1.1 MB)", user, dev, 140 * 999 * 8, hipMemcpyDeviceToHost);
        tcopy("D2H into pageable (1.1 MB)", pageable, dev, 140 * 999 * 8, hipMemcpyDeviceToHost);
        t0 = now_us();
        er = hipHostUnregister(user);
        printf("hipHostUnregister: %s, %.0f us\n", hipGetErrorString(er), now_us() - t0);
        // registering again (the pages are resident now)
        t0 = now_us();
        er = hipHostRegister(user, n * 8, hipHostRegisterDefault);
        printf("hipHostRegister again: %s, %.0f us\n", hipGetErrorString(er), now_us() - t0);
        if (er == hipSuccess) (void)hipHostUnregister(user);
        // small registrations (the residual vector, 1.1 MB; the trajectory vector, 1.2 MB)
        t0 = now_us();
        er = hipHostRegister(user, 140 * 999 * 8, hipHostRegisterDefault);
        printf("hipHostRegister(1.1 MB): %s, %.0f us\n", hipGetErrorString(er), now_us() - t0);
        if (er == hipSuccess) (void)hipHostUnregister(user);
        free(pageable);
        (void)hipFree(dev);
    }
    free(raw);
    return 0;
}
