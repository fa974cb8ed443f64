// Microbenchmark (GPU box): what a kernel that ONLY stores costs per launch, at the byte counts of this library's launches -- the
// practical ceiling of the store-bound kernels.  One workgroup of 256 threads per `chunk` bytes (the per-interval
// output of the workload), 512-byte non-temporal wave stores, back-to-back launches on one stream over a ring of buffers beyond twice
// the Infinity Cache.   hipcc -O3 --offload-arch=gfx950 tests/hip/fill_floor.hip -o tests/hip/fill_floor && tests/hip/fill_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill(double* __restrict__ out, long long chunk_doubles, long long total_doubles, int per_wg) {
    for (int k = 0; k < per_wg; ++k) {
        const long long c = (long long)blockIdx.x * per_wg + k;
        double* p = out + c * chunk_doubles;
        if (c * chunk_doubles >= total_doubles) return;
        const double v = (double)c;
        for (long long i = threadIdx.x; i < chunk_doubles; i += 256) __builtin_nontemporal_store(v, p + i);
    }
}

int main() {
    struct Case { const char* name; long long chunk_bytes; int n; } cases[] = {
        {"config 3 F+dF, T=1000", 42616 - 1176, 999}, {"config 3 dF+mu_d2F one call, T=1000", 42616 - 1176 + 14656 + 64, 999},
        {"config 5 mu_d2F, T=500", 9280 * 8, 499}, {"config 5 F+dF, T=500", 308040 - 4296, 499}, {"config 5 one call, T=500", 308040 - 4296 + 9280 * 8, 499},
        {"config 3 F+dF, T=8000", 42616 - 1176, 7999}, {"config 3 one call, T=8000", 42616 - 1176 + 14656 + 64, 7999},
        {"config 3 F+dF, T=32000", 42616 - 1176, 31999}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        const long long chunk = c.chunk_bytes / 8, total = chunk * c.n;
        const int nbuf = (int)((640ll << 20) / (total * 8)) + 2;
        std::vector<double*> bufs(nbuf);
        for (auto& b : bufs) CK(hipMalloc(&b, total * 8));
        for (int grid_mode = 0; grid_mode < 2; ++grid_mode) {
            const int per_wg = grid_mode == 0 ? 1 : (c.n + 1023) / 1024;           // one workgroup per interval, or a grid of <= 1024
            const int grid = (c.n + per_wg - 1) / per_wg;
            if (grid_mode == 1 && per_wg == 1) continue;
            for (int i = 0; i < 3 * nbuf; ++i) fill<<<grid, 256, 0, st>>>(bufs[i % nbuf], chunk, total, per_wg);
            CK(hipStreamSynchronize(st));
            const int reps = 200;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) fill<<<grid, 256, 0, st>>>(bufs[i % nbuf], chunk, total, per_wg);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("%-40s %8.1f MB  grid %5d x %2d intervals: %8.2f us per launch  %5.2f TB/s  (%.2f of 8 TB/s)\n", c.name, total * 8 / 1e6, grid, per_wg, us,
                   total * 8 / us / 1e6, total * 8 / us / 1e6 / 8.0);
        }
        for (auto& b : bufs) CK(hipFree(b));
    }
    return 0;
}
