// Workgroup dispatch rate: how long does a launch of G one-wave workgroups take when the waves do (almost) nothing?  A kernel with
// one interval per 64-thread workgroup (qc_mfma_hess*.hip beyond one device round: 7999 workgroups at BASELINE config 4 on one GPU)
// cannot go faster than this.  LDS: static bytes per workgroup; REGS: registers held per lane (an asm-visible array);
// `work` = dependent fma rounds per wave (~4 cycles each).
//   hipcc -O3 --offload-arch=gfx950 tests/hip/dispatch_rate.hip -o tests/hip/dispatch_rate && tests/hip/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS, int REGS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(double* out, int work) {
    __shared__ double sm[LDS > 0 ? LDS / 8 : 1];
    double r[REGS / 2];
#pragma unroll
    for (int i = 0; i < REGS / 2; ++i) r[i] = threadIdx.x + i;
    if (LDS > 0) sm[threadIdx.x] = r[0];
    for (int w = 0; w < work; ++w) {
#pragma unroll
        for (int i = 0; i < REGS / 2; ++i) r[i] = __builtin_fma(r[i], 1.0000001, 0.5);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < REGS / 2; ++i) s += r[i];
    if (LDS > 0) s += sm[(threadIdx.x + 1) & 63];
    if (s == 12345.678) out[blockIdx.x] = s;      // never true: keeps the work
}
template <int LDS, int REGS, int THREADS>
static double run(int grid, double* d, int work, int reps = 500) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<LDS, REGS, THREADS>), dim3(grid), dim3(THREADS), 0, 0, d, work);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LDS, REGS, THREADS>), dim3(grid), dim3(THREADS), 0, 0, d, work);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}
int main() {
    double* d; hipMalloc(&d, (size_t)64 << 20);
    for (int work : {0, 20, 100}) {
        for (int grid : {1000, 2000, 4000, 8000, 16000, 32000}) {
            printf("work %3d, grid %5d | 64 thr: 0 KB/32 regs %.2f us, 10 KB/160 regs %.2f, 14 KB/200 regs %.2f, 40 KB/200 regs %.2f | 128 thr 10 KB/100 regs, grid/2: %.2f | 256 thr 10 KB/64 regs, grid/4: %.2f\n",
                   work, grid, run<0, 32, 64>(grid, d, work), run<10240, 160, 64>(grid, d, work), run<14336, 200, 64>(grid, d, work), run<40960, 200, 64>(grid, d, work),
                   run<10240, 100, 128>(grid / 2, d, work), run<10240, 64, 256>(grid / 4, d, work));
        }
    }
    return 0;
}
