// Launch-to-launch time of back-to-back kernels whose waves all run for a fixed time (a spin on s_memrealtime), by workgroup shape:
// is the part of a launch that lies outside its waves' lifetime (dispatch, end-of-kernel) the same for one-wave and two-wave
// workgroups, with and without a large static LDS block and a barrier?  (Round 3: the two-wave Hessian kernels had in-kernel spans
// 0.9 us shorter than the one-wave kernel and launch-to-launch times 0.4 - 0.9 us longer.)
//   hipcc -O3 --offload-arch=gfx950 tests/hip/launch_gap2.hip -o tests/hip/launch_gap2 && tests/hip/launch_gap2
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS, bool BAR>
__global__ void k(double* out, int ticks, int nstore) {
    __shared__ double sm[LDS > 0 ? LDS : 1];
    if (LDS > 0) sm[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
    if (BAR) __syncthreads();
    // nstore 512-byte stores per wave at the end (the Hessian kernel's 28 per interval)
    double* p = out + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 64 * 32 + (threadIdx.x & 63);
    const double v = LDS > 0 ? sm[(threadIdx.x + 1) % 64] : 1.0;
    for (int i = 0; i < nstore; ++i) __builtin_nontemporal_store(v + i, p + 64 * i);
}
template <int LDS, bool BAR>
static double run(int grid, int block, double* d, int ticks, int nstore, int reps = 2000) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<LDS, BAR>), dim3(grid), dim3(block), 0, 0, d, ticks, nstore);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LDS, BAR>), dim3(grid), dim3(block), 0, 0, d, ticks, nstore);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}
int main() {
    double* d; hipMalloc(&d, (size_t)64 << 20);
    for (int ticks : {0, 300, 600}) {            // 100 MHz: 0, 3, 6 us
        for (int nstore : {0, 28}) {
            printf("waves alive %.0f us, %2d stores per wave | 999 x  64 threads: no LDS %.2f us, 40 KB LDS %.2f | 999 x 128 threads: no LDS %.2f, 35 KB LDS + barrier %.2f"
                   " | 1998 x 64 threads, 20 KB LDS %.2f | 500 x 128 threads, 35 KB + barrier %.2f\n", ticks / 100.0, nstore,
                   run<0, false>(999, 64, d, ticks, nstore), run<5000, false>(999, 64, d, ticks, nstore), run<0, false>(999, 128, d, ticks, nstore),
                   run<4400, true>(999, 128, d, ticks, nstore), run<2500, false>(1998, 64, d, ticks, nstore), run<4400, true>(500, 128, d, ticks, nstore));
        }
    }
    return 0;
}
