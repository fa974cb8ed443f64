// Launch-to-launch time of back-to-back trivial kernels on one stream, by workgroup shape: how much of a short kernel's "launch
// overhead" depends on the number of waves per workgroup, on static LDS, and on the presence of a barrier.
//   hipcc -O3 --offload-arch=gfx950 tests/hip/launch_gap.hip -o tests/hip/launch_gap && tests/hip/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS, bool BAR>
__global__ void k(double* out) {
    __shared__ double sm[LDS > 0 ? LDS : 1];
    if (LDS > 0) sm[threadIdx.x] = threadIdx.x;
    if (BAR) __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (LDS > 0 ? sm[(threadIdx.x + 1) % 64] : 1.0);
}
template <int LDS, bool BAR>
static double run(int grid, int block, double* d, int reps = 2000) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<LDS, BAR>), dim3(grid), dim3(block), 0, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LDS, BAR>), dim3(grid), dim3(block), 0, 0, d);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}
int main() {
    double* d; hipMalloc(&d, 1 << 20);
    const int waves = 2000;
    for (int block : {64, 128, 256, 512}) {
        const int grid = waves * 64 / block;
        printf("%4d workgroups x %3d threads (%d waves): no LDS %.2f us, no LDS + barrier %.2f, 18 KB LDS %.2f, 18 KB LDS + barrier %.2f\n", grid, block, waves,
               run<0, false>(grid, block, d), run<0, true>(grid, block, d), run<2304, false>(grid, block, d), run<2304, true>(grid, block, d));
    }
    for (int block : {64, 128}) {
        const int grid = 999;
        printf("%4d workgroups x %3d threads: no LDS %.2f us, 18 KB LDS + barrier %.2f\n", grid, block, run<0, false>(grid, block, d), run<2304, true>(grid, block, d));
    }
    return 0;
}
