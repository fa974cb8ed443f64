// Microbenchmark (GPU box): back-to-back launch cost of an (almost) empty kernel as a function of
// grid size, block size and register allocation.  Decides the workgroup shape of the MFMA kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NREG>
__global__ void empty_k(double* out, int never) {
    double r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (double)(threadIdx.x + i);
    if (never) {   // keeps NREG doubles live in registers without executing anything at run time
#pragma unroll
        for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(r[i]));
        double s = 0;
#pragma unroll
        for (int i = 0; i < NREG; ++i) s += r[i];
        out[threadIdx.x] = s;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NREG>
int run(int grid, int block) {
    double* d; CK(hipMalloc(&d, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; ++i) empty_k<NREG><<<grid, block>>>(d, 0);
    CK(hipDeviceSynchronize());
    const int reps = 2000;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) empty_k<NREG><<<grid, block>>>(d, 0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("regs(f64) %3d  grid %5d x %4d (%5d waves): %6.2f us/launch\n", NREG, grid, block, grid * block / 64, ms * 1e3 / reps);
    CK(hipFree(d));
    return 0;
}

int main() {
    for (int block : {64, 128, 256, 512, 1024})
        for (int waves : {1024, 2048, 4096}) {
            run<4>(waves * 64 / block, block);
            run<100>(waves * 64 / block, block);
        }
    return 0;
}
