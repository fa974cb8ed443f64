// Microbenchmark (GPU box): issue interval of v_mfma_f64_16x16x4_f64 from one wave — independent
// accumulators vs one dependent accumulator chain — in shader cycles (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void rate(double* out, unsigned long long* cyc, double a, double b) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, (double)i};
    const double av = a + threadIdx.x, bv = b - threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int blocks, int threads) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * threads * 8); hipMalloc(&cyc, blocks * 8);
    rate<NACC><<<blocks, threads>>>(out, cyc, 1e-3, 2e-3);
    rate<NACC><<<blocks, threads>>>(out, cyc, 1e-3, 2e-3);
    unsigned long long h[4096];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0, mn = ~0ull;
    for (int i = 0; i < blocks; ++i) { if (h[i] > mx) mx = h[i]; if (h[i] < mn) mn = h[i]; }
    printf("%d accumulators, %4d blocks x %3d threads: %.1f .. %.1f cycles per MFMA\n", NACC, blocks, threads,
           (double)mn / (64.0 * NACC), (double)mx / (64.0 * NACC));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(1, 64); run<2>(1, 64); run<4>(1, 64); run<8>(1, 64);
    run<1>(1024, 64); run<4>(1024, 64);          // one wave per SIMD, whole chip
    run<4>(1024, 128);                            // two waves per SIMD
    return 0;
}
