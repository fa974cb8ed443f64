// Microbenchmark (GPU box): cache-policy bits of global_store on gfx950 for the evaluator's write-once output
// (999 wave-private regions of 40 KB, the pattern of the MFMA kernel's copy waves).  All 8 combinations of sc0 sc1 nt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__device__ inline void st(double* p, double v) {
    if constexpr (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void fill_regions(double* __restrict__ p, int nwaves, int region_d, int pitch_d, double v) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    double* q = p + (size_t)wave * pitch_d;
    for (int i = lane; i < region_d; i += 64) st<MODE>(q + i, v);
}

template <int MODE>
int run(std::vector<double*>& bufs, int nint, int pitch_d, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 400, nbuf = (int)bufs.size();
    float ms, best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 20; ++i) fill_regions<MODE><<<nint, 64>>>(bufs[i % nbuf], nint, pitch_d, pitch_d, 1.0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) fill_regions<MODE><<<nint, 64>>>(bufs[i % nbuf], nint, pitch_d, pitch_d, (double)i);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-12s %6.2f us/launch (%5.2f TB/s)\n", name, best * 1e3 / reps, (double)nint * pitch_d * 8 / (best / reps * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int nint = 999, pitch_d = 5040, nbuf = 18;
    std::vector<double*> bufs(nbuf);
    for (auto& b : bufs) CK(hipMalloc(&b, (size_t)nint * pitch_d * 8));
    run<0>(bufs, nint, pitch_d, "plain");
    run<1>(bufs, nint, pitch_d, "nt");
    run<2>(bufs, nint, pitch_d, "sc0");
    run<3>(bufs, nint, pitch_d, "sc0 nt");
    run<4>(bufs, nint, pitch_d, "sc1");
    run<5>(bufs, nint, pitch_d, "sc1 nt");
    run<6>(bufs, nint, pitch_d, "sc0 sc1");
    run<7>(bufs, nint, pitch_d, "sc0 sc1 nt");
    return 0;
}
