// Microbenchmark (GPU box): the floor of the config-3 F + dF launch AT THE REAL KERNEL'S GRID AND WAVE SHAPE.
// One workgroup per interval (999 workgroups), every interval writes its 5040-double Jacobian block + 140 residuals with
// 512-byte wave stores (the address pattern of qc_mfma16_pade4_kernel), 17 ring buffers as in bench.py.  Variants:
//   waves    how many waves of the workgroup issue the stores (1: the copy wave's situation, 2, 4)
//   prologue 0: stores only;  1: the storing waves first wait for the interval's loads (2 knots = 2.3 KB and the 14 KB of
//            generator images, all requested in one batch, a value derived from them is what gets stored) -- the one
//            dependent memory round trip every real interval has;  2: as 1, but ONE wave loads and hands the value over
//            through LDS behind a barrier (the real kernel's structure: copy wave loads, compute wave waits)
//   mfma     dependent f64 MFMAs (16x16x4) between the loads and the first store (8 in the real copy wave)
// Prints us per launch (back-to-back launches on one stream, hipEvents) -- compare with bench.py's step_us_stream_events.
//   hipcc -O3 --offload-arch=gfx950 tests/hip/store_floor.hip -o tests/hip/store_floor && tests/hip/store_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kJac = 5040, kF = 140, kZ = 147, kImg = 7 * 256;   // doubles

template <int WAVES, int PROLOGUE, int MFMA>
__global__ __launch_bounds__(WAVES * 64 < 128 ? 128 : WAVES * 64) void floor_kernel(const double* __restrict__ Z, const double* __restrict__ Gx,
                                                                                  double* __restrict__ F, double* __restrict__ J, int n_int,
                                                                                  unsigned long long* __restrict__ stamps) {
    __shared__ double hand[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (b >= n_int) return;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    if (stamps) ts0 = __builtin_amdgcn_s_memrealtime();
    double v = (double)b;
    if (PROLOGUE) {
        if (PROLOGUE == 1 ? (w < WAVES) : (w == 0)) {
            const double* z0 = Z + (size_t)b * kZ;
            const v2d* g = reinterpret_cast<const v2d*>(Gx) + lane;
            v2d acc = {0.0, 0.0};
            double zs = 0.0;
#pragma unroll
            for (int i = 0; i < 14; ++i) acc += g[i * 64];                 // 7 images x 2 KB, 16 B per lane per load
#pragma unroll
            for (int i = 0; i < 4; ++i) zs += z0[(lane + 64 * i) % (2 * kZ)];   // both knots
            v = acc[0] + acc[1] + zs;
            if (MFMA) {
                v4d c = {v, v, v, v};
#pragma unroll
                for (int i = 0; i < MFMA; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, c, 0, 0, 0);
                v = c[0] + c[1];
            }
            if (PROLOGUE == 2) hand[lane] = v;
        }
        if (PROLOGUE == 2) {
            __syncthreads();
            v = hand[lane];
        }
    }
    if (w >= WAVES) return;
    if (stamps) { __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(v)); ts1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
    double* Jb = J + (size_t)b * kJac;
    // 5040 doubles = 78.75 wave stores of 64 doubles: store s of the interval goes to wave s % WAVES
    for (int s = w; s < 79; s += WAVES) {
        const int i = s * 64 + lane;
        if (i < kJac) __builtin_nontemporal_store(v + s, Jb + i);
    }
    if (w == WAVES - 1)
        for (int i = lane; i < kF; i += 64) __builtin_nontemporal_store(v, F + (size_t)b * kF + i);
    if (stamps) {
        __builtin_amdgcn_sched_barrier(0);
        ts2 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && w == 0) { stamps[b * 4] = ts0; stamps[b * 4 + 1] = ts1; stamps[b * 4 + 2] = ts2; stamps[b * 4 + 3] = ts3; }
    }
}

template <int WAVES, int PROLOGUE, int MFMA>
static int run(const char* name, const double* dZ, const double* dG, std::vector<double*>& Jb, std::vector<double*>& Fb, int n_int) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 600, nbuf = (int)Jb.size();
    const int threads = WAVES * 64 < 128 ? 128 : WAVES * 64;
    for (int i = 0; i < 40; ++i) floor_kernel<WAVES, PROLOGUE, MFMA><<<n_int, threads>>>(dZ, dG, Fb[i % nbuf], Jb[i % nbuf], n_int, nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) floor_kernel<WAVES, PROLOGUE, MFMA><<<n_int, threads>>>(dZ, dG, Fb[i % nbuf], Jb[i % nbuf], n_int, nullptr);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double us = best * 1e3 / reps, bytes = (double)n_int * (kJac + kF + kZ) * 8;
    printf("%-44s %6.2f us/launch  %5.2f TB/s  (frac of 8 TB/s %.3f)\n", name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
    {   // in-kernel phases (s_memrealtime, 100 MHz) of wave 0 of every workgroup, in the middle of a run of launches
        unsigned long long* dst; CK(hipMalloc(&dst, (size_t)n_int * 4 * 8));
        for (int i = 0; i < 30; ++i) floor_kernel<WAVES, PROLOGUE, MFMA><<<n_int, threads>>>(dZ, dG, Fb[i % nbuf], Jb[i % nbuf], n_int, i == 29 ? dst : nullptr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st((size_t)n_int * 4);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        CK(hipFree(dst));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < n_int; ++b) if (st[b * 4] < t0) t0 = st[b * 4];
        const char* what[4] = {"start", "loads done", "stores issued", "drained"};
        for (int k = 0; k < 4; ++k) {
            std::vector<double> v(n_int);
            for (int b = 0; b < n_int; ++b) v[b] = (double)(st[b * 4 + k] - t0) * 0.01;
            std::sort(v.begin(), v.end());
            printf("      %-14s min %5.2f  median %5.2f  max %5.2f us\n", what[k], v[0], v[n_int / 2], v[n_int - 1]);
        }
    }
    return 0;
}

int main() {
    const int n_int = 999, nbuf = 17;
    double *dZ, *dG;
    CK(hipMalloc(&dZ, (size_t)(n_int + 1) * kZ * 8)); CK(hipMemset(dZ, 0, (size_t)(n_int + 1) * kZ * 8));
    CK(hipMalloc(&dG, kImg * 8)); CK(hipMemset(dG, 0, kImg * 8));
    std::vector<double*> Jb(nbuf), Fb(nbuf);
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&Jb[i], (size_t)n_int * kJac * 8)); CK(hipMalloc(&Fb[i], (size_t)n_int * kF * 8)); }
#define RUN(W, P, M) if (run<W, P, M>("waves " #W "  prologue " #P "  mfma " #M, dZ, dG, Jb, Fb, n_int)) return 1
    RUN(1, 0, 0); RUN(2, 0, 0); RUN(4, 0, 0);
    RUN(1, 1, 0); RUN(2, 1, 0); RUN(4, 1, 0);
    RUN(1, 1, 8); RUN(2, 1, 8); RUN(4, 1, 8);
    RUN(1, 2, 8); RUN(2, 2, 8); RUN(4, 2, 8);
    return 0;
}
