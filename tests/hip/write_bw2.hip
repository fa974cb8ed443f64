// Microbenchmark (GPU box): does the ACCESS PATTERN of the MFMA kernel's copy waves limit the store rate?
// pattern 0: grid-strided fill (each wave-instruction 512 contiguous bytes, neighbouring waves neighbouring chunks)
// pattern 1: every wave owns one contiguous region of `region` bytes and writes it front to back
//            (what one copy wave does with its 32 KB of B/F copies, one region per 42 KB interval block)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill_strided(double* __restrict__ p, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, p + i);
}
// nwaves regions, region_d doubles written, regions pitch_d doubles apart
__global__ void fill_regions(double* __restrict__ p, int nwaves, int region_d, int pitch_d, double v) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    double* q = p + (size_t)wave * pitch_d;
    for (int i = lane; i < region_d; i += 64) __builtin_nontemporal_store(v, q + i);
}
// same, with s_memrealtime stamps: [wave][0] start, [1] all stores issued, [2] all stores acknowledged
__global__ void fill_regions_stamped(double* __restrict__ p, int nwaves, int region_d, int pitch_d, double v, unsigned long long* st) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double* q = p + (size_t)wave * pitch_d;
    for (int i = lane; i < region_d; i += 64) __builtin_nontemporal_store(v, q + i);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { st[wave * 3] = t0; st[wave * 3 + 1] = t1; st[wave * 3 + 2] = t2; }
}

int main() {
    const int nint = 999, pitch_d = 5040, nbuf = 18;
    const size_t n = (size_t)nint * pitch_d;
    std::vector<double*> bufs(nbuf);
    for (auto& b : bufs) CK(hipMalloc(&b, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 400;
    float ms;
    for (int block : {64, 256}) {
        // pattern 0 on the same bytes
        for (int i = 0; i < 20; ++i) fill_strided<<<1024, 256>>>(bufs[i % nbuf], n, 1.0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) fill_strided<<<1024, 256>>>(bufs[i % nbuf], n, (double)i);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("strided fill        %6.1f MB: %6.2f us/launch\n", n * 8 / 1e6, ms * 1e3 / reps);
        // pattern 1: one region of 5040 doubles (whole interval block) per wave
        for (int region_d : {5040, 4096}) {
            const int grid = (nint * 64 + block - 1) / block;
            for (int i = 0; i < 20; ++i) fill_regions<<<grid, block>>>(bufs[i % nbuf], nint, region_d, pitch_d, 1.0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) fill_regions<<<grid, block>>>(bufs[i % nbuf], nint, region_d, pitch_d, (double)i);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("wave-private regions block %3d, %5d doubles of each %d: %6.1f MB: %6.2f us/launch (%5.2f TB/s)\n", block, region_d, pitch_d,
                   (double)nint * region_d * 8 / 1e6, ms * 1e3 / reps, (double)nint * region_d * 8 / (ms / reps * 1e-3) / 1e12);
        }
        // two waves per interval, half a block each
        {
            const int grid = (2 * nint * 64 + block - 1) / block;
            for (int i = 0; i < 20; ++i) fill_regions<<<grid, block>>>(bufs[i % nbuf], 2 * nint, 2520, 2520, 1.0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) fill_regions<<<grid, block>>>(bufs[i % nbuf], 2 * nint, 2520, 2520, (double)i);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("two waves per interval block %3d: %6.2f us/launch\n", block, ms * 1e3 / reps);
        }
        {
            const int grid = (4 * nint * 64 + block - 1) / block;
            for (int i = 0; i < 20; ++i) fill_regions<<<grid, block>>>(bufs[i % nbuf], 4 * nint, 1260, 1260, 1.0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) fill_regions<<<grid, block>>>(bufs[i % nbuf], 4 * nint, 1260, 1260, (double)i);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("four waves per interval block %3d: %6.2f us/launch\n", block, ms * 1e3 / reps);
        }
    }
    {   // in-kernel phases of the pure store pattern (one wave per interval, 40 KB each)
        unsigned long long* dst; CK(hipMalloc(&dst, nint * 3 * 8));
        std::vector<unsigned long long> hst(nint * 3);
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 30; ++i) fill_regions_stamped<<<nint, 64>>>(bufs[i % nbuf], nint, 5040, pitch_d, (double)i, dst);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hst.data(), dst, nint * 3 * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0, t2 = 0, s1 = 0;
            for (int w = 0; w < nint; ++w) { if (hst[w*3] < t0) t0 = hst[w*3]; }
            for (int w = 0; w < nint; ++w) { if (hst[w*3+1] > t1) t1 = hst[w*3+1]; if (hst[w*3+2] > t2) t2 = hst[w*3+2]; if (hst[w*3] > s1) s1 = hst[w*3]; }
            printf("stamped pure-store kernel: last wave start +%.2f us, last store issued +%.2f us, last store acknowledged +%.2f us\n",
                   (s1 - t0) * 0.01, (t1 - t0) * 0.01, (t2 - t0) * 0.01);
        }
    }
    return 0;
}
