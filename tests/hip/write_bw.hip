// Microbenchmark (GPU box): what a write-only stream of the bench's size can reach on this device.
// Fills a ring of buffers (total > 2 x Infinity Cache) with 8-byte or 16-byte per-lane stores,
// plain / nt / sc1, and reports per-launch time and TB/s for 42.6 MB (config 3) and 341 MB (config 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int W, int MODE>
__global__ void fill(double* __restrict__ p, size_t n, double v) {   // n doubles
    const size_t stride = (size_t)gridDim.x * blockDim.x * W;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W; i < n; i += stride) {
        if (W == 2) {
            v2d x = {v, v};
            if (MODE == 2) __builtin_nontemporal_store(x, reinterpret_cast<v2d*>(p + i));
            else *reinterpret_cast<v2d*>(p + i) = x;
        } else {
            if (MODE == 1) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + i), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 2) __builtin_nontemporal_store(v, p + i);
            else p[i] = v;
        }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W, int MODE>
int run(const char* name, size_t bytes, int grid, int block) {
    const size_t n = bytes / 8;
    const int nbuf = (int)((size_t)(700u << 20) / bytes) + 2;
    std::vector<double*> bufs(nbuf);
    for (auto& b : bufs) CK(hipMalloc(&b, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) fill<W, MODE><<<grid, block>>>(bufs[i % nbuf], n, 1.0);
    CK(hipDeviceSynchronize());
    const int reps = 200;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fill<W, MODE><<<grid, block>>>(bufs[i % nbuf], n, (double)i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %7.1f MB grid %5d x %4d : %7.2f us/launch  %5.2f TB/s\n", name, bytes / 1e6, grid, block, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e12);
    for (auto& b : bufs) CK(hipFree(b));
    return 0;
}

int main() {
    const size_t sizes[2] = {42574560 / 16 * 16, 340886560 / 16 * 16};
    for (size_t bytes : sizes) {
        for (int grid : {256, 1024, 4096}) {
            run<2, 0>("16B/lane plain", bytes, grid, 256);
            run<2, 2>("16B/lane nt", bytes, grid, 256);
            run<1, 0>("8B/lane plain", bytes, grid, 256);
            run<1, 2>("8B/lane nt", bytes, grid, 256);
            run<1, 1>("8B/lane sc1", bytes, grid, 256);
        }
    }
    return 0;
}
