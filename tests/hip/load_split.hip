// The load phase of the 2N = 16 Hessian kernels: 999 workgroups, each fetching 7 generator-image tiles (2 x 16 bytes per lane each, the
// same 14 KB for every workgroup: L2 hits) and 3 knot tiles (distinct per workgroup: HBM) -- (a) all by one wave of a one-wave
// workgroup, (b) all by wave 0 of a two-wave workgroup, (c) split between the two waves.  Reported: time from the wave's first
// instruction to "all its loads back" (s_memrealtime, 100 MHz), median and maximum over the workgroups.
//   hipcc -O3 --offload-arch=gfx950 tests/hip/load_split.hip -o tests/hip/load_split && tests/hip/load_split
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>   // 0: one-wave workgroups; 1: two waves, wave 0 loads everything; 2: two waves, split 4 + 1 / 3 + 2 tiles
__global__ void k(const double* img, const double* knots, double* out, unsigned long long* ts) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v2d* ip = reinterpret_cast<const v2d*>(img) + lane;
    const v2d* kp = reinterpret_cast<const v2d*>(knots + (size_t)blockIdx.x * 160) + (lane & 31);
    v2d acc = {0.0, 0.0};
    int i0 = 0, i1 = 7, k0 = 0, k1 = 3;
    if (MODE == 1 && wave == 1) { i1 = 0; k1 = 0; }
    if (MODE == 2) { if (wave == 0) { i1 = 4; k1 = 1; } else { i0 = 4; k0 = 1; } }
    v2d x[14], y[6];
#pragma unroll
    for (int i = 0; i < 7; ++i) if (i >= i0 && i < i1) { x[2 * i] = ip[i * 128]; x[2 * i + 1] = ip[i * 128 + 64]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i >= k0 && i < k1) { y[2 * i] = kp[i * 16]; y[2 * i + 1] = kp[i * 16 + 8]; }
#pragma unroll
    for (int i = 0; i < 7; ++i) if (i >= i0 && i < i1) acc += x[2 * i] + x[2 * i + 1];
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i >= k0 && i < k1) acc += y[2 * i] + y[2 * i + 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1];
    if (lane == 0) { ts[(blockIdx.x * 2 + wave) * 2] = t0; ts[(blockIdx.x * 2 + wave) * 2 + 1] = t1; }
}
template <int MODE>
static void run(const char* what, int block, const double* img, const double* knots, double* out, unsigned long long* ts) {
    const int grid = 999;
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(block), 0, 0, img, knots, out, ts);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (int b = 0; b < grid; ++b) first = std::min(first, h[b * 4]);
    for (int w = 0; w < block / 64; ++w) {
        std::vector<double> back;
        for (int b = 0; b < grid; ++b) back.push_back((h[(b * 2 + w) * 2 + 1] - first) * 0.01);
        std::sort(back.begin(), back.end());
        printf("%-58s wave %d: loads back at median %.2f us, max %.2f us after the first wave's entry\n", what, w, back[grid / 2], back[grid - 1]);
    }
}
int main() {
    double *img, *knots, *out; unsigned long long* ts;
    hipMalloc(&img, 7 * 2048); hipMalloc(&knots, 1000 * 160 * 8); hipMalloc(&out, 999 * 128 * 8); hipMalloc(&ts, 999 * 4 * 8);
    hipMemset(img, 0, 7 * 2048); hipMemset(knots, 0, 1000 * 160 * 8);
    run<0>("one-wave workgroups, 10 tiles per wave", 64, img, knots, out, ts);
    run<1>("two-wave workgroups, wave 0 loads the 10 tiles", 128, img, knots, out, ts);
    run<2>("two-wave workgroups, 5 tiles each", 128, img, knots, out, ts);
    return 0;
}
