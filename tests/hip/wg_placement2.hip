// Where do the two waves of a 128-thread workgroup go, four workgroups resident per compute unit (the launch shape of
// qc_mfma16_pade4_fused_kernel / qc_mfma16_pade4_kernel at BASELINE config 3: 999 workgroups, ~36 KB of LDS, <= 256 VGPRs)?  Records
// HW_REG_HW_ID of both waves of every workgroup; prints, per compute unit, (simd of wave 0 / simd of wave 1) in start order, and how
// many compute units have TWO first waves (the MFMA-heavy role of those kernels) on one SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement2 wg_placement2.hip && ./wg_placement2 [grid = 999]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec { unsigned xcc, hw[2]; unsigned long long t0; };

__global__ __launch_bounds__(128, 2) void probe(Rec* out, double* sink, int spin) {
    __shared__ double lds[36 * 128];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
    lds[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    double acc = lds[(threadIdx.x * 7) & 127];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 1e-9;
    if (acc == 12345.678) sink[0] = acc;
    if (lane == 0) { out[blockIdx.x].hw[w] = hw; if (w == 0) { out[blockIdx.x].xcc = xcc; out[blockIdx.x].t0 = t0; } }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 999;
    Rec* d; double* sink;
    hipMalloc(&d, sizeof(Rec) * grid); hipMalloc(&sink, 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(128), 0, 0, d, sink, 3000);
    hipDeviceSynchronize();
    std::vector<Rec> r(grid);
    hipMemcpy(r.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < grid; ++b) tmin = std::min(tmin, r[b].t0);
    for (int b = 0; b < grid; ++b) per_cu[(r[b].xcc << 16) | (r[b].hw[0] & 0xff00)].push_back(b);
    int shown = 0, hist[8] = {0}, clash = 0, split = 0, same_slot = 0;
    for (auto& kv : per_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return r[a].t0 < r[b].t0; });
        hist[std::min<size_t>(v.size(), 7)]++;
        int first_on[4] = {0, 0, 0, 0};
        for (int b : v) { first_on[(r[b].hw[0] >> 4) & 3]++; split += ((r[b].hw[0] >> 4) & 3) != ((r[b].hw[1] >> 4) & 3); same_slot += (r[b].hw[0] & 15) == (r[b].hw[1] & 15); }
        clash += *std::max_element(first_on, first_on + 4) > 1;
        if (shown++ < 16) {
            printf("xcc %u se %u cu %2u:", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 8) & 15);
            for (int b : v) printf("  wg %3d (+%4.2f us; simd %u/%u slot %u/%u)", b, (r[b].t0 - tmin) / 100.0, (r[b].hw[0] >> 4) & 3, (r[b].hw[1] >> 4) & 3, r[b].hw[0] & 15, r[b].hw[1] & 15);
            printf("\n");
        }
    }
    printf("grid %d: %zu compute units;", grid, per_cu.size());
    for (int k = 1; k < 8; ++k) if (hist[k]) printf(" %d host %d workgroups,", hist[k], k);
    printf("\nworkgroups whose two waves sit on different SIMDs: %d of %d; with the same wave slot number: %d\n", split, grid, same_slot);
    printf("compute units with two or more FIRST waves on one SIMD: %d\n", clash);
    return 0;
}
