// Where does the dispatcher put the workgroups of a one-wave grid of 512-thread workgroups, two resident per compute unit (the launch
// shape of qc_mfma32_ell_kernel at BASELINE config 5: 499 workgroups, <= 128 VGPRs, ~46 KB of LDS)?  Records per workgroup the XCC id,
// HW_REG_HW_ID of its waves 0 and 7 and the time its first wave started; prints, per compute unit, the workgroups it hosted in start order.
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip && ./wg_placement [grid = 499]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec { unsigned xcc, hw0, hw7; unsigned long long t0; };

__global__ __launch_bounds__(512, 4) void probe(Rec* out, double* sink, int spin) {
    __shared__ double lds[46 * 128];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
    lds[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    double acc = lds[(threadIdx.x * 7) & 511];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 1e-9;      // ~ a few microseconds of residency, so that every workgroup is resident at once
    if (acc == 12345.678) sink[0] = acc;
    if (lane == 0 && w == 0) { out[blockIdx.x].xcc = xcc; out[blockIdx.x].hw0 = hw; out[blockIdx.x].t0 = t0; }
    if (lane == 0 && w == 7) out[blockIdx.x].hw7 = hw;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 499;
    Rec* d; double* sink;
    hipMalloc(&d, sizeof(Rec) * grid); hipMalloc(&sink, 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, d, sink, 3000);
    hipDeviceSynchronize();
    std::vector<Rec> r(grid);
    hipMemcpy(r.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    std::map<unsigned, std::vector<int>> per_cu;
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < grid; ++b) tmin = std::min(tmin, r[b].t0);
    for (int b = 0; b < grid; ++b) per_cu[(r[b].xcc << 16) | (r[b].hw0 & 0xff00)].push_back(b);
    int shown = 0, one = 0, two = 0, more = 0, first_lower = 0, both = 0, slot_rule = 0;
    for (auto& kv : per_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return r[a].t0 < r[b].t0; });
        one += v.size() == 1; two += v.size() == 2; more += v.size() > 2;
        if (v.size() == 2) {
            ++both;
            first_lower += v[0] < v[1];
            slot_rule += (r[v[0]].hw0 & 15) < (r[v[1]].hw0 & 15);
        }
        if (shown++ < 24) {
            printf("xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 12) & 1, (kv.first >> 8) & 15);
            for (int b : v) printf("  wg %3d (t0 +%5.2f us, wave slots %u/%u simd %u/%u)", b, (r[b].t0 - tmin) / 100.0, r[b].hw0 & 15, r[b].hw7 & 15, (r[b].hw0 >> 4) & 3, (r[b].hw7 >> 4) & 3);
            printf("\n");
        }
    }
    printf("grid %d: %zu compute units used; %d host one workgroup, %d two, %d more\n", grid, per_cu.size(), one, two, more);
    printf("of the %d pairs: the earlier workgroup has the LOWER blockIdx in %d, the lower wave slot (wave 0) in %d\n", both, first_lower, slot_rule);
    int lowhalf = 0;
    for (auto& kv : per_cu) if (kv.second.size() == 2) lowhalf += (kv.second[0] < grid / 2) != (kv.second[1] < grid / 2);
    printf("pairs with exactly one member in the lower half of the grid (blockIdx < %d): %d\n", grid / 2, lowhalf);
    return 0;
}
