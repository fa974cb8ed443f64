// Device -> pinned host copies of one 12.8 / 14.7 MB block: one copy-engine transfer against 2 / 4 concurrent transfers of equal
// parts on separate streams, and against a transfer with a 4 MB upload running beside it (the Hessian call's multipliers).
//   hipcc -O3 --offload-arch=gfx950 tests/hip/d2h_split.hip -o tests/hip/d2h_split && tests/hip/d2h_split
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t sizes[2] = {(size_t)12800000, (size_t)14700000};
    hipStream_t st[8];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char *d, *h, *d2, *h2;
    CK(hipMalloc((void**)&d, 16 << 20)); CK(hipHostMalloc((void**)&h, 16 << 20, hipHostMallocDefault));
    CK(hipMalloc((void**)&d2, 4 << 20)); CK(hipHostMalloc((void**)&h2, 4 << 20, hipHostMallocDefault));
    for (size_t bytes : sizes) {
        for (int parts : {1, 2, 3, 4, 8}) {
            for (int up = 0; up < 2; ++up) {
                double best = 1e30;
                for (int r = 0; r < 12; ++r) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now_us();
                    const size_t per = (bytes / parts + 4095) & ~(size_t)4095;
                    if (up) CK(hipMemcpyAsync(d2, h2, 4 << 20, hipMemcpyHostToDevice, st[7]));
                    for (int p = 0; p < parts; ++p) {
                        const size_t o = p * per, len = std::min(per, bytes - o);
                        CK(hipMemcpyAsync(h + o, d + o, len, hipMemcpyDeviceToHost, st[p]));
                    }
                    for (int p = 0; p < parts; ++p) CK(hipStreamSynchronize(st[p]));
                    const double t1 = now_us();
                    if (up) CK(hipStreamSynchronize(st[7]));
                    best = std::min(best, t1 - t0);
                }
                printf("%.1f MB device -> pinned host in %d concurrent part(s)%s: best %.0f us = %.1f GB/s\n", bytes / 1e6, parts,
                       up ? " + a 4 MB upload beside it" : "", best, bytes / best / 1e3);
            }
        }
    }
    return 0;
}
