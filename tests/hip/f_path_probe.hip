// Probe for the residual-only host call (config 3: 1.18 MB of knots up, 1.12 MB of residuals down, a ~5 us kernel between):
// which way of moving the two vectors costs what, wall clock per call, caller-side.  Build: hipcc -O2 --offload-arch=gfx950 f_path_probe.hip -o f_path_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one workgroup per interval: reads 2 x 147 doubles of Z, writes 140 doubles of F (a stand-in with the real kernel's traffic)
__global__ void stand_in(const double* __restrict__ Z, double* __restrict__ F, int zdim, int ddim) {
    const int b = blockIdx.x, t = threadIdx.x;
    const double* z0 = Z + (size_t)b * zdim;
    double acc = 0.0;
    for (int i = t; i < 2 * zdim; i += blockDim.x) acc += z0[i];
    __shared__ double s[256];
    s[t] = acc;
    __syncthreads();
    if (t < ddim) __builtin_nontemporal_store(s[t % 64] + z0[t], F + (size_t)b * ddim + t);
}

int main() {
    const int T = 1000, zdim = 147, ddim = 140, n_int = T - 1;
    const size_t zb = (size_t)T * zdim * 8, fb = (size_t)n_int * ddim * 8;
    double *dZ, *dF;
    CK(hipMalloc(&dZ, zb)); CK(hipMalloc(&dF, fb));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    double* Zp = (double*)aligned_alloc(4096, (zb + 4095) / 4096 * 4096);
    double* Fp = (double*)aligned_alloc(4096, (fb + 4095) / 4096 * 4096);
    for (size_t i = 0; i < zb / 8; ++i) Zp[i] = 1e-3 * (double)(i % 977);
    memset(Fp, 0, fb);
    double *Zh, *Fh;   // library-owned pinned memory
    CK(hipHostMalloc(&Zh, zb, hipHostMallocDefault)); CK(hipHostMalloc(&Fh, fb, hipHostMallocDefault));
    memcpy(Zh, Zp, zb);
    auto wait = [&] { hipError_t e; while ((e = hipEventQuery(ev)) == hipErrorNotReady) __builtin_ia32_pause(); CK(e); };
    auto run = [&](const char* name, auto&& fn) {
        std::vector<double> ts;
        for (int i = 0; i < 60; ++i) { const double t0 = now_us(); fn(); ts.push_back(now_us() - t0); }
        std::sort(ts.begin() + 10, ts.end());
        printf("%-78s median %6.1f us  min %6.1f\n", name, ts[10 + 25], ts[10]);
    };
    run("pageable Z -> H2D copy -> kernel -> D2H copy -> pageable F (the current path)", [&] {
        CK(hipMemcpyAsync(dZ, Zp, zb, hipMemcpyHostToDevice, st));
        stand_in<<<n_int, 256, 0, st>>>(dZ, dF, zdim, ddim);
        CK(hipMemcpyAsync(Fp, dF, fb, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st)); wait(); });
    run("pinned Z (library memory) -> H2D copy -> kernel -> D2H copy -> pinned F", [&] {
        CK(hipMemcpyAsync(dZ, Zh, zb, hipMemcpyHostToDevice, st));
        stand_in<<<n_int, 256, 0, st>>>(dZ, dF, zdim, ddim);
        CK(hipMemcpyAsync(Fh, dF, fb, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st)); wait(); });
    run("pageable Z -> H2D copy -> kernel writes pinned F directly", [&] {
        CK(hipMemcpyAsync(dZ, Zp, zb, hipMemcpyHostToDevice, st));
        stand_in<<<n_int, 256, 0, st>>>(dZ, Fh, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); });
    run("pageable Z -> H2D copy -> kernel writes pinned F directly -> memcpy to pageable F (1 thread)", [&] {
        CK(hipMemcpyAsync(dZ, Zp, zb, hipMemcpyHostToDevice, st));
        stand_in<<<n_int, 256, 0, st>>>(dZ, Fh, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); memcpy(Fp, Fh, fb); });
    run("pinned Z read by the kernel directly, pinned F written directly (no copy at all)", [&] {
        stand_in<<<n_int, 256, 0, st>>>(Zh, Fh, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); });
    run("memcpy pageable Z -> pinned (1 thread), kernel reads / writes pinned directly", [&] {
        memcpy(Zh, Zp, zb);
        stand_in<<<n_int, 256, 0, st>>>(Zh, Fh, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); });
    // registering the caller's arrays once (what an explicit registration call of the boundary would do)
    CK(hipHostRegister(Zp, zb, hipHostRegisterDefault)); CK(hipHostRegister(Fp, fb, hipHostRegisterDefault));
    double *Zd = nullptr, *Fd = nullptr;
    CK(hipHostGetDevicePointer((void**)&Zd, Zp, 0)); CK(hipHostGetDevicePointer((void**)&Fd, Fp, 0));
    run("registered caller arrays: H2D copy -> kernel -> D2H copy", [&] {
        CK(hipMemcpyAsync(dZ, Zp, zb, hipMemcpyHostToDevice, st));
        stand_in<<<n_int, 256, 0, st>>>(dZ, dF, zdim, ddim);
        CK(hipMemcpyAsync(Fp, dF, fb, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st)); wait(); });
    run("registered caller arrays: H2D copy -> kernel writes the caller's F directly", [&] {
        CK(hipMemcpyAsync(dZ, Zp, zb, hipMemcpyHostToDevice, st));
        stand_in<<<n_int, 256, 0, st>>>(dZ, Fd, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); });
    run("registered caller arrays: the kernel reads Z and writes F in place (no copy at all)", [&] {
        stand_in<<<n_int, 256, 0, st>>>(Zd, Fd, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); });
    {
        const double t0 = now_us();
        CK(hipHostUnregister(Fp));
        const double t1 = now_us();
        CK(hipHostRegister(Fp, fb, hipHostRegisterDefault));
        printf("hipHostUnregister %.0f us, hipHostRegister %.0f us (1.12 MB)\n", t1 - t0, now_us() - t1);
    }
    run("kernel alone (device to device) + event", [&] {
        stand_in<<<n_int, 256, 0, st>>>(dZ, dF, zdim, ddim);
        CK(hipEventRecord(ev, st)); wait(); });
    run("H2D copy of pageable Z alone + event", [&] { CK(hipMemcpyAsync(dZ, Zp, zb, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev, st)); wait(); });
    return 0;
}
