// Microbenchmark (GPU box): do vector instructions of the SAME wave issue in the shadow of f64 MFMAs?
// One wave; per iteration 8 independent v_mfma_f64_16x16x4_f64 (64 cycles of the matrix pipe each) and, between them,
// K independent vector instructions of one kind.  Prints shader cycles per iteration (s_memtime).
//   hipcc -O3 --offload-arch=gfx950 tests/hip/mfma_valu_overlap.hip -o tests/hip/mfma_valu_overlap && tests/hip/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

// KIND 0: nothing, 1: v_fma_f64, 2: v_add_u32, 3: v_mov_b32 dpp (row_ror), 4: ds_write_b64 (LDS)
template <int NMFMA, int KIND, int K>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, int iters) {
    __shared__ double lds[64 * 8];
    v4d acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    double f[8];
    int n[8];
    for (int q = 0; q < 8; ++q) { f[q] = threadIdx.x * 1e-3 + q; n[q] = threadIdx.x + q; }
    const double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // (inline asm: the builtin makes the compiler shuttle the accumulators between VGPRs and AGPRs every iteration)
            if (q < NMFMA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < K; ++v) {
                if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[v & 7]) : "v"(a));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[v & 7]) : "v"(it));
                if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(n[v & 7]));
                if (KIND == 4) asm volatile("ds_write_b64 %0, %1" ::"v"((int)threadIdx.x * 8), "v"(f[v & 7]) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3] + f[q] + n[q];
    out[threadIdx.x] = s + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// Two waves of ONE SIMD (waves 0 and 4 of a 320-thread workgroup): wave 0 issues MFMAs, wave 4 vector instructions.
// mode bit 0: wave 0 active, bit 1: wave 4 active.  cyc[0], cyc[1] = cycles per iteration of each.
__global__ __launch_bounds__(320) void k2(double* out, unsigned long long* cyc, int iters, int mode) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double a = 1.0 + lane * 1e-9, b = 0.5;
    if (w == 0 && (mode & 1)) {
        v4d acc[8];
        for (int q = 0; q < 8; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0.0;
        for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        out[lane] = s;
        if (lane == 0) { cyc[0] = t1 - t0; cyc[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
    }
    if (w == 4 && (mode & 2)) {
        double f[8];
        for (int q = 0; q < 8; ++q) f[q] = lane * 1e-3 + q;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 96; ++v) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[v & 7]) : "v"(a));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0.0;
        for (int q = 0; q < 8; ++q) s += f[q];
        out[64 + lane] = s;
        if (lane == 0) { cyc[1] = t1 - t0; cyc[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
    }
}

static int run2(int mode, double* out, unsigned long long* cyc) {
    const int iters = 2000;
    unsigned long long c[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(cyc, 0, 32));
        hipLaunchKernelGGL(k2, dim3(1), dim3(320), 0, 0, out, cyc, iters, mode);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(c, cyc, 32, hipMemcpyDeviceToHost));
    printf("two waves of one SIMD, mode %d: MFMA wave %8.1f cycles / 8 MFMAs, vector wave %8.1f cycles / 96 v_fma_f64   (HW_ID SIMD %d / %d, CU %d / %d)\n",
           mode, (double)c[0] / iters, (double)c[1] / iters, (int)((c[2] >> 4) & 3), (int)((c[3] >> 4) & 3), (int)((c[2] >> 8) & 15), (int)((c[3] >> 8) & 15));
    return 0;
}

template <int NMFMA, int KIND, int K>
static int run(const char* name, double* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NMFMA, KIND, K>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<NMFMA, KIND, K>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %8.1f cycles per iteration (%d MFMAs + %d x %d vector instructions)\n", name, (double)c / iters, NMFMA, 8, K);
    return 0;
}

int main() {
    double* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 128 * 8));
    CK(hipMalloc(&cyc, 32));
    run<8, 0, 0>("8 MFMA", out, cyc);
    run<4, 0, 0>("4 MFMA", out, cyc);
    run<8, 1, 2>("8 MFMA + 16 v_fma_f64 interleaved", out, cyc);
    run<8, 1, 8>("8 MFMA + 64 v_fma_f64 interleaved", out, cyc);
    run<0, 1, 12>("96 v_fma_f64", out, cyc);
    run<8, 1, 12>("8 MFMA + 96 v_fma_f64 interleaved", out, cyc);
    run<8, 1, 4>("8 MFMA + 32 v_fma_f64 interleaved", out, cyc);
    run<0, 2, 12>("96 v_add_u32", out, cyc);
    run<8, 2, 12>("8 MFMA + 96 v_add_u32 interleaved", out, cyc);
    run<0, 3, 12>("96 v_mov_b32_dpp", out, cyc);
    run<8, 3, 12>("8 MFMA + 96 v_mov_b32_dpp interleaved", out, cyc);
    run<0, 4, 4>("32 ds_write_b64", out, cyc);
    run<8, 4, 4>("8 MFMA + 32 ds_write_b64 interleaved", out, cyc);
    run2(1, out, cyc);
    run2(2, out, cyc);
    run2(3, out, cyc);
    return 0;
}
