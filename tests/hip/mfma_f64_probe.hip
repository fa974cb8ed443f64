// Hardware probe (run on the GPU box): checks the lane maps of v_mfma_f64_16x16x4_f64 that
// qc_mfma_kernels.hip relies on, with exact integer data and an asymmetric B
// (cdna_hip_programming.md section 3: "Always A=I-check with ASYMMETRIC B").
//   A (16x4):  lane l holds A[i = l & 15][k = l >> 4]
//   B (4x16):  lane l holds B[k = l >> 4][j = l & 15]
//   C/D:       lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
// and the DPP row_ror:8 half-row swap.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, double* D, double* S) {
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    v4d acc = {0, 0, 0, 0};
    for (int kk = 0; kk < 4; ++kk) {
        const double a = A[i * 16 + 4 * kk + g];        // A row-major [16][16]
        const double b = B[(4 * kk + g) * 16 + i];      // B row-major [16][16]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + i] = acc[r];
    int lo = __double2loint((double)l), hi = __double2hiint((double)l);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    S[l] = __hiloint2double(hi, lo);
}

int main() {
    double hA[256], hB[256], hD[256], ref[256], hS[64];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { hA[i * 16 + j] = (double)((i * 7 + j * 3) % 11 - 5); hB[i * 16 + j] = (double)((i * 5 + j * 13) % 17 - 8); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD, *dS;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 2048); hipMalloc(&dS, 512);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD, dS);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(hS, dS, 512, hipMemcpyDeviceToHost);
    int bad = 0, badS = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    for (int l = 0; l < 64; ++l) badS += hS[l] != (double)((l & ~15) | ((l + 8) & 15));
    printf("mfma_f64_16x16x4 layout mismatches: %d ; row_ror:8 swap mismatches: %d\n", bad, badS);
    if (badS) { for (int l = 0; l < 32; ++l) printf("%g ", hS[l]); printf("\n"); }
    return (bad || badS) ? 1 : 0;
}
