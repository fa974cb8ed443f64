// Hardware probe: is v_mfma_f64_16x16x4_f64 a chain of FMAs in k order, D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C))))?
// What the sparse-drive (row-gather) forms of the 2N = 16 kernels rely on for bit-identity with the dense-image kernels: with ONE non-zero
// A[i][k] = w per row, a product accumulated into C must equal fma(w, B[k][j], C) exactly (the zero terms add nothing), and with two
// non-zeros per row fma(w2, y2, fma(w1, y1, C)) in ascending k.  Random operands with full mantissas.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_fma_probe mfma_f64_fma_probe.hip && ./mfma_f64_fma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4d __attribute__((ext_vector_type(4)));

// D = C + A B over 16 x 16 x 16 as four chained MFMAs (the kernels' mm16 with an accumulator input)
__global__ void probe(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    v4d acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(g + 4 * r) * 16 + i];
    for (int kk = 0; kk < 4; ++kk) {
        const double a = A[i * 16 + 4 * kk + g];        // A row-major [16][16]
        const double b = B[(4 * kk + g) * 16 + i];      // B row-major [16][16]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + i] = acc[r];
}

static double rnd() { return ldexp((double)rand() / RAND_MAX - 0.5, rand() % 7 - 3) * (1.0 + 1e-9 * rand()); }

int main() {
    double hA[256], hB[256], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    srand(7);
    int bad1 = 0, bad1z = 0, bad2 = 0, bad2alt = 0, badfull = 0, n1 = 0, n2 = 0, nf = 0;
    for (int trial = 0; trial < 200; ++trial) {
        const int mode = trial % 3;      // 0: one non-zero per row; 1: two per row; 2: dense (full chain of 16 in k order)
        for (int i = 0; i < 256; ++i) { hA[i] = 0.0; hB[i] = rnd(); hC[i] = (trial & 8) ? 0.0 : rnd(); }
        int c1[16], c2[16];
        for (int i = 0; i < 16; ++i) {
            c1[i] = rand() % 16; c2[i] = (c1[i] + 1 + rand() % 15) % 16;
            if (mode == 2) { for (int k = 0; k < 16; ++k) hA[i * 16 + k] = rnd(); continue; }
            hA[i * 16 + c1[i]] = rnd();
            if (mode == 1) hA[i * 16 + c2[i]] = rnd();
        }
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const double d = hD[i * 16 + j], c = hC[i * 16 + j];
            if (mode == 0) {
                ++n1;
                const double want = fma(hA[i * 16 + c1[i]], hB[c1[i] * 16 + j], c);
                bad1 += d != want;
                if (c == 0.0) bad1z += d != hA[i * 16 + c1[i]] * hB[c1[i] * 16 + j];
            } else if (mode == 1) {
                ++n2;
                const int lo = c1[i] < c2[i] ? c1[i] : c2[i], hi = c1[i] < c2[i] ? c2[i] : c1[i];
                const double asc = fma(hA[i * 16 + hi], hB[hi * 16 + j], fma(hA[i * 16 + lo], hB[lo * 16 + j], c));
                const double desc = fma(hA[i * 16 + lo], hB[lo * 16 + j], fma(hA[i * 16 + hi], hB[hi * 16 + j], c));
                bad2 += d != asc;
                bad2alt += d != desc;
            } else {
                ++nf;
                double t = c;
                for (int k = 0; k < 16; ++k) t = fma(hA[i * 16 + k], hB[k * 16 + j], t);
                badfull += d != t;
            }
        }
    }
    printf("one non-zero per row:  %d of %d entries differ from fma(w, y, C)   (C = 0 cases differing from w * y: %d)\n", bad1, n1, bad1z);
    printf("two non-zeros per row: %d of %d differ from the ascending-k fma chain (%d from the descending one)\n", bad2, n2, bad2alt);
    printf("dense rows:            %d of %d differ from the 16-term fma chain in k order\n", badfull, nf);
    return 0;
}
