// f64-MFMA kernel for the order-4 Pade integrator at 2N = 32 (4 qubits, BASELINE config 5): every
// matrix is 2 x 2 tiles of 16 x 16, every n x N (32 x 16) matrix two full-width tiles, all products
// v_mfma_f64_16x16x4_f64 with register operands (lane maps: qc_mfma_kernels.hip header).
//
// One 512-thread workgroup (8 wavefronts) per interval:
//   waves 0-3  assemble one tile each of G = G_0 + sum_k a_k G_k from the lane-ordered generator image
//              (72 KB for m = 8, read ONCE per interval) and publish it in LDS (8 KB); one barrier.
//   waves 4-7  "copy waves": wave 4+c owns tile (I, J) = (c>>1, c&1) of B^T and F^T:
//              G_B tiles by identity products, (G^2)^T[I][J] = sum_K G_B[K][I] * G_A[J][K]  (16 MFMAs),
//              then the 2N = 32 tile stores of the I_N (x) B / -I_N (x) F copies (84 % of the bytes).
//   waves 0-3  "compute waves": G D, Q_0 = -h/2 S + h^2/12 G D, Q_1 = h^2/12 D (16 MFMAs, each wave),
//              then drives k = w, w+4, ...:  d/da_k = G_k Q_0 + G (G_k Q_1), transposed for the store
//              (56 MFMAs per drive).  Wave 0 also produces the residual and d/dh (48 MFMAs).
// Outputs are stored transposed (lane <-> row): each store instruction writes four whole 128-byte lines.
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kThreads32 = 512;
constexpr int kMaxGrid32 = 1024;
constexpr int kMU32 = 8;

// A-layout image of tile `tile` (= 2 I + K) of generator `mat`: [mat][tile][pair][lane][2]
__device__ inline v4d load_GA32(const double* __restrict__ Gx, int mat, int tile, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(Gx) + (mat * 4 + tile) * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline v4d lds_tile(const double* __restrict__ base, int tile, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(base) + tile * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline void lds_put_tile(double* __restrict__ base, int tile, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + tile * 128 + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}

// A0*B0 + A1*B1 (two 16x16x16 products summed): two accumulator chains of four MFMAs
__device__ inline v4d mm16x2(const v4d& a0, const v4d& b0, const v4d& a1, const v4d& b1) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], z, 0, 0, 0);
    v4d acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], z, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[2], b0[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[3], b0[3], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[2], b1[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[3], b1[3], acc1, 0, 0, 0);
    return acc0 + acc1;
}

// Store transposed tile I of a 32 x 16 matrix (or tile (I, J) of a 32 x 32 one with colbase = 16 J... see callers):
// lane (g, j) reg r holds X[rowbase + j][colbase + 4r + g]; column-major with 32 rows per column.
__device__ inline void store_T32(double* __restrict__ p, const v4d& x, int rowbase, int colbase, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8m<2>(p + (colbase + 4 * r + g) * 32 + rowbase + j, x[r]);   // non-temporal, compile-time
}

template <bool JAC>
__global__ __launch_bounds__(kThreads32, 2) void qc_mfma32_pade4_kernel(const QcParams P, const double* __restrict__ Z,
                                                                       double* __restrict__ F, double* __restrict__ J) {
    __shared__ __attribute__((aligned(16))) double GaL[4 * 256];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gx = P.Gx;
    const v4d IdB = identity_B(g, j);

    for (int vb = blockIdx.x; vb < P.n_int; vb += gridDim.x) {
        const int b = qc_xcd_remap(vb, P.n_int);
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
        double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
        const double h = ft ? z0[P.off_dt] : P.dt_fixed;
        const double hc1 = h * c1, hc2 = h * h * c2;

        // knot data for the compute waves (requested before the assembly barrier)
        v4d S[2], D[2];
        if (w < 4) {
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const double* u0p = z0 + P.off_U + j * 32 + 16 * I + g;
                const double* u1p = z1 + P.off_U + j * 32 + 16 * I + g;
                const v4d u0 = {u0p[0], u0p[4], u0p[8], u0p[12]};
                const v4d u1 = {u1p[0], u1p[4], u1p[8], u1p[12]};
                S[I] = u1 + u0;
                D[I] = u1 - u0;
            }
            // ---- assemble tile w of G and publish it -------------------------------------------------
            v4d Gt = load_GA32(Gx, 0, w, lane);
            for (int k0 = 0; k0 < m; k0 += kMU32) {
                v4d gk[kMU32];
                double ak[kMU32];
#pragma unroll
                for (int u = 0; u < kMU32; ++u) {
                    const int k = (k0 + u < m) ? k0 + u : m - 1;
                    gk[u] = load_GA32(Gx, k + 1, w, lane);
                    ak[u] = (k0 + u < m) ? z0[P.off_a + k] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < kMU32; ++u) Gt += ak[u] * gk[u];
            }
            lds_put_tile(GaL, w, lane, Gt);
        }
        __syncthreads();

        if (w >= 4) {
            // ================= copy wave: tile (I, Jt) of B^T and F^T, N copies each =====================
            if (JAC) {
                const int c = w - 4, I = c >> 1, Jt = c & 1;
                // (G^T)[I][K] as A operand = B-layout of G[K][I] = G[K][I] * Id;  (G^T)[K][Jt] as B operand = A-layout of G[Jt][K]
                const v4d GbK0 = mm16(lds_tile(GaL, 0 * 2 + I, lane), IdB);
                const v4d GbK1 = mm16(lds_tile(GaL, 1 * 2 + I, lane), IdB);
                const v4d G2T = mm16x2(GbK0, lds_tile(GaL, Jt * 2 + 0, lane), GbK1, lds_tile(GaL, Jt * 2 + 1, lane));
                const v4d GT = lds_tile(GaL, Jt * 2 + I, lane);      // D-layout of (G^T)[I][Jt] = A-layout of G[Jt][I]
                v4d Fm, Bm;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ev = (I == Jt ? IdB[r] : 0.0) + hc2 * G2T[r];
                    Fm[r] = -(ev + hc1 * GT[r]);
                    Bm[r] = ev - hc1 * GT[r];
                }
                // lane (g, j) reg r = B^T[16I+4r+g][16Jt+j] = B[16Jt+j][16I+4r+g]
                double* pF = Jb + P.jo_F;
                double* pB = Jb + P.jo_B;
                for (int q = 0; q < 16; ++q) {
                    store_T32(pF + q * 1024, Fm, 16 * Jt, 16 * I, g, j);
                    store_T32(pB + q * 1024, Bm, 16 * Jt, 16 * I, g, j);
                }
                if (w == 4) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, false);
            }
        } else {
            // ===================== compute waves ===========================================================
            v4d Ga[4];
#pragma unroll
            for (int tI = 0; tI < 4; ++tI) Ga[tI] = lds_tile(GaL, tI, lane);
            v4d GD[2];
#pragma unroll
            for (int I = 0; I < 2; ++I) GD[I] = mm16x2(Ga[2 * I], D[0], Ga[2 * I + 1], D[1]);
            if (w == 0) {
                v4d GS[2], G2D[2];
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    GS[I] = mm16x2(Ga[2 * I], S[0], Ga[2 * I + 1], S[1]);
                    G2D[I] = mm16x2(Ga[2 * I], GD[0], Ga[2 * I + 1], GD[1]);
                }
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    const v4d dl = D[I] - hc1 * GS[I] + hc2 * G2D[I];
                    if (Fb) store_T32(Fb, mm16(dl, IdB), 16 * I, 0, g, j);
                    if (JAC && ft) {
                        const v4d dh = (-c1) * GS[I] + (2.0 * c2 * h) * G2D[I];
                        store_T32(Jb + P.jo_h, mm16(dh, IdB), 16 * I, 0, g, j);
                    }
                }
                if (!JAC) deriv_rows_generic(P, z0, z1, h, Fb, nullptr, lane, false);
            }
            if (JAC) {
                v4d Q0[2], Q1[2];
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    Q0[I] = (-hc1) * S[I] + hc2 * GD[I];
                    Q1[I] = hc2 * D[I];
                }
                for (int k = w; k < m; k += 4) {
                    v4d Gk[4];
#pragma unroll
                    for (int tI = 0; tI < 4; ++tI) Gk[tI] = load_GA32(Gx, k + 1, tI, lane);
                    v4d R0[2], R1[2];
#pragma unroll
                    for (int I = 0; I < 2; ++I) {
                        R0[I] = mm16x2(Gk[2 * I], Q0[0], Gk[2 * I + 1], Q0[1]);
                        R1[I] = mm16x2(Gk[2 * I], Q1[0], Gk[2 * I + 1], Q1[1]);
                    }
                    double* pa = Jb + P.jo_a + (size_t)k * 512;
#pragma unroll
                    for (int I = 0; I < 2; ++I) {
                        const v4d Y = R0[I] + mm16x2(Ga[2 * I], R1[0], Ga[2 * I + 1], R1[1]);
                        store_T32(pa, mm16(Y, IdB), 16 * I, 0, g, j);
                    }
                }
            }
        }
        __syncthreads();   // GaL is rewritten by the next interval of a persistent grid
    }
}

}  // namespace

size_t qc_mfma32_gx_doubles(const QcParams& P) { return (size_t)(P.m + 1) * 4 * 256; }

// [matrix][tile = 2I+K][pair][lane][2]:  lane (g, i) reg kk = X[16I + i][16K + 4kk + g]  (X column-major 32 x 32)
void qc_mfma32_pack_G(const QcParams& P, const double* G, double* Gx) {
    const int n = 32, M = P.m + 1;
    for (int mat = 0; mat < M; ++mat) {
        const double* A = G + (size_t)mat * n * n;
        for (int tile = 0; tile < 4; ++tile) {
            const int I = tile >> 1, K = tile & 1;
            for (int pr = 0; pr < 2; ++pr)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 2; ++e) {
                        const int g = l >> 4, i = l & 15, kk = 2 * pr + e;
                        Gx[(((size_t)mat * 4 + tile) * 2 + pr) * 128 + l * 2 + e] = A[(size_t)(16 * K + 4 * kk + g) * n + 16 * I + i];
                    }
        }
    }
}

hipError_t qc_launch_mfma32_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    const int grid = P.n_int < kMaxGrid32 ? P.n_int : kMaxGrid32;
    if (dJ) hipLaunchKernelGGL(qc_mfma32_pade4_kernel<true>, dim3(grid), dim3(kThreads32), 0, st, P, dZ, dF, dJ);
    else hipLaunchKernelGGL(qc_mfma32_pade4_kernel<false>, dim3(grid), dim3(kThreads32), 0, st, P, dZ, dF, dJ);
    return hipGetLastError();
}
