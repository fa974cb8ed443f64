// f64-MFMA kernel for the order-4 Pade integrator at 32 < 2N <= 64 (17 .. 32 levels; 5 qubits: N = 32): every matrix is
// 4 x 4 tiles of 16 x 16 and lives in LDS (column-major, 66-double columns), every product is a chain of 16
// v_mfma_f64_16x16x4_f64 whose operands are read from LDS as the chain advances (lane maps: qc_mfma_kernels.hip header).
// Systems with fewer than 32 levels are zero-padded to the 64 x 64 tiles; stores are masked.
//
// With X column-major in LDS the two operand fragments of tile products are
//   row fragment  R of X:  lane (g, i), step q  ->  X[16 R + i][4 q + g]      (consecutive lanes = consecutive rows)
//   col fragment  C of X:  lane (g, j), step q  ->  X[4 q + g][16 C + j]
// mfma(a = row fragment R of A, b = col fragment C of B) accumulates tile (R, C) of A B in the D layout
// (lane (g, j) reg r = (A B)[16 R + 4 r + g][16 C + j]); with the operands SWAPPED, mfma(a = col fragment C of B,
// b = row fragment R of A), the same 16 instructions accumulate the TRANSPOSED tile (lane (g, j) reg r =
// (A B)[16 R + j][16 C + 4 r + g]): lanes run along rows, so every 8-byte-per-lane store writes four whole 128-byte lines
// of a column-major output block.  All outputs are produced that way; nothing is transposed after the fact.
//
// One 1024-thread workgroup (16 wavefronts, 152 KB of LDS) per interval (up to four per interval when there are fewer
// intervals than CUs: each takes every `parts`-th drive and copy), one workgroup per CU; the hardware deals the
// waves round-robin over the 4 SIMDs, so every SIMD hosts two compute and two copy waves.
//   phase 0   all threads: G = G_0 + sum_k a_k G_k (zero-padded global images, L2 hits), [S | D] = [U1 + U0 | U1 - U0].
//   phase A   compute wave (I, J'): tiles (I, J') of G S and G D (32 MFMAs) -> Q_0 = -c1 h S + c2 h^2 G D,
//             Q_h = -c1 S + 2 c2 h G D, D' = c2 h^2 D;   copy wave: two tiles of G^2 (32 MFMAs) -> B = I - c1 h G + c2 h^2 G^2.
//   phase B   compute wave: residual D + G Q_0 and d/dh = G Q_h (32 MFMAs), then per drive k:
//             d/da_k = G_k Q_0 + G (G_k D')  (48 MFMAs; the 64 x 32 product G_k D' is exchanged through a double-buffered LDS
//             block, one workgroup barrier per drive; the G_k fragments come straight from global memory / L2 and are
//             requested BEFORE the previous drive's stores are issued, so no load ever queues behind a store).
//             copy wave c: 8 columns of B and of -F = -(B + 2 c1 h G), kept in registers, written N times (I_N (x) B,
//             -I_N (x) F: 92 % of the bytes), a share of the copies between two drive barriers.
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kThreads64 = 1024;
constexpr int kLd = 66;                   // column stride of the LDS matrices (col fragments conflict-free)
constexpr int kMat = 64 * kLd;            // a 64-column matrix
constexpr int kHalf = 32 * kLd;           // a 32-column matrix
constexpr int oG = 0;                     // G
constexpr int oX = kMat;                  // [S | D], later [Q_0 | D']
constexpr int oB = 2 * kMat;              // B
constexpr int oQh = 3 * kMat;             // Q_h (32 columns)
constexpr int oV = 3 * kMat + kHalf;      // G_k D', double-buffered (2 x 32 columns)
constexpr int kLdsDoubles64 = 3 * kMat + 3 * kHalf;

__device__ __forceinline__ v4d mfma4(double a, double b, const v4d& c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// Two tile products sharing the b-operand fragment `fb` (registers); the a-operands are read from LDS as the chains advance:
//   acc0 += sum_q mfma(pa0[q * sa], fb[q]),  acc1 += sum_q mfma(pa1[q * sa], fb[q])
__device__ __forceinline__ void mm_a2_reg(const double* __restrict__ pa0, const double* __restrict__ pa1, int sa,
                                          const double (&fb)[16], v4d& acc0, v4d& acc1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        acc0 = mfma4(pa0[q * sa], fb[q], acc0);
        acc1 = mfma4(pa1[q * sa], fb[q], acc1);
    }
}
// one tile product, two accumulator chains (even / odd steps)
__device__ __forceinline__ v4d mm_a1_reg(const double* __restrict__ pa, int sa, const double (&fb)[16], const v4d& init) {
    v4d acc0 = init, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
        acc0 = mfma4(pa[q * sa], fb[q], acc0);
        acc1 = mfma4(pa[(q + 1) * sa], fb[q + 1], acc1);
    }
    return acc0 + acc1;
}

// Row fragment I of generator image `mat` (zero-padded 64 x 64, column-major, global memory): 16 loads of 8 bytes per lane,
// each a run of four whole 128-byte lines.
__device__ __forceinline__ void load_drive_fragment(const double* __restrict__ Gx, int mat, int I, int g, int j, double (&f)[16]) {
    const double* p = Gx + (size_t)mat * 4096 + g * 64 + 16 * I + j;
#pragma unroll
    for (int q = 0; q < 16; ++q) f[q] = p[q * 256];
}
// Makes the compiler wait for the fragment HERE (its own vmcnt wait precedes the first use of a loaded register; a store
// issued in between would otherwise have to complete first, microseconds when the copy waves saturate the memory pipeline).
__device__ __forceinline__ void settle_fragment(double (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(f[q]));
}

template <bool JAC>
__global__ __launch_bounds__(kThreads64, 1) void qc_mfma64_pade4_kernel(const QcParams P, const int parts, const double* __restrict__ Z,
                                                                       double* __restrict__ F, double* __restrict__ J) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* __restrict__ Gs = sm + oG;
    double* __restrict__ Xs = sm + oX;
    double* __restrict__ Bs = sm + oB;
    double* __restrict__ Qhs = sm + oQh;
    double* __restrict__ Vs = sm + oV;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int n = P.n, nc = P.nc, m = P.m;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gx = P.Gx;

    // Few intervals (T <~ 128: fewer workgroups than CUs): `parts` workgroups share an interval -- each repeats phases 0 / A,
    // then takes the drives k = part, part + parts, ... and the copies q = part, part + parts, ...; part 0 also stores the
    // residual, d/dh and the derivative-integrator rows.
    const int part = JAC ? (int)(blockIdx.x % (unsigned)parts) : 0;
    const int step = JAC ? parts : 1;
    const int ml = part < m ? (m - part + step - 1) / step : 0;        // drives of this workgroup
    const int b = qc_xcd_remap(blockIdx.x / (unsigned)step, P.n_int);
    const long long t = P.t_begin + b;
    const double* __restrict__ z0 = Z + t * (long long)P.zdim;
    const double* __restrict__ z1 = z0 + P.zdim;
    double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
    double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);
    const double hc1 = h * c1, hc2 = h * h * c2;

    // ---------------- phase 0: G and [S | D] into LDS ---------------------------------------------------------
    {
        double acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = Gx[tid + 1024 * e];
        for (int k = 0; k < m; ++k) {
            const double ak = z0[P.off_a + k];
            const double* __restrict__ Gk = Gx + (size_t)(k + 1) * 4096;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += ak * Gk[tid + 1024 * e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 1024 * e;
            Gs[(idx >> 6) * kLd + (idx & 63)] = acc[e];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + 1024 * e, r = idx & 63, c = idx >> 6;
            const bool ok = r < n && c < nc;
            const int off = ok ? P.off_U + c * n + r : P.off_U;
            const double u0 = ok ? z0[off] : 0.0, u1 = ok ? z1[off] : 0.0;
            Xs[c * kLd + r] = u1 + u0;
            Xs[(32 + c) * kLd + r] = u1 - u0;
        }
    }
    if (w == 15 && part == 0) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, false);   // derivative-integrator rows: loads, then a few stores
    __syncthreads();

    const bool compute = w < 8;
    if (compute) {
        // ================= compute wave (I, Jp): tile row I, state tile column Jp ========================================
        const int I = w & 3, Jp = w >> 2;
        const double* rowG = Gs + g * kLd + 16 * I + j;                // row fragment I of G, step q at + 4 q kLd
        const double* colS = Xs + (16 * Jp + j) * kLd + g;             // col fragment Jp of S (later Q_0), step q at + 4 q
        const double* colD = Xs + (32 + 16 * Jp + j) * kLd + g;        // col fragment Jp of D (later D')
        v4d GS = {0.0, 0.0, 0.0, 0.0}, GD = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 16; ++q) {                                 // normal orientation: a = row fragment of G
            const double a = rowG[q * 4 * kLd];
            GS = mfma4(a, colS[q * 4], GS);
            GD = mfma4(a, colD[q * 4], GD);
        }
        double* sN = Xs + (16 * Jp + j) * kLd + 16 * I + g;           // tile (I, Jp) of S in the D layout: + 4 r
        double* dT = Xs + (32 + 16 * Jp + g) * kLd + 16 * I + j;      // tile (I, Jp) of D, transposed layout: + 4 r kLd
        v4d Q0, Qh, Dt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double S = sN[4 * r];
            Q0[r] = -hc1 * S + hc2 * GD[r];
            Qh[r] = -c1 * S + 2.0 * c2 * h * GD[r];
            Dt[r] = dT[4 * r * kLd];
        }
        __syncthreads();                                               // every wave has read [S | D]
        double* qhN = Qhs + (16 * Jp + j) * kLd + 16 * I + g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sN[4 * r] = Q0[r];
            qhN[4 * r] = Qh[r];
            dT[4 * r * kLd] = hc2 * Dt[r];
        }
        __syncthreads();                                               // [Q_0 | D'], Q_h and B are complete

        double fG[16];                                                 // row fragment I of G: b-operand of every transposed product
#pragma unroll
        for (int q = 0; q < 16; ++q) fG[q] = rowG[q * 4 * kLd];
        double fk[16];                                                 // row fragment I of the current drive's generator
        if (JAC && ml > 0) load_drive_fragment(Gx, part + 1, I, g, j, fk);
        const double* colQh = Qhs + (16 * Jp + j) * kLd + g;
        v4d res = Dt, dh = {0.0, 0.0, 0.0, 0.0};
        if constexpr (JAC) {
            mm_a2_reg(colS, colQh, 4, fG, res, dh);                    // (D + G Q_0)^T, (G Q_h)^T
        } else {                                                       // one chain, the same summation order as above:
#pragma unroll                                                         // the residual-only launch returns bit-identical values
            for (int q = 0; q < 16; ++q) res = mfma4(colS[q * 4], fG[q], res);
        }
        if (JAC && ml > 0) settle_fragment(fk);
        const int row = 16 * I + j;
        const bool rok = row < n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * Jp + 4 * r + g;
            if (rok && col < nc && part == 0) {
                if (Fb) qc_st8m<2>(Fb + col * n + row, res[r]);
                if constexpr (JAC) { if (ft) qc_st8m<2>(Jb + P.jo_h + col * n + row, dh[r]); }
            }
        }
        if constexpr (JAC) {
            for (int kl = 0; kl < ml; ++kl) {
                const int k = part + kl * step;
                double* __restrict__ Vk = Vs + (kl & 1) * kHalf;
                v4d T1L = {0.0, 0.0, 0.0, 0.0}, T1R = {0.0, 0.0, 0.0, 0.0};
                mm_a2_reg(colS, colD, 4, fk, T1L, T1R);                // (G_k Q_0)^T, (G_k D')^T
                double* vT = Vk + (16 * Jp + g) * kLd + 16 * I + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) vT[4 * r * kLd] = T1R[r];
                if (kl + 1 < ml) load_drive_fragment(Gx, k + step + 1, I, g, j, fk);
                __syncthreads();                                       // G_k D' complete (the other buffer is free again)
                const v4d Y = mm_a1_reg(Vk + (16 * Jp + j) * kLd + g, 4, fG, T1L);   // + (G (G_k D'))^T
                if (kl + 1 < ml) settle_fragment(fk);
                double* pa = Jb + P.jo_a + (size_t)k * P.s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * Jp + 4 * r + g;
                    if (rok && col < nc) qc_st8m<2>(pa + col * n + row, Y[r]);
                }
            }
        }
    } else {
        // ================= copy wave c: two tiles of G^2 -> B; then 8 columns of B and -F, written nc times ===========
        const int c = w - 8, I = c & 3, Jq = c >> 2;
        const double* rowG = Gs + g * kLd + 16 * I + j;
        if constexpr (JAC) {
            const double* col0 = Gs + (32 * Jq + j) * kLd + g;         // col fragments 2 Jq and 2 Jq + 1 of G
            const double* col1 = col0 + 16 * kLd;
            v4d S0 = {0.0, 0.0, 0.0, 0.0}, S1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double a = rowG[q * 4 * kLd];
                S0 = mfma4(a, col0[q * 4], S0);
                S1 = mfma4(a, col1[q * 4], S1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * I + 4 * r + g, cA = 32 * Jq + j, cB = cA + 16;
                Bs[cA * kLd + rr] = (rr == cA ? 1.0 : 0.0) - hc1 * Gs[cA * kLd + rr] + hc2 * S0[r];
                Bs[cB * kLd + rr] = (rr == cB ? 1.0 : 0.0) - hc1 * Gs[cB * kLd + rr] + hc2 * S1[r];
            }
        }
        __syncthreads();
        __syncthreads();
        if constexpr (JAC) {
            double bc[8], fc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int col = 8 * c + e;
                bc[e] = Bs[col * kLd + lane];
                fc[e] = -(bc[e] + 2.0 * hc1 * Gs[col * kLd + lane]);
            }
            const bool rok = lane < n;
            const size_t n2 = (size_t)n * n;
            double* __restrict__ pF = Jb + P.jo_F + (size_t)(8 * c) * n + lane;
            double* __restrict__ pB = Jb + P.jo_B + (size_t)(8 * c) * n + lane;
            const int ncl = part < nc ? (nc - part + step - 1) / step : 0;   // copies of this workgroup: q = part + ql * step
            const int share = ml > 0 ? (ncl + ml - 1) / ml : ncl;             // copies between two drive barriers
            int q0 = 0;
            for (int kl = 0; kl <= ml; ++kl) {
                const int q1 = kl < ml ? (q0 + share < ncl ? q0 + share : ncl) : ncl;
                for (int ql = q0; ql < q1; ++ql) {
                    const size_t qo = (size_t)(part + ql * step) * n2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (rok && 8 * c + e < n) {
                            qc_st8m<2>(pF + qo + (size_t)e * n, fc[e]);
                            qc_st8m<2>(pB + qo + (size_t)e * n, bc[e]);
                        }
                    }
                }
                q0 = q1;
                if (kl < ml) __syncthreads();
            }
        }
    }
}

}  // namespace

size_t qc_mfma64_gx_doubles(const QcParams& P) { return (size_t)2 * (P.m + 1) * 4096; }

// Zero-padded 64 x 64 column-major copies of the (m+1) generators, followed by those of their transposes (Hessian kernel).
void qc_mfma64_pack_G(const QcParams& P, const double* G, double* Gx) {
    const int n = P.n, M = P.m + 1;
    for (int mat = 0; mat < M; ++mat)
        for (int c = 0; c < 64; ++c)
            for (int r = 0; r < 64; ++r) {
                const bool in = r < n && c < n;
                Gx[(size_t)mat * 4096 + c * 64 + r] = in ? G[(size_t)mat * n * n + (size_t)c * n + r] : 0.0;
                Gx[(size_t)(M + mat) * 4096 + c * 64 + r] = in ? G[(size_t)mat * n * n + (size_t)r * n + c] : 0.0;
            }
}

bool qc_mfma64_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n > 32 && P.n <= 64 && P.nc <= 32;
}

hipError_t qc_launch_mfma64_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    constexpr size_t lds = (size_t)kLdsDoubles64 * sizeof(double);
    // per launch: the attribute belongs to the (function, device) pair and the call is a host-side table update
    hipError_t e = dJ ? hipFuncSetAttribute(reinterpret_cast<const void*>(qc_mfma64_pade4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                      : hipFuncSetAttribute(reinterpret_cast<const void*>(qc_mfma64_pade4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (P.n_int <= 0) return hipSuccess;
    // fewer intervals than CUs: up to 4 workgroups share an interval (drives and copies dealt round-robin)
    int parts = 1;
    if (dJ) { while (parts < 4 && 2 * parts * P.n_int <= 256) parts *= 2; }
    if (dJ) hipLaunchKernelGGL((qc_mfma64_pade4_kernel<true>), dim3((unsigned)P.n_int * parts), dim3(kThreads64), lds, st, P, parts, dZ, dF, dJ);
    else hipLaunchKernelGGL((qc_mfma64_pade4_kernel<false>), dim3(P.n_int), dim3(kThreads64), lds, st, P, 1, dZ, dF, dJ);
    return hipGetLastError();
}
