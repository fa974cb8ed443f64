// f64-MFMA kernel for the EXPONENTIAL integrator at 2N = 32 (4 qubits), up to 8 drives: residual
//     delta = U_t+1 - exp(h G(a_t)) U_t                                          (reference README.md:79)
// and the Jacobian blocks  d/dU_t = -I_N (x) E,  d/dU_t+1 = I,  d/da_j = -L_j U_t,  d/dh = -G E U_t   (E = exp(h G),
// L_j = L_exp(h G; h G_j)).  The 2N = 16 kernel (qc_mfma_exp.hip) explains the algorithm: scaling and squaring with
// ||Y||_1 <= 1/8, a degree-8 Taylor polynomial and its Frechet derivatives in Horner form on R_k = P_k / (k-1)!,
//     R_k = Y R_k+1 + I/(k-1)!          R'_k,j = G_j R_k+1 + Y R'_k+1,j,
// squarings E <- E E, L_j <- E L_j + L_j E with the left factors obtained as LDS-transposed tiles (a D-layout tile read
// as the A operand acts as its transpose), the factor h / 2^sq applied to the outputs once.
//
// Here every matrix is 2 x 2 tiles of 16 x 16 and one 512-thread workgroup (8 wavefronts) serves one interval:
//   wave k      owns drive k: its four A-layout image tiles of G_j and its chain R'_j (4 tiles) stay in registers;
//   waves 0-3   additionally own one tile of the shared chain R (and of E in the squarings), published through a
//               double-buffered LDS block; one barrier per Horner step / squaring.
// Per step a drive wave issues 64 MFMAs (16 products), the R owners 8 more.  Outputs leave transposed (lane <-> row,
// whole 128-byte lines): wave w stores copies w and w + 8 of -E from the transposed tiles of E, its drive's columns as
// (L_j U_t)^T = U_t^T L_j^T; wave 0 the residual and d/dh.  About 7200 MFMAs per interval at m = 8 with 3 squarings:
// MFMA-pipe-bound (~110 us floor for config 5 at T = 500); the LDS kernel needs 4.1 ms for the same problem.
#include <stdlib.h>

#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kE32Deg = 8;          // with ||Y||_1 <= 1/8: truncation (1/8)^9 / 9! = 4e-14; one step fewer than degree 10 at 1/4
constexpr double kE32Th = 0.125;
constexpr int kE32Mmax = 8;
constexpr int kE32Threads = 512;

__device__ inline v4d e32_tile(const double* __restrict__ base, int tile, int lane) {   // [tile][pair][lane][2], global or LDS
    const v2d* p = reinterpret_cast<const v2d*>(base) + tile * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline void e32_put(double* __restrict__ base, int tile, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + tile * 128 + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}
template <int CTRL>
__device__ inline double e32_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double e32_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// acc (+)= A0 * B0 + A1 * B1 for one output tile: one chain of 8 MFMAs on top of `c`
__device__ __forceinline__ v4d e32_mac2(const v4d& a0, const v4d& b0, const v4d& a1, const v4d& b1, v4d c) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b0[kk], c, 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], c, 0, 0, 0);
    return c;
}

// Q'[I][J] = sum_K ( A1[I][K] R[K][J] + A2[I][K] Q[K][J] )   for the four output tiles, MFMAs interleaved over the tiles.
// Tiles of a 32 x 32 matrix are indexed 2 * (row block) + (column block).
__device__ __forceinline__ void e32_chain4(const v4d (&A1)[4], const v4d (&R)[4], const v4d (&A2)[4], const v4d (&Q)[4], v4d (&out)[4]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d acc[4] = {z, z, z, z};
#pragma unroll
    for (int K = 0; K < 2; ++K) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int I = t >> 1, J = t & 1;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1[2 * I + K][kk], R[2 * K + J][kk], acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int K = 0; K < 2; ++K) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int I = t >> 1, J = t & 1;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A2[2 * I + K][kk], Q[2 * K + J][kk], acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = acc[t];
}

// lane (g, j) reg r = X[rowbase + j][colbase + 4 r + g] of a column-major block with `ld` rows per column at p
__device__ inline void e32_store_T(double* __restrict__ p, const v4d& x, int ld, int rowbase, int colbase, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (colbase + 4 * r + g < ld && rowbase + j < ld) qc_st8m<2>(p + (size_t)(colbase + 4 * r + g) * ld + rowbase + j, x[r]);
}

// ELL: every drive generator has at most ONE entry per row (Pauli strings; P.ell16 = qc_exp_ell_build's tables, qc_mfma_exp_hess.hip): the
// product G_j R of a Horner step is a row gather from a row-major copy of R that the owners of R publish next to the tiles -- 32 MFMAs
// per drive wave and step instead of 64.  fma(w, x, acc) per element: what the dense product adds besides exact zeros.
constexpr int kRS = 33;             // row stride of the row-major copies (doubles): the lanes of a gather fall on distinct banks

template <bool JAC, bool ELL = false>
__global__ __launch_bounds__(kE32Threads, 1) void qc_mfma32_exp_kernel(const QcParams P, const double* __restrict__ Z,
                                                                       double* __restrict__ F, double* __restrict__ J) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    __shared__ __attribute__((aligned(16))) double GL[4 * 256];          // G (unscaled), A-layout tiles 2I+K
    __shared__ __attribute__((aligned(16))) double RL[2][4 * 256];       // the shared chain R / E, D-layout tiles 2K+J, double-buffered
    __shared__ double RR[ELL ? 2 : 1][ELL ? 32 * kRS : 1];               // ELL: R row-major (the gathers' source), double-buffered
    __shared__ __attribute__((aligned(16))) double U1L[2 * 256];         // U_t+1 (wave 0), parked from the first loads to the residual
    __shared__ __attribute__((aligned(16))) double RT[2][4 * 256];       // E transposed tile by tile (read as an A operand: acts as the tile), by its owners
    __shared__ double TS[8 * 16 * 17];                                   // per-wave transpose scratch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15;
    const bool ft = P.off_dt >= 0;
    const bool drive = JAC && w < m;
    const double* __restrict__ GxA = P.Gx;                               // A-layout images [mat][2I+K]
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    double* __restrict__ scr = TS + w * (16 * 17);

    const int b = qc_xcd_remap((int)blockIdx.x, P.n_int);
#ifdef QC_X32_STAMPS      // diagnostic variant build (profiles/stamps_exp32.py): wave 0 -> slots 0-7, wave 5 -> slots 8-15
    constexpr bool DIAG = true;
    QC_STAMP_DECL;
#define X32_STAMP(k) QC_STAMP(P, b, lane, k)
    X32_STAMP(0);
#else
#define X32_STAMP(k)
#endif
    const long long t = P.t_begin + b;
    const double* __restrict__ z0 = Z + t * (long long)P.zdim;
    const double* __restrict__ z1 = z0 + P.zdim;
    double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
    double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);

    // ---- loads: this wave's half tile of the generator images (assembly), its drive's images, U_t ---------------------
    v4d Gj[4];
    double tw[2][4];                                                     // ELL: weight and source offset (column x kRS + j) of rows 16 I + 4 r + g of this wave's drive
    int tc[2][4];
    if constexpr (ELL) {
        const double* __restrict__ bw = reinterpret_cast<const double*>(P.ell16);
        const int* __restrict__ bc = reinterpret_cast<const int*>(reinterpret_cast<const char*>(P.ell16) + kE32Mmax * 32 * 8);
        const int k = drive ? w : 0;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double wt = bw[k * 32 + 16 * I + 4 * r + g];
                tw[I][r] = drive ? wt : 0.0;
                tc[I][r] = bc[k * 32 + 16 * I + 4 * r + g] * kRS + j;
            }
        }
    } else {
        const int kmat = drive ? w + 1 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Gj[q] = e32_tile(GxA + (size_t)kmat * 1024, q, lane);
    }
    // state columns: N = 16 for a unitary, K <= 16 for K kets, 1 for a density operator (N^2 = 16 levels).  Tile columns
    // >= nc re-read column 0 and are never stored (the kernel is MFMA-bound: the run-time masks cost nothing here).
    // Systems with 9 .. 15 levels: nr = 2N < 32 rows, zero-padded to the 2 x 2 tiles (the exponential of the padded generator is
    // the exponential of the true one plus an identity block that is never stored).
    const int nc = P.nc, jc = j < nc ? j : 0, nr = P.n;
    v4d U[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + 4 * r + g; U[I][r] = row < nr ? z0[P.off_U + jc * nr + row] : 0.0; }
    }
    // U_t+1 for the residual: requested HERE by the wave that needs it at the end and parked in LDS (a load issued behind the interval's
    // stores waits for all of them -- it sat in wave 0's tail, the longest of the workgroup)
    if (w == 0 && Fb) {
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            v4d u1;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = 16 * I + 4 * r + g; u1[r] = row < nr ? z1[P.off_U + jc * nr + row] : 0.0; }
            e32_put(U1L, I, lane, u1);
        }
    }
    {
        const v2d* __restrict__ ab = reinterpret_cast<const v2d*>(GxA) + (w >> 1) * 128 + (w & 1) * 64 + lane;
        v2d img[kE32Mmax + 1];
        double ak[kE32Mmax];
#pragma unroll
        for (int u = 0; u <= kE32Mmax; ++u) img[u] = ab[(size_t)(u <= m ? u : 0) * 512];
#pragma unroll
        for (int u = 0; u < kE32Mmax; ++u) ak[u] = z0[P.off_a + (u < m ? u : 0)];
        v2d Gh = img[0];
#pragma unroll
        for (int u = 0; u < kE32Mmax; ++u) Gh += (u < m ? ak[u] : 0.0) * img[u + 1];
        reinterpret_cast<v2d*>(GL)[(w >> 1) * 128 + (w & 1) * 64 + lane] = Gh;
    }
    if (w < 4) {           // R_deg+1 = I/deg!, tile (I, J) by wave 2 I + J: the owners read their operand tiles of R from LDS in every step
        double f0 = 1.0;   // (indexing the register copy R[] with the wave's tile index put the array into scratch memory: four
#pragma unroll             //  scratch loads inside every owner chain -- 2.8 instead of 2.0 us per Horner step, profiles/r06_exp_hess.txt)
        for (int k = 2; k <= kE32Deg; ++k) f0 *= (double)k;
        v4d r0;
#pragma unroll
        for (int r = 0; r < 4; ++r) r0[r] = ((w >> 1) == (w & 1) && 4 * r + g == j) ? 1.0 / f0 : 0.0;
        e32_put(RL[0], w, lane, r0);
        if constexpr (ELL) {   // ... and row-major for the first step's gathers
#pragma unroll
            for (int r = 0; r < 4; ++r) RR[0][(16 * (w >> 1) + 4 * r + g) * kRS + 16 * (w & 1) + j] = r0[r];
        }
    }
    X32_STAMP(1);
    __syncthreads();
    X32_STAMP(2);

    // ---- ||h G||_1 (every wave, redundantly): lane (g, i) reg kk of tile (I, K) holds G[16I+i][16K+4kk+g] ---------------
    v4d Y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) Y[q] = e32_tile(GL, q, lane);
    int sq = 0;
    {
        double best = 0.0;
        bool bad = false;
#pragma unroll
        for (int K = 0; K < 2; ++K) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double c = fabs(h * Y[K][kk]) + fabs(h * Y[2 + K][kk]);
                c += e32_dpp<0x128>(c);
                c += e32_dpp<0x124>(c);
                c += e32_dpp<0x122>(c);
                c += e32_dpp<0x121>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = e32_readlane(c, 16 * r);
                    if (!(v == v) || v > 1e300) bad = true;
                    best = fmax(best, v);
                }
            }
        }
        if (!bad && best > kE32Th) {
            int e;
            (void)frexp(best / kE32Th, &e);
            sq = e;
            if (ldexp(kE32Th, e - 1) >= best) sq = e - 1;
            sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
        }
    }
    const double sc = ldexp(1.0, -sq);
#pragma unroll
    for (int q = 0; q < 4; ++q) Y[q] = (h * sc) * Y[q];

    // ---- Horner: R_deg+1 = I/deg!, R' = 0 --------------------------------------------------------------------------------
    double fact = 1.0;
#pragma unroll
    for (int k = 2; k <= kE32Deg; ++k) fact *= (double)k;
    double ck = 1.0 / fact;
    v4d R[4] = {ck * IdB, zero, zero, ck * IdB};          // tiles (0,0), (0,1), (1,0), (1,1)
    v4d Q[4] = {zero, zero, zero, zero};
    int cur = 0;
    X32_STAMP(3);
#pragma unroll 1
    for (int k = kE32Deg; k >= 1; --k) {
        ck *= (double)k;                                   // 1/(k-1)!
        if (w < 4) {                                       // tile (I, J) of R_k = Y R_k+1 + ck I
            const int I = w >> 1, Jt = w & 1;
            const v4d c0 = I == Jt ? ck * IdB : zero;
            const v4d yA = I ? Y[2] : Y[0], yB = I ? Y[3] : Y[1];      // (selects, not Y[2 * I]: see above)
            const v4d rn = e32_mac2(yA, e32_tile(RL[cur], Jt, lane), yB, e32_tile(RL[cur], 2 + Jt, lane), c0);
            e32_put(RL[cur ^ 1], w, lane, rn);
            if constexpr (ELL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) RR[cur ^ 1][(16 * I + 4 * r + g) * kRS + 16 * Jt + j] = rn[r];
            }
            if (k == 1) e32_put(RT[cur ^ 1], w, lane, lds_transpose16(scr, rn, g, j));      // the squarings and the outputs read the transposed tiles
        }
        if constexpr (ELL) {
            if (drive) {   // Q_j <- Y Q_j + G_j R: the gathers requested, the products, the gathered terms added
                const double* __restrict__ rr = RR[cur];
                double x[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[t][r] = rr[tc[t >> 1][r] + 16 * (t & 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
                v4d acc[4] = {zero, zero, zero, zero};
#pragma unroll
                for (int K = 0; K < 2; ++K) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[2 * (t >> 1) + K][kk], Q[2 * K + (t & 1)][kk], acc[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Q[t][r] = __builtin_fma(tw[t >> 1][r], x[t][r], acc[t][r]);
                }
            }
        } else {
            if (drive) e32_chain4(Gj, R, Y, Q, Q);
        }
        __syncthreads();
        cur ^= 1;
        if (!ELL || k == 1) {             // (the row-gather form's drive chains gather from RR: no wave needs R in registers until the last step)
#pragma unroll
            for (int q = 0; q < 4; ++q) R[q] = e32_tile(RL[cur], q, lane);
        }
    }
    X32_STAMP(4);
    // ---- squarings: E <- E E, L_j <- E L_j + L_j E (left factors: E^T tiles published by E's owners, L_j^T by LDS transposes) ---------
    for (int s = 0; s < sq; ++s) {
        v4d Et[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) Et[q] = e32_tile(RT[cur], q, lane);           // Et[2I+K] read as A acts as E[I][K]
        if (w < 4) {
            const int I = w >> 1, Jt = w & 1;
            const v4d en = e32_mac2(e32_tile(RT[cur], 2 * I, lane), e32_tile(RL[cur], Jt, lane), e32_tile(RT[cur], 2 * I + 1, lane),
                                    e32_tile(RL[cur], 2 + Jt, lane), zero);      // (operand tiles from LDS by tile index: no register array is indexed)
            e32_put(RL[cur ^ 1], w, lane, en);
            e32_put(RT[cur ^ 1], w, lane, lds_transpose16(scr, en, g, j));
        }
        if (drive) {
            // E L_j first (its operands are there), the transposes of L_j in its shadow, then L_j E
            const v4d z = {0.0, 0.0, 0.0, 0.0};
            v4d acc[4] = {z, z, z, z};
#pragma unroll
            for (int K = 0; K < 2; ++K) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Et[2 * (t >> 1) + K][kk], Q[2 * K + (t & 1)][kk], acc[t], 0, 0, 0);
                }
            }
            v4d Lt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) Lt[q] = lds_transpose16(scr, Q[q], g, j);
#pragma unroll
            for (int K = 0; K < 2; ++K) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[2 * (t >> 1) + K][kk], R[2 * K + (t & 1)][kk], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) Q[q] = acc[q];
        }
        __syncthreads();
        cur ^= 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) R[q] = e32_tile(RL[cur], q, lane);
    }

    X32_STAMP(5);
    // ---- outputs -----------------------------------------------------------------------------------------------------------
    v4d Et[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) Et[q] = e32_tile(RT[cur], q, lane);               // transposed tiles: A operands acting as E[I][K], and what is stored
    if (w == 0) {
        // E U_t, the residual, d/dh = -G E U_t (transposed for the stores through the LDS scratch) -- in front of this wave's copies of -E
        // and its drive's block: the residual and d/dh are a chain of products and transposes, the longest tail of the workgroup when last
        v4d EU[2];
#pragma unroll
        for (int I = 0; I < 2; ++I) EU[I] = e32_mac2(Et[2 * I], U[0], Et[2 * I + 1], U[1], zero);
        if (Fb) {
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const v4d dT = lds_transpose16(scr, e32_tile(U1L, I, lane) - EU[I], g, j);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * r + g < nc && 16 * I + j < nr) qc_st8m<2>(Fb + (4 * r + g) * nr + 16 * I + j, dT[r]);
            }
        }
        if (JAC && ft) {
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const v4d ge = e32_mac2(e32_tile(GL, 2 * I, lane), EU[0], e32_tile(GL, 2 * I + 1, lane), EU[1], zero);
                const v4d hT = lds_transpose16(scr, -ge, g, j);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * r + g < nc && 16 * I + j < nr) qc_st8m<2>(Jb + P.jo_h + (4 * r + g) * nr + 16 * I + j, hT[r]);
            }
        }
    }
    if constexpr (JAC) {
        // copies w and w + 8 of -E:  Et[2K+J] lane (g, j) reg r = E[16K+j][16J+4r+g]
        double* pF = Jb + P.jo_F;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (w + 8 * c < nc) {
                double* p = pF + (size_t)(w + 8 * c) * nr * nr;
#pragma unroll
                for (int q = 0; q < 4; ++q) e32_store_T(p, -Et[q], nr, 16 * (q >> 1), 16 * (q & 1), g, j);
            }
        }
        if (w == 7) for (int i = lane; i < P.s; i += 64) Jb[P.jo_B + i] = 1.0;
        if (drive) {
            // d/da_j = -(h/2^sq) L_j U_t, transposed: (L_j U_t)^T[.., 16J..] = sum_K U_t[K]^T (L_j^T)[K][J],  (L_j^T)[K][J] = (L_j[J][K])^T
            v4d Lt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) Lt[q] = lds_transpose16(scr, Q[q], g, j);     // Lt[2I+K] = (L_j[I][K])^T in D layout
            const double fac = -(h * sc);
            double* pa = Jb + P.jo_a + (size_t)w * P.s;
#pragma unroll
            for (int Jt = 0; Jt < 2; ++Jt) {
                const v4d x = e32_mac2(U[0], Lt[2 * Jt], U[1], Lt[2 * Jt + 1], zero);   // K = 0: (L_j[J][0])^T, K = 1: (L_j[J][1])^T
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * r + g < nc && 16 * Jt + j < nr) qc_st8m<2>(pa + (4 * r + g) * nr + 16 * Jt + j, fac * x[r]);
            }
        }
    }
    if (w == 6) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, false);
#ifdef QC_X32_STAMPS
    X32_STAMP(6);
    __builtin_amdgcn_s_waitcnt(0);        // (every store acknowledged)
    X32_STAMP(7);
    if (P.stamps != nullptr && lane == 0 && (w == 0 || w == 5)) {
#pragma unroll
        for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + (w == 0 ? 0 : 8) + k_] = qc_ts_[k_];
    }
#endif
}

}  // namespace

bool qc_mfma32_exp_supported(const QcParams& P) {
    return P.integrator == QC_EXPONENTIAL && P.n > 16 && P.n <= 32 && P.nc <= 16 && P.m <= kE32Mmax;
}

hipError_t qc_launch_mfma32_exp(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    static const bool ell_off = getenv("QC_EXP_ELL") && atoi(getenv("QC_EXP_ELL")) == 0;      // A/B diagnostics
    if (dJ && P.ell16 != nullptr && !ell_off) hipLaunchKernelGGL((qc_mfma32_exp_kernel<true, true>), dim3(P.n_int), dim3(kE32Threads), 0, st, P, dZ, dF, dJ);
    else if (dJ) hipLaunchKernelGGL(qc_mfma32_exp_kernel<true>, dim3(P.n_int), dim3(kE32Threads), 0, st, P, dZ, dF, dJ);
    else hipLaunchKernelGGL(qc_mfma32_exp_kernel<false>, dim3(P.n_int), dim3(kE32Threads), 0, st, P, dZ, dF, dJ);
    return hipGetLastError();
}
