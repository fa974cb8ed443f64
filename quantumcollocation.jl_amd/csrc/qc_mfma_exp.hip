// f64-MFMA kernel for the EXPONENTIAL integrator at 2N = 16 (3 qubits), up to 8 drives: residual
//     delta = U_t+1 - exp(h G(a_t)) U_t                                          (reference README.md:79)
// and its Jacobian blocks  d/dU_t = -I_N (x) E,  d/dU_t+1 = I,  d/da_j = -L_j U_t,  d/dh = -G E U_t,
// with E = exp(h G), L_j = L_exp(h G; h G_j) the Frechet derivative.  ONE wavefront per interval; every matrix is one
// 16 x 16 tile in registers (lane maps: qc_mfma_kernels.hip header).
//
// Scaling and squaring, Y = h G / 2^sq with ||Y||_1 <= 1/8, degree-8 Taylor polynomial (truncation 4e-14)
// in Horner form on R_k = P_k / (k-1)!  (P_k = I + Y/k P_k+1):
//     R_k  = Y R_k+1 + I/(k-1)!          R'_k,j = G_j R_k+1 + Y R'_k+1,j           k = 8 .. 1,   R_9 = I/8!,  R'_9 = 0
// so a step is 4 + 8 m MFMAs with the constant A-layout tiles Y, G_j as A operands and the previous outputs as B
// operands (D layout = B layout): nothing but MFMAs touches the accumulators, the identity enters as the C operand.
// R_1 = exp(Y), R'_1,j = L_exp(Y; G_j).  Squarings  E <- E E,  L_j <- E L_j + L_j E  need E and L_j as LEFT factors:
// a D-layout tile read as the A operand acts as its transpose, so E^T and L_j^T are made by LDS transposes (8 LDS
// operations each, no MFMA) and used as A operands.  L_j is linear in its direction, so the factor h / 2^sq is applied
// once to the outputs.  Outputs leave transposed (lane <-> row, whole 128-byte lines per store): the copies of -E
// from E^T, the drive columns as (L_j U_t)^T = U_t^T L_j^T (A = U_t tile, B = L_j^T), residual and d/dh through LDS.
// MFMAs per interval (m = 6, 4 squarings): 8 x 52 + 4 x 52 + 8 + 24 = 656;  the LDS kernel spends 463 us on config 3.
#include <stdlib.h>

#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kXDeg = 8;          // with ||Y||_1 <= 1/8: truncation (1/8)^9 / 9! = 4e-14; one step fewer than degree 10 at 1/4
constexpr double kXTh = 0.125;
constexpr int kXMmax = 8;

__device__ inline v4d ximg(const double* __restrict__ Gx, int mat, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(Gx) + mat * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
template <int CTRL>
__device__ inline double xdpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double xreadlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// one Horner step (or, with c = 0, one squaring with A operands (Et, Et, Kt_j) in place of (Y, G_j, Y)):
//   R <- A0 R + c I,   Q_j <- A1_j R + A2 Q_j       (A2_j per drive when PER_DRIVE_A2, the squaring's  L_j^T-as-A E term)
template <int M, bool JAC>
__device__ __forceinline__ void horner_step(const v4d& Y, const v4d (&Gj)[M], v4d& R, v4d (&Q)[M], const v4d& cI) {
    v4d accR = cI;
    v4d acc[M];
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    accR = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[0], R[0], accR, 0, 0, 0);
    if constexpr (JAC) {
#pragma unroll
        for (int q = 0; q < M; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gj[q][0], R[0], z, 0, 0, 0);
    }
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
        accR = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], R[kk], accR, 0, 0, 0);
        if constexpr (JAC) {
#pragma unroll
            for (int q = 0; q < M; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gj[q][kk], R[kk], acc[q], 0, 0, 0);
        }
    }
    if constexpr (JAC) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int q = 0; q < M; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Q[q][kk], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < M; ++q) Q[q] = acc[q];
    }
    R = accR;
}

// The same step for drive generators with at most ONE entry per row (P.ell16 = qc_exp_ell_build's tables, qc_mfma_exp_hess.hip):
// G_j R is a row gather from a row-major LDS copy of R -- 4 + 4 m MFMAs a step instead of 4 + 8 m.  The gathers are requested before
// the products and added behind them: fma(w, x, acc), what the dense product adds besides exact zeros.
template <int M>
__device__ __forceinline__ void horner_step_ell(const v4d& Y, const double (&tw)[M][4], const int (&tc)[M][4], double* __restrict__ scr, v4d& R,
                                                v4d (&Q)[M], const v4d& cI, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(4 * r + g) * 17 + j] = R[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double x[M][4];
#pragma unroll
    for (int q = 0; q < M; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[q][r] = scr[tc[q][r]];
    }
    __builtin_amdgcn_sched_barrier(0);
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d accR = cI, acc[M];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        accR = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], R[kk], accR, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < M; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Q[q][kk], kk == 0 ? z : acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < M; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Q[q][r] = __builtin_fma(tw[q][r], x[q][r], acc[q][r]);
    }
    R = accR;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next step's copy follows this step's gathers
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// squaring: E <- Et^T E (= E E),  L_j <- Et^T L_j + Kt_j^T E (= E L_j + L_j E);  Et, Kt_j are the transposed tiles
template <int M, bool JAC>
__device__ __forceinline__ void square_step(const v4d& Et, const v4d (&Kt)[M], v4d& E, v4d (&L)[M]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d accE = z, acc[M];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        accE = __builtin_amdgcn_mfma_f64_16x16x4f64(Et[kk], E[kk], accE, 0, 0, 0);
        if constexpr (JAC) {
#pragma unroll
            for (int q = 0; q < M; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Et[kk], L[q][kk], kk == 0 ? z : acc[q], 0, 0, 0);
        }
    }
    if constexpr (JAC) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int q = 0; q < M; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kt[q][kk], E[kk], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < M; ++q) L[q] = acc[q];
    }
    E = accE;
}

// kW wavefronts per interval, each with its own kMU drives (and, redundantly, the shared chain R: 4 MFMAs per step).  The
// waves never exchange data: no barrier.  Two waves per interval fill the chip on short trajectories (T = 200: 18.3 -> 11.9 us);
// at T = 1000 one wave per interval already occupies every SIMD and the kernel is bound by MFMA issue at the clock the
// device sustains under FP64 matrix load (two waves there: 30.8 vs 30.3 us, the redundant R chain costs what the overlap gains).
// (Two-wave forms: TWO intervals per four-wave workgroup -- placement only, the waves never synchronise: a four-wave workgroup takes
//  one wave slot on each SIMD of its CU, where two-wave workgroups left SIMDs empty at 257 - 512 intervals; qc_mfma_exp_hess.hip.)
template <bool JAC, int kMU, int kW, bool ELL = false, int kIPW = 1>
__global__ __launch_bounds__(64 * kW * kIPW, kW) void qc_mfma16_exp_kernel(const QcParams P, const double* __restrict__ Z, double* __restrict__ F,
                                                                    double* __restrict__ J) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    static_assert(kIPW == 1 || kW == 2, "two intervals per workgroup: the two-wave forms");
    __shared__ double scr_all[kIPW * kW * (kMU + 1) * 16 * 17];      // per-wave transpose scratch: E and the kMU L_j tiles in one LDS round trip
    const int lane = threadIdx.x & 63;
    const int wq = kIPW * kW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;      // wave of the workgroup
    const int wv = kW > 1 ? wq % kW : 0;                      // wave of its interval
    double* __restrict__ scr = scr_all + wq * ((kMU + 1) * 16 * 17);
    const int d0 = wv * kMU;                          // first drive of this wave
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool ft = P.off_dt >= 0;
    const double* __restrict__ Gx = P.Gx;            // A-layout images [mat][pair][lane][2]
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    {   // one interval per workgroup: a grid-stride loop lets the compiler hoist interval-invariant tiles (the twelve scaled
        // identities, 96 VGPRs) out of the body and spill
        const int b = qc_xcd_remap((int)blockIdx.x, (P.n_int + kIPW - 1) / kIPW) * kIPW + wq / kW;
        if (b >= P.n_int) return;                             // (an odd interval count: the last workgroup's second pair has nothing to do)
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
        double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
        const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);

        // ---- loads (one batch): knots, generator images --------------------------------------------------------
        // K kets: columns >= nc re-read column 0 and are never stored; N < 8 levels: nr = 2N < 16 rows, zero-padded to the tile
        // (the exponential of the padded generator is the exponential of the true one plus an identity block that is not stored)
        const int nc = P.nc, jc = jj < nc ? jj : 0, nr = P.n;
        v4d u0, u1;                                              // [U_t | U_t]  (both 8-column halves hold the same matrix)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + g;
            u0[r] = row < nr ? z0[P.off_U + jc * nr + row] : 0.0;
            u1[r] = row < nr ? z1[P.off_U + jc * nr + row] : 0.0;
        }
        v4d Gj[kMU];
        double tw[kMU][4];                                        // ELL: weight and LDS offset (column x 17 + j) of rows 4 r + g of the wave's drives
        int tc[kMU][4];
        v4d Ga = ximg(Gx, 0, lane);
        {   // G = G_0 + sum over ALL drives (every wave assembles it); unconditional clamped loads, one batch
            constexpr int kMA = kXMmax;       // every drive enters G, whatever subset this wave differentiates
            v4d img[kMA];
            double ak[kMA];
#pragma unroll
            for (int u = 0; u < kMA; ++u) {
                const int k = u < m ? u : (m > 0 ? m - 1 : 0);
                img[u] = ximg(Gx, m > 0 ? k + 1 : 0, lane);
                ak[u] = (u < m) ? z0[P.off_a + k] : 0.0;
            }
            if constexpr (ELL) {   // rows 4 r + g of the wave's drives (unused drive slots: weight 0, column 0 -- their chains stay zero)
                const double* __restrict__ bw = reinterpret_cast<const double*>(P.ell16);
                const int* __restrict__ bc = reinterpret_cast<const int*>(reinterpret_cast<const char*>(P.ell16) + kXMmax * 32 * 8);
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tw[u][r] = bw[(d0 + u) * 32 + 4 * r + g];
                        tc[u][r] = bc[(d0 + u) * 32 + 4 * r + g] * 17 + j;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
                    const int k = d0 + u;                          // this wave's drives (their own loads: no dynamic register index)
                    Gj[u] = ximg(Gx, k < m ? k + 1 : 0, lane);
                }
            }
#pragma unroll
            for (int u = 0; u < kMA; ++u) Ga += ak[u] * img[u];
        }
        if constexpr (!ELL) {
#pragma unroll
            for (int u = 0; u < kMU; ++u) if (d0 + u >= m) Gj[u] = zero;   // unused drive slots: their chains stay zero
        }

        // ---- ||h G||_1 = largest column sum: lane (g, i) reg kk holds G[i][4kk+g]; rows of 16 lanes share a column ------
        int sq = 0;
        {
            double best = 0.0;
            bool bad = false;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double c = fabs(h * Ga[kk]);
                c += xdpp<0x128>(c);
                c += xdpp<0x124>(c);
                c += xdpp<0x122>(c);
                c += xdpp<0x121>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = xreadlane(c, 16 * r);
                    if (!(v == v) || v > 1e300) bad = true;
                    best = fmax(best, v);
                }
            }
            if (!bad && best > kXTh) {
                int e;
                (void)frexp(best / kXTh, &e);
                sq = e;
                if (ldexp(kXTh, e - 1) >= best) sq = e - 1;
                sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
            }
        }
        const double sc = ldexp(1.0, -sq);
        const v4d Y = (h * sc) * Ga;

        // ---- Horner: R_deg+1 = I/deg!, R' = 0 ----------------------------------------------------------------------
        double fact = 1.0;
#pragma unroll
        for (int k = 2; k <= kXDeg; ++k) fact *= (double)k;        // deg!
        double ck = 1.0 / fact;                                     // 1/deg!
        v4d R = ck * IdB;
        v4d Q[kMU];
#pragma unroll
        for (int u = 0; u < kMU; ++u) Q[u] = zero;
#pragma unroll 1
        for (int k = kXDeg; k >= 1; --k) {
            ck *= (double)k;                                        // 1/(k-1)!
            if constexpr (ELL) horner_step_ell<kMU>(Y, tw, tc, scr, R, Q, ck * IdB, g, j);
            else horner_step<kMU, JAC>(Y, Gj, R, Q, ck * IdB);
        }
        // ---- squarings ------------------------------------------------------------------------------------------------
        for (int q = 0; q < sq; ++q) {
            v4d Et, Kt[kMU];
            if constexpr (JAC) {
                v4d in[kMU + 1], out[kMU + 1];
                in[0] = R;
#pragma unroll
                for (int u = 0; u < kMU; ++u) in[u + 1] = Q[u];          // unused drive slots hold zeros
                lds_transpose16_multi<kMU + 1>(scr, in, out, g, j);
                Et = out[0];
#pragma unroll
                for (int u = 0; u < kMU; ++u) Kt[u] = out[u + 1];
            } else {
                Et = lds_transpose16(scr, R, g, j);
            }
            square_step<kMU, JAC>(Et, Kt, R, Q);
        }
        // ---- outputs --------------------------------------------------------------------------------------------------
        const v4d Et = lds_transpose16(scr, R, g, j);              // E^T: A operand acting as E, and the tile to store
        const v4d EU = mm16(Et, u0);                                // [E U_t | E U_t]
        if (Fb && wv == 0) {
            const v4d dT = lds_transpose16(scr, u1 - EU, g, j);     // delta^T: lane j <-> row
#pragma unroll
            for (int r = 0; r < 2; ++r) if (4 * r + g < nc && j < nr) qc_st8m<2>(Fb + (4 * r + g) * nr + j, dT[r]);
        }
        if constexpr (JAC) {
            double* pF = Jb + P.jo_F;
            const v4d mE = -Et;
#pragma unroll
            for (int q = wv * (8 / kW); q < (wv + 1) * (8 / kW); ++q) {
                if (q < nc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * r + g < nr && j < nr) qc_st8m<2>(pF + q * nr * nr + (4 * r + g) * nr + j, mE[r]);
                }
            }
            if (wv == kW - 1) for (int i = lane; i < P.s; i += 64) Jb[P.jo_B + i] = 1.0;
            // d/da_j = -(h/2^sq) L_j U_t, transposed:  U_t^T L_j^T  (A = U_t tile, B = L_j^T)
            const double fac = -(h * sc);
            v4d Kt[kMU], XT[kMU], ua[kMU];
            lds_transpose16_multi<kMU>(scr, Q, Kt, g, j);
#pragma unroll
            for (int u = 0; u < kMU; ++u) ua[u] = u0;
            mm16_multi<kMU>(ua, Kt, XT);
#pragma unroll
            for (int u = 0; u < kMU; ++u) {
                if (d0 + u < m) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        if (4 * r + g < nc && j < nr) qc_st8m<2>(Jb + P.jo_a + (size_t)(d0 + u) * P.s + (4 * r + g) * nr + j, fac * XT[u][r]);
                }
            }
            if (ft && wv == 0) {   // d/dh = -G E U_t
                const v4d GEU = mm16(Ga, EU);
                const v4d hT = lds_transpose16(scr, -GEU, g, j);
#pragma unroll
                for (int r = 0; r < 2; ++r) if (4 * r + g < nc && j < nr) qc_st8m<2>(Jb + P.jo_h + (4 * r + g) * nr + j, hT[r]);
            }
        }
        if (wv == kW - 1) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, false);
    }
}

}  // namespace

bool qc_mfma_exp_supported(const QcParams& P) {
    return P.integrator == QC_EXPONENTIAL && P.n <= 16 && P.nc <= 8 && P.m <= kXMmax;
}

hipError_t qc_launch_mfma_exp(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    const int grid = P.n_int;
    static const bool ell_off = getenv("QC_EXP_ELL") && atoi(getenv("QC_EXP_ELL")) == 0;      // A/B diagnostics
    const bool ell = P.ell16 != nullptr && !ell_off;      // drive generators with one entry per row: the row-gather form of the Horner steps
#define QC_XJ1(MU_, W_, I_) do { const int wgs = (grid + I_ - 1) / I_; \
                            if (ell) hipLaunchKernelGGL((qc_mfma16_exp_kernel<true, MU_, W_, true, I_>), dim3(wgs), dim3(64 * W_ * I_), 0, st, P, dZ, dF, dJ); \
                            else hipLaunchKernelGGL((qc_mfma16_exp_kernel<true, MU_, W_, false, I_>), dim3(wgs), dim3(64 * W_ * I_), 0, st, P, dZ, dF, dJ); } while (0)
    // two-wave forms beyond one workgroup per CU: two intervals per (four-wave) workgroup (up to 256 intervals the pairing would only leave CUs empty)
#define QC_XJ(MU_, W_) do { if (W_ == 2 && grid > 256) QC_XJ1(MU_, W_, (W_ == 2 ? 2 : 1)); else QC_XJ1(MU_, W_, 1); } while (0)
    if (!dJ) {   // residual only: no Frechet chains, one wave
        hipLaunchKernelGGL((qc_mfma16_exp_kernel<false, 1, 1>), dim3(grid), dim3(64), 0, st, P, dZ, dF, dJ);
    } else if (P.m <= 1) {
        QC_XJ(1, 1);
    } else if (!ell && P.n_int >= 768) {   // enough intervals to put a wave on (nearly) every SIMD: one wave per interval, no redundant R chain
        // (dense drive images only.  The row-gather form keeps two waves per interval at every length: with the drive products gone the
        //  duplicated R chain costs less than what the second wave hides of the gathers' round trips -- T = 1000 / 8000: 24.6 / 150.2
        //  against 24.9 / 157.3 us)
        if (P.m <= 2) hipLaunchKernelGGL((qc_mfma16_exp_kernel<true, 2, 1>), dim3(grid), dim3(64), 0, st, P, dZ, dF, dJ);
        else if (P.m <= 4) hipLaunchKernelGGL((qc_mfma16_exp_kernel<true, 4, 1>), dim3(grid), dim3(64), 0, st, P, dZ, dF, dJ);
        else if (P.m <= 6) hipLaunchKernelGGL((qc_mfma16_exp_kernel<true, 6, 1>), dim3(grid), dim3(64), 0, st, P, dZ, dF, dJ);
        else hipLaunchKernelGGL((qc_mfma16_exp_kernel<true, 8, 1>), dim3(grid), dim3(64), 0, st, P, dZ, dF, dJ);
    } else {     // short trajectories: two waves per interval, ceil(m/2) drives each (config 2, T = 200: 18.3 -> 11.9 us)
        const int mh = (P.m + 1) / 2;
        if (mh == 1) QC_XJ(1, 2);
        else if (mh == 2) QC_XJ(2, 2);
        else if (mh == 3) QC_XJ(3, 2);
        else QC_XJ(4, 2);
    }
#undef QC_XJ
#undef QC_XJ1
    return hipGetLastError();
}
