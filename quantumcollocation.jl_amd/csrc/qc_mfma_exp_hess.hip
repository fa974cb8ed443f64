// f64-MFMA kernel for the Hessian of the Lagrangian of the EXPONENTIAL integrator at 2N <= 16 (up to 3 qubits), up to 8 drives:
//     mu^T delta,   delta = U_t+1 - exp(h G(a_t)) U_t                                     (reference README.md:79)
// The reference solves `PiccoloOptions(integrator=:exponential)` problems with the Hessian left on
// (unitary_smooth_pulse_problem.jl:224-240,242-266; `eval_hessian=false` is spelled out where it is wanted,
// unitary_robustness_problem.jl:205,247), so Ipopt asks this integrator for mu_d2F.
//
// delta is linear in U_t+1: no block touches knot t+1.  With X = h G, E = exp(X), M = reshape(mu, 2N, nc), W = M U_t^T, V = W^T,
// L(X; A) the Frechet derivative of exp and L2(X; A, B) the second one:
//     (U_t, a_j) = -L(X; h G_j)^T M                 (U_t, h) = -(G E)^T M
//     (a_i, a_j) = -<W, L2(X; h G_i, h G_j)>        (a_j, h) = -<W, G_j E + G L(X; h G_j)>          (h, h) = -<W, G G E>
//     (dx_i, h)  = -mu_i                             (derivative integrators)
// FORWARD OVER REVERSE: <W, L2(X; A, B)> = <B^T, L2(X; V, A)> (the trace under the double integral that defines L2 is invariant
// under cyclic shifts), so ONE second-order chain per drive, in the directions (V, G_i), serves the whole row (a_i, a_j), j >= i:
//     (a_i, a_j) = -h <G_j^T, L2(X; V, h G_i)>
// m second-order chains instead of m (m + 1) / 2; the A-layout image of G_j, read lane for lane against a D-layout tile, IS G_j^T.
//
// One wavefront per kMU drives, kW waves per interval, every matrix one 16 x 16 tile in registers (lane maps: qc_mfma_kernels.hip).
// Scaling and squaring as qc_mfma_exp.hip, Y = h G / 2^sq with ||Y||_1 <= 1/8, Taylor degree 10 (the second derivative of the
// truncated series loses two orders: (1/8)^9 / 9! = 2e-14) in Horner form on R_k = P_k / (k-1)!:
//     R_k   = Y R_k+1 + I/(k-1)!                  Q_k,j = G_j R_k+1 + Y Q_k+1,j            QV_k = V R_k+1 + Y QV_k+1
//     P_k,j = V Q_k+1,j + G_j QV_k+1 + Y P_k+1,j
// (3 + 5 kMU products a step) and the squarings
//     P_j <- E P_j + P_j E + LV L_j + L_j LV      L_j <- E L_j + L_j E      LV <- E LV + LV E      E <- E E
// (3 + 6 kMU products; the left factors are transposed tiles read back from a per-wave LDS scratch).  The directions are linear,
// so the factors h / 2^sq (per G_j) and 1 / 2^sq (V) are applied once to the outputs.  Blocks leave transposed (lane <-> row,
// whole lines per store): (M^T L_j), (M^T G E) with the B-layout tile of M as the A operand.
// MFMAs per interval (m = 6, one wave, 4 squarings): 10 x 132 + 4 x 156 + 56 = 2000.
#include <stdlib.h>

#include <vector>

#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kXHDeg = 10;
constexpr double kXHTh = 0.125;
constexpr int kXHMmax = 8;

__device__ inline v4d ximg(const double* __restrict__ Gx, int mat, int lane) { return load_image_tile(Gx + mat * 256, lane); }
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
// sum over the 64 lanes, the same value in every lane (fixed order: bit-reproducible)
__device__ inline double xsum64(double c) {
    c += xdpp<0x128>(c);
    c += xdpp<0x124>(c);
    c += xdpp<0x122>(c);
    c += xdpp<0x121>(c);
    return (xreadlane(c, 0) + xreadlane(c, 16)) + (xreadlane(c, 32) + xreadlane(c, 48));
}
// N such sums at once, stage by stage (no instruction waits for its predecessor); per value the additions of xsum64: the same bits
template <int N>
__device__ __forceinline__ void xsum64_multi(double (&x)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += xdpp<0x128>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += xdpp<0x124>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += xdpp<0x122>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += xdpp<0x121>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = (xreadlane(x[q], 0) + xreadlane(x[q], 16)) + (xreadlane(x[q], 32) + xreadlane(x[q], 48));
}
__device__ inline double dot4(const v4d& a, const v4d& b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }

// acc[q] (+)= A[q] * B[q] for NQ independent products, MFMAs interleaved round-robin (mm16_multi with accumulation)
template <int NQ, bool FIRST>
__device__ __forceinline__ void mma(const v4d (&a)[NQ], const v4d (&b)[NQ], v4d (&acc)[NQ]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], b[q][kk], (FIRST && kk == 0) ? z : acc[q], 0, 0, 0);
    }
}
// the same with one A operand for every product
template <int NQ, bool FIRST>
__device__ __forceinline__ void mma1(const v4d& a, const v4d (&b)[NQ], v4d (&acc)[NQ]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[q][kk], (FIRST && kk == 0) ? z : acc[q], 0, 0, 0);
    }
}
// ... and with one B operand
template <int NQ, bool FIRST>
__device__ __forceinline__ void mmb(const v4d (&a)[NQ], const v4d& b, v4d (&acc)[NQ]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], b[kk], (FIRST && kk == 0) ? z : acc[q], 0, 0, 0);
    }
}

// tile q of the wave's scratch: written in D layout (row-major, 17-double rows), read back transposed
__device__ __forceinline__ void scr_put(double* __restrict__ scr, int q, const v4d& x, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[q * 272 + (4 * r + g) * 17 + j] = x[r];
}
__device__ __forceinline__ v4d scr_get_T(const double* __restrict__ scr, int q, int g, int j) {
    v4d y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = scr[q * 272 + j * 17 + 4 * r + g];
    return y;
}
// the tile's values in vector registers HERE (an MFMA result otherwise stays in its accumulation registers until the compiler
// finds it convenient to copy it, and the next phase's accumulators take new ones)
__device__ __forceinline__ void pin_v(v4d& x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(x[r]));
}
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ELL: every drive generator has at most ONE entry per row (Pauli strings; P.ell16 = qc_exp_ell_build's tables): the two products
// with a drive image in every Horner step, G_j R and G_j QV, are row gathers from row-major LDS copies of R and QV -- 3 + 3 kMU products
// a step instead of 3 + 5 kMU.  fma(w, x, acc) per element: what the dense product adds besides exact zeros.
// (Two waves per SIMD up to three drives a wave -- 256 registers -- is what the launch's time rests on: demanded of the compiler, which
//  otherwise lands on either side of the line with any small change: 252 -> 264 registers when the tables' layout changed, 61.5 -> 71.2 us.)
// Two-wave forms put TWO intervals into one four-wave workgroup (the waves of an interval never synchronise: the pairing is placement
// only).  The dispatcher gives the first wave of a two-wave workgroup the SIMD behind the previous workgroup's FIRST wave: at 257 - 512
// intervals 243 waves shared a SIMD while as many SIMDs stayed empty -- T = 500 took 58 us, T = 1000 60.5 (profiles/r06_exp16_timeline.txt).
// A four-wave workgroup takes one wave slot on each SIMD of its CU.  (Up to 256 intervals -- one workgroup per CU or fewer -- the pairing
// would only leave CUs empty: kIPW = 1 there.  T = 500: 58.0 -> 35.6 us; T = 200, config 2: 23.1 with one interval per workgroup, 24.5 with two.)
template <int kMU, int kW, bool ELL, int kIPW = 1>
__global__ __launch_bounds__(64 * kW * kIPW) __attribute__((amdgpu_waves_per_eu(kMU <= 3 ? 2 : 1))) void qc_mfma16_exp_hess_kernel(const QcParams P, const double* __restrict__ Z, const double* __restrict__ Mu,
                                                                     double* __restrict__ H) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();
    constexpr int kTiles = 2 * kMU + 2;                       // E, LV, L_j, P_j
    static_assert(kIPW == 1 || kW == 2, "two intervals per workgroup: the two-wave forms");
    __shared__ double scr_all[kIPW * kW * kTiles * 272];
    const int lane = threadIdx.x & 63;
    const int wq = kIPW * kW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;      // wave of the workgroup
    const int wv = kW > 1 ? wq % kW : 0;                      // wave of its interval
    double* __restrict__ scr = scr_all + wq * (kTiles * 272);
    const int d0 = wv * kMU;                                  // first drive of this wave
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool ft = P.off_dt >= 0;
    const double* __restrict__ Gx = P.Gx;                    // A-layout images [mat][pair][lane][2]
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    const int b = qc_xcd_remap((int)blockIdx.x, (P.n_int + kIPW - 1) / kIPW) * kIPW + wq / kW;
    if (b >= P.n_int) return;                                 // (an odd interval count: the last workgroup's second pair has nothing to do)
#ifdef QC_XH_STAMPS       // diagnostic variant build (profiles/stamps_exp16.py): wave 0 -> slots 0-7, wave 1 -> slots 8-15
    constexpr bool DIAG = true;
    QC_STAMP_DECL;
#define XH_STAMP(k) QC_STAMP(P, b, lane, k)
    XH_STAMP(0);
#else
#define XH_STAMP(k)
#endif
    const long long t = P.t_begin + b;
    const double* __restrict__ z0 = Z + t * (long long)P.zdim;
    const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);
    const int nc = P.nc, nr = P.n;

    // ---- loads (one batch): state and multipliers in the A layout (lane (g, i) reg kk = X[i][4 kk + g], zero beyond nc columns /
    //      nr rows) and the multipliers in the B layout (lane (g, j) reg r = M[4 r + g][j]; columns >= nc re-read column 0 and are
    //      never stored), the generator images
    v4d aU, aM, bM;
    {
        const int jc = jj < nc ? jj : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * r + g;                              // A layout: column of the n x nc matrix, row j
            const bool in = c < nc && j < nr;
            const int off = in ? c * nr + j : 0;
            const double u = z0[P.off_U + off], mm_ = mu[off];
            aU[r] = in ? u : 0.0;
            aM[r] = in ? mm_ : 0.0;
            const int row = 4 * r + g;                            // B layout: row, column jc
            const bool inb = row < nr;
            const double mb = mu[inb ? jc * nr + row : 0];
            bM[r] = inb ? mb : 0.0;
        }
    }
    v4d Gj[kMU];
    double tw[kMU][4];                                        // ELL: weight and LDS offset (column x 17 + j) of rows 4 r + g of the wave's drives
    int tc[kMU][4];
    v4d Ga = ximg(Gx, 0, lane);
    {   // G = G_0 + sum over ALL drives (every wave assembles it); unconditional clamped loads, one batch
        v4d img[kXHMmax];
        double ak[kXHMmax];
#pragma unroll
        for (int u = 0; u < kXHMmax; ++u) {
            const int k = u < m ? u : (m > 0 ? m - 1 : 0);
            img[u] = ximg(Gx, m > 0 ? k + 1 : 0, lane);
            ak[u] = (u < m) ? z0[P.off_a + k] : 0.0;
        }
        if constexpr (ELL) {   // rows 4 r + g of the wave's drives (unused drive slots: weight 0, column 0 -- their chains stay zero)
            const double* __restrict__ bw = reinterpret_cast<const double*>(P.ell16);
            const int* __restrict__ bc = reinterpret_cast<const int*>(reinterpret_cast<const char*>(P.ell16) + kXHMmax * 32 * 8);
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
                const int k = d0 + u;
                Gj[u] = ximg(Gx, k < m ? k + 1 : 0, lane);
            }
        }
#pragma unroll
        for (int u = 0; u < kXHMmax; ++u) Ga += ak[u] * img[u];
    }
    if constexpr (!ELL) {
#pragma unroll
        for (int u = 0; u < kMU; ++u) if (d0 + u >= m) Gj[u] = zero;   // unused drive slots: their chains stay zero
    }

    // ---- ||h G||_1 = largest column sum -> squaring count (as qc_mfma_exp.hip) ------------------------------------------------
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
        if (!bad && best > kXHTh) {
            int e;
            (void)frexp(best / kXHTh, &e);
            sq = e;
            if (ldexp(kXHTh, e - 1) >= best) sq = e - 1;
            sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
        }
    }
    const double sc = ldexp(1.0, -sq), hs = h * sc;
    const v4d Y = hs * Ga;

    // ---- W = M U^T and V = U M^T (D layout).  Read as an A operand, the D-layout tile of W acts as W^T = V. ------------------------
    const v4d Wd = mm16(aM, aU);
    const v4d Vd = mm16(aU, aM);

    XH_STAMP(1);
    // ---- Horner: R_deg+1 = I/deg!, every derivative chain 0 -------------------------------------------------------------------
    double fact = 1.0;
#pragma unroll
    for (int k = 2; k <= kXHDeg; ++k) fact *= (double)k;
    double ck = 1.0 / fact;
    v4d R = ck * IdB, QV = zero;
    v4d Q[kMU], Pm[kMU];
#pragma unroll
    for (int u = 0; u < kMU; ++u) { Q[u] = zero; Pm[u] = zero; }
#pragma unroll 1
    for (int k = kXHDeg; k >= 1; --k) {
        ck *= (double)k;                                      // 1/(k-1)!
        if constexpr (ELL) {
            scr_put(scr, 0, R, g, j);                             // row-major copies of R_k+1 and QV_k+1 for the gathers
            scr_put(scr, 1, QV, g, j);
            lds_order();
            // Two phases, each: the gathers REQUESTED (all of them, before the products -- left to itself the compiler, at the edge of
            // the second wave per SIMD, reads one value at a time into one register pair: 24 LDS round trips a step), the products,
            // then the gathered terms added.  The phases' accumulators and gathered values share registers.
            {   // P_j <- V Q_j + Y P_j + G_j QV
                double x[kMU][4];
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = scr[272 + tc[u][r]];
                }
                __builtin_amdgcn_sched_barrier(0);
                v4d acc[kMU];
                mma1<kMU, true>(Wd, Q, acc);
                mma1<kMU, false>(Y, Pm, acc);
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Pm[u][r] = __builtin_fma(tw[u][r], x[u][r], acc[u][r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // Q_j <- Y Q_j + G_j R;   QV <- V R + Y QV;   R <- Y R + I/(k-1)!
                double x[kMU][4];
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = scr[tc[u][r]];
                }
                __builtin_amdgcn_sched_barrier(0);
                v4d a2[2] = {Wd, Y}, b2[2] = {R, R}, o2[2];
                mma<2, true>(a2, b2, o2);                         // V R, Y R
                v4d accq[kMU + 1], bq[kMU + 1];
#pragma unroll
                for (int u = 0; u < kMU; ++u) bq[u] = Q[u];
                bq[kMU] = QV;
                mma1<kMU + 1, true>(Y, bq, accq);                 // Y Q_j, Y QV
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Q[u][r] = __builtin_fma(tw[u][r], x[u][r], accq[u][r]);
                }
                QV = accq[kMU] + o2[0];
                R = o2[1] + ck * IdB;
            }
            __builtin_amdgcn_sched_barrier(0);
            lds_order();                                          // the next step's copies follow this step's gathers
        } else {
        {   // second-order chains first: they read the old Q, QV
            v4d acc[kMU];
            mma1<kMU, true>(Wd, Q, acc);                      // V Q_j
            mmb<kMU, false>(Gj, QV, acc);                     // + G_j QV
            mma1<kMU, false>(Y, Pm, acc);                     // + Y P_j
#pragma unroll
            for (int u = 0; u < kMU; ++u) { Pm[u] = acc[u]; pin_v(Pm[u]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            v4d acc[kMU + 2], bb[kMU + 2], aa[kMU + 2];
            // G_j R (kMU), V R, Y R, then + Y Q_j, + Y QV
#pragma unroll
            for (int u = 0; u < kMU; ++u) { aa[u] = Gj[u]; bb[u] = R; }
            aa[kMU] = Wd; bb[kMU] = R;
            aa[kMU + 1] = Y; bb[kMU + 1] = R;
            mma<kMU + 2, true>(aa, bb, acc);
            v4d acc2[kMU + 1], b2[kMU + 1];
#pragma unroll
            for (int u = 0; u < kMU; ++u) { acc2[u] = acc[u]; b2[u] = Q[u]; }
            acc2[kMU] = acc[kMU]; b2[kMU] = QV;
            mma1<kMU + 1, false>(Y, b2, acc2);
#pragma unroll
            for (int u = 0; u < kMU; ++u) Q[u] = acc2[u];
            QV = acc2[kMU];
            R = acc[kMU + 1] + ck * IdB;
        }
        }
    }
    XH_STAMP(2);
    // ---- squarings --------------------------------------------------------------------------------------------------------------
    for (int q = 0; q < sq; ++q) {
        scr_put(scr, 0, R, g, j);
        scr_put(scr, 1, QV, g, j);
#pragma unroll
        for (int u = 0; u < kMU; ++u) { scr_put(scr, 2 + u, Q[u], g, j); scr_put(scr, 2 + kMU + u, Pm[u], g, j); }
        lds_order();
        const v4d Et = scr_get_T(scr, 0, g, j), LVt = scr_get_T(scr, 1, g, j);
        // (three phases with scheduling fences between them: their accumulators share registers -- interleaved by the compiler they
        //  take eight tiles of accumulation registers, and with those the kernel loses its second wave per SIMD)
        {
            v4d acc[kMU], at[kMU];
            mma1<kMU, true>(Et, Pm, acc);                     // E P_j
            mma1<kMU, false>(LVt, Q, acc);                    // + LV L_j
#pragma unroll
            for (int u = 0; u < kMU; ++u) at[u] = scr_get_T(scr, 2 + kMU + u, g, j);
            mmb<kMU, false>(at, R, acc);                      // + P_j E
#pragma unroll
            for (int u = 0; u < kMU; ++u) at[u] = scr_get_T(scr, 2 + u, g, j);
            mmb<kMU, false>(at, QV, acc);                     // + L_j LV
#pragma unroll
            for (int u = 0; u < kMU; ++u) { Pm[u] = acc[u]; pin_v(Pm[u]); }
            __builtin_amdgcn_sched_barrier(0);
            // L_j <- E L_j + L_j E (at still holds L_j^T)
            v4d accl[kMU];
            mma1<kMU, true>(Et, Q, accl);
            mmb<kMU, false>(at, R, accl);
#pragma unroll
            for (int u = 0; u < kMU; ++u) { Q[u] = accl[u]; pin_v(Q[u]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            v4d a2[2] = {Et, Et}, b2[2] = {QV, R}, acc2[2];
            mma<2, true>(a2, b2, acc2);                       // E LV, E E
            v4d a1[1] = {LVt}, acc1[1] = {acc2[0]};
            mmb<1, false>(a1, R, acc1);                       // + LV E
            QV = acc1[0];
            R = acc2[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_wave_barrier();
    }
    XH_STAMP(3);
    // ---- outputs ------------------------------------------------------------------------------------------------------------------
    // (U_t, a_j) = -(h/2^sq) L_j^T M, stored transposed: M^T L_j  (A = the B-layout tile of M, acting as M^T)
    {
        v4d XT[kMU];
        mma1<kMU, true>(bM, Q, XT);
        const double fac = -hs;
#pragma unroll
        for (int u = 0; u < kMU; ++u) {
            if (d0 + u < m) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (4 * r + g < nc && j < nr) qc_st8m<2>(Hb + P.ho_Ua + (size_t)(d0 + u) * P.s + (4 * r + g) * nr + j, fac * XT[u][r]);
            }
        }
    }
    XH_STAMP(4);
    // The scalar blocks: every per-lane partial first, ONE batched reduction, one store instruction (lane q stores sum q).  One at a time
    // -- an image load, a 64-lane sum and a store per pair, up to 15 in a row on the first wave -- they were a chain of round trips at
    // the end of every wave's life, and at T = 1000 the launch IS one wave's life (profiles/r06_exp_hess.txt).
    //   (a_i, a_j) = -h (h / 4^sq) <G_j^T, P_i> for the wave's drives i and every j >= i: the A-layout image of G_j, lane for lane
    //   (a_j, h)   = -( <G_j^T, E V> + (h/2^sq) <G^T W, L_j> )                  (h, h) = -<G^T W, G E>
    const double faa = -(h * hs * sc);
#pragma unroll
    for (int half = 0; half < 2; ++half) {                    // (four images at a time: all eight cost the second wave per SIMD in registers)
        constexpr int kNA = 4 * kMU;
        double pv[kNA];
        {
            v4d img[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) img[q] = ximg(Gx, 4 * half + q < m ? 4 * half + q + 1 : 0, lane);
#pragma unroll
            for (int u = 0; u < kMU; ++u) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[4 * u + q] = dot4(img[q], Pm[u]);
            }
        }
        xsum64_multi<kNA>(pv);
        double mine = 0.0;
#pragma unroll
        for (int q = 0; q < kNA; ++q) mine = lane == q ? pv[q] : mine;
        const int u = lane >> 2, jd = 4 * half + (lane & 3), i = d0 + u;
        if (lane < kNA && i < m && jd >= i && jd < m) Hb[P.ho_aa + jd * (jd + 1) / 2 + i] = faa * mine;
        __builtin_amdgcn_sched_barrier(0);                    // (the second half's images are not requested next to the first's)
        if (m <= 4) break;
    }
    if (ft) {
        // T2 = G^T W (A = the D-layout tile of G, acting as G^T), E V (A = E^T), G E
        const v4d Gd = lds_transpose16(scr, Ga, g, j);        // Ga read as a D-layout tile is G^T: its transpose is D-layout(G)
        const v4d Et = lds_transpose16(scr, R, g, j);
        v4d a3[3] = {Gd, Et, Ga}, b3[3] = {Wd, Vd, R}, o3[3];
        mma<3, true>(a3, b3, o3);
        const v4d T2 = o3[0], EV = o3[1], GE = o3[2];
        if constexpr (ELL) {
#pragma unroll
            for (int u = 0; u < kMU; ++u) Gj[u] = ximg(Gx, d0 + u < m ? d0 + u + 1 : 0, lane);
        }
        double pv[kMU + 1];
#pragma unroll
        for (int u = 0; u < kMU; ++u) pv[u] = dot4(Gj[u], EV) + hs * dot4(T2, Q[u]);
        pv[kMU] = dot4(T2, GE);
        if (wv == 0) {
            // (U_t, h) = -(G E)^T M, transposed: M^T (G E)
            const v4d XT = mm16(bM, GE);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (4 * r + g < nc && j < nr) qc_st8m<2>(Hb + P.ho_Uh + (4 * r + g) * nr + j, -XT[r]);
        }
        xsum64_multi<kMU + 1>(pv);
        double mine = 0.0;
#pragma unroll
        for (int q = 0; q <= kMU; ++q) mine = lane == q ? pv[q] : mine;
        if (lane < kMU) {
            if (d0 + lane < m) Hb[P.ho_ah + d0 + lane] = -mine;
        } else if (lane == kMU && wv == 0) {
            Hb[P.ho_hh] = -mine;
        }
    }
    if (wv == kW - 1) qc_hess_tail(P, mu, Hb, lane, 64);
#ifdef QC_XH_STAMPS
    XH_STAMP(5);
    __builtin_amdgcn_s_waitcnt(0);
    XH_STAMP(6);
    if (P.stamps != nullptr && lane == 0 && wv < 2) {
#pragma unroll
        for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + 8 * wv + k_] = qc_ts_[k_];
    }
#endif
}

}  // namespace

bool qc_mfma_exp_hess_supported(const QcParams& P) {
    return P.integrator == QC_EXPONENTIAL && P.n <= 16 && P.nc <= 8 && P.m <= kXHMmax && P.hess_nnz > 0 && P.Gx != nullptr;
}

// Rows of the drive generators of an exponential-integrator handle at 2N <= 32 whose drives have at most ONE entry per row:
// blob = [8][32] weights (doubles), [8][32] columns (ints); unused rows and drive slots: weight 0, column 0.  Read by the four MFMA
// kernels of the exponential integrator (qc_mfma_exp*.hip, qc_mfma32_exp*.hip).  (P.ell16 of a Pade handle is qc_mfma16_ell_build's
// table; the two never meet: the integrator decides.)
bool qc_exp_ell_build(const QcParams& P, const double* G, std::vector<char>* blob) {
    if (P.integrator != QC_EXPONENTIAL || P.n > 32 || P.m < 1 || P.m > kXHMmax) return false;
    const int n = P.n, m = P.m;
    blob->assign(kXHMmax * 32 * 8 + kXHMmax * 32 * 4, 0);
    double* tw = reinterpret_cast<double*>(blob->data());
    int* tc = reinterpret_cast<int*>(blob->data() + kXHMmax * 32 * 8);
    for (int k = 0; k < m; ++k)
        for (int a = 0; a < n; ++a) {
            int cnt = 0;
            for (int c = 0; c < n; ++c) {
                const double v = G[(size_t)(k + 1) * n * n + (size_t)c * n + a];      // drive k, row a, column c (column-major)
                if (v == 0.0) continue;
                if (++cnt > 1) return false;
                tw[k * 32 + a] = v;
                tc[k * 32 + a] = c;
            }
        }
    return true;
}

hipError_t qc_launch_mfma_exp_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    const int grid = P.n_int;
    static const bool ell_off = getenv("QC_EXP_ELL") && atoi(getenv("QC_EXP_ELL")) == 0;      // A/B diagnostics
    const bool ell = P.ell16 != nullptr && !ell_off;
#define QC_XH1(MU_, W_, I_) do { const int wgs = (grid + I_ - 1) / I_; \
                            if (ell) hipLaunchKernelGGL((qc_mfma16_exp_hess_kernel<MU_, W_, true, I_>), dim3(wgs), dim3(64 * W_ * I_), 0, st, P, dZ, dMu, dH); \
                            else hipLaunchKernelGGL((qc_mfma16_exp_hess_kernel<MU_, W_, false, I_>), dim3(wgs), dim3(64 * W_ * I_), 0, st, P, dZ, dMu, dH); } while (0)
    // two-wave forms beyond one workgroup per CU: two intervals per (four-wave) workgroup -- see the kernel
#define QC_XH(MU_, W_) do { if (W_ == 2 && grid > 256) QC_XH1(MU_, W_, (W_ == 2 ? 2 : 1)); else QC_XH1(MU_, W_, 1); } while (0)
    // Measured at config 3 (T = 1000, m = 6; profiles/r06_exp_hess.txt): two waves of three drives 70.6 us = 0.81 of the f64 MFMA peak
    // counting the shared chains both waves run (0.74 counting them once), one wave of six drives 70.9 us (2000 MFMAs per interval,
    // 0.74 of peak, 300 registers), three waves of two 93.9 us (the shared chains three times): the launch is bound by the matrix pipes.
    if (P.m <= 1) QC_XH(1, 1);
    else if (P.m <= 2) QC_XH(2, 1);
    else if (P.m <= 3) QC_XH(3, 1);
    else if (P.m <= 4) QC_XH(2, 2);
    else if (P.m <= 6) QC_XH(3, 2);
    else QC_XH(4, 2);
#undef QC_XH
#undef QC_XH1
    return hipGetLastError();
}
