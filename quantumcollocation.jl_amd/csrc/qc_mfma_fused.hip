// F, dF and mu_d2F of every interval in ONE launch: order-4 Pade, 2N = 16 (a unitary on 8 levels: BASELINE configs 3 and 4), up to 6
// drives, exactly antisymmetric generators (every Hermitian Hamiltonian).  Ipopt asks for the constraint Jacobian and the Hessian of
// the Lagrangian at the same accepted point (reference call sites test/scripts/integrator_test_1qubit.jl:46,52; SURVEY.md 7, "fuse
// residual + Jacobian (+ Hessian) into one launch"); as two launches (qc_mfma_kernels.hip, qc_mfma_hess.hip) the second one pays a
// second launch floor, re-reads the knots, re-assembles G, and -- one wave per SIMD, a serial chain of loads, two MFMA stages,
// transposes and reductions -- runs at a fifth of the HBM rate with nothing to overlap.  Here it rides on the F + dF kernel, which is
// bound by its 42.6 MB of stores and leaves the matrix pipes idle:
//
//   wave 1 ("copy wave")     the copy wave of qc_mfma16_pade4_kernel -- every global load of the interval in one batch, G, (G^2)^T,
//                            B^T / F^T, the 2N tile copies, the derivative-integrator rows -- plus the interval's multipliers (one
//                            more tile request in the same batch), and the parts of the Hessian it can do from what it holds in
//                            registers while its stores drain: stage A (Y = G [M | c2 h^2 D], T_k = G_k [M | c2 h^2 D]) between
//                            its tile copies, the (a, a) sums from the T_k, the derivative-integrator entries and the padding
//   wave 0 ("compute wave")  the compute wave of qc_mfma16_pade4_kernel (residual, d/dh, the drive columns), then -- G, the knots'
//                            tiles, the images in LDS, [-N_k | -N_k+1] and Y handed over by the copy wave: no global load at
//                            all -- stage B, the LDS transposes, the matrix blocks' stores and the (a, h) / (h, h) sums of
//                            qc_mfma16_pade4_hess_anti_kernel, operation for operation
// Measured at config 3 (T = 1000; profiles/r03_fused_variants.txt, r03_fused_timeline.txt): two launches 16.9 us; the plain
// composition 15.5 us (the compute wave at its 256-register budget spills 9 registers, and a scratch reload behind the wave's own
// stores waits for them); S / D of the scalar blocks re-read from LDS instead of held through the MFMA stages (244 registers, no
// scratch) 14.5 us; the (a, a) sums -- three quarters of the scalar blocks -- handed to the copy wave, which has issued its stores by
// then, 13.85 us; no tile copy before the hand-off (the F + dF kernel stores four there: its compute wave has slack, this one is the
// critical path) 13.5 us; stage A of the Hessian on the copy wave, between its tile copies, 13.1 us = 59.6 MB at 4.5 TB/s.  Parking the compute wave's F + dF tiles in LDS until the Hessian's MFMA stages were
// through (its stores then leave after the copy waves' flood): slower, 15.3 us.
// Two waves per interval as before, so T = 1000 still fits the device in one round (a third wave per interval would not: 3 x 999
// waves > 2048 slots at this register budget); the arithmetic of both halves is the arithmetic of the two kernels, so the values are
// bit-identical to two launches (tests/test_gpu_parity.py::test_fused_launch_is_bit_identical).
// LDS per workgroup (doubles): G 256 | U_t 256 | U_t+1 256 | M 256 | images kMU x 256 (rewritten with the stage-A tiles T_k once the
// images are in registers) | scratch (kMU + 1) x 272 (transposes of both halves; the reduction rows alias it) = 35.7 KB at kMU = 6:
// four workgroups per CU, as the F + dF kernel has.  (kMU = 8 would need 44 KB: seven and eight drives take two launches.)
// ELL (round 5): drive generators with ONE entry per row -- Pauli strings, what configs 3 and 4 drive with.  Every product with a drive
// image, G_k X, is then a row gather from a row-major LDS copy of X: T_k[a][j] = w_k[a] X[c_k[a]][j].  v_mfma_f64_16x16x4_f64 is an
// ascending-k chain of fused multiply-adds (tests/hip/mfma_f64_fma_probe.hip), so with the other fifteen entries of the row exact zeros
// the dense product's accumulator holds fma(w, x, acc) and nothing else: the gathered form has the SAME BITS as the dense-image kernels
// (finite inputs), and the one call stays bit-identical to the two launches.  120 f64 MFMAs per interval (64 cycles each on the SIMD's
// one matrix pipe: 3.6 us of pipe time per SIMD at T = 1000, in front of and between the stores of a store-bound kernel) become 48; the
// copy wave's stage A no longer paces its tile copies; the compute wave holds no image in a register.  Used for trajectories of more
// than one and up to four device rounds (qc_mfma16_fused_gathers below): within one round the launch follows its store stream and the images
// are faster; beyond four rounds the box decides.
#include <stdlib.h>

#include <vector>

#include "qc_mfma_hess_common.h"

namespace {

using namespace qc_mfma;

constexpr int kFuThreads = 128;
// One interval per workgroup, the whole trajectory in one launch.  (Rounds 3's first form sent longer trajectories out in launches of
// 1024 workgroups -- one round of the device --: T = 2000 / 4000 / 8000 took 26.5 / 53.7 / 108.7 us against 25.9 / 46.1 / 84.1 us in one
// launch, the hardware refilling each CU as its workgroups retire.)
constexpr int kFuMaxGrid = 1 << 24;
constexpr int kFuMaxStamped = 1024;          // the time-stamped build (QC_STAMPS=1) records one round
constexpr int kFuDF = 4;                     // derivative integrators served from registers (F + dF part)
constexpr int kLdsGa = 0, kLdsU0 = 256, kLdsU1 = 512, kLdsM = 768, kLdsGk = 1024;
// Leading arguments = what the first load requests depend on; preloaded into scalar registers at wave launch (Makefile:
// -amdgpu-kernarg-preload-count, as for the two kernels this one is made of).  Zt / mu0: the first knot / the first interval's
// multipliers of THIS launch.
template <int kMU, int VAR, bool ELL>
__global__ __launch_bounds__(kFuThreads, 2) void qc_mfma16_pade4_fused_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                             const double* __restrict__ hot_mu0, int hot_n_int, int hot_zdim,
                                                                             int hot_off_a, int hot_off_dt, int hot_m, int hot_off_U,
                                                                             int hot_f_stride, const QcParams P, double* __restrict__ F,
                                                                             double* __restrict__ J, double* __restrict__ H) {
    using R = FuRows<kMU>;
    constexpr int kLdsScr = kLdsGk + kMU * 256;
    constexpr bool LATE = (VAR & 1) != 0;           // S / D for the scalar blocks re-read from LDS at the end instead of held through the MFMA stages
    constexpr bool DIAG = (VAR & 8) != 0;           // time stamps (QC_STAMPS=1 handles; profiles/stamps_fused.py), never in a timed run
    QC_STAMP_DECL;
    QC_STAMP(P, 0, 0, 0);                           // wave entry (both roles: slots 0.. compute wave, 10.. copy wave below)
    constexpr bool AACOPY = (VAR & 4) != 0;         // the (a, a) sums -- three quarters of the scalar blocks -- on the copy wave, which has nothing left to do by then
    // transposes: the F + dF part needs kMU / 2 + 1 tiles at once, the Hessian part kMU + 1 -- in two rounds of at most 4 when the copy
    // wave's reduction rows need the space
    constexpr int kTr1 = AACOPY ? (kMU + 1 < 4 ? kMU + 1 : 4) : kMU + 1, kTr2 = kMU + 1 - kTr1;
    constexpr int kScrTiles = kTr1 > kMU / 2 + 1 ? kTr1 : kMU / 2 + 1;
    constexpr int kLdsRedC = kLdsScr + kScrTiles * 272;                     // the copy wave's reduction rows ((a, a) sums)
    constexpr bool SACOPY = (VAR & 32) != 0;        // stage A of the Hessian on the copy wave too, between its tile copies (needs AACOPY)
    static_assert(!SACOPY || AACOPY, "stage A on the copy wave goes with the (a, a) sums there");
    static_assert(!ELL || SACOPY, "the row-gather form is built on the final variant");
    // (SACOPY: the region also carries the tiles [-N_k | -N_k+1] and Y from the copy wave to the compute wave)
    constexpr int kRedCLen = !AACOPY ? 0 : (SACOPY && (kMU / 2 + 1) * 256 > R::kAA * kFuStride ? (kMU / 2 + 1) * 256 : R::kAA * kFuStride);
    constexpr int kLdsFlag = kLdsRedC + kRedCLen;                           // one word: "the compute wave has taken the hand-over tiles"
    constexpr int kLdsScal = kLdsFlag + 2;                                  // ([0] that flag, [1] fu_scalar_run_store's counter;) the staged scalar run
    constexpr int kLdsTotal = kLdsScal + kFuScalMax;
    static_assert(kLdsTotal * 8 <= 40960, "four workgroups per CU");
    static_assert((AACOPY ? R::kPair + 1 : R::kRows) * kFuStride <= kScrTiles * 272, "the compute wave's reduction rows alias the transpose scratch");
    __shared__ __attribute__((aligned(16))) double sm[kLdsTotal];
    QcKernargTouch<sizeof(QcParams) + 128> touch;
    touch.request();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = 1 - wave;                      // 0 compute wave (the workgroup's first), 1 copy wave
    const int m = hot_m;
    const bool ft = hot_off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gx = hot_Gx;

    // the copy wave's generator images depend on nothing but the kernel arguments
    v4d g0_img, gk_img[kMU];
    double ell_w[2] = {0.0, 0.0};                   // ELL: the drives' rows (entries lane and 64 + lane of [drive][16]), to LDS with the hand-off
    int ell_c[2] = {0, 0};
    if (role == 1) {
        g0_img = fu_load_GA(Gx, 0, lane);
#pragma unroll
        for (int u = 0; u < kMU; ++u) {
            const int k = u < m ? u : (m > 0 ? m - 1 : 0);
            gk_img[u] = fu_load_GA(Gx, m > 0 ? k + 1 : 0, lane);
        }
        if constexpr (ELL) {
            typedef const __attribute__((address_space(1))) double* gdp;
            typedef const __attribute__((address_space(1))) int* gip;
            const gdp tw = (gdp)(unsigned long long)P.ell16;
            const gip tc = (gip)(unsigned long long)((const char*)P.ell16 + 6 * 16 * 8);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 64 * q + lane;
                ell_w[q] = tw[e < 16 * m ? e : 0];
                ell_c[q] = tc[e < 16 * m ? e : 0];
            }
        }
    }
    double* __restrict__ const ellTW = sm + kLdsGk;                         // ELL: the image region holds the drives' rows instead
    int* __restrict__ const ellTC = reinterpret_cast<int*>(sm + kLdsGk + 96);
    touch.consume();
    if ((int)blockIdx.x >= hot_n_int) return;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const v4d IdB = identity_B(g, j);
    const int b = qc_xcd_remap(blockIdx.x, hot_n_int);
    const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
    const double* __restrict__ z1 = z0 + hot_zdim;
    const double* __restrict__ mu = hot_mu0 + (long long)b * hot_f_stride;
    const double av_pre = role == 1 ? load_amp_lanes(z0, hot_off_a, m, lane) : 0.0;
    const double h = ft ? load_uniform(z0 + hot_off_dt) : opaque_scalar(P.dt_fixed);
    double* __restrict__ Jb = J + (size_t)b * P.J_stride + P.J_off;
    double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double hc1 = h * c1, hc2 = h * h * c2;
    // the Hessian block's scalar entries leave in one piece (fu_scalar_run_store) where the copy wave's register path serves the
    // derivative integrators' entries; otherwise one by one, as qc_mfma16_pade4_hess_anti_kernel stores them
    const bool hfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
    const bool staged = hfast && P.scal_run && fu_scalar_run_len(P) <= kFuScalMax;      // (scal_run: the scalar kinds close the block in the default order)
    double* __restrict__ scal = sm + kLdsScal;
    int* __restrict__ scal_count = reinterpret_cast<int*>(sm + kLdsFlag) + 1;

    if (role == 1) {
        // ================= copy wave (qc_mfma16_pade4_kernel's, plus the multipliers) ======================================
        __builtin_amdgcn_s_setprio(3);
        if constexpr (DIAG) qc_ts_[10] = qc_ts_[0];
        const v4d u0 = fu_load_col(z0 + hot_off_U, jj, g);
        const v4d u1 = fu_load_col(z1 + hot_off_U, jj, g);
        const v4d mt = fu_load_col(mu, jj, g);                  // M = reshape(mu_t[0:s], 16, 8): both lane halves hold the same 8 columns
        double dxv[kFuDF], dfv[kFuDF];
        const bool dfast = P.n_deriv <= kFuDF;
#pragma unroll
        for (int d = 0; d < kFuDF; ++d) {
            const int i = lane < P.ddim_i[d] ? lane : 0;
            dxv[d] = z0[P.dx_off[d] + i];
            dfv[d] = z1[P.x_off[d] + i] - z0[P.x_off[d] + i];
        }
        // multipliers of the derivative integrators' rows (the Hessian's closing entries -mu_i), with the batch as well
        double mud[2] = {0.0, 0.0};
        if (hfast) {
#pragma unroll
            for (int d = 0; d < 2; ++d) mud[d] = mu[P.drow[d] + (lane < P.ddim_i[d] ? lane : 0)];
        }
        v4d Ga = g0_img;
#pragma unroll
        for (int u = 0; u < kMU; ++u) {
            const double a = (u < m) ? bcast_lane(av_pre, u) : 0.0;
            Ga += a * gk_img[u];
        }
        QC_STAMP(P, b, lane, 11);                   // copy wave: G assembled (its loads are back)
        const v4d Gb = mm16(Ga, IdB);
        const v4d G2T = mm16(Gb, Ga);
        v4d Fm, Bm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double ev = IdB[r] + hc2 * G2T[r];
            Fm[r] = -(ev + hc1 * Ga[r]);       // -F^T
            Bm[r] = ev - hc1 * Ga[r];          //  B^T
        }
        // copies stored before the hand-off: 4 in the F + dF kernel, where nothing waits for the compute wave; here the compute wave's
        // chain is twice as long and is what the launch waits for (profiles/r03_fused_timeline.txt)
        constexpr int kEarly = (VAR & 16) ? 0 : 4;     // (1 and 2 copies: 13.6 / 13.9 us against 13.5 with none, 13.85 with four)
#pragma unroll
        for (int q = 0; q < kEarly; ++q) {
            if (q < P.copies) {
                fu_store_tile_T(Jb + P.jo_F + q * 256, Fm, g, j);
                fu_store_tile_T(Jb + P.jo_B + q * 256, Bm, g, j);
            }
        }
        fu_lds_put(sm + kLdsGa, lane, Ga);
        fu_lds_put(sm + kLdsU0, lane, u0);
        fu_lds_put(sm + kLdsU1, lane, u1);
        fu_lds_put(sm + kLdsM, lane, mt);
        if constexpr (ELL) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 64 * q + lane;
                if (e < 16 * kMU) {
                    ellTW[e] = e < 16 * m ? ell_w[q] : 0.0;
                    ellTC[e] = e < 16 * m ? ell_c[q] : 0;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < kMU; ++u)
                if (u < m) fu_lds_put(sm + kLdsGk + u * 256, lane, gk_img[u]);
        }
        if (lane == 0) {
            if constexpr (SACOPY) reinterpret_cast<int*>(sm + kLdsFlag)[0] = 0;
            scal_count[0] = 0;
        }
        __syncthreads();
        v4d Tq[kMU];                               // SACOPY: the stage-A tiles T_k = G_k [M | c2 h^2 D], kept for the (a, a) sums
        if constexpr (!SACOPY) {
            double* pF = Jb + P.jo_F;
            double* pB = Jb + P.jo_B;
            const int ncop = P.copies;
            for (int q = kEarly; q < ncop; ++q) {
                fu_store_tile_T(pF + q * 256, Fm, g, j);
                fu_store_tile_T(pB + q * 256, Bm, g, j);
            }
        } else {
            // Stage A of the Hessian -- Y = G [M | c2 h^2 D] and T_k = G_k [M | c2 h^2 D]: this wave holds G, the images, M and the
            // knots' tiles in registers -- rides between the tile copies: the wave issues a store every ~100 cycles because the CU's
            // store queue is full, and the seven independent accumulator chains (mm16_multi's order, so the same bits) run in the
            // shadow of those stalls.  The compute wave's chain is shorter by these 28 products.
            static_assert(kEarly == 0, "all copies behind the hand-off");
            v4d MD;
#pragma unroll
            for (int r = 0; r < 4; ++r) MD[r] = left ? mt[r] : hc2 * (u1[r] - u0[r]);
            const v4d zero = {0.0, 0.0, 0.0, 0.0};
            v4d acc[kMU + 1];
            double* pF = Jb + P.jo_F;
            double* pB = Jb + P.jo_B;
            const int ncop = P.copies;
            if constexpr (ELL) {
                // T_k by row gathers from a row-major copy of [M | c2 h^2 D] (in the hand-over region, which this wave fills afterwards):
                // LDS reads, which issue beside the stores and cost the matrix pipe nothing
                double* __restrict__ xs = sm + kLdsRedC;
                fu_put_rows(xs, MD, g, j);
                fu_lds_order();
                v4d Tg[kMU];
                fu_gather_all<kMU>(ellTW, ellTC, xs, g, j, Tg);
#pragma unroll
                for (int u = 0; u < kMU; ++u) acc[1 + u] = Tg[u];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], MD[kk], kk ? acc[0] : zero, 0, 0, 0);
                if constexpr (!ELL) {
#pragma unroll
                    for (int u = 0; u < kMU; ++u) acc[1 + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(gk_img[u][kk], MD[kk], kk ? acc[1 + u] : zero, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 2 * kk; q < 2 * kk + 2; ++q) {
                    if (q < ncop) {
                        fu_store_tile_T(pF + q * 256, Fm, g, j);
                        fu_store_tile_T(pB + q * 256, Bm, g, j);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int q = 8; q < ncop; ++q) {       // (never: copies == N = 8 here)
                fu_store_tile_T(pF + q * 256, Fm, g, j);
                fu_store_tile_T(pB + q * 256, Bm, g, j);
            }
#pragma unroll
            for (int u = 0; u < kMU; ++u) Tq[u] = acc[1 + u];
            // hand-over to the compute wave's stage B: [-N_k | -N_k+1] per drive pair, and Y
            if constexpr (ELL) fu_lds_order();     // (the gathers above have read the region)
            double* __restrict__ hand = sm + kLdsRedC;
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) fu_lds_put(hand + 256 * p2, lane, fu_sel(left, Tq[2 * p2], swap8(Tq[2 * p2 + 1])));
            fu_lds_put(hand + 256 * (kMU / 2), lane, acc[0]);
            __syncthreads();                       // second barrier: the compute wave is through its F + dF part by now
        }
        __builtin_amdgcn_s_setprio(0);
        QC_STAMP(P, b, lane, 12);                   // copy wave: the 2N tile copies issued
        {   // derivative integrator rows of F and dF
            int jo = P.jo_d;
            bool all_fast = dfast;
#pragma unroll
            for (int d = 0; d < kFuDF; ++d) {
                if (d < P.n_deriv) {
                    const int dim = P.ddim_i[d], r0 = P.drow[d];
                    if (dfast && dim <= 64) {
                        if (lane < dim) {
                            if (Fb) Fb[r0 + lane] = dfv[d] - h * dxv[d];
                            Jb[jo + lane] = -1.0;
                            Jb[jo + dim + lane] = 1.0;
                            Jb[jo + 2 * dim + lane] = -h;
                            if (ft) Jb[jo + 3 * dim + lane] = -dxv[d];
                        }
                    } else {
                        all_fast = false;
                    }
                    jo += (ft ? 4 : 3) * dim;
                }
            }
            if (!all_fast) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, dfast);
        }
        // the Hessian block's derivative-integrator entries d2/d(dx_i) dh = -mu_i and its alignment padding
        if (hfast) {
            // (one run with the other scalar entries when they are staged: see fu_scalar_run_store)
            int o = P.ho_d;
            if (staged) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    if (lane < P.ddim_i[d]) scal[o - P.ho_aa + lane] = -mud[d];
                    o += P.ddim_i[d];
                }
                for (int i = lane; i < P.h_pad; i += 64) scal[P.hess_nnz - P.ho_aa + i] = 0.0;
            } else {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    if (lane < P.ddim_i[d]) Hb[o + lane] = -mud[d];
                    o += P.ddim_i[d];
                }
                for (int i = lane; i < P.h_pad; i += 64) Hb[P.hess_nnz + i] = 0.0;
            }
        } else {
            qc_hess_tail(P, mu, Hb, lane, 64);
        }
        QC_STAMP(P, b, lane, 13);                   // copy wave: every store of its own issued
        if constexpr (AACOPY && SACOPY) {
            // (a_u, a_v) = -sum T_u . swap8(T_v) over all lanes, from this wave's own registers; the rows go where the hand-over
            // tiles were, once the compute wave says it has them
            double* __restrict__ redc = sm + kLdsRedC;
            double aa[R::kAA];
#pragma unroll
            for (int v = 0; v < kMU; ++v) {
                const v4d Tsw = swap8(Tq[v]);
#pragma unroll
                for (int u = 0; u <= v; ++u) {
                    // dot4(-T_u, swap8(T_v)) with the roundings qc_mfma16_pade4_hess_anti_kernel's compilation has (first product
                    // rounded on its own, the other three fused in order): written out, because the compiler's choice of WHICH
                    // product stays un-fused depends on the shape of the surrounding code, and the values must be the same bits
                    const double t0 = Tq[u][0] * Tsw[0];
                    double r = __builtin_fma(-Tq[u][1], Tsw[1], -t0);
                    r = __builtin_fma(-Tq[u][2], Tsw[2], r);
                    aa[v * (v + 1) / 2 + u] = __builtin_fma(-Tq[u][3], Tsw[3], r);
                }
            }
            QC_STAMP(P, b, lane, 14);
            while (__hip_atomic_load(reinterpret_cast<int*>(sm + kLdsFlag), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int i = 0; i < R::kAA; ++i) redc[i * kFuStride + lane] = aa[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            fu_reduce_rows<kMU>(P, redc, Hb, lane, m, ft, 0, R::kAA, 0, staged, scal);
        } else if constexpr (AACOPY) {
            // (a_u, a_v) = -sum T_u . swap8(T_v) over all lanes, from the stage-A tiles the compute wave has parked in LDS: this wave's
            // stores are issued and it would only wait for them to drain
            __syncthreads();
            QC_STAMP(P, b, lane, 14);               // copy wave: second barrier passed (stage-A tiles there)
            const double* __restrict__ tsave = sm + kLdsGk;
            double* __restrict__ redc = sm + kLdsRedC;
            v4d Tn[kMU];
#pragma unroll
            for (int u = 0; u < kMU; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Tn[u][r] = -tsave[(u * 4 + r) * 64 + lane];
            }
#pragma unroll
            for (int v = 0; v < kMU; ++v) {
                v4d Tsw;
#pragma unroll
                for (int r = 0; r < 4; ++r) Tsw[r] = tsave[(v * 4 + r) * 64 + (lane ^ 8)];
#pragma unroll
                for (int u = 0; u <= v; ++u) redc[(v * (v + 1) / 2 + u) * kFuStride + lane] = fu_dot4(Tn[u], Tsw);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            fu_reduce_rows<kMU>(P, redc, Hb, lane, m, ft, 0, R::kAA, 0, staged, scal);
        }
        if (staged) fu_scalar_run_store(P, scal, scal_count, Hb, lane);
        QC_STAMP(P, b, lane, 15);                   // copy wave: done
        QC_STAMP_FLUSH(P, b, lane, 10, 15);
        return;
    }

    // ===================== compute wave: F + dF ======================================================================
    __builtin_amdgcn_s_setprio(1);
    __syncthreads();                              // the copy wave's hand-off
    QC_STAMP(P, b, lane, 1);                      // compute wave: hand-off received
    double* __restrict__ scr = sm + kLdsScr;
    v4d Ga = fu_lds_get(sm + kLdsGa, lane);
    v4d u0 = fu_lds_get(sm + kLdsU0, lane);
    v4d u1 = fu_lds_get(sm + kLdsU1, lane);
    auto store_jac_tiles = [&](const v4d& ET, const v4d (&YT)[kMU / 2]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * r + g;              // tile column: < 8 residual column c, >= 8 d/dh column c-8
            if (c < 8) {
                if (Fb) qc_st8m<2>(Fb + c * 16 + j, ET[r]);
            } else if (ft) {
                qc_st8m<2>(Jb + P.jo_h + (c - 8) * 16 + j, ET[r]);
            }
        }
        double* pa = Jb + P.jo_a;
#pragma unroll
        for (int u = 0; u < kMU; u += 2) {
            if (u < m) {
                const bool two = u + 1 < m;
                double* p = pa + (size_t)u * 128;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * r + g;      // tile columns < 8: drive u column c; >= 8: drive u+1 column c-8
                    if (r < 2) qc_st8m<2>(p + c * 16 + j, YT[u >> 1][r]);
                    else if (two) qc_st8m<2>(p + 128 + (c - 8) * 16 + j, YT[u >> 1][r]);
                }
            }
        }
    };
    {
        v4d W, Wsw;                               // W = [S | D], Wsw = [D | S]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double sm_ = u1[r] + u0[r], df = u1[r] - u0[r];
            W[r] = left ? sm_ : df;
            Wsw[r] = left ? df : sm_;
        }
        const v4d P1 = mm16(Ga, W);               // [GS | GD]
        const v4d P1sw = swap8(P1);               // [GD | GS]
        v4d Q;                                    // [Q0 | Q1] = [-h c1 S + h^2 c2 GD | h^2 c2 D]
#pragma unroll
        for (int r = 0; r < 4; ++r) Q[r] = left ? (-hc1 * W[r] + hc2 * P1sw[r]) : (hc2 * W[r]);
        constexpr int NB = kMU + 1;
        v4d sB[NB];                               // P2 = G P1sw, R_k = G_k Q
        if constexpr (ELL) {
            // R_k by row gathers from a row-major copy of Q (transpose scratch), beside the one dense chain G P1sw
            fu_put_rows(scr, Q, g, j);
            fu_lds_order();
            const v4d zero = {0.0, 0.0, 0.0, 0.0};
            v4d p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[0], P1sw[0], zero, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], P1sw[kk], p2, 0, 0, 0);
            v4d Rg[kMU];
            fu_gather_all<kMU>(ellTW, ellTC, scr, g, j, Rg);
            sB[0] = p2;
#pragma unroll
            for (int u = 0; u < kMU; ++u) sB[u + 1] = Rg[u];
            fu_lds_order();                       // (the transposes below rewrite the scratch)
        } else {
            v4d aB[NB], bB[NB];
            aB[0] = Ga;
            bB[0] = P1sw;
#pragma unroll
            for (int u = 0; u < kMU; ++u) {       // an unused slot (u >= m) repeats the last drive; never stored
                aB[u + 1] = fu_lds_get(sm + kLdsGk + (u < m ? u : m - 1) * 256, lane);
                bB[u + 1] = Q;
            }
            mm16_multi<NB>(aB, bB, sB);
        }
        const v4d P2 = sB[0];                     // [G^2 D | G^2 S]
        v4d E;                                    // [delta | d/dh]
        {
            v4d dl, dh;
            const double d1 = -c1, d2 = 2.0 * c2 * h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dl[r] = Wsw[r] - hc1 * P1[r] + hc2 * P2[r];
                dh[r] = d1 * P1[r] + d2 * P2[r];
            }
            const v4d dhs = swap8(dh);
#pragma unroll
            for (int r = 0; r < 4; ++r) E[r] = left ? dl[r] : dhs[r];
        }
        constexpr int NC = kMU / 2 + 1;
        v4d sC[NC], Y[NC];
        {
            v4d aT[kMU / 2], bT[kMU / 2], dT[kMU / 2];
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) {
                const v4d R1 = sB[2 * p2 + 1], R2 = sB[2 * p2 + 2];
                const v4d R1sw = swap8(R1), R2sw = swap8(R2);
                aT[p2] = Ga;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bT[p2][r] = left ? R1sw[r] : R2[r];          // [G_k Q1 | G_k+1 Q1]
                    Y[p2 + 1][r] = left ? R1[r] : R2sw[r];       // [G_k Q0 | G_k+1 Q0]
                }
            }
            mm16_multi<kMU / 2>(aT, bT, dT);
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) sC[p2 + 1] = dT[p2];
        }
        v4d ET, YT[kMU / 2];
        {
            v4d tin[kMU / 2 + 1], tout[kMU / 2 + 1];
            tin[0] = E;
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) tin[p2 + 1] = Y[p2 + 1] + sC[p2 + 1];
            lds_transpose16_multi<kMU / 2 + 1>(scr, tin, tout, g, j);      // (the scratch region: the hand-off block is read again below)
            ET = tout[0];
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) YT[p2] = tout[p2 + 1];
        }
        QC_STAMP(P, b, lane, 2);                  // compute wave: F + dF products and transposes through
        store_jac_tiles(ET, YT);
    }

    QC_STAMP(P, b, lane, 3);                      // compute wave: F + dF stores issued
    // ===================== compute wave: mu_d2F (qc_mfma16_pade4_hess_anti_kernel's body, inputs from LDS) ==================
    {
        double* __restrict__ tsave = sm + kLdsGk;     // the stage-A tiles T_k go where the images were, once those are in registers
        double* __restrict__ red = scr;               // the reduction rows alias the transpose scratch (read back before they are written)
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        v4d gA[kMU];
        if constexpr (!ELL) {
#pragma unroll
            for (int u = 0; u < kMU; ++u) gA[u] = fu_lds_get(sm + kLdsGk + (u < m ? u : (m > 0 ? m - 1 : 0)) * 256, lane);
        }
        const v4d mv = fu_lds_get(sm + kLdsM, lane);
        Ga = fu_lds_get(sm + kLdsGa, lane);
        u0 = fu_lds_get(sm + kLdsU0, lane);
        u1 = fu_lds_get(sm + kLdsU1, lane);
        const double c2h2 = 2.0 * c2 * h, hh2 = 0.5 * h;
        v4d Sc, Db, MD;                                     // c1 [S | S], [D | D], [M | c2 h^2 D]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Sc[r] = c1 * (u1[r] + u0[r]);
            Db[r] = u1[r] - u0[r];
            MD[r] = left ? mv[r] : hc2 * Db[r];
        }
        v4d Y, PNn[kMU / 2];                               // G [M | c2 h^2 D];  [-N_k | -N_k+1] per drive pair
        if constexpr (SACOPY) {
            // stage A came from the copy wave (its registers held everything it needs; it ran between that wave's tile copies)
            __syncthreads();
            const double* __restrict__ hand = sm + kLdsRedC;
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) PNn[p2] = fu_lds_get(hand + 256 * p2, lane);
            Y = fu_lds_get(hand + 256 * (kMU / 2), lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tiles are in registers: the copy wave may reuse the region
            if (lane == 0) __hip_atomic_store(reinterpret_cast<int*>(sm + kLdsFlag), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            QC_STAMP(P, b, lane, 5);              // compute wave: second barrier passed, hand-over taken
        } else {
            // ---- stage A: G MD and G_k MD, interleaved
            v4d T[kMU];
            {
                constexpr int NA = 1 + kMU;
                v4d aA[NA], bA[NA], dA[NA];
                aA[0] = Ga;
                bA[0] = MD;
#pragma unroll
                for (int u = 0; u < kMU; ++u) {
                    aA[1 + u] = gA[u];
                    bA[1 + u] = MD;
                }
                mm16_multi<NA>(aA, bA, dA);
                Y = dA[0];
#pragma unroll
                for (int u = 0; u < kMU; ++u) T[u] = dA[1 + u];
            }
#pragma unroll
            for (int u = 0; u < kMU; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tsave[(u * 4 + r) * 64 + lane] = T[u][r];
            }
            QC_STAMP(P, b, lane, 4);              // compute wave: stage A through, tiles parked
            if constexpr (AACOPY) __syncthreads();    // the copy wave takes the (a, a) sums from here
            QC_STAMP(P, b, lane, 5);              // compute wave: second barrier passed
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) PNn[p2] = fu_sel(left, T[2 * p2], swap8(T[2 * p2 + 1]));
        }
        // ---- stage B
        v4d Q[kMU / 2], Y2;                                 // 2 c2 h [N'' + N' pairs], [M2 | .]
        {
            v4d YL, Gs;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                YL[r] = left ? c2h2 * Y[r] : 0.0;           // [2 c2 h (-M1) | 0]
                Gs[r] = c2h2 * Ga[r];
            }
            const v4d YR = swap8(YL);                       // [0 | 2 c2 h (-M1)]
            Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[0], Y[0], zero, 0, 0, 0);
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[0], PNn[p2][0], zero, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) {
                Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], Y[kk], Y2, 0, 0, 0);
#pragma unroll
                for (int p2 = 0; p2 < kMU / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[kk], PNn[p2][kk], Q[p2], 0, 0, 0);
            }
            if constexpr (ELL) {
                // + G_2p YL + G_2p+1 YR with YL = [y | 0], YR = [0 | y]: a left lane's accumulator takes the one term w_2p[a] y[c_2p[a]][j],
                // a right lane's w_2p+1[a] y[c_2p+1[a]][j - 8]; every other term of the two dense chains is an exact zero
                fu_put_rows(scr, YL, g, j);
                fu_lds_order();
#pragma unroll
                for (int p2 = 0; p2 < kMU / 2; ++p2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = (left ? 2 * p2 : 2 * p2 + 1) * 16 + 4 * r + g;
                        Q[p2][r] = __builtin_fma(ellTW[row], scr[ellTC[row] + jj], Q[p2][r]);
                    }
                }
                fu_lds_order();                   // (the transposes below rewrite the scratch)
                (void)YR;
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int p2 = 0; p2 < kMU / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[2 * p2][kk], YL[kk], Q[p2], 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int p2 = 0; p2 < kMU / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[2 * p2 + 1][kk], YR[kk], Q[p2], 0, 0, 0);
                }
            }
        }
        QC_STAMP(P, b, lane, 6);                  // compute wave: stage B issued
        // ---- matrix blocks: combine, transpose through LDS, store
        v4d ET, XT[kMU];
        {
            v4d tin[kMU + 1], tout[kMU + 1];
            const v4d ty = c1 * Y, ts = c2h2 * Y2;
            tin[0] = fu_sel(left, ty - ts, swap8(ty + ts));    // (U_t, h) | (h, U_t+1)
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) {
                const v4d lin = hc1 * PNn[p2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tin[1 + 2 * p2][r] = __builtin_fma(-hh2, Q[p2][r], lin[r]);
                    tin[2 + 2 * p2][r] = __builtin_fma(hh2, Q[p2][r], lin[r]);
                }
            }
            if constexpr (kTr2 == 0) {
                lds_transpose16_multi<kMU + 1>(scr, tin, tout, g, j);
            } else {
                v4d i1[kTr1], o1[kTr1], i2[kTr2], o2[kTr2];
#pragma unroll
                for (int q = 0; q < kTr1; ++q) i1[q] = tin[q];
#pragma unroll
                for (int q = 0; q < kTr2; ++q) i2[q] = tin[kTr1 + q];
                lds_transpose16_multi<kTr1>(scr, i1, o1, g, j);
                lds_transpose16_multi<kTr2>(scr, i2, o2, g, j);
#pragma unroll
                for (int q = 0; q < kTr1; ++q) tout[q] = o1[q];
#pragma unroll
                for (int q = 0; q < kTr2; ++q) tout[kTr1 + q] = o2[q];
            }
            ET = tout[0];
#pragma unroll
            for (int u = 0; u < kMU; ++u) XT[u] = tout[1 + u];
        }
        {
            const unsigned lo = 8u * (16u * g + j);
            if (ft) {
                double* __restrict__ eb = Hb + P.ho_Uh;     // (U_t, h): columns 0..7, (h, U_t+1): columns 8..15 of the tile
                double* __restrict__ fb = Hb + P.ho_hU;
                fu_st_off(eb, lo, ET[0]);
                fu_st_off(eb, lo + 512u, ET[1]);
                fu_st_off(fb, lo, ET[2]);
                fu_st_off(fb, lo + 512u, ET[3]);
            }
            double* __restrict__ xb = Hb + P.ho_Ua;
            double* __restrict__ yb = Hb + P.ho_aU;
#pragma unroll
            for (int u = 0; u < kMU; u += 2) {
                if (u < m) {
                    fu_st_off(xb, lo + 1024u * u, XT[u][0]);
                    fu_st_off(yb, lo + 1024u * u, XT[u + 1][0]);
                    fu_st_off(xb, lo + 1024u * u + 512u, XT[u][1]);
                    fu_st_off(yb, lo + 1024u * u + 512u, XT[u + 1][1]);
                    if (u + 1 < m) {
                        fu_st_off(xb, lo + 1024u * (u + 1), XT[u][2]);
                        fu_st_off(yb, lo + 1024u * (u + 1), XT[u + 1][2]);
                        fu_st_off(xb, lo + 1024u * (u + 1) + 512u, XT[u][3]);
                        fu_st_off(yb, lo + 1024u * (u + 1) + 512u, XT[u + 1][3]);
                    }
                }
            }
        }
        QC_STAMP(P, b, lane, 7);                  // compute wave: Hessian matrix blocks' stores issued
        // ---- scalar blocks
        constexpr int kRowShift = AACOPY ? R::kAA : 0;      // the compute wave's rows start at the pair rows when the copy wave has the rest
        if constexpr (!AACOPY) {
            v4d Tn[kMU];                                    // -T_u
#pragma unroll
            for (int u = 0; u < kMU; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Tn[u][r] = -tsave[(u * 4 + r) * 64 + lane];
            }
#pragma unroll
            for (int v = 0; v < kMU; ++v) {
                v4d Tsw;                                    // swap8(T_v)
#pragma unroll
                for (int r = 0; r < 4; ++r) Tsw[r] = tsave[(v * 4 + r) * 64 + (lane ^ 8)];
#pragma unroll
                for (int u = 0; u <= v; ++u) red[(v * (v + 1) / 2 + u) * kFuStride + lane] = fu_dot4(Tn[u], Tsw);
            }
        }
        if constexpr (LATE) {
            // S and D again, from the knots' tiles still in LDS: 16 registers less through the two MFMA stages, where this wave sits
            // at its 256-register budget (the same operations on the same values: the same bits)
            asm volatile("" ::: "memory");
            const v4d w0 = fu_lds_get(sm + kLdsU0, lane), w1 = fu_lds_get(sm + kLdsU1, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Sc[r] = c1 * (w1[r] + w0[r]);
                Db[r] = w1[r] - w0[r];
            }
        }
        if (ft) {
#pragma unroll
            for (int p2 = 0; p2 < kMU / 2; ++p2) red[(R::kAA - kRowShift + p2) * kFuStride + lane] = fu_dot4(Q[p2], Db) + fu_dot4(PNn[p2], Sc);
            red[(R::kAA - kRowShift + R::kPair) * kFuStride + lane] = (2.0 * c2) * fu_dot4(Y2, Db);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        fu_reduce_rows<kMU>(P, red, Hb, lane, m, ft, kRowShift, R::kRows, kRowShift, staged, scal);
        if (staged) fu_scalar_run_store(P, scal, scal_count, Hb, lane);
        QC_STAMP(P, b, lane, 8);                  // compute wave: every store issued
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 9);              // ... and drained
        }
        QC_STAMP_FLUSH(P, b, lane, 0, 9);
    }
}

}  // namespace

// The row-gather form (ELL) serves trajectories of one to four device rounds (four workgroups per CU: 1024 intervals a round): there the
// matrix pipes decide how fast workgroups retire and make room -- T = 1500 / 2000 / 3000 / 4000: 21.6 / 25.5 / 36.3 / 44.0 us against 22.7 /
// 26.4 / 37.3 / 45.9 with the dense images, on every box measured -- while a single round follows its store stream and is 0.3 - 0.7 us
// FASTER with the images (T = 1000: 12.8 against 13.5), and long streams are decided by the box: T = 6000 ... 32000 lose 2 - 4 % with the
// gathers on two boxes and win 6 - 8 % (T = 16000: 143 - 148 against 155 - 156 us) on a third (profiles/r05_fused_ell16.txt: A/B inside one
// process on the same buffers, profiles/fused_ab.py -- across processes long streams are bimodal).  QC_FUSED_ELL=0 / 1: never / always.
constexpr int kFuEllMinIntervals = 1025, kFuEllMaxIntervals = 4096;
bool qc_mfma16_fused_gathers(const QcParams& P) {
#ifdef QC_FUSED_ELL_DYNAMIC      /* experiment builds: the switch is read at every launch (A/B inside one process, on the same buffers) */
    const int mode = getenv("QC_FUSED_ELL") ? atoi(getenv("QC_FUSED_ELL")) : -1;
#else
    static const int mode = getenv("QC_FUSED_ELL") ? atoi(getenv("QC_FUSED_ELL")) : -1;
#endif
    return P.ell16 != nullptr && mode != 0 && (mode == 1 || (P.n_int >= kFuEllMinIntervals && P.n_int <= kFuEllMaxIntervals));
}

bool qc_mfma16_fused_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n == 16 && P.nc == 8 && P.antisym && P.m >= 1 && P.m <= 6 && P.hess_nnz > 0 && P.store_mode == 2 &&
           (P.stamps == nullptr || (P.m > 4 && P.n_int <= kFuMaxStamped)) && P.dbg_skip == 0 && P.Gx != nullptr && P.copies == P.nc;
}

hipError_t qc_launch_mfma16_fused(const QcParams& P, const double* dZ, const double* dMu, double* dF, double* dJ, double* dH, hipStream_t st) {
    for (int b0 = 0; b0 < P.n_int; b0 += kFuMaxGrid) {
        const int n = P.n_int - b0 < kFuMaxGrid ? P.n_int - b0 : kFuMaxGrid;
        const double* Zt = dZ + (P.t_begin + b0) * (long long)P.zdim;
        const double* mu0 = dMu + (P.t_begin + b0) * P.F_stride + P.F_off;
        double* Fp = dF ? dF + (size_t)b0 * P.F_stride : nullptr;
        double* Jp = dJ + (size_t)b0 * P.J_stride;
        double* Hp = dH + (size_t)b0 * P.H_stride;
#define QC_FU(MU_, V_) hipLaunchKernelGGL((qc_mfma16_pade4_fused_kernel<MU_, V_, false>), dim3(n), dim3(kFuThreads), 0, st, P.Gx, Zt, mu0, n, P.zdim, P.off_a, P.off_dt, \
                                          P.m, P.off_U, (int)P.F_stride, P, Fp, Jp, Hp)
#define QC_FU_ELL(MU_) hipLaunchKernelGGL((qc_mfma16_pade4_fused_kernel<MU_, 53, true>), dim3(n), dim3(kFuThreads), 0, st, P.Gx, Zt, mu0, n, P.zdim, P.off_a, P.off_dt, \
                                          P.m, P.off_U, (int)P.F_stride, P, Fp, Jp, Hp)
        // VAR bits: 1 S / D of the scalar blocks re-read from LDS (no scratch spills), 4 the (a, a) sums on the copy wave, 8 time stamps,
        // 16 no copy before the hand-off, 32 stage A on the copy wave.  QC_FUSED_VARIANT=0: the plain composition of the two kernels,
        // 21: everything but bit 32 -- for comparison.
        static const bool plain = getenv("QC_FUSED_VARIANT") && atoi(getenv("QC_FUSED_VARIANT")) == 0;
        static const int var = getenv("QC_FUSED_VARIANT") ? atoi(getenv("QC_FUSED_VARIANT")) : -1;
        const bool ell = qc_mfma16_fused_gathers(P) && var < 0;
        if (P.stamps != nullptr && ell) hipLaunchKernelGGL((qc_mfma16_pade4_fused_kernel<6, 61, true>), dim3(n), dim3(kFuThreads), 0, st, P.Gx, Zt, mu0, n, P.zdim,
                                                           P.off_a, P.off_dt, P.m, P.off_U, (int)P.F_stride, P, Fp, Jp, Hp);
        else if (P.stamps != nullptr) QC_FU(6, 61);
        else if (ell && P.m <= 2) QC_FU_ELL(2);
        else if (ell && P.m <= 4) QC_FU_ELL(4);
        else if (ell) QC_FU_ELL(6);
        else if (P.m <= 2) QC_FU(2, 53);
        else if (P.m <= 4) QC_FU(4, 53);
        else if (plain) QC_FU(6, 0);
        else if (var == 21) QC_FU(6, 21);
        else QC_FU(6, 53);
#undef QC_FU
#undef QC_FU_ELL
    }
    return hipGetLastError();
}

// Rows of the drive generators for the row-gather form: every drive generator of a handle the one-call kernel serves has at most ONE
// entry per row.  blob = [6][16] weights (doubles), [6][16] columns x kXS (ints); unused rows / drives: weight 0, column 0.
bool qc_mfma16_ell_build(const QcParams& P, const double* G, std::vector<char>* blob) {
    if (P.integrator != QC_PADE || P.p != 2 || P.n != 16 || P.nc != 8 || !P.antisym || P.m < 1 || P.m > 6 || P.hess_nnz == 0) return false;
    const int n = 16, m = P.m;
    blob->assign(6 * 16 * 8 + 6 * 16 * 4, 0);
    double* tw = reinterpret_cast<double*>(blob->data());
    int* tc = reinterpret_cast<int*>(blob->data() + 6 * 16 * 8);
    for (int k = 0; k < m; ++k)
        for (int a = 0; a < n; ++a) {
            int cnt = 0;
            for (int c = 0; c < n; ++c) {
                const double v = G[(size_t)(k + 1) * n * n + (size_t)c * n + a];      // drive k, row a, column c (column-major)
                if (v == 0.0) continue;
                if (++cnt > 1) return false;
                tw[k * 16 + a] = v;
                tc[k * 16 + a] = c * kXS;
            }
        }
    return true;
}
