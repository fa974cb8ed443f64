// mu_d2F of every interval, order-4 Pade, 2N = 16 (a unitary on 8 levels: BASELINE configs 3 and 4), up to 6 drives, exactly
// antisymmetric generators: the one-wave kernel of qc_mfma_hess.hip (qc_mfma16_pade4_hess_anti_kernel) with a second wave per interval
// that takes the last drive pair's share of the back half and the (a, a) sums.  Launches of up to 1024 intervals (one round of the
// device); longer ones keep the one-wave kernel.  (Reference call site: test/scripts/integrator_test_1qubit.jl:50-52.)
//
// What a launch of the one-wave kernel waits for (tests/hip/launch_gap2.hip, profiles/r03_launch_gap2.txt: a launch = the waves'
// lifetime + 1.1 us, and bytes stored at a wave's end drain for bytes / 6.8 TB/s AFTER it): the longest wave (7.4 - 8.1 us: loads 1.4,
// stage A 1.0, stage B 1.6, transposes 0.6, 28 stores 0.9, scalar blocks 1.7) and, close behind, the drain of the matrix blocks, whose
// stores leave between 4.9 and 6.3 us.  Both are moved here, with the arithmetic of every value unchanged (the same operations in the
// same order: the same bits as the one-wave kernel and as the fused kernel of qc_mfma_fused.hip, tests/test_gpu_parity.py):
//
//   wave 0  loads, G, stage A as before; parks the T_k, G and Y in LDS; barrier (never waits: wave 1 is already there).  Stage B of Y2
//           and of all drive pairs but the last; their tiles -- (U, h), (h, U) and the pairs' (U, a), (a, U) blocks: five sevenths of
//           the bytes at six drives -- combined, transposed, stored; the (a, h) sums of those pairs and (h, h)
//   wave 1  sleeps 1.3 us, then requests what it needs besides the parked tiles -- the two images of the LAST drive pair and the knots'
//           tiles -- (requested at entry they queued in front of wave 0's loads in the CU's one vector-memory pipeline: its loads back at
//           2.0 instead of 1.4 us); after the barrier the last pair's chain of twelve MFMAs, its transposes and stores, its (a, h) sum, and
//           the 21 (a, a) products from the parked T_k with their reduction: three quarters of the one-wave kernel's 1.7 us tail
//
// Timeline (profiles/r03_hess2_timeline.txt, stamped build, T = 1000): wave 0's blocks stored at 5.2 us, wave 1's at 4.5, the waves
// done at 6.3 / 6.1 (6.85 max) against 7.4 (8.2) in the one-wave kernel; launches 8.4 - 8.65 against 8.5 - 8.75 us at T = 1000, 7.0
// against 7.5 at T = 500, 6.1 - 6.8 against 7.0 at T = 250 (DESIGN.md 5.1a has the three forms that were measured and not kept).
//
// LDS per workgroup (doubles, kMU = 6): parked T_k 6 x 256 | wave 0's transposes 5 x 272 (its reduction rows alias them) | wave 1's
// reduction rows 22 x 65 (its transposes alias them) | G, Y 2 x 256 = 37.8 KB: four workgroups per CU.
#include <stdlib.h>

#include "qc_mfma_hess_common.h"

namespace {

using namespace qc_mfma;

constexpr int kH2Threads = 128;

template <int kMU>
struct H2 {
    static constexpr int kPairs = kMU / 2;
    static constexpr int kP1 = kPairs > 1 ? kPairs - 1 : 1;  // drive pairs of stage B's first part; the last pair (if there are two or more) follows
    static constexpr int kP2 = kPairs - kP1;
    static constexpr int kAA = kMU * (kMU + 1) / 2;
    static constexpr int kNT1 = 1 + 2 * kP1;
    static constexpr int kSave = 0, kSaveLen = kMU * 256;
    static constexpr int kScr = kSave + kSaveLen, kScrLen = kNT1 * 272;
    static constexpr int kRedH = kScr + kScrLen, kRedHLen = (kAA + kP2) * kFuStride;      // wave 1's rows; its transposes alias them (before)
    static constexpr int kHand = kRedH + kRedHLen, kHandLen = kP2 > 0 ? 512 : 0;           // G | Y for wave 1's drive pair
    static constexpr int kTotal = kHand + kHandLen;
    static_assert(2 * kP2 * 272 <= kRedHLen, "wave 1's transpose scratch aliases its reduction rows");
    static_assert((kP1 + 1) * kFuStride <= kScrLen, "wave 0's reduction rows alias its transpose scratch");
    static_assert(kTotal * 8 <= 40960, "four workgroups per CU");
};

// Sums the 64 per-lane partials of the scalar-block rows held in `slots` [0, n_slots) at red + slot * kFuStride, in the fixed order
// of qc_mfma16_pade4_hess_anti_kernel (fu_reduce_rows), and stores them.  Slot s is row s of the (a, a) block for s < n_aa_slots,
// then drive pair `first_pair + (s - n_aa_slots)` of the (a, h) block, then (with_hh) the (h, h) entry.
template <int kMU>
__device__ __forceinline__ void h2_reduce(const QcParams& P, const double* __restrict__ red, double* __restrict__ Hb, int lane, int m, bool ft,
                                          int n_aa_slots, int n_pair_slots, int first_pair, bool with_hh) {
    const int naa = m * (m + 1) / 2;
    const int half = lane >> 5;
    const int n_slots = n_aa_slots + n_pair_slots + (with_hh ? 1 : 0);
    const int slot = lane & 31;                    // (at most 32 slots: 21 + 3 + 1 at six drives)
    const bool in = slot < n_slots;
    const bool aa_row = slot < n_aa_slots;
    const bool pair_row = !aa_row && slot < n_aa_slots + n_pair_slots;
    const bool hh_row = in && !aa_row && !pair_row;
    const int drive = 2 * (first_pair + slot - n_aa_slots) + half;
    const bool wanted = in && ((aa_row && slot < naa) || (ft && ((pair_row && drive < m) || (hh_row && half == 0))));
    const double* rp = red + (in ? slot : 0) * kFuStride + 8 * half;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a0 += rp[16 * i] + rp[16 * i + 4];
        a1 += rp[16 * i + 1] + rp[16 * i + 5];
        a2 += rp[16 * i + 2] + rp[16 * i + 6];
        a3 += rp[16 * i + 3] + rp[16 * i + 7];
    }
    const double own = (a0 + a1) + (a2 + a3);
    const double both = own + xor32_f64(own, lane);
    if (wanted) {
        if (aa_row) {
            if (half == 0) Hb[P.ho_aa + slot] = both;
        } else {
            Hb[pair_row ? P.ho_ah + drive : P.ho_hh] = own;
        }
    }
}

// One drive pair's two transposed tiles -> the (U_t, a) / (a, U_t+1) blocks of drives u and u + 1 (qc_mfma16_pade4_hess_anti_kernel's stores)
__device__ __forceinline__ void h2_store_pair(double* __restrict__ xb, double* __restrict__ yb, unsigned lo, int u, int m, const v4d& XA, const v4d& XB) {
    if (u < m) {
        fu_st_off(xb, lo + 1024u * u, XA[0]);
        fu_st_off(yb, lo + 1024u * u, XB[0]);
        fu_st_off(xb, lo + 1024u * u + 512u, XA[1]);
        fu_st_off(yb, lo + 1024u * u + 512u, XB[1]);
        if (u + 1 < m) {
            fu_st_off(xb, lo + 1024u * (u + 1), XA[2]);
            fu_st_off(yb, lo + 1024u * (u + 1), XB[2]);
            fu_st_off(xb, lo + 1024u * (u + 1) + 512u, XA[3]);
            fu_st_off(yb, lo + 1024u * (u + 1) + 512u, XB[3]);
        }
    }
}

// Leading arguments: preloaded into scalar registers at wave launch (Makefile: -amdgpu-kernarg-preload-count), as in the one-wave kernel.
template <int kMU, bool DIAG>
__global__ __launch_bounds__(kH2Threads, 2) void qc_mfma16_pade4_hess2_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                             const double* __restrict__ hot_mu0, int hot_n_int, int hot_zdim,
                                                                             int hot_off_a, int hot_off_dt, int hot_m, int hot_off_U,
                                                                             int hot_f_stride, const QcParams P, double* __restrict__ H) {
    using L = H2<kMU>;
    QC_STAMP_DECL;
    QC_STAMP(P, 0, 0, 0);
    __shared__ __attribute__((aligned(16))) double sm[L::kTotal];
    QcKernargTouch<sizeof(QcParams) + 128> touch;
    touch.request();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = hot_m;
    const bool ft = hot_off_dt >= 0;
    double* __restrict__ tsave = sm + L::kSave;

    if (wave == 1) {
        // ================= wave 1: the last drive pair's stage B and blocks, the (a, a) sums ================================
        const double* __restrict__ Gx1 = hot_Gx;
        v4d gL[2];                                // images of the last pair's drives (from L2 / L1: wave 0 requests the same lines)
        if constexpr (L::kP2 > 0) {
            // This wave needs its loads 2.8 us from now; requested at once they queue in the CU's one vector-memory pipeline in front
            // of wave 0's, whose every load is on the critical path (wave 0's loads back at 2.0 instead of 1.4 us: the launch 9.0
            // instead of 8.6 us).  Its requests leave when wave 0's are through.
            __builtin_amdgcn_s_sleep(44);         // 44 x 64 cycles = 1.3 us
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = 2 * L::kP1 + q, k = u < m ? u : (m > 0 ? m - 1 : 0);
                gL[q] = load_image_tile(Gx1 + (m > 0 ? k + 1 : 0) * 256, lane);
            }
        }
        touch.consume();
        if ((int)blockIdx.x >= hot_n_int) return;
        const int b = qc_xcd_remap(blockIdx.x, hot_n_int);
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
        double* __restrict__ redh = sm + L::kRedH;
        const int g = lane >> 4, j = lane & 15, jj = j & 7;
        const bool left = j < 8;
        v4d Sc, Db;
        double hc1 = 0.0, c2h2 = 0.0, hh2 = 0.0;
        if constexpr (L::kP2 > 0) {
            const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
            const double* __restrict__ z1 = z0 + hot_zdim;
            const double c1 = P.c[1], c2 = P.c[2];
            const double h = ft ? load_uniform(z0 + hot_off_dt) : opaque_scalar(P.dt_fixed);
            const v4d u0 = load_col16_T(z0 + hot_off_U + jj * 16, g);
            const v4d u1 = load_col16_T(z1 + hot_off_U + jj * 16, g);
            hc1 = h * c1, c2h2 = 2.0 * c2 * h, hh2 = 0.5 * h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Sc[r] = c1 * (u1[r] + u0[r]);
                Db[r] = u1[r] - u0[r];
            }
        }
        if constexpr (DIAG) qc_ts_[10] = qc_ts_[0];
        __syncthreads();                          // the T_k (and G, Y) are parked
        QC_STAMP(P, b, lane, 11);                 // wave 1: released
        if constexpr (L::kP2 > 0) {
            constexpr int p2 = L::kP1;
            const v4d zero = {0.0, 0.0, 0.0, 0.0};
            const v4d Ga = fu_lds_get(sm + L::kHand, lane), Y = fu_lds_get(sm + L::kHand + 256, lane);
            v4d PN, YL, Gs;                       // [-N_k | -N_k+1] = sel(left, T_k, swap8(T_k+1)): the swapped half read at lane ^ 8
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ta = tsave[((2 * p2) * 4 + r) * 64 + lane], tb = tsave[((2 * p2 + 1) * 4 + r) * 64 + (lane ^ 8)];
                PN[r] = left ? ta : tb;
                YL[r] = left ? c2h2 * Y[r] : 0.0;
                Gs[r] = c2h2 * Ga[r];
            }
            const v4d YR = swap8(YL);
            v4d Q = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[0], PN[0], zero, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) Q = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[kk], PN[kk], Q, 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Q = __builtin_amdgcn_mfma_f64_16x16x4f64(gL[0][kk], YL[kk], Q, 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Q = __builtin_amdgcn_mfma_f64_16x16x4f64(gL[1][kk], YR[kk], Q, 0, 0, 0);
            v4d tin[2], tout[2];
            const v4d lin = hc1 * PN;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tin[0][r] = __builtin_fma(-hh2, Q[r], lin[r]);
                tin[1][r] = __builtin_fma(hh2, Q[r], lin[r]);
            }
            lds_transpose16_multi<2>(redh, tin, tout, g, j);      // (the rows below are written after these reads)
            h2_store_pair(Hb + P.ho_Ua, Hb + P.ho_aU, 8u * (16u * g + j), 2 * p2, m, tout[0], tout[1]);
            QC_STAMP(P, b, lane, 14);             // wave 1: its pair's blocks stored
            if (ft) redh[L::kAA * kFuStride + lane] = fu_dot4(Q, Db) + fu_dot4(PN, Sc);
        }
        v4d T[kMU];
#pragma unroll
        for (int u = 0; u < kMU; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) T[u][r] = tsave[(u * 4 + r) * 64 + lane];
        }
#pragma unroll
        for (int v = 0; v < kMU; ++v) {
            v4d Tsw;                              // swap8(T_v): read at lane ^ 8
#pragma unroll
            for (int r = 0; r < 4; ++r) Tsw[r] = tsave[(v * 4 + r) * 64 + (lane ^ 8)];
#pragma unroll
            for (int u = 0; u <= v; ++u) {
                // (a_u, a_v) = -sum T_u . swap8(T_v), with the roundings qc_mfma16_pade4_hess_anti_kernel's compilation has (first product
                // rounded on its own, the other three fused in order): written out, because the compiler's choice of WHICH product stays
                // un-fused depends on the shape of the surrounding code, and the values must be the same bits (qc_mfma_fused.hip)
                const double t0 = T[u][0] * Tsw[0];
                double r = __builtin_fma(-T[u][1], Tsw[1], -t0);
                r = __builtin_fma(-T[u][2], Tsw[2], r);
                redh[(v * (v + 1) / 2 + u) * kFuStride + lane] = __builtin_fma(-T[u][3], Tsw[3], r);
            }
        }
        QC_STAMP(P, b, lane, 12);                 // wave 1: products through
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        h2_reduce<kMU>(P, redh, Hb, lane, m, ft, L::kAA, L::kP2, L::kP1, false);
        QC_STAMP(P, b, lane, 13);                 // wave 1: done
        QC_STAMP_FLUSH(P, b, lane, 10, 14);
        return;
    }

    // ===================== wave 0 ==========================================================================================
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gx = hot_Gx;
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    // the generator images depend on nothing but the kernel arguments: requested before any address of the interval is computed
    v4d gA[kMU];
    const v4d G0 = load_image_tile(Gx, lane);
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
        const int k = u < m ? u : (m > 0 ? m - 1 : 0);
        gA[u] = load_image_tile(Gx + (m > 0 ? k + 1 : 0) * 256, lane);
    }
    if ((int)blockIdx.x >= hot_n_int) return;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const int b = qc_xcd_remap(blockIdx.x, hot_n_int);
    const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
    const double* __restrict__ z1 = z0 + hot_zdim;
    const double* __restrict__ mu = hot_mu0 + (long long)b * hot_f_stride;
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double av = load_amp_lanes(z0, hot_off_a, m, lane);
    const double h = ft ? load_uniform(z0 + hot_off_dt) : opaque_scalar(P.dt_fixed);
    const v4d u0 = load_col16_T(z0 + hot_off_U + jj * 16, g);
    const v4d u1 = load_col16_T(z1 + hot_off_U + jj * 16, g);
    const v4d mv = load_col16_T(mu + jj * 16, g);               // M = reshape(mu_t[0:s], 16, 8): both lane halves hold the same 8 columns
    const bool dfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
    double mud[2] = {0.0, 0.0};
    if (dfast) {
#pragma unroll
        for (int d = 0; d < 2; ++d) mud[d] = mu[P.drow[d] + (lane < P.ddim_i[d] ? lane : 0)];
    }
    QC_STAMP(P, b, lane, 1);                      // every load of the interval requested
    touch.consume();
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QC_STAMP(P, b, lane, 2);                  // ... and back
    }
    v4d Ga = G0;
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
        const double a = (u < m) ? bcast_lane(av, u) : 0.0;
        Ga += a * gA[u];
    }
    const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h, hh2 = 0.5 * h;
    v4d Sc, Db, MD;                               // c1 [S | S], [D | D], [M | c2 h^2 D]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Sc[r] = c1 * (u1[r] + u0[r]);
        Db[r] = u1[r] - u0[r];
        MD[r] = left ? mv[r] : hc2 * Db[r];
    }
    // ---- stage A: G MD and G_k MD, interleaved
    v4d Y, T[kMU];
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
    QC_STAMP(P, b, lane, 3);                      // stage A issued
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tsave[(u * 4 + r) * 64 + lane] = T[u][r];
    }
    if constexpr (L::kP2 > 0) {
        fu_lds_put(sm + L::kHand, lane, Ga);
        fu_lds_put(sm + L::kHand + 256, lane, Y);
    }
    __syncthreads();                              // releases wave 1 (it has been waiting here: no delay for this wave)
    QC_STAMP(P, b, lane, 4);                      // T_k parked, barrier passed
    // ---- stage B, first part: Y2 and the drive pairs [0, kP1), interleaved
    double* __restrict__ tscr = sm + L::kScr;
    const unsigned lo = 8u * (16u * g + j);       // lane (g, j), register r of a transposed tile: byte offset 8 (16 (4 r + g) + j)
    double* __restrict__ xb = Hb + P.ho_Ua;
    double* __restrict__ yb = Hb + P.ho_aU;
    v4d PNn[L::kP1], Q[L::kP1], Y2, YL, YR, Gs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        YL[r] = left ? c2h2 * Y[r] : 0.0;         // [2 c2 h (-M1) | 0]
        Gs[r] = c2h2 * Ga[r];
    }
    YR = swap8(YL);                               // [0 | 2 c2 h (-M1)]
#pragma unroll
    for (int p2 = 0; p2 < L::kP1; ++p2) PNn[p2] = fu_sel(left, T[2 * p2], swap8(T[2 * p2 + 1]));
    Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[0], Y[0], zero, 0, 0, 0);
#pragma unroll
    for (int p2 = 0; p2 < L::kP1; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[0], PNn[p2][0], zero, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
        Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], Y[kk], Y2, 0, 0, 0);
#pragma unroll
        for (int p2 = 0; p2 < L::kP1; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[kk], PNn[p2][kk], Q[p2], 0, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int p2 = 0; p2 < L::kP1; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[2 * p2][kk], YL[kk], Q[p2], 0, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int p2 = 0; p2 < L::kP1; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[2 * p2 + 1][kk], YR[kk], Q[p2], 0, 0, 0);
    }
    QC_STAMP(P, b, lane, 5);                      // stage B, first part issued
    {
        v4d tin[L::kNT1], tout[L::kNT1];
        const v4d ty = c1 * Y, ts = c2h2 * Y2;
        tin[0] = fu_sel(left, ty - ts, swap8(ty + ts));    // (U_t, h) | (h, U_t+1)
#pragma unroll
        for (int p2 = 0; p2 < L::kP1; ++p2) {
            const v4d lin = hc1 * PNn[p2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tin[1 + 2 * p2][r] = __builtin_fma(-hh2, Q[p2][r], lin[r]);
                tin[2 + 2 * p2][r] = __builtin_fma(hh2, Q[p2][r], lin[r]);
            }
        }
        lds_transpose16_multi<L::kNT1>(tscr, tin, tout, g, j);
        if (ft) {
            double* __restrict__ eb = Hb + P.ho_Uh;     // (U_t, h): columns 0..7, (h, U_t+1): columns 8..15 of the tile
            double* __restrict__ fb = Hb + P.ho_hU;
            fu_st_off(eb, lo, tout[0][0]);
            fu_st_off(eb, lo + 512u, tout[0][1]);
            fu_st_off(fb, lo, tout[0][2]);
            fu_st_off(fb, lo + 512u, tout[0][3]);
        }
#pragma unroll
        for (int p2 = 0; p2 < L::kP1; ++p2) h2_store_pair(xb, yb, lo, 2 * p2, m, tout[1 + 2 * p2], tout[2 + 2 * p2]);
    }
    QC_STAMP(P, b, lane, 6);                      // first part's matrix blocks stored
    QC_STAMP(P, b, lane, 7);                      // (as 6)
    // derivative integrators: d2/d(dx_i) dh = -mu_i; alignment padding (explicit zeros)
    if (dfast) {
        int o = P.ho_d;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (lane < P.ddim_i[d]) Hb[o + lane] = -mud[d];
            o += P.ddim_i[d];
        }
        for (int i = lane; i < P.h_pad; i += 64) Hb[P.hess_nnz + i] = 0.0;
    } else {
        qc_hess_tail(P, mu, Hb, lane, 64);
    }
    // ---- (a, h) and (h, h): rows alias the transpose scratch
    if (ft) {
#pragma unroll
        for (int p2 = 0; p2 < L::kP1; ++p2) tscr[p2 * kFuStride + lane] = fu_dot4(Q[p2], Db) + fu_dot4(PNn[p2], Sc);
        tscr[L::kP1 * kFuStride + lane] = (2.0 * c2) * fu_dot4(Y2, Db);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    h2_reduce<kMU>(P, tscr, Hb, lane, m, ft, 0, L::kP1, 0, true);
    QC_STAMP(P, b, lane, 8);                      // every store issued
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QC_STAMP(P, b, lane, 9);                  // ... and drained
    }
    QC_STAMP_FLUSH(P, b, lane, 0, 9);
}

}  // namespace

// Serves launches of up to kH2MaxInt intervals -- one round of the device (four workgroups per CU): measured against the one-wave kernel
// (profiles/r03_hess2_ab.txt) 6.1 - 6.8 / 7.0 / 8.6 us against 7.1 / 7.5 / 8.7 at T = 250 / 500 / 1000; beyond one round the one-wave
// kernel's persistent grid is faster (T = 2000: 15.1 against 17.3 us; T = 8000: 47.7 against 52.9).  QC_HESS_TWO_WAVES=0: never.
constexpr int kH2MaxInt = 1024;

bool qc_mfma16_hess2_supported(const QcParams& P) {
    static const bool off = getenv("QC_HESS_TWO_WAVES") && atoi(getenv("QC_HESS_TWO_WAVES")) == 0;
    return !off && P.n_int <= kH2MaxInt && P.integrator == QC_PADE && P.p == 2 && P.n == 16 && P.nc == 8 && P.antisym && P.m >= 1 && P.m <= 6 && P.hess_nnz > 0 &&
           (P.stamps == nullptr || P.m > 4) && P.dbg_skip == 0 && P.Gx != nullptr;
}

hipError_t qc_launch_mfma16_hess2(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    const double* Zt = dZ + P.t_begin * (long long)P.zdim;
    const double* mu0 = dMu + P.t_begin * P.F_stride + P.F_off;
#define QC_H2(MU_, D_) hipLaunchKernelGGL((qc_mfma16_pade4_hess2_kernel<MU_, D_>), dim3(P.n_int), dim3(kH2Threads), 0, st, P.Gx, Zt, mu0, P.n_int, P.zdim, \
                                          P.off_a, P.off_dt, P.m, P.off_U, (int)P.F_stride, P, dH)
    if (P.stamps != nullptr) QC_H2(6, true);
    else if (P.m <= 2) QC_H2(2, false);
    else if (P.m <= 4) QC_H2(4, false);
    else QC_H2(6, false);
#undef QC_H2
    return hipGetLastError();
}
