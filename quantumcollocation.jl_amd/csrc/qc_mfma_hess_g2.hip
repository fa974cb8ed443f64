// mu_d2F alone for drive generators with ONE entry per row (Pauli strings: BASELINE configs 3 / 4), order-4 Pade, 2N = 16, a whole
// unitary, Hermitian Hamiltonians: the launch Ipopt issues after the Jacobian has returned (MOI hands mu to
// eval_hessian_lagrangian only; reference call test/scripts/integrator_test_1qubit.jl:50-52).  One wavefront per interval.
//
// Round 6.  The row-gather instantiation of qc_mfma16_pade4_hess_anti_kernel (qc_mfma_hess.hip) carried the six stage-A tiles
// T_k = G_k [M | c2 h^2 D] through the whole wave for the (a, a) block -- 48 registers, 48 lane rotations, 84 multiply-adds per lane
// and 21 reduction rows through LDS (13 KB): 211 registers and 14.6 KB, eight waves per CU; a long trajectory was bound by how many
// intervals are in flight (timing-only ablations, profiles/r06_hess_ablation.txt: without the scalar blocks 36.4 -> 31.5 us at
// T = 8000, with neither stores nor scalar blocks 20.4 us -- a chain of latencies, two waves per SIMD).  Here:
//   * (a_u, a_v) = c2 h^2 <G_u G_v + G_v G_u, M D^T>  ((h^2/12) <M, (G_i G_j + G_j G_i) D>).  With one entry per row,
//     G_u G_v has one entry per row as well: the block is 2 x 16 gathered entries of the GRAM matrix Q = M (c2 h^2 D)^T per pair
//     (two MFMAs; operands read back from the row-major copy of [M | c2 h^2 D] the gathers need anyway), each lane (g, r) adds its
//     row r for the pairs g, g + 4, ... from a per-handle table (qc_mfma16_ell_build: weights and LDS offsets, 24 bytes per lane
//     and slot, requested with the first loads), 16 partial sums per pair meet in LDS.  No T_k at all: the pair tiles
//     [-N_2p | -N_2p+1] are gathered directly.
//   * 4 instead of 25 reduction rows; the scratch regions alias each other in the order they die: 9.6 KB at six drives.
// Everything else -- stage B, the matrix blocks and their transposed stores, the (a, h) / (h, h) sums -- operation for operation as
// the row-gather instantiation (same bits); the (a, a) entries agree with it to rounding (fewer terms, not the same sum).
#include <stdlib.h>

#include "qc_mfma_hess_common.h"

namespace {

using namespace qc_mfma;

constexpr int kG2Stride = 65;

__device__ inline v4d g2_img(const double* __restrict__ Gx, int mat, int lane) { return load_image_tile(Gx + mat * 256, lane); }
__device__ inline double g2_dot4(const v4d& a, const v4d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ inline v4d g2_sel(bool c, const v4d& a, const v4d& b) { return v4d{c ? a[0] : b[0], c ? a[1] : b[1], c ? a[2] : b[2], c ? a[3] : b[3]}; }
template <int MODE>
__device__ __forceinline__ void g2_st_off(double* __restrict__ ubase, unsigned byteoff, double v) {
    qc_st8m<MODE>(reinterpret_cast<double*>(reinterpret_cast<char*>(ubase) + byteoff), v);
}

template <int kHM, bool DIAG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) void qc_mfma16_pade4_hess_g2_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                     const double* __restrict__ hot_mu0, const void* __restrict__ hot_ell, int hot_n_int,
                                                                     int hot_zdim, int hot_off_a, int hot_off_dt, int hot_m, int hot_off_U, int hot_f_stride,
                                                                     const QcParams P,
                                                                     double* __restrict__ H) {
    QC_STAMP_DECL;
    QC_STAMP(P, 0, 0, 0);                                         // kernel entry
    QcKernargTouch<sizeof(QcParams) + 128> touch;
    touch.request();
    constexpr int kPairs = kHM / 2, kAA = kHM * (kHM + 1) / 2, kSlots = (kAA + 3) / 4;
    // LDS: row-major copies of [M | c2 h^2 D] and [2 c2 h (-M1) | 0], dead once the drive terms are gathered: the transposes (two tiles at
    // a time) and then the (a, h) / (h, h) rows reuse them; the Gram matrix and the pair partials keep regions of their own -- the (a, a)
    // block is made AFTER the matrix blocks' stores (nine tenths of the interval's bytes) are on their way.  9.6 KB at six drives.
    constexpr int kLds = 3 * 272 + kSlots * 64;
    static_assert((kPairs + 1) * kG2Stride <= 2 * 272, "reduction rows alias the scratch");
    __shared__ double lds[kLds];
    __shared__ double ellTW[96];
    __shared__ int ellTC[96];
    double* __restrict__ rMD = lds;
    double* __restrict__ rYL = lds + 272;
    double* __restrict__ rQ = lds + 544;
    double* __restrict__ rPart = lds + 816;
    double* __restrict__ tscr = lds;                              // transposes: after the early regions are dead
    double* __restrict__ red = lds;                               // (a, h) / (h, h) rows: after the transposed tiles are back in registers
    const int lane = threadIdx.x;
    const int m = hot_m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const bool ft = hot_off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    // ---- EVERY request of the wave in one batch, from the preloaded arguments alone: the interval's state first (HBM: the longest
    //      way), then this lane's table entries and the generator images (L2).  Round 6: the drives' rows used to be copied into LDS
    //      between the images and the interval's loads -- the copy waits for its data, so the interval's loads left one round trip
    //      late (stamped: 2.1 us from kernel entry to the last request at T = 1000, 4.1 us at T = 8000; profiles/r06_hess_g2_timeline.txt).
    const int b = qc_xcd_remap((int)blockIdx.x, hot_n_int);
    const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
    const double* __restrict__ z1 = z0 + hot_zdim;
    const double* __restrict__ mu = hot_mu0 + (long long)b * hot_f_stride;
    const double av = load_amp_lanes(z0, hot_off_a, m, lane);
    const double hv = load_uniform(z0 + (ft ? hot_off_dt : 0));
    const v4d u0 = load_col16_T(z0 + hot_off_U + jj * 16, g);
    const v4d u1 = load_col16_T(z1 + hot_off_U + jj * 16, g);
    const v4d mv = load_col16_T(mu + jj * 16, g);
    typedef const __attribute__((address_space(1))) double* gdp;
    typedef const __attribute__((address_space(1))) int* gip;
    typedef const __attribute__((address_space(1))) v2d* gv2;
    const char* ell = reinterpret_cast<const char*>(hot_ell);
    double tw0, tw1 = 0.0;                                        // the drives' rows: 96 entries, lanes l and 64 + l (qc_mfma_hess_common.h)
    int tc0, tc1 = 0;
    {
        const gdp twp = (gdp)(unsigned long long)ell;
        const gip tcp = (gip)(unsigned long long)(ell + 6 * 16 * 8);
        tw0 = twp[lane];
        tc0 = tcp[lane];
        if (lane < 32) {
            tw1 = twp[64 + lane];
            tc1 = tcp[64 + lane];
        }
    }
    v2d pw[kSlots];
    int poA[kSlots], poB[kSlots];
    {
        const gv2 wp = (gv2)(unsigned long long)(ell + 1152) + lane;
        const gip op = (gip)(unsigned long long)(ell + 1152 + 6 * 64 * 16) + 2 * lane;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            pw[s] = wp[s * 64];
            poA[s] = op[s * 128];
            poB[s] = op[s * 128 + 1];
        }
    }
    v4d gA[kHM];
    const v4d G0 = g2_img(hot_Gx, 0, lane);
#pragma unroll
    for (int u = 0; u < kHM; ++u) {
        const int k = u < m ? u : (m > 0 ? m - 1 : 0);
        gA[u] = g2_img(hot_Gx, m > 0 ? k + 1 : 0, lane);
    }
    // (what follows reads the argument block)
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double h = ft ? hv : opaque_scalar(P.dt_fixed);
    const bool dfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
    double mud[2] = {0.0, 0.0};
    if (dfast) {
#pragma unroll
        for (int d = 0; d < 2; ++d) mud[d] = mu[P.drow[d] + (lane < P.ddim_i[d] ? lane : 0)];
    }
    {   // the drives' rows into LDS: weight 0 / column 0 beyond the handle's drives
        ellTW[lane] = lane < 16 * m ? tw0 : 0.0;
        ellTC[lane] = lane < 16 * m ? tc0 : 0;
        if (lane < 32) {
            ellTW[64 + lane] = 64 + lane < 16 * m ? tw1 : 0.0;
            ellTC[64 + lane] = 64 + lane < 16 * m ? tc1 : 0;
        }
    }
    QC_STAMP(P, b, lane, 1);                                      // every load requested
    touch.consume();
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QC_STAMP(P, b, lane, 2);                                  // ... and back
    }
    fu_lds_order();                                               // (the drives' rows are in LDS)
    v4d Ga = G0;
#pragma unroll
    for (int u = 0; u < kHM; ++u) {
        const double a = (u < m) ? bcast_lane(av, u) : 0.0;
        Ga += a * gA[u];
    }
    const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h, hh2 = 0.5 * h;
    v4d Sc, Db, MD;                                               // c1 [S | S], [D | D], [M | c2 h^2 D]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Sc[r] = c1 * (u1[r] + u0[r]);
        Db[r] = u1[r] - u0[r];
        MD[r] = left ? mv[r] : hc2 * Db[r];
    }
    // ---- stage A.  The pair chains Q_p start from (2 c2 h G) [-N_2p | -N_2p+1], which needs the gathered tiles but not Y: Y, the Gram
    //      matrix and the three pair chains go to the matrix pipe together (18 independent-enough MFMAs), Y2 = G Y behind them.
    fu_put_rows(rMD, MD, g, j);
    fu_lds_order();
    v4d PNn[kPairs];                                              // [-N_2p | -N_2p+1], gathered directly (left lanes: drive 2p, right: 2p+1)
    const int row0 = (left ? 0 : 16) + g;                         // table row of (pair 0, register 0): + 32 per pair, + 4 per register
#pragma unroll
    for (int p2 = 0; p2 < kPairs; ++p2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 32 * p2 + 4 * r;
            PNn[p2][r] = __builtin_fma(ellTW[row], rMD[ellTC[row] + jj], 0.0);
        }
    }
    // Gram operands.  A = M: lane (g, i) reg kk = M[i][4 kk + g];  B = (c2 h^2 D)^T: lane (g, jc) reg kk = c2 h^2 D[jc][4 kk + g];  K = 8
    const double qa0 = rMD[j * kXS + g], qa1 = rMD[j * kXS + 4 + g];
    const double qb0 = rMD[j * kXS + 8 + g], qb1 = rMD[j * kXS + 12 + g];
    v4d Gs;
#pragma unroll
    for (int r = 0; r < 4; ++r) Gs[r] = c2h2 * Ga[r];
    v4d Y, Qg, Q[kPairs], Y2;
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[0], MD[0], zero, 0, 0, 0);        // (one accumulator chain each: the bits of the dense-image form)
#pragma unroll
    for (int p2 = 0; p2 < kPairs; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[0], PNn[p2][0], zero, 0, 0, 0);
    Qg = __builtin_amdgcn_mfma_f64_16x16x4f64(qa0, qb0, zero, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
        Y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], MD[kk], Y, 0, 0, 0);
#pragma unroll
        for (int p2 = 0; p2 < kPairs; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[kk], PNn[p2][kk], Q[p2], 0, 0, 0);
        if (kk == 1) Qg = __builtin_amdgcn_mfma_f64_16x16x4f64(qa1, qb1, Qg, 0, 0, 0);
    }
    QC_STAMP(P, b, lane, 3);                                      // stage A issued
    // ---- stage B: Y2 = G Y;  Q_p += G_2p [2 c2 h (-M1) | 0] + G_2p+1 [0 | 2 c2 h (-M1)] as row gathers ---------------------------------
    {
        v4d YL;
#pragma unroll
        for (int r = 0; r < 4; ++r) YL[r] = left ? c2h2 * Y[r] : 0.0;             // [2 c2 h (-M1) | 0]
        Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[0], Y[0], zero, 0, 0, 0);
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], Y[kk], Y2, 0, 0, 0);
        fu_put_rows(rYL, YL, g, j);
        fu_put_rows(rQ, Qg, g, j);
        fu_lds_order();
#pragma unroll
        for (int p2 = 0; p2 < kPairs; ++p2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 32 * p2 + 4 * r;
                Q[p2][r] = __builtin_fma(ellTW[row], rYL[ellTC[row] + jj], Q[p2][r]);
            }
        }
    }
    QC_STAMP(P, b, lane, 5);                                      // stage B issued, drive terms gathered
    // ---- (a, a), first half: this lane's row of the Gram matrix for its pairs, into LDS (the table's registers die here; the sums are made
    //      behind the matrix blocks' stores)
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const double t = pw[s][0] * rQ[poA[s]];
        rPart[s * 64 + lane] = __builtin_fma(pw[s][1], rQ[poB[s]], t);
    }
    // ---- matrix blocks, PAIR BY PAIR: combine, transpose the pair's two tiles through LDS, store -- the first stores leave as soon as the
    //      first pair is done, not behind the whole interval's arithmetic (at one device round the launch ends 14.7 MB of stores after
    //      the first one; stamped: first store 5.2 us after the kernel's entry with all seven tiles transposed at once).  A pair's tiles
    //      die with its stores; its share of the (a, h) sums stays behind in two registers.
    const unsigned lo = 8u * (16u * g + j);                        // lane (g, j), register r of a transposed tile: element j of column 4 r + g
    double pah[kPairs];
#pragma unroll
    for (int p2 = 0; p2 < kPairs; ++p2) {
        const int u = 2 * p2;
        pah[p2] = g2_dot4(Q[p2], Db) + g2_dot4(PNn[p2], Sc);
        v4d tin[2], tout[2];
        const v4d lin = hc1 * PNn[p2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tin[0][r] = __builtin_fma(-hh2, Q[p2][r], lin[r]);
            tin[1][r] = __builtin_fma(hh2, Q[p2][r], lin[r]);
        }
        lds_transpose16_multi<2>(tscr, tin, tout, g, j);
        if (u < m) {
            double* __restrict__ xb = Hb + P.ho_Ua;
            double* __restrict__ yb = Hb + P.ho_aU;
            g2_st_off<2>(xb, lo + 1024u * u, tout[0][0]);
            g2_st_off<2>(yb, lo + 1024u * u, tout[1][0]);
            g2_st_off<2>(xb, lo + 1024u * u + 512u, tout[0][1]);
            g2_st_off<2>(yb, lo + 1024u * u + 512u, tout[1][1]);
            if (u + 1 < m) {
                g2_st_off<2>(xb, lo + 1024u * (u + 1), tout[0][2]);
                g2_st_off<2>(yb, lo + 1024u * (u + 1), tout[1][2]);
                g2_st_off<2>(xb, lo + 1024u * (u + 1) + 512u, tout[0][3]);
                g2_st_off<2>(yb, lo + 1024u * (u + 1) + 512u, tout[1][3]);
            }
        }
        if (p2 == 0) QC_STAMP(P, b, lane, 6);                     // first pair's stores issued
    }
    const double phh = (2.0 * c2) * g2_dot4(Y2, Db);
    if (ft) {   // (U_t, h) | (h, U_t+1) = c1 Y -/+ 2 c2 h Y2
        const v4d ty = c1 * Y, ts = c2h2 * Y2;
        const v4d et = lds_transpose16(tscr, g2_sel(left, ty - ts, swap8(ty + ts)), g, j);
        double* __restrict__ eb = Hb + P.ho_Uh;
        double* __restrict__ fb = Hb + P.ho_hU;
        g2_st_off<2>(eb, lo, et[0]);
        g2_st_off<2>(eb, lo + 512u, et[1]);
        g2_st_off<2>(fb, lo, et[2]);
        g2_st_off<2>(fb, lo + 512u, et[3]);
    }
    QC_STAMP(P, b, lane, 7);                                      // matrix blocks' stores issued
    // ---- (a, a), second half: 16 partial sums per pair -------------------------------------------------------------------------------------
    {
        const int naa = m * (m + 1) / 2;
        const int q = lane < kAA ? lane : 0;
        const double* pp = rPart + (q >> 2) * 64 + (q & 3) * 16;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s0 += pp[4 * i];
            s1 += pp[4 * i + 1];
            s2 += pp[4 * i + 2];
            s3 += pp[4 * i + 3];
        }
        if (lane < naa) Hb[P.ho_aa + lane] = (s0 + s1) + (s2 + s3);
    }
    QC_STAMP(P, b, lane, 4);                                      // (a, a) stored
    // ---- (a, h), (h, h): rows of per-lane partial sums, reduced in the fixed order of the row-gather instantiation -------------------
    if (ft) {
#pragma unroll
        for (int p2 = 0; p2 < kPairs; ++p2) red[p2 * kG2Stride + lane] = pah[p2];
        red[kPairs * kG2Stride + lane] = phh;
        fu_lds_order();
        const int half = lane >> 5, row = lane & 31;
        const bool pair_row = row < kPairs, hh_row = row == kPairs;
        const int drive = 2 * row + half;
        const double* rp = red + (row <= kPairs ? row : 0) * kG2Stride + 8 * half;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0 += rp[16 * i] + rp[16 * i + 4];
            a1 += rp[16 * i + 1] + rp[16 * i + 5];
            a2 += rp[16 * i + 2] + rp[16 * i + 6];
            a3 += rp[16 * i + 3] + rp[16 * i + 7];
        }
        const double own = (a0 + a1) + (a2 + a3);
        if (pair_row && drive < m) Hb[P.ho_ah + drive] = own;
        if (hh_row && half == 0) Hb[P.ho_hh] = own;
    }
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
    QC_STAMP(P, b, lane, 8);                                      // every store issued
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QC_STAMP(P, b, lane, 9);                                  // drained
        QC_STAMP_FLUSH(P, b, lane, 0, 9);
    }
}

}  // namespace

// Appends this handle's PAIR TABLE to the blob of qc_mfma16_ell_build (qc_mfma_fused.hip: [6][16] weights, [6][16] columns x kXS =
// 1152 bytes): for slot s and lane l = (g, r) the pair q = g + 4 s = v (v + 1) / 2 + u (u <= v) and row r of
//     (a_u, a_v) = sum_r  w_u(r) w_v(c_u(r)) Q[r][c_v(c_u(r))]  +  w_v(r) w_u(c_v(r)) Q[r][c_u(c_v(r))],        Q = M (c2 h^2 D)^T
// as [6][64] (wA, wB) doubles, then [6][64] (offA, offB) ints -- offsets into the row-major LDS copy of Q (row stride kXS).
void qc_mfma16_ell_pair_table(const QcParams& P, std::vector<char>* blob) {
    const int m = P.m;
    const size_t base = blob->size();                              // 1152
    blob->resize(base + 6 * 64 * 16 + 6 * 64 * 8, 0);              // (before any pointer into it is taken)
    const double* tw = reinterpret_cast<const double*>(blob->data());
    const int* tc = reinterpret_cast<const int*>(blob->data() + 6 * 16 * 8);
    double* pw = reinterpret_cast<double*>(blob->data() + base);
    int* po = reinterpret_cast<int*>(blob->data() + base + 6 * 64 * 16);
    for (int s = 0; s < 6; ++s)
        for (int l = 0; l < 64; ++l) {
            const int q = (l >> 4) + 4 * s, r = l & 15;
            int v = 0;
            while ((v + 1) * (v + 2) / 2 <= q) ++v;
            const int u = q - v * (v + 1) / 2;
            double wA = 0.0, wB = 0.0;
            int oA = 0, oB = 0;
            if (v < m) {
                const int cu = tc[u * 16 + r] / kXS, cv = tc[v * 16 + r] / kXS;
                wA = tw[u * 16 + r] * tw[v * 16 + cu];
                oA = r * kXS + tc[v * 16 + cu] / kXS;
                wB = tw[v * 16 + r] * tw[u * 16 + cv];
                oB = r * kXS + tc[u * 16 + cv] / kXS;
            }
            pw[(s * 64 + l) * 2] = wA;
            pw[(s * 64 + l) * 2 + 1] = wB;
            po[(s * 64 + l) * 2] = oA;
            po[(s * 64 + l) * 2 + 1] = oB;
        }
}

bool qc_mfma16_hess_g2(const QcParams& P) {
    static const bool off = getenv("QC_HESS_G2") && atoi(getenv("QC_HESS_G2")) == 0;
    static const bool ell_off = getenv("QC_HESS_ELL") && atoi(getenv("QC_HESS_ELL")) == 0;
    return !off && !ell_off && P.ell16 != nullptr && P.antisym && P.n == 16 && P.nc == 8 && P.m >= 1 && P.m <= 6 && (P.stamps == nullptr || P.m > 4);
}

hipError_t qc_launch_mfma16_hess_g2(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    const double* Zt = dZ + P.t_begin * (long long)P.zdim;
    const double* mu0 = dMu + P.t_begin * P.F_stride + P.F_off;
#define QC_G2(HM_) hipLaunchKernelGGL((qc_mfma16_pade4_hess_g2_kernel<HM_>), dim3(P.n_int), dim3(64), 0, st, P.Gx, Zt, mu0, P.ell16, P.n_int, P.zdim, \
                                      P.off_a, P.off_dt, P.m, P.off_U, (int)P.F_stride, P, dH)
    if (P.stamps != nullptr && P.m > 4)       // diagnostic timeline (QC_STAMPS=1; profiles/stamps_hess1.py)
        hipLaunchKernelGGL((qc_mfma16_pade4_hess_g2_kernel<6, true>), dim3(P.n_int), dim3(64), 0, st, P.Gx, Zt, mu0, P.ell16, P.n_int, P.zdim, P.off_a, P.off_dt,
                           P.m, P.off_U, (int)P.F_stride, P, dH);
    else if (P.m <= 2) QC_G2(2);
    else if (P.m <= 4) QC_G2(4);
    else QC_G2(6);
#undef QC_G2
    return hipGetLastError();
}
