// f64-MFMA kernel for the Hessian of the Lagrangian of the EXPONENTIAL integrator at 16 < 2N <= 32 (4 qubits; the density operator of
// 2 qubits: N^2 = 16 levels), up to 8 drives.  The 2N = 16 kernel (qc_mfma_exp_hess.hip) states the mathematics -- the reference solves
// `integrator=:exponential` problems with the Hessian left on, unitary_smooth_pulse_problem.jl:224-266,
// density_operator_smooth_pulse_problem.jl:68,104-106 --:
//     (U_t, a_j) = -L(X; h G_j)^T M      (U_t, h) = -(G E)^T M      (a_i, a_j) = -h <G_j^T, L2(X; V, h G_i)>
//     (a_j, h) = -( <G_j^T, E V> + <G^T W, L(X; h G_j)> )      (h, h) = -<G^T W, G E>          X = h G, E = exp(X), W = M U_t^T, V = W^T
// forward over reverse (one second-order chain per drive), scaling and squaring with ||Y||_1 <= 1/8, Taylor degree 10 in Horner form:
//     R_k = Y R_k+1 + I/(k-1)!      QV_k = V R_k+1 + Y QV_k+1      Q_k,j = G_j R_k+1 + Y Q_k+1,j      P_k,j = V Q_k+1,j + G_j QV_k+1 + Y P_k+1,j
//     squarings:  P_j <- E P_j + P_j E + LV L_j + L_j LV     L_j <- E L_j + L_j E     LV <- E LV + LV E     E <- E E
// Layout as qc_mfma32_exp.hip: every matrix 2 x 2 tiles of 16 x 16, one 512-thread workgroup per interval, wave k = drive k with its
// chains Q_j, P_j (8 tiles) and its generator's images in registers; the SHARED chains R and QV (and E, LV in the squarings) belong
// tile by tile to waves 0 - 3 and are published through double-buffered LDS blocks, in the operand layout and transposed (the left
// factors of the squarings); Y, V and the shared chains are read from LDS where a product needs them (a tile is 2 KB: 16 cycles of
// LDS against 256 of the matrix pipe).  One barrier per Horner step / squaring.  About 2570 MFMAs per drive wave at 4 squarings --
// 20 k per interval at eight drives: the matrix pipes' (qc_mfma32_exp.hip, F + dF: 7200).
#include <stdlib.h>

#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kXDeg = 10;           // the second derivative of the truncated series loses two orders: (1/8)^9 / 9! = 2e-14
constexpr double kXTh = 0.125;
constexpr int kXMmax = 8;
constexpr int kXThreads = 512;

__device__ inline v4d x_tile(const double* __restrict__ base, int tile, int lane) {   // [tile][pair][lane][2], global or LDS
    const v2d* p = reinterpret_cast<const v2d*>(base) + tile * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline v4d x_gtile(const double* base, int tile, int lane) { return load_image_tile(base + tile * 256, lane); }   // global (generator images)
__device__ inline void x_put(double* __restrict__ base, int tile, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + tile * 128 + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}
template <int CTRL>
__device__ inline double x_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double x_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ inline double x_sum64(double c) {     // sum over the 64 lanes, fixed order
    c += x_dpp<0x128>(c);
    c += x_dpp<0x124>(c);
    c += x_dpp<0x122>(c);
    c += x_dpp<0x121>(c);
    return (x_readlane(c, 0) + x_readlane(c, 16)) + (x_readlane(c, 32) + x_readlane(c, 48));
}
__device__ inline double x_dot4(const v4d& a, const v4d& b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }

// c += a * b for one 16 x 16 x 16 tile product (one accumulator chain)
__device__ __forceinline__ void x_mac(const v4d& a, const v4d& b, v4d& c) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], c, 0, 0, 0);
}
// acc[2I+J] += sum_K A(I, K) * B(K, J) for the four tiles, MFMAs interleaved over the tiles.  A tiles come from `A` (registers, index
// 2I+K) or from an LDS block `AL` at tile index ai(I, K); B tiles likewise.
template <typename FA, typename FB>
__device__ __forceinline__ void x_prod4(FA a_of, FB b_of, v4d (&acc)[4]) {
#pragma unroll
    for (int K = 0; K < 2; ++K) {
        const v4d a0 = a_of(0, K), a1 = a_of(1, K), b0 = b_of(K, 0), b1 = b_of(K, 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b0[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b1[kk], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b0[kk], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], acc[3], 0, 0, 0);
        }
    }
}

// ELL: every drive generator has at most ONE entry per row (Pauli strings; P.ell16 = qc_exp_ell_build's tables, qc_mfma_exp_hess.hip): the
// products G_j R and G_j QV of a Horner step are row gathers from row-major copies of R and QV that their owners publish next to the
// tiles -- 96 MFMAs per drive wave and step instead of 160.  fma(w, x, acc) per element: what the dense product adds besides exact zeros.
constexpr int kRS = 33;             // row stride of the row-major copies (doubles): the lanes of a gather fall on distinct banks
__device__ __forceinline__ void x_pin(v4d& x) {       // the tile's values in vector registers HERE (qc_mfma_exp_hess.hip, pin_v)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(x[r]));
}

template <bool ELL>
__global__ __launch_bounds__(kXThreads, 1) void qc_mfma32_exp_hess_kernel(const QcParams P, const double* __restrict__ Z, const double* __restrict__ Mu,
                                                                          double* __restrict__ H) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();
    __shared__ double RR[ELL ? 2 : 1][ELL ? 32 * kRS : 1];              // ELL: R and QV row-major (the gathers' sources), double-buffered
    __shared__ double QR[ELL ? 2 : 1][ELL ? 32 * kRS : 1];
    // tile index of a 32 x 32 matrix: 2 * (row block) + (column block)
    __shared__ __attribute__((aligned(16))) double GL[4 * 256];          // G (unscaled), A-layout tiles
    __shared__ __attribute__((aligned(16))) double YL[4 * 256];          // Y = (h / 2^sq) G, A-layout tiles
    __shared__ __attribute__((aligned(16))) double WL[4 * 256];          // W = M U^T, D-layout tiles: tile (K, I) read as an A operand acts as V(I, K)
    __shared__ __attribute__((aligned(16))) double VL[4 * 256];          // V = U M^T, D-layout tiles (the B operand of E V)
    __shared__ __attribute__((aligned(16))) double RL[2][4 * 256];       // shared chain R / E, D-layout, double-buffered
    __shared__ __attribute__((aligned(16))) double RT[2][4 * 256];       // ... transposed tile by tile (read as an A operand: acts as the tile)
    __shared__ __attribute__((aligned(16))) double QL[2][4 * 256];       // shared chain QV / LV
    __shared__ __attribute__((aligned(16))) double QT[2][4 * 256];
    __shared__ double TS[8 * 16 * 17];                                   // per-wave transpose scratch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15;
    const bool ft = P.off_dt >= 0;
    const bool drive = w < m;
    const bool owner = w < 4;
    const int oI = w >> 1, oJ = w & 1;                                    // the shared tile an owner wave computes
    const double* __restrict__ GxA = P.Gx;                               // A-layout images [mat][2I+K]
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    double* __restrict__ scr = TS + w * (16 * 17);

    const int b = qc_xcd_remap((int)blockIdx.x, P.n_int);
#ifdef QC_X32_STAMPS      // diagnostic variant build (profiles/stamps_exp32.py hess): wave 0 -> slots 0-7, wave 7 -> slots 8-15
    constexpr bool DIAG = true;
    QC_STAMP_DECL;
#define X32_STAMP(k) QC_STAMP(P, b, lane, k)
    X32_STAMP(0);
#else
#define X32_STAMP(k)
#endif
    const long long t = P.t_begin + b;
    const double* __restrict__ z0 = Z + t * (long long)P.zdim;
    const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);
    const int nc = P.nc, nr = P.n;

    // ---- loads: this wave's drive images, its half tile of the generator assembly; the owners: M and U_t in the A layout --------------
    v4d Gj[4];
    double tw[2][4];                                                     // ELL: weight and source offset (column x kRS + j) of rows 16 I + 4 r + g of this wave's drive
    int tc[2][4];
    if constexpr (ELL) {
        const double* __restrict__ bw = reinterpret_cast<const double*>(P.ell16);
        const int* __restrict__ bc = reinterpret_cast<const int*>(reinterpret_cast<const char*>(P.ell16) + kXMmax * 32 * 8);
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
        for (int q = 0; q < 4; ++q) Gj[q] = x_gtile(GxA + (size_t)kmat * 1024, q, lane);
        if (!drive) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Gj[q] = zero;                     // (idle chains stay zero)
        }
    }
    // B layout of M (rows in registers, column j; columns >= nc re-read column 0 and are never stored): the A operand of the blocks M^T (.)
    v4d bM[2];
    {
        const int jc = j < nc ? j : 0;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = 16 * I + 4 * r + g; bM[I][r] = row < nr ? mu[jc * nr + row] : 0.0; }
        }
    }
    {
        const v2d* __restrict__ ab = reinterpret_cast<const v2d*>(GxA) + (w >> 1) * 128 + (w & 1) * 64 + lane;
        v2d img[kXMmax + 1];
        double ak[kXMmax];
#pragma unroll
        for (int u = 0; u <= kXMmax; ++u) img[u] = ab[(size_t)(u <= m ? u : 0) * 512];
#pragma unroll
        for (int u = 0; u < kXMmax; ++u) ak[u] = z0[P.off_a + (u < m ? u : 0)];
        v2d Gh = img[0];
#pragma unroll
        for (int u = 0; u < kXMmax; ++u) Gh += (u < m ? ak[u] : 0.0) * img[u + 1];
        reinterpret_cast<v2d*>(GL)[(w >> 1) * 128 + (w & 1) * 64 + lane] = Gh;
    }
    if (owner) {   // tile (oI, oJ) of W = M U^T and of V = U M^T: A layout, lane (g, i) reg kk = X[16 I + i][4 kk + g], zero beyond nc / nr
        v4d aMI, aUJ, aUI, aMJ;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = 4 * kk + g;
            const bool ci = c < nc;
            const int rI = 16 * oI + j, rJ = 16 * oJ + j;
            const bool inI = ci && rI < nr, inJ = ci && rJ < nr;
            const double mI = mu[inI ? c * nr + rI : 0], uJ = z0[P.off_U + (inJ ? c * nr + rJ : 0)];
            const double uI = z0[P.off_U + (inI ? c * nr + rI : 0)], mJ = mu[inJ ? c * nr + rJ : 0];
            aMI[kk] = inI ? mI : 0.0;
            aUJ[kk] = inJ ? uJ : 0.0;
            aUI[kk] = inI ? uI : 0.0;
            aMJ[kk] = inJ ? mJ : 0.0;
        }
        x_put(WL, w, lane, mm16(aMI, aUJ));                               // W(I, J) = M_I U_J^T
        x_put(VL, w, lane, mm16(aUI, aMJ));                               // V(I, J) = U_I M_J^T
    }
    __syncthreads();

    // ---- ||h G||_1 (every wave, redundantly) -> squaring count; Y = (h / 2^sq) G into LDS -------------------------------------------
    int sq = 0;
    {
        v4d Gt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) Gt[q] = x_tile(GL, q, lane);
        double best = 0.0;
        bool bad = false;
#pragma unroll
        for (int K = 0; K < 2; ++K) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double c = fabs(h * Gt[K][kk]) + fabs(h * Gt[2 + K][kk]);
                c += x_dpp<0x128>(c);
                c += x_dpp<0x124>(c);
                c += x_dpp<0x122>(c);
                c += x_dpp<0x121>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = x_readlane(c, 16 * r);
                    if (!(v == v) || v > 1e300) bad = true;
                    best = fmax(best, v);
                }
            }
        }
        if (!bad && best > kXTh) {
            int e;
            (void)frexp(best / kXTh, &e);
            sq = e;
            if (ldexp(kXTh, e - 1) >= best) sq = e - 1;
            sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
        }
        if (owner) x_put(YL, w, lane, (h * ldexp(1.0, -sq)) * Gt[w]);
    }
    const double sc = ldexp(1.0, -sq), hs = h * sc;

    // ---- Horner: R_deg+1 = I/deg!, every derivative chain 0 -------------------------------------------------------------------------
    double fact = 1.0;
#pragma unroll
    for (int k = 2; k <= kXDeg; ++k) fact *= (double)k;
    double ck = 1.0 / fact;
    if (owner) {
        const v4d r0 = oI == oJ ? ck * IdB : zero;
        x_put(RL[0], w, lane, r0);
        x_put(RT[0], w, lane, r0);                                        // (a multiple of the identity: its own transpose)
        x_put(QL[0], w, lane, zero);
        x_put(QT[0], w, lane, zero);
        if constexpr (ELL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                RR[0][(16 * oI + 4 * r + g) * kRS + 16 * oJ + j] = r0[r];
                QR[0][(16 * oI + 4 * r + g) * kRS + 16 * oJ + j] = 0.0;
            }
        }
    }
    __syncthreads();                                                      // (YL and the chains' start)
    v4d Q[4] = {zero, zero, zero, zero}, Pm[4] = {zero, zero, zero, zero};
    int cur = 0;
    X32_STAMP(1);
    auto Yt = [&](int I, int K) { return x_tile(YL, 2 * I + K, lane); };  // A operand Y(I, K)
    auto Vt = [&](int I, int K) { return x_tile(WL, 2 * K + I, lane); };  // A operand V(I, K) = the D-layout tile W(K, I)
    auto Gt = [&](int I, int K) { return Gj[2 * I + K]; };
#pragma unroll 1
    for (int k = kXDeg; k >= 1; --k) {
        ck *= (double)k;                                                  // 1/(k-1)!
        const double* __restrict__ Rc = RL[cur];
        const double* __restrict__ Qc = QL[cur];
        auto Rb = [&](int K, int J) { return x_tile(Rc, 2 * K + J, lane); };
        auto QVb = [&](int K, int J) { return x_tile(Qc, 2 * K + J, lane); };
        if (owner) {   // tile (oI, oJ) of R_k = Y R_k+1 + ck I and of QV_k = V R_k+1 + Y QV_k+1
            v4d rn = oI == oJ ? ck * IdB : zero, qn = zero;
#pragma unroll
            for (int K = 0; K < 2; ++K) {
                const v4d rb = Rb(K, oJ);
                x_mac(Yt(oI, K), rb, rn);
                x_mac(Vt(oI, K), rb, qn);
                x_mac(Yt(oI, K), QVb(K, oJ), qn);
            }
            x_put(RL[cur ^ 1], w, lane, rn);
            x_put(QL[cur ^ 1], w, lane, qn);
            if constexpr (ELL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    RR[cur ^ 1][(16 * oI + 4 * r + g) * kRS + 16 * oJ + j] = rn[r];
                    QR[cur ^ 1][(16 * oI + 4 * r + g) * kRS + 16 * oJ + j] = qn[r];
                }
            }
            if (k == 1) {                                                 // the squarings (and the outputs) read the transposed tiles
                x_put(RT[cur ^ 1], w, lane, lds_transpose16(scr, rn, g, j));
                x_put(QT[cur ^ 1], w, lane, lds_transpose16(scr, qn, g, j));
            }
        }
        if constexpr (ELL) {
            if (drive) {   // two phases, each: its gathers requested, its products, the gathered terms added (qc_mfma_exp_hess.hip)
                {
                    const double* __restrict__ qr = QR[cur];
                    double x[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[t][r] = qr[tc[t >> 1][r] + 16 * (t & 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    v4d acc[4] = {zero, zero, zero, zero};
                    x_prod4(Vt, [&](int K, int J) { return Q[2 * K + J]; }, acc);      // V Q_j
                    x_prod4(Yt, [&](int K, int J) { return Pm[2 * K + J]; }, acc);     // + Y P_j
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Pm[t][r] = __builtin_fma(tw[t >> 1][r], x[t][r], acc[t][r]);      // + G_j QV
                        x_pin(Pm[t]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const double* __restrict__ rr = RR[cur];
                    double x[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[t][r] = rr[tc[t >> 1][r] + 16 * (t & 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    v4d acq[4] = {zero, zero, zero, zero};
                    x_prod4(Yt, [&](int K, int J) { return Q[2 * K + J]; }, acq);      // Y Q_j
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Q[t][r] = __builtin_fma(tw[t >> 1][r], x[t][r], acq[t][r]);       // + G_j R
                    }
                }
            }
        } else {
        if (drive) {
            v4d acc[4] = {zero, zero, zero, zero};
            x_prod4(Vt, [&](int K, int J) { return Q[2 * K + J]; }, acc);      // V Q_j
            x_prod4(Gt, QVb, acc);                                             // + G_j QV
            x_prod4(Yt, [&](int K, int J) { return Pm[2 * K + J]; }, acc);     // + Y P_j
#pragma unroll
            for (int q = 0; q < 4; ++q) Pm[q] = acc[q];
            v4d acq[4] = {zero, zero, zero, zero};
            x_prod4(Gt, Rb, acq);                                              // G_j R
            x_prod4(Yt, [&](int K, int J) { return Q[2 * K + J]; }, acq);      // + Y Q_j
#pragma unroll
            for (int q = 0; q < 4; ++q) Q[q] = acq[q];
        }
        }
        __syncthreads();
        cur ^= 1;
    }
    X32_STAMP(2);
    // ---- squarings ------------------------------------------------------------------------------------------------------------------
    for (int s = 0; s < sq; ++s) {
        const double* __restrict__ Rc = RL[cur];
        const double* __restrict__ Qc = QL[cur];
        const double* __restrict__ RTc = RT[cur];
        const double* __restrict__ QTc = QT[cur];
        auto Eb = [&](int K, int J) { return x_tile(Rc, 2 * K + J, lane); };       // B operand E(K, J)
        auto LVb = [&](int K, int J) { return x_tile(Qc, 2 * K + J, lane); };
        auto Ea = [&](int I, int K) { return x_tile(RTc, 2 * I + K, lane); };      // A operand acting as E(I, K)
        auto LVa = [&](int I, int K) { return x_tile(QTc, 2 * I + K, lane); };
        if (owner) {   // E <- E E, LV <- E LV + LV E
            v4d en = zero, ln = zero;
#pragma unroll
            for (int K = 0; K < 2; ++K) {
                const v4d ea = Ea(oI, K), eb = Eb(K, oJ);
                x_mac(ea, eb, en);
                x_mac(ea, LVb(K, oJ), ln);
                x_mac(LVa(oI, K), eb, ln);
            }
            x_put(RL[cur ^ 1], w, lane, en);
            x_put(QL[cur ^ 1], w, lane, ln);
            x_put(RT[cur ^ 1], w, lane, lds_transpose16(scr, en, g, j));
            x_put(QT[cur ^ 1], w, lane, lds_transpose16(scr, ln, g, j));
        }
        if (drive) {
            v4d accP[4] = {zero, zero, zero, zero}, accL[4] = {zero, zero, zero, zero};
            x_prod4(Ea, [&](int K, int J) { return Pm[2 * K + J]; }, accP);    // E P_j
            x_prod4(LVa, [&](int K, int J) { return Q[2 * K + J]; }, accP);    // + LV L_j
            x_prod4(Ea, [&](int K, int J) { return Q[2 * K + J]; }, accL);     // E L_j
            {
                v4d Lt[4];                                                    // (L_j(I, K))^T in the D layout: read as A, acts as L_j(I, K)
#pragma unroll
                for (int q = 0; q < 4; ++q) Lt[q] = lds_transpose16(scr, Q[q], g, j);
                auto La = [&](int I, int K) { return Lt[2 * I + K]; };
                x_prod4(La, LVb, accP);                                       // + L_j LV
                x_prod4(La, Eb, accL);                                        // + L_j E
            }
            {
                v4d Pt[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) Pt[q] = lds_transpose16(scr, Pm[q], g, j);
                x_prod4([&](int I, int K) { return Pt[2 * I + K]; }, Eb, accP);   // + P_j E
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { Pm[q] = accP[q]; Q[q] = accL[q]; }
        }
        __syncthreads();
        cur ^= 1;
    }
    X32_STAMP(3);
    // ---- shared outputs: E V, G^T W, G E (tile (oI, oJ) each) into the buffers the chains no longer use -----------------------------
    double* __restrict__ EVL = QL[cur ^ 1];
    double* __restrict__ T2L = QT[cur ^ 1];
    double* __restrict__ GEL = RL[cur ^ 1];
    if (owner) {
        v4d ev = zero, t2 = zero, ge = zero;
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            x_mac(x_tile(RT[cur], 2 * oI + K, lane), x_tile(VL, 2 * K + oJ, lane), ev);                     // E(I, K) V(K, J)
            x_mac(x_tile(GL, 2 * oI + K, lane), x_tile(RL[cur], 2 * K + oJ, lane), ge);                     // G(I, K) E(K, J)
            // (G^T)(I, K) = (G(K, I))^T: the A-layout tile of G(K, I) IS the D-layout tile of its transpose; transposed once more it is the
            // D-layout tile of G(K, I), which read as an A operand acts as (G(K, I))^T
            x_mac(lds_transpose16(scr, x_tile(GL, 2 * K + oI, lane), g, j), x_tile(WL, 2 * K + oJ, lane), t2);
        }
        x_put(EVL, w, lane, ev);
        x_put(T2L, w, lane, t2);
        x_put(GEL, w, lane, ge);
    }
    __syncthreads();
    X32_STAMP(4);
    // ---- this drive's blocks -------------------------------------------------------------------------------------------------------
    if (drive) {
        // (U_t, a_w) = -(h / 2^sq) L_w^T M, stored transposed: (M^T L_w)[c][16 J + j] = sum_K (M_K)^T L_w(K, J)
        double* __restrict__ pa = Hb + P.ho_Ua + (size_t)w * P.s;
#pragma unroll
        for (int Jt = 0; Jt < 2; ++Jt) {
            v4d x = zero;
            x_mac(bM[0], Q[Jt], x);
            x_mac(bM[1], Q[2 + Jt], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * r + g < nc && 16 * Jt + j < nr) qc_st8m<2>(pa + (4 * r + g) * nr + 16 * Jt + j, -hs * x[r]);
        }
        X32_STAMP(5);
        // (a_w, a_jd), jd >= w: -h (h / 4^sq) <G_jd^T, P_w>; the A-layout tile (J', K') of G_jd, lane for lane, IS the D-layout tile (K', J') of G_jd^T
        const double faa = -(h * hs * sc);
        for (int jd = w; jd < m; ++jd) {
            double v = 0.0;
#pragma unroll
            for (int Kp = 0; Kp < 2; ++Kp) {
#pragma unroll
                for (int Jp = 0; Jp < 2; ++Jp) v += x_dot4(x_gtile(GxA + (size_t)(jd + 1) * 1024, 2 * Jp + Kp, lane), Pm[2 * Kp + Jp]);
            }
            v = x_sum64(v);
            if (lane == 0) Hb[P.ho_aa + jd * (jd + 1) / 2 + w] = faa * v;
        }
        X32_STAMP(6);
        if (ft) {   // (a_w, h) = -( <G_w^T, E V> + (h / 2^sq) <G^T W, L_w> )
            double v = 0.0;
#pragma unroll
            for (int Kp = 0; Kp < 2; ++Kp) {
#pragma unroll
                for (int Jp = 0; Jp < 2; ++Jp) {
                    v += x_dot4(x_gtile(GxA + (size_t)(w + 1) * 1024, 2 * Jp + Kp, lane), x_tile(EVL, 2 * Kp + Jp, lane));
                    v += hs * x_dot4(x_tile(T2L, 2 * Kp + Jp, lane), Q[2 * Kp + Jp]);
                }
            }
            v = x_sum64(v);
            if (lane == 0) Hb[P.ho_ah + w] = -v;
        }
    }
    if (w == (m < 8 ? m : 7)) {
        // (an idle wave where there is one:) (U_t, h) = -(G E)^T M, transposed: M^T (G E);  (h, h) = -<G^T W, G E>
        if (ft) {
#pragma unroll
            for (int Jt = 0; Jt < 2; ++Jt) {
                v4d x = zero;
                x_mac(bM[0], x_tile(GEL, Jt, lane), x);
                x_mac(bM[1], x_tile(GEL, 2 + Jt, lane), x);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * r + g < nc && 16 * Jt + j < nr) qc_st8m<2>(Hb + P.ho_Uh + (4 * r + g) * nr + 16 * Jt + j, -x[r]);
            }
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += x_dot4(x_tile(T2L, q, lane), x_tile(GEL, q, lane));
            v = x_sum64(v);
            if (lane == 0) Hb[P.ho_hh] = -v;
        }
        qc_hess_tail(P, mu, Hb, lane, 64);
    }
#ifdef QC_X32_STAMPS
    X32_STAMP(7);
    if (P.stamps != nullptr && lane == 0 && (w == 0 || w == 7)) {
#pragma unroll
        for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + (w == 0 ? 0 : 8) + k_] = qc_ts_[k_];
    }
#endif
}

}  // namespace

bool qc_mfma32_exp_hess_supported(const QcParams& P) {
    return P.integrator == QC_EXPONENTIAL && P.n > 16 && P.n <= 32 && P.nc <= 16 && P.m <= kXMmax && P.hess_nnz > 0 && P.Gx != nullptr;
}

hipError_t qc_launch_mfma32_exp_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    static const bool ell_off = getenv("QC_EXP_ELL") && atoi(getenv("QC_EXP_ELL")) == 0;      // A/B diagnostics
    if (P.ell16 != nullptr && !ell_off) hipLaunchKernelGGL(qc_mfma32_exp_hess_kernel<true>, dim3(P.n_int), dim3(kXThreads), 0, st, P, dZ, dMu, dH);
    else hipLaunchKernelGGL(qc_mfma32_exp_hess_kernel<false>, dim3(P.n_int), dim3(kXThreads), 0, st, P, dZ, dMu, dH);
    return hipGetLastError();
}
