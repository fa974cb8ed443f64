// f64-MFMA Hessian-of-Lagrangian kernel, order-4 Pade, 2N = 16, up to 8 drives: ONE wavefront per
// interval, operands in registers (lane maps: qc_mfma_kernels.hip header), reductions through LDS.
//
// With M = reshape(mu_t[0:s], 16, 8), M1 = G^T M, M2 = G^T M1, N_k = G_k^T M, N'_k = G_k^T M1,
// N''_k = G^T N_k, V_k = G_k D  (h = dt, c1 = 1/2, c2 = 1/12):
//   (U_t,  a_k)    = -c1 h N_k - c2 h^2 (N''_k + N'_k)          (a_k, U_t+1) = -c1 h N_k + c2 h^2 (N''_k + N'_k)
//   (U_t,  h)      = -(c1 M1 + 2 c2 h M2)                        (h,  U_t+1) = -c1 M1 + 2 c2 h M2
//   (a_i, a_k)     = c2 h^2 (<N_i, V_k> + <N_k, V_i>)
//   (a_k, h)       = <N_k, -c1 S + 2 c2 h G D> + 2 c2 h <N'_k, D>
//   (h, h)         = 2 c2 <M1, G D>
//   (dx_i, h)      = -mu_i   (derivative integrators)
// Left multiplication by a transpose uses the B-layout image as the A operand
// (A-layout(X^T) = B-layout(X)); the B-layout tile of G is assembled from the B-layout images like the A-layout one, and
// the transposes for the stores go through LDS.  MFMAs per interval: 12 + 8 m + 4 ceil(m/2) (72 for m = 6).
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kHMmax = 8;                     // at most this many drives (held in registers)
constexpr int kHVals = kHMmax * (kHMmax + 1) / 2 + kHMmax + 1;
constexpr int kHStride = 65;                  // LDS row stride (doubles) of the reduction scratch

__device__ inline v4d load_img(const double* __restrict__ Gx, int mat, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(Gx) + mat * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline double dot4(const v4d& a, const v4d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ inline v4d sel(bool c, const v4d& a, const v4d& b) {
    return v4d{c ? a[0] : b[0], c ? a[1] : b[1], c ? a[2] : b[2], c ? a[3] : b[3]};
}

// ANTI: every generator is exactly antisymmetric (QcParams.antisym): B-layout(G_k) = A-layout(G_k^T) = -A-layout(G_k), so the
// transposed images are not loaded at all (half the L2 -> CU traffic of the interval's prologue, 48 registers fewer)
// ONCE: one interval per workgroup, no persistent loop (whose invariants the compiler hoists in front of the first load; see
// qc_mfma16_pade4_kernel)
template <int kHM, bool KET, bool BATCH, bool ANTI, bool ONCE = false>
__global__ __launch_bounds__(64) void qc_mfma16_pade4_hess_kernel(const QcParams Pk, const double* __restrict__ Z,
                                                                  const double* __restrict__ Mu, double* __restrict__ H,
                                                                  const QcParams* __restrict__ Pb) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    const QcParams& P = BATCH ? Pb[blockIdx.y] : Pk;      // BATCH: one launch for several handles (qc_mfma_kernels.hip)
    __shared__ double red[kHVals * kHStride];
    __shared__ double tscr[(kHM + 1) * 16 * 17];
    const int lane = threadIdx.x;
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ GxA = P.Gx;                          // A-layout images
    const double* __restrict__ GxB = P.Gx + (size_t)(m + 1) * 256;  // B-layout images (= A-layout of the transposes)
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    int vb = blockIdx.x;
    if (vb >= P.n_int) return;
    do {
        const int b = qc_xcd_remap(vb, P.n_int);
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;

        // ---- loads: knots, multipliers, generator images (one batch) ----------------------------------
        // K kets (nc < 8): the tile columns >= nc re-read column 0; the multipliers there are zeroed, which zeroes every
        // quantity derived from M in those columns (the scalar blocks sum over whole tiles), and they are never stored
        const int nc = KET ? P.nc : 8, jc = (!KET || jj < nc) ? jj : 0;     // KET = false: the masks fold away at compile time
        const int nr = KET ? P.n : 16;                                      // rows per column (N < 8 levels: zero-padded tile)
        v4d u0, u1, mraw;
        if constexpr (!KET) {
            const double* u0p = z0 + P.off_U + jc * 16 + g;
            const double* u1p = z1 + P.off_U + jc * 16 + g;
            const double* mp = mu + jc * 16 + g;
            u0 = v4d{u0p[0], u0p[4], u0p[8], u0p[12]};
            u1 = v4d{u1p[0], u1p[4], u1p[8], u1p[12]};
            mraw = v4d{mp[0], mp[4], mp[8], mp[12]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * r + g;
                const bool in = row < nr;
                u0[r] = in ? z0[P.off_U + jc * nr + row] : 0.0;
                u1[r] = in ? z1[P.off_U + jc * nr + row] : 0.0;
                mraw[r] = in ? mu[jc * nr + row] : 0.0;
            }
        }
        const v4d mv = (!KET || jj < nc) ? mraw : v4d{0.0, 0.0, 0.0, 0.0};
        const double h = ft ? load_uniform(z0 + P.off_dt) : P.dt_fixed;
        v4d gA[kHM], gB[kHM];
        double ak[kHM];
        v4d Ga = load_img(GxA, 0, lane);
        v4d Gb;                                             // B-layout of G = A-layout of G^T, assembled like Ga (no identity product)
        if constexpr (!ANTI) Gb = load_img(GxB, 0, lane);
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            const int k = u < m ? u : (m > 0 ? m - 1 : 0);
            gA[u] = load_img(GxA, m > 0 ? k + 1 : 0, lane);
            if constexpr (!ANTI) gB[u] = load_img(GxB, m > 0 ? k + 1 : 0, lane);
            ak[u] = (u < m) ? z0[P.off_a + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            Ga += ak[u] * gA[u];
            if constexpr (!ANTI) Gb += ak[u] * gB[u];
        }
        if constexpr (ANTI) {
            Gb = -Ga;
#pragma unroll
            for (int u = 0; u < kHM; ++u) gB[u] = -gA[u];
        }
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;

        // ---- products in five dependency stages, each a batch of independent 16x16x16 products whose MFMAs are
        //      interleaved (mm16_multi):  1: G_B   2: M1, [GS|GD]   3: M2, [N_k|N'_k], [V_k|.]   4: (U,h)^T, N''   5: (U,a)^T
        const v4d TM0 = sel(left, mv, zero);                // [M | 0]
        v4d W, Wsw;                                         // [S | D], [D | S]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double sm_ = u1[r] + u0[r], df = u1[r] - u0[r];
            W[r] = left ? sm_ : df;
            Wsw[r] = left ? df : sm_;
        }
        v4d Y1, P1sw;
        {
            v4d a2[2] = {Gb, Ga}, b2[2] = {TM0, W}, d2[2];
            mm16_multi<2>(a2, b2, d2);
            Y1 = d2[0];                                     // [M1 | 0]
            P1sw = swap8(d2[1]);                            // [GD | GS]
        }
        const v4d TM = sel(left, TM0, swap8(Y1));           // [M | M1]
        v4d Y2, NN[kHM], VV[kHM];                           // [M2 | 0], [N_k | N'_k], [V_k | .]
        {
            constexpr int N3 = 1 + 2 * kHM;
            v4d a3[N3], b3[N3], d3[N3];
            a3[0] = Gb;
            b3[0] = Y1;
#pragma unroll
            for (int u = 0; u < kHM; ++u) {                 // unused slots (u >= m) repeat the last drive; zeroed below
                a3[1 + u] = gB[u];
                b3[1 + u] = TM;
                a3[1 + kHM + u] = gA[u];
                b3[1 + kHM + u] = Wsw;
            }
            mm16_multi<N3>(a3, b3, d3);
            Y2 = d3[0];
#pragma unroll
            for (int u = 0; u < kHM; ++u) {
                NN[u] = (u < m) ? d3[1 + u] : zero;
                VV[u] = (u < m) ? d3[1 + kHM + u] : zero;
            }
        }
        v4d PN[kHM / 2], PN1[kHM / 2], PN2[kHM / 2];
        {
            constexpr int N4 = kHM / 2;
            v4d a4[N4], b4[N4];
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) {
                PN[p2] = sel(left, NN[2 * p2], swap8(NN[2 * p2 + 1]));      // [N_k | N_k+1]
                PN1[p2] = sel(left, swap8(NN[2 * p2]), NN[2 * p2 + 1]);     // [N'_k | N'_k+1]
                a4[p2] = Gb;
                b4[p2] = PN[p2];
            }
            mm16_multi<N4>(a4, b4, PN2);                                    // [N''_k | N''_k+1]
        }
        // transposes for the line-wide stores go through the padded LDS scratch (an identity product costs 4 MFMAs each), all
        // kHM + 1 tiles in ONE LDS round trip
        const v4d uh = -(c1 * Y1 + c2h2 * Y2), hu = (-c1) * Y1 + c2h2 * Y2;
        v4d ET, XT[kHM];                                    // (U_t, h) | (h, U_t+1);  (U_t, a) and (a, U_t+1) tiles, transposed
        {
            v4d tin[kHM + 1], tout[kHM + 1];
            tin[0] = sel(left, uh, swap8(hu));
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) {
                const v4d q = hc2 * (PN2[p2] + PN1[p2]), lin = (-hc1) * PN[p2];
                tin[1 + 2 * p2] = lin - q;
                tin[2 + 2 * p2] = lin + q;
            }
            lds_transpose16_multi<kHM + 1>(tscr, tin, tout, g, j);
            ET = tout[0];
#pragma unroll
            for (int u = 0; u < kHM; ++u) XT[u] = tout[1 + u];
        }
        if (ft) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;
                if (!KET || ((c & 7) < nc && j < nr)) qc_st8m<2>(Hb + (c < 8 ? P.ho_Uh + c * nr : P.ho_hU + (c - 8) * nr) + j, ET[r]);
            }
        }
#pragma unroll
        for (int u = 0; u < kHM; u += 2) {
            if (u < m) {
                const bool two = u + 1 < m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * r + g;           // tile columns < 8: drive u column c; >= 8: drive u+1 column c-8
                    if ((r < 2 || two) && (!KET || ((c & 7) < nc && j < nr))) {
                        const size_t o = (size_t)(u + (r < 2 ? 0 : 1)) * (KET ? P.s : 128) + (c & 7) * nr + j;
                        qc_st8m<2>(Hb + P.ho_Ua + o, XT[u][r]);
                        qc_st8m<2>(Hb + P.ho_aU + o, XT[u + 1][r]);
                    }
                }
            }
        }
        // ---- scalar blocks: per-lane partial sums, reduced through LDS ----------------------------------------
        {
            int idx = 0;
#pragma unroll
            for (int v = 0; v < kHM; ++v) {
#pragma unroll
                for (int u = 0; u <= v; ++u) {
                    if (v < m) red[(v * (v + 1) / 2 + u) * kHStride + lane] = left ? hc2 * (dot4(NN[u], VV[v]) + dot4(NN[v], VV[u])) : 0.0;
                }
            }
            idx = m * (m + 1) / 2;
            if (ft) {
                const v4d wl = (-c1) * W + c2h2 * P1sw;     // left: -c1 S + 2 c2 h GD
                const v4d wr = c2h2 * W;                    // right: 2 c2 h D
#pragma unroll
                for (int u = 0; u < kHM; ++u)
                    if (u < m) red[(idx + u) * kHStride + lane] = dot4(NN[u], left ? wl : wr);
                red[(idx + m) * kHStride + lane] = left ? 2.0 * c2 * dot4(Y1, P1sw) : 0.0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
            const int naa = m * (m + 1) / 2;
            const int nval = naa + (ft ? m + 1 : 0);
            if (lane < nval) {
                const double* row = red + lane * kHStride;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
                for (int c = 0; c < 64; c += 4) { a0 += row[c]; a1 += row[c + 1]; a2 += row[c + 2]; a3 += row[c + 3]; }
                const double sum = (a0 + a1) + (a2 + a3);
                const int o = lane < naa ? P.ho_aa + lane : (lane < naa + m ? P.ho_ah + (lane - naa) : P.ho_hh);
                Hb[o] = sum;
            }
        }
        qc_hess_tail(P, mu, Hb, lane, 64);   // derivative integrators: d2/d(dx_i) dh = -mu_i; alignment padding
        __builtin_amdgcn_wave_barrier();   // the scratch rows are rewritten by the next interval
    } while (!ONCE && (vb += gridDim.x) < P.n_int);
}

}  // namespace

bool qc_mfma_hess_supported(const QcParams& P) {
    if (qc_mfma32_hess_supported(P) || qc_mfma64_hess_supported(P) || qc_mfma16_padeP_hess_supported(P)) return true;
    return P.integrator == QC_PADE && P.p == 2 && P.n <= 16 && P.nc <= 8 && P.m <= kHMmax;
}

hipError_t qc_launch_mfma16_hess_batch(const QcParams& P0, const QcParams* dPb, int count, const double* dZ, const double* dMu, double* dH,
                                       hipStream_t st) {
    const int grid = P0.n_int < 4096 ? P0.n_int : 4096;
    // (the batched launch reads each handle's parameters from device memory; it keeps the general form)
#define QC_B(HM_)                                                                                                                      \
    do {                                                                                                                            \
        if (P0.nc != 8 || P0.n != 16)                                                                                               \
            hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM_, true, true, false>), dim3(grid, count), dim3(64), 0, st, P0, dZ, dMu, dH, dPb); \
        else                                                                                                                        \
            hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM_, false, true, false>), dim3(grid, count), dim3(64), 0, st, P0, dZ, dMu, dH, dPb); \
    } while (0)
    if (P0.m <= 2) QC_B(2); else if (P0.m <= 4) QC_B(4); else if (P0.m <= 6) QC_B(6); else QC_B(8);
#undef QC_B
    return hipGetLastError();
}

template <int HM, bool KET>
static void launch_hess16(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st, int grid) {
    if (grid == P.n_int) {   // one interval per workgroup: the loop-free instantiations
        if (P.antisym) hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM, KET, false, true, true>), dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH, nullptr);
        else hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM, KET, false, false, true>), dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH, nullptr);
        return;
    }
    if (P.antisym) hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM, KET, false, true>), dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH, nullptr);
    else hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM, KET, false, false>), dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH, nullptr);
}

hipError_t qc_launch_mfma_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    if (qc_mfma16_padeP_hess_supported(P)) return qc_launch_mfma16_padeP_hess(P, dZ, dMu, dH, st);
    if (P.n > 32) return qc_launch_mfma64_hess(P, dZ, dMu, dH, st);
    if (P.n > 16) return qc_launch_mfma32_hess(P, dZ, dMu, dH, st);
    const int grid = P.n_int < 4096 ? P.n_int : 4096;
    if (P.nc != 8 || P.n != 16) {
        if (P.m <= 2) launch_hess16<2, true>(P, dZ, dMu, dH, st, grid);
        else if (P.m <= 4) launch_hess16<4, true>(P, dZ, dMu, dH, st, grid);
        else if (P.m <= 6) launch_hess16<6, true>(P, dZ, dMu, dH, st, grid);
        else launch_hess16<8, true>(P, dZ, dMu, dH, st, grid);
        return hipGetLastError();
    }
    if (P.m <= 2) launch_hess16<2, false>(P, dZ, dMu, dH, st, grid);
    else if (P.m <= 4) launch_hess16<4, false>(P, dZ, dMu, dH, st, grid);
    else if (P.m <= 6) launch_hess16<6, false>(P, dZ, dMu, dH, st, grid);
    else launch_hess16<8, false>(P, dZ, dMu, dH, st, grid);
    return hipGetLastError();
}
