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
// (A-layout(X^T) = B-layout(X)); G_B = G I.  MFMAs per interval: 16 + 8 m + 12 ceil(m/2) + 4 (104 for m = 6).
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kHM = 8;                        // drives held in registers
constexpr int kHVals = kHM * (kHM + 1) / 2 + kHM + 1;
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

__global__ __launch_bounds__(64) void qc_mfma16_pade4_hess_kernel(const QcParams P, const double* __restrict__ Z,
                                                                  const double* __restrict__ Mu, double* __restrict__ H) {
    __shared__ double red[kHVals * kHStride];
    const int lane = threadIdx.x;
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const int mode = P.store_mode;
    const double* __restrict__ GxA = P.Gx;                          // A-layout images
    const double* __restrict__ GxB = P.Gx + (size_t)(m + 1) * 256;  // B-layout images (= A-layout of the transposes)
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    for (int vb = blockIdx.x; vb < P.n_int; vb += gridDim.x) {
        const int b = qc_xcd_remap(vb, P.n_int);
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;

        // ---- loads: knots, multipliers, generator images (one batch) ----------------------------------
        const double* u0p = z0 + P.off_U + jj * 16 + g;
        const double* u1p = z1 + P.off_U + jj * 16 + g;
        const double* mp = mu + jj * 16 + g;
        const v4d u0 = {u0p[0], u0p[4], u0p[8], u0p[12]};
        const v4d u1 = {u1p[0], u1p[4], u1p[8], u1p[12]};
        const v4d mv = {mp[0], mp[4], mp[8], mp[12]};
        const double h = ft ? z0[P.off_dt] : P.dt_fixed;
        v4d gA[kHM], gB[kHM];
        double ak[kHM];
        v4d Ga = load_img(GxA, 0, lane);
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            const int k = u < m ? u : (m > 0 ? m - 1 : 0);
            gA[u] = load_img(GxA, m > 0 ? k + 1 : 0, lane);
            gB[u] = load_img(GxB, m > 0 ? k + 1 : 0, lane);
            ak[u] = (u < m) ? z0[P.off_a + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kHM; ++u) Ga += ak[u] * gA[u];
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;

        // ---- shared products -----------------------------------------------------------------------------
        const v4d Gb = mm16(Ga, IdB);                       // B-layout of G = A-layout of G^T
        const v4d TM0 = sel(left, mv, zero);                // [M | 0]
        const v4d Y1 = mm16(Gb, TM0);                       // [M1 | 0]
        const v4d Y2 = mm16(Gb, Y1);                        // [M2 | 0]
        v4d W, Wsw;                                         // [S | D], [D | S]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double sm_ = u1[r] + u0[r], df = u1[r] - u0[r];
            W[r] = left ? sm_ : df;
            Wsw[r] = left ? df : sm_;
        }
        const v4d P1sw = swap8(mm16(Ga, W));                // [GD | GS]
        const v4d TM = sel(left, TM0, swap8(Y1));           // [M | M1]

        if (ft) {   // (U_t, h) on the left half, (h, U_t+1) on the right half, transposed store
            const v4d uh = -(c1 * Y1 + c2h2 * Y2), hu = (-c1) * Y1 + c2h2 * Y2;
            const v4d ET = mm16(sel(left, uh, swap8(hu)), IdB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;
                qc_st8(Hb + (c < 8 ? P.ho_Uh + c * 16 : P.ho_hU + (c - 8) * 16) + j, ET[r], mode);
            }
        }

        // ---- per drive: [N_k | N'_k] and [V_k | .] -------------------------------------------------------
        v4d NN[kHM], VV[kHM];
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            if (u < m) {
                NN[u] = mm16(gB[u], TM);
                VV[u] = mm16(gA[u], Wsw);
            } else {
                NN[u] = zero;
                VV[u] = zero;
            }
        }
        // ---- (U, a) blocks, two drives per tile ------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < kHM; u += 2) {
            if (u < m) {
                const bool two = u + 1 < m;
                const v4d PN = sel(left, NN[u], swap8(NN[u + 1]));      // [N_k | N_k+1]
                const v4d PN1 = sel(left, swap8(NN[u]), NN[u + 1]);     // [N'_k | N'_k+1]
                const v4d PN2 = mm16(Gb, PN);                           // [N''_k | N''_k+1]
                const v4d q = hc2 * (PN2 + PN1), lin = (-hc1) * PN;
                const v4d X0T = mm16(lin - q, IdB), X1T = mm16(lin + q, IdB);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r < 2 || two) {
                        qc_st8(Hb + P.ho_Ua + (size_t)u * 128 + (4 * r + g) * 16 + j, X0T[r], mode);
                        qc_st8(Hb + P.ho_aU + (size_t)u * 128 + (4 * r + g) * 16 + j, X1T[r], mode);
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
        if (ft) {   // derivative integrators: d2/d(dx_i) dh = -mu_i
            int r0 = P.s, o = P.ho_d;
            for (int d = 0; d < P.n_deriv; ++d) {
                for (int i = lane; i < P.ddim_i[d]; i += 64) Hb[o + i] = -mu[r0 + i];
                r0 += P.ddim_i[d];
                o += P.ddim_i[d];
            }
        }
        __builtin_amdgcn_wave_barrier();   // the scratch rows are rewritten by the next interval
    }
}

}  // namespace

bool qc_mfma_hess_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n == 16 && P.nc == P.N && P.m <= kHM;
}

hipError_t qc_launch_mfma_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    const int grid = P.n_int < 4096 ? P.n_int : 4096;
    hipLaunchKernelGGL(qc_mfma16_pade4_hess_kernel, dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH);
    return hipGetLastError();
}
