// Hessian of the Lagrangian for the EXPONENTIAL integrator, any Hilbert dimension, any drive count: one workgroup per interval,
// scratch in LDS (or in the global workspace when it does not fit), VALU arithmetic.  Serves what qc_mfma_exp_hess.hip does not
// and is its independent cross-check (QC_KERNEL_LDS).
//
// The reference builds `PiccoloOptions(integrator=:exponential)` problems and solves them with the Hessian left on
// (unitary_smooth_pulse_problem.jl:224-240,242-266; where it is not wanted the templates say `eval_hessian=false`,
// unitary_robustness_problem.jl:205,247): Ipopt asks such a problem for mu_d2F.
//
//     delta = U1 - E U0,   E = exp(X),  X = h G(a),  G(a) = G_0 + sum_j a_j G_j              (README.md:79)
// is linear in U1: every block of the Hessian that touches knot t+1 vanishes.  With M = reshape(mu, n, nc), W = M U0^T, V = W^T,
// L_j = L_exp(X; h G_j) (Frechet derivative), L2 the second Frechet derivative:
//     (U0, a_j) = -vec(L_j^T M)                     (U0, h) = -vec((G E)^T M) = -vec(E^T (G^T M))
//     (a_i, a_j) = -<W, L2(X; h G_i, h G_j)>        (a_j, h) = -<W, G_j E + G L_j>        (h, h) = -<M, G G E U0>
//     (dx_i, h) = -mu_i  (derivative integrators)
// FORWARD OVER REVERSE for the (a, a) block: <W, L2(X; A, B)> = <B^T, L2(X; V, A)> (cyclic invariance of the trace under the double
// integral that defines L2), so ONE second-order chain per drive -- directions (V, h G_j) -- serves every pair:
//     (a_i, a_j) = -h <G_i^T, P_j>,   P_j = L2(X; V, h G_j)
// instead of one chain per pair (m against m (m + 1) / 2).  All chains come from ONE scaled Taylor polynomial (degree kDeg at
// ||Y||_1 <= 1/4, Y = X / 2^sq) differentiated term by term,
//     A_k = A_k-1 Y / k            DV_k = (DV_k-1 Y + A_k-1 V') / k            D_k,j = (D_k-1,j Y + A_k-1 Y_j) / k
//     S_k,j = (S_k-1,j Y + DV_k-1 Y_j + D_k-1,j V') / k                         (V' = V / 2^sq, Y_j = h G_j / 2^sq)
// followed by sq squarings
//     P_j <- E P_j + P_j E + LV L_j + L_j LV      L_j <- E L_j + L_j E      LV <- E LV + LV E      E <- E E
// No linear solve; the only data-dependent branch is the squaring count (non-finite input: sq = 0, NaNs propagate).
#include "qc_internal.h"

namespace {

constexpr int kThreadsLds = 256;
constexpr int kThreadsGws = 1024;
constexpr int kDeg = 12;             // as qc_lds_exp_kernel: (1/4)^13 / 13! < 3e-18

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

struct ExpHessLayout {
    int z0, mu, GM, EU, X2, red, Y, Vs, A0, A1, DV0, DV1, E0, LV0, Ec, LVc, ch, total;   // ch: 6 matrices per drive of a chunk
};

__host__ __device__ inline ExpHessLayout layout(const QcParams& P, int cj) {
    ExpHessLayout L;
    const int n2 = P.n * P.n, nN = P.n * P.nc;
    int o = 0;
    L.z0 = o; o += even_up(P.zdim);
    L.mu = o; o += even_up(P.s);
    L.GM = o; o += nN;
    L.EU = o; o += nN;
    L.X2 = o; o += nN;
    L.red = o; o += 16;
    L.Y = o; o += n2;
    L.Vs = o; o += n2;
    L.A0 = o; o += n2;
    L.A1 = o; o += n2;
    L.DV0 = o; o += n2;
    L.DV1 = o; o += n2;
    L.E0 = o; o += n2;
    L.LV0 = o; o += n2;
    L.Ec = o; o += n2;
    L.LVc = o; o += n2;
    L.ch = o; o += 6 * cj * n2;
    L.total = o;
    return L;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// G(a)[r][c] from the constant generators and the amplitudes in z0 (recomputed where it is needed: m fused multiply-adds
// against n^2 doubles of scratch)
__device__ inline double g_of(const QcParams& P, const double* __restrict__ z0, int idx, int n2) {
    double g = P.G[idx];
    for (int j = 0; j < P.m; ++j) g = fma(z0[P.off_a + j], P.G[(size_t)(j + 1) * n2 + idx], g);
    return g;
}

template <bool GWS>
__global__ __launch_bounds__(GWS ? kThreadsGws : kThreadsLds) void qc_lds_exp_hess_kernel(const QcParams P, const double* __restrict__ Z,
                                                                                        const double* __restrict__ Mu, double* __restrict__ H,
                                                                                        int cj) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();
    constexpr int kThreads = GWS ? kThreadsGws : kThreadsLds;
    constexpr int kWaves = kThreads / 64;
    extern __shared__ __attribute__((aligned(16))) double lds_sm[];
    double* sm;
    if constexpr (GWS) sm = P.ws + (size_t)blockIdx.x * P.ws_stride; else sm = lds_sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = qc_xcd_remap(blockIdx.x, gridDim.x);
    const long long t = P.t_begin + b;
    const int n = P.n, N = P.nc, s = P.s, m = P.m;
    const int n2 = n * n, nN = n * N;
    const ExpHessLayout L = layout(P, cj);
    double* z0 = sm + L.z0;
    double* Mm = sm + L.mu;          // M = reshape(mu[0:s], n, nc)
    double* GM = sm + L.GM;
    double* EU = sm + L.EU;
    double* X2 = sm + L.X2;
    double* red = sm + L.red;
    double* Y = sm + L.Y;
    double* Vs = sm + L.Vs;
    double* Abuf[2] = {sm + L.A0, sm + L.A1};
    double* DVbuf[2] = {sm + L.DV0, sm + L.DV1};
    double* E0 = sm + L.E0;
    double* LV0 = sm + L.LV0;
    double* Ec = sm + L.Ec;
    double* LVc = sm + L.LVc;
    double* Dbuf[2] = {sm + L.ch, sm + L.ch + cj * n2};
    double* Ls = sm + L.ch + 2 * cj * n2;
    double* Sbuf[2] = {sm + L.ch + 3 * cj * n2, sm + L.ch + 4 * cj * n2};
    double* Ps = sm + L.ch + 5 * cj * n2;
    const bool ft = P.off_dt >= 0;
    double* Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double* zt = Z + t * (long long)P.zdim;
    const double* mut = Mu + t * P.F_stride + P.F_off;

    for (int i = tid; i < P.zdim; i += kThreads) z0[i] = zt[i];
    for (int i = tid; i < s; i += kThreads) Mm[i] = mut[i];
    __syncthreads();
    const double h = ft ? z0[P.off_dt] : P.dt_fixed;
    const double* U0 = z0 + P.off_U;

    // ||h G||_1 (largest column sum) -> squaring count
    {
        double best = 0.0;
        for (int c = wave; c < n; c += kWaves) {
            double acc = 0.0;
            for (int r = lane; r < n; r += 64) acc += fabs(h * g_of(P, z0, r + n * c, n2));
            acc = wave_sum(acc);
            best = fmax(best, acc);
            if (!(acc == acc)) best = acc;
        }
        if (lane == 0) red[wave] = best;
    }
    __syncthreads();
    int sq = 0;
    {
        double nrm = 0.0;
        bool bad = false;
        for (int w = 0; w < kWaves; ++w) { const double v = red[w]; if (!(v == v) || v > 1e300) bad = true; nrm = fmax(nrm, v); }
        if (!bad && nrm > 0.25) {
            int e;
            (void)frexp(nrm / 0.25, &e);
            sq = e;
            if (ldexp(0.25, e - 1) >= nrm) sq = e - 1;
            if (sq < 0) sq = 0;
            if (sq > 60) sq = 60;
        }
    }
    const double sc = ldexp(1.0, -sq), hs = h * sc;
    for (int idx = tid; idx < n2; idx += kThreads) {
        const int r = idx % n, c = idx / n;
        Y[idx] = hs * g_of(P, z0, idx, n2);
        double v = 0.0;                                  // V = U0 M^T, scaled like Y
        for (int k = 0; k < N; ++k) v = fma(U0[r + n * k], Mm[c + n * k], v);
        Vs[idx] = sc * v;
    }
    __syncthreads();

    const int nchunks = m > 0 ? (m + cj - 1) / cj : 1;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int j0 = ch * cj;
        const int jc = m > 0 ? min(cj, m - j0) : 0;
        // ---- Taylor polynomial and its first / second directional derivatives --------------------------------------------
        for (int idx = tid; idx < n2; idx += kThreads) {
            const double id = (idx % n == idx / n) ? 1.0 : 0.0;
            Abuf[0][idx] = id;
            DVbuf[0][idx] = 0.0;
            if (ch == 0) { E0[idx] = id; LV0[idx] = 0.0; }
        }
        for (int idx = tid; idx < jc * n2; idx += kThreads) { Dbuf[0][idx] = 0.0; Ls[idx] = 0.0; Sbuf[0][idx] = 0.0; Ps[idx] = 0.0; }
        __syncthreads();
        for (int k = 1; k <= kDeg; ++k) {
            const double* Ap = Abuf[(k - 1) & 1];
            double* An = Abuf[k & 1];
            const double* DVp = DVbuf[(k - 1) & 1];
            double* DVn = DVbuf[k & 1];
            const double* Dp = Dbuf[(k - 1) & 1];
            double* Dn = Dbuf[k & 1];
            const double* Sp = Sbuf[(k - 1) & 1];
            double* Sn = Sbuf[k & 1];
            const double inv = 1.0 / (double)k;
            for (int idx = tid; idx < n2; idx += kThreads) {
                const int r = idx % n, c = idx / n;
                double acc = 0.0, av = 0.0;
                for (int q = 0; q < n; ++q) {
                    const double ap = Ap[r + n * q];
                    acc = fma(ap, Y[q + n * c], acc);
                    av = fma(DVp[r + n * q], Y[q + n * c], fma(ap, Vs[q + n * c], av));
                }
                acc *= inv;
                av *= inv;
                An[idx] = acc;
                DVn[idx] = av;
                if (ch == 0) { E0[idx] += acc; LV0[idx] += av; }
            }
            for (int idx = tid; idx < jc * n2; idx += kThreads) {
                const int r = idx % n, c = (idx / n) % n, jj = idx / n2;
                const double* __restrict__ Gj = P.G + (size_t)(j0 + jj + 1) * n2;
                const double* Dj = Dp + jj * n2;
                const double* Sj = Sp + jj * n2;
                double d1 = 0.0, d2 = 0.0, s1 = 0.0, s2 = 0.0;
                for (int q = 0; q < n; ++q) {
                    const double gj = Gj[q + n * c], y = Y[q + n * c], dj = Dj[r + n * q];
                    d1 = fma(dj, y, d1);
                    d2 = fma(Ap[r + n * q], gj, d2);
                    s1 = fma(Sj[r + n * q], y, fma(dj, Vs[q + n * c], s1));
                    s2 = fma(DVp[r + n * q], gj, s2);
                }
                const double dn = (d1 + hs * d2) * inv, sn = (s1 + hs * s2) * inv;
                Dn[idx] = dn;
                Sn[idx] = sn;
                Ls[idx] += dn;
                Ps[idx] += sn;
            }
            __syncthreads();
        }
        // ---- squarings (from the sums of chunk 0's polynomial: E and LV evolve with the chunk's chains) ---------------------
        for (int idx = tid; idx < n2; idx += kThreads) { Ec[idx] = E0[idx]; LVc[idx] = LV0[idx]; }
        __syncthreads();
        for (int q = 0; q < sq; ++q) {
            double* En = Abuf[0];
            double* LVn = Abuf[1];
            double* Ln = Dbuf[0];
            double* Pn = Sbuf[0];
            for (int idx = tid; idx < jc * n2; idx += kThreads) {
                const int r = idx % n, c = (idx / n) % n, jj = idx / n2;
                const double* Lj = Ls + jj * n2;
                const double* Pj = Ps + jj * n2;
                double l = 0.0, p = 0.0;
                for (int k = 0; k < n; ++k) {
                    const double erk = Ec[r + n * k], ekc = Ec[k + n * c], lrk = Lj[r + n * k], lkc = Lj[k + n * c];
                    l = fma(erk, lkc, fma(lrk, ekc, l));
                    p = fma(erk, Pj[k + n * c], fma(Pj[r + n * k], ekc, p));
                    p = fma(LVc[r + n * k], lkc, fma(lrk, LVc[k + n * c], p));
                }
                Ln[idx] = l;
                Pn[idx] = p;
            }
            for (int idx = tid; idx < n2; idx += kThreads) {
                const int r = idx % n, c = idx / n;
                double e = 0.0, lv = 0.0;
                for (int k = 0; k < n; ++k) {
                    const double erk = Ec[r + n * k], ekc = Ec[k + n * c];
                    e = fma(erk, ekc, e);
                    lv = fma(erk, LVc[k + n * c], fma(LVc[r + n * k], ekc, lv));
                }
                En[idx] = e;
                LVn[idx] = lv;
            }
            __syncthreads();
            for (int idx = tid; idx < jc * n2; idx += kThreads) { Ls[idx] = Ln[idx]; Ps[idx] = Pn[idx]; }
            for (int idx = tid; idx < n2; idx += kThreads) { Ec[idx] = En[idx]; LVc[idx] = LVn[idx]; }
            __syncthreads();
        }
        // ---- this chunk's blocks ---------------------------------------------------------------------------------------------
        // (U0, a_j) = -L_j^T M
        for (int idx = tid; idx < jc * nN; idx += kThreads) {
            const int r = idx % n, c = (idx / n) % N, jj = idx / nN;
            const double* Lj = Ls + jj * n2;
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(Lj[k + n * r], Mm[k + n * c], acc);
            Hb[P.ho_Ua + (size_t)(j0 + jj) * s + c * n + r] = -acc;
        }
        // scalars, one wave each: (a_i, a_j) = -h <G_i^T, P_j> for i <= j;  (a_j, h) = -<W, G_j E + G L_j>  (W = V^T)
        for (int jj = 0; jj < jc; ++jj) {
            const int j = j0 + jj;
            const double* Pj = Ps + jj * n2;
            const double* Lj = Ls + jj * n2;
            const int nsc = j + 1 + (ft ? 1 : 0);
            for (int q = wave; q < nsc; q += kWaves) {
                double acc = 0.0;
                if (q <= j) {
                    const double* __restrict__ Gi = P.G + (size_t)(q + 1) * n2;
                    for (int idx = lane; idx < n2; idx += 64) {
                        const int r = idx % n, c = idx / n;
                        acc = fma(Gi[c + n * r], Pj[idx], acc);
                    }
                    acc = wave_sum(acc);
                    if (lane == 0) Hb[P.ho_aa + j * (j + 1) / 2 + q] = -h * acc;
                } else {
                    const double* __restrict__ Gj = P.G + (size_t)(j + 1) * n2;
                    for (int idx = lane; idx < n2; idx += 64) {
                        const int r = idx % n, c = idx / n;
                        double x = 0.0;
                        for (int k = 0; k < n; ++k) x = fma(Gj[r + n * k], Ec[k + n * c], fma(g_of(P, z0, r + n * k, n2), Lj[k + n * c], x));
                        acc = fma(Vs[c + n * r], x, acc);            // W[r][c] = V[c][r]
                    }
                    acc = wave_sum(acc);
                    if (lane == 0) Hb[P.ho_ah + j] = -acc / sc;      // Vs = V / 2^sq (a power of two: exact)
                }
            }
        }
        __syncthreads();
    }
    // ---- (U0, h) = -E^T (G^T M),  (h, h) = -<M, G G E U0>,  (dx, h) ------------------------------------------------------------
    if (ft) {
        for (int idx = tid; idx < nN; idx += kThreads) {
            const int r = idx % n, c = idx / n;
            double gm = 0.0, eu = 0.0;
            for (int k = 0; k < n; ++k) {
                gm = fma(g_of(P, z0, k + n * r, n2), Mm[k + n * c], gm);
                eu = fma(Ec[r + n * k], U0[k + n * c], eu);
            }
            GM[idx] = gm;
            EU[idx] = eu;
        }
        __syncthreads();
        for (int idx = tid; idx < nN; idx += kThreads) {
            const int r = idx % n, c = idx / n;
            double acc = 0.0, x = 0.0;
            for (int k = 0; k < n; ++k) {
                acc = fma(Ec[k + n * r], GM[k + n * c], acc);
                x = fma(g_of(P, z0, r + n * k, n2), EU[k + n * c], x);
            }
            Hb[P.ho_Uh + idx] = -acc;
            X2[idx] = x;                                  // G E U0
        }
        __syncthreads();
        double acc = 0.0;                                 // <G^T M, G E U0> = <M, G G E U0>
        for (int idx = tid; idx < nN; idx += kThreads) acc = fma(GM[idx], X2[idx], acc);
        acc = wave_sum(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < kWaves; ++w) tot += red[w];
            Hb[P.ho_hh] = -tot;
        }
    }
    qc_hess_tail(P, mut, Hb, tid, kThreads);
}

template <typename K>
static hipError_t raise_lds_limit(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

int chunk_of(const QcParams& P) {
    int cj = P.m > 0 ? P.m : 1;
    if (P.use_ws) return cj < 2 ? cj : 2;
    while (cj > 1 && (size_t)layout(P, cj).total * sizeof(double) > 64 * 1024) cj = (cj + 1) / 2;
    return cj;
}

}  // namespace

size_t qc_lds_exp_hess_bytes(const QcParams& P) { return (size_t)layout(P, chunk_of(P)).total * sizeof(double); }

hipError_t qc_launch_lds_exp_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, size_t lds, hipStream_t st) {
    const int cj = chunk_of(P);
    if (P.use_ws) {
        hipLaunchKernelGGL(qc_lds_exp_hess_kernel<true>, dim3(P.n_int), dim3(kThreadsGws), 0, st, P, dZ, dMu, dH, cj);
        return hipGetLastError();
    }
    hipError_t e = raise_lds_limit(&qc_lds_exp_hess_kernel<false>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qc_lds_exp_hess_kernel<false>, dim3(P.n_int), dim3(kThreadsLds), lds, st, P, dZ, dMu, dH, cj);
    return hipGetLastError();
}
