// LDS/VALU kernels: any Hilbert dimension N, any drive count, any even Pade order, both integrators.
// One 256-thread workgroup per interval (t, t+1); every small dense product is staged in LDS.
// These kernels serve the small configurations (1-2 qubits, odd dimensions such as qutrits, high
// Pade orders) and are the independent cross-check of the MFMA kernels (qc_mfma_kernels.hip).
//
// Mathematics: SURVEY.md Appendix A (A.2 coefficients, A.3 residual/Jacobian, A.4 Hessian), written
// for general order 2p as
//     delta     = D + sum_{k=1..p} c_k h^k G^k W_k,            W_k = D (k even), -S (k odd)
//     d/dh      = sum_k k c_k h^{k-1} G^k W_k
//     d/da_j    = sum_{i=0..p-1} G^i G_j Q_i,                   Q_i = sum_{k=i+1..p} c_k h^k G^{k-1-i} W_k
//     B, F      = sum_k (-+1)^k c_k h^k G^k
// with S = U_{t+1} + U_t, D = U_{t+1} - U_t, G = G_0 + sum_j a_j G_j, h = dt_t.
#include "qc_internal.h"

namespace {

constexpr int kThreadsLds = 256;     // workgroup size with the scratch in LDS
constexpr int kThreadsGws = 1024;    // ... and with the scratch in the global workspace (large systems, few intervals: use the whole CU)

struct LdsJacLayout {
    int z0, z1, Gp, PD, PS, Q, R, total;  // offsets in doubles
};

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

__host__ __device__ inline LdsJacLayout jac_layout(const QcParams& P) {
    LdsJacLayout L;
    const int n2 = P.n * P.n, nN = P.n * P.nc;
    const int p = P.p > 0 ? P.p : 1;
    int o = 0;
    L.z0 = o; o += even_up(P.zdim);
    L.z1 = o; o += even_up(P.zdim);
    L.Gp = o; o += p * n2;
    L.PD = o; o += (p + 1) * nN;
    L.PS = o; o += (p + 1) * nN;
    L.Q = o;  o += p * nN;
    L.R = o;  o += P.jchunk * p * nN;
    L.total = o;
    return L;
}

// C (n x ncol, col-major) = A (n x n, col-major) * X (n x ncol, col-major); all in LDS unless noted.
template <int NT>
__device__ inline void matmul_lds(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ X,
                                  int n, int ncol, int tid) {
    for (int idx = tid; idx < n * ncol; idx += NT) {
        const int r = idx % n, c = idx / n;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < n; ++k) acc = fma(A[r + n * k], X[k + n * c], acc);
        C[idx] = acc;
    }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// GWS: the per-interval scratch lives in a global-memory workspace instead of LDS (systems too large for 160 KB of LDS per
// interval: N >= ~18 levels).  Same code, one workgroup per interval; __syncthreads orders the workgroup's global accesses.
template <bool JAC, bool GWS>
__global__ __launch_bounds__(GWS ? kThreadsGws : kThreadsLds) void qc_lds_pade_kernel(const QcParams P, const double* __restrict__ Z,
                                                               double* __restrict__ F, double* __restrict__ J) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    constexpr int kThreads = GWS ? kThreadsGws : kThreadsLds;
    extern __shared__ __attribute__((aligned(16))) double lds_sm[];
    double* sm;
    if constexpr (GWS) sm = P.ws + (size_t)blockIdx.x * P.ws_stride; else sm = lds_sm;
    const int tid = threadIdx.x;
    const int b = qc_xcd_remap(blockIdx.x, gridDim.x);
    const long long t = P.t_begin + b;
    const int n = P.n, N = P.nc, s = P.s, m = P.m, p = P.p;
    const int n2 = n * n, nN = n * N;
    const LdsJacLayout L = jac_layout(P);
    double* z0 = sm + L.z0;
    double* z1 = sm + L.z1;
    double* Gp = sm + L.Gp;
    double* PD = sm + L.PD;
    double* PS = sm + L.PS;
    double* Q = sm + L.Q;
    double* R = sm + L.R;
    const bool ft = P.off_dt >= 0;

    const double* zt = Z + t * (long long)P.zdim;
    for (int i = tid; i < P.zdim; i += kThreads) {
        z0[i] = zt[i];
        z1[i] = zt[P.zdim + i];
    }
    __syncthreads();
    const double h = ft ? z0[P.off_dt] : P.dt_fixed;

    // G = G_0 + sum_j a_j G_j ; D ; -S
    for (int idx = tid; idx < n2; idx += kThreads) {
        double g = P.G[idx];
        for (int j = 0; j < m; ++j) g = fma(z0[P.off_a + j], P.G[(size_t)(j + 1) * n2 + idx], g);
        Gp[idx] = g;
    }
    for (int idx = tid; idx < nN; idx += kThreads) {
        const double u0 = z0[P.off_U + idx], u1 = z1[P.off_U + idx];
        PD[idx] = u1 - u0;
        PS[idx] = -(u1 + u0);
    }
    __syncthreads();
    // powers G^{k+1}, and the two Krylov streams G^k D, G^k (-S)
    for (int k = 1; k <= p; ++k) {
        if (k < p) matmul_lds<kThreads>(Gp + k * n2, Gp, Gp + (k - 1) * n2, n, n, tid);
        matmul_lds<kThreads>(PD + k * nN, Gp, PD + (k - 1) * nN, n, N, tid);
        matmul_lds<kThreads>(PS + k * nN, Gp, PS + (k - 1) * nN, n, N, tid);
        __syncthreads();
    }

    double* Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    double* Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;

    // residual and d/dh
    for (int idx = tid; idx < s; idx += kThreads) {
        double acc = PD[idx], dacc = 0.0, hk = 1.0;
        for (int k = 1; k <= p; ++k) {
            const double v = ((k & 1) ? PS : PD)[k * nN + idx];
            dacc = fma(P.c[k] * (double)k * hk, v, dacc);  // k c_k h^{k-1}
            hk *= h;
            acc = fma(P.c[k] * hk, v, acc);
        }
        if (Fb) Fb[idx] = acc;
        if (JAC && ft) Jb[P.jo_h + idx] = dacc;
    }
    // derivative integrators
    {
        int jo = P.jo_d;
        for (int d = 0; d < P.n_deriv; ++d) {
            const int dim = P.ddim_i[d], r0 = P.drow[d];
            for (int i = tid; i < dim; i += kThreads) {
                const double dx = z0[P.dx_off[d] + i];
                if (Fb) Fb[r0 + i] = z1[P.x_off[d] + i] - z0[P.x_off[d] + i] - h * dx;
                if (JAC) {
                    Jb[jo + i] = -1.0;
                    Jb[jo + dim + i] = 1.0;
                    Jb[jo + 2 * dim + i] = -h;
                    if (ft) Jb[jo + 3 * dim + i] = -dx;
                }
            }
            jo += (ft ? 4 : 3) * dim;
        }
    }
    if (!JAC) return;

    // -F and B, N copies each (I_N (x) .)
    for (int idx = tid; idx < n2; idx += kThreads) {
        const double diag = (idx % n == idx / n) ? 1.0 : 0.0;
        double fv = diag, bv = diag, hk = 1.0;
        for (int k = 1; k <= p; ++k) {
            hk *= h;
            const double g = P.c[k] * hk * Gp[(k - 1) * n2 + idx];
            fv += g;
            bv += (k & 1) ? -g : g;
        }
        for (int q = 0; q < N; ++q) {
            Jb[P.jo_F + q * n2 + idx] = -fv;
            Jb[P.jo_B + q * n2 + idx] = bv;
        }
    }
    // Q_i
    for (int idx = tid; idx < p * nN; idx += kThreads) {
        const int i = idx / nN, e = idx % nN;
        double acc = 0.0, hk = 1.0;
        for (int k = 1; k <= p; ++k) {
            hk *= h;
            if (k >= i + 1) acc = fma(P.c[k] * hk, ((k & 1) ? PS : PD)[(k - 1 - i) * nN + e], acc);
        }
        Q[idx] = acc;
    }
    __syncthreads();
    // d/da_j = sum_i G^i (G_j Q_i), jchunk drives per pass
    for (int j0 = 0; j0 < m; j0 += P.jchunk) {
        const int jc = min(P.jchunk, m - j0);
        for (int idx = tid; idx < jc * p * nN; idx += kThreads) {
            const int r = idx % n;
            const int c = (idx / n) % N;
            const int i = (idx / nN) % p;
            const int jj = idx / (nN * p);
            const double* __restrict__ Gj = P.G + (size_t)(j0 + jj + 1) * n2;
            const double* __restrict__ X = Q + i * nN + c * n;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < n; ++k) acc = fma(Gj[r + n * k], X[k], acc);
            R[idx] = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < jc * nN; idx += kThreads) {
            const int r = idx % n;
            const int c = (idx / n) % N;
            const int jj = idx / nN;
            const double* Rj = R + jj * p * nN;
            double acc = Rj[c * n + r];
            for (int i = 1; i < p; ++i) {
                const double* A = Gp + (i - 1) * n2;
                const double* X = Rj + i * nN + c * n;
#pragma unroll 8
                for (int k = 0; k < n; ++k) acc = fma(A[r + n * k], X[k], acc);
            }
            Jb[P.jo_a + (size_t)(j0 + jj) * s + c * n + r] = acc;
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
//  Hessian of the Lagrangian  mu_t^T delta_t  (written for general order 2p).
//  With M = reshape(mu_t[0:s], n, N), M_q = (G^T)^q M, P^(k)_q = G^q W_k (W_k = D for even k, -S odd):
//    (U1,h)   = sum_k (-1)^k k c_k h^{k-1} M_k            (U0,h) = - sum_k k c_k h^{k-1} M_k
//    (h,h)    = sum_k k(k-1) c_k h^{k-2} <M, P^(k)_k>
//    (U1,a_j) = sum_r (G^r)^T G_j^T YB_r,  YB_r = sum_{k>r} (-1)^k c_k h^k M_{k-1-r}
//    (U0,a_j) = -sum_r (G^r)^T G_j^T YF_r, YF_r = sum_{k>r}        c_k h^k M_{k-1-r}
//    (a_j,h)  = <Lam', G_j>,  Lam' = sum_i M_i Q'_i^T,  Q'_i = sum_{k>i} k c_k h^{k-1} P^(k)_{k-1-i}
//    (a_i,a_j)= S_ij + S_ji,  S_ij = sum_{k>=2} c_k h^k sum_{al=0..k-2} <A_i^al, C_j^(par(k), k-2-al)>
//               A_i^al = G_i^T M_al,   C_j^(s,0) = G_j P^s_0,  C_j^(s,t) = G C_j^(s,t-1) + G_j P^s_t
//    derivative integrators: d2/d(dx_i) dh = -mu_i
//  (U,U) blocks vanish.  Checked against complex-step differentiation in tests/test_oracle_math.py via the oracle.
// ------------------------------------------------------------------------------------------------
struct LdsHessLayout {
    int z0, z1, mu, Gp, PD, PS, Mq, YB, YF, Qp, Lam, S, chunk, total;
    int cj;  // drives per chunk
};

__host__ __device__ inline LdsHessLayout hess_layout(const QcParams& P, int cj) {
    LdsHessLayout L;
    const int n2 = P.n * P.n, nN = P.n * P.nc, p = P.p, m = P.m > 0 ? P.m : 1;
    int o = 0;
    L.z0 = o; o += even_up(P.zdim);
    L.z1 = o; o += even_up(P.zdim);
    L.mu = o; o += even_up(P.ddim);
    L.Gp = o; o += p * n2;
    L.PD = o; o += (p + 1) * nN;
    L.PS = o; o += (p + 1) * nN;
    L.Mq = o; o += (p + 1) * nN;
    L.YB = o; o += p * nN;
    L.YF = o; o += p * nN;
    L.Qp = o; o += p * nN;
    L.Lam = o; o += n2;
    L.S = o; o += even_up(m * m);
    L.cj = cj;
    // chunk space: max( RB+RF for cj drives , A for cj drives + C for cj drives )
    const int pm1 = p > 1 ? p - 1 : 1;
    const int a = 2 * cj * p * nN, b = cj * pm1 * nN + cj * 2 * pm1 * nN;
    L.chunk = o; o += a > b ? a : b;
    L.total = o;
    return L;
}

// C (n x ncol) = A^T (A is n x n col-major) * X
template <int NT>
__device__ inline void matmul_T_lds(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ X,
                                    int n, int ncol, int tid) {
    for (int idx = tid; idx < n * ncol; idx += NT) {
        const int r = idx % n, c = idx / n;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < n; ++k) acc = fma(A[k + n * r], X[k + n * c], acc);
        C[idx] = acc;
    }
}


template <bool GWS>
__global__ __launch_bounds__(GWS ? kThreadsGws : kThreadsLds) void qc_lds_pade_hess_kernel(const QcParams P, const double* __restrict__ Z,
                                                                    const double* __restrict__ Mu, double* __restrict__ H,
                                                                    int cj) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    constexpr int kThreads = GWS ? kThreadsGws : kThreadsLds;
    extern __shared__ __attribute__((aligned(16))) double lds_sm[];
    double* sm;
    if constexpr (GWS) sm = P.ws + (size_t)blockIdx.x * P.ws_stride; else sm = lds_sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = qc_xcd_remap(blockIdx.x, gridDim.x);
    const long long t = P.t_begin + b;
    const int n = P.n, N = P.nc, s = P.s, m = P.m, p = P.p;
    const int n2 = n * n, nN = n * N;
    const LdsHessLayout L = hess_layout(P, cj);
    double* z0 = sm + L.z0;
    double* z1 = sm + L.z1;
    double* mu = sm + L.mu;
    double* Gp = sm + L.Gp;
    double* PD = sm + L.PD;
    double* PS = sm + L.PS;
    double* Mq = sm + L.Mq;
    double* YB = sm + L.YB;
    double* YF = sm + L.YF;
    double* Qp = sm + L.Qp;
    double* Lam = sm + L.Lam;
    double* S = sm + L.S;
    double* CH = sm + L.chunk;
    const bool ft = P.off_dt >= 0;
    double* Hb = H + (size_t)b * P.H_stride + P.H_off;

    const double* zt = Z + t * (long long)P.zdim;
    const double* mut = Mu + t * P.F_stride + P.F_off;
    for (int i = tid; i < P.zdim; i += kThreads) { z0[i] = zt[i]; z1[i] = zt[P.zdim + i]; }
    for (int i = tid; i < P.ddim; i += kThreads) mu[i] = mut[i];
    for (int i = tid; i < m * m; i += kThreads) S[i] = 0.0;
    __syncthreads();
    const double h = ft ? z0[P.off_dt] : P.dt_fixed;

    for (int idx = tid; idx < n2; idx += kThreads) {
        double g = P.G[idx];
        for (int j = 0; j < m; ++j) g = fma(z0[P.off_a + j], P.G[(size_t)(j + 1) * n2 + idx], g);
        Gp[idx] = g;
    }
    for (int idx = tid; idx < nN; idx += kThreads) {
        const double u0 = z0[P.off_U + idx], u1 = z1[P.off_U + idx];
        PD[idx] = u1 - u0;
        PS[idx] = -(u1 + u0);
        Mq[idx] = mu[idx];
    }
    __syncthreads();
    for (int k = 1; k <= p; ++k) {
        if (k < p) matmul_lds<kThreads>(Gp + k * n2, Gp, Gp + (k - 1) * n2, n, n, tid);
        matmul_lds<kThreads>(PD + k * nN, Gp, PD + (k - 1) * nN, n, N, tid);
        matmul_lds<kThreads>(PS + k * nN, Gp, PS + (k - 1) * nN, n, N, tid);
        matmul_T_lds<kThreads>(Mq + k * nN, Gp, Mq + (k - 1) * nN, n, N, tid);
        __syncthreads();
    }

    // ---- elementwise blocks: (U,h), YB, YF, Q' ---------------------------------------------------------
    for (int idx = tid; idx < nN; idx += kThreads) {
        if (ft) {
            double vB = 0.0, vF = 0.0, hk1 = 1.0;      // h^{k-1}
            for (int k = 1; k <= p; ++k) {
                const double v = P.c[k] * (double)k * hk1 * Mq[k * nN + idx];
                vF += v;
                vB += (k & 1) ? -v : v;
                hk1 *= h;
            }
            Hb[P.ho_hU + idx] = vB;
            Hb[P.ho_Uh + idx] = -vF;
        }
        for (int r = 0; r < p; ++r) {
            double yb = 0.0, yf = 0.0, qp = 0.0, hk = 1.0, hk1 = 1.0;
            for (int k = 1; k <= p; ++k) {
                hk1 = hk;          // h^{k-1}
                hk *= h;           // h^k
                if (k >= r + 1) {
                    const double mv = P.c[k] * hk * Mq[(k - 1 - r) * nN + idx];
                    yf += mv;
                    yb += (k & 1) ? -mv : mv;
                    qp = fma(P.c[k] * (double)k * hk1, ((k & 1) ? PS : PD)[(k - 1 - r) * nN + idx], qp);
                }
            }
            YB[r * nN + idx] = yb;
            YF[r * nN + idx] = yf;
            Qp[r * nN + idx] = qp;
        }
    }
    qc_hess_tail(P, mu, Hb, tid, kThreads);   // derivative integrators: -mu; alignment padding
    __syncthreads();
    if (ft) {
        // (h,h): one wave reduces
        if (wave == 0) {
            double acc = 0.0;
            for (int e = lane; e < nN; e += 64) {
                double hk2 = 1.0;   // h^{k-2}
                for (int k = 2; k <= p; ++k) {
                    acc = fma(P.c[k] * (double)(k * (k - 1)) * hk2 * Mq[e], ((k & 1) ? PS : PD)[k * nN + e], acc);
                    hk2 *= h;
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) Hb[P.ho_hh] = acc;
        }
        // Lam' = sum_i M_i Q'_i^T  (n x n)
        for (int idx = tid; idx < n2; idx += kThreads) {
            const int r = idx % n, c = idx / n;
            double acc = 0.0;
            for (int i = 0; i < p; ++i)
                for (int col = 0; col < N; ++col) acc = fma(Mq[i * nN + col * n + r], Qp[i * nN + col * n + c], acc);
            Lam[idx] = acc;
        }
        __syncthreads();
        for (int j = wave; j < m; j += kThreads / 64) {       // (a_j, h) = <Lam', G_j>
            const double* Gj = P.G + (size_t)(j + 1) * n2;
            double acc = 0.0;
            for (int e = lane; e < n2; e += 64) acc = fma(Lam[e], Gj[e], acc);
            acc = wave_sum(acc);
            if (lane == 0) Hb[P.ho_ah + j] = acc;
        }
    }
    __syncthreads();

    // ---- (U, a_j) blocks, cj drives per pass ---------------------------------------------------------------
    double* RB = CH;
    double* RF = CH + cj * p * nN;
    for (int j0 = 0; j0 < m; j0 += cj) {
        const int jc = min(cj, m - j0);
        for (int idx = tid; idx < jc * p * nN; idx += kThreads) {
            const int r = idx % n, c = (idx / n) % N, rr = (idx / nN) % p, jj = idx / (nN * p);
            const double* __restrict__ Gj = P.G + (size_t)(j0 + jj + 1) * n2;     // G_j^T X: sum_k G_j[k][r] X[k][c]
            double ab = 0.0, af = 0.0;
#pragma unroll 8
            for (int k = 0; k < n; ++k) {
                const double gv = Gj[k + n * r];
                ab = fma(gv, YB[rr * nN + c * n + k], ab);
                af = fma(gv, YF[rr * nN + c * n + k], af);
            }
            RB[idx] = ab;
            RF[idx] = af;
        }
        __syncthreads();
        for (int idx = tid; idx < jc * nN; idx += kThreads) {
            const int r = idx % n, c = (idx / n) % N, jj = idx / nN;
            double ab = RB[jj * p * nN + c * n + r], af = RF[jj * p * nN + c * n + r];
            for (int rr = 1; rr < p; ++rr) {
                const double* A = Gp + (rr - 1) * n2;                               // (G^rr)^T
                const double* XB = RB + jj * p * nN + rr * nN + c * n;
                const double* XF = RF + jj * p * nN + rr * nN + c * n;
#pragma unroll 8
                for (int k = 0; k < n; ++k) {
                    const double gv = A[k + n * r];
                    ab = fma(gv, XB[k], ab);
                    af = fma(gv, XF[k], af);
                }
            }
            Hb[P.ho_aU + (size_t)(j0 + jj) * s + c * n + r] = ab;
            Hb[P.ho_Ua + (size_t)(j0 + jj) * s + c * n + r] = -af;
        }
        __syncthreads();
    }

    // ---- (a_i, a_j) ---------------------------------------------------------------------------------------
    if (p >= 2) {
        const int pm1 = p - 1;
        double* Ai = CH;                       // [ii][al][nN]
        double* Cj = CH + cj * pm1 * nN;       // [jj][par(2)][t][nN]
        for (int j0 = 0; j0 < m; j0 += cj) {
            const int jc = min(cj, m - j0);
            // C_j^(s,t), t = 0..p-2, sequential in t
            for (int tq = 0; tq < pm1; ++tq) {
                for (int idx = tid; idx < jc * 2 * nN; idx += kThreads) {
                    const int r = idx % n, c = (idx / n) % N, par = (idx / nN) & 1, jj = idx / (2 * nN);
                    const double* __restrict__ Gj = P.G + (size_t)(j0 + jj + 1) * n2;
                    const double* Pt = (par ? PS : PD) + tq * nN + c * n;
                    double acc = 0.0;
#pragma unroll 8
                    for (int k = 0; k < n; ++k) acc = fma(Gj[r + n * k], Pt[k], acc);
                    if (tq > 0) {
                        const double* Cp = Cj + ((jj * 2 + par) * pm1 + tq - 1) * nN + c * n;
#pragma unroll 8
                        for (int k = 0; k < n; ++k) acc = fma(Gp[r + n * k], Cp[k], acc);
                    }
                    Cj[((jj * 2 + par) * pm1 + tq) * nN + c * n + r] = acc;
                }
                __syncthreads();
            }
            for (int i0 = 0; i0 < m; i0 += cj) {
                const int ic = min(cj, m - i0);
                for (int idx = tid; idx < ic * pm1 * nN; idx += kThreads) {
                    const int r = idx % n, c = (idx / n) % N, al = (idx / nN) % pm1, ii = idx / (nN * pm1);
                    const double* __restrict__ Gi = P.G + (size_t)(i0 + ii + 1) * n2;
                    const double* Mx = Mq + al * nN + c * n;
                    double acc = 0.0;
#pragma unroll 8
                    for (int k = 0; k < n; ++k) acc = fma(Gi[k + n * r], Mx[k], acc);   // G_i^T M_al
                    Ai[idx] = acc;
                }
                __syncthreads();
                for (int pr = wave; pr < ic * jc; pr += kThreads / 64) {
                    const int ii = pr % ic, jj = pr / ic;
                    double acc = 0.0;
                    for (int e = lane; e < nN; e += 64) {
                        double hk = h;
                        for (int k = 2; k <= p; ++k) {
                            hk *= h;
                            const double ck = P.c[k] * hk;
                            const double* Cb = Cj + ((jj * 2 + (k & 1)) * pm1) * nN + e;
                            double tsum = 0.0;
                            for (int al = 0; al <= k - 2; ++al) tsum = fma(Ai[(ii * pm1 + al) * nN + e], Cb[(k - 2 - al) * nN], tsum);
                            acc = fma(ck, tsum, acc);
                        }
                    }
                    acc = wave_sum(acc);
                    if (lane == 0) S[(i0 + ii) * m + (j0 + jj)] = acc;
                }
                __syncthreads();
            }
        }
    }
    for (int idx = tid; idx < m * (m + 1) / 2; idx += kThreads) {
        int j = 0;
        while ((j + 1) * (j + 2) / 2 <= idx) ++j;
        const int i = idx - j * (j + 1) / 2;
        Hb[P.ho_aa + idx] = (p >= 2) ? S[i * m + j] + S[j * m + i] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
//  Exponential integrator (reference unitary_smooth_pulse_problem.jl:168-170, README.md:79):
//      delta = U1 - E U0,  E = exp(hG)
//      d/dU1 = I,  d/dU0 = -I_N (x) E,  d/dh = -G E U0,  d/da_j = -L_exp(hG; h G_j) U0      
//  E and the Frechet derivatives L_j come from ONE scaled Taylor polynomial (degree kExpDeg at
//  ||Y||_1 <= 1/4, truncation < 3e-18) differentiated term by term,
//      A_k = A_{k-1} Y / k,     D_k = (D_{k-1} Y + A_{k-1} E_j) / k,     Y = hG / 2^sq, E_j = h G_j / 2^sq
//  followed by sq squarings  L <- E L + L E,  E <- E E.  No linear solve, no branch on the data
//  except the squaring count, which is taken from ||hG||_1 (non-finite input: sq = 0, NaNs propagate).
// ------------------------------------------------------------------------------------------------
constexpr int kExpDeg = 12;

struct LdsExpLayout {
    int z0, z1, Gm, Y, A0, A1, E0, Ecur, Etmp, D0, D1, Ls, EU, red, total;
};

__host__ __device__ inline LdsExpLayout exp_layout(const QcParams& P, int cj) {
    LdsExpLayout L;
    const int n2 = P.n * P.n, nN = P.n * P.nc;
    int o = 0;
    L.z0 = o; o += even_up(P.zdim);
    L.z1 = o; o += even_up(P.zdim);
    L.Gm = o; o += n2;
    L.Y = o; o += n2;
    L.A0 = o; o += n2;
    L.A1 = o; o += n2;
    L.E0 = o; o += n2;
    L.Ecur = o; o += n2;
    L.Etmp = o; o += n2;
    L.D0 = o; o += cj * n2;
    L.D1 = o; o += cj * n2;
    L.Ls = o; o += cj * n2;
    L.EU = o; o += nN;
    L.red = o; o += 16;          // one slot per wave (up to 16 waves in the global-workspace mode)
    L.total = o;
    return L;
}

template <bool JAC, bool GWS>
__global__ __launch_bounds__(GWS ? kThreadsGws : kThreadsLds) void qc_lds_exp_kernel(const QcParams P, const double* __restrict__ Z,
                                                              double* __restrict__ F, double* __restrict__ J, int cj) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    constexpr int kThreads = GWS ? kThreadsGws : kThreadsLds;
    extern __shared__ __attribute__((aligned(16))) double lds_sm[];
    double* sm;
    if constexpr (GWS) sm = P.ws + (size_t)blockIdx.x * P.ws_stride; else sm = lds_sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = qc_xcd_remap(blockIdx.x, gridDim.x);
    const long long t = P.t_begin + b;
    const int n = P.n, N = P.nc, s = P.s, m = P.m;
    const int n2 = n * n, nN = n * N;
    const LdsExpLayout L = exp_layout(P, cj);
    double* z0 = sm + L.z0;
    double* z1 = sm + L.z1;
    double* Gm = sm + L.Gm;
    double* Y = sm + L.Y;
    double* Abuf[2] = {sm + L.A0, sm + L.A1};
    double* E0 = sm + L.E0;
    double* Ecur = sm + L.Ecur;
    double* Etmp = sm + L.Etmp;
    double* Dbuf[2] = {sm + L.D0, sm + L.D1};
    double* Ls = sm + L.Ls;
    double* EU = sm + L.EU;
    double* red = sm + L.red;
    const bool ft = P.off_dt >= 0;
    double* Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    double* Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;

    const double* zt = Z + t * (long long)P.zdim;
    for (int i = tid; i < P.zdim; i += kThreads) { z0[i] = zt[i]; z1[i] = zt[P.zdim + i]; }
    __syncthreads();
    const double h = ft ? z0[P.off_dt] : P.dt_fixed;
    for (int idx = tid; idx < n2; idx += kThreads) {
        double g = P.G[idx];
        for (int j = 0; j < m; ++j) g = fma(z0[P.off_a + j], P.G[(size_t)(j + 1) * n2 + idx], g);
        Gm[idx] = g;
    }
    __syncthreads();
    // ||hG||_1 = max column sum: wave w reduces columns w, w+4, ...
    {
        double best = 0.0;
        for (int c = wave; c < n; c += kThreads / 64) {
            double acc = 0.0;
            for (int r = lane; r < n; r += 64) acc += fabs(h * Gm[r + n * c]);
            acc = wave_sum(acc);
            best = fmax(best, acc);     // fmax drops NaN: handled below through isfinite(h*G) of the sum
            if (!(acc == acc)) best = acc;
        }
        if (lane == 0) red[wave] = best;
    }
    __syncthreads();
    int sq = 0;
    {
        double nrm = 0.0;
        bool bad = false;
        for (int w = 0; w < kThreads / 64; ++w) { const double v = red[w]; if (!(v == v) || v > 1e300) bad = true; nrm = fmax(nrm, v); }
        if (!bad && nrm > 0.25) {
            int e;
            (void)frexp(nrm / 0.25, &e);      // nrm/0.25 = f * 2^e, f in [0.5, 1)  ->  ceil(log2) <= e
            sq = e;
            if (ldexp(0.25, e - 1) >= nrm) sq = e - 1;
            if (sq < 0) sq = 0;
            if (sq > 60) sq = 60;
        }
    }
    const double sc = ldexp(1.0, -sq);
    for (int idx = tid; idx < n2; idx += kThreads) {
        Y[idx] = h * sc * Gm[idx];
        const double id = (idx % n == idx / n) ? 1.0 : 0.0;
        E0[idx] = id;
    }
    __syncthreads();

    const int nchunks = (JAC && m > 0) ? (m + cj - 1) / cj : 1;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int j0 = ch * cj;
        const int jc = (JAC && m > 0) ? min(cj, m - j0) : 0;
        // Taylor polynomial: A_0 = I, D_0 = 0
        for (int idx = tid; idx < n2; idx += kThreads) Abuf[0][idx] = (idx % n == idx / n) ? 1.0 : 0.0;
        for (int idx = tid; idx < jc * n2; idx += kThreads) { Dbuf[0][idx] = 0.0; Ls[idx] = 0.0; }
        __syncthreads();
        for (int k = 1; k <= kExpDeg; ++k) {
            const double* Ap = Abuf[(k - 1) & 1];
            double* An = Abuf[k & 1];
            const double* Dp = Dbuf[(k - 1) & 1];
            double* Dn = Dbuf[k & 1];
            const double inv = 1.0 / (double)k;
            for (int idx = tid; idx < n2; idx += kThreads) {
                const int r = idx % n, c = idx / n;
                double acc = 0.0;
                for (int q = 0; q < n; ++q) acc = fma(Ap[r + n * q], Y[q + n * c], acc);
                acc *= inv;
                An[idx] = acc;
                if (ch == 0) E0[idx] += acc;
            }
            for (int idx = tid; idx < jc * n2; idx += kThreads) {
                const int r = idx % n, c = (idx / n) % n, jj = idx / n2;
                const double* __restrict__ Gj = P.G + (size_t)(j0 + jj + 1) * n2;
                const double* Dj = Dp + jj * n2;
                double acc = 0.0, acc2 = 0.0;
                for (int q = 0; q < n; ++q) {
                    acc = fma(Dj[r + n * q], Y[q + n * c], acc);
                    acc2 = fma(Ap[r + n * q], Gj[q + n * c], acc2);
                }
                acc = (acc + h * sc * acc2) * inv;
                Dn[idx] = acc;
                Ls[idx] += acc;
            }
            __syncthreads();
        }
        // squarings
        for (int idx = tid; idx < n2; idx += kThreads) Ecur[idx] = E0[idx];
        __syncthreads();
        for (int q = 0; q < sq; ++q) {
            double* Ln = Dbuf[0];
            for (int idx = tid; idx < jc * n2; idx += kThreads) {
                const int r = idx % n, c = (idx / n) % n, jj = idx / n2;
                const double* Lj = Ls + jj * n2;
                double acc = 0.0;
#pragma unroll 8
                for (int k = 0; k < n; ++k) acc = fma(Ecur[r + n * k], Lj[k + n * c], fma(Lj[r + n * k], Ecur[k + n * c], acc));
                Ln[idx] = acc;
            }
            matmul_lds<kThreads>(Etmp, Ecur, Ecur, n, n, tid);
            __syncthreads();
            for (int idx = tid; idx < jc * n2; idx += kThreads) Ls[idx] = Ln[idx];
            for (int idx = tid; idx < n2; idx += kThreads) Ecur[idx] = Etmp[idx];
            __syncthreads();
        }
        // d/da_j = -L_j U0
        for (int idx = tid; idx < jc * nN; idx += kThreads) {
            const int r = idx % n, c = (idx / n) % N, jj = idx / nN;
            const double* Lj = Ls + jj * n2;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < n; ++k) acc = fma(Lj[r + n * k], z0[P.off_U + c * n + k], acc);
            Jb[P.jo_a + (size_t)(j0 + jj) * s + c * n + r] = -acc;
        }
        __syncthreads();
    }
    // E U0, residual, -I (x) E, identity block, d/dh
    for (int idx = tid; idx < nN; idx += kThreads) {
        const int r = idx % n, c = idx / n;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < n; ++k) acc = fma(Ecur[r + n * k], z0[P.off_U + c * n + k], acc);
        EU[idx] = acc;
        if (Fb) Fb[idx] = z1[P.off_U + idx] - acc;
        if (JAC) Jb[P.jo_B + idx] = 1.0;
    }
    {
        int jo = P.jo_d;
        for (int d = 0; d < P.n_deriv; ++d) {
            const int dim = P.ddim_i[d], r0 = P.drow[d];
            for (int i = tid; i < dim; i += kThreads) {
                const double dx = z0[P.dx_off[d] + i];
                if (Fb) Fb[r0 + i] = z1[P.x_off[d] + i] - z0[P.x_off[d] + i] - h * dx;
                if (JAC) {
                    Jb[jo + i] = -1.0;
                    Jb[jo + dim + i] = 1.0;
                    Jb[jo + 2 * dim + i] = -h;
                    if (ft) Jb[jo + 3 * dim + i] = -dx;
                }
            }
            jo += (ft ? 4 : 3) * dim;
        }
    }
    if (!JAC) return;
    for (int idx = tid; idx < n2; idx += kThreads) {
        const double v = -Ecur[idx];
        for (int q = 0; q < N; ++q) Jb[P.jo_F + q * n2 + idx] = v;
    }
    __syncthreads();
    if (ft) {
        for (int idx = tid; idx < nN; idx += kThreads) {
            const int r = idx % n, c = idx / n;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < n; ++k) acc = fma(Gm[r + n * k], EU[c * n + k], acc);
            Jb[P.jo_h + idx] = -acc;
        }
    }
}

}  // namespace

static int exp_chunk(const QcParams& P) {
    int cj = P.m > 0 ? P.m : 1;
    if (P.use_ws) return cj < 2 ? cj : 2;       // global workspace: small chunks keep the workspace small
    while (cj > 1 && (size_t)exp_layout(P, cj).total * sizeof(double) > 64 * 1024) cj = (cj + 1) / 2;
    return cj;
}

size_t qc_lds_bytes_jac(const QcParams& P) {
    if (P.integrator == QC_EXPONENTIAL) return (size_t)exp_layout(P, exp_chunk(P)).total * sizeof(double);
    return (size_t)jac_layout(P).total * sizeof(double);
}

template <typename K>
static hipError_t raise_lds_limit(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t qc_launch_lds_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, size_t lds, hipStream_t st) {
    const dim3 grid(P.n_int), block(P.use_ws ? kThreadsGws : kThreadsLds);
    if (P.use_ws) {   // scratch in the global workspace, no dynamic LDS
        if (P.integrator == QC_EXPONENTIAL) {
            const int cj = exp_chunk(P);
            if (dJ) hipLaunchKernelGGL((qc_lds_exp_kernel<true, true>), grid, block, 0, st, P, dZ, dF, dJ, cj);
            else hipLaunchKernelGGL((qc_lds_exp_kernel<false, true>), grid, block, 0, st, P, dZ, dF, dJ, cj);
        } else if (dJ) {
            hipLaunchKernelGGL((qc_lds_pade_kernel<true, true>), grid, block, 0, st, P, dZ, dF, dJ);
        } else {
            hipLaunchKernelGGL((qc_lds_pade_kernel<false, true>), grid, block, 0, st, P, dZ, dF, dJ);
        }
        return hipGetLastError();
    }
    if (P.integrator == QC_EXPONENTIAL) {
        const int cj = exp_chunk(P);
        hipError_t e = dJ ? raise_lds_limit(&qc_lds_exp_kernel<true, false>, lds) : raise_lds_limit(&qc_lds_exp_kernel<false, false>, lds);
        if (e != hipSuccess) return e;
        if (dJ) hipLaunchKernelGGL((qc_lds_exp_kernel<true, false>), grid, block, lds, st, P, dZ, dF, dJ, cj);
        else hipLaunchKernelGGL((qc_lds_exp_kernel<false, false>), grid, block, lds, st, P, dZ, dF, dJ, cj);
        return hipGetLastError();
    }
    if (dJ) {
        hipError_t e = raise_lds_limit(&qc_lds_pade_kernel<true, false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((qc_lds_pade_kernel<true, false>), grid, block, lds, st, P, dZ, dF, dJ);
    } else {
        hipError_t e = raise_lds_limit(&qc_lds_pade_kernel<false, false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((qc_lds_pade_kernel<false, false>), grid, block, lds, st, P, dZ, dF, dJ);
    }
    return hipGetLastError();
}

static int hess_chunk(const QcParams& P) {
    int cj = P.m > 0 ? P.m : 1;
    if (P.use_ws) return cj < 2 ? cj : 2;
    while (cj > 1 && (size_t)hess_layout(P, cj).total * sizeof(double) > 96 * 1024) cj = (cj + 1) / 2;
    return cj;
}

size_t qc_lds_bytes_hess(const QcParams& P) {
    if (P.integrator != QC_PADE) return qc_lds_exp_hess_bytes(P);
    return (size_t)hess_layout(P, hess_chunk(P)).total * sizeof(double);
}

hipError_t qc_launch_lds_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, size_t lds, hipStream_t st) {
    if (P.integrator != QC_PADE) return qc_launch_lds_exp_hess(P, dZ, dMu, dH, lds, st);
    const int cj = hess_chunk(P);
    if (P.use_ws) {
        hipLaunchKernelGGL(qc_lds_pade_hess_kernel<true>, dim3(P.n_int), dim3(kThreadsGws), 0, st, P, dZ, dMu, dH, cj);
        return hipGetLastError();
    }
    hipError_t e = raise_lds_limit(&qc_lds_pade_hess_kernel<false>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qc_lds_pade_hess_kernel<false>, dim3(P.n_int), dim3(kThreadsLds), lds, st, P, dZ, dMu, dH, cj);
    return hipGetLastError();
}
