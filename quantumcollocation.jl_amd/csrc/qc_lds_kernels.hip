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

constexpr int kThreads = 256;

struct LdsJacLayout {
    int z0, z1, Gp, PD, PS, Q, R, total;  // offsets in doubles
};

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

__host__ __device__ inline LdsJacLayout jac_layout(const QcParams& P) {
    LdsJacLayout L;
    const int n2 = P.n * P.n, nN = P.n * P.N;
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
__device__ inline void matmul_lds(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ X,
                                  int n, int ncol, int tid) {
    for (int idx = tid; idx < n * ncol; idx += kThreads) {
        const int r = idx % n, c = idx / n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(A[r + n * k], X[k + n * c], acc);
        C[idx] = acc;
    }
}

template <bool JAC>
__global__ __launch_bounds__(kThreads) void qc_lds_pade_kernel(const QcParams P, const double* __restrict__ Z,
                                                               double* __restrict__ F, double* __restrict__ J) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    const int b = qc_xcd_remap(blockIdx.x, gridDim.x);
    const long long t = P.t_begin + b;
    const int n = P.n, N = P.N, s = P.s, m = P.m, p = P.p;
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
        if (k < p) matmul_lds(Gp + k * n2, Gp, Gp + (k - 1) * n2, n, n, tid);
        matmul_lds(PD + k * nN, Gp, PD + (k - 1) * nN, n, N, tid);
        matmul_lds(PS + k * nN, Gp, PS + (k - 1) * nN, n, N, tid);
        __syncthreads();
    }

    double* Fb = F ? F + (size_t)b * P.ddim : nullptr;
    double* Jb = JAC ? J + (size_t)b * P.jac_nnz : nullptr;

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
        int r0 = s, jo = P.jo_d;
        for (int d = 0; d < P.n_deriv; ++d) {
            const int dim = P.ddim_i[d];
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
            r0 += dim;
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
                for (int k = 0; k < n; ++k) acc = fma(A[r + n * k], X[k], acc);
            }
            Jb[P.jo_a + (size_t)(j0 + jj) * s + c * n + r] = acc;
        }
        __syncthreads();
    }
}

}  // namespace

size_t qc_lds_bytes_jac(const QcParams& P) { return (size_t)jac_layout(P).total * sizeof(double); }

hipError_t qc_launch_lds_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, size_t lds, hipStream_t st) {
    if (P.integrator != QC_PADE) return hipErrorNotSupported;
    const dim3 grid(P.n_int), block(kThreads);
    if (dJ) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qc_lds_pade_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(qc_lds_pade_kernel<true>, grid, block, lds, st, P, dZ, dF, dJ);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qc_lds_pade_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(qc_lds_pade_kernel<false>, grid, block, lds, st, P, dZ, dF, dJ);
    }
    return hipGetLastError();
}

size_t qc_lds_bytes_hess(const QcParams& P) { (void)P; return 0; }

hipError_t qc_launch_lds_hess(const QcParams&, const double*, const double*, double*, size_t, hipStream_t) {
    return hipErrorNotSupported;
}
