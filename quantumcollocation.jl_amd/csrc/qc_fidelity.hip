// Fidelity of the final knot: value, gradient and dense upper-triangular Hessian of
//     F(U~) = |tr(U_goal' U)| / n   over a subspace block          (iso_vec_unitary_fidelity,
//     l(U~) = |1 - F(U~)|                                           unitary_minimum_time_problem.jl:77;
// the loss of UnitaryInfidelityObjective, docstring unitary_smooth_pulse_problem.jl:23-28).
// tr = g_r.u + i g_i.u with constant vectors g_r, g_i built from the goal, so
//     grad F = (t_r g_r + t_i g_i) / (n^2 F),   hess F = (g_r g_r^T + g_i g_i^T) / (n^2 F) - grad F grad F^T / F.
// One 256-thread workgroup; this is a few hundred FLOPs on 128..512 numbers: it exists so that a
// device-resident consumer needs no host round trip for the last knot, not for speed.
#include <string>
#include <vector>

#include "qc_internal.h"

struct qc_fidelity {
    int N = 0, s = 0, n_sub = 0, device = 0, kind = QC_FID_UNITARY;
    double *dgr = nullptr, *dgi = nullptr, *dU = nullptr, *dOut = nullptr;   // dOut: [value(2: F, l) | gradF (s) | hessF (s(s+1)/2)]
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {

__global__ __launch_bounds__(256) void qc_fidelity_kernel(const double* __restrict__ u, const double* __restrict__ gr,
                                                          const double* __restrict__ gi, int s, int n_sub, int kind,
                                                          double* __restrict__ val, double* __restrict__ grad,
                                                          double* __restrict__ hess) {
    __shared__ double red[2][4];
    __shared__ double sg[2048];   // g_r, g_i staged (s <= 1024 handled through global otherwise)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double ar = 0.0, ai = 0.0;
    for (int i = tid; i < s; i += 256) {
        const double ui = u[i];
        ar = fma(gr[i], ui, ar);
        ai = fma(gi[i], ui, ai);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ar += __shfl_xor(ar, off, 64);
        ai += __shfl_xor(ai, off, 64);
    }
    if (lane == 0) { red[0][wave] = ar; red[1][wave] = ai; }
    __syncthreads();
    const double tr = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double ti = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double n = (double)n_sub;
    // unitary: F = |t| / n;   ket: F = |t|^2 (iso_fidelity);   density operator against a pure goal: F = Re t = psi' rho psi
    const double Fv = kind == QC_FID_UNITARY ? sqrt(tr * tr + ti * ti) / n : (kind == QC_FID_KET ? tr * tr + ti * ti : tr);
    const double inv = 1.0 / (n * n * Fv);
    if (tid == 0) {
        val[0] = Fv;
        val[1] = fabs(1.0 - Fv);
    }
    if (kind != QC_FID_UNITARY) {
        for (int i = tid; i < s; i += 256) {
            if (grad) grad[i] = kind == QC_FID_KET ? 2.0 * (tr * gr[i] + ti * gi[i]) : gr[i];
        }
        if (!hess) return;
        const long long nh2 = (long long)s * (s + 1) / 2;
        for (long long e = tid; e < nh2; e += 256) {
            int j = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while ((long long)(j + 1) * (j + 2) / 2 <= e) ++j;
            while ((long long)j * (j + 1) / 2 > e) --j;
            const int i = (int)(e - (long long)j * (j + 1) / 2);
            hess[e] = kind == QC_FID_KET ? 2.0 * (gr[i] * gr[j] + gi[i] * gi[j]) : 0.0;
        }
        return;
    }
    const bool stage = s <= 1024;
    for (int i = tid; i < s; i += 256) {
        const double gv = (tr * gr[i] + ti * gi[i]) * inv;
        if (grad) grad[i] = gv;
        if (stage) { sg[i] = gr[i]; sg[1024 + i] = gi[i]; }
    }
    if (!hess) return;
    __syncthreads();
    const long long nh = (long long)s * (s + 1) / 2;
    for (long long e = tid; e < nh; e += 256) {
        // upper triangle, column-major: e = j (j+1)/2 + i, i <= j
        int j = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while ((long long)(j + 1) * (j + 2) / 2 <= e) ++j;
        while ((long long)j * (j + 1) / 2 > e) --j;
        const int i = (int)(e - (long long)j * (j + 1) / 2);
        const double gri = stage ? sg[i] : gr[i], grj = stage ? sg[j] : gr[j];
        const double gii = stage ? sg[1024 + i] : gi[i], gij = stage ? sg[1024 + j] : gi[j];
        const double dFi = (tr * gri + ti * gii) * inv, dFj = (tr * grj + ti * gij) * inv;
        hess[e] = (gri * grj + gii * gij) * inv - dFi * dFj / Fv;
    }
}

thread_local std::string g_ferr;
int ffail(qc_fidelity* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_ferr = msg;
    return code;
}

}  // namespace

#define QCF_HIP(h, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return ffail(h, QC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* qc_fidelity_last_error(const qc_fidelity* h) { return h ? h->err.c_str() : g_ferr.c_str(); }

extern "C" int qc_fidelity_create(int32_t N, const double* goal_iso, const int32_t* subspace, int32_t n_sub, int32_t device,
                                  qc_fidelity** out) {
    if (!out) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: out is NULL");
    *out = nullptr;
    if (N < 1 || N > 64 || !goal_iso) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: bad N or goal");
    if (subspace && (n_sub < 1 || n_sub > N)) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: bad subspace size");
    std::vector<int> sub;
    if (subspace) {
        for (int k = 0; k < n_sub; ++k) {
            if (subspace[k] < 0 || subspace[k] >= N) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: subspace index out of range");
            sub.push_back(subspace[k]);
        }
    } else {
        for (int k = 0; k < N; ++k) sub.push_back(k);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ffail(nullptr, QC_ERR_NO_DEVICE, "qc_fidelity_create: no HIP device visible");
    if (device < 0 || device >= ndev) return ffail(nullptr, QC_ERR_NO_DEVICE, "qc_fidelity_create: device ordinal out of range");
    qc_fidelity* h = new qc_fidelity();
    h->N = N;
    h->s = 2 * N * N;
    h->n_sub = (int)sub.size();
    h->device = device;
    std::vector<double> gr(h->s, 0.0), gi(h->s, 0.0);
    for (int j : sub)
        for (int i : sub) {
            const int re = j * 2 * N + i, im = j * 2 * N + N + i;
            const double Gre = goal_iso[re], Gim = goal_iso[im];
            gr[re] = Gre;  gr[im] = Gim;
            gi[re] = -Gim; gi[im] = Gre;
        }
    auto bail = [&](hipError_t e, const char* what) { std::string m = std::string(what) + ": " + hipGetErrorString(e); delete h; return ffail(nullptr, QC_ERR_HIP, m); };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    const size_t nout = 2 + (size_t)h->s + (size_t)h->s * (h->s + 1) / 2;
    if ((e = hipMalloc((void**)&h->dgr, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dgi, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dU, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dOut, nout * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpy(h->dgr, gr.data(), h->s * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipMemcpy(h->dgi, gi.data(), h->s * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    *out = h;
    return QC_OK;
}

// Ket and density-operator fidelities share the handle: only the constant vectors g_r, g_i and the formula differ.
extern "C" int qc_fidelity_create_kind(int32_t kind, int32_t N, const double* goal_ket_iso, int32_t device, qc_fidelity** out) {
    if (!out) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_kind: out is NULL");
    *out = nullptr;
    if (kind != QC_FID_KET && kind != QC_FID_DENSITY) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_kind: kind must be QC_FID_KET or QC_FID_DENSITY");
    if (N < 1 || N > 64 || !goal_ket_iso) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_kind: bad N or goal");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ffail(nullptr, QC_ERR_NO_DEVICE, "qc_fidelity_create_kind: no HIP device visible");
    if (device < 0 || device >= ndev) return ffail(nullptr, QC_ERR_NO_DEVICE, "qc_fidelity_create_kind: device ordinal out of range");
    qc_fidelity* h = new qc_fidelity();
    h->N = N;
    h->kind = kind;
    h->n_sub = 1;
    h->device = device;
    h->s = kind == QC_FID_KET ? 2 * N : 2 * N * N;
    std::vector<double> gr(h->s, 0.0), gi(h->s, 0.0);
    const double* gre = goal_ket_iso;        // [Re psi_goal; Im psi_goal]
    const double* gim = goal_ket_iso + N;
    if (kind == QC_FID_KET) {
        // <g|psi> = (g_re . p_re + g_im . p_im) + i (g_re . p_im - g_im . p_re)
        for (int i = 0; i < N; ++i) {
            gr[i] = gre[i];  gr[N + i] = gim[i];
            gi[i] = -gim[i]; gi[N + i] = gre[i];
        }
    } else {
        // psi' rho psi = sum_ij conj(psi_i) rho_ij psi_j = <P, rho>_F with P = psi psi'; real for Hermitian rho:
        // Re <P, rho> = sum Re P_ij Re rho_ij + Im P_ij Im rho_ij on the iso-vec [vec(Re rho); vec(Im rho)] (column-major)
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) {
                const double pre = gre[i] * gre[j] + gim[i] * gim[j];       // Re (psi_i conj(psi_j))
                const double pim = gim[i] * gre[j] - gre[i] * gim[j];       // Im (psi_i conj(psi_j))
                gr[j * N + i] = pre;
                gr[N * N + j * N + i] = pim;
            }
    }
    auto bail = [&](hipError_t e, const char* what) { std::string m = std::string(what) + ": " + hipGetErrorString(e); qc_fidelity_destroy(h); return ffail(nullptr, QC_ERR_HIP, m); };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    const size_t nout = 2 + (size_t)h->s + (size_t)h->s * (h->s + 1) / 2;
    if ((e = hipMalloc((void**)&h->dgr, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dgi, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dU, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dOut, nout * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpy(h->dgr, gr.data(), h->s * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipMemcpy(h->dgi, gi.data(), h->s * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    *out = h;
    return QC_OK;
}

extern "C" void qc_fidelity_destroy(qc_fidelity* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (double* p : {h->dgr, h->dgi, h->dU, h->dOut}) if (p) (void)hipFree(p);
    delete h;
}

extern "C" int qc_fidelity_eval_dev(qc_fidelity* h, const double* dU, double* dval2, double* dgrad, double* dhess, void* stream) {
    if (!h) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_eval_dev: NULL handle");
    if (!dU || !dval2) return ffail(h, QC_ERR_INVALID, "qc_fidelity_eval_dev: NULL buffer");
    hipLaunchKernelGGL(qc_fidelity_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dU, h->dgr, h->dgi, h->s, h->n_sub, h->kind, dval2, dgrad, dhess);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ffail(h, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

extern "C" int qc_fidelity_eval(qc_fidelity* h, const double* U_iso, double* fidelity, double* infidelity, double* grad, double* hess) {
    if (!h) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_eval: NULL handle");
    if (!U_iso) return ffail(h, QC_ERR_INVALID, "qc_fidelity_eval: NULL input");
    QCF_HIP(h, hipSetDevice(h->device));
    QCF_HIP(h, hipMemcpyAsync(h->dU, U_iso, (size_t)h->s * 8, hipMemcpyHostToDevice, h->stream));
    double* dval = h->dOut;
    double* dgrad = h->dOut + 2;
    double* dhess = h->dOut + 2 + h->s;
    int rc = qc_fidelity_eval_dev(h, h->dU, dval, grad ? dgrad : nullptr, hess ? dhess : nullptr, h->stream);
    if (rc) return rc;
    double v[2];
    QCF_HIP(h, hipMemcpyAsync(v, dval, 16, hipMemcpyDeviceToHost, h->stream));
    if (grad) QCF_HIP(h, hipMemcpyAsync(grad, dgrad, (size_t)h->s * 8, hipMemcpyDeviceToHost, h->stream));
    if (hess) QCF_HIP(h, hipMemcpyAsync(hess, dhess, (size_t)h->s * (h->s + 1) / 2 * 8, hipMemcpyDeviceToHost, h->stream));
    QCF_HIP(h, hipStreamSynchronize(h->stream));
    if (fidelity) *fidelity = v[0];
    if (infidelity) *infidelity = v[1];
    return QC_OK;
}
