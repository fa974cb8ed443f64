// Trajectory cost terms: the quadratic regularisers on a / da / dda and the minimum-time
// term, summed over all knots, with gradient and upper-triangular Hessian:
//     J(Z) = sum_t  1/2 sum_k R_k (sc_t (v_tk - b_tk))^2  +  D sum_{t < n_mt} dt_t,      sc_t = dt_t (or 1)
// (`QuadraticRegularizer(name, traj, R; baseline, timestep_name)`, reference call sites
// unitary_smooth_pulse_problem.jl:151-153; `MinimumTimeObjective(traj; D)`, unitary_minimum_time_problem.jl:67-69).
// The per-knot weighting by dt is how QuantumCollocationCore 0.3 is recalled to define the regulariser (its source
// is not vendored: SURVEY 8c); QC_REG_PLAIN drops it.
//
// One wavefront per knot: the lanes sweep the knot's zdim entries (coalesced 8-byte loads), write the whole
// gradient row (zeros where nothing is regularised, so the caller never memsets), reduce q_t = sum_k R_k dv^2 with
// DPP-free shuffles and lane 0 finishes the dt entries.  J is reduced in a fixed order (per-knot partials, then one
// workgroup), so repeated evaluations are bit-identical.  This is O(T zdim) bytes: latency-, not bandwidth-bound.
#include <string>
#include <vector>

#include "qc_internal.h"

struct qc_terms {
    qc_terms_desc d{};
    int n_reg = 0, device = 0, cross = 0;
    int64_t hess_per_knot = 0;
    int* dslot = nullptr;          // zdim: index into the regulariser list or -1
    double* dR = nullptr;          // n_reg
    double* dbase = nullptr;       // n_reg x T or NULL
    double* dpart = nullptr;       // T partial sums
    double *dZ = nullptr, *dJ = nullptr, *dgrad = nullptr, *dhess = nullptr;   // staging for the host-pointer entry
    std::vector<int> index;
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {

struct TermsParams {
    long long T;
    int zdim, off_dt, n_reg, plain, cross;
    long long global_dim, n_mt, hess_per_knot;
    double dt_fixed, D;
    const int* slot;
    const double* R;
    const double* base;
};

__global__ __launch_bounds__(256) void qc_terms_kernel(TermsParams P, const double* __restrict__ Z, double* __restrict__ part,
                                                       double* __restrict__ grad, double* __restrict__ hess) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grad && blockIdx.x == 0)
        for (long long i = threadIdx.x; i < P.global_dim; i += 256) grad[P.T * P.zdim + i] = 0.0;
    if (t >= P.T) return;
    const double* z = Z + t * P.zdim;
    const double dt = P.off_dt >= 0 ? z[P.off_dt] : P.dt_fixed;
    const double sc = P.plain ? 1.0 : dt;
    double q = 0.0;
    for (int j = lane; j < P.zdim; j += 64) {
        const int k = P.slot[j];
        double g = 0.0;
        if (k >= 0) {
            const double w = P.R[k];
            const double dv = z[j] - (P.base ? P.base[t * P.n_reg + k] : 0.0);
            g = w * sc * sc * dv;
            q = fma(w * dv, dv, q);
            if (hess) {
                double* hk = hess + t * P.hess_per_knot;
                hk[k] = w * sc * sc;
                if (P.cross) hk[P.n_reg + k] = 2.0 * dt * w * dv;
            }
        }
        if (grad && j != P.off_dt) grad[t * P.zdim + j] = g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    if (lane == 0) {
        const double mt = (P.off_dt >= 0 && t < P.n_mt) ? P.D : 0.0;
        part[t] = 0.5 * sc * sc * q + mt * dt;
        if (grad && P.off_dt >= 0) grad[t * P.zdim + P.off_dt] = (P.plain ? 0.0 : dt * q) + mt;
        if (hess && P.cross) hess[t * P.hess_per_knot + 2 * P.n_reg] = q;
    }
}

// fixed-order sum of the per-knot partials: thread i adds part[i], part[i+256], ...; then a binary tree
__global__ __launch_bounds__(256) void qc_terms_sum_kernel(const double* __restrict__ part, long long T, double* __restrict__ J) {
    __shared__ double red[256];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < T; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) J[0] = red[0];
}

thread_local std::string g_terr;
int tfail(qc_terms* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_terr = msg;
    return code;
}

}  // namespace

#define QCT_HIP(h, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return tfail(h, QC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* qc_terms_last_error(const qc_terms* h) { return h ? h->err.c_str() : g_terr.c_str(); }

static int terms_validate(const qc_terms_desc* d, int* cross) {
    if (!d) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: NULL descriptor");
    if (d->T < 1 || d->zdim < 1 || d->global_dim < 0) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: bad T / zdim / global_dim");
    if (d->off_dt >= d->zdim) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: off_dt outside the knot");
    if (d->n_reg < 0 || d->n_reg > d->zdim) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: bad n_reg");
    if (d->n_reg > 0 && (!d->reg_index || !d->reg_R)) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: NULL regulariser arrays");
    for (int k = 0; k < d->n_reg; ++k) {
        const int j = d->reg_index[k];
        if (j < 0 || j >= d->zdim || j == d->off_dt || (k > 0 && j <= d->reg_index[k - 1]))
            return tfail(nullptr, QC_ERR_INVALID, "qc_terms: reg_index must be strictly increasing, inside the knot and not the timestep");
    }
    if (d->weighting != QC_REG_DT_SCALED && d->weighting != QC_REG_PLAIN) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: unknown weighting (QC_REG_DT_SCALED = 2, QC_REG_PLAIN = 3; the values 0 and 1 of ABI <= 0.3 are retired)");
    if (d->min_time_D != 0.0 && d->off_dt < 0) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: a minimum-time term needs a free timestep");
    if (d->min_time_knots < 0 || d->min_time_knots > d->T) return tfail(nullptr, QC_ERR_INVALID, "qc_terms: min_time_knots out of range");
    *cross = (d->weighting == QC_REG_DT_SCALED && d->off_dt >= 0 && d->n_reg > 0) ? 1 : 0;
    return QC_OK;
}

extern "C" int qc_terms_desc_hess_nnz(const qc_terms_desc* d, int64_t* nnz) {
    int cross = 0;
    int rc = terms_validate(d, &cross);
    if (rc) return rc;
    if (!nnz) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_desc_hess_nnz: NULL output");
    *nnz = d->T * ((int64_t)d->n_reg * (1 + cross) + cross);
    return QC_OK;
}

extern "C" int qc_terms_desc_hess_structure(const qc_terms_desc* d, int64_t* rows, int64_t* cols, int one_based) {
    int cross = 0;
    int rc = terms_validate(d, &cross);
    if (rc) return rc;
    if (!rows || !cols) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_desc_hess_structure: NULL output");
    const int64_t b = one_based ? 1 : 0;
    int64_t e = 0;
    for (int64_t t = 0; t < d->T; ++t) {
        const int64_t c0 = t * d->zdim + b;
        for (int k = 0; k < d->n_reg; ++k, ++e) rows[e] = cols[e] = c0 + d->reg_index[k];
        if (!cross) continue;
        for (int k = 0; k < d->n_reg; ++k, ++e) {
            const int64_t a = c0 + d->reg_index[k], c = c0 + d->off_dt;
            rows[e] = a < c ? a : c;
            cols[e] = a < c ? c : a;
        }
        rows[e] = cols[e] = c0 + d->off_dt;
        ++e;
    }
    return QC_OK;
}

extern "C" int qc_terms_create(const qc_terms_desc* d, qc_terms** out) {
    if (!out) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_create: out is NULL");
    *out = nullptr;
    int cross = 0;
    int rc = terms_validate(d, &cross);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tfail(nullptr, QC_ERR_NO_DEVICE, "qc_terms_create: no HIP device visible");
    if (d->device < 0 || d->device >= ndev) return tfail(nullptr, QC_ERR_NO_DEVICE, "qc_terms_create: device ordinal out of range");
    qc_terms* h = new qc_terms();
    h->d = *d;
    h->n_reg = d->n_reg;
    h->device = d->device;
    h->cross = cross;
    h->hess_per_knot = (int64_t)d->n_reg * (1 + cross) + cross;
    h->index.assign(d->reg_index, d->reg_index + d->n_reg);
    h->d.reg_index = nullptr;   // caller-owned arrays are not retained
    h->d.reg_R = nullptr;
    h->d.reg_baseline = nullptr;
    std::vector<int> slot(d->zdim, -1);
    for (int k = 0; k < d->n_reg; ++k) slot[d->reg_index[k]] = k;
    auto bail = [&](hipError_t e, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        qc_terms_destroy(h);
        return tfail(nullptr, QC_ERR_HIP, m);
    };
    hipError_t e;
    const size_t Zlen = (size_t)d->T * d->zdim + (size_t)d->global_dim;
    const size_t nh = (size_t)d->T * h->hess_per_knot;
    if ((e = hipSetDevice(d->device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipMalloc((void**)&h->dslot, slot.size() * 4)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpy(h->dslot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if (d->n_reg > 0) {
        if ((e = hipMalloc((void**)&h->dR, (size_t)d->n_reg * 8)) != hipSuccess) return bail(e, "hipMalloc");
        if ((e = hipMemcpy(h->dR, d->reg_R, (size_t)d->n_reg * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
        if (d->reg_baseline) {
            const size_t nb = (size_t)d->n_reg * d->T * 8;
            if ((e = hipMalloc((void**)&h->dbase, nb)) != hipSuccess) return bail(e, "hipMalloc");
            if ((e = hipMemcpy(h->dbase, d->reg_baseline, nb, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
        }
    }
    if ((e = hipMalloc((void**)&h->dpart, (size_t)d->T * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dZ, Zlen * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dJ, 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dgrad, Zlen * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if (nh && (e = hipMalloc((void**)&h->dhess, nh * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    *out = h;
    return QC_OK;
}

extern "C" void qc_terms_destroy(qc_terms* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->dslot) (void)hipFree(h->dslot);
    for (double* p : {h->dR, h->dbase, h->dpart, h->dZ, h->dJ, h->dgrad, h->dhess}) if (p) (void)hipFree(p);
    delete h;
}

extern "C" int qc_terms_hess_nnz(const qc_terms* h, int64_t* nnz) {
    if (!h || !nnz) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_hess_nnz: NULL argument");
    *nnz = h->d.T * h->hess_per_knot;
    return QC_OK;
}

extern "C" int qc_terms_hess_structure(const qc_terms* h, int64_t* rows, int64_t* cols, int one_based) {
    if (!h) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_hess_structure: NULL handle");
    qc_terms_desc d = h->d;
    static const double dummy = 0.0;
    d.reg_index = h->index.data();
    d.reg_R = &dummy;
    return qc_terms_desc_hess_structure(&d, rows, cols, one_based);
}

extern "C" int qc_terms_eval_dev(qc_terms* h, const double* dZ, double* dJ, double* dgrad, double* dhvals, void* stream) {
    if (!h) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_eval_dev: NULL handle");
    if (!dZ || !dJ) return tfail(h, QC_ERR_INVALID, "qc_terms_eval_dev: NULL buffer");
    TermsParams P;
    P.T = h->d.T;
    P.zdim = h->d.zdim;
    P.off_dt = h->d.off_dt;
    P.n_reg = h->n_reg;
    P.plain = h->d.weighting == QC_REG_PLAIN;
    P.cross = h->cross;
    P.global_dim = h->d.global_dim;
    P.n_mt = h->d.min_time_knots;
    P.hess_per_knot = h->hess_per_knot;
    P.dt_fixed = h->d.dt_fixed;
    P.D = h->d.min_time_D;
    P.slot = h->dslot;
    P.R = h->dR;
    P.base = h->dbase;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((h->d.T + 3) / 4);
    hipLaunchKernelGGL(qc_terms_kernel, dim3(grid), dim3(256), 0, s, P, dZ, h->dpart, dgrad, h->hess_per_knot ? dhvals : nullptr);
    hipLaunchKernelGGL(qc_terms_sum_kernel, dim3(1), dim3(256), 0, s, h->dpart, (long long)h->d.T, dJ);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tfail(h, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

extern "C" int qc_terms_eval(qc_terms* h, const double* Z, double* J, double* grad, double* hvals) {
    if (!h) return tfail(nullptr, QC_ERR_INVALID, "qc_terms_eval: NULL handle");
    if (!Z) return tfail(h, QC_ERR_INVALID, "qc_terms_eval: NULL input");
    const size_t Zlen = (size_t)h->d.T * h->d.zdim + (size_t)h->d.global_dim;
    const size_t nh = (size_t)h->d.T * h->hess_per_knot;
    QCT_HIP(h, hipSetDevice(h->device));
    QCT_HIP(h, hipMemcpyAsync(h->dZ, Z, Zlen * 8, hipMemcpyHostToDevice, h->stream));
    int rc = qc_terms_eval_dev(h, h->dZ, h->dJ, grad ? h->dgrad : nullptr, hvals ? h->dhess : nullptr, h->stream);
    if (rc) return rc;
    double j = 0.0;
    QCT_HIP(h, hipMemcpyAsync(&j, h->dJ, 8, hipMemcpyDeviceToHost, h->stream));
    if (grad) QCT_HIP(h, hipMemcpyAsync(grad, h->dgrad, Zlen * 8, hipMemcpyDeviceToHost, h->stream));
    if (hvals && nh) QCT_HIP(h, hipMemcpyAsync(hvals, h->dhess, nh * 8, hipMemcpyDeviceToHost, h->stream));
    QCT_HIP(h, hipStreamSynchronize(h->stream));
    if (J) *J = j;
    return QC_OK;
}
