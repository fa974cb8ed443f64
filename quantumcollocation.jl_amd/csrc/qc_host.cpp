// Host side of libqcolloc_hip.so: descriptor validation, value-block layout, sparsity structure,
// handle lifetime, host-buffer staging.  The arithmetic of the path lives in the .hip kernels; there
// is deliberately no CPU evaluation path in this library.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <string>
#include <vector>

#include "qc_internal.h"


static thread_local std::string g_err;

int qc_fail(std::string* err, int code, const std::string& msg) {
    if (err) *err = msg;
    g_err = msg;
    return code;
}
#define fail qc_fail

#define QC_STR2(x) #x
#define QC_STR(x) QC_STR2(x)
extern "C" const char* qc_version(void) {
    return "qcolloc-hip " QC_STR(QC_VERSION_MAJOR) "." QC_STR(QC_VERSION_MINOR)
           " (gfx950, fp64; kernels: lds, mfma16, mfma32, mfma64, mfma16-exp, mfma32-exp)";
}

extern "C" int32_t qc_abi_version(void) { return QC_VERSION_MAJOR * 1000 + QC_VERSION_MINOR; }

extern "C" const char* qc_last_error(const qc_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
//  Isomorphism helpers (reference trajectory_initialization.jl:137)
// ------------------------------------------------------------------------------------------------
extern "C" int qc_operator_to_iso_vec(int32_t N, const double* U_re, const double* U_im, double* v) {
    if (N <= 0 || !U_re || !U_im || !v) return fail(nullptr, QC_ERR_INVALID, "qc_operator_to_iso_vec: bad argument");
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
            v[j * 2 * N + i] = U_re[j * N + i];
            v[j * 2 * N + N + i] = U_im[j * N + i];
        }
    return QC_OK;
}

extern "C" int qc_iso_vec_to_operator(int32_t N, const double* v, double* U_re, double* U_im) {
    if (N <= 0 || !U_re || !U_im || !v) return fail(nullptr, QC_ERR_INVALID, "qc_iso_vec_to_operator: bad argument");
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
            U_re[j * N + i] = v[j * 2 * N + i];
            U_im[j * N + i] = v[j * 2 * N + N + i];
        }
    return QC_OK;
}

extern "C" int qc_generator_from_hamiltonian(int32_t N, const double* H_re, const double* H_im, double* G) {
    if (N <= 0 || !H_re || !H_im || !G) return fail(nullptr, QC_ERR_INVALID, "qc_generator_from_hamiltonian: bad argument");
    const int n = 2 * N;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
            const double re = H_re[j * N + i], im = H_im[j * N + i];
            G[j * n + i] = im;             // top-left     Im H
            G[(j + N) * n + i] = re;       // top-right    Re H
            G[j * n + N + i] = -re;        // bottom-left -Re H
            G[(j + N) * n + N + i] = im;   // bottom-right Im H
        }
    return QC_OK;
}

extern "C" int qc_pade_coefficients(int32_t order, double* out) {
    if (order < 2 || (order & 1) || order / 2 > QC_MAX_P || !out)
        return fail(nullptr, QC_ERR_INVALID, "qc_pade_coefficients: order must be even, 2..20");
    const int p = order / 2;
    // c_k = (2p-k)! p! / ((2p)! k! (p-k)!)  ->  c_0 = 1, c_k = c_{k-1} * (p-k+1) / (k (2p-k+1))
    out[0] = 1.0;
    for (int k = 1; k <= p; ++k) out[k] = out[k - 1] * (double)(p - k + 1) / ((double)k * (double)(2 * p - k + 1));
    return QC_OK;
}

// ------------------------------------------------------------------------------------------------
//  Descriptor -> parameters
// ------------------------------------------------------------------------------------------------
static bool overlaps(int a0, int a1, int b0, int b1) { return a0 < b1 && b0 < a1; }

int qc_build_params(const qc_desc* d, QcParams* P, qc_dims_t* dims, std::string* err) {
    if (!d) return fail(err, QC_ERR_INVALID, "descriptor is NULL");
    if (d->N < 1 || d->N > 64) return fail(err, QC_ERR_INVALID, "N must be in 1..64");
    if (d->m < 0 || d->m > 64) return fail(err, QC_ERR_INVALID, "m must be in 0..64");
    if (d->T < 2) return fail(err, QC_ERR_INVALID, "T must be >= 2");
    if (d->global_dim < 0) return fail(err, QC_ERR_INVALID, "global_dim must be >= 0");
    memset(P, 0, sizeof(*P));
    P->N = d->N;
    P->n = 2 * d->N;
    if (d->state_cols < 0 || d->state_cols > 64) return fail(err, QC_ERR_INVALID, "state_cols must be in 0..64");
    P->nc = d->state_cols > 0 ? d->state_cols : d->N;
    P->s = 2 * d->N * P->nc;
    P->copies = P->nc;
    P->m = d->m;
    P->zdim = d->zdim;
    P->off_U = d->off_U;
    P->off_a = d->off_a;
    P->off_dt = d->off_dt < 0 ? -1 : d->off_dt;
    P->dt_fixed = d->dt_fixed;
    if (d->zdim < P->s + P->m + (P->off_dt >= 0 ? 1 : 0)) return fail(err, QC_ERR_INVALID, "zdim too small for U, a (and dt)");
    if (d->off_U < 0 || d->off_U + P->s > d->zdim) return fail(err, QC_ERR_INVALID, "off_U out of range");
    if (d->off_a < 0 || d->off_a + P->m > d->zdim) return fail(err, QC_ERR_INVALID, "off_a out of range");
    if (P->off_dt >= d->zdim) return fail(err, QC_ERR_INVALID, "off_dt out of range");
    if (P->m > 0 && overlaps(d->off_U, d->off_U + P->s, d->off_a, d->off_a + P->m))
        return fail(err, QC_ERR_INVALID, "U and a components overlap");
    if (P->off_dt >= 0 && (overlaps(P->off_dt, P->off_dt + 1, d->off_U, d->off_U + P->s) ||
                           overlaps(P->off_dt, P->off_dt + 1, d->off_a, d->off_a + P->m)))
        return fail(err, QC_ERR_INVALID, "dt overlaps U or a");
    if (d->integrator != QC_PADE && d->integrator != QC_EXPONENTIAL) return fail(err, QC_ERR_INVALID, "unknown integrator");
    P->integrator = d->integrator;
    if (d->integrator == QC_PADE) {
        if (d->pade_order < 2 || (d->pade_order & 1) || d->pade_order / 2 > QC_MAX_P)
            return fail(err, QC_ERR_INVALID, "pade_order must be even, 2..20");
        P->p = d->pade_order / 2;
        qc_pade_coefficients(d->pade_order, P->c);
    } else {
        P->p = 0;
    }
    if (d->n_deriv < 0 || d->n_deriv > QC_MAX_DERIV) return fail(err, QC_ERR_INVALID, "n_deriv out of range");
    P->n_deriv = d->n_deriv;
    P->ddim = P->s;
    if (d->row_placement != QC_ROWS_STACKED && d->row_placement != QC_ROWS_BY_COMPONENT)
        return fail(err, QC_ERR_INVALID, "unknown row_placement");
    const bool by_comp = d->row_placement == QC_ROWS_BY_COMPONENT;
    if (by_comp && d->rows_per_interval <= 0) return fail(err, QC_ERR_INVALID, "QC_ROWS_BY_COMPONENT needs rows_per_interval (Z.dims.states)");
    for (int i = 0; i < d->n_deriv; ++i) {
        const int dim = d->deriv_dim[i], xo = d->deriv_x_off[i], dxo = d->deriv_dx_off[i];
        if (dim < 1 || xo < 0 || dxo < 0 || xo + dim > d->zdim || dxo + dim > d->zdim)
            return fail(err, QC_ERR_INVALID, "derivative integrator component out of range");
        if (overlaps(xo, xo + dim, dxo, dxo + dim)) return fail(err, QC_ERR_INVALID, "derivative integrator x and dx overlap");
        if (P->off_dt >= 0 && (overlaps(P->off_dt, P->off_dt + 1, xo, xo + dim) || overlaps(P->off_dt, P->off_dt + 1, dxo, dxo + dim)))
            return fail(err, QC_ERR_INVALID, "derivative integrator component overlaps dt");
        P->x_off[i] = xo;
        P->dx_off[i] = dxo;
        P->ddim_i[i] = dim;
        if (by_comp) {
            const long long ro = d->deriv_row_off[i];
            if (ro < 0 || ro + dim > d->rows_per_interval) return fail(err, QC_ERR_INVALID, "deriv_row_off out of the per-interval row block");
            if (overlaps((int)ro, (int)ro + dim, (int)d->row_offset, (int)d->row_offset + P->s))
                return fail(err, QC_ERR_INVALID, "derivative integrator rows overlap the state integrator's rows");
            for (int k = 0; k < i; ++k)
                if (overlaps((int)ro, (int)ro + dim, d->deriv_row_off[k], d->deriv_row_off[k] + d->deriv_dim[k]))
                    return fail(err, QC_ERR_INVALID, "derivative integrator rows overlap each other");
            P->drow[i] = (int)(ro - d->row_offset);
        } else {
            P->drow[i] = P->ddim;
        }
        P->ddim += dim;
    }
    long long tb = d->t_begin, te = d->t_end;
    if (tb == 0 && te == 0) te = d->T - 1;
    if (tb < 0 || te > d->T - 1 || tb > te) return fail(err, QC_ERR_INVALID, "interval range [t_begin, t_end) out of [0, T-1)");
    if (te - tb > 0x7fffffffLL / 2) return fail(err, QC_ERR_INVALID, "too many intervals for one handle");
    P->t_begin = tb;
    P->n_int = (int)(te - tb);

    const int n2 = P->n * P->n, s = P->s, m = P->m;
    const bool ft = P->off_dt >= 0;
    // Block orders (ABI 0.6): permutations of the block kinds; all zeros = the default order
    int jord[QC_JAC_BLOCKS], hord[QC_HESS_BLOCKS];
    {
        bool jzero = true, hzero = true;
        for (int i = 0; i < QC_JAC_BLOCKS; ++i) jzero = jzero && d->jac_block_order[i] == 0;
        for (int i = 0; i < QC_HESS_BLOCKS; ++i) hzero = hzero && d->hess_block_order[i] == 0;
        unsigned seen = 0;
        for (int i = 0; i < QC_JAC_BLOCKS; ++i) {
            jord[i] = jzero ? i : d->jac_block_order[i];
            if (jord[i] < 0 || jord[i] >= QC_JAC_BLOCKS || (seen & (1u << jord[i]))) return fail(err, QC_ERR_INVALID, "jac_block_order is not a permutation of QC_JB_*");
            seen |= 1u << jord[i];
        }
        seen = 0;
        for (int i = 0; i < QC_HESS_BLOCKS; ++i) {
            hord[i] = hzero ? i : d->hess_block_order[i];
            if (hord[i] < 0 || hord[i] >= QC_HESS_BLOCKS || (seen & (1u << hord[i]))) return fail(err, QC_ERR_INVALID, "hess_block_order is not a permutation of QC_HB_*");
            seen |= 1u << hord[i];
        }
    }
    // Jacobian block offsets
    int o = 0;
    {
        int dlen = 0;
        for (int i = 0; i < P->n_deriv; ++i) dlen += (ft ? 4 : 3) * P->ddim_i[i];
        const int len[QC_JAC_BLOCKS] = {P->nc * n2, (P->integrator == QC_PADE) ? P->nc * n2 : s, s * m, ft ? s : 0, dlen};
        int* const off[QC_JAC_BLOCKS] = {&P->jo_F, &P->jo_B, &P->jo_a, &P->jo_h, &P->jo_d};
        for (int i = 0; i < QC_JAC_BLOCKS; ++i) { *off[jord[i]] = o; o += len[jord[i]]; }
    }
    P->jac_nnz = o;
    // Hessian block offsets.  The exponential integrator's residual U_t+1 - exp(h G) U_t is linear in U_t+1: its (a, U_t+1) and
    // (h, U_t+1) blocks are structurally empty (the reference solves :exponential problems with the Hessian left on,
    // unitary_smooth_pulse_problem.jl:224-266).
    // Default order: the four kinds of matrix blocks first -- whole 128-byte lines each when the interval's block is line-aligned
    // (hess_align = 16, 2N x N a multiple of 16) --, then the scalar blocks as ONE contiguous run that the one-call kernel assembles and
    // stores in one piece (round 5: the scalar entries used to sit between the blocks; partly written lines at the tail of that launch
    // cost 0.7 us)
    o = 0;
    const int ub = P->integrator == QC_PADE ? 1 : 0;
    {
        int dlen = 0;
        if (ft) for (int i = 0; i < P->n_deriv; ++i) dlen += P->ddim_i[i];
        const int len[QC_HESS_BLOCKS] = {s * m, ub * s * m, ft ? s : 0, ft ? ub * s : 0, m * (m + 1) / 2, ft ? m : 0, ft ? 1 : 0, dlen};
        int* const off[QC_HESS_BLOCKS] = {&P->ho_Ua, &P->ho_aU, &P->ho_Uh, &P->ho_hU, &P->ho_aa, &P->ho_ah, &P->ho_hh, &P->ho_d};
        for (int i = 0; i < QC_HESS_BLOCKS; ++i) { *off[hord[i]] = o; o += len[hord[i]]; }
        P->scal_run = hord[4] == QC_HB_AA && hord[5] == QC_HB_AH && hord[6] == QC_HB_HH && hord[7] == QC_HB_D;
    }
    P->hess_nnz = o;
    // placement inside the problem's vectors
    if (d->rows_per_interval < 0 || d->row_offset < 0 || d->jac_per_interval < 0 || d->jac_offset < 0 || d->hess_per_interval < 0 || d->hess_offset < 0)
        return fail(err, QC_ERR_INVALID, "composition strides/offsets must be >= 0");
    P->F_stride = d->rows_per_interval > 0 ? d->rows_per_interval : P->ddim;
    P->F_off = d->row_offset;
    P->J_stride = d->jac_per_interval > 0 ? d->jac_per_interval : P->jac_nnz;
    P->J_off = d->jac_offset;
    // Line alignment of the per-interval Hessian blocks: explicit zeros after the handle's own values.
    if (d->hess_align < 0 || d->hess_align > 64) return fail(err, QC_ERR_INVALID, "hess_align must be in 0..64 (16 = whole 128-byte lines)");
    if (d->hess_tail_zeros < 0 || d->hess_tail_zeros > 4096) return fail(err, QC_ERR_INVALID, "hess_tail_zeros must be in 0..4096");
    if (d->hess_per_interval > 0) {
        P->h_pad = P->hess_nnz ? d->hess_tail_zeros : 0;
    } else {
        const int al = d->hess_align == 0 ? 1 : d->hess_align;   // 0 = the reference's structure, no padding (ABI 0.5)
        P->h_pad = P->hess_nnz ? (al - P->hess_nnz % al) % al : 0;
    }
    P->H_stride = d->hess_per_interval > 0 ? d->hess_per_interval : P->hess_nnz + P->h_pad;
    P->H_off = d->hess_offset;
    const long long own_rows_end = by_comp ? P->F_off + P->s : P->F_off + P->ddim;   // (derivative rows were checked above)
    if (own_rows_end > P->F_stride || P->J_off + P->jac_nnz > P->J_stride || (P->hess_nnz && P->H_off + P->hess_nnz + P->h_pad > P->H_stride))
        return fail(err, QC_ERR_INVALID, "composition offset + own size exceeds the per-interval block");

    if (dims) {
        memset(dims, 0, sizeof(*dims));
        dims->n_rows = (int64_t)P->F_stride * (d->T - 1);
        dims->n_cols = (int64_t)d->zdim * d->T + d->global_dim;
        dims->ddim = P->ddim;
        dims->jac_nnz_interval = P->jac_nnz;
        dims->hess_nnz_interval = P->hess_nnz + P->h_pad;
        dims->n_intervals = P->n_int;
        dims->F_len = (int64_t)P->F_stride * P->n_int;
        dims->jac_nnz = (int64_t)P->jac_nnz * P->n_int;
        dims->hess_nnz = (int64_t)(P->hess_nnz + P->h_pad) * P->n_int;
        dims->Z_len = dims->n_cols;
        dims->kernel = 0;
    }
    return QC_OK;
}

// Local structure: rows in [0, ddim), cols in [0, 2*zdim) with col >= zdim meaning knot t+1.  Every block is written at ITS offset
// (P.jo_* / P.ho_*: qc_desc.jac_block_order / hess_block_order decide where the blocks sit; inside a block the order is fixed).
void qc_local_jac_structure(const QcParams& P, std::vector<int32_t>* R, std::vector<int32_t>* C) {
    R->assign(P.jac_nnz, 0); C->assign(P.jac_nnz, 0);
    const int n = P.n, N = P.nc, s = P.s, m = P.m, zd = P.zdim;
    const bool ft = P.off_dt >= 0;
    int o = 0;
    auto put = [&](int r, int c) { (*R)[o] = r; (*C)[o] = c; ++o; };
    o = P.jo_F;
    for (int q = 0; q < N; ++q)
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) put(q * n + r, P.off_U + q * n + c);
    o = P.jo_B;
    if (P.integrator == QC_PADE) {
        for (int q = 0; q < N; ++q)
            for (int c = 0; c < n; ++c)
                for (int r = 0; r < n; ++r) put(q * n + r, zd + P.off_U + q * n + c);
    } else {
        for (int i = 0; i < s; ++i) put(i, zd + P.off_U + i);
    }
    o = P.jo_a;
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < s; ++i) put(i, P.off_a + j);
    o = P.jo_h;
    if (ft)
        for (int i = 0; i < s; ++i) put(i, P.off_dt);
    o = P.jo_d;
    for (int d = 0; d < P.n_deriv; ++d) {
        const int dim = P.ddim_i[d], r0 = P.drow[d];   // relative to the handle's row block (F_off is added by the caller)
        for (int i = 0; i < dim; ++i) put(r0 + i, P.x_off[d] + i);
        for (int i = 0; i < dim; ++i) put(r0 + i, zd + P.x_off[d] + i);
        for (int i = 0; i < dim; ++i) put(r0 + i, P.dx_off[d] + i);
        if (ft) for (int i = 0; i < dim; ++i) put(r0 + i, P.off_dt);
    }
}

void qc_local_hess_structure(const QcParams& P, std::vector<int32_t>* R, std::vector<int32_t>* C) {
    R->assign(P.hess_nnz + P.h_pad, 0); C->assign(P.hess_nnz + P.h_pad, 0);
    const int s = P.s, m = P.m, zd = P.zdim;
    const bool ft = P.off_dt >= 0, pade = P.integrator == QC_PADE;   // exponential: no entry touches knot t+1
    int o = 0;
    auto up = [&](int i, int j) { (*R)[o] = std::min(i, j); (*C)[o] = std::max(i, j); ++o; };
    o = P.ho_Ua;
    for (int j = 0; j < m; ++j) for (int i = 0; i < s; ++i) up(P.off_U + i, P.off_a + j);
    o = P.ho_aU;
    if (pade) for (int j = 0; j < m; ++j) for (int i = 0; i < s; ++i) up(P.off_a + j, zd + P.off_U + i);
    if (ft) {
        o = P.ho_Uh;
        for (int i = 0; i < s; ++i) up(P.off_U + i, P.off_dt);
        o = P.ho_hU;
        if (pade) for (int i = 0; i < s; ++i) up(P.off_dt, zd + P.off_U + i);
    }
    o = P.ho_aa;
    for (int j = 0; j < m; ++j) for (int i = 0; i <= j; ++i) up(P.off_a + i, P.off_a + j);
    if (ft) {
        o = P.ho_ah;
        for (int j = 0; j < m; ++j) up(P.off_a + j, P.off_dt);
        o = P.ho_hh;
        up(P.off_dt, P.off_dt);
        o = P.ho_d;
        for (int d = 0; d < P.n_deriv; ++d) for (int i = 0; i < P.ddim_i[d]; ++i) up(P.dx_off[d] + i, P.off_dt);
    }
    // alignment padding: explicit zeros, recorded as duplicates of the first entry (COO duplicates are summed)
    for (int i = 0; i < P.h_pad; ++i) { (*R)[P.hess_nnz + i] = (*R)[0]; (*C)[P.hess_nnz + i] = (*C)[0]; }
}

static void expand_structure(const QcParams& P, const std::vector<int32_t>& lr, const std::vector<int32_t>& lc,
                             long long row_stride, long long row_off, int64_t* rows, int64_t* cols, int one_based) {
    const int64_t o = one_based ? 1 : 0;
    const size_t k = lr.size();
    for (int b = 0; b < P.n_int; ++b) {
        const long long t = P.t_begin + b;
        for (size_t e = 0; e < k; ++e) {
            rows[(size_t)b * k + e] = t * row_stride + row_off + lr[e] + o;
            cols[(size_t)b * k + e] = t * (long long)P.zdim + lc[e] + o;
        }
    }
}

extern "C" int qc_desc_dims(const qc_desc* d, qc_dims_t* out) {
    QcParams P;
    std::string err;
    if (!out) return fail(nullptr, QC_ERR_INVALID, "qc_desc_dims: out is NULL");
    return qc_build_params(d, &P, out, &err);
}

extern "C" int qc_desc_jac_structure(const qc_desc* d, int64_t* rows, int64_t* cols, int one_based) {
    QcParams P; qc_dims_t dims; std::string err;
    int rc = qc_build_params(d, &P, &dims, &err);
    if (rc) return rc;
    if (!rows || !cols) return fail(nullptr, QC_ERR_INVALID, "qc_desc_jac_structure: NULL output");
    std::vector<int32_t> lr, lc;
    qc_local_jac_structure(P, &lr, &lc);
    expand_structure(P, lr, lc, P.F_stride, P.F_off, rows, cols, one_based);
    return QC_OK;
}

extern "C" int qc_desc_hess_structure(const qc_desc* d, int64_t* rows, int64_t* cols, int one_based) {
    QcParams P; qc_dims_t dims; std::string err;
    int rc = qc_build_params(d, &P, &dims, &err);
    if (rc) return rc;
    if (P.hess_nnz == 0) return QC_OK;
    if (!rows || !cols) return fail(nullptr, QC_ERR_INVALID, "qc_desc_hess_structure: NULL output");
    std::vector<int32_t> lr, lc;
    qc_local_hess_structure(P, &lr, &lc);
    // Hessian rows AND cols are variable indices: both use the zdim stride.
    expand_structure(P, lr, lc, P.zdim, 0, rows, cols, one_based);
    return QC_OK;
}

// ------------------------------------------------------------------------------------------------
//  Handle
// ------------------------------------------------------------------------------------------------
extern "C" int qc_create(const qc_desc* d, qc_handle** out) {
    if (!out) return fail(nullptr, QC_ERR_INVALID, "qc_create: out is NULL");
    *out = nullptr;
    QcParams P; qc_dims_t dims; std::string err;
    int rc = qc_build_params(d, &P, &dims, &err);
    if (rc) return rc;
    if (!d->G_drift || (d->m > 0 && !d->G_drives)) return fail(nullptr, QC_ERR_INVALID, "qc_create: G_drift/G_drives is NULL");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, QC_ERR_NO_DEVICE, "qc_create: no HIP device visible (this library has no CPU path)");
    if (d->device < 0 || d->device >= ndev) return fail(nullptr, QC_ERR_NO_DEVICE, "qc_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) != hipSuccess) return fail(nullptr, QC_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, QC_ERR_NO_DEVICE, std::string("qc_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

    qc_handle* h = new qc_handle();
    h->desc = *d;
    h->desc.G_drift = nullptr;
    h->desc.G_drives = nullptr;
    h->device = d->device;
    h->prm = P;
    h->dims = dims;

    // kernel selection
    int kernel = d->kernel;
    const bool mfma_ok = qc_mfma_supported(P);
    if (kernel == QC_KERNEL_AUTO) kernel = mfma_ok ? QC_KERNEL_MFMA : QC_KERNEL_LDS;
    if (kernel == QC_KERNEL_MFMA && !mfma_ok) {
        delete h;
        return fail(nullptr, QC_ERR_UNSUPPORTED, "qc_create: no MFMA kernel serves this descriptor (order-4 Pade up to 32 levels, other Pade orders up to 8 levels, "
                                                   "the exponential integrator up to 16 levels; see qc_desc.kernel)");
    }
    if (kernel != QC_KERNEL_MFMA && kernel != QC_KERNEL_LDS) { delete h; return fail(nullptr, QC_ERR_INVALID, "qc_create: unknown kernel id"); }
    static std::atomic<unsigned long long> next_serial{1};
    h->serial = next_serial.fetch_add(1);
    h->kernel = kernel;
    h->dims.kernel = kernel;

    auto bail = [&](int code, const std::string& msg) { std::string m2 = msg; qc_destroy(h); return fail(nullptr, code, m2); };
#define QC_HIP_C(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bail(QC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

    qc_device_guard guard(h->device);
    QC_HIP_C(guard.err);
    const size_t n2 = (size_t)P.n * P.n;
    std::vector<double> G((size_t)(P.m + 1) * n2);
    memcpy(G.data(), d->G_drift, n2 * sizeof(double));
    if (P.m) memcpy(G.data() + n2, d->G_drives, (size_t)P.m * n2 * sizeof(double));
    {   // exact antisymmetry of every generator (G = iso(-iH) of a Hermitian H): lets the Hessian kernels skip the transposed images
        bool anti = true;
        for (int mat = 0; anti && mat <= P.m; ++mat) {
            const double* A = G.data() + (size_t)mat * n2;
            for (int c = 0; anti && c < P.n; ++c)
                for (int r = 0; r <= c; ++r)
                    if (A[(size_t)c * P.n + r] != -A[(size_t)r * P.n + c]) { anti = false; break; }
        }
        h->prm.antisym = anti ? 1 : 0;
        if (const char* e = getenv("QC_NO_ANTISYM")) if (atoi(e)) h->prm.antisym = 0;   // diagnostic: force the general path
    }
    QC_HIP_C(hipMalloc((void**)&h->dG, G.size() * sizeof(double)));
    QC_HIP_C(hipMemcpy(h->dG, G.data(), G.size() * sizeof(double), hipMemcpyHostToDevice));
    h->prm.G = h->dG;
    // Outputs are written once and never re-read by the kernel: non-temporal stores measured fastest on
    // MI355X (profiles/README.md: plain 14.4, sc1 12.8, nt 11.9 us per config-3 evaluation).
    h->prm.store_mode = 2;
    if (const char* e = getenv("QC_STORE_MODE")) h->prm.store_mode = std::max(0, std::min(2, atoi(e)));   // diagnostic override
    // Diagnostic ablation for the profiling scripts: it produces WRONG results, so it needs the explicit opt-in
    // QC_DIAGNOSTICS=1 next to it and says so on stderr every time a handle is created with it.
    if (const char* e = getenv("QC_DEBUG_SKIP")) {
        const char* opt = getenv("QC_DIAGNOSTICS");
        if (atoi(e) != 0 && opt && atoi(opt) == 1) {
            h->prm.dbg_skip = atoi(e);
            fprintf(stderr, "qcolloc: QC_DEBUG_SKIP=%d is active: outputs of this handle are NOT valid results\n", h->prm.dbg_skip);
        } else if (atoi(e) != 0) {
            fprintf(stderr, "qcolloc: QC_DEBUG_SKIP ignored (set QC_DIAGNOSTICS=1 to enable the ablation)\n");
        }
    }
    if (kernel == QC_KERNEL_MFMA) {
        std::vector<double> Gx(qc_mfma_gx_doubles(P));
        qc_mfma_pack_G(P, G.data(), Gx.data());
        QC_HIP_C(hipMalloc((void**)&h->dGx, Gx.size() * sizeof(double)));
        QC_HIP_C(hipMemcpy(h->dGx, Gx.data(), Gx.size() * sizeof(double), hipMemcpyHostToDevice));
        h->prm.Gx = h->dGx;
        // drive generators with at most two entries per row (Pauli strings, ladder pairs): row-gather tables for the kernels that
        // never touch a dense drive image (QC_NO_ELL=1: the dense kernels, for A/B runs)
        const bool no_ell = getenv("QC_NO_ELL") && atoi(getenv("QC_NO_ELL"));
        std::vector<char> blob;
        int slots = 0;
        const int R = no_ell ? 0 : qc_mfma32_ell_build(h->prm, G.data(), &blob, &slots);
        if (R > 0) {
            QC_HIP_C(hipMalloc(&h->dEll, blob.size()));
            QC_HIP_C(hipMemcpy(h->dEll, blob.data(), blob.size(), hipMemcpyHostToDevice));
            h->prm.ell = h->dEll;
            h->prm.ell_R = R;
            h->prm.ell_slots = slots;
        }
        std::vector<char> blob16;
        if (!no_ell && qc_mfma16_ell_build(h->prm, G.data(), &blob16)) {
            qc_mfma16_ell_pair_table(h->prm, &blob16);
            QC_HIP_C(hipMalloc(&h->dEll16, blob16.size()));
            QC_HIP_C(hipMemcpy(h->dEll16, blob16.data(), blob16.size(), hipMemcpyHostToDevice));
            h->prm.ell16 = h->dEll16;
        } else if (!no_ell && qc_exp_ell_build(h->prm, G.data(), &blob16)) {      // the exponential integrator's MFMA kernels (2N <= 32)
            QC_HIP_C(hipMalloc(&h->dEll16, blob16.size()));
            QC_HIP_C(hipMemcpy(h->dEll16, blob16.data(), blob16.size(), hipMemcpyHostToDevice));
            h->prm.ell16 = h->dEll16;
        }
    }
    // LDS budget of the LDS kernels
    {
        // choose the j-chunk so the LDS kernel fits in 160 KiB (64 KiB keeps >= 2 blocks per CU when possible)
        QcParams& Q = h->prm;
        Q.jchunk = std::max(1, Q.m);
        while (Q.jchunk > 1 && qc_lds_bytes_jac(Q) > 64 * 1024) Q.jchunk = (Q.jchunk + 1) / 2;
        h->lds_bytes_jac = qc_lds_bytes_jac(Q);
        h->lds_bytes_hess = qc_lds_bytes_hess(Q);
        if (h->lds_bytes_jac > 160 * 1024 || h->lds_bytes_hess > 160 * 1024) {
            // Too large for LDS: the same kernels run with their scratch in a global-memory workspace (slow, but the
            // library does not refuse the problem: 5 qubits, N = 32, are 2 N = 64 rows).
            Q.use_ws = 1;
            Q.jchunk = std::min(std::max(1, Q.m), 2);
            h->lds_bytes_jac = qc_lds_bytes_jac(Q);
            h->lds_bytes_hess = qc_lds_bytes_hess(Q);
            const size_t per = (std::max(h->lds_bytes_jac, h->lds_bytes_hess) / sizeof(double) + 1) & ~(size_t)1;
            const size_t total = per * (size_t)std::max(1, Q.n_int);
            if (total * sizeof(double) > ((size_t)16 << 30))
                return bail(QC_ERR_UNSUPPORTED, "qc_create: the global workspace for this system would exceed 16 GiB");
            QC_HIP_C(hipMalloc((void**)&h->dWs, total * sizeof(double)));
            Q.ws = h->dWs;
            Q.ws_stride = (long long)per;
        }
    }
    if (const char* e = getenv("QC_HOST_COMPACT")) h->host_compact = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("QC_HOST_LANDING")) h->host_landing = atoi(e) ? 1 : 0;
    if (const char* e = getenv("QC_STAMPS")) {
        if (atoi(e) && P.n_int > 0) {
            QC_HIP_C(hipMalloc((void**)&h->dStamps, (size_t)P.n_int * 16 * sizeof(unsigned long long)));
            QC_HIP_C(hipMemset(h->dStamps, 0, (size_t)P.n_int * 16 * sizeof(unsigned long long)));
            h->prm.stamps = h->dStamps;
        }
    }
    QC_HIP_C(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    QC_HIP_C(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    QC_HIP_C(hipEventCreateWithFlags(&h->ev_staged, hipEventDisableTiming));
    QC_HIP_C(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
#undef QC_HIP_C
    *out = h;
    return QC_OK;
}

extern "C" void qc_destroy(qc_handle* h) {
    if (!h) return;
    if (!h->shards.empty() || h->fan) {   // multi-device handle: no device state of its own
        if (h->fan) qc_fanout_destroy(h->fan);
        if (h->rccl) qc_rccl_destroy(h->rccl);
        for (qc_handle* sh : h->shards) qc_destroy(sh);
        delete h;
        return;
    }
    qc_device_guard guard(h->device);
    // both streams idle before any pinned or device block is freed (a failed call may have left chunk kernels in flight on either)
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    double* bufs[] = {h->dG, h->dGx, h->dZ, h->dF, h->dJ, h->dMu, h->dH, (double*)h->dStamps, h->dRE, h->dRQ, h->dRS, h->dRinit, h->dRout, h->dRZ, h->dWs, h->dHs};
    if (h->hJc) (void)hipHostFree(h->hJc);
    if (h->hFc) (void)hipHostFree(h->hFc);
    if (h->hZ) (void)hipHostFree(h->hZ);
    for (int i = 0; i < QC_HOST_RING; ++i) {
        if (h->rearm[i]) qc_rearm_destroy(h->rearm[i]);     // (its jobs write into hC[i])
        if (h->hC[i]) (void)hipHostFree(h->hC[i]);
    }
    if (h->dC) (void)hipFree(h->dC);
    if (h->dEll) (void)hipFree(h->dEll);
    if (h->dEll16) (void)hipFree(h->dEll16);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->dBatch) (void)hipFree(h->dBatch);
    if (h->dBatchLand) (void)hipFree(h->dBatchLand);
    for (hipEvent_t ev : h->chunk_events) if (ev) (void)hipEventDestroy(ev);
    for (double* b : bufs) if (b) (void)hipFree(b);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->ev_staged) (void)hipEventDestroy(h->ev_staged);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int qc_debug_read_stamps(qc_handle* h, uint64_t* out, int64_t count) {
    if (!h || !out) return fail(nullptr, QC_ERR_INVALID, "qc_debug_read_stamps: NULL argument");
    if (!h->shards.empty()) return fail(&h->err, QC_ERR_UNSUPPORTED, "qc_debug_read_stamps: use the shard handles of a multi-device handle");
    if (!h->dStamps) return fail(&h->err, QC_ERR_UNSUPPORTED, "handle was not created with QC_STAMPS=1");
    if (count > (int64_t)h->prm.n_int * 16) count = (int64_t)h->prm.n_int * 16;
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    QC_HIP(h, hipDeviceSynchronize());
    QC_HIP(h, hipMemcpy(out, h->dStamps, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return QC_OK;
}

extern "C" const char* qc_kernel_name(const qc_handle* h, int32_t which) {
    if (!h) return "none";
    if (!h->shards.empty()) return qc_kernel_name(h->shards[0], which);
    const QcParams& P = h->prm;
    const bool mfma = h->kernel == QC_KERNEL_MFMA;
    if (which == 0) {
        if (!mfma) return P.use_ws ? "lds-gws" : "lds";
        if (P.integrator == QC_EXPONENTIAL) return P.n > 16 ? (P.ell16 != nullptr ? "mfma32-exp-gather" : "mfma32-exp") : (P.ell16 != nullptr ? "mfma16-exp-gather" : "mfma16-exp");
        if (qc_mfma16_padeP_supported(P)) return "mfma16-padeP";
        if (P.n > 16 && P.n <= 32 && P.ell) return "mfma32-pade4-ell";
        return P.n > 32 ? "mfma64-pade4" : (P.n > 16 ? "mfma32-pade4" : "mfma16-pade4");
    }
    if (P.integrator != QC_PADE) {      // exponential integrator: mu_d2F alone; F + dF + mu_d2F as two launches
        if (which == 2) return "two-launches";
        if (mfma && qc_mfma_exp_hess_supported(P)) return P.ell16 != nullptr ? "mfma16-exp-hess-gather" : "mfma16-exp-hess";
        if (mfma && qc_mfma32_exp_hess_supported(P)) return P.ell16 != nullptr ? "mfma32-exp-hess-gather" : "mfma32-exp-hess";
        return P.use_ws ? "lds-gws-exp-hess" : "lds-exp-hess";
    }
    if (which == 2) return mfma && qc_mfma16_fused_supported(P) ? (qc_mfma16_fused_gathers(P) ? "mfma16-pade4-fused-gather" : "mfma16-pade4-fused") : (mfma && P.ell && P.hess_nnz ? "mfma32-pade4-fused-ell" : "two-launches");
    if (mfma && qc_mfma_hess_supported(P)) {
        if (qc_mfma16_padeP_hess_supported(P)) return "mfma16-padeP-hess";
        if (P.n == 16 && qc_mfma16_hess_gathers(P)) return "mfma16-pade4-hess-gather";      // one entry per drive-generator row: the one-wave kernel's row-gather form
        if (qc_mfma16_hess2_supported(P)) return "mfma16-pade4-hess2";
        if (P.n > 16 && P.n <= 32 && P.ell) return "mfma32-pade4-hess-ell";
        return P.n > 32 ? "mfma64-pade4-hess" : (P.n > 16 ? "mfma32-pade4-hess" : "mfma16-pade4-hess");
    }
    return P.use_ws ? "lds-gws-hess" : "lds-hess";
}

extern "C" int qc_dims(const qc_handle* h, qc_dims_t* out) {
    if (!h || !out) return fail(nullptr, QC_ERR_INVALID, "qc_dims: NULL argument");
    *out = h->dims;
    return QC_OK;
}

extern "C" int qc_jac_structure(const qc_handle* h, int64_t* rows, int64_t* cols, int one_based) {
    if (!h || !rows || !cols) return fail(nullptr, QC_ERR_INVALID, "qc_jac_structure: NULL argument");
    std::vector<int32_t> lr, lc;
    qc_local_jac_structure(h->prm, &lr, &lc);
    expand_structure(h->prm, lr, lc, h->prm.F_stride, h->prm.F_off, rows, cols, one_based);
    return QC_OK;
}

extern "C" int qc_hess_structure(const qc_handle* h, int64_t* rows, int64_t* cols, int one_based) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_hess_structure: NULL handle");
    if (h->prm.hess_nnz == 0) return QC_OK;
    if (!rows || !cols) return fail(nullptr, QC_ERR_INVALID, "qc_hess_structure: NULL output");
    std::vector<int32_t> lr, lc;
    qc_local_hess_structure(h->prm, &lr, &lc);
    expand_structure(h->prm, lr, lc, h->prm.zdim, 0, rows, cols, one_based);
    return QC_OK;
}

