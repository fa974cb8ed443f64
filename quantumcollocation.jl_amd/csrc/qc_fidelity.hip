// Fidelity of the final knot: value, gradient and dense upper-triangular Hessian of
//     F(U~) = |tr(U_goal' U)| / n   over a subspace block          (iso_vec_unitary_fidelity,
//     l(U~) = |1 - F(U~)|                                           unitary_minimum_time_problem.jl:77;
// the loss of UnitaryInfidelityObjective, docstring unitary_smooth_pulse_problem.jl:23-28).
// tr = g_r.u + i g_i.u with constant vectors g_r, g_i built from the goal, so
//     grad F = (t_r g_r + t_i g_i) / (n^2 F),   hess F = (g_r g_r^T + g_i g_i^T) / (n^2 F) - grad F grad F^T / F.
// One 256-thread workgroup; this is a few hundred FLOPs on 128..512 numbers: it exists so that a
// device-resident consumer needs no host round trip for the last knot, not for speed.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "qc_internal.h"

struct qc_fidelity {
    int N = 0, s = 0, n_sub = 0, device = 0, kind = QC_FID_UNITARY;
    int form = QC_FID_FORM_ABS;      // |tr| / n  or  |tr|^2 / n^2
    int K = 0;                       // free phases (global variables behind the state in the input vector)
    double *dgr = nullptr, *dgi = nullptr, *dU = nullptr, *dOut = nullptr;   // dOut: [value(2: F, l) | gradF (P) | hessF (P(P+1)/2)], P = s + K
    // free phases: m_r(u) = sum_p (CR + i CI)[r][p] u_p,  tr = sum_r exp(i theta_r) m_r,  theta_r = sum_k phi_k lam[k][r]
    double *dCR = nullptr, *dCI = nullptr, *dLam = nullptr, *dWork = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {

// Work array of the free-phase form (doubles): [ t (2) | a1 (P) | b1 (P) | a2uk (K s) | b2uk (K s) | a2kl (K K) | b2kl (K K) | mr (n) | mi (n) | c (n) | sn (n) ]
// a1 / b1 = d Re tr / dx, d Im tr / dx over x = [u ; phi];  a2uk / b2uk = d2 tr / (du dphi_k);  a2kl / b2kl = d2 tr / (dphi_k dphi_l).
__device__ inline size_t fid_work_doubles(int s, int K, int n) { return 2 + 2 * (size_t)(s + K) + 2 * (size_t)K * s + 2 * (size_t)K * K + 4 * (size_t)n; }

__global__ __launch_bounds__(256) void qc_fidelity_kernel(const double* __restrict__ u, const double* __restrict__ gr,
                                                          const double* __restrict__ gi, int s, int n_sub, int kind, int form, int K,
                                                          const double* __restrict__ CR, const double* __restrict__ CI,
                                                          const double* __restrict__ lam, double* __restrict__ work,
                                                          double* __restrict__ val, double* __restrict__ grad,
                                                          double* __restrict__ hess) {
    __shared__ double red[2][4];
    __shared__ double sg[2048];   // g_r, g_i staged (s <= 1024 handled through global otherwise)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double n = (double)n_sub;
    if (K > 0) {
        // ---------------- free phases: tr(U_goal' R(phi) U), R = V exp(i diag(theta)) V' ------------------------------
        const int P = s + K, nn = n_sub;
        double* tt = work;
        double* a1 = work + 2;
        double* b1 = a1 + P;
        double* a2uk = b1 + P;
        double* b2uk = a2uk + (size_t)K * s;
        double* a2kl = b2uk + (size_t)K * s;
        double* b2kl = a2kl + (size_t)K * K;
        double* mr = b2kl + (size_t)K * K;
        double* mi = mr + nn;
        double* cs = mi + nn;
        double* sn = cs + nn;
        const double* phi = u + s;
        for (int r = wave; r < nn; r += 4) {          // m_r: one wave per r
            double ar = 0.0, ai = 0.0;
            for (int p = lane; p < s; p += 64) {
                const double up = u[p];
                ar = fma(CR[(size_t)r * s + p], up, ar);
                ai = fma(CI[(size_t)r * s + p], up, ai);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                ar += __shfl_xor(ar, off, 64);
                ai += __shfl_xor(ai, off, 64);
            }
            if (lane == 0) {
                double th = 0.0;
                for (int k = 0; k < K; ++k) th = fma(phi[k], lam[(size_t)k * nn + r], th);
                mr[r] = ar; mi[r] = ai; cs[r] = cos(th); sn[r] = sin(th);
            }
        }
        __syncthreads();
        if (tid == 0) {                               // scalars: t, t_k, t_kl (fixed order: bit-reproducible)
            double tr_ = 0.0, ti_ = 0.0;
            for (int r = 0; r < nn; ++r) {
                tr_ += cs[r] * mr[r] - sn[r] * mi[r];
                ti_ += cs[r] * mi[r] + sn[r] * mr[r];
            }
            tt[0] = tr_; tt[1] = ti_;
            for (int k = 0; k < K; ++k) {
                double ak = 0.0, bk = 0.0;
                for (int r = 0; r < nn; ++r) {
                    const double wr = cs[r] * mr[r] - sn[r] * mi[r], wi = cs[r] * mi[r] + sn[r] * mr[r], l = lam[(size_t)k * nn + r];
                    ak -= l * wi;  bk += l * wr;      // i lam w
                }
                a1[s + k] = ak; b1[s + k] = bk;
                for (int l2 = 0; l2 < K; ++l2) {
                    double akl = 0.0, bkl = 0.0;
                    for (int r = 0; r < nn; ++r) {
                        const double wr = cs[r] * mr[r] - sn[r] * mi[r], wi = cs[r] * mi[r] + sn[r] * mr[r];
                        const double ll = lam[(size_t)k * nn + r] * lam[(size_t)l2 * nn + r];
                        akl -= ll * wr;  bkl -= ll * wi;
                    }
                    a2kl[k * K + l2] = akl; b2kl[k * K + l2] = bkl;
                }
            }
        }
        for (int p = tid; p < s; p += 256) {          // coefficient vectors of u at this phi, and their phi-derivatives
            double g_r = 0.0, g_i = 0.0;
            for (int r = 0; r < nn; ++r) {
                const double cr = CR[(size_t)r * s + p], ci = CI[(size_t)r * s + p];
                g_r += cs[r] * cr - sn[r] * ci;
                g_i += cs[r] * ci + sn[r] * cr;
            }
            a1[p] = g_r; b1[p] = g_i;
            for (int k = 0; k < K; ++k) {
                double hr = 0.0, hi = 0.0;
                for (int r = 0; r < nn; ++r) {
                    const double cr = CR[(size_t)r * s + p], ci = CI[(size_t)r * s + p], l = lam[(size_t)k * nn + r];
                    hr -= l * (cs[r] * ci + sn[r] * cr);
                    hi += l * (cs[r] * cr - sn[r] * ci);
                }
                a2uk[(size_t)k * s + p] = hr; b2uk[(size_t)k * s + p] = hi;
            }
        }
        __syncthreads();
        const double tr = tt[0], ti = tt[1];
        const double S = tr * tr + ti * ti, n2 = n * n;
        const double Fv = form == QC_FID_FORM_ABS2 ? S / n2 : sqrt(S) / n;
        if (tid == 0) { val[0] = Fv; val[1] = fabs(1.0 - Fv); }
        const double sc1 = form == QC_FID_FORM_ABS2 ? 2.0 / n2 : 1.0 / (n2 * Fv);
        if (grad) for (int p = tid; p < P; p += 256) grad[p] = (tr * a1[p] + ti * b1[p]) * sc1;
        if (!hess) return;
        const long long nh = (long long)P * (P + 1) / 2;
        for (long long e = tid; e < nh; e += 256) {
            int j = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while ((long long)(j + 1) * (j + 2) / 2 <= e) ++j;
            while ((long long)j * (j + 1) / 2 > e) --j;
            const int i = (int)(e - (long long)j * (j + 1) / 2);          // i <= j
            double a2 = 0.0, b2 = 0.0;                                    // second derivatives of (Re tr, Im tr)
            if (j >= s) {
                if (i >= s) { a2 = a2kl[(i - s) * K + (j - s)]; b2 = b2kl[(i - s) * K + (j - s)]; }
                else { a2 = a2uk[(size_t)(j - s) * s + i]; b2 = b2uk[(size_t)(j - s) * s + i]; }
            }
            const double q = a1[i] * a1[j] + b1[i] * b1[j] + tr * a2 + ti * b2;
            if (form == QC_FID_FORM_ABS2) hess[e] = 2.0 * q / n2;
            else {
                const double Fi = (tr * a1[i] + ti * b1[i]) * sc1, Fj = (tr * a1[j] + ti * b1[j]) * sc1;
                hess[e] = q * sc1 - Fi * Fj / Fv;
            }
        }
        return;
    }
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
    // unitary: F = |t| / n (or |t|^2 / n^2);   ket: F = |t|^2 (iso_fidelity);   density operator against a pure goal: F = Re t = psi' rho psi
    const bool sq = kind == QC_FID_UNITARY && form == QC_FID_FORM_ABS2;
    const double Fv = kind == QC_FID_UNITARY ? (sq ? (tr * tr + ti * ti) / (n * n) : sqrt(tr * tr + ti * ti) / n)
                                             : (kind == QC_FID_KET ? tr * tr + ti * ti : tr);
    const double inv = 1.0 / (n * n * Fv);
    if (tid == 0) {
        val[0] = Fv;
        val[1] = fabs(1.0 - Fv);
    }
    if (kind != QC_FID_UNITARY || sq) {
        const double two = sq ? 2.0 / (n * n) : 2.0;    // |t|^2 forms: grad = two (t_r g_r + t_i g_i), hess = two (g_r g_r' + g_i g_i')
        const bool quad = sq || kind == QC_FID_KET;
        for (int i = tid; i < s; i += 256) {
            if (grad) grad[i] = quad ? two * (tr * gr[i] + ti * gi[i]) : gr[i];
        }
        if (!hess) return;
        const long long nh2 = (long long)s * (s + 1) / 2;
        for (long long e = tid; e < nh2; e += 256) {
            int j = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while ((long long)(j + 1) * (j + 2) / 2 <= e) ++j;
            while ((long long)j * (j + 1) / 2 > e) --j;
            const int i = (int)(e - (long long)j * (j + 1) / 2);
            hess[e] = quad ? two * (gr[i] * gr[j] + gi[i] * gi[j]) : 0.0;
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

// Eigen-decomposition A = V diag(w) V' of a small complex Hermitian matrix (cyclic Jacobi, column-major d x d, re / im planes).
// Host-side set-up only (once per handle, like packing the generator images); d is the dimension of one phase operator (2 for a qubit).
static bool hermitian_eig(int d, const double* Are, const double* Aim, std::vector<double>* w, std::vector<double>* Vre, std::vector<double>* Vim) {
    std::vector<double> ar(Are, Are + (size_t)d * d), ai(Aim, Aim + (size_t)d * d);
    Vre->assign((size_t)d * d, 0.0);
    Vim->assign((size_t)d * d, 0.0);
    for (int k = 0; k < d; ++k) (*Vre)[(size_t)k * d + k] = 1.0;
    auto at = [d](std::vector<double>& m, int r, int c) -> double& { return m[(size_t)c * d + r]; };
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < d; ++p) for (int q = p + 1; q < d; ++q) off += at(ar, p, q) * at(ar, p, q) + at(ai, p, q) * at(ai, p, q);
        if (off < 1e-300) break;
        for (int p = 0; p < d; ++p)
            for (int q = p + 1; q < d; ++q) {
                const double xr = at(ar, p, q), xi = at(ai, p, q), mag = sqrt(xr * xr + xi * xi);
                if (mag == 0.0) continue;
                // unitary rotation in the (p, q) plane: columns p, q <- [c, -s e^{i ph}; s e^{-i ph}, c] with A_pq = mag e^{i ph}
                const double app = at(ar, p, p), aqq = at(ar, q, q);
                const double theta = 0.5 * atan2(2.0 * mag, app - aqq);
                const double c = cos(theta), sn = sin(theta);
                const double er = xr / mag, ei = xi / mag;                // e^{i ph}
                // J = [[c, -sn e^{i ph}], [sn e^{-i ph}, c]] acting on columns (p, q): A <- J' A J, V <- V J
                auto rot_cols = [&](std::vector<double>& mr, std::vector<double>& mi) {
                    for (int r = 0; r < d; ++r) {
                        const double pr = at(mr, r, p), pi = at(mi, r, p), qr = at(mr, r, q), qi = at(mi, r, q);
                        // new_p = c p + sn e^{-i ph} q ;  new_q = -sn e^{i ph} p + c q
                        at(mr, r, p) = c * pr + sn * (er * qr + ei * qi);
                        at(mi, r, p) = c * pi + sn * (er * qi - ei * qr);
                        at(mr, r, q) = -sn * (er * pr - ei * pi) + c * qr;
                        at(mi, r, q) = -sn * (er * pi + ei * pr) + c * qi;
                    }
                };
                rot_cols(ar, ai);
                for (int cc = 0; cc < d; ++cc) {   // rows: A <- J' A  (row p <- c row_p + sn e^{i ph} row_q ; row q <- -sn e^{-i ph} row_p + c row_q)
                    const double pr = at(ar, p, cc), pi = at(ai, p, cc), qr = at(ar, q, cc), qi = at(ai, q, cc);
                    at(ar, p, cc) = c * pr + sn * (er * qr - ei * qi);
                    at(ai, p, cc) = c * pi + sn * (er * qi + ei * qr);
                    at(ar, q, cc) = -sn * (er * pr + ei * pi) + c * qr;
                    at(ai, q, cc) = -sn * (er * pi - ei * pr) + c * qi;
                }
                rot_cols(*Vre, *Vim);
            }
    }
    w->resize(d);
    for (int k = 0; k < d; ++k) (*w)[k] = at(ar, k, k);
    double off = 0.0, dia = 0.0;
    for (int p = 0; p < d; ++p) for (int q = 0; q < d; ++q) (p == q ? dia : off) += at(ar, p, q) * at(ar, p, q) + at(ai, p, q) * at(ai, p, q);
    return off <= 1e-24 * (dia > 0.0 ? dia : 1.0);
}

extern "C" int qc_hermitian_eig(int32_t d, const double* A_re, const double* A_im, double* w, double* V_re, double* V_im) {
    if (d < 1 || d > 64 || !A_re || !A_im || !w || !V_re || !V_im) return ffail(nullptr, QC_ERR_INVALID, "qc_hermitian_eig: bad argument");
    std::vector<double> ww, vr, vi;
    if (!hermitian_eig(d, A_re, A_im, &ww, &vr, &vi)) return ffail(nullptr, QC_ERR_INVALID, "qc_hermitian_eig: did not converge");
    memcpy(w, ww.data(), (size_t)d * 8);
    memcpy(V_re, vr.data(), (size_t)d * d * 8);
    memcpy(V_im, vi.data(), (size_t)d * d * 8);
    return QC_OK;
}

extern "C" int qc_fidelity_create_desc(const qc_fidelity_desc* d, qc_fidelity** out) {
    if (!out) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_desc: out is NULL");
    *out = nullptr;
    if (!d) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_desc: descriptor is NULL");
    if (d->kind == QC_FID_KET || d->kind == QC_FID_DENSITY) {
        if (d->n_phases || d->subspace || d->form != QC_FID_FORM_ABS)
            return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_desc: subspace, form and free phases apply to QC_FID_UNITARY only");
        return qc_fidelity_create_kind(d->kind, d->N, d->goal_iso, d->device, out);
    }
    if (d->kind != QC_FID_UNITARY) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create_desc: unknown kind");
    const int N = d->N;
    if (N < 1 || N > 64 || !d->goal_iso) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: bad N or goal");
    if (d->subspace && (d->n_sub < 1 || d->n_sub > N)) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: bad subspace size");
    if (d->form != QC_FID_FORM_ABS && d->form != QC_FID_FORM_ABS2) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: unknown form");
    std::vector<int> sub;
    if (d->subspace) {
        for (int k = 0; k < d->n_sub; ++k) {
            if (d->subspace[k] < 0 || d->subspace[k] >= N) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: subspace index out of range");
            sub.push_back(d->subspace[k]);
        }
    } else {
        for (int k = 0; k < N; ++k) sub.push_back(k);
    }
    const int n = (int)sub.size(), K = d->n_phases;
    if (K < 0 || K > 16) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: n_phases must be in 0..16");
    // ---- free phases: simultaneous eigenbasis of the commuting operators O_k = I (x) .. Op_k .. (x) I on the subspace ----
    std::vector<double> Vr, Vi, lam;        // V (n x n, column-major), lam[k][r]
    if (K > 0) {
        if (!d->phase_dims || !d->phase_ops) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: phase_dims / phase_ops missing");
        long long prod = 1;
        for (int k = 0; k < K; ++k) {
            if (d->phase_dims[k] < 1 || d->phase_dims[k] > 64) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: bad phase operator dimension");
            prod *= d->phase_dims[k];
        }
        if (prod != n) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: the phase operators' dimensions must multiply to the subspace size");
        Vr.assign(1, 1.0); Vi.assign(1, 0.0);
        lam.assign((size_t)K * n, 0.0);
        int cur = 1;                        // dimension of the Kronecker product built so far
        const double* op = d->phase_ops;
        std::vector<std::vector<double>> ws(K);
        for (int k = 0; k < K; ++k) {
            const int dk = d->phase_dims[k];
            for (int c = 0; c < dk; ++c)
                for (int r = 0; r < dk; ++r) {
                    const double re = op[(size_t)c * dk + r], im = op[(size_t)dk * dk + (size_t)c * dk + r];
                    const double re_t = op[(size_t)r * dk + c], im_t = op[(size_t)dk * dk + (size_t)r * dk + c];
                    if (fabs(re - re_t) > 1e-12 * (1.0 + fabs(re)) || fabs(im + im_t) > 1e-12 * (1.0 + fabs(im)))
                        return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: phase operators must be Hermitian");
                }
            std::vector<double> vr, vi;
            if (!hermitian_eig(dk, op, op + (size_t)dk * dk, &ws[k], &vr, &vi))
                return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_create: eigen-decomposition of a phase operator did not converge");
            // V <- V (x) V_k  (first operator = most significant index, Julia's kron / reduce(kron, ...))
            std::vector<double> nr((size_t)cur * dk * cur * dk), ni(nr.size());
            for (int c1 = 0; c1 < cur; ++c1) for (int c2 = 0; c2 < dk; ++c2)
                for (int r1 = 0; r1 < cur; ++r1) for (int r2 = 0; r2 < dk; ++r2) {
                    const double ar = Vr[(size_t)c1 * cur + r1], ai = Vi[(size_t)c1 * cur + r1];
                    const double br = vr[(size_t)c2 * dk + r2], bi = vi[(size_t)c2 * dk + r2];
                    const size_t o = (size_t)(c1 * dk + c2) * (cur * dk) + (r1 * dk + r2);
                    nr[o] = ar * br - ai * bi;
                    ni[o] = ar * bi + ai * br;
                }
            Vr.swap(nr); Vi.swap(ni);
            cur *= dk;
            op += 2 * (size_t)dk * dk;
        }
        // eigenvalue of O_k on composite index r = (r_1, ..., r_K), r_1 most significant
        for (int r = 0; r < n; ++r) {
            int rem = r;
            for (int k = K - 1; k >= 0; --k) {
                const int dk = d->phase_dims[k];
                lam[(size_t)k * n + r] = ws[k][rem % dk];
                rem /= dk;
            }
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ffail(nullptr, QC_ERR_NO_DEVICE, "qc_fidelity_create: no HIP device visible");
    if (d->device < 0 || d->device >= ndev) return ffail(nullptr, QC_ERR_NO_DEVICE, "qc_fidelity_create: device ordinal out of range");
    qc_fidelity* h = new qc_fidelity();
    h->N = N;
    h->s = 2 * N * N;
    h->n_sub = n;
    h->device = d->device;
    h->form = d->form;
    h->K = K;
    const double* goal_iso = d->goal_iso;
    std::vector<double> gr(h->s, 0.0), gi(h->s, 0.0);
    for (int j : sub)
        for (int i : sub) {
            const int re = j * 2 * N + i, im = j * 2 * N + N + i;
            const double Gre = goal_iso[re], Gim = goal_iso[im];
            gr[re] = Gre;  gr[im] = Gim;
            gi[re] = -Gim; gi[im] = Gre;
        }
    std::vector<double> CR, CI;
    if (K > 0) {
        // m_r = sum_{a,b} conj(V_ar) U_ab W_br,  W = G' V  (all on the subspace);  d m_r / d Re U_ab = c, d m_r / d Im U_ab = i c
        std::vector<double> Wr((size_t)n * n, 0.0), Wi((size_t)n * n, 0.0);      // W[b][r], column r
        for (int r = 0; r < n; ++r)
            for (int b = 0; b < n; ++b) {
                double sr = 0.0, si = 0.0;
                for (int c = 0; c < n; ++c) {     // conj(G[c][b]) V[c][r]
                    const double gre = goal_iso[sub[b] * 2 * N + sub[c]], gim = goal_iso[sub[b] * 2 * N + N + sub[c]];
                    const double vr = Vr[(size_t)r * n + c], vi = Vi[(size_t)r * n + c];
                    sr += gre * vr + gim * vi;
                    si += gre * vi - gim * vr;
                }
                Wr[(size_t)r * n + b] = sr; Wi[(size_t)r * n + b] = si;
            }
        CR.assign((size_t)n * h->s, 0.0);
        CI.assign((size_t)n * h->s, 0.0);
        for (int r = 0; r < n; ++r)
            for (int b = 0; b < n; ++b)
                for (int a = 0; a < n; ++a) {
                    const double var = Vr[(size_t)r * n + a], vai = -Vi[(size_t)r * n + a];       // conj(V_ar)
                    const double cr = var * Wr[(size_t)r * n + b] - vai * Wi[(size_t)r * n + b];
                    const double ci = var * Wi[(size_t)r * n + b] + vai * Wr[(size_t)r * n + b];
                    const int re = sub[b] * 2 * N + sub[a], im = re + N;
                    CR[(size_t)r * h->s + re] = cr;  CI[(size_t)r * h->s + re] = ci;       // d/d Re U_ab = c
                    CR[(size_t)r * h->s + im] = -ci; CI[(size_t)r * h->s + im] = cr;       // d/d Im U_ab = i c
                }
    }
    auto bail = [&](hipError_t e, const char* what) { std::string m = std::string(what) + ": " + hipGetErrorString(e); qc_fidelity_destroy(h); return ffail(nullptr, QC_ERR_HIP, m); };
    hipError_t e;
    qc_device_guard guard(d->device);
    if (guard.err != hipSuccess) return bail(guard.err, "hipSetDevice");
    const size_t P = (size_t)h->s + K;
    const size_t nout = 2 + P + P * (P + 1) / 2;
    if ((e = hipMalloc((void**)&h->dgr, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dgi, h->s * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dU, P * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&h->dOut, nout * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpy(h->dgr, gr.data(), h->s * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipMemcpy(h->dgi, gi.data(), h->s * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if (K > 0) {
        const size_t nw = 2 + 2 * P + 2 * (size_t)K * h->s + 2 * (size_t)K * K + 4 * (size_t)n;
        if ((e = hipMalloc((void**)&h->dCR, CR.size() * 8)) != hipSuccess) return bail(e, "hipMalloc");
        if ((e = hipMalloc((void**)&h->dCI, CI.size() * 8)) != hipSuccess) return bail(e, "hipMalloc");
        if ((e = hipMalloc((void**)&h->dLam, lam.size() * 8)) != hipSuccess) return bail(e, "hipMalloc");
        if ((e = hipMalloc((void**)&h->dWork, nw * 8)) != hipSuccess) return bail(e, "hipMalloc");
        if ((e = hipMemcpy(h->dCR, CR.data(), CR.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
        if ((e = hipMemcpy(h->dCI, CI.data(), CI.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
        if ((e = hipMemcpy(h->dLam, lam.data(), lam.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    }
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    *out = h;
    return QC_OK;
}

extern "C" int qc_fidelity_create(int32_t N, const double* goal_iso, const int32_t* subspace, int32_t n_sub, int32_t device,
                                  qc_fidelity** out) {
    qc_fidelity_desc d;
    memset(&d, 0, sizeof(d));
    d.kind = QC_FID_UNITARY;
    d.N = N;
    d.goal_iso = goal_iso;
    d.subspace = subspace;
    d.n_sub = n_sub;
    d.device = device;
    return qc_fidelity_create_desc(&d, out);
}

extern "C" int32_t qc_fidelity_input_len(const qc_fidelity* h) { return h ? h->s + h->K : 0; }

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
    qc_device_guard guard(device);
    if (guard.err != hipSuccess) return bail(guard.err, "hipSetDevice");
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
    qc_device_guard guard(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (double* p : {h->dgr, h->dgi, h->dU, h->dOut, h->dCR, h->dCI, h->dLam, h->dWork}) if (p) (void)hipFree(p);
    delete h;
}

extern "C" int qc_fidelity_eval_dev(qc_fidelity* h, const double* dU, double* dval2, double* dgrad, double* dhess, void* stream) {
    if (!h) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_eval_dev: NULL handle");
    if (!dU || !dval2) return ffail(h, QC_ERR_INVALID, "qc_fidelity_eval_dev: NULL buffer");
    qc_device_guard guard(h->device);
    if (guard.err != hipSuccess) return ffail(h, QC_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    hipLaunchKernelGGL(qc_fidelity_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dU, h->dgr, h->dgi, h->s, h->n_sub, h->kind, h->form, h->K,
                       h->dCR, h->dCI, h->dLam, h->dWork, dval2, dgrad, dhess);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ffail(h, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

extern "C" int qc_fidelity_eval(qc_fidelity* h, const double* U_iso, double* fidelity, double* infidelity, double* grad, double* hess) {
    if (!h) return ffail(nullptr, QC_ERR_INVALID, "qc_fidelity_eval: NULL handle");
    if (!U_iso) return ffail(h, QC_ERR_INVALID, "qc_fidelity_eval: NULL input");
    qc_device_guard guard(h->device);
    QCF_HIP(h, guard.err);
    const size_t P = (size_t)h->s + h->K;    // input = [state ; free phases]
    QCF_HIP(h, hipMemcpyAsync(h->dU, U_iso, P * 8, hipMemcpyHostToDevice, h->stream));
    double* dval = h->dOut;
    double* dgrad = h->dOut + 2;
    double* dhess = h->dOut + 2 + P;
    int rc = qc_fidelity_eval_dev(h, h->dU, dval, grad ? dgrad : nullptr, hess ? dhess : nullptr, h->stream);
    if (rc) return rc;
    double v[2];
    QCF_HIP(h, hipMemcpyAsync(v, dval, 16, hipMemcpyDeviceToHost, h->stream));
    if (grad) QCF_HIP(h, hipMemcpyAsync(grad, dgrad, P * 8, hipMemcpyDeviceToHost, h->stream));
    if (hess) QCF_HIP(h, hipMemcpyAsync(hess, dhess, P * (P + 1) / 2 * 8, hipMemcpyDeviceToHost, h->stream));
    QCF_HIP(h, hipStreamSynchronize(h->stream));
    if (fidelity) *fidelity = v[0];
    if (infidelity) *infidelity = v[1];
    return QC_OK;
}
