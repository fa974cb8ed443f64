/*
 * CPU ORACLE in C (test infrastructure, NOT product code) — a second restatement of the knot-point
 * dynamics evaluator, used (a) to cross-check oracle/qc_oracle.py and (b) as bench.py's
 * `cpu_baseline` (kind "port": the reference evaluator is Julia inside the un-vendored
 * QuantumCollocationCore 0.3 and cannot be built or run here).
 *
 * PARITY UNPINNED against the reference (no golden vectors exist for this path: SURVEY.md §8c);
 * pinned against mathematics through oracle/qc_oracle.py (tests/test_oracle_math.py,
 * tests/test_oracle_c.py).
 *
 * Follows, per interval and like the reference's per-knot loop (`Threads.@threads for t = 1:T-1`
 * [RECALL], call shapes reference test/scripts/integrator_test_1qubit.jl:41-52):
 *   G(a_t) assembly                      PiccoloQuantumObjects QuantumSystem.G      (call site unitary_smooth_pulse_problem.jl:199)
 *   B, F = I -+ c1 h G + c2 h^2 G^2 ...  UnitaryPadeIntegrator                      (unitary_smooth_pulse_problem.jl:14,165-167)
 *   delta = B U_{t+1} - F U_t            README.md:74-80 / docstring :10-30
 *   d/dU blocks I_N (x) B, -I_N (x) F;  d/da_j, d/dh columns                      
 *   x_{t+1} - x_t - h dx_t               DerivativeIntegrator                       (:15-16,177-178)
 *   mu-contracted Hessian blocks                                                    (exponential: interval_hess_exp)
 * Value order = the canonical block order of oracle/qc_oracle.py::jac_structure_local.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QCO_MAX_DERIV 8
#define QCO_MAX_P 10

typedef struct {
    int N, m;
    long long T;
    int zdim, off_U, off_a, off_dt;
    double dt_fixed;
    int integrator; /* 0 = Pade, 1 = exponential */
    int order;
    int n_deriv;
    int x_off[QCO_MAX_DERIV], dx_off[QCO_MAX_DERIV], ddim[QCO_MAX_DERIV];
    const double* G_drift;  /* n*n col-major */
    const double* G_drives; /* m * n*n */
    int ncol;               /* columns of the iso state: 0 -> N (unitary), K -> K kets back to back */
} qco_problem;

static int qco_nc(const qco_problem* P) { return P->ncol > 0 ? P->ncol : P->N; }

/* C (r x c) = A (r x k) * B (k x c), column-major */
static void mm(double* C, const double* A, const double* B, int r, int k, int c) {
    for (int j = 0; j < c; ++j) {
        double* Cj = C + (size_t)j * r;
        for (int i = 0; i < r; ++i) Cj[i] = 0.0;
        for (int l = 0; l < k; ++l) {
            const double b = B[(size_t)j * k + l];
            const double* Al = A + (size_t)l * r;
            for (int i = 0; i < r; ++i) Cj[i] += Al[i] * b;
        }
    }
}
/* C (k x c) = A^T (A is r x k) * B (r x c) */
static void mtm(double* C, const double* A, const double* B, int r, int k, int c) {
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < k; ++i) {
            double acc = 0.0;
            for (int l = 0; l < r; ++l) acc += A[(size_t)i * r + l] * B[(size_t)j * r + l];
            C[(size_t)j * k + i] = acc;
        }
}
static double dot(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static void pade_coeffs(int order, double* c) {
    const int p = order / 2;
    c[0] = 1.0;
    for (int k = 1; k <= p; ++k) c[k] = c[k - 1] * (double)(p - k + 1) / ((double)k * (double)(2 * p - k + 1));
}

int qco_ddim(const qco_problem* P) {
    int d = 2 * P->N * qco_nc(P);
    for (int i = 0; i < P->n_deriv; ++i) d += P->ddim[i];
    return d;
}
int qco_jac_nnz(const qco_problem* P) {
    const int n = 2 * P->N, nc = qco_nc(P), s = n * nc, ft = P->off_dt >= 0;
    int o = nc * n * n + (P->integrator == 0 ? nc * n * n : s) + s * P->m + (ft ? s : 0);
    for (int i = 0; i < P->n_deriv; ++i) o += (ft ? 4 : 3) * P->ddim[i];
    return o;
}
int qco_hess_nnz(const qco_problem* P) {
    /* the exponential integrator is linear in U_{t+1}: no (a, U_{t+1}), (h, U_{t+1}) blocks */
    const int n = 2 * P->N, s = n * qco_nc(P), m = P->m, ft = P->off_dt >= 0, ub = P->integrator == 0 ? 2 : 1;
    int o = ub * s * m + m * (m + 1) / 2;
    if (ft) {
        o += m + ub * s + 1;
        for (int i = 0; i < P->n_deriv; ++i) o += P->ddim[i];
    }
    return o;
}

typedef struct {
    double *G, *Gp, *B, *F, *dB, *dF, *T1, *T2, *T3, *X1, *X2, *big, *bigE, *bigT;
} qco_ws;

static int ws_alloc(qco_ws* w, int n, int p) {
    const size_t n2 = (size_t)n * n;
    memset(w, 0, sizeof(*w));
    w->G = malloc(n2 * 8);
    w->Gp = malloc(n2 * 8 * (size_t)(p + 2));
    w->B = malloc(n2 * 8); w->F = malloc(n2 * 8); w->dB = malloc(n2 * 8); w->dF = malloc(n2 * 8);
    w->T1 = malloc(n2 * 8); w->T2 = malloc(n2 * 8); w->T3 = malloc(n2 * 8);
    w->X1 = malloc(n2 * 8); w->X2 = malloc(n2 * 8);
    w->big = malloc(4 * n2 * 8); w->bigE = malloc(4 * n2 * 8); w->bigT = malloc(4 * n2 * 8);
    return w->bigT != NULL;
}
static void ws_free(qco_ws* w) {
    free(w->G); free(w->Gp); free(w->B); free(w->F); free(w->dB); free(w->dF); free(w->T1); free(w->T2); free(w->T3);
    free(w->X1); free(w->X2); free(w->big); free(w->bigE); free(w->bigT);
}

/* E = exp(X), k x k, scaled Taylor (degree 24) + squaring; T1/T2 are k*k scratch */
static void expm_taylor(double* E, const double* X, int k, double* T1, double* T2) {
    const size_t k2 = (size_t)k * k;
    double nrm = 0.0;
    for (int j = 0; j < k; ++j) {
        double cs = 0.0;
        for (int i = 0; i < k; ++i) cs += fabs(X[(size_t)j * k + i]);
        if (cs > nrm) nrm = cs;
    }
    int sq = 0;
    if (nrm > 0.25) { sq = (int)ceil(log2(nrm / 0.25)); if (sq < 0) sq = 0; }
    const double sc = ldexp(1.0, -sq);
    for (size_t i = 0; i < k2; ++i) { T1[i] = 0.0; E[i] = 0.0; }
    for (int i = 0; i < k; ++i) { T1[(size_t)i * k + i] = 1.0; E[(size_t)i * k + i] = 1.0; }
    for (int d = 1; d < 25; ++d) {
        mm(T2, T1, X, k, k, k);
        for (size_t i = 0; i < k2; ++i) { T1[i] = T2[i] * sc / d; E[i] += T1[i]; }
    }
    for (int q = 0; q < sq; ++q) { mm(T2, E, E, k, k, k); memcpy(E, T2, k2 * 8); }
}

/* dGpow[k](Gj) = sum_{i<k} G^i Gj G^{k-1-i}; accumulate coefB*..., coefF*... into dB, dF */
static void accumulate_dpow(const qco_ws* w, const double* Gj, int n, int k, double cb, double cf, double* dB, double* dF,
                            double* T1, double* T2) {
    const size_t n2 = (size_t)n * n;
    for (int i = 0; i < k; ++i) {
        const double* L = Gj;   /* G^i Gj G^{k-1-i}; products with G^0 = I are skipped */
        if (i > 0) { mm(T1, w->Gp + (size_t)i * n2, Gj, n, n, n); L = T1; }
        if (k - 1 - i > 0) { mm(T2, L, w->Gp + (size_t)(k - 1 - i) * n2, n, n, n); L = T2; }
        for (size_t e = 0; e < n2; ++e) { dB[e] += cb * L[e]; dF[e] += cf * L[e]; }
    }
}

static void interval_F_jac(const qco_problem* P, const qco_ws* w, const double* z0, const double* z1, double* Fo, double* Jo) {
    const int n = 2 * P->N, N = qco_nc(P) /* state columns */, s = n * N, m = P->m, ft = P->off_dt >= 0;
    const size_t n2 = (size_t)n * n;
    const double h = ft ? z0[P->off_dt] : P->dt_fixed;
    const double* U0 = z0 + P->off_U;
    const double* U1 = z1 + P->off_U;
    const double* a = z0 + P->off_a;
    memcpy(w->G, P->G_drift, n2 * 8);
    for (int j = 0; j < m; ++j)
        for (size_t e = 0; e < n2; ++e) w->G[e] += a[j] * P->G_drives[(size_t)j * n2 + e];
    int jo = 0;
    if (P->integrator == 0) {
        const int p = P->order / 2;
        double c[QCO_MAX_P + 1];
        pade_coeffs(P->order, c);
        /* powers G^0..G^p */
        memset(w->Gp, 0, n2 * 8);
        for (int i = 0; i < n; ++i) w->Gp[(size_t)i * n + i] = 1.0;
        for (int k = 1; k <= p; ++k) mm(w->Gp + (size_t)k * n2, w->Gp + (size_t)(k - 1) * n2, w->G, n, n, n);
        memset(w->B, 0, n2 * 8); memset(w->F, 0, n2 * 8);
        memset(w->dB, 0, n2 * 8); memset(w->dF, 0, n2 * 8); /* d/dh */
        double hk = 1.0;
        for (int k = 0; k <= p; ++k) {
            const double sg = (k & 1) ? -1.0 : 1.0;
            const double* Gk = w->Gp + (size_t)k * n2;
            for (size_t e = 0; e < n2; ++e) { w->B[e] += sg * c[k] * hk * Gk[e]; w->F[e] += c[k] * hk * Gk[e]; }
            if (k + 1 <= p) {
                const double* Gk1 = w->Gp + (size_t)(k + 1) * n2;
                const double dc = c[k + 1] * (k + 1) * hk; /* (k+1) c_{k+1} h^k */
                for (size_t e = 0; e < n2; ++e) { w->dB[e] += -sg * dc * Gk1[e]; w->dF[e] += dc * Gk1[e]; }
            }
            hk *= h;
        }
        /* residual */
        if (Fo) {
            mm(w->X1, w->B, U1, n, n, N);
            mm(w->X2, w->F, U0, n, n, N);
            for (int e = 0; e < s; ++e) Fo[e] = w->X1[e] - w->X2[e];
        }
        if (Jo) {
            for (int q = 0; q < N; ++q) for (size_t e = 0; e < n2; ++e) Jo[jo + (size_t)q * n2 + e] = -w->F[e];
            jo += N * (int)n2;
            for (int q = 0; q < N; ++q) for (size_t e = 0; e < n2; ++e) Jo[jo + (size_t)q * n2 + e] = w->B[e];
            jo += N * (int)n2;
            double* dh_col = NULL;
            if (ft) {
                mm(w->X1, w->dB, U1, n, n, N);
                mm(w->X2, w->dF, U0, n, n, N);
                dh_col = Jo + jo + (size_t)s * m;
                for (int e = 0; e < s; ++e) dh_col[e] = w->X1[e] - w->X2[e];
            }
            for (int j = 0; j < m; ++j) {
                const double* Gj = P->G_drives + (size_t)j * n2;
                memset(w->dB, 0, n2 * 8); memset(w->dF, 0, n2 * 8);
                double hq = 1.0;
                for (int k = 1; k <= p; ++k) {
                    hq *= h;
                    accumulate_dpow(w, Gj, n, k, ((k & 1) ? -1.0 : 1.0) * c[k] * hq, c[k] * hq, w->dB, w->dF, w->T1, w->T2);
                }
                mm(w->X1, w->dB, U1, n, n, N);
                mm(w->X2, w->dF, U0, n, n, N);
                for (int e = 0; e < s; ++e) Jo[jo + (size_t)j * s + e] = w->X1[e] - w->X2[e];
            }
            jo += s * m + (ft ? s : 0);
        }
    } else {
        /* exponential integrator: delta = U1 - exp(hG) U0 (README.md:79) */
        double* E = w->B;
        for (size_t e = 0; e < n2; ++e) w->T3[e] = h * w->G[e];
        expm_taylor(E, w->T3, n, w->T1, w->T2);
        mm(w->X1, E, U0, n, n, N);
        if (Fo) for (int e = 0; e < s; ++e) Fo[e] = U1[e] - w->X1[e];
        if (Jo) {
            for (int q = 0; q < N; ++q) for (size_t e = 0; e < n2; ++e) Jo[jo + (size_t)q * n2 + e] = -E[e];
            jo += N * (int)n2;
            for (int e = 0; e < s; ++e) Jo[jo + e] = 1.0;
            jo += s;
            const int k2 = 2 * n;
            for (int j = 0; j < m; ++j) {
                const double* Gj = P->G_drives + (size_t)j * n2;
                memset(w->big, 0, (size_t)k2 * k2 * 8);
                for (int cc = 0; cc < n; ++cc)
                    for (int rr = 0; rr < n; ++rr) {
                        w->big[(size_t)cc * k2 + rr] = h * w->G[(size_t)cc * n + rr];
                        w->big[(size_t)(cc + n) * k2 + n + rr] = h * w->G[(size_t)cc * n + rr];
                        w->big[(size_t)(cc + n) * k2 + rr] = h * Gj[(size_t)cc * n + rr];
                    }
                /* workspace was allocated for 2n x 2n matrices: dB/dF are free scratch in this branch */
                expm_taylor(w->bigE, w->big, k2, w->bigT, w->dB);
                for (int cc = 0; cc < n; ++cc)
                    for (int rr = 0; rr < n; ++rr) w->T1[(size_t)cc * n + rr] = w->bigE[(size_t)(cc + n) * k2 + rr];
                mm(w->X2, w->T1, U0, n, n, N);
                for (int e = 0; e < s; ++e) Jo[jo + (size_t)j * s + e] = -w->X2[e];
            }
            jo += s * m;
            if (ft) {
                mm(w->X2, w->G, w->X1, n, n, N);
                for (int e = 0; e < s; ++e) Jo[jo + e] = -w->X2[e];
                jo += s;
            }
        }
    }
    int r0 = s;
    for (int d = 0; d < P->n_deriv; ++d) {
        const int dim = P->ddim[d];
        for (int i = 0; i < dim; ++i) {
            const double dx = z0[P->dx_off[d] + i];
            if (Fo) Fo[r0 + i] = z1[P->x_off[d] + i] - z0[P->x_off[d] + i] - h * dx;
            if (Jo) {
                Jo[jo + i] = -1.0;
                Jo[jo + dim + i] = 1.0;
                Jo[jo + 2 * dim + i] = -h;
                if (ft) Jo[jo + 3 * dim + i] = -dx;
            }
        }
        r0 += dim;
        jo += (ft ? 4 : 3) * dim;
    }
}

int qco_eval_F_jac(const qco_problem* P, const double* Z, double* F, double* J, long long t_begin, long long t_end,
                   int nthreads) {
    const int n = 2 * P->N, p = P->integrator == 0 ? P->order / 2 : 1;
    const int ddim = qco_ddim(P), nnz = qco_jac_nnz(P);
    int ok = 1;
    if (P->integrator == 0 && (P->order < 2 || (P->order & 1) || P->order / 2 > QCO_MAX_P)) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        qco_ws w;
        /* the 2n x 2n Frechet block needs its own Taylor scratch: allocate generously */
        if (!ws_alloc(&w, 2 * n, p)) {
#pragma omp atomic write
            ok = 0;
        } else {
#pragma omp for schedule(static)
            for (long long t = t_begin; t < t_end; ++t) {
                const double* z0 = Z + (size_t)t * P->zdim;
                interval_F_jac(P, &w, z0, z0 + P->zdim, F ? F + (size_t)(t - t_begin) * ddim : NULL,
                               J ? J + (size_t)(t - t_begin) * nnz : NULL);
            }
        }
        ws_free(&w);
    }
    return ok ? 0 : -2;
}

/* ---- Hessian of mu^T delta (Pade), canonical order of hess_structure_local ---------------------- */
static void d2pow_accumulate(const qco_ws* w, const double* Gi, const double* Gj, int n, int k, double cb, double cf,
                             double* dB, double* dF, double* T1, double* T2, double* T3) {
    const size_t n2 = (size_t)n * n;
    for (int al = 0; al <= k - 2; ++al)
        for (int be = 0; be <= k - 2 - al; ++be) {
            const int ga = k - 2 - al - be;
            for (int pass = 0; pass < 2; ++pass) {
                const double* A = pass ? Gj : Gi;
                const double* Bm = pass ? Gi : Gj;
                mm(T1, w->Gp + (size_t)al * n2, A, n, n, n);
                mm(T2, T1, w->Gp + (size_t)be * n2, n, n, n);
                mm(T1, T2, Bm, n, n, n);
                mm(T3, T1, w->Gp + (size_t)ga * n2, n, n, n);
                for (size_t e = 0; e < n2; ++e) { dB[e] += cb * T3[e]; dF[e] += cf * T3[e]; }
            }
        }
}

static void interval_hess(const qco_problem* P, const qco_ws* w, const double* z0, const double* z1, const double* mu, double* Ho) {
    const int n = 2 * P->N, N = qco_nc(P) /* state columns */, s = n * N, m = P->m, ft = P->off_dt >= 0;
    const size_t n2 = (size_t)n * n;
    const int p = P->order / 2;
    const double h = ft ? z0[P->off_dt] : P->dt_fixed;
    const double* U0 = z0 + P->off_U;
    const double* U1 = z1 + P->off_U;
    const double* a = z0 + P->off_a;
    const double* M = mu; /* n x N */
    double c[QCO_MAX_P + 1];
    pade_coeffs(P->order, c);
    memcpy(w->G, P->G_drift, n2 * 8);
    for (int j = 0; j < m; ++j)
        for (size_t e = 0; e < n2; ++e) w->G[e] += a[j] * P->G_drives[(size_t)j * n2 + e];
    memset(w->Gp, 0, n2 * 8);
    for (int i = 0; i < n; ++i) w->Gp[(size_t)i * n + i] = 1.0;
    for (int k = 1; k <= p; ++k) mm(w->Gp + (size_t)k * n2, w->Gp + (size_t)(k - 1) * n2, w->G, n, n, n);
    /* hess_structure_local's order: (U, a) | (a, U) | (U, h) | (h, U) | (a, a) | (a, h) | (h, h) | (dx, h) */
    int o_Ua = 0, o_aU = s * m, o_Uh = 2 * s * m, o_hU = o_Uh + (ft ? s : 0), o_aa = o_hU + (ft ? s : 0);
    int o_ah = o_aa + m * (m + 1) / 2, o_hh = o_ah + (ft ? m : 0), o_d = o_hh + (ft ? 1 : 0);
    for (int j = 0; j < m; ++j) {
        const double* Gj = P->G_drives + (size_t)j * n2;
        /* (U, a_j): vec(dB^T M), -vec(dF^T M) */
        memset(w->dB, 0, n2 * 8); memset(w->dF, 0, n2 * 8);
        double hq = 1.0;
        for (int k = 1; k <= p; ++k) {
            hq *= h;
            accumulate_dpow(w, Gj, n, k, ((k & 1) ? -1.0 : 1.0) * c[k] * hq, c[k] * hq, w->dB, w->dF, w->T1, w->T2);
        }
        mtm(w->X1, w->dB, M, n, n, N);
        mtm(w->X2, w->dF, M, n, n, N);
        for (int e = 0; e < s; ++e) { Ho[o_aU + (size_t)j * s + e] = w->X1[e]; Ho[o_Ua + (size_t)j * s + e] = -w->X2[e]; }
        if (ft) { /* (a_j, h) */
            memset(w->dB, 0, n2 * 8); memset(w->dF, 0, n2 * 8);
            double hk1 = 1.0;
            for (int k = 1; k <= p; ++k) {
                accumulate_dpow(w, Gj, n, k, ((k & 1) ? -1.0 : 1.0) * c[k] * k * hk1, c[k] * k * hk1, w->dB, w->dF, w->T1, w->T2);
                hk1 *= h;
            }
            mm(w->X1, w->dB, U1, n, n, N);
            mm(w->X2, w->dF, U0, n, n, N);
            for (int e = 0; e < s; ++e) w->X1[e] -= w->X2[e];
            Ho[o_ah + j] = dot(M, w->X1, s);
        }
        /* (a_i, a_j), i <= j */
        for (int i = 0; i <= j; ++i) {
            const double* Gi = P->G_drives + (size_t)i * n2;
            memset(w->dB, 0, n2 * 8); memset(w->dF, 0, n2 * 8);
            double hq2 = h;
            for (int k = 2; k <= p; ++k) {
                hq2 *= h;
                d2pow_accumulate(w, Gi, Gj, n, k, ((k & 1) ? -1.0 : 1.0) * c[k] * hq2, c[k] * hq2, w->dB, w->dF, w->T1, w->T2, w->T3);
            }
            mm(w->X1, w->dB, U1, n, n, N);
            mm(w->X2, w->dF, U0, n, n, N);
            for (int e = 0; e < s; ++e) w->X1[e] -= w->X2[e];
            Ho[o_aa + j * (j + 1) / 2 + i] = dot(M, w->X1, s);
        }
    }
    if (ft) {
        memset(w->dB, 0, n2 * 8); memset(w->dF, 0, n2 * 8);   /* d/dh  */
        memset(w->B, 0, n2 * 8); memset(w->F, 0, n2 * 8);     /* d2/dh2 */
        double hk1 = 1.0;
        for (int k = 1; k <= p; ++k) {
            const double sg = (k & 1) ? -1.0 : 1.0;
            const double* Gk = w->Gp + (size_t)k * n2;
            for (size_t e = 0; e < n2; ++e) { w->dB[e] += sg * c[k] * k * hk1 * Gk[e]; w->dF[e] += c[k] * k * hk1 * Gk[e]; }
            if (k >= 2) {
                const double hk2 = hk1 / h;  /* h^{k-2} */
                const double cc = c[k] * k * (k - 1) * (k == 2 ? 1.0 : hk2);
                for (size_t e = 0; e < n2; ++e) { w->B[e] += sg * cc * Gk[e]; w->F[e] += cc * Gk[e]; }
            }
            hk1 *= h;
        }
        mtm(w->X1, w->dB, M, n, n, N);
        mtm(w->X2, w->dF, M, n, n, N);
        for (int e = 0; e < s; ++e) { Ho[o_hU + e] = w->X1[e]; Ho[o_Uh + e] = -w->X2[e]; }
        mm(w->X1, w->B, U1, n, n, N);
        mm(w->X2, w->F, U0, n, n, N);
        for (int e = 0; e < s; ++e) w->X1[e] -= w->X2[e];
        Ho[o_hh] = dot(M, w->X1, s);
        int r0 = s, o = o_d;
        for (int d = 0; d < P->n_deriv; ++d) {
            for (int i = 0; i < P->ddim[d]; ++i) Ho[o + i] = -mu[r0 + i];
            r0 += P->ddim[d];
            o += P->ddim[d];
        }
    }
}

/* ---- Hessian of mu^T delta for the exponential integrator, delta = U1 - exp(h G(a)) U0 (README.md:79) -----------
 * The reference solves `integrator=:exponential` problems with the Hessian left on
 * (unitary_smooth_pulse_problem.jl:224-240,242-266; `eval_hessian=false` is spelled out where it is wanted,
 * unitary_robustness_problem.jl:205,247).  delta is linear in U1: every U_{t+1} block vanishes.  With E = exp(hG),
 * L_j = L_exp(hG; h G_j), L2_ij = the second Frechet derivative in the directions h G_i, h G_j, M = reshape(mu):
 *     (U0, a_j) = -vec(L_j^T M)   (U0, h) = -vec((G E)^T M)   (a_i, a_j) = -<M, L2_ij U0>
 *     (a_j, h) = -<M, (G_j E + G L_j) U0>   (h, h) = -<M, G^2 E U0>   (dx_i, h) = -mu_i
 * Here by differentiating the scaled Taylor polynomial and the squarings term by term in FORWARD mode (one chain per
 * drive, one per drive pair) -- oracle/qc_oracle.py takes the 3n x 3n block-triangular exponential instead, the HIP
 * kernels a forward-over-reverse form: three routes to the same numbers. */
#define QCO_EXP_HDEG 18

static void interval_hess_exp(const qco_problem* P, const double* z0, const double* z1, const double* mu, double* Ho, double* W) {
    const int n = 2 * P->N, N = qco_nc(P), s = n * N, m = P->m, ft = P->off_dt >= 0;
    const size_t n2 = (size_t)n * n;
    const int np = m * (m + 1) / 2;
    const double h = ft ? z0[P->off_dt] : P->dt_fixed;
    const double* U0 = z0 + P->off_U;
    const double* a = z0 + P->off_a;
    const double* M = mu;
    (void)z1;
    /* workspace carve-up */
    double* G = W;               W += n2;
    double* Y = W;               W += n2;
    double* E = W;               W += n2;
    double* A0 = W;              W += n2;
    double* A1 = W;              W += n2;
    double* T1 = W;              W += n2;
    double* T2 = W;              W += n2;
    double* L = W;               W += (size_t)m * n2;     /* running sums / squared values of the first derivatives */
    double* D0 = W;              W += (size_t)m * n2;
    double* D1 = W;              W += (size_t)m * n2;
    double* L2 = W;              W += (size_t)np * n2;
    double* S0 = W;              W += (size_t)np * n2;
    double* S1 = W;              W += (size_t)np * n2;
    double* X1 = W;              W += n2;
    double* X2 = W;              W += n2;
    memcpy(G, P->G_drift, n2 * 8);
    for (int j = 0; j < m; ++j)
        for (size_t e = 0; e < n2; ++e) G[e] += a[j] * P->G_drives[(size_t)j * n2 + e];
    double nrm = 0.0;
    for (int j = 0; j < n; ++j) {
        double cs = 0.0;
        for (int i = 0; i < n; ++i) cs += fabs(h * G[(size_t)j * n + i]);
        if (cs > nrm) nrm = cs;
    }
    int sq = 0;
    if (nrm > 0.25) { sq = (int)ceil(log2(nrm / 0.25)); if (sq < 0) sq = 0; }
    const double sc = ldexp(1.0, -sq), hs = h * sc;    /* Y = hs G, directions hs G_j */
    for (size_t e = 0; e < n2; ++e) { Y[e] = hs * G[e]; E[e] = 0.0; A0[e] = 0.0; }
    for (int i = 0; i < n; ++i) { E[(size_t)i * n + i] = 1.0; A0[(size_t)i * n + i] = 1.0; }
    memset(L, 0, (size_t)m * n2 * 8); memset(D0, 0, (size_t)m * n2 * 8);
    memset(L2, 0, (size_t)np * n2 * 8); memset(S0, 0, (size_t)np * n2 * 8);
    double *Ap = A0, *An = A1, *Dp = D0, *Dn = D1, *Sp = S0, *Sn = S1;
    for (int k = 1; k <= QCO_EXP_HDEG; ++k) {
        const double inv = 1.0 / k;
        /* second-order terms first: they use the (k-1)-th first-order terms */
        for (int j = 0; j < m; ++j)
            for (int i = 0; i <= j; ++i) {
                const size_t pi = (size_t)(j * (j + 1) / 2 + i) * n2;
                mm(T1, Sp + pi, Y, n, n, n);
                mm(T2, Dp + (size_t)i * n2, P->G_drives + (size_t)j * n2, n, n, n);
                for (size_t e = 0; e < n2; ++e) T1[e] += hs * T2[e];
                mm(T2, Dp + (size_t)j * n2, P->G_drives + (size_t)i * n2, n, n, n);
                for (size_t e = 0; e < n2; ++e) { Sn[pi + e] = (T1[e] + hs * T2[e]) * inv; L2[pi + e] += Sn[pi + e]; }
            }
        for (int j = 0; j < m; ++j) {
            mm(T1, Dp + (size_t)j * n2, Y, n, n, n);
            mm(T2, Ap, P->G_drives + (size_t)j * n2, n, n, n);
            for (size_t e = 0; e < n2; ++e) { Dn[(size_t)j * n2 + e] = (T1[e] + hs * T2[e]) * inv; L[(size_t)j * n2 + e] += Dn[(size_t)j * n2 + e]; }
        }
        mm(T1, Ap, Y, n, n, n);
        for (size_t e = 0; e < n2; ++e) { An[e] = T1[e] * inv; E[e] += An[e]; }
        double* t;
        t = Ap; Ap = An; An = t;  t = Dp; Dp = Dn; Dn = t;  t = Sp; Sp = Sn; Sn = t;
    }
    for (int q = 0; q < sq; ++q) {
        /* L2_ij <- E L2_ij + L2_ij E + L_i L_j + L_j L_i;  L_j <- E L_j + L_j E;  E <- E E */
        for (int j = 0; j < m; ++j)
            for (int i = 0; i <= j; ++i) {
                const size_t pi = (size_t)(j * (j + 1) / 2 + i) * n2;
                mm(T1, E, L2 + pi, n, n, n);
                mm(T2, L2 + pi, E, n, n, n);
                for (size_t e = 0; e < n2; ++e) T1[e] += T2[e];
                mm(T2, L + (size_t)i * n2, L + (size_t)j * n2, n, n, n);
                for (size_t e = 0; e < n2; ++e) T1[e] += T2[e];
                mm(T2, L + (size_t)j * n2, L + (size_t)i * n2, n, n, n);
                for (size_t e = 0; e < n2; ++e) S0[pi + e] = T1[e] + T2[e];
            }
        memcpy(L2, S0, (size_t)np * n2 * 8);
        for (int j = 0; j < m; ++j) {
            mm(T1, E, L + (size_t)j * n2, n, n, n);
            mm(T2, L + (size_t)j * n2, E, n, n, n);
            for (size_t e = 0; e < n2; ++e) D0[(size_t)j * n2 + e] = T1[e] + T2[e];
        }
        memcpy(L, D0, (size_t)m * n2 * 8);
        mm(T1, E, E, n, n, n);
        memcpy(E, T1, n2 * 8);
    }
    /* value order of hess_structure_local (exponential: no U_{t+1} blocks): (U, a) | (U, h) | (a, a) | (a, h) | (h, h) | (dx, h) */
    const int o_Ua = 0, o_Uh = s * m, o_aa = o_Uh + (ft ? s : 0), o_ah = o_aa + np, o_hh = o_ah + (ft ? m : 0), o_d = o_hh + (ft ? 1 : 0);
    mm(X1, E, U0, n, n, N);                 /* E U0 */
    for (int j = 0; j < m; ++j) {
        mtm(X2, L + (size_t)j * n2, M, n, n, N);
        for (int e = 0; e < s; ++e) Ho[o_Ua + (size_t)j * s + e] = -X2[e];
        for (int i = 0; i <= j; ++i) {
            mm(X2, L2 + (size_t)(j * (j + 1) / 2 + i) * n2, U0, n, n, N);
            Ho[o_aa + j * (j + 1) / 2 + i] = -dot(M, X2, s);
        }
        if (ft) {   /* (a_j, h) = -<M, G_j E U0 + G L_j U0> */
            mm(X2, P->G_drives + (size_t)j * n2, X1, n, n, N);
            double v = dot(M, X2, s);
            mm(T1, L + (size_t)j * n2, U0, n, n, N);
            mm(X2, G, T1, n, n, N);
            Ho[o_ah + j] = -(v + dot(M, X2, s));
        }
    }
    if (ft) {
        mm(T1, G, E, n, n, n);              /* G E */
        mtm(X2, T1, M, n, n, N);
        for (int e = 0; e < s; ++e) Ho[o_Uh + e] = -X2[e];
        mm(X2, G, X1, n, n, N);             /* G E U0 */
        mm(T1, G, X2, n, n, N);             /* G^2 E U0 */
        Ho[o_hh] = -dot(M, T1, s);
        int r0 = s, o = o_d;
        for (int d = 0; d < P->n_deriv; ++d) {
            for (int i = 0; i < P->ddim[d]; ++i) Ho[o + i] = -mu[r0 + i];
            r0 += P->ddim[d];
            o += P->ddim[d];
        }
    }
}

static int qco_eval_hess_exp(const qco_problem* P, const double* Z, const double* mu, double* H, long long t_begin, long long t_end) {
    const int n = 2 * P->N, m = P->m, ddim = qco_ddim(P), nnz = qco_hess_nnz(P);
    const size_t n2 = (size_t)n * n, per = (9 + 3 * (size_t)m + 3 * (size_t)(m * (m + 1) / 2)) * n2;
    int ok = 1;
#pragma omp parallel
    {
        double* W = malloc(per * 8);
        if (!W) {
#pragma omp atomic write
            ok = 0;
        } else {
#pragma omp for schedule(static)
            for (long long t = t_begin; t < t_end; ++t) {
                const double* z0 = Z + (size_t)t * P->zdim;
                interval_hess_exp(P, z0, z0 + P->zdim, mu + (size_t)t * ddim, H + (size_t)(t - t_begin) * nnz, W);
            }
        }
        free(W);
    }
    return ok ? 0 : -2;
}

int qco_eval_hess(const qco_problem* P, const double* Z, const double* mu, double* H, long long t_begin, long long t_end,
                  int nthreads) {
#ifdef _OPENMP
    if (P->integrator != 0 && nthreads > 0) omp_set_num_threads(nthreads);
#endif
    if (P->integrator != 0) return qco_eval_hess_exp(P, Z, mu, H, t_begin, t_end);
    const int n = 2 * P->N, p = P->order / 2, ddim = qco_ddim(P), nnz = qco_hess_nnz(P);
    int ok = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        qco_ws w;
        if (!ws_alloc(&w, n, p)) {
#pragma omp atomic write
            ok = 0;
        } else {
#pragma omp for schedule(static)
            for (long long t = t_begin; t < t_end; ++t) {
                const double* z0 = Z + (size_t)t * P->zdim;
                interval_hess(P, &w, z0, z0 + P->zdim, mu + (size_t)t * ddim, H + (size_t)(t - t_begin) * nnz);
            }
        }
        ws_free(&w);
    }
    return ok ? 0 : -2;
}

int qco_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
