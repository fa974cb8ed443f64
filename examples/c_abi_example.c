/* Pure-C use of the drop-in boundary (include/qcolloc.h): a 1-qubit UnitaryPadeIntegrator + two DerivativeIntegrators,
 * T = 6 knots, free timestep -- the integrator list of unitary_smooth_pulse_problem.jl:175-179.  Builds the generators
 * from Hamiltonians, queries dims and structure, evaluates F, dF, mu_d2F on host buffers and prints checksums; then the
 * integrator list of a two-system UnitarySamplingProblem through the "_list" entry points.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_example.c -o /tmp/c_abi_example \
 *       -Lquantumcollocation.jl_amd/csrc -lqcolloc_hip -Wl,-rpath,$PWD/quantumcollocation.jl_amd/csrc -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qcolloc.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != QC_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, qc_last_error(h)); return 1; } \
    } while (0)

int main(void) {
    qc_handle* h = NULL;
    enum { N = 2, M = 2, T = 6, S = 2 * N * N, ZDIM = S + 3 * M + 1 };
    /* Hamiltonians, column-major real / imaginary planes: drift 0.1 Z, drives X and Y */
    const double Zre[4] = {0.1, 0, 0, -0.1}, Zim[4] = {0, 0, 0, 0};
    const double Xre[4] = {0, 1, 1, 0}, Xim[4] = {0, 0, 0, 0};
    const double Yre[4] = {0, 0, 0, 0}, Yim[4] = {0, 1, -1, 0};   /* Y = [[0,-i],[i,0]], column-major */
    double G0[16], Gd[2 * 16];
    CHECK(qc_generator_from_hamiltonian(N, Zre, Zim, G0));
    CHECK(qc_generator_from_hamiltonian(N, Xre, Xim, Gd));
    CHECK(qc_generator_from_hamiltonian(N, Yre, Yim, Gd + 16));

    qc_desc d;
    memset(&d, 0, sizeof d);
    d.N = N; d.m = M; d.T = T; d.zdim = ZDIM; d.global_dim = 0;
    d.off_U = 0; d.off_a = S; d.off_dt = S + 3 * M; d.dt_fixed = 0.0;     /* knot = [U~ (8), a (2), da (2), dda (2), dt (1)] */
    d.integrator = QC_PADE; d.pade_order = 4;
    d.n_deriv = 2;
    d.deriv_x_off[0] = S;     d.deriv_dx_off[0] = S + M;     d.deriv_dim[0] = M;   /* DerivativeIntegrator(a, da)   */
    d.deriv_x_off[1] = S + M; d.deriv_dx_off[1] = S + 2 * M; d.deriv_dim[1] = M;   /* DerivativeIntegrator(da, dda) */
    d.G_drift = G0; d.G_drives = Gd;
    d.device = 0; d.kernel = QC_KERNEL_AUTO;

    qc_dims_t dims;
    CHECK(qc_desc_dims(&d, &dims));
    printf("rows %lld cols %lld jac_nnz %lld hess_nnz %lld\n", (long long)dims.n_rows, (long long)dims.n_cols, (long long)dims.jac_nnz,
           (long long)dims.hess_nnz);
    CHECK(qc_create(&d, &h));

    double* Z = calloc((size_t)dims.Z_len, sizeof(double));
    for (int t = 0; t < T; ++t) {
        double* z = Z + (size_t)t * ZDIM;
        const double th = 0.3 * t;                       /* U_t = exp(-i th X): [Re U; Im U] per column */
        z[0] = cos(th); z[1] = 0; z[2] = 0; z[3] = -sin(th);
        z[4] = 0; z[5] = cos(th); z[6] = -sin(th); z[7] = 0;
        for (int k = 0; k < 3 * M; ++k) z[S + k] = 0.1 * sin(1.0 + t + 0.7 * k);
        z[S + 3 * M] = 0.2;
    }
    double* F = malloc((size_t)dims.F_len * sizeof(double));
    double* J = malloc((size_t)dims.jac_nnz * sizeof(double));
    double* H = malloc((size_t)dims.hess_nnz * sizeof(double));
    double* mu = malloc((size_t)dims.n_rows * sizeof(double));
    int64_t* rows = malloc((size_t)dims.jac_nnz * sizeof(int64_t));
    int64_t* cols = malloc((size_t)dims.jac_nnz * sizeof(int64_t));
    for (long long i = 0; i < dims.n_rows; ++i) mu[i] = 1.0;
    CHECK(qc_eval_F_jac(h, Z, F, J));
    CHECK(qc_eval_hess(h, Z, mu, H));
    CHECK(qc_jac_structure(h, rows, cols, 1));
    double sF = 0, sJ = 0, sH = 0;
    for (long long i = 0; i < dims.F_len; ++i) sF += F[i] * (1 + i % 7);
    for (long long i = 0; i < dims.jac_nnz; ++i) sJ += J[i] * (1 + i % 11);
    for (long long i = 0; i < dims.hess_nnz; ++i) sH += H[i] * (1 + i % 13);
    printf("checksums F %.15e dF %.15e mu_d2F %.15e first entry (%lld,%lld) %s\n", sF, sJ, sH, (long long)rows[0], (long long)cols[0], qc_version());
    /* Ipopt's order of calls: residuals at a new x, then the Jacobian and the Hessian of the Lagrangian at the
       SAME x with new_x = false -- the knots already on the device are used, Z is not read or uploaded again. */
    {
        double* J2 = malloc((size_t)dims.jac_nnz * sizeof(double));
        double* H2 = malloc((size_t)dims.hess_nnz * sizeof(double));
        if (qc_abi_version() != QC_VERSION_MAJOR * 1000 + QC_VERSION_MINOR) { fprintf(stderr, "header / library mismatch\n"); return 1; }
        CHECK(qc_set_new_x(h, 1));
        CHECK(qc_eval_F(h, Z, F));
        {
            const int64_t gen = qc_knot_generation(h);    /* uploads so far: the elision below is valid while nobody else has uploaded */
            CHECK(qc_set_new_x(h, 0));
            CHECK(qc_eval_jac(h, Z, J2));                 /* (Z is not read) */
            CHECK(qc_eval_hess(h, Z, mu, H2));
            CHECK(qc_set_new_x(h, 1));
            if (qc_knot_generation(h) != gen) { fprintf(stderr, "an elided call uploaded\n"); return 1; }
        }
        printf("ipopt order: %s\n", memcmp(J, J2, (size_t)dims.jac_nnz * sizeof(double)) == 0 &&
                                     memcmp(H, H2, (size_t)dims.hess_nnz * sizeof(double)) == 0 ? "same values" : "DIFFERENT VALUES");
        free(J2); free(H2);
    }
    /* A long-lived result array in pinned memory of the library: the residual kernel writes it in place (no device-to-host copy). */
    {
        void* p = NULL;
        CHECK(qc_host_alloc((int64_t)dims.F_len * (int64_t)sizeof(double), &p));
        CHECK(qc_eval_F(h, Z, (double*)p));
        printf("pinned residuals: %s\n", memcmp(F, p, (size_t)dims.F_len * sizeof(double)) == 0 ? "same values" : "DIFFERENT VALUES");
        CHECK(qc_host_free(p));
    }
    qc_destroy(h);
    free(Z); free(F); free(J); free(H); free(mu); free(rows); free(cols);
    /* An integrator list with two state integrators -- a UnitarySamplingProblem over two systems (drift +0.1 Z and -0.1 Z) that share
       the controls of one merged trajectory, knot = [U~_1 (8), U~_2 (8), a, da, dda, dt]; list [U_1, U_2, D(a, da), D(da, dda)]
       (unitary_sampling_problem.jl:134-155).  One composed handle per unitary integrator, each with its slot of the problem's
       per-interval blocks; the host-buffer "_list" calls evaluate the whole list. */
    {
        enum { ZD2 = 2 * S + 3 * M + 1 };
        const double Wre[4] = {-0.1, 0, 0, 0.1};
        double G0b[16];
        qc_handle* hs[2] = {NULL, NULL};
        qc_desc dd[2];
        qc_dims_t own[2], total;
        int64_t rows_pi = 0, jac_pi = 0, hess_pi = 0, ro = 0, jo = 0, ho = 0;
        CHECK(qc_generator_from_hamiltonian(N, Wre, Zim, G0b));
        for (int k = 0; k < 2; ++k) {
            memset(&dd[k], 0, sizeof dd[k]);
            dd[k].N = N; dd[k].m = M; dd[k].T = T; dd[k].zdim = ZD2;
            dd[k].off_U = k * S; dd[k].off_a = 2 * S; dd[k].off_dt = 2 * S + 3 * M;
            dd[k].integrator = QC_PADE; dd[k].pade_order = 4;
            dd[k].G_drift = k ? G0b : G0; dd[k].G_drives = Gd;
            dd[k].hess_align = 1;                       /* the shared Hessian block is padded as a whole (hess_tail_zeros), not per handle */
            if (k == 1) {                               /* the derivative integrators follow the last unitary integrator */
                dd[k].n_deriv = 2;
                dd[k].deriv_x_off[0] = 2 * S;     dd[k].deriv_dx_off[0] = 2 * S + M;     dd[k].deriv_dim[0] = M;
                dd[k].deriv_x_off[1] = 2 * S + M; dd[k].deriv_dx_off[1] = 2 * S + 2 * M; dd[k].deriv_dim[1] = M;
            }
            CHECK(qc_desc_dims(&dd[k], &own[k]));
            rows_pi += own[k].ddim; jac_pi += own[k].jac_nnz_interval; hess_pi += own[k].hess_nnz_interval;
        }
        for (int k = 0; k < 2; ++k) {
            dd[k].rows_per_interval = rows_pi; dd[k].row_offset = ro;
            dd[k].jac_per_interval = jac_pi;   dd[k].jac_offset = jo;
            dd[k].hess_per_interval = hess_pi; dd[k].hess_offset = ho;
            CHECK(qc_create(&dd[k], &hs[k]));
            ro += own[k].ddim; jo += own[k].jac_nnz_interval; ho += own[k].hess_nnz_interval;
        }
        h = hs[0];                                      /* errors of the list calls are recorded on the first handle */
        CHECK(qc_dims(hs[0], &total));
        {
            const size_t n_int = (size_t)total.n_intervals, nF = n_int * (size_t)rows_pi, nJ = n_int * (size_t)jac_pi, nH = n_int * (size_t)hess_pi;
            double* Z2 = calloc((size_t)T * ZD2, sizeof(double));
            double* F2 = malloc(nF * sizeof(double));
            double* J2 = malloc(nJ * sizeof(double));
            double* H2 = malloc(nH * sizeof(double));
            double* mu2 = malloc(nF * sizeof(double));
            double sF2 = 0, sJ2 = 0, sH2 = 0;
            for (int t = 0; t < T; ++t) {
                double* z = Z2 + (size_t)t * ZD2;
                for (int k = 0; k < 2; ++k) {
                    const double th = (0.3 + 0.05 * k) * t;
                    double* u = z + k * S;
                    u[0] = cos(th); u[3] = -sin(th); u[5] = cos(th); u[6] = -sin(th);
                }
                for (int k = 0; k < 3 * M; ++k) z[2 * S + k] = 0.1 * sin(1.0 + t + 0.7 * k);
                z[2 * S + 3 * M] = 0.2;
            }
            for (size_t i = 0; i < nF; ++i) mu2[i] = 1.0;
            CHECK(qc_eval_F_jac_list(hs, 2, Z2, F2, J2));
            CHECK(qc_eval_hess_list(hs, 2, Z2, mu2, H2));
            for (size_t i = 0; i < nF; ++i) sF2 += F2[i] * (1 + i % 7);
            for (size_t i = 0; i < nJ; ++i) sJ2 += J2[i] * (1 + i % 11);
            for (size_t i = 0; i < nH; ++i) sH2 += H2[i] * (1 + i % 13);
            printf("list checksums F %.15e dF %.15e mu_d2F %.15e sizes %zu %zu %zu\n", sF2, sJ2, sH2, nF, nJ, nH);
            free(Z2); free(F2); free(J2); free(H2); free(mu2);
        }
        qc_destroy(hs[0]);
        qc_destroy(hs[1]);
    }
    return 0;
}
