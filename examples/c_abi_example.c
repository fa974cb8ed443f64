/* Pure-C use of the drop-in boundary (include/qcolloc.h): a 1-qubit UnitaryPadeIntegrator + two DerivativeIntegrators,
 * T = 6 knots, free timestep -- the integrator list of unitary_smooth_pulse_problem.jl:175-179.  Builds the generators
 * from Hamiltonians, queries dims and structure, evaluates F, dF, mu_d2F on host buffers and prints checksums.
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
    return 0;
}
