/*
 * qcolloc.h — C ABI of libqcolloc_hip.so: MI355X (gfx950) knot-point evaluator for the
 * direct-collocation unitary dynamics constraint of QuantumCollocation.jl.
 *
 * This header IS the drop-in boundary.  Every entry point replaces one observable piece of the
 * `QuantumDynamics` object that QuantumCollocationCore builds inside
 * `QuantumControlProblem(traj, J, integrators; ...)`
 * (reference src/problem_templates/unitary_smooth_pulse_problem.jl:181-190) and that the reference's
 * own micro-benchmark exercises (reference test/scripts/integrator_test_1qubit.jl:41-52):
 *
 *   reference (Julia)                                   this ABI
 *   -------------------------------------------------   -----------------------------------------
 *   QuantumSystem(H_drift, H_drives).G_drift/G_drives    qc_generator_from_hamiltonian
 *       (unitary_smooth_pulse_problem.jl:199)
 *   operator_to_iso_vec / iso_vec_to_operator            qc_operator_to_iso_vec / qc_iso_vec_to_operator
 *       (trajectory_initialization.jl:40-41,137,413-418)
 *   UnitaryPadeIntegrator(state, control, sys, traj;     qc_desc{integrator=QC_PADE, pade_order,...}
 *       order)  (unitary_smooth_pulse_problem.jl:165-167)
 *   UnitaryExponentialIntegrator(...)  (:168-170)        qc_desc{integrator=QC_EXPONENTIAL}
 *   QuantumStatePadeIntegrator(state, control, sys, traj)  qc_desc{state_cols = number of kets}
 *       (quantum_state_smooth_pulse_problem.jl:146-152)
 *   DerivativeIntegrator(x, dx, traj)  (:177-178)        qc_desc.deriv_{x_off,dx_off,dim}[i]
 *   QuantumDynamics(integrators, traj)                   qc_create
 *       (integrator_test_1qubit.jl:41)
 *   dynamics.F(Z.datavec)               (:45)            qc_eval_F        / qc_eval_F_dev
 *   dynamics.dF(Z.datavec)              (:46)            qc_eval_jac      / qc_eval_F_jac(_dev)
 *   dynamics.dF_structure               (:46)            qc_jac_structure
 *   dynamics.mu_d2F(Z.datavec, mu)      (:52)            qc_eval_hess     / qc_eval_hess_dev
 *   dynamics.mu_d2F_structure           (:52)            qc_hess_structure
 *   shapes (Z.dims.states*(Z.T-1), Z.dim*Z.T+Z.global_dim)  (:44,48)   qc_dims
 *   K unitary integrators of a UnitarySamplingProblem    qc_eval_F_list / _jac_list / _F_jac_list / _hess_list (host buffers),
 *       (unitary_sampling_problem.jl:134-155), the           qc_eval_F_jac_dev_multi / qc_eval_hess_dev_multi (device buffers)
 *       members of a UnitaryDirectSumProblem                  (one handle per state integrator, qc_desc.rows_per_interval ...)
 *       (unitary_direct_sum_problem.jl:127-130), the kets
 *       x systems of a QuantumStateSamplingProblem
 *   UnitaryBangBangProblem's [U, D(a, da)], order 12      one handle: n_deriv = 1, pade_order = 12; the slack components only
 *       (unitary_bang_bang_problem.jl:163-175,205-215)        widen zdim
 *   DensityOperatorExponentialIntegrator                 qc_desc{N = levels^2, state_cols = 1, QC_EXPONENTIAL}
 *       (density_operator_smooth_pulse_problem.jl:104-106)   with the Lindblad generators as G_drift / G_drives
 *   iso_vec_unitary_fidelity, UnitaryInfidelityObjective,  qc_fidelity_create / qc_fidelity_eval(_dev)
 *   FinalUnitaryFidelityConstraint                           (unitary_minimum_time_problem.jl:77-84)
 *   iso_fidelity, QuantumStateObjective,                 qc_fidelity_create_kind(QC_FID_KET | QC_FID_DENSITY)
 *   DensityOperatorPureStateInfidelityObjective
 *   QuadraticRegularizer, MinimumTimeObjective           qc_terms_create / qc_terms_eval(_dev)
 *       (unitary_smooth_pulse_problem.jl:151-153, unitary_minimum_time_problem.jl:67-69)
 *   unitary_rollout / rollout / open_rollout             qc_rollout / qc_rollout_dev
 *       (trajectory_initialization.jl:426,493,547)
 *
 * Conventions
 *   - All matrices are column-major (Julia order).  All arrays are caller-owned; nothing is retained
 *     after a call returns, except that G_drift/G_drives are copied to the device by qc_create.
 *   - Return value 0 = QC_OK, negative = error; text via qc_last_error (thread-local for
 *     handle-less calls).  Nothing throws or exits across this boundary.
 *   - A handle from qc_create is bound to ONE HIP device and evaluates the interval range [t_begin, t_end) of the
 *     T-1 intervals; a handle from qc_create_multi spreads that range over several devices inside this
 *     library (one process, N GPUs: what a Julia/Ipopt consumer needs).  Every entry
 *     point selects the handle's device itself and restores the caller's current device before returning.
 *     A handle is not thread-safe and may have ONE evaluation in flight at a time: several kernels keep
 *     scratch in the handle (lds-gws, mfma64 Hessian, rollout, batched launches), so a second "_dev"
 *     call on another stream must be ordered after the first by the caller (different handles are
 *     independent).  Non-finite inputs are evaluated, not rejected (Ipopt probes wild points).
 *   - "_dev" entry points take DEVICE pointers and a hipStream_t (passed as void*) and are
 *     asynchronous on that stream.  The plain entry points take HOST pointers and return after the
 *     results are in the caller's buffers.
 *   - Every wait of a host-buffer call on the device has a deadline (environment QC_HOST_TIMEOUT_MS, default 30 s).  A call that runs
 *     into it returns QC_ERR_HIP WITHOUT synchronising (a hung device would hang that, too): work of that call may still be queued and
 *     may still write into the output buffers it was given and into memory of qc_host_alloc.  Those buffers must stay allocated and
 *     are not to be read until the handle's next call has returned (it drains the handle's streams before it reuses anything) or the
 *     handle has been destroyed (qc_destroy drains them too).
 *   - There is no CPU fallback: without a visible gfx950 device qc_create fails with QC_ERR_NO_DEVICE.
 */
#ifndef QCOLLOC_H
#define QCOLLOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QC_VERSION_MAJOR 0
#define QC_VERSION_MINOR 6

enum {
    QC_OK = 0,
    QC_ERR_INVALID = -1,    /* bad descriptor / argument */
    QC_ERR_NO_DEVICE = -2,  /* no HIP device, or not gfx950 */
    QC_ERR_HIP = -3,        /* a HIP runtime call failed */
    QC_ERR_UNSUPPORTED = -4 /* valid request this build cannot serve (e.g. qc_multi_all_gather_dev with two shards on one device) */
};

enum { QC_PADE = 0, QC_EXPONENTIAL = 1 };

/* Kernel selection (qc_desc.kernel). AUTO picks the MFMA path for systems with N <= 16 levels (sizes other than 8 / 16
 * levels zero-padded to the tiles) with the Pade order 4 or the exponential integrator, for up to 8 levels with any Pade
 * order, and for 17 .. 32 levels (5 qubits) with the Pade order 4; else the generic LDS/VALU path (its scratch in a global
 * workspace when it exceeds the LDS).  The Hessian of a 17 .. 32-level MFMA handle allocates 128 MiB of device scratch at
 * its first evaluation.  Forcing a path that cannot serve the descriptor is QC_ERR_UNSUPPORTED. */
enum { QC_KERNEL_AUTO = 0, QC_KERNEL_LDS = 1, QC_KERNEL_MFMA = 2 };

#define QC_MAX_DERIV 8
#define QC_HESS_ALIGN_LINE 16   /* qc_desc.hess_align of the line-aligned (padded) Hessian value layout */
enum { QC_ROWS_STACKED = 0, QC_ROWS_BY_COMPONENT = 1 };

typedef struct qc_handle qc_handle;

/* Value blocks of an interval (qc_desc.jac_block_order / hess_block_order; the enumerators' order is the default order).
 * Jacobian: d/dU_t = -I (x) F (N dense n x n blocks) | d/dU_t+1 = I (x) B (exponential integrator: the identity, s entries) | d/da
 * (s x m, column-major) | d/ddt (s; free timestep) | the derivative integrators' entries.
 * Hessian (upper triangle): (U_t, a) | (a, U_t+1) | (U_t, dt) | (dt, U_t+1) | (a, a) upper | (a, dt) | (dt, dt) | (dx, dt);
 * blocks that do not exist for a handle (fixed timestep; exponential integrator: nothing touches knot t+1) have length 0. */
enum { QC_JB_F = 0, QC_JB_B = 1, QC_JB_A = 2, QC_JB_H = 3, QC_JB_D = 4, QC_JAC_BLOCKS = 5 };
enum { QC_HB_UA = 0, QC_HB_AU = 1, QC_HB_UH = 2, QC_HB_HU = 3, QC_HB_AA = 4, QC_HB_AH = 5, QC_HB_HH = 6, QC_HB_D = 7, QC_HESS_BLOCKS = 8 };

/* Problem descriptor.  Offsets are 0-based positions inside one knot vector z_t of length zdim
 * (reference component order [U~, a, da, dda, dt]: trajectory_initialization.jl:357-382; other
 * templates order differently, e.g. integrator_test_1qubit.jl:24-30, hence parameters). */
typedef struct qc_desc {
    int32_t N;           /* Hilbert-space dimension; iso dimension n = 2N; state length s = 2N^2 */
    int32_t m;           /* number of drives (system.n_drives) */
    int64_t T;           /* number of knot points (Z.T) */
    int32_t zdim;        /* knot dimension (Z.dim) */
    int64_t global_dim;  /* trailing global variables after zdim*T (Z.global_dim); columns only */
    int32_t off_U;       /* offset of U~ (length s) */
    int32_t off_a;       /* offset of the control a (length m) */
    int32_t off_dt;      /* offset of the timestep, or -1 for a fixed timestep */
    double dt_fixed;     /* timestep when off_dt < 0 */
    int32_t integrator;  /* QC_PADE | QC_EXPONENTIAL */
    int32_t pade_order;  /* even, 2..20 (reference uses 4 by default, 12 in unitary_bang_bang_problem.jl:208) */
    int32_t n_deriv;     /* number of DerivativeIntegrators, rows appended after the unitary rows */
    int32_t deriv_x_off[QC_MAX_DERIV];
    int32_t deriv_dx_off[QC_MAX_DERIV];
    int32_t deriv_dim[QC_MAX_DERIV];
    const double* G_drift;  /* n*n, column-major:   iso(-i H_drift) */
    const double* G_drives; /* m matrices of n*n, column-major each */
    int32_t device;         /* HIP device ordinal this handle is bound to */
    int32_t kernel;         /* QC_KERNEL_* */
    int64_t t_begin;        /* first interval (0-based) this handle evaluates */
    int64_t t_end;          /* one past the last interval; t_begin = t_end = 0 means [0, T-1) */
    int32_t state_cols;     /* columns of the iso state matrix: 0 or N = a unitary U~ (2N x N, length 2N^2);
                             * K >= 1 = K kets psi~ = [Re psi; Im psi] stored back to back (2N x K, length 2NK), i.e.
                             * K QuantumStatePadeIntegrators over the same system
                             * (quantum_state_smooth_pulse_problem.jl:146-152).  off_U is the first ket's offset. */
    int32_t hess_align;     /* 0 (default) or 1: the Hessian value block of an interval holds EXACTLY the structural entries -- the
                             * reference's mu_d2F_structure, entry for entry (config 3: 1 832 per interval; observable as
                             * length(dynamics.mu_d2F_structure), test/scripts/integrator_test_1qubit.jl:48-52).  This is what every
                             * host-buffer entry point and both bindings use unless told otherwise (ABI 0.5; through ABI 0.4 the value
                             * 0 meant 16).
                             * k > 1 (QC_HESS_ALIGN_LINE = 16 doubles = 128 B for device-resident consumers): the block is padded with
                             * explicit zeros to a multiple of k doubles, so that every interval's block starts on a cache-line
                             * boundary (partial lines written by two workgroups cost 1.1 - 1.4x the HBM write traffic).
                             * The padding entries appear in the structure as duplicates of the interval's first structure entry with
                             * value 0 (COO duplicates are summed: reference test/test_utils.jl:14-20).  An opt-in of "_dev" consumers
                             * that own their sparse assembly; a host-buffer call is bound by PCIe and gains nothing from it.
                             * Ignored for composed handles (hess_per_interval > 0): see hess_tail_zeros. */
    /* Composition (all 0 = this handle is the whole dynamics).  A problem whose integrator list holds several
     * unitary integrators (UnitarySamplingProblem: one per system over a merged trajectory,
     * unitary_sampling_problem.jl:134-155) is served by one handle per unitary integrator; each handle's rows and
     * values are placed inside the problem's per-interval blocks.  Composed handles are evaluated through the "_dev" / "_dev_multi"
     * entry points (device buffers) or the "_list" entry points (host buffers), never through qc_eval_F / _jac / _hess. */
    int64_t rows_per_interval;  /* dynamics rows of the whole problem per interval (Z.dims.states) */
    int64_t row_offset;         /* first row of this handle inside that block */
    int64_t jac_per_interval;   /* Jacobian values of the whole problem per interval */
    int64_t jac_offset;
    int64_t hess_per_interval;  /* Hessian values of the whole problem per interval */
    int64_t hess_offset;
    /* Row placement.  QC_ROWS_STACKED (0): this handle's rows are [state integrator (s) | derivative integrators in
     * order], starting at row_offset.  QC_ROWS_BY_COMPONENT (1): the state integrator's rows start at row_offset and
     * derivative integrator i's rows at deriv_row_off[i] inside the per-interval block of rows_per_interval rows (which
     * must then be given): every integrator's rows sit at its state component's position among the trajectory's state
     * components, and components WITHOUT an integrator leave structurally empty rows, so that
     * n_rows == Z.dims.states * (T - 1) as the reference's own harness declares
     * (test/scripts/integrator_test_script.jl:23-44, integrator_test_1qubit.jl:24-44).  Empty rows are never written by
     * the "_dev" entry points (zero them once); the host-buffer entry points deliver them as 0. */
    int32_t row_placement;
    int32_t hess_tail_zeros;    /* composed handles only: explicit zero entries this handle appends after its own Hessian
                                 * values (the composer pads the shared per-interval block through its last handle) */
    int32_t deriv_row_off[QC_MAX_DERIV];
    /* Order of the value blocks INSIDE an interval (ABI 0.6).  The values of an interval are a sequence of blocks; which block comes
     * where is this library's choice, and the reference's own intra-knot COO order is not known here ("keep a
     * permutation hook").  Once `julia/reconcile.jl` has printed Core's order, a binding that wants the value vectors in THAT order --
     * without a 40 MB gather on the host per dF call -- names it here: a permutation of QC_JB_* / QC_HB_* (all zeros = the default
     * order, the enumerators' order).  Structures, block offsets and every kernel follow it; the order of the entries inside a block
     * (column-major) and of the derivative integrators' pieces stays.  Costs: the host-buffer Jacobian path sends one copy of the
     * replicated blocks over PCIe only in the default order (another order: the full values), and the one-call launch stores the
     * scalar Hessian entries one by one unless the four scalar kinds close the block in the default order. */
    int32_t jac_block_order[QC_JAC_BLOCKS];
    int32_t hess_block_order[QC_HESS_BLOCKS];
} qc_desc;

typedef struct qc_dims_t {
    int64_t n_rows;          /* ddim*(T-1): rows of dF, length of F and mu (whole problem) */
    int64_t n_cols;          /* zdim*T + global_dim */
    int64_t ddim;            /* dynamics rows per interval */
    int64_t jac_nnz_interval;
    int64_t hess_nnz_interval;
    int64_t n_intervals;     /* t_end - t_begin, intervals this handle owns */
    int64_t F_len;           /* rows per interval of the F vector this handle writes into (ddim, or rows_per_interval when
                              * given) * n_intervals */
    int64_t jac_nnz;         /* jac_nnz_interval * n_intervals */
    int64_t hess_nnz;        /* hess_nnz_interval * n_intervals (0 if no analytic Hessian); hess_nnz_interval includes the
                              * alignment padding (qc_desc.hess_align / hess_tail_zeros) */
    int64_t Z_len;           /* zdim*T + global_dim: length of the Z argument (always the full vector) */
    int32_t kernel;          /* QC_KERNEL_LDS or QC_KERNEL_MFMA actually selected */
    int32_t reserved;
} qc_dims_t;

/* ---- host-side helpers (no GPU needed) ------------------------------------------------------ */

/* U (N x N complex, column-major, split re/im planes) -> iso-vec of length 2N^2:
 * vec(vcat(real(U), imag(U)))  (reference trajectory_initialization.jl:137). */
int qc_operator_to_iso_vec(int32_t N, const double* U_re, const double* U_im, double* iso_vec);
int qc_iso_vec_to_operator(int32_t N, const double* iso_vec, double* U_re, double* U_im);

/* H (N x N complex Hermitian, column-major re/im planes) -> G = iso(-iH) = [[Im H, Re H],[-Re H, Im H]]
 * (2N x 2N, column-major), i.e. QuantumSystem's G_drift / G_drives entries. */
int qc_generator_from_hamiltonian(int32_t N, const double* H_re, const double* H_im, double* G);

/* Eigen-decomposition A = V diag(w) V' of a complex Hermitian d x d matrix (column-major, re / im planes), d <= 64: the
 * host-side set-up step of the free-phase fidelity (phase operators, qc_fidelity_desc), exposed so that it can be checked. */
int qc_hermitian_eig(int32_t d, const double* A_re, const double* A_im, double* w, double* V_re, double* V_im);

/* Pade coefficients c_0..c_p (p = order/2) of the diagonal approximant. out has p+1 entries. */
int qc_pade_coefficients(int32_t order, double* out);

/* Sizes implied by a descriptor, without touching a GPU (G pointers may be NULL). */
int qc_desc_dims(const qc_desc* d, qc_dims_t* out);
/* Sparsity structure implied by a descriptor, without touching a GPU.  Global COO indices of this
 * handle's slice, knot-major, in the exact order of the value vectors.  one_based != 0 gives Julia
 * indices.  rows/cols must hold jac_nnz (resp. hess_nnz) entries.  The Hessian structure is
 * upper-triangular (row <= col). */
int qc_desc_jac_structure(const qc_desc* d, int64_t* rows, int64_t* cols, int one_based);
int qc_desc_hess_structure(const qc_desc* d, int64_t* rows, int64_t* cols, int one_based);

/* ---- handle ----------------------------------------------------------------------------------- */

int qc_create(const qc_desc* d, qc_handle** out);
void qc_destroy(qc_handle* h);
/* Message of the last error on this handle (or, with h == NULL, of the last handle-less call on
 * this thread).  Never NULL. */
const char* qc_last_error(const qc_handle* h);

int qc_dims(const qc_handle* h, qc_dims_t* out);
/* Names of the device kernels this handle's evaluations run on (diagnostic; static strings, never NULL):
 * which = 0: F / F + dF ("mfma16-pade4", "mfma16-padeP", "mfma32-pade4-ell" (sparse drive generators) / "mfma32-pade4",
 * "mfma64-pade4", "mfma16-exp" / "mfma16-exp-gather" (drive generators with one entry per row), "mfma32-exp" / "mfma32-exp-gather", "lds", "lds-gws");
 * which = 1: mu_d2F ("mfma16-pade4-hess-gather" (drive generators with one entry per row) / "mfma16-pade4-hess2" /
 * "mfma16-pade4-hess", "mfma16-padeP-hess", "mfma32-pade4-hess-ell" / "mfma32-pade4-hess", "mfma64-pade4-hess", "lds-hess",
 * "lds-gws-hess"; exponential integrator: "mfma16-exp-hess", "mfma16-exp-hess-gather"
 * (drive generators with one entry per row: Pauli strings), "mfma32-exp-hess" / "mfma32-exp-hess-gather", "lds-exp-hess", "lds-gws-exp-hess");
 * which = 2: qc_eval_F_jac_hess_dev ("mfma16-pade4-fused-gather" / "mfma16-pade4-fused", "mfma32-pade4-fused-ell", or
 * "two-launches"). */
const char* qc_kernel_name(const qc_handle* h, int32_t which);
int qc_jac_structure(const qc_handle* h, int64_t* rows, int64_t* cols, int one_based);
int qc_hess_structure(const qc_handle* h, int64_t* rows, int64_t* cols, int one_based);

/* ---- evaluation, host buffers (what Ipopt's callbacks hand over) ------------------------------ */
/* Z: full trajectory vector (Z_len doubles).  F: F_len.  vals: jac_nnz.  mu: the FULL multiplier
 * vector of length n_rows (as MOI passes it); hvals: hess_nnz. */
int qc_eval_F(qc_handle* h, const double* Z, double* F);
int qc_eval_jac(qc_handle* h, const double* Z, double* vals);
int qc_eval_F_jac(qc_handle* h, const double* Z, double* F, double* vals);
int qc_eval_hess(qc_handle* h, const double* Z, const double* mu, double* hvals);

/* Ipopt's `new_x` flag (the C interface hands it to every callback; MOI evaluators track it themselves): new_x = 0 declares
 * that the trajectory vector of the following host-buffer calls is the one the handle's previous host-buffer call received --
 * the accepted trial point, at which Ipopt asks for the Jacobian and the Hessian after the residuals -- so the knots already
 * on the device are used and Z is not read at all.  Stays in force until qc_set_new_x(h, 1) (the default: every call copies
 * its Z).  A handle that has not seen a Z yet copies regardless.  Multi-device handles pass the flag on to their shards. */
int qc_set_new_x(qc_handle* h, int new_x);
/* How many times the handle has copied a trajectory vector's knots to the device so far (host-buffer calls with new_x = 1; -1 for a
 * NULL handle).  A binding that elides uploads remembers the value after the call that put ITS x on the device and sets new_x = 0
 * only while the value is unchanged: any other host-buffer call on the same handle in between (another closure, a second evaluator)
 * moves it.  qc_rollout keeps its trajectory vector in a buffer of its own and does not count. */
int64_t qc_knot_generation(const qc_handle* h);

/* Pinned host memory for long-lived arrays (optional).  The host-buffer calls move Z, mu, the residuals and the Hessian values between
 * the caller's arrays and the device with the GPU's copy engine; for ordinary (pageable) memory the runtime pins the pages for the
 * duration of every call.  Memory from qc_host_alloc is pinned for good (hipHostMalloc, visible to every device):
 *   - qc_eval_F writes the residuals into it straight from the kernel -- no device-to-host copy at all (config 3: 0.07 ms per call
 *     instead of 0.09, profiles/r04_f_path_probe.txt); layouts with rows no kernel writes (QC_ROWS_BY_COMPONENT) keep the copy;
 *   - every other transfer from / into it skips the per-call pinning.
 * The values are the same to the bit.  The bindings take the result vectors their closures hand out from here, an evaluator its
 * residual cache.  (Pinning arrays the CALLER allocated -- hipHostRegister -- was built first and withdrawn: under allocator churn in a
 * long-running process the GPU's writes into such ranges faulted intermittently, profiles/NOTES.md.)  Process-wide, thread-safe, not
 * tied to a handle; QC_ERR_NO_DEVICE without a GPU. */
int qc_host_alloc(int64_t bytes, void** out);
int qc_host_free(void* p);

/* ---- evaluation, device-resident (asynchronous on `stream`, a hipStream_t) --------------------- */
/* dZ: device pointer to the full Z vector (8-byte aligned).  dF may be NULL (skip residual store);
 * dvals may be NULL (residual only).  dF/dvals/dhvals point at THIS HANDLE'S slice
 * (interval t_begin first) and must be 8-byte aligned (the kernels issue 8-byte stores only). */
int qc_eval_F_jac_dev(qc_handle* h, const double* dZ, double* dF, double* dvals, void* stream);
int qc_eval_hess_dev(qc_handle* h, const double* dZ, const double* dmu, double* dhvals, void* stream);

/* dynamics.dF(Z) and dynamics.mu_d2F(Z, mu) (and F) at the same Z in one call -- what Ipopt asks for at every accepted point
 * (integrator_test_1qubit.jl:46,52).  One kernel launch where a fused kernel serves the handle (qc_kernel_name(h, 2):
 * "mfma16-pade4-fused" -- order-4 Pade, a unitary on 8 levels, Hermitian Hamiltonians, up to 6 drives: BASELINE configs 3 / 4), else
 * the two launches of qc_eval_F_jac_dev and qc_eval_hess_dev on `stream`; the values are bit-identical either way.  dF may be NULL. */
int qc_eval_F_jac_hess_dev(qc_handle* h, const double* dZ, const double* dmu, double* dF, double* dvals, double* dhvals, void* stream);

/* Several handles over the same trajectory in ONE launch: the K unitary integrators of a `UnitarySamplingProblem`
 * (reference unitary_sampling_problem.jl:134-155), each created with its slot of the shared per-interval blocks
 * (rows_per_interval / row_offset / ...).  dF / dvals / dhvals point at the SHARED vectors.  Handles whose shapes or
 * kernels differ are evaluated one launch each, with the same result. */
int qc_eval_F_jac_dev_multi(qc_handle* const* hs, int32_t count, const double* dZ, double* dF, double* dvals, void* stream);
int qc_eval_hess_dev_multi(qc_handle* const* hs, int32_t count, const double* dZ, const double* dmu, double* dhvals, void* stream);

/* The same integrator lists with HOST buffers -- what a CPU consumer (Ipopt through the MOI evaluator) of a
 * `UnitarySamplingProblem` (unitary_sampling_problem.jl:134-155: [U_1 .. U_K, D, D], shared controls), a
 * `UnitaryDirectSumProblem` (unitary_direct_sum_problem.jl:127-130: [U_1, D, D, U_2, D, D, ...], own controls per member) or
 * a `QuantumStateSamplingProblem` (quantum_state_sampling_problem.jl:98-122) calls as dynamics.F(Z), dynamics.dF(Z),
 * dynamics.mu_d2F(Z, mu).  `hs` = the composed handles of the list in integrator order, all on one device over one
 * trajectory and interval range.  Z is uploaded once, the handles are evaluated as in the "_dev_multi" calls, and F / vals /
 * hvals receive the whole problem's vectors: rows_per_interval, jac_per_interval, hess_per_interval values per interval,
 * interval-major.  The calls return when the arrays are complete.  qc_set_new_x(hs[0], 0) / qc_knot_generation(hs[0])
 * apply to the list (hs[0] owns the staging).  Errors are recorded on hs[0]. */
int qc_eval_F_list(qc_handle* const* hs, int32_t count, const double* Z, double* F);
int qc_eval_jac_list(qc_handle* const* hs, int32_t count, const double* Z, double* vals);
int qc_eval_F_jac_list(qc_handle* const* hs, int32_t count, const double* Z, double* F, double* vals);
int qc_eval_hess_list(qc_handle* const* hs, int32_t count, const double* Z, const double* mu, double* hvals);

/* ---- one handle over several GPUs ("n_gpus, device_ids[]"; 8e) ------------------------------------ */
/* The reference's consumer is ONE process (Julia + Ipopt, unitary_smooth_pulse_problem.jl:181-190, solve! at :219), so
 * the knot sharding of SURVEY 8(e) lives behind this boundary: qc_create_multi splits the descriptor's interval range
 * [t_begin, t_end) into n_shards contiguous chunks of ceil(len / n_shards) intervals (trailing shards may be short or
 * empty), shard i bound to HIP device device_ids[i] (ordinals may repeat: several shards on one device).  The result
 * is an ordinary qc_handle: qc_dims / qc_*_structure / qc_eval_F / qc_eval_jac / qc_eval_F_jac / qc_eval_hess /
 * qc_rollout behave exactly as on a single-device handle over the same range and return bit-identical arrays; every
 * shard has its own host thread, stream and pinned staging, copies only its knots (+ 1 halo knot) to its device and
 * writes its contiguous slice of the caller's arrays, so the N PCIe links run in parallel.  There is no collective on
 * this path (a CPU consumer needs none).  The "_dev" entry points take ONE device's pointers: use them on the shard
 * handles (qc_multi_shard), or qc_multi_eval_F_jac_dev / qc_multi_all_gather_dev below. */
int qc_create_multi(const qc_desc* d, int32_t n_shards, const int32_t* device_ids, qc_handle** out);
/* Number of shards of a multi-device handle (0 for a single-device handle). */
int32_t qc_multi_count(const qc_handle* h);
/* Borrowed pointer to shard i's single-device handle (owned by h; NULL if out of range).  Its interval range, device and
 * sizes are in its qc_dims; its device ordinal via qc_multi_shard_info. */
qc_handle* qc_multi_shard(qc_handle* h, int32_t i);
int qc_multi_shard_info(const qc_handle* h, int32_t i, int32_t* device, int64_t* t_begin, int64_t* t_end);
/* Device-resident evaluation of every shard, each on its own device and internal stream, asynchronous; dZ[i] / dF[i] /
 * dvals[i] are DEVICE pointers on shard i's device: dZ[i] the full Z vector, dF[i] / dvals[i] FULL-LENGTH vectors of
 * the multi handle's range (shard i writes its slice at its offset; a chunk-padded length, qc_multi_padded_len, is
 * needed for the all-gather).  dF or dvals may be NULL.  qc_multi_sync waits for all shards.
 * Ordering: the launches go out on each shard's INTERNAL stream, which knows nothing of the streams that produced dZ[i] / dmu[i]:
 * the inputs must be complete on their devices before these calls (synchronise the producing streams or devices first), and the
 * outputs may be read after qc_multi_sync. */
int qc_multi_eval_F_jac_dev(qc_handle* h, const double* const* dZ, double* const* dF, double* const* dvals);
int qc_multi_eval_hess_dev(qc_handle* h, const double* const* dZ, const double* const* dmu, double* const* dhvals);
int qc_multi_sync(qc_handle* h);
/* RCCL all-gather over xGMI of the per-shard slices (north_star / SURVEY 8e: "ncclAllGather on the per-GPU value
 * blocks when a device-resident full Jacobian is wanted"): in place on the full-length vectors, bufs[i] on shard i's
 * device holding that shard's slice at offset i * chunk * per_interval; afterwards every device holds all slices.
 * per_interval = doubles per interval (jac_nnz_interval, ddim or hess_nnz_interval); every buffer must hold
 * qc_multi_padded_len(h, per_interval) doubles.  Needs distinct devices (one RCCL rank per device; librccl is loaded
 * on first use, QC_ERR_UNSUPPORTED when it is absent or devices repeat).  Ordered after the shards' evaluations on their
 * internal streams; qc_multi_sync waits for it. */
int64_t qc_multi_padded_len(const qc_handle* h, int64_t per_interval);
int qc_multi_all_gather_dev(qc_handle* h, double* const* bufs, int64_t per_interval);

/* sizeof of the ABI structs as this build sees them: a binding asserts them against its own mirror at load time. */
int64_t qc_sizeof_desc(void);
int64_t qc_sizeof_dims(void);
int64_t qc_sizeof_terms_desc(void);

/* ---- fidelity of the final knot (SURVEY 8f "next" row 1) -------------------------------------------- */
/* F(U~) = |tr(U_goal' U)| / n over the subspace block (`iso_vec_unitary_fidelity(U_T, U_G, subspace=...)`,
 * reference unitary_minimum_time_problem.jl:77) and the loss l = |1 - F| of `UnitaryInfidelityObjective`
 * (docstring unitary_smooth_pulse_problem.jl:23-28); `FinalUnitaryFidelityConstraint` (:80-84) is F - F_min >= 0.
 * grad: dF/dU~ (length 2N^2); hess: d2F/dU~2, dense upper triangle, column-major (entry (i<=j) at j(j+1)/2 + i).
 * For the loss: grad l = -sign(1-F) grad F, hess l = -sign(1-F) hess F.  The normalisation (|tr|/n) follows the
 * docstring; PiccoloQuantumObjects 0.3's own definition could not be inspected. */
typedef struct qc_fidelity qc_fidelity;
int qc_fidelity_create(int32_t N, const double* goal_iso, const int32_t* subspace /* 0-based, or NULL */, int32_t n_sub,
                       int32_t device, qc_fidelity** out);
/* Ket and density-operator states (quantum_state_smooth_pulse_problem.jl:133 `QuantumStateObjective`,
 * quantum_state_minimum_time_problem.jl:50-62 `iso_fidelity` / `FinalQuantumStateFidelityConstraint`,
 * density_operator_smooth_pulse_problem.jl:55 `DensityOperatorPureStateInfidelityObjective`):
 *   QC_FID_KET      state psi~ (2N),      F = |<psi_goal|psi>|^2
 *   QC_FID_DENSITY  state rho~ (2N^2),    F = psi_goal' rho psi_goal   (linear)
 * goal_ket_iso = [Re psi_goal; Im psi_goal] in both cases; loss |1 - F|, gradients / Hessians as for the unitary kind. */
#define QC_FID_UNITARY 0
#define QC_FID_KET 1
#define QC_FID_DENSITY 2
int qc_fidelity_create_kind(int32_t kind, int32_t N, const double* goal_ket_iso, int32_t device, qc_fidelity** out);
/* Descriptor form: everything above plus the two choices that PiccoloQuantumObjects 0.3 (not vendored) makes and this
 * repository cannot inspect, and the free-phase fidelity of the minimum-time / smooth-pulse templates:
 *   form      QC_FID_FORM_ABS   F = |tr(U_goal' U)| / n      (the reference's docstring, unitary_smooth_pulse_problem.jl:23-28; default)
 *             QC_FID_FORM_ABS2  F = |tr(U_goal' U)|^2 / n^2
 *   n_phases  K > 0: `iso_vec_unitary_free_phase_fidelity(U, U_goal, phases, phase_operators; subspace)` and
 *             `FinalUnitaryFreePhaseFidelityConstraint` (unitary_minimum_time_problem.jl:86-100, phases = global
 *             variables `traj.global_data[phase_name]`, trajectory_initialization.jl:370-380):
 *                 F(U, phi) = |tr(U_goal' R(phi) U)| / n,   R(phi) = (x)_k exp(i phi_k Op_k)   (Julia's reduce(kron, ...)),
 *             Op_k Hermitian d_k x d_k with prod d_k = subspace size (Paulis Z for virtual-Z corrections,
 *             unitary_smooth_pulse_problem.jl:345-346).  The evaluation input is then [U~ (2N^2) ; phi (K)], gradient and
 *             Hessian cover all 2N^2 + K variables (qc_fidelity_input_len). */
#define QC_FID_FORM_ABS 0
#define QC_FID_FORM_ABS2 1
typedef struct qc_fidelity_desc {
    int32_t kind;               /* QC_FID_UNITARY | QC_FID_KET | QC_FID_DENSITY (subspace / form / phases: unitary only) */
    int32_t N;
    const double* goal_iso;     /* unitary: iso-vec of U_goal (2N^2); ket / density: [Re psi_goal; Im psi_goal] */
    const int32_t* subspace;    /* 0-based level indices, or NULL = all levels */
    int32_t n_sub;
    int32_t form;               /* QC_FID_FORM_* */
    int32_t n_phases;           /* K, 0..16 */
    int32_t device;
    const int32_t* phase_dims;  /* K dimensions d_k */
    const double* phase_ops;    /* operator k after operator k-1: d_k x d_k real plane, then d_k x d_k imaginary plane, column-major */
} qc_fidelity_desc;
int qc_fidelity_create_desc(const qc_fidelity_desc* d, qc_fidelity** out);
int32_t qc_fidelity_input_len(const qc_fidelity* h);   /* 2N^2 + K (state length for kets / density operators) */
void qc_fidelity_destroy(qc_fidelity* h);
const char* qc_fidelity_last_error(const qc_fidelity* h);
/* host buffers; any of fidelity / infidelity / grad / hess may be NULL */
int qc_fidelity_eval(qc_fidelity* h, const double* U_iso, double* fidelity, double* infidelity, double* grad, double* hess);
/* device buffers, asynchronous on `stream`: dval2 = {F, |1-F|}; dgrad / dhess may be NULL */
int qc_fidelity_eval_dev(qc_fidelity* h, const double* dU, double* dval2, double* dgrad, double* dhess, void* stream);

/* ---- rollouts (SURVEY 8f "next" row 4) -------------------------------------------------------------- */
/* x_{t+1} = exp(dt_t G(a_t)) x_t for t = 0 .. T-2 from x_0 = init (2N x cols, column-major), controls and timesteps read
 * from Z at the handle's offsets; out receives the (2N cols) x T state matrix, one column per knot: `unitary_rollout`,
 * `rollout`, `open_rollout` (reference call sites trajectory_initialization.jl:426,493,547; `unitary_rollout_fidelity`,
 * unitary_smooth_pulse_problem.jl:218, is this followed by qc_fidelity_eval on the last column).  The propagator is the
 * matrix exponential whatever integrator the handle was created with (the reference's default `expv`); the whole
 * trajectory is covered whatever the handle's shard.  2N <= 64. */
int qc_rollout(qc_handle* h, const double* Z, const double* init, double* out);
int qc_rollout_dev(qc_handle* h, const double* dZ, const double* dinit, double* dout, void* stream);

/* ---- trajectory cost terms (SURVEY 8f "next" row 3) ----------------------------------------------- */
/* J(Z) = sum_t 1/2 sum_k R_k (sc_t (v_tk - b_tk))^2 + D sum_{t < min_time_knots} dt_t,   sc_t = 1 (default) or dt_t:
 * the `QuadraticRegularizer(name, traj, R; baseline, timestep_name)` terms on a / da / dda (reference call sites
 * unitary_smooth_pulse_problem.jl:151-153) flattened into one list of regularised scalar entries of a knot, plus
 * `MinimumTimeObjective(traj; D)` (unitary_minimum_time_problem.jl:67-69; the last knot's timestep drives no
 * interval, hence min_time_knots = T-1 there).  QC_REG_DT_SCALED (2, the default of every binding) puts the timestep
 * inside the square: every problem template hands the regulariser the timestep's name
 * (`QuadraticRegularizer(name, traj, R; timestep_name=timestep_name)`, unitary_smooth_pulse_problem.jl:151-153,
 * unitary_sampling_problem.jl:116-118, quantum_state_sampling_problem.jl:82-84), which only a definition that reads dt_t
 * needs, and that is how QuantumCollocationCore 0.3 is recalled to define it (not vendored, SURVEY 8c;
 * julia/reconcile.jl settles it).  QC_REG_PLAIN (3) is the docstring's "1/2 sum_t R_a a_t^2 + ..."
 * (unitary_smooth_pulse_problem.jl:13).  The values 0 and 1 are RETIRED (ABI 0.1 - 0.3 gave them both meanings in turn): a
 * descriptor that carries one of them is refused with QC_ERR_INVALID, so a binding built against an older header fails loudly
 * instead of computing the other regulariser; bindings also compare qc_abi_version() at load time.
 * Gradient: dense, Z_len entries (zeros included).  Hessian: upper triangle, per knot
 * [ (v_k,v_k) k=0..n_reg-1 | (v_k,dt) k=0..n_reg-1 | (dt,dt) ]; the last two groups exist only for QC_REG_DT_SCALED
 * with a free timestep. */
#define QC_REG_DT_SCALED 2
#define QC_REG_PLAIN 3
typedef struct qc_terms_desc {
    int64_t T;
    int32_t zdim;
    int32_t off_dt;              /* offset of the timestep inside a knot, -1: fixed */
    int64_t global_dim;
    double dt_fixed;
    int32_t n_reg;               /* number of regularised scalar entries of a knot */
    int32_t weighting;           /* QC_REG_DT_SCALED | QC_REG_PLAIN */
    const int32_t* reg_index;    /* n_reg offsets inside a knot, strictly increasing, != off_dt */
    const double* reg_R;         /* n_reg weights */
    const double* reg_baseline;  /* NULL, or n_reg x T (entry-fastest) values subtracted before squaring */
    double min_time_D;           /* 0: no minimum-time term */
    int64_t min_time_knots;      /* timesteps of knots 0..min_time_knots-1 are summed */
    int32_t device;
    int32_t reserved0;
} qc_terms_desc;
typedef struct qc_terms qc_terms;
int qc_terms_desc_hess_nnz(const qc_terms_desc* d, int64_t* nnz);
int qc_terms_desc_hess_structure(const qc_terms_desc* d, int64_t* rows, int64_t* cols, int one_based);
int qc_terms_create(const qc_terms_desc* d, qc_terms** out);
void qc_terms_destroy(qc_terms* h);
const char* qc_terms_last_error(const qc_terms* h);
int qc_terms_hess_nnz(const qc_terms* h, int64_t* nnz);
int qc_terms_hess_structure(const qc_terms* h, int64_t* rows, int64_t* cols, int one_based);
/* host buffers; J / grad / hvals may be NULL */
int qc_terms_eval(qc_terms* h, const double* Z, double* J, double* grad, double* hvals);
/* device buffers, asynchronous on `stream`; dgrad / dhvals may be NULL */
int qc_terms_eval_dev(qc_terms* h, const double* dZ, double* dJ, double* dgrad, double* dhvals, void* stream);

/* Diagnostic only: when the environment variable QC_STAMPS=1 is set at qc_create, the MFMA kernel
 * records 16 s_memrealtime (100 MHz) checkpoints per interval; this copies them out (synchronises the
 * device).  Not part of the evaluated path; a handle created without QC_STAMPS returns
 * QC_ERR_UNSUPPORTED. */
int qc_debug_read_stamps(qc_handle* h, uint64_t* out, int64_t count);

/* Diagnostic only: the rate (GB/s of values written) at which this host replicates the compact Jacobian form of handle h into a
 * full value array with the library's own worker team -- no GPU work, no transfer; the host-side bound of qc_eval_jac that
 * bench.py reports next to the PCIe bound.  QC_ERR_UNSUPPORTED when the handle's Jacobian has no replicated blocks. */
int qc_debug_host_expand_rate(qc_handle* h, int32_t reps, double* GBps);

/* Library/build identification: "qcolloc-hip <major>.<minor> (gfx950, ...)" */
const char* qc_version(void);
/* QC_VERSION_MAJOR * 1000 + QC_VERSION_MINOR of the build.  A binding compares it with the header version it mirrors when
 * it loads the library (struct sizes alone do not reveal a renumbered constant). */
int32_t qc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QCOLLOC_H */
