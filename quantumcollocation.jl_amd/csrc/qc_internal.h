// Internal declarations shared by the host code and the HIP kernels of libqcolloc_hip.so.
// Not part of the ABI (that is include/qcolloc.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "qcolloc.h"

#define QC_MAX_P 10  // Pade order up to 20

// Per-interval value-block layout of the Jacobian (canonical order, DESIGN.md "COO order"):
//   [ -F copies (N * n^2) | B copies (N * n^2)  or  identity (s) | d/da (s*m) | d/dh (s) | derivative integrators ]
// and of the Hessian:
//   [ (U_t,a) s*m | (a,U_t+1) s*m | (a,a) upper m(m+1)/2 | (a,h) m | (U_t,h) s | (h,U_t+1) s | (h,h) 1 | (dx,h) ... ]
struct QcParams {
    int N, n, s, m, zdim, ddim;
    int off_U, off_a, off_dt;
    double dt_fixed;
    int integrator;
    int p;                   // Pade degree = order/2
    double c[QC_MAX_P + 1];  // Pade coefficients c_0..c_p
    int n_deriv;
    int dx_off[QC_MAX_DERIV], x_off[QC_MAX_DERIV], ddim_i[QC_MAX_DERIV];
    long long t_begin;       // first interval of this handle
    int n_int;               // number of intervals of this handle
    int jac_nnz, hess_nnz;   // per interval
    int jo_F, jo_B, jo_a, jo_h, jo_d;                           // Jacobian sub-block offsets (doubles)
    int ho_Ua, ho_aU, ho_aa, ho_ah, ho_Uh, ho_hU, ho_hh, ho_d;  // Hessian sub-block offsets
    int jchunk;              // LDS kernel: drives processed per phase
    int store_mode;          // 0 plain, 1 write-through (sc1), 2 non-temporal; see qc_st8
    int dbg_skip;            // diagnostic ablation (QC_DEBUG_SKIP): bit0 skip copy wave, bit1 skip compute wave
    const double* G;         // device: (m+1) matrices n*n, column-major; index 0 = drift
    const double* Gx;        // device: kernel-specific re-laid-out copy of G (MFMA path), or nullptr
    unsigned long long* stamps;  // diagnostic: 16 s_memrealtime slots per interval, or nullptr (normal)
};

struct qc_handle {
    qc_desc desc;  // G pointers nulled after create
    QcParams prm;
    qc_dims_t dims;
    int device = 0;
    int kernel = QC_KERNEL_LDS;
    size_t lds_bytes_jac = 0, lds_bytes_hess = 0;
    double* dG = nullptr;
    double* dGx = nullptr;
    // staging for the host-pointer entry points
    double *dZ = nullptr, *dF = nullptr, *dJ = nullptr, *dMu = nullptr, *dH = nullptr;
    unsigned long long* dStamps = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
};

// Fills prm/dims from a descriptor (host only). Returns QC_OK or error with message in `err`.
int qc_build_params(const qc_desc* d, QcParams* prm, qc_dims_t* dims, std::string* err);
void qc_local_jac_structure(const QcParams& P, std::vector<int32_t>* rows, std::vector<int32_t>* cols);
void qc_local_hess_structure(const QcParams& P, std::vector<int32_t>* rows, std::vector<int32_t>* cols);

// Kernel launchers (defined in the .hip files). All asynchronous on `stream`.
size_t qc_lds_bytes_jac(const QcParams& P);
size_t qc_lds_bytes_hess(const QcParams& P);
hipError_t qc_launch_lds_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, size_t lds, hipStream_t st);
hipError_t qc_launch_lds_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, size_t lds,
                              hipStream_t st);
bool qc_mfma_supported(const QcParams& P);
bool qc_mfma_hess_supported(const QcParams& P);
size_t qc_mfma_gx_doubles(const QcParams& P);
void qc_mfma_pack_G(const QcParams& P, const double* G_host, double* Gx_host);
hipError_t qc_launch_mfma_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
size_t qc_mfma32_gx_doubles(const QcParams& P);
void qc_mfma32_pack_G(const QcParams& P, const double* G_host, double* Gx_host);
hipError_t qc_launch_mfma32_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
hipError_t qc_launch_mfma_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);

// Diagnostic time stamp (only when the handle was created with QC_STAMPS=1; never in a timed run):
// slot k of interval b <- s_memrealtime (100 MHz).
#define QC_STAMP(P, b, lane, k)                                                                  \
    do {                                                                                         \
        if ((P).stamps != nullptr) {                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                     \
            if ((lane) == 0) (P).stamps[(size_t)(b) * 16 + (k)] = t_;                            \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    } while (0)

// Streaming store of one output value.  The outputs are written once and never re-read by the kernel;
// mode 1 (sc1, agent-scope relaxed atomic store) writes through the XCD's L2 so the bytes leave for
// HBM while the wave keeps computing instead of being flushed at the end of the kernel.
__device__ inline void qc_st8(double* p, double v, int mode) {
    if (mode == 1) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 2) {
        __builtin_nontemporal_store(v, p);
    } else {
        *p = v;
    }
}

// XCD-aware block -> local interval map: blocks b and b+8 share an XCD (round-robin dispatch,
// MI355X_MICROARCH.md "Workgroup dispatch"), so each XCD gets a contiguous run of intervals and the
// shared knot z_{t+1} of neighbouring intervals is served by one L2.  Bijective for any nb.
__host__ __device__ inline int qc_xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}
