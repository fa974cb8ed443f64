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
// and of the Hessian (qc_build_params: the matrix blocks first, then the scalar entries as one run):
//   [ (U_t,a) s*m | (a,U_t+1) s*m | (U_t,h) s | (h,U_t+1) s | (a,a) upper m(m+1)/2 | (a,h) m | (h,h) 1 | (dx,h) ... ]
// The exponential integrator is linear in U_t+1: its (a,U_t+1) and (h,U_t+1) blocks are structurally empty (length 0).
struct QcParams {
    int N, n, s, m, zdim, ddim;
    int nc;                  // columns of the iso state matrix (N for a unitary, K for K kets); s = n * nc
    int off_U, off_a, off_dt;
    double dt_fixed;
    int integrator;
    int p;                   // Pade degree = order/2
    double c[QC_MAX_P + 1];  // Pade coefficients c_0..c_p
    int n_deriv;
    int dx_off[QC_MAX_DERIV], x_off[QC_MAX_DERIV], ddim_i[QC_MAX_DERIV];
    int drow[QC_MAX_DERIV];  // first row of derivative integrator i relative to this handle's row block (stacked: s + sum of the
                             // earlier dims; QC_ROWS_BY_COMPONENT: deriv_row_off[i] - row_offset, possibly negative)
    long long t_begin;       // first interval of this handle
    int n_int;               // number of intervals of this handle
    int jac_nnz, hess_nnz;   // per interval (own values; hess_nnz excludes the padding)
    int antisym;             // every generator is exactly antisymmetric (Hermitian Hamiltonians): G_k^T = -G_k bit for bit, so the
                             // transposed (B-layout) generator images need not be loaded (Hessian kernels)
    int copies;              // copies of the -F / B blocks the MFMA order-4 kernels write (nc; 1 = compact form of the host path)
    int h_pad;               // explicit zeros this handle writes after its hess_nnz values (line alignment of the interval blocks)
    // placement of this handle's rows / values inside the problem's vectors (composed problems: several integrator
    // groups share one row block and one value block per interval; default = own sizes, offset 0)
    long long F_stride, F_off, J_stride, J_off, H_stride, H_off;
    int jo_F, jo_B, jo_a, jo_h, jo_d;                           // Jacobian sub-block offsets (doubles)
    int ho_Ua, ho_aU, ho_aa, ho_ah, ho_Uh, ho_hU, ho_hh, ho_d;  // Hessian sub-block offsets
    int scal_run;            // the Hessian block's four scalar kinds [(a,a) | (a,h) | (h,h) | (dx,h)] close the block in this order (the default;
                             // qc_desc.hess_block_order may say otherwise): the one-call kernel stages them in LDS and stores them in one piece
    int jchunk;              // LDS kernel: drives processed per phase
    int store_mode;          // 0 plain, 1 write-through (sc1), 2 non-temporal; see qc_st8
    int dbg_skip;            // diagnostic ablation (QC_DEBUG_SKIP): bit0 skip copy wave, bit1 skip compute wave
    const double* G;         // device: (m+1) matrices n*n, column-major; index 0 = drift
    const double* Gx;        // device: kernel-specific re-laid-out copy of G (MFMA path), or nullptr
    int use_ws;              // LDS kernels: per-interval scratch in the global workspace `ws` (system too large for LDS)
    double* ws;              // device: n_int * ws_stride doubles, or nullptr
    long long ws_stride;
    double* hs;              // device: scratch of the 4 x 4-tile Hessian kernel (qc_mfma64_hess.hip), allocated on first use
    unsigned long long* stamps;  // diagnostic: 16 s_memrealtime slots per interval, or nullptr (normal)
    const void* ell;         // device: row-gather tables of sparse drive generators (qc_mfma32_ell.hip), or nullptr
    int ell_R, ell_slots;    // entries per generator row (1 or 2); drives touching one entry of G at most (0: G from the dense images)
    const void* ell16;       // device: rows of the drive generators of a 2N = 16 handle whose drives have ONE entry per row (Pauli strings):
                             // [drive][16] weights (doubles), then [drive][16] columns x 17 (ints) -- qc_mfma_fused.hip; else nullptr
};

struct qc_fanout;
void qc_fanout_destroy(qc_fanout* f);
struct qc_rccl_state;
void qc_rccl_destroy(qc_rccl_state* r);
#define QC_HOST_RING 4      // pinned blocks of the host path taking turns (QC_HOST_NBUF uses the first 2 .. 4 of them)
struct qc_rearm;
void qc_rearm_destroy(qc_rearm* r);      // waits for the background re-arm jobs of a pinned block

// Selects a HIP device for the lifetime of the object and restores the caller's current device afterwards (every entry
// point uses one: the library never leaves the calling thread on another device than it found).
struct qc_device_guard {
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit qc_device_guard(int device) : dev(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    }
    ~qc_device_guard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    qc_device_guard(const qc_device_guard&) = delete;
    qc_device_guard& operator=(const qc_device_guard&) = delete;
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
    void* dEll = nullptr;      // tables of the sparse-drive kernels (qc_mfma32_ell.hip)
    void* dEll16 = nullptr;    // ... and of the 2N = 16 one-call kernel (qc_mfma_fused.hip)
    // staging for the host-pointer entry points
    double *dZ = nullptr, *dF = nullptr, *dJ = nullptr, *dMu = nullptr, *dH = nullptr;
    unsigned long long* dStamps = nullptr;
    double* hJc = nullptr;     // compact Jacobian values (one copy of the replicated blocks): pinned host staging, device-visible
    double* hFc = nullptr;     // residuals of the direct-to-host path: pinned host staging, device-visible
    double* hZ = nullptr;      // pinned staging of this handle's knots for the host-to-device copy
    // One-launch host path ("landing watch", qc_host_eval.cpp): per interval [ residual rows | compact Jacobian values ], written by
    // the kernel into dC, copied into hC by the copy engine; hC holds a sentinel word wherever the copy has not arrived yet
    // Several pinned blocks take turns: the one a call has consumed is re-armed with the sentinel by pool workers AFTER the call
    // has returned (off the caller's critical path, and off the memory traffic next to the copy engine's writes); the next call
    // that wants that block waits for its re-arm jobs first.
    double* dC = nullptr;
    double* hC[QC_HOST_RING] = {};
    bool hC_armed[QC_HOST_RING] = {};    // completely sentinel-filled once `rearm[i]` has drained (cleared when a call fails midway)
    struct qc_rearm* rearm[QC_HOST_RING] = {};
    int hC_next = 0;
    // Who laid out dC / hC[] last: 1 = this handle's own host-buffer calls, else a hash of the list it led (members, block layout,
    // intervals) -- and for how many doubles.  A different layout takes the blocks afresh (zeroed dC, re-armed hC[]): rows that the new
    // layout's kernels never write must not show the previous layout's values.
    unsigned long long stage_tag = 0;
    size_t stage_cap = 0;
    unsigned long long plain_tag = 0;   // whose rows / values dF, dJ, dH hold (0: nobody's yet, 1: the handle's own host-buffer calls, else the hash of a list
                                        // it leads): zeroed again when another layout takes them (claim_plain)
    bool needs_drain = false;  // a wait on the device ran into QC_HOST_TIMEOUT_MS: the streams still hold that call's work -- the next call drains them first
    hipStream_t drain_s1 = nullptr, drain_s2 = nullptr;   // ... and the same for a MEMBER of a list whose leader's call timed out: the leader's
    int drain_dev = -1;                                   //     streams (by value: the leader may be destroyed before the member's next call)
    int new_x = 1;             // qc_set_new_x: 0 = the knots on the device are current, Z is not read
    bool z_valid = false;      // dZ holds this handle's knots of SOME host-buffer call
    unsigned long long z_gen = 0;    // uploads of the knots so far (qc_knot_generation: what a binding that elides uploads compares)
    hipEvent_t ev_done = nullptr;    // end of the one launch of a host-buffer call
    int host_landing = 1;      // QC_HOST_LANDING=0: the chunked launches of round 2 instead of one watched copy (A/B diagnostics)
    QcParams* dBatch = nullptr;                // device copy of the parameter blocks of a batched launch led by this handle
    std::vector<unsigned long long> batch_members;   // serial numbers of the handles the cached blocks belong to
    QcParams* dBatchLand = nullptr;            // ... and of the batched launch of a host-buffer list call (the landing layout of qc_eval_*_list)
    std::vector<unsigned long long> land_members;
    size_t land_blk = 0;
    unsigned long long serial = 0;             // unique per created handle (a recycled address is not the same handle)
    // multi-device handle (qc_create_multi): no device state of its own, `shards` own the devices
    std::vector<qc_handle*> shards;
    std::vector<long long> shard_F_off, shard_J_off, shard_H_off;   // offsets (doubles) of each shard's slice in the full vectors
    long long shard_chunk = 0;                 // intervals per shard (ceil), the all-gather's chunk
    struct qc_fanout* fan = nullptr;           // one worker thread per shard
    struct qc_rccl_state* rccl = nullptr;      // communicators of qc_multi_all_gather_dev (first use)
    std::vector<hipEvent_t> chunk_events;
    int host_compact = 1;      // QC_HOST_COMPACT=0 disables the compact D2H path of the host-buffer entry points
    double* dWs = nullptr;     // global workspace of the LDS kernels for systems beyond the LDS budget
    double* dHs = nullptr;     // scratch of the 4 x 4-tile Hessian kernel (first Hessian call)
    double *dRE = nullptr, *dRQ = nullptr, *dRS = nullptr, *dRinit = nullptr, *dRout = nullptr, *dRZ = nullptr;   // rollout scratch / staging
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // odd chunks of the direct-to-host path (kernel boundaries of one stream overlap the other's stores)
    hipEvent_t ev_staged = nullptr;  // the knots are on the device (recorded on `stream`, waited for by `stream2`)
    std::string err;
};

int qc_fail(std::string* err, int code, const std::string& msg);   // records the message (handle and thread-local), returns code
#define QC_HIP(h, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            return qc_fail(&(h)->err, QC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                                           \
    } while (0)

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
size_t qc_lds_exp_hess_bytes(const QcParams& P);     // exponential integrator's mu_d2F, any size: qc_lds_exp_hess.hip
hipError_t qc_launch_lds_exp_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, size_t lds, hipStream_t st);
bool qc_mfma_supported(const QcParams& P);
bool qc_mfma_hess_supported(const QcParams& P);
bool qc_mfma_compact_supported(const QcParams& P);   // the F + dF kernel honours QcParams.copies (order-4 kernels, 2N <= 32)
size_t qc_mfma_gx_doubles(const QcParams& P);
void qc_mfma_pack_G(const QcParams& P, const double* G_host, double* Gx_host);
hipError_t qc_launch_mfma_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
size_t qc_mfma32_gx_doubles(const QcParams& P);
void qc_mfma32_pack_G(const QcParams& P, const double* G_host, double* Gx_host);
hipError_t qc_launch_mfma32_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
bool qc_mfma16_padeP_supported(const QcParams& P);
hipError_t qc_launch_mfma16_padeP(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
bool qc_mfma16_padeP_hess_supported(const QcParams& P);
hipError_t qc_launch_mfma16_padeP_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
bool qc_mfma64_supported(const QcParams& P);
size_t qc_mfma64_gx_doubles(const QcParams& P);
void qc_mfma64_pack_G(const QcParams& P, const double* G_host, double* Gx_host);
hipError_t qc_launch_mfma64_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
bool qc_mfma64_hess_supported(const QcParams& P);
size_t qc_mfma64_hess_scratch_doubles(const QcParams& P);
hipError_t qc_launch_mfma64_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
bool qc_mfma16_batchable(const QcParams& P);
hipError_t qc_launch_mfma16_F_jac_batch(const QcParams& P0, const QcParams* dPb, int count, const double* dZ, double* dF, double* dJ,
                                        hipStream_t st);
hipError_t qc_launch_mfma16_hess_batch(const QcParams& P0, const QcParams* dPb, int count, const double* dZ, const double* dMu, double* dH,
                                       hipStream_t st);
bool qc_mfma_exp_supported(const QcParams& P);
hipError_t qc_launch_mfma_exp(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
bool qc_mfma_exp_hess_supported(const QcParams& P);   // mu_d2F of the exponential integrator, 2N <= 16: qc_mfma_exp_hess.hip
hipError_t qc_launch_mfma_exp_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
bool qc_mfma32_exp_hess_supported(const QcParams& P);  // ... 16 < 2N <= 32: qc_mfma32_exp_hess.hip
hipError_t qc_launch_mfma32_exp_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
bool qc_mfma32_exp_supported(const QcParams& P);
hipError_t qc_launch_mfma32_exp(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
bool qc_mfma32_hess_supported(const QcParams& P);
// sparse drive generators (at most 2 entries per row), 2N = 32, Hermitian Hamiltonians: qc_mfma32_ell.hip
int qc_mfma32_ell_build(const QcParams& P, const double* G_host, std::vector<char>* blob, int* slots_out);
bool qc_mfma16_ell_build(const QcParams& P, const double* G_host, std::vector<char>* blob);   // 2N = 16, one entry per drive row: qc_mfma_fused.hip
bool qc_exp_ell_build(const QcParams& P, const double* G_host, std::vector<char>* blob);   // exponential integrator, 2N <= 32, at most one entry per drive row: qc_mfma_exp_hess.hip
void qc_mfma16_ell_pair_table(const QcParams& P, std::vector<char>* blob);                    // ... its pair table for the (a, a) block: qc_mfma_hess_g2.hip
bool qc_mfma16_hess_g2(const QcParams& P);
hipError_t qc_launch_mfma16_hess_g2(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
bool qc_mfma16_hess_gathers(const QcParams& P);      // ... the mu_d2F launches (qc_mfma_hess.hip, qc_mfma_hess2.hip)
bool qc_mfma16_fused_gathers(const QcParams& P);     // ... and whether the one-call launch of this handle takes the row-gather form
hipError_t qc_launch_mfma32_ell_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
hipError_t qc_launch_mfma32_ell_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st);
hipError_t qc_launch_mfma32_ell_fused(const QcParams& P, const double* dZ, const double* dMu, double* dF, double* dJ, double* dH, hipStream_t st);
hipError_t qc_launch_mfma32_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
// F + dF + mu_d2F in one launch (qc_mfma_fused.hip): 2N = 16, a unitary on 8 levels, antisymmetric generators, 1 .. 6 drives
bool qc_mfma16_fused_supported(const QcParams& P);
bool qc_mfma16_hess2_supported(const QcParams& P);
hipError_t qc_launch_mfma16_hess2(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);
hipError_t qc_launch_mfma16_fused(const QcParams& P, const double* dZ, const double* dMu, double* dF, double* dJ, double* dH, hipStream_t st);
hipError_t qc_launch_mfma_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st);

hipError_t qc_launch_pack_jac(const double* dJ, double* dJc, int n_int, int jac_nnz, int comp_len, int n2, int jo_F, int jo_B, int head2,
                              int tail_src, hipStream_t st);
bool qc_rollout_supported(const QcParams& P);
void qc_rollout_scratch(const QcParams& P, long long T, size_t* nE, size_t* nQ, size_t* nS, int* chunk, int* n_chunks);
hipError_t qc_launch_rollout(const QcParams& P, long long T, const double* dZ, const double* dinit, double* dout, double* dE, double* dQ,
                             double* dS, hipStream_t st);

// Diagnostic time stamps (DIAG instantiations only; handle created with QC_STAMPS=1; never in a timed run).
// QC_STAMP records s_memrealtime (100 MHz) into a per-wave register array; QC_STAMP_FLUSH writes the slots
// the wave owns at its very end.  (Writing each stamp to memory where it is taken stalls the wave behind the
// store queue it is trying to observe.)
#define QC_STAMP_DECL unsigned long long qc_ts_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define QC_STAMP(P, b, lane, k)                                                                  \
    do {                                                                                         \
        if constexpr (DIAG) {                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            qc_ts_[k] = __builtin_amdgcn_s_memrealtime();                                        \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    } while (0)
#define QC_STAMP_CYCLES(k)                                                                       \
    do {                                                                                         \
        if constexpr (DIAG) qc_ts_[k] = __builtin_amdgcn_s_memtime();                            \
    } while (0)
#define QC_STAMP_FLUSH(P, b, lane, first, last)                                                  \
    do {                                                                                         \
        if constexpr (DIAG) {                                                                    \
            if ((P).stamps != nullptr && (lane) == 0) {                                          \
                _Pragma("unroll") for (int k_ = (first); k_ <= (last); ++k_)                     \
                    (P).stamps[(size_t)(b) * 16 + k_] = qc_ts_[k_];                              \
            }                                                                                    \
        }                                                                                        \
    } while (0)

// Streaming store of one output value.  The outputs are written once and never re-read by the kernel.
// mode 0 plain, 1 write-through (sc1: agent-scope relaxed atomic store), 2 non-temporal (default; measured
// fastest, profiles/README.md).  The mode is a COMPILE-TIME constant in the MFMA kernels: as a run-time
// switch it put a scalar branch ladder (and SGPR spills) around each of the ~80 stores of a wave and halved
// the store issue rate.
template <int MODE>
__device__ inline void qc_st8m(double* p, double v) {
    if constexpr (MODE == 1) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (MODE == 2) {
        __builtin_nontemporal_store(v, p);
    } else {
        *p = v;
    }
}
__device__ inline void qc_st8(double* p, double v, int mode) {
    if (mode == 1) qc_st8m<1>(p, v);
    else if (mode == 2) qc_st8m<2>(p, v);
    else qc_st8m<0>(p, v);
}

// Kernel-argument warm-up.  QcParams travels by value in the kernarg segment (~0.6 KB = 10 cache lines); the compiler reads
// its fields with scalar loads in several DEPENDENT batches (load, wait, branch, load more ...), and every batch that touches a
// new line is a scalar-cache miss served from L2 / HBM (the command processor has just written the segment): four to five
// serial round trips of 0.3 - 0.5 us in front of the first global load of every wave (with the segment in host memory,
// HIP_FORCE_DEV_KERNARG=0, the config-3 launch takes 16.3 instead of 11.3 us).  One dword of EVERY line is requested here in a
// single batch, so the later batches hit the scalar cache.  BYTES = size of the kernel's argument block.
template <int BYTES>
__device__ __forceinline__ void qc_kernarg_touch() {
    typedef __attribute__((address_space(4))) const int kint;
    kint* k = (kint*)__builtin_amdgcn_kernarg_segment_ptr();
    int w = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) w += k[o / 4];
    asm volatile("" ::"s"(w));   // keeps the loads; nothing depends on the sum
}

// The same in two halves: the requests at one point, the (only) wait for them at another -- for kernels whose first vector
// load requests depend on preloaded arguments only and must not stand behind a scalar wait (qc_mfma_kernels.hip).
template <int BYTES>
struct QcKernargTouch {
    static constexpr int kN = (BYTES + 63) / 64;
    int v[kN];
    __device__ __forceinline__ void request() {
        typedef __attribute__((address_space(4))) const int kint;
        kint* k = (kint*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
        for (int i = 0; i < kN; ++i) v[i] = k[i * 16];
    }
    __device__ __forceinline__ void consume() {
#pragma unroll
        for (int i = 0; i < kN; ++i) asm volatile("" ::"s"(v[i]));
    }
};

// Tail of an interval's Hessian block: the derivative integrators' entries d2/d(dx_i) dh = -mu_i (free timestep only) and the
// alignment padding (explicit zeros; qc_desc.hess_align).  `mu`, `Hb` point at this handle's rows / values of the interval.
__device__ inline void qc_hess_tail(const QcParams& P, const double* __restrict__ mu, double* __restrict__ Hb, int tid, int nthreads) {
    if (P.off_dt >= 0) {
        int o = P.ho_d;
        for (int d = 0; d < P.n_deriv; ++d) {
            for (int i = tid; i < P.ddim_i[d]; i += nthreads) Hb[o + i] = -mu[P.drow[d] + i];
            o += P.ddim_i[d];
        }
    }
    for (int i = tid; i < P.h_pad; i += nthreads) Hb[P.hess_nnz + i] = 0.0;
}

// XCD-aware block -> local interval map: blocks b and b+8 share an XCD (round-robin dispatch,
// MI355X_MICROARCH.md "Workgroup dispatch"), so each XCD gets a contiguous run of intervals and the
// shared knot z_{t+1} of neighbouring intervals is served by one L2.  Bijective for any nb.
__host__ __device__ inline int qc_xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}
