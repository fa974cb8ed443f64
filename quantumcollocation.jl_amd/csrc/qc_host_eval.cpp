// Host side of libqcolloc_hip.so, part 2: the evaluation entry points -- device-resident launches, the host-buffer
// path (H2D of the knots, kernels, compact D2H with host-side replication), rollouts, and the multi-device handle
// (one process, N GPUs: per-shard threads / streams / pinned staging, optional in-library RCCL all-gather).
// Descriptor validation, structures and handle lifetime are in qc_host.cpp.  No CPU evaluation path anywhere.
#include <dlfcn.h>
#include <link.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "qc_internal.h"
#include "qc_host_team.h"

#define fail qc_fail
using qc_team::HostGroup;
using qc_team::HostPool;
using qc_team::host_pool;
using qc_team::host_copy;
using qc_team::host_trace;
using qc_team::now_us;
using qc_team::cpu_pause;
using qc_team::kLandSentinel;
using qc_team::LandJob;
using qc_team::land_piece_intervals;

static int check_align(qc_handle* h, const void* p, size_t a, const char* what) {
    if (p && ((uintptr_t)p % a) != 0) return fail(&h->err, QC_ERR_INVALID, std::string(what) + " is not sufficiently aligned");
    return QC_OK;
}

static bool is_multi(const qc_handle* h) { return !h->shards.empty(); }

// A wait on the device that ran into QC_HOST_TIMEOUT_MS: the call returns an error instead of spinning for ever.  Nothing is
// synchronised here (a hung device would hang that, too): the handle's streams may still hold work that writes into the call's output
// buffers (an asynchronous copy into the caller's array, the residual kernel writing into pinned memory in place) and into the
// handle's staging.  The handle remembers (needs_drain): its next call -- or qc_destroy -- waits for the streams before anything is
// reused, and qcolloc.h tells the caller to keep the failed call's output buffers allocated until then.
static int timed_out(qc_handle* h, const char* what) {
    for (int i = 0; i < QC_HOST_RING; ++i) h->hC_armed[i] = false;
    h->needs_drain = true;
    return fail(&h->err, QC_ERR_HIP, std::string("timed out after ") + std::to_string((long long)(qc_team::timeout_us() / 1e3)) +
                                         " ms waiting for " + what + " on the device (QC_HOST_TIMEOUT_MS)");
}

// first thing of every host-buffer call (device selected): finish what a timed-out call left on the streams.  Polled against the
// same deadline: on a device that is still hung the call returns QC_ERR_HIP again and the handle keeps the flag -- an
// unbounded hipStreamSynchronize here would have made the deadline work once per handle only.
static int drain_stream(qc_handle* h, hipStream_t st, bool foreign = false) {
    if (!st) return QC_OK;
    const double t0 = now_us();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return QC_OK;
        if (foreign && e != hipErrorNotReady) { (void)hipGetLastError(); return QC_OK; }   // the leader is gone: qc_destroy waited for its streams
        if (e != hipErrorNotReady) return fail(&h->err, QC_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        if ((spins & 63) == 63 && now_us() - t0 > qc_team::timeout_us())
            return fail(&h->err, QC_ERR_HIP, "the device has still not finished the work of a call that timed out (QC_HOST_TIMEOUT_MS): "
                                               "keep that call's buffers allocated and call again, or destroy the handle");
        cpu_pause();
    }
}
static int drain_if_needed(qc_handle* h) {
    if (!h->needs_drain) return QC_OK;
    int rc;
    if ((rc = drain_stream(h, h->stream)) || (rc = drain_stream(h, h->stream2))) return rc;      // (needs_drain stays set)
    h->needs_drain = false;
    return QC_OK;
}
// A list call runs on its LEADER's streams and staging but writes the caller's arrays for every member: after a time-out every
// member must drain the leader's streams before its own next call reuses or releases anything.
static int drain_leader_if_needed(qc_handle* h) {
    if (h->drain_dev < 0) return QC_OK;
    qc_device_guard guard(h->drain_dev);
    QC_HIP(h, guard.err);
    int rc;
    if ((rc = drain_stream(h, h->drain_s1, true)) || (rc = drain_stream(h, h->drain_s2, true))) return rc;
    h->drain_dev = -1;
    return QC_OK;
}
static void mark_members(qc_handle* const* hs, int count) {      // (the leader's streams by value: the leader may be destroyed first)
    for (int i = 1; i < count; ++i) {
        hs[i]->drain_s1 = hs[0]->stream;
        hs[i]->drain_s2 = hs[0]->stream2;
        hs[i]->drain_dev = hs[0]->device;
    }
}
// dF / dJ / dH are zeroed when they are made and hold the rows / values of ONE layout at a time: the handle's own host-buffer calls
// (tag 1) or a list it leads (a hash of the members).  Another layout zeroes them again: rows that the new layout's kernels never
// write (QC_ROWS_BY_COMPONENT; other members' rows) must not show the previous layout's numbers.
static int claim_plain(qc_handle* h, unsigned long long tag) {
    if (h->plain_tag == tag) return QC_OK;
    const QcParams& P = h->prm;
    const size_t n_int = (size_t)P.n_int;
    if (h->plain_tag != 0) {      // (fresh buffers are zeroed by ensure_zeroed)
        if (h->dF) QC_HIP(h, hipMemsetAsync(h->dF, 0, n_int * (size_t)P.F_stride * sizeof(double), h->stream));
        if (h->dJ) QC_HIP(h, hipMemsetAsync(h->dJ, 0, n_int * (size_t)P.J_stride * sizeof(double), h->stream));
        if (h->dH) QC_HIP(h, hipMemsetAsync(h->dH, 0, n_int * (size_t)P.H_stride * sizeof(double), h->stream));
    }
    h->plain_tag = tag;
    return QC_OK;
}

#define QC_NOT_MULTI(h, name)                                                                                        \
    do {                                                                                                             \
        if (is_multi(h))                                                                                             \
            return fail(&(h)->err, QC_ERR_INVALID, std::string(name) + ": a multi-device handle has no single device; " \
                        "use its shard handles (qc_multi_shard) or the qc_multi_* entry points");                     \
    } while (0)

// ------------------------------------------------------------------------------------------------
//  Device-resident evaluation
// ------------------------------------------------------------------------------------------------
extern "C" int qc_eval_F_jac_dev(qc_handle* h, const double* dZ, double* dF, double* dvals, void* stream) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_eval_F_jac_dev: NULL handle");
    QC_NOT_MULTI(h, "qc_eval_F_jac_dev");
    if (!dZ || (!dF && !dvals)) return fail(&h->err, QC_ERR_INVALID, "qc_eval_F_jac_dev: NULL buffer");
    int rc;
    if ((rc = check_align(h, dZ, 8, "dZ"))) return rc;
    if ((rc = check_align(h, dF, 8, "dF"))) return rc;
    if ((rc = check_align(h, dvals, 8, "dvals"))) return rc;
    if (h->prm.n_int == 0) return QC_OK;
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    hipError_t e;
    // (Keeping the parameter block in device memory instead of the kernarg segment -- the batched launch's mechanism with one
    // handle -- was measured: 10.9 instead of 10.55 us per config-3 launch.)
    if (h->kernel == QC_KERNEL_MFMA) e = qc_launch_mfma_F_jac(h->prm, dZ, dF, dvals, (hipStream_t)stream);
    else e = qc_launch_lds_F_jac(h->prm, dZ, dF, dvals, h->lds_bytes_jac, (hipStream_t)stream);
    if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

// ------------------------------------------------------------------------------------------------
//  Several handles in one launch (the systems of a sampling problem)
// ------------------------------------------------------------------------------------------------
// Returns 1 when the handles can share a launch (and hs[0]->dBatch holds their parameter blocks), 0 when not, < 0 on error.
static int prepare_batch(qc_handle* const* hs, int32_t count, bool hessian) {
    qc_handle* h0 = hs[0];
    for (int i = 0; i < count; ++i) {
        if (is_multi(hs[i])) return fail(&h0->err, QC_ERR_INVALID, "batched launch: multi-device handles are not accepted");
        if (hs[i]->device != h0->device) return fail(&h0->err, QC_ERR_INVALID, "batched launch: the handles are bound to different devices");
    }
    if (count < 2 || count > 65535) return 0;
    for (int i = 0; i < count; ++i) {
        const qc_handle* h = hs[i];
        if (h->kernel != QC_KERNEL_MFMA || !qc_mfma16_batchable(h->prm)) return 0;
        if (h->prm.m > 8 && hessian) return 0;
        if (hessian && h->prm.antisym != h0->prm.antisym) return 0;   // (the antisymmetric generators' Hessian kernel is a different one)
        if (h->prm.n_int != h0->prm.n_int || h->prm.t_begin != h0->prm.t_begin || h->prm.zdim != h0->prm.zdim || h->prm.m != h0->prm.m ||
            h->prm.n != h0->prm.n || h->prm.nc != h0->prm.nc)
            return 0;
    }
    bool same = h0->dBatch != nullptr && (int)h0->batch_members.size() == count;
    for (int i = 0; same && i < count; ++i) same = h0->batch_members[i] == hs[i]->serial;
    if (same) return 1;
    qc_device_guard guard(h0->device);
    QC_HIP(h0, guard.err);
    if (h0->dBatch) { (void)hipFree(h0->dBatch); h0->dBatch = nullptr; }
    std::vector<QcParams> blocks(count);
    for (int i = 0; i < count; ++i) blocks[i] = hs[i]->prm;
    QC_HIP(h0, hipMalloc((void**)&h0->dBatch, sizeof(QcParams) * count));
    QC_HIP(h0, hipMemcpy(h0->dBatch, blocks.data(), sizeof(QcParams) * count, hipMemcpyHostToDevice));
    h0->batch_members.clear();
    for (int i = 0; i < count; ++i) h0->batch_members.push_back(hs[i]->serial);
    return 1;
}

extern "C" int qc_eval_F_jac_dev_multi(qc_handle* const* hs, int32_t count, const double* dZ, double* dF, double* dvals, void* stream) {
    if (!hs || count < 1) return fail(nullptr, QC_ERR_INVALID, "qc_eval_F_jac_dev_multi: no handles");
    for (int i = 0; i < count; ++i) if (!hs[i]) return fail(nullptr, QC_ERR_INVALID, "qc_eval_F_jac_dev_multi: NULL handle");
    qc_handle* h0 = hs[0];
    if (!dZ || (!dF && !dvals)) return fail(&h0->err, QC_ERR_INVALID, "qc_eval_F_jac_dev_multi: NULL buffer");
    int rc;
    if ((rc = check_align(h0, dZ, 8, "dZ"))) return rc;
    if ((rc = check_align(h0, dF, 8, "dF"))) return rc;
    if ((rc = check_align(h0, dvals, 8, "dvals"))) return rc;
    const int ok = prepare_batch(hs, count, false);
    if (ok < 0) return ok;
    if (ok == 0 || h0->prm.n_int == 0) {   // shapes differ or not the batchable kernel: one launch per handle
        for (int i = 0; i < count; ++i) if ((rc = qc_eval_F_jac_dev(hs[i], dZ, dF, dvals, stream))) return rc;
        return QC_OK;
    }
    qc_device_guard guard(h0->device);
    QC_HIP(h0, guard.err);
    hipError_t e = qc_launch_mfma16_F_jac_batch(h0->prm, h0->dBatch, count, dZ, dF, dvals, (hipStream_t)stream);
    if (e != hipSuccess) return fail(&h0->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

extern "C" int qc_eval_hess_dev_multi(qc_handle* const* hs, int32_t count, const double* dZ, const double* dmu, double* dhvals, void* stream) {
    if (!hs || count < 1) return fail(nullptr, QC_ERR_INVALID, "qc_eval_hess_dev_multi: no handles");
    for (int i = 0; i < count; ++i) if (!hs[i]) return fail(nullptr, QC_ERR_INVALID, "qc_eval_hess_dev_multi: NULL handle");
    qc_handle* h0 = hs[0];
    if (!dZ || !dmu || !dhvals) return fail(&h0->err, QC_ERR_INVALID, "qc_eval_hess_dev_multi: NULL buffer");
    int rc;
    if ((rc = check_align(h0, dZ, 8, "dZ"))) return rc;
    if ((rc = check_align(h0, dmu, 8, "dmu"))) return rc;
    if ((rc = check_align(h0, dhvals, 8, "dhvals"))) return rc;
    const int ok = prepare_batch(hs, count, true);
    if (ok < 0) return ok;
    bool hess_ok = ok == 1 && h0->prm.n_int > 0;
    for (int i = 0; hess_ok && i < count; ++i) hess_ok = hs[i]->prm.hess_nnz > 0 && qc_mfma_hess_supported(hs[i]->prm);
    if (!hess_ok) {
        for (int i = 0; i < count; ++i) if ((rc = qc_eval_hess_dev(hs[i], dZ, dmu, dhvals, stream))) return rc;
        return QC_OK;
    }
    qc_device_guard guard(h0->device);
    QC_HIP(h0, guard.err);
    hipError_t e = qc_launch_mfma16_hess_batch(h0->prm, h0->dBatch, count, dZ, dmu, dhvals, (hipStream_t)stream);
    if (e != hipSuccess) return fail(&h0->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

extern "C" int qc_eval_hess_dev(qc_handle* h, const double* dZ, const double* dmu, double* dhvals, void* stream) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_eval_hess_dev: NULL handle");
    QC_NOT_MULTI(h, "qc_eval_hess_dev");
    if (h->prm.hess_nnz == 0) return QC_OK;   // no drives and a fixed timestep: the constraint is linear
    if (!dZ || !dmu || !dhvals) return fail(&h->err, QC_ERR_INVALID, "qc_eval_hess_dev: NULL buffer");
    int rc;
    if ((rc = check_align(h, dZ, 8, "dZ"))) return rc;
    if ((rc = check_align(h, dmu, 8, "dmu"))) return rc;
    if ((rc = check_align(h, dhvals, 8, "dhvals"))) return rc;
    if (h->prm.n_int == 0) return QC_OK;
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    hipError_t e;
    if (h->kernel == QC_KERNEL_MFMA && qc_mfma64_hess_supported(h->prm) && !h->dHs) {   // 128 MiB of scratch, first Hessian call only
        QC_HIP(h, hipMalloc((void**)&h->dHs, qc_mfma64_hess_scratch_doubles(h->prm) * sizeof(double)));
        h->prm.hs = h->dHs;
    }
    if (h->kernel == QC_KERNEL_MFMA && qc_mfma_exp_hess_supported(h->prm))
        e = qc_launch_mfma_exp_hess(h->prm, dZ, dmu, dhvals, (hipStream_t)stream);
    else if (h->kernel == QC_KERNEL_MFMA && qc_mfma32_exp_hess_supported(h->prm))
        e = qc_launch_mfma32_exp_hess(h->prm, dZ, dmu, dhvals, (hipStream_t)stream);
    else if (h->kernel == QC_KERNEL_MFMA && qc_mfma_hess_supported(h->prm))
        e = qc_launch_mfma_hess(h->prm, dZ, dmu, dhvals, (hipStream_t)stream);
    else
        e = qc_launch_lds_hess(h->prm, dZ, dmu, dhvals, h->lds_bytes_hess, (hipStream_t)stream);
    if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

// Jacobian and Hessian of the Lagrangian at one point in ONE launch where a fused kernel serves the handle (qc_mfma_fused.hip),
// else the two launches, on the same stream: the result is the same bit for bit either way.
extern "C" int qc_eval_F_jac_hess_dev(qc_handle* h, const double* dZ, const double* dmu, double* dF, double* dvals, double* dhvals, void* stream) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_eval_F_jac_hess_dev: NULL handle");
    QC_NOT_MULTI(h, "qc_eval_F_jac_hess_dev");
    if (!dZ || !dvals || (h->prm.hess_nnz && (!dmu || !dhvals))) return fail(&h->err, QC_ERR_INVALID, "qc_eval_F_jac_hess_dev: NULL buffer");
    int rc;
    if ((rc = check_align(h, dZ, 8, "dZ")) || (rc = check_align(h, dmu, 8, "dmu")) || (rc = check_align(h, dF, 8, "dF")) ||
        (rc = check_align(h, dvals, 8, "dvals")) || (rc = check_align(h, dhvals, 8, "dhvals")))
        return rc;
    if (h->prm.n_int == 0) return QC_OK;
    static const bool allow = !(getenv("QC_NO_FUSED") && atoi(getenv("QC_NO_FUSED")));
    if (allow && h->kernel == QC_KERNEL_MFMA && qc_mfma16_fused_supported(h->prm)) {
        qc_device_guard guard(h->device);
        QC_HIP(h, guard.err);
        const hipError_t e = qc_launch_mfma16_fused(h->prm, dZ, dmu, dF, dvals, dhvals, (hipStream_t)stream);
        if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        return QC_OK;
    }
    if (allow && h->kernel == QC_KERNEL_MFMA && h->prm.ell && h->prm.hess_nnz) {      // sparse drive generators at 2N = 32: qc_mfma32_ell.hip
        qc_device_guard guard(h->device);
        QC_HIP(h, guard.err);
        const hipError_t e = qc_launch_mfma32_ell_fused(h->prm, dZ, dmu, dF, dvals, dhvals, (hipStream_t)stream);
        if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        return QC_OK;
    }
    if (h->prm.hess_nnz == 0) return qc_eval_F_jac_dev(h, dZ, dF, dvals, stream);
    // Two kernels, one after the other.  (mu_d2F on a second stream beside F + dF, between two events, was measured at config 5,
    // where the first is bound by its stores and the second by the matrix pipes: 67 us against 53 -- the fork and join cost more
    // than the overlap gives.)
    if ((rc = qc_eval_F_jac_dev(h, dZ, dF, dvals, stream))) return rc;
    return qc_eval_hess_dev(h, dZ, dmu, dhvals, stream);
}

// ------------------------------------------------------------------------------------------------
//  Host-buffer evaluation (H2D -> kernel -> D2H, synchronous)
// ------------------------------------------------------------------------------------------------
static int ensure(qc_handle* h, double** p, size_t count) {
    if (*p || count == 0) return QC_OK;
    QC_HIP(h, hipMalloc((void**)p, count * sizeof(double)));
    return QC_OK;
}
static int ensure_zeroed(qc_handle* h, double** p, size_t count) {   // rows no kernel writes (QC_ROWS_BY_COMPONENT) stay 0
    if (*p || count == 0) return QC_OK;
    QC_HIP(h, hipMalloc((void**)p, count * sizeof(double)));
    // on the handle's stream: it is a non-blocking stream, a memset on the null stream would race with the kernel behind it
    QC_HIP(h, hipMemsetAsync(*p, 0, count * sizeof(double), h->stream));
    return QC_OK;
}
static int ensure_pinned(qc_handle* h, double** p, size_t count, bool zero) {
    if (*p || count == 0) return QC_OK;
    QC_HIP(h, hipHostMalloc((void**)p, count * sizeof(double), hipHostMallocDefault));
    if (zero) memset(*p, 0, count * sizeof(double));
    return QC_OK;
}

// A handle whose VALUE vectors are shared with other handles (the integrator groups of a sampling problem) cannot use the
// host-buffer entry points; a handle that only places its rows inside a wider row block (QC_ROWS_BY_COMPONENT) can.
static bool shares_values(const qc_handle* h) {
    const QcParams& P = h->prm;
    return P.J_stride != P.jac_nnz || (P.hess_nnz && P.H_stride != P.hess_nnz + P.h_pad) || P.J_off || P.H_off ||
           (h->desc.row_placement == QC_ROWS_STACKED && (P.F_stride != P.ddim || P.F_off));
}

// Compact transfer of the Jacobian values to a host buffer.  Of the 5040 values of a config-3 interval 4096 are the
// N copies of -F and of B (I_N (x) B): only one copy of each crosses PCIe (3.5x fewer bytes), worker threads
// replicate it into the caller's array while the next chunk is in flight.  The device-resident entry points are not
// affected (they always produce the full value vector).  Two forms:
//   direct  the order-4 MFMA kernels (2N = 16 and 32: BASELINE configs 1-5) write the compact form -- and the residuals --
//           STRAIGHT into pinned host memory, one launch per chunk of intervals (QcParams.copies = 1): no full-size value
//           vector in HBM, no gather kernels;
//   packed  every other kernel writes the full vector to HBM and qc_pack_jac_kernel gathers the compact form chunk by chunk.
// Diagnostics: QC_HOST_COMPACT=0 (plain full copy), =2 (packed form even where direct is available), QC_HOST_THREADS,
// QC_HOST_CHUNKS.
struct CompactPlan {
    bool useful;
    int n2, head2, tail_src, tail_len, comp_len, copies, second_copies;
};

static CompactPlan compact_plan(const QcParams& P) {
    CompactPlan c{};
    c.n2 = P.n * P.n;
    c.copies = P.nc;
    const bool pade = P.integrator == QC_PADE;
    c.second_copies = pade ? P.nc : 0;            // the exponential integrator's d/dU_{t+1} block is an identity (s entries, no copies)
    c.head2 = pade ? 2 * c.n2 : c.n2;
    c.tail_src = pade ? P.jo_a : P.jo_B;
    c.tail_len = P.jac_nnz - c.tail_src;
    c.comp_len = c.head2 + c.tail_len;
    c.useful = P.nc > 1 && P.jo_F == 0 && P.jo_B == P.nc * c.n2 && (!pade || P.jo_a == 2 * P.nc * c.n2);
    return c;
}

static void expand_intervals(const QcParams& P, const CompactPlan& cp, const double* comp, double* vals, int b0, int b1) {
    const qc_copy_fn cpy = host_copy();
    for (int b = b0; b < b1; ++b) {
        const double* src = comp + (size_t)b * cp.comp_len;
        double* dst = vals + (size_t)b * P.jac_nnz;
        for (int c = 0; c < cp.copies; ++c) cpy(dst + P.jo_F + (size_t)c * cp.n2, src, (size_t)cp.n2);
        for (int c = 0; c < cp.second_copies; ++c) cpy(dst + P.jo_B + (size_t)c * cp.n2, src + cp.n2, (size_t)cp.n2);
        cpy(dst + cp.tail_src, src + cp.head2, (size_t)cp.tail_len);
    }
    qc_host_copy_fence();
}

static int usable_cores() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2 CPU quota ("max" or "<quota> <period>")
        long long q = 0, per = 0;
        if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, q / per));
        fclose(f);
    }
    return std::max(1, n);
}

static int pool_workers(int shards) {
    static const int cores = usable_cores();
    int w = std::min(cores, shards > 1 ? 16 : 8);   // measured best on the MI355X host (16-CPU quota): 8 workers, 16 chunks per handle
    if (const char* ev = getenv("QC_HOST_THREADS")) w = std::max(1, std::min(64, atoi(ev)));
    return std::max(1, w);
}

// The worker pool, the landing watch and the pinned ring's re-arm jobs are in qc_host_team.h: plain C++, no HIP, so that
// tests/host_team_test.cpp can drive them on the CPU under -fsanitize=thread / address with a thread standing in for the copy engine.
namespace {
// CPU sets of the L3 domains (core complexes) on the NUMA node the device hangs off, from sysfs; empty when anything is unreadable.
// A core complex of the MI355X hosts' EPYC has one link to the memory controllers (~30 - 60 GB/s of writes): the replication
// team writes 41.5 MB per config-3 call and only keeps up with the PCIe link when its members sit on DIFFERENT complexes, next to
// the memory the GPU's copy engine writes.  Left to the scheduler inside a 16-CPU quota on a 256-thread host they land anywhere,
// often several to a complex and on the far socket (run-to-run spread 0.30 - 0.45 ms per call).
static bool parse_cpulist(const char* path, cpu_set_t* set) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    CPU_ZERO(set);
    int a, b;
    bool any = false;
    for (;;) {
        if (fscanf(f, "%d", &a) != 1) break;
        b = a;
        int c = fgetc(f);
        if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
        for (int i = a; i <= b && i < CPU_SETSIZE; ++i) { CPU_SET(i, set); any = true; }
        if (c != ',') break;
    }
    fclose(f);
    return any;
}
static std::vector<cpu_set_t> device_l3_domains(int device) {
    std::vector<cpu_set_t> out;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return out;
    for (char* p = bus; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return out;
    cpu_set_t node_set, allowed;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    if (!parse_cpulist(path, &node_set)) return out;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
    cpu_set_t seen;
    CPU_ZERO(&seen);
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &node_set) || !CPU_ISSET(c, &allowed) || CPU_ISSET(c, &seen)) continue;
        cpu_set_t dom;
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
        if (!parse_cpulist(path, &dom)) return std::vector<cpu_set_t>();
        cpu_set_t use;
        CPU_AND(&use, &dom, &allowed);
        CPU_OR(&seen, &seen, &dom);
        if (CPU_COUNT(&use) > 0) out.push_back(use);
    }
    return out;
}

}  // namespace

// Chunks of the compact transfer: at most two per worker, at least 16 intervals and ~256 KB of compact values each -- a chunk is
// a kernel launch (~5 us of latency each, serialised on two streams), so the small problems (config 1: 40 KB, config 2: 0.5 MB
// in all) go out in one or two.
static int chunk_count(const qc_handle* h, int n_int, int workers, size_t bytes_per_interval) {
    const int min_chunk = 16;
    const size_t min_bytes = 256 << 10;
    const size_t by_bytes = std::max<size_t>(1, (size_t)n_int * bytes_per_interval / min_bytes);
    int n_chunks = std::max(1, std::min<int>(std::min<size_t>(2 * workers, by_bytes), (n_int + min_chunk - 1) / min_chunk));
    if (const char* ev = getenv("QC_HOST_CHUNKS")) n_chunks = std::max(1, std::min(n_int, atoi(ev)));
    (void)h;
    return n_chunks;
}

static int ensure_events(qc_handle* h, int n) {
    if ((int)h->chunk_events.size() < n) {
        const size_t old = h->chunk_events.size();
        h->chunk_events.resize(n, nullptr);
        for (size_t k = old; k < h->chunk_events.size(); ++k) QC_HIP(h, hipEventCreateWithFlags(&h->chunk_events[k], hipEventDisableTiming));
    }
    return QC_OK;
}

// Parameters of the compact value layout [ -F (n2) | B (n2) | drive columns ... ] with ONE copy of the replicated blocks,
// for the kernels that can write it themselves (QcParams.copies).
static QcParams compact_params(const QcParams& P, const CompactPlan& cp) {
    QcParams C = P;
    const int shift = cp.head2 - cp.tail_src;   // every block behind the replicated ones moves up
    C.copies = 1;
    C.jo_F = 0;
    C.jo_B = cp.n2;
    C.jo_a = P.jo_a + shift;
    C.jo_h = P.jo_h + shift;
    C.jo_d = P.jo_d + shift;
    C.jac_nnz = cp.comp_len;
    C.J_stride = cp.comp_len;
    C.J_off = 0;
    return C;
}

// Chunked tail of a host evaluation: `produce(k, b0, b1)` enqueues the work that makes chunk k's compact values (and
// residuals) appear in the pinned buffers; the worker pool expands each chunk into the caller's arrays as it lands.
static thread_local double g_call_begin = 0.0;   // entry of the host-buffer call being traced (QC_HOST_TRACE)

// (One launch over all intervals in order, with per-chunk completion counters added to by the kernel in pinned host memory and
// polled here, was built and measured: every interval then has to drain its PCIe writes (s_waitcnt vmcnt(0)) before it can be
// counted, 137 us per round of 128 intervals -- 1.2 ms per evaluation instead of 0.5.  Rejected.)
// `two_streams`: odd chunks are produced on h->stream2 (produce() picks the stream by the chunk's parity).
static int run_chunks(qc_handle* h, const CompactPlan& cp, double* F, double* vals, int shards,
                      const std::function<int(int, int, int)>& produce, bool two_streams = false) {
    const QcParams& P = h->prm;
    const double t_begin = now_us();
    HostPool& pool = host_pool();
    const int workers = pool_workers(shards);
    pool.ensure(workers);
    // (residuals only: ~1 MB in all; QC_HOST_F_CHUNKS, default 1)
    static const int f_chunks = getenv("QC_HOST_F_CHUNKS") ? std::max(1, atoi(getenv("QC_HOST_F_CHUNKS"))) : 1;
    int n_chunks = vals ? chunk_count(h, P.n_int, workers, (size_t)cp.comp_len * sizeof(double)) : std::max(1, std::min(f_chunks, P.n_int / 64));
    // Chunk boundaries.  QC_HOST_TAPER=t (default 0: equal chunks): sizes fall linearly from the first chunk to the last by the
    // factor t -- large chunks keep the link busy, and only the LAST chunk's replication is not overlapped with a transfer.
    static const double taper = getenv("QC_HOST_TAPER") ? std::max(1.0, atof(getenv("QC_HOST_TAPER"))) : 1.0;
    std::vector<int> bound(1, 0);
    if (taper <= 1.0 || n_chunks < 3) {
        const int per = (P.n_int + n_chunks - 1) / n_chunks;
        for (int b = per; b < P.n_int; b += per) bound.push_back(b);
    } else {
        double wsum = 0.0;
        for (int k = 0; k < n_chunks; ++k) wsum += taper - (taper - 1.0) * k / (n_chunks - 1);
        double acc = 0.0;
        for (int k = 0; k + 1 < n_chunks; ++k) {
            acc += taper - (taper - 1.0) * k / (n_chunks - 1);
            const int b = (int)(P.n_int * acc / wsum + 0.5);
            if (b > bound.back() && b < P.n_int) bound.push_back(b);
        }
    }
    bound.push_back(P.n_int);
    n_chunks = (int)bound.size() - 1;
    int rc;
    if ((rc = ensure_events(h, n_chunks))) return rc;
    for (int k = 0; k < n_chunks; ++k) {
        const int b0 = bound[k], b1 = bound[k + 1];
        hipError_t er = hipSuccess;
        if (!(rc = produce(k, b0, b1))) er = hipEventRecord(h->chunk_events[k], (two_streams && (k & 1)) ? h->stream2 : h->stream);
        if (rc || er != hipSuccess) {   // earlier chunks' kernels are still writing into the pinned blocks: let them finish before anyone reuses or frees those
            (void)hipStreamSynchronize(h->stream);
            if (two_streams) (void)hipStreamSynchronize(h->stream2);
            return rc ? rc : fail(&h->err, QC_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(er));
        }
    }
    HostGroup grp;
    int rc_wait = QC_OK;
    const double t_launched = now_us();
    double t_first = 0, t_last = 0;
    const double* comp = h->hJc;
    const double* hF = F ? h->hFc : nullptr;
    const QcParams* Pp = &P;
    for (int k = 0; k < n_chunks; ++k) {
        // (hipEventQuery in a loop: the calling thread has nothing else to do, and a blocking wait reacts 20 - 50 us late.  Not for
        //  the shards of a multi-device handle: N spinning shard threads next to the replication workers exceed a 16-CPU quota.
        //  QC_HOST_WAIT=1 restores hipEventSynchronize everywhere)
        static const bool blocking_env = getenv("QC_HOST_WAIT") && atoi(getenv("QC_HOST_WAIT")) == 1;
        const bool blocking_wait = blocking_env || shards > 1;
        hipError_t ew = hipSuccess;
        if (blocking_wait) ew = hipEventSynchronize(h->chunk_events[k]);
        else {
            unsigned spins = 0;
            while ((ew = hipEventQuery(h->chunk_events[k])) == hipErrorNotReady) {
                if ((++spins & qc_team::deadline_check_mask()) == 0 && now_us() - t_begin > qc_team::timeout_us()) return timed_out(h, "a chunk of the compact transfer");
                cpu_pause();
            }
        }
        if (ew != hipSuccess) { rc_wait = QC_ERR_HIP; break; }
        if (k == 0) t_first = now_us();
        if (k == n_chunks - 1) t_last = now_us();
        const int b0 = bound[k], b1 = bound[k + 1];
        // several jobs per chunk, so that the LAST chunk's replication (the un-overlapped tail) is shared by the workers
        static const int max_pieces = getenv("QC_HOST_PIECES") ? std::max(1, atoi(getenv("QC_HOST_PIECES"))) : 4;
        const int pieces = std::max(1, std::min(std::min(workers, max_pieces), (b1 - b0) / 4));
        const int step = (b1 - b0 + pieces - 1) / pieces;
        // (small outputs -- configs 1 and 2, short trajectories -- are expanded by the calling thread: waking a worker costs
        //  20 - 40 us, more than copying 100 KB)
        const size_t out_bytes = (size_t)(b1 - b0) * ((vals ? (size_t)Pp->jac_nnz : 0) + (hF ? (size_t)Pp->F_stride : 0)) * sizeof(double);
        if (out_bytes <= (128u << 10)) {
            if (vals) expand_intervals(*Pp, cp, comp, vals, b0, b1);
            if (hF) memcpy(F + (size_t)b0 * Pp->F_stride, hF + (size_t)b0 * Pp->F_stride, (size_t)(b1 - b0) * Pp->F_stride * sizeof(double));
            continue;
        }
        // (the LAST chunk's first piece is expanded by this thread, which has nothing left to wait for: a worker's wake-up
        //  alone costs 20 - 40 us)
        const bool last = k == n_chunks - 1;
        auto piece = [=](int c0, int c1) {
            if (vals) expand_intervals(*Pp, cp, comp, vals, c0, c1);
            if (hF) memcpy(F + (size_t)c0 * Pp->F_stride, hF + (size_t)c0 * Pp->F_stride, (size_t)(c1 - c0) * Pp->F_stride * sizeof(double));
        };
        for (int c0 = b0 + (last ? step : 0); c0 < b1; c0 += step) {
            const int c1 = std::min(b1, c0 + step);
            pool.push([=] { piece(c0, c1); }, &grp);
        }
        if (last) piece(b0, std::min(b1, b0 + step));
    }
    grp.wait();
    const double t_done = now_us();
    // Every chunk's event has completed, and the last two were the last operations of the two streams: both are idle.  (An explicit
    // hipStreamSynchronize on each cost 38 us per call for nothing.)  After a failed wait the streams are drained the slow way.
    if (rc_wait) {
        (void)hipStreamSynchronize(h->stream);
        if (two_streams) (void)hipStreamSynchronize(h->stream2);
    }
    if (host_trace())
        fprintf(stderr, "qcolloc host trace: %d chunks, %d workers: chunk loop entered +%.0f us after the call, launches issued +%.0f, first chunk "
                "landed +%.0f, last chunk landed +%.0f, replication done +%.0f, stream idle +%.0f us\n", n_chunks, workers, t_begin - g_call_begin,
                t_launched - g_call_begin, t_first - g_call_begin, t_last - g_call_begin, t_done - g_call_begin, now_us() - g_call_begin);
    if (rc_wait) return fail(&h->err, QC_ERR_HIP, "hipEventSynchronize failed in the compact transfer");
    return QC_OK;
}

// memcpy shared by the pool's workers (staging of the trajectory vector into pinned memory: 1.2 MB at config 3, 9.4 MB at config 4)
static void pool_memcpy(double* dst, const double* src, size_t n, int workers) {
    const size_t min_piece = 16384;   // doubles
    const int pieces = (int)std::max<size_t>(1, std::min<size_t>((size_t)workers, n / min_piece));
    if (pieces == 1) { memcpy(dst, src, n * sizeof(double)); return; }
    HostPool& pool = host_pool();
    HostGroup grp;
    const size_t step = (n + pieces - 1) / pieces;
    for (size_t o = step; o < n; o += step) {
        const size_t len = std::min(step, n - o);
        pool.push([=] { memcpy(dst + o, src + o, len * sizeof(double)); }, &grp);
    }
    memcpy(dst, src, std::min(step, n) * sizeof(double));   // the caller takes the first piece
    grp.wait();
}

// ------------------------------------------------------------------------------------------------
//  Pinned blocks handed to the caller (qc_host_alloc)
// ------------------------------------------------------------------------------------------------
namespace {
struct HostRange { char* base; size_t bytes; };
std::mutex g_reg_mu;
std::vector<HostRange> g_reg;
// [p, p + bytes) inside a block of qc_host_alloc: its device pointer on the CURRENT device; nullptr when it is ordinary memory
double* registered_device_ptr(const void* p, size_t bytes) {
    const char* q = static_cast<const char*>(p);
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (const HostRange& r : g_reg) found = found || (q >= r.base && q + bytes <= r.base + r.bytes);
    }
    if (!found) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<double*>(d);
}
}  // namespace

extern "C" int qc_host_alloc(int64_t bytes, void** out) {
    if (!out || bytes <= 0) return fail(nullptr, QC_ERR_INVALID, "qc_host_alloc: NULL output or empty size");
    *out = nullptr;
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, e == hipErrorNoDevice || e == hipErrorInvalidDevice ? QC_ERR_NO_DEVICE : QC_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_reg.push_back(HostRange{static_cast<char*>(p), (size_t)bytes});
    *out = p;
    return QC_OK;
}

extern "C" int qc_host_free(void* p) {
    if (!p) return QC_OK;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        size_t i = 0;
        for (; i < g_reg.size(); ++i) if (g_reg[i].base == static_cast<char*>(p)) break;
        if (i == g_reg.size()) return fail(nullptr, QC_ERR_INVALID, "qc_host_free: not a block of qc_host_alloc");
        g_reg.erase(g_reg.begin() + (long)i);
    }
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, QC_ERR_HIP, std::string("hipHostFree: ") + hipGetErrorString(e)); }
    return QC_OK;
}

extern "C" int qc_set_new_x(qc_handle* h, int new_x) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_set_new_x: NULL handle");
    h->new_x = new_x ? 1 : 0;
    for (qc_handle* sh : h->shards) sh->new_x = h->new_x;
    return QC_OK;
}

extern "C" int64_t qc_knot_generation(const qc_handle* h) {
    if (!h) return -1;
    if (h->shards.empty()) return (int64_t)h->z_gen;
    unsigned long long g = 0;   // every shard uploads its knots in every call that uploads at all (empty shards never do)
    for (const qc_handle* sh : h->shards) g = std::max(g, sh->z_gen);
    return (int64_t)g;
}

// ------------------------------------------------------------------------------------------------
//  One launch and one copy per host-buffer call, watched as it lands ("landing watch")
// ------------------------------------------------------------------------------------------------
// The host-buffer calls are bound by the PCIe link (config 3: 12.7 MB of residuals and compact Jacobian values, 14.7 MB of
// Hessian values per call).  What tests/hip/landing_probe.hip measured on the MI355X host (profiles/r03_landing_probe.txt):
//   * the copy engine moves 12.8 MB device -> host in 234 us (54.5 GB/s), into pinned AND into pageable memory (the runtime pins
//     the caller's pages in place; the call then blocks for the duration); 1.2 MB host -> device take 30 us either way;
//   * kernel stores into pinned host memory reach 44 - 47 GB/s whatever the grid, and they land in NO usable order: the L2
//     acknowledges a store long before it crosses the link and writes back in its own order -- of one launch over 999 intervals
//     the first complete interval is seen after 220 us of 440 (that variant was built first: 0.46 ms per call);
//   * round 2's sixteen chunk launches with an event each: 36 - 47 GB/s and 16 launch latencies.
// So: ONE kernel writes the call's compact output into HBM (3 - 9 us), ONE asynchronous copy brings it to a pinned block in
// address order at the link's rate, and the data is its own completion flag -- the pinned block holds a sentinel word (a
// signalling NaN that no arithmetic produces) wherever the copy has not arrived yet; a team of host threads (the calling thread
// and pool workers) claims pieces of the interval range as their first block is seen to have landed, waits block by block until
// no word of a block is the sentinel, replicates the block into the caller's array and re-arms it.  No events between
// chunks, no chunk launches.  Correctness does not rest on the sentinel being unique: the calling thread polls the copy's
// completion event, and once that has been seen every remaining word is taken as it is (a result that happens to equal the
// sentinel -- possible only if the caller's input carries that NaN payload -- costs the overlap, not the answer).
// Residual-only and Hessian calls need no host-side replication: their output goes device -> caller's array in one copy.
// Brings this handle's knots [t_begin, t_end] to the device unless qc_set_new_x(h, 0) says they are there already.  From where
// they lie: the runtime pins the caller's pages in place (30 us for config 3's 1.2 MB, the same as from pinned memory).
static int upload_knots(qc_handle* h, const double* Z) {
    if (h->new_x == 0 && h->z_valid) return QC_OK;
    const QcParams& P = h->prm;
    int rc;
    if ((rc = ensure(h, &h->dZ, (size_t)h->dims.Z_len))) return rc;
    const size_t o = (size_t)P.t_begin * P.zdim, n = (size_t)(P.n_int + 1) * P.zdim;
    QC_HIP(h, hipMemcpyAsync(h->dZ + o, Z + o, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->z_valid = true;
    ++h->z_gen;
    return QC_OK;
}

// waits for the handle's completion event without a blocking runtime call (those react 20 - 50 us late); the shards of a
// multi-device handle yield between polls (N spinning shard threads next to the team would exceed a small CPU quota)
static int wait_done(qc_handle* h, int shards) {
    hipError_t e;
    const double t0 = now_us();
    unsigned spins = 0;
    while ((e = hipEventQuery(h->ev_done)) == hipErrorNotReady) {
        if ((++spins & qc_team::deadline_check_mask()) == 0 && now_us() - t0 > qc_team::timeout_us()) return timed_out(h, "the evaluation");
        if (shards > 1) sched_yield(); else cpu_pause();
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        return fail(&h->err, QC_ERR_HIP, std::string("the evaluation failed on the device: ") + hipGetErrorString(e));
    }
    return QC_OK;
}

// Watches the copy into J.src land (recorded as h->ev_done on h->stream by the caller) and replicates it into the caller's arrays.
static int land_run(qc_handle* h, LandJob& J, int shards, double t_begin, double t_issued) {
    HostPool& pool = host_pool();
    const int workers = pool_workers(shards);
    pool.ensure(workers, h->device, device_l3_domains);
    // helpers: enough to keep up with the link (the replication of config 3 writes 41.5 MB per call), never more than pieces;
    // small outputs are replicated by the calling thread alone (a worker's wake-up costs 20 - 40 us)
    const size_t out_bytes = (size_t)J.n_int * ((J.dst_stride ? J.dst_stride : (size_t)J.lay.jac_nnz) + (J.F ? J.f_len : 0)) * sizeof(double);
    int helpers = std::max(0, workers / std::max(1, shards) - (shards > 1 ? 1 : 0));
    if (out_bytes <= (256u << 10)) helpers = 0;
    J.t_begin = t_begin;
    const hipEvent_t ev = h->ev_done;
    const int done = qc_team::land_team(J, pool, helpers, [ev]() -> int {
        const hipError_t e = hipEventQuery(ev);
        return e == hipSuccess ? qc_team::LAND_DONE : (e == hipErrorNotReady ? qc_team::LAND_PENDING : qc_team::LAND_FAILED);
    });
    const double t_consumed = now_us();
    const int np = (int)J.bound.size() - 1;
    int rc = QC_OK;
    if (done == qc_team::LAND_TIMEOUT) return timed_out(h, "the copy of the compact values");
    if (done != qc_team::LAND_DONE && done != qc_team::LAND_FAILED) rc = wait_done(h, shards);      // (every piece consumed before the completion was seen)
    if (host_trace())
        fprintf(stderr, "qcolloc host trace (%d pieces, %d helpers): inputs, launch and copy issued +%.0f us, first piece done +%.0f, last piece done +%.0f, "
                "team done +%.0f us; copy seen complete +%.0f with %d pieces done; %d of %d blocks were waited for\n", np, std::min(helpers, np - 1), t_issued - t_begin,
                J.t_first_piece.load() ? J.t_first_piece.load() - t_begin : 0.0, J.t_last_piece.load() ? J.t_last_piece.load() - t_begin : 0.0,
                t_consumed - t_begin, J.t_event.load() ? J.t_event.load() - t_begin : -1.0, J.pieces_at_event.load(), J.blocks_waited.load(), J.n_int);
    if (host_trace()) {
        fprintf(stderr, "    members (cpu:pieces):");
        for (int i = 0; i < std::min(64, J.n_members.load()); ++i) fprintf(stderr, " %d:%d", J.member_cpu[i], J.member_pieces[i]);
        fprintf(stderr, "\n");
    }
    if (!rc && done == qc_team::LAND_FAILED) { (void)hipStreamSynchronize(h->stream); rc = fail(&h->err, QC_ERR_HIP, "the evaluation failed on the device"); }
    return rc;
}

struct qc_rearm { qc_team::Rearm r; };
void qc_rearm_destroy(qc_rearm* r) {
    if (!r) return;
    r->r.grp.wait();
    delete r;
}

static int eval_host_chunked(qc_handle* h, const double* Z, double* F, double* vals, int shards);

// dC / hC[] change hands between layouts (the handle's own host-buffer calls, a list it leads, another list): taken afresh
static int claim_staging(qc_handle* h, unsigned long long tag, size_t cap) {
    if (h->stage_tag == tag && h->stage_cap == cap) return QC_OK;
    for (int i = 0; i < QC_HOST_RING; ++i) {
        if (h->rearm[i]) h->rearm[i]->r.grp.wait();
        if (h->hC[i]) { (void)hipHostFree(h->hC[i]); h->hC[i] = nullptr; }
        h->hC_armed[i] = false;
    }
    if (h->dC) { QC_HIP(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->dC); h->dC = nullptr; }
    h->stage_tag = tag;
    h->stage_cap = cap;
    return QC_OK;
}


// The ring of pinned blocks: every block holds the watched output of this handle (residual rows + compact Jacobian values).
// ring_take hands out the next block, armed (its re-arm jobs of the previous turn waited for).
static size_t ring_capacity(const qc_handle* h) {
    const QcParams& P = h->prm;
    const CompactPlan cp = compact_plan(P);
    const size_t jac = (size_t)P.F_stride + (size_t)(cp.useful ? cp.comp_len : P.jac_nnz);
    return (size_t)P.n_int * jac;
}
static int ring_take(qc_handle* h, int* index, size_t cap = 0) {
    static const int ring = std::max(2, std::min(QC_HOST_RING, getenv("QC_HOST_NBUF") ? atoi(getenv("QC_HOST_NBUF")) : 3));
    const int ib = h->hC_next;
    h->hC_next = (ib + 1) % ring;
    if (!cap) cap = ring_capacity(h);
    int rc;
    if ((rc = ensure_pinned(h, &h->hC[ib], cap, false))) return rc;
    if (h->rearm[ib]) h->rearm[ib]->r.grp.wait();             // the re-arm jobs of this block's previous use (normally long done)
    if (!h->hC_armed[ib]) {
        // (non-temporal stores: fenced before the copy engine may write the block, or a write-combining buffer draining late
        //  would put the sentinel over words that have already landed)
        qc_host_fill(h->hC[ib], cap, kLandSentinel);
        qc_host_copy_fence();
        h->hC_armed[ib] = true;
    }
    *index = ib;
    return QC_OK;
}
// the consumed part of a block is re-armed behind the caller's back by whichever workers are idle (qc_host_team.h)
static void ring_rearm_later(qc_handle* h, int ib, size_t used) {
    if (!h->rearm[ib]) h->rearm[ib] = new qc_rearm();
    qc_team::rearm_later(host_pool(), h->rearm[ib]->r, h->hC[ib], used);
}

extern "C" int qc_debug_host_expand_rate(qc_handle* h, int32_t reps, double* GBps) {
    if (!h || !GBps || reps < 1) return fail(h ? &h->err : nullptr, QC_ERR_INVALID, "qc_debug_host_expand_rate: bad argument");
    qc_handle* s = is_multi(h) ? h->shards[0] : h;
    const QcParams& P = s->prm;
    const CompactPlan cp = compact_plan(P);
    if (!cp.useful || P.n_int == 0) return fail(&h->err, QC_ERR_UNSUPPORTED, "qc_debug_host_expand_rate: no replicated blocks in this handle's Jacobian");
    const int n_int = P.n_int;
    std::vector<double> comp((size_t)n_int * cp.comp_len, 1.0), vals((size_t)n_int * P.jac_nnz);
    HostPool& pool = host_pool();
    const int workers = pool_workers(1);
    pool.ensure(workers);
    const int per = land_piece_intervals((size_t)cp.comp_len * sizeof(double));
    double best = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        std::atomic<int> next{0};
        auto work = [&] {
            for (;;) {
                const int b0 = next.fetch_add(per);
                if (b0 >= n_int) return;
                expand_intervals(P, cp, comp.data(), vals.data(), b0, std::min(n_int, b0 + per));
            }
        };
        HostGroup grp;
        const double t0 = now_us();
        pool.push_many(work, workers, &grp);
        work();
        grp.wait();
        const double dt = now_us() - t0;
        if (r > 0) best = std::max(best, (double)n_int * P.jac_nnz * sizeof(double) / dt / 1e3);   // (the first pass touches fresh pages)
    }
    *GBps = best;
    return QC_OK;
}

static int eval_host(qc_handle* h, const double* Z, double* F, double* vals, int shards) {
    if (shares_values(h)) return fail(&h->err, QC_ERR_UNSUPPORTED, "composed handles write into shared vectors: use the _dev entry points");
    if (!Z || (!F && !vals)) return fail(&h->err, QC_ERR_INVALID, "qc_eval: NULL buffer");
    const QcParams& P = h->prm;
    if (P.n_int == 0) return QC_OK;
    const CompactPlan cp = compact_plan(P);
    const bool compact = vals && h->host_compact && cp.useful;
    const bool direct = (compact || !vals) && h->host_compact == 1 && h->kernel == QC_KERNEL_MFMA && qc_mfma_compact_supported(P);
    if (vals && (!direct || h->host_landing != 1)) return eval_host_chunked(h, Z, F, vals, shards);
    const double t_begin = now_us();
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    int rc;
    if ((rc = drain_if_needed(h)) || (rc = drain_leader_if_needed(h))) return rc;
    if ((rc = upload_knots(h, Z))) return rc;
    if (!vals) {
        // residuals only (a line-search trial): kernel -> HBM -> one copy into the caller's array.  (Rows no kernel writes --
        // QC_ROWS_BY_COMPONENT -- stay zero in the device vector.)
        // Memory of qc_host_alloc takes the residuals straight from the kernel: no device-to-host copy, no pinning
        // (QC_HOST_F_DIRECT=0: the copy, for A/B runs).  Not where the layout has rows no kernel writes: the copy delivers their zeros.
        static const bool f_direct = !(getenv("QC_HOST_F_DIRECT") && atoi(getenv("QC_HOST_F_DIRECT")) == 0);
        const bool rows_all_written = P.F_stride == P.ddim && P.F_off == 0;
        if (double* Fdev = (f_direct && rows_all_written) ? registered_device_ptr(F, (size_t)h->dims.F_len * sizeof(double)) : nullptr) {
            if ((rc = qc_eval_F_jac_dev(h, h->dZ, Fdev, nullptr, h->stream))) return rc;
            QC_HIP(h, hipEventRecord(h->ev_done, h->stream));
            rc = wait_done(h, shards);
            if (host_trace()) fprintf(stderr, "qcolloc host trace (residuals only, written in place): done +%.0f us\n", now_us() - t_begin);
            return rc;
        }
        if ((rc = claim_plain(h, 1)) || (rc = ensure_zeroed(h, &h->dF, (size_t)h->dims.F_len))) return rc;
        if ((rc = qc_eval_F_jac_dev(h, h->dZ, h->dF, nullptr, h->stream))) return rc;
        QC_HIP(h, hipMemcpyAsync(F, h->dF, (size_t)h->dims.F_len * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        QC_HIP(h, hipEventRecord(h->ev_done, h->stream));
        rc = wait_done(h, shards);
        if (host_trace()) fprintf(stderr, "qcolloc host trace (residuals only): done +%.0f us\n", now_us() - t_begin);
        return rc;
    }
    // residuals + compact Jacobian values, interleaved per interval in ONE device block, so that one copy brings both and a
    // team member finds everything an interval needs in one place
    LandJob J;
    J.lay.jac_nnz = P.jac_nnz; J.lay.jo_F = P.jo_F; J.lay.jo_B = P.jo_B; J.lay.n2 = cp.n2; J.lay.copies = cp.copies;
    J.lay.second_copies = cp.second_copies; J.lay.head2 = cp.head2; J.lay.tail_src = cp.tail_src; J.lay.tail_len = cp.tail_len;
    J.n_int = P.n_int;
    // (a layout with rows no kernel writes keeps its residual rows in the block even when only the values are asked for: the block
    //  layout of such a handle must not change between calls, or stale values would show through the never-written rows)
    const bool every_row_written = P.F_stride == P.ddim && P.F_off == 0;
    const bool with_F = F != nullptr || !every_row_written;
    J.f_len = with_F ? (size_t)P.F_stride : 0;
    J.blk = J.f_len + (size_t)cp.comp_len;
    J.vals = vals;
    J.F = F;
    const size_t cap = (size_t)P.n_int * ((size_t)P.F_stride + (size_t)cp.comp_len);
    if ((rc = claim_staging(h, 1, cap))) return rc;           // (the blocks may have been a list's: ADVICE round 4)
    if ((rc = ensure_zeroed(h, &h->dC, cap))) return rc;      // (zeroed once per layout: residual rows no kernel writes are delivered as 0)
    int ib;
    if ((rc = ring_take(h, &ib))) return rc;
    J.src = h->hC[ib];
    J.rearm_inline = qc_team::land_inline_rearm();
    QcParams C = compact_params(P, cp);
    C.J_stride = (long long)J.blk;
    C.J_off = (long long)J.f_len;
    C.F_stride = (long long)J.blk;
    const hipError_t e = qc_launch_mfma_F_jac(C, h->dZ, with_F ? h->dC : nullptr, h->dC, h->stream);
    if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    const size_t used = (size_t)P.n_int * J.blk;
    hipError_t ec = hipMemcpyAsync(h->hC[ib], h->dC, used * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (ec == hipSuccess) ec = hipEventRecord(h->ev_done, h->stream);
    if (ec != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        h->hC_armed[ib] = false;
        return fail(&h->err, QC_ERR_HIP, std::string("copy of the compact values: ") + hipGetErrorString(ec));
    }
    rc = land_run(h, J, shards, t_begin, now_us());
    if (rc) { h->hC_armed[ib] = false; return rc; }
    if (!J.rearm_inline) ring_rearm_later(h, ib, used);
    return QC_OK;
}

// The chunked host path of round 2 (one launch per chunk of intervals, events between them) -- what kernels that cannot write
// the compact form themselves still use, and the A/B reference of the one-launch path (QC_HOST_LANDING=0).
static int eval_host_chunked(qc_handle* h, const double* Z, double* F, double* vals, int shards) {
    const QcParams& P = h->prm;
    if (host_trace()) g_call_begin = now_us();
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    int rc;
    if ((rc = drain_if_needed(h)) || (rc = drain_leader_if_needed(h)) || (rc = claim_plain(h, 1))) return rc;
    // Only the knots this handle touches cross PCIe: [t_begin, t_end] inclusive, through a pinned staging buffer (an
    // asynchronous copy from pageable memory is staged by the runtime in small pieces and blocks the calling thread).
    const size_t z0 = (size_t)P.t_begin * P.zdim;
    const bool z_skip = h->new_x == 0 && h->z_valid;
    if ((rc = ensure_pinned(h, &h->hZ, (size_t)h->dims.Z_len, false))) return rc;
    if ((rc = ensure(h, &h->dZ, (size_t)h->dims.Z_len))) return rc;
    HostPool& pool = host_pool();
    const int workers = pool_workers(shards);
    pool.ensure(workers);
    // knots [k0, k1) of this handle's range: pageable -> pinned (worker threads) -> device (asynchronous on the handle's stream)
    auto stage_knots = [&](size_t k0, size_t k1) -> int {
        if (z_skip) return QC_OK;
        const size_t o = z0 + k0 * P.zdim, n = (k1 - k0) * P.zdim;
        pool_memcpy(h->hZ + o, Z + o, n, workers);
        QC_HIP(h, hipMemcpyAsync(h->dZ + o, h->hZ + o, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        return QC_OK;
    };

    const CompactPlan cp = compact_plan(P);
    const bool compact = vals && h->host_compact && cp.useful;
    const bool direct = (compact || !vals) && h->host_compact == 1 && h->kernel == QC_KERNEL_MFMA && qc_mfma_compact_supported(P);
    if (direct) {
        // Direct form: the kernel writes residuals and compact values straight into pinned host memory, one launch per chunk of
        // intervals; the first chunk's knots are staged and copied ahead of the rest, so that the first launch (and with it the
        // PCIe write stream) starts after ~75 KB instead of the whole trajectory vector.  (Reading the knots from the pinned
        // buffer inside the kernel -- no H2D copy at all -- was measured slower: every chunk launch then begins with a PCIe
        // round trip, 23 instead of 16 us per chunk at config 3.)
        if (vals && (rc = ensure_pinned(h, &h->hJc, (size_t)P.n_int * cp.comp_len, false))) return rc;
        if (F && (rc = ensure_pinned(h, &h->hFc, (size_t)h->dims.F_len, true))) return rc;
        const QcParams C = vals ? compact_params(P, cp) : P;
        static const bool two = !(getenv("QC_HOST_STREAMS") && atoi(getenv("QC_HOST_STREAMS")) == 1);
        rc = run_chunks(h, cp, F, vals, shards, [&](int k, int b0, int b1) -> int {
            int rs = QC_OK;
            if (k == 0) rs = stage_knots(0, (size_t)b1 + 1);                                   // chunk 0 and its halo knot
            else if (k == 1) {
                rs = stage_knots((size_t)b0 + 1, (size_t)P.n_int + 1);                         // everything else
                if (!rs && two) {   // the second stream starts behind the copies
                    QC_HIP(h, hipEventRecord(h->ev_staged, h->stream));
                    QC_HIP(h, hipStreamWaitEvent(h->stream2, h->ev_staged, 0));
                }
            }
            if (rs) return rs;
            QcParams Ck = C;
            Ck.t_begin = C.t_begin + b0;
            Ck.n_int = b1 - b0;
            hipError_t e = qc_launch_mfma_F_jac(Ck, h->dZ, F ? h->hFc + (size_t)b0 * C.F_stride : nullptr,
                                                vals ? h->hJc + (size_t)b0 * cp.comp_len : nullptr, (two && (k & 1)) ? h->stream2 : h->stream);
            if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            return QC_OK;
        }, two);
        if (!z_skip) { h->z_valid = rc == QC_OK; ++h->z_gen; }
        return rc;
    }
    if ((rc = stage_knots(0, (size_t)P.n_int + 1))) return rc;
    if (!z_skip) { h->z_valid = true; ++h->z_gen; }
    if (F && (rc = ensure_zeroed(h, &h->dF, (size_t)h->dims.F_len))) return rc;
    if (vals && (rc = ensure(h, &h->dJ, (size_t)h->dims.jac_nnz))) return rc;
    if ((rc = qc_eval_F_jac_dev(h, h->dZ, F ? h->dF : nullptr, vals ? h->dJ : nullptr, h->stream))) return rc;
    if (F && h->dims.F_len) QC_HIP(h, hipMemcpyAsync(F, h->dF, (size_t)h->dims.F_len * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (compact) {
        // gather kernels write the compact form into the pinned host buffer (device-visible), one launch per chunk: a shader
        // copy moves ~50 GB/s over PCIe here, hipMemcpyAsync into pinned memory (SDMA engine) only 27 GB/s
        if ((rc = ensure_pinned(h, &h->hJc, (size_t)P.n_int * cp.comp_len, false))) return rc;
        return run_chunks(h, cp, nullptr, vals, shards, [&](int, int b0, int b1) -> int {
            hipError_t e = qc_launch_pack_jac(h->dJ + (size_t)b0 * P.jac_nnz, h->hJc + (size_t)b0 * cp.comp_len, b1 - b0, P.jac_nnz, cp.comp_len,
                                              cp.n2, P.jo_F, P.jo_B, cp.head2, cp.tail_src, h->stream);
            if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            return QC_OK;
        });
    }
    if (vals && h->dims.jac_nnz)
        QC_HIP(h, hipMemcpyAsync(vals, h->dJ, (size_t)h->dims.jac_nnz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    QC_HIP(h, hipStreamSynchronize(h->stream));
    return QC_OK;
}

static int hess_host(qc_handle* h, const double* Z, const double* mu, double* hvals, int shards) {
    if (h->prm.hess_nnz == 0) return QC_OK;   // no drives and a fixed timestep: the constraint is linear
    if (!Z || !mu || !hvals) return fail(&h->err, QC_ERR_INVALID, "qc_eval_hess: NULL buffer");
    if (shares_values(h)) return fail(&h->err, QC_ERR_UNSUPPORTED, "composed handles write into shared vectors: use the _dev entry points");
    const QcParams& P = h->prm;
    if (P.n_int == 0) return QC_OK;
    const double t_begin = now_us();
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    int rc;
    if ((rc = drain_if_needed(h)) || (rc = drain_leader_if_needed(h)) || (rc = claim_plain(h, 1))) return rc;
    if ((rc = ensure(h, &h->dMu, (size_t)h->dims.n_rows))) return rc;
    if ((rc = ensure(h, &h->dH, (size_t)h->dims.hess_nnz))) return rc;
    if ((rc = upload_knots(h, Z))) return rc;
    const size_t m0 = (size_t)P.t_begin * P.F_stride, mn = (size_t)P.n_int * P.F_stride;
    // The values have no replicated blocks: kernel -> HBM, then the copy engine straight into the caller's array (14.7 MB at
    // config 3: 0.27 ms at the link's 54.5 GB/s; the runtime pins the caller's pages in place, and the call blocks meanwhile).
    // QC_HOST_HESS_CHUNKS > 1 sends the multipliers up and the values down in pieces on two streams; measured at config 3
    // (profiles/r03_host_path.txt): 1 piece 0.344 ms, 2 pieces 0.405, 4 pieces 0.46 - 0.48, 8 pieces 0.66 -- a copy into pageable
    // memory returns when it is done, so the pieces do not overlap and each pays its own set-up.  Default 1.
    // Also measured and not kept (profiles/r03_host_hess_watched.txt): the values through a pinned ring block with the team copying
    // them out behind the copy engine as for the Jacobian, the first quarter's kernel and download started while the rest of the
    // multipliers go up on the second stream -- bit-identical, 0.357 (one part) and 0.378 - 0.388 ms (two parts) against 0.346 for
    // this path: an upload running beside the download slows it (49 against 54.5 GB/s) and the team finishes 20 us after the link.
    static const int want = getenv("QC_HOST_HESS_CHUNKS") ? std::max(1, atoi(getenv("QC_HOST_HESS_CHUNKS"))) : 1;
    const bool chunkable = h->kernel == QC_KERNEL_MFMA && qc_mfma_hess_supported(P) && !qc_mfma64_hess_supported(P) && !qc_mfma16_padeP_hess_supported(P) &&
                           h->host_compact != 0;
    const int n_chunks = chunkable ? std::max(1, std::min(want, P.n_int / 64)) : 1;
    if (n_chunks == 1) {
        QC_HIP(h, hipMemcpyAsync(h->dMu + m0, mu + m0, mn * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if ((rc = qc_eval_hess_dev(h, h->dZ, h->dMu, h->dH, h->stream))) return rc;
        QC_HIP(h, hipMemcpyAsync(hvals, h->dH, (size_t)h->dims.hess_nnz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        QC_HIP(h, hipEventRecord(h->ev_done, h->stream));
        rc = wait_done(h, shards);
    } else {
        if ((rc = ensure_events(h, n_chunks))) return rc;
        // stream2: the multipliers, piece by piece; stream: kernel and download of piece k behind piece k's upload
        QC_HIP(h, hipEventRecord(h->ev_staged, h->stream));              // (the knots, and whatever the stream did before)
        QC_HIP(h, hipStreamWaitEvent(h->stream2, h->ev_staged, 0));
        const int per = (P.n_int + n_chunks - 1) / n_chunks;
        for (int k = 0; k < n_chunks; ++k) {
            const int b0 = k * per, b1 = std::min(P.n_int, b0 + per);
            if (b0 >= b1) break;
            const size_t mo = m0 + (size_t)b0 * P.F_stride, ml = (size_t)(b1 - b0) * P.F_stride;
            QC_HIP(h, hipMemcpyAsync(h->dMu + mo, mu + mo, ml * sizeof(double), hipMemcpyHostToDevice, h->stream2));
            QC_HIP(h, hipEventRecord(h->chunk_events[k], h->stream2));
        }
        hipError_t e = hipSuccess;
        for (int k = 0; k < n_chunks && e == hipSuccess; ++k) {
            const int b0 = k * per, b1 = std::min(P.n_int, b0 + per);
            if (b0 >= b1) break;
            QcParams Ck = P;
            Ck.t_begin = P.t_begin + b0;
            Ck.n_int = b1 - b0;
            e = hipStreamWaitEvent(h->stream, h->chunk_events[k], 0);
            if (e == hipSuccess) e = qc_launch_mfma_hess(Ck, h->dZ, h->dMu, h->dH + (size_t)b0 * P.H_stride, h->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(hvals + (size_t)b0 * P.H_stride, h->dH + (size_t)b0 * P.H_stride, (size_t)(b1 - b0) * P.H_stride * sizeof(double),
                                   hipMemcpyDeviceToHost, h->stream);
        }
        if (e == hipSuccess) e = hipEventRecord(h->ev_done, h->stream);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(h->stream2);
            (void)hipStreamSynchronize(h->stream);
            return fail(&h->err, QC_ERR_HIP, std::string("qc_eval_hess: ") + hipGetErrorString(e));
        }
        rc = wait_done(h, shards);
    }
    if (host_trace()) fprintf(stderr, "qcolloc host trace (Hessian values, %d piece(s)): done +%.0f us\n", n_chunks, now_us() - t_begin);
    return rc;
}

// ------------------------------------------------------------------------------------------------
//  Multi-device handle: one worker thread per shard, each driving its own device / stream / pinned staging
// ------------------------------------------------------------------------------------------------
struct qc_fanout {
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> task;
        bool has = false, done = true, stop = false;
        int rc = 0;
        void loop() {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [&] { return stop || has; });
                if (stop) return;
                has = false;
                lk.unlock();
                const int r = task();
                lk.lock();
                rc = r;
                done = true;
                cv.notify_all();
            }
        }
    };
    std::vector<std::unique_ptr<Worker>> w;
    explicit qc_fanout(int n) {
        for (int i = 0; i < n; ++i) {
            w.emplace_back(new Worker());
            Worker* p = w.back().get();
            p->th = std::thread([p] { p->loop(); });
        }
    }
    // runs task(i) for every shard concurrently; returns the first non-zero status (by shard order)
    int run(const std::function<int(int)>& task) {
        for (size_t i = 0; i < w.size(); ++i) {
            Worker* p = w[i].get();
            std::lock_guard<std::mutex> lk(p->mu);
            p->task = [&task, i] { return task((int)i); };
            p->has = true;
            p->done = false;
            p->cv.notify_all();
        }
        int rc = 0;
        for (auto& up : w) {
            Worker* p = up.get();
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->done; });
            if (!rc && p->rc) rc = p->rc;
        }
        return rc;
    }
    ~qc_fanout() {
        for (auto& up : w) {
            { std::lock_guard<std::mutex> lk(up->mu); up->stop = true; }
            up->cv.notify_all();
            up->th.join();
        }
    }
};
void qc_fanout_destroy(qc_fanout* f) { delete f; }

static int multi_run(qc_handle* h, const std::function<int(int)>& task) {
    const int n = (int)h->shards.size();
    int rc;
    if (n == 1) rc = task(0);
    else rc = h->fan->run(task);
    if (rc) {
        for (qc_handle* sh : h->shards)
            if (!sh->err.empty()) { h->err = sh->err; break; }
        return fail(&h->err, rc, h->err.empty() ? std::string("a shard reported an error") : h->err);
    }
    return QC_OK;
}

extern "C" int qc_create_multi(const qc_desc* d, int32_t n_shards, const int32_t* device_ids, qc_handle** out) {
    if (!out) return fail(nullptr, QC_ERR_INVALID, "qc_create_multi: out is NULL");
    *out = nullptr;
    if (n_shards < 1 || n_shards > 1024 || !device_ids) return fail(nullptr, QC_ERR_INVALID, "qc_create_multi: n_shards must be in 1..1024 with a device list");
    QcParams P; qc_dims_t dims; std::string err;
    int rc = qc_build_params(d, &P, &dims, &err);
    if (rc) return rc;
    // Composed descriptors (one handle per state integrator of a sampling / direct-sum problem, qc_desc.rows_per_interval ...) are
    // sharded like any other: every member of the list is created with the SAME device list, so that shard s of every member covers
    // the same intervals on the same device, and the "_list" entry points evaluate shard by shard (list_eval_multi below).
    qc_handle* h = new qc_handle();
    h->desc = *d;
    h->desc.G_drift = nullptr;
    h->desc.G_drives = nullptr;
    h->device = device_ids[0];
    h->prm = P;
    h->dims = dims;
    const long long len = P.n_int, tb = P.t_begin;
    const long long chunk = (len + n_shards - 1) / n_shards;
    h->shard_chunk = chunk;
    for (int i = 0; i < n_shards; ++i) {
        qc_desc di = *d;
        di.device = device_ids[i];
        long long b0 = std::min(len, (long long)i * chunk), b1 = std::min(len, (long long)(i + 1) * chunk);
        di.t_begin = tb + b0;
        di.t_end = tb + b1;
        if (di.t_begin == 0 && di.t_end == 0) { di.t_begin = di.t_end = 1; }   // an empty first shard of a 1-interval problem cannot happen (chunk >= 1), kept for safety
        qc_handle* sh = nullptr;
        rc = qc_create(&di, &sh);
        if (rc) {
            const std::string msg = std::string("qc_create_multi: shard ") + std::to_string(i) + ": " + qc_last_error(nullptr);
            qc_destroy(h);
            return fail(nullptr, rc, msg);
        }
        h->shards.push_back(sh);
        h->shard_F_off.push_back(b0 * P.F_stride);
        h->shard_J_off.push_back(b0 * P.J_stride);
        h->shard_H_off.push_back(b0 * P.H_stride);
    }
    h->kernel = h->shards[0]->kernel;
    h->dims.kernel = h->kernel;
    {
        // Which way do the Jacobian values come home?  The compact form (one copy of the replicated blocks per link, 3.5x fewer
        // bytes at config 3) leaves the replication to ONE host: N x 41.5 MB per evaluation at the 170 - 235 GB/s its threads reach.
        // With the links working in parallel the plain full copy -- 41.5 MB per link at 54.5 GB/s, nothing for the host to do --
        // overtakes that from about four devices on (0.76 ms per evaluation at any N, against 1.4 - 2 ms of replication at N = 8).
        // QC_HOST_MULTI_FULL = the number of distinct devices from which the shards copy in full (default 4; 0 = never).
        std::vector<int> distinct;
        for (int i = 0; i < n_shards; ++i)
            if (std::find(distinct.begin(), distinct.end(), device_ids[i]) == distinct.end()) distinct.push_back(device_ids[i]);
        const int from = getenv("QC_HOST_MULTI_FULL") ? atoi(getenv("QC_HOST_MULTI_FULL")) : 4;
        if (from > 0 && (int)distinct.size() >= from && !getenv("QC_HOST_COMPACT"))
            for (qc_handle* sh : h->shards) sh->host_compact = 0;
    }
    if (n_shards > 1) h->fan = new qc_fanout(n_shards);
    *out = h;
    return QC_OK;
}

extern "C" int32_t qc_multi_count(const qc_handle* h) { return h ? (int32_t)h->shards.size() : 0; }

extern "C" qc_handle* qc_multi_shard(qc_handle* h, int32_t i) {
    if (!h || i < 0 || i >= (int32_t)h->shards.size()) return nullptr;
    return h->shards[i];
}

extern "C" int qc_multi_shard_info(const qc_handle* h, int32_t i, int32_t* device, int64_t* t_begin, int64_t* t_end) {
    if (!h || i < 0 || i >= (int32_t)h->shards.size()) return fail(nullptr, QC_ERR_INVALID, "qc_multi_shard_info: not a multi-device handle or shard out of range");
    const qc_handle* sh = h->shards[i];
    if (device) *device = sh->device;
    if (t_begin) *t_begin = sh->prm.t_begin;
    if (t_end) *t_end = sh->prm.t_begin + sh->prm.n_int;
    return QC_OK;
}

extern "C" int64_t qc_multi_padded_len(const qc_handle* h, int64_t per_interval) {
    if (!h || h->shards.empty() || per_interval < 0) return 0;
    return (int64_t)h->shard_chunk * (int64_t)h->shards.size() * per_interval;
}

extern "C" int qc_multi_eval_F_jac_dev(qc_handle* h, const double* const* dZ, double* const* dF, double* const* dvals) {
    if (!h || h->shards.empty()) return fail(h ? &h->err : nullptr, QC_ERR_INVALID, "qc_multi_eval_F_jac_dev: not a multi-device handle");
    if (!dZ || (!dF && !dvals)) return fail(&h->err, QC_ERR_INVALID, "qc_multi_eval_F_jac_dev: NULL pointer list");
    for (size_t i = 0; i < h->shards.size(); ++i) {   // launches are asynchronous: no threads needed
        qc_handle* sh = h->shards[i];
        const int rc = qc_eval_F_jac_dev(sh, dZ[i], dF && dF[i] ? dF[i] + h->shard_F_off[i] : nullptr,
                                         dvals && dvals[i] ? dvals[i] + h->shard_J_off[i] : nullptr, sh->stream);
        if (rc) return fail(&h->err, rc, sh->err);
    }
    return QC_OK;
}

extern "C" int qc_multi_eval_hess_dev(qc_handle* h, const double* const* dZ, const double* const* dmu, double* const* dhvals) {
    if (!h || h->shards.empty()) return fail(h ? &h->err : nullptr, QC_ERR_INVALID, "qc_multi_eval_hess_dev: not a multi-device handle");
    if (!dZ || !dmu || !dhvals) return fail(&h->err, QC_ERR_INVALID, "qc_multi_eval_hess_dev: NULL pointer list");
    for (size_t i = 0; i < h->shards.size(); ++i) {
        qc_handle* sh = h->shards[i];
        const int rc = qc_eval_hess_dev(sh, dZ[i], dmu[i], dhvals[i] ? dhvals[i] + h->shard_H_off[i] : nullptr, sh->stream);
        if (rc) return fail(&h->err, rc, sh->err);
    }
    return QC_OK;
}

extern "C" int qc_multi_sync(qc_handle* h) {
    if (!h || h->shards.empty()) return fail(h ? &h->err : nullptr, QC_ERR_INVALID, "qc_multi_sync: not a multi-device handle");
    for (qc_handle* sh : h->shards) {
        qc_device_guard guard(sh->device);
        QC_HIP(h, guard.err);
        QC_HIP(h, hipStreamSynchronize(sh->stream));
    }
    return QC_OK;
}

// ---- RCCL all-gather of the shard slices (loaded on first use; the library itself links only libamdhip64) ----------
namespace {
typedef struct ncclComm* ncclComm_t;
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // A copy that is already in the process wins (torch bundles its own librccl under its own directory: a bare-name
        // RTLD_NOLOAD probe does not find that one, and a second RCCL runtime next to it must not happen): walk the loaded objects
        // and take the first whose file name starts with "librccl", by its full path.
        std::string loaded;
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* out) -> int {
            const char* path = info->dlpi_name;
            if (!path || !*path) return 0;
            const char* base = strrchr(path, '/');
            base = base ? base + 1 : path;
            if (strncmp(base, "librccl.so", 10) != 0) return 0;   // (not librccl-net.so and friends)
            *static_cast<std::string*>(out) = path;
            return 1;
        }, &loaded);
        if (!loaded.empty()) api.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (api.lib && !dlsym(api.lib, "ncclAllGather")) api.lib = nullptr;   // whatever that object was, it is not the collective library
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd && api.GetErrorString;
    });
    return api;
}
constexpr int kNcclFloat64 = 8;   // ncclDataType_t ncclFloat64 (rccl.h)
}  // namespace

struct qc_rccl_state {
    std::vector<ncclComm_t> comms;
};
void qc_rccl_destroy(qc_rccl_state* r) {
    if (!r) return;
    RcclApi& api = rccl_api();
    if (api.ok) for (ncclComm_t c : r->comms) if (c) (void)api.CommDestroy(c);
    delete r;
}

extern "C" int qc_multi_all_gather_dev(qc_handle* h, double* const* bufs, int64_t per_interval) {
    if (!h || h->shards.empty()) return fail(h ? &h->err : nullptr, QC_ERR_INVALID, "qc_multi_all_gather_dev: not a multi-device handle");
    if (!bufs || per_interval <= 0) return fail(&h->err, QC_ERR_INVALID, "qc_multi_all_gather_dev: bad argument");
    const int n = (int)h->shards.size();
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        devs[i] = h->shards[i]->device;
        if (!bufs[i]) return fail(&h->err, QC_ERR_INVALID, "qc_multi_all_gather_dev: NULL buffer");
        for (int k = 0; k < i; ++k)
            if (devs[k] == devs[i]) return fail(&h->err, QC_ERR_UNSUPPORTED, "qc_multi_all_gather_dev: one RCCL rank per device: the shards must be on distinct devices");
    }
    RcclApi& api = rccl_api();
    if (!api.ok) return fail(&h->err, QC_ERR_UNSUPPORTED, "qc_multi_all_gather_dev: librccl could not be loaded");
    if (!h->rccl) {
        std::unique_ptr<qc_rccl_state> st(new qc_rccl_state());
        st->comms.assign(n, nullptr);
        const int r = api.CommInitAll(st->comms.data(), n, devs.data());
        if (r != 0) return fail(&h->err, QC_ERR_HIP, std::string("ncclCommInitAll: ") + api.GetErrorString(r));
        h->rccl = st.release();
    }
    const size_t count = (size_t)h->shard_chunk * (size_t)per_interval;
    int r = api.GroupStart();
    for (int i = 0; r == 0 && i < n; ++i) {
        qc_device_guard guard(devs[i]);
        // in place: rank i's contribution already sits at offset i * count of its own full-length vector
        r = api.AllGather(bufs[i] + (size_t)i * count, bufs[i], count, kNcclFloat64, h->rccl->comms[i], h->shards[i]->stream);
    }
    const int r2 = api.GroupEnd();
    if (r != 0 || r2 != 0) return fail(&h->err, QC_ERR_HIP, std::string("ncclAllGather: ") + api.GetErrorString(r != 0 ? r : r2));
    return QC_OK;
}

// ------------------------------------------------------------------------------------------------
//  Host-buffer entry points (single- and multi-device handles)
// ------------------------------------------------------------------------------------------------
static int eval_host_any(qc_handle* h, const double* Z, double* F, double* vals) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_eval: NULL handle");
    if (!is_multi(h)) return eval_host(h, Z, F, vals, 1);
    if (!Z || (!F && !vals)) return fail(&h->err, QC_ERR_INVALID, "qc_eval: NULL buffer");
    const int n = (int)h->shards.size();
    return multi_run(h, [&](int i) {
        return eval_host(h->shards[i], Z, F ? F + h->shard_F_off[i] : nullptr, vals ? vals + h->shard_J_off[i] : nullptr, n);
    });
}

extern "C" int qc_eval_F(qc_handle* h, const double* Z, double* F) { return eval_host_any(h, Z, F, nullptr); }
extern "C" int qc_eval_jac(qc_handle* h, const double* Z, double* vals) { return eval_host_any(h, Z, nullptr, vals); }
extern "C" int qc_eval_F_jac(qc_handle* h, const double* Z, double* F, double* vals) {
    if (h && (!F || !vals)) return fail(&h->err, QC_ERR_INVALID, "qc_eval_F_jac: NULL buffer");
    return eval_host_any(h, Z, F, vals);
}

extern "C" int qc_eval_hess(qc_handle* h, const double* Z, const double* mu, double* hvals) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_eval_hess: NULL handle");
    if (!is_multi(h)) return hess_host(h, Z, mu, hvals, 1);
    if (h->prm.hess_nnz == 0) return QC_OK;
    if (!Z || !mu || !hvals) return fail(&h->err, QC_ERR_INVALID, "qc_eval_hess: NULL buffer");
    const int n = (int)h->shards.size();
    return multi_run(h, [&](int i) { return hess_host(h->shards[i], Z, mu, hvals + h->shard_H_off[i], n); });
}

// ------------------------------------------------------------------------------------------------
//  Integrator lists with several state integrators, host buffers (UnitarySamplingProblem, UnitaryDirectSumProblem,
//  QuantumStateSamplingProblem): one composed handle per state integrator, ONE upload of the knots, the batched launch where
//  the handles allow it, and the whole-problem value vectors copied straight into the caller's arrays.  hs[0] owns the
//  staging buffers and the stream; qc_set_new_x / qc_knot_generation on hs[0] mean what they mean on a single handle.
// ------------------------------------------------------------------------------------------------
static int list_check(qc_handle* const* hs, int32_t count, const char* who) {
    if (!hs || count < 1) return fail(nullptr, QC_ERR_INVALID, std::string(who) + ": no handles");
    for (int i = 0; i < count; ++i) {
        if (!hs[i]) return fail(nullptr, QC_ERR_INVALID, std::string(who) + ": NULL handle");
        if (is_multi(hs[i])) return fail(&hs[0]->err, QC_ERR_INVALID, std::string(who) + ": single- and multi-device handles cannot be mixed in one list");
    }
    const qc_handle* h0 = hs[0];
    const QcParams& P0 = h0->prm;
    for (int i = 1; i < count; ++i) {
        const QcParams& P = hs[i]->prm;
        if (hs[i]->device != h0->device || P.zdim != P0.zdim || hs[i]->desc.T != h0->desc.T || P.t_begin != P0.t_begin || P.n_int != P0.n_int ||
            hs[i]->dims.Z_len != h0->dims.Z_len || P.F_stride != P0.F_stride || P.J_stride != P0.J_stride)
            return fail(&hs[0]->err, QC_ERR_INVALID, std::string(who) + ": the handles do not describe one problem (device, trajectory, interval range "
                        "or the per-interval block sizes differ)");
    }
    return QC_OK;
}

// The Jacobian values of a list through the compact form and the landing watch (eval_host's mechanism): per interval ONE block
// [ the problem's residual rows | handle 0's values | handle 1's values | ... ] in HBM, every handle that can write the compact form
// of its values itself (one copy of the N replicated blocks) does, one copy brings the blocks to a pinned ring block of hs[0],
// and the host team replicates segment by segment into the caller's array.  Returns 1 when it served the call, 0 when the
// list has nothing to gain from it (the plain copies follow), < 0 on error.
// The members' landing-layout parameter blocks in device memory, for the ONE batched launch (gridDim.y = count) that writes every
// member's segment of the per-interval blocks: cached on hs[0] while the members and the block layout stay the same.
static int land_batch_params(qc_handle* const* hs, int32_t count, const std::vector<QcParams>& blocks, size_t blk) {
    qc_handle* h0 = hs[0];
    bool same = h0->dBatchLand != nullptr && (int)h0->land_members.size() == count && h0->land_blk == blk;
    for (int i = 0; same && i < count; ++i) same = h0->land_members[i] == hs[i]->serial;
    if (same) return QC_OK;
    if (h0->dBatchLand) { QC_HIP(h0, hipStreamSynchronize(h0->stream)); (void)hipFree(h0->dBatchLand); h0->dBatchLand = nullptr; }
    QC_HIP(h0, hipMalloc((void**)&h0->dBatchLand, sizeof(QcParams) * count));
    QC_HIP(h0, hipMemcpy(h0->dBatchLand, blocks.data(), sizeof(QcParams) * count, hipMemcpyHostToDevice));
    h0->land_members.clear();
    for (int i = 0; i < count; ++i) h0->land_members.push_back(hs[i]->serial);
    h0->land_blk = blk;
    return QC_OK;
}

static int list_eval_landing(qc_handle* const* hs, int32_t count, double* F, double* vals, int shards) {
    qc_handle* h = hs[0];
    const QcParams& P0 = h->prm;
    if (!vals || h->host_landing != 1 || h->host_compact != 1) return 0;
    struct Seg { CompactPlan cp; bool direct; size_t len, off; };
    std::vector<Seg> seg((size_t)count);
    const size_t f_len = (size_t)P0.F_stride;       // always in the block: its layout must not depend on what a call asks for
    size_t blk = f_len;
    bool any = false;
    for (int i = 0; i < count; ++i) {
        const QcParams& P = hs[i]->prm;
        Seg& S = seg[(size_t)i];
        S.cp = compact_plan(P);
        S.direct = S.cp.useful && hs[i]->kernel == QC_KERNEL_MFMA && qc_mfma_compact_supported(P);
        S.len = S.direct ? (size_t)S.cp.comp_len : (size_t)P.jac_nnz;
        S.off = blk;
        blk += S.len;
        any = any || S.direct;
    }
    if (!any) return 0;
    const double t_begin = now_us();
    const size_t n_int = (size_t)P0.n_int, cap = n_int * blk;
    int rc;
    // the layout's identity: the members (a recycled address is not the same handle: serial numbers), their row / value placement and
    // the block size -- two lists with equal block sizes but different row ownership must not share a zeroed-once block
    unsigned long long tag = 1469598103934665603ull;
    auto mix = [&tag](unsigned long long v) { tag = (tag ^ v) * 1099511628211ull; };
    mix((unsigned long long)count); mix((unsigned long long)blk); mix((unsigned long long)n_int);
    for (int i = 0; i < count; ++i) { mix(hs[i]->serial); mix((unsigned long long)hs[i]->prm.F_off); mix((unsigned long long)seg[(size_t)i].off); }
    tag |= 2ull;                                              // (never 0 = untouched, never 1 = the handle's own calls)
    if ((rc = claim_staging(h, tag, cap))) return rc;
    if ((rc = ensure_zeroed(h, &h->dC, cap))) return rc;      // (zeroed once per layout: rows no handle of the list owns are delivered as 0)
    int ib;
    if ((rc = ring_take(h, &ib, cap))) return rc;
    std::vector<QcParams> Cs((size_t)count);
    // one launch for every member where the batched kernel serves them all (2N <= 16, order 4, equal shapes: the K systems of a
    // sampling problem, the members of a direct sum of equal systems), as the "_dev_multi" entry points do; else one launch each
    bool batch = count >= 2 && count <= 65535;
    static const bool no_batch = getenv("QC_LIST_BATCH") && atoi(getenv("QC_LIST_BATCH")) == 0;     // A/B diagnostics
    for (int i = 0; i < count; ++i) {
        const Seg& S = seg[(size_t)i];
        QcParams& C = Cs[(size_t)i];
        C = S.direct ? compact_params(hs[i]->prm, S.cp) : hs[i]->prm;
        C.J_stride = (long long)blk;
        C.J_off = (long long)S.off;
        C.F_stride = (long long)blk;                          // (F_off = the handle's first row inside the problem's rows: unchanged)
        const QcParams& Pi = hs[i]->prm;
        batch = batch && !no_batch && S.direct && hs[i]->kernel == QC_KERNEL_MFMA && qc_mfma16_batchable(Pi) && Pi.m == P0.m && Pi.n == P0.n && Pi.nc == P0.nc;
    }
    if (batch) {
        if ((rc = land_batch_params(hs, count, Cs, blk))) return rc;
        const hipError_t e = qc_launch_mfma16_F_jac_batch(Cs[0], h->dBatchLand, count, h->dZ, h->dC, h->dC, h->stream);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(h->stream);
            return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        }
    } else {
        for (int i = 0; i < count; ++i) {
            const QcParams& C = Cs[(size_t)i];
            const hipError_t e = hs[i]->kernel == QC_KERNEL_MFMA ? qc_launch_mfma_F_jac(C, h->dZ, h->dC, h->dC, h->stream)
                                                                  : qc_launch_lds_F_jac(C, h->dZ, h->dC, h->dC, hs[i]->lds_bytes_jac, h->stream);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(h->stream);
                return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            }
        }
    }
    hipError_t ec = hipMemcpyAsync(h->hC[ib], h->dC, cap * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (ec == hipSuccess) ec = hipEventRecord(h->ev_done, h->stream);
    if (ec != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        h->hC_armed[ib] = false;
        return fail(&h->err, QC_ERR_HIP, std::string("copy of the compact values: ") + hipGetErrorString(ec));
    }
    LandJob J;
    auto layout_of = [](const QcParams& P, const Seg& S) {
        qc_team::LandLayout L;
        L.jac_nnz = P.jac_nnz;
        if (S.direct) {
            L.jo_F = P.jo_F; L.jo_B = P.jo_B; L.n2 = S.cp.n2; L.copies = S.cp.copies; L.second_copies = S.cp.second_copies;
            L.head2 = S.cp.head2; L.tail_src = S.cp.tail_src; L.tail_len = S.cp.tail_len;
        } else {
            L.tail_len = P.jac_nnz;                           // nothing replicated: the segment is the values
        }
        return L;
    };
    J.lay = layout_of(P0, seg[0]);
    J.dst_stride = (size_t)P0.J_stride;
    J.dst_off0 = (size_t)P0.J_off;
    for (int i = 1; i < count; ++i) {
        qc_team::LandSeg S;
        S.lay = layout_of(hs[i]->prm, seg[(size_t)i]);
        S.src_off = seg[(size_t)i].off;
        S.dst_off = (size_t)hs[i]->prm.J_off;
        J.more.push_back(S);
    }
    J.n_int = (int)n_int;
    J.f_len = f_len;
    J.blk = blk;
    J.vals = vals;
    J.F = F;
    J.src = h->hC[ib];
    J.rearm_inline = qc_team::land_inline_rearm();
    rc = land_run(h, J, shards, t_begin, now_us());
    if (rc) { h->hC_armed[ib] = false; return rc; }
    if (!J.rearm_inline) ring_rearm_later(h, ib, cap);
    return 1;
}

static int list_eval(qc_handle* const* hs, int32_t count, const double* Z, const double* mu, double* F, double* vals, double* hvals, const char* who,
                     int shards = 1);

// A list of multi-device handles (every member created by qc_create_multi with the same device list): shard s of every member covers
// the same intervals on the same device, so the list is evaluated shard by shard -- one host thread per shard (the leader's), each
// uploading its knots, launching on its device and landing its slice of the caller's arrays over its own PCIe link.  Z and mu are the
// full vectors; F / vals / hvals the whole problem's.
static int list_eval_multi(qc_handle* const* hs, int32_t count, const double* Z, const double* mu, double* F, double* vals, double* hvals, const char* who) {
    qc_handle* h0 = hs[0];
    const int n = (int)h0->shards.size();
    for (int i = 0; i < count; ++i) {
        if (!hs[i]) return fail(nullptr, QC_ERR_INVALID, std::string(who) + ": NULL handle");
        if (!is_multi(hs[i]) || (int)hs[i]->shards.size() != n)
            return fail(&h0->err, QC_ERR_INVALID, std::string(who) + ": every member of a multi-device list must be a multi-device handle over the same device list");
        if (hs[i]->prm.F_stride != h0->prm.F_stride || hs[i]->prm.J_stride != h0->prm.J_stride)
            return fail(&h0->err, QC_ERR_INVALID, std::string(who) + ": the handles do not describe one problem (per-interval block sizes differ)");
    }
    if (!Z || (!F && !vals && !hvals) || (hvals && !mu)) return fail(&h0->err, QC_ERR_INVALID, std::string(who) + ": NULL buffer");
    std::vector<std::vector<qc_handle*>> sub((size_t)n, std::vector<qc_handle*>((size_t)count));
    for (int sidx = 0; sidx < n; ++sidx)
        for (int i = 0; i < count; ++i) sub[(size_t)sidx][(size_t)i] = hs[i]->shards[(size_t)sidx];
    return multi_run(h0, [&](int sidx) {
        // (offsets in doubles of shard sidx's slice: the same for every member, they share the per-interval strides)
        const long long b0 = std::min<long long>(h0->prm.n_int, (long long)sidx * h0->shard_chunk);
        return list_eval(sub[(size_t)sidx].data(), count, Z, mu, F ? F + b0 * h0->prm.F_stride : nullptr, vals ? vals + b0 * h0->prm.J_stride : nullptr,
                         hvals ? hvals + b0 * h0->prm.H_stride : nullptr, who, n);
    });
}

static int list_eval(qc_handle* const* hs, int32_t count, const double* Z, const double* mu, double* F, double* vals, double* hvals, const char* who,
                     int shards) {
    int rc;
    if (hs && count >= 1 && hs[0] && is_multi(hs[0])) return list_eval_multi(hs, count, Z, mu, F, vals, hvals, who);
    if ((rc = list_check(hs, count, who))) return rc;
    qc_handle* h = hs[0];
    const QcParams& P = h->prm;
    if (!Z || (!F && !vals && !hvals) || (hvals && !mu)) return fail(&h->err, QC_ERR_INVALID, std::string(who) + ": NULL buffer");
    if (hvals) {
        for (int i = 1; i < count; ++i)
            if ((hs[i]->prm.hess_nnz != 0) != (P.hess_nnz != 0) || (P.hess_nnz && hs[i]->prm.H_stride != P.H_stride))
                return fail(&h->err, QC_ERR_INVALID, std::string(who) + ": the handles do not share one per-interval Hessian block");
        if (P.hess_nnz == 0) hvals = nullptr;     // linear constraints: nothing to write
    }
    if (P.n_int == 0) return QC_OK;
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    const size_t n_int = (size_t)P.n_int;
    const size_t nF = n_int * (size_t)P.F_stride, nJ = n_int * (size_t)P.J_stride, nH = hvals ? n_int * (size_t)P.H_stride : 0;
    if ((rc = drain_if_needed(h))) return rc;
    for (int i = 1; i < count; ++i) if ((rc = drain_leader_if_needed(hs[i]))) return rc;
    if ((rc = upload_knots(h, Z))) return rc;
    if (vals) {
        if ((rc = list_eval_landing(hs, count, F, vals, shards)) < 0) {
            if (h->needs_drain) mark_members(hs, count);
            return rc;
        }
        if (rc == 1) {
            if (!hvals) return QC_OK;
            vals = nullptr;
            F = nullptr;
        }
    }
    {   // a DIFFERENT list led by this handle (other members: other rows and values owned), or the handle's own calls before
        unsigned long long tag = 1469598103934665603ull;
        for (int i = 0; i < count; ++i) tag = (tag ^ hs[i]->serial) * 1099511628211ull;
        tag |= 2ull;
        if ((rc = claim_plain(h, tag))) return rc;
    }
    if (F && (rc = ensure_zeroed(h, &h->dF, nF))) return rc;       // rows no handle of the list owns stay 0
    if (vals && (rc = ensure_zeroed(h, &h->dJ, nJ))) return rc;
    if (F || vals) {
        if ((rc = qc_eval_F_jac_dev_multi(hs, count, h->dZ, F ? h->dF : nullptr, vals ? h->dJ : nullptr, h->stream))) return rc;
        if (F) QC_HIP(h, hipMemcpyAsync(F, h->dF, nF * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (vals) QC_HIP(h, hipMemcpyAsync(vals, h->dJ, nJ * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (hvals) {
        // (the kernels index the multipliers from interval 0 of the trajectory: this handle's slice goes to its place in a full-length vector)
        const size_t nMu = n_int * (size_t)P.F_stride, m0 = (size_t)P.t_begin * (size_t)P.F_stride;
        if ((rc = ensure(h, &h->dMu, (size_t)(h->desc.T - 1) * (size_t)P.F_stride))) return rc;
        if ((rc = ensure_zeroed(h, &h->dH, nH))) return rc;
        QC_HIP(h, hipMemcpyAsync(h->dMu + m0, mu + m0, nMu * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if ((rc = qc_eval_hess_dev_multi(hs, count, h->dZ, h->dMu, h->dH, h->stream))) return rc;
        QC_HIP(h, hipMemcpyAsync(hvals, h->dH, nH * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    QC_HIP(h, hipEventRecord(h->ev_done, h->stream));
    rc = wait_done(h, shards);
    if (h->needs_drain) mark_members(hs, count);      // timed out: the members' caller-owned arrays may still be written by the leader's streams
    return rc;
}

extern "C" int qc_eval_F_list(qc_handle* const* hs, int32_t count, const double* Z, double* F) {
    if (hs && count >= 1 && hs[0] && !F) return fail(&hs[0]->err, QC_ERR_INVALID, "qc_eval_F_list: NULL buffer");
    return list_eval(hs, count, Z, nullptr, F, nullptr, nullptr, "qc_eval_F_list");
}
extern "C" int qc_eval_jac_list(qc_handle* const* hs, int32_t count, const double* Z, double* vals) {
    if (hs && count >= 1 && hs[0] && !vals) return fail(&hs[0]->err, QC_ERR_INVALID, "qc_eval_jac_list: NULL buffer");
    return list_eval(hs, count, Z, nullptr, nullptr, vals, nullptr, "qc_eval_jac_list");
}
extern "C" int qc_eval_F_jac_list(qc_handle* const* hs, int32_t count, const double* Z, double* F, double* vals) {
    if (hs && count >= 1 && hs[0] && (!F || !vals)) return fail(&hs[0]->err, QC_ERR_INVALID, "qc_eval_F_jac_list: NULL buffer");
    return list_eval(hs, count, Z, nullptr, F, vals, nullptr, "qc_eval_F_jac_list");
}
extern "C" int qc_eval_hess_list(qc_handle* const* hs, int32_t count, const double* Z, const double* mu, double* hvals) {
    if (hs && count >= 1 && hs[0] && !hvals) return fail(&hs[0]->err, QC_ERR_INVALID, "qc_eval_hess_list: NULL buffer");
    return list_eval(hs, count, Z, mu, nullptr, nullptr, hvals, "qc_eval_hess_list");
}

// ------------------------------------------------------------------------------------------------
//  Rollouts
// ------------------------------------------------------------------------------------------------
extern "C" int qc_rollout_dev(qc_handle* h, const double* dZ, const double* dinit, double* dout, void* stream) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_rollout_dev: NULL handle");
    QC_NOT_MULTI(h, "qc_rollout_dev");
    if (!dZ || !dinit || !dout) return fail(&h->err, QC_ERR_INVALID, "qc_rollout_dev: NULL buffer");
    if (!qc_rollout_supported(h->prm)) return fail(&h->err, QC_ERR_UNSUPPORTED, "qc_rollout: state dimension 2N > 64");
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    size_t nE, nQ, nS;
    int chunk, n_chunks;
    qc_rollout_scratch(h->prm, h->desc.T, &nE, &nQ, &nS, &chunk, &n_chunks);
    int rc;
    if ((rc = ensure(h, &h->dRE, nE))) return rc;
    if ((rc = ensure(h, &h->dRQ, nQ))) return rc;
    if ((rc = ensure(h, &h->dRS, nS))) return rc;
    hipError_t e = qc_launch_rollout(h->prm, h->desc.T, dZ, dinit, dout, h->dRE, h->dRQ, h->dRS, (hipStream_t)stream);
    if (e != hipSuccess) return fail(&h->err, QC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QC_OK;
}

extern "C" int qc_rollout(qc_handle* h, const double* Z, const double* init, double* out) {
    if (!h) return fail(nullptr, QC_ERR_INVALID, "qc_rollout: NULL handle");
    if (is_multi(h)) {   // a rollout is a sequential scan over the whole trajectory: shard 0's device serves it
        const int rc = qc_rollout(h->shards[0], Z, init, out);
        if (rc) h->err = h->shards[0]->err;
        return rc;
    }
    if (!Z || !init || !out) return fail(&h->err, QC_ERR_INVALID, "qc_rollout: NULL buffer");
    qc_device_guard guard(h->device);
    QC_HIP(h, guard.err);
    const size_t ns = (size_t)h->prm.n * h->prm.nc, T = (size_t)h->desc.T;
    int rc;
    // The rollout's trajectory vector has a device buffer of its own: h->dZ holds the knots of the last EVALUATION, which a later
    // call under qc_set_new_x(h, 0) reuses (on a multi-device handle only shard 0 would have seen the rollout's vector).
    if ((rc = ensure(h, &h->dRZ, (size_t)h->dims.Z_len))) return rc;
    if ((rc = ensure(h, &h->dRinit, ns))) return rc;
    if ((rc = ensure(h, &h->dRout, ns * T))) return rc;
    QC_HIP(h, hipMemcpyAsync(h->dRZ, Z, (size_t)h->dims.Z_len * sizeof(double), hipMemcpyHostToDevice, h->stream));
    QC_HIP(h, hipMemcpyAsync(h->dRinit, init, ns * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if ((rc = qc_rollout_dev(h, h->dRZ, h->dRinit, h->dRout, h->stream))) return rc;
    QC_HIP(h, hipMemcpyAsync(out, h->dRout, ns * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    QC_HIP(h, hipStreamSynchronize(h->stream));
    return QC_OK;
}

// ------------------------------------------------------------------------------------------------
//  ABI struct sizes (bindings assert them at load time)
// ------------------------------------------------------------------------------------------------
extern "C" int64_t qc_sizeof_desc(void) { return (int64_t)sizeof(qc_desc); }
extern "C" int64_t qc_sizeof_dims(void) { return (int64_t)sizeof(qc_dims_t); }
extern "C" int64_t qc_sizeof_terms_desc(void) { return (int64_t)sizeof(qc_terms_desc); }
