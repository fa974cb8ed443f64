// The host-side team of the host-buffer path (qc_host_eval.cpp): the process-wide worker pool, the "landing watch" -- host threads
// that replicate the compact Jacobian form out of a pinned block WHILE the GPU's copy engine is still filling it, the data being its
// own completion flag -- and the background re-arm of the pinned ring.  Plain C++17: no HIP, no handle.  The copy engine appears as a
// callback that says whether the copy has completed, so tests/host_team_test.cpp can drive every line of this file on the CPU, under
// -fsanitize=thread and -fsanitize=address, with a thread standing in for the engine (in order, in random order, stalling).
#pragma once

#include <sched.h>
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
#include <thread>
#include <utility>
#include <vector>

// qc_host_copy.cpp (host compiler, x86 intrinsics, run-time dispatch)
typedef void (*qc_copy_fn)(double*, const double*, size_t);
void qc_host_copy_select(int mode, qc_copy_fn* fn);
void qc_host_copy_fence();
size_t qc_host_scan(const double* p, size_t n, unsigned long long sentinel);
void qc_host_fill(double* p, size_t n, unsigned long long sentinel);

namespace qc_team {

inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline bool host_trace() { static const bool on = getenv("QC_HOST_TRACE") && atoi(getenv("QC_HOST_TRACE")); return on; }
inline void cpu_pause() { __builtin_ia32_pause(); }

// How long a host thread waits for the device before the call gives up with an error instead of spinning for ever on a lost copy or a
// hung device (QC_HOST_TIMEOUT_MS; default 30 s -- a config-4 evaluation takes 3 ms).
inline std::atomic<double>& timeout_override_us() { static std::atomic<double> v{0.0}; return v; }   // tests only (> 0: instead of the environment's)
inline double timeout_us() {
    static const double v = [] {
        const char* e = getenv("QC_HOST_TIMEOUT_MS");
        const double ms = e ? atof(e) : 30000.0;
        return (ms > 0.0 ? ms : 30000.0) * 1e3;
    }();
    const double o = timeout_override_us().load(std::memory_order_relaxed);
    return o > 0.0 ? o : v;
}

// the clock is read every 1024th turn of a spin loop (a deadline below a millisecond -- tests -- is checked on every turn)
inline unsigned deadline_check_mask() { return timeout_us() < 1e3 ? 0u : 1023u; }

inline qc_copy_fn host_copy() {
    static qc_copy_fn fn = [] {
        qc_copy_fn f = nullptr;
        const char* ev = getenv("QC_HOST_NT");
        qc_host_copy_select(ev ? atoi(ev) : 1, &f);
        return f;
    }();
    return fn;
}

struct HostGroup {   // completion of one call's jobs
    std::mutex mu;
    std::condition_variable cv;
    int outstanding = 0;
    void add() { std::lock_guard<std::mutex> lk(mu); ++outstanding; }
    void done() { std::lock_guard<std::mutex> lk(mu); if (--outstanding == 0) cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return outstanding == 0; }); }
};

// Process-wide pool of replication workers, shared by every handle (the shards of a multi-device handle push into it concurrently).
// Workers block on a condition variable between jobs (threads spinning in hipEventSynchronize per chunk were tried first: on a
// CPU-quota-limited host they starve the copying ones).
struct HostPool {
    std::vector<std::thread> th;
    std::vector<cpu_set_t> domains;       // where the members go: the core complexes next to every device served so far (or empty)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<std::function<void()>, HostGroup*>> q;
    bool stop = false;
    // (Workers that poll for ~100 us before blocking were measured on the 16-CPU-quota host: 0.55 instead of 0.48 ms per
    // config-3 evaluation -- the polling threads eat the quota the launching thread needs.)
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;   // (stop: only once the queue has drained -- re-arm jobs outlive their call, and their group is waited for)
            auto job = std::move(q.front());
            q.pop_front();
            lk.unlock();
            job.first();
            job.second->done();
            lk.lock();
        }
    }
    std::vector<int> devices_seen;        // devices whose NUMA node's complexes are in `domains`
    // `n` members at least; `device` >= 0 and `domains_of`: the complexes of a device not served before join the list
    // (the shards of a multi-device handle sit on both sockets) -- the sysfs walk itself needs the HIP runtime and stays with the caller
    void ensure(int n, int device = -1, const std::function<std::vector<cpu_set_t>(int)>& domains_of = nullptr) {
        std::lock_guard<std::mutex> lk(mu);
        // QC_HOST_AFFINITY=0: leave the members to the scheduler
        static const bool pin = !(getenv("QC_HOST_AFFINITY") && atoi(getenv("QC_HOST_AFFINITY")) == 0);
        bool repin = false;
        if (pin && domains_of && device >= 0 && std::find(devices_seen.begin(), devices_seen.end(), device) == devices_seen.end()) {
            devices_seen.push_back(device);
            for (const cpu_set_t& d : domains_of(device)) {
                bool known = false;
                for (const cpu_set_t& e : domains) known = known || CPU_EQUAL(&d, &e);
                if (!known) { domains.push_back(d); repin = true; }
            }
        }
        while ((int)th.size() < n) { th.emplace_back([this] { run(); }); repin = true; }
        if (repin && !domains.empty()) {
            // member i on complex (i + 1) mod n: complex 0 is left to the calling thread's side of the work when it happens to be there
            for (size_t i = 0; i < th.size(); ++i) {
                const cpu_set_t& set = domains[(i + 1) % domains.size()];
                (void)pthread_setaffinity_np(th[i].native_handle(), sizeof(cpu_set_t), &set);
            }
        }
    }
    void push(std::function<void()> fn, HostGroup* g) {
        g->add();
        { std::lock_guard<std::mutex> lk(mu); q.emplace_back(std::move(fn), g); }
        cv.notify_one();
    }
    // `count` team members running the same function: one wake-up call for all of them (a notify per job is a futex call each)
    void push_many(const std::function<void()>& fn, int count, HostGroup* g) {
        if (count <= 0) return;
        for (int i = 0; i < count; ++i) g->add();
        { std::lock_guard<std::mutex> lk(mu); for (int i = 0; i < count; ++i) q.emplace_back(fn, g); }
        cv.notify_all();
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};
inline HostPool& host_pool() {
    static HostPool* p = new HostPool();   // intentionally leaked: worker threads must not be joined from a static destructor
    return *p;
}

// ------------------------------------------------------------------------------------------------
//  Landing watch
// ------------------------------------------------------------------------------------------------
// The host-buffer calls are bound by the PCIe link (config 3: 12.7 MB of residuals and compact Jacobian values, 14.7 MB of
// Hessian values per call).  What tests/hip/landing_probe.hip measured on the MI355X host (profiles/r03_landing_probe.txt):
//   * the copy engine moves 12.8 MB device -> host in 234 us (54.5 GB/s), into pinned AND into pageable memory (the runtime pins
//     the caller's pages in place; the call then blocks for the duration); 1.2 MB host -> device take 30 us either way;
//   * kernel stores into pinned host memory reach 44 - 47 GB/s whatever the grid, and they land in NO usable order: the L2
//     acknowledges a store long before it crosses the link and writes back in its own order;
//   * round 2's sixteen chunk launches with an event each: 36 - 47 GB/s and 16 launch latencies.
// So: ONE kernel writes the call's compact output into HBM (3 - 9 us), ONE asynchronous copy brings it to a pinned block in
// address order at the link's rate, and the data is its own completion flag -- the pinned block holds a sentinel word (a
// signalling NaN that no arithmetic produces) wherever the copy has not arrived yet; a team of host threads (the calling thread
// and pool workers) claims pieces of the interval range as their first block is seen to have landed, waits block by block until
// no word of a block is the sentinel, replicates the block into the caller's array and re-arms it.  No events between
// chunks, no chunk launches.  Correctness does not rest on the sentinel being unique: the calling thread polls the copy's
// completion, and once that has been seen every remaining word is taken as it is (a result that happens to equal the
// sentinel -- possible only if the caller's input carries that NaN payload -- costs the overlap, not the answer).
constexpr unsigned long long kLandSentinel = 0x7FF4C0DEC0DE5A5Aull;

// where the replicated blocks go in the caller's value array (one interval): `copies` copies of the first n2 compact values from
// jo_F on, `second_copies` of the next n2 from jo_B on, then tail_len values at tail_src (compact position head2)
struct LandLayout {
    int jac_nnz = 0, jo_F = 0, jo_B = 0, n2 = 0, copies = 0, second_copies = 0, head2 = 0, tail_src = 0, tail_len = 0;
};
// A further value segment of the interval blocks (integrator lists: one segment per state integrator): its compact values start
// `src_off` doubles into the interval's block and are replicated to `dst_off` doubles into the interval's values.
struct LandSeg {
    LandLayout lay;
    size_t src_off = 0, dst_off = 0;
};

enum { LAND_PENDING = 0, LAND_DONE = 1, LAND_FAILED = 2, LAND_TIMEOUT = 3 };
typedef std::function<int()> land_poll_fn;   // LAND_PENDING / LAND_DONE / LAND_FAILED: has the copy into the block completed?  (calling thread only)

struct LandJob {
    LandLayout lay;
    // the watched pinned block: one sub-block of `blk` doubles per interval = [ residual rows (f_len) | compact Jacobian values ]
    double* src = nullptr;
    size_t blk = 0, f_len = 0;
    bool rearm_inline = false;
    double* vals = nullptr;                             // caller's Jacobian values (replication target)
    size_t dst_stride = 0, dst_off0 = 0;                // values per interval in `vals` (0: lay.jac_nnz) and where `lay`'s values start in them
    std::vector<LandSeg> more;                          // segments behind the first (`lay`, at f_len)
    double* F = nullptr;                                // caller's residuals, or nullptr
    int n_int = 0;
    double t_begin = 0.0;                               // the deadline counts from here
    std::vector<int> bound;                             // piece k = intervals [bound[k], bound[k + 1])
    std::unique_ptr<std::atomic<int>[]> claimed;
    std::atomic<int> remaining{0};
    std::atomic<int> done{LAND_PENDING};                // LAND_DONE: the copy's completion has been seen; LAND_FAILED / LAND_TIMEOUT: give up
    std::atomic<double> t_first_piece{0.0}, t_last_piece{0.0}, t_event{0.0};   // trace (QC_HOST_TRACE)
    std::atomic<int> pieces_at_event{0}, blocks_waited{0};       // trace: pieces done when the copy's end was seen; blocks a member had to wait for
    std::atomic<int> n_members{0};
    int member_cpu[64], member_pieces[64];                       // trace: where each team member ran, how many pieces it took
};

inline bool land_inline_rearm() {   // QC_HOST_REARM=inline: every block re-armed by the member that consumed it, inside the call (A/B diagnostics)
    static const bool v = getenv("QC_HOST_REARM") && !strcmp(getenv("QC_HOST_REARM"), "inline");
    return v;
}
inline bool land_nowatch() {        // QC_HOST_NOWATCH=1: the team does not look at the block before the copy's completion (how long does the copy take alone?)
    static const bool v = getenv("QC_HOST_NOWATCH") && atoi(getenv("QC_HOST_NOWATCH"));
    return v;
}
inline bool land_gave_up(const LandJob& J) { const int d = J.done.load(std::memory_order_acquire); return d == LAND_FAILED || d == LAND_TIMEOUT; }

inline void land_poll(LandJob& J, const land_poll_fn& poll) {   // calling thread only
    const int e = poll();
    if (e == LAND_DONE) {
        if (host_trace() && !J.done.load()) { J.t_event.store(now_us()); J.pieces_at_event.store((int)J.bound.size() - 1 - J.remaining.load()); }
        int expect = LAND_PENDING;
        J.done.compare_exchange_strong(expect, LAND_DONE, std::memory_order_acq_rel);
    } else if (e != LAND_PENDING) J.done.store(LAND_FAILED, std::memory_order_release);
}
inline void land_check_deadline(LandJob& J) {
    if (now_us() - J.t_begin > timeout_us()) {
        int expect = LAND_PENDING;
        J.done.compare_exchange_strong(expect, LAND_TIMEOUT, std::memory_order_acq_rel);
    }
}

// spins until no word of p[0 .. n) holds the sentinel, or the copy is known to be complete or lost (`poll`: this is the calling
// thread, the only one that asks the runtime -- it must keep asking while it waits for a block)
inline void land_wait(LandJob& J, const double* p, size_t n, const land_poll_fn* poll) {
    size_t off = 0;
    unsigned spins = 0;
    while (off < n) {
        off += qc_host_scan(p + off, n - off, kLandSentinel);
        if (off >= n) return;
        if (J.done.load(std::memory_order_acquire)) return;
        if (spins == 0 && host_trace()) J.blocks_waited.fetch_add(1, std::memory_order_relaxed);
        ++spins;
        if (poll && (spins & 15) == 0) land_poll(J, *poll);
        if ((spins & deadline_check_mask()) == 0) land_check_deadline(J);
        cpu_pause();
    }
}

inline bool land_started(const LandJob& J, int b) {   // has interval b's block arrived?  (cheap: its first and last word)
    const volatile unsigned long long* u = (const volatile unsigned long long*)(J.src + (size_t)b * J.blk);
    return u[0] != kLandSentinel && u[J.blk - 1] != kLandSentinel;
}

inline void land_replicate(const LandLayout& L, const double* src, double* dst, qc_copy_fn cpy) {
    for (int c = 0; c < L.copies; ++c) cpy(dst + L.jo_F + (size_t)c * L.n2, src, (size_t)L.n2);
    for (int c = 0; c < L.second_copies; ++c) cpy(dst + L.jo_B + (size_t)c * L.n2, src + L.n2, (size_t)L.n2);
    cpy(dst + L.tail_src, src + L.head2, (size_t)L.tail_len);
}
inline void land_piece(LandJob& J, int k, const land_poll_fn* poll) {
    const LandLayout& L = J.lay;
    const qc_copy_fn cpy = host_copy();
    for (int b = J.bound[k]; b < J.bound[k + 1]; ++b) {
        double* blk = J.src + (size_t)b * J.blk;
        land_wait(J, blk, J.blk, poll);
        if (land_gave_up(J)) return;                       // (what is in the block is not a result)
        if (J.F) memcpy(J.F + (size_t)b * J.f_len, blk, J.f_len * sizeof(double));
        double* row = J.vals + (size_t)b * (J.dst_stride ? J.dst_stride : (size_t)L.jac_nnz);
        land_replicate(L, blk + J.f_len, row + J.dst_off0, cpy);
        for (const LandSeg& S : J.more) land_replicate(S.lay, blk + S.src_off, row + S.dst_off, cpy);
        if (J.rearm_inline) qc_host_fill(blk, J.blk, kLandSentinel);
    }
    qc_host_copy_fence();
}

// Team member: claims pieces whose first block has landed (the copy engine writes in address order, so that is the lowest
// unclaimed piece; any order would work), until none is left.  `poll`: the calling thread also polls the copy's completion.
inline void land_consume(LandJob& J, const land_poll_fn* poll) {
    const int np = (int)J.bound.size() - 1;
    int lo = 0, mine = 0;
    unsigned idle = 0;
    const int me = host_trace() ? J.n_members.fetch_add(1) : 0;
    while (J.remaining.load(std::memory_order_acquire) > 0 && !land_gave_up(J)) {
        bool got = false;
        const bool all = J.done.load(std::memory_order_acquire) == LAND_DONE;
        while (lo < np && J.claimed[lo].load(std::memory_order_relaxed)) ++lo;
        // (only a few pieces beyond the frontier are looked at: reads of lines the copy engine is about to write cost it a snoop each)
        for (int k = lo; k < np && (all || k < lo + 4); ++k) {
            if (J.claimed[k].load(std::memory_order_relaxed)) continue;
            if (!all && (land_nowatch() || !land_started(J, J.bound[k]))) continue;
            int expect = 0;
            if (!J.claimed[k].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) continue;
            land_piece(J, k, poll);
            if (host_trace()) {
                const double t = now_us();
                double z = 0.0;
                J.t_first_piece.compare_exchange_strong(z, t);
                J.t_last_piece.store(t);
            }
            J.remaining.fetch_sub(1, std::memory_order_acq_rel);
            got = true;
            ++mine;
            break;
        }
        if (poll && !all) land_poll(J, *poll);
        if (!got) {
            if ((++idle & deadline_check_mask()) == 0) land_check_deadline(J);
            cpu_pause();
        }
    }
    if (host_trace() && me < 64) { J.member_cpu[me] = sched_getcpu(); J.member_pieces[me] = mine; }
}

inline int land_piece_intervals(size_t block_bytes) {   // ~128 KB of pinned block per piece, at least 2 intervals (QC_HOST_PIECE_KB)
    static const size_t kb = getenv("QC_HOST_PIECE_KB") ? (size_t)std::max(1, atoi(getenv("QC_HOST_PIECE_KB"))) : 128;
    return (int)std::max<size_t>(2, (kb << 10) / std::max<size_t>(1, block_bytes));
}

// Cuts the interval range into pieces, runs the calling thread and `helpers` pool workers over them until the block is consumed,
// the copy is reported lost, or the deadline passes.  Returns J.done's final value (LAND_DONE also when every piece was consumed
// before the completion was seen: the caller then waits for the completion itself).
inline int land_team(LandJob& J, HostPool& pool, int helpers, const land_poll_fn& poll) {
    J.bound.assign(1, 0);
    const int per = land_piece_intervals(J.blk * sizeof(double));
    // The last pieces are cut finer (QC_HOST_TAIL_SPLIT parts each, default 4; 1: uniform): when the copy's last bytes land every
    // member is idle, and the call ends one piece's replication later -- a quarter piece instead of a whole one.
    static const int tail_split = getenv("QC_HOST_TAIL_SPLIT") ? std::max(1, atoi(getenv("QC_HOST_TAIL_SPLIT"))) : 4;
    const int fine = std::max(1, per / tail_split);
    const int tail_from = tail_split > 1 ? std::max(0, J.n_int - 2 * per) : J.n_int;
    for (int b = per; b < J.n_int; b += (b >= tail_from ? fine : per)) J.bound.push_back(b);
    J.bound.push_back(J.n_int);
    const int np = (int)J.bound.size() - 1;
    J.claimed.reset(new std::atomic<int>[np]);
    for (int k = 0; k < np; ++k) J.claimed[k].store(0, std::memory_order_relaxed);
    J.remaining.store(np, std::memory_order_release);
    helpers = std::max(0, std::min(helpers, np - 1));
    HostGroup grp;
    LandJob* Jp = &J;
    pool.push_many([Jp] { land_consume(*Jp, nullptr); }, helpers, &grp);
    land_consume(J, &poll);
    grp.wait();
    return J.done.load(std::memory_order_acquire);
}

// ------------------------------------------------------------------------------------------------
//  The ring of pinned blocks: background re-arm
// ------------------------------------------------------------------------------------------------
// the consumed part of a block is re-armed with the sentinel behind the caller's back by whichever workers are idle; the next call
// that wants the block waits for the group first (normally long done)
struct Rearm { HostGroup grp; };
inline void rearm_later(HostPool& pool, Rearm& r, double* base, size_t used) {
    // QC_HOST_REARM_JOBS workers share it (default 2: a trickle that the next call's transfers hardly notice, done well before
    // the block's next turn in the ring of 3; eight workers re-arm in a burst that slowed the next call's upload of Z from 30
    // to 100 - 190 us when calls follow each other without a pause)
    static const size_t jobs = getenv("QC_HOST_REARM_JOBS") ? (size_t)std::max(1, atoi(getenv("QC_HOST_REARM_JOBS"))) : 2;
    const size_t piece = std::max<size_t>(size_t(1) << 17, (used + jobs - 1) / jobs);      // doubles: at least 1 MB each
    for (size_t o = 0; o < used; o += piece) {
        const size_t len = std::min(piece, used - o);
        pool.push([base, o, len] { qc_host_fill(base + o, len, kLandSentinel); qc_host_copy_fence(); }, &r.grp);
    }
}

}  // namespace qc_team
