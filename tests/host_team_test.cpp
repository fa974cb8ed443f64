// CPU driver of the host team (quantumcollocation.jl_amd/csrc/qc_host_team.h): the worker pool, the landing watch's lock-free piece
// claiming, the sentinel scans, the deferred re-arm jobs and the deadline, with a THREAD standing in for the GPU's copy engine.
// Built and run by tests/run_sanitizers_host.sh under -fsanitize=thread and -fsanitize=address (tests/test_host_team.py).
//   usage: host_team_test [iterations]      exit code 0 = every scenario passed
#include <stdio.h>

#include <random>

#include "qc_host_team.h"

using namespace qc_team;

namespace {

struct Shape { int n_int, f_len; LandLayout lay; int comp_len; };
Shape config3_like(int n_int) {   // one interval of BASELINE config 3: 8 copies of two 16 x 16 blocks, 944 further values, 140 residual rows
    Shape s;
    s.n_int = n_int;
    s.f_len = 140;
    s.lay.n2 = 256; s.lay.copies = 8; s.lay.second_copies = 8; s.lay.jo_F = 0; s.lay.jo_B = 2048; s.lay.head2 = 512;
    s.lay.tail_src = 4096; s.lay.tail_len = 944; s.lay.jac_nnz = 5040;
    s.comp_len = s.lay.head2 + s.lay.tail_len;
    return s;
}

// The stand-in for the copy engine: writes `n` doubles of `payload` into `dst` in pieces (TSan reports that involve this function are
// suppressed: the real writer is a device, the words it writes are their own completion flags -- tests/tsan_host_team.supp).
enum Mode { IN_ORDER, RANDOM_ORDER, STALL };
__attribute__((noinline)) void fake_engine_write(double* dst, const double* payload, size_t n, Mode mode, unsigned seed, std::atomic<int>* complete,
                                                 std::atomic<bool>* release) {
    const size_t piece = 512;   // 4 KB
    std::vector<size_t> order;
    for (size_t o = 0; o < n; o += piece) order.push_back(o);
    std::mt19937 rng(seed);
    if (mode == RANDOM_ORDER) std::shuffle(order.begin(), order.end(), rng);
    size_t k = 0;
    for (size_t o : order) {
        if (mode == STALL && k == order.size() / 2) {          // the copy is lost: nothing more arrives until the test lets go
            while (!release->load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            return;
        }
        memcpy(dst + o, payload + o, std::min(piece, n - o) * sizeof(double));
        if ((++k & 15) == 0) std::this_thread::sleep_for(std::chrono::microseconds(20 + rng() % 60));   // a link, not a memcpy
    }
    complete->store(1, std::memory_order_release);
}

struct Ring {     // three "pinned" blocks taking turns, as qc_handle's hC[] / rearm[] / hC_armed[]
    double* blk[3] = {nullptr, nullptr, nullptr};
    Rearm rearm[3];
    size_t cap = 0;
    int next = 0;
    explicit Ring(size_t doubles) : cap(doubles) {
        for (auto& b : blk) {
            b = static_cast<double*>(aligned_alloc(4096, (cap * sizeof(double) + 4095) / 4096 * 4096));
            qc_host_fill(b, cap, kLandSentinel);
            qc_host_copy_fence();
        }
    }
    ~Ring() { for (int i = 0; i < 3; ++i) { rearm[i].grp.wait(); free(blk[i]); } }
    int take() {
        const int i = next;
        next = (next + 1) % 3;
        rearm[i].grp.wait();
        return i;
    }
};

int fails = 0;
#define EXPECT(c, ...) do { if (!(c)) { ++fails; fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

// one host-buffer call: arm -> "copy" -> team -> verify -> re-arm later
void one_call(Ring& ring, const Shape& s, Mode mode, unsigned seed, int helpers, bool with_F, bool expect_timeout) {
    const size_t blk = (size_t)(with_F ? s.f_len : 0) + s.comp_len, used = (size_t)s.n_int * blk;
    const int ib = ring.take();
    double* block = ring.blk[ib];
    for (size_t i = 0; i < used; i += 97) EXPECT(*(unsigned long long*)(block + i) == kLandSentinel, "block %d not armed at word %zu", ib, i);
    std::vector<double> payload(used), vals((size_t)s.n_int * s.lay.jac_nnz, -1.0), F(with_F ? (size_t)s.n_int * s.f_len : 0, -1.0);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (double& x : payload) x = u(rng);
    std::atomic<int> complete{0};
    std::atomic<bool> release{false};
    std::thread engine(fake_engine_write, block, payload.data(), used, mode, seed, &complete, &release);
    LandJob J;
    J.lay = s.lay;
    J.n_int = s.n_int;
    J.f_len = with_F ? s.f_len : 0;
    J.blk = blk;
    J.src = block;
    J.vals = vals.data();
    J.F = with_F ? F.data() : nullptr;
    J.t_begin = now_us();
    const int done = land_team(J, host_pool(), helpers, [&]() -> int { return complete.load(std::memory_order_acquire) ? LAND_DONE : LAND_PENDING; });
    if (expect_timeout) {
        EXPECT(done == LAND_TIMEOUT, "a stalled copy ended with state %d after %.0f ms, expected the deadline", done, (now_us() - J.t_begin) / 1e3);
        EXPECT(now_us() - J.t_begin < 20.0 * timeout_us(), "the deadline took %.0f ms", (now_us() - J.t_begin) / 1e3);
        release.store(true, std::memory_order_release);
        engine.join();
        qc_host_fill(block, used, kLandSentinel);      // (what qc_host_eval.cpp does after a failed call: re-arm from scratch)
        qc_host_copy_fence();
        return;
    }
    engine.join();
    EXPECT(done == LAND_DONE || done == LAND_PENDING, "state %d", done);
    EXPECT(J.remaining.load() == 0, "%d pieces left", J.remaining.load());
    // the replication, by definition
    size_t bad = 0;
    for (int b = 0; b < s.n_int; ++b) {
        const double* src = payload.data() + (size_t)b * blk + J.f_len;
        const double* v = vals.data() + (size_t)b * s.lay.jac_nnz;
        for (int c = 0; c < s.lay.copies; ++c) bad += memcmp(v + s.lay.jo_F + (size_t)c * s.lay.n2, src, s.lay.n2 * 8) != 0;
        for (int c = 0; c < s.lay.second_copies; ++c) bad += memcmp(v + s.lay.jo_B + (size_t)c * s.lay.n2, src + s.lay.n2, s.lay.n2 * 8) != 0;
        bad += memcmp(v + s.lay.tail_src, src + s.lay.head2, s.lay.tail_len * 8) != 0;
        if (with_F) bad += memcmp(F.data() + (size_t)b * s.f_len, payload.data() + (size_t)b * blk, s.f_len * 8) != 0;
    }
    EXPECT(bad == 0, "%zu replicated blocks differ (mode %d, seed %u, helpers %d)", bad, (int)mode, seed, helpers);
    rearm_later(host_pool(), ring.rearm[ib], block, used);
}

// an integrator list: three value segments per interval block (compact, plain, compact), one replicated array of all three
void one_list_call(Mode mode, unsigned seed, int helpers, bool with_F) {
    const int n_int = 90, f_len = 52;
    LandLayout big = config3_like(1).lay, plain, small;
    plain.jac_nnz = 300; plain.tail_len = 300;                                  // nothing replicated: the segment is the values
    small.n2 = 16; small.copies = 2; small.second_copies = 2; small.jo_F = 0; small.jo_B = 32; small.head2 = 32;
    small.tail_src = 64; small.tail_len = 40; small.jac_nnz = 104;
    const LandLayout lays[3] = {big, plain, small};
    size_t comp[3], src_off[3], dst_off[3], blk = f_len, stride = 0;
    for (int i = 0; i < 3; ++i) {
        comp[i] = (size_t)lays[i].head2 + lays[i].tail_len;
        src_off[i] = blk; blk += comp[i];
        dst_off[i] = stride; stride += lays[i].jac_nnz;
    }
    const size_t used = (size_t)n_int * blk;
    Ring ring(used);
    const int ib = ring.take();
    double* block = ring.blk[ib];
    std::vector<double> payload(used), vals((size_t)n_int * stride, -1.0), F((size_t)n_int * f_len, -1.0);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (double& x : payload) x = u(rng);
    std::atomic<int> complete{0};
    std::atomic<bool> release{false};
    std::thread engine(fake_engine_write, block, payload.data(), used, mode, seed, &complete, &release);
    LandJob J;
    J.lay = lays[0];
    J.dst_stride = stride;
    J.dst_off0 = dst_off[0];
    for (int i = 1; i < 3; ++i) { LandSeg S; S.lay = lays[i]; S.src_off = src_off[i]; S.dst_off = dst_off[i]; J.more.push_back(S); }
    J.n_int = n_int;
    J.f_len = f_len;                                                            // (a list's blocks always carry the residual rows)
    J.blk = blk;
    J.src = block;
    J.vals = vals.data();
    J.F = with_F ? F.data() : nullptr;
    J.t_begin = now_us();
    const int done = land_team(J, host_pool(), helpers, [&]() -> int { return complete.load(std::memory_order_acquire) ? LAND_DONE : LAND_PENDING; });
    engine.join();
    EXPECT(done == LAND_DONE || done == LAND_PENDING, "list: state %d", done);
    EXPECT(J.remaining.load() == 0, "list: %d pieces left", J.remaining.load());
    size_t bad = 0;
    for (int b = 0; b < n_int; ++b)
        for (int i = 0; i < 3; ++i) {
            const LandLayout& L = lays[i];
            const double* src = payload.data() + (size_t)b * blk + src_off[i];
            const double* v = vals.data() + (size_t)b * stride + dst_off[i];
            for (int c = 0; c < L.copies; ++c) bad += memcmp(v + L.jo_F + (size_t)c * L.n2, src, L.n2 * 8) != 0;
            for (int c = 0; c < L.second_copies; ++c) bad += memcmp(v + L.jo_B + (size_t)c * L.n2, src + L.n2, L.n2 * 8) != 0;
            bad += memcmp(v + L.tail_src, src + L.head2, L.tail_len * 8) != 0;
        }
    for (int b = 0; with_F && b < n_int; ++b) bad += memcmp(F.data() + (size_t)b * f_len, payload.data() + (size_t)b * blk, f_len * 8) != 0;
    EXPECT(bad == 0, "list: %zu replicated blocks differ (mode %d, seed %u, helpers %d)", bad, (int)mode, seed, helpers);
    rearm_later(host_pool(), ring.rearm[ib], block, used);
}

void caller(int id, int iterations) {
    const Shape s = config3_like(160 + 37 * id);
    Ring ring((size_t)s.n_int * (s.f_len + s.comp_len));
    for (int it = 0; it < iterations; ++it) {
        const Mode mode = (it % 3 == 1) ? RANDOM_ORDER : IN_ORDER;
        one_call(ring, s, mode, 1000u * id + it, /*helpers=*/it % 4, /*with_F=*/it % 2 == 0, false);
    }
}

}  // namespace

int main(int argc, char** argv) {
    timeout_override_us().store(20e6);               // the sanitizers slow everything down: 20 s for the calls that must succeed
    const int iterations = argc > 1 ? atoi(argv[1]) : 12;
    host_pool().ensure(6);
    // several callers at once over the shared pool, as the shards of a multi-device handle
    std::vector<std::thread> callers;
    for (int id = 0; id < 3; ++id) callers.emplace_back(caller, id, iterations);
    for (auto& t : callers) t.join();
    // a copy that never completes: the team gives up at the deadline instead of spinning for ever, and the block is usable again
    {
        const Shape s = config3_like(120);
        Ring ring((size_t)s.n_int * (s.f_len + s.comp_len));
        one_call(ring, s, IN_ORDER, 7, 2, true, false);
        timeout_override_us().store(250e3);          // ... and a quarter of a second for the ones that must not
        one_call(ring, s, STALL, 8, 3, true, true);
        one_call(ring, s, STALL, 9, 0, false, true);
        timeout_override_us().store(20e6);
        for (int k = 0; k < 4; ++k) one_call(ring, s, k & 1 ? RANDOM_ORDER : IN_ORDER, 10 + k, 2, true, false);
    }
    // the value segments of an integrator list
    for (int k = 0; k < 6; ++k) one_list_call(k % 3 == 1 ? RANDOM_ORDER : IN_ORDER, 40 + k, k % 4, k % 2 == 0);
    // queued jobs are not dropped when a pool stops: its destructor drains the queue before the workers leave
    {
        std::atomic<int> ran{0};
        HostGroup grp;
        {
            HostPool pool;
            pool.ensure(2);
            for (int i = 0; i < 64; ++i) pool.push([&ran] { ran.fetch_add(1); std::this_thread::sleep_for(std::chrono::microseconds(50)); }, &grp);
        }
        grp.wait();
        EXPECT(ran.load() == 64, "only %d of 64 queued jobs ran before the pool stopped", ran.load());
    }
    printf("host team test: %d failure(s)\n", fails);
    return fails ? 1 : 0;
}
