// f64-MFMA Hessian-of-Lagrangian kernel for SPARSE DRIVE GENERATORS, order-4 Pade, 2N = 32 (4 qubits: BASELINE config 5).
//
// Drive Hamiltonians in quantum control are almost always sparse: a Pauli string has ONE entry per row, a ladder pair a + a^dagger
// two.  qc_mfma32_hess.hip treats every G_k as a dense 32 x 32 matrix -- 64 MFMAs per drive and interval, 16 KB of generator images
// per wave held in registers, 64 KB of LDS for the cross-drive sums -- and is bound by neither HBM nor the matrix pipes (one
// workgroup per CU at 200 registers; intervals one after the other at 45 % pipe occupancy).  Here a drive generator is R <= 2
// (weight, column) pairs per row ("ELL"), built once by qc_create; with Hermitian Hamiltonians (G^T = -G exactly) and
//     E = G M,   Y_k = G_k M,   Q = M D^T        (M = reshape(mu_t[0:s], 32, 16), D = U_t+1 - U_t, S = U_t+1 + U_t, h = dt)
// the blocks of SURVEY A.4 become
//   (U_t, a_k)  =  c1 h Y_k - c2 h^2 (G_k E + G Y_k)        (a_k, U_t+1) =  c1 h Y_k + c2 h^2 (G_k E + G Y_k)
//   (U_t, h)    =  c1 E - 2 c2 h G E                         (h, U_t+1)   =  c1 E + 2 c2 h G E
//   (a_i, a_k)  =  c2 h^2 < G_i G_k + G_k G_i , Q >          (the products of the CONSTANT generators are tabulated at create time:
//                                                             <N_i, V_k> = tr(M^T G_i G_k D) = <G_i G_k, M D^T>)
//   (a_k, h)    = -< Y_k , -c1 S + 2 c2 h G D > - 2 c2 h < E , G_k D >          (h, h) = -2 c2 < E , G D >
// in which G_k (anything) is a row gather from an LDS copy and the only dense products are G x (32 x 16): E, G D, G E and G Y_k
// per drive -- 16 MFMAs per drive instead of 64, 224 per interval instead of 544 -- plus 16 for the Gram matrix Q.  No generator
// image lives in a register, the cross-drive sums need Q (8 KB) instead of every drive's N_k and V_k (64 KB): a workgroup takes
// 53 KB of LDS and < 128 registers, so TWO intervals are resident per compute unit and one's loads, barriers, reductions and stores
// hide behind the other's products.  One 512-thread workgroup per interval, wave k = drive k:
//   phase 0   wave w assembles half of one A-layout tile of G = G_0 + sum a_k G_k (as qc_mfma32_hess.hip: the same nine image
//             loads, the same sums, bit-identical G); waves 4-7 fetch M, U_t, U_t+1 -> LDS (tile format + plain row-major copies,
//             the gather sources); every wave loads its drive's ELL rows into registers.  Barrier.
//   phase 1   waves 0,1: E tile;  2,3: (G D) tile and W = -c1 S + 2 c2 h G D;  4-7: one tile of Q.  Every drive wave: Y_k by row
//             gathers straight into the A operand of (G Y_k)^T = Y_k^T G^T (16 MFMAs, two accumulator chains).  Barrier.
//   phase 2   drive wave: Y_k^T and (G_k E)^T by row gathers in the store layout, the two matrix blocks (lane <-> row: whole
//             128-byte lines per store); (a_k, h); its share of the m (m + 1) / 2 pair sums over the tabulated entries of Q.
//             Waves 0,1: (G E)^T and the (U_t, h) / (h, U_t+1) column block; wave 2: (h, h); wave 5: derivative-integrator tail.
// Matrix layouts and lane maps: qc_mfma_kernels.hip header, qc_mfma32_kernels.hip (tile index 2 I + K, [pair][lane][2]).
#include <algorithm>
#include <cmath>
#include <vector>

#include "qc_mfma_common.h"

// No fused multiply-adds of the compiler's choosing in this file: the Hessian values of the one-call instantiation must equal those of
// the Hessian-only instantiation bit for bit (include/qcolloc.h, qc_eval_F_jac_hess_dev), and which a * b + c contracts depends on the
// code around it.  (The products that matter are MFMAs; the vector arithmetic here is a few hundred instructions per wave.)
#pragma clang fp contract(off)

namespace {

using namespace qc_mfma;

constexpr int kEThreads = 512;
constexpr int kEMax = 8;            // drives (one wave each)
constexpr int kPS = 17;             // row stride of the plain 32 x 16 LDS copies (doubles)
constexpr int kQS = 33;             // row stride of the plain 32 x 32 Gram matrix

// Byte offsets of the tables inside the device blob (host and device agree through this one function).
//   tw / tc    [k][q][32]     row a of drive k: weight and column (pre-multiplied by kPS) of its q-th entry
//   gw / gk    [slot][1024]   assembly plan of G = G_0 + sum_k a_k G_k in the order of the A-layout image: the slot-th drive that
//                             touches the entry (ascending k, so the sum has the order of the dense one) and its weight; k = 0, w = 0
//                             where fewer drives do
struct EllLayout { size_t tw, gw, tc, gk, bytes; };
__host__ __device__ inline EllLayout ell_layout(int m, int R, int slots) {
    EllLayout o;
    o.tw = 0;
    o.gw = o.tw + (size_t)m * R * 32 * 8;
    o.tc = o.gw + (size_t)slots * 1024 * 8;
    o.gk = o.tc + (size_t)m * R * 32 * 4;
    o.bytes = o.gk + (size_t)slots * 1024 * 4;
    return o;
}
constexpr int kMaxSlots = 4;        // drives touching one entry of G beyond this: the handle stays with the dense-image kernels

__device__ inline v4d tile_ld(const double* __restrict__ base, int tile, int lane) {   // [tile][pair][lane][2]
    const v2d* p = reinterpret_cast<const v2d*>(base) + tile * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline void tile_st(double* __restrict__ base, int tile, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + tile * 128 + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}
__device__ inline double dot4(const v4d& a, const v4d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
template <int CTRL>
__device__ inline double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// N sums over the 64 lanes at once (DPP row rotations, then four read-lanes): fixed order, bit-reproducible, wave-uniform results
template <int N>
__device__ __forceinline__ void wave_sum_multi(double (&x)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x128>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x124>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x122>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x121>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = (readlane_f64(x[q], 0) + readlane_f64(x[q], 16)) + (readlane_f64(x[q], 32) + readlane_f64(x[q], 48));
}

// (G Y)[I] in the B/D layout from Y's two B/D-layout tiles: sum_K G_A[2 I + K] * Y[K]  (8 MFMAs, two chains)
__device__ __forceinline__ v4d gy_tile(const double* __restrict__ GL, int I, int lane, const v4d& y0, const v4d& y1) {
    const v4d a0 = tile_ld(GL, 2 * I, lane), a1 = tile_ld(GL, 2 * I + 1, lane);
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], y0[0], z, 0, 0, 0);
    v4d c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], y1[0], z, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], y0[kk], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], y1[kk], c1, 0, 0, 0);
    }
    return c0 + c1;
}
// (G Y)^T[J]: lane (g, j) reg r = (G Y)[16 J + j][4 r + g] -- the store layout -- from the same operands the other way round:
// A = Y[K] (B/D-layout registers read as an A operand are the transposed tile), B = G_A[2 J + K]
__device__ __forceinline__ v4d gyT_tile(const double* __restrict__ GL, int J, int lane, const v4d& y0, const v4d& y1) {
    const v4d b0 = tile_ld(GL, 2 * J, lane), b1 = tile_ld(GL, 2 * J + 1, lane);
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[0], b0[0], z, 0, 0, 0);
    v4d c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[0], b1[0], z, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[kk], b0[kk], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[kk], b1[kk], c1, 0, 0, 0);
    }
    return c0 + c1;
}

// lane (g, j) reg r = X[16 J + j][4 r + g] of a column-major 32-row block at p: four whole 128-byte lines per instruction
__device__ inline void store_T32(double* __restrict__ p, const v4d& x, int J, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8m<2>(p + (4 * r + g) * 32 + 16 * J + j, x[r]);
}

// operand-layout (B/D) tile I of a plain row-major 32 x 16 matrix: lane (g, j) reg r = X[16 I + 4 r + g][j]
__device__ __forceinline__ v4d plain_ld(const double* __restrict__ src, int I, int g, int j) {
    v4d x;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = src[(16 * I + 4 * r + g) * kPS + j];
    return x;
}
__device__ __forceinline__ void plain_st(double* __restrict__ dst, int I, int g, int j, const v4d& x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * I + 4 * r + g) * kPS + j] = x[r];
}
// ... and its transpose (store layout): lane (g, j) reg r = X[16 J + j][4 r + g]
__device__ __forceinline__ v4d plain_ld_T(const double* __restrict__ src, int J, int g, int j) {
    v4d x;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = src[(16 * J + j) * kPS + 4 * r + g];
    return x;
}

// load_col16_T (qc_mfma_common.h) in two halves: the request (32 contiguous bytes per lane) and, once the data is needed, the
// transposition between lane groups and registers -- the loader waves must not wait for their knots in front of barrier A
struct Col16Raw { double x0, x1, x2, x3; };
__device__ __forceinline__ Col16Raw col16_request(const double* __restrict__ col, int g) {
    typedef double v2d_ __attribute__((ext_vector_type(2)));
    typedef v2d_ __attribute__((aligned(8))) v2d_u;
    const v2d_u* q = reinterpret_cast<const v2d_u*>(col + 4 * g);
    const v2d_ q0 = q[0], q1 = q[1];
    return Col16Raw{q0[0], q0[1], q1[0], q1[1]};
}
__device__ __forceinline__ v4d col16_finish(Col16Raw r) {
    swap32_f64(r.x0, r.x2);
    swap32_f64(r.x1, r.x3);
    swap16_rows_f64(r.x0, r.x1);
    swap16_rows_f64(r.x2, r.x3);
    return v4d{r.x0, r.x1, r.x2, r.x3};
}

// One row-gathered 32 x 16 product in the OPERAND layout (B/D layout: lane (g, j) reg kk of tile K = X[16 K + 4 kk + g][j], X = G_k S):
// the row's R (weight, column) pairs come from the LDS tables (broadcast reads: four distinct rows per instruction)
template <int R>
__device__ __forceinline__ void gather_rows_operand(const double* __restrict__ tw, const int* __restrict__ tc, const double* __restrict__ src,
                                                    int g, int j, v4d (&out)[2]) {
#pragma unroll
    for (int K = 0; K < 2; ++K) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int a = 16 * K + 4 * kk + g;
            double y = tw[a] * src[tc[a] + j];
#pragma unroll
            for (int q = 1; q < R; ++q) y += tw[q * 32 + a] * src[tc[q * 32 + a] + j];
            out[K][kk] = y;
        }
    }
}
// ... and in the STORE layout (lane (g, j) reg r of tile J = X[16 J + j][4 r + g]): row a = 16 J + j is the lane's own
template <int R>
__device__ __forceinline__ v4d gather_rows_store(const double* __restrict__ tw, const int* __restrict__ tc, const double* __restrict__ src,
                                                 int J, int g, int j) {
    const int a = 16 * J + j;
    v4d out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double y = tw[a] * src[tc[a] + 4 * r + g];
#pragma unroll
        for (int q = 1; q < R; ++q) y += tw[q * 32 + a] * src[tc[q * 32 + a] + 4 * r + g];
        out[r] = y;
    }
    return out;
}

// Two transposed tiles of the SAME columns, rows 0-15 (x0) and 16-31 (x1), as whole 256-byte columns (qc_mfma32_kernels.hip):
// v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the other
__device__ inline void swap16_f64(double a, double b, double& x, double& y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
struct ColumnPair { v4d e, o; };
__device__ inline ColumnPair merge_rows32(const v4d& x0, const v4d& x1) {
    ColumnPair c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double x, y;
        swap16_f64(x0[r], x1[r], x, y);
        c.e[r] = x;
        c.o[r] = y;
    }
    return c;
}
__device__ inline void store_T32_columns(double* __restrict__ p, const ColumnPair& c, int colbase, int g, int j) {
    double* __restrict__ q = p + (colbase + (g & 2)) * 32 + 16 * (g & 1) + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        qc_st8m<2>(q + (4 * r) * 32, c.e[r]);
        qc_st8m<2>(q + (4 * r + 1) * 32, c.o[r]);
    }
}

// The (a_lo, a_hi) entry of the Hessian is c2 h^2 <G_lo G_hi + G_hi G_lo, Q>, Q = M D^T.  With R entries per generator row,
//   (G_x G_y)[a][d] = sum_{q, q'} w_x[q][a] w_y[q'][c] at d = c_y[q'][c], c = c_x[q][a],
// so a pair is a sum of 2 * 32 * R^2 terms (both orders x rows x entry choices), every factor read from the drives' rows in LDS (TW:
// weights, TCr: raw columns; rows with fewer than R entries carry weight 0 and column 0: a zero term).
// Round 5, second form: ONE LANE PER PAIR (at most 36 pairs: lane p < npairs).  A wave takes a block of four rows and adds up, for every
// pair at once, the block's 8 R^2 terms in a fixed order; the eight block sums of a pair meet in LDS and the wave that arrives last adds
// them in block order and stores all pairs with one instruction.  No sum across lanes at all.  (First form of the round: lane = (order,
// row) for five pairs per wave, per-pair constants made from the tables, five 64-lane wave sums per wave -- 420 instructions in every
// wave's tail, 3 us of issue slots per SIMD, 1.7 us of the config-5 launch: profiles/r05_ell32_tail.txt.  Rounds 3 - 4 tabulated the
// merged entries per pair at create time and every workgroup fetched its 28 KB of them from L2: a third of the request phase's bytes.)
// The drives' rows sit kTS entries apart (odd): lanes of different drives read different LDS banks.
template <int R> constexpr int kTSof = R * 32 + 1;
// pair index p = hi (hi + 1) / 2 + lo, lo <= hi  ->  (lo, hi), per lane
__device__ __forceinline__ void pair_decode(int p, int& lo, int& hi) {
    static_assert(kEMax <= 8, "thresholds below: hi < 8");
    hi = (p >= 1) + (p >= 3) + (p >= 6) + (p >= 10) + (p >= 15) + (p >= 21) + (p >= 28);
    lo = p - hi * (hi + 1) / 2;
}
// Sum of the terms of rows [4 blk, 4 blk + 4) of the lane's pair (lo, hi): rows ascending, order (lo, hi) before (hi, lo), q before q'.
// Eight terms at a time (four rows with one entry per row, one row with two): three LDS round trips per batch -- the first factors,
// the second factors at the columns just read, Q's entries -- then the batch's products are added in term order.
template <int R>
__device__ __forceinline__ double pair_block_sum(const double* __restrict__ TW, const int* __restrict__ TCr, const double* __restrict__ Qp, int lo, int hi, int blk) {
    constexpr int kTS = kTSof<R>;
    constexpr int kRows = R == 1 ? 4 : 1;         // rows per batch
    constexpr int kT = 2 * kRows * R * R;         // terms per batch
    static_assert(kT == 8, "eight terms per batch");
    double acc = 0.0;
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += kRows) {
        double w1[2 * kRows * R], w2[kT], qv[kT];
        int c1[2 * kRows * R], d2[kT];
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int x = o ? hi : lo, at = x * kTS + q * 32 + 4 * blk + i0 + i;
                    w1[(2 * i + o) * R + q] = TW[at];
                    c1[(2 * i + o) * R + q] = TCr[at];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 2 * kRows * R; ++f) {
#pragma unroll
            for (int q2 = 0; q2 < R; ++q2) {
                const int y = ((f / R) & 1) ? lo : hi, at = y * kTS + q2 * 32 + c1[f];
                w2[f * R + q2] = TW[at];
                d2[f * R + q2] = TCr[at];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < kT; ++e) qv[e] = Qp[(4 * blk + i0 + e / (2 * R * R)) * kQS + d2[e]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < kT; ++e) acc += (w1[e / R] * w2[e]) * qv[e];
    }
    return acc;
}

// Hand-offs between the waves of a workgroup through counters in LDS (the copy waves of the fused kernel must not stand at a
// workgroup barrier while the knots are still on their way from HBM: their stores are what the launch time follows).  A wave's LDS
// operations execute in order, so data written before the counter is visible to whoever has seen the counter.
enum { FL_U = 0, FL_M, FL_GD, FL_E, FL_Q, FL_COUNT };
__device__ __forceinline__ void flag_signal(int* flags, int f, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&flags[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void flag_wait(int* flags, int f, int count) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < count) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// JAC: F + dF (the I_N (x) B / -I_N (x) F copies by two "copy waves", everything else by the six compute waves);
// HESS: mu_d2F.  JAC && HESS: both in one launch -- every Hessian value by the same operations in the same order as the HESS-only
// instantiation (bit-identical), the knots and multipliers read once.
// SLOTS: how many drives touch one entry of G at most (1 for Pauli strings on distinct qubits; instantiated for 1, 2, 4)
template <int R, bool JAC, bool HESS, bool DIAG, int SLOTS>
__global__ __launch_bounds__(kEThreads, 4) void qc_mfma32_ell_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                     const double* __restrict__ hot_mu0, const char* __restrict__ hot_ell,
                                                                     const int hot_n_int, const int hot_zdim, const int hot_m,
                                                                     const int hot_off_a, const int hot_off_dt, const int hot_off_U,
                                                                     const int hot_f_stride, const int hot_unused, const QcParams Pk,
                                                                     double* __restrict__ F, double* __restrict__ Jv, double* __restrict__ H) {
    constexpr int kFirst = JAC ? 2 : 0;           // first compute wave (waves 0, 1 of a JAC instantiation are the copy waves)
    constexpr int kCW = 8 - kFirst;               // compute waves
    constexpr int kDrivesPerWave = (kEMax + kCW - 1) / kCW;
    // The pair sums (pair_sums below) need Q only.  mu_d2F alone with R = 1 makes them at the very end, behind its drives (measured with the
    // round's first form: summing them before the first drive delays every drive's products by 1 us and the launch by 0.7); the one-call
    // form and R = 2 make them EARLY, as soon as Q is there (their drives have no registers to spare for anything held across them).
    constexpr bool kEarlyPairs = HESS && (JAC || R != 1);      // (the one-call form with the sums at the end instead: the same 36.6 - 36.8 us)
    constexpr int kDF = 2;                        // derivative integrators whose data is requested early and parked in LDS (JAC): the templates have two
    QcKernargTouch<sizeof(QcParams) + 96> touch;
    touch.request();
    __shared__ __attribute__((aligned(16))) double GL[4 * 256];      // G, A-layout tiles 2 I + K
    // the 32 x 16 matrices, plain row-major with rows of kPS doubles -- gather sources and operand tiles alike:
    // M, D, S, E = G M, G D, W = -c1 S + 2 c2 h G D
    __shared__ double Mp[32 * kPS], Dp[32 * kPS], Sp[32 * kPS], Ep[32 * kPS], GDp[32 * kPS], Wp[32 * kPS];
    __shared__ double Qp[32 * kQS];                                  // Q = M D^T
    constexpr int kTS = kTSof<R>;                                    // entries between two drives' rows (odd: see pair_block_sum)
    __shared__ double TW[kEMax * kTS];                               // the drives' rows: weights ...
    __shared__ int TC[kEMax * kTS];                                  // ... and columns (x kPS)
    __shared__ int TCr[kEMax * kTS];                                 // ... and the raw columns (pair sums)
    __shared__ double PS[HESS ? 8 * 64 : 1];                         // the pairs' block sums: [block of four rows][pair]
    __shared__ int ps_count;
    __shared__ int flags[FL_COUNT];
    __shared__ double DerL[2 * kDF * 64];                            // derivative-integrator data (JAC), parked from the first loads to the end
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int m = hot_m;
    const bool ft = hot_off_dt >= 0;
    const int cw = w - kFirst;                    // compute-wave index (negative: copy wave)
    const int b = qc_xcd_remap((int)blockIdx.x, hot_n_int);
    const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
    const double* __restrict__ z1 = z0 + hot_zdim;
    const double* __restrict__ mu = hot_mu0 + (long long)b * hot_f_stride;
    const QcParams& P = Pk;
    // roles among the compute waves: the last four fetch the knots' state tiles and the multipliers and produce the shared products
    const bool f_only = JAC && Jv == nullptr;                        // residuals alone (a line-search trial): the same operations, nothing else
    const bool ld_m = HESS && (cw == kCW - 4 || cw == kCW - 3);      // M tile I = cw - (kCW - 4)
    const bool ld_u = cw == kCW - 2 || cw == kCW - 1;                // U_t, U_t+1 tiles I = cw - (kCW - 2)
    QC_STAMP_DECL;
    QC_STAMP(P, b, lane, 0);
    // mu_d2F alone -- issue priority by phase (round 5).  Two workgroups share a compute unit and the second runs in the issue slots
    // the first one's (older) waves leave (profiles/NOTES.md: it trails by 2.5 - 3 us at every stage).  Up to its drive's stores a wave
    // runs at priority 1, in its tail (pair sums, reductions, scalar stores: nothing the launch waits for) at 0: the second workgroup's
    // products and epilogue then outrank the first one's tail, its blocks leave 0.6 us earlier (8.1 -> 7.5 us) and the launch takes 13.1
    // instead of 13.7 us.  (Priorities by the workgroup's place alone changed nothing; a wave stepping down once its products are
    // issued, or the producers of G D / E stepping up, neither.)
    if constexpr (HESS && !JAC) __builtin_amdgcn_s_setprio(1);

    // ---- phase 0: G (every wave its half tile), the tables, then the state (requested last: G must not wait for it) -----------
    const double h = ft ? z0[hot_off_dt] : opaque_scalar(P.dt_fixed);
    const EllLayout lay = ell_layout(m, R, SLOTS);
    const int npairs = m * (m + 1) / 2;
    Col16Raw raw0 = {0.0, 0.0, 0.0, 0.0}, raw1 = raw0;      // loader waves: M tile, or U_t / U_t+1 tiles (as requested; transposed behind barrier A)
    double mud[2] = {0.0, 0.0};
    double dvx[kDF], dva[kDF], dvb[kDF];
    const bool dfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
    bool rows_fast = JAC && P.n_deriv <= kDF;     // every derivative integrator fits a wave: its rows come from the early requests
#pragma unroll
    for (int d = 0; d < kDF; ++d) rows_fast = rows_fast && P.ddim_i[d] <= 64;
    {
        // half (w & 1) of A-layout tile (w >> 1) of G = G_0 + sum_k a_k G_k: two entries per lane, in image order
        const int e0 = (w >> 1) * 256 + (w & 1) * 128 + 2 * lane;
        // (no branch anywhere in the request phase: where paths meet the compiler cannot count what is outstanding and waits for all)
        v2d Gh = reinterpret_cast<const v2d*>(hot_Gx)[e0 >> 1];
        // the amplitudes by ONE vector load, lane u = a_u, handed to the lanes that need them through the LDS crossbar (a load per
        // entry would be a second round trip behind the plan's); the plan: the few drives that touch an entry, in ascending order
        const double amp = z0[hot_off_a + (lane < m ? lane : 0)];
        const double* __restrict__ gw = reinterpret_cast<const double*>(hot_ell + lay.gw);
        const int* __restrict__ gk = reinterpret_cast<const int*>(hot_ell + lay.gk);
        v2d wv[SLOTS];
        int k0[SLOTS], k1[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            wv[sl] = reinterpret_cast<const v2d*>(gw + (size_t)sl * 1024)[e0 >> 1];
            k0[sl] = gk[(size_t)sl * 1024 + e0];
            k1[sl] = gk[(size_t)sl * 1024 + e0 + 1];
        }
        // drive w's rows -> LDS (one (weight, column) pair per lane; any wave will do, wave k takes drive k)
        const int trow = (w < m ? w : 0) * R * 32 + (lane < 32 * R ? lane : 0);
        const double twv = reinterpret_cast<const double*>(hot_ell + lay.tw)[trow];
        const int tcv = reinterpret_cast<const int*>(hot_ell + lay.tc)[trow];
        // ---- everything G needs has been requested; what follows is requested behind it and waited for later (loads return in
        //      order: the assembly below waits for its own only -- the scheduling fences keep the compiler from mixing the two groups)
        __builtin_amdgcn_sched_barrier(0);
        // EVERY wave issues the same requests here, without a branch (a wave that has no use for one reads a harmless address):
        // behind a branch the compiler cannot count what is outstanding where the paths meet, and waits for everything.
        {
            const int Iu = cw - (kCW - 2), Im = cw - (kCW - 4);
            const double* pa = ld_u ? z0 + hot_off_U + j * 32 + 16 * Iu : (ld_m ? mu + j * 32 + 16 * Im : z0);
            const double* pb = ld_u ? z1 + hot_off_U + j * 32 + 16 * Iu : z0;
            raw0 = col16_request(pa, ld_u || ld_m ? g : 0);
            raw1 = col16_request(pb, ld_u ? g : 0);
        }
        if constexpr (HESS) {             // derivative integrators: d2/d(dx_i) dh = -mu_i, requested here, written at the end (wave cw = 0)
#pragma unroll
            for (int d = 0; d < 2; ++d) mud[d] = mu[dfast ? P.drow[d] + (lane < P.ddim_i[d] ? lane : 0) : 0];
        }
        if constexpr (JAC) {              // derivative integrators' rows: dx_t, x_t, x_t+1 (every wave asks, wave cw = 1 uses them)
#pragma unroll
            for (int d = 0; d < kDF; ++d) {   // (unused slots have zero offsets and dimensions: the requests stay in bounds)
                const int i = (rows_fast && lane < P.ddim_i[d]) ? lane : 0;
                dvx[d] = z0[P.dx_off[d] + i];
                dva[d] = z0[P.x_off[d] + i];
                dvb[d] = z1[P.x_off[d] + i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) Gh += v2d{__shfl(amp, k0[sl]), __shfl(amp, k1[sl])} * wv[sl];
        if (w < m && lane < 32 * R && !f_only) { TW[w * kTS + lane] = twv; TC[w * kTS + lane] = tcv; TCr[w * kTS + lane] = tcv / kPS; }
        reinterpret_cast<v2d*>(GL)[e0 >> 1] = Gh;
        if (tid < FL_COUNT) flags[tid] = 0;
        if (tid == FL_COUNT) ps_count = 0;
    }
    touch.consume();
    double* __restrict__ Hb = HESS ? H + (size_t)b * P.H_stride + P.H_off : nullptr;
    double* __restrict__ Jb = (JAC && !f_only) ? Jv + (size_t)b * P.J_stride + P.J_off : nullptr;
    double* __restrict__ Fb = (JAC && F) ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    const double c1 = P.c[1], c2 = P.c[2];
    QC_STAMP(P, b, lane, 1);
#ifdef QC_ELL_STAMP_BARRIER
    // diagnostic build (profiles/stamps_ell32_barrier.py): when does EACH wave arrive at barrier A?  slots 0 - 7 = waves 0 - 7
    if constexpr (DIAG) { if (P.stamps != nullptr && lane == 0) P.stamps[(size_t)b * 16 + w] = qc_ts_[1]; }
#endif
    __syncthreads();                  // G, the tables and the zeroed counters; the state is still on its way
    const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;
    QC_STAMP(P, b, lane, 2);
    // The pair sums (pair_block_sum above): this wave's blocks of rows for every pair (lane = pair), handed over through LDS; the wave
    // that arrives last adds the eight block sums of each pair in block order and stores the pairs' run.  Needs Q (FL_Q).
    auto pair_sums = [&]() {
        int plo, phi;
        pair_decode(lane < npairs ? lane : 0, plo, phi);       // (lanes beyond the last pair repeat pair 0: never stored)
        flag_wait(flags, FL_Q, 1);
        for (int blk = cw; blk < 8; blk += kCW) PS[blk * 64 + lane] = pair_block_sum<R>(TW, TCr, Qp, plo, phi, blk);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&ps_count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == kCW - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            double sum = PS[lane];
#pragma unroll
            for (int blk = 1; blk < 8; ++blk) sum += PS[blk * 64 + lane];
            if (lane < npairs) Hb[P.ho_aa + lane] = hc2 * sum;
        }
    };

    if (JAC && cw < 0) {
        if (f_only) return;
        // ================= copy wave: block row I of B^T and F^T (tiles (I, 0), (I, 1)), N copies each ======================
        // B^T = I - hc1 G^T + hc2 (G^2)^T, -F^T = -(I + hc1 G^T + hc2 (G^2)^T); with G^T = -G the operands of
        // (G^2)^T[I][J] = sum_K G^T[I][K] G^T[K][J] are A-layout tiles of G both: -(G_A[I][K] as A) x (G_A[J][K] as B)
        const int I = w;
        const v4d IdB = identity_B(g, j);
        const v4d aI0 = tile_ld(GL, 2 * I, lane), aI1 = tile_ld(GL, 2 * I + 1, lane);
        v4d Fm[2], Bm[2];
#pragma unroll
        for (int Jt = 0; Jt < 2; ++Jt) {
            const v4d bJ0 = tile_ld(GL, 2 * Jt, lane), bJ1 = tile_ld(GL, 2 * Jt + 1, lane);
            const v4d z = {0.0, 0.0, 0.0, 0.0};
            v4d a0 = z, a1 = z;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI0[kk], bJ0[kk], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI1[kk], bJ1[kk], a1, 0, 0, 0);
            }
            const v4d G2T = -(a0 + a1);
            const v4d GT = tile_ld(GL, 2 * Jt + I, lane);     // B/D layout of (G^T)[I][Jt] = A layout of G[Jt][I]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ev = (I == Jt ? IdB[r] : 0.0) + hc2 * G2T[r];
                Fm[Jt][r] = -(ev + hc1 * GT[r]);
                Bm[Jt][r] = ev - hc1 * GT[r];
            }
        }
        // lane (g, j) reg r = B^T[16 I + 4 r + g][16 Jt + j] = B[16 Jt + j][16 I + 4 r + g]
        double* __restrict__ pF = Jb + P.jo_F;
        double* __restrict__ pB = Jb + P.jo_B;
        const ColumnPair Fc = merge_rows32(Fm[0], Fm[1]), Bc = merge_rows32(Bm[0], Bm[1]);
        QC_STAMP(P, b, lane, 3);
        const int ncop = P.copies;    // N copies of each block; 1 when the host path asks for the compact form
        for (int q = 0; q < ncop; ++q) {
            store_T32_columns(pF + q * 1024, Fc, 16 * I, g, j);
            store_T32_columns(pB + q * 1024, Bc, 16 * I, g, j);
        }
        QC_STAMP(P, b, lane, 4);
    } else {
        // ================= compute wave ====================================================================================
        if (ld_m) {
            const int I = cw - (kCW - 4);
            plain_st(Mp, I, g, j, col16_finish(raw0));                         // lane (g, j) reg r = M[16 I + 4 r + g][j]
            flag_signal(flags, FL_M, lane);
        }
        if (ld_u) {
            const int I = cw - (kCW - 2);
            const v4d st0 = col16_finish(raw0), st1 = col16_finish(raw1);
            const v4d dd = st1 - st0, ss = st1 + st0;
            plain_st(Dp, I, g, j, dd);
            plain_st(Sp, I, g, j, ss);
            flag_signal(flags, FL_U, lane);
        }
        if (JAC && cw == 1 && rows_fast) {
#pragma unroll
            for (int d = 0; d < kDF; ++d) {
                DerL[(2 * d) * 64 + lane] = dvx[d];
                DerL[(2 * d + 1) * 64 + lane] = dvb[d] - dva[d];
            }
        }
        flag_wait(flags, FL_U, 2);
        if constexpr (HESS) flag_wait(flags, FL_M, 2);
        QC_STAMP(P, b, lane, 3);
        // ---- the shared products (their producers first: the other waves wait for them in their first drive's epilogue) ----
        if (cw == kCW - 2) {              // G D (both tiles), W = -c1 S + 2 c2 h G D
            const v4d d0 = plain_ld(Dp, 0, g, j), d1 = plain_ld(Dp, 1, g, j);
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const v4d gd = gy_tile(GL, I, lane, d0, d1);
                plain_st(GDp, I, g, j, gd);
                if constexpr (HESS) plain_st(Wp, I, g, j, (-c1) * plain_ld(Sp, I, g, j) + c2h2 * gd);
            }
            flag_signal(flags, FL_GD, lane);
        }
        if (HESS && cw == kCW - 4) {      // E = G M
            const v4d m0 = plain_ld(Mp, 0, g, j), m1 = plain_ld(Mp, 1, g, j);
#pragma unroll
            for (int I = 0; I < 2; ++I) plain_st(Ep, I, g, j, gy_tile(GL, I, lane, m0, m1));
            flag_signal(flags, FL_E, lane);
        }
        if (HESS && cw == kCW - 3) {      // Q[16 I + i][16 J + j] = sum_c M[16 I + i][c] D[16 J + j][c]
#pragma unroll
            for (int IJ = 0; IJ < 4; ++IJ) {
                const int I = IJ >> 1, J = IJ & 1;
                v4d q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)   // A: lane (g, i) = M[16 I + i][4 kk + g];  B: lane (g, j) = D[16 J + j][4 kk + g]
                    q = __builtin_amdgcn_mfma_f64_16x16x4f64(Mp[(16 * I + j) * kPS + 4 * kk + g], Dp[(16 * J + j) * kPS + 4 * kk + g], q, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Qp[(16 * I + 4 * r + g) * kQS + 16 * J + j] = q[r];
            }
            flag_signal(flags, FL_Q, lane);
        }
        if constexpr (kEarlyPairs) pair_sums();      // early: as soon as Q is there (see kEarlyPairs)
        // ---- this wave's drives ---------------------------------------------------------------------------------------------
        double pv[kDrivesPerWave];
#pragma unroll
        for (int t = 0; t < kDrivesPerWave; ++t) {
            pv[t] = 0.0;
            const int k = cw + kCW * t;
            if (k < m && !f_only) {
                const double* __restrict__ tw = TW + k * kTS;
                const int* __restrict__ tc = TC + k * kTS;
                v4d Vk[2], Yk[2], TV[2], TY[2];
                gather_rows_operand<R>(tw, tc, Dp, g, j, Vk);                   // V_k = G_k D
                if constexpr (HESS) gather_rows_operand<R>(tw, tc, Mp, g, j, Yk);   // Y_k = G_k M
                {   // (G V_k)^T and (G Y_k)^T: A = the gathered tiles (operand-layout registers read as an A operand are the transposed
                    // tile), B = G_A[2 J + K]; the chains of the output tiles interleave
                    const v4d z = {0.0, 0.0, 0.0, 0.0};
                    v4d v0 = z, v1 = z, y0 = z, y1 = z;
#pragma unroll
                    for (int K = 0; K < 2; ++K) {
                        const v4d b0 = tile_ld(GL, K, lane), b1 = tile_ld(GL, 2 + K, lane);      // G_A tiles (J = 0, K), (J = 1, K)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            if constexpr (JAC) {
                                v0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vk[K][kk], b0[kk], v0, 0, 0, 0);
                                v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vk[K][kk], b1[kk], v1, 0, 0, 0);
                            }
                            if constexpr (HESS) {
                                y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Yk[K][kk], b0[kk], y0, 0, 0, 0);
                                y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Yk[K][kk], b1[kk], y1, 0, 0, 0);
                            }
                        }
                    }
                    TV[0] = v0; TV[1] = v1; TY[0] = y0; TY[1] = y1;
                }
                if (t == 0) QC_STAMP(P, b, lane, 4);
                // mu_d2F alone with one entry per row: the drive's first store is what the launch follows (HBM is busy from it to the end), and
                // the epilogue in front of it was nine LDS round trips one after the other (the compiler keeps the register count low and
                // reuses the destinations).  Here: the two store-layout rows of the drive are fetched while the products run, the eight
                // gathered values per lane go out as one batch behind the flags, the blocks are stored, and the (a_k, h) partial -- which only
                // the closing reduction needs -- comes last.  Same operations on the same values as the other order: the same bits.
                constexpr bool kStoresFirst = HESS && !JAC && R == 1;
                double tws[2] = {0.0, 0.0};
                int tcs[2] = {0, 0};
                if constexpr (kStoresFirst) {
#pragma unroll
                    for (int J = 0; J < 2; ++J) { tws[J] = tw[16 * J + j]; tcs[J] = tc[16 * J + j]; }
                }
                flag_wait(flags, FL_GD, 1);
                if constexpr (HESS) flag_wait(flags, FL_E, 1);
                if (t == 0) QC_STAMP(P, b, lane, 5);
                auto partial_ah = [&]() {     // (a_k, h), per lane: the last use of the gathered operand tiles
                    if (ft)
                        pv[t] = -(dot4(Yk[0], plain_ld(Wp, 0, g, j)) + dot4(Yk[1], plain_ld(Wp, 1, g, j))) -
                                c2h2 * (dot4(plain_ld(Ep, 0, g, j), Vk[0]) + dot4(plain_ld(Ep, 1, g, j), Vk[1]));
                };
                if constexpr (HESS && !kStoresFirst) partial_ah();
                if constexpr (JAC) {      // d/da_k = -c1 h G_k S + c2 h^2 (G_k (G D) + G (G_k D)), transposed for the store
                    double* __restrict__ pa = Jb + P.jo_a + (size_t)k * 512;
                    v4d yT[2];
#pragma unroll
                    for (int J = 0; J < 2; ++J)
                        yT[J] = (-hc1) * gather_rows_store<R>(tw, tc, Sp, J, g, j) + hc2 * (gather_rows_store<R>(tw, tc, GDp, J, g, j) + TV[J]);
                    store_T32_columns(pa, merge_rows32(yT[0], yT[1]), 0, g, j);
                }
                if constexpr (HESS) {
                    double* __restrict__ pUa = Hb + P.ho_Ua + (size_t)k * 512;
                    double* __restrict__ paU = Hb + P.ho_aU + (size_t)k * 512;
                    v4d lo[2], hi[2];
                    if constexpr (kStoresFirst) {
                        v4d ms[2], es[2];         // M and E at the drive's columns, the lane's rows: every read requested before the first use
#pragma unroll
                        for (int J = 0; J < 2; ++J) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { ms[J][r] = Mp[tcs[J] + 4 * r + g]; es[J][r] = Ep[tcs[J] + 4 * r + g]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int J = 0; J < 2; ++J) {
                            const v4d yt = tws[J] * ms[J];                                  // Y_k^T      (gather_rows_store<1>'s products)
                            const v4d get = tws[J] * es[J];                                 // (G_k E)^T
                            const v4d lin = hc1 * yt, qd = hc2 * (get + TY[J]);
                            lo[J] = lin - qd;
                            hi[J] = lin + qd;
                        }
                    } else {
#pragma unroll
                        for (int J = 0; J < 2; ++J) {
                            const v4d yt = gather_rows_store<R>(tw, tc, Mp, J, g, j);       // Y_k^T
                            const v4d get = gather_rows_store<R>(tw, tc, Ep, J, g, j);      // (G_k E)^T
                            const v4d lin = hc1 * yt, qd = hc2 * (get + TY[J]);
                            lo[J] = lin - qd;
                            hi[J] = lin + qd;
                        }
                    }
                    store_T32_columns(pUa, merge_rows32(lo[0], lo[1]), 0, g, j);        // whole 256-byte columns per piece
                    store_T32_columns(paU, merge_rows32(hi[0], hi[1]), 0, g, j);
                }
                if constexpr (kStoresFirst) partial_ah();
                if (t == 0) QC_STAMP(P, b, lane, 6);
            }
        }
        if constexpr (HESS && !JAC) __builtin_amdgcn_s_setprio(0);      // the tail (see the kernel's entry)
        if constexpr (HESS) {     // the (a_k, h) of this wave's drives; the pair sums where they have not been made early
            if (ft) {
                wave_sum_multi<kDrivesPerWave>(pv);
                if (lane == 0) {
#pragma unroll
                    for (int t = 0; t < kDrivesPerWave; ++t) {
                        const int k = cw + kCW * t;
                        if (k < m) Hb[P.ho_ah + k] = pv[t];
                    }
                }
            }
            if constexpr (!kEarlyPairs) pair_sums();
        }
        // ---- the shared blocks ---------------------------------------------------------------------------------------------------
        if (JAC && cw == kCW - 1) {       // residual and d/dh: D - hc1 G S + hc2 G (G D);  -c1 G S + 2 c2 h G (G D)
            flag_wait(flags, FL_GD, 1);
            const v4d gd0 = plain_ld(GDp, 0, g, j), gd1 = plain_ld(GDp, 1, g, j);
            const v4d s0 = plain_ld(Sp, 0, g, j), s1 = plain_ld(Sp, 1, g, j);
            v4d res[2], dh[2];
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const v4d gs = gyT_tile(GL, J, lane, s0, s1);                   // (G S)^T[J]
                const v4d ggd = gyT_tile(GL, J, lane, gd0, gd1);                // (G (G D))^T[J]
                res[J] = plain_ld_T(Dp, J, g, j) - hc1 * gs + hc2 * ggd;        // D^T[J] - ...
                dh[J] = (-c1) * gs + c2h2 * ggd;
            }
            if (Fb) store_T32_columns(Fb, merge_rows32(res[0], res[1]), 0, g, j);
            if (ft && !f_only) store_T32_columns(Jb + P.jo_h, merge_rows32(dh[0], dh[1]), 0, g, j);
        }
        // mu_d2F alone: on the wave that shares its SIMD with a loader wave and carries no shared product (waves w and w + 4 share a
        // SIMD: E sits on wave 4 beside wave 0, Q on 5 beside 1, G D on 6 beside 2) -- 48 MFMAs per SIMD everywhere instead of 64 / 32:
        // 12.7 instead of 13.2 us (on wave 1, beside Q: no gain).  The one-call form keeps it with E's producer.
        constexpr int kUhWave = JAC ? kCW - 4 : 3;
        if (HESS && ft && cw == kUhWave) {    // (U_t, h)^T and (h, U_t+1)^T: c1 E -+ 2 c2 h G E
            if (kUhWave != kCW - 4) flag_wait(flags, FL_E, 1);
            const v4d e0 = plain_ld(Ep, 0, g, j), e1 = plain_ld(Ep, 1, g, j);
            v4d lo[2], hi[2];
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const v4d ge = gyT_tile(GL, J, lane, e0, e1);                   // (G E)^T[J]
                const v4d et = plain_ld_T(Ep, J, g, j);                         // E^T[J]
                lo[J] = c1 * et - c2h2 * ge;
                hi[J] = c1 * et + c2h2 * ge;
            }
            store_T32_columns(Hb + P.ho_Uh, merge_rows32(lo[0], lo[1]), 0, g, j);
            store_T32_columns(Hb + P.ho_hU, merge_rows32(hi[0], hi[1]), 0, g, j);
        }
        if (HESS && ft && cw == kCW - 3) {    // (h, h) = -2 c2 <E, G D>
            flag_wait(flags, FL_GD, 1);
            flag_wait(flags, FL_E, 1);
            double s1[1] = {dot4(plain_ld(Ep, 0, g, j), plain_ld(GDp, 0, g, j)) + dot4(plain_ld(Ep, 1, g, j), plain_ld(GDp, 1, g, j))};
            wave_sum_multi<1>(s1);
            if (lane == 0) Hb[P.ho_hh] = -2.0 * c2 * s1[0];
        }
        if (HESS && cw == 0) {            // derivative integrators' (dx, h) entries and the alignment padding
            if (dfast) {
                int o = P.ho_d;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    if (lane < P.ddim_i[d]) Hb[o + lane] = -mud[d];
                    o += P.ddim_i[d];
                }
                for (int i = lane; i < P.h_pad; i += 64) Hb[P.hess_nnz + i] = 0.0;
            } else {
                qc_hess_tail(Pk, mu, Hb, lane, 64);
            }
        }
        if (JAC && cw == 1) {             // derivative-integrator rows and their Jacobian entries
            if (rows_fast) {
                int jo = P.jo_d;
#pragma unroll
                for (int d = 0; d < kDF; ++d) {
                    if (d < P.n_deriv) {
                        const int dim = P.ddim_i[d], r0 = P.drow[d];
                        if (lane < dim) {
                            const double dx = DerL[(2 * d) * 64 + lane], df = DerL[(2 * d + 1) * 64 + lane];
                            if (Fb) Fb[r0 + lane] = df - h * dx;
                            if (Jb) {
                                Jb[jo + lane] = -1.0;
                                Jb[jo + dim + lane] = 1.0;
                                Jb[jo + 2 * dim + lane] = -h;
                                if (ft) Jb[jo + 3 * dim + lane] = -dx;
                            }
                        }
                        jo += (ft ? 4 : 3) * dim;
                    }
                }
            } else {
                deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, false);
            }
        }
        QC_STAMP(P, b, lane, 7);
    }
#ifndef QC_ELL_STAMP_BARRIER
    if constexpr (DIAG) {
        // slots 0-7: the first compute wave; 8-15: wave 0 of a JAC instantiation (copy wave), else the last compute wave
        const bool first = cw == 0, second = JAC ? w == 0 : w == 7;
        if (P.stamps != nullptr && lane == 0 && (first || second)) {
#pragma unroll
            for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + (first ? 0 : 8) + k_] = qc_ts_[k_];
        }
    }
#endif
}

}  // namespace

// ---- host side: are the drives sparse enough, and the tables ---------------------------------------------------------
// G: (m + 1) column-major n x n matrices, index 0 = drift (dense is fine: only the DRIVES are row-gathered).
// Returns the ELL width R (1 or 2) and fills `blob` / `slots`, or 0 when this kernel does not serve the handle.
int qc_mfma32_ell_build(const QcParams& P, const double* G, std::vector<char>* blob, int* slots_out) {
    if (P.integrator != QC_PADE || P.p != 2 || P.n != 32 || P.nc != 16 || !P.antisym || P.m < 1 || P.m > kEMax || P.hess_nnz == 0) return 0;
    const int n = 32, m = P.m;
    auto Gk = [&](int k, int a, int c) { return G[(size_t)(k + 1) * n * n + (size_t)c * n + a]; };   // drive k, row a, column c
    int R = 0;
    for (int k = 0; k < m; ++k)
        for (int a = 0; a < n; ++a) {
            int cnt = 0;
            for (int c = 0; c < n; ++c) cnt += Gk(k, a, c) != 0.0;
            R = std::max(R, cnt);
        }
    if (R < 1 || R > 2) return 0;
    // how many drives touch one entry of G
    int slots = 0;
    for (int a = 0; a < n; ++a)
        for (int c = 0; c < n; ++c) {
            int cnt = 0;
            for (int k = 0; k < m; ++k) cnt += Gk(k, a, c) != 0.0;
            slots = std::max(slots, cnt);
        }
    if (slots > kMaxSlots) return 0;                     // (e.g. five diagonal drives: the dense-image kernels serve the handle)
    slots = slots <= 1 ? 1 : (slots <= 2 ? 2 : 4);      // the instantiated plan depths; unused slots carry weight 0
    const EllLayout lay = ell_layout(m, R, slots);
    blob->assign(lay.bytes, 0);
    double* tw = reinterpret_cast<double*>(blob->data() + lay.tw);
    double* gw = reinterpret_cast<double*>(blob->data() + lay.gw);
    int* tc = reinterpret_cast<int*>(blob->data() + lay.tc);
    int* gk = reinterpret_cast<int*>(blob->data() + lay.gk);
    for (int k = 0; k < m; ++k)
        for (int a = 0; a < n; ++a) {
            int q = 0;
            for (int c = 0; c < n; ++c) {
                const double v = Gk(k, a, c);
                if (v == 0.0) continue;
                tw[(k * R + q) * 32 + a] = v;
                tc[(k * R + q) * 32 + a] = c * kPS;
                ++q;
            }
            // (rows with fewer than R entries keep weight 0 and column 0: a valid address, a zero term)
        }
    // assembly plan in the order of the A-layout image (qc_mfma32_pack_G): entry [tile = 2 I + K][pair][lane = 16 g + i][e] = X[16 I + i][16 K + 4 (2 pair + e) + g]
    for (int tile = 0; tile < 4; ++tile)
        for (int pr = 0; pr < 2; ++pr)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 2; ++e) {
                    const int gg = l >> 4, i = l & 15, kk = 2 * pr + e;
                    const int a = 16 * (tile >> 1) + i, c = 16 * (tile & 1) + 4 * kk + gg;
                    const int idx = (tile * 2 + pr) * 128 + l * 2 + e;
                    int sl = 0;
                    for (int k = 0; k < m; ++k)
                        if (Gk(k, a, c) != 0.0) { gw[(size_t)sl * 1024 + idx] = Gk(k, a, c); gk[(size_t)sl * 1024 + idx] = k; ++sl; }
                }
    *slots_out = slots;
    return R;
}

#define QC_ELL_ARGS(F_, J_, H_) P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu ? dMu + P.t_begin * P.F_stride + P.F_off : nullptr, (const char*)P.ell, \
                    P.n_int, P.zdim, P.m, P.off_a, P.off_dt, P.off_U, (int)P.F_stride, 0, P, F_, J_, H_
#define QC_ELL_GO(R_, JAC_, HESS_, DIAG_, S_, F_, J_, H_) \
    hipLaunchKernelGGL((qc_mfma32_ell_kernel<R_, JAC_, HESS_, DIAG_, S_>), dim3(P.n_int), dim3(kEThreads), 0, st, QC_ELL_ARGS(F_, J_, H_))
#define QC_ELL_LAUNCH(JAC_, HESS_, F_, J_, H_)                                                                                    \
    do {                                                                                                                          \
        const int S = P.ell_slots;                                                                                                \
        if (P.stamps != nullptr && P.ell_R == 1 && S == 1) QC_ELL_GO(1, JAC_, HESS_, true, 1, F_, J_, H_);   /* diagnostic timeline */ \
        else if (P.ell_R == 1 && S == 1) QC_ELL_GO(1, JAC_, HESS_, false, 1, F_, J_, H_);                                         \
        else if (P.ell_R == 1 && S == 2) QC_ELL_GO(1, JAC_, HESS_, false, 2, F_, J_, H_);                                         \
        else if (P.ell_R == 1) QC_ELL_GO(1, JAC_, HESS_, false, 4, F_, J_, H_);                                                   \
        else if (S == 1) QC_ELL_GO(2, JAC_, HESS_, false, 1, F_, J_, H_);                                                         \
        else if (S == 2) QC_ELL_GO(2, JAC_, HESS_, false, 2, F_, J_, H_);                                                         \
        else QC_ELL_GO(2, JAC_, HESS_, false, 4, F_, J_, H_);                                                                     \
    } while (0)

hipError_t qc_launch_mfma32_ell_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    QC_ELL_LAUNCH(false, true, nullptr, nullptr, dH);
    return hipGetLastError();
}
// F + dF, or the residuals alone (dJ == NULL): the same instantiation, so the residuals are the same to the bit
hipError_t qc_launch_mfma32_ell_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    const double* dMu = nullptr;
    QC_ELL_LAUNCH(true, false, dF, dJ, nullptr);
    return hipGetLastError();
}
// F + dF + mu_d2F in one launch
hipError_t qc_launch_mfma32_ell_fused(const QcParams& P, const double* dZ, const double* dMu, double* dF, double* dJ, double* dH, hipStream_t st) {
    QC_ELL_LAUNCH(true, true, dF, dJ, dH);
    return hipGetLastError();
}
