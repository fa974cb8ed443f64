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

namespace {

using namespace qc_mfma;

constexpr int kEThreads = 512;
constexpr int kEMax = 8;            // drives (one wave each)
constexpr int kPS = 17;             // row stride of the plain 32 x 16 LDS copies (doubles)
constexpr int kQS = 33;             // row stride of the plain 32 x 32 Gram matrix

// Byte offsets of the tables inside the device blob (host and device agree through this one function).
//   tw / tc    [k][q][32]     row a of drive k: weight and column (pre-multiplied by kPS) of its q-th entry
//   pw / po    [pair][L]      entries of G_i G_k + G_k G_i: weight, offset into the plain Gram matrix (row * kQS + col)
//   gw / gk    [slot][1024]   assembly plan of G = G_0 + sum_k a_k G_k in the order of the A-layout image: the slot-th drive that
//                             touches the entry (ascending k, so the sum has the order of the dense one) and its weight; k = 0, w = 0
//                             where fewer drives do
struct EllLayout { size_t tw, pw, gw, tc, po, gk, bytes; };
__host__ __device__ inline EllLayout ell_layout(int m, int R, int L, int slots) {
    const size_t npairs = (size_t)m * (m + 1) / 2;
    EllLayout o;
    o.tw = 0;
    o.pw = o.tw + (size_t)m * R * 32 * 8;
    o.gw = o.pw + npairs * L * 8;
    o.tc = o.gw + (size_t)slots * 1024 * 8;
    o.po = o.tc + (size_t)m * R * 32 * 4;
    o.gk = o.po + npairs * L * 4;
    o.bytes = o.gk + (size_t)slots * 1024 * 4;
    return o;
}
constexpr int kMaxSlots = 4;        // drives touching one entry of G beyond this: the dense images assemble G

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

template <int R, bool DIAG>
__global__ __launch_bounds__(kEThreads, 2) void qc_mfma32_ell_hess_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                          const double* __restrict__ hot_mu0, const char* __restrict__ hot_ell,
                                                                          const int hot_n_int, const int hot_zdim, const int hot_m,
                                                                          const int hot_off_a, const int hot_off_dt, const int hot_off_U,
                                                                          const int hot_f_stride, const int hot_slots, const QcParams Pk,
                                                                          double* __restrict__ H) {
    constexpr int L = R == 1 ? 64 : 256;          // padded length of a pair list (fixed by R: at most 32 * 2 R^2 entries)
    constexpr int kPairsPerWave = (kEMax * (kEMax + 1) / 2 + kEMax - 1) / kEMax;    // 5
    QcKernargTouch<sizeof(QcParams) + 96> touch;
    touch.request();
    __shared__ __attribute__((aligned(16))) double GL[4 * 256];      // G, A-layout tiles 2 I + K
    __shared__ __attribute__((aligned(16))) double MT[2 * 256];      // M, B/D-layout tiles
    __shared__ __attribute__((aligned(16))) double DT[2 * 256];      // D
    __shared__ __attribute__((aligned(16))) double ST[2 * 256];      // S
    __shared__ __attribute__((aligned(16))) double ET[2 * 256];      // E = G M
    __shared__ __attribute__((aligned(16))) double GDT[2 * 256];     // G D
    __shared__ __attribute__((aligned(16))) double WT[2 * 256];      // -c1 S + 2 c2 h G D
    __shared__ double Mp[32 * kPS], Dp[32 * kPS], Ep[32 * kPS];      // plain row-major copies: the gather sources
    __shared__ double Qp[32 * kQS];                                  // Q = M D^T
    __shared__ double TW[kEMax * R * 32];                            // the drives' rows: weights ...
    __shared__ int TC[kEMax * R * 32];                               // ... and columns (x kPS)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int m = hot_m;
    const bool ft = hot_off_dt >= 0;
    const bool drive = w < m;
    const int b = qc_xcd_remap((int)blockIdx.x, hot_n_int);
    const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
    const double* __restrict__ z1 = z0 + hot_zdim;
    const double* __restrict__ mu = hot_mu0 + (long long)b * hot_f_stride;
    const QcParams& P = Pk;
    QC_STAMP_DECL;
    QC_STAMP(P, b, lane, 0);

    // ---- phase 0 ------------------------------------------------------------------------------------------------
    const double h = ft ? z0[hot_off_dt] : opaque_scalar(P.dt_fixed);
    const EllLayout lay = ell_layout(m, R, L, hot_slots);
    const int npairs = m * (m + 1) / 2;
    double pwv[kPairsPerWave];        // R = 1: this wave's pair entries, one per lane and pair, requested with everything else
    int pov[kPairsPerWave];
    {
        // half (w & 1) of A-layout tile (w >> 1) of G = G_0 + sum_k a_k G_k: two entries per lane, in image order
        const int e0 = (w >> 1) * 256 + (w & 1) * 128 + 2 * lane;
        v2d Gh = reinterpret_cast<const v2d*>(hot_Gx)[e0 >> 1];
        if (hot_slots > 0) {          // sparse drives: the few drives that touch an entry, in ascending order (the dense sum's order)
            const double* __restrict__ gw = reinterpret_cast<const double*>(hot_ell + lay.gw);
            const int* __restrict__ gk = reinterpret_cast<const int*>(hot_ell + lay.gk);
            // (the amplitudes by ONE vector load, lane u = a_u, handed to the lanes that need them through the LDS crossbar: a load
            //  per entry would be a second round trip behind the plan's)
            const double amp = z0[hot_off_a + (lane < m ? lane : 0)];
            v2d wv[kMaxSlots];
            int k0[kMaxSlots], k1[kMaxSlots];
#pragma unroll
            for (int sl = 0; sl < kMaxSlots; ++sl) {
                const int so = sl < hot_slots ? sl : 0;
                wv[sl] = reinterpret_cast<const v2d*>(gw + (size_t)so * 1024)[e0 >> 1];
                k0[sl] = gk[(size_t)so * 1024 + e0];
                k1[sl] = gk[(size_t)so * 1024 + e0 + 1];
            }
#pragma unroll
            for (int sl = 0; sl < kMaxSlots; ++sl)
                if (sl < hot_slots) Gh += v2d{__shfl(amp, k0[sl]), __shfl(amp, k1[sl])} * wv[sl];
        } else {
            const v2d* __restrict__ ab = reinterpret_cast<const v2d*>(hot_Gx) + (e0 >> 1);
            v2d img[kEMax];
            double ak[kEMax];
#pragma unroll
            for (int u = 0; u < kEMax; ++u) img[u] = ab[(size_t)(u < m ? u + 1 : 0) * 512];
#pragma unroll
            for (int u = 0; u < kEMax; ++u) ak[u] = z0[hot_off_a + (u < m ? u : 0)];
#pragma unroll
            for (int u = 0; u < kEMax; ++u) Gh += (u < m ? ak[u] : 0.0) * img[u + 0];
        }
        // this drive's rows -> LDS (one (weight, column) pair per lane)
        double twv = 0.0;
        int tcv = 0;
        if (drive && lane < 32 * R) {
            twv = reinterpret_cast<const double*>(hot_ell + lay.tw)[w * R * 32 + lane];
            tcv = reinterpret_cast<const int*>(hot_ell + lay.tc)[w * R * 32 + lane];
        }
        if constexpr (R == 1) {
            const double* __restrict__ pw = reinterpret_cast<const double*>(hot_ell + lay.pw);
            const int* __restrict__ po = reinterpret_cast<const int*>(hot_ell + lay.po);
#pragma unroll
            for (int t = 0; t < kPairsPerWave; ++t) {
                const int p = w + m * t;
                const bool ok = drive && p < npairs;
                pwv[t] = ok ? pw[(size_t)p * L + lane] : 0.0;
                pov[t] = ok ? po[(size_t)p * L + lane] : 0;
            }
        }
        if (w >= 4) {
            const int I = w & 1;
            if (w < 6) {
                const v4d mt = load_col16_T(mu + j * 32 + 16 * I, g);        // lane (g, j) reg r = M[16 I + 4 r + g][j]
                tile_st(MT, I, lane, mt);
#pragma unroll
                for (int r = 0; r < 4; ++r) Mp[(16 * I + 4 * r + g) * kPS + j] = mt[r];
            } else {
                const v4d u0 = load_col16_T(z0 + hot_off_U + j * 32 + 16 * I, g);
                const v4d u1 = load_col16_T(z1 + hot_off_U + j * 32 + 16 * I, g);
                const v4d dd = u1 - u0;
                tile_st(ST, I, lane, u1 + u0);
                tile_st(DT, I, lane, dd);
#pragma unroll
                for (int r = 0; r < 4; ++r) Dp[(16 * I + 4 * r + g) * kPS + j] = dd[r];
            }
        }
        if (drive && lane < 32 * R) { TW[w * R * 32 + lane] = twv; TC[w * R * 32 + lane] = tcv; }
        reinterpret_cast<v2d*>(GL)[e0 >> 1] = Gh;
    }
    touch.consume();
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double c1 = P.c[1], c2 = P.c[2];
    QC_STAMP(P, b, lane, 1);
    __syncthreads();
    const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;
    const double* __restrict__ tw = TW + (drive ? w : 0) * R * 32;
    const int* __restrict__ tc = TC + (drive ? w : 0) * R * 32;
    QC_STAMP(P, b, lane, 2);

    // ---- phase 1 ------------------------------------------------------------------------------------------------
    if (w < 2) {                      // E tile I = w
        const v4d e = gy_tile(GL, w, lane, tile_ld(MT, 0, lane), tile_ld(MT, 1, lane));
        tile_st(ET, w, lane, e);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ep[(16 * w + 4 * r + g) * kPS + j] = e[r];
    } else if (w < 4) {               // (G D) tile I = w - 2, W tile
        const int I = w - 2;
        const v4d gd = gy_tile(GL, I, lane, tile_ld(DT, 0, lane), tile_ld(DT, 1, lane));
        tile_st(GDT, I, lane, gd);
        tile_st(WT, I, lane, (-c1) * tile_ld(ST, I, lane) + c2h2 * gd);
    } else {                          // Q tile (I, J): Q[16 I + i][16 J + j] = sum_c M[16 I + i][c] D[16 J + j][c]
        const int I = (w - 4) >> 1, J = (w - 4) & 1;
        const v4d z = {0.0, 0.0, 0.0, 0.0};
        v4d q = z;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)   // A: lane (g, i) = M[16 I + i][4 kk + g];  B: lane (g, j) = D[16 J + j][4 kk + g]
            q = __builtin_amdgcn_mfma_f64_16x16x4f64(Mp[(16 * I + j) * kPS + 4 * kk + g], Dp[(16 * J + j) * kPS + 4 * kk + g], q, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Qp[(16 * I + 4 * r + g) * kQS + 16 * J + j] = q[r];
    }
    v4d Yk[2], TY[2];
    if (drive) {
        gather_rows_operand<R>(tw, tc, Mp, g, j, Yk);               // Y_k = G_k M
        // (G Y_k)^T: A = Y_k[K] (operand-layout registers read as an A operand are the transposed tile), B = G_A[2 J + K]
        const v4d z = {0.0, 0.0, 0.0, 0.0};
        v4d t0 = z, t1 = z;
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            const v4d b0 = tile_ld(GL, K, lane), b1 = tile_ld(GL, 2 + K, lane);      // G_A tiles (J = 0, K), (J = 1, K)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Yk[K][kk], b0[kk], t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Yk[K][kk], b1[kk], t1, 0, 0, 0);
            }
        }
        TY[0] = t0;
        TY[1] = t1;
    }
    QC_STAMP(P, b, lane, 3);
    __syncthreads();
    QC_STAMP(P, b, lane, 4);

    // ---- phase 2 ------------------------------------------------------------------------------------------------
    if (drive) {
        double* __restrict__ pUa = Hb + P.ho_Ua + (size_t)w * 512;
        double* __restrict__ paU = Hb + P.ho_aU + (size_t)w * 512;
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            const v4d yt = gather_rows_store<R>(tw, tc, Mp, J, g, j);       // Y_k^T
            const v4d get = gather_rows_store<R>(tw, tc, Ep, J, g, j);      // (G_k E)^T
            const v4d lin = hc1 * yt, qd = hc2 * (get + TY[J]);
            store_T32(pUa, lin - qd, J, g, j);
            store_T32(paU, lin + qd, J, g, j);
        }
        QC_STAMP(P, b, lane, 5);
        // (a_k, h) and this wave's pairs: one batched reduction
        double pv[1 + kPairsPerWave];
        pv[0] = 0.0;
        if (ft) {
            v4d Vk[2];
            gather_rows_operand<R>(tw, tc, Dp, g, j, Vk);           // V_k = G_k D
            pv[0] = -(dot4(Yk[0], tile_ld(WT, 0, lane)) + dot4(Yk[1], tile_ld(WT, 1, lane))) -
                    c2h2 * (dot4(tile_ld(ET, 0, lane), Vk[0]) + dot4(tile_ld(ET, 1, lane), Vk[1]));
        }
        if constexpr (R == 1) {
#pragma unroll
            for (int t = 0; t < kPairsPerWave; ++t) pv[1 + t] = pwv[t] * Qp[pov[t]];
        } else {
            const double* __restrict__ pw = reinterpret_cast<const double*>(hot_ell + lay.pw);
            const int* __restrict__ po = reinterpret_cast<const int*>(hot_ell + lay.po);
#pragma unroll
            for (int t = 0; t < kPairsPerWave; ++t) {
                const int p = w + m * t;                 // pairs dealt round-robin over the drive waves
                double acc = 0.0;
                if (p < npairs) {
#pragma unroll
                    for (int e = 0; e < L / 64; ++e) acc += pw[(size_t)p * L + 64 * e + lane] * Qp[po[(size_t)p * L + 64 * e + lane]];
                }
                pv[1 + t] = acc;
            }
        }
        wave_sum_multi<1 + kPairsPerWave>(pv);
        if (lane == 0) {
            if (ft) Hb[P.ho_ah + w] = pv[0];
#pragma unroll
            for (int t = 0; t < kPairsPerWave; ++t) {
                const int p = w + m * t;
                if (p < npairs) Hb[P.ho_aa + p] = hc2 * pv[1 + t];
            }
        }
    }
    QC_STAMP(P, b, lane, 6);
    if (ft) {
        if (w < 2) {                  // (U_t, h)^T and (h, U_t+1)^T, column block J = w
            const int J = w;
            const v4d ge = gyT_tile(GL, J, lane, tile_ld(ET, 0, lane), tile_ld(ET, 1, lane));     // (G E)^T[J]
            v4d et;                                                                                // E^T[J]
#pragma unroll
            for (int r = 0; r < 4; ++r) et[r] = Ep[(16 * J + j) * kPS + 4 * r + g];
            store_T32(Hb + P.ho_Uh, c1 * et - c2h2 * ge, J, g, j);
            store_T32(Hb + P.ho_hU, c1 * et + c2h2 * ge, J, g, j);
        } else if (w == 2) {          // (h, h)
            double s[1] = {dot4(tile_ld(ET, 0, lane), tile_ld(GDT, 0, lane)) + dot4(tile_ld(ET, 1, lane), tile_ld(GDT, 1, lane))};
            wave_sum_multi<1>(s);
            if (lane == 0) Hb[P.ho_hh] = -2.0 * c2 * s[0];
        }
    }
    if (w == 5) qc_hess_tail(Pk, mu, Hb, lane, 64);     // derivative integrators' (dx, h) entries and the alignment padding
    if constexpr (DIAG) {
        QC_STAMP(P, b, lane, 7);
        if (P.stamps != nullptr && lane == 0 && (w == 0 || w == 5)) {   // slots 0-7: wave 0, 8-15: wave 5
#pragma unroll
            for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + (w == 0 ? 0 : 8) + k_] = qc_ts_[k_];
        }
    }
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
    const int L = R == 1 ? 64 : 256;
    // G_i G_k + G_k G_i, structurally (an entry that cancels to exactly zero is kept out: it adds nothing)
    const int npairs = m * (m + 1) / 2;
    std::vector<std::vector<std::pair<int, double>>> lists(npairs);
    for (int hi = 0; hi < m; ++hi)
        for (int lo = 0; lo <= hi; ++lo) {
            std::vector<double> Pm((size_t)n * n, 0.0);
            for (int a = 0; a < n; ++a)
                for (int c = 0; c < n; ++c) {
                    const double x = Gk(lo, a, c), y = Gk(hi, a, c);
                    if (x != 0.0) for (int d = 0; d < n; ++d) Pm[(size_t)a * n + d] += x * Gk(hi, c, d);
                    if (y != 0.0) for (int d = 0; d < n; ++d) Pm[(size_t)a * n + d] += y * Gk(lo, c, d);
                }
            auto& li = lists[hi * (hi + 1) / 2 + lo];
            for (int a = 0; a < n; ++a)
                for (int d = 0; d < n; ++d)
                    if (Pm[(size_t)a * n + d] != 0.0) li.emplace_back(a * kQS + d, Pm[(size_t)a * n + d]);
            if ((int)li.size() > L) return 0;            // (cannot happen: at most 2 R^2 entries per row)
        }
    // how many drives touch one entry of G
    int slots = 0;
    for (int a = 0; a < n; ++a)
        for (int c = 0; c < n; ++c) {
            int cnt = 0;
            for (int k = 0; k < m; ++k) cnt += Gk(k, a, c) != 0.0;
            slots = std::max(slots, cnt);
        }
    if (slots > kMaxSlots) slots = 0;                    // the kernel assembles G from the dense images then
    const EllLayout lay = ell_layout(m, R, L, slots);
    blob->assign(lay.bytes, 0);
    double* tw = reinterpret_cast<double*>(blob->data() + lay.tw);
    double* pw = reinterpret_cast<double*>(blob->data() + lay.pw);
    double* gw = reinterpret_cast<double*>(blob->data() + lay.gw);
    int* tc = reinterpret_cast<int*>(blob->data() + lay.tc);
    int* po = reinterpret_cast<int*>(blob->data() + lay.po);
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
    for (int p = 0; p < npairs; ++p)
        for (size_t e = 0; e < lists[p].size(); ++e) {
            po[(size_t)p * L + e] = lists[p][e].first;
            pw[(size_t)p * L + e] = lists[p][e].second;
        }
    // assembly plan in the order of the A-layout image (qc_mfma32_pack_G): entry [tile = 2 I + K][pair][lane = 16 g + i][e] = X[16 I + i][16 K + 4 (2 pair + e) + g]
    for (int tile = 0; tile < 4 && slots > 0; ++tile)
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

hipError_t qc_launch_mfma32_ell_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    const int grid = P.n_int;
#define QC_ELL_ARGS P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu + P.t_begin * P.F_stride + P.F_off, (const char*)P.ell, P.n_int, P.zdim, P.m, \
                    P.off_a, P.off_dt, P.off_U, (int)P.F_stride, P.ell_slots, P, dH
    if (P.stamps != nullptr) {
        if (P.ell_R == 1) hipLaunchKernelGGL((qc_mfma32_ell_hess_kernel<1, true>), dim3(grid), dim3(kEThreads), 0, st, QC_ELL_ARGS);
        else hipLaunchKernelGGL((qc_mfma32_ell_hess_kernel<2, true>), dim3(grid), dim3(kEThreads), 0, st, QC_ELL_ARGS);
    } else if (P.ell_R == 1) hipLaunchKernelGGL((qc_mfma32_ell_hess_kernel<1, false>), dim3(grid), dim3(kEThreads), 0, st, QC_ELL_ARGS);
    else hipLaunchKernelGGL((qc_mfma32_ell_hess_kernel<2, false>), dim3(grid), dim3(kEThreads), 0, st, QC_ELL_ARGS);
#undef QC_ELL_ARGS
    return hipGetLastError();
}
