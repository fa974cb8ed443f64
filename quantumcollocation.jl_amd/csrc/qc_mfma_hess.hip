// f64-MFMA Hessian-of-Lagrangian kernel, order-4 Pade, 2N = 16, up to 8 drives: ONE wavefront per
// interval, operands in registers (lane maps: qc_mfma_kernels.hip header), reductions through LDS.
//
// With M = reshape(mu_t[0:s], 16, 8), M1 = G^T M, M2 = G^T M1, N_k = G_k^T M, N'_k = G_k^T M1,
// N''_k = G^T N_k, V_k = G_k D  (h = dt, c1 = 1/2, c2 = 1/12):
//   (U_t,  a_k)    = -c1 h N_k - c2 h^2 (N''_k + N'_k)          (a_k, U_t+1) = -c1 h N_k + c2 h^2 (N''_k + N'_k)
//   (U_t,  h)      = -(c1 M1 + 2 c2 h M2)                        (h,  U_t+1) = -c1 M1 + 2 c2 h M2
//   (a_i, a_k)     = c2 h^2 (<N_i, V_k> + <N_k, V_i>)
//   (a_k, h)       = -c1 <N_k, S> + 2 c2 h <N''_k + N'_k, D>        (<N_k, G D> = <G^T N_k, D>: the product G D is never formed)
//   (h, h)         = 2 c2 <M2, D>                                    (<M1, G D> = <G^T M1, D>)
//   (dx_i, h)      = -mu_i   (derivative integrators)
// Left multiplication by a transpose uses the B-layout image as the A operand
// (A-layout(X^T) = B-layout(X)); the B-layout tile of G is assembled from the B-layout images like the A-layout one, and
// the transposes for the stores go through LDS.  MFMAs per interval: 8 + 8 m + 4 ceil(m/2) (68 for m = 6).
#include <algorithm>

#include "qc_mfma_hess_common.h"

namespace {

using namespace qc_mfma;

constexpr int kHMmax = 8;                     // at most this many drives (held in registers)
constexpr int kHVals = kHMmax * (kHMmax + 1) / 2 + kHMmax + 1;
constexpr int kHStride = 65;                  // LDS row stride (doubles) of the reduction scratch

__device__ inline v4d load_img(const double* __restrict__ Gx, int mat, int lane) { return load_image_tile(Gx + mat * 256, lane); }
__device__ inline double dot4(const v4d& a, const v4d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ inline v4d sel(bool c, const v4d& a, const v4d& b) {
    return v4d{c ? a[0] : b[0], c ? a[1] : b[1], c ? a[2] : b[2], c ? a[3] : b[3]};
}

// (Exactly antisymmetric generators, QcParams.antisym, take qc_mfma16_pade4_hess_anti_kernel below; this is the general form,
// also used by the batched launch.)
// ONCE: one interval per workgroup, no persistent loop (whose invariants the compiler hoists in front of the first load; see
// qc_mfma16_pade4_kernel)
// DIAG: time stamps of the wave's phases (QC_STAMPS=1 handles; profiles/stamps_hess.py), never in a timed run
template <int kHM, bool KET, bool BATCH, bool ONCE = false>
__global__ __launch_bounds__(64) void qc_mfma16_pade4_hess_kernel(const QcParams Pk, const double* __restrict__ Z,
                                                                  const double* __restrict__ Mu, double* __restrict__ H,
                                                                  const QcParams* __restrict__ Pb) {
    constexpr bool DIAG = false;                  // (time stamps: the antisymmetric kernel only)
    QC_STAMP_DECL;
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    const QcParams& P = BATCH ? Pb[blockIdx.y] : Pk;      // BATCH: one launch for several handles (qc_mfma_kernels.hip)
    __shared__ double red[kHVals * kHStride];
    __shared__ double tscr[(kHM + 1) * 16 * 17];
    const int lane = threadIdx.x;
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ GxA = P.Gx;                          // A-layout images
    const double* __restrict__ GxB = P.Gx + (size_t)(m + 1) * 256;  // B-layout images (= A-layout of the transposes)
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    int vb = blockIdx.x;
    if (vb >= P.n_int) return;
    do {
        const int b = qc_xcd_remap(vb, P.n_int);
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;

        // ---- loads, one batch, in the order the first products need them: amplitudes and generator images (-> G), then the
        //      knots and the multipliers.  The amplitudes come by ONE vector load (load_amp_lanes; per drive they were m
        //      scalar loads, each behind its own wait: m dependent round trips in front of the first product).
        // K kets (nc < 8): the tile columns >= nc re-read column 0; the multipliers there are zeroed, which zeroes every
        // quantity derived from M in those columns (the scalar blocks sum over whole tiles), and they are never stored
        const int nc = KET ? P.nc : 8, jc = (!KET || jj < nc) ? jj : 0;     // KET = false: the masks fold away at compile time
        const int nr = KET ? P.n : 16;                                      // rows per column (N < 8 levels: zero-padded tile)
        const double av = load_amp_lanes(z0, P.off_a, m, lane);
        const double h = ft ? load_uniform(z0 + P.off_dt) : opaque_scalar(P.dt_fixed);
        v4d gA[kHM], gB[kHM];
        v4d Gb = load_img(GxB, 0, lane);                    // B-layout of G = A-layout of G^T, assembled like Ga (no identity product)
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            const int k = u < m ? u : (m > 0 ? m - 1 : 0);
            gA[u] = load_img(GxA, m > 0 ? k + 1 : 0, lane);
            gB[u] = load_img(GxB, m > 0 ? k + 1 : 0, lane);
        }
        v4d u0, u1, mraw;
        if constexpr (!KET) {
            u0 = load_col16_T(z0 + P.off_U + jc * 16, g);     // 2 requests of 16 bytes per lane instead of 4 of 8 (qc_mfma_common.h)
            u1 = load_col16_T(z1 + P.off_U + jc * 16, g);
            mraw = load_col16_T(mu + jc * 16, g);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * r + g;
                const bool in = row < nr;
                u0[r] = in ? z0[P.off_U + jc * nr + row] : 0.0;
                u1[r] = in ? z1[P.off_U + jc * nr + row] : 0.0;
                mraw[r] = in ? mu[jc * nr + row] : 0.0;
            }
        }
        const v4d mv = (!KET || jj < nc) ? mraw : v4d{0.0, 0.0, 0.0, 0.0};
        // multipliers of the derivative integrators' rows (their entries -mu_i close the block): requested HERE -- a load issued
        // at the end of the wave queues behind the wave's own ~50 stores.  Unused slots have zero dims / offsets: in bounds.
        const bool dfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
        double mud[2] = {0.0, 0.0};
        if (dfast) {
#pragma unroll
            for (int d = 0; d < 2; ++d) mud[d] = mu[P.drow[d] + (lane < P.ddim_i[d] ? lane : 0)];
        }
        QC_STAMP(P, b, lane, 1);                  // every load of the interval requested
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 2);              // ... and back
        }
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            const double a = (u < m) ? bcast_lane(av, u) : 0.0;
            Gb += a * gB[u];
        }
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;

        // ---- products in three dependency stages, each a batch of independent 16x16x16 products whose MFMAs are
        //      interleaved (mm16_multi):  1: M1   2: M2, [N_k|N'_k], [V_k|.]   3: N''
        const v4d TM0 = sel(left, mv, zero);                // [M | 0]
        v4d Sb, Db, Dv;                                     // [S | S], [D | D], [c2 h^2 D | 0]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Sb[r] = u1[r] + u0[r];                          // (both halves of the lanes hold the same 8 columns)
            Db[r] = u1[r] - u0[r];
            Dv[r] = left ? hc2 * Db[r] : 0.0;
        }
        const v4d Y1 = mm16(Gb, TM0);                       // [M1 | 0]
        QC_STAMP(P, b, lane, 3);                  // stage 1 issued
        const v4d TM = sel(left, TM0, swap8(Y1));           // [M | M1]
        v4d Y2, NN[kHM], VV[kHM];                           // [M2 | 0], [N_k | N'_k], [c2 h^2 V_k | 0]
        {
            constexpr int N3 = 1 + 2 * kHM;
            v4d a3[N3], b3[N3], d3[N3];
            a3[0] = Gb;
            b3[0] = Y1;
#pragma unroll
            for (int u = 0; u < kHM; ++u) {                 // unused slots (u >= m) repeat the last drive; zeroed below
                a3[1 + u] = gB[u];
                b3[1 + u] = TM;
                a3[1 + kHM + u] = gA[u];
                b3[1 + kHM + u] = Dv;
            }
            mm16_multi<N3>(a3, b3, d3);
            QC_STAMP(P, b, lane, 4);              // stage 2 issued
            Y2 = d3[0];
#pragma unroll
            for (int u = 0; u < kHM; ++u) {
                NN[u] = (u < m) ? d3[1 + u] : zero;
                VV[u] = (u < m) ? d3[1 + kHM + u] : zero;
            }
        }
        v4d PN[kHM / 2], PN1[kHM / 2], PN2[kHM / 2];
        {
            constexpr int N4 = kHM / 2;
            v4d a4[N4], b4[N4];
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) {
                PN[p2] = sel(left, NN[2 * p2], swap8(NN[2 * p2 + 1]));      // [N_k | N_k+1]
                PN1[p2] = sel(left, swap8(NN[2 * p2]), NN[2 * p2 + 1]);     // [N'_k | N'_k+1]
                a4[p2] = Gb;
                b4[p2] = PN[p2];
            }
            mm16_multi<N4>(a4, b4, PN2);                                    // [N''_k | N''_k+1]
            QC_STAMP(P, b, lane, 5);              // stage 3 issued
        }
        // transposes for the line-wide stores go through the padded LDS scratch (an identity product costs 4 MFMAs each), all
        // kHM + 1 tiles in ONE LDS round trip
        const v4d uh = -(c1 * Y1 + c2h2 * Y2), hu = (-c1) * Y1 + c2h2 * Y2;
        v4d ET, XT[kHM];                                    // (U_t, h) | (h, U_t+1);  (U_t, a) and (a, U_t+1) tiles, transposed
        {
            v4d tin[kHM + 1], tout[kHM + 1];
            tin[0] = sel(left, uh, swap8(hu));
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) {
                const v4d q = hc2 * (PN2[p2] + PN1[p2]), lin = (-hc1) * PN[p2];
                tin[1 + 2 * p2] = lin - q;
                tin[2 + 2 * p2] = lin + q;
            }
            lds_transpose16_multi<kHM + 1>(tscr, tin, tout, g, j);
            if constexpr (DIAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 6);              // tiles transposed
            ET = tout[0];
#pragma unroll
            for (int u = 0; u < kHM; ++u) XT[u] = tout[1 + u];
        }
        if (ft) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;
                if (!KET || ((c & 7) < nc && j < nr)) qc_st8m<2>(Hb + (c < 8 ? P.ho_Uh + c * nr : P.ho_hU + (c - 8) * nr) + j, ET[r]);
            }
        }
#pragma unroll
        for (int u = 0; u < kHM; u += 2) {
            if (u < m) {
                const bool two = u + 1 < m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * r + g;           // tile columns < 8: drive u column c; >= 8: drive u+1 column c-8
                    if ((r < 2 || two) && (!KET || ((c & 7) < nc && j < nr))) {
                        const size_t o = (size_t)(u + (r < 2 ? 0 : 1)) * (KET ? P.s : 128) + (c & 7) * nr + j;
                        qc_st8m<2>(Hb + P.ho_Ua + o, XT[u][r]);
                        qc_st8m<2>(Hb + P.ho_aU + o, XT[u + 1][r]);
                    }
                }
            }
        }
        QC_STAMP(P, b, lane, 7);                  // matrix blocks' stores issued
        // ---- scalar blocks: per-lane partial sums, reduced through LDS.  They follow the matrix blocks' stores, which are
        //      nine tenths of the interval's bytes.  (Vector instructions of a wave do NOT issue in the shadow of its own f64
        //      MFMAs -- tests/hip/mfma_valu_overlap.hip --, so placing these sums between the products of stage 3 only delays
        //      the stores.)  No branches: the unused drives' tiles are zero; the right halves of the V tiles are zero, so no lane
        //      mask.  Row v (v + 1) / 2 + u whatever m is; the rows of unused drives receive zeros and are overwritten by the
        //      (a, h) / (h, h) rows below (LDS operations of a wave execute in order).
#pragma unroll
        for (int v = 0; v < kHM; ++v) {
#pragma unroll
            for (int u = 0; u <= v; ++u) red[(v * (v + 1) / 2 + u) * kHStride + lane] = dot4(NN[u], VV[v]) + dot4(NN[v], VV[u]);
        }
        const int naa = m * (m + 1) / 2;
        if (ft) {
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) {      // drive 2 p2 in the left half of the pair tiles, 2 p2 + 1 in the right
                const double tv = c2h2 * dot4(PN2[p2] + PN1[p2], Db) - c1 * dot4(PN[p2], Sb);
                red[(naa + 2 * p2) * kHStride + lane] = left ? tv : 0.0;
                red[(naa + 2 * p2 + 1) * kHStride + lane] = left ? 0.0 : tv;
            }
            red[(naa + m) * kHStride + lane] = left ? 2.0 * c2 * dot4(Y2, Db) : 0.0;    // after the pair rows: row naa + m may be one of them
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
            const int nval = naa + (ft ? m + 1 : 0);
            // up to 32 values (kHM <= 6): lanes l and l + 32 sum one half of row l each; otherwise one lane per row
            constexpr bool kSplit = kHM * (kHM + 1) / 2 + kHM + 1 <= 32;
            const int rowi = kSplit ? (lane & 31) : lane;
            const int ncol = kSplit ? 32 : 64;
            const double* row = red + rowi * kHStride + (kSplit ? (lane >> 5) * 32 : 0);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            if (rowi < nval) {
#pragma unroll
                for (int c = 0; c < ncol; c += 4) { a0 += row[c]; a1 += row[c + 1]; a2 += row[c + 2]; a3 += row[c + 3]; }
            }
            double sum = (a0 + a1) + (a2 + a3);
            if constexpr (kSplit) sum += __shfl_xor(sum, 32, 64);
            if (lane < nval) {
                const int o = lane < naa ? P.ho_aa + lane : (lane < naa + m ? P.ho_ah + (lane - naa) : P.ho_hh);
                Hb[o] = sum;
            }
        }
        // derivative integrators: d2/d(dx_i) dh = -mu_i; alignment padding (explicit zeros)
        if (dfast) {
            int o = P.ho_d;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (lane < P.ddim_i[d]) Hb[o + lane] = -mud[d];
                o += P.ddim_i[d];
            }
            for (int i = lane; i < P.h_pad; i += 64) Hb[P.hess_nnz + i] = 0.0;
        } else {
            qc_hess_tail(P, mu, Hb, lane, 64);
        }
        QC_STAMP(P, b, lane, 8);                  // every store issued
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 9);              // drained
            QC_STAMP_FLUSH(P, b, lane, 0, 9);
        }
        __builtin_amdgcn_wave_barrier();   // the scratch rows are rewritten by the next interval
    } while (!ONCE && (vb += gridDim.x) < P.n_int);
}


// ---- antisymmetric generators (every Hermitian Hamiltonian gives them): G_k^T = -G_k -----------------------------------------------
// The evaluation every physical system takes.  A sign-free formulation on the A-layout images alone, in TWO dependency stages (the
// general kernel above needs three), with half the (a, a) dot products and without the lane masks / accumulator copies of the
// general form.  What shaped it (tests/hip/mfma_valu_overlap.hip, profiles/r02_hess_timeline*.txt): a wave's vector instructions
// do not issue in the shadow of its own f64 MFMAs (a wave with 68 MFMAs and 1100 vector instructions spends 4350 + 5000 cycles,
// not max of the two), so the vector instruction count is as much the kernel's cost as the products.
//   MD = [M | c2 h^2 D]                                  (one B operand for all of stage A)
//   stage A (1 + m products):   Y = G MD = [-M1 | .]        T_k = G_k MD = [-N_k | c2 h^2 V_k]
//   stage B (1 + 3 ceil(m/2)):  Y2 = G Y = [M2 | .]
//        Q_p = (2 c2 h G) [-N_k | -N_k+1] + G_k [2 c2 h (-M1) | 0] + G_k+1 [0 | 2 c2 h (-M1)]        (ONE accumulator chain per pair)
//            = 2 c2 h [N''_k + N'_k | N''_k+1 + N'_k+1]
//   (U_t, a) | (a, U_t+1) pair tiles = c1 h [-N_k | -N_k+1] -/+ (h / 2) Q_p        (U_t, h) | (h, U_t+1) = c1 Y -/+ 2 c2 h Y2
//   (a_u, a_v) = - sum over ALL lanes of T_u . swap8(T_v)     (left lanes: -N_u . c2 h^2 V_v,  right lanes: c2 h^2 V_u . -N_v)
//   (a_k, h)   = <Q_p, D> + c1 <[-N_k | -N_k+1], S>  by halves of the lanes;      (h, h) = 2 c2 <M2, D>  (left lanes)
// Unused drive slots (u >= m) repeat the last drive; nothing derived from them is stored, so nothing is zeroed.
// (Tried and removed: four intervals per workgroup with the generator images fetched once and shared through LDS -- the images
// are 14 of the 21 KB a wave requests and the load phase is bound by the compute unit's 64 B / clock vector-memory path.  In the
// timeline the loads were back 0.25 us earlier, the launch took 10.15 instead of 9.76 us: the 256-thread, 124 KB-LDS workgroups
// cost more to dispatch and to retire than the loads saved.)
// The scalar blocks: one row of per-lane partial sums per value in LDS; lanes l and l + 32 sum the left / right lanes' columns of
// row l, so a row can carry two values (the pair rows) without a lane mask.
template <int kHM>
struct HessAntiRows {
    static constexpr int kAA = kHM * (kHM + 1) / 2, kPair = kHM / 2, kRows = kAA + kPair + 1;
};

template <int MODE>
__device__ __forceinline__ void st_off(double* __restrict__ ubase, unsigned byteoff, double v) {   // uniform base + 32-bit lane offset
    qc_st8m<MODE>(reinterpret_cast<double*>(reinterpret_cast<char*>(ubase) + byteoff), v);
}

// The leading scalar / pointer arguments are PRELOADED into scalar registers while the wave is launched (this file is compiled
// with -amdgpu-kernarg-preload-count): they are what the first load requests depend on (qc_mfma_kernels.hip).  mu0 = the
// multipliers of this handle's first interval; f_stride = rows per interval.
#define QC_HESS_HOT_ARGS(P) (P).Gx, dZ + (P).t_begin * (long long)(P).zdim, dMu + (P).t_begin * (P).F_stride + (P).F_off, (P).n_int, (P).zdim, (P).off_a, \
                            (P).off_dt, (P).m, (P).off_U, (int)(P).F_stride
// ELL: every drive generator has ONE entry per row (P.ell16; qc_mfma_hess_common.h): the 24 + 24 products with drive images become row
// gathers from row-major LDS copies -- 20 instead of 68 f64 MFMAs per interval (64 cycles each in dependent chains), the same bits.
template <int kHM, bool KET, bool BATCH, bool ONCE, bool DIAG = false, bool ELL = false>
__global__ __launch_bounds__(64) void qc_mfma16_pade4_hess_anti_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                       const double* __restrict__ hot_mu0, int hot_n_int,
                                                                       int hot_zdim, int hot_off_a, int hot_off_dt, int hot_m, int hot_off_U,
                                                                       int hot_f_stride, const QcParams Pk, const double* __restrict__ Z,
                                                                       const double* __restrict__ Mu, double* __restrict__ H,
                                                                       const QcParams* __restrict__ Pb) {
    QC_STAMP_DECL;
    QC_STAMP(Pk, 0, 0, 0);                        // kernel entry
    QcKernargTouch<sizeof(QcParams) + 128> touch; // requested here, waited for behind the first load requests (qc_internal.h)
    touch.request();
    const QcParams& P = BATCH ? Pb[blockIdx.y] : Pk;      // BATCH: one launch for several handles (qc_mfma_kernels.hip)
    const int h_n_int = BATCH ? P.n_int : hot_n_int, h_zdim = BATCH ? P.zdim : hot_zdim, h_off_a = BATCH ? P.off_a : hot_off_a;
    const int h_off_dt = BATCH ? P.off_dt : hot_off_dt, h_off_U = BATCH ? P.off_U : hot_off_U;
    if constexpr (DIAG) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        QC_STAMP(Pk, 0, 0, 10);                   // kernel arguments in the scalar cache
    }
    using R = HessAntiRows<kHM>;
    // LDS: the parked stage-A tiles, and ONE scratch region that serves the transposes (in rounds of at most four tiles) and then the
    // reduction rows of the scalar blocks, which are written when the transposed tiles are back in registers.  26.8 KB at six drives:
    // SIX workgroups per CU instead of the four that three separate arrays (40.5 KB) allowed -- a long trajectory is bound by how many
    // intervals are in flight (one wave each, a 4.7 us latency chain), not by HBM.
    constexpr int kTrRound = 4;
    constexpr int kScrTiles = kHM + 1 < kTrRound ? kHM + 1 : kTrRound;
    constexpr int kScrLen = kScrTiles * 272 > R::kRows * kHStride ? kScrTiles * 272 : R::kRows * kHStride;
    // (ELL keeps the T_k in registers -- it holds no drive image through stage B --: 14.6 KB of LDS, eight workgroups per CU)
    constexpr bool kPark = !ELL;
    __shared__ double tsave[kPark ? kHM * 256 : 1];   // the stage-A tiles T_k, parked for the (a, a) sums (registers: see below)
    __shared__ double tscr[kScrLen];
    __shared__ double ellTW[ELL ? 96 : 1];        // ELL: the drives' rows (weights; columns x kXS)
    __shared__ int ellTC[ELL ? 96 : 1];
    static_assert(!ELL || (!KET && !BATCH), "the row-gather form serves whole unitaries of one handle");
    double* __restrict__ red = tscr;
    const int lane = threadIdx.x;
    const int m = BATCH ? P.m : hot_m;
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const bool ft = h_off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ GxA = BATCH ? P.Gx : hot_Gx;
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    // The generator images depend on nothing but the kernel arguments: requested before any address of the interval is computed
    // (and once for all intervals of a persistent grid)
    v4d gA[kHM];
    const v4d G0 = load_img(GxA, 0, lane);
#pragma unroll
    for (int u = 0; u < kHM; ++u) {
        const int k = u < m ? u : (m > 0 ? m - 1 : 0);
        gA[u] = load_img(GxA, m > 0 ? k + 1 : 0, lane);
    }
    if constexpr (ELL) {
        fu_load_tables(P.ell16, m, lane, ellTW, ellTC);
        fu_lds_order();
    }
    int vb = blockIdx.x;
    if (vb >= h_n_int) return;
    do {
        const int b = qc_xcd_remap(vb, h_n_int);
        const double* __restrict__ z0 = BATCH ? Z + (P.t_begin + b) * (long long)h_zdim : hot_Zt + (long long)b * h_zdim;
        const double* __restrict__ z1 = z0 + h_zdim;
        const double* __restrict__ mu = BATCH ? Mu + (P.t_begin + b) * P.F_stride + P.F_off : hot_mu0 + (long long)b * hot_f_stride;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;

        // ---- loads: one batch, amplitudes and images (-> G) first (see the general kernel) -------------------------------
        const int nc = KET ? P.nc : 8, jc = (!KET || jj < nc) ? jj : 0;
        const int nr = KET ? P.n : 16;
        const double av = load_amp_lanes(z0, h_off_a, m, lane);
        const double h = ft ? load_uniform(z0 + h_off_dt) : opaque_scalar(P.dt_fixed);
        QC_STAMP(P, b, lane, 11);                 // first two loads requested
        v4d Ga = G0;
        v4d u0, u1, mraw;
        if constexpr (!KET) {
            u0 = load_col16_T(z0 + h_off_U + jc * 16, g);     // 2 requests of 16 bytes per lane instead of 4 of 8 (qc_mfma_common.h)
            u1 = load_col16_T(z1 + h_off_U + jc * 16, g);
            mraw = load_col16_T(mu + jc * 16, g);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * r + g;
                const bool in = row < nr;
                u0[r] = in ? z0[P.off_U + jc * nr + row] : 0.0;
                u1[r] = in ? z1[P.off_U + jc * nr + row] : 0.0;
                mraw[r] = in ? mu[jc * nr + row] : 0.0;
            }
        }
        const bool dfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
        double mud[2] = {0.0, 0.0};
        if (dfast) {
#pragma unroll
            for (int d = 0; d < 2; ++d) mud[d] = mu[P.drow[d] + (lane < P.ddim_i[d] ? lane : 0)];
        }
        QC_STAMP(P, b, lane, 1);                  // every load of the interval requested
        touch.consume();                          // (the argument block's lines: the one scalar wait, behind every request)
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 2);              // ... and back
        }
        const v4d mv = (!KET || jj < nc) ? mraw : zero;     // K kets: the re-read columns carry zero multipliers (general kernel)
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
            const double a = (u < m) ? bcast_lane(av, u) : 0.0;
            Ga += a * gA[u];
        }
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h, hh2 = 0.5 * h;
        v4d Sc, Db, MD;                                     // c1 [S | S], [D | D], [M | c2 h^2 D]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Sc[r] = c1 * (u1[r] + u0[r]);                   // (both halves of the lanes hold the same 8 columns)
            Db[r] = u1[r] - u0[r];
            MD[r] = left ? mv[r] : hc2 * Db[r];
        }
        // ---- stage A: G MD and G_k MD, interleaved ----------------------------------------------------------------------
        v4d Y, T[kHM];
        if constexpr (ELL) {
            fu_put_rows(tscr, MD, g, j);          // row-major copy of [M | c2 h^2 D]
            fu_lds_order();
            {   // (mm16_multi's single accumulator chain: the bits of the dense-image form)
                v4d a1[1] = {Ga}, b1[1] = {MD}, d1[1];
                mm16_multi<1>(a1, b1, d1);
                Y = d1[0];
            }
            fu_gather_all<kHM>(ellTW, ellTC, tscr, g, j, T);
            fu_lds_order();                       // (stage B rewrites the copy)
        } else {
            constexpr int NA = 1 + kHM;
            v4d aA[NA], bA[NA], dA[NA];
            aA[0] = Ga;
            bA[0] = MD;
#pragma unroll
            for (int u = 0; u < kHM; ++u) {
                aA[1 + u] = gA[u];
                bA[1 + u] = MD;
            }
            mm16_multi<NA>(aA, bA, dA);
            Y = dA[0];
#pragma unroll
            for (int u = 0; u < kHM; ++u) T[u] = dA[1 + u];
        }
        QC_STAMP(P, b, lane, 3);                  // stage A issued
        // The T_k are needed again for the (a, a) sums at the very end.  Kept in registers they push the kernel over 256
        // registers, which costs an accumulator-register copy per MFMA result word (144 v_accvgpr_read); LDS writes issue in the
        // shadow of the MFMAs, and the swapped halves come back for free (read at lane ^ 8) instead of by 48 DPP moves.
#pragma unroll
        for (int u = 0; u < kHM; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { if constexpr (kPark) tsave[(u * 4 + r) * 64 + lane] = T[u][r]; }
        }
        // ---- stage B ------------------------------------------------------------------------------------------------------
        v4d PNn[kHM / 2], Q[kHM / 2], Y2;                   // [-N_k | -N_k+1], 2 c2 h [N'' + N' pairs], [M2 | .]
        {
            v4d YL, Gs;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                YL[r] = left ? c2h2 * Y[r] : 0.0;           // [2 c2 h (-M1) | 0]
                Gs[r] = c2h2 * Ga[r];
            }
            const v4d YR = swap8(YL);                       // [0 | 2 c2 h (-M1)]
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) PNn[p2] = sel(left, T[2 * p2], swap8(T[2 * p2 + 1]));
            // 12-deep accumulator chains of the pairs, interleaved with each other and with Y2's (no MFMA waits for its predecessor)
            Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[0], Y[0], zero, 0, 0, 0);
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[0], PNn[p2][0], zero, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) {
                Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ga[kk], Y[kk], Y2, 0, 0, 0);
#pragma unroll
                for (int p2 = 0; p2 < kHM / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[kk], PNn[p2][kk], Q[p2], 0, 0, 0);
            }
            if constexpr (ELL) {
                fu_put_rows(tscr, YL, g, j);      // row-major copy of [2 c2 h (-M1) | 0]
                fu_lds_order();
#pragma unroll
                for (int p2 = 0; p2 < kHM / 2; ++p2) fu_gather_pair(ellTW, ellTC, tscr, p2, left, g, jj, Q[p2]);
                fu_lds_order();                   // (the transposes below rewrite the scratch)
                (void)YR;
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int p2 = 0; p2 < kHM / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[2 * p2][kk], YL[kk], Q[p2], 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int p2 = 0; p2 < kHM / 2; ++p2) Q[p2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gA[2 * p2 + 1][kk], YR[kk], Q[p2], 0, 0, 0);
                }
            }
        }
        QC_STAMP(P, b, lane, 5);                  // stage B issued
        // ---- matrix blocks: combine, transpose through LDS (all tiles in one round trip), store -------------------------
        v4d ET, XT[kHM];
        {
            v4d tin[kHM + 1], tout[kHM + 1];
            const v4d ty = c1 * Y, ts = c2h2 * Y2;
            tin[0] = sel(left, ty - ts, swap8(ty + ts));    // (U_t, h) | (h, U_t+1)
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) {
                const v4d lin = hc1 * PNn[p2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tin[1 + 2 * p2][r] = __builtin_fma(-hh2, Q[p2][r], lin[r]);
                    tin[2 + 2 * p2][r] = __builtin_fma(hh2, Q[p2][r], lin[r]);
                }
            }
            lds_transpose16_rounds<kHM + 1, kTrRound>(tscr, tin, tout, g, j);
            if constexpr (DIAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 6);              // tiles transposed
            ET = tout[0];
#pragma unroll
            for (int u = 0; u < kHM; ++u) XT[u] = tout[1 + u];
        }
        if constexpr (!KET) {
            // lane (g, j), register r of a transposed tile is element j of column 4 r + g: byte offset 8 (16 (4 r + g) + j); the
            // register / drive parts are immediates
            const unsigned lo = 8u * (16u * g + j);
            if (ft) {
                double* __restrict__ eb = Hb + P.ho_Uh;     // (U_t, h): columns 0..7, (h, U_t+1): columns 8..15 of the tile
                double* __restrict__ fb = Hb + P.ho_hU;
                st_off<2>(eb, lo, ET[0]);
                st_off<2>(eb, lo + 512u, ET[1]);
                st_off<2>(fb, lo, ET[2]);
                st_off<2>(fb, lo + 512u, ET[3]);
            }
            double* __restrict__ xb = Hb + P.ho_Ua;
            double* __restrict__ yb = Hb + P.ho_aU;
#pragma unroll
            for (int u = 0; u < kHM; u += 2) {
                if (u < m) {
                    st_off<2>(xb, lo + 1024u * u, XT[u][0]);
                    st_off<2>(yb, lo + 1024u * u, XT[u + 1][0]);
                    st_off<2>(xb, lo + 1024u * u + 512u, XT[u][1]);
                    st_off<2>(yb, lo + 1024u * u + 512u, XT[u + 1][1]);
                    if (u + 1 < m) {
                        st_off<2>(xb, lo + 1024u * (u + 1), XT[u][2]);
                        st_off<2>(yb, lo + 1024u * (u + 1), XT[u + 1][2]);
                        st_off<2>(xb, lo + 1024u * (u + 1) + 512u, XT[u][3]);
                        st_off<2>(yb, lo + 1024u * (u + 1) + 512u, XT[u + 1][3]);
                    }
                }
            }
        } else {
            if (ft) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * r + g;
                    if ((c & 7) < nc && j < nr) qc_st8m<2>(Hb + (c < 8 ? P.ho_Uh + c * nr : P.ho_hU + (c - 8) * nr) + j, ET[r]);
                }
            }
#pragma unroll
            for (int u = 0; u < kHM; u += 2) {
                if (u < m) {
                    const bool two = u + 1 < m;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * r + g;           // tile columns < 8: drive u column c; >= 8: drive u+1 column c-8
                        if ((r < 2 || two) && (c & 7) < nc && j < nr) {
                            const size_t o = (size_t)(u + (r < 2 ? 0 : 1)) * P.s + (c & 7) * nr + j;
                            qc_st8m<2>(Hb + P.ho_Ua + o, XT[u][r]);
                            qc_st8m<2>(Hb + P.ho_aU + o, XT[u + 1][r]);
                        }
                    }
                }
            }
        }
        QC_STAMP(P, b, lane, 7);                  // matrix blocks' stores issued
        // ---- scalar blocks (after the matrix blocks' stores: nine tenths of the interval's bytes are on their way) -----------
        {
            v4d Tn[kHM];                                    // -T_u
#pragma unroll
            for (int u = 0; u < kHM; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Tn[u][r] = kPark ? -tsave[(u * 4 + r) * 64 + lane] : -T[u][r];
            }
#pragma unroll
            for (int v = 0; v < kHM; ++v) {
                v4d Tsw;                                    // swap8(T_v)
#pragma unroll
                for (int r = 0; r < 4; ++r) Tsw[r] = kPark ? tsave[(v * 4 + r) * 64 + (lane ^ 8)] : swap8(T[v][r]);
#pragma unroll
                for (int u = 0; u <= v; ++u) red[(v * (v + 1) / 2 + u) * kHStride + lane] = dot4(Tn[u], Tsw);
            }
        }
        if (ft) {
#pragma unroll
            for (int p2 = 0; p2 < kHM / 2; ++p2) red[(R::kAA + p2) * kHStride + lane] = dot4(Q[p2], Db) + dot4(PNn[p2], Sc);
            red[(R::kAA + R::kPair) * kHStride + lane] = (2.0 * c2) * dot4(Y2, Db);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
            const int naa = m * (m + 1) / 2;
            const int half = lane >> 5;
#pragma unroll
            for (int base = 0; base < R::kRows; base += 32) {
                const int row = base + (lane & 31);
                const bool pair_row = row >= R::kAA && row < R::kAA + R::kPair, hh_row = row == R::kAA + R::kPair;
                const int drive = 2 * (row - R::kAA) + half;
                const bool wanted = row < naa || (ft && ((pair_row && drive < m) || (hh_row && half == 0)));
                // the columns of the left (half 0) / right (half 1) lanes: c = 16 i + 8 half + t
                const double* rp = red + (row < R::kRows ? row : 0) * kHStride + 8 * half;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a0 += rp[16 * i] + rp[16 * i + 4];
                    a1 += rp[16 * i + 1] + rp[16 * i + 5];
                    a2 += rp[16 * i + 2] + rp[16 * i + 6];
                    a3 += rp[16 * i + 3] + rp[16 * i + 7];
                }
                const double own = (a0 + a1) + (a2 + a3);
                const double both = own + xor32_f64(own, lane);
                if (wanted) {
                    if (row < naa) {
                        if (half == 0) Hb[P.ho_aa + row] = both;
                    } else {
                        Hb[pair_row ? P.ho_ah + drive : P.ho_hh] = own;
                    }
                }
            }
        }
        // derivative integrators: d2/d(dx_i) dh = -mu_i; alignment padding (explicit zeros)
        if (dfast) {
            int o = P.ho_d;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (lane < P.ddim_i[d]) Hb[o + lane] = -mud[d];
                o += P.ddim_i[d];
            }
            for (int i = lane; i < P.h_pad; i += 64) Hb[P.hess_nnz + i] = 0.0;
        } else {
            qc_hess_tail(P, mu, Hb, lane, 64);
        }
        QC_STAMP(P, b, lane, 8);                  // every store issued
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 9);              // drained
            QC_STAMP_FLUSH(P, b, lane, 0, 11);
        }
        __builtin_amdgcn_wave_barrier();   // the scratch rows are rewritten by the next interval
    } while (!ONCE && (vb += gridDim.x) < h_n_int);
}

}  // namespace

bool qc_mfma_hess_supported(const QcParams& P) {
    if (qc_mfma32_hess_supported(P) || qc_mfma64_hess_supported(P) || qc_mfma16_padeP_hess_supported(P)) return true;
    return P.integrator == QC_PADE && P.p == 2 && P.n <= 16 && P.nc <= 8 && P.m <= kHMmax;
}

hipError_t qc_launch_mfma16_hess_batch(const QcParams& P0, const QcParams* dPb, int count, const double* dZ, const double* dMu, double* dH,
                                       hipStream_t st) {
    const int grid = P0.n_int < 4096 ? P0.n_int : 4096;
    // (the batched launch reads each handle's parameters from device memory; it keeps the general form)
#define QC_B(HM_)                                                                                                                      \
    do {                                                                                                                            \
        if (P0.antisym && (P0.nc != 8 || P0.n != 16))                                                                               \
            hipLaunchKernelGGL((qc_mfma16_pade4_hess_anti_kernel<HM_, true, true, false>), dim3(grid, count), dim3(64), 0, st, QC_HESS_HOT_ARGS(P0), P0, dZ, dMu, dH, dPb); \
        else if (P0.antisym)                                                                                                        \
            hipLaunchKernelGGL((qc_mfma16_pade4_hess_anti_kernel<HM_, false, true, false>), dim3(grid, count), dim3(64), 0, st, QC_HESS_HOT_ARGS(P0), P0, dZ, dMu, dH, dPb); \
        else if (P0.nc != 8 || P0.n != 16)                                                                                               \
            hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM_, true, true>), dim3(grid, count), dim3(64), 0, st, P0, dZ, dMu, dH, dPb); \
        else                                                                                                                        \
            hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM_, false, true>), dim3(grid, count), dim3(64), 0, st, P0, dZ, dMu, dH, dPb); \
    } while (0)
    if (P0.m <= 2) QC_B(2); else if (P0.m <= 4) QC_B(4); else if (P0.m <= 6) QC_B(6); else QC_B(8);
#undef QC_B
    return hipGetLastError();
}

// mu_d2F alone takes the row-gather form of the one-wave kernel wherever the handle's drives allow it (P.ell16): the launch is a latency
// chain per interval, mostly the 68 dependent f64 MFMAs -- 20 with the gathers -- and, holding no drive image through stage B, the form
// keeps its stage-A tiles in registers instead of LDS: 14 KB of LDS and 217 registers, EIGHT workgroups per CU (the dense-image form:
// six; round 4: four).  T = 1000 / 2000 / 4000 / 8000 / 32000: 8.1 / 11.5 / 20.4 / 38.1 / 120 us against 8.7 (two-wave kernel) / 15.7 / 23.9 /
// 42.3 / 140 with the gathers but six per CU, and 8.7 / 15.4 / 25.9 / 47.2 / 173 in round 4 (profiles/r05_hess_long.txt).  QC_HESS_ELL=0: never.
bool qc_mfma16_hess_gathers(const QcParams& P) {
    static const bool off = getenv("QC_HESS_ELL") && atoi(getenv("QC_HESS_ELL")) == 0;
    return !off && P.ell16 != nullptr && P.antisym && P.n == 16 && P.nc == 8 && P.m >= 1 && P.m <= 6 && P.stamps == nullptr;
}

template <int HM, bool KET>
static void launch_hess16(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st, int grid) {
    const bool once = grid == P.n_int;      // one interval per workgroup: the loop-free instantiations
    if (P.antisym) {
        if constexpr (HM == 6 && !KET) {
            if (once && P.stamps != nullptr) {   // diagnostic timeline (QC_STAMPS=1)
                hipLaunchKernelGGL((qc_mfma16_pade4_hess_anti_kernel<HM, KET, false, true, true>), dim3(grid), dim3(64), 0, st, QC_HESS_HOT_ARGS(P), P, dZ, dMu, dH, nullptr);
                return;
            }
        }
        if constexpr (!KET && HM <= 6) {
            if (once && qc_mfma16_hess_gathers(P)) {      // drive generators with one entry per row: the row-gather instantiation (loop-free launches)
                hipLaunchKernelGGL((qc_mfma16_pade4_hess_anti_kernel<HM, KET, false, true, false, true>), dim3(grid), dim3(64), 0, st, QC_HESS_HOT_ARGS(P), P, dZ, dMu, dH, nullptr);
                return;
            }
        }
        if (once) hipLaunchKernelGGL((qc_mfma16_pade4_hess_anti_kernel<HM, KET, false, true>), dim3(grid), dim3(64), 0, st, QC_HESS_HOT_ARGS(P), P, dZ, dMu, dH, nullptr);
        else hipLaunchKernelGGL((qc_mfma16_pade4_hess_anti_kernel<HM, KET, false, false>), dim3(grid), dim3(64), 0, st, QC_HESS_HOT_ARGS(P), P, dZ, dMu, dH, nullptr);
        return;
    }
    if (once) hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM, KET, false, true>), dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH, nullptr);
    else hipLaunchKernelGGL((qc_mfma16_pade4_hess_kernel<HM, KET, false>), dim3(grid), dim3(64), 0, st, P, dZ, dMu, dH, nullptr);
}

hipError_t qc_launch_mfma_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    if (qc_mfma16_padeP_hess_supported(P)) return qc_launch_mfma16_padeP_hess(P, dZ, dMu, dH, st);
    if (P.n > 32) return qc_launch_mfma64_hess(P, dZ, dMu, dH, st);
    if (P.n > 16) return P.ell ? qc_launch_mfma32_ell_hess(P, dZ, dMu, dH, st) : qc_launch_mfma32_hess(P, dZ, dMu, dH, st);
    if (qc_mfma16_hess_g2(P)) return qc_launch_mfma16_hess_g2(P, dZ, dMu, dH, st);       // one entry per drive-generator row: qc_mfma_hess_g2.hip (round 6)
    // two waves per interval up to one round of the device (qc_mfma_hess2.hip) -- unless the drives allow the one-wave kernel's row-gather
    // form, which keeps eight workgroups per CU resident and is faster at every length (T = 750 / 1000: 7.85 / 8.12 against 8.08 / 8.74 us)
    if (qc_mfma16_hess2_supported(P) && !qc_mfma16_hess_gathers(P)) return qc_launch_mfma16_hess2(P, dZ, dMu, dH, st);
    // One interval per workgroup at any length (round 5: the loop-free instantiation needs 228 registers and 25.3 KB of LDS at six
    // drives -- six workgroups per CU, the hardware refilling each CU as its workgroups retire; the persistent loop's instantiation
    // needs 308 registers, four per CU).  T = 3000 / 4000 / 8000 / 32000: 20.1 / 25.8 / 45.9 / 156.9 us against 20.8 / 25.9 / 47.0 / 172.8
    // with the persistent grid of 1024 (profiles/r05_hess_long.txt).  QC_HESS_ONCE_MAX=2048 QC_HESS_GRID=<n>: the persistent grid beyond 2048 intervals.
    static const int grid_cap = getenv("QC_HESS_GRID") ? std::max(1, atoi(getenv("QC_HESS_GRID"))) : 1024;
    static const int once_max = getenv("QC_HESS_ONCE_MAX") ? atoi(getenv("QC_HESS_ONCE_MAX")) : (1 << 30);
    // Six workgroups per CU live 8.2 instead of 6.5 - 7.4 us each: between 1.5 and 2 rounds of the device that is a loss (two rounds either
    // way: T = 2000 17.7 against 15.4 us) -- there the persistent instantiation (four per CU) with one round of workgroups stays.
    const bool window = P.antisym && P.m > 4 && P.m <= 6 && P.n_int > 1536 && P.n_int <= 2048 && !qc_mfma16_hess_gathers(P);   // (the row-gather form: eight per CU, one round)
    const int grid = (P.n_int <= once_max && !window) ? P.n_int : (P.n_int < grid_cap ? P.n_int : grid_cap);
    if (P.nc != 8 || P.n != 16) {
        if (P.m <= 2) launch_hess16<2, true>(P, dZ, dMu, dH, st, grid);
        else if (P.m <= 4) launch_hess16<4, true>(P, dZ, dMu, dH, st, grid);
        else if (P.m <= 6) launch_hess16<6, true>(P, dZ, dMu, dH, st, grid);
        else launch_hess16<8, true>(P, dZ, dMu, dH, st, grid);
        return hipGetLastError();
    }
    if (P.m <= 2) launch_hess16<2, false>(P, dZ, dMu, dH, st, grid);
    else if (P.m <= 4) launch_hess16<4, false>(P, dZ, dMu, dH, st, grid);
    else if (P.m <= 6) launch_hess16<6, false>(P, dZ, dMu, dH, st, grid);
    else launch_hess16<8, false>(P, dZ, dMu, dH, st, grid);
    return hipGetLastError();
}
