// Register-resident f64-MFMA kernels (v_mfma_f64_16x16x4_f64) for the order-4 Pade integrator.
//
// n = 2N = 16 (3 qubits, BASELINE configs 3 and 4): TWO WAVEFRONTS PER INTERVAL, one LDS hand-off, one barrier.
//   wave 1 ("copy wave")    issues EVERY global load of the interval in one batch (generator images, both knots,
//                           derivative-integrator data), assembles G, hands G / U_t / U_t+1 / the G_j images to the
//                           compute wave through LDS, then G -> (G^2)^T -> B^T, F^T -> the 2N tile stores of the
//                           I_N (x) B / -I_N (x) F blocks (80 % of the interval's bytes: this wave lives in the store queue)
//   wave 0 ("compute wave") no global loads (one issued during the copy burst waits > 3 us); residual, d/dh and the m
//                           drive columns as four batches of independent, interleaved MFMA products; its 16 stores are
//                           issued only after the whole chain (a store issued mid-burst stalls the wave for microseconds)
// so the bandwidth-bound copies of one wave overlap the MFMA chains of the other
// (profiles/r01_mfma_v2_timeline.txt shows the serial one-wave version: 3.4 us prologue, 4.4 us store-bound,
// 5.3 us MFMA-latency-bound; profiles/r01_mfma_v4_timeline.txt this one).
//
// Every 16x16x16 product is 4 MFMAs whose operands never leave the register file:
//   * A-operand layout of a 16x16 matrix X:  lane (g = l>>4, i = l&15), reg kk holds X[i][4kk+g]
//   * B-operand layout == C/D layout:        lane (g, j = l&15),        reg r  holds X[4r+g][j]
//     so a product's result is directly the B operand of the next left-multiplication, and the
//     C/D registers of X read as an A operand are X^T: "mm16(X, I)" transposes a tile in 4 MFMAs.
//   * Two n x N (16 x 8) matrices share one 16-column tile; the halves are exchanged with a DPP
//     row_ror:8 (no LDS, no cross-wave traffic).
//   * Output tiles are produced TRANSPOSED (lane <-> row): register r of the wave is then the 512
//     contiguous bytes of columns 4r..4r+3 in column-major memory, i.e. every global store
//     instruction writes four whole 128-byte lines.  B^T and F^T come for free from
//     (G^2)^T = mm16(G_B, G_A); the n x N outputs take one identity product each.
//   * The constant generators G_0, G_j are read from a lane-ordered A-layout image in global memory
//     (14 KB for m = 6: L2/Infinity-Cache resident), 16 bytes per lane per load, once per interval.
//   * Store flavour (non-temporal) and diagnostics are template parameters: as run-time switches they put a scalar
//     branch ladder around every store and halved the store issue rate.
//
// Per interval (S = U1+U0, D = U1-U0, h = dt):                                          MFMAs
//   copy wave:    G_B = G I ; (G^2)^T = G^T G^T ; B^T, F^T = I -+ h/2 G^T + h^2/12 (G^2)^T      8
//   compute wave: P1 = G [S | D] = [GS | GD] ;  P2 = G [GD | GS] = [G^2 D | .]                   8
//                 E  = [delta | d/dh] = [D - h/2 GS + h^2/12 G^2 D | -1/2 GS + h/6 G^2 D],  E^T  4
//                 Q  = [ -h/2 S + h^2/12 GD | h^2/12 D ]
//                 per drive pair (j, j+1):  R_j = G_j Q, R_j+1 = G_j+1 Q,
//                      T = G [R_j(right) | R_j+1(right)],
//                      Y = [R_j(left) + T(left) | R_j+1(left) + T(right)] = [d/da_j | d/da_j+1], Y^T   16
// = 20 + 8 m MFMAs per interval (68 for m = 6).
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kIntervalsPerWG = 1;         // interval pairs per workgroup: waves [0, IPW) compute, [IPW, 2 IPW) copy.
                                           // Measured on MI355X (bench.py, config 3): IPW 1 -> 11.55 us, 2 -> 11.60 us,
                                           // 4 -> 12.0 us per evaluation (the wider barrier couples four intervals); with the
                                           // hand-off through one LDS flag per wave pair instead of the workgroup barriers
                                           // (pairs independent of each other) 11.34 / 11.75 / 12.04 us against 11.17 us.
constexpr int kThreads = 128 * kIntervalsPerWG;
constexpr int kMaxGrid = 1024;             // persistent beyond this many workgroups

// A-layout image of generator `mat` (0 = drift): [matrix][pair(2)][lane(64)][2] doubles
__device__ inline v4d load_GA(const double* __restrict__ Gx, int mat, int lane) { return load_image_tile(Gx + mat * 256, lane); }

constexpr int kDF = 4;   // derivative integrators handled from registers in the copy wave


// G = G_0 + sum_k a_k G_k (A-layout).  The first kMU drive images and the amplitudes are requested in one batch (no load
// waits on another) and returned in gk/ak for reuse by the drive-column loop.  Two halves, so that a caller can put other
// load requests between them: assemble_G_request issues the loads, assemble_G_combine waits for them and forms G.
template <int kMU>
__device__ inline void request_images(int m, const double* __restrict__ Gx, int lane, v4d& g0, v4d (&gk)[kMU]) {
    g0 = load_GA(Gx, 0, lane);
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
        const int k = u < m ? u : (m > 0 ? m - 1 : 0);     // clamped: the load is unconditional
        gk[u] = load_GA(Gx, m > 0 ? k + 1 : 0, lane);
    }
}
template <int kMU>
__device__ inline void assemble_G_request(const QcParams& P, const double* __restrict__ Gx, const double* __restrict__ z0, int lane,
                                          v4d& g0, v4d (&gk)[kMU], double& av) {
    av = load_amp_lanes(z0, P.off_a, P.m, lane);   // every amplitude in one vector load (qc_mfma_common.h)
    request_images(P.m, Gx, lane, g0, gk);
}
template <int kMU>
__device__ inline v4d assemble_G_combine(const QcParams& P, const double* __restrict__ Gx, int lane, const v4d& g0, const v4d (&gk)[kMU],
                                         double av, double (&ak)[kMU]) {
    const int m = P.m;
    v4d Ga = g0;
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
        ak[u] = (u < m) ? bcast_lane(av, u) : 0.0;
        Ga += ak[u] * gk[u];
    }
    for (int k = kMU; k < m; ++k) Ga += bcast_lane(av, k) * load_GA(Gx, k + 1, lane);
    return Ga;
}
template <int kMU>
__device__ inline v4d assemble_G(const QcParams& P, const double* __restrict__ Gx, const double* __restrict__ z0,
                                 int lane, v4d (&gk)[kMU], double (&ak)[kMU]) {
    v4d g0;
    double av;
    assemble_G_request(P, Gx, z0, lane, g0, gk, av);
    return assemble_G_combine(P, Gx, lane, g0, gk, av, ak);
}

// Store a transposed tile: lane (g, j) reg r holds X[j][4r+g] of a 16 x 16 column-major block at p.
template <int MODE>
__device__ inline void store_tile_T(double* __restrict__ p, const v4d& x, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8m<MODE>(p + (4 * r + g) * 16 + j, x[r]);
}

// LDS hand-off block of one workgroup (doubles): the copy wave loads everything the interval needs from
// global memory BEFORE any store of the workgroup is issued, and passes it on; the compute wave issues no
// global load at all, so its MFMA chain never waits behind the store burst in the memory pipeline.
// The same for the masked instantiation (KET): an nr x nr block (nr <= 16 rows per column), only rows / columns < nr
template <int MODE>
__device__ inline void store_tile_T_masked(double* __restrict__ p, const v4d& x, int nr, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * r + g < nr && j < nr) qc_st8m<MODE>(p + (4 * r + g) * nr + j, x[r]);
}
// State tile [U | U] (both 8-column halves): lane (g, j) reg r = U[4r+g][col].  MASK: the state has nr <= 16 rows per column
// (systems with N < 8 levels are zero-padded to the 16 x 16 tile) and columns >= nc re-read column 0 (never stored).
template <bool MASK>
__device__ inline v4d load_state_tile(const double* __restrict__ zU, int col, int nr, int g) {
    if constexpr (!MASK) {
        // (As two 16-byte requests per lane + lane-group transposes, load_col16_T, which gains 0.27 us in the Hessian kernel: 9.14 -
        //  9.18 against 8.87 - 8.95 us here, same run.  These loads are not what the first store waits for.)
        const double* p = zU + col * 16 + g;
        return v4d{p[0], p[4], p[8], p[12]};
    } else {
        const double* p = zU + col * nr;
        v4d v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (4 * r + g < nr) ? p[4 * r + g] : 0.0;
        return v;
    }
}

constexpr int kLdsGa = 0, kLdsU0 = 256, kLdsU1 = 512, kLdsGk = 768;   // + MU * 256 generator images

__device__ inline void lds_put(double* __restrict__ base, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}
__device__ inline v4d lds_get(const double* __restrict__ base, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(base) + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}

// BATCH: blockIdx.y selects one of several handles' parameter blocks in device memory (the systems of a sampling
// problem: same shapes, different generators and output slots) so that they share ONE launch; the parameters are then
// read with scalar loads from global memory instead of the kernarg segment, everything else is identical.
// ONCE: one interval per workgroup (grid = number of workgroups needed, no persistent loop).  With the loop, the compiler hoists
// every loop-invariant mask, address and predicate (~350 instructions, SGPRs spilled to VGPR lanes) in front of it, i.e. in
// front of the FIRST GLOBAL LOAD of the copy wave: 1.6 us between a wave's first instruction and its first load request
// (profiles/r02_stamps_d.txt).  Without the loop the loads are scheduled first and the set-up runs under their latency.
// QcHot: what the first load requests of a wave depend on, as the FIRST kernel arguments -- this file is compiled with
// -amdgpu-kernarg-preload-count, which has the hardware place the leading argument dwords in scalar registers while the wave is
// launched, so the image and amplitude requests do not wait for a scalar load from the argument block (the by-value parameter
// block, 0.55 KB, follows; it is read under the latency of those requests).
// (Scalars and pointers only: a by-value struct is passed by reference and is not preloaded.)
#define QC_HOT_ARGS(P) (P).Gx, dZ, (P).t_begin, (P).n_int, (P).zdim, (P).off_a, (P).off_dt, (P).m

template <bool JAC, int MODE, bool DIAG, int kMU, bool KET, bool BATCH, bool ONCE = false>
__global__ __launch_bounds__(JAC ? kThreads : 64, 2) void qc_mfma16_pade4_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ Z, long long hot_t_begin, int hot_n_int,
                                                                      int hot_zdim, int hot_off_a, int hot_off_dt, int hot_m, const QcParams Pk,
                                                                      double* __restrict__ F, double* __restrict__ J,
                                                                      const QcParams* __restrict__ Pb) {
    const QcParams& P = BATCH ? Pb[blockIdx.y] : Pk;
    const int h_n_int = BATCH ? P.n_int : hot_n_int, h_zdim = BATCH ? P.zdim : hot_zdim, h_off_a = BATCH ? P.off_a : hot_off_a;
    const int h_off_dt = BATCH ? P.off_dt : hot_off_dt;
    const long long h_t_begin = BATCH ? P.t_begin : hot_t_begin;
    constexpr int kLdsBlock = kLdsGk + kMU * 256;
    __shared__ __attribute__((aligned(16))) double sm_all[JAC ? kIntervalsPerWG * kLdsBlock : 2];
    unsigned long long t_entry = 0, t_kernarg = 0;
    if constexpr (DIAG) {
        t_entry = __builtin_amdgcn_s_memrealtime();
        // a stamp that cannot issue before one field of (nearly) every line of the argument block has arrived
        const long long dep = (long long)Pk.N + Pk.n_deriv + Pk.drow[7] + Pk.jac_nnz + Pk.jo_d + Pk.ho_d + (long long)Pk.G + (long long)Pk.stamps +
                              (long long)Z + (long long)Pb + (long long)(Pk.c[2] != 0.0) + (long long)(Pk.dt_fixed != 0.0);
        asm volatile("s_memrealtime %0" : "=s"(t_kernarg) : "s"(dep));
    }
    // One dword of every line of the argument block is REQUESTED here (one batch of scalar-cache misses instead of one per use,
    // qc_internal.h) and waited for behind the image requests, which depend on preloaded arguments only.
#ifndef QC_NO_KERNARG_TOUCH
    QcKernargTouch<sizeof(QcParams) + 96> touch;
    touch.request();
#endif
    // (Demanding every argument of the prologue at one point -- one batch of scalar loads -- was measured: 10.0 vs 9.6 us.)
    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wave = JAC ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    // 0 compute wave, 1 copy wave (the first wave of the workgroup computes).  Swapping the roles in every other workgroup, so
    // that the compute waves of a CU's workgroups do not share SIMDs, was measured (blockIdx bits 0, 3, 8, 9): 10.4 - 11.3 us
    // against 10.4 us -- the matrix pipes are not what the launch waits for.
    const int role = JAC ? 1 - wave / kIntervalsPerWG : 0;
    const int slot = wave % kIntervalsPerWG;      // which of the workgroup's intervals
    double* __restrict__ sm = sm_all + (JAC ? slot * kLdsBlock : 0);
    const int ipw = JAC ? kIntervalsPerWG : 1;
    const int n_wg = (h_n_int + ipw - 1) / ipw;
    const int m = BATCH ? P.m : hot_m;
    // KET = false: a unitary on N = 8 levels, every mask below folds away at compile time (as run-time tests they cost the
    // headline kernel 0.8 us per launch).  KET = true: the masked instantiation -- K <= 8 state columns (kets) and / or
    // N < 8 levels (2N = nr < 16 rows, zero-padded to the tile).
    const int nc = KET ? P.nc : 8;
    const int nr = KET ? P.n : 16;
    const bool ft = h_off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gx = BATCH ? P.Gx : hot_Gx;

    // The copy wave's generator images depend on nothing but the kernel arguments: they are requested before any address of the
    // interval is computed (and once for all intervals of a persistent grid).
    v4d g0_img, gk_img[kMU];
    if (JAC && role == 1) request_images(m, Gx, lane0, g0_img, gk_img);
#ifndef QC_NO_KERNARG_TOUCH
    touch.consume();   // (consumed behind the amplitude / timestep requests as well, with the fixed timestep preloaded too so that no
                       //  scalar wait stands in front of them: 8.52 - 8.66 against 8.49 - 8.56 us, same run -- not better)
#endif
    int vb = blockIdx.x;
    if (vb >= n_wg) return;   // (never: the grid has at most n_wg workgroups)
    do {
        // Persistent grids: an opaque copy of the lane index per pass -- what derives from it (tile masks, LDS and store offsets,
        // the identity tile) is computed where it is used instead of being hoisted out of the loop and held, at this kernel's
        // 256-register budget spilled, through the products (22 of the 96 instantiations kept 20 - 260 bytes of scratch).
        int lane = lane0;
        if constexpr (!ONCE) asm volatile("" : "+v"(lane));
        const int g = lane >> 4, j = lane & 15, jj = j & 7;
        const int jc = (!KET || jj < nc) ? jj : 0;
        const bool left = j < 8;
        const v4d IdB = identity_B(g, j);
        const int b_raw = qc_xcd_remap(vb, n_wg) * ipw + slot;   // local interval of this wave pair
        const bool active = b_raw < h_n_int;                      // the last workgroup may be partly empty;
        const int b = active ? b_raw : h_n_int - 1;               // its idle waves still take part in the barriers
        const long long t = h_t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)h_zdim;
        const double* __restrict__ z1 = z0 + h_zdim;
        // what G depends on besides the images -- the amplitudes and the timestep -- is requested as soon as the knot's address
        // exists, before the output addresses are computed (both waves: the compute wave needs the timestep too)
        const double av_pre = (JAC && role == 1) ? load_amp_lanes(z0, h_off_a, m, lane) : 0.0;
        const double h_pre = ft ? load_uniform(z0 + h_off_dt) : opaque_scalar(P.dt_fixed);
        double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
        double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
        QC_STAMP_DECL;

        if (JAC && role == 1) {
            // ================= copy wave =====================================================================
            if (!active) { __syncthreads(); if constexpr (!ONCE) __syncthreads(); continue; }
            __builtin_amdgcn_s_setprio(3);   // critical path: nothing reaches HBM before this wave's first store
            if constexpr (DIAG) {
                qc_ts_[9] = t_entry;
                if (!(P.dbg_skip & 4)) { qc_ts_[10] = t_kernarg; qc_ts_[11] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | (1ull << 40); }   // HW_REG_HW_ID: where the wave runs
            }
            QC_STAMP(P, b, lane, 0);
            // Every global load of the interval in one batch, in the order the wave needs them: first what G depends on (the
            // timestep, the amplitudes, the generator images), then the state tiles and the derivative-integrator data that are only
            // passed on.  G, its two products and B^T / F^T are formed as soon as the first group is back (the compiler's counted
            // vmcnt leaves the second group in flight); the hand-off to the compute wave follows, then the stores.
            const double h = h_pre;
            double ak[kMU];
            const double av = av_pre;   // every amplitude in one vector load (qc_mfma_common.h)
            const v4d& g0 = g0_img;
            const v4d (&gk)[kMU] = gk_img;
            // (Dropping these twenty loads altogether -- wrong results, timing only -- gains 0.23 us: the vector-memory pipeline of
            // the CU is not what the first store waits for; it waits for the round trip of the amplitudes and the images.)
            // (The compute wave fetching the state tiles itself, so that the hand-off need not wait for them: 9.78 vs 9.17 us.)
            const v4d u0 = load_state_tile<KET>(z0 + P.off_U, jc, nr, g);
            const v4d u1 = load_state_tile<KET>(z1 + P.off_U, jc, nr, g);
            double dxv[kDF], dfv[kDF];       // derivative integrators, register fast path (<= kDF of <= 64 rows)
            const bool dfast = P.n_deriv <= kDF;
#pragma unroll
            for (int d = 0; d < kDF; ++d) {
                // unused slots (d >= n_deriv) have zero offsets/dims in QcParams: the loads stay in bounds
                const int i = lane < P.ddim_i[d] ? lane : 0;
                dxv[d] = z0[P.dx_off[d] + i];
                dfv[d] = z1[P.x_off[d] + i] - z0[P.x_off[d] + i];
            }
            if (DIAG && (P.dbg_skip & 4)) {   // QC_DEBUG_SKIP=4: when do the scalar loads (amplitudes, h) and when do the vector loads arrive?
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                QC_STAMP(P, b, lane, 10);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                QC_STAMP(P, b, lane, 11);
            }
            const v4d Ga = assemble_G_combine(P, Gx, lane, g0, gk, av, ak);
            const double hc1 = h * c1, hc2 = h * h * c2;
            QC_STAMP(P, b, lane, 1);
            bool skip = false;
            if constexpr (DIAG) skip = (P.dbg_skip & 1) != 0;
            // A-layout(G^T) = B-layout(G) = Gb;  B-layout(G^T) = D-layout(G^T) = A-layout(G) = Ga.
            const v4d Gb = mm16(Ga, IdB);
            const v4d G2T = mm16(Gb, Ga);
            v4d Fm, Bm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ev = IdB[r] + hc2 * G2T[r];
                Fm[r] = -(ev + hc1 * Ga[r]);       // -F^T
                Bm[r] = ev - hc1 * Ga[r];          //  B^T
            }
            // The first kEarly copies of -F^T / B^T leave BEFORE the hand-off: the hand-off waits for the state tiles (the last loads of
            // the batch) and for the barrier, and nothing reaches HBM before this wave's first store.  Measured, copies stored
            // first -> step: 0 -> 9.42-9.66 us, 1 -> 9.36, 3 -> 9.18-9.34, 4 -> 8.95-9.37, 5 -> 9.57-9.69, 6 -> 10.0, 8 -> 10.6
            // (beyond half of them the compute wave starts too late).  Only in the loop-free unitary instantiation: in the persistent
            // and the masked ones it measured slower (T = 2000: 17.7 vs 17.15 us, config 2: 5.8 vs 5.2-5.5 us).  With the early copies
            // this wave drains a microsecond before the compute wave; giving it the residual / d/dh block (12 MFMAs) in return made
            // the launch slower (9.45 vs 9.05 us): its MFMAs share a SIMD with another workgroup's compute wave.
#ifndef QC_EARLY_COPIES
#define QC_EARLY_COPIES 4
#endif
            constexpr int kEarly = (!KET && ONCE) ? QC_EARLY_COPIES : 0;
            if constexpr (kEarly > 0) {
                if (!skip) {
#pragma unroll
                    for (int q = 0; q < kEarly; ++q) {
                        if (q < P.copies) {
                            store_tile_T<MODE>(Jb + P.jo_F + q * 256, Fm, g, j);
                            store_tile_T<MODE>(Jb + P.jo_B + q * 256, Bm, g, j);
                        }
                    }
                }
            }
            // hand-off to the compute wave
            lds_put(sm + kLdsGa, lane, Ga);
            lds_put(sm + kLdsU0, lane, u0);
            lds_put(sm + kLdsU1, lane, u1);
#pragma unroll
            for (int u = 0; u < kMU; ++u)
                if (u < m) lds_put(sm + kLdsGk + u * 256, lane, gk[u]);
            __syncthreads();
            if (!skip) {
                double* pF = Jb + P.jo_F;
                double* pB = Jb + P.jo_B;
                // 8 bytes per lane, 512 contiguous bytes per instruction.  (Pairing lanes for 16-byte stores measured 8 % slower.)
                // P.copies = N copies of each block (I_N (x) B); 1 when the host path asks for the compact form
                const int ncop = P.copies;
                for (int q = kEarly; q < ncop; ++q) {
                    if constexpr (!KET) {
                        store_tile_T<MODE>(pF + q * 256, Fm, g, j);
                        store_tile_T<MODE>(pB + q * 256, Bm, g, j);
                    } else {
                        store_tile_T_masked<MODE>(pF + q * nr * nr, Fm, nr, g, j);
                        store_tile_T_masked<MODE>(pB + q * nr * nr, Bm, nr, g, j);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                QC_STAMP(P, b, lane, 2);
                {   // derivative integrator rows: residual x_{t+1} - x_t - h dx_t and the 4 (3) diagonal blocks
                    int jo = P.jo_d;
                    bool all_fast = dfast;
#pragma unroll
                    for (int d = 0; d < kDF; ++d) {
                        if (d < P.n_deriv) {
                            const int dim = P.ddim_i[d], r0 = P.drow[d];
                            if (dfast && dim <= 64) {
                                if (lane < dim) {
                                    if (Fb) Fb[r0 + lane] = dfv[d] - h * dxv[d];
                                    Jb[jo + lane] = -1.0;
                                    Jb[jo + dim + lane] = 1.0;
                                    Jb[jo + 2 * dim + lane] = -h;
                                    if (ft) Jb[jo + 3 * dim + lane] = -dxv[d];
                                }
                            } else {
                                all_fast = false;
                            }
                            jo += (ft ? 4 : 3) * dim;
                        }
                    }
                    if (!all_fast) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, dfast);
                }
                if constexpr (DIAG) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    QC_STAMP(P, b, lane, 3);
                }
                QC_STAMP_FLUSH(P, b, lane, 0, 3);
                QC_STAMP_FLUSH(P, b, lane, 9, 11);
            }
            if constexpr (!ONCE) __syncthreads();   // the hand-off block is rewritten by the next interval of a persistent grid
            continue;
        }

        // ===================== compute wave ===========================================================
        if (!active) { if constexpr (JAC) { __syncthreads(); if constexpr (!ONCE) __syncthreads(); } continue; }
        __builtin_amdgcn_s_setprio(1);
        QC_STAMP(P, b, lane, 4);
        QC_STAMP_CYCLES(13);
        const double h = h_pre;   // requested at the top, back long before the hand-off arrives
        v4d u0, u1, Ga;
        v4d gk[kMU];
        if constexpr (JAC) {
            __syncthreads();                      // wait for the copy wave's hand-off
            Ga = lds_get(sm + kLdsGa, lane);
            u0 = lds_get(sm + kLdsU0, lane);
            u1 = lds_get(sm + kLdsU1, lane);
        } else {                                  // residual-only launch: a single wave, loads for itself
            u0 = load_state_tile<KET>(z0 + P.off_U, jc, nr, g);
            u1 = load_state_tile<KET>(z1 + P.off_U, jc, nr, g);
            double ak[kMU];
            Ga = assemble_G(P, Gx, z0, lane, gk, ak);
        }
        const double hc1 = h * c1, hc2 = h * h * c2;
        QC_STAMP(P, b, lane, 5);
        bool skipc = false;
        if constexpr (DIAG) skipc = (P.dbg_skip & 2) != 0;
        if (!skipc) {
            v4d W, Wsw;                               // W = [S | D], Wsw = [D | S]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double sm_ = u1[r] + u0[r], df = u1[r] - u0[r];
                W[r] = left ? sm_ : df;
                Wsw[r] = left ? df : sm_;
            }
            // The products are issued in four dependency stages, each a batch of independent 16x16x16 products
            // whose MFMAs are interleaved (mm16_multi):  A: P1   B: P2, R_0..R_{MU-1}   C: E^T, T_0..T_{MU/2-1}   D: Y_p^T.
            const v4d P1 = mm16(Ga, W);               // [GS | GD]
            const v4d P1sw = swap8(P1);               // [GD | GS]
            v4d Q;                                    // [Q0 | Q1] = [-h c1 S + h^2 c2 GD | h^2 c2 D]
#pragma unroll
            for (int r = 0; r < 4; ++r) Q[r] = left ? (-hc1 * W[r] + hc2 * P1sw[r]) : (hc2 * W[r]);
            constexpr int NB = JAC ? kMU + 1 : 1;
            v4d sB[NB];                               // stage B results: P2 = G P1sw, R_k = G_k Q = [G_k Q0 | G_k Q1]
            {
                v4d aB[NB], bB[NB];
                aB[0] = Ga;
                bB[0] = P1sw;
                if constexpr (JAC) {
#pragma unroll
                    for (int u = 0; u < kMU; ++u) {   // an unused slot (u >= m) repeats the last drive; never stored
                        aB[u + 1] = lds_get(sm + kLdsGk + (u < m ? u : m - 1) * 256, lane);
                        bB[u + 1] = Q;
                    }
                }
                mm16_multi<NB>(aB, bB, sB);
            }
            const v4d P2 = sB[0];                     // [G^2 D | G^2 S]
            QC_STAMP(P, b, lane, 6);
            v4d E;                                    // [delta | d/dh]: formed on the left half, d/dh moved to the right
            {
                v4d dl, dh;
                const double d1 = -c1, d2 = 2.0 * c2 * h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dl[r] = Wsw[r] - hc1 * P1[r] + hc2 * P2[r];
                    dh[r] = d1 * P1[r] + d2 * P2[r];
                }
                const v4d dhs = swap8(dh);
#pragma unroll
                for (int r = 0; r < 4; ++r) E[r] = left ? dl[r] : dhs[r];
            }
            // JAC: the transposes for the line-wide stores (E^T and the drive-pair tiles) go through the hand-off block in LDS,
            // which nobody reads any more after stage B, instead of through identity products: 4 + 2 kMU MFMAs (a quarter of the
            // wave's matrix-pipe time) off the dependent chain, one LDS round trip for all tiles (lds_transpose16_multi).
            constexpr int NC = JAC ? kMU / 2 + 1 : 1;
            v4d sC[NC];                               // stage C results: (E^T,) T_p = G [R_2p(right) | R_2p+1(right)]
            v4d Y[NC];                                // Y[p+1] = [R_2p(left) | R_2p+1(left)]
            {
                v4d aC[NC], bC[NC];
                aC[0] = E;                            // C/D registers read as an A operand are the transpose
                bC[0] = IdB;
                if constexpr (JAC) {
#pragma unroll
                    for (int p2 = 0; p2 < kMU / 2; ++p2) {
                        const v4d R1 = sB[2 * p2 + 1], R2 = sB[2 * p2 + 2];
                        const v4d R1sw = swap8(R1), R2sw = swap8(R2);
                        aC[p2 + 1] = Ga;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            bC[p2 + 1][r] = left ? R1sw[r] : R2[r];      // [G_k Q1 | G_k+1 Q1]
                            Y[p2 + 1][r] = left ? R1[r] : R2sw[r];       // [G_k Q0 | G_k+1 Q0]
                        }
                    }
                }
                if constexpr (JAC) {                  // T_p only; E is transposed through LDS below
                    if constexpr (kMU / 2 > 0) {
                        v4d aT[kMU / 2], bT[kMU / 2], dT[kMU / 2];
#pragma unroll
                        for (int p2 = 0; p2 < kMU / 2; ++p2) { aT[p2] = aC[p2 + 1]; bT[p2] = bC[p2 + 1]; }
                        mm16_multi<kMU / 2>(aT, bT, dT);
#pragma unroll
                        for (int p2 = 0; p2 < kMU / 2; ++p2) sC[p2 + 1] = dT[p2];
                    }
                } else {
                    mm16_multi<NC>(aC, bC, sC);
                }
            }
            v4d ET;                                   // lane (g, j) reg r = E[j][4r+g]
            v4d YT[kMU / 2 > 0 ? kMU / 2 : 1];        // transposes of [d/da_k | d/da_k+1]
            if constexpr (JAC) {
                v4d tin[kMU / 2 + 1], tout[kMU / 2 + 1];
                tin[0] = E;
#pragma unroll
                for (int p2 = 0; p2 < kMU / 2; ++p2) tin[p2 + 1] = Y[p2 + 1] + sC[p2 + 1];
                lds_transpose16_multi<kMU / 2 + 1>(sm, tin, tout, g, j);
                ET = tout[0];
#pragma unroll
                for (int p2 = 0; p2 < kMU / 2; ++p2) YT[p2] = tout[p2 + 1];
            } else {
                ET = sC[0];
            }
            // The compute wave's few stores are issued AFTER its MFMA chain: a store issued while the copy
            // waves flood the CU's store queue stalls this wave for microseconds.
            auto store_ET = [&]() {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * r + g;          // tile column: < 8 residual column c, >= 8 d/dh column c-8
                    if (c < 8) {
                        if (Fb && (!KET || (c < nc && j < nr))) qc_st8m<MODE>(Fb + c * nr + j, ET[r]);
                    } else if (JAC && ft && (!KET || (c - 8 < nc && j < nr))) {
                        qc_st8m<MODE>(Jb + P.jo_h + (c - 8) * nr + j, ET[r]);
                    }
                }
            };
            QC_STAMP(P, b, lane, 7);
            if constexpr (!JAC) {   // residual-only launch has no copy wave: derivative residual rows here
                store_ET();
                deriv_rows_generic(P, z0, z1, h, Fb, nullptr, lane, false);
            } else {
                // ---- drive columns: d/da_k = G_k Q0 + G (G_k Q1), two drives per tile ----------------------------
                double* pa = Jb + P.jo_a;
                auto store_pair = [&](int k, const v4d& YTk) {
                    const bool two = k + 1 < m;
                    const int sk = KET ? P.s : 128;
                    double* p = pa + (size_t)k * sk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * r + g;      // tile columns < 8: drive k column c; >= 8: drive k+1 column c-8
                        if (r < 2) { if (!KET || (c < nc && j < nr)) qc_st8m<MODE>(p + c * nr + j, YTk[r]); }
                        else if (two && (!KET || (c - 8 < nc && j < nr))) qc_st8m<MODE>(p + sk + (c - 8) * nr + j, YTk[r]);
                    }
                };
                QC_STAMP(P, b, lane, 8);
                store_ET();
#pragma unroll
                for (int u = 0; u < kMU; u += 2)
                    if (u < m) store_pair(u, YT[u >> 1]);
                // drives beyond the hand-off block: generator images straight from global memory
                for (int k = kMU; k < m; k += 2) {
                    const v4d R1 = mm16(load_GA(Gx, k + 1, lane), Q);
                    const v4d R2 = mm16(load_GA(Gx, k + 2 <= m ? k + 2 : k + 1, lane), Q);
                    const v4d R1sw = swap8(R1), R2sw = swap8(R2);
                    v4d X, Yk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        X[r] = left ? R1sw[r] : R2[r];
                        Yk[r] = left ? R1[r] : R2sw[r];
                    }
                    Yk += mm16(Ga, X);
                    store_pair(k, mm16(Yk, IdB));
                }
            }
            if constexpr (DIAG) {   // diagnostic: time until this wave's stores have left the CU
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                QC_STAMP(P, b, lane, 12);
                QC_STAMP_CYCLES(14);
            }
            if constexpr (DIAG) qc_ts_[15] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | (1ull << 40);   // HW_REG_HW_ID
            QC_STAMP_FLUSH(P, b, lane, 4, 8);        // (slots 9 - 11 are the copy wave's: kernel entry, arguments read, HW_ID)
            QC_STAMP_FLUSH(P, b, lane, 12, 15);
        }
        if constexpr (JAC && !ONCE) __syncthreads();   // pairs with the copy wave's end-of-interval barrier
    } while (!ONCE && (vb += gridDim.x) < n_wg);
}

}  // namespace

bool qc_mfma_compact_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && !qc_mfma16_padeP_supported(P) && ((P.n <= 16 && P.nc <= 8) || (P.n <= 32 && P.nc <= 16)) && P.m <= 32;
}

bool qc_mfma_supported(const QcParams& P) {
    if (qc_mfma_exp_supported(P) || qc_mfma32_exp_supported(P) || qc_mfma64_supported(P) || qc_mfma16_padeP_supported(P)) return true;
    return P.integrator == QC_PADE && P.p == 2 && ((P.n <= 16 && P.nc <= 8) || (P.n <= 32 && P.nc <= 16)) && P.m <= 32;
}

size_t qc_mfma_gx_doubles(const QcParams& P) {
    if (P.n > 32) return qc_mfma64_gx_doubles(P);
    return P.n > 16 ? qc_mfma32_gx_doubles(P) : (size_t)2 * (P.m + 1) * 256;
}

// Packs the (m+1) generators (column-major n x n, index 0 = drift) into the lane-ordered A-layout
// images [layout][matrix][pair][lane][2]:  layout 0 (A operand of X): lane (g, i) reg kk = X[i][4kk+g];
// layout 1 (B layout of X = A operand of X^T, used by the Hessian kernel): lane (g, i) reg kk = X[4kk+g][i].
void qc_mfma_pack_G(const QcParams& P, const double* G, double* Gx) {
    if (P.n > 32) { qc_mfma64_pack_G(P, G, Gx); return; }
    if (P.n > 16) { qc_mfma32_pack_G(P, G, Gx); return; }
    const int n = P.n, M = P.m + 1;      // n < 16: the images are zero-padded to the 16 x 16 tile
    for (int mat = 0; mat < M; ++mat) {
        const double* A = G + (size_t)mat * n * n;
        for (int pr = 0; pr < 2; ++pr)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 2; ++e) {
                    const int g = l >> 4, i = l & 15, kk = 2 * pr + e, c = 4 * kk + g;
                    const bool in = i < n && c < n;
                    Gx[((size_t)mat * 2 + pr) * 128 + l * 2 + e] = in ? A[(size_t)c * n + i] : 0.0;                 // X[i][4kk+g]
                    Gx[((size_t)(M + mat) * 2 + pr) * 128 + l * 2 + e] = in ? A[(size_t)i * n + c] : 0.0;         // X[4kk+g][i]
                }
    }
}

template <bool JAC, bool DIAG, int MU>
static void launch16m(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st, int grid, int threads) {
    if (P.nc != 8 || P.n != 16) {   // K < 8 kets and / or N < 8 levels: the masked instantiation (non-temporal stores, no diagnostics)
        const int n_wg_k = JAC ? (P.n_int + kIntervalsPerWG - 1) / kIntervalsPerWG : P.n_int;
        if (grid == n_wg_k) hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 2, false, MU, true, false, true>), dim3(grid), dim3(threads), 0, st, QC_HOT_ARGS(P), P, dF, dJ, nullptr);
        else hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 2, false, MU, true, false>), dim3(grid), dim3(threads), 0, st, QC_HOT_ARGS(P), P, dF, dJ, nullptr);
        return;
    }
    const int n_wg = JAC ? (P.n_int + kIntervalsPerWG - 1) / kIntervalsPerWG : P.n_int;
    switch (P.store_mode) {
        case 0: hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 0, DIAG, MU, false, false>), dim3(grid), dim3(threads), 0, st, QC_HOT_ARGS(P), P, dF, dJ, nullptr); break;
        case 1: hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 1, DIAG, MU, false, false>), dim3(grid), dim3(threads), 0, st, QC_HOT_ARGS(P), P, dF, dJ, nullptr); break;
        default:
            if (grid == n_wg)   // one interval per workgroup: the loop-free instantiation (also for the stamped diagnostic build)
                hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 2, DIAG, MU, false, false, true>), dim3(grid), dim3(threads), 0, st, QC_HOT_ARGS(P), P, dF, dJ, nullptr);
            else
                hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 2, DIAG, MU, false, false>), dim3(grid), dim3(threads), 0, st, QC_HOT_ARGS(P), P, dF, dJ, nullptr);
            break;
    }
}

// The number of drive images held in registers / LDS is a compile-time constant (2, 4, 6 or 8: the next even
// number >= m, at most 8), so that small systems do not pay the register pressure of the largest.
template <bool JAC, bool DIAG>
static void launch16(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st, int grid, int threads) {
    if (P.m <= 2) launch16m<JAC, DIAG, 2>(P, dZ, dF, dJ, st, grid, threads);
    else if (P.m <= 4) launch16m<JAC, DIAG, 4>(P, dZ, dF, dJ, st, grid, threads);
    else if (P.m <= 6) launch16m<JAC, DIAG, 6>(P, dZ, dF, dJ, st, grid, threads);
    else launch16m<JAC, DIAG, 8>(P, dZ, dF, dJ, st, grid, threads);
}

// One launch for `count` handles (gridDim.y = count).  The caller has checked qc_mfma16_batchable for every handle.
bool qc_mfma16_batchable(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n <= 16 && P.nc <= 8 && P.m <= 32 && P.store_mode == 2 && P.stamps == nullptr &&
           P.dbg_skip == 0 && P.Gx != nullptr;
}

template <bool JAC, int MU>
static void launch16_batch(const QcParams& P0, const QcParams* dPb, int count, const double* dZ, double* dF, double* dJ, hipStream_t st, int grid,
                           int threads) {
    if (P0.nc != 8 || P0.n != 16)   // kets / padded systems: the masked instantiation
        hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 2, false, MU, true, true>), dim3(grid, count), dim3(threads), 0, st, QC_HOT_ARGS(P0), P0, dF, dJ, dPb);
    else
        hipLaunchKernelGGL((qc_mfma16_pade4_kernel<JAC, 2, false, MU, false, true>), dim3(grid, count), dim3(threads), 0, st, QC_HOT_ARGS(P0), P0, dF, dJ, dPb);
}

hipError_t qc_launch_mfma16_F_jac_batch(const QcParams& P0, const QcParams* dPb, int count, const double* dZ, double* dF, double* dJ,
                                        hipStream_t st) {
    const int n_wg = dJ ? (P0.n_int + kIntervalsPerWG - 1) / kIntervalsPerWG : P0.n_int;
    const int grid = n_wg < kMaxGrid ? n_wg : kMaxGrid;
    const int threads = dJ ? kThreads : 64;
#define QC_B(J_, MU_) launch16_batch<J_, MU_>(P0, dPb, count, dZ, dF, dJ, st, grid, threads)
    if (dJ) { if (P0.m <= 2) QC_B(true, 2); else if (P0.m <= 4) QC_B(true, 4); else if (P0.m <= 6) QC_B(true, 6); else QC_B(true, 8); }
    else    { if (P0.m <= 2) QC_B(false, 2); else if (P0.m <= 4) QC_B(false, 4); else if (P0.m <= 6) QC_B(false, 6); else QC_B(false, 8); }
#undef QC_B
    return hipGetLastError();
}

hipError_t qc_launch_mfma_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    if (P.integrator == QC_EXPONENTIAL) return P.n > 16 ? qc_launch_mfma32_exp(P, dZ, dF, dJ, st) : qc_launch_mfma_exp(P, dZ, dF, dJ, st);
    if (qc_mfma16_padeP_supported(P)) return qc_launch_mfma16_padeP(P, dZ, dF, dJ, st);
    if (P.n > 32) return qc_launch_mfma64_F_jac(P, dZ, dF, dJ, st);
    if (P.n > 16) {
        // sparse drive generators: the row-gather kernel (qc_mfma32_ell.hip; QC_ELL_JAC=0: the dense-image kernel, for A/B runs);
        static const bool ell_jac = !(getenv("QC_ELL_JAC") && atoi(getenv("QC_ELL_JAC")) == 0);
        if (P.ell && ell_jac && P.n == 32 && P.nc == 16) return qc_launch_mfma32_ell_F_jac(P, dZ, dF, dJ, st);
        return qc_launch_mfma32_F_jac(P, dZ, dF, dJ, st);
    }
    const int n_wg = dJ ? (P.n_int + kIntervalsPerWG - 1) / kIntervalsPerWG : P.n_int;
    const int grid = n_wg < kMaxGrid ? n_wg : kMaxGrid;
    const bool diag = P.stamps != nullptr || P.dbg_skip != 0;
    if (dJ) { if (diag) launch16<true, true>(P, dZ, dF, dJ, st, grid, kThreads); else launch16<true, false>(P, dZ, dF, dJ, st, grid, kThreads); }
    else    { if (diag) launch16<false, true>(P, dZ, dF, dJ, st, grid, 64); else launch16<false, false>(P, dZ, dF, dJ, st, grid, 64); }
    return hipGetLastError();
}
