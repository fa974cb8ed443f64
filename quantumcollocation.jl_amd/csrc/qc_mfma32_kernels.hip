// f64-MFMA kernel for the order-4 Pade integrator at 2N = 32 (4 qubits, BASELINE config 5): every
// matrix is 2 x 2 tiles of 16 x 16, every n x N (32 x 16) matrix two full-width tiles, all products
// v_mfma_f64_16x16x4_f64 with register operands (lane maps: qc_mfma_kernels.hip header).
//
// One 512-thread workgroup (8 wavefronts) per PAIR of intervals, one workgroup per CU.  The hardware deals a
// workgroup's waves round-robin over the 4 SIMDs, so with waves 0-3 = compute and waves 4-7 = copy every SIMD
// hosts exactly one MFMA-heavy wave and one store-heavy wave (two 4-wave workgroups per CU put two compute
// waves on one SIMD about half the time: their MFMA chains then ran 1.7x longer and became the tail).
//   phase 1     the 8 waves share the (m+1) x 4 tiles of the generator images (72 KB for m = 8): each tile is
//               requested ONCE per workgroup and parked in LDS; timestep and amplitudes by scalar loads.  Barrier.
//               NOTHING else is requested here: the images alone keep the CU's vector-memory pipeline busy for ~1.2 us.
//   phase 2     wave w assembles tile w&3 of G = G_0 + sum_k a_k G_k for interval w>>2 from the LDS images.  Barrier.
//               The copy waves issue no global load at all; the compute waves request their knot data (and waves 1 / 3
//               the derivative-integrator data) behind barrier 2 -- they have the slack (done at half the launch time).
//   waves 4-7   "copy waves": wave 4 + 2 s + I owns block row I of B^T and F^T of interval s (two tiles):
//               G_B tiles by identity products, (G^2)^T[I][J] = sum_K G_B[K][I] * G_A[J][K]  (24 MFMAs),
//               then the 2N = 32 tile stores x 2 tiles x 2 matrices of the I_N (x) B / -I_N (x) F copies.
//   waves 0-3   "compute waves": wave 2 s + c of interval s:  G D, Q_0 = -h/2 S + h^2/12 G D (16 MFMAs), then
//               drives k = c, c+2, ...:  d/da_k = G_k Q_0 + h^2/12 G (G_k D), transposed for the store (56 MFMAs
//               per drive).  Wave 2 s also produces the residual and d/dh (48 MFMAs).
// Outputs are stored transposed (lane <-> row): each store instruction writes four whole 128-byte lines;
// stores are non-temporal and compile-time (a run-time store-mode switch halves the store issue rate).
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kThreads32 = 512;
constexpr int kMaxGrid32 = 1024;
constexpr int kMU32 = 8;            // drives whose images live in the LDS block (more: fetched from global memory)
constexpr int kDF32 = 4;            // derivative integrators served from registers

// A-layout image of tile `tile` (= 2 I + K) of generator `mat`: [mat][tile][pair][lane][2]
__device__ inline v4d load_GA32(const double* __restrict__ Gx, int mat, int tile, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(Gx) + (mat * 4 + tile) * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline v4d lds_tile(const double* __restrict__ base, int tile, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(base) + tile * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline void lds_put_tile(double* __restrict__ base, int tile, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + tile * 128 + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}

// the value lane `src` holds, as a wave-uniform scalar
__device__ inline double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// A0*B0 + A1*B1 (two 16x16x16 products summed): two accumulator chains of four MFMAs
__device__ inline v4d mm16x2(const v4d& a0, const v4d& b0, const v4d& a1, const v4d& b1) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], z, 0, 0, 0);
    v4d acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], z, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[2], b0[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[3], b0[3], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[2], b1[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[3], b1[3], acc1, 0, 0, 0);
    return acc0 + acc1;
}

// Store a transposed tile: lane (g, j) reg r holds X[rowbase + j][colbase + 4r + g]; column-major, 32 rows per column.
__device__ inline void store_T32(double* __restrict__ p, const v4d& x, int rowbase, int colbase, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8m<2>(p + (colbase + 4 * r + g) * 32 + rowbase + j, x[r]);   // non-temporal, compile-time
}
// Two transposed tiles of the SAME columns, rows 0-15 (x0) and 16-31 (x1), as full 256-byte columns: v_permlane16_swap (gfx950)
// exchanges the odd 16-lane rows of one register with the even rows of the other, after which lanes 0-31 of a register hold
// one whole column and lanes 32-63 another -- two contiguous 256-byte pieces per store instruction instead of four 128-byte
// halves whose partners are written by another instruction at another time (DRAM row locality: config 5's F + dF streams
// 154 MB per launch).
__device__ inline void swap16_f64(double a, double b, double& x, double& y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);     // 16-lane rows: [a.0, b.0, a.2, b.2]
    y = __hiloint2double((int)hi[1], (int)lo[1]);     //               [a.1, b.1, a.3, b.3]
}
struct ColumnPair { v4d e, o; };                      // registers r: columns 4r + (g & 2) and 4r + (g & 2) + 1, rows 16 (g & 1) + j
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
// the masked forms (KET instantiation): the state block has nr <= 32 rows per column and nc <= 16 columns; the operator
// blocks are nr x nr (systems with 9 .. 15 levels are zero-padded to the 2 x 2 tiles)
__device__ inline void store_T32_cols(double* __restrict__ p, const v4d& x, int rowbase, int nc, int nr, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * r + g < nc && rowbase + j < nr) qc_st8m<2>(p + (4 * r + g) * nr + rowbase + j, x[r]);
}
__device__ inline void store_T32_masked(double* __restrict__ p, const v4d& x, int rowbase, int colbase, int nr, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (colbase + 4 * r + g < nr && rowbase + j < nr) qc_st8m<2>(p + (colbase + 4 * r + g) * nr + rowbase + j, x[r]);
}

// KET: K < 16 ket states (column-masked loads and stores); false = unitary, every mask folds away at compile time
// SINGLE: one interval per workgroup (fewer intervals than CUs: T <~ 256) -- the four compute waves share the interval's drives
// (k = w, w + 4, ...), the four copy waves its block rows and copies (halves), so a workgroup is done in about half the time
// ONCE: the grid covers every pair (always, below 2 x kMaxGrid32 intervals): no persistent loop, so nothing is hoisted in front
// of the first load request (with the loop: ~280 scalar instructions and six dependent batches of argument reads, 1.45 us).
template <bool JAC, bool DIAG, bool KET, bool SINGLE = false, bool ONCE = false>
__global__ __launch_bounds__(kThreads32, 2) void qc_mfma32_pade4_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                       int hot_n_int, int hot_zdim, int hot_off_a, int hot_off_dt, int hot_m,
                                                                       const QcParams P, double* __restrict__ F, double* __restrict__ J) {
    // (the leading arguments are preloaded into scalar registers at wave launch -- this file is compiled with
    //  -amdgpu-kernarg-preload-count --: the image and amplitude requests depend on nothing else; hot_Zt = the handle's first knot)
    unsigned long long t_entry = 0;
    if constexpr (DIAG) t_entry = __builtin_amdgcn_s_memrealtime();
    QcKernargTouch<sizeof(QcParams) + 96> touch;   // one batch of scalar-cache misses instead of one per use (qc_internal.h):
    touch.request();                               // requested here, waited for behind the image requests
    __shared__ __attribute__((aligned(16))) double GaL[2 * 4 * 256];                    // G tiles of the two intervals
    __shared__ __attribute__((aligned(16))) double ImgL[(kMU32 + 1) * 4 * 256];        // image tile t of matrix k at (k * 4 + t) * 256
    __shared__ double DerL[2 * 2 * kDF32 * 64];                                          // derivative-integrator data parked until the end
    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = hot_m;
    const int mL = m < kMU32 ? m : kMU32;
    const bool ft = hot_off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gx = hot_Gx;
    const int n_wg = SINGLE ? hot_n_int : (hot_n_int + 1) / 2;

    for (int vb = blockIdx.x; vb < n_wg; vb += gridDim.x) {
        // Opaque per-pass copies of the lane and wave indices: what derives from them (LDS and store offsets, role flags) is
        // computed where it is used instead of being hoisted out of the persistent loop and held -- at the kernel's 256-register
        // budget: spilled -- through the products (qc_mfma32_hess.hip).
        int lane = lane0, w = w0;
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(w));
        const bool copy_role = w >= 4;
        const int slot = (w & 3) >> 1;      // which interval of the pair
        const int sub = w & 1;              // compute: drive parity; copy: block row I
        const int g = lane >> 4, j = lane & 15;
        const int nc = KET ? P.nc : 16;
        const int nr = KET ? P.n : 32;
        const int jc = (!KET || j < nc) ? j : 0;
        const v4d IdB = identity_B(g, j);
        const int b_raw = SINGLE ? qc_xcd_remap(vb, n_wg) : 2 * qc_xcd_remap(vb, n_wg) + slot;
        const bool active = b_raw < hot_n_int;                // an odd interval count leaves the last slot empty;
        const int b = active ? b_raw : hot_n_int - 1;         // its waves still load images and take part in the barriers
        const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
        const double* __restrict__ z1 = z0 + hot_zdim;
        QC_STAMP_DECL;
        QC_STAMP(P, b, lane, 0);

        // ---- phase 1: the generator images, the timestep and the amplitudes ----------------------------------------------
        constexpr int kPer = (kMU32 + 2) / 2;             // image tiles: matrix parity by wave half, tile by w & 3 -> each of
        v4d img[kPer];                                    // the (mL+1) x 4 tiles exactly once; 5 matrices per wave for kMU32 = 8
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int mat = 2 * u + (w >> 2);
            img[u] = load_GA32(Gx, mat <= mL ? mat : mL, w & 3, lane);
        }
        // The timestep and the first kMU32 amplitudes: ONE vector load (lane u < m: a_u, the other lanes: the timestep), read
        // out lane by lane behind barrier 1.  As scalar loads they were a trap twice over: every scalar wait that follows one
        // -- a kernel argument read a little later -- waits for its round trip through L2 as well (shared counter, out-of-order
        // return), and without the persistent loop the compiler put a wait behind EACH of the nine.
        const int ia = lane < mL ? hot_off_a + lane : (ft ? hot_off_dt : 0);
        const double av = z0[ia];
        asm volatile("" ::: "memory");   // no load sinks below this line, no LDS store rises above it
        touch.consume();                 // (the argument block's lines: the scalar wait, behind every request of the phase)
        // (output addresses: they depend on fields of the argument block, i.e. on a scalar wait -- behind the requests as well)
        double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
        double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
#ifdef QC_STAMP_PROLOGUE
        QC_STAMP(P, b, lane, 5);         // every load requested
#if QC_STAMP_PROLOGUE != 2
        if constexpr (DIAG) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        QC_STAMP(P, b, lane, 6);         // this wave's loads are back
#endif
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int mat = 2 * u + (w >> 2);
            if (mat <= mL) lds_put_tile(ImgL, mat * 4 + (w & 3), lane, img[u]);
        }
#ifdef QC_STAMP_PROLOGUE
        QC_STAMP(P, b, lane, 7);         // LDS written, at the barrier
#endif
        __syncthreads();
        QC_STAMP(P, b, lane, 1);
        const double h = ft ? lane_value(av, 63) : opaque_scalar(P.dt_fixed);

        // ---- phase 2: tile w & 3 of G for interval (w >> 2) ... the copy half assembles for slot of (w-4)>>1 --------
        {
            // waves 0-3 assemble the 4 tiles of slot 0's G?  No: wave w assembles tile (w & 3) of the interval whose
            // amplitudes it holds; waves {0,1} and {4,5} hold slot 0, waves {2,3} and {6,7} slot 1.  Tiles of slot s:
            // compute waves 2s, 2s+1 take tiles 0,1; copy waves 4+2s, 5+2s take tiles 2,3.
            const int tile = (copy_role ? 2 : 0) + sub;
            v4d Gt = lds_tile(ImgL, tile, lane);
#pragma unroll
            for (int u = 0; u < kMU32; ++u) {
                if (u < m) Gt += lane_value(av, u) * lds_tile(ImgL, (u + 1) * 4 + tile, lane);
                if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two tiles in flight, not all eight (register pressure)
            }
            for (int k = kMU32; k < m; ++k) Gt += z0[P.off_a + k] * load_GA32(Gx, k + 1, tile, lane);
            lds_put_tile(GaL + slot * 1024, tile, lane, Gt);
        }
#ifdef QC_STAMP_PROLOGUE
        QC_STAMP(P, b, lane, 8);
#endif
#ifdef QC_STAMP_PROLOGUE
        QC_STAMP(P, b, lane, 9);
#endif
        __syncthreads();
        QC_STAMP(P, b, lane, 2);
        const double hc1 = h * c1, hc2 = h * h * c2;
        const double* __restrict__ GaS = GaL + slot * 1024;

        if (copy_role) {
            // ================= copy wave: block row I of B^T and F^T (tiles (I,0), (I,1)), N copies each ========
            bool skip_copy = false;
            if constexpr (DIAG) skip_copy = (P.dbg_skip & 1) != 0;
            if (JAC && active && !skip_copy) {
                const int I = sub;
                // (G^T)[I][K] as A operand = B-layout of G[K][I] = G[K][I] * Id;  (G^T)[K][Jt] as B operand = A-layout of G[Jt][K]
                const v4d GbK0 = mm16(lds_tile(GaS, 0 * 2 + I, lane), IdB);
                const v4d GbK1 = mm16(lds_tile(GaS, 1 * 2 + I, lane), IdB);
                v4d Fm[2], Bm[2];
#pragma unroll
                for (int Jt = 0; Jt < 2; ++Jt) {
                    const v4d G2T = mm16x2(GbK0, lds_tile(GaS, Jt * 2 + 0, lane), GbK1, lds_tile(GaS, Jt * 2 + 1, lane));
                    const v4d GT = lds_tile(GaS, Jt * 2 + I, lane);      // D-layout of (G^T)[I][Jt] = A-layout of G[Jt][I]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double ev = (I == Jt ? IdB[r] : 0.0) + hc2 * G2T[r];
                        Fm[Jt][r] = -(ev + hc1 * GT[r]);
                        Bm[Jt][r] = ev - hc1 * GT[r];
                    }
                }
                // lane (g, j) reg r = B^T[16I+4r+g][16Jt+j] = B[16Jt+j][16I+4r+g]
                double* pF = Jb + P.jo_F;
                double* pB = Jb + P.jo_B;
                const int ncop = P.copies;   // N copies of each block; 1 when the host path asks for the compact form
                ColumnPair Fc, Bc;
                if constexpr (!KET) {
                    Fc = merge_rows32(Fm[0], Fm[1]);
                    Bc = merge_rows32(Bm[0], Bm[1]);
                }
                for (int q = SINGLE ? slot : 0; q < ncop; q += SINGLE ? 2 : 1) {   // SINGLE: waves 4, 5 even copies, 6, 7 odd
                    if constexpr (KET) {
                        const size_t o = (size_t)q * nr * nr;
                        store_T32_masked(pF + o, Fm[0], 0, 16 * I, nr, g, j);
                        store_T32_masked(pF + o, Fm[1], 16, 16 * I, nr, g, j);
                        store_T32_masked(pB + o, Bm[0], 0, 16 * I, nr, g, j);
                        store_T32_masked(pB + o, Bm[1], 16, 16 * I, nr, g, j);
                        continue;
                    }
                    store_T32_columns(pF + q * 1024, Fc, 16 * I, g, j);
                    store_T32_columns(pB + q * 1024, Bc, 16 * I, g, j);
                }
            }
        } else if (active) {
            // ===================== compute waves ===========================================================
            // The knots' state tiles and the derivative-integrator data are requested HERE, behind barrier 2, by the waves that
            // use them.  Requested with the images they delayed barrier 1 by 2 us (the CU's vector-memory pipeline needs ~1.2 us
            // for the 80 requests of 1 KB that bring the 72 KB of images; a strided state request costs it ~200 cycles, sixteen of
            // them per wave 1.6 us), requested between the barriers they delayed barrier 2 by as much -- and either way the copy
            // waves' first store, which is what the launch time follows.  The compute waves have the slack: they are done after
            // 15 us of the launch's 36 (profiles/stamps_jac32.py, stamps_jac32_prologue.py).
            const int csub = SINGLE ? (w & 3) : sub;           // this wave's first drive;  SINGLE: four compute waves per interval
            constexpr int cstep = SINGLE ? 4 : 2;
            v4d U0[2], U1[2];
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                if constexpr (!KET) {
                    U0[I] = load_col16_T(z0 + P.off_U + jc * 32 + 16 * I, g);   // 2 x 16 bytes per lane, transposed in registers
                    U1[I] = load_col16_T(z1 + P.off_U + jc * 32 + 16 * I, g);
                } else {   // columns >= nc re-read column 0 (never stored); rows >= nr are the zero padding
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * I + 4 * r + g;
                        U0[I][r] = row < nr ? z0[P.off_U + jc * nr + row] : 0.0;
                        U1[I][r] = row < nr ? z1[P.off_U + jc * nr + row] : 0.0;
                    }
                }
            }
            const bool dfast = P.n_deriv <= kDF32;
            const bool deriv_wave = JAC && csub == 1;          // waves 1 and 3 (SINGLE: wave 1) also write the derivative rows
            double* __restrict__ derl = DerL + slot * (2 * kDF32 * 64);
            if (deriv_wave) {   // requested before this wave's first store, parked in LDS, used after the drives
                double dvx[kDF32], dva[kDF32], dvb[kDF32];
#pragma unroll
                for (int d = 0; d < kDF32; ++d) {   // unused slots have zero offsets/dims: the loads stay in bounds
                    const int i = lane < P.ddim_i[d] ? lane : 0;
                    dvx[d] = z0[P.dx_off[d] + i];
                    dva[d] = z0[P.x_off[d] + i];
                    dvb[d] = z1[P.x_off[d] + i];
                }
#pragma unroll
                for (int d = 0; d < kDF32; ++d) {
                    derl[(2 * d) * 64 + lane] = dvx[d];
                    derl[(2 * d + 1) * 64 + lane] = dvb[d] - dva[d];
                }
            }
            v4d Ga[4];
#pragma unroll
            for (int tI = 0; tI < 4; ++tI) Ga[tI] = lds_tile(GaS, tI, lane);
            const v4d S[2] = {U1[0] + U0[0], U1[1] + U0[1]};
            const v4d D[2] = {U1[0] - U0[0], U1[1] - U0[1]};
            v4d GD[2];
#pragma unroll
            for (int I = 0; I < 2; ++I) GD[I] = mm16x2(Ga[2 * I], D[0], Ga[2 * I + 1], D[1]);
            if (csub == 0) {
                v4d GS[2], G2D[2];
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    GS[I] = mm16x2(Ga[2 * I], S[0], Ga[2 * I + 1], S[1]);
                    G2D[I] = mm16x2(Ga[2 * I], GD[0], Ga[2 * I + 1], GD[1]);
                }
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    const v4d dl = D[I] - hc1 * GS[I] + hc2 * G2D[I];
                    if (Fb) { if constexpr (KET) store_T32_cols(Fb, mm16(dl, IdB), 16 * I, nc, nr, g, j); else store_T32(Fb, mm16(dl, IdB), 16 * I, 0, g, j); }
                    if (JAC && ft) {
                        const v4d dh = (-c1) * GS[I] + (2.0 * c2 * h) * G2D[I];
                        if constexpr (KET) store_T32_cols(Jb + P.jo_h, mm16(dh, IdB), 16 * I, nc, nr, g, j); else store_T32(Jb + P.jo_h, mm16(dh, IdB), 16 * I, 0, g, j);
                    }
                }
                if (!JAC) deriv_rows_generic(P, z0, z1, h, Fb, nullptr, lane, false);
            }
            bool skip_drives = false;
            if constexpr (DIAG) skip_drives = (P.dbg_skip & 2) != 0;
            if (JAC && !skip_drives) {
                v4d Q0[2];                          // Q_1 = h^2 c2 D is applied as a scale on G (G_k D) below
#pragma unroll
                for (int I = 0; I < 2; ++I) Q0[I] = (-hc1) * S[I] + hc2 * GD[I];
                for (int k = csub; k < m; k += cstep) {
                    v4d Gk[4];
#pragma unroll
                    for (int tI = 0; tI < 4; ++tI)   // LDS image block for the first kMU32 drives, global memory beyond
                        Gk[tI] = k < kMU32 ? lds_tile(ImgL, (k + 1) * 4 + tI, lane) : load_GA32(Gx, k + 1, tI, lane);
                    v4d R1[2];                      // G_k D
#pragma unroll
                    for (int I = 0; I < 2; ++I) R1[I] = mm16x2(Gk[2 * I], D[0], Gk[2 * I + 1], D[1]);
                    double* pa = Jb + P.jo_a + (size_t)k * (KET ? P.s : 512);
                    v4d YT[2];
#pragma unroll
                    for (int I = 0; I < 2; ++I) {
                        const v4d R0 = mm16x2(Gk[2 * I], Q0[0], Gk[2 * I + 1], Q0[1]);          // G_k Q_0
                        const v4d Y = R0 + hc2 * mm16x2(Ga[2 * I], R1[0], Ga[2 * I + 1], R1[1]);   // + h^2 c2 G (G_k D)
                        YT[I] = mm16(Y, IdB);
                        if constexpr (KET) store_T32_cols(pa, YT[I], 16 * I, nc, nr, g, j);
                    }
                    if constexpr (!KET) store_T32_columns(pa, merge_rows32(YT[0], YT[1]), 0, g, j);
                }
            }
            if (deriv_wave) {   // derivative integrator rows
                int jo = P.jo_d;
                bool all_fast = dfast;
#pragma unroll
                for (int d = 0; d < kDF32; ++d) {
                    if (d < P.n_deriv) {
                        const int dim = P.ddim_i[d], r0 = P.drow[d];
                        if (dfast && dim <= 64) {
                            if (lane < dim) {
                                const double dx = derl[(2 * d) * 64 + lane], df = derl[(2 * d + 1) * 64 + lane];
                                if (Fb) Fb[r0 + lane] = df - h * dx;
                                Jb[jo + lane] = -1.0;
                                Jb[jo + dim + lane] = 1.0;
                                Jb[jo + 2 * dim + lane] = -h;
                                if (ft) Jb[jo + 3 * dim + lane] = -dx;
                            }
                        } else {
                            all_fast = false;
                        }
                        jo += (ft ? 4 : 3) * dim;
                    }
                }
                if (!all_fast) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, dfast);
            }
        }
        if constexpr (DIAG) {
            QC_STAMP(P, b, lane, 3);                       // everything issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QC_STAMP(P, b, lane, 4);                       // everything acknowledged
            if (P.stamps != nullptr && lane == 0 && active) {   // 4 waves x 4 slots per interval: [compute 0, compute 1, copy 0, copy 1]
                const int wi = (copy_role ? 2 : 0) + sub;
#pragma unroll
                for (int k_ = 0; k_ < 4; ++k_) P.stamps[(size_t)b * 16 + wi * 4 + k_] = qc_ts_[k_ == 3 ? 4 : (k_ == 0 ? 1 : k_ + 1)];
#ifdef QC_STAMP_PROLOGUE
#if QC_STAMP_PROLOGUE == 2
                P.stamps[(size_t)b * 16 + wi * 4 + 0] = qc_ts_[1];     // behind barrier 1
                P.stamps[(size_t)b * 16 + wi * 4 + 1] = qc_ts_[8];     // G tile assembled and written
                P.stamps[(size_t)b * 16 + wi * 4 + 2] = qc_ts_[9];     // knot loads requested, at barrier 2
                P.stamps[(size_t)b * 16 + wi * 4 + 3] = qc_ts_[2];     // behind barrier 2
#else
                P.stamps[(size_t)b * 16 + wi * 4 + 0] = qc_ts_[0];
                P.stamps[(size_t)b * 16 + wi * 4 + 1] = qc_ts_[5];
                P.stamps[(size_t)b * 16 + wi * 4 + 2] = qc_ts_[6];
                P.stamps[(size_t)b * 16 + wi * 4 + 3] = qc_ts_[7];
                if (wi == 1) P.stamps[(size_t)b * 16 + wi * 4 + 0] = qc_ts_[1];
#endif
                if (false)
#endif
                if (wi == 1) {   // compute wave 1 records the prologue instead: kernel entry, addresses known (arguments read), loads landed
                    P.stamps[(size_t)b * 16 + 4] = t_entry;
                    P.stamps[(size_t)b * 16 + 5] = qc_ts_[0];
                    P.stamps[(size_t)b * 16 + 6] = qc_ts_[1];
                }
            }
        }
        if constexpr (ONCE) break;
        __syncthreads();   // the LDS blocks are rewritten by the next pair of a persistent grid
    }
}

}  // namespace

size_t qc_mfma32_gx_doubles(const QcParams& P) { return (size_t)2 * (P.m + 1) * 4 * 256; }

// A-layout images [matrix][tile = 2I+K][pair][lane][2]:  lane (g, i) reg kk = X[16I + i][16K + 4kk + g]  (X column-major
// 32 x 32), followed by the B-layout images [matrix][tile = 2K+J][pair][lane][2]: lane (g, j) reg kk = X[16K + 4kk + g][16J + j]
// (= A-layout images of the transposes; used by the Hessian kernel).
void qc_mfma32_pack_G(const QcParams& P, const double* G, double* Gx) {
    const int n = P.n, M = P.m + 1;      // n < 32: zero-padded to the 2 x 2 tiles
    double* GxB = Gx + (size_t)M * 1024;
    for (int mat = 0; mat < M; ++mat) {
        const double* A = G + (size_t)mat * n * n;
        for (int tile = 0; tile < 4; ++tile) {
            const int K = tile >> 1, J = tile & 1;
            for (int pr = 0; pr < 2; ++pr)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 2; ++e) {
                        const int g = l >> 4, j = l & 15, kk = 2 * pr + e;
                        const int rr = 16 * K + 4 * kk + g, cc = 16 * J + j;
                        GxB[(((size_t)mat * 4 + tile) * 2 + pr) * 128 + l * 2 + e] = (rr < n && cc < n) ? A[(size_t)cc * n + rr] : 0.0;
                    }
        }
    }
    for (int mat = 0; mat < M; ++mat) {
        const double* A = G + (size_t)mat * n * n;
        for (int tile = 0; tile < 4; ++tile) {
            const int I = tile >> 1, K = tile & 1;
            for (int pr = 0; pr < 2; ++pr)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 2; ++e) {
                        const int g = l >> 4, i = l & 15, kk = 2 * pr + e;
                        const int rr = 16 * I + i, cc = 16 * K + 4 * kk + g;
                        Gx[(((size_t)mat * 4 + tile) * 2 + pr) * 128 + l * 2 + e] = (rr < n && cc < n) ? A[(size_t)cc * n + rr] : 0.0;
                    }
        }
    }
}

hipError_t qc_launch_mfma32_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    const int n_wg = (P.n_int + 1) / 2;
    const int grid = n_wg < kMaxGrid32 ? n_wg : kMaxGrid32;
    const bool once = n_wg <= kMaxGrid32;
    const bool diag = P.stamps != nullptr || P.dbg_skip != 0;
    const bool ket = P.nc != 16 || P.n != 32;
#define QC_LAUNCH32(JAC_, DIAG_, KET_, SINGLE_, ONCE_, GRID_) \
    hipLaunchKernelGGL((qc_mfma32_pade4_kernel<JAC_, DIAG_, KET_, SINGLE_, ONCE_>), dim3(GRID_), dim3(kThreads32), 0, st, P.Gx,                 \
                       dZ + P.t_begin * (long long)P.zdim, P.n_int, P.zdim, P.off_a, P.off_dt, P.m, P, dF, dJ)
    if (dJ && !diag && P.n_int <= 256) {   // fewer intervals than CUs: one interval per workgroup, all eight waves on it
        if (ket) QC_LAUNCH32(true, false, true, true, true, P.n_int);
        else QC_LAUNCH32(true, false, false, true, true, P.n_int);
        return hipGetLastError();
    }
    if (ket) {
        if (dJ && once) QC_LAUNCH32(true, false, true, false, true, grid);
        else if (dJ) QC_LAUNCH32(true, false, true, false, false, grid);
        else if (once) QC_LAUNCH32(false, false, true, false, true, grid);
        else QC_LAUNCH32(false, false, true, false, false, grid);
        return hipGetLastError();
    }
    if (dJ && diag && once) QC_LAUNCH32(true, true, false, false, true, grid);
    else if (dJ && diag) QC_LAUNCH32(true, true, false, false, false, grid);
    else if (dJ && once) QC_LAUNCH32(true, false, false, false, true, grid);
    else if (dJ) QC_LAUNCH32(true, false, false, false, false, grid);
    else if (once) QC_LAUNCH32(false, false, false, false, true, grid);
    else QC_LAUNCH32(false, false, false, false, false, grid);
#undef QC_LAUNCH32
    return hipGetLastError();
}
