// Register-resident f64-MFMA kernel for Pade integrators of ANY even order 2p (p = 1 .. 10) at 2N <= 16 (up to 8 levels,
// zero-padded to the 16 x 16 tile; K <= 8 state columns).  The order-4 kernel of qc_mfma_kernels.hip stays the tuned
// special case; this one serves `pade_order` = 2, 6, 8, ... 20 (the reference's bang-bang test solves with order 12),
// which otherwise run on the VALU/LDS kernel (config 3, order 12: 78 us per evaluation there, 19.6 us here; at order 4 this
// kernel takes 11.7 us against 11.4 us of the tuned one).
//
// Mathematics (qc_lds_kernels.hip header; S = U1 + U0, D = U1 - U0, W_k = D for even k, -S for odd k, h = dt):
//     delta  = D + G Q_0            d/dh = G Q_h            d/da_j = sum_{i=0..p-1} G^i G_j Q_i     (Horner in G)
//     Q_i    = sum_{k=i+1..p} c_k h^k G^{k-1-i} W_k         Q_h    = sum_{k=1..p} k c_k h^{k-1} G^{k-1} W_k
//     B^T, F^T = sum_k (-+1)^k c_k h^k (G^T)^k
// Tile conventions as in qc_mfma_kernels.hip: a 16 x 16 tile holds two 16 x 8 matrices side by side ([S | D],
// [Q_2t | Q_2t+1], [d/da_j | d/da_j+1]); halves are exchanged with DPP row_ror:8; results of left-multiplications stay in
// the B/D register layout; outputs are transposed by an identity product so that stores write whole 128-byte lines.
//
// Four wavefronts per interval (256 threads, 27 KB of LDS at order 12):
//   all      wave w sums the generator images k = w, w+4, ... (one L2 round trip for m <= 8); partial sums meet in LDS.
//   wave 0   (G^T)^k chain (p-1 dependent products) -> B^T, -F^T, then the 2N tile copies of I (x) B, -I (x) F
//   wave 1   Krylov chain K_q = G^q [S | D] (p-1 dependent products, tiles parked in LDS), the packed tiles
//            [Q_2t | Q_2t+1] -> LDS, then [delta | d/dh] = [D | 0] + G [Q_0 | Q_h]
//   waves 2, 3, 1   one pair of drives each (round-robin): per tile t from the top, R = G_j [Q_2t | Q_2t+1] for both drives
//            and two Horner steps T <- G T + [G_j Q_i | G_j+1 Q_i];  2 ceil(p/2) + p - 1 products per pair.
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kPThreads = 256;

__device__ inline v4d load_GA(const double* __restrict__ Gx, int mat, int lane) {   // A-layout image [matrix][pair][lane][2]
    const v2d* p = reinterpret_cast<const v2d*>(Gx) + mat * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
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
// LDS operations of one wave execute in order; this only keeps the compiler from moving them across
__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS map (doubles): 4 partial sums of G | c_0 .. c_p | K_0 .. K_p-1 | [Q_0|Q_1], [Q_2|Q_3], ...
__host__ __device__ inline int lds_doubles_padeP(int p) { return 1024 + 16 + 256 * p + 256 * ((p + 1) / 2); }

template <bool JAC>
__global__ __launch_bounds__(kPThreads) void qc_mfma16_padeP_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                   int hot_n_int, int hot_zdim, int hot_off_a, int hot_off_dt, int hot_m, int hot_p, int hot_off_U,
                                                                   const QcParams P, double* __restrict__ F, double* __restrict__ J) {
    // (the leading arguments are preloaded into scalar registers at wave launch -- -amdgpu-kernarg-preload-count, qc_mfma_kernels.hip --:
    //  the first load requests depend on them only; hot_Zt = the handle's first knot)
    QcKernargTouch<sizeof(QcParams) + 96> touch;   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    touch.request();
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int p = hot_p, m = hot_m;
    double* __restrict__ PartL = sm;
    double* __restrict__ ChL = sm + 1024;
    int* FlagL = reinterpret_cast<int*>(sm + 1039);                     // hand-off flag (last slot of the coefficient block)
    double* __restrict__ KL = sm + 1040;
    double* __restrict__ QL = KL + 256 * p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const int nc = P.nc, nr = P.n;
    const bool ft = hot_off_dt >= 0;
    const double* __restrict__ Gx = hot_Gx;
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    const int b = qc_xcd_remap(blockIdx.x, hot_n_int);
    const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
    const double* __restrict__ z1 = z0 + hot_zdim;
    const double h = ft ? z0[hot_off_dt] : opaque_scalar(P.dt_fixed);

    // ---- all waves: partial sums of G = G_0 + sum_k a_k G_k (wave w takes k = w, w+4, ...), two images per round trip ----
    v4d u0 = zero, u1 = zero;
    if (w == 1) {   // state tiles [U | U]: lane (g, j) reg r = U[4r+g][j & 7]; columns >= nc and rows >= nr are zero
        const bool cok = jj < nc;
        const double* p0 = z0 + hot_off_U + (cok ? jj : 0) * nr;
        const double* p1 = z1 + hot_off_U + (cok ? jj : 0) * nr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = cok && 4 * r + g < nr;
            u0[r] = ok ? p0[4 * r + g] : 0.0;
            u1[r] = ok ? p1[4 * r + g] : 0.0;
        }
    }
    {
        v4d part = w == 0 ? load_GA(Gx, 0, lane) : zero;
        for (int k = w; k < m; k += 8) {
            const int k2 = k + 4 < m ? k + 4 : k;
            const v4d ga = load_GA(Gx, k + 1, lane), gb = load_GA(Gx, k2 + 1, lane);
            const double aa = z0[hot_off_a + k], ab = k + 4 < m ? z0[hot_off_a + k2] : 0.0;
            part += aa * ga;
            part += ab * gb;
        }
        lds_put(PartL + 256 * w, lane, part);
        touch.consume();   // the argument block's lines: the scalar wait, behind the first requests
        if (w == 0) {                                                   // Pade coefficients -> LDS (read with per-lane indices later)
            for (int k = 0; k <= p; ++k) if (lane == 0) ChL[k] = P.c[k];
            if (lane == 0) *FlagL = 0;
        }
    }
    double* __restrict__ Jb = JAC ? J + (size_t)b * P.J_stride + P.J_off : nullptr;
    double* __restrict__ Fb = F ? F + (size_t)b * P.F_stride + P.F_off : nullptr;
    if (w == 3) deriv_rows_generic(P, z0, z1, h, Fb, Jb, lane, false);   // derivative-integrator rows (a few loads and stores, before the wave's other stores)
    // images of this wave's first pair of drives, requested before anything waits
    const int n_pairs = (m + 1) / 2;
    const int my_first = w == 2 ? 0 : (w == 3 ? 1 : (w == 1 ? 2 : n_pairs));   // pairs 0, 1, 2 -> waves 2, 3, 1, then round-robin
    v4d Gj = zero, Gj1 = zero;
    if (JAC && my_first < n_pairs) {
        const int k = 2 * my_first;
        Gj = load_GA(Gx, k + 1, lane);
        Gj1 = load_GA(Gx, (k + 1 < m ? k + 1 : k) + 1, lane);
    }
    __syncthreads();                                                   // ---- barrier A: partial sums complete
    const v4d Ga = (lds_get(PartL, lane) + lds_get(PartL + 256, lane)) + (lds_get(PartL + 512, lane) + lds_get(PartL + 768, lane));

    if (w == 0) {
        // ================= wave 0: (G^T)^k chain, B^T, -F^T, tile copies ================================================
        if constexpr (JAC) {
            const v4d Gb = mm16(Ga, IdB);                              // A-layout of G^T
            v4d T = Ga;                                                // (G^T)^1 in B/D layout
            v4d Fm = IdB, Bm = IdB;
            double hk = 1.0;
            for (int k = 1; k <= p; ++k) {
                hk *= h;
                const double ck = ChL[k] * hk;
                Fm += ck * T;
                Bm += ((k & 1) ? -ck : ck) * T;
                if (k < p) T = mm16(Gb, T);
            }
            Fm = -Fm;
            double* pF = Jb + P.jo_F;
            double* pB = Jb + P.jo_B;
            for (int q = 0; q < nc; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (4 * r + g < nr && j < nr) {
                        qc_st8m<2>(pF + q * nr * nr + (4 * r + g) * nr + j, Fm[r]);
                        qc_st8m<2>(pB + q * nr * nr + (4 * r + g) * nr + j, Bm[r]);
                    }
                }
            }
        }
        return;
    }

    v4d Wsw = zero;
    if (w == 1) {
        // ================= wave 1: Krylov tiles, Q tiles =================================================================
        v4d W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double sv = u1[r] + u0[r], dv = u1[r] - u0[r];
            W[r] = left ? sv : dv;                                     // [S | D]
            Wsw[r] = left ? dv : sv;                                   // [D | S]
        }
        v4d K = W;
        lds_put(KL, lane, K);
        for (int q = 1; q < p; ++q) {
            K = mm16(Ga, K);
            lds_put(KL + 256 * q, lane, K);
        }
        wave_lds_sync();
        const int tiles = JAC ? (p + 1) / 2 : 1;                       // the residual-only launch needs Q_0 only
        for (int tq = 0; tq < tiles; ++tq) {
            const int i = 2 * tq + (left ? 0 : 1);                     // this half-tile's Q index
            v4d acc = zero;
            double hk = 1.0;
            for (int e = 0; e <= i; ++e) hk *= h;                      // h^(i+1)
            for (int q = 0; q + 1 + 2 * tq <= p; ++q) {
                const int k = q + 1 + i;
                const v4d Kq = lds_get(KL + 256 * q, lane), Ksw = swap8(Kq);
                const double ck = k <= p ? ChL[k <= p ? k : p] * hk : 0.0;
                const double sg = (k & 1) ? -ck : ck;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double sv = left ? Kq[r] : Ksw[r], dv = left ? Ksw[r] : Kq[r];
                    acc[r] += sg * ((k & 1) ? sv : dv);                // W_k = D (k even), -S (k odd)
                }
                hk *= h;
            }
            lds_put(QL + 256 * tq, lane, acc);
        }
    }
    // ---- hand-off B: the Q tiles are complete.  A flag in LDS instead of a workgroup barrier: wave 0 does not take part (it
    // would otherwise hold its 2N tile copies back until the Krylov chain is done, or hold the drive waves back until its
    // stores are issued).
    if (w == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(FlagL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        while (__hip_atomic_load(FlagL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    if (w == 1) {
        // [Q_0 | Q_h] -> [delta | d/dh]
        v4d QE = zero;
        {
            double hk = left ? h : 1.0;                                // left: c_k h^k, right: k c_k h^(k-1)
            for (int q = 0; q < p; ++q) {
                const int k = q + 1;
                const v4d Kq = lds_get(KL + 256 * q, lane), Ksw = swap8(Kq);
                const double ck = ChL[k] * hk * (left ? 1.0 : (double)k);
                const double sg = (k & 1) ? -ck : ck;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double sv = left ? Kq[r] : Ksw[r], dv = left ? Ksw[r] : Kq[r];
                    QE[r] += sg * ((k & 1) ? sv : dv);
                }
                hk *= h;
            }
        }
        v4d E = mm16(Ga, QE);
#pragma unroll
        for (int r = 0; r < 4; ++r) E[r] = left ? Wsw[r] + E[r] : E[r];
        const v4d ET = mm16(E, IdB);                                   // lane (g, j) reg r = E[j][4r+g]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * r + g;                                   // tile column: < 8 residual column c, >= 8 d/dh column c-8
            if (c < 8) {
                if (Fb && c < nc && j < nr) qc_st8m<2>(Fb + c * nr + j, ET[r]);
            } else if (JAC && ft && c - 8 < nc && j < nr) {
                qc_st8m<2>(Jb + P.jo_h + (c - 8) * nr + j, ET[r]);
            }
        }
    }
    if constexpr (JAC) {
        // ================= drive pairs (waves 2, 3, 1 round-robin) ===========================================================
        const int tmax = (p + 1) / 2 - 1;
        for (int pi = my_first; pi < n_pairs; pi += 3) {
            const int k = 2 * pi;
            if (pi != my_first) {
                Gj = load_GA(Gx, k + 1, lane);
                Gj1 = load_GA(Gx, (k + 1 < m ? k + 1 : k) + 1, lane);
            }
            v4d T = zero;
            bool first = true;
            for (int tq = tmax; tq >= 0; --tq) {
                const v4d Qt = lds_get(QL + 256 * tq, lane);
                v4d a2[2] = {Gj, Gj1}, b2[2] = {Qt, Qt}, R[2];
                mm16_multi<2>(a2, b2, R);                              // [G_j Q_2t | G_j Q_2t+1], [G_j+1 Q_2t | G_j+1 Q_2t+1]
                const v4d R1sw = swap8(R[0]), R2sw = swap8(R[1]);
                if (2 * tq + 1 <= p - 1) {                             // Horner step i = 2t+1:  + [G_j Q_i | G_j+1 Q_i]
                    v4d A1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) A1[r] = left ? R1sw[r] : R[1][r];
                    T = first ? A1 : mm16(Ga, T) + A1;
                    first = false;
                }
                {                                                      // Horner step i = 2t
                    v4d A0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) A0[r] = left ? R[0][r] : R2sw[r];
                    T = first ? A0 : mm16(Ga, T) + A0;
                    first = false;
                }
            }
            const v4d YT = mm16(T, IdB);                               // lane (g, j) reg r = [d/da_k | d/da_k+1][j][4r+g]
            const bool two = k + 1 < m;
            double* pa = Jb + P.jo_a + (size_t)k * P.s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;
                if (c < 8) { if (c < nc && j < nr) qc_st8m<2>(pa + c * nr + j, YT[r]); }
                else if (two && c - 8 < nc && j < nr) qc_st8m<2>(pa + P.s + (c - 8) * nr + j, YT[r]);
            }
        }
    }
}

}  // namespace

bool qc_mfma16_padeP_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p >= 1 && P.p <= QC_MAX_P && P.p != 2 && P.n <= 16 && P.nc <= 8;
}

hipError_t qc_launch_mfma16_padeP(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    if (P.n_int <= 0) return hipSuccess;
    const size_t lds = (size_t)lds_doubles_padeP(P.p) * sizeof(double);
    const double* dZt = dZ + P.t_begin * (long long)P.zdim;
    if (dJ) hipLaunchKernelGGL((qc_mfma16_padeP_kernel<true>), dim3(P.n_int), dim3(kPThreads), lds, st, P.Gx, dZt, P.n_int, P.zdim, P.off_a, P.off_dt, P.m, P.p, P.off_U, P, dF, dJ);
    else hipLaunchKernelGGL((qc_mfma16_padeP_kernel<false>), dim3(P.n_int), dim3(kPThreads), lds, st, P.Gx, dZt, P.n_int, P.zdim, P.off_a, P.off_dt, P.m, P.p, P.off_U, P, dF, dJ);
    return hipGetLastError();
}
