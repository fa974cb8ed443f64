// f64-MFMA Hessian-of-Lagrangian kernel for Pade integrators of ANY even order 2p at 2N <= 16 (up to 8 levels, zero-padded;
// K <= 8 state columns; up to 8 drives).  Companion of qc_mfma_padeP.hip; the order-4 case has its own kernel
// (qc_mfma_hess.hip).  Mathematics: qc_lds_kernels.hip ("Hessian of the Lagrangian", general order), with
// M = reshape(mu_t[0:s], 2N, N), M_q = (G^T)^q M, K_q = G^q [S | D], W_k = D (k even), -S (k odd):
//   (h,U1) | (U0,h)  = sum_k (-1)^k k c_k h^(k-1) M_k  |  -sum_k k c_k h^(k-1) M_k
//   (h,h)            = sum_{k>=2} k (k-1) c_k h^(k-2) <M_k, W_k>
//   (a_j,U1) | (U0,a_j) = Horner in G^T of  G_j^T [YB_r | YF_r],  YB_r | YF_r = sum_{k>r} ((-1)^k | 1) c_k h^k M_(k-1-r)   (second half negated)
//   (a_j,h)          = sum_i <A_j^i, Q'_i>,   A_j^i = G_j^T M_i,   Q'_i = sum_{k>i} k c_k h^(k-1) G^(k-1-i) W_k
//   (a_i,a_j)        = S_ij + S_ji,  S_ij = sum_{al+t <= p-2} c_k h^k <A_i^al, C_j^(par(k), t)>,  k = al + t + 2,
//                      C_j^t = G C_j^(t-1) + G_j K_t  (a tile: S-parity half | D-parity half)
// Tile conventions of qc_mfma_kernels.hip: a 16 x 16 register tile holds two 16 x 8 matrices side by side, halves are exchanged
// with DPP row_ror:8, results of left-multiplications stay in the B/D layout, value blocks are stored transposed (identity
// product) so that stores write whole 128-byte lines.
//
// One 512-thread workgroup (8 wavefronts) per interval:
//   all      wave w sums the generator images k = w, w+8 (one L2 round trip); partial sums meet in LDS.            barrier A
//   wave 0   M_q chain (p products) -> LDS, (h,U1) | (U0,h), (h,h), the tiles [YB_r | YF_r] -> LDS
//   wave 1   Krylov chain K_q (p-1 products) -> LDS, the packed tiles [Q'_2t | Q'_2t+1] -> LDS                    barrier H
//   drive j -> wave (j + 2) mod 8:  matrix blocks (2p-1 products), A_j tiles -> LDS, (a_j,h)                          barrier C
//                                   C_j chain (2p-3 products); as each C_j^t appears, its inner products with every A_i^al
//                                   (tiles from LDS) are accumulated per i; wave sums -> S[i][j] in LDS                 barrier E
//   all      (a_i,a_j) = S_ij + S_ji
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kPHThreads = 512;
constexpr int kPHMaxM = 8;

__device__ inline v4d load_GA(const double* __restrict__ Gx, int mat, int lane) {   // image [matrix][pair][lane][2]
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
__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline double dot4(const v4d& a, const v4d& b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }
__device__ inline double wave_sum(double v) {   // fixed order: bit-reproducible
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// LDS map (tiles of 256 doubles): X_0 .. X_p-1 | K_0 .. K_p-1 | Y_0 .. Y_p-1 | Q'_0 .. | A[m][tp] ; then c_k, S.  The 8 partial
// sums of G share the A block (dead after barrier A; the A tiles are written after barrier H)
struct PHLayout {
    int part, X, K, Y, Qp, A, coef, S, total, tp;
};
__host__ __device__ inline PHLayout ph_layout(int p, int m) {
    PHLayout L;
    L.tp = (p + 1) / 2;
    int o = 0;
    L.X = o;    o += p * 256;                       // X_0 .. X_p-1 (X_p never leaves wave 0's registers)
    L.K = o;    o += p * 256;
    L.Y = o;    o += p * 256;
    L.Qp = o;   o += L.tp * 256;
    L.A = o;    L.part = o;
    { const int a = (m > 0 ? m : 1) * L.tp; o += (a > 8 ? a : 8) * 256; }
    L.coef = o; o += 16;
    L.S = o;    o += 64;
    L.total = o;
    return L;
}

__global__ __launch_bounds__(kPHThreads, 4) void qc_mfma16_padeP_hess_kernel(const QcParams P, const double* __restrict__ Z,
                                                                        const double* __restrict__ Mu, double* __restrict__ H) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int p = P.p, m = P.m;
    const PHLayout L = ph_layout(p, m);
    double* __restrict__ PartL = sm + L.part;
    double* __restrict__ XL = sm + L.X;
    double* __restrict__ KL = sm + L.K;
    double* __restrict__ YL = sm + L.Y;
    double* __restrict__ QpL = sm + L.Qp;
    double* __restrict__ AL = sm + L.A;
    double* __restrict__ ChL = sm + L.coef;
    double* __restrict__ SL = sm + L.S;
    const int tp = L.tp;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const int nc = P.nc, nr = P.n;
    const bool ft = P.off_dt >= 0;
    const double* __restrict__ Gx = P.Gx;
    const v4d IdB = identity_B(g, j);
    const v4d zero = {0.0, 0.0, 0.0, 0.0};

    const int b = qc_xcd_remap(blockIdx.x, P.n_int);
    const long long t = P.t_begin + b;
    const double* __restrict__ z0 = Z + t * (long long)P.zdim;
    const double* __restrict__ z1 = z0 + P.zdim;
    const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
    double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
    const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);

    // ---- loads: multipliers (wave 0), states (wave 1), generator images (all), this wave's drive images -------------------
    v4d t0 = zero, t1 = zero, tm = zero;   // waves 0, 1: U_t, U_t+1 tiles [U | U];  wave 0 also [M | M]
    if (w <= 1) {
        const bool cok = jj < nc;
        const int cb = (cok ? jj : 0) * nr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = cok && 4 * r + g < nr;
            const int off = cb + (ok ? 4 * r + g : 0);
            if (w == 0) { const double v = mu[off]; tm[r] = ok ? v : 0.0; }
            const double a0 = z0[P.off_U + off], a1 = z1[P.off_U + off];
            t0[r] = ok ? a0 : 0.0;
            t1[r] = ok ? a1 : 0.0;
        }
    }
    if (w == 7) qc_hess_tail(P, mu, Hb, lane, 64);   // derivative integrators: d2/d(dx_i) dh = -mu_i; alignment padding
    {
        v4d part = w == 0 ? load_GA(Gx, 0, lane) : zero;
        for (int k = w; k < m; k += 8) part += z0[P.off_a + k] * load_GA(Gx, k + 1, lane);
        lds_put(PartL + 256 * w, lane, part);
        if (w == 0) {
            for (int k = 0; k <= p; ++k) if (lane == 0) ChL[k] = P.c[k];
        }
        if (tid < 64) SL[tid] = 0.0;
    }
    const int my_first = (w + 6) & 7;                  // drive j -> wave (j + 2) mod 8
    v4d GjA = zero, GjT = zero;
    if (my_first < m) {
        GjA = load_GA(Gx, my_first + 1, lane);                 // A-layout of G_j
        GjT = load_GA(Gx, (m + 1) + my_first + 1, lane);       // A-layout of G_j^T
    }
    __syncthreads();                                                   // ---- barrier A
    v4d Ga = lds_get(PartL, lane);
#pragma unroll
    for (int ww = 1; ww < 8; ++ww) Ga += lds_get(PartL + 256 * ww, lane);
    const v4d Gb = mm16(Ga, IdB);                                      // A-layout of G^T

    if (w == 0) {
        // ---- M_q chain: X_q = [M_q | M_q] -------------------------------------------------------------------------------
        v4d X = tm;
        lds_put(XL, lane, X);
        v4d UH = zero;                                                 // [(h,U1) | (U0,h)]
        v4d W;                                                         // [S | D]
#pragma unroll
        for (int r = 0; r < 4; ++r) W[r] = left ? t1[r] + t0[r] : t1[r] - t0[r];
        double hk1 = 1.0, hk2 = 1.0, hh = 0.0;                         // h^(k-1), h^(k-2);  (h,h) partial of this lane
        for (int k = 1; k <= p; ++k) {
            X = mm16(Gb, X);
            if (k < p) lds_put(XL + 256 * k, lane, X);
            const double wk = ChL[k] * (double)k * hk1;
            const double sg = left ? ((k & 1) ? -wk : wk) : -wk;
            UH += sg * X;
            // (h,h) = sum_{k>=2} k (k-1) c_k h^(k-2) <M_k, W_k>: left lanes hold S (odd k, negated), right lanes D (even k)
            if (k >= 2) {
                const bool mine = (k & 1) ? left : !left;
                const double w2 = ChL[k] * (double)(k * (k - 1)) * hk2;
                hh += mine ? ((k & 1) ? -w2 : w2) * dot4(X, W) : 0.0;
                hk2 *= h;
            }
            hk1 *= h;
        }
        if (ft) {
            hh = wave_sum(hh);
            if (lane == 0) Hb[P.ho_hh] = hh;
            const v4d UT = mm16(UH, IdB);                              // lane (g, j) reg r = UH[j][4r+g]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;
                if (c < 8) { if (c < nc && j < nr) qc_st8m<2>(Hb + P.ho_hU + c * nr + j, UT[r]); }
                else if (c - 8 < nc && j < nr) qc_st8m<2>(Hb + P.ho_Uh + (c - 8) * nr + j, UT[r]);
            }
        }
        wave_lds_sync();
        // ---- [YB_r | YF_r] = sum_{k>r} ((-1)^k | 1) c_k h^k M_(k-1-r) -------------------------------------------------------
        for (int r = 0; r < p; ++r) {
            v4d acc = zero;
            double hk = 1.0;
            for (int e = 0; e <= r; ++e) hk *= h;                      // h^(r+1)
            for (int k = r + 1; k <= p; ++k) {
                const double ck = ChL[k] * hk;
                acc += (left ? ((k & 1) ? -ck : ck) : ck) * lds_get(XL + 256 * (k - 1 - r), lane);
                hk *= h;
            }
            lds_put(YL + 256 * r, lane, acc);
        }
    } else if (w == 1) {
        // ---- Krylov chain K_q = G^q [S | D], Q' tiles --------------------------------------------------------------------
        v4d K;
#pragma unroll
        for (int r = 0; r < 4; ++r) K[r] = left ? t1[r] + t0[r] : t1[r] - t0[r];
        lds_put(KL, lane, K);
        for (int q = 1; q < p; ++q) {
            K = mm16(Ga, K);
            lds_put(KL + 256 * q, lane, K);
        }
        wave_lds_sync();
        for (int tq = 0; tq < tp; ++tq) {
            const int i = 2 * tq + (left ? 0 : 1);
            v4d acc = zero;
            double hk1 = 1.0;
            for (int e = 0; e < i; ++e) hk1 *= h;                      // h^i = h^(k-1) at k = i+1
            for (int q = 0; q + 1 + 2 * tq <= p; ++q) {
                const int k = q + 1 + i;
                const v4d Kq = lds_get(KL + 256 * q, lane), Ksw = swap8(Kq);
                const double ck = k <= p ? ChL[k <= p ? k : p] * (double)k * hk1 : 0.0;
                const double sg = (k & 1) ? -ck : ck;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double sv = left ? Kq[r] : Ksw[r], dv = left ? Ksw[r] : Kq[r];
                    acc[r] += sg * ((k & 1) ? sv : dv);
                }
                hk1 *= h;
            }
            lds_put(QpL + 256 * tq, lane, acc);
        }
    }
    __syncthreads();                                                   // ---- barrier H: X, Y, K, Q' complete

    // ---- drives, part 1: matrix blocks, A tiles, (a_j,h) ------------------------------------------------------------------
    for (int dj = my_first; dj < m; dj += 8) {
        if (dj != my_first) {
            GjA = load_GA(Gx, dj + 1, lane);
            GjT = load_GA(Gx, (m + 1) + dj + 1, lane);
        }
        {
            v4d T = zero;
            for (int r = p - 1; r >= 0; --r) {
                const v4d R = mm16(GjT, lds_get(YL + 256 * r, lane));  // G_j^T [YB_r | YF_r]
                T = (r == p - 1) ? R : mm16(Gb, T) + R;
            }
            const v4d TT = mm16(T, IdB);                               // lane (g, j) reg r = T[j][4r+g]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;
                if (c < 8) { if (c < nc && j < nr) qc_st8m<2>(Hb + P.ho_aU + (size_t)dj * P.s + c * nr + j, TT[r]); }
                else if (c - 8 < nc && j < nr) qc_st8m<2>(Hb + P.ho_Ua + (size_t)dj * P.s + (c - 8) * nr + j, -TT[r]);
            }
        }
        double ah = 0.0;
        for (int tq = 0; tq < tp; ++tq) {
            const v4d Xa = lds_get(XL + 256 * (2 * tq), lane), Xb = lds_get(XL + 256 * (2 * tq + 1 < p ? 2 * tq + 1 : p - 1), lane);   // (i = p pairs with Q'_p = 0)
            v4d Mt;                                                    // [M_2t | M_2t+1]  (both halves of an X tile are equal)
#pragma unroll
            for (int r = 0; r < 4; ++r) Mt[r] = left ? Xa[r] : Xb[r];
            const v4d At = mm16(GjT, Mt);                              // [A_j^2t | A_j^2t+1]
            lds_put(AL + (size_t)(dj * tp + tq) * 256, lane, At);
            ah += dot4(At, lds_get(QpL + 256 * tq, lane));             // Q'_i = 0 for i >= p
        }
        if (ft) {
            ah = wave_sum(ah);
            if (lane == 0) Hb[P.ho_ah + dj] = ah;
        }
    }
    __syncthreads();                                                   // ---- barrier C: every A tile is in LDS

    // ---- drives, part 2: C_j chain and its inner products with every A_i^al ---------------------------------------------------
    if (p >= 2) {
        for (int dj = my_first; dj < m; dj += 8) {
            if (dj != my_first || m > 8) GjA = load_GA(Gx, dj + 1, lane);
            double acc[kPHMaxM];
#pragma unroll
            for (int i = 0; i < kPHMaxM; ++i) acc[i] = 0.0;
            v4d C = zero;
            for (int tq = 0; tq <= p - 2; ++tq) {
                const v4d GK = mm16(GjA, lds_get(KL + 256 * tq, lane));
                C = tq == 0 ? GK : mm16(Ga, C) + GK;                  // [C^(S,t) | C^(D,t)]  (S taken positive; W_odd = -S below)
                const v4d Csw = swap8(C);
                v4d CvS, CvD;                                          // S- / D-parity value of C at this lane's (row, column & 7)
#pragma unroll
                for (int r = 0; r < 4; ++r) { CvS[r] = left ? C[r] : Csw[r]; CvD[r] = left ? Csw[r] : C[r]; }
                // Terms al = 0 .. p-2-t (k = al + t + 2, weight c_k h^k, the parity of k picks the half of C), two per A tile
                // [A^al0 | A^al1].  With Ae / Ao the even / odd half of the A tile seen from this lane,
                //   sum_lanes w0 Ae.Cv0 + w1 Ao.Cv1 = sum_lanes a . PW,   PW = P1 + swap8(P2),
                //   P1 = left ? w0 Cv0 : w1 Cv1,  P2 = left ? w1 Cv1 : w0 Cv0
                // (a sum over all lanes is invariant under swapping the halves of both factors), so the loop over the
                // drives i is one LDS tile read and four FMAs per tile.  Both halves compute the same sum: weights carry 1/2.
                double hk = h * h;
                for (int e = 0; e < tq; ++e) hk *= h;                  // h^(t+2) at al = 0
                for (int at = 0; 2 * at + tq <= p - 2; ++at) {
                    const int k0 = 2 * at + tq + 2, k1 = k0 + 1;
                    const double w0 = 0.5 * ChL[k0] * hk * ((k0 & 1) ? -1.0 : 1.0);
                    const double w1 = k1 <= p ? 0.5 * ChL[k1 <= p ? k1 : p] * hk * h * ((k1 & 1) ? -1.0 : 1.0) : 0.0;
                    const bool s0 = (k0 & 1) != 0;                     // k0 odd: S-parity half for al0, D-parity for al1
                    v4d P1, P2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v0 = w0 * (s0 ? CvS[r] : CvD[r]), v1 = w1 * (s0 ? CvD[r] : CvS[r]);
                        P1[r] = left ? v0 : v1;
                        P2[r] = left ? v1 : v0;
                    }
                    const v4d PW = P1 + swap8(P2);
#pragma unroll
                    for (int i = 0; i < kPHMaxM; ++i) {
                        if (i < m) acc[i] += dot4(lds_get(AL + (size_t)(i * tp + at) * 256, lane), PW);
                    }
                    hk *= h * h;
                }
            }
#pragma unroll
            for (int i = 0; i < kPHMaxM; ++i) {
                if (i < m) {
                    const double sij = wave_sum(acc[i]);
                    if (lane == 0) SL[i * 8 + dj] = sij;
                }
            }
        }
    }
    __syncthreads();                                                   // ---- barrier E: S complete
    if (tid < 64) {
        const int i = tid >> 3, k = tid & 7;
        if (i <= k && k < m) Hb[P.ho_aa + k * (k + 1) / 2 + i] = SL[i * 8 + k] + SL[k * 8 + i];
    }
}

}  // namespace

bool qc_mfma16_padeP_hess_supported(const QcParams& P) {
    if (!(P.integrator == QC_PADE && P.p >= 1 && P.p <= QC_MAX_P && P.p != 2 && P.n <= 16 && P.nc <= 8 && P.m <= kPHMaxM && P.Gx != nullptr))
        return false;
    return (size_t)ph_layout(P.p, P.m).total * sizeof(double) <= 160 * 1024;
}

hipError_t qc_launch_mfma16_padeP_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    if (P.n_int <= 0) return hipSuccess;
    const size_t lds = (size_t)ph_layout(P.p, P.m).total * sizeof(double);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qc_mfma16_padeP_hess_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(qc_mfma16_padeP_hess_kernel, dim3(P.n_int), dim3(kPHThreads), lds, st, P, dZ, dMu, dH);
    return hipGetLastError();
}
