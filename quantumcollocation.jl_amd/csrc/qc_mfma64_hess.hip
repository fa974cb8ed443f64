// f64-MFMA Hessian-of-Lagrangian kernel, order-4 Pade, 32 < 2N <= 64 (17 .. 32 levels; 5 qubits: N = 32), up to 14 drives.
// Same construction as qc_mfma64_kernels.hip: matrices column-major in LDS (66-double columns), every tile product a chain of
// 16 v_mfma_f64_16x16x4_f64 with the operands SWAPPED so that the accumulator holds the transposed tile (lanes along rows:
// each 8-byte-per-lane store writes whole 128-byte lines of the column-major value blocks).
//
// The blocks (h = dt, c1 = 1/2, c2 = 1/12; M = reshape(mu_t[0:s], 2N, N), M1 = G^T M, M2 = G^T M1,
// N_k = G_k^T M, V_k = G_k D, X_k = G_k^T M1 + G^T N_k, R_h = -c1 S + 2 c2 h G D):
//   (U_t, a_k)   = -c1 h N_k - c2 h^2 X_k        (a_k, U_t+1) = -c1 h N_k + c2 h^2 X_k
//   (U_t, h)     = -(c1 M1 + 2 c2 h M2)           (h, U_t+1)   = -c1 M1 + 2 c2 h M2
//   (a_i, a_k)   = c2 h^2 (<N_i, V_k> + <N_k, V_i>)      (a_k, h) = <N_k, R_h> + 2 c2 h <M1, V_k>      (h, h) = 2 c2 <M1, G D>
//   (dx_i, h)    = -mu_i   (derivative integrators)
//
// One 1024-thread workgroup (16 wavefronts, 118 KB of LDS) per CU walks the intervals b = blockIdx, blockIdx + grid, ...
//   waves 0-7  "matrix waves" (I, J'): tile (I, J') of M1, M2, and per drive of N_k, G_k^T M1 (sharing the fragment of
//              G_k^T, read from a transposed global image), then G^T N_k (N_k exchanged through a double-buffered LDS block,
//              one workgroup barrier per drive) and the two matrix blocks of the drive.
//   waves 8-15 "aux waves" (I, J'): tile (I, J') of G D -> R_h, and per drive of V_k.
// The scalar blocks are inner products between {N_0 .. N_m-1, M1} and {V_0 .. V_m-1, R_h, G D}: a 16 x 16 Gram matrix over
// 2048 elements.  Every tile of those matrices is also written to a per-workgroup global scratch laid out
// [element / 4][slot (16)][element % 4]; at the end of the interval the Gram matrix is 512 MFMAs (32 per wave, operands
// fetched as whole 512-byte rows of the scratch), summed over the waves through LDS in a fixed order (bit-reproducible).
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kHT64 = 1024;
constexpr int kHMax64 = 14;               // drives: m + 2 Gram slots <= 16
constexpr int kHGrid64 = 256;             // one workgroup per CU
constexpr int kLd = 66;
constexpr int kMat = 64 * kLd;
constexpr int kHalf = 32 * kLd;
constexpr int oG = 0;                     // G
constexpr int oM = kMat;                  // [M | M1]
constexpr int oD = 2 * kMat;              // D
constexpr int oN = 2 * kMat + kHalf;      // N_k, double-buffered; first S (phase 0-1 only); last the Gram partial sums
constexpr int kLdsDoublesH64 = 3 * kMat + kHalf;
constexpr int kScratchSet = 512 * 64;     // doubles of one slot set [512 chunks][16 slots][4]

__device__ __forceinline__ v4d mfma4(double a, double b, const v4d& c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// acc0 += sum_q mfma(pa0[4 q], fb[q]),  acc1 += sum_q mfma(pa1[4 q], fb[q]):  two col fragments (LDS) against one row fragment
__device__ __forceinline__ void mm2(const double* __restrict__ pa0, const double* __restrict__ pa1, const double (&fb)[16], v4d& acc0, v4d& acc1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        acc0 = mfma4(pa0[q * 4], fb[q], acc0);
        acc1 = mfma4(pa1[q * 4], fb[q], acc1);
    }
}
// one tile product, two accumulator chains (even / odd steps)
__device__ __forceinline__ v4d mm1(const double* __restrict__ pa, const double (&fb)[16], const v4d& init) {
    v4d acc0 = init, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
        acc0 = mfma4(pa[q * 4], fb[q], acc0);
        acc1 = mfma4(pa[(q + 1) * 4], fb[q + 1], acc1);
    }
    return acc0 + acc1;
}
// the same with the b-operand fragment read from LDS as well (pb[q * sb])
__device__ __forceinline__ v4d mm1p(const double* __restrict__ pa, const double* __restrict__ pb, int sb, const v4d& init) {
    v4d acc0 = init, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
        acc0 = mfma4(pa[q * 4], pb[q * sb], acc0);
        acc1 = mfma4(pa[(q + 1) * 4], pb[(q + 1) * sb], acc1);
    }
    return acc0 + acc1;
}
// row fragment I of a zero-padded 64 x 64 column-major global image
__device__ __forceinline__ void load_fragment(const double* __restrict__ img, int I, int g, int j, double (&f)[16]) {
    const double* p = img + g * 64 + 16 * I + j;
#pragma unroll
    for (int q = 0; q < 16; ++q) f[q] = p[q * 256];
}
__device__ __forceinline__ void settle_fragment(double (&f)[16]) {   // see qc_mfma64_kernels.hip
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(f[q]));
}
// transposed-layout tile (I, Jp) of a 64 x 32 matrix (lane (g, j) reg r = X[16 I + j][16 Jp + 4 r + g]) -> slot `slot` of a scratch set
__device__ __forceinline__ void scratch_put(double* __restrict__ S, int slot, int I, int Jp, int g, int j, const v4d& x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(size_t)(((16 * Jp + 4 * r + g) * 16 + 4 * I + (j >> 2)) * 64 + slot * 4 + (j & 3))] = x[r];
}
// the same tile -> a column-major value block (n rows per column), masked to the n x nc corner
__device__ __forceinline__ void block_put(double* __restrict__ p, int n, int nc, int I, int Jp, int g, int j, const v4d& x) {
    const int row = 16 * I + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 16 * Jp + 4 * r + g;
        if (row < n && col < nc) qc_st8m<2>(p + col * n + row, x[r]);
    }
}

__global__ __launch_bounds__(kHT64, 1) void qc_mfma64_pade4_hess_kernel(const QcParams P, const double* __restrict__ Z,
                                                                       const double* __restrict__ Mu, double* __restrict__ H,
                                                                       double* __restrict__ scratch) {
    qc_kernarg_touch<sizeof(QcParams) + 64>();   // one batch of scalar-cache misses instead of one per use (qc_internal.h)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* __restrict__ Gs = sm + oG;
    double* __restrict__ Ms = sm + oM;
    double* __restrict__ Ds = sm + oD;
    double* __restrict__ Ns = sm + oN;
    const int tid0 = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int n = P.n, nc = P.nc, m = P.m;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ Gimg = P.Gx;                              // (m+1) images of G_k
    const double* __restrict__ GTimg = P.Gx + (size_t)(m + 1) * 4096;    // (m+1) images of G_k^T
    double* __restrict__ SA = scratch + (size_t)blockIdx.x * 2 * kScratchSet;   // slots N_0 .. N_m-1, M1
    double* __restrict__ SB = SA + kScratchSet;                                 // slots V_0 .. V_m-1, R_h, G D
    const bool matrix = w < 8;
    const int I = w & 3, Jp = (w >> 2) & 1;

#pragma unroll 1
    for (int b = blockIdx.x; b < P.n_int; b += gridDim.x) {
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
        const double h = ft ? z0[P.off_dt] : opaque_scalar(P.dt_fixed);
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;
        // The thread coordinates are re-derived from an opaque copy of the thread id in every interval: everything computed
        // from them (a few dozen LDS / scratch / image / output offsets) is otherwise hoisted out of the interval loop and spilled.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, g = lane >> 4, j = lane & 15;

        unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // diagnostic time stamps (QC_STAMPS=1), waves 0 and 8
        const bool stamp = P.stamps != nullptr && (w == 0 || w == 8);
#define QC_TS(k) do { if (stamp) ts[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
        QC_TS(0);
        // ---------------- phase 0: G, M, S, D into LDS -------------------------------------------------------------
        {
            double acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = Gimg[tid + 1024 * e];
            // (requesting 4 or 8 images per pass before the first FMA measured no faster: 7.1 / 7.4 / 9.1 us for 1 / 4 / 8 --
            // the 360 KB of images per interval are bound by throughput, not by the round trip)
            for (int k = 0; k < m; ++k) {
                const double ak = z0[P.off_a + k];
                const double* __restrict__ Gk = Gimg + (size_t)(k + 1) * 4096;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += ak * Gk[tid + 1024 * e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = tid + 1024 * e;
                Gs[(idx >> 6) * kLd + (idx & 63)] = acc[e];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = tid + 1024 * e, r = idx & 63, c = idx >> 6;
                const bool ok = r < n && c < nc;
                const int off = ok ? c * n + r : 0;
                const double u0 = ok ? z0[P.off_U + off] : 0.0, u1 = ok ? z1[P.off_U + off] : 0.0;
                Ms[c * kLd + r] = ok ? mu[off] : 0.0;
                Ns[c * kLd + r] = u1 + u0;                 // S (until the first drive overwrites the block)
                Ds[c * kLd + r] = u1 - u0;
            }
        }
        if (w == 15) qc_hess_tail(P, mu, Hb, lane, 64);    // derivative integrators: d2/d(dx_i) dh = -mu_i; alignment padding
        __syncthreads();
        QC_TS(1);

        const double* colM = Ms + (16 * Jp + j) * kLd + g;            // col fragment Jp of M; of M1 at + 32 kLd
        const double* colD = Ds + (16 * Jp + j) * kLd + g;            // col fragment Jp of D
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        // b-operand of the products with G: matrix waves row fragment I of G^T (= col fragment I of G, step 4), aux waves row
        // fragment I of G (step 4 kLd); read from LDS as the chains advance (16 more live registers would spill)
        const double* pG = matrix ? Gs + (16 * I + j) * kLd + g : Gs + g * kLd + 16 * I + j;
        const int sG = matrix ? 4 : 4 * kLd;
        double fk[16];     // row fragment I of the current drive's G_k^T (matrix) / G_k (aux)
        if (m > 0) load_fragment((matrix ? GTimg : Gimg) + 4096, I, g, j, fk);

        // ---------------- phase 1: M1 (matrix waves), G D and R_h (aux waves) -----------------------------------------
        v4d M1t = zero;
        if (matrix) {
            M1t = mm1p(colM, pG, sG, zero);                                  // (G^T M)^T tile
            double* p = Ms + (32 + 16 * Jp + g) * kLd + 16 * I + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[4 * r * kLd] = M1t[r];
        } else {
            const v4d GD = mm1p(colD, pG, sG, zero);                         // (G D)^T tile
            if (ft) {
                const double* sT = Ns + (16 * Jp + g) * kLd + 16 * I + j;
                v4d Rh;
#pragma unroll
                for (int r = 0; r < 4; ++r) Rh[r] = -c1 * sT[4 * r * kLd] + c2h2 * GD[r];
                if (m > 0) settle_fragment(fk);
                scratch_put(SB, m, I, Jp, g, j, Rh);
                scratch_put(SB, m + 1, I, Jp, g, j, GD);
            }
        }
        __syncthreads();                                                // M1 complete; S no longer needed
        QC_TS(2);
        if (matrix) {
            if (ft) {
                const v4d M2 = mm1p(colM + 32 * kLd, pG, sG, zero);          // (G^T M1)^T tile
                if (m > 0) settle_fragment(fk);
                scratch_put(SA, m, I, Jp, g, j, M1t);
                block_put(Hb + P.ho_Uh, n, nc, I, Jp, g, j, -(c1 * M1t + c2h2 * M2));
                block_put(Hb + P.ho_hU, n, nc, I, Jp, g, j, (-c1) * M1t + c2h2 * M2);
            }
        }

        QC_TS(3);
        // ---------------- drives -----------------------------------------------------------------------------------
        for (int k = 0; k < m; ++k) {
            if (k == 1) QC_TS(4);
            if (k == 2) QC_TS(5);
            double* __restrict__ Nk = Ns + (k & 1) * kHalf;
            if (matrix) {
                v4d Nt = zero, Pt = zero;
                mm2(colM, colM + 32 * kLd, fk, Nt, Pt);                 // (G_k^T M)^T, (G_k^T M1)^T
                double* p = Nk + (16 * Jp + g) * kLd + 16 * I + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) p[4 * r * kLd] = Nt[r];
                if (k + 1 < m) load_fragment(GTimg + (size_t)(k + 2) * 4096, I, g, j, fk);
                __syncthreads();                                        // N_k complete (the other buffer is free again)
                const v4d Xt = mm1p(Nk + (16 * Jp + j) * kLd + g, pG, sG, Pt);   // + (G^T N_k)^T
                if (k + 1 < m) settle_fragment(fk);
                scratch_put(SA, k, I, Jp, g, j, Nt);
                block_put(Hb + P.ho_Ua + (size_t)k * P.s, n, nc, I, Jp, g, j, (-hc1) * Nt - hc2 * Xt);
                block_put(Hb + P.ho_aU + (size_t)k * P.s, n, nc, I, Jp, g, j, (-hc1) * Nt + hc2 * Xt);
            } else {
                const v4d Vt = mm1(colD, fk, zero);                     // (G_k D)^T
                if (k + 1 < m) load_fragment(Gimg + (size_t)(k + 2) * 4096, I, g, j, fk);
                __syncthreads();
                if (k + 1 < m) settle_fragment(fk);
                scratch_put(SB, k, I, Jp, g, j, Vt);
            }
        }

        QC_TS(6);
        // ---------------- scalar blocks: Gram matrix of the scratch slots ----------------------------------------------
        // scratch written by this workgroup, read by this workgroup (one CU, one L1): a workgroup-scope release is enough.
        // (__threadfence() is an agent-scope release: it waited 40-60 us for the interval's non-temporal block stores.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
            const int perm = j * 4 + g;             // lane (i, g) <-> slot i, element g of the chunk
            const double* pa = SA + (size_t)(32 * w) * 64 + perm;
            const double* pb = SB + (size_t)(32 * w) * 64 + perm;
            v4d c0 = zero, c1v = zero;
#pragma unroll 4
            for (int q = 0; q < 32; q += 2) {
                c0 = mfma4(pa[q * 64], pb[q * 64], c0);
                c1v = mfma4(pa[(q + 1) * 64], pb[(q + 1) * 64], c1v);
            }
            const v4d c = c0 + c1v;                                     // lane (g, j) reg r = sum over this wave's elements of A_{4r+g} B_j
#pragma unroll
            for (int r = 0; r < 4; ++r) Ns[w * 256 + (4 * r + g) * 16 + j] = c[r];
        }
        __syncthreads();
        if (tid < 256) {
            double sum = 0.0;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) sum += Ns[ww * 256 + tid];
            Ms[tid] = sum;                                              // C[i][j] at i * 16 + j  (M is no longer needed)
        }
        __syncthreads();
        if (tid < 256) {
            const int i = tid >> 4, k = tid & 15;
            if (i <= k && k < m) Hb[P.ho_aa + k * (k + 1) / 2 + i] = hc2 * (Ms[i * 16 + k] + Ms[k * 16 + i]);
            if (ft && i == 0 && k < m) Hb[P.ho_ah + k] = Ms[k * 16 + m] + c2h2 * Ms[m * 16 + k];
            if (ft && tid == 255) Hb[P.ho_hh] = 2.0 * c2 * Ms[m * 16 + m + 1];
        }
        __syncthreads();                                                // LDS and scratch are rewritten by the next interval
        QC_TS(7);
        if (stamp && lane == 0) {
#pragma unroll
            for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + (w == 0 ? 0 : 8) + k_] = ts[k_];
        }
#undef QC_TS
    }
}

}  // namespace

bool qc_mfma64_hess_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n > 32 && P.n <= 64 && P.nc <= 32 && P.m <= kHMax64 && P.Gx != nullptr;
}

size_t qc_mfma64_hess_scratch_doubles(const QcParams& P) { return (size_t)kHGrid64 * 2 * kScratchSet; }

hipError_t qc_launch_mfma64_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    constexpr size_t lds = (size_t)kLdsDoublesH64 * sizeof(double);
    if (P.hs == nullptr) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qc_mfma64_pade4_hess_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (P.n_int <= 0) return hipSuccess;
    const int grid = P.n_int < kHGrid64 ? P.n_int : kHGrid64;
    hipLaunchKernelGGL(qc_mfma64_pade4_hess_kernel, dim3(grid), dim3(kHT64), lds, st, P, dZ, dMu, dH, P.hs);
    return hipGetLastError();
}
