// Device helpers shared by the two-wave 2N = 16 kernels of qc_mfma_fused.hip (F + dF + mu_d2F in one launch) and qc_mfma_hess2.hip
// (mu_d2F alone): tiles through LDS, the scalar-block rows and their fixed-order reduction (qc_mfma16_pade4_hess_anti_kernel's).
#pragma once
#include "qc_mfma_common.h"

namespace qc_mfma {

constexpr int kFuStride = 65;                // row stride of the reduction scratch (qc_mfma_hess.hip: kHStride)

__device__ inline v4d fu_load_GA(const double* __restrict__ Gx, int mat, int lane) { return load_image_tile(Gx + mat * 256, lane); }
__device__ inline double fu_dot4(const v4d& a, const v4d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ inline v4d fu_sel(bool c, const v4d& a, const v4d& b) { return v4d{c ? a[0] : b[0], c ? a[1] : b[1], c ? a[2] : b[2], c ? a[3] : b[3]}; }
__device__ inline void fu_lds_put(double* __restrict__ base, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}
__device__ inline v4d fu_lds_get(const double* __restrict__ base, int lane) {
    const v2d* p = reinterpret_cast<const v2d*>(base) + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
// lane (g, j) reg r = col[4 r + g] of column `col` of a 16-row column-major block
__device__ inline v4d fu_load_col(const double* __restrict__ base, int col, int g) {
    const double* p = base + col * 16 + g;
    return v4d{p[0], p[4], p[8], p[12]};
}
// transposed tile: lane (g, j) reg r holds X[j][4r+g] of a 16 x 16 column-major block at p
__device__ inline void fu_store_tile_T(double* __restrict__ p, const v4d& x, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8m<2>(p + (4 * r + g) * 16 + j, x[r]);
}
__device__ __forceinline__ void fu_st_off(double* __restrict__ ubase, unsigned byteoff, double v) {
    qc_st8m<2>(reinterpret_cast<double*>(reinterpret_cast<char*>(ubase) + byteoff), v);
}

// ---- drive generators with ONE entry per row (Pauli strings: BASELINE configs 3 / 4), qc_mfma16_ell_build's tables ----------------------
// Every product with a drive image, G_u X, is then a row gather from a row-major LDS copy of X, with the SAME BITS as the dense-image
// product (v_mfma_f64_16x16x4_f64 is an ascending-k chain of fused multiply-adds, tests/hip/mfma_f64_fma_probe.hip: fifteen exact zeros
// and one fma(w, x, acc) per element).  Used by the one-call kernel (qc_mfma_fused.hip) and the mu_d2F kernels (qc_mfma_hess.hip, hess2).
constexpr int kXS = 17;                      // row stride of the row-major LDS copies the gathers read (doubles)

// T_u = G_u X for every drive, G_u with one entry per row: lane (g, j) reg r = w_u[4 r + g] * X[c_u[4 r + g]][j], X from its row-major
// copy xs (tw: [u][16] weights, tc: [u][16] columns x kXS, both in LDS; unused drive slots carry weight 0, column 0).  fma(w, x, +0):
// what the dense product's accumulator chain holds.
template <int kMU>
__device__ __forceinline__ void fu_gather_all(const double* __restrict__ tw, const int* __restrict__ tc, const double* __restrict__ xs, int g, int j,
                                              v4d (&out)[kMU]) {
#pragma unroll
    for (int u = 0; u < kMU; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = u * 16 + 4 * r + g;
            out[u][r] = __builtin_fma(tw[row], xs[tc[row] + j], 0.0);
        }
    }
}
// lane (g, j) reg r = X[4 r + g][j]  ->  xs[(4 r + g) * kXS + j]
__device__ __forceinline__ void fu_put_rows(double* __restrict__ xs, const v4d& x, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(4 * r + g) * kXS + j] = x[r];
}
__device__ __forceinline__ void fu_lds_order() {     // a wave's LDS operations execute in order; this keeps the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The tables from the handle's blob (P.ell16: [6][16] weights, [6][16] columns x kXS) into LDS: 96 entries, lanes l and 64 + l.
__device__ __forceinline__ void fu_load_tables(const void* blob, int m, int lane, double* __restrict__ tw_lds, int* __restrict__ tc_lds) {
    typedef const __attribute__((address_space(1))) double* gdp;
    typedef const __attribute__((address_space(1))) int* gip;
    const gdp tw = (gdp)(unsigned long long)blob;
    const gip tc = (gip)(unsigned long long)((const char*)blob + 6 * 16 * 8);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = 64 * q + lane;
        if (e < 96) {
            tw_lds[e] = e < 16 * m ? tw[e] : 0.0;
            tc_lds[e] = e < 16 * m ? tc[e] : 0;
        }
    }
}
// Stage B's drive terms, Q_p += G_2p [y | 0] + G_2p+1 [0 | y]: a left lane's accumulator takes the one term w_2p[a] y[c_2p[a]][j], a right
// lane's w_2p+1[a] y[c_2p+1[a]][j - 8]; every other term of the two dense chains is an exact zero.  ys: the row-major copy of [y | 0].
__device__ __forceinline__ void fu_gather_pair(const double* __restrict__ tw, const int* __restrict__ tc, const double* __restrict__ ys, int pair, bool left,
                                               int g, int jj, v4d& Q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (left ? 2 * pair : 2 * pair + 1) * 16 + 4 * r + g;
        Q[r] = __builtin_fma(tw[row], ys[tc[row] + jj], Q[r]);
    }
}

template <int kMU>
struct FuRows {
    static constexpr int kAA = kMU * (kMU + 1) / 2, kPair = kMU / 2, kRows = kAA + kPair + 1;
};

// Sums the 64 per-lane partials of the scalar-block rows [row_begin, row_end) (row r at red + (r - row_shift) * kFuStride) in the fixed
// order of qc_mfma16_pade4_hess_anti_kernel -- lanes l and l + 32 take the left / right lanes' columns of row l -- and stores them.
template <int kMU>
__device__ __forceinline__ void fu_reduce_rows(const QcParams& P, const double* __restrict__ red, double* __restrict__ Hb, int lane, int m, bool ft,
                                               int row_begin, int row_end, int row_shift, bool staged = false, double* __restrict__ scal = nullptr) {
    using R = FuRows<kMU>;
    const int naa = m * (m + 1) / 2;
    const int half = lane >> 5;
    for (int base = row_begin; base < row_end; base += 32) {
        const int row = base + (lane & 31);
        const bool in = row < row_end;
        const bool pair_row = row >= R::kAA && row < R::kAA + R::kPair, hh_row = row == R::kAA + R::kPair;
        const int drive = 2 * (row - R::kAA) + half;
        const bool wanted = in && (row < naa || (ft && ((pair_row && drive < m) || (hh_row && half == 0))));
        const double* rp = red + ((in ? row : row_begin) - row_shift) * kFuStride + 8 * half;
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
            // staged: the interval's scalar run [(a, a) | (a, h) | (h, h) | (dx, h) | padding] is collected in LDS (`scal`, P.ho_aa at
            // index 0) and stored in one piece by fu_scalar_run_store; otherwise the entries go out one by one.  (Two branches, not one
            // store through a selected pointer: that would be a FLAT store, which queues behind every global store of the wave.)
            if (staged) {
                if (row < naa) {
                    if (half == 0) scal[row] = both;
                } else if (row >= R::kAA) {
                    scal[(pair_row ? P.ho_ah + drive : P.ho_hh) - P.ho_aa] = own;
                }
            } else {
                if (row < naa) {
                    if (half == 0) Hb[P.ho_aa + row] = both;
                } else if (row >= R::kAA) {
                    Hb[pair_row ? P.ho_ah + drive : P.ho_hh] = own;
                }
            }
        }
    }
}

// The scalar entries of an interval's Hessian block -- (a, a), (a, h), (h, h), the derivative integrators' (dx, h) and the alignment
// padding: one contiguous run behind the matrix blocks (qc_host.cpp) -- come from both waves of the workgroup, a few lanes at a time,
// at the very end of their lives.  Stored where they arise they are a dozen partly written lines per interval at the tail of the launch:
// 0.7 us of the one-call launch at config 3 (profiles/NOTES.md, round 5).  Here each wave leaves its entries in LDS (`scal`, P.ho_aa
// at index 0) and calls this; the wave that arrives LAST stores the run in one piece (three whole lines at config 3).  `count`: one
// LDS word, zero before either wave can get here.  Same values, same bits.
constexpr int kFuScalMax = 96;               // entries of the staged run (48 at config 3; the one-call kernel has 105 doubles of LDS to spare at six drives)
__device__ __forceinline__ int fu_scalar_run_len(const QcParams& P) { return P.hess_nnz + P.h_pad - P.ho_aa; }
__device__ __forceinline__ void fu_scalar_run_store(const QcParams& P, const double* __restrict__ scal, int* __restrict__ count, double* __restrict__ Hb, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old == 0) return;                    // the other wave is still at work: it stores
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int n = fu_scalar_run_len(P);
    for (int i = lane; i < n; i += 64) Hb[P.ho_aa + i] = scal[i];
}

}  // namespace qc_mfma
