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
        const double both = own + __shfl_xor(own, 32, 64);
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
