// Register-resident f64-MFMA kernels (v_mfma_f64_16x16x4_f64) for the order-4 Pade integrator.
//
// n = 2N = 16 (3 qubits, BASELINE configs 3 and 4): ONE WAVEFRONT PER INTERVAL.  Every 16x16x16
// product is 4 MFMAs whose operands never leave the register file:
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
//   * The constant generators G_0, G_j are staged once per workgroup in LDS, pre-packed in both
//     operand layouts in lane order (conflict-free ds_read_b128); the knot data of the first
//     interval is requested before that staging so the two latencies overlap.
//
// Per interval (S = U1+U0, D = U1-U0, h = dt):                                          MFMAs
//   (G^2)^T = G^T G^T ;  B^T, F^T = I -+ h/2 G^T + h^2/12 (G^2)^T   (stored N times each)    4
//   P1 = G [S | D]      = [GS | GD]                                                          4
//   P2 = G [GD | GS]    = [G^2 D | .]                                                        4
//   E  = [delta | d/dh] = [D - h/2 GS + h^2/12 G^2 D | -1/2 GS + h/6 G^2 D],  E^T            4
//   Q  = [ -h/2 S + h^2/12 GD | h^2/12 D ]
//   per drive pair (j, j+1):  R_j = G_j Q, R_j+1 = G_j+1 Q, T = G [R_j(right) | R_j+1(right)],
//        Y = [R_j(left) + T(left) | R_j+1(left) + T(right)] = [d/da_j | d/da_j+1],  Y^T     16
// = 16 + 8 m MFMAs (64 for m = 6).  The B/F copies (80 % of the interval's bytes) are stored first
// so HBM writes start while the drive columns are still being computed.
#include "qc_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kWaves = 4;                  // intervals per workgroup
constexpr int kThreads = 64 * kWaves;
constexpr int kMaxGrid = 1024;             // persistent beyond this many workgroups
constexpr int kStage = 8;                  // 16-byte loads in flight per thread while staging

__device__ inline double swap8(double x) {  // exchange the two 8-column halves of a 16-column tile
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);  // row_ror:8
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline v4d swap8(v4d x) { return v4d{swap8(x[0]), swap8(x[1]), swap8(x[2]), swap8(x[3])}; }

// D = A * B (16x16x16): A in A-layout regs, B in B-layout regs
__device__ inline v4d mm16(const v4d& a, const v4d& b) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc, 0, 0, 0);
    return acc;
}

// LDS image of the constants: [layout(2)][matrix(m+1)][pair(2)][lane(64)][2] doubles
__device__ inline v4d lds_mat(const double* __restrict__ base, int mat, int lane) {
    const v2d lo = *reinterpret_cast<const v2d*>(base + ((mat * 2 + 0) * 64 + lane) * 2);
    const v2d hi = *reinterpret_cast<const v2d*>(base + ((mat * 2 + 1) * 64 + lane) * 2);
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}

// Store a transposed tile: lane (g, j) reg r holds X[j][4r+g] of a 16 x 16 column-major block at p.
__device__ inline void store_tile_T(double* __restrict__ p, const v4d& x, int g, int j, int mode) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8(p + (4 * r + g) * 16 + j, x[r], mode);
}

struct KnotRegs {
    v4d u0, u1;
};

__device__ inline KnotRegs load_knots(const QcParams& P, const double* __restrict__ Z, long long t, int g, int jj) {
    const double* u0p = Z + t * (long long)P.zdim + P.off_U + jj * 16 + g;
    const double* u1p = u0p + P.zdim;
    KnotRegs k;
    k.u0 = v4d{u0p[0], u0p[4], u0p[8], u0p[12]};
    k.u1 = v4d{u1p[0], u1p[4], u1p[8], u1p[12]};
    return k;
}

template <bool JAC>
__global__ __launch_bounds__(kThreads) void qc_mfma16_pade4_kernel(const QcParams P, const double* __restrict__ Z,
                                                                   double* __restrict__ F, double* __restrict__ J,
                                                                   int n_wg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = P.m;
    const int matsz = 256;                       // doubles per packed matrix
    const double* ldsA = sm;                     // A-operand images
    const double* ldsB = sm + (m + 1) * matsz;   // B/D-operand images
    const int g = lane >> 4, j = lane & 15, jj = j & 7;
    const bool left = j < 8;
    const bool ft = P.off_dt >= 0;
    const double c1 = P.c[1], c2 = P.c[2];
    const int sm_mode = P.store_mode;

    // knot data of the first interval: in flight while the constants are staged
    int vb = blockIdx.x;
    int b = qc_xcd_remap(vb, n_wg) * kWaves + wave;
    KnotRegs kn = load_knots(P, Z, P.t_begin + (b < P.n_int ? b : 0), g, jj);

    {   // stage the constants: 2 (m+1) 256 doubles, kStage 16-byte loads in flight per thread
        const int total2 = (m + 1) * matsz;
        const v2d* __restrict__ src = reinterpret_cast<const v2d*>(P.Gx);
        v2d* dst = reinterpret_cast<v2d*>(sm);
        for (int base = tid; base < total2; base += kThreads * kStage) {
            v2d tmp[kStage];
#pragma unroll
            for (int u = 0; u < kStage; ++u) {
                const int i = base + u * kThreads;
                tmp[u] = i < total2 ? src[i] : v2d{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < kStage; ++u) {
                const int i = base + u * kThreads;
                if (i < total2) dst[i] = tmp[u];
            }
        }
    }
    __syncthreads();

    const v4d IdB = {(g == j) ? 1.0 : 0.0, (4 + g == j) ? 1.0 : 0.0, (8 + g == j) ? 1.0 : 0.0, (12 + g == j) ? 1.0 : 0.0};

    for (; vb < n_wg; vb += gridDim.x) {
        b = qc_xcd_remap(vb, n_wg) * kWaves + wave;   // local interval of this wave
        const v4d u0 = kn.u0, u1 = kn.u1;
        {   // request the next interval's knots now (persistent grids only)
            const int vn = vb + gridDim.x;
            if (vn < n_wg) {
                const int bn = qc_xcd_remap(vn, n_wg) * kWaves + wave;
                kn = load_knots(P, Z, P.t_begin + (bn < P.n_int ? bn : 0), g, jj);
            }
        }
        if (b >= P.n_int) continue;
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        const double h = ft ? z0[P.off_dt] : P.dt_fixed;

        // ---- G in both operand layouts ---------------------------------------------------------------
        v4d Ga = lds_mat(ldsA, 0, lane), Gb = lds_mat(ldsB, 0, lane);
        for (int k = 0; k < m; ++k) {
            const double a = z0[P.off_a + k];
            const v4d xa = lds_mat(ldsA, k + 1, lane), xb = lds_mat(ldsB, k + 1, lane);
            Ga += a * xa;
            Gb += a * xb;
        }
        double* __restrict__ Jb = JAC ? J + (size_t)b * P.jac_nnz : nullptr;
        double* __restrict__ Fb = F ? F + (size_t)b * P.ddim : nullptr;
        const double hc1 = h * c1, hc2 = h * h * c2;

        if (JAC) {
            // ---- B^T, F^T and their N copies (issued first: 80 % of the interval's bytes) ------------
            // A-layout(G^T) = B-layout(G) = Gb;  B-layout(G^T) = A-layout(G) = Ga;  D-layout(G^T) = Ga.
            const v4d G2T = mm16(Gb, Ga);
            v4d Fm, Bm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ev = IdB[r] + hc2 * G2T[r];
                Fm[r] = -(ev + hc1 * Ga[r]);       // -F^T
                Bm[r] = ev - hc1 * Ga[r];          //  B^T
            }
            double* pF = Jb + P.jo_F;
            double* pB = Jb + P.jo_B;
#pragma unroll
            for (int q = 0; q < 8; ++q) store_tile_T(pF + q * 256, Fm, g, j, sm_mode);
#pragma unroll
            for (int q = 0; q < 8; ++q) store_tile_T(pB + q * 256, Bm, g, j, sm_mode);
        }

        // ---- Krylov products -----------------------------------------------------------------------
        v4d W, Wsw;                               // W = [S | D], Wsw = [D | S]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double sm_ = u1[r] + u0[r], df = u1[r] - u0[r];
            W[r] = left ? sm_ : df;
            Wsw[r] = left ? df : sm_;
        }
        const v4d P1 = mm16(Ga, W);               // [GS | GD]
        const v4d P1sw = swap8(P1);               // [GD | GS]
        const v4d P2 = mm16(Ga, P1sw);            // [G^2 D | G^2 S]
        {   // E = [delta | d/dh] (values are formed on the left half, d/dh moved to the right half)
            v4d dl, dh;
            const double d1 = -c1, d2 = 2.0 * c2 * h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dl[r] = Wsw[r] - hc1 * P1[r] + hc2 * P2[r];
                dh[r] = d1 * P1[r] + d2 * P2[r];
            }
            const v4d dhs = swap8(dh);
            v4d E;
#pragma unroll
            for (int r = 0; r < 4; ++r) E[r] = left ? dl[r] : dhs[r];
            const v4d ET = mm16(E, IdB);          // lane (g, j) reg r = E[j][4r+g]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + g;          // tile column: < 8 residual column c, >= 8 d/dh column c-8
                if (c < 8) {
                    if (Fb) qc_st8(Fb + c * 16 + j, ET[r], sm_mode);
                } else if (JAC && ft) {
                    qc_st8(Jb + P.jo_h + (c - 8) * 16 + j, ET[r], sm_mode);
                }
            }
        }
        // derivative integrators (a few lanes)
        {
            int r0 = P.s, jo = P.jo_d;
            for (int d = 0; d < P.n_deriv; ++d) {
                const int dim = P.ddim_i[d];
                for (int i = lane; i < dim; i += 64) {
                    const double dx = z0[P.dx_off[d] + i];
                    if (Fb) Fb[r0 + i] = z1[P.x_off[d] + i] - z0[P.x_off[d] + i] - h * dx;
                    if (JAC) {
                        Jb[jo + i] = -1.0;
                        Jb[jo + dim + i] = 1.0;
                        Jb[jo + 2 * dim + i] = -h;
                        if (ft) Jb[jo + 3 * dim + i] = -dx;
                    }
                }
                r0 += dim;
                jo += (ft ? 4 : 3) * dim;
            }
        }
        if (!JAC) continue;

        // ---- drive columns ---------------------------------------------------------------------------
        v4d Q;                                    // [Q0 | Q1] = [-h c1 S + h^2 c2 GD | h^2 c2 D]
#pragma unroll
        for (int r = 0; r < 4; ++r) Q[r] = left ? (-hc1 * W[r] + hc2 * P1sw[r]) : (hc2 * W[r]);
        double* pa = Jb + P.jo_a;
        int k = 0;
        for (; k + 1 < m; k += 2) {
            const v4d R1 = mm16(lds_mat(ldsA, k + 1, lane), Q);    // [G_k Q0 | G_k Q1]
            const v4d R2 = mm16(lds_mat(ldsA, k + 2, lane), Q);    // [G_k+1 Q0 | G_k+1 Q1]
            const v4d R1sw = swap8(R1), R2sw = swap8(R2);
            v4d X, Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[r] = left ? R1sw[r] : R2[r];     // [G_k Q1 | G_k+1 Q1]
                Y[r] = left ? R1[r] : R2sw[r];     // [G_k Q0 | G_k+1 Q0]
            }
            const v4d Tt = mm16(Ga, X);            // [G G_k Q1 | G G_k+1 Q1]
            Y += Tt;                               // [d/da_k | d/da_k+1]
            store_tile_T(pa + (size_t)k * 128, mm16(Y, IdB), g, j, sm_mode);
        }
        if (k < m) {                               // odd drive count: last column alone
            const v4d R1 = mm16(lds_mat(ldsA, k + 1, lane), Q);
            const v4d Tt = mm16(Ga, swap8(R1));
            const v4d YT = mm16(R1 + Tt, IdB);     // left half valid: tile columns 0..7
            double* p = pa + (size_t)k * 128;
#pragma unroll
            for (int r = 0; r < 2; ++r) qc_st8(p + (4 * r + g) * 16 + j, YT[r], sm_mode);
        }
    }
}

}  // namespace

bool qc_mfma_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n == 16 && P.m <= 32;
}

size_t qc_mfma_gx_doubles(const QcParams& P) { return (size_t)2 * (P.m + 1) * 256; }

// Packs the (m+1) generators (column-major n x n, index 0 = drift) into the LDS image
// [layout][matrix][pair][lane][2]:  A-layout lane (g, i) reg kk = X[i][4kk+g];  B/D-layout lane (g, j) reg r = X[4r+g][j].
void qc_mfma_pack_G(const QcParams& P, const double* G, double* Gx) {
    const int n = 16, M = P.m + 1;
    for (int mat = 0; mat < M; ++mat) {
        const double* A = G + (size_t)mat * n * n;
        auto At = [&](int row, int col) { return A[(size_t)col * n + row]; };  // col-major
        for (int pr = 0; pr < 2; ++pr)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 2; ++e) {
                    const int g = l >> 4, i = l & 15, r = 2 * pr + e;
                    Gx[(((size_t)0 * M + mat) * 2 + pr) * 128 + l * 2 + e] = At(i, 4 * r + g);      // A-layout
                    Gx[(((size_t)1 * M + mat) * 2 + pr) * 128 + l * 2 + e] = At(4 * r + g, i);      // B/D-layout
                }
    }
}

hipError_t qc_launch_mfma_F_jac(const QcParams& P, const double* dZ, double* dF, double* dJ, hipStream_t st) {
    const int n_wg = (P.n_int + kWaves - 1) / kWaves;
    const int grid = n_wg < kMaxGrid ? n_wg : kMaxGrid;
    const size_t lds = qc_mfma_gx_doubles(P) * sizeof(double);
    if (dJ) hipLaunchKernelGGL(qc_mfma16_pade4_kernel<true>, dim3(grid), dim3(kThreads), lds, st, P, dZ, dF, dJ, n_wg);
    else hipLaunchKernelGGL(qc_mfma16_pade4_kernel<false>, dim3(grid), dim3(kThreads), lds, st, P, dZ, dF, dJ, n_wg);
    return hipGetLastError();
}

hipError_t qc_launch_mfma_hess(const QcParams&, const double*, const double*, double*, hipStream_t) {
    return hipErrorNotSupported;
}
