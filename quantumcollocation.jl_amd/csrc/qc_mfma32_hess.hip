// f64-MFMA Hessian-of-Lagrangian kernel, order-4 Pade, 2N = 32 (4 qubits, BASELINE config 5), up to 8 drives.
//
// One 512-thread workgroup (8 wavefronts) per interval, wave k owns drive k.  Every 32 x 32 matrix is 2 x 2 tiles
// of 16 x 16, every 32 x 16 matrix (M = reshape(mu_t[0:s], 32, 16), S, D, ...) two tiles; all products are
// v_mfma_f64_16x16x4_f64 with register operands (lane maps: qc_mfma_kernels.hip header).
//
// The blocks (h = dt, c1 = 1/2, c2 = 1/12; M1 = G^T M, M2 = G^T M1, N_k = G_k^T M, V_k = G_k D):
//   (U_t, a_k)   = -c1 h N_k - c2 h^2 X_k        (a_k, U_t+1) = -c1 h N_k + c2 h^2 X_k,   X_k = G_k^T M1 + G^T N_k
//   (U_t, h)     = -(c1 M1 + 2 c2 h M2)           (h, U_t+1)   = -c1 M1 + 2 c2 h M2
//   (a_i, a_k)   = c2 h^2 (<N_i, V_k> + <N_k, V_i>)
//   (a_k, h)     = <N_k, -c1 S + 2 c2 h G D> + 2 c2 h <M1, V_k>          (h, h) = 2 c2 <M1, G D>
//   (dx_i, h)    = -mu_i   (derivative integrators)
// The matrix blocks are produced ALREADY TRANSPOSED (lane <-> row of the stored column-major block, so every store
// instruction writes whole 128-byte lines) without a transposing product: registers holding a tile in B/D layout,
// read as the A operand, are the transposed tile, hence
//   N_k^T = M^T G_k            A = M tiles,   B = B-layout image of G_k
//   X_k^T = M1^T G_k + N_k^T G A = M1 / N_k tiles (as produced),  B = B-layout tiles of G_k / G
//   M1^T  = M^T G,  M2^T = M1^T G   likewise.
// N_k, V_k, M1, G D are also needed untransposed for the scalar blocks: N_k = G_k^T M uses the B-layout image as the
// A operand (A-layout(X^T) = B-layout(X)).  Per drive: 20 16x16x16 products (80 MFMAs); shared work 16 products.
//
//   phase 0   wave w assembles one of the 8 tiles of G = G_0 + sum_k a_k G_k (4 A-layout, 4 B-layout) -> LDS.  Barrier.
//   phase 1   waves 0,1: M1 tile w; waves 2,3: (G D) tile w-2  -> LDS.  Every drive wave: N_k, V_k, N_k^T
//             (6 interleaved accumulator chains); N_k, V_k -> LDS.  Barrier.
//   phase 2   drive wave: X_k^T, the two matrix blocks of its drive, its (a_k, h) entry; waves 4,5: (U_t,h) / (h,U_t+1)
//             column block w-4; wave 6: (h,h); wave 7: (dx,h); then the (a_i,a_k) pairs round-robin over the waves
//             from the N / V tiles in LDS.  Sums are reduced by a fixed xor-butterfly: bit-reproducible.
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kHThreads32 = 512;
constexpr int kHMax32 = 8;          // drives (one wave each)
constexpr int kHGrid32 = 2048;

__device__ inline v4d g_tile(const double* __restrict__ base, int tile, int lane) {   // global or LDS: [tile][pair][lane][2]
    const v2d* p = reinterpret_cast<const v2d*>(base) + tile * 128 + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
__device__ inline void put_tile(double* __restrict__ base, int tile, int lane, const v4d& x) {
    v2d* p = reinterpret_cast<v2d*>(base) + tile * 128 + lane;
    p[0] = v2d{x[0], x[1]};
    p[64] = v2d{x[2], x[3]};
}
__device__ inline double dot4(const v4d& a, const v4d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ inline double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// NQ independent outputs d[q] = a0[q] * b0[q] + a1[q] * b1[q], MFMAs interleaved round-robin over the outputs
template <int NQ>
__device__ __forceinline__ void mm16x2_multi(const v4d (&a0)[NQ], const v4d (&b0)[NQ], const v4d (&a1)[NQ], const v4d (&b1)[NQ],
                                             v4d (&d)[NQ]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NQ; ++q) d[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q][0], b0[q][0], z, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) d[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q][kk], b0[q][kk], d[q], 0, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) d[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q][kk], b1[q][kk], d[q], 0, 0, 0);
    }
}

// lane (g, j) reg r = X[16 J + j][4 r + g] of a column-major 32-row block at p  (a transposed-land tile)
__device__ inline void store_T(double* __restrict__ p, const v4d& x, int J, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qc_st8m<2>(p + (4 * r + g) * 32 + 16 * J + j, x[r]);
}

__global__ __launch_bounds__(kHThreads32, 1) void qc_mfma32_pade4_hess_kernel(const QcParams P, const double* __restrict__ Z,
                                                                              const double* __restrict__ Mu, double* __restrict__ H) {
    __shared__ __attribute__((aligned(16))) double GL[8 * 256];                 // G: tiles 0-3 A-layout (2I+K), 4-7 B-layout (4+2K+J)
    __shared__ __attribute__((aligned(16))) double M1L[2 * 256];                // M1 tiles
    __shared__ __attribute__((aligned(16))) double GDL[2 * 256];                // (G D) tiles
    __shared__ __attribute__((aligned(16))) double NVL[kHMax32 * 4 * 256];      // per drive: N_k[0], N_k[1], V_k[0], V_k[1]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = P.m;
    const int g = lane >> 4, j = lane & 15;
    const bool ft = P.off_dt >= 0;
    const bool drive = w < m;
    const double c1 = P.c[1], c2 = P.c[2];
    const double* __restrict__ GxA = P.Gx;                              // A-layout images [mat][2I+K]
    const double* __restrict__ GxB = P.Gx + (size_t)(m + 1) * 1024;     // B-layout images [mat][2K+J]

    for (int vb = blockIdx.x; vb < P.n_int; vb += gridDim.x) {
        const int b = qc_xcd_remap(vb, P.n_int);
        const long long t = P.t_begin + b;
        const double* __restrict__ z0 = Z + t * (long long)P.zdim;
        const double* __restrict__ z1 = z0 + P.zdim;
        const double* __restrict__ mu = Mu + t * P.F_stride + P.F_off;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
        const double h = ft ? z0[P.off_dt] : P.dt_fixed;
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;

        // ---- loads: this wave's G tile images, its drive's images, M, S, D -------------------------------------
        const double* __restrict__ asm_base = (w < 4 ? GxA : GxB) + (size_t)(w & 3) * 256;
        v4d Gt = g_tile(asm_base, 0, lane);
        v4d GkA[4], GkB[4];
        {
            const int mat = drive ? w + 1 : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                GkA[q] = g_tile(GxA + (size_t)mat * 1024, q, lane);
                GkB[q] = g_tile(GxB + (size_t)mat * 1024, q, lane);
            }
        }
        v4d Mt[2], S[2], D[2];
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const int o = j * 32 + 16 * I + g;
            const double* u0p = z0 + P.off_U + o;
            const double* u1p = z1 + P.off_U + o;
            const double* mp = mu + o;
            const v4d u0 = {u0p[0], u0p[4], u0p[8], u0p[12]};
            const v4d u1 = {u1p[0], u1p[4], u1p[8], u1p[12]};
            Mt[I] = v4d{mp[0], mp[4], mp[8], mp[12]};
            S[I] = u1 + u0;
            D[I] = u1 - u0;
        }
#pragma unroll
        for (int u = 0; u < kHMax32; ++u) {
            if (u < m) Gt += z0[P.off_a + u] * g_tile(asm_base + (size_t)(u + 1) * 1024, 0, lane);
        }
        put_tile(GL, w, lane, Gt);
        __syncthreads();

        // ---- phase 1 ---------------------------------------------------------------------------------------------
        if (w < 2) {          // M1[I] = sum_K (G[K][I])^T M[K]
            v4d a0[1] = {g_tile(GL, 4 + w, lane)}, a1[1] = {g_tile(GL, 4 + 2 + w, lane)}, b0[1] = {Mt[0]}, b1[1] = {Mt[1]}, d[1];
            mm16x2_multi<1>(a0, b0, a1, b1, d);
            put_tile(M1L, w, lane, d[0]);
        } else if (w < 4) {   // (G D)[I] = sum_K G[I][K] D[K]
            const int I = w - 2;
            v4d a0[1] = {g_tile(GL, 2 * I, lane)}, a1[1] = {g_tile(GL, 2 * I + 1, lane)}, b0[1] = {D[0]}, b1[1] = {D[1]}, d[1];
            mm16x2_multi<1>(a0, b0, a1, b1, d);
            put_tile(GDL, I, lane, d[0]);
        }
        v4d Nk[2], Vk[2], NkT[2];
        if (drive) {
            // N_k[I] = sum_K (G_k[K][I])^T M[K];  V_k[I] = sum_K G_k[I][K] D[K];  N_k^T[J] = sum_K M[K]^T G_k[K][J]
            v4d a0[6] = {GkB[0], GkB[1], GkA[0], GkA[2], Mt[0], Mt[0]};
            v4d b0[6] = {Mt[0], Mt[0], D[0], D[0], GkB[0], GkB[1]};
            v4d a1[6] = {GkB[2], GkB[3], GkA[1], GkA[3], Mt[1], Mt[1]};
            v4d b1[6] = {Mt[1], Mt[1], D[1], D[1], GkB[2], GkB[3]};
            v4d d[6];
            mm16x2_multi<6>(a0, b0, a1, b1, d);
            Nk[0] = d[0]; Nk[1] = d[1]; Vk[0] = d[2]; Vk[1] = d[3]; NkT[0] = d[4]; NkT[1] = d[5];
            double* nv = NVL + w * 1024;
            put_tile(nv, 0, lane, Nk[0]);
            put_tile(nv, 1, lane, Nk[1]);
            put_tile(nv, 2, lane, Vk[0]);
            put_tile(nv, 3, lane, Vk[1]);
        }
        __syncthreads();

        // ---- phase 2 ---------------------------------------------------------------------------------------------
        const v4d M1a = g_tile(M1L, 0, lane), M1b = g_tile(M1L, 1, lane);
        if (drive) {
            // X_k^T[J] = sum_K M1[K]^T G_k[K][J] + sum_K N_k[K]^T G[K][J]
            v4d a0[2] = {M1a, M1a}, b0[2] = {GkB[0], GkB[1]}, a1[2] = {M1b, M1b}, b1[2] = {GkB[2], GkB[3]}, x0[2];
            mm16x2_multi<2>(a0, b0, a1, b1, x0);
            v4d a2[2] = {Nk[0], Nk[0]}, b2[2] = {g_tile(GL, 4, lane), g_tile(GL, 5, lane)}, a3[2] = {Nk[1], Nk[1]},
                b3[2] = {g_tile(GL, 6, lane), g_tile(GL, 7, lane)}, x1[2];
            mm16x2_multi<2>(a2, b2, a3, b3, x1);
            double* pUa = Hb + P.ho_Ua + (size_t)w * 512;
            double* paU = Hb + P.ho_aU + (size_t)w * 512;
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const v4d lin = (-hc1) * NkT[J], q = hc2 * (x0[J] + x1[J]);
                store_T(pUa, lin - q, J, g, j);
                store_T(paU, lin + q, J, g, j);
            }
            if (ft) {   // (a_k, h)
                const v4d gd0 = g_tile(GDL, 0, lane), gd1 = g_tile(GDL, 1, lane);
                const v4d wl0 = (-c1) * S[0] + c2h2 * gd0, wl1 = (-c1) * S[1] + c2h2 * gd1;
                const double part = (dot4(Nk[0], wl0) + dot4(Nk[1], wl1)) + c2h2 * (dot4(M1a, Vk[0]) + dot4(M1b, Vk[1]));
                const double sum = wave_sum(part);
                if (lane == 0) Hb[P.ho_ah + w] = sum;
            }
        }
        if (ft) {
            if (w == 4 || w == 5) {   // (U_t,h)^T and (h,U_t+1)^T, column block J
                const int J = w - 4;
                const v4d gb0 = g_tile(GL, 4 + J, lane), gb1 = g_tile(GL, 4 + 2 + J, lane);
                v4d a0[2] = {Mt[0], M1a}, b0[2] = {gb0, gb0}, a1[2] = {Mt[1], M1b}, b1[2] = {gb1, gb1}, d[2];
                mm16x2_multi<2>(a0, b0, a1, b1, d);      // M1^T[J], M2^T[J]
                store_T(Hb + P.ho_Uh, -(c1 * d[0] + c2h2 * d[1]), J, g, j);
                store_T(Hb + P.ho_hU, (-c1) * d[0] + c2h2 * d[1], J, g, j);
            } else if (w == 6) {      // (h, h)
                const double part = dot4(M1a, g_tile(GDL, 0, lane)) + dot4(M1b, g_tile(GDL, 1, lane));
                const double sum = wave_sum(part);
                if (lane == 0) Hb[P.ho_hh] = 2.0 * c2 * sum;
            } else if (w == 7) {      // derivative integrators: d2/d(dx_i) dh = -mu_i
                int r0 = P.s, o = P.ho_d;
                for (int d = 0; d < P.n_deriv; ++d) {
                    for (int i = lane; i < P.ddim_i[d]; i += 64) Hb[o + i] = -mu[r0 + i];
                    r0 += P.ddim_i[d];
                    o += P.ddim_i[d];
                }
            }
        }
        // (a_u, a_v), u <= v, at v(v+1)/2 + u
        {
            const int npair = m * (m + 1) / 2;
            int v = 0, u = 0;
            for (int p = 0; p < npair; ++p) {
                if ((p & 7) == w) {
                    const double* nu = NVL + u * 1024;
                    const double* nvp = NVL + v * 1024;
                    const double part = (dot4(g_tile(nu, 0, lane), g_tile(nvp, 2, lane)) + dot4(g_tile(nu, 1, lane), g_tile(nvp, 3, lane))) +
                                        (dot4(g_tile(nvp, 0, lane), g_tile(nu, 2, lane)) + dot4(g_tile(nvp, 1, lane), g_tile(nu, 3, lane)));
                    const double sum = wave_sum(part);
                    if (lane == 0) Hb[P.ho_aa + p] = hc2 * sum;
                }
                if (++u > v) { u = 0; ++v; }
            }
        }
        __syncthreads();   // the LDS blocks are rewritten by the next interval of a persistent grid
    }
}

}  // namespace

bool qc_mfma32_hess_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n == 32 && P.nc == P.N && P.m <= kHMax32 && P.Gx != nullptr;
}

hipError_t qc_launch_mfma32_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    const int grid = P.n_int < kHGrid32 ? P.n_int : kHGrid32;
    hipLaunchKernelGGL(qc_mfma32_pade4_hess_kernel, dim3(grid), dim3(kHThreads32), 0, st, P, dZ, dMu, dH);
    return hipGetLastError();
}
