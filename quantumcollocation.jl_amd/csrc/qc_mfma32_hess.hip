// f64-MFMA Hessian-of-Lagrangian kernel, order-4 Pade, 2N = 32 (4 qubits, BASELINE config 5), up to 8 drives.
//
// One 512-thread workgroup (8 wavefronts) per compute unit; it walks a contiguous run of intervals, wave k owns drive k
// and keeps that drive's generator images (A- and B-layout, 16 KB) in registers for the whole run.  Every 32 x 32
// matrix is 2 x 2 tiles of 16 x 16, every 32 x 16 matrix (M = reshape(mu_t[0:s], 32, 16), S, D, ...) two tiles; all
// products are v_mfma_f64_16x16x4_f64 with register operands (lane maps: qc_mfma_kernels.hip header).
//
// The blocks (h = dt, c1 = 1/2, c2 = 1/12; M1 = G^T M, M2 = G^T M1, N_k = G_k^T M, V_k = G_k D):
//   (U_t, a_k)   = -c1 h N_k - c2 h^2 X_k        (a_k, U_t+1) = -c1 h N_k + c2 h^2 X_k,   X_k = G_k^T M1 + G^T N_k
//   (U_t, h)     = -(c1 M1 + 2 c2 h M2)           (h, U_t+1)   = -c1 M1 + 2 c2 h M2
//   (a_i, a_k)   = c2 h^2 (<N_i, V_k> + <N_k, V_i>)
//   (a_k, h)     = <N_k, -c1 S + 2 c2 h G D> + 2 c2 h <M1, V_k>          (h, h) = 2 c2 <M1, G D>
//   (dx_i, h)    = -mu_i   (derivative integrators)
// The matrix blocks are produced ALREADY TRANSPOSED (lane <-> row of the stored column-major block, so every store
// instruction writes whole 128-byte lines): registers holding a tile in B/D layout, read as the A operand, are the
// transposed tile, hence X_k^T = M1^T G_k + N_k^T G and M2^T = M1^T G take the tiles M1, N_k as they were produced
// (A operand) against B-layout tiles of G_k / G.  Plain transposes (N_k^T, M1^T, B-layout of G from its A-layout) go
// through a padded LDS scratch instead of an identity product.  N_k = G_k^T M uses the B-layout image as the A
// operand (A-layout(X^T) = B-layout(X)).  Per drive and interval: 16 16x16x16 products (64 MFMAs); shared: 8.
//
//   phase 0   wave w assembles half of one A-layout tile of G = G_0 + sum_k a_k G_k (nine 1 KB image loads, the only
//             per-interval L2 traffic besides the knots); waves 4-7 fetch M, U_t, U_t+1 tiles -> LDS.  Barrier.
//   phase 1   waves 0,1: B-layout tiles of G (LDS transposes) and M1 tile w; waves 2,3: (G D) tile w-2  -> LDS.
//             Every drive wave: N_k, V_k (4 interleaved accumulator chains), N_k^T;  N_k, V_k -> LDS.  Barrier.
//   phase 2   drive wave: X_k^T and the two matrix blocks of its drive; its scalar blocks (a_k,h), (a_k,a_k) and the
//             pairs {k, (k+d) mod m} from its registers and the partner's tiles in LDS.  Waves w and w+4 share a SIMD:
//             waves 0-3 run the MFMA chain first, waves 4-7 the LDS/VALU part first.  Waves 4,5: (U_t,h) / (h,U_t+1)
//             column block; wave 6: (h,h).  Sums by DPP row rotations + v_readlane in a fixed order: bit-reproducible.
// Measured history (config 5, T = 500; profiles/README.md): LDS kernel 1782 us; one interval per workgroup with all
// images re-read 45.7 us (load phase L2-bandwidth-bound, 7 us); this design 28.9 us (MFMA pipes ~55 % busy).
#include "qc_mfma_common.h"

namespace {

using namespace qc_mfma;

constexpr int kHThreads32 = 512;
constexpr int kHMax32 = 8;          // drives (one wave each)
constexpr int kHCUs = 256;          // compute units of an MI355X

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
// Sum over the 64 lanes without the LDS crossbar: four DPP row rotations leave every lane of a 16-lane row with the
// row sum, four v_readlane pick the rows up.  Fixed order: bit-reproducible.  The result is wave-uniform.
template <int CTRL>
__device__ inline double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ inline double row_sum(double x) {
    x += dpp_f64<0x128>(x);   // row_ror:8
    x += dpp_f64<0x124>(x);   // row_ror:4
    x += dpp_f64<0x122>(x);   // row_ror:2
    x += dpp_f64<0x121>(x);   // row_ror:1
    return x;
}
__device__ inline double wave_sum(double x) {
    x = row_sum(x);
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}

// N sums at once: every stage of the reduction for all N values before the next stage, so that no instruction waits for its
// predecessor (one at a time, each of the 4 DPP stages and the read-lane tail is a chain of dependent instructions).  Same
// order of additions per value as wave_sum: bit-identical results.
template <int N>
__device__ __forceinline__ void wave_sum_multi(double (&x)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x128>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x124>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x122>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] += dpp_f64<0x121>(x[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = (readlane_f64(x[q], 0) + readlane_f64(x[q], 16)) + (readlane_f64(x[q], 32) + readlane_f64(x[q], 48));
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
// FULL: 16 levels and 16 columns exactly (a 4-qubit unitary): no masks -- each mask is an exec-mask save / restore pair around its
// store, 85 of them per interval, and the scalar registers they occupy are spilled to vector-register lanes
template <bool FULL>
__device__ inline void store_T(double* __restrict__ p, const v4d& x, int J, int g, int j, int nc, int nr) {
    if constexpr (FULL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) qc_st8m<2>(p + (4 * r + g) * 32 + 16 * J + j, x[r]);
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * r + g < nc && 16 * J + j < nr) qc_st8m<2>(p + (4 * r + g) * nr + 16 * J + j, x[r]);
}

// ANTI: every generator is exactly antisymmetric (QcParams.antisym): the B-layout image tile (K, J) is minus the A-layout
// tile (J, K), so only the A-layout images are fetched (half of the workgroup's one-time 128 KB image load)
template <bool DIAG, bool ANTI, bool FULL = false>
__global__ __launch_bounds__(kHThreads32, 1) void qc_mfma32_pade4_hess_kernel(const double* __restrict__ hot_Gx, const double* __restrict__ hot_Zt,
                                                                              const double* __restrict__ hot_mu0, const int hot_n_int, const int per_wg,
                                                                              const int hot_zdim, const int hot_m, const int hot_off_a, const int hot_off_dt,
                                                                              const int hot_off_U, const int hot_f_stride, const QcParams Pk,
                                                                              double* __restrict__ H) {
    // (the leading arguments are preloaded into scalar registers at wave launch -- -amdgpu-kernarg-preload-count --: every load
    //  request of phase 0 depends on them only; hot_Zt / hot_mu0 = the handle's first knot / first interval's multipliers)
    QcKernargTouch<sizeof(QcParams) + 96> touch;   // one batch of scalar-cache misses instead of one per use (qc_internal.h):
    touch.request();                               // requested here, waited for behind the drive images' requests
    __shared__ __attribute__((aligned(16))) double GL[8 * 256];                 // G: tiles 0-3 A-layout (2I+K), 4-7 B-layout (4+2K+J)
    __shared__ __attribute__((aligned(16))) double ML[2 * 256];                 // M tiles
    __shared__ __attribute__((aligned(16))) double DL[2 * 256];                 // D = U_t+1 - U_t tiles
    __shared__ __attribute__((aligned(16))) double SL[2 * 256];                 // S = U_t+1 + U_t tiles
    __shared__ __attribute__((aligned(16))) double M1L[2 * 256];                // M1 tiles
    __shared__ __attribute__((aligned(16))) double GDL[2 * 256];                // (G D) tiles
    __shared__ __attribute__((aligned(16))) double WLL[2 * 256];                // (-c1 S + 2 c2 h G D) tiles
    __shared__ __attribute__((aligned(16))) double NVL[kHMax32 * 4 * 256];      // per drive: N_k[0], N_k[1], V_k[0], V_k[1]
    __shared__ double TS[8 * 16 * 17];                                          // per-wave transpose scratch
    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = hot_m;
    const bool ft = hot_off_dt >= 0;
    const bool drive0 = w0 < m0;
    const double* __restrict__ GxA = hot_Gx;                              // A-layout images [mat][2I+K]
    const double* __restrict__ GxB = hot_Gx + (size_t)(m0 + 1) * 1024;    // B-layout images [mat][2K+J]

    // The drive's images stay in registers for every interval of this workgroup: they are the bulk of the L2 traffic
    // (16 KB per wave), and with one interval per workgroup the kernel was L2-bandwidth-bound in its load phase.
    v4d GkA[4], GkB[4];
    {
        const int kmat = drive0 ? w0 + 1 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            GkA[q] = g_tile(GxA + (size_t)kmat * 1024, q, lane0);
            if constexpr (!ANTI) GkB[q] = g_tile(GxB + (size_t)kmat * 1024, q, lane0);
        }
        if constexpr (ANTI) {   // B tile 2K+J = -(A tile 2J+K)
            GkB[0] = -GkA[0]; GkB[1] = -GkA[2]; GkB[2] = -GkA[1]; GkB[3] = -GkA[3];
        }
    }
    touch.consume();
    const int n_wg = (hot_n_int + per_wg - 1) / per_wg;
    const int b0 = qc_xcd_remap((int)blockIdx.x, n_wg) * per_wg;

    for (int it = 0; it < per_wg; ++it) {
        const int b = b0 + it;
        if (b >= hot_n_int) break;
        // An opaque copy of the lane index per interval: what derives from it (LDS and store offsets) is recomputed instead of
        // being hoisted out of the interval loop and held through the products (251 -> 238 registers).
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        int w = w0;                                   // (and of the wave index: what derives from it lives in scalar registers)
        asm volatile("" : "+s"(w));
        const int g = lane >> 4, j = lane & 15;
        double* __restrict__ scr = TS + w * (16 * 17);
        // The same for the parameter block: its fields are scalar loads from the kernel-argument segment; hoisted out of the loop
        // they outnumber the scalar registers and are spilled to vector-register lanes (282 v_readlane / 163 v_writelane per
        // interval, vector instructions all).  Read through an opaque pointer they are re-read where used (scalar-cache hits).
        typedef const __attribute__((address_space(4))) QcParams* kparams_t;
        // (the parameter block follows the preloaded arguments: 3 pointers + 8 ints = 56 bytes, 8-byte aligned)
        constexpr int kParamsOffset = 3 * 8 + 8 * 4;
        static_assert(kParamsOffset % 8 == 0, "QcParams is 8-byte aligned in the kernel-argument segment");
        kparams_t Pq = (kparams_t)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + kParamsOffset);
        asm volatile("" : "+s"(Pq));
        const auto& P = *Pq;
        int m = hot_m;                                // (an opaque copy: with it the drive-count conditions -- sixteen 64-bit masks when hoisted)
        asm volatile("" : "+s"(m));
        const bool drive = w < m;
        const double c1 = P.c[1], c2 = P.c[2];
        const double* __restrict__ z0 = hot_Zt + (long long)b * hot_zdim;
        const double* __restrict__ z1 = z0 + hot_zdim;
        const double* __restrict__ mu = hot_mu0 + (long long)b * hot_f_stride;
        double* __restrict__ Hb = H + (size_t)b * P.H_stride + P.H_off;
        // (the fixed timestep as an opaque value: qc_mfma_common.h -- as a second load the compiler merges the two arms into one
        //  FLAT load, fenced by vmcnt(0))
        const double h = ft ? z0[hot_off_dt] : opaque_scalar(P.dt_fixed);   // requested with the other loads; first used behind the barrier
        // (Timestep and amplitudes by one vector load read out with v_readlane, as in qc_mfma32_kernels.hip, where the scalar
        //  loads each cost a round trip in front of the next batch of requests: 21.76 against 21.43 us here, same run.  The eight
        //  waves of a workgroup read the same words: one scalar-cache miss, seven hits.)
        QC_STAMP_DECL;
        QC_STAMP(P, b, lane, 0);

        // ---- phase 0: wave w assembles half (w & 1) of A-layout tile (w >> 1) of G; waves 4-7 fetch M, U_t, U_t+1 ----
        // (every load unconditional and issued before the first use: a load behind a branch on m is not hoisted.
        //  Requesting these inputs one interval ahead was tried twice.  Round 1: the registers it holds across the products spill,
        //  and a spill reload queued behind the interval's stores costs more than the 1.2 us of load latency it hides.  Round 2,
        //  with room in the register file: double-buffered LDS blocks, the next interval's loads requested at the start of
        //  phase 1 and written to LDS at its end.  The second interval of a workgroup got 1.6 us shorter (no phase 0), the
        //  first 0.9 us longer (eight waves' staging loads in the compute unit's vector-memory issue path next to the products)
        //  plus the staging of the first interval in front of the loop: 23.6 - 24.8 us against 23.25 us at two intervals per
        //  workgroup.  It would pay from about four intervals per workgroup (T > 1000 at 4 qubits).)
        const bool dfast = ft && P.n_deriv <= 2 && P.ddim_i[0] <= 64 && P.ddim_i[1] <= 64;
        double mud[2] = {0.0, 0.0};
        {
            const v2d* __restrict__ ab = reinterpret_cast<const v2d*>(GxA) + (w >> 1) * 128 + (w & 1) * 64 + lane;
            v2d img[kHMax32 + 1];
            double ak[kHMax32];
#pragma unroll
            for (int u = 0; u <= kHMax32; ++u) img[u] = ab[(size_t)(u <= m ? u : 0) * 512];
#pragma unroll
            for (int u = 0; u < kHMax32; ++u) ak[u] = z0[hot_off_a + (u < m ? u : 0)];
            if (w >= 4) {
                // K < 16 kets: tile columns >= nc re-read column 0; the multipliers there are zeroed (the scalar blocks sum over
                // whole tiles) and nothing of them is stored.  The kernel is MFMA-bound: run-time masks cost nothing here.
                // Systems with 9 .. 15 levels: nr = 2N < 32 rows per column, zero-padded to the 2 x 2 tiles.
                const int I = w & 1;
                const int nr = FULL ? 32 : P.n, cb = ((FULL || j < P.nc) ? j : 0) * nr;
                auto ld4 = [&](const double* base) {
                    if constexpr (FULL) return load_col16_T(base + cb + 16 * I, g);   // 2 x 16 bytes per lane, transposed in registers (qc_mfma_common.h)
                    v4d v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int row = 16 * I + 4 * r + g; v[r] = (FULL || row < nr) ? base[cb + row] : 0.0; }
                    return v;
                };
                if (w < 6) {
                    const v4d mraw = ld4(mu);
                    put_tile(ML, I, lane, (FULL || j < P.nc) ? mraw : v4d{0.0, 0.0, 0.0, 0.0});
                } else {
                    const v4d u0 = ld4(z0 + hot_off_U);
                    const v4d u1 = ld4(z1 + hot_off_U);
                    put_tile(SL, I, lane, u1 + u0);
                    put_tile(DL, I, lane, u1 - u0);
                }
            }
            if (w == 5 && dfast) {   // derivative integrators: d2/d(dx_i) dh = -mu_i, a plain copy.  Only REQUESTED here (a load issued
                                     // after the stores of phase 2 would wait for all of them); written behind the barrier.  (The
                                     // general form, qc_hess_tail, is a loop of load -> store round trips: run here it held wave 5
                                     // -- and with it barrier 1 -- for two of them.)
#pragma unroll
                for (int d = 0; d < 2; ++d) mud[d] = mu[P.drow[d] + (lane < P.ddim_i[d] ? lane : 0)];   // unused slots: zero dims, in bounds
            }
            v2d Gh = img[0];
#pragma unroll
            for (int u = 0; u < kHMax32; ++u) Gh += (u < m ? ak[u] : 0.0) * img[u + 1];
            reinterpret_cast<v2d*>(GL)[(w >> 1) * 128 + (w & 1) * 64 + lane] = Gh;
        }
        QC_STAMP(P, b, lane, 1);      // loads arrived, G half tile published
        __syncthreads();
        const double hc1 = h * c1, hc2 = h * h * c2, c2h2 = 2.0 * c2 * h;
        QC_STAMP(P, b, lane, 2);

        // ---- phase 1 ---------------------------------------------------------------------------------------------
        if (w == 5) {   // the tail of the interval's block: derivative-integrator entries and the alignment padding
            if (dfast) {
                int o = P.ho_d;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    if (lane < P.ddim_i[d]) Hb[o + lane] = -mud[d];
                    o += P.ddim_i[d];
                }
                for (int i = lane; i < P.h_pad; i += 64) Hb[P.hess_nnz + i] = 0.0;
            } else {
                qc_hess_tail(Pk, mu, Hb, lane, 64);
            }
        }
        const v4d Mt[2] = {g_tile(ML, 0, lane), g_tile(ML, 1, lane)};
        const v4d D[2] = {g_tile(DL, 0, lane), g_tile(DL, 1, lane)};
        if (w < 2) {          // B-layout tiles (K, I) of G by identity products, then M1[I] = sum_K (G[K][I])^T M[K]
            const v4d gb[2] = {lds_transpose16(scr, g_tile(GL, w, lane), g, j), lds_transpose16(scr, g_tile(GL, 2 + w, lane), g, j)};
            put_tile(GL, 4 + w, lane, gb[0]);
            put_tile(GL, 4 + 2 + w, lane, gb[1]);
            v4d a1[1] = {gb[0]}, a2[1] = {gb[1]}, b1[1] = {Mt[0]}, b2[1] = {Mt[1]}, d[1];
            mm16x2_multi<1>(a1, b1, a2, b2, d);
            put_tile(M1L, w, lane, d[0]);
        } else if (w < 4) {   // (G D)[I] = sum_K G[I][K] D[K]
            const int I = w - 2;
            v4d a0[1] = {g_tile(GL, 2 * I, lane)}, a1[1] = {g_tile(GL, 2 * I + 1, lane)}, b0[1] = {D[0]}, b1[1] = {D[1]}, d[1];
            mm16x2_multi<1>(a0, b0, a1, b1, d);
            put_tile(GDL, I, lane, d[0]);
            put_tile(WLL, I, lane, (-c1) * g_tile(SL, I, lane) + c2h2 * d[0]);
        }
        v4d Nk[2], Vk[2], NkT[2];
        if (drive) {
            // N_k[I] = sum_K (G_k[K][I])^T M[K];  V_k[I] = sum_K G_k[I][K] D[K];  N_k^T[J] = (N_k[J])^T through LDS
            v4d a0[4] = {GkB[0], GkB[1], GkA[0], GkA[2]};
            v4d b0[4] = {Mt[0], Mt[0], D[0], D[0]};
            v4d a1[4] = {GkB[2], GkB[3], GkA[1], GkA[3]};
            v4d b1[4] = {Mt[1], Mt[1], D[1], D[1]};
            v4d d[4];
            mm16x2_multi<4>(a0, b0, a1, b1, d);
            Nk[0] = d[0]; Nk[1] = d[1]; Vk[0] = d[2]; Vk[1] = d[3];
            NkT[0] = lds_transpose16(scr, Nk[0], g, j);
            NkT[1] = lds_transpose16(scr, Nk[1], g, j);
            double* nv = NVL + w * 1024;
            put_tile(nv, 0, lane, Nk[0]);
            put_tile(nv, 1, lane, Nk[1]);
            put_tile(nv, 2, lane, Vk[0]);
            put_tile(nv, 3, lane, Vk[1]);
        }
        QC_STAMP(P, b, lane, 3);      // phase-1 products done
        __syncthreads();
        QC_STAMP(P, b, lane, 4);

        // ---- phase 2 ---------------------------------------------------------------------------------------------
        const v4d M1a = g_tile(M1L, 0, lane), M1b = g_tile(M1L, 1, lane);
        if (drive) {
            // X_k^T[J] = sum_K M1[K]^T G_k[K][J] + sum_K N_k[K]^T G[K][J], then the two matrix blocks of the drive
            auto matrix_blocks = [&]() {
                v4d a0[2] = {M1a, M1a}, b0[2] = {GkB[0], GkB[1]}, a1[2] = {M1b, M1b}, b1[2] = {GkB[2], GkB[3]}, x0[2];
                mm16x2_multi<2>(a0, b0, a1, b1, x0);
                v4d a2[2] = {Nk[0], Nk[0]}, b2[2] = {g_tile(GL, 4, lane), g_tile(GL, 5, lane)}, a3[2] = {Nk[1], Nk[1]},
                    b3[2] = {g_tile(GL, 6, lane), g_tile(GL, 7, lane)}, x1[2];
                mm16x2_multi<2>(a2, b2, a3, b3, x1);
                double* pUa = Hb + P.ho_Ua + (size_t)w * (FULL ? 512 : P.s);
                double* paU = Hb + P.ho_aU + (size_t)w * (FULL ? 512 : P.s);
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    const v4d lin = (-hc1) * NkT[J], q = hc2 * (x0[J] + x1[J]);
                    store_T<FULL>(pUa, lin - q, J, g, j, P.nc, P.n);
                    store_T<FULL>(paU, lin + q, J, g, j, P.nc, P.n);
                }
            };
            // scalar blocks of this drive from registers + the partner's tiles in LDS: (a_k,h), (a_k,a_k), and the pairs
            // {k, (k+d) mod m}: d = 1 .. (m-1)/2 for every k, d = m/2 (m even) for k < m/2  ->  each unordered pair once
            auto scalar_blocks = [&]() {
                constexpr int kPairs = kHMax32 / 2;
                double pv[2 + kPairs];                        // per-lane partial sums: (a_k,h), (a_k,a_k), the pairs at distance 1 .. 4
                pv[0] = ft ? (dot4(Nk[0], g_tile(WLL, 0, lane)) + dot4(Nk[1], g_tile(WLL, 1, lane))) +
                                 c2h2 * (dot4(M1a, Vk[0]) + dot4(M1b, Vk[1]))
                           : 0.0;
                pv[1] = 2.0 * hc2 * (dot4(Nk[0], Vk[0]) + dot4(Nk[1], Vk[1]));
#pragma unroll
                for (int dd = 1; dd <= kPairs; ++dd) {
                    pv[1 + dd] = 0.0;
                    if (2 * dd < m || (2 * dd == m && w < dd)) {
                        const int k = w + dd < m ? w + dd : w + dd - m;
                        const double* nk = NVL + k * 1024;
                        const v4d n0 = g_tile(nk, 0, lane), n1 = g_tile(nk, 1, lane), v0 = g_tile(nk, 2, lane), v1 = g_tile(nk, 3, lane);
                        pv[1 + dd] = hc2 * ((dot4(Nk[0], v0) + dot4(Nk[1], v1)) + (dot4(n0, Vk[0]) + dot4(n1, Vk[1])));
                    }
                }
                // ONE reduction for the six values (one at a time each is a chain of dependent DPP / read-lane instructions
                // behind its own LDS wait); the same order of additions per value: bit-identical
                wave_sum_multi<2 + kPairs>(pv);
                if (lane == 0) {
                    if (ft) Hb[P.ho_ah + w] = pv[0];
                    Hb[P.ho_aa + w * (w + 1) / 2 + w] = pv[1];
#pragma unroll
                    for (int dd = 1; dd <= kPairs; ++dd) {
                        if (2 * dd < m || (2 * dd == m && w < dd)) {
                            const int k = w + dd < m ? w + dd : w + dd - m;
                            const int lo = k < w ? k : w, hi = k < w ? w : k;
                            Hb[P.ho_aa + hi * (hi + 1) / 2 + lo] = pv[1 + dd];
                        }
                    }
                }
            };
            auto scalar_blocks_serial = [&]() {
                if (ft) {
                    const double v = wave_sum((dot4(Nk[0], g_tile(WLL, 0, lane)) + dot4(Nk[1], g_tile(WLL, 1, lane))) +
                                              c2h2 * (dot4(M1a, Vk[0]) + dot4(M1b, Vk[1])));
                    if (lane == 0) Hb[P.ho_ah + w] = v;
                }
                {
                    const double v = wave_sum(2.0 * hc2 * (dot4(Nk[0], Vk[0]) + dot4(Nk[1], Vk[1])));
                    if (lane == 0) Hb[P.ho_aa + w * (w + 1) / 2 + w] = v;
                }
#pragma unroll
                for (int dd = 1; dd <= kHMax32 / 2; ++dd) {
                    if (2 * dd < m || (2 * dd == m && w < dd)) {
                        const int k = w + dd < m ? w + dd : w + dd - m;
                        const double* nk = NVL + k * 1024;
                        const v4d n0 = g_tile(nk, 0, lane), n1 = g_tile(nk, 1, lane), v0 = g_tile(nk, 2, lane), v1 = g_tile(nk, 3, lane);
                        const double v = wave_sum(hc2 * ((dot4(Nk[0], v0) + dot4(Nk[1], v1)) + (dot4(n0, Vk[0]) + dot4(n1, Vk[1]))));
                        const int lo = k < w ? k : w, hi = k < w ? w : k;
                        if (lane == 0) Hb[P.ho_aa + hi * (hi + 1) / 2 + lo] = v;
                    }
                }
            };
            // waves w and w + 4 share a SIMD: one runs its MFMA chain while the other does LDS/VALU work
            // (the scalar blocks with ONE batched reduction behind the matrix blocks: 1.4 instead of 2.3 us; in front of them,
            //  next to the other wave's MFMAs: 3.0 instead of 2.5 us -- measured both ways in one run)
            if (w < 4) {
                matrix_blocks();
                QC_STAMP(P, b, lane, 5);
                scalar_blocks();
            } else {
                scalar_blocks_serial();
                QC_STAMP(P, b, lane, 5);
                matrix_blocks();
            }
        }
        QC_STAMP(P, b, lane, 6);      // drive blocks and scalars done
        if (ft) {
            // (waves 0 - 2: with the batched reduction they are through their drive blocks 0.9 us before waves 4 - 7)
            if (w == 0 || w == 1) {   // (U_t,h)^T and (h,U_t+1)^T, column block J
                const int J = w;
                const v4d gb0 = g_tile(GL, 4 + J, lane), gb1 = g_tile(GL, 4 + 2 + J, lane);
                v4d a0[1] = {M1a}, b0[1] = {gb0}, a1[1] = {M1b}, b1[1] = {gb1}, d[1];
                mm16x2_multi<1>(a0, b0, a1, b1, d);      // M2^T[J] = sum_K M1[K]^T G[K][J]
                const v4d m1t = lds_transpose16(scr, J == 0 ? M1a : M1b, g, j);   // M1^T[J] = (M1[J])^T
                store_T<FULL>(Hb + P.ho_Uh, -(c1 * m1t + c2h2 * d[0]), J, g, j, P.nc, P.n);
                store_T<FULL>(Hb + P.ho_hU, (-c1) * m1t + c2h2 * d[0], J, g, j, P.nc, P.n);
            } else if (w == 2) {      // (h, h)
                const double sum = wave_sum(dot4(M1a, g_tile(GDL, 0, lane)) + dot4(M1b, g_tile(GDL, 1, lane)));
                if (lane == 0) Hb[P.ho_hh] = 2.0 * c2 * sum;
            }
        }
        if constexpr (DIAG) {
            QC_STAMP(P, b, lane, 7);
            if (P.stamps != nullptr && lane == 0 && (w == 0 || w == 5)) {   // slots 0-7: wave 0, 8-15: wave 5
#pragma unroll
                for (int k_ = 0; k_ < 8; ++k_) P.stamps[(size_t)b * 16 + (w == 0 ? 0 : 8) + k_] = qc_ts_[k_];
            }
        }
        __syncthreads();   // the LDS blocks are rewritten by the workgroup's next interval
    }
}

}  // namespace

bool qc_mfma32_hess_supported(const QcParams& P) {
    return P.integrator == QC_PADE && P.p == 2 && P.n > 16 && P.n <= 32 && P.nc <= 16 && P.m <= kHMax32 && P.Gx != nullptr;
}

hipError_t qc_launch_mfma32_hess(const QcParams& P, const double* dZ, const double* dMu, double* dH, hipStream_t st) {
    // one workgroup per CU (200 VGPRs, 121 KB LDS); each takes a contiguous run of intervals and keeps its drive images
    const int per_wg = (P.n_int + kHCUs - 1) / kHCUs;
    const int grid = (P.n_int + per_wg - 1) / per_wg;
    if (P.stamps != nullptr && P.antisym && P.n == 32 && P.nc == 16) hipLaunchKernelGGL((qc_mfma32_pade4_hess_kernel<true, true, true>), dim3(grid), dim3(kHThreads32), 0, st, P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu + P.t_begin * P.F_stride + P.F_off, P.n_int, per_wg, P.zdim, P.m, P.off_a, P.off_dt, P.off_U, (int)P.F_stride, P, dH);
    else if (P.stamps != nullptr) hipLaunchKernelGGL((qc_mfma32_pade4_hess_kernel<true, false>), dim3(grid), dim3(kHThreads32), 0, st, P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu + P.t_begin * P.F_stride + P.F_off, P.n_int, per_wg, P.zdim, P.m, P.off_a, P.off_dt, P.off_U, (int)P.F_stride, P, dH);
    else if (P.antisym && P.n == 32 && P.nc == 16) hipLaunchKernelGGL((qc_mfma32_pade4_hess_kernel<false, true, true>), dim3(grid), dim3(kHThreads32), 0, st, P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu + P.t_begin * P.F_stride + P.F_off, P.n_int, per_wg, P.zdim, P.m, P.off_a, P.off_dt, P.off_U, (int)P.F_stride, P, dH);
    else if (P.antisym) hipLaunchKernelGGL((qc_mfma32_pade4_hess_kernel<false, true>), dim3(grid), dim3(kHThreads32), 0, st, P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu + P.t_begin * P.F_stride + P.F_off, P.n_int, per_wg, P.zdim, P.m, P.off_a, P.off_dt, P.off_U, (int)P.F_stride, P, dH);
    else hipLaunchKernelGGL((qc_mfma32_pade4_hess_kernel<false, false>), dim3(grid), dim3(kHThreads32), 0, st, P.Gx, dZ + P.t_begin * (long long)P.zdim, dMu + P.t_begin * P.F_stride + P.F_off, P.n_int, per_wg, P.zdim, P.m, P.off_a, P.off_dt, P.off_U, (int)P.F_stride, P, dH);
    return hipGetLastError();
}
