// Shared device helpers of the f64-MFMA kernels (v_mfma_f64_16x16x4_f64 lane maps; see the header
// comment of qc_mfma_kernels.hip and tests/hip/mfma_f64_probe.hip).
#pragma once
#include "qc_internal.h"

namespace qc_mfma {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ inline double swap8(double x) {  // exchange the two 8-column halves of a 16-column tile
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);  // row_ror:8
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline v4d swap8(v4d x) { return v4d{swap8(x[0]), swap8(x[1]), swap8(x[2]), swap8(x[3])}; }

// D = A * B (16x16x16): A in A-layout regs, B in B-layout regs.  Two accumulators halve the
// dependent-MFMA chain (a dependent f64 MFMA issues every ~100 cycles, an independent one every 64).
__device__ inline v4d mm16(const v4d& a, const v4d& b) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], z, 0, 0, 0);
    v4d acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], z, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
    return acc0 + acc1;
}


// NQ independent 16x16x16 products with their MFMAs interleaved round-robin: every accumulator chain has NQ-1
// other MFMAs between two of its own, so (for NQ >= 2) no MFMA waits on its predecessor, and no VALU touches
// a result until the whole batch is issued (the two-accumulator mm16 above pays ~80 idle cycles of MFMA->VALU
// wait states after every product: the s_nop 15 / s_nop 2 pairs in the ISA).
template <int NQ>
__device__ __forceinline__ void mm16_multi(const v4d (&a)[NQ], const v4d (&b)[NQ], v4d (&d)[NQ]) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NQ; ++q) d[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][0], b[q][0], z, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) d[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], b[q][kk], d[q], 0, 0, 0);
    }
}

// (A 16-byte-per-lane paired-lane store of transposed tiles was tried for the copy waves and removed: 12.4 vs 11.5 us.)

// D-layout(X) -> D-layout(X^T) of a 16 x 16 tile through a padded (17-double rows) per-wave LDS scratch: four 8-byte writes
// and reads per lane, conflict-free up to 2-way; replaces a transposing identity product (4 MFMAs = 256 cycles of the pipe).
// LDS operations of one wave execute in order, so no barrier is needed.
__device__ inline v4d lds_transpose16(double* __restrict__ scr, const v4d& x, int g, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(4 * r + g) * 17 + j] = x[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v4d y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = scr[j * 17 + 4 * r + g];
    __builtin_amdgcn_wave_barrier();
    return y;
}

// N tiles at once (scratch of N * 16 * 17 doubles): all writes, one wait, all reads -- one LDS round trip instead of N
template <int N>
__device__ __forceinline__ void lds_transpose16_multi(double* __restrict__ scr, const v4d (&x)[N], v4d (&y)[N], int g, int j) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[q * 272 + (4 * r + g) * 17 + j] = x[q][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < N; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) y[q][r] = scr[q * 272 + j * 17 + 4 * r + g];
    }
    __builtin_amdgcn_wave_barrier();
}

// The same in rounds of at most kRound tiles through a scratch of kRound tiles (less LDS; each round is one write / read trip).
template <int N, int kRound>
__device__ __forceinline__ void lds_transpose16_rounds(double* __restrict__ scr, const v4d (&x)[N], v4d (&y)[N], int g, int j) {
    if constexpr (N <= kRound) {
        lds_transpose16_multi<N>(scr, x, y, g, j);
    } else {
        v4d a[kRound], b[kRound], c[N - kRound], d[N - kRound];
#pragma unroll
        for (int q = 0; q < kRound; ++q) a[q] = x[q];
#pragma unroll
        for (int q = 0; q < N - kRound; ++q) c[q] = x[kRound + q];
        lds_transpose16_multi<kRound>(scr, a, b, g, j);
        lds_transpose16_rounds<N - kRound, kRound>(scr, c, d, g, j);
#pragma unroll
        for (int q = 0; q < kRound; ++q) y[q] = b[q];
#pragma unroll
        for (int q = 0; q < N - kRound; ++q) y[kRound + q] = d[q];
    }
}

// Drive amplitudes of a knot.  Written as `a_k = z0[off_a + k]` per drive, the compiler emits one SCALAR load per amplitude,
// each behind its own `s_waitcnt lgkmcnt(0)` (the offset is re-read from the kernel arguments under a branch on k < m): the
// m loads then return one after the other -- m dependent HBM round trips in front of the first product of every wave.  Here
// the whole control vector is fetched by ONE vector load (lane l holds a_min(l, m-1)) and amplitude k is broadcast from lane k.
__device__ inline double load_amp_lanes(const double* __restrict__ z0, int off_a, int m, int lane) {
    const int k = lane < m ? lane : (m > 0 ? m - 1 : 0);
    return m > 0 ? z0[off_a + k] : 0.0;
}
__device__ inline double bcast_lane(double v, int src_lane) {   // src_lane wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
// A 16-row column piece in the B-operand / C-D lane map -- lane (g, j) reg r = col[4 r + g], `col` the lane's column -- fetched as
// 32 CONTIGUOUS bytes per lane (rows 4g .. 4g+3: two 16-byte requests) instead of four 8-byte requests of stride 4, then
// transposed between the lane groups g and the registers r: v_permlane32_swap exchanges (g bit 1, r bit 1),
// v_permlane16_swap (g bit 0, r bit 0) (gfx950).  The compute unit has ONE vector-memory pipeline; with four waves per CU each
// requesting images and state tiles at kernel start, the load phase is bound by the number of requests, not by latency.
__device__ inline void swap32_f64(double& a, double& b) {   // a: [a.0 a.1 b.0 b.1], b: [a.2 a.3 b.2 b.3]  (16-lane rows)
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// x of lane l ^ 32, without the LDS crossbar (what __shfl_xor(x, 32) compiles to is two ds_bpermute_b32: an LDS round trip)
__device__ inline double xor32_f64(double x, int lane) {
    double a = x, b = x;
    swap32_f64(a, b);                     // a: [x.lo | x.lo], b: [x.hi | x.hi]  (32-lane halves)
    return lane < 32 ? b : a;
}
__device__ inline void swap16_rows_f64(double& a, double& b) {   // a: [a.0 b.0 a.2 b.2], b: [a.1 b.1 a.3 b.3]
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ inline v4d load_col16_T(const double* __restrict__ col, int g) {
    typedef double v2d_ __attribute__((ext_vector_type(2)));
    typedef v2d_ __attribute__((aligned(8))) v2d_u;                      // the trajectory vector is 8-byte aligned only
    const v2d_u* q = reinterpret_cast<const v2d_u*>(col + 4 * g);
    const v2d_ q0 = q[0], q1 = q[1];
    double x0 = q0[0], x1 = q0[1], x2 = q1[0], x3 = q1[1];                // x_r' = col[4 g + r']
    swap32_f64(x0, x2);
    swap32_f64(x1, x3);
    swap16_rows_f64(x0, x1);
    swap16_rows_f64(x2, x3);
    return v4d{x0, x1, x2, x3};                                           // x_r = col[4 r + g]
}

// A wave-uniform double (the timestep) through the VECTOR memory path as well: every lane loads the same address and keeps
// its own copy in a VGPR.  (A scalar load would share lgkmcnt with the LDS traffic and the kernel-argument loads; a broadcast
// by v_readlane right behind the load makes the compiler wait for it before it issues the next load.)
__device__ inline double load_uniform(const double* __restrict__ p) {
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));       // opaque: keeps the address in a VGPR
    return p[zero];
}

// The fixed timestep as a VALUE.  Written `ft ? z0[off_dt] : P.dt_fixed`, both arms are loads (global memory / kernel-argument
// segment) and the compiler merges them into ONE load through a selected pointer: a FLAT load, in front of which it waits
// for every outstanding memory operation of the wave -- vmcnt(0), which on gfx9 includes the stores of the previous interval
// of a persistent loop.  An opaque copy of the kernel argument keeps the arms apart.
__device__ inline double opaque_scalar(double x) {
    asm volatile("" : "+s"(x));
    return x;
}

// A generator image tile: [pair(2)][lane(64)][2] doubles at `tile`, always device (hipMalloc) memory.  The pointer comes out of
// the parameter block; in the batched launches that block itself lives in global memory, a pointer loaded from memory has no
// known address space, and every access through it would be a FLAT load (which the compiler fences with vmcnt(0) /
// lgkmcnt(0)).  The integer round trip states the address space.
__device__ inline v4d load_image_tile(const double* tile, int lane) {
    typedef const __attribute__((address_space(1))) v2d* gptr;
    const gptr p = (gptr)(unsigned long long)tile + lane;
    const v2d lo = p[0], hi = p[64];
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}

// Identity in B/D layout: lane (g, j) reg r = (4r + g == j)
__device__ inline v4d identity_B(int g, int j) {
    return v4d{(g == j) ? 1.0 : 0.0, (4 + g == j) ? 1.0 : 0.0, (8 + g == j) ? 1.0 : 0.0, (12 + g == j) ? 1.0 : 0.0};
}

// Derivative integrators x_{t+1} - x_t - h dx_t (reference unitary_smooth_pulse_problem.jl:15-16), generic
// form: any count, any dimension.  `skip_small` skips the integrators the register fast path served.
__device__ inline void deriv_rows_generic(const QcParams& P, const double* __restrict__ z0, const double* __restrict__ z1,
                                          double h, double* __restrict__ Fb, double* __restrict__ Jb, int lane, bool skip_small) {
    const bool ft = P.off_dt >= 0;
    int jo = P.jo_d;
    for (int d = 0; d < P.n_deriv; ++d) {
        const int dim = P.ddim_i[d], r0 = P.drow[d];
        if (!(skip_small && dim <= 64)) {
            for (int i = lane; i < dim; i += 64) {
                const double dx = z0[P.dx_off[d] + i];
                if (Fb) Fb[r0 + i] = z1[P.x_off[d] + i] - z0[P.x_off[d] + i] - h * dx;
                if (Jb) {
                    Jb[jo + i] = -1.0;
                    Jb[jo + dim + i] = 1.0;
                    Jb[jo + 2 * dim + i] = -h;
                    if (ft) Jb[jo + 3 * dim + i] = -dx;
                }
            }
        }
        jo += (ft ? 4 : 3) * dim;
    }
}

}  // namespace qc_mfma
