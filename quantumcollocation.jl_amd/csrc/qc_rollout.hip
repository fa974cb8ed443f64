// Rollouts: x_{t+1} = exp(dt_t G(a_t)) x_t for every knot of a trajectory, from a given x_0
// (`unitary_rollout`, `rollout`, `open_rollout`; reference call sites trajectory_initialization.jl:426,493,547 and
// `unitary_rollout_fidelity`, unitary_smooth_pulse_problem.jl:218).  Unlike the constraint path this is a sequential
// recurrence; it is evaluated as a three-level scan of matrix products:
//   1. propagators   one workgroup per interval: E_t = exp(dt_t G(a_t)) by a scaled Taylor polynomial (degree 12 at
//                    ||Y||_1 <= 1/4) and repeated squaring, all in LDS  ->  scratch, (T-1) n^2 doubles
//   2. chunk totals  one workgroup per chunk of ~sqrt(T) intervals: Q_c = E_last ... E_first
//   3. chunk starts  one workgroup: x at the first knot of every chunk, S_{c+1} = Q_c S_c
//   4. states        one workgroup per chunk: x_{t+1} = E_t x_t from S_c, written to the output
// so the dependent chain is ~3 sqrt(T) small products instead of T.  FP64 VALU/LDS throughout (n = 2N <= 64); this is a
// few MFLOP per trajectory and runs once per solve or callback, not per Ipopt iteration.
#include <string>
#include <type_traits>

#include "qc_internal.h"

namespace {

constexpr int kRT = 256;
constexpr int kRDeg = 12;

struct RollParams {
    int n, nc, m, zdim, off_a, off_dt, n_int, chunk, n_chunks;
    double dt_fixed;
    const double* G;
};

// C (n x p) = A (n x n) * B (n x p), all column-major in LDS; C must not alias A or B.
// NT = n at compile time (0: any n): with the inner loop unrolled its 2 n LDS reads go out together; with a run-time trip count every
// fused multiply-add waited for its own pair of reads -- 16 dependent LDS round trips per product at 2N = 16, 0.9 us of every 1.3 us step
// of the chains below (profiles/r06_variants.txt: 122 -> 55 us for a T = 1000 rollout).  Same operations in the same order: same bits.
template <int NT>
__device__ __forceinline__ void mm_lds(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ B, int n_rt, int p, int tid) {
    const int n = NT ? NT : n_rt;
    for (int idx = tid; idx < n * p; idx += kRT) {
        const int r = idx % n, c = idx / n;
        double acc = 0.0;
        if constexpr (NT != 0) {
#pragma unroll
            for (int q = 0; q < NT; ++q) acc = fma(A[r + NT * q], B[q + NT * c], acc);
        } else {
            for (int q = 0; q < n; ++q) acc = fma(A[r + n * q], B[q + n * c], acc);
        }
        C[idx] = acc;
    }
}

// The chains of kernels 2 - 4 read one matrix per step from global memory.  Requested inside the step, every step paid an L2 round trip
// between two barriers (~1.5 us: 122 us for a T = 1000 rollout at 2N = 16, of which ~75 were these waits); here the NEXT matrix is
// requested into registers before the current product and written to LDS behind it.  Same arithmetic in the same order: same bits.
constexpr int kRPre = 16;           // matrix entries per thread: n <= 64 -> n^2 / kRT <= 16
struct RollPrefetch {
    double v[kRPre];
    __device__ __forceinline__ void fetch(const double* __restrict__ src, int n2, int tid) {
#pragma unroll
        for (int i = 0; i < kRPre; ++i) { const int idx = tid + i * kRT; v[i] = idx < n2 ? src[idx] : 0.0; }
    }
    __device__ __forceinline__ void commit(double* __restrict__ dst, int n2, int tid) const {
#pragma unroll
        for (int i = 0; i < kRPre; ++i) { const int idx = tid + i * kRT; if (idx < n2) dst[idx] = v[i]; }
    }
};

template <int NT>
__global__ __launch_bounds__(kRT) void qc_rollout_prop_kernel(RollParams R, const double* __restrict__ Z, double* __restrict__ E) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[kRT / 64];
    const int tid = threadIdx.x, n = NT ? NT : R.n, n2 = n * n;
    double* Y = sm;
    double* A0 = sm + n2;
    double* A1 = sm + 2 * n2;
    double* Em = sm + 3 * n2;
    const long long t = blockIdx.x;
    const double* z = Z + t * (long long)R.zdim;
    const double h = R.off_dt >= 0 ? z[R.off_dt] : R.dt_fixed;
    for (int idx = tid; idx < n2; idx += kRT) {
        double g = R.G[idx];
        for (int k = 0; k < R.m; ++k) g = fma(z[R.off_a + k], R.G[(size_t)(k + 1) * n2 + idx], g);
        Y[idx] = h * g;
    }
    __syncthreads();
    // ||h G||_1 = largest column sum (thread c sums column c)
    double cs = 0.0;
    if (tid < n) for (int r = 0; r < n; ++r) cs += fabs(Y[r + n * tid]);
    bool bad = !(cs == cs) || cs > 1e300;
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(cs, off, 64);
        cs = fmax(cs, o);
        bad = bad || __shfl_xor((int)bad, off, 64);
    }
    if ((tid & 63) == 0) red[tid >> 6] = bad ? -1.0 : cs;
    __syncthreads();
    double nrm = 0.0;
    bool anybad = false;
    for (int w = 0; w < kRT / 64; ++w) { if (red[w] < 0.0) anybad = true; nrm = fmax(nrm, red[w]); }
    int sq = 0;
    if (!anybad && nrm > 0.25) {
        int e;
        (void)frexp(nrm / 0.25, &e);
        sq = e;
        if (ldexp(0.25, e - 1) >= nrm) sq = e - 1;
        sq = sq < 0 ? 0 : (sq > 60 ? 60 : sq);
    }
    const double sc = ldexp(1.0, -sq);
    for (int idx = tid; idx < n2; idx += kRT) {
        Y[idx] *= sc;
        const double id = (idx % n == idx / n) ? 1.0 : 0.0;
        A0[idx] = id;
        Em[idx] = id;
    }
    __syncthreads();
    for (int k = 1; k <= kRDeg; ++k) {
        const double* Ap = (k & 1) ? A0 : A1;
        double* An = (k & 1) ? A1 : A0;
        const double inv = 1.0 / (double)k;
        for (int idx = tid; idx < n2; idx += kRT) {
            const int r = idx % n, c = idx / n;
            double acc = 0.0;
            if constexpr (NT != 0) {
#pragma unroll
                for (int q = 0; q < NT; ++q) acc = fma(Ap[r + NT * q], Y[q + NT * c], acc);
            } else {
                for (int q = 0; q < n; ++q) acc = fma(Ap[r + n * q], Y[q + n * c], acc);
            }
            acc *= inv;
            An[idx] = acc;
            Em[idx] += acc;
        }
        __syncthreads();
    }
    for (int q = 0; q < sq; ++q) {
        mm_lds<NT>(A0, Em, Em, n, n, tid);
        __syncthreads();
        for (int idx = tid; idx < n2; idx += kRT) Em[idx] = A0[idx];
        __syncthreads();
    }
    double* Eo = E + (size_t)t * n2;
    for (int idx = tid; idx < n2; idx += kRT) Eo[idx] = Em[idx];
}

// Q_c = E_{last} ... E_{first} of chunk c
template <int NT>
__global__ __launch_bounds__(kRT) void qc_rollout_total_kernel(RollParams R, const double* __restrict__ E, double* __restrict__ Q) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, n = NT ? NT : R.n, n2 = n * n;
    double* Qa = sm;
    double* Qb = sm + n2;
    double* Et = sm + 2 * n2;
    const int c = blockIdx.x;
    const int t0 = c * R.chunk, t1 = min(R.n_int, t0 + R.chunk);
    for (int idx = tid; idx < n2; idx += kRT) Qa[idx] = E[(size_t)t0 * n2 + idx];
    __syncthreads();
    double* cur = Qa;
    double* nxt = Qb;
    RollPrefetch pre;
    if (t0 + 1 < t1) pre.fetch(E + (size_t)(t0 + 1) * n2, n2, tid);
    for (int t = t0 + 1; t < t1; ++t) {
        pre.commit(Et, n2, tid);
        __syncthreads();
        if (t + 1 < t1) pre.fetch(E + (size_t)(t + 1) * n2, n2, tid);      // in flight during the product
        mm_lds<NT>(nxt, Et, cur, n, n, tid);
        __syncthreads();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int idx = tid; idx < n2; idx += kRT) Q[(size_t)c * n2 + idx] = cur[idx];
}

// S_0 = init, S_{c+1} = Q_c S_c
template <int NT>
__global__ __launch_bounds__(kRT) void qc_rollout_starts_kernel(RollParams R, const double* __restrict__ Q, const double* __restrict__ init,
                                                                double* __restrict__ S) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, n = NT ? NT : R.n, n2 = n * n, ns = n * R.nc;
    double* Xa = sm;
    double* Xb = sm + ns;
    double* Qt = sm + 2 * ns;
    for (int idx = tid; idx < ns; idx += kRT) { Xa[idx] = init[idx]; S[idx] = init[idx]; }
    __syncthreads();
    double* cur = Xa;
    double* nxt = Xb;
    RollPrefetch pre;
    if (R.n_chunks > 1) pre.fetch(Q, n2, tid);
    for (int c = 0; c + 1 < R.n_chunks; ++c) {
        pre.commit(Qt, n2, tid);
        __syncthreads();
        if (c + 2 < R.n_chunks) pre.fetch(Q + (size_t)(c + 1) * n2, n2, tid);
        mm_lds<NT>(nxt, Qt, cur, n, R.nc, tid);
        __syncthreads();
        for (int idx = tid; idx < ns; idx += kRT) S[(size_t)(c + 1) * ns + idx] = nxt[idx];
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
}

// x at every knot of chunk c from its start state; out is (n nc) x T, one column per knot
template <int NT>
__global__ __launch_bounds__(kRT) void qc_rollout_states_kernel(RollParams R, const double* __restrict__ E, const double* __restrict__ S,
                                                                double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, n = NT ? NT : R.n, n2 = n * n, ns = n * R.nc;
    double* Xa = sm;
    double* Xb = sm + ns;
    double* Et = sm + 2 * ns;
    const int c = blockIdx.x;
    const int t0 = c * R.chunk, t1 = min(R.n_int, t0 + R.chunk);
    // (the first knot of chunk c > 0 is the last knot the previous chunk steps to: ONE workgroup writes that column -- the two values
    //  differ in the last bit, S_c being a product in another order, and two writers made the result depend on which came last)
    for (int idx = tid; idx < ns; idx += kRT) {
        const double v = S[(size_t)c * ns + idx];
        Xa[idx] = v;
        if (c == 0) out[idx] = v;
    }
    __syncthreads();
    double* cur = Xa;
    double* nxt = Xb;
    RollPrefetch pre;
    if (t0 < t1) pre.fetch(E + (size_t)t0 * n2, n2, tid);
    for (int t = t0; t < t1; ++t) {
        pre.commit(Et, n2, tid);
        __syncthreads();
        if (t + 1 < t1) pre.fetch(E + (size_t)(t + 1) * n2, n2, tid);
        mm_lds<NT>(nxt, Et, cur, n, R.nc, tid);
        __syncthreads();
        for (int idx = tid; idx < ns; idx += kRT) out[(size_t)(t + 1) * ns + idx] = nxt[idx];
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
}

// Gather of the non-replicated part of the Jacobian values for the host-buffer entry points (qc_host.cpp, "compact
// transfer"): per interval [first -F block | first B block (Pade) | everything from the drive columns on].
__global__ __launch_bounds__(256) void qc_pack_jac_kernel(const double* __restrict__ J, double* __restrict__ C, int n_int, int jac_nnz,
                                                          int comp_len, int n2, int jo_F, int jo_B, int head2, int tail_src) {
    for (int b = blockIdx.x; b < n_int; b += gridDim.x) {
        const double* __restrict__ src = J + (size_t)b * jac_nnz;
        double* __restrict__ dst = C + (size_t)b * comp_len;
        for (int e = threadIdx.x; e < comp_len; e += 256) {
            const int o = e < n2 ? jo_F + e : (e < head2 ? jo_B + (e - n2) : tail_src + (e - head2));
            dst[e] = src[o];
        }
    }
}

}  // namespace

hipError_t qc_launch_pack_jac(const double* dJ, double* dJc, int n_int, int jac_nnz, int comp_len, int n2, int jo_F, int jo_B, int head2,
                              int tail_src, hipStream_t st) {
    const int grid = n_int < 4096 ? n_int : 4096;
    hipLaunchKernelGGL(qc_pack_jac_kernel, dim3(grid), dim3(256), 0, st, dJ, dJc, n_int, jac_nnz, comp_len, n2, jo_F, jo_B, head2, tail_src);
    return hipGetLastError();
}

bool qc_rollout_supported(const QcParams& P) { return P.n <= 64; }

void qc_rollout_scratch(const QcParams& P, long long T, size_t* nE, size_t* nQ, size_t* nS, int* chunk, int* n_chunks) {
    const long long n_int = T - 1;
    int ch = 1;
    while ((long long)ch * ch < n_int) ++ch;
    const int nch = (int)((n_int + ch - 1) / ch);
    *chunk = ch;
    *n_chunks = nch;
    *nE = (size_t)n_int * P.n * P.n;
    *nQ = (size_t)nch * P.n * P.n;
    *nS = (size_t)nch * P.n * P.nc;
}

hipError_t qc_launch_rollout(const QcParams& P, long long T, const double* dZ, const double* dinit, double* dout, double* dE, double* dQ,
                             double* dS, hipStream_t st) {
    RollParams R;
    R.n = P.n; R.nc = P.nc; R.m = P.m; R.zdim = P.zdim; R.off_a = P.off_a; R.off_dt = P.off_dt; R.dt_fixed = P.dt_fixed; R.G = P.G;
    R.n_int = (int)(T - 1);
    size_t nE, nQ, nS;
    qc_rollout_scratch(P, T, &nE, &nQ, &nS, &R.chunk, &R.n_chunks);
    const size_t n2 = (size_t)P.n * P.n, ns = (size_t)P.n * P.nc;
    const size_t lds_prop = 4 * n2 * 8, lds_tot = 3 * n2 * 8, lds_st = (2 * ns + n2) * 8;
    // the matrix size at compile time for the common ones (1 - 4 qubits: 2N = 4, 8, 16, 32): the products' inner loops unroll
    auto run = [&](auto nt) -> hipError_t {
        constexpr int NT = decltype(nt)::value;
        hipError_t e;
        if (lds_prop > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qc_rollout_prop_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prop);
            if (e != hipSuccess) return e;
        }
        if (lds_tot > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qc_rollout_total_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tot);
            if (e != hipSuccess) return e;
        }
        if (lds_st > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qc_rollout_starts_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_st);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qc_rollout_states_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_st);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(qc_rollout_prop_kernel<NT>, dim3(R.n_int), dim3(kRT), lds_prop, st, R, dZ, dE);
        if (R.n_chunks > 1) hipLaunchKernelGGL(qc_rollout_total_kernel<NT>, dim3(R.n_chunks - 1), dim3(kRT), lds_tot, st, R, dE, dQ);
        hipLaunchKernelGGL(qc_rollout_starts_kernel<NT>, dim3(1), dim3(kRT), lds_st, st, R, dQ, dinit, dS);
        hipLaunchKernelGGL(qc_rollout_states_kernel<NT>, dim3(R.n_chunks), dim3(kRT), lds_st, st, R, dE, dS, dout);
        return hipGetLastError();
    };
    switch (P.n) {
        case 4: return run(std::integral_constant<int, 4>());
        case 8: return run(std::integral_constant<int, 8>());
        case 16: return run(std::integral_constant<int, 16>());
        case 32: return run(std::integral_constant<int, 32>());
        default: return run(std::integral_constant<int, 0>());
    }
}
