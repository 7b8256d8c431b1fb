// D3: alpha = (L L^T)^-1 R for a few right-hand sides (q <= 8), and the D5
// tail (row reductions over W = K(X*,X) L^-T).
//
// The skinny solves walk the 64-column diagonal blocks whose inverses potrf
// left in the workspace: per block one launch that (every workgroup,
// redundantly, from L2) forms x_b = inv(L_bb) z_b and then subtracts
// L[rows below, block] x_b from the running right-hand side.  Right-hand
// sides are kept "RHS-major" (q x n) so the updates are coalesced.
// HBM-read bound: n^2/2 * sizeof(T) bytes per direction.
#include "common.hpp"

namespace cimrgp {

namespace {

constexpr int SB   = 64;
constexpr int MAXQ = 8;
constexpr int LSI  = SB + 1;

// (n x q) row-major  <->  (q x n)
template <typename T>
__global__ void k_transpose_nq(const T* __restrict__ src, T* __restrict__ dst, int64_t n, int q, int to_qn)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * q) return;
    if (to_qn) { const int64_t c = e / n, i = e - c * n; dst[e] = src[i * q + c]; }
    else       { const int64_t i = e / q, c = e - i * q; dst[e] = src[c * n + i]; }
}

template <typename T>
static __device__ __forceinline__ void load_slab(const T* __restrict__ inv, T* s)
{
    for (int e = threadIdx.x; e < SB * SB; e += blockDim.x) s[(e >> 6) * LSI + (e & 63)] = inv[e];
}

// forward: z_b = inv(L_bb) w_b ; w[rows > block] -= L[rows, block] z_b
template <typename T>
__global__ __launch_bounds__(256)
void k_fwd_step(const T* __restrict__ L, int64_t ld, int n, const T* __restrict__ inv,
                T* __restrict__ work, T* __restrict__ out, int q, int b0, int sw, int rows_per_wg)
{
    __shared__ T sinv[SB * LSI];
    __shared__ T sw_in[MAXQ][SB];
    __shared__ T sx[MAXQ][SB];
    const int tid = threadIdx.x;
    load_slab(inv, sinv);
    for (int e = tid; e < q * SB; e += 256) {
        const int c = e >> 6, u = e & 63;
        sw_in[c][u] = (u < sw) ? work[(int64_t)c * n + b0 + u] : (T)0;
    }
    __syncthreads();
    {
        const int t = tid & 63;
        for (int c = tid >> 6; c < q; c += 4) {
            T s = (T)0;
            for (int u = 0; u <= t; ++u) s += sinv[t * LSI + u] * sw_in[c][u];
            sx[c][t] = s;
            if (blockIdx.x == 0 && t < sw) out[(int64_t)c * n + b0 + t] = s;
        }
    }
    __syncthreads();
    const int l16 = tid & 15, slot = tid >> 4;
    const int rbeg = b0 + sw + blockIdx.x * rows_per_wg;
    const int rend = min(n, rbeg + rows_per_wg);
    for (int r = rbeg + slot; r < rend; r += 16) {
        const T* lp = L + (int64_t)r * ld + b0 + l16 * 4;
        T lv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lv[e] = (l16 * 4 + e < sw) ? lp[e] : (T)0;
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) {
            if (c < q) {
                T s = (T)0;
#pragma unroll
                for (int e = 0; e < 4; ++e) s += lv[e] * sx[c][l16 * 4 + e];
                s += __shfl_xor(s, 8, 16);
                s += __shfl_xor(s, 4, 16);
                s += __shfl_xor(s, 2, 16);
                s += __shfl_xor(s, 1, 16);
                if (l16 == 0) work[(int64_t)c * n + r] -= s;
            }
        }
    }
}

// backward: a_b = inv(L_bb)^T w_b ; w[cols < block] -= L[block, cols]^T a_b
template <typename T>
__global__ __launch_bounds__(256)
void k_bwd_step(const T* __restrict__ L, int64_t ld, int n, const T* __restrict__ inv,
                T* __restrict__ work, T* __restrict__ out, int q, int b0, int sw)
{
    __shared__ T sinv[SB * LSI];
    __shared__ T sw_in[MAXQ][SB];
    __shared__ T sx[MAXQ][SB];
    const int tid = threadIdx.x;
    load_slab(inv, sinv);
    for (int e = tid; e < q * SB; e += 256) {
        const int c = e >> 6, u = e & 63;
        sw_in[c][u] = (u < sw) ? work[(int64_t)c * n + b0 + u] : (T)0;
    }
    __syncthreads();
    {
        const int t = tid & 63;
        for (int c = tid >> 6; c < q; c += 4) {
            T s = (T)0;
            for (int u = t; u < sw; ++u) s += sinv[u * LSI + t] * sw_in[c][u];
            sx[c][t] = s;
            if (blockIdx.x == 0 && t < sw) out[(int64_t)c * n + b0 + t] = s;
        }
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j < b0) {
        T acc[MAXQ];
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) acc[c] = (T)0;
        const T* lp = L + (int64_t)b0 * ld + j;
        for (int u = 0; u < sw; ++u) {
            const T lv = lp[(int64_t)u * ld];
#pragma unroll
            for (int c = 0; c < MAXQ; ++c)
                if (c < q) acc[c] += lv * sx[c][u];
        }
#pragma unroll
        for (int c = 0; c < MAXQ; ++c)
            if (c < q) work[(int64_t)c * n + j] -= acc[c];
    }
}

// D5 tail: one wave per row of W.
template <typename T>
__global__ __launch_bounds__(256)
void k_predict_from_w(const T* __restrict__ W, int ns, int n, int64_t ldw, const T* __restrict__ z, int q,
                      T sf2_plus, const T* __restrict__ bias, T* __restrict__ mean, T* __restrict__ var,
                      int accumulate)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= ns) return;
    const T* wp = W + (int64_t)row * ldw;
    T ss = (T)0;
    T sm[MAXQ];
#pragma unroll
    for (int c = 0; c < MAXQ; ++c) sm[c] = (T)0;
    const bool want_mean = (mean != nullptr) && (z != nullptr);
    for (int j = lane; j < n; j += 64) {
        const T w = wp[j];
        ss += w * w;
        if (want_mean) {
#pragma unroll
            for (int c = 0; c < MAXQ; ++c)
                if (c < q) sm[c] += w * z[(int64_t)j * q + c];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
#pragma unroll
        for (int c = 0; c < MAXQ; ++c)
            if (c < q) sm[c] += __shfl_xor(sm[c], off, 64);
    }
    if (lane == 0) {
        if (var) { const T v = sf2_plus - ss; var[row] = accumulate ? var[row] + v : v; }
        if (want_mean) {
#pragma unroll
            for (int c = 0; c < MAXQ; ++c) {
                if (c < q) {
                    const T m = sm[c] + (bias ? bias[c] : (T)0);
                    T* o = mean + (int64_t)row * q + c;
                    *o = accumulate ? *o + m : m;
                }
            }
        }
    }
}

}  // namespace

template <typename T>
int potrs_run(const T* l, int64_t n, int64_t ld, const T* ws, T* rhs, int q, T* z_out, T* scratch, hipStream_t st)
{
    const char* fn = "cimrgp_potrs";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(q >= 1 && q <= MAXQ, fn, "number of right-hand sides must be in [1, 8]");
    CIMRGP_REQUIRE(n < (1ll << 31), fn, "matrix too large");
    T* work = scratch;             // (q x n) running right-hand side
    T* res  = scratch + q * n;     // (q x n) solved blocks
    const unsigned tg = (unsigned)((n * q + 255) / 256);
    hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg), dim3(256), 0, st, (const T*)rhs, work, n, q, 1);
    CIMRGP_LAUNCH_CHECK(fn);
    const int rows_per_wg = 128;
    for (int64_t b0 = 0; b0 < n; b0 += SB) {
        const int sw = (int)((n - b0 < SB) ? (n - b0) : SB);
        const int64_t below = n - (b0 + sw);
        const unsigned grid = (unsigned)((below + rows_per_wg - 1) / rows_per_wg);
        hipLaunchKernelGGL((k_fwd_step<T>), dim3(grid ? grid : 1), dim3(256), 0, st, l, ld, (int)n,
                           ws + (b0 / SB) * (SB * SB), work, res, q, (int)b0, sw, rows_per_wg);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    if (z_out) {
        hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg), dim3(256), 0, st, (const T*)res, z_out, n, q, 0);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    hipError_t e = hipMemcpyAsync(work, res, sizeof(T) * (size_t)(q * n), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return check_hip(e, fn, "hipMemcpyAsync");
    const int64_t last = ((n - 1) / SB) * SB;
    for (int64_t b0 = last; b0 >= 0; b0 -= SB) {
        const int sw = (int)((n - b0 < SB) ? (n - b0) : SB);
        const unsigned grid = (unsigned)((b0 + 255) / 256);
        hipLaunchKernelGGL((k_bwd_step<T>), dim3(grid ? grid : 1), dim3(256), 0, st, l, ld, (int)n,
                           ws + (b0 / SB) * (SB * SB), work, res, q, (int)b0, sw);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg), dim3(256), 0, st, (const T*)res, rhs, n, q, 0);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int predict_from_w_run(const T* w, int64_t ns, int64_t n, int64_t ldw, const T* z, int q, double sf2,
                       double extra, const T* bias, T* mean, T* var, int accumulate, hipStream_t st)
{
    const char* fn = "cimrgp_predict_from_w";
    if (ns <= 0) return 0;
    CIMRGP_REQUIRE(q >= 0 && q <= MAXQ, fn, "number of outputs must be <= 8");
    CIMRGP_REQUIRE(ns < (1ll << 31) && n < (1ll << 31), fn, "too many points");
    hipLaunchKernelGGL((k_predict_from_w<T>), dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, st,
                       w, (int)ns, (int)n, ldw, z, q, (T)(sf2 + extra), bias, mean, var, accumulate);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template int potrs_run<double>(const double*, int64_t, int64_t, const double*, double*, int, double*, double*, hipStream_t);
template int potrs_run<float>(const float*, int64_t, int64_t, const float*, float*, int, float*, float*, hipStream_t);
template int predict_from_w_run<double>(const double*, int64_t, int64_t, int64_t, const double*, int, double, double,
                                        const double*, double*, double*, int, hipStream_t);
template int predict_from_w_run<float>(const float*, int64_t, int64_t, int64_t, const float*, int, double, double,
                                       const float*, float*, float*, int, hipStream_t);

}  // namespace cimrgp
