// D6 / a11: small element-wise and reduction kernels of the residual chain.
// All HBM-bound and tiny (O(n q)); they exist so that the product path never
// leaves the device between blocks.
#include "common.hpp"

namespace cimrgp {

namespace {

constexpr int MAXQ = 8;

template <typename T>
static __device__ __forceinline__ T block_sum_1024(T v, T* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    T s = (T)0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];   // fixed order: deterministic
    return s;
}

// stats[c] = mean_i (y - fbar)[i][c];  stats[q] = pooled population variance
// of the centred block.  One workgroup (the block is at most a few MB).
template <typename T>
__global__ __launch_bounds__(1024)
void k_block_stats(const T* __restrict__ y, const T* __restrict__ fbar, int64_t n, int q, T* __restrict__ stats)
{
    __shared__ T red[16];
    __shared__ T smean[MAXQ];
    for (int c = 0; c < q; ++c) {
        T s = (T)0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
            s += y[i * q + c] - (fbar ? fbar[i * q + c] : (T)0);
        s = block_sum_1024(s, red);
        if (threadIdx.x == 0) { smean[c] = s / (T)n; stats[c] = smean[c]; }
    }
    __syncthreads();
    T s2 = (T)0;
    for (int64_t e = threadIdx.x; e < n * q; e += blockDim.x) {
        const int c = (int)(e % q);
        const T r = y[e] - (fbar ? fbar[e] : (T)0) - smean[c];
        s2 += r * r;
    }
    s2 = block_sum_1024(s2, red);
    if (threadIdx.x == 0) stats[q] = s2 / (T)(n * q);
}

template <typename T>
__global__ void k_residual(const T* __restrict__ y, const T* __restrict__ fbar, const T* __restrict__ bias,
                           int64_t total, int q, T* __restrict__ r)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % q);
    r[e] = y[e] - (fbar ? fbar[e] : (T)0) - (bias ? bias[c] : (T)0);
}

template <typename T>
__global__ void k_train_mean(const T* __restrict__ r, const T* __restrict__ alpha, const T* __restrict__ bias,
                             const T* __restrict__ noise, int64_t total, int q, T* __restrict__ out, int accumulate)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % q);
    const T v = r[e] - noise[0] * alpha[e] + (bias ? bias[c] : (T)0);
    out[e] = accumulate ? out[e] + v : v;
}

template <typename T>
__global__ void k_add_diag(T* __restrict__ k, int64_t n, int64_t ld, const T* __restrict__ noise)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) k[i * ld + i] += noise[0];
}

template <typename T>
__global__ void k_noise_from_stats(const T* __restrict__ stats, int q, T frac, T floor_value, T* __restrict__ noise)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const T v = frac * stats[q];
        noise[0] = (v > floor_value) ? v : floor_value;
    }
}

template <typename T>
__global__ __launch_bounds__(1024)
void k_logdet_half(const T* __restrict__ l, int64_t n, int64_t ld, double* __restrict__ out)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += log((double)l[i * ld + i]);
    s = block_sum_1024(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace

template <typename T>
int misc_block_stats(const T* y, const T* fbar, int64_t n, int q, T* stats, hipStream_t st)
{
    const char* fn = "cimrgp_block_stats";
    CIMRGP_REQUIRE(n > 0, fn, "empty block");
    CIMRGP_REQUIRE(q >= 1 && q <= MAXQ, fn, "number of outputs must be in [1, 8]");
    hipLaunchKernelGGL((k_block_stats<T>), dim3(1), dim3(1024), 0, st, y, fbar, n, q, stats);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int misc_residual(const T* y, const T* fbar, const T* bias, int64_t n, int q, T* r, hipStream_t st)
{
    const char* fn = "cimrgp_residual";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(q >= 1, fn, "q must be positive");
    const int64_t total = n * q;
    hipLaunchKernelGGL((k_residual<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, fbar, bias, total, q, r);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int misc_train_mean(const T* r, const T* alpha, const T* bias, const T* noise, int64_t n, int q, T* out,
                    int accumulate, hipStream_t st)
{
    const char* fn = "cimrgp_train_mean";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(q >= 1, fn, "q must be positive");
    const int64_t total = n * q;
    hipLaunchKernelGGL((k_train_mean<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       r, alpha, bias, noise, total, q, out, accumulate);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int misc_add_diag(T* k, int64_t n, int64_t ld, const T* noise, hipStream_t st)
{
    const char* fn = "cimrgp_add_diag";
    if (n <= 0) return 0;
    hipLaunchKernelGGL((k_add_diag<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k, n, ld, noise);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int misc_noise_from_stats(const T* stats, int q, double frac, double floor_value, T* noise, hipStream_t st)
{
    const char* fn = "cimrgp_noise_from_stats";
    hipLaunchKernelGGL((k_noise_from_stats<T>), dim3(1), dim3(64), 0, st, stats, q, (T)frac, (T)floor_value, noise);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int misc_logdet_half(const T* l, int64_t n, int64_t ld, double* out, hipStream_t st)
{
    const char* fn = "cimrgp_logdet_half";
    CIMRGP_REQUIRE(n > 0, fn, "empty matrix");
    hipLaunchKernelGGL((k_logdet_half<T>), dim3(1), dim3(1024), 0, st, l, n, ld, out);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

#define CIMRGP_INST(T)                                                                                   \
    template int misc_block_stats<T>(const T*, const T*, int64_t, int, T*, hipStream_t);                 \
    template int misc_residual<T>(const T*, const T*, const T*, int64_t, int, T*, hipStream_t);          \
    template int misc_train_mean<T>(const T*, const T*, const T*, const T*, int64_t, int, T*, int, hipStream_t); \
    template int misc_add_diag<T>(T*, int64_t, int64_t, const T*, hipStream_t);                          \
    template int misc_noise_from_stats<T>(const T*, int, double, double, T*, hipStream_t);               \
    template int misc_logdet_half<T>(const T*, int64_t, int64_t, double*, hipStream_t);
CIMRGP_INST(double)
CIMRGP_INST(float)
#undef CIMRGP_INST

}  // namespace cimrgp
