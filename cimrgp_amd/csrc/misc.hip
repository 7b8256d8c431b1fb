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

// ---------------------------------------------------------------------------
// Gradient of the log marginal likelihood (the `.optimize()` step of
// RegressionInput.py:63).  With G = alpha alpha^T - q K^-1 and the RBF parametrisation
// K = sf E + noise I,  E_ij = exp(-d2_ij / (2 l^2)):
//     dLML/dlog(sf)    = 1/2 sum_ij G_ij sf E_ij
//     dLML/dlog(l)     = 1/2 sum_ij G_ij sf E_ij d2_ij / l^2
//     dLML/dlog(noise) = 1/2 noise sum_i G_ii
// One pass over the LOWER triangle of K^-1 (off-diagonal entries count twice); the kernel
// matrix is re-evaluated on the fly from X, nothing n x n besides K^-1 is read.  Per-tile
// partial sums, then a fixed-order final reduction (deterministic).
// ---------------------------------------------------------------------------
namespace cimrgp {
namespace {

constexpr int LG_T = 64;
constexpr int LG_MAXD = 8;

// ARD: one length-scale per input dimension.  The caller passes inputs already divided by their
// length-scales (so the kernel is evaluated with l = 1) and gets d/dlog l_k = 1/2 sum G K d_k^2 per
// dimension: partial record = [sf | l_1 .. l_8 | trace] (LG_NP entries) instead of [sf | l | trace].
constexpr int LG_NP = LG_MAXD + 2;

template <typename T, bool ARD>
__global__ __launch_bounds__(256)
void k_lml_grad_tiles(const T* __restrict__ x, int n, int d, const T* __restrict__ kinv, int64_t ld,
                      const T* __restrict__ alpha, int q, T neg_half_inv_l2, T sf2, T inv_l2,
                      double* __restrict__ partial)
{
    __shared__ T sa[LG_T * LG_MAXD], sb[LG_T * LG_MAXD];
    __shared__ T aa[LG_T * 8], ab[LG_T * 8];
    __shared__ double red[ARD ? LG_NP : 3][4];
    const int id = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > id) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
    const int tj = id - ti * (ti + 1) / 2;
    const int row0 = ti * LG_T, col0 = tj * LG_T;
    const int tid = threadIdx.x;
    for (int e = tid; e < LG_T * LG_MAXD; e += 256) {
        const int r = e / LG_MAXD, k = e - r * LG_MAXD;
        sa[e] = (k < d && row0 + r < n) ? x[(int64_t)(row0 + r) * d + k] : (T)0;
        sb[e] = (k < d && col0 + r < n) ? x[(int64_t)(col0 + r) * d + k] : (T)0;
        aa[e] = (k < q && row0 + r < n) ? alpha[(int64_t)(row0 + r) * q + k] : (T)0;
        ab[e] = (k < q && col0 + r < n) ? alpha[(int64_t)(col0 + r) * q + k] : (T)0;
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    double s_sf = 0.0, s_l = 0.0, s_tr = 0.0;
    double s_lk[ARD ? LG_MAXD : 1];
#pragma unroll
    for (int k = 0; k < (ARD ? LG_MAXD : 1); ++k) s_lk[k] = 0.0;
    const int gc = col0 + tx;
    for (int rr = ty; rr < LG_T; rr += 4) {
        const int gr = row0 + rr;
        if (gr < n && gc < n && gc <= gr) {
            T d2 = (T)0;
            T dk2[LG_MAXD];
#pragma unroll
            for (int k = 0; k < LG_MAXD; ++k) {
                const T df = sa[rr * LG_MAXD + k] - sb[tx * LG_MAXD + k];
                dk2[k] = df * df;
                d2 += dk2[k];
            }
            T aat = (T)0;
#pragma unroll
            for (int k = 0; k < 8; ++k) aat += aa[rr * 8 + k] * ab[tx * 8 + k];
            const T g = aat - (T)q * kinv[(int64_t)gr * ld + gc];
            const T kf = sf2 * exp(d2 * neg_half_inv_l2);
            const double wgt = (gr == gc) ? 1.0 : 2.0;
            s_sf += wgt * (double)(g * kf);
            if (ARD) {
#pragma unroll
                for (int k = 0; k < LG_MAXD; ++k) s_lk[k] += wgt * (double)(g * kf * dk2[k]);
            } else {
                s_l  += wgt * (double)(g * kf * d2 * inv_l2);
            }
            if (gr == gc) s_tr += (double)g;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_sf += __shfl_xor(s_sf, off, 64);
        s_l  += __shfl_xor(s_l, off, 64);
        s_tr += __shfl_xor(s_tr, off, 64);
        if (ARD) {
#pragma unroll
            for (int k = 0; k < LG_MAXD; ++k) s_lk[k] += __shfl_xor(s_lk[k], off, 64);
        }
    }
    if (ARD) {
        if (tx == 0) {
            red[0][ty] = s_sf;
#pragma unroll
            for (int k = 0; k < LG_MAXD; ++k) red[1 + k][ty] = s_lk[k];
            red[LG_NP - 1][ty] = s_tr;
        }
        __syncthreads();
        if (tid < LG_NP) partial[(int64_t)id * LG_NP + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    } else {
        if (tx == 0) { red[0][ty] = s_sf; red[1][ty] = s_l; red[2][ty] = s_tr; }
        __syncthreads();
        if (tid < 3) partial[(int64_t)id * 3 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    }
}

__global__ __launch_bounds__(1024)
void k_lml_grad_final(const double* __restrict__ partial, int64_t ntiles, double noise, double* __restrict__ out,
                      int np, int nout)
{
    // np entries per tile record, the last one the trace term; the first nout - 1 are copied out
    __shared__ double red[16];
    for (int c = 0; c < nout; ++c) {
        const int src = (c == nout - 1) ? np - 1 : c;
        double s = 0.0;
        for (int64_t t = threadIdx.x; t < ntiles; t += blockDim.x) s += partial[t * np + src];
        s = block_sum_1024(s, red);
        if (threadIdx.x == 0) out[c] = 0.5 * s * (c == nout - 1 ? noise : 1.0);
        __syncthreads();
    }
}

}  // namespace

template <typename T>
int lml_grad_run(const T* x, int64_t n, int d, const T* kinv, int64_t ld, const T* alpha, int q,
                 double ell, double sf2, double noise, double* out3, double* scratch, hipStream_t st, bool ard)
{
    const char* fn = "cimrgp_lml_grad";
    CIMRGP_REQUIRE(n > 0 && n < (1ll << 30), fn, "bad size");
    CIMRGP_REQUIRE(d >= 1 && d <= LG_MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(q >= 1 && q <= 8, fn, "number of outputs must be in [1, 8]");
    const int64_t tm = (n + LG_T - 1) / LG_T, tiles = tm * (tm + 1) / 2;
    CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
    if (ard) {
        // inputs are pre-scaled by the length-scales: unit length-scale here; out = [sf | l_1..l_d | noise]
        hipLaunchKernelGGL((k_lml_grad_tiles<T, true>), dim3((unsigned)tiles), dim3(256), 0, st, x, (int)n, d, kinv, ld, alpha, q,
                           (T)(-0.5), (T)sf2, (T)1, scratch);
        CIMRGP_LAUNCH_CHECK(fn);
        hipLaunchKernelGGL(k_lml_grad_final, dim3(1), dim3(1024), 0, st, (const double*)scratch, tiles, noise, out3, LG_NP, d + 2);
    } else {
        hipLaunchKernelGGL((k_lml_grad_tiles<T, false>), dim3((unsigned)tiles), dim3(256), 0, st, x, (int)n, d, kinv, ld, alpha, q,
                           (T)(-0.5 / (ell * ell)), (T)sf2, (T)(1.0 / (ell * ell)), scratch);
        CIMRGP_LAUNCH_CHECK(fn);
        hipLaunchKernelGGL(k_lml_grad_final, dim3(1), dim3(1024), 0, st, (const double*)scratch, tiles, noise, out3, 3, 3);
    }
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template int lml_grad_run<double>(const double*, int64_t, int, const double*, int64_t, const double*, int, double, double,
                                  double, double*, double*, hipStream_t, bool);
template int lml_grad_run<float>(const float*, int64_t, int, const float*, int64_t, const float*, int, double, double,
                                 double, double*, double*, hipStream_t, bool);

}  // namespace cimrgp
