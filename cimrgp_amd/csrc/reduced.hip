// Reduced-rank (Hilbert-space) block path of the reference, SURVEY.md 8f rank 2:
//   a4/a6  Laplacian eigenfunction matrix Phi (KernelClass.py:9-37, MRGP.py:337-357)
//   a7/a9  the N-dependent sums behind update_scale_given_axis / update_bias_given_noise /
//          update_noise (Posteriors.py:298-342, 345-372, 396-412)
//   a11/a14  Phi E[au]^T + bias and the matching variance (Stats.py:316-348, MRGP.py:782-803)
// Everything that does not scale with N (Bingham axes, ARD, Gamma/Normal updates) stays on the
// host.  All three kernels are HBM-bound skinny passes over Phi (n x m, m <= 64).
#include "common.hpp"

namespace cimrgp {

namespace {

constexpr int RB_MAXM = 64;
constexpr int RB_MAXQ = 8;
constexpr int RB_MAXD = 8;

// phi[n][m]: phi_i(x) = prod_k L_k^-1/2 sin(pi (i+1) (x_k + L_k) / (2 L_k))
template <typename T>
__global__ __launch_bounds__(256)
void k_laplace_basis(const T* __restrict__ x, int64_t n, int d, const double* __restrict__ interval, int m,
                     T* __restrict__ phi)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * m) return;
    const int64_t r = e / m;
    const int i = (int)(e - r * m) + 1;
    double v = 1.0;
    for (int k = 0; k < d; ++k) {
        const double L = interval[k];
        const double up = M_PI * (double)i * ((double)x[r * d + k] + L);
        v *= sin(up / (2.0 * L)) / sqrt(L);
    }
    phi[e] = (T)v;
}

// Per-workgroup partial sums over a slice of rows; layout of one record (doubles):
//   [0, m q)        G[i][c] = sum_n phi[n][i] r0[n][c],   r0 = y - fbar - Phi E[au]^T
//   [m q, +m)       s1[i]   = sum_n phi[n][i]
//   [.., +m)        s2[i]   = sum_n phi[n][i]^2
//   [.., +q)        sr[c]   = sum_n r0[n][c]
//   [.., +1)        sn      = sum_n |r0[n]|^2
//   [.., +1)        sv      = sum_n fvar[n]
template <typename T>
__global__ __launch_bounds__(256)
void k_basis_moments(const T* __restrict__ phi, const T* __restrict__ y, const T* __restrict__ fbar,
                     const T* __restrict__ fvar, const double* __restrict__ eau, int64_t n, int m, int q,
                     int rows_per_wg, double* __restrict__ partial)
{
    __shared__ double s_eau[RB_MAXQ * RB_MAXM];
    __shared__ double acc[RB_MAXM * RB_MAXQ + 2 * RB_MAXM + RB_MAXQ + 2];
    const int rec = m * q + 2 * m + q + 2;
    const int tid = threadIdx.x;
    for (int e = tid; e < q * m; e += 256) s_eau[e] = eau[e];
    for (int e = tid; e < rec; e += 256) acc[e] = 0.0;
    __syncthreads();
    // thread = (row slot rs = tid / 64, basis lane i = tid & 63): 4 rows at a time
    const int i = tid & 63, rs = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
    double g[RB_MAXQ], s1 = 0.0, s2 = 0.0, sr[RB_MAXQ], sn = 0.0, sv = 0.0;
#pragma unroll
    for (int c = 0; c < RB_MAXQ; ++c) { g[c] = 0.0; sr[c] = 0.0; }
    for (int64_t r = r0 + rs; r < r1; r += 4) {
        const double p = (i < m) ? (double)phi[r * m + i] : 0.0;
        // residual of this row: every lane needs it; computed by a wave-wide reduction over i
        double res[RB_MAXQ];
#pragma unroll
        for (int c = 0; c < RB_MAXQ; ++c) {
            if (c < q) {
                double t = (i < m) ? p * s_eau[c * m + i] : 0.0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                res[c] = (double)y[r * q + c] - (fbar ? (double)fbar[r * q + c] : 0.0) - t;
            } else {
                res[c] = 0.0;
            }
        }
        s1 += p;
        s2 += p * p;
#pragma unroll
        for (int c = 0; c < RB_MAXQ; ++c) g[c] += p * res[c];
        if (i == 0) {
#pragma unroll
            for (int c = 0; c < RB_MAXQ; ++c) { sr[c] += res[c]; sn += res[c] * res[c]; }
            if (fvar) sv += (double)fvar[r];
        }
    }
    // combine the 4 row slots (fixed order: slot 0..3 through shared memory atomics would not be
    // deterministic, so serialise by slot)
    for (int s = 0; s < 4; ++s) {
        if (rs == s) {
            if (i < m) {
                for (int c = 0; c < q; ++c) acc[i * q + c] += g[c];
                acc[m * q + i] += s1;
                acc[m * q + m + i] += s2;
            }
            if (i == 0) {
                for (int c = 0; c < q; ++c) acc[m * q + 2 * m + c] += sr[c];
                acc[m * q + 2 * m + q] += sn;
                acc[m * q + 2 * m + q + 1] += sv;
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < rec; e += 256) partial[(int64_t)blockIdx.x * rec + e] = acc[e];
}

__global__ __launch_bounds__(256)
void k_basis_moments_final(const double* __restrict__ partial, int nwg, int rec, double* __restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rec) return;
    double s = 0.0;
    for (int w = 0; w < nwg; ++w) s += partial[(int64_t)w * rec + e];     // fixed order
    out[e] = s;
}

// mean[n][c] (+)= bias[c] + sum_i phi[n][i] eau[c][i];  var[n] (+)= bias_var + sum_i phi[n][i]^2 c2[i]
template <typename T>
__global__ __launch_bounds__(256)
void k_basis_apply(const T* __restrict__ phi, int64_t n, int m, const double* __restrict__ eau, int q,
                   const double* __restrict__ bias, const double* __restrict__ c2, double bias_var,
                   T* __restrict__ mean, T* __restrict__ var, int accumulate)
{
    __shared__ double s_eau[RB_MAXQ * RB_MAXM];
    __shared__ double s_c2[RB_MAXM];
    for (int e = threadIdx.x; e < q * m; e += 256) s_eau[e] = eau[e];
    for (int e = threadIdx.x; e < m; e += 256) s_c2[e] = c2 ? c2[e] : 0.0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double mu[RB_MAXQ], v = bias_var;
#pragma unroll
    for (int c = 0; c < RB_MAXQ; ++c) mu[c] = (c < q && bias) ? bias[c] : 0.0;
    for (int i = 0; i < m; ++i) {
        const double p = (double)phi[r * m + i];
        v += p * p * s_c2[i];
#pragma unroll
        for (int c = 0; c < RB_MAXQ; ++c)
            if (c < q) mu[c] += p * s_eau[c * m + i];
    }
    if (mean) {
        for (int c = 0; c < q; ++c) {
            T* o = mean + r * q + c;
            *o = accumulate ? (T)((double)*o + mu[c]) : (T)mu[c];
        }
    }
    if (var) var[r] = accumulate ? (T)((double)var[r] + v) : (T)v;
}

}  // namespace

template <typename T>
int laplace_basis_run(const T* x, int64_t n, int d, const double* interval, int m, T* phi, hipStream_t st)
{
    const char* fn = "cimrgp_laplace_basis";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(d >= 1 && d <= RB_MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(m >= 1 && m <= RB_MAXM, fn, "number of basis functions must be in [1, 64]");
    const int64_t total = n * m;
    hipLaunchKernelGGL((k_laplace_basis<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, n, d, interval, m, phi);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int basis_moments_run(const T* phi, const T* y, const T* fbar, const T* fvar, const double* eau, int64_t n, int m, int q,
                      double* out, double* scratch, hipStream_t st)
{
    const char* fn = "cimrgp_basis_moments";
    CIMRGP_REQUIRE(n > 0, fn, "empty block");
    CIMRGP_REQUIRE(m >= 1 && m <= RB_MAXM && q >= 1 && q <= RB_MAXQ, fn, "m must be in [1, 64], q in [1, 8]");
    const int rows_per_wg = 256;
    const int nwg = (int)((n + rows_per_wg - 1) / rows_per_wg);
    const int rec = m * q + 2 * m + q + 2;
    hipLaunchKernelGGL((k_basis_moments<T>), dim3((unsigned)nwg), dim3(256), 0, st, phi, y, fbar, fvar, eau, n, m, q,
                       rows_per_wg, scratch);
    CIMRGP_LAUNCH_CHECK(fn);
    hipLaunchKernelGGL(k_basis_moments_final, dim3((unsigned)((rec + 255) / 256)), dim3(256), 0, st,
                       (const double*)scratch, nwg, rec, out);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int basis_apply_run(const T* phi, int64_t n, int m, const double* eau, int q, const double* bias, const double* c2,
                    double bias_var, T* mean, T* var, int accumulate, hipStream_t st)
{
    const char* fn = "cimrgp_basis_apply";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(m >= 1 && m <= RB_MAXM && q >= 1 && q <= RB_MAXQ, fn, "m must be in [1, 64], q in [1, 8]");
    hipLaunchKernelGGL((k_basis_apply<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, phi, n, m, eau, q, bias, c2,
                       bias_var, mean, var, accumulate);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

#define CIMRGP_INST(T)                                                                                         \
    template int laplace_basis_run<T>(const T*, int64_t, int, const double*, int, T*, hipStream_t);            \
    template int basis_moments_run<T>(const T*, const T*, const T*, const T*, const double*, int64_t, int, int, double*, \
                                      double*, hipStream_t);                                                   \
    template int basis_apply_run<T>(const T*, int64_t, int, const double*, int, const double*, const double*, double, T*, \
                                    T*, int, hipStream_t);
CIMRGP_INST(double)
CIMRGP_INST(float)
#undef CIMRGP_INST

}  // namespace cimrgp
