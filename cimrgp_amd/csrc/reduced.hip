// Reduced-rank (Hilbert-space) block path of the reference, SURVEY.md 8f rank 2:
//   a4/a6  Laplacian eigenfunction matrix Phi (KernelClass.py:9-37, MRGP.py:337-357)
//   a7/a9  the N-dependent sums behind update_scale_given_axis / update_bias_given_noise /
//          update_noise (Posteriors.py:298-342, 345-372, 396-412)
//   a11/a14  Phi E[au]^T + bias and the matching variance (Stats.py:316-348, MRGP.py:782-803)
// Everything that does not scale with N (Bingham axes, ARD, Gamma/Normal updates) stays on the
// host.
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

// ---------------------------------------------------------------------------------------------
// Phi is never read from HBM by the two kernels below: a row of it costs one sincos per input
// dimension plus the three-term recurrence  sin((i+1)t) = 2 cos(t) sin(i t) - sin((i-1)t)
// (|error| < 2e-13 for i <= 64, measured), far cheaper than the 8 m bytes it would stream.
// HBM traffic per row is then x (d), y and f_bar (2q), f_var (1) elements -- the algorithmic
// minimum -- and both kernels are bound by the FP64 vector rate of the recurrence + reductions.
// ---------------------------------------------------------------------------------------------
template <int D>
struct SineRows {
    double prev[D], cur[D], twoc[D], norm;

    // dimensions k >= d (template rounded up) are held at the constant 1
    __device__ __forceinline__ void start(const double* xr, const double* __restrict__ interval, int d, bool live)
    {
        norm = live ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < d) {
                const double L = interval[k];
                const double theta = M_PI * (xr[k] + L) / (2.0 * L);
                double sn, cs;
                sincos(theta, &sn, &cs);
                prev[k] = 0.0;
                cur[k] = sn;
                twoc[k] = 2.0 * cs;
                norm *= 1.0 / sqrt(L);
            } else {
                prev[k] = 1.0;
                cur[k] = 1.0;
                twoc[k] = 2.0;
            }
        }
    }
    __device__ __forceinline__ double value() const
    {
        double v = norm;
#pragma unroll
        for (int k = 0; k < D; ++k) v *= cur[k];
        return v;
    }
    __device__ __forceinline__ void advance()
    {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double nx = fma(twoc[k], cur[k], -prev[k]);
            prev[k] = cur[k];
            cur[k] = nx;
        }
    }
};

template <typename T, int D>
__device__ __forceinline__ void load_row(const T* __restrict__ x, int64_t row, int d, double* xr)
{
#pragma unroll
    for (int k = 0; k < D; ++k) xr[k] = (k < d) ? (double)x[row * d + k] : 0.0;
}

constexpr int MOM_ROWS = 128;          // rows per workgroup step: one per thread (2 waves; 38 KB LDS at m = 32)
constexpr int MOM_WAVES = MOM_ROWS / 64;
constexpr int MOM_MAX_WGS = 1024;

// One record of partial sums per workgroup (doubles):
//   [0, m q)   G[i][c] = sum_n phi[n][i] r0[n][c],   r0 = y - fbar - Phi E[au]^T
//   [.., +m)   colsum phi          [.., +m)  colsum phi^2
//   [.., +q)   sum_n r0[n][c]      [.., +1)  sum_n |r0[n]|^2      [.., +1)  sum_n fvar[n]
// Step = 128 rows.  Phase A (thread = row): generate the row of Phi into LDS, form its residual.
// Phase B (lane = basis index, rows split over the wave's lane groups and the waves): reduce
// the LDS tile against the residuals.  Dynamic LDS: tile[128][m+1] + res[Q][128] + eau[Q][m] + combine.
template <typename T, int D, int Q>
__global__ __launch_bounds__(MOM_ROWS)
void k_basis_moments(const T* __restrict__ x, int d, const double* __restrict__ interval, const T* __restrict__ y,
                     const T* __restrict__ fbar, const T* __restrict__ fvar, const double* __restrict__ eau, int64_t n,
                     int m, int q, double* __restrict__ partial)
{
    extern __shared__ double lds[];
    const int pitch = m + 1;                               // odd pitch: conflict-free in both phases
    double* tile = lds;                                    // [MOM_ROWS][pitch]
    double* res = tile + MOM_ROWS * pitch;                 // [Q][MOM_ROWS]
    double* s_eau = res + Q * MOM_ROWS;                    // [Q][m]
    double* s_red = s_eau + Q * m;                         // [waves][m (Q + 2)] cross-wave combine
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < Q * m; e += MOM_ROWS) s_eau[e] = (e / m < q) ? eau[e] : 0.0;

    // phase-B geometry: width = m rounded up to a power of two (<= 64); 64 / width row groups
    int width = 1;
    while (width < m) width <<= 1;
    const int groups = 64 / width;
    const int bi = lane & (width - 1), grp = lane / width;

    double g[Q], s1 = 0.0, s2 = 0.0, sr[Q], sn = 0.0, sv = 0.0;
#pragma unroll
    for (int c = 0; c < Q; ++c) { g[c] = 0.0; sr[c] = 0.0; }
    __syncthreads();

    const int64_t steps = (n + MOM_ROWS - 1) / MOM_ROWS;
    for (int64_t step = blockIdx.x; step < steps; step += gridDim.x) {
        const int64_t row = step * MOM_ROWS + tid;
        const bool live = row < n;
        const int64_t rr = live ? row : n - 1;
        // ---- phase A
        double xr[D], t[Q];
        load_row<T, D>(x, rr, d, xr);
        double yv[Q];
#pragma unroll
        for (int c = 0; c < Q; ++c) {
            const bool on = c < q;
            const double yy = on ? (double)y[rr * q + (on ? c : 0)] : 0.0;
            const double ff = (on && fbar) ? (double)fbar[rr * q + c] : 0.0;
            yv[c] = yy - ff;
            t[c] = 0.0;
        }
        const double fv = (fvar && live) ? (double)fvar[rr] : 0.0;
        SineRows<D> gen;
        gen.start(xr, interval, d, live);
        double* my = tile + tid * pitch;
        for (int i = 0; i < m; ++i) {
            const double p = gen.value();
            my[i] = p;
#pragma unroll
            for (int c = 0; c < Q; ++c) t[c] = fma(p, s_eau[c * m + i], t[c]);
            gen.advance();
        }
#pragma unroll
        for (int c = 0; c < Q; ++c) {
            const double r = live ? yv[c] - t[c] : 0.0;
            res[c * MOM_ROWS + tid] = r;
            sr[c] += r;
            sn = fma(r, r, sn);
        }
        sv += fv;
        __syncthreads();
        // ---- phase B: wave w reduces rows [64 w, 64 w + 64), lane group grp takes every groups-th
        if (bi < m) {
            const int base = wave * 64;
            for (int r = grp; r < 64; r += groups) {
                const double p = tile[(base + r) * pitch + bi];
                s1 += p;
                s2 = fma(p, p, s2);
#pragma unroll
                for (int c = 0; c < Q; ++c) g[c] = fma(p, res[c * MOM_ROWS + base + r], g[c]);
            }
        }
        __syncthreads();
    }
    // ---- combine: lane groups (shuffles), then waves (LDS, fixed order), then the per-row sums
    for (int off = width; off < 64; off <<= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
#pragma unroll
        for (int c = 0; c < Q; ++c) g[c] += __shfl_xor(g[c], off, 64);
    }
    const int per_wave = m * (Q + 2);
    if (lane < m) {
#pragma unroll
        for (int c = 0; c < Q; ++c) s_red[wave * per_wave + lane * Q + c] = g[c];
        s_red[wave * per_wave + m * Q + lane] = s1;
        s_red[wave * per_wave + m * Q + m + lane] = s2;
    }
    // per-row sums: wave reduction then one value per wave into the tile area (free by now)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sn += __shfl_xor(sn, off, 64);
        sv += __shfl_xor(sv, off, 64);
#pragma unroll
        for (int c = 0; c < Q; ++c) sr[c] += __shfl_xor(sr[c], off, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < Q; ++c) tile[wave * (Q + 2) + c] = sr[c];
        tile[wave * (Q + 2) + Q] = sn;
        tile[wave * (Q + 2) + Q + 1] = sv;
    }
    __syncthreads();
    const int rec = m * q + 2 * m + q + 2;
    // partial[e][workgroup]: the final pass then reads each element's partials contiguously
    (void)rec;
    double* out = partial + blockIdx.x;
    const int64_t pitch_wg = gridDim.x;
    for (int e = tid; e < m * q; e += MOM_ROWS) {
        const int i = e / q, c = e - i * q;
        double v = 0.0;
        for (int w = 0; w < MOM_WAVES; ++w) v += s_red[w * per_wave + i * Q + c];
        out[e * pitch_wg] = v;
    }
    for (int e = tid; e < 2 * m; e += MOM_ROWS) {
        double v = 0.0;
        for (int w = 0; w < MOM_WAVES; ++w) v += s_red[w * per_wave + m * Q + e];
        out[(m * q + e) * pitch_wg] = v;
    }
    if (tid < q + 2) {
        const int src = (tid < q) ? tid : Q + (tid - q);
        double v = 0.0;
        for (int w = 0; w < MOM_WAVES; ++w) v += tile[w * (Q + 2) + src];
        out[(m * q + 2 * m + tid) * pitch_wg] = v;
    }
}

// One workgroup per output element: 256 strided partial sums, then a fixed-shape tree -- the
// result depends on the launch geometry only, never on timing.
__global__ __launch_bounds__(256)
void k_basis_moments_final(const double* __restrict__ partial, int nwg, double* __restrict__ out)
{
    __shared__ double tree[256];
    const double* src = partial + (int64_t)blockIdx.x * nwg;
    double s = 0.0;
    for (int w = threadIdx.x; w < nwg; w += 256) s += src[w];
    tree[threadIdx.x] = s;
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half) tree[threadIdx.x] += tree[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = tree[0];
}

// mean[n][c] (+)= bias[c] + sum_i phi[n][i] eau[c][i];  var[n] (+)= bias_var + sum_i phi[n][i]^2 c2[i]
// Thread = row; Phi regenerated by the recurrence, coefficients broadcast from LDS.
template <typename T, int D, int Q>
__global__ __launch_bounds__(256)
void k_basis_apply(const T* __restrict__ x, int64_t n, int d, const double* __restrict__ interval, int m,
                   const double* __restrict__ eau, int q, const double* __restrict__ bias, const double* __restrict__ c2,
                   double bias_var, T* __restrict__ mean, T* __restrict__ var, int accumulate)
{
    __shared__ double s_eau[Q * RB_MAXM];
    __shared__ double s_c2[RB_MAXM];
    for (int e = threadIdx.x; e < Q * m; e += 256) s_eau[e] = (e / m < q) ? eau[e] : 0.0;
    for (int e = threadIdx.x; e < m; e += 256) s_c2[e] = c2 ? c2[e] : 0.0;
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    double xr[D], mu[Q], v = bias_var;
    load_row<T, D>(x, row, d, xr);
#pragma unroll
    for (int c = 0; c < Q; ++c) mu[c] = (c < q && bias) ? bias[c] : 0.0;
    SineRows<D> gen;
    gen.start(xr, interval, d, true);
    for (int i = 0; i < m; ++i) {
        const double p = gen.value();
        v = fma(p * p, s_c2[i], v);
#pragma unroll
        for (int c = 0; c < Q; ++c) mu[c] = fma(p, s_eau[c * m + i], mu[c]);
        gen.advance();
    }
    if (mean) {
#pragma unroll
        for (int c = 0; c < Q; ++c) {
            if (c < q) {
                T* o = mean + row * q + c;
                *o = accumulate ? (T)((double)*o + mu[c]) : (T)mu[c];
            }
        }
    }
    if (var) var[row] = accumulate ? (T)((double)var[row] + v) : (T)v;
}

template <typename T, int D>
int moments_launch(int qt, dim3 grid, size_t lds, hipStream_t st, const T* x, int d, const double* interval, const T* y,
                   const T* fbar, const T* fvar, const double* eau, int64_t n, int m, int q, double* partial)
{
    switch (qt) {
        case 2: hipLaunchKernelGGL((k_basis_moments<T, D, 2>), grid, dim3(MOM_ROWS), lds, st, x, d, interval, y, fbar, fvar, eau, n, m, q, partial); break;
        case 4: hipLaunchKernelGGL((k_basis_moments<T, D, 4>), grid, dim3(MOM_ROWS), lds, st, x, d, interval, y, fbar, fvar, eau, n, m, q, partial); break;
        default: hipLaunchKernelGGL((k_basis_moments<T, D, 8>), grid, dim3(MOM_ROWS), lds, st, x, d, interval, y, fbar, fvar, eau, n, m, q, partial); break;
    }
    return 0;
}

template <typename T, int D>
int apply_launch(int qt, dim3 grid, hipStream_t st, const T* x, int64_t n, int d, const double* interval, int m,
                 const double* eau, int q, const double* bias, const double* c2, double bias_var, T* mean, T* var, int acc)
{
    switch (qt) {
        case 2: hipLaunchKernelGGL((k_basis_apply<T, D, 2>), grid, dim3(256), 0, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, acc); break;
        case 4: hipLaunchKernelGGL((k_basis_apply<T, D, 4>), grid, dim3(256), 0, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, acc); break;
        default: hipLaunchKernelGGL((k_basis_apply<T, D, 8>), grid, dim3(256), 0, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, acc); break;
    }
    return 0;
}

inline int q_template(int q) { return q <= 2 ? 2 : (q <= 4 ? 4 : 8); }
inline int d_template(int d) { return d <= 3 ? d : (d == 4 ? 4 : 8); }

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

int basis_moments_workgroups(int64_t n)
{
    const int64_t steps = (n + MOM_ROWS - 1) / MOM_ROWS;
    return (int)(steps < MOM_MAX_WGS ? steps : MOM_MAX_WGS);
}

template <typename T>
int basis_moments_run(const T* x, int64_t n, int d, const double* interval, int m, const T* y, const T* fbar, const T* fvar,
                      const double* eau, int q, double* out, double* scratch, hipStream_t st)
{
    const char* fn = "cimrgp_basis_moments";
    CIMRGP_REQUIRE(n > 0, fn, "empty block");
    CIMRGP_REQUIRE(d >= 1 && d <= RB_MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(m >= 1 && m <= RB_MAXM && q >= 1 && q <= RB_MAXQ, fn, "m must be in [1, 64], q in [1, 8]");
    const int nwg = basis_moments_workgroups(n);
    const int rec = m * q + 2 * m + q + 2;
    const int qt = q_template(q);
    const size_t lds = sizeof(double) * ((size_t)MOM_ROWS * (m + 1) + (size_t)qt * MOM_ROWS + (size_t)qt * m + MOM_WAVES * (size_t)m * (qt + 2));
    const dim3 grid((unsigned)nwg);
#define CIMRGP_MOM(DD) \
    do { \
        if (lds > 64 * 1024) { \
            (void)hipFuncSetAttribute((const void*)k_basis_moments<T, DD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            (void)hipFuncSetAttribute((const void*)k_basis_moments<T, DD, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            (void)hipFuncSetAttribute((const void*)k_basis_moments<T, DD, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        } \
        moments_launch<T, DD>(qt, grid, lds, st, x, d, interval, y, fbar, fvar, eau, n, m, q, scratch); \
    } while (0)
    switch (d_template(d)) {
        case 1: CIMRGP_MOM(1); break;
        case 2: CIMRGP_MOM(2); break;
        case 3: CIMRGP_MOM(3); break;
        case 4: CIMRGP_MOM(4); break;
        default: CIMRGP_MOM(8); break;
    }
#undef CIMRGP_MOM
    CIMRGP_LAUNCH_CHECK(fn);
    hipLaunchKernelGGL(k_basis_moments_final, dim3((unsigned)rec), dim3(256), 0, st, (const double*)scratch, nwg, out);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int basis_apply_run(const T* x, int64_t n, int d, const double* interval, int m, const double* eau, int q, const double* bias,
                    const double* c2, double bias_var, T* mean, T* var, int accumulate, hipStream_t st)
{
    const char* fn = "cimrgp_basis_apply";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(d >= 1 && d <= RB_MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(m >= 1 && m <= RB_MAXM && q >= 1 && q <= RB_MAXQ, fn, "m must be in [1, 64], q in [1, 8]");
    const dim3 grid((unsigned)((n + 255) / 256));
    const int qt = q_template(q);
    switch (d_template(d)) {
        case 1: apply_launch<T, 1>(qt, grid, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, accumulate); break;
        case 2: apply_launch<T, 2>(qt, grid, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, accumulate); break;
        case 3: apply_launch<T, 3>(qt, grid, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, accumulate); break;
        case 4: apply_launch<T, 4>(qt, grid, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, accumulate); break;
        default: apply_launch<T, 8>(qt, grid, st, x, n, d, interval, m, eau, q, bias, c2, bias_var, mean, var, accumulate); break;
    }
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

#define CIMRGP_INST(T)                                                                                         \
    template int laplace_basis_run<T>(const T*, int64_t, int, const double*, int, T*, hipStream_t);            \
    template int basis_moments_run<T>(const T*, int64_t, int, const double*, int, const T*, const T*, const T*,    \
                                      const double*, int, double*, double*, hipStream_t);                      \
    template int basis_apply_run<T>(const T*, int64_t, int, const double*, int, const double*, int, const double*, \
                                    const double*, double, T*, T*, int, hipStream_t);
CIMRGP_INST(double)
CIMRGP_INST(float)
#undef CIMRGP_INST

}  // namespace cimrgp
