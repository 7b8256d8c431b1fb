// One C call per layer for a batch of equal-sized blocks (round 3): the reference's independent-over-l
// loops -- fit: Posteriors.py:35-59 (update_scale_given_axis over the regions of a resolution),
// predict: MRGP.py:782-803 (per-region predictions of a resolution, concatenated) -- with every step
// launched ONCE for all the blocks (blockIdx.y = block).  Regions are contiguous ranges of the layer's
// arrays (Inputs.py:57-60), so a block is a row offset (`starts`) into x / y / f_bar.
//
//   fit      statistics -> bias, noise | residual rows | Gram + noise | factorisation with the residual
//            rows carried (z = L^-1 r) | backward solve (alpha) | training-point prediction
//   predict  cross-Gram | row-wise solve W = K* L^-T | mean = W z + bias, var = sf - sum W^2 (+ noise)
//
// Round 2 issued five launches per block from a Python loop for the front end and one row-wise solve
// per block for the prediction: a layer of 128 blocks of 2048 points was launch-bound (22 ms for 0.37
// Tflop).
#include "common.hpp"

namespace cimrgp {

namespace {

constexpr int LY_MAXQ = 8;

template <typename T>
static __device__ __forceinline__ T ly_block_sum(T v, T* red)
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

// Per block b: column means of (y - f_bar) and the pooled population variance about them
// (k_block_stats' arithmetic, misc.hip), then
//   bias[b]  = the shared bias if given, else the column means
//   noise[b] = the fixed value if >= 0, else the shared noise if given, else max(frac var, floor)
// (RegressionInput.py:62: labels.var() * 0.01).
template <typename T>
__global__ __launch_bounds__(1024)
void k_layer_stats(const T* __restrict__ y, const T* __restrict__ fbar, const int64_t* __restrict__ starts, int64_t n, int q,
                   T noise_fixed, T frac, T floor_value, const T* __restrict__ shared_bias,
                   const T* __restrict__ shared_noise, T* __restrict__ bias, T* __restrict__ noise)
{
    __shared__ T red[16];
    __shared__ T smean[LY_MAXQ];
    const int b = blockIdx.x;
    const T* yb = y + starts[b] * q;
    const T* fb = fbar ? fbar + starts[b] * q : nullptr;
    for (int c = 0; c < q; ++c) {
        T s = (T)0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
            s += yb[i * q + c] - (fb ? fb[i * q + c] : (T)0);
        s = ly_block_sum(s, red);
        if (threadIdx.x == 0) smean[c] = s / (T)n;
    }
    __syncthreads();
    T s2 = (T)0;
    for (int64_t e = threadIdx.x; e < n * q; e += blockDim.x) {
        const int c = (int)(e % q);
        const T r = yb[e] - (fb ? fb[e] : (T)0) - smean[c];
        s2 += r * r;
    }
    s2 = ly_block_sum(s2, red);
    if (threadIdx.x == 0) {
        for (int c = 0; c < q; ++c) bias[(int64_t)b * q + c] = shared_bias ? shared_bias[c] : smean[c];
        T v;
        if (noise_fixed >= (T)0) v = noise_fixed;
        else if (shared_noise) v = shared_noise[0];
        else {
            v = frac * (s2 / (T)(n * q));
            if (!(v > floor_value)) v = floor_value;
        }
        noise[b] = v;
    }
}

// rows[b][c][j] = y - f_bar - bias: the block's residual, transposed, as the q rows carried through the
// factorisation
template <typename T>
__global__ void k_layer_rows(const T* __restrict__ y, const T* __restrict__ fbar, const int64_t* __restrict__ starts,
                             int64_t n, int q, const T* __restrict__ bias, T* __restrict__ rows, int64_t ldr, int64_t srows)
{
    const int b = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * q) return;
    const int64_t j = e / q;
    const int c = (int)(e - j * q);
    const int64_t g = (starts[b] + j) * q + c;
    rows[(int64_t)b * srows + (int64_t)c * ldr + j] = y[g] - (fbar ? fbar[g] : (T)0) - bias[(int64_t)b * q + c];
}

// z[b][j][c] = alpha0[b][j][c] = rows[b][c][j] (z = L^-1 r after the factorisation; alpha0 is solved in place)
template <typename T>
__global__ void k_layer_z(const T* __restrict__ rows, int64_t ldr, int64_t srows, int64_t n, int q, T* __restrict__ z,
                          T* __restrict__ alpha, T* __restrict__ work, int64_t swork)
{
    const int b = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * q) return;
    const int64_t j = e / q;
    const int c = (int)(e - j * q);
    const T v = rows[(int64_t)b * srows + (int64_t)c * ldr + j];
    z[(int64_t)b * n * q + e] = v;
    alpha[(int64_t)b * n * q + e] = v;
    // the backward solve's running right-hand side (q x n per problem, potrs_run: `work`), so that it need not transpose alpha first
    if (work) work[(int64_t)b * swork + (int64_t)c * n + j] = v;
}

// train_out += K_noiseless alpha + bias = (y - f_bar - bias) - noise alpha + bias = y - f_bar - noise alpha
// (no second pass over the Gram matrix: K alpha = r - noise alpha)
template <typename T>
__global__ void k_layer_train_mean(const T* __restrict__ y, const T* __restrict__ fbar, const int64_t* __restrict__ starts,
                                   int64_t n, int q, const T* __restrict__ alpha, const T* __restrict__ noise,
                                   T* __restrict__ out)
{
    const int b = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * q) return;
    const int64_t g = starts[b] * q + e;
    out[g] += y[g] - (fbar ? fbar[g] : (T)0) - noise[b] * alpha[(int64_t)b * n * q + e];
}

}  // namespace

template <typename T>
int layer_fit_run(const LayerFit<T>& a, hipStream_t st)
{
    const char* fn = "cimrgp_layer_fit";
    CIMRGP_REQUIRE(a.batch >= 1 && a.batch < 65536, fn, "batch count out of range");
    CIMRGP_REQUIRE(a.n > 0, fn, "empty blocks");
    CIMRGP_REQUIRE(a.q >= 1 && a.q <= LY_MAXQ, fn, "number of outputs must be in [1, 8]");
    const unsigned nb = (unsigned)a.batch;
    const unsigned ge = (unsigned)((a.n * a.q + 255) / 256);
    hipLaunchKernelGGL((k_layer_stats<T>), dim3(nb), dim3(1024), 0, st, a.y, a.fbar, a.starts, a.n, a.q,
                       (T)a.noise_fixed, (T)a.noise_frac, (T)a.noise_floor, a.shared_bias, a.shared_noise, a.bias, a.noise);
    CIMRGP_LAUNCH_CHECK(fn);
    hipLaunchKernelGGL((k_layer_rows<T>), dim3(ge, nb), dim3(256), 0, st, a.y, a.fbar, a.starts, a.n, a.q,
                       (const T*)a.bias, a.rows, a.ldr, a.srows);
    CIMRGP_LAUNCH_CHECK(fn);
    int rc = rbf_gram_batched_run<T>(a.x, a.starts, a.n, a.x, a.starts, a.n, a.d, a.ell, a.sf2, (const T*)a.noise, a.k, a.ldk,
                                     a.sk, a.batch, true, st);
    if (rc) return rc;
    PotrfBatch bt;
    bt.count = a.batch;
    bt.sk = a.sk;
    bt.sws = a.sws;
    bt.sb = a.srows;
    rc = potrf_batched_run<T>(a.k, a.n, a.ldk, a.ws, a.info, a.rows, a.q, a.ldr, bt, st);
    if (rc) return rc;
    hipLaunchKernelGGL((k_layer_z<T>), dim3(ge, nb), dim3(256), 0, st, (const T*)a.rows, a.ldr, a.srows, a.n, a.q, a.z, a.alpha,
                       a.scratch, 2 * (int64_t)a.q * a.n);
    CIMRGP_LAUNCH_CHECK(fn);
    rc = potrs_run<T>(a.k, a.n, a.ldk, a.ws, a.alpha, a.q, nullptr, a.scratch, true, st, bt, true);
    if (rc) return rc;
    hipLaunchKernelGGL((k_layer_train_mean<T>), dim3(ge, nb), dim3(256), 0, st, a.y, a.fbar, a.starts, a.n, a.q,
                       (const T*)a.alpha, (const T*)a.noise, a.train_out);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int layer_predict_run(const LayerPredict<T>& a, hipStream_t st)
{
    const char* fn = "cimrgp_layer_predict";
    CIMRGP_REQUIRE(a.batch >= 1 && a.batch < 65536, fn, "batch count out of range");
    CIMRGP_REQUIRE(a.q >= 1 && a.q <= LY_MAXQ, fn, "number of outputs must be in [1, 8]");
    if (a.ns <= 0 || a.n <= 0) return 0;
    CIMRGP_REQUIRE(a.ldw >= a.n, fn, "leading dimension of W smaller than n");
    // W_b = K(xs_b, x_b)
    int rc = rbf_gram_batched_run<T>(a.xs, a.t_starts, a.ns, a.x, a.starts, a.n, a.d, a.ell, a.sf2, (const T*)nullptr, a.w, a.ldw,
                                     a.sw, a.batch, false, st);
    if (rc) return rc;
    // W_b <- W_b L_b^-T
    PotrfBatch bt;
    bt.count = a.batch;
    bt.sk = a.sl;
    bt.sws = a.sws;
    bt.sb = a.sw;
    rc = solve_rows_run<T>(a.l, a.n, a.ldl, a.ws, a.w, a.ns, a.ldw, st, bt);
    if (rc) return rc;
    // mean += W z + bias, var += sf - sum W^2 (+ noise)
    return predict_from_w_run<T>((const T*)a.w, a.ns, a.n, a.ldw, a.z, a.q, a.sf2, 0.0, a.noise, a.bias, a.mean, a.var, 1, st,
                                 a.batch, a.t_starts, a.sw);
}

// The targets of ONE block as carried rows and back (cimrgp_block_posterior).
template <typename T>
__global__ void k_rhs_rows(const T* __restrict__ y, int64_t n, int q, T* __restrict__ rows, int64_t ldr)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * q) return;
    const int64_t j = e / q;
    const int c = (int)(e - j * q);
    rows[(int64_t)c * ldr + j] = y[e];
}

template <typename T>
int rhs_rows_run(const T* y, int64_t n, int q, T* rows, int64_t ldr, hipStream_t st)
{
    if (n <= 0 || q <= 0) return 0;
    hipLaunchKernelGGL((k_rhs_rows<T>), dim3((unsigned)((n * q + 255) / 256)), dim3(256), 0, st, y, n, q, rows, ldr);
    CIMRGP_LAUNCH_CHECK("cimrgp_block_posterior");
    return 0;
}

template <typename T>
int rows_to_z_run(const T* rows, int64_t ldr, int64_t n, int q, T* z, T* alpha, hipStream_t st, T* work)
{
    if (n <= 0 || q <= 0) return 0;
    hipLaunchKernelGGL((k_layer_z<T>), dim3((unsigned)((n * q + 255) / 256), 1), dim3(256), 0, st, rows, ldr, (int64_t)0, n, q, z, alpha,
                       work, (int64_t)0);
    CIMRGP_LAUNCH_CHECK("cimrgp_block_posterior");
    return 0;
}

template int rhs_rows_run<double>(const double*, int64_t, int, double*, int64_t, hipStream_t);
template int rhs_rows_run<float>(const float*, int64_t, int, float*, int64_t, hipStream_t);
template int rows_to_z_run<double>(const double*, int64_t, int64_t, int, double*, double*, hipStream_t, double*);
template int rows_to_z_run<float>(const float*, int64_t, int64_t, int, float*, float*, hipStream_t, float*);
template int layer_fit_run<double>(const LayerFit<double>&, hipStream_t);
template int layer_fit_run<float>(const LayerFit<float>&, hipStream_t);
template int layer_predict_run<double>(const LayerPredict<double>&, hipStream_t);
template int layer_predict_run<float>(const LayerPredict<float>&, hipStream_t);

}  // namespace cimrgp
