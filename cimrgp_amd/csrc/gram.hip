// D1: tiled RBF Gram / cross-Gram builder and D4: fused predictive mean.
//
// Gram: one workgroup = one 64 x 64 tile; the row and column input tiles are
// staged in LDS once; each thread produces a 4 x 4 patch whose 4 columns are
// contiguous, so a 16-lane row group stores 16 x 32 B (f64) = 512 contiguous
// bytes per matrix row.  HBM-write bound: n^2 * sizeof(T) bytes,
// with the exp as the secondary limiter in f64.
#include "common.hpp"

namespace cimrgp {

namespace {

constexpr int GTILE = 64;
constexpr int MAXD  = 8;

// D = compile-time input dimension (1, 2) or 0 = run-time d <= 8 with fully
// unrolled, predicated loops (run-time indexed register arrays would spill).
template <typename T, bool SYMM, int D>
static __device__ __forceinline__ void gram_tile(const T* __restrict__ xa, int na, const T* __restrict__ xb, int nb, int d,
                                                  T neg_half_inv_l2, T sf2, T diag_add, T* __restrict__ K, int64_t ld,
                                                  int tiles_n, int lower_only)
{
    __shared__ T sa[GTILE * MAXD];
    __shared__ T sb[GTILE * MAXD];
    int ti, tj;
    if (SYMM && lower_only) {
        const int id = blockIdx.x;
        ti = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
        while (ti * (ti + 1) / 2 > id) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
        tj = id - ti * (ti + 1) / 2;
    } else {
        ti = blockIdx.x / tiles_n;
        tj = blockIdx.x - ti * tiles_n;
    }
    const int row0 = ti * GTILE, col0 = tj * GTILE;
    const int tid = threadIdx.x;
    for (int e = tid; e < GTILE * MAXD; e += 256) {
        const int r = e / MAXD, k = e - r * MAXD;
        sa[e] = (k < d && row0 + r < na) ? xa[(int64_t)(row0 + r) * d + k] : (T)0;
        sb[e] = (k < d && col0 + r < nb) ? xb[(int64_t)(col0 + r) * d + k] : (T)0;
    }
    __syncthreads();

    // Store layout (round 3): a lane owns the 16 bytes it stores with ONE instruction -- EPL = 2 doubles / 4 floats
    // of one row -- and the lanes of a wave are adjacent along the row: every store instruction writes whole
    // 512-byte runs (2 rows x 32 lanes in FP64, 4 rows x 16 lanes in FP32).  (Rounds 1-2: 4 adjacent columns =
    // 32 bytes per lane in two instructions, each of which wrote every other 16 bytes of its run: 3.5 TB/s with
    // the exponential taken out, against 5.8 TB/s for a plain fill of the same bytes.)
    constexpr int EPL = 16 / (int)sizeof(T);    // elements per lane and row
    constexpr int LPR = GTILE / EPL;            // lanes per tile row
    constexpr int RPI = 64 / LPR;               // rows per wave and store instruction
    constexpr int NIT = GTILE / (4 * RPI);      // row iterations: 4 waves x RPI rows each
    const int lane = tid & 63, wave = tid >> 6;
    const int cx = (lane % LPR) * EPL;          // first column of this lane inside the tile
    const int ry = lane / LPR;                  // row of this lane inside a wave's row group
    constexpr int DD = D ? D : MAXD;
    T xc[EPL][DD];
#pragma unroll
    for (int b = 0; b < EPL; ++b)
#pragma unroll
        for (int k = 0; k < DD; ++k) xc[b][k] = sb[(cx + b) * MAXD + k];

#pragma unroll
    for (int a = 0; a < NIT; ++a) {
        const int r  = (a * 4 + wave) * RPI + ry;
        const int gr = row0 + r;
        if (gr >= na) continue;
        T out[EPL];
#pragma unroll
        for (int b = 0; b < EPL; ++b) {
            T d2 = (T)0;
#pragma unroll
            for (int k = 0; k < DD; ++k) {
                if (D || k < d) {
                    const T df = sa[r * MAXD + k] - xc[b][k];
                    d2 += df * df;
                }
            }
            T v = sf2 * exp(d2 * neg_half_inv_l2);
            if (SYMM && (gr == col0 + cx + b)) v += diag_add;
            out[b] = v;
        }
        const int gc = col0 + cx;
        T* dst = K + (int64_t)gr * ld + gc;
        if (gc + EPL - 1 < nb && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            // non-temporal: the matrix is written once and read next by another kernel (round 4: 65.5 -> 59-61 us at
            // n = 8192, lower tiles)
            typedef double d2v __attribute__((ext_vector_type(2)));
            typedef float f4v __attribute__((ext_vector_type(4)));
            if (sizeof(T) == 8) { d2v v = {(double)out[0], (double)out[1]}; __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(dst)); }
            else { f4v v = {(float)out[0], (float)out[1], (float)out[EPL - 2], (float)out[EPL - 1]}; __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(dst)); }
        } else {
#pragma unroll
            for (int b = 0; b < EPL; ++b)
                if (gc + b < nb) dst[b] = out[b];
        }
    }
}

template <typename T, bool SYMM, int D>
__global__ __launch_bounds__(256)
void k_rbf_gram(const T* __restrict__ xa, int na, const T* __restrict__ xb, int nb, int d,
                T neg_half_inv_l2, T sf2, T diag_add, T* __restrict__ K, int64_t ld,
                int tiles_n, int lower_only)
{
    gram_tile<T, SYMM, D>(xa, na, xb, nb, d, neg_half_inv_l2, sf2, diag_add, K, ld, tiles_n, lower_only);
}

// The lower triangle of a symmetric Gram matrix in 64 x 128 tiles (round 5): a tile row is 128 columns = 1 KB in FP64, so
// every store instruction of a wave writes ONE whole 1-KB run of one matrix row (FP32: two 512-byte runs); tile rows 2p
// and 2p+1 both need p + 1 tiles, so pair p starts at tile p (p + 1).  In the tile that holds the diagonal, lanes whose
// columns lie right of the tile's last row have nothing below the diagonal and skip.
constexpr int GWIDE = 128;
template <typename T, int D>
__global__ __launch_bounds__(256)
void k_rbf_gram_lower_wide(const T* __restrict__ x, int n, int d, T neg_half_inv_l2, T sf2, T diag_add, T* __restrict__ K, int64_t ld)
{
    __shared__ T sa[GTILE * MAXD];
    __shared__ T sb[GWIDE * MAXD];
    const int id = blockIdx.x;
    int p = (int)((sqrtf(4.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while (p * (p + 1) > id) --p;
    while ((p + 1) * (p + 2) <= id) ++p;
    const int rem = id - p * (p + 1);
    const int odd = rem >= p + 1;
    const int ti = 2 * p + odd, tj = rem - odd * (p + 1);
    const int row0 = ti * GTILE, col0 = tj * GWIDE;
    const int tid = threadIdx.x;
    for (int e = tid; e < GWIDE * MAXD; e += 256) {
        const int r = e / MAXD, k = e - r * MAXD;
        if (r < GTILE) sa[e] = (k < d && row0 + r < n) ? x[(int64_t)(row0 + r) * d + k] : (T)0;
        sb[e] = (k < d && col0 + r < n) ? x[(int64_t)(col0 + r) * d + k] : (T)0;
    }
    __syncthreads();
    constexpr int EPL = 16 / (int)sizeof(T);    // elements per lane and row
    constexpr int LPR = GWIDE / EPL;            // lanes per tile row: 64 (FP64) / 32 (FP32)
    constexpr int RPI = 64 / LPR;               // rows per wave and store instruction
    constexpr int NIT = GTILE / (4 * RPI);
    const int lane = tid & 63, wave = tid >> 6;
    const int cx = (lane % LPR) * EPL;
    const int ry = lane / LPR;
    const int gc = col0 + cx;
    if (gc > row0 + GTILE - 1) return;          // (behind the only barrier) all of this lane's columns are above the diagonal in every row of the tile
    constexpr int DD = D ? D : MAXD;
    T xc[EPL][DD];
#pragma unroll
    for (int b = 0; b < EPL; ++b)
#pragma unroll
        for (int k = 0; k < DD; ++k) xc[b][k] = sb[(cx + b) * MAXD + k];
#pragma unroll
    for (int a = 0; a < NIT; ++a) {
        const int r = (a * 4 + wave) * RPI + ry;
        const int gr = row0 + r;
        if (gr >= n) continue;
        T out[EPL];
#pragma unroll
        for (int b = 0; b < EPL; ++b) {
            T d2 = (T)0;
#pragma unroll
            for (int k = 0; k < DD; ++k) {
                if (D || k < d) {
                    const T df = sa[r * MAXD + k] - xc[b][k];
                    d2 += df * df;
                }
            }
            T v = sf2 * exp(d2 * neg_half_inv_l2);
            if (gr == gc + b) v += diag_add;
            out[b] = v;
        }
        T* dst = K + (int64_t)gr * ld + gc;
        if (gc + EPL - 1 < n && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            typedef double d2v __attribute__((ext_vector_type(2)));
            typedef float f4v __attribute__((ext_vector_type(4)));
            if (sizeof(T) == 8) { d2v v = {(double)out[0], (double)out[1]}; __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(dst)); }
            else { f4v v = {(float)out[0], (float)out[1], (float)out[EPL - 2], (float)out[EPL - 1]}; __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(dst)); }
        } else {
#pragma unroll
            for (int b = 0; b < EPL; ++b)
                if (gc + b < n) dst[b] = out[b];
        }
    }
}

// The blocks of one layer in one launch (blockIdx.y = block): block b takes its `na` rows of inputs at
// row a_starts[b] of xa and its `nb` columns at row b_starts[b] of xb (regions are contiguous ranges of
// the layer's arrays, Inputs.py:57-60), writes matrix b of the arena (stride kstride) and, on the
// diagonal of a symmetric matrix, adds ITS noise (diag_dev[b]).
template <typename T, bool SYMM, int D>
__global__ __launch_bounds__(256)
void k_rbf_gram_batched(const T* __restrict__ xa, const int64_t* __restrict__ a_starts, int na,
                        const T* __restrict__ xb, const int64_t* __restrict__ b_starts, int nb, int d,
                        T neg_half_inv_l2, T sf2, const T* __restrict__ diag_dev, T* __restrict__ K, int64_t ld,
                        int64_t kstride, int tiles_n, int lower_only)
{
    const int b = blockIdx.y;
    gram_tile<T, SYMM, D>(xa + a_starts[b] * d, na, xb + b_starts[b] * d, nb, d, neg_half_inv_l2, sf2,
                          diag_dev ? diag_dev[b] : (T)0, K + (int64_t)b * kstride, ld, tiles_n, lower_only);
}

// D4: mean[i][c] (+)= bias[c] + sum_j k(xs_i, x_j) alpha[j][c].
// Workgroup = 8 test points x 32 partial sums over the training points; the
// cross-Gram row is never written anywhere.  Deterministic (fixed-order LDS
// reduction, no atomics).
constexpr int PM_TS = 8;
constexpr int PM_PH = 32;
constexpr int MAXQ  = 8;

// Q = number of outputs at compile time (a run-time guard inside the loop serialises the loads).
template <typename T, int D, int Q>
__global__ __launch_bounds__(256)
void k_predict_mean(const T* __restrict__ x, int n, int d, const T* __restrict__ alpha, int q,
                    const T* __restrict__ xs, int ns, T neg_half_inv_l2, T sf2,
                    const T* __restrict__ bias, T* __restrict__ mean, int accumulate)
{
    __shared__ T red[PM_PH][PM_TS][Q];
    const int tid = threadIdx.x;
    const int t  = tid & (PM_TS - 1);
    const int ph = tid / PM_TS;
    const int gi = blockIdx.x * PM_TS + t;
    constexpr int DD = D ? D : MAXD;
    T xt[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) xt[k] = ((D || k < d) && gi < ns) ? xs[(int64_t)gi * d + k] : (T)0;
    T sum[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) sum[c] = (T)0;
    for (int j = ph; j < n; j += PM_PH) {
        T d2 = (T)0;
#pragma unroll
        for (int k = 0; k < DD; ++k) {
            if (D || k < d) {
                const T df = xt[k] - x[(int64_t)j * d + k];
                d2 += df * df;
            }
        }
        const T kv = sf2 * exp(d2 * neg_half_inv_l2);
#pragma unroll
        for (int c = 0; c < Q; ++c) sum[c] += kv * alpha[(int64_t)j * Q + c];
    }
#pragma unroll
    for (int c = 0; c < Q; ++c) red[ph][t][c] = sum[c];
    __syncthreads();
    if (tid < PM_TS * q) {
        const int tt = tid / q, c = tid - tt * q;
        const int g = blockIdx.x * PM_TS + tt;
        if (g < ns) {
            T s = bias ? bias[c] : (T)0;
            for (int p = 0; p < PM_PH; ++p) s += red[p][tt][c];
            T* o = mean + (int64_t)g * q + c;
            *o = accumulate ? (*o + s) : s;
        }
    }
}

}  // namespace

template <typename T>
int rbf_gram_run(const T* xa, int64_t na, const T* xb, int64_t nb, int d, double ell, double sf2,
                 double diag_add, T* k, int64_t ld, bool symm, bool lower_only, hipStream_t st)
{
    const char* fn = symm ? "cimrgp_rbf_gram" : "cimrgp_rbf_cross";
    if (na <= 0 || nb <= 0) return 0;
    CIMRGP_REQUIRE(d >= 1 && d <= MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(ell > 0.0, fn, "length-scale must be positive");
    CIMRGP_REQUIRE(ld >= nb, fn, "leading dimension smaller than the number of columns");
    CIMRGP_REQUIRE(na < (1ll << 30) && nb < (1ll << 30), fn, "matrix too large");
    const int64_t tm = (na + GTILE - 1) / GTILE, tn = (nb + GTILE - 1) / GTILE;
    const T c = (T)(-0.5 / (ell * ell));
#define CIMRGP_GRAM_LAUNCH(SYMM_, D_, tiles_, diag_, lo_)                                   \
    hipLaunchKernelGGL((k_rbf_gram<T, SYMM_, D_>), dim3((unsigned)(tiles_)), dim3(256), 0, st, \
                       xa, (int)na, xb, (int)nb, d, c, (T)sf2, (T)(diag_), k, ld, (int)tn, (lo_))
    if (symm && lower_only && na == nb && xa == xb) {
        // the lower triangle in 64 x 128 tiles: pairs of tile rows, P (P + 1) tiles in the full pairs (+ P + 1 for an odd last row)
        const int64_t pairs = tm / 2;
        const int64_t tiles = pairs * (pairs + 1) + ((tm & 1) ? pairs + 1 : 0);
        CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
#define CIMRGP_GRAMW_LAUNCH(D_) hipLaunchKernelGGL((k_rbf_gram_lower_wide<T, D_>), dim3((unsigned)tiles), dim3(256), 0, st, \
                                                   xa, (int)na, d, c, (T)sf2, (T)diag_add, k, ld)
        if (d == 1)      CIMRGP_GRAMW_LAUNCH(1);
        else if (d == 2) CIMRGP_GRAMW_LAUNCH(2);
        else             CIMRGP_GRAMW_LAUNCH(0);
#undef CIMRGP_GRAMW_LAUNCH
    } else if (symm) {
        const int64_t tiles = lower_only ? tm * (tm + 1) / 2 : tm * tn;
        CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
        const int lo = lower_only ? 1 : 0;
        if (d == 1)      CIMRGP_GRAM_LAUNCH(true, 1, tiles, diag_add, lo);
        else if (d == 2) CIMRGP_GRAM_LAUNCH(true, 2, tiles, diag_add, lo);
        else             CIMRGP_GRAM_LAUNCH(true, 0, tiles, diag_add, lo);
    } else {
        const int64_t tiles = tm * tn;
        CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
        if (d == 1)      CIMRGP_GRAM_LAUNCH(false, 1, tiles, 0, 0);
        else if (d == 2) CIMRGP_GRAM_LAUNCH(false, 2, tiles, 0, 0);
        else             CIMRGP_GRAM_LAUNCH(false, 0, tiles, 0, 0);
    }
#undef CIMRGP_GRAM_LAUNCH
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int rbf_gram_batched_run(const T* xa, const int64_t* a_starts, int64_t na, const T* xb, const int64_t* b_starts, int64_t nb,
                         int d, double ell, double sf2, const T* diag_dev, T* k, int64_t ld, int64_t kstride, int batch,
                         bool symm, hipStream_t st)
{
    const char* fn = "cimrgp_layer";
    if (na <= 0 || nb <= 0 || batch <= 0) return 0;
    CIMRGP_REQUIRE(d >= 1 && d <= MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(ell > 0.0, fn, "length-scale must be positive");
    CIMRGP_REQUIRE(ld >= nb, fn, "leading dimension smaller than the number of columns");
    CIMRGP_REQUIRE(na < (1ll << 30) && nb < (1ll << 30) && batch < 65536, fn, "batch too large");
    const int64_t tm = (na + GTILE - 1) / GTILE, tn = (nb + GTILE - 1) / GTILE;
    const int64_t tiles = symm ? tm * (tm + 1) / 2 : tm * tn;
    CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
    const T c = (T)(-0.5 / (ell * ell));
#define CIMRGP_GRAMB_LAUNCH(SYMM_, D_)                                                                      \
    hipLaunchKernelGGL((k_rbf_gram_batched<T, SYMM_, D_>), dim3((unsigned)tiles, (unsigned)batch), dim3(256), 0, st, \
                       xa, a_starts, (int)na, xb, b_starts, (int)nb, d, c, (T)sf2, diag_dev, k, ld, kstride, (int)tn, symm ? 1 : 0)
    if (symm) { if (d == 1) CIMRGP_GRAMB_LAUNCH(true, 1); else if (d == 2) CIMRGP_GRAMB_LAUNCH(true, 2); else CIMRGP_GRAMB_LAUNCH(true, 0); }
    else      { if (d == 1) CIMRGP_GRAMB_LAUNCH(false, 1); else if (d == 2) CIMRGP_GRAMB_LAUNCH(false, 2); else CIMRGP_GRAMB_LAUNCH(false, 0); }
#undef CIMRGP_GRAMB_LAUNCH
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int predict_mean_run(const T* x, int64_t n, int d, const T* alpha, int q, const T* xs, int64_t ns,
                     double ell, double sf2, const T* bias, T* mean, int accumulate, hipStream_t st)
{
    const char* fn = "cimrgp_predict_mean";
    if (ns <= 0) return 0;
    CIMRGP_REQUIRE(d >= 1 && d <= MAXD, fn, "input dimension must be in [1, 8]");
    CIMRGP_REQUIRE(q >= 1 && q <= MAXQ, fn, "number of outputs must be in [1, 8]");
    CIMRGP_REQUIRE(ell > 0.0, fn, "length-scale must be positive");
    CIMRGP_REQUIRE(n < (1ll << 31) && ns < (1ll << 31), fn, "too many points");
    const unsigned grid = (unsigned)((ns + PM_TS - 1) / PM_TS);
#define CIMRGP_PM_LAUNCH(D_, Q_)                                                              \
    hipLaunchKernelGGL((k_predict_mean<T, D_, Q_>), dim3(grid), dim3(256), 0, st, x, (int)n, d, \
                       alpha, q, xs, (int)ns, (T)(-0.5 / (ell * ell)), (T)sf2, bias, mean, accumulate)
#define CIMRGP_PM_D(Q_)                                     \
    { if (d == 1)      CIMRGP_PM_LAUNCH(1, Q_);             \
      else if (d == 2) CIMRGP_PM_LAUNCH(2, Q_);             \
      else             CIMRGP_PM_LAUNCH(0, Q_); }
    switch (q) {
        case 1: CIMRGP_PM_D(1); break;
        case 2: CIMRGP_PM_D(2); break;
        case 3: CIMRGP_PM_D(3); break;
        case 4: CIMRGP_PM_D(4); break;
        case 5: CIMRGP_PM_D(5); break;
        case 6: CIMRGP_PM_D(6); break;
        case 7: CIMRGP_PM_D(7); break;
        default: CIMRGP_PM_D(8); break;
    }
#undef CIMRGP_PM_D
#undef CIMRGP_PM_LAUNCH
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template int rbf_gram_run<double>(const double*, int64_t, const double*, int64_t, int, double, double, double,
                                  double*, int64_t, bool, bool, hipStream_t);
template int rbf_gram_run<float>(const float*, int64_t, const float*, int64_t, int, double, double, double,
                                 float*, int64_t, bool, bool, hipStream_t);
template int rbf_gram_batched_run<double>(const double*, const int64_t*, int64_t, const double*, const int64_t*, int64_t, int,
                                          double, double, const double*, double*, int64_t, int64_t, int, bool, hipStream_t);
template int rbf_gram_batched_run<float>(const float*, const int64_t*, int64_t, const float*, const int64_t*, int64_t, int,
                                         double, double, const float*, float*, int64_t, int64_t, int, bool, hipStream_t);
template int predict_mean_run<double>(const double*, int64_t, int, const double*, int, const double*, int64_t,
                                      double, double, const double*, double*, int, hipStream_t);
template int predict_mean_run<float>(const float*, int64_t, int, const float*, int, const float*, int64_t,
                                     double, double, const float*, float*, int, hipStream_t);

}  // namespace cimrgp
