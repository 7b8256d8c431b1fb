// D3: alpha = (L L^T)^-1 R for a few right-hand sides (q <= 8), and the D5
// tail (row reductions over W = K(X*,X) L^-T).
//
// The skinny solves walk the 256-column panels: per panel ONE launch in which
// every workgroup first solves the 256x256 diagonal block redundantly (four
// sub-steps using the 64x64 inverses potrf left in the workspace, all from L2)
// and then subtracts its slice of  L[rows below, panel] z_p  (forward) or
// L[panel, columns left]^T a_p  (backward) from the running right-hand side.
// Right-hand sides are kept "RHS-major" (q x n) so the updates are coalesced.
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

constexpr int PW = CIMRGP_NB;      // panel width of the skinny solves (256)

// Sum the 4 per-wave partials held in LDS part[4][MAXQ][64] into one value per (c, t).
// ---------------------------------------------------------------------------
// Forward panel step:  z_p = L_pp^-1 w_p ;  w[rows below the panel] -= L[rows, panel] z_p.
// Every workgroup first solves the 256-wide diagonal block redundantly (four
// 64-wide sub-steps: row-dot with the already solved part, then a product with
// the stored 64x64 inverse; everything comes from L2), workgroup 0 publishes it,
// then each workgroup updates its own slice of rows (16 lanes per row, 4
// elements per lane, coalesced 16-byte loads, all loads of a pass in flight).
// ---------------------------------------------------------------------------
// dst[c][lr] (-)= sum_k M[lr][k] x[c][k]  for lr < nrows (<= 64), k < kw (<= 256): 16 lanes per
// row, 4 contiguous elements per lane and 64-column segment, 16 rows per pass, every load of a
// pass independent of the others (memory-level parallelism instead of a dependent chain).
template <typename T, bool ASSIGN>
static __device__ __forceinline__ void rowdot64(const T* __restrict__ M, int64_t ldm, int nrows, int kw,
                                                 const T (*x)[CIMRGP_NB], int xoff, T (*dst)[CIMRGP_NB], int doff, int q)
{
    const int l16 = threadIdx.x & 15, slot = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int lr = slot + 16 * pass;
        T sum[MAXQ];
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) sum[c] = (T)0;
        if (lr < nrows) {
            const T* mp = M + (int64_t)lr * ldm;
#pragma unroll
            for (int seg = 0; seg < CIMRGP_NB / 64; ++seg) {
                const int kk = seg * 64 + l16 * 4;
                if (seg * 64 < kw) {
                    T mv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) mv[e] = (kk + e < kw) ? mp[kk + e] : (T)0;
#pragma unroll
                    for (int c = 0; c < MAXQ; ++c)
                        if (c < q) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) sum[c] += mv[e] * x[c][xoff + kk + e];
                        }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) {
            if (c < q) {
                T sv = sum[c];
                sv += __shfl_xor(sv, 8, 16);
                sv += __shfl_xor(sv, 4, 16);
                sv += __shfl_xor(sv, 2, 16);
                sv += __shfl_xor(sv, 1, 16);
                if (l16 == 0 && lr < nrows) {
                    if (ASSIGN) dst[c][doff + lr] = sv;
                    else        dst[c][doff + lr] -= sv;
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void k_fwd_panel(const T* __restrict__ L, int64_t ld, int n, const T* __restrict__ inv64,
                 T* __restrict__ work, T* __restrict__ out, int q, int k0, int w, int rows_per_wg)
{
    __shared__ T zs[MAXQ][PW];          // running right-hand side of the panel, then the solution
    __shared__ T tmp[MAXQ][PW];
    const int tid = threadIdx.x;
    for (int e = tid; e < q * PW; e += 256) {
        const int c = e / PW, u = e - c * PW;
        zs[c][u] = (u < w) ? work[(int64_t)c * n + k0 + u] : (T)0;
    }
    __syncthreads();
    const int nsub = (w + SB - 1) / SB;
    for (int s = 0; s < nsub; ++s) {
        const int c0 = SB * s;
        const int sw = min(SB, w - c0);
        // rhs_s = zs_s - L[s-block rows, panel cols < c0] z[< c0]
        if (c0 > 0) {
            rowdot64<T, false>(L + (int64_t)(k0 + c0) * ld + k0, ld, sw, c0, zs, 0, zs, c0, q);
            __syncthreads();
        }
        // z_s = I_s rhs_s   (through tmp: every row needs the whole rhs_s)
        rowdot64<T, true>(inv64 + (int64_t)((k0 + c0) / SB) * (SB * SB), SB, sw, SB, zs, c0, tmp, c0, q);
        __syncthreads();
        for (int e = tid; e < q * SB; e += 256) {
            const int c = e >> 6, u = e & 63;
            const T v = (u < sw) ? tmp[c][c0 + u] : (T)0;
            zs[c][c0 + u] = v;
            if (blockIdx.x == 0 && u < sw) out[(int64_t)c * n + k0 + c0 + u] = v;
        }
        __syncthreads();
    }
    // rows below the panel
    const int l16 = tid & 15, slot = tid >> 4;
    const int rbeg = k0 + w + blockIdx.x * rows_per_wg;
    const int rend = min(n, rbeg + rows_per_wg);
#pragma unroll 4
    for (int r = rbeg + slot; r < rend; r += 16) {
        const T* lp = L + (int64_t)r * ld + k0;
        T sum[MAXQ];
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) sum[c] = (T)0;
#pragma unroll
        for (int seg = 0; seg < PW / 64; ++seg) {
            const int kk = seg * 64 + l16 * 4;
            T lv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) lv[e] = (kk + e < w) ? lp[kk + e] : (T)0;
#pragma unroll
            for (int c = 0; c < MAXQ; ++c)
                if (c < q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[c] += lv[e] * zs[c][kk + e];
                }
        }
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) {
            if (c < q) {
                T sv = sum[c];
                sv += __shfl_xor(sv, 8, 16);
                sv += __shfl_xor(sv, 4, 16);
                sv += __shfl_xor(sv, 2, 16);
                sv += __shfl_xor(sv, 1, 16);
                if (l16 == 0) work[(int64_t)c * n + r] -= sv;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Backward panel step:  a_p = L_pp^-T w_p ;  w[cols left of the panel] -= L[panel, cols]^T a_p.
// Column-sum form throughout (thread = column, coalesced along the row).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void k_bwd_panel(const T* __restrict__ L, int64_t ld, int n, const T* __restrict__ inv64,
                 T* __restrict__ work, T* __restrict__ out, int q, int k0, int w)
{
    __shared__ T zs[MAXQ][PW];
    __shared__ T part[4][MAXQ][SB];
    const int tid = threadIdx.x;
    const int t = tid & 63, pr = tid >> 6;
    for (int e = tid; e < q * PW; e += 256) {
        const int c = e / PW, u = e - c * PW;
        zs[c][u] = (u < w) ? work[(int64_t)c * n + k0 + u] : (T)0;
    }
    __syncthreads();
    const int nsub = (w + SB - 1) / SB;
    for (int s = nsub - 1; s >= 0; --s) {
        const int c0 = SB * s;
        const int sw = min(SB, w - c0);
        const int hi = c0 + sw;                       // solved part of the panel: [hi, w)
        // (1) rhs_s[t] = zs[c0+t] - sum_{u in [hi, w)} L[k0+u][k0+c0+t] a[u]
        T acc[MAXQ];
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) acc[c] = (T)0;
        if (t < sw) {
            const T* lp = L + (int64_t)k0 * ld + k0 + c0 + t;
#pragma unroll 8
            for (int u = hi + pr; u < w; u += 4) {
                const T lv = lp[(int64_t)u * ld];
#pragma unroll
                for (int c = 0; c < MAXQ; ++c)
                    if (c < q) acc[c] += lv * zs[c][u];
            }
        }
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) part[pr][c][t] = acc[c];
        __syncthreads();
        if (pr == 0) {
#pragma unroll
            for (int c = 0; c < MAXQ; ++c)
                if (c < q) zs[c][c0 + t] -= part[0][c][t] + part[1][c][t] + part[2][c][t] + part[3][c][t];
        }
        __syncthreads();
        // (2) a_s[t] = sum_{u >= t} I_s[u][t] rhs_s[u]
        const T* ip = inv64 + (int64_t)((k0 + c0) / SB) * (SB * SB) + t;
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) acc[c] = (T)0;
#pragma unroll 8
        for (int u = t + pr; u < sw; u += 4) {
            const T iv = ip[u * SB];
#pragma unroll
            for (int c = 0; c < MAXQ; ++c)
                if (c < q) acc[c] += iv * zs[c][c0 + u];
        }
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) part[pr][c][t] = acc[c];
        __syncthreads();
        if (pr == 0) {
#pragma unroll
            for (int c = 0; c < MAXQ; ++c) {
                if (c < q) {
                    const T v = part[0][c][t] + part[1][c][t] + part[2][c][t] + part[3][c][t];
                    zs[c][c0 + t] = v;
                    if (blockIdx.x == 0 && t < sw) out[(int64_t)c * n + k0 + c0 + t] = v;
                }
            }
        }
        __syncthreads();
    }
    // columns left of the panel: thread = column, 4-way split over the panel's rows
    const int col = blockIdx.x * SB + t;
    T acc[MAXQ];
#pragma unroll
    for (int c = 0; c < MAXQ; ++c) acc[c] = (T)0;
    if (col < k0) {
        const T* lp = L + (int64_t)k0 * ld + col;
        const int ubeg = pr * SB, uend = min(w, ubeg + SB);
#pragma unroll 8
        for (int u = ubeg; u < uend; ++u) {
            const T lv = lp[(int64_t)u * ld];
#pragma unroll
            for (int c = 0; c < MAXQ; ++c)
                if (c < q) acc[c] += lv * zs[c][u];
        }
    }
#pragma unroll
    for (int c = 0; c < MAXQ; ++c) part[pr][c][t] = acc[c];
    __syncthreads();
    if (pr == 0 && col < k0) {
#pragma unroll
        for (int c = 0; c < MAXQ; ++c)
            if (c < q) work[(int64_t)c * n + col] -= part[0][c][t] + part[1][c][t] + part[2][c][t] + part[3][c][t];
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
    const int rows_per_wg = 64;
    for (int64_t k0 = 0; k0 < n; k0 += PW) {
        const int w = (int)((n - k0 < PW) ? (n - k0) : PW);
        const int64_t below = n - (k0 + w);
        const unsigned grid = (unsigned)((below + rows_per_wg - 1) / rows_per_wg);
        hipLaunchKernelGGL((k_fwd_panel<T>), dim3(grid ? grid : 1), dim3(256), 0, st, l, ld, (int)n,
                           ws, work, res, q, (int)k0, w, rows_per_wg);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    if (z_out) {
        hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg), dim3(256), 0, st, (const T*)res, z_out, n, q, 0);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    hipError_t e = hipMemcpyAsync(work, res, sizeof(T) * (size_t)(q * n), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return check_hip(e, fn, "hipMemcpyAsync");
    const int64_t last = ((n - 1) / PW) * PW;
    for (int64_t k0 = last; k0 >= 0; k0 -= PW) {
        const int w = (int)((n - k0 < PW) ? (n - k0) : PW);
        const unsigned grid = (unsigned)((k0 + SB - 1) / SB);
        hipLaunchKernelGGL((k_bwd_panel<T>), dim3(grid ? grid : 1), dim3(256), 0, st, l, ld, (int)n,
                           ws, work, res, q, (int)k0, w);
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
