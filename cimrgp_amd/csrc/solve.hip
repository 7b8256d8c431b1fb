// D3: alpha = (L L^T)^-1 R for a few right-hand sides (q <= 8), and the D5
// tail (row reductions over W = K(X*,X) L^-T).
//
// The skinny solves walk the 256-column panels: per panel one narrow launch applies
// the stored 256x256 inverse of the diagonal block (left in the workspace by potrf),
// a second one subtracts  L[rows below, panel] z_p  (forward) or
// L[panel, columns left]^T a_p  (backward) from the running right-hand side.
// Right-hand sides are kept "RHS-major" (q x n) so the updates are coalesced.
// HBM-read bound: n^2/2 * sizeof(T) bytes per direction.
#include "common.hpp"

namespace cimrgp {

namespace {

#ifndef CIMRGP_STAMP
constexpr int SB   = 64;
#endif
constexpr int MAXQ = 8;

// (n x q) row-major  <->  (q x n)
template <typename T>
__global__ void k_transpose_nq(const T* __restrict__ src, T* __restrict__ dst, int64_t n, int q, int to_qn,
                               int64_t ssrc = 0, int64_t sdst = 0)
{
    src += (int64_t)blockIdx.y * ssrc;               // batch of independent solves: blockIdx.y selects the problem
    dst += (int64_t)blockIdx.y * sdst;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * q) return;
    if (to_qn) { const int64_t c = e / n, i = e - c * n; dst[e] = src[i * q + c]; }
    else       { const int64_t i = e / q, c = e - i * q; dst[e] = src[c * n + i]; }
}

constexpr int PW = CIMRGP_NB;      // panel width of the skinny solves (256)
constexpr int ST = 1024;           // threads per workgroup of the skinny solves
constexpr int FWD_ROWS = ST / 64;  // rows per workgroup of k_fwd_update (a wave per row)

// ---------------------------------------------------------------------------
// Q = number of right-hand sides at compile time: a run-time `if (c < q)` inside the load loops
// is a branch per iteration and serialises the loads (one memory round trip each).
//
// Forward panel step in two launches (round 2; see the backward pair below for why):
//   k_fwd_alpha   z_p = L_pp^-1 w_p = invT_p^T w_p  (column sums over the rows <= column of the upper
//                 triangular invT_p): 16 workgroups x 16 columns, 64 row parts of 4 rows per column,
//                 fixed-order reduction in LDS;
//   k_fwd_update  w[rows below the panel] -= L[rows, panel] z_p : 16 rows per workgroup, a wave per
//                 row (round 5; 64 rows of 16 lanes before), z_p (w x Q) read once per workgroup.
// ---------------------------------------------------------------------------
template <typename T, int Q>
__global__ __launch_bounds__(ST)
void k_fwd_alpha(const T* __restrict__ invT, int n, const T* __restrict__ work, T* __restrict__ out, int k0, int w)
{
    __shared__ T wsh[Q][PW];
    __shared__ T red[64][Q][16];
    const int tid = threadIdx.x;
    const int c16 = tid & 15, part = tid >> 4;                    // 64 row parts of 4 rows
    const int col = blockIdx.x * 16 + c16;
    const T* bp = invT + (int64_t)(k0 / PW) * (PW * PW) + col;
    T bv[4];
    const int rbeg = part * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (rbeg + e <= col) ? bp[(int64_t)(rbeg + e) * PW] : (T)0;
    for (int e = tid; e < Q * PW; e += ST) {
        const int c = e / PW, u = e - c * PW;
        wsh[c][u] = (u < w) ? work[(int64_t)c * n + k0 + u] : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        T sv = bv[0] * wsh[c][rbeg];
#pragma unroll
        for (int e = 1; e < 4; ++e) sv += bv[e] * wsh[c][rbeg + e];
        red[part][c][c16] = sv;
    }
    __syncthreads();
    if (tid < Q * 16) {
        const int c = tid >> 4, u = tid & 15;
        T sv = (T)0;
#pragma unroll 8
        for (int pp = 0; pp < 64; ++pp) sv += red[pp][c][u];
        const int oc = blockIdx.x * 16 + u;
        if (oc < w) out[(int64_t)c * n + k0 + oc] = sv;
    }
}

// Round 5: 16 rows per workgroup, a whole wave per row (4 elements per lane) instead of 64 rows x 16 lanes: a workgroup
// streams 32 KB of L instead of 128 KB and four times as many share a step (the step's time is one workgroup's time).
template <typename T, int Q>
__global__ __launch_bounds__(ST)
void k_fwd_update(const T* __restrict__ L, int64_t ld, int n, T* __restrict__ work, const T* __restrict__ zp, int k0, int w)
{
    constexpr int EPL = PW / 64;                                  // elements per lane: a 256-column panel row over one wave
    __shared__ T zs[Q][PW];
    const int tid = threadIdx.x;
    const int lane = tid & 63, slot = tid >> 6;                   // 16 rows per workgroup
    const int r = k0 + w + blockIdx.x * FWD_ROWS + slot;
    T lv[EPL];
    T wold[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) wold[c] = (lane == 0 && r < n) ? work[(int64_t)c * n + r] : (T)0;    // requested up front
    if (r < n) {
        const T* lp = L + (int64_t)r * ld + k0 + lane * EPL;
        if (w == PW) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) lv[e] = lp[e];
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) lv[e] = (lane * EPL + e < w) ? lp[e] : (T)0;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) lv[e] = (T)0;
    }
    for (int e = tid; e < Q * PW; e += ST) {
        const int c = e / PW, u = e - c * PW;
        zs[c][u] = (u < w) ? zp[(int64_t)c * n + k0 + u] : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        T sv = (T)0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) sv += lv[e] * zs[c][lane * EPL + e];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (lane == 0 && r < n) work[(int64_t)c * n + r] = wold[c] - sv;
    }
}

// ---------------------------------------------------------------------------
// Backward panel step in two narrow-latency launches (round 2).  Round 1 used ONE launch per panel
// in which every workgroup first applied the 256x256 inverse redundantly (512 KB of invT streamed
// through EVERY workgroup, L2-resident) before its update could start: 23 us per panel against 8.7 us
// for the pair below (backward solve + prediction tail at N = 8192: 0.82 -> 0.35 ms).
//   k_bwd_alpha   a_p = invT_p w_p : one wave per row of invT_p (upper triangular: columns >= row),
//                 2 KB per row, 64 workgroups of 4 rows;
//   k_bwd_update  w[cols left of the panel] -= L[panel, cols]^T a_p : thread = column, 16-way split
//                 over the panel's rows, 64 columns per workgroup; a_p (w x Q) read once per workgroup.
// Both keep a fixed summation order (bit-identical on repetition).
// ---------------------------------------------------------------------------
template <typename T, int Q>
__global__ __launch_bounds__(256)
void k_bwd_alpha(const T* __restrict__ invT, int n, const T* __restrict__ work, T* __restrict__ out, int k0, int w,
                 int64_t sws = 0, int64_t sscr = 0)
{
    invT += (int64_t)blockIdx.y * sws;
    work += (int64_t)blockIdx.y * sscr;
    out += (int64_t)blockIdx.y * sscr;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= w) return;
    const T* rp = invT + (int64_t)(k0 / PW) * (PW * PW) + (int64_t)r * PW + lane * 4;
    T bv[4], wv[Q][4];
    const bool live = lane * 4 + 3 >= r;             // this lane's columns reach the diagonal or beyond
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = live ? rp[e] : (T)0;
#pragma unroll
    for (int c = 0; c < Q; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = lane * 4 + e;
            wv[c][e] = (live && u < w) ? work[(int64_t)c * n + k0 + u] : (T)0;
        }
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        T sv = bv[0] * wv[c][0];
#pragma unroll
        for (int e = 1; e < 4; ++e) sv += bv[e] * wv[c][e];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (lane == 0) out[(int64_t)c * n + k0 + r] = sv;
    }
}

// Round 4: TWO panels per backward step (62 -> 32 dependent launches at n = 8192) through the 512 x 512 inverse
//     [invT_p  X; 0  invT_p+1]  (X = -A^-T B^T C^-T, built behind the 256 x 256 inverses: potrf.hip, build_invT):
//   a_p = invT_p w_p + X w_p+1,  a_p+1 = invT_p+1 w_p+1 : one wave per row of the pair, 128 workgroups of 4 rows.
template <typename T, int Q>
__global__ __launch_bounds__(256)
void k_bwd_alpha2(const T* __restrict__ invT, const T* __restrict__ xoff, int n, const T* __restrict__ work, T* __restrict__ out,
                  int k0, int64_t sws = 0, int64_t sscr = 0)
{
    invT += (int64_t)blockIdx.y * sws;
    xoff += (int64_t)blockIdx.y * sws;
    work += (int64_t)blockIdx.y * sscr;
    out += (int64_t)blockIdx.y * sscr;
    const int lane = threadIdx.x & 63;
    const int r2 = blockIdx.x * 4 + (threadIdx.x >> 6);          // row of the pair, 0 .. 511
    const bool top = r2 < PW;
    const int r = top ? r2 : r2 - PW;
    const T* rp = invT + (int64_t)(k0 / PW + (top ? 0 : 1)) * (PW * PW) + (int64_t)r * PW + lane * 4;
    const int wcol = k0 + (top ? 0 : PW) + lane * 4;              // the right-hand side columns this lane multiplies
    const bool live = lane * 4 + 3 >= r;                          // this lane's columns reach the diagonal or beyond
    T bv[4], xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = live ? rp[e] : (T)0;
    if (top) {
        const T* xp = xoff + (int64_t)(k0 / (2 * PW)) * (PW * PW) + (int64_t)r * PW + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = xp[e];
    }
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        T sv = (T)0;
        if (live) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sv += bv[e] * work[(int64_t)c * n + wcol + e];
        }
        if (top) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sv += xv[e] * work[(int64_t)c * n + k0 + PW + lane * 4 + e];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (lane == 0) out[(int64_t)c * n + k0 + r2] = sv;
    }
}

// PWU = rows of the step: one panel (256) or a pair (512).  COLS = columns per workgroup: 64, or (round 5, single matrices)
// one 128-byte line per row -- a workgroup then streams 64 KB instead of 256 KB and four times as many of them share a
// step; the step's time is one workgroup's time (k_bwd_update at n = 8192: 8.9-10 us per step whatever the width left).
template <typename T, int Q, int PWU = PW, int COLS = SB>
__global__ __launch_bounds__(ST)
void k_bwd_update(const T* __restrict__ L, int64_t ld, int n, T* __restrict__ work, const T* __restrict__ alpha,
                  int k0, int w, int64_t sk = 0, int64_t sscr = 0)
{
    L += (int64_t)blockIdx.y * sk;
    work += (int64_t)blockIdx.y * sscr;
    alpha += (int64_t)blockIdx.y * sscr;
    constexpr int PARTS = ST / COLS;                              // row parts: 16 (64 columns) ... 64 (16 columns)
    constexpr int RPP = PWU / PARTS;                              // rows per part
    static_assert(ST % COLS == 0 && PWU % PARTS == 0 && RPP >= 1, "k_bwd_update: the step's rows split evenly over the parts");
    __shared__ T zs[Q][PWU];
    __shared__ T red[PARTS][Q][COLS];
    const int tid = threadIdx.x;
    const int t = tid % COLS, part = tid / COLS;
    const int col = blockIdx.x * COLS + t;
    const int ubeg = part * RPP;
    // the value this thread will update at the end (threads 0 .. Q COLS - 1), requested now: one round trip less behind the sums
    const int oc = tid / COLS, ocol = blockIdx.x * COLS + (tid - oc * COLS);
    const bool owner = (tid < Q * COLS) && (ocol < k0);
    const T wold = owner ? work[(int64_t)oc * n + ocol] : (T)0;
    T lv[RPP];
    if (col < k0) {
        const T* lp = L + (int64_t)k0 * ld + col;
        if (w == PWU) {
#pragma unroll
            for (int e = 0; e < RPP; ++e) lv[e] = lp[(int64_t)(ubeg + e) * ld];
        } else {
#pragma unroll
            for (int e = 0; e < RPP; ++e) lv[e] = (ubeg + e < w) ? lp[(int64_t)(ubeg + e) * ld] : (T)0;
        }
    } else {
#pragma unroll
        for (int e = 0; e < RPP; ++e) lv[e] = (T)0;
    }
    for (int e = tid; e < Q * PWU; e += ST) {
        const int c = e / PWU, u = e - c * PWU;
        zs[c][u] = (u < w) ? alpha[(int64_t)c * n + k0 + u] : (T)0;
    }
    __syncthreads();
    T acc[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) acc[c] = (T)0;
#pragma unroll
    for (int e = 0; e < RPP; ++e) {
#pragma unroll
        for (int c = 0; c < Q; ++c)
            acc[c] += lv[e] * zs[c][ubeg + e];
    }
#pragma unroll
    for (int c = 0; c < Q; ++c)
        red[part][c][t] = acc[c];
    __syncthreads();
    // one thread per (output, column): the parts in their fixed order
    if (owner) {
        T sv = (T)0;
#pragma unroll
        for (int pp = 0; pp < PARTS; ++pp) sv += red[pp][oc][tid - oc * COLS];
        work[(int64_t)oc * n + ocol] = wold - sv;
    }
}

// D5 tail: one wave per row of W.
template <typename T, int Q>
__global__ __launch_bounds__(256)
void k_predict_from_w(const T* __restrict__ W, int ns, int n, int64_t ldw, const T* __restrict__ z, int q,
                      T sf2_plus, const T* __restrict__ extra_dev, const T* __restrict__ bias, T* __restrict__ mean,
                      T* __restrict__ var, int accumulate, const int64_t* __restrict__ t_starts = nullptr, int64_t sw = 0)
{
    if (t_starts) {
        // the blocks of one layer in one launch (blockIdx.y = block): block b's rows of W are matrix b of the
        // arena, its outputs start at test row t_starts[b]; z, bias and the extra variance are per block
        const int b = blockIdx.y;
        W += (int64_t)b * sw;
        if (z) z += (int64_t)b * n * q;
        if (bias) bias += (int64_t)b * q;
        if (extra_dev) extra_dev += b;
        if (mean) mean += t_starts[b] * q;
        if (var) var += t_starts[b];
    }
    // Round 5: one ROW per workgroup, a quarter of it per wave (until then one row per wave: 2048 waves for the bench's
    // W, two per SIMD, 4 KB in flight each -- 134 MB in 60-80 us; four times the waves keep four times the bytes in flight).
    // The four partial sums are combined in wave order (fixed: bit-identical on repetition).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const T* wp = W + (int64_t)row * ldw;
    T ss = (T)0;
    T sm[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) sm[c] = (T)0;
    const bool want_mean = (mean != nullptr) && (z != nullptr);
    // rows start on a 128-byte line (ldw is a multiple of 16 elements): two elements per lane and
    // load, eight independent loads in flight per lane
    const int n2 = n & ~1;
    const int chunk = (((n2 + 3) / 4 + 127) / 128) * 128;        // elements per wave: whole 128-element sweeps
    const int jend = (n2 < (wave + 1) * chunk) ? n2 : (wave + 1) * chunk;
#pragma unroll 8
    for (int j = wave * chunk + 2 * lane; j < jend; j += 128) {
        const T w0 = wp[j], w1 = wp[j + 1];
        ss += w0 * w0;
        ss += w1 * w1;
        if (want_mean) {
#pragma unroll
            for (int c = 0; c < Q; ++c) {
                sm[c] += w0 * z[(int64_t)j * q + c];
                sm[c] += w1 * z[(int64_t)(j + 1) * q + c];
            }
        }
    }
    if (n2 < n && lane == 0 && wave == 3) {
        const T w = wp[n2];
        ss += w * w;
        if (want_mean) {
#pragma unroll
            for (int c = 0; c < Q; ++c) sm[c] += w * z[(int64_t)n2 * q + c];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
#pragma unroll
        for (int c = 0; c < Q; ++c)
            sm[c] += __shfl_xor(sm[c], off, 64);
    }
    __shared__ T part[4][Q + 1];
    if (lane == 0) {
        part[wave][Q] = ss;
#pragma unroll
        for (int c = 0; c < Q; ++c) part[wave][c] = sm[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ss = part[0][Q];
#pragma unroll
        for (int c = 0; c < Q; ++c) sm[c] = part[0][c];
#pragma unroll
        for (int w4 = 1; w4 < 4; ++w4) {
            ss += part[w4][Q];
#pragma unroll
            for (int c = 0; c < Q; ++c) sm[c] += part[w4][c];
        }
    }
    if (threadIdx.x == 0) {
        if (var) { const T v = sf2_plus + (extra_dev ? extra_dev[0] : (T)0) - ss; var[row] = accumulate ? var[row] + v : v; }
        if (want_mean) {
#pragma unroll
            for (int c = 0; c < Q; ++c) {
                {
                    const T m = sm[c] + (bias ? bias[c] : (T)0);
                    T* o = mean + (int64_t)row * q + c;
                    *o = accumulate ? *o + m : m;
                }
            }
        }
    }
}

}  // namespace

// Dispatch the run-time number of right-hand sides to a compile-time constant QQ.
#define CIMRGP_Q_SWITCH(q_, ...)                                   \
    switch (q_) {                                                  \
        case 1: { constexpr int QQ = 1; __VA_ARGS__; } break;      \
        case 2: { constexpr int QQ = 2; __VA_ARGS__; } break;      \
        case 3: { constexpr int QQ = 3; __VA_ARGS__; } break;      \
        case 4: { constexpr int QQ = 4; __VA_ARGS__; } break;      \
        case 5: { constexpr int QQ = 5; __VA_ARGS__; } break;      \
        case 6: { constexpr int QQ = 6; __VA_ARGS__; } break;      \
        case 7: { constexpr int QQ = 7; __VA_ARGS__; } break;      \
        default: { constexpr int QQ = 8; __VA_ARGS__; } break;     \
    }

template <typename T>
int potrs_run(const T* l, int64_t n, int64_t ld, const T* ws, T* rhs, int q, T* z_out, T* scratch,
              bool backward_only, hipStream_t st, PotrfBatch bt, bool work_ready)
{
    const char* fn = "cimrgp_potrs";
    if (n <= 0) return 0;
    CIMRGP_REQUIRE(bt.count == 1 || backward_only, fn, "only the backward half is batched");
    CIMRGP_REQUIRE(bt.count >= 1 && bt.count < 65536, fn, "batch count out of range");
    const unsigned nbatch = (unsigned)bt.count;
    const int64_t sscr = (bt.count > 1) ? 2 * (int64_t)q * n : 0;     // scratch per problem: [work | res]
    const int64_t srhs = (bt.count > 1) ? (int64_t)q * n : 0;         // right-hand sides per problem (n x q)
    CIMRGP_REQUIRE(q >= 1 && q <= MAXQ, fn, "number of right-hand sides must be in [1, 8]");
    CIMRGP_REQUIRE(n < (1ll << 31), fn, "matrix too large");
    T* work = scratch;             // (q x n) running right-hand side
    T* res  = scratch + q * n;     // (q x n) solved blocks
    const unsigned tg = (unsigned)((n * q + 255) / 256);
    // work_ready (backward half only): the caller has put the right-hand sides into `work` itself (k_layer_z writes them
    // there beside z: one launch less in front of a chain of dependent launches)
    CIMRGP_REQUIRE(!work_ready || backward_only, fn, "a prepared right-hand side is for the backward half");
    if (!work_ready) {
        hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg, nbatch), dim3(256), 0, st, (const T*)rhs, work, n, q, 1, srhs, sscr);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    const T* invT = ws + ((n + SB - 1) / SB) * (SB * SB);
    for (int64_t k0 = 0; k0 < n && !backward_only; k0 += PW) {
        const int w = (int)((n - k0 < PW) ? (n - k0) : PW);
        const int64_t below = n - (k0 + w);
        const unsigned grid = (unsigned)((below + FWD_ROWS - 1) / FWD_ROWS);
        CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_fwd_alpha<T, QQ>), dim3((unsigned)((w + 15) / 16)), dim3(ST), 0, st,
                                              invT, (int)n, (const T*)work, res, (int)k0, w));
        CIMRGP_LAUNCH_CHECK(fn);
        if (grid) {
            CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_fwd_update<T, QQ>), dim3(grid), dim3(ST), 0, st, l, ld, (int)n,
                                                  work, (const T*)res, (int)k0, w));
            CIMRGP_LAUNCH_CHECK(fn);
        }
    }
    if (!backward_only) {
        if (z_out) {
            hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg), dim3(256), 0, st, (const T*)res, z_out, n, q, 0);
            CIMRGP_LAUNCH_CHECK(fn);
        }
        hipError_t e = hipMemcpyAsync(work, res, sizeof(T) * (size_t)(q * n), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return check_hip(e, fn, "hipMemcpyAsync");
    }   // backward_only: `work` already holds z (RHS-major), put there by the first transpose
    // single matrices: one 128-byte line of L per row and workgroup (k_bwd_update); a batch of blocks has workgroups enough
    constexpr int LINE = 128 / (int)sizeof(T);
    const bool narrow = (nbatch == 1);
    const int64_t last = ((n - 1) / PW) * PW;
    const int64_t npairs = bwd_pairs(n);                           // pairs of full panels: two per step (k_bwd_alpha2)
    const T* xoff = invT + ((n + PW - 1) / PW) * (PW * PW);
    for (int64_t k0 = last; k0 >= 2 * npairs * PW; k0 -= PW) {    // what lies behind the pairs: at most two panels, singly
        const int w = (int)((n - k0 < PW) ? (n - k0) : PW);
        const unsigned grid = (unsigned)((k0 + SB - 1) / SB);
        CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_bwd_alpha<T, QQ>), dim3((unsigned)((w + 3) / 4), nbatch), dim3(256), 0, st,
                                              invT, (int)n, (const T*)work, res, (int)k0, w, bt.sws, sscr));
        CIMRGP_LAUNCH_CHECK(fn);
        if (grid && narrow) {
            CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_bwd_update<T, QQ, PW, LINE>), dim3((unsigned)((k0 + LINE - 1) / LINE), nbatch), dim3(ST), 0, st,
                                                  l, ld, (int)n, work, (const T*)res, (int)k0, w, bt.sk, sscr));
            CIMRGP_LAUNCH_CHECK(fn);
        } else if (grid) {
            CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_bwd_update<T, QQ>), dim3(grid, nbatch), dim3(ST), 0, st, l, ld, (int)n,
                                                  work, (const T*)res, (int)k0, w, bt.sk, sscr));
            CIMRGP_LAUNCH_CHECK(fn);
        }
    }
    for (int64_t j = npairs - 1; j >= 0; --j) {
        const int64_t k0 = 2 * PW * j;
        const unsigned grid = (unsigned)((k0 + SB - 1) / SB);
        CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_bwd_alpha2<T, QQ>), dim3(2 * PW / 4, nbatch), dim3(256), 0, st,
                                              invT, xoff, (int)n, (const T*)work, res, (int)k0, bt.sws, sscr));
        CIMRGP_LAUNCH_CHECK(fn);
        if (grid && narrow) {
            CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_bwd_update<T, QQ, 2 * PW, LINE>), dim3((unsigned)((k0 + LINE - 1) / LINE), nbatch), dim3(ST), 0, st,
                                                  l, ld, (int)n, work, (const T*)res, (int)k0, 2 * PW, bt.sk, sscr));
            CIMRGP_LAUNCH_CHECK(fn);
        } else if (grid) {
            CIMRGP_Q_SWITCH(q, hipLaunchKernelGGL((k_bwd_update<T, QQ, 2 * PW>), dim3(grid, nbatch), dim3(ST), 0, st, l, ld, (int)n,
                                                  work, (const T*)res, (int)k0, 2 * PW, bt.sk, sscr));
            CIMRGP_LAUNCH_CHECK(fn);
        }
    }
    hipLaunchKernelGGL((k_transpose_nq<T>), dim3(tg, nbatch), dim3(256), 0, st, (const T*)res, rhs, n, q, 0, sscr, srhs);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
int predict_from_w_run(const T* w, int64_t ns, int64_t n, int64_t ldw, const T* z, int q, double sf2,
                       double extra, const T* extra_dev, const T* bias, T* mean, T* var, int accumulate, hipStream_t st,
                       int batch, const int64_t* t_starts, int64_t sw)
{
    const char* fn = "cimrgp_predict_from_w";
    if (ns <= 0 || batch <= 0) return 0;
    CIMRGP_REQUIRE(q >= 0 && q <= MAXQ, fn, "number of outputs must be <= 8");
    CIMRGP_REQUIRE(ns < (1ll << 31) && n < (1ll << 31) && batch < 65536, fn, "too many points");
    CIMRGP_REQUIRE(batch == 1 || t_starts != nullptr, fn, "a batch needs the blocks' test offsets");
    CIMRGP_Q_SWITCH(q > 0 ? q : 1, hipLaunchKernelGGL((k_predict_from_w<T, QQ>), dim3((unsigned)ns, (unsigned)batch), dim3(256), 0, st,
                                                      w, (int)ns, (int)n, ldw, z, q, (T)(sf2 + extra), extra_dev, bias, mean, var, accumulate,
                                                      t_starts, sw));
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template int potrs_run<double>(const double*, int64_t, int64_t, const double*, double*, int, double*, double*, bool, hipStream_t, PotrfBatch, bool);
template int potrs_run<float>(const float*, int64_t, int64_t, const float*, float*, int, float*, float*, bool, hipStream_t, PotrfBatch, bool);
template int predict_from_w_run<double>(const double*, int64_t, int64_t, int64_t, const double*, int, double, double,
                                        const double*, const double*, double*, double*, int, hipStream_t, int, const int64_t*, int64_t);
template int predict_from_w_run<float>(const float*, int64_t, int64_t, int64_t, const float*, int, double, double,
                                       const float*, const float*, float*, float*, int, hipStream_t, int, const int64_t*, int64_t);

}  // namespace cimrgp
