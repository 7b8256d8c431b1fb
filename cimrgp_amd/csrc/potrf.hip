// Blocked right-looking Cholesky (lower, row-major, in place) and the row-wise
// triangular solve that shares its panel kernels.
//
//   for each panel of CIMRGP_NB = 256 columns:
//     for each 64-column sub-block s of the panel (left-looking inside the panel):
//        k_diag64   one workgroup: A_ss -= L_s,prev L_s,prev^T (MFMA), factor the
//                   64x64 block from registers and form its inverse alongside
//        k_trsm64   rows below: X = (P_s - P_prev L_s,prev^T) inv(L_ss)^T  (MFMA, in place)
//     gemm_nt (lower)  trailing matrix -= panel * panel^T       (MFMA, K = 256)
//
// The inverted 64x64 diagonal blocks stay in the workspace (slab c0/64) and
// are what cimrgp_potrs / cimrgp_trsm_rows use afterwards.
#include "common.hpp"

#include <vector>

namespace cimrgp {

// ---- optional per-launch timing of the trailing update (bench.py roofline) ----
// Events are recorded on the launch stream around every lower-triangular
// trailing-update launch while profiling is on; collect() waits for them.
namespace {
struct TrailRec { hipEvent_t start, stop; double flops; };
std::vector<TrailRec> g_recs;
std::vector<TrailRec> g_free;
bool g_profile = false;
}  // namespace

int profile_begin()
{
    g_profile = true;
    return 0;
}

int profile_collect(double* total_ms, double* total_flops, int64_t* launches)
{
    double ms = 0.0, fl = 0.0;
    int64_t cnt = 0;
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.stop);
        if (e != hipSuccess) return check_hip(e, "cimrgp_profile_collect", "hipEventSynchronize");
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.start, r.stop);
        if (e != hipSuccess) return check_hip(e, "cimrgp_profile_collect", "hipEventElapsedTime");
        ms += t; fl += r.flops; ++cnt;
        g_free.push_back(r);
    }
    g_recs.clear();
    g_profile = false;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = cnt;
    return 0;
}

static TrailRec* rec_open(hipStream_t st, double flops)
{
    if (!g_profile) return nullptr;
    TrailRec r;
    if (!g_free.empty()) { r = g_free.back(); g_free.pop_back(); }
    else {
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return nullptr;
    }
    r.flops = flops;
    (void)hipEventRecord(r.start, st);
    g_recs.push_back(r);
    return &g_recs.back();
}

namespace {

constexpr int SB = 64;   // diagonal sub-block
constexpr int64_t ROWS_START_BELOW = 4608;   // carried rows start once the trailing matrix is smaller than this


template <typename T> struct Tile64 {
    static constexpr int ROWB  = SB * (int)sizeof(T);   // bytes of 64 k-values per row
    static constexpr int LROW  = ROWB + 16;              // padded LDS row stride
    static constexpr int CPR   = ROWB / 16;              // 16-byte chunks per row
    static constexpr int NSTEP = ROWB / 32;              // slot-steps (4 slots x 8 B) per 64 k
    static constexpr int KPS   = 32 / (int)sizeof(T);    // k values per slot-step
    static constexpr int BYTES = SB * LROW;
};

// Cooperative, coalesced load of a 64 x kw strip (row stride ld elements) into a
// padded LDS tile; rows >= mrows and columns >= kw are zero-filled.
template <typename T, int ROWS = SB>
static __device__ __forceinline__ void load_tile64(unsigned char* dst, const T* __restrict__ src, int64_t ld,
                                                    int mrows, int kw)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    for (int e = threadIdx.x; e < ROWS * TL::CPR; e += 256) {
        const int r = e / TL::CPR, c = e - r * TL::CPR;
        const int kcol = c * X::EPC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < mrows && kcol < kw) {
            v = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + kcol);
            if (kcol + X::EPC > kw) v = mask_chunk<T>(v, kcol, kw);
        }
        *reinterpret_cast<uint4*>(dst + r * TL::LROW + c * 16) = v;
    }
}

// Same strip, split in two halves so the global loads of the next k chunk can be in
// flight (in registers) while the current chunk is multiplied.
template <typename T, int ROWS>
static __device__ __forceinline__ void gload_tile64(uint4 (&regs)[ROWS * Tile64<T>::CPR / 256], const T* __restrict__ src,
                                                     int64_t ld, int mrows, int kw)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
#pragma unroll
    for (int p = 0; p < ROWS * TL::CPR / 256; ++p) {
        const int e = threadIdx.x + 256 * p;
        const int r = e / TL::CPR, c = e - r * TL::CPR;
        const int kcol = c * X::EPC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < mrows && kcol < kw) {
            v = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + kcol);
            if (kcol + X::EPC > kw) v = mask_chunk<T>(v, kcol, kw);
        }
        regs[p] = v;
    }
}

template <typename T, int ROWS>
static __device__ __forceinline__ void swrite_tile64(unsigned char* dst, const uint4 (&regs)[ROWS * Tile64<T>::CPR / 256])
{
    using TL = Tile64<T>;
#pragma unroll
    for (int p = 0; p < ROWS * TL::CPR / 256; ++p) {
        const int e = threadIdx.x + 256 * p;
        const int r = e / TL::CPR, c = e - r * TL::CPR;
        *reinterpret_cast<uint4*>(dst + r * TL::LROW + c * 16) = regs[p];
    }
}

// acc[ct] += A(16 rows of this wave) * B(rows 16 ct ..)^T over one 64-wide k chunk.
template <typename T, bool TRI>
static __device__ __forceinline__ void mma_chunk64(typename Mx<T>::acc_t (&acc)[4], const unsigned char* as,
                                                    const unsigned char* bs, int wave, int lane)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned char* pa = as + (wave * 16 + frow) * TL::LROW + fslot * 8;
    const unsigned char* pb = bs + frow * TL::LROW + fslot * 8;
#pragma unroll
    for (int s = 0; s < TL::NSTEP; ++s) {
        const uint2 a = *reinterpret_cast<const uint2*>(pa + s * 32);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            // TRI: B is lower triangular (B[j][k] = 0 for k > j): column tile ct needs k <= 16 ct + 15
            if (!TRI || (s * TL::KPS <= 16 * ct + 15)) {
                const uint2 b = *reinterpret_cast<const uint2*>(pb + ct * 16 * TL::LROW + s * 32);
                acc[ct] = X::mma(a, b, acc[ct]);
            }
        }
    }
}

// 32-row variant: wave = (rt = row tile 0/1, ch = column half 0/1): 16 rows x 32 columns.
template <typename T, bool TRI>
static __device__ __forceinline__ void mma_chunk32(typename Mx<T>::acc_t (&acc)[2], const unsigned char* as,
                                                    const unsigned char* bs, int rt, int ch, int lane)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned char* pa = as + (rt * 16 + frow) * TL::LROW + fslot * 8;
    const unsigned char* pb = bs + (ch * 32 + frow) * TL::LROW + fslot * 8;
#pragma unroll
    for (int s = 0; s < TL::NSTEP; ++s) {
        // TRI: column tile ct = 2 ch + c needs k <= 16 ct + 15; ch is run-time (wave-uniform)
        if (!TRI || (s * TL::KPS <= 16 * (2 * ch + 1) + 15)) {
            const uint2 a = *reinterpret_cast<const uint2*>(pa + s * 32);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!TRI || (s * TL::KPS <= 16 * (2 * ch + c) + 15)) {
                    const uint2 b = *reinterpret_cast<const uint2*>(pb + c * 16 * TL::LROW + s * 32);
                    acc[c] = X::mma(a, b, acc[c]);
                }
            }
        }
    }
}

// Wave-local broadcast of lane `src`'s value (src wave-uniform): v_readlane, no LDS.
static __device__ __forceinline__ double bcast_lane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ float bcast_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

template <typename T> static __device__ __forceinline__ T rsqrt_refined(T d);
template <> __device__ __forceinline__ double rsqrt_refined<double>(double d)
{
    double r = __builtin_amdgcn_rsq(d);          // ~2^-26; two Newton steps -> full f64
    {
        const double e0 = fma(-d * r, r, 1.0);
        r = fma(0.5 * r, e0, r);
    }
    const double e = fma(-d * r, r, 1.0);       // one Newton step: full f64 accuracy
    return fma(0.5 * r, e, r);
}
template <> __device__ __forceinline__ float rsqrt_refined<float>(float d)
{
    float r = rsqrtf(d);
    const float e = fmaf(-d * r, r, 1.0f);
    return fmaf(0.5f * r, e, r);
}

// ---------------------------------------------------------------------------
// Diagonal 64x64 sub-block of a panel (left-looking inside the panel):
//   S = A_ss - Lrow Lrow^T        Lrow = the kprev panel columns left of the
//                                 block, already final (MFMA, K = kprev <= 192)
//   S = L L^T  in place, inv slab = L^-1 (lower; zero elsewhere).
// Factor and inverse advance together, one barrier per column.  Thread
// (i = tid & 63, g = tid >> 6) keeps row i, columns k = g + 4u (u < 16) of the
// unscaled Schur complement S and of the unscaled inverse Mi in REGISTERS; per
// column only column j of S and row j of Mi travel through LDS (double
// buffered).  Column j of L is S[:,j] r_j and row j of L^-1 is Mi[j,:] r_j,
// r_j = 1/sqrt(S[j][j]); the row operations that reduce S are applied to Mi.
// ---------------------------------------------------------------------------
// LDS-only barrier: waits for this wave's LDS traffic, not for global stores in
// flight (a plain __syncthreads() also drains vmcnt, i.e. every store's round trip).
static __device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Register layout of the column loop: thread (i = tid & 63, g = tid >> 6) keeps row i,
// columns k = 16 g + u (u < 16) of ONE combined 64x64 array A:
//     A[i][k] = S[i][k]   for k <= i   (Schur complement, lower triangle)
//     A[i][k] = Mi[k][i]  for k >  i   (unscaled inverse, stored transposed in the upper triangle)
// At pivot j everything a thread needs is column j of A: A[i][j] is S[i][j] for i >= j and
// Mi[j][i] for i < j, so the wave that owns column j publishes it with ONE LDS store, and
//     A[i][k] -= (A[k][j] r) * h_i   for every k > j,    r = 1/sqrt(A[j][j]),
//     h_i = L[i][j] = A[i][j] r (i > j),  L^-1[j][i] = A[i][j] r (i < j),  r (i = j, from 0)
// covers the Schur update, the inverse update and the birth of column j of Mi in one
// formula (for j < i < k it touches a not-yet-born Mi slot, which is reset at pivot i).
// Waves left of the pivot column have nothing to do; register indices are compile-time
// constants (u0 unrolled, wave index looped).
constexpr int DG_NW = 8;              // waves of the diagonal kernel: two per SIMD (the loop is VALU-issue bound)
constexpr int DG_NS = SB / DG_NW;     // register slots (columns) per lane
constexpr int DG_NT = 64 * DG_NW;

template <typename T>
__global__ __launch_bounds__(DG_NT)
void k_diag64(T* __restrict__ D, int64_t ld, int w, const T* __restrict__ Lrow, int kprev,
              T* __restrict__ inv, int32_t* info, int col_base)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    constexpr int LS = SB + 1;
    __shared__ __attribute__((aligned(16))) unsigned char chunk[TL::BYTES];   // Lrow chunk, later L^-1 out
    __shared__ T S[SB * LS];                                                  // Schur block, later L out
    constexpr int BC = 4;                                                     // pivots per barrier
    __shared__ __attribute__((aligned(16))) T comb4[2][SB][BC];               // BC pivot-time columns of A
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave id: provably uniform
    // latency-bound chain running next to MFMA-bound update workgroups: win issue arbitration
    __builtin_amdgcn_s_setprio(3);

    STAMP(0);
    for (int e = tid; e < SB * SB; e += DG_NT) {
        const int r = e >> 6, c = e & 63;
        T v = (r == c) ? (T)1 : (T)0;
        if (r < w && c <= r) v = D[(int64_t)r * ld + c];
        S[r * LS + c] = v;
    }
    STAMP(1);
    if (kprev > 0) {
        acc_t acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = acc_zero<T>();
        // chunk kc+1 travels global -> registers while chunk kc is multiplied
        constexpr int NR = SB * TL::CPR / DG_NT;      // 16-byte pieces per thread and chunk
        uint4 regs[NR];
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + DG_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            regs[p] = (r < w) ? *reinterpret_cast<const uint4*>(Lrow + (int64_t)r * ld + c * X::EPC) : make_uint4(0, 0, 0, 0);
        }
        for (int kc = 0; kc < kprev; kc += SB) {
            __syncthreads();
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int e = tid + DG_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                *reinterpret_cast<uint4*>(chunk + r * TL::LROW + c * 16) = regs[p];
            }
            __syncthreads();
            if (kc + SB < kprev) {
#pragma unroll
                for (int p = 0; p < NR; ++p) {
                    const int e = tid + DG_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                    regs[p] = (r < w) ? *reinterpret_cast<const uint4*>(Lrow + kc + SB + (int64_t)r * ld + c * X::EPC)
                                      : make_uint4(0, 0, 0, 0);
                }
            }
            if (g < 4) mma_chunk64<T, false>(acc, chunk, chunk, g, lane);
        }
        if (g < 4) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = g * 16 + X::crow(lane, r), col = ct * 16 + (lane & 15);
                    if (row < w && col <= row) S[row * LS + col] -= acc[ct][r];
                }
        }
    }
    __syncthreads();

    T a[DG_NS];
#pragma unroll
    for (int u = 0; u < DG_NS; ++u) a[u] = S[i * LS + DG_NS * g + u];   // strict upper part of S is zero
    T* lout = S;                                   // L[i][j]    at lout[i * LS + j]
    T* iout = reinterpret_cast<T*>(chunk);         // L^-1[j][c] at iout[j * SB + c]
    __syncthreads();
    for (int e = tid; e < SB * SB; e += DG_NT) iout[e] = (T)0;
    STAMP(2);

    // Column loop, BC pivots per barrier.  The wave that owns columns j0..j0+BC-1 (same register
    // slots u0..u0+BC-1 of every lane) eliminates them among themselves with wave-local broadcasts
    // (v_readlane, no LDS, no barrier), publishes the BC pivot-time columns with 16-byte LDS
    // stores, and after ONE barrier every wave applies the rank-BC update to its
    // slots right of the block.  A lane that is itself a pivot row of the block (i = j0 + p)
    // starts its not-yet-born inverse entries from 0 and takes contributions from pivots >= p only.
    for (int gg = 0; gg < DG_NW; ++gg) {
#pragma unroll
        for (int ub = 0; ub < DG_NS / BC; ++ub) {
            const int u0 = BC * ub;
            const int j0 = DG_NS * gg + u0;
            T (*cb)[BC] = comb4[(j0 / BC) & 1];
            T colv[BC], rr[BC];
            if (g == gg) {
#pragma unroll
                for (int t = 0; t < BC; ++t) {
                    const int j = j0 + t;
                    colv[t] = a[u0 + t];
                    const T d = bcast_lane(colv[t], j);
                    const T r = rsqrt_refined<T>(d);
                    rr[t] = r;
                    const T h = (i == j) ? r : colv[t] * r;
                    const T nhr = -h * r;
#pragma unroll
                    for (int t2 = t + 1; t2 < BC; ++t2) {
                        const T ak = bcast_lane(colv[t], j0 + t2);          // A[k][j] at pivot time
                        a[u0 + t2] = fma(nhr, ak, (i == j) ? (T)0 : a[u0 + t2]);
                    }
                    if (!(d > (T)0) && i == j && j < w) atomicCAS(info, 0, col_base + j + 1);
                    if (i >= j) lout[i * LS + j] = (i == j) ? d * r : h;
                    else        iout[j * SB + i] = h;
                    if (i == j) iout[j * SB + j] = r;
                }
#pragma unroll
                for (int t = 0; t < BC; ++t) cb[i][t] = colv[t];
            }
            lds_barrier();
            if (g >= gg) {
                if (g != gg) {
#pragma unroll
                    for (int t = 0; t < BC; ++t) {
                        colv[t] = cb[i][t];
                        rr[t] = rsqrt_refined<T>(cb[j0 + t][t]);
                    }
                }
                const bool in_block = (i >= j0) && (i < j0 + BC);
                T nhr[BC];
#pragma unroll
                for (int t = 0; t < BC; ++t) {
                    const T h = (i == j0 + t) ? rr[t] : colv[t] * rr[t];
                    nhr[t] = (in_block && i > j0 + t) ? (T)0 : -h * rr[t];
                }
#pragma unroll
                for (int u = 0; u < DG_NS; ++u) {
                    if (g > gg || u >= u0 + BC) {               // columns right of the block
                        const int k = DG_NS * g + u;
                        T v = in_block ? (T)0 : a[u];
#pragma unroll
                        for (int t = 0; t < BC; ++t) v = fma(nhr[t], cb[k][t], v);
                        a[u] = v;
                    }
                }
            }
        }
    }
    STAMP(3);
    __syncthreads();
    for (int e = tid; e < SB * SB; e += DG_NT) {
        const int r = e >> 6, c = e & 63;
        if (r < w && c <= r) D[(int64_t)r * ld + c] = lout[r * LS + c];
        inv[e] = (r < w && c < w) ? iout[e] : (T)0;
    }
    STAMP(4);
}

// ---------------------------------------------------------------------------
// Panel solve for one 64-column sub-block, left-looking inside the panel, in place:
//   T = P_s - Pprev Lrow^T        Pprev = this row's kprev earlier panel columns (final),
//                                 Lrow  = the diagonal block's rows, same columns
//   X = T invL^T                  invL = 64x64 inverse of the diagonal block (lower)
// One workgroup = 32 rows (each wave 16 rows x 32 columns = 2 MFMA tiles); a
// row is read completely before it is overwritten and no other workgroup
// touches it.  Two row sets share one launch (matrix rows below the block and
// the extra right-hand-side rows of a row-wise solve).
// ---------------------------------------------------------------------------
constexpr int TR = 32;   // rows per workgroup of the panel solve

template <typename T>
static __device__ __forceinline__ void trsm64_body(T* __restrict__ Prow, int64_t ldp, int mrows, int kw, int kprev,
                                                    const T* __restrict__ Lrow, int64_t ldl,
                                                    const T* __restrict__ invL)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    // 32 + 32 + 64 rows: 67.6 KB (f64) -- fits next to one resident trailing-update workgroup
    __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * TR + SB) * TL::LROW];
    unsigned char* ps = smem;                          // P_s, then T          (32 rows)
    unsigned char* as = smem + TR * TL::LROW;          // chunk of Pprev       (32 rows)
    unsigned char* bs = smem + 2 * TR * TL::LROW;      // chunk of Lrow, finally invL (64 rows)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, ch = wave >> 1;
    __builtin_amdgcn_s_setprio(3);               // panel chain: ahead of co-resident update waves
    const T* Pprev = Prow - kprev;               // the panel's earlier columns of the same rows

    STAMP(8);
    uint4 ra[TR * TL::CPR / 256], rb[SB * TL::CPR / 256];
    if (kprev > 0) {
        gload_tile64<T, TR>(ra, Pprev, ldp, mrows, SB);
        gload_tile64<T, SB>(rb, Lrow, ldl, kw, SB);
    }
    load_tile64<T, TR>(ps, Prow, ldp, mrows, kw);
    acc_t acc[2];
    acc[0] = acc_zero<T>(); acc[1] = acc_zero<T>();
    for (int kc = 0; kc < kprev; kc += SB) {
        __syncthreads();                          // previous chunk's fragments have been read
        swrite_tile64<T, TR>(as, ra);
        swrite_tile64<T, SB>(bs, rb);
        __syncthreads();
        if (kc + SB < kprev) {                    // next chunk in flight during the MFMAs
            gload_tile64<T, TR>(ra, Pprev + kc + SB, ldp, mrows, SB);
            gload_tile64<T, SB>(rb, Lrow + kc + SB, ldl, kw, SB);
        }
        mma_chunk32<T, false>(acc, as, bs, rt, ch, lane);
    }
    __syncthreads();
    STAMP(9);
    // T = P_s - acc (each lane owns its accumulator elements), and stage invL
    if (kprev > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + (lane & 15);
                T* t = reinterpret_cast<T*>(ps + row * TL::LROW) + col;
                *t -= acc[c][r];
            }
    }
    for (int e = tid; e < SB * TL::CPR; e += 256) {
        const int r = e / TL::CPR, c = e - r * TL::CPR;
        *reinterpret_cast<uint4*>(bs + r * TL::LROW + c * 16) =
            *reinterpret_cast<const uint4*>(invL + r * SB + c * X::EPC);
    }
    __syncthreads();
    acc[0] = acc_zero<T>(); acc[1] = acc_zero<T>();
    STAMP(10);
    mma_chunk32<T, true>(acc, ps, bs, rt, ch, lane);
    STAMP(11);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int gc = (2 * ch + c) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = rt * 16 + X::crow(lane, r);
            if (lr < mrows && gc < kw) Prow[(int64_t)lr * ldp + gc] = acc[c][r];
        }
    }
    STAMP(12);
}

template <typename T>
__global__ __launch_bounds__(256)
void k_trsm64(T* __restrict__ P1, int64_t ld1, int M1, int nb1,
              T* __restrict__ P2, int64_t ld2, int M2,
              int kw, int kprev, const T* __restrict__ Lrow, int64_t ldl, const T* __restrict__ invL)
{
    const bool second = (int)blockIdx.x >= nb1;
    T* P = second ? P2 : P1;
    const int64_t ldp = second ? ld2 : ld1;
    const int M = second ? M2 : M1;
    const int row0 = (second ? (int)blockIdx.x - nb1 : (int)blockIdx.x) * TR;
    trsm64_body<T>(P + (int64_t)row0 * ldp, ldp, min(TR, M - row0), kw, kprev, Lrow, ldl, invL);
}

// All four sub-steps of a panel for rows that take no part in the factorisation itself (the
// right-hand-side rows of a row-wise solve): one launch per panel instead of four.  A row
// block only ever reads its own earlier results, written by this same workgroup.
template <typename T>
__global__ __launch_bounds__(256)
void k_trsm256(T* __restrict__ P, int64_t ldp, int M, int w, const T* __restrict__ Lpanel, int64_t ldl,
               const T* __restrict__ inv64)
{
    const int row0 = (int)blockIdx.x * TR;
    const int mrows = min(TR, M - row0);
    for (int c0 = 0; c0 < w; c0 += SB) {
        if (c0) __syncthreads();                 // this workgroup's stores of the previous sub-step are visible
        trsm64_body<T>(P + (int64_t)row0 * ldp + c0, ldp, mrows, min(SB, w - c0), c0,
                       Lpanel + (int64_t)c0 * ldl, ldl, inv64 + (int64_t)(c0 / SB) * (SB * SB));
    }
}

// ---------------------------------------------------------------------------
// 256x256 inverses of the diagonal blocks, for the skinny solves: the identity is
// carried through the panel solve, batched over ALL panels (blockIdx.y), one launch
// per 64-column sub-step:  invT_p = I L_pp^-T = (L_pp^-1)^T  (upper triangular,
// row r = column r of L_pp^-1), stored 256 x 256 row-major per panel.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_invT_init(T* __restrict__ invT, int n)
{
    const int p = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // element of the 256 x 256 block
    const int r = e >> 8, c = e & 255;
    const int w = min(CIMRGP_NB, n - p * CIMRGP_NB);
    invT[(int64_t)p * (CIMRGP_NB * CIMRGP_NB) + e] = (r == c && r < w) ? (T)1 : (T)0;
}

template <typename T>
__global__ __launch_bounds__(256)
void k_invT_step(T* __restrict__ invT, const T* __restrict__ L, int64_t ld, int n, const T* __restrict__ inv64, int s)
{
    const int p = blockIdx.y;
    const int k0 = p * CIMRGP_NB, c0 = k0 + SB * s;
    const int kw = min(SB, n - c0);
    if (kw <= 0) return;
    T* Prow = invT + (int64_t)p * (CIMRGP_NB * CIMRGP_NB) + (int64_t)blockIdx.x * TR * CIMRGP_NB + SB * s;
    trsm64_body<T>(Prow, CIMRGP_NB, TR, kw, SB * s, L + (int64_t)c0 * ld + k0, ld, inv64 + (int64_t)(c0 / SB) * (SB * SB));
}

}  // namespace

// One pass over the panels.  With FACTOR the matrix itself is factored; with
// rows (b != nullptr) the extra rows are carried through the same panel
// operations, which turns them into  B L^-T.
template <typename T, bool FACTOR>
static int panel_sweep(T* kmat, int64_t n, int64_t ld, T* ws, int32_t* info,
                       T* b, int64_t m, int64_t ldb, hipStream_t st)
{
    const char* fn = FACTOR ? "cimrgp_potrf" : "cimrgp_trsm_rows";
    const bool rows = (b != nullptr && m > 0);
    for (int64_t k0 = 0; k0 < n; k0 += CIMRGP_NB) {
        const int64_t w = (n - k0 < CIMRGP_NB) ? (n - k0) : CIMRGP_NB;
        const int64_t k1 = k0 + w;
        if (!FACTOR && rows) {
            hipLaunchKernelGGL((k_trsm256<T>), dim3((unsigned)((m + TR - 1) / TR)), dim3(256), 0, st,
                               b + k0, ldb, (int)m, (int)w, (const T*)(kmat + k0 * ld + k0), ld,
                               (const T*)(ws + (k0 / SB) * (SB * SB)));
            CIMRGP_LAUNCH_CHECK(fn);
        }
        for (int64_t c0 = k0; FACTOR && c0 < k1; c0 += SB) {
            const int sw = (int)((k1 - c0 < SB) ? (k1 - c0) : SB);
            const int kprev = (int)(c0 - k0);
            const int64_t pc = c0 + sw;            // first row after this sub-block
            T* inv = ws + (c0 / SB) * (SB * SB);
            const T* lrow = kmat + c0 * ld + k0;   // rows of the diagonal block, earlier panel columns
            if (FACTOR) {
                hipLaunchKernelGGL((k_diag64<T>), dim3(1), dim3(DG_NT), 0, st,
                                   kmat + c0 * ld + c0, ld, sw, lrow, kprev, inv, info, (int)c0);
                CIMRGP_LAUNCH_CHECK(fn);
            }
            const int64_t m1 = FACTOR ? (n - pc) : 0;
            const int nb1 = (int)((m1 + TR - 1) / TR);
            const int nb2 = rows ? (int)((m + TR - 1) / TR) : 0;
            if (nb1 + nb2 > 0) {
                hipLaunchKernelGGL((k_trsm64<T>), dim3((unsigned)(nb1 + nb2)), dim3(256), 0, st,
                                   kmat + pc * ld + c0, ld, (int)m1, nb1,
                                   rows ? b + c0 : nullptr, ldb, rows ? (int)m : 0,
                                   sw, kprev, lrow, ld, (const T*)inv);
                CIMRGP_LAUNCH_CHECK(fn);
            }
        }
        if (n > k1) {
            if (FACTOR) {
                const double mm = (double)(n - k1);
                TrailRec* rec = rec_open(st, mm * (mm + 1.0) * (double)w);   // lower SYRK: M(M+1)K flop
                int rc = gemm_nt_sub<T>(kmat + k1 * ld + k1, ld, kmat + k1 * ld + k0, ld,
                                        kmat + k1 * ld + k0, ld, n - k1, n - k1, (int)w, true, st);
                if (rec) (void)hipEventRecord(rec->stop, st);
                if (rc) return rc;
            }
            if (rows) {
                int rc = gemm_nt_sub<T>(b + k1, ldb, b + k0, ldb, kmat + k1 * ld + k0, ld,
                                        m, n - k1, (int)w, false, st);
                if (rc) return rc;
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Factorisation with one-panel look-ahead.  The trailing update of panel p is
// split into the columns of panel p+1 ("head", rectangular) and the rest (lower
// SYRK).  As soon as the head is done, panel p+1 is factored on a second,
// high-priority stream while the main stream is still busy with the rest; the
// latency-bound panel chain (64 sequential columns per diagonal block) hides
// behind the MFMA-bound update for as long as the trailing matrix is large.
// Fork/join by events only (graph-capturable); the side stream and the event
// pool are created once per device and reused.
// ---------------------------------------------------------------------------
namespace {
struct LookAhead {
    hipStream_t side = nullptr;        // panel chain (high priority)
    hipStream_t rows = nullptr;        // carried rows: lags behind the factorisation, lowest priority
    std::vector<hipEvent_t> ev;
    int device = -1;
};
LookAhead g_la[16];

LookAhead* lookahead_ctx(size_t nevents)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    LookAhead& la = g_la[dev];
    if (la.side == nullptr) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&la.side, hipStreamNonBlocking, hi) != hipSuccess) { la.side = nullptr; return nullptr; }
        if (hipStreamCreateWithPriority(&la.rows, hipStreamNonBlocking, lo) != hipSuccess) la.rows = nullptr;
        la.device = dev;
    }
    while (la.ev.size() < nevents) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        la.ev.push_back(e);
    }
    return &la;
}

template <typename T>
int factor_panel(T* kmat, int64_t n, int64_t ld, T* ws, int32_t* info, int64_t k0, int64_t w, hipStream_t st)
{
    const char* fn = "cimrgp_potrf";
    const int64_t k1 = k0 + w;
    for (int64_t c0 = k0; c0 < k1; c0 += SB) {
        const int sw = (int)((k1 - c0 < SB) ? (k1 - c0) : SB);
        const int kprev = (int)(c0 - k0);
        const int64_t pc = c0 + sw;
        T* inv = ws + (c0 / SB) * (SB * SB);
        const T* lrow = kmat + c0 * ld + k0;
        hipLaunchKernelGGL((k_diag64<T>), dim3(1), dim3(DG_NT), 0, st,
                           kmat + c0 * ld + c0, ld, sw, lrow, kprev, inv, info, (int)c0);
        CIMRGP_LAUNCH_CHECK(fn);
        const int64_t m1 = n - pc;
        const int nb1 = (int)((m1 + TR - 1) / TR);
        if (nb1 > 0) {
            hipLaunchKernelGGL((k_trsm64<T>), dim3((unsigned)nb1), dim3(256), 0, st,
                               kmat + pc * ld + c0, ld, (int)m1, nb1, (T*)nullptr, (int64_t)0, 0,
                               sw, kprev, lrow, ld, (const T*)inv);
            CIMRGP_LAUNCH_CHECK(fn);
        }
    }
    return 0;
}
}  // namespace

// Workspace layout: [ceil(n/64) slabs of 64x64 inverses][ceil(n/256) blocks of 256x256 invT].
template <typename T>
static int build_invT(const T* kmat, int64_t n, int64_t ld, T* ws, hipStream_t st)
{
    const char* fn = "cimrgp_potrf";
    const int64_t nslab = (n + SB - 1) / SB, npan = (n + CIMRGP_NB - 1) / CIMRGP_NB;
    T* invT = ws + nslab * (SB * SB);
    hipLaunchKernelGGL((k_invT_init<T>), dim3(CIMRGP_NB * CIMRGP_NB / 256, (unsigned)npan), dim3(256), 0, st, invT, (int)n);
    CIMRGP_LAUNCH_CHECK(fn);
    for (int s = 0; s < CIMRGP_NB / SB; ++s) {
        hipLaunchKernelGGL((k_invT_step<T>), dim3(CIMRGP_NB / TR, (unsigned)npan), dim3(256), 0, st,
                           invT, kmat, ld, (int)n, (const T*)ws, s);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    return 0;
}

#define CIMRGP_HIP_TRY(call, what) \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) return check_hip(e__, "cimrgp_potrf", what); } while (0)

template <typename T>
int potrf_run(T* k, int64_t n, int64_t ld, T* ws, int32_t* info, T* b, int64_t m, int64_t ldb, hipStream_t st)
{
    const bool rows = (b != nullptr && m > 0);
    CIMRGP_HIP_TRY(hipMemsetAsync(info, 0, sizeof(int32_t), st), "hipMemsetAsync(info)");
    const int64_t npanels = (n + CIMRGP_NB - 1) / CIMRGP_NB;
    LookAhead* la = (npanels > 2) ? lookahead_ctx((size_t)(4 * npanels + 8)) : nullptr;
    if (la == nullptr) {
        int rc0 = panel_sweep<T, true>(k, n, ld, ws, info, b, m, ldb, st);
        return rc0 ? rc0 : build_invT<T>(k, n, ld, ws, st);
    }

    hipStream_t sp = la->side;
    size_t ne = 0;
    // The side stream runs the whole latency-bound chain in stream order -- "head" update of the
    // next panel's columns, then that panel's factorisation -- so that no inter-queue signal
    // sits between two links of the chain; the caller's stream runs the bulk of each trailing
    // update (and, off the chain, the carried rows).  Cross-stream edges: "panel final"
    // (side -> main, before the bulk update that reads it) and "bulk update done" (main -> side,
    // before the next head touches columns the bulk update wrote).
    // (Reserving CUs for the chain with a CU-masked bulk stream was measured and rejected: a
    // masked queue ran the trailing update 20 % slower even with 8 of 256 CUs masked.)
    hipEvent_t ev_start = la->ev[ne++];
    CIMRGP_HIP_TRY(hipEventRecord(ev_start, st), "hipEventRecord");
    CIMRGP_HIP_TRY(hipStreamWaitEvent(sp, ev_start, 0), "hipStreamWaitEvent");
    int rc = factor_panel<T>(k, n, ld, ws, info, 0, (n < CIMRGP_NB) ? n : CIMRGP_NB, sp);
    if (rc) return rc;
    hipEvent_t ev_panel = la->ev[ne++];
    CIMRGP_HIP_TRY(hipEventRecord(ev_panel, sp), "hipEventRecord");
    hipEvent_t ev_rest = nullptr;                      // bulk update of the previous panel
    int64_t rows_next = 0;                             // first panel the carried rows have not seen yet
    for (int64_t k0 = 0; k0 < n; k0 += CIMRGP_NB) {
        const int64_t w  = (n - k0 < CIMRGP_NB) ? (n - k0) : CIMRGP_NB;
        const int64_t k1 = k0 + w;
        CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_panel, 0), "hipStreamWaitEvent");   // panel k0 is final
        hipEvent_t ev_final = nullptr;                 // the same fact, for the carried rows' queue
        if (rows && la->rows) {
            ev_final = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_final, st), "hipEventRecord");
        }
        if (k1 < n) {
            const int64_t wn = (n - k1 < CIMRGP_NB) ? (n - k1) : CIMRGP_NB;   // next panel
            const int64_t k2 = k1 + wn;
            // chain: head (columns of the next panel, all rows below), then the next panel
            if (ev_rest) CIMRGP_HIP_TRY(hipStreamWaitEvent(sp, ev_rest, 0), "hipStreamWaitEvent");
            rc = gemm_nt_sub<T>(k + k1 * ld + k1, ld, k + k1 * ld + k0, ld, k + k1 * ld + k0, ld,
                                n - k1, wn, (int)w, false, sp);
            if (rc) return rc;
            rc = factor_panel<T>(k, n, ld, ws, info, k1, wn, sp);
            if (rc) return rc;
            ev_panel = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_panel, sp), "hipEventRecord");
            // bulk: lower SYRK beyond the next panel, concurrently with the chain
            ev_rest = nullptr;
            if (n > k2) {
                const double mm = (double)(n - k2);
                TrailRec* rec = rec_open(st, mm * (mm + 1.0) * (double)w);
                rc = gemm_nt_sub<T>(k + k2 * ld + k2, ld, k + k2 * ld + k0, ld, k + k2 * ld + k0, ld,
                                    n - k2, n - k2, (int)w, true, st);
                if (rec) (void)hipEventRecord(rec->stop, st);
                if (rc) return rc;
                ev_rest = la->ev[ne++];
                CIMRGP_HIP_TRY(hipEventRecord(ev_rest, st), "hipEventRecord");
            }
        }
        if (rows) {
            // Panel k0 is final: solve + update the carried rows.  They form their own chain (panel
            // p+1 of the rows needs panel p of the rows) that depends on the factorisation only
            // through "panel k0 final", so it runs on a third queue and lags behind: nothing of it
            // is issued while the trailing updates are still large (that phase is MFMA-bound and
            // the rows would only take compute units away from the critical path); from then on it
            // fills the compute units the latency-bound panel chain leaves idle.
            hipStream_t sq = la->rows ? la->rows : st;
            const bool defer = (sq != st) && (n - k1 > ROWS_START_BELOW) && (k1 < n);
            if (!defer) {
                if (sq != st) CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_final, 0), "hipStreamWaitEvent");
                for (int64_t r0 = rows_next; r0 <= k0; r0 += CIMRGP_NB) {
                    const int64_t rw = (n - r0 < CIMRGP_NB) ? (n - r0) : CIMRGP_NB;
                    const int64_t r1 = r0 + rw;
                    hipLaunchKernelGGL((k_trsm256<T>), dim3((unsigned)((m + TR - 1) / TR)), dim3(256), 0, sq,
                                       b + r0, ldb, (int)m, (int)rw, (const T*)(k + r0 * ld + r0), ld,
                                       (const T*)(ws + (r0 / SB) * (SB * SB)));
                    CIMRGP_LAUNCH_CHECK("cimrgp_potrf_rows");
                    if (n > r1) {
                        rc = gemm_nt_sub<T>(b + r1, ldb, b + r0, ldb, k + r1 * ld + r0, ld, m, n - r1, (int)rw, false, sq);
                        if (rc) return rc;
                    }
                }
                rows_next = k1;
            }
        }
    }
    if (rows && la->rows) {
        hipEvent_t ev_rows_done = la->ev[ne++];
        CIMRGP_HIP_TRY(hipEventRecord(ev_rows_done, la->rows), "hipEventRecord");
        CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_rows_done, 0), "hipStreamWaitEvent");
    }
    return build_invT<T>(k, n, ld, ws, st);
}

template <typename T>
int solve_rows_run(const T* l, int64_t n, int64_t ld, const T* ws, T* b, int64_t m, int64_t ldb, hipStream_t st)
{
    return panel_sweep<T, false>(const_cast<T*>(l), n, ld, const_cast<T*>(ws), nullptr, b, m, ldb, st);
}

template int potrf_run<double>(double*, int64_t, int64_t, double*, int32_t*, double*, int64_t, int64_t, hipStream_t);
template int potrf_run<float>(float*, int64_t, int64_t, float*, int32_t*, float*, int64_t, int64_t, hipStream_t);
template int solve_rows_run<double>(const double*, int64_t, int64_t, const double*, double*, int64_t, int64_t, hipStream_t);
template int solve_rows_run<float>(const float*, int64_t, int64_t, const float*, float*, int64_t, int64_t, hipStream_t);

}  // namespace cimrgp
