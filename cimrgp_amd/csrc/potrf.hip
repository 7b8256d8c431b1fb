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
#include <functional>
#include "gemm_tile.hpp"
#include <cstdlib>
#include <mutex>

#include <vector>

namespace cimrgp {

// ---- optional per-launch timing of the trailing update (bench.py roofline) ----
// Events are recorded on the launch stream around every lower-triangular
// trailing-update launch while profiling is on; collect() waits for them.
namespace {
struct TrailRec { hipEvent_t start, stop; double flops, bytes; };
std::vector<TrailRec> g_recs;
std::vector<TrailRec> g_free;
bool g_profile = false;
std::mutex g_profile_mutex;            // factorisations on different caller streams may be enqueued by different host threads
}  // namespace

int profile_begin()
{
    std::lock_guard<std::mutex> guard(g_profile_mutex);
    g_profile = true;
    return 0;
}

// Stops recording without touching the records so far (profile_begin resumes): bench.py brackets the launches of
// every fourth step only -- two event records per launch on the update queue cost ~1 % of an N = 8192 step.
int profile_pause()
{
    std::lock_guard<std::mutex> guard(g_profile_mutex);
    g_profile = false;
    return 0;
}

int profile_collect(double* total_ms, double* total_flops, int64_t* launches, double* total_bytes)
{
    std::lock_guard<std::mutex> guard(g_profile_mutex);
    double ms = 0.0, fl = 0.0, by = 0.0;
    int64_t cnt = 0;
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.stop);
        if (e != hipSuccess) return check_hip(e, "cimrgp_profile_collect", "hipEventSynchronize");
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.start, r.stop);
        if (e != hipSuccess) return check_hip(e, "cimrgp_profile_collect", "hipEventElapsedTime");
        ms += t; fl += r.flops; by += r.bytes; ++cnt;
        g_free.push_back(r);
    }
    g_recs.clear();
    g_profile = false;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = cnt;
    return 0;
}

// Opens a record (start event on `st`) and returns its stop event, to be recorded behind the launch;
// nullptr when profiling is off.
// flops = M (M + 1) K of a lower update; bytes = its algorithmic traffic: C (lower) read and written, the panel once
static hipEvent_t rec_open(hipStream_t st, double flops, double bytes = 0.0)
{
    std::lock_guard<std::mutex> guard(g_profile_mutex);
    if (!g_profile) return nullptr;
    TrailRec r;
    if (!g_free.empty()) { r = g_free.back(); g_free.pop_back(); }
    else {
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return nullptr;
    }
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.start, st);
    g_recs.push_back(r);
    return r.stop;
}

namespace {

constexpr int SB = 64;   // diagonal sub-block
// Far part of the trailing matrix updated once per group of panels while larger than this.
// Measured: whole potrf at N = 65536 1650 -> 1505 ms (56.9 -> 62.3 TF/s); at N = 8192 pairing
// (thresholds 2048..6144) raises the update kernel's rate (49 -> 56 % of peak) but not the
// end-to-end time (longer-lived update workgroups, longer slot waits of the chain), so it
// starts above that size.

constexpr int64_t ROWS_PAIR_ABOVE_SOLVE = 1024;   // stand-alone row-wise solve: pair the updates while more columns remain

// (the schedule's thresholds are `knobs()`, common.hpp: constants in the product build)

// 16 bytes in flight between global memory and LDS.  A first-class vector: arrays of HIP's uint4
// struct filled from global memory stay in scratch (the optimiser does not split the struct copy).
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ v4u v4u_zero() { v4u z = {0u, 0u, 0u, 0u}; return z; }

template <typename T> struct Tile64 {
    static constexpr int ROWB  = SB * (int)sizeof(T);   // bytes of 64 k-values per row
    static constexpr int LROW  = ROWB + 16;              // padded LDS row stride
    static constexpr int CPR   = ROWB / 16;              // 16-byte chunks per row
    static constexpr int NSTEP = ROWB / 32;              // slot-steps (4 slots x 8 B) per 64 k
    static constexpr int KPS   = 32 / (int)sizeof(T);    // k values per slot-step
    static constexpr int BYTES = SB * LROW;
};

// mask_chunk for the first-class vector: zero the elements whose k index is >= kvalid
template <typename T> static __device__ __forceinline__ v4u mask_v4u(v4u v, int kfirst, int kvalid)
{
    const uint4 m = mask_chunk<T>(make_uint4(v.x, v.y, v.z, v.w), kfirst, kvalid);
    v4u o = {m.x, m.y, m.z, m.w};
    return o;
}

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
        v4u v = v4u_zero();
        if (r < mrows && kcol < kw) {
            v = *reinterpret_cast<const v4u*>(src + (int64_t)r * ld + kcol);
            if (kcol + X::EPC > kw) v = mask_v4u<T>(v, kcol, kw);
        }
        *reinterpret_cast<v4u*>(dst + r * TL::LROW + c * 16) = v;
    }
}

// Same strip, split in two halves so the global loads of the next k chunk can be in
// flight (in registers) while the current chunk is multiplied.
template <typename T, int ROWS>
static __device__ __forceinline__ void gload_tile64(v4u (&regs)[ROWS * Tile64<T>::CPR / 256], const T* __restrict__ src,
                                                     int64_t ld, int mrows, int kw)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
#pragma unroll
    for (int p = 0; p < ROWS * TL::CPR / 256; ++p) {
        const int e = threadIdx.x + 256 * p;
        const int r = e / TL::CPR, c = e - r * TL::CPR;
        const int kcol = c * X::EPC;
        v4u v = v4u_zero();
        if (r < mrows && kcol < kw) {
            v = *reinterpret_cast<const v4u*>(src + (int64_t)r * ld + kcol);
            if (kcol + X::EPC > kw) v = mask_v4u<T>(v, kcol, kw);
        }
        regs[p] = v;
    }
}

template <typename T, int ROWS>
static __device__ __forceinline__ void swrite_tile64(unsigned char* dst, const v4u (&regs)[ROWS * Tile64<T>::CPR / 256])
{
    using TL = Tile64<T>;
#pragma unroll
    for (int p = 0; p < ROWS * TL::CPR / 256; ++p) {
        const int e = threadIdx.x + 256 * p;
        const int r = e / TL::CPR, c = e - r * TL::CPR;
        *reinterpret_cast<v4u*>(dst + r * TL::LROW + c * 16) = regs[p];
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
    // unrolled by 4 only: fully unrolled, the 80 fragment reads of a chunk are all issued up front and
    // hold 160 registers (the four-wave kernels must stay within 256 to fit beside an update workgroup)
#pragma unroll 4
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

// hardware reciprocal estimate (v_rcp_f64 / v_rcp_f32): the pivot recurrence refines it itself
static __device__ __forceinline__ double rcp_seed(double d) { return __builtin_amdgcn_rcp(d); }
static __device__ __forceinline__ float rcp_seed(float d) { return __builtin_amdgcn_rcpf(d); }

// ---------------------------------------------------------------------------
// Riders (round 3).  The launches of a panel's chain are latency-bound: one workgroup factors a 64 x 64
// block for 17-31 us while the 30-250 panel-solve workgroups beside it are gone after ~10 us and most
// compute units idle.  In the one-queue sweeps (small matrices, batches, the tail of a large
// factorisation) the trailing updates therefore no longer get launches of their own: they are cut into
// 64 x 64 tiles (gemm_tile, W = 2) that ride as extra workgroups of the chain's launches -- the update by
// the PREVIOUS panel in the launches of this panel's chain, and the update of the next panel's columns by
// this panel sub-block by sub-block (K = 64) as soon as a sub-block is final, so that the next chain
// never waits for a "head" update either (fused_sweep has the schedule).  One queue, no inter-queue
// signal, and the chain's workgroup 0 is dispatched first in its launch.
// ---------------------------------------------------------------------------
template <typename T> struct RiderJob {
    T* c; const T* a; const T* b;      // C[m x n] -= A[m x k] B[n x k]^T
    int64_t ldc, lda, ldb;
    int m, n, k;
    int lower;                          // 1: lower-triangular grid of tiles (square C), 0: rectangular
    int tiles_n;                        // tiles per row of the rectangular grid
    int first, count;                   // this launch runs tiles [first, first + count) of the job
    int skip00;                         // tile (0, 0) is left alone: the chain's workgroup 0 owns it
    int rows_job;                       // 1: c and a are carried rows (batch stride sb), 0: the matrix (stride sk)
};
constexpr int MAX_RIDER_JOBS = 6;
template <typename T> struct Riders {
    RiderJob<T> job[MAX_RIDER_JOBS];
    int njobs;
    int total;                          // sum of the jobs' counts = extra workgroups of the launch
};
template <typename T> static Riders<T> no_riders() { Riders<T> r; r.njobs = 0; r.total = 0; return r; }

constexpr int RIDER_LDS = 4 * 64 * (128 + 16);      // gemm_tile<.., W = 2>: 2 stages x 2 operands x 64 rows x 144 bytes
// LDS of a chain kernel that also hosts riders: the larger of the panel solve's area and a rider's (f32: the rider's)
template <typename T> struct ChainLds {
    static constexpr int TRSM = (2 * 32 + 64) * (64 * (int)sizeof(T) + 16);          // = TrsmLds<T>::BYTES
    static constexpr int BYTES = TRSM > RIDER_LDS ? TRSM : RIDER_LDS;
};

// Workgroup number r (0 <= r < rd.total) of a launch's riders; the first 256 threads of the workgroup.
template <typename T>
static __device__ __forceinline__ void run_rider(unsigned char* smem, const Riders<T>& rd, int r, int64_t sk, int64_t sb)
{
    int j = 0;
    while (j + 1 < rd.njobs && r >= rd.job[j].count) { r -= rd.job[j].count; ++j; }       // uniform
    const RiderJob<T>& jb = rd.job[j];
    const int id = jb.first + r;
    int ti, tj;
    if (jb.lower) {
        ti = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
        while (ti * (ti + 1) / 2 > id) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
        tj = id - ti * (ti + 1) / 2;
    } else {
        ti = id / jb.tiles_n;
        tj = id - ti * jb.tiles_n;
    }
    if (jb.skip00 && ti == 0 && tj == 0) return;
    const int64_t off_ca = (int64_t)blockIdx.y * (jb.rows_job ? sb : sk);
    T* c = jb.c + off_ca;
    const T* a = jb.a + off_ca;
    const T* b = jb.b + (int64_t)blockIdx.y * sk;
    constexpr int BKE = 128 / (int)sizeof(T);
    const bool interior = (ti + 1) * 64 <= jb.m && (tj + 1) * 64 <= jb.n && (jb.k % BKE) == 0;
    if (jb.lower) {
        if (interior) gemm_tile<T, true, false, 2>(smem, c, jb.ldc, a, jb.lda, b, jb.ldb, jb.m, jb.n, jb.k, ti, tj);
        else          gemm_tile<T, true, true, 2>(smem, c, jb.ldc, a, jb.lda, b, jb.ldb, jb.m, jb.n, jb.k, ti, tj);
    } else {
        if (interior) gemm_tile<T, false, false, 2>(smem, c, jb.ldc, a, jb.lda, b, jb.ldb, jb.m, jb.n, jb.k, ti, tj);
        else          gemm_tile<T, false, true, 2>(smem, c, jb.ldc, a, jb.lda, b, jb.ldb, jb.m, jb.n, jb.k, ti, tj);
    }
}

// ---------------------------------------------------------------------------
// Diagonal 64x64 sub-block of a panel (left-looking inside the panel):
//   S = A_ss - Lrow Lrow^T        Lrow = the kprev panel columns left of the
//                                 block, already final (MFMA, K = kprev <= 192)
//   S = L L^T  in place, inv slab = L^-1 (lower; zero elsewhere).
// Factor and inverse advance together on the unscaled Schur complement S and the
// unscaled inverse Mi: column j of L is S[:,j] r_j and row j of L^-1 is
// Mi[j,:] r_j, r_j = 1/sqrt(S[j][j]); the row operations that reduce S are
// applied to Mi.
// ---------------------------------------------------------------------------
// (lds_barrier, lds_settle: common.hpp)
#ifdef RACE_NOPRIO          /* tools/lab/race_probe.hip only: the chain kernels at the default wave priority */
#define CHAIN_SETPRIO() do { } while (0)
#else
#define CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(3)
#endif
#ifdef CIMRGP_RACE_PDUMP    /* tools/lab/race_probe.hip only, the lightest record: what the pivot wave READ as gathered, per block (one 16-byte store per lane) */
__device__ float* g_race_pbuf;                // [4 sub-blocks of the panel][16 blocks][64 lanes][4]
#endif
// S and Mi are held as ONE combined 64x64 array A:
//     A[i][k] = S[i][k]   for k <= i   (Schur complement, lower triangle)
//     A[i][k] = Mi[k][i]  for k >  i   (unscaled inverse, stored transposed in the upper triangle)
// At pivot j everything needed is column j of A: A[i][j] is S[i][j] for i >= j and
// Mi[j][i] for i < j, and
//     A[i][k] -= (A[k][j] r) * h_i   for every k > j,    r = 1/sqrt(A[j][j]),
//     h_i = L[i][j] = A[i][j] r (i > j),  L^-1[j][i] = A[i][j] r (i < j),  r (i = j, from 0)
// covers the Schur update, the inverse update and the birth of column j of Mi in one
// formula (for j < i < k it touches a not-yet-born Mi slot, which is reset at pivot i).
constexpr int DG_TW = 8;              // tile waves of the diagonal kernel (two 16x16 tiles each)
constexpr int DG_NW = DG_TW + 1;      // + the pivot wave
constexpr int DG_NT = 64 * DG_NW;

// One 16x16x4 matrix-core step (K = 4 = the pivots of one block).
static __device__ __forceinline__ Mx<double>::acc_t mfma_k4(double a, double b, Mx<double>::acc_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ Mx<float>::acc_t mfma_k4(float a, float b, Mx<float>::acc_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Version 3 of the diagonal kernel.  The combined array A (see above) lives in MATRIX-CORE
// ACCUMULATOR layout in eight "tile" waves: wave g < 8 owns the two 16x16 tiles (row tile g >> 1,
// column tiles 2 (g & 1) + {0, 1}).  A ninth wave, the pivot wave, works in "lane = row" layout
// on four columns at a time.  Per block p of BC = 4 pivots, ONE barrier:
//   pivot wave   takes block p's columns (gathered by the tile waves one iteration earlier, i.e.
//                updated through block p-2), applies block p-1's rank-4 update to them itself
//                (16 FMAs per lane, coefficients by v_readlane from its own registers), eliminates
//                the 4 pivots among themselves (wave-local broadcasts, 4 dependent rsqrt/scale
//                steps) and publishes the rank-4 update's two operands (64 x 4 each);
//   tile waves   meanwhile apply block p-1's update to their tiles with one matrix-core
//                instruction per tile, then gather block p+1's columns for the pivot wave.
// The pivot-time columns, kept for all 64 pivots, and the 64 reciprocal roots ARE the result:
// L[i][j] = cs[i][j] r_j (i >= j), L^-1[j][i] = cs[i][j] r_j (i < j), L^-1[j][j] = r_j.
// (Version 1 kept A in "lane = row" registers in all waves and applied the rank-4 update with
// scalar FMAs whose per-column coefficients every wave read as LDS broadcasts: 2 MB of LDS
// return traffic per block, 1.1 us per 4 pivots of which the pivots themselves were 0.24 us.)
// One block of BC = 4 pivots in the pivot wave (lane = row i), shared by the nine-wave and the four-wave
// factorisation: takes the block's columns `nx` as gathered (updated through block p-3; rows of blocks p-2 and
// p-1 gathered as ZERO: their slots restart there), applies the rank-4 updates of blocks p-2 and p-1 to them
// itself, eliminates the 4 pivots and publishes the rank-4 update's two operands (hs_row: this row's left operand, UNMASKED --
// the tile waves zero the three entries of a block's own rows that lie below their pivots; cs: the
// pivot-time columns, kept for all 64 pivots).
//
// Round 4: this wave is ISSUE-bound, not latency-bound.  A lone wave issues one instruction per ~5 clocks
// (tools/diag_probe.hip: 4 pivots = 750-790 clocks with an 11-deep dependent chain per pivot and just the
// same with a 5-deep one), and an iteration of rounds 1-3 was ~300 instructions of this wave against ~500
// clocks of the tile waves' work.  So everything that is not the recurrence left the loop:
//   * the recurrence needs -A[i][j] / d only: a reciprocal (seed + three fused steps, x0 (1 + e)(1 + e^2)),
//     no root.  The reciprocal roots -- which only scale the OUTPUT -- and the first non-positive pivot are
//     taken after the loop from the pivots themselves, d_j = cs[j][j] (a pivot-time column is final from
//     its own pivot on): diag_roots;
//   * row j itself takes the generic -A[j][j] / d = -1 instead of -1/d, i.e. column i of the unscaled
//     inverse is kept scaled by d_i (its birth needs no division and no select; the recurrence of an
//     inverse column is linear in it; the epilogue multiplies by r_i^2): no special case for lane j;
//   * "restart from zero" of a slot is one select on the high word (tiny_if): as an addend a double
//     below 2^-1042 is zero.
template <typename T> static __device__ __forceinline__ T tiny_if(T v, bool c);
template <> __device__ __forceinline__ double tiny_if<double>(double v, bool c)
{
    return __hiloint2double(c ? 0 : __double2hiint(v), __double2loint(v));
}
template <> __device__ __forceinline__ float tiny_if<float>(float v, bool c) { return c ? 0.0f : v; }

template <typename T>
static __device__ __forceinline__ void pivot_block(int p, T (&nx)[4], T (&cv)[4], T (&hsr)[4], T (&hsr2)[4],
                                                    T (&sm1)[4][4], T (&sm2)[4][4],
                                                    T* __restrict__ hs_row, T* __restrict__ cs, int i)
{
    constexpr int LS = SB + 2;
    constexpr int BC = 4;
    constexpr int NP = SB / BC;
    const int j0 = BC * p;
    // The 4 x 4 coefficients of an earlier block's rank-4 update = its pivot-time columns at the rows of block p,
    // published in `cs` by THIS wave: sm1 (block p-1) and sm2 (block p-2) were requested at
    // the end of the previous block, behind its publishing stores and ahead of its barrier (below).  One chain of fused
    // operations per column (fewest instructions).
#ifndef CIMRGP_GATHER_LATE         /* the gathered columns carry the updates through block p-3: two self-updates here */
    // Rows of block p-2 and p-1 arrive as zero (their slots restart there); a row of block p-1 takes nothing from
    // block p-2 (not born yet: reset between the two updates).
    if (p > 1) {
        const bool rows_prev = (unsigned)(i - (j0 - BC)) < (unsigned)BC;
#pragma unroll
        for (int t2 = 0; t2 < BC; ++t2) {
            T u = nx[t2];
#pragma unroll
            for (int t = 0; t < BC; ++t) u = fma(hsr2[t], sm2[t2][t], u);
            nx[t2] = tiny_if<T>(u, rows_prev);
        }
    }
#endif
    if (p > 0) {
#pragma unroll
        for (int t2 = 0; t2 < BC; ++t2) {
            T u = nx[t2];
#pragma unroll
            for (int t = 0; t < BC; ++t) u = fma(hsr[t], sm1[t2][t], u);
            nx[t2] = u;
        }
    }
#pragma unroll
    for (int t = 0; t < BC; ++t) cv[t] = nx[t];
    if (p == 8) STAMPW(17, DG_TW);
    T nh[BC];
#pragma unroll
    for (int t = 0; t < BC; ++t) {
        const int j = j0 + t;
        const T d = bcast_lane(cv[t], j);
        const T x0 = rcp_seed(d);
        const T e = fma(-d, x0, (T)1);
        const T sx0 = -cv[t] * x0;
        const T sx1 = fma(sx0, e, sx0);
        nh[t] = fma(sx1, e * e, sx1);                                // -A[i][j] / d  (row j: -1)
#pragma unroll
        for (int t2 = t + 1; t2 < BC; ++t2) {
            const T ak = bcast_lane(cv[t], j0 + t2);                // A[k][j] at pivot time
            cv[t2] = fma(nh[t], ak, tiny_if<T>(cv[t2], i == j));    // row j: its slots right of the pivot are born here
        }
    }
    if (p == 8) STAMPW(18, DG_TW);
    T* cp = &cs[i * LS + j0];
#pragma unroll
    for (int t = 0; t < BC; ++t) {
        hs_row[t] = nh[t];
        cp[t] = cv[t];
    }
    // Round 5: the NEXT block's coefficients are requested here, behind the publishing stores and ahead of the barrier.
    // (1) They are this wave's own words (rows j0+4 .. j0+7 of the columns just published and of the block before), so
    //     nothing is waited for that is not there; behind the barrier only the gathered columns remain to be read.
    // (2) The barrier's wait for these LOADS is what makes the stores above visible to the tile waves that read `hs`
    //     and `cs` right behind the barrier: the LDS executes a wave's accesses in order, so the loads' data returns
    //     only after the stores have been performed -- the wait for a store's lgkmcnt alone does not imply that
    //     (lds_settle in common.hpp has the measurement).  The last block settles on its last word instead.
    if (p + 1 < NP) {
        const T* sp = &cs[(j0 + BC) * LS + j0];
#pragma unroll
        for (int t2 = 0; t2 < BC; ++t2)
#pragma unroll
            for (int t = 0; t < BC; ++t) sm1[t2][t] = sp[t2 * LS + t];
#ifndef CIMRGP_GATHER_LATE
        if (p > 0) {
#pragma unroll
            for (int t2 = 0; t2 < BC; ++t2)
#pragma unroll
                for (int t = 0; t < BC; ++t) sm2[t2][t] = sp[t2 * LS + t - BC];
        }
#endif
    } else {
        lds_settle(&cp[BC - 1]);
    }
    // this wave's own copies of the left operand, for the next blocks' self-updates: a row of THIS block takes
    // nothing from the pivots above it (its slots right of the block are born at its own pivot)
#pragma unroll
    for (int t = 0; t < BC; ++t) hsr2[t] = hsr[t];
    const unsigned u = (unsigned)(i - j0);
    hsr[0] = tiny_if<T>(nh[0], u - 1u < 3u);
    hsr[1] = tiny_if<T>(nh[1], u - 2u < 2u);
    hsr[2] = tiny_if<T>(nh[2], u == 3u);
    hsr[3] = nh[3];
    if (p == 8) STAMPW(19, DG_TW);
}

// After the pivot loop (all waves past a barrier): the reciprocal roots r_j = 1 / sqrt(d_j) from the pivots
// d_j = cs[j][j], and the first non-positive pivot among the block's w columns.  One wave, lane = j.
template <typename T>
static __device__ __forceinline__ void diag_roots(const T* __restrict__ cs, T* __restrict__ rall, int lane, int w,
                                                   int32_t* info, int col_base)
{
    constexpr int LS = SB + 2;
    const T d = cs[lane * LS + lane];
    rall[lane] = rsqrt_refined<T>(d);
    lds_settle(&rall[lane]);                 // every wave reads the roots right behind the next barrier
    const unsigned long long badmask = __ballot(lane < w && !(d > (T)0));
    if (badmask != 0ull && lane == 0) atomicCAS(info, 0, col_base + __ffsll((long long)badmask));
}

// The factorisation proper, shared by all four chain kernels.  On entry `cs` holds the Schur complement S
// (lower triangle, identity padding beyond w, zeros above the diagonal) and every wave has passed a barrier
// behind its writer; `pcol`, `hs`, `rall` are LDS areas nobody reads any more.  Writes L into D (lower) and
// L^-1 into `inv`.
//   tw >= 0   tile wave number tw of NTW: owns the 16 x 16 tiles t = tw + NTW k of the combined array
//             (t = 4 row-tile + column-tile) in matrix-core accumulator layout;
//   pivot     the pivot wave (pivot_block);
//   neither   keeps the barriers only.
// Round 4: WHICH SIMD the pivot wave shares matters more than how many tile waves there are.  The vector ALU
// of a SIMD issues one wave-wide instruction per 4 clocks whatever wave it comes from, and the pivot wave's
// ~140 instructions per block of 4 pivots are the critical path: with two tile waves on its SIMD (rounds 1-3:
// nine waves, waves 0 / 4 / 8 on SIMD 0) an iteration took the SUM of the pivot wave's and the tile waves'
// issue time (tools/diag_probe.hip, 200 launches back to back: 13.5 us per kernel; tile waves idle 10.1;
// pivot wave idle 7.5).  So the nine-wave kernels leave waves 0 and 4 idle in the loop and deal the sixteen
// tiles to the six waves of the other three SIMDs (nine_tile_wave), the four-wave kernels keep three tile
// waves beside a pivot wave with a SIMD of its own.
// Tile ownership: a tile wave owns tiles of ONE 16-column block BCOL (two in the four-wave form's last wave), rows
// BR0 .. BR0 + NB - 1.  All its tiles then share the right operand, are final together (one uniform early-out per
// block of pivots), and its step is straight-line code: every operand is requested before the first multiply
// (with the sixteen tiles dealt round-robin, each tile sat behind its own branches and paid its own LDS round
// trip: 240 clocks per tile, 14.3 us per kernel with the pivot wave idle against 7.5 for two tiles per wave).
// Each role runs its OWN copy of the pivot loop (same number of barriers): with the roles told apart inside one
// loop the compiler moved the accumulators between per-branch register assignments every iteration.
//   nine-wave kernels (6 tile waves): column 3: rows 0-1 | 2-3, column 2: rows 0-1 | 2-3, column 1: all, column 0: all
//   four-wave kernels (3 tile waves): column 3 | column 2 | columns 1 and 0
template <typename T, int NB, int BCOL, int BR0>
struct TileGroup {
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    static constexpr int LS = SB + 2;
    static constexpr int BC = 4;
    static constexpr int NP = SB / BC;
    acc_t acc[NB > 0 ? NB : 1];
    __device__ __forceinline__ void load(const T* cs, int lane)
    {
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[k][r] = cs[((BR0 + k) * 16 + X::crow(lane, r)) * LS + BCOL * 16 + (lane & 15)];
    }
    // The gather of block p+2 for the pivot wave, THEN block p's rank-4 update of the group's tiles: the columns leave
    // with the updates through block p-1 and the pivot wave applies two blocks itself (its coefficients are in
    // registers by then: pivot_block).
    // Why this order (rounds 4 and 5; tools/lab/race_probe.hip, HISTORY.md).  With the gather BEHIND the multiplies --
    // rounds 1-3 -- FP32 factorisations beside a running FP32 update came out wrong in 1.7-3 % of the runs, always in
    // the last rows of a block's last columns.  Round 4 blamed matrix-core results landing late and moved the gather
    // ahead; round 5 found what it was: the gathering wave's LAST one or two ds_write instructions -- issued right
    // ahead of `s_waitcnt lgkmcnt(0); s_barrier` -- were not yet in the LDS array when the pivot wave read their
    // words right behind the barrier: it read, bit for bit, what the words held two blocks earlier.  1024 idle cycles
    // between the multiplies and the stores changed nothing (the matrix cores were never late); reading the last
    // stored word back ahead of the barrier cured it (0 of 1999 runs against 60 of 1999).  The LDS executes one wave's
    // accesses in issue order, so a LOAD's returned data proves the wave's earlier stores performed; a store's own
    // lgkmcnt does not prove them visible to another wave.  Here the operand loads below ARE that proof: they are issued
    // behind the gather's stores (the compiler-only fence keeps them there) and the multiplies need their data, so the
    // stores have been performed long before this wave reaches the barrier -- by construction, not by timing.
    // (The same read-back behind the multiplies, -DCIMRGP_GATHER_LATE, is correct too and one self-update shorter in
    // the pivot wave, but puts the load's round trip on the tile wave's path: 11.5 / 14.6 us per kernel against
    // 10.7 / 12.3 in this order: profiles/r05_chain_kernels.txt.)
    //   right operand = the pivot-time columns at this column block's rows (zero for columns that are final),
    //   left operand per tile = the pivot wave's -A[i][j] / d (a row of the block takes nothing from the pivots
    //   above it: the pivot wave publishes unmasked),
    //   rows of the block: their slots right of it restart from 0 (the one tile concerned: a uniform branch
    //   ahead of the operand reads, so that no join sits between the reads and the multiplies).
    // (no __restrict__ on these: `hs` and `cs` are rewritten by the PIVOT wave between the barriers, at addresses
    // that repeat every second block -- a tile wave that only reads them must not be told they are its own)
    __device__ __forceinline__ void step(int p, const T* hs, const T* cs, T* pcol, int lane)
    {
        if (NB == 0) return;
        const int j0 = BC * p, bc0 = j0 >> 4, jb = j0 & 15;
        if (BCOL < bc0) return;                                   // uniform: every column of the group is final
        const int fcol = lane & 15, fk = lane >> 4;
        const int col = BCOL * 16 + fcol;
        const int g0 = j0 + 2 * BC, gbc = g0 >> 4, gjb = g0 & 15;   // block p+2
#ifndef CIMRGP_GATHER_LATE
        if (p + 2 < NP && BCOL == gbc && fcol >= gjb && fcol < gjb + BC) {
            T* pc = pcol + ((p & 1) * SB) * BC + (fcol - gjb);
#pragma unroll
            for (int k = 0; k < NB; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // rows of blocks p and p+1 restart from zero in these columns (pivot_block adds to what it is given)
                    const int grow = (BR0 + k) * 16 + X::crow(lane, r);
                    pc[grow * BC] = (grow >= j0 && grow < g0) ? (T)0 : acc[k][r];
                }
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);      // compiler only: the operand loads below stay BEHIND the gather's stores
#endif
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (BR0 + k == bc0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rin = X::crow(lane, r);
                    if (rin >= jb && rin < jb + BC && col >= j0 + BC) acc[k][r] = (T)0;
                }
            }
        const T* hsp = hs + ((p & 1) * SB) * BC + fk;
        T bf = cs[col * LS + j0 + fk];
        T af[NB > 0 ? NB : 1];
#pragma unroll
        for (int k = 0; k < NB; ++k) af[k] = hsp[((BR0 + k) * 16 + fcol) * BC];
        if (col < j0 + BC) bf = (T)0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int arow = (BR0 + k) * 16 + fcol;
            if (arow > j0 + fk && arow < j0 + BC) af[k] = (T)0;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] = mfma_k4(af[k], bf, acc[k]);
#ifdef CIMRGP_GATHER_LATE     /* tools/lab/race_probe.hip: the hand-over right behind the multiplies (the columns leave updated through block p) */
#ifdef RACE_NOPS_AFTER       /* idle wait states (16 per s_nop 15) between the multiplies and the LDS writes of their results */
#pragma unroll
        for (int z = 0; z < RACE_NOPS_AFTER; ++z)
#pragma unroll
            for (int k = 0; k < NB; ++k) asm volatile("s_nop 15" : "+v"(acc[k]));
#endif
        if (p + 2 < NP && BCOL == gbc && fcol >= gjb && fcol < gjb + BC) {
            T* pc = pcol + ((p & 1) * SB) * BC + (fcol - gjb);
#pragma unroll
            for (int k = 0; k < NB; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = (BR0 + k) * 16 + X::crow(lane, r);
                    pc[grow * BC] = (grow >= j0 + BC && grow < g0) ? (T)0 : acc[k][r];
                }
#ifndef CIMRGP_RACE_UNSETTLED       /* (the probe reproduces the wrong results without this line) */
            lds_settle(pc + ((BR0 + NB - 1) * 16 + X::crow(lane, 3)) * BC);
#endif
        }
#endif
    }
};

// one tile wave's whole pivot loop (two groups; the second may be empty)
template <typename T, int NB, int BCOL, int BR0, int NB2 = 0, int BCOL2 = 0, int BR02 = 0>
static __device__ __forceinline__ void tile_wave_loop(const T* hs, const T* cs, T* pcol, int lane)
{
    TileGroup<T, NB, BCOL, BR0> ga;
    TileGroup<T, NB2, BCOL2, BR02> gb;
    ga.load(cs, lane);
    if (NB2 > 0) gb.load(cs, lane);
    lds_barrier();                                   // every wave has taken its share of S: cs may be overwritten
    for (int p = 0; p < SB / 4; ++p) {
        lds_barrier();
#if defined(DIAG_EXP) && DIAG_EXP == 1
        continue;
#endif
        ga.step(p, hs, cs, pcol, lane);
        if (NB2 > 0) gb.step(p, hs, cs, pcol, lane);
    }
}

template <typename T, int NTW, int NT>
static __device__ __forceinline__ void diag_tail_lds(int tw, bool pivot, T* __restrict__ pcol, T* __restrict__ hs, T* __restrict__ cs,
                                                      T* __restrict__ rall, T* __restrict__ D, int64_t ld,
                                                      int w, T* __restrict__ inv, int32_t* info, int col_base)
{
    static_assert(NTW == 6 || NTW == 3, "tile ownership below");
    constexpr int LS = SB + 2;       // even pitch: a row's 4 block columns are one 16-byte-aligned pair of stores
    constexpr int BC = 4;
    constexpr int NP = SB / BC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane;
    if (pivot) {
        // ---- pivot wave: the recurrence and nothing else
        T cv[BC], hsr[BC], hsr2[BC];                 // this row's block columns, left operands of the last two blocks
        T sm1[BC][BC], sm2[BC][BC];                  // coefficients of the self-updates, requested one block ahead (pivot_block)
#pragma unroll
        for (int t = 0; t < BC; ++t) {
            cv[t] = (T)0; hsr[t] = (T)0; hsr2[t] = (T)0;
#pragma unroll
            for (int t2 = 0; t2 < BC; ++t2) { sm1[t][t2] = (T)0; sm2[t][t2] = (T)0; }
        }
        T first[3][BC];                              // blocks 0, 1 and 2 straight from S (zero above the diagonal)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int t = 0; t < BC; ++t) first[b][t] = cs[i * LS + b * BC + t];
        lds_barrier();                               // every wave has taken its share of S: cs may be overwritten
        STAMP(2);
        for (int p = 0; p < NP; ++p) {
            if (p == 8) STAMPW(16, DG_TW);
            T nx[BC];
#ifdef CIMRGP_GATHER_LATE
            if (p < 2) {
#else
            if (p < 3) {
#endif
#pragma unroll
                for (int t = 0; t < BC; ++t) nx[t] = (p == 0) ? first[0][t] : (p == 1) ? first[1][t] : first[2][t];
            } else {
                // this row's share of block p as gathered (two 16-byte reads, in flight during the FMAs below)
                const T* gp = &pcol[((p & 1) * SB + i) * BC];
#pragma unroll
                for (int t = 0; t < BC; ++t) nx[t] = gp[t];
            }
#ifdef CIMRGP_RACE_PDUMP
            if (g_race_pbuf != nullptr && blockIdx.y == 0) {
                float* out = g_race_pbuf + ((((col_base >> 6) & 3) * 16 + p) * 64 + i) * 4;
#pragma unroll
                for (int t = 0; t < BC; ++t) out[t] = (float)nx[t];
            }
#endif
#if !defined(DIAG_EXP) || DIAG_EXP != 2      /* timing-only builds of tools/diag_probe.hip: 1 = tile waves idle, 2 = pivot wave idle */
            pivot_block<T>(p, nx, cv, hsr, hsr2, sm1, sm2, &hs[((p & 1) * SB + i) * BC], cs, i);
#endif
            lds_barrier();
            if (p == 8) STAMPW(20, DG_TW);
            if (p == 9) STAMPW(21, DG_TW);
        }
    } else if (tw < 0) {
        lds_barrier();
        for (int p = 0; p < NP; ++p) lds_barrier();
    } else if (NTW == 6) {
        if (tw == 0)      tile_wave_loop<T, 2, 3, 0>(hs, cs, pcol, lane);
        else if (tw == 1) tile_wave_loop<T, 2, 3, 2>(hs, cs, pcol, lane);
        else if (tw == 2) tile_wave_loop<T, 2, 2, 0>(hs, cs, pcol, lane);
        else if (tw == 3) tile_wave_loop<T, 2, 2, 2>(hs, cs, pcol, lane);
        else if (tw == 4) tile_wave_loop<T, 4, 1, 0>(hs, cs, pcol, lane);
        else              tile_wave_loop<T, 4, 0, 0>(hs, cs, pcol, lane);
    } else {
        if (tw == 0)      tile_wave_loop<T, 4, 3, 0>(hs, cs, pcol, lane);
        else if (tw == 1) tile_wave_loop<T, 4, 2, 0>(hs, cs, pcol, lane);
        else              tile_wave_loop<T, 4, 1, 0, 4, 0, 0>(hs, cs, pcol, lane);
    }
    STAMP(3);
    __syncthreads();
    if (tid < 64) diag_roots<T>(cs, rall, lane, w, info, col_base);
    __syncthreads();
    for (int e = tid; e < SB * SB; e += NT) {
        const int r = e >> 6, c = e & 63;
        if (r < w && c <= r) D[(int64_t)r * ld + c] = cs[r * LS + c] * rall[c];
        T v = (T)0;
        if (r < w && c < r) v = cs[c * LS + r] * rall[r] * (rall[c] * rall[c]);      // inverse column c is kept scaled by d_c
        if (r < w && c == r) v = rall[r];
        inv[e] = v;
    }
    STAMP(4);
}

// Nine-wave kernels: wave 8 is the pivot wave; it shares SIMD 0 with waves 0 and 4 (waves of a workgroup are
// dealt to the four SIMDs round-robin: tools/diag_probe.hip prints it), which therefore own nothing in the
// pivot loop; tile wave numbers 0..5 go to waves 1, 2, 3, 5, 6, 7.
constexpr int NINE_TW = 6;
static __device__ __forceinline__ int nine_tile_wave(int g) { return ((g & 3) == 0) ? -1 : (g < 4 ? g - 1 : g - 2); }

// Entry of the nine-wave kernels: the eight tile waves hold S in accumulator layout (wave g: row tile
// (g >> 1) & 3, column tiles 2 (g & 1) + {0, 1}) from their left-looking products; it goes to `cs` (dead on
// entry), and behind one barrier -- after which `pcol`, `hs`, `rall` must be dead too -- the pivot loop runs
// with the roles above.
template <typename T>
static __device__ __forceinline__ void diag_tail(typename Mx<T>::acc_t (&acc)[2], T* __restrict__ pcol, T* __restrict__ hs,
                                                  T* __restrict__ cs, T* __restrict__ rall, T* __restrict__ D, int64_t ld,
                                                  int w, T* __restrict__ inv, int32_t* info, int col_base)
{
    using X = Mx<T>;
    constexpr int LS = SB + 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave id: provably uniform
    if (g < DG_TW) {
        const int br = (g >> 1) & 3, ch = g & 1;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cs[(br * 16 + X::crow(lane, r)) * LS + (2 * ch + c) * 16 + (lane & 15)] = acc[c][r];
        lds_settle(&cs[(br * 16 + X::crow(lane, 3)) * LS + (2 * ch + 1) * 16 + (lane & 15)]);
    }
    lds_barrier();
    diag_tail_lds<T, NINE_TW, DG_NT>(nine_tile_wave(g), g == DG_TW, pcol, hs, cs, rall, D, ld, w, inv, info, col_base);
}

// LDS of the factorisation proper, in bytes: pivot columns (cs), gathered columns and left operand
// (double buffered), reciprocal roots.
template <typename T> struct DiagLds {
    static constexpr int CS   = SB * (SB + 2) * (int)sizeof(T);
    static constexpr int PCOL = 2 * SB * 4 * (int)sizeof(T);
    static constexpr int RALL = SB * (int)sizeof(T);
};

template <typename T>
__global__ __launch_bounds__(DG_NT)
void k_diag64(T* __restrict__ D, int64_t ld, int w, const T* __restrict__ Lrow, int kprev,
              T* __restrict__ inv, int32_t* info, int col_base, int64_t sk, int64_t sws, int64_t sb, Riders<T> rd)
{
    static_assert(Tile64<T>::BYTES <= RIDER_LDS, "the prologue chunk and a rider's tiles share one LDS area");
    __shared__ __attribute__((aligned(16))) unsigned char chunk[RIDER_LDS];           // Lrow chunk of the prologue / a rider's tiles
    if (blockIdx.x != 0) {                           // riders: update tiles in the shadow of the pivot loop
        if (threadIdx.x >= 256) return;
        run_rider<T>(chunk, rd, (int)blockIdx.x - 1, sk, sb);
        return;
    }
    // batch of independent factorisations (blocks of one layer): blockIdx.y selects the matrix
    D += (int64_t)blockIdx.y * sk;
    Lrow += (int64_t)blockIdx.y * sk;
    inv += (int64_t)blockIdx.y * sws;
    info += blockIdx.y;
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    constexpr int TT = 64 * DG_TW;                                            // threads of the tile waves
    static_assert(DG_TW == 8 && DG_NW == 9, "tile ownership below assumes 8 tile waves + 1 pivot wave");
    __shared__ __attribute__((aligned(16))) unsigned char pcol_[DiagLds<T>::PCOL];   // gathered pivot columns, double buffered
    __shared__ __attribute__((aligned(16))) unsigned char hs_[DiagLds<T>::PCOL];     // left operand (per row), double buffered
    __shared__ __attribute__((aligned(16))) unsigned char cs_[DiagLds<T>::CS];       // right operand = pivot-time columns, kept
    __shared__ __attribute__((aligned(16))) unsigned char rall_[DiagLds<T>::RALL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave id: provably uniform
    const bool tile_wave = g < DG_TW;
    const int br = (g >> 1) & 3, ch = g & 1;
    const int fcol = lane & 15;
    // latency-bound chain running next to MFMA-bound update workgroups: win issue arbitration
    CHAIN_SETPRIO();

    STAMP(0);
    // the block's own elements are requested first (they depend on nothing), so that their round
    // trip overlaps the left-looking update below instead of following it
    T dval[2][4];
    if (tile_wave) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                dval[c][r] = (row < w && col <= row) ? D[(int64_t)row * ld + col] : (T)0;
            }
    }
    // S = A_ss - Lrow Lrow^T, straight into the accumulator layout
    acc_t pacc[2];
    pacc[0] = acc_zero<T>();
    pacc[1] = acc_zero<T>();
    if (kprev > 0) {
        // kprev <= 192: all (up to three) 64-column chunks of Lrow are requested at once, so only
        // one global round trip is exposed; each then goes registers -> LDS -> matrix cores
        constexpr int NR = SB * TL::CPR / TT;         // 16-byte pieces per thread and chunk
        constexpr int MAXC = (CIMRGP_NB - SB) / SB;
        uint4 regs[MAXC][NR];
        if (tile_wave) {
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                if (q * SB < kprev) {
#pragma unroll
                    for (int p = 0; p < NR; ++p) {
                        const int e = tid + TT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                        regs[q][p] = (r < w) ? *reinterpret_cast<const uint4*>(Lrow + q * SB + (int64_t)r * ld + c * X::EPC)
                                             : make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            if (q * SB < kprev) {                      // uniform
                if (q) __syncthreads();                // the previous chunk has been consumed
                if (tile_wave) {
#pragma unroll
                    for (int p = 0; p < NR; ++p) {
                        const int e = tid + TT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                        *reinterpret_cast<uint4*>(chunk + r * TL::LROW + c * 16) = regs[q][p];
                    }
                }
                __syncthreads();
                if (tile_wave) mma_chunk32<T, false>(pacc, chunk, chunk, br, ch, lane);
            }
        }
    }
    STAMP(1);
    acc_t acc[2];
    acc[0] = acc_zero<T>();
    acc[1] = acc_zero<T>();
    if (tile_wave) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                T v = (row == col) ? (T)1 : (T)0;                   // identity padding; strict upper part is zero
                if (row < w && col <= row) v = dval[c][r] - pacc[c][r];
                acc[c][r] = v;
            }
    }
    diag_tail<T>(acc, reinterpret_cast<T*>(pcol_), reinterpret_cast<T*>(hs_), reinterpret_cast<T*>(cs_),
                 reinterpret_cast<T*>(rall_), D, ld, w, inv, info, col_base);
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
template <typename T> struct TrsmLds { static constexpr int BYTES = (2 * TR + SB) * Tile64<T>::LROW; };

template <typename T>
static __device__ __forceinline__ void trsm64_body(unsigned char* smem, T* __restrict__ Prow, int64_t ldp, int mrows, int kw, int kprev,
                                                    const T* __restrict__ Lrow, int64_t ldl,
                                                    const T* __restrict__ invL)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    // smem: 32 + 32 + 64 rows = TRSM_LDS bytes: 67.6 KB (f64) -- fits next to one resident trailing-update workgroup
    unsigned char* ps = smem;                          // P_s, then T          (32 rows)
    unsigned char* as = smem + TR * TL::LROW;          // chunk of Pprev       (32 rows)
    unsigned char* bs = smem + 2 * TR * TL::LROW;      // chunk of Lrow, finally invL (64 rows)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, ch = wave >> 1;
    CHAIN_SETPRIO();               // panel chain: ahead of co-resident update waves
    const T* Pprev = Prow - kprev;               // the panel's earlier columns of the same rows

    STAMP(8);
    v4u ra[TR * TL::CPR / 256], rb[SB * TL::CPR / 256];
    // the inverted diagonal block is needed last but depends on nothing: requested first, parked in
    // registers, so that its round trip hides behind the whole K loop instead of following it
    v4u rinv[SB * TL::CPR / 256];
#pragma unroll
    for (int p = 0; p < SB * TL::CPR / 256; ++p) {
        const int e = tid + 256 * p, r = e / TL::CPR, c = e - r * TL::CPR;
        rinv[p] = *reinterpret_cast<const v4u*>(invL + r * SB + c * X::EPC);
    }
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
    swrite_tile64<T, SB>(bs, rinv);
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

// The same solve for G consecutive 32-row tiles in ONE workgroup (batched launches, round 3).  In a batch of
// many blocks the panel solves are not latency- but L2-bandwidth-bound: every 32-row workgroup streams the
// block's 64 x kprev strip of L and its 64 x 64 inverse (80-130 KB) for 16-48 KB of its own rows -- 600 MB per
// launch for 128 blocks of 2048.  Here a chunk of L is staged once per G tiles and the inverse once per
// workgroup; the tiles' operand strips follow one another through the same LDS area.
constexpr int TRSM_GROUP = 4;
template <typename T, int G>
static __device__ __forceinline__ void trsm64_group(unsigned char* smem, T* __restrict__ Prow, int64_t ldp, int mrows, int kw, int kprev,
                                                     const T* __restrict__ Lrow, int64_t ldl, const T* __restrict__ invL)
{
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    unsigned char* ps = smem;                          // P_s of one tile, then T
    unsigned char* as = smem + TR * TL::LROW;          // chunk of one tile's earlier panel columns
    unsigned char* bs = smem + 2 * TR * TL::LROW;      // chunk of Lrow, finally invL
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, ch = wave >> 1;
    CHAIN_SETPRIO();
    const T* Pprev = Prow - kprev;
    v4u ra[TR * TL::CPR / 256], rb[SB * TL::CPR / 256], rinv[SB * TL::CPR / 256];
#pragma unroll
    for (int p = 0; p < SB * TL::CPR / 256; ++p) {
        const int e = tid + 256 * p, r = e / TL::CPR, c = e - r * TL::CPR;
        rinv[p] = *reinterpret_cast<const v4u*>(invL + r * SB + c * X::EPC);
    }
    acc_t acc[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc[g][0] = acc_zero<T>(); acc[g][1] = acc_zero<T>(); }
    if (kprev > 0) {
        gload_tile64<T, SB>(rb, Lrow, ldl, kw, SB);
        gload_tile64<T, TR>(ra, Pprev, ldp, mrows, SB);
    }
    for (int kc = 0; kc < kprev; kc += SB) {
        __syncthreads();                               // the previous chunk's fragments have been read
        swrite_tile64<T, SB>(bs, rb);
        if (kc + SB < kprev) gload_tile64<T, SB>(rb, Lrow + kc + SB, ldl, kw, SB);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g) __syncthreads();                    // the previous tile's fragments have been read
            swrite_tile64<T, TR>(as, ra);
            __syncthreads();
            // next operand strip in flight during the multiplies: the next tile's, or tile 0's of the next chunk
            if (g + 1 < G) gload_tile64<T, TR>(ra, Pprev + (int64_t)(g + 1) * TR * ldp + kc, ldp, mrows - (g + 1) * TR, SB);
            else if (kc + SB < kprev) gload_tile64<T, TR>(ra, Pprev + kc + SB, ldp, mrows, SB);
            mma_chunk32<T, false>(acc[g], as, bs, rt, ch, lane);
        }
    }
    __syncthreads();
    swrite_tile64<T, SB>(bs, rinv);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int rows_g = mrows - g * TR;             // workgroup-uniform
        if (rows_g > 0) {
            T* Pg = Prow + (int64_t)g * TR * ldp;
            __syncthreads();                           // the previous tile's T has been read; invL is in place
            load_tile64<T, TR>(ps, Pg, ldp, rows_g < TR ? rows_g : TR, kw);
            __syncthreads();
            if (kprev > 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + (lane & 15);
                        T* t = reinterpret_cast<T*>(ps + row * TL::LROW) + col;
                        *t -= acc[g][c][r];
                    }
            }
            __syncthreads();
            acc_t x[2];
            x[0] = acc_zero<T>(); x[1] = acc_zero<T>();
            mma_chunk32<T, true>(x, ps, bs, rt, ch, lane);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int gc = (2 * ch + c) * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = rt * 16 + X::crow(lane, r);
                    if (lr < rows_g && gc < kw) Pg[(int64_t)lr * ldp + gc] = x[c][r];
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2)
void k_trsm64(T* __restrict__ P1, int64_t ld1, int M1, int nb1,
              T* __restrict__ P2, int64_t ld2, int M2,
              int kw, int kprev, const T* __restrict__ Lrow, int64_t ldl, const T* __restrict__ invL,
              int64_t sk, int64_t sws, int64_t sb, int nchain, Riders<T> rd, int trg)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[ChainLds<T>::BYTES];
    if ((int)blockIdx.x >= nchain) {                 // riders
        run_rider<T>(smem, rd, (int)blockIdx.x - nchain, sk, sb);
        return;
    }
    P1 += (int64_t)blockIdx.y * sk;                  // batch: see k_diag64
    if (P2) P2 += (int64_t)blockIdx.y * sb;
    Lrow += (int64_t)blockIdx.y * sk;
    invL += (int64_t)blockIdx.y * sws;
    const bool second = (int)blockIdx.x >= nb1;
    T* P = second ? P2 : P1;
    const int64_t ldp = second ? ld2 : ld1;
    const int M = second ? M2 : M1;
    // trg = rows per workgroup: TR, or TR x TRSM_GROUP in batched launches (nb1 counts workgroups of that size)
    const int row0 = (second ? (int)blockIdx.x - nb1 : (int)blockIdx.x) * trg;
    if (trg == TR) trsm64_body<T>(smem, P + (int64_t)row0 * ldp, ldp, min(TR, M - row0), kw, kprev, Lrow, ldl, invL);
    else           trsm64_group<T, TRSM_GROUP>(smem, P + (int64_t)row0 * ldp, ldp, min(trg, M - row0), kw, kprev, Lrow, ldl, invL);
}

// ---------------------------------------------------------------------------
// One link of the panel chain in ONE launch (round 2).  Sub-block s of the panel has been factored
// (its inverse is in the workspace); this launch does
//   workgroups 1..   the panel solve of sub-block s for the rows BELOW the next diagonal block
//                    (and the carried rows): 32-row workgroups, exactly k_trsm64's body;
//   workgroup 0      the next diagonal block: its 64 rows' panel solve
//                    X = (P_s - Pprev Lrow^T) inv(L_ss)^T, the Schur complement
//                    S = A - [Pprev X][Pprev X]^T of the diagonal block -- every operand chunk serves
//                    both products from one LDS tile -- and the factorisation of S (diag_tail).
// The tall solve, which nothing on the chain waits for, runs in the shadow of workgroup 0's pivot
// loop, and the next diagonal block never waits for a launch of its own: per link
// max(solve, X + S + pivots) instead of diag + solve.  Operand tiles and the factorisation's LDS
// areas overlay each other (67.6 KB in all, as the panel solve alone).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(DG_NT)
void k_link(T* __restrict__ A, int64_t ld, int n, int c0, int k0, int wn,
            T* __restrict__ P2, int64_t ld2, int M2, T* __restrict__ ws, int32_t* info,
            int64_t sk, int64_t sws, int64_t sb, int nchain, Riders<T> rd)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[ChainLds<T>::BYTES];
    if ((int)blockIdx.x >= nchain) {                 // riders
        if (threadIdx.x >= 256) return;
        run_rider<T>(smem, rd, (int)blockIdx.x - nchain, sk, sb);
        return;
    }
    A += (int64_t)blockIdx.y * sk;                   // batch: see k_diag64
    ws += (int64_t)blockIdx.y * sws;
    if (P2) P2 += (int64_t)blockIdx.y * sb;
    info += blockIdx.y;
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    const int kprev = c0 - k0;
    const T* invL = ws + (int64_t)(c0 / SB) * (SB * SB);
    const T* Lrow = A + (int64_t)c0 * ld + k0;       // rows of the factored diagonal block, earlier panel columns
    const int r0 = c0 + SB;                          // first row (and column) of the next diagonal block
    if (blockIdx.x != 0) {
        // the solve uses four waves; the other five leave as whole waves (s_barrier counts the waves that
        // have not terminated, so the body's barriers are among the remaining four)
        if (threadIdx.x >= 256) return;
        const int pc = r0 + wn;
        const int M1 = n - pc, nb1 = (M1 + TR - 1) / TR;
        const int b = (int)blockIdx.x - 1;
        const bool second = b >= nb1;
        T* P = second ? P2 : A + (int64_t)pc * ld + c0;
        const int64_t ldp = second ? ld2 : ld;
        const int M = second ? M2 : M1;
        const int row0 = (second ? b - nb1 : b) * TR;
        trsm64_body<T>(smem, P + (int64_t)row0 * ldp, ldp, min(TR, M - row0), SB, kprev, Lrow, ld, invL);
        return;
    }
    constexpr int TT = 64 * DG_TW;                   // threads of the tile waves
    constexpr int NR = SB * TL::CPR / TT;            // 16-byte pieces per thread and 64 x 64 tile
    static_assert(SB * TL::LROW >= DiagLds<T>::CS, "pivot columns overlay the second operand tile");
    static_assert(SB * TL::LROW >= 2 * DiagLds<T>::PCOL + DiagLds<T>::RALL, "gather buffers overlay the first operand tile");
    unsigned char* bufA = smem;                      // own rows' chunk (both operands of S, left operand of T), then T, then X
    unsigned char* bufB = smem + SB * TL::LROW;      // Lrow chunk, then inv(L_ss)
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool tile_wave = g < DG_TW;
    const int br = (g >> 1) & 3, ch = g & 1;
    const int fcol = lane & 15;
    CHAIN_SETPRIO();
    T* Arow = A + (int64_t)r0 * ld;                  // the next diagonal block's rows
    T* D = Arow + r0;

    // everything that depends on nothing is requested first: the diagonal block and P_s in
    // accumulator layout, inv(L_ss) and the first operand chunks in staging registers
    T dval[2][4], pval[2][4];
    v4u rI[NR], rA[NR], rB[NR];
    if (tile_wave) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                dval[c][r] = (row < wn && col <= row) ? D[(int64_t)row * ld + col] : (T)0;
                pval[c][r] = (row < wn) ? Arow[(int64_t)row * ld + c0 + col] : (T)0;
            }
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + TT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            rI[p] = *reinterpret_cast<const v4u*>(invL + r * SB + c * X::EPC);
            if (kprev > 0) {
                rA[p] = (r < wn) ? *reinterpret_cast<const v4u*>(Arow + k0 + (int64_t)r * ld + c * X::EPC) : v4u_zero();
                rB[p] = *reinterpret_cast<const v4u*>(Lrow + (int64_t)r * ld + c * X::EPC);
            }
        }
    }
    acc_t accT[2], accS[2];
    accT[0] = acc_zero<T>(); accT[1] = acc_zero<T>();
    accS[0] = acc_zero<T>(); accS[1] = acc_zero<T>();
    for (int kc = 0; kc < kprev; kc += SB) {
        if (kc) __syncthreads();                     // the previous chunk has been consumed
        if (tile_wave) {
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int e = tid + TT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                *reinterpret_cast<v4u*>(bufA + r * TL::LROW + c * 16) = rA[p];
                *reinterpret_cast<v4u*>(bufB + r * TL::LROW + c * 16) = rB[p];
            }
        }
        __syncthreads();
        if (tile_wave) {
            if (kc + SB < kprev) {                   // next chunk in flight during the multiplies
#pragma unroll
                for (int p = 0; p < NR; ++p) {
                    const int e = tid + TT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                    rA[p] = (r < wn) ? *reinterpret_cast<const v4u*>(Arow + k0 + kc + SB + (int64_t)r * ld + c * X::EPC)
                                     : v4u_zero();
                    rB[p] = *reinterpret_cast<const v4u*>(Lrow + kc + SB + (int64_t)r * ld + c * X::EPC);
                }
            }
            mma_chunk32<T, false>(accT, bufA, bufB, br, ch, lane);
            mma_chunk32<T, false>(accS, bufA, bufA, br, ch, lane);
        }
    }
    if (kprev > 0) __syncthreads();
    // T = P_s - accT (each lane owns its accumulator elements) as the left operand, inv(L_ss) as the right one
    if (tile_wave) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                *(reinterpret_cast<T*>(bufA + row * TL::LROW) + col) = pval[c][r] - accT[c][r];
            }
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + TT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            *reinterpret_cast<v4u*>(bufB + r * TL::LROW + c * 16) = rI[p];
        }
    }
    lds_barrier();
    acc_t accX[2];
    accX[0] = acc_zero<T>(); accX[1] = acc_zero<T>();
    if (tile_wave) {
        mma_chunk32<T, true>(accX, bufA, bufB, br, ch, lane);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                if (row < wn) Arow[(int64_t)row * ld + c0 + col] = accX[c][r];
            }
    }
    lds_barrier();                                   // T and inv(L_ss) have been read
    if (tile_wave) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                *(reinterpret_cast<T*>(bufA + row * TL::LROW) + col) = accX[c][r];     // rows >= wn are zero
            }
    }
    lds_barrier();
    acc_t acc[2];
    acc[0] = acc_zero<T>();
    acc[1] = acc_zero<T>();
    if (tile_wave) {
        mma_chunk32<T, false>(accS, bufA, bufA, br, ch, lane);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = br * 16 + X::crow(lane, r), col = (2 * ch + c) * 16 + fcol;
                T v = (row == col) ? (T)1 : (T)0;                   // identity padding; strict upper part is zero
                if (row < wn && col <= row) v = dval[c][r] - accS[c][r];
                acc[c][r] = v;
            }
    }
    // The second tile is dead since the barrier above: S goes there (diag_tail) and becomes the pivot columns;
    // the first is read by the last multiply until diag_tail's barrier and then holds the gather buffers.
    T* pcol = reinterpret_cast<T*>(bufA);
    T* hs   = reinterpret_cast<T*>(bufA + DiagLds<T>::PCOL);
    T* rall = reinterpret_cast<T*>(bufA + 2 * DiagLds<T>::PCOL);
    diag_tail<T>(acc, pcol, hs, reinterpret_cast<T*>(bufB), rall, D, ld, wn,
                 ws + (int64_t)(r0 / SB) * (SB * SB), info, r0);
}

// ---------------------------------------------------------------------------
// Four-wave forms of the diagonal factorisation and of the link (round 2).  A nine-wave workgroup
// needs a compute unit that BOTH resident trailing-update workgroups have left (three of its waves
// share one SIMD's registers), so beside a running update it waits for the update's last generation;
// a four-wave workgroup (one wave per SIMD, <= 256 registers) fits beside ONE update workgroup and
// is dispatched, by queue priority, as soon as any update workgroup retires.
//   waves 0..2   tile waves: the sixteen 16x16 tiles of the combined array dealt round-robin
//                (tile t = 4 row-tile + column-tile belongs to wave t mod 3: 6 / 5 / 5 tiles);
//   wave 3       the pivot wave, alone on its SIMD.
// The left-looking products before the factorisation use all four waves in row-tile layout
// (wave w = rows 16 w ..: mma_chunk64) and hand the Schur complement over through LDS (`cs`, whose
// first use inside the loop is the pivot wave's write of block 0 after the first barrier).
// ---------------------------------------------------------------------------
constexpr int Q_TW = 3;               // tile waves
constexpr int Q_NT = 256;             // threads

// Schur complement from row-tile accumulators to `cs` (identity padding beyond w, zeros above the diagonal)
template <typename T>
static __device__ __forceinline__ void schur_to_lds(T* __restrict__ cs, const T (&dval)[4][4],
                                                     const typename Mx<T>::acc_t (&sub)[4], int wave, int lane, int w)
{
    using X = Mx<T>;
    constexpr int LS = SB + 2;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
            T v = (row == col) ? (T)1 : (T)0;
            if (row < w && col <= row) v = dval[c][r] - sub[c][r];
            cs[row * LS + col] = v;
        }
    // the other waves read these words right behind the caller's barrier: read the last one back first (lds_settle)
    lds_settle(&cs[(wave * 16 + X::crow(lane, 3)) * LS + 3 * 16 + (lane & 15)]);
}

template <typename T>
__global__ __launch_bounds__(Q_NT, 2)
void k_diag64q(T* __restrict__ D, int64_t ld, int w, const T* __restrict__ Lrow, int kprev,
               T* __restrict__ inv, int32_t* info, int col_base, int64_t sk, int64_t sws, int64_t sb, Riders<T> rd)
{
    __shared__ __attribute__((aligned(16))) unsigned char chunk[RIDER_LDS];           // Lrow chunk / a rider's tiles
    if (blockIdx.x != 0) {                           // riders
        run_rider<T>(chunk, rd, (int)blockIdx.x - 1, sk, sb);
        return;
    }
    D += (int64_t)blockIdx.y * sk;
    Lrow += (int64_t)blockIdx.y * sk;
    inv += (int64_t)blockIdx.y * sws;
    info += blockIdx.y;
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    __shared__ __attribute__((aligned(16))) unsigned char pcol_[DiagLds<T>::PCOL];
    __shared__ __attribute__((aligned(16))) unsigned char hs_[DiagLds<T>::PCOL];
    __shared__ __attribute__((aligned(16))) unsigned char cs_[DiagLds<T>::CS];
    __shared__ __attribute__((aligned(16))) unsigned char rall_[DiagLds<T>::RALL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    CHAIN_SETPRIO();
    T dval[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
            dval[c][r] = (row < w && col <= row) ? D[(int64_t)row * ld + col] : (T)0;
        }
    acc_t pacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pacc[c] = acc_zero<T>();
    if (kprev > 0) {
        // all (up to four: kprev = 256 when the block takes its head update here) chunks of Lrow are
        // requested at once: ONE exposed round trip -- beside a running trailing update a dependent global
        // round trip costs several microseconds, not one
        constexpr int NR = SB * TL::CPR / Q_NT;
        constexpr int MAXC = CIMRGP_NB / SB;
        v4u regs[MAXC][NR];
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            if (q * SB < kprev) {
#pragma unroll
                for (int p = 0; p < NR; ++p) {
                    const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                    regs[q][p] = (r < w) ? *reinterpret_cast<const v4u*>(Lrow + q * SB + (int64_t)r * ld + c * X::EPC) : v4u_zero();
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            if (q * SB < kprev) {                      // uniform
                if (q) __syncthreads();                // the previous chunk has been consumed
#pragma unroll
                for (int p = 0; p < NR; ++p) {
                    const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
                    *reinterpret_cast<v4u*>(chunk + r * TL::LROW + c * 16) = regs[q][p];
                }
                __syncthreads();
                mma_chunk64<T, false>(pacc, chunk, chunk, wave, lane);
            }
        }
    }
    schur_to_lds<T>(reinterpret_cast<T*>(cs_), dval, pacc, wave, lane, w);
    __syncthreads();
    diag_tail_lds<T, Q_TW, Q_NT>(wave < Q_TW ? wave : -1, wave == Q_TW, reinterpret_cast<T*>(pcol_), reinterpret_cast<T*>(hs_),
                                 reinterpret_cast<T*>(cs_), reinterpret_cast<T*>(rall_), D, ld, w, inv, info, col_base);
}

// k_link with four-wave workgroups throughout (see k_link for what a link does).
template <typename T>
__global__ __launch_bounds__(Q_NT, 2)
void k_linkq(T* __restrict__ A, int64_t ld, int n, int c0, int k0, int wn,
             T* __restrict__ P2, int64_t ld2, int M2, T* __restrict__ ws, int32_t* info,
             int64_t sk, int64_t sws, int64_t sb, int nchain, Riders<T> rd, int trg)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[ChainLds<T>::BYTES];
    if ((int)blockIdx.x >= nchain) {                 // riders
        run_rider<T>(smem, rd, (int)blockIdx.x - nchain, sk, sb);
        return;
    }
    A += (int64_t)blockIdx.y * sk;
    ws += (int64_t)blockIdx.y * sws;
    if (P2) P2 += (int64_t)blockIdx.y * sb;
    info += blockIdx.y;
    using X = Mx<T>;
    using TL = Tile64<T>;
    using acc_t = typename X::acc_t;
    const int kprev = c0 - k0;
    const T* invL = ws + (int64_t)(c0 / SB) * (SB * SB);
    const T* Lrow = A + (int64_t)c0 * ld + k0;
    const int r0 = c0 + SB;
    if (blockIdx.x != 0) {
        // trg = rows per solve workgroup: TR, or TR x TRSM_GROUP in batched launches (k_trsm64)
        const int pc = r0 + wn;
        const int M1 = n - pc, nb1 = (M1 + trg - 1) / trg;
        const int b = (int)blockIdx.x - 1;
        const bool second = b >= nb1;
        T* P = second ? P2 : A + (int64_t)pc * ld + c0;
        const int64_t ldp = second ? ld2 : ld;
        const int M = second ? M2 : M1;
        const int row0 = (second ? b - nb1 : b) * trg;
        if (trg == TR) trsm64_body<T>(smem, P + (int64_t)row0 * ldp, ldp, min(TR, M - row0), SB, kprev, Lrow, ld, invL);
        else           trsm64_group<T, TRSM_GROUP>(smem, P + (int64_t)row0 * ldp, ldp, min(trg, M - row0), SB, kprev, Lrow, ld, invL);
        return;
    }
    constexpr int NR = SB * TL::CPR / Q_NT;
    static_assert(SB * TL::LROW >= DiagLds<T>::CS, "pivot columns overlay the second operand tile");
    static_assert(SB * TL::LROW >= 2 * DiagLds<T>::PCOL + DiagLds<T>::RALL, "gather buffers overlay the first operand tile");
    unsigned char* bufA = smem;
    unsigned char* bufB = smem + SB * TL::LROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    CHAIN_SETPRIO();
    T* Arow = A + (int64_t)r0 * ld;
    T* D = Arow + r0;

    T pval[4][4];
    v4u rI[NR], rA[NR], rB[NR];
    // inv(L_ss) is requested when the staging registers of the last operand chunk fall free (the
    // kernel is held to 256 registers so that a workgroup fits beside a trailing-update workgroup)
    auto fetch_inv = [&]() {
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            rI[p] = *reinterpret_cast<const v4u*>(invL + r * SB + c * X::EPC);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
                pval[c][r] = (row < wn) ? Arow[(int64_t)row * ld + c0 + col] : (T)0;
            }
    };
    if (kprev > 0) {
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            rA[p] = (r < wn) ? *reinterpret_cast<const v4u*>(Arow + k0 + (int64_t)r * ld + c * X::EPC) : v4u_zero();
            rB[p] = *reinterpret_cast<const v4u*>(Lrow + (int64_t)r * ld + c * X::EPC);
        }
    } else {
        fetch_inv();
    }
    acc_t accT[4], accS[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { accT[c] = acc_zero<T>(); accS[c] = acc_zero<T>(); }
    auto stage = [&]() {                             // staging registers -> operand tiles
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            *reinterpret_cast<v4u*>(bufA + r * TL::LROW + c * 16) = rA[p];
            *reinterpret_cast<v4u*>(bufB + r * TL::LROW + c * 16) = rB[p];
        }
    };
    // all chunks but the last: the next chunk in flight during the multiplies
    for (int kc = 0; kc + SB < kprev; kc += SB) {
        if (kc) __syncthreads();
        stage();
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
            rA[p] = (r < wn) ? *reinterpret_cast<const v4u*>(Arow + k0 + kc + SB + (int64_t)r * ld + c * X::EPC) : v4u_zero();
            rB[p] = *reinterpret_cast<const v4u*>(Lrow + kc + SB + (int64_t)r * ld + c * X::EPC);
        }
        mma_chunk64<T, false>(accT, bufA, bufB, wave, lane);
        mma_chunk64<T, false>(accS, bufA, bufA, wave, lane);
    }
    // the last chunk: inv(L_ss) and P_s in flight instead (the staging registers are free by then)
    if (kprev > 0) {
        if (kprev > SB) __syncthreads();
        stage();
        __syncthreads();
        fetch_inv();
        mma_chunk64<T, false>(accT, bufA, bufB, wave, lane);
        mma_chunk64<T, false>(accS, bufA, bufA, wave, lane);
    }
    if (kprev > 0) __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
            *(reinterpret_cast<T*>(bufA + row * TL::LROW) + col) = pval[c][r] - accT[c][r];
        }
#pragma unroll
    for (int p = 0; p < NR; ++p) {
        const int e = tid + Q_NT * p, r = e / TL::CPR, c = e - r * TL::CPR;
        *reinterpret_cast<v4u*>(bufB + r * TL::LROW + c * 16) = rI[p];
    }
    // the diagonal block itself: needed last, requested now that the staging registers are free
    T dval[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
            dval[c][r] = (row < wn && col <= row) ? D[(int64_t)row * ld + col] : (T)0;
        }
    lds_barrier();
    acc_t accX[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) accX[c] = acc_zero<T>();
    mma_chunk64<T, true>(accX, bufA, bufB, wave, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
            if (row < wn) Arow[(int64_t)row * ld + c0 + col] = accX[c][r];
        }
    lds_barrier();                                   // T and inv(L_ss) have been read
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + X::crow(lane, r), col = c * 16 + (lane & 15);
            *(reinterpret_cast<T*>(bufA + row * TL::LROW) + col) = accX[c][r];
        }
    lds_barrier();
    mma_chunk64<T, false>(accS, bufA, bufA, wave, lane);
    // S into the second tile (dead since the barrier above), which becomes `cs`; the first tile is
    // read by the multiply above until the barrier below and then holds the gather buffers
    schur_to_lds<T>(reinterpret_cast<T*>(bufB), dval, accS, wave, lane, wn);
    lds_barrier();
    T* pcol = reinterpret_cast<T*>(bufA);
    T* hs   = reinterpret_cast<T*>(bufA + DiagLds<T>::PCOL);
    T* rall = reinterpret_cast<T*>(bufA + 2 * DiagLds<T>::PCOL);
    diag_tail_lds<T, Q_TW, Q_NT>(wave < Q_TW ? wave : -1, wave == Q_TW, pcol, hs, reinterpret_cast<T*>(bufB), rall, D, ld, wn,
                                 ws + (int64_t)(r0 / SB) * (SB * SB), info, r0);
}

// All four sub-steps of a panel for rows that take no part in the factorisation itself (the
// right-hand-side rows of a row-wise solve): one launch per panel instead of four.  A row
// block only ever reads its own earlier results, written by this same workgroup.
template <typename T>
__global__ __launch_bounds__(256)
void k_trsm256(T* __restrict__ P, int64_t ldp, int M, int w, const T* __restrict__ Lpanel, int64_t ldl,
               const T* __restrict__ inv64, int64_t sp = 0, int64_t sk = 0, int64_t sws = 0)
{
    P += (int64_t)blockIdx.y * sp;                   // batch: see k_diag64
    Lpanel += (int64_t)blockIdx.y * sk;
    inv64 += (int64_t)blockIdx.y * sws;
    const int row0 = (int)blockIdx.x * TR;
    const int mrows = min(TR, M - row0);
    __shared__ __attribute__((aligned(16))) unsigned char smem[TrsmLds<T>::BYTES];
    for (int c0 = 0; c0 < w; c0 += SB) {
        if (c0) __syncthreads();                 // this workgroup's stores of the previous sub-step are visible
        trsm64_body<T>(smem, P + (int64_t)row0 * ldp + c0, ldp, mrows, min(SB, w - c0), c0,
                       Lpanel + (int64_t)c0 * ldl, ldl, inv64 + (int64_t)(c0 / SB) * (SB * SB));
    }
}

// ---------------------------------------------------------------------------
// Round 4: ONE launch per panel for the carried rows' own chain.  Panel p of the rows needs
//     W_p = (B_p - W_{p-1} L[p, p-1]^T) L_pp^-T
// -- the update by the previous panel (everything older has been applied by the bulk "far" updates, which are
// off this chain) and the 256-wide solve.  Until round 4 these were two launches of general kernels (a 64-tile
// update of 132 workgroups, 38-65 us, then k_trsm256, 34-55 us: both latency-bound) and the rows fell eleven
// panels behind a factorisation whose own chain takes 93-125 us per panel.  Here a workgroup owns 16 rows (one
// MFMA row tile) for both parts:
//   * wave w owns the four 16-column tiles  64 j + 16 w  (j = 0..3: one of every 64-column sub-block), so that in
//     sub-step j all four waves work on sub-block j and each already holds its share of it;
//   * the right operand (rows of L / of the 64 x 64 inverses) goes from global memory STRAIGHT into the
//     matrix-core operand registers -- lane (n, q) takes 32 contiguous bytes of row n per 128-byte chunk of K
//     (the k order inside a chunk is a permutation, the same one on both operands), so a wave's request is 16
//     rows x 128 bytes, whole cache lines, and nobody waits at a barrier for a staging buffer; the loads run a
//     ring of chunks ahead of the multiplies, through the barriers (LDS-scoped fences: a __syncthreads() would
//     drain them);
//   * the left operands (W_{p-1}'s rows, the solved sub-blocks, the 16 x 64 sub-block being solved) live in LDS.
// Sub-step j: T_j = B_j - W_{p-1} L[j, p-1]^T (phase 1, all j at once) - sum_{i<j} W_i L[j, i]^T, then
// W_j = T_j inv_j^T through LDS (two barriers per sub-step).  Full panels only (w = 256, previous panel 256 or
// none); ragged last panels keep the two-launch form.
// ---------------------------------------------------------------------------
template <typename T> struct RowsStep {
    static constexpr int R = 16;                                   // rows per workgroup
    static constexpr int CHE = 128 / (int)sizeof(T);               // elements per 128-byte chunk of K
    static constexpr int NCP = CIMRGP_NB / CHE;                    // chunks of the previous panel: 16 (f64) / 8 (f32)
    static constexpr int NCS = SB / CHE;                           // chunks of a 64-column sub-block: 4 / 2
    static constexpr int LPE = 32 / (int)sizeof(T);                // elements a lane takes per chunk (32 bytes)
    static constexpr int ASTR = CIMRGP_NB * (int)sizeof(T) + 16;   // LDS row strides: 16 bytes of padding
    static constexpr int TSTR = SB * (int)sizeof(T) + 16;
    static constexpr int BYTES = 2 * R * ASTR + R * TSTR;
    static constexpr int RING1 = 3;                                // phase 1: chunks in flight (4 tiles each)
    static constexpr int RING2 = 8;                                // phase 2: chunks in flight (1 tile each)
    static constexpr int P2_TOTAL = NCS * (1 + 2 + 3 + 4);
};

template <typename T, bool HAS_PREV>
__global__ __launch_bounds__(256)
void k_rows_step(T* __restrict__ P, int64_t ldp, int M, const T* __restrict__ Lrow, int64_t ldl, const T* __restrict__ inv64,
                 const T* __restrict__ Pprev = nullptr, int64_t ldprev = 0, int b_zero = 0, int ny = 1,
                 int64_t sp = 0, int64_t sl = 0, int64_t sws = 0, int64_t sprev = 0,
                 int64_t sp2 = 0, int64_t sl2 = 0, int64_t sws2 = 0)
{
    // Pprev: the previous panel's solved rows live elsewhere (row r of this launch at Pprev + r * ldprev) instead of in
    // the 256 columns left of P; b_zero: B_p = 0 (1) or the identity (2), not read.  blockIdx.y = i + ny * j: problem i of ny with strides
    // (sp, sl, sws, sprev), inside matrix j of a batch with strides (sp2, sl2, sws2; Pprev moves with sp2).  These
    // serve the 512-wide inverses of the skinny backward solve (build_invT).
    {
        const int yi = (int)blockIdx.y % ny, yj = (int)blockIdx.y / ny;
        P += (int64_t)yi * sp + (int64_t)yj * sp2;
        Lrow += (int64_t)yi * sl + (int64_t)yj * sl2;
        inv64 += (int64_t)yi * sws + (int64_t)yj * sws2;
        if (Pprev) Pprev += (int64_t)yi * sprev + (int64_t)yj * sp2;
    }
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    using RS = RowsStep<T>;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    constexpr int KPREV = HAS_PREV ? CIMRGP_NB : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RS::BYTES];
    unsigned char* aprev = smem;                       // W_{p-1}: R rows x 256
    unsigned char* wcur  = smem + RS::R * RS::ASTR;    // W_p as it is solved
    unsigned char* tbuf  = smem + 2 * RS::R * RS::ASTR;   // T_j: R rows x 64
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = (int)blockIdx.x * RS::R;
    const int mrows = min(RS::R, M - row0);
    T* Prow = P + (int64_t)row0 * ldp;                 // this workgroup's rows, first column of panel p
    const int fn = lane & 15, fq = lane >> 4;
    const int ctile = 16 * wave + fn;                  // this lane's column inside a 64-column sub-block

    // W_{p-1}'s rows: requested first, written to LDS after everything else has been requested
    v4u stg[HAS_PREV ? RS::NCP / 2 : 1];
    const int sr = tid >> 4, st16 = tid & 15;
    if (HAS_PREV) {
        // rows past the end: a valid row's values, never stored
        const T* src = Pprev ? Pprev + (int64_t)(row0 + min(sr, mrows - 1)) * ldprev : Prow - KPREV + (int64_t)min(sr, mrows - 1) * ldp;
#pragma unroll
        for (int i = 0; i < RS::NCP / 2; ++i) stg[i] = *reinterpret_cast<const v4u*>(src + (i * 16 + st16) * X::EPC);
    }
    acc_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[j][r] = (b_zero == 0) ? Prow[(int64_t)min(X::crow(lane, r), mrows - 1) * ldp + SB * j + ctile]
                      : (b_zero == 2 && row0 + X::crow(lane, r) == SB * j + ctile) ? (T)1 : (T)0;

    // right-operand rows of this lane: rows SB j + ctile of the panel's row block of L, and of the inverses
    const T* lp[4];
    const T* ip[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lp[j] = Lrow + (int64_t)(SB * j + ctile) * ldl + fq * RS::LPE;
        ip[j] = inv64 + (int64_t)j * (SB * SB) + ctile * SB + fq * RS::LPE;
    }
    // phase 2's operand stream in the order it is consumed: sub-step j = j NCS chunks of L[j, 0 .. 64 j) (behind
    // the previous panel's 256 columns), then NCS chunks of inv_j
    auto p2_addr = [&](int p) -> const T* {
        int j = 0, base = 0;
#pragma unroll
        for (j = 0; j < 4; ++j) {
            const int len = (j + 1) * RS::NCS;
            if (p < base + len) break;
            base += len;
        }
        const int c = p - base;
        return (c < j * RS::NCS) ? lp[j] + KPREV + c * RS::CHE : ip[j] + (c - j * RS::NCS) * RS::CHE;
    };
    v4u ring2[RS::RING2][2];
#define ROWS_P2_LOAD(p_)                                                                   \
    {                                                                                      \
        const T* q_ = p2_addr(p_);                                                         \
        ring2[(p_) % RS::RING2][0] = *reinterpret_cast<const v4u*>(q_);                    \
        ring2[(p_) % RS::RING2][1] = *reinterpret_cast<const v4u*>(q_ + X::EPC);           \
    }
    // the four 8-byte k-slots of a lane's 32 bytes
#define ROWS_SLOT(v_, s_) ((s_) == 0 ? make_uint2((v_)[0].x, (v_)[0].y) : (s_) == 1 ? make_uint2((v_)[0].z, (v_)[0].w) \
                           : (s_) == 2 ? make_uint2((v_)[1].x, (v_)[1].y) : make_uint2((v_)[1].z, (v_)[1].w))

    if (HAS_PREV) {
        v4u ring1[RS::RING1][4][2];
#define ROWS_P1_LOAD(c_)                                                                   \
    {                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
            const T* q_ = lp[j] + (c_) * RS::CHE;                                          \
            ring1[(c_) % RS::RING1][j][0] = *reinterpret_cast<const v4u*>(q_);             \
            ring1[(c_) % RS::RING1][j][1] = *reinterpret_cast<const v4u*>(q_ + X::EPC);    \
        }                                                                                  \
    }
#pragma unroll
        for (int c = 0; c < RS::RING1; ++c) ROWS_P1_LOAD(c)
#pragma unroll
        for (int i = 0; i < RS::NCP / 2; ++i)
            *reinterpret_cast<v4u*>(aprev + sr * RS::ASTR + (i * 16 + st16) * 16) = stg[i];
        lds_barrier();                                                   // W_{p-1}'s rows are in LDS
        const unsigned char* abase = aprev + fn * RS::ASTR + fq * 32;
#pragma unroll
        for (int c = 0; c < RS::NCP; ++c) {
            v4u a[2];
            a[0] = *reinterpret_cast<const v4u*>(abase + c * 128);
            a[1] = *reinterpret_cast<const v4u*>(abase + c * 128 + 16);
            if (c == RS::NCP - 1) {
                // the ring drains: phase 2's first chunks take its place
#pragma unroll
                for (int p = 0; p < RS::RING2; ++p) ROWS_P2_LOAD(p)
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint2 an = ROWS_SLOT(a, s);          // negated by the multiply itself (Mx<T>::mma_neg)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = X::mma_neg(an, ROWS_SLOT(ring1[c % RS::RING1][j], s), acc[j]);
            }
            if (c + RS::RING1 < RS::NCP) ROWS_P1_LOAD(c + RS::RING1)
        }
#undef ROWS_P1_LOAD
    } else {
#pragma unroll
        for (int p = 0; p < RS::RING2; ++p) ROWS_P2_LOAD(p)
    }

    int pos = 0;                                       // position in phase 2's stream (a constant once unrolled)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // T_j's share of this wave: the solved sub-blocks 0 .. j-1 of this panel against L[j, 0 .. 64 j)
        const unsigned char* wbase = wcur + fn * RS::ASTR + fq * 32;
#pragma unroll
        for (int c = 0; c < j * RS::NCS; ++c, ++pos) {
            v4u a[2];
            a[0] = *reinterpret_cast<const v4u*>(wbase + c * 128);
            a[1] = *reinterpret_cast<const v4u*>(wbase + c * 128 + 16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[j] = X::mma_neg(ROWS_SLOT(a, s), ROWS_SLOT(ring2[pos % RS::RING2], s), acc[j]);
            if (pos + RS::RING2 < RS::P2_TOTAL) ROWS_P2_LOAD(pos + RS::RING2)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<T*>(tbuf + X::crow(lane, r) * RS::TSTR + ctile * (int)sizeof(T)) = acc[j][r];
        lds_barrier();                                                   // T_j complete
        acc_t x = acc_zero<T>();
        const unsigned char* tbase = tbuf + fn * RS::TSTR + fq * 32;
#pragma unroll
        for (int c = 0; c < RS::NCS; ++c, ++pos) {
            v4u a[2];
            a[0] = *reinterpret_cast<const v4u*>(tbase + c * 128);
            a[1] = *reinterpret_cast<const v4u*>(tbase + c * 128 + 16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                x = X::mma(ROWS_SLOT(a, s), ROWS_SLOT(ring2[pos % RS::RING2], s), x);
            if (pos + RS::RING2 < RS::P2_TOTAL) ROWS_P2_LOAD(pos + RS::RING2)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = X::crow(lane, r);
            if (j < 3) *reinterpret_cast<T*>(wcur + lr * RS::ASTR + (SB * j + ctile) * (int)sizeof(T)) = x[r];
            if (lr < mrows) Prow[(int64_t)lr * ldp + SB * j + ctile] = x[r];
        }
        if (j < 3) lds_barrier();                                        // W_j is in LDS; T's buffer is free
    }
#undef ROWS_P2_LOAD
#undef ROWS_SLOT
}

// ---------------------------------------------------------------------------
// 256x256 inverses of the diagonal blocks, for the skinny solves: the identity is
// carried through the panel solve, batched over ALL panels (blockIdx.y):
// invT_p = I L_pp^-T = (L_pp^-1)^T  (upper triangular, row r = column r of L_pp^-1),
// stored 256 x 256 row-major per panel.  (Rounds 1-2: an init pass and one launch per
// 64-column sub-step.)
// ---------------------------------------------------------------------------
// ONE launch (round 3): a 32-row strip of the 256 x 256 block depends on no other strip, so its
// workgroup runs the four 64-column sub-steps itself (its own earlier columns are read back from global memory
// behind a workgroup barrier), writes the identity it starts from instead of a separate init pass, and skips
// what is known to be zero: strip i has nothing left of column 32 i.  (Five dependent launches at the end of
// every factorisation were 53 us of a N = 8192 step; 0.94 ms of a 128 x 2048 layer.)
template <typename T>
__global__ __launch_bounds__(256)
void k_invT_panel(T* __restrict__ invT, const T* __restrict__ L, int64_t ld, int n, const T* __restrict__ inv64,
                  int64_t sk = 0, int64_t sws = 0, int p0 = 0)
{
    invT += (int64_t)blockIdx.z * sws;
    L += (int64_t)blockIdx.z * sk;
    inv64 += (int64_t)blockIdx.z * sws;
    const int p = p0 + (int)blockIdx.y, strip = blockIdx.x;
    const int k0 = p * CIMRGP_NB;
    const int w = min(CIMRGP_NB, n - k0);
    T* blk = invT + (int64_t)p * (CIMRGP_NB * CIMRGP_NB) + (int64_t)strip * TR * CIMRGP_NB;
    for (int e = threadIdx.x; e < TR * CIMRGP_NB; e += 256) {
        const int r = strip * TR + (e >> 8), c = e & 255;
        blk[e] = (r == c && r < w) ? (T)1 : (T)0;
    }
    __shared__ __attribute__((aligned(16))) unsigned char smem[TrsmLds<T>::BYTES];
    const int first = (strip * TR) / SB;                 // first 64-column block with a non-zero in this strip
    for (int s = first; s < CIMRGP_NB / SB; ++s) {
        const int c0 = k0 + SB * s;
        const int kw = min(SB, n - c0);
        if (kw <= 0) break;
        __syncthreads();                                  // the strip's earlier columns (and the identity) are stored; LDS is free
        trsm64_body<T>(smem, blk + SB * s, CIMRGP_NB, TR, kw, SB * (s - first), L + (int64_t)c0 * ld + k0 + SB * first, ld,
                       inv64 + (int64_t)(c0 / SB) * (SB * SB));
    }
}

// The panel chain's wait for the head tiles of a combined (head-first) persistent update: ONE workgroup polls the
// count of stored head tiles (k_gemm_nt_pers adds 1 per tile behind an agent-scope release) and ends; the
// chain's next launch follows it in queue order.  One resident wave cannot starve the update of compute
// units (a poll inside the wide panel-solve launch could: its hundreds of workgroups would hold the units the
// persistent workgroups are waiting for).  The poll is bounded by the constant 100 MHz clock (s_memrealtime):
// after GATE_TIMEOUT_TICKS = 2 s without the count the factorisation is flagged with CIMRGP_INFO_WATCHDOG
// (include/cimrgp.h: a SCHEDULE failure, not a numerical one) instead of hanging the device.
constexpr long long GATE_TIMEOUT_TICKS = 200000000ll;
__global__ void k_gate(const int* __restrict__ flag, int expected, int32_t* info)
{
    if (threadIdx.x == 0) {
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
            __builtin_amdgcn_s_sleep(64);
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > GATE_TIMEOUT_TICKS) {
                atomicCAS(info, 0, CIMRGP_INFO_WATCHDOG);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// The other direction (round 5): "panel final" from the chain's queue to the update's queue as a device word instead of
// an event.  k_post follows the panel's last kernel in queue order and stores the panel's ordinal; on the update's queue
// a k_gate in front of the next persistent update polls it.  Two kernels on ONE queue follow each other within a
// microsecond or two; an event recorded on one queue and waited for on another cost 10-20 us between the end of one
// update and the start of the next (profiles/r05_timeline_n8192.txt), twelve times per factorisation at N = 8192.
__global__ void k_post(int* __restrict__ flag, int value)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// One pass over the panels.  With FACTOR the matrix itself is factored; with
// rows (b != nullptr) the extra rows are carried through the same panel
// operations, which turns them into  B L^-T.
// How many panels share one pass over the far part of the matrix being updated (K = 256 x that
// many).  Stand-alone rate of the update at K = 256 / 512 / 768 / 1024: 51.5 / 59.0 / 62.1 / 63.5
// TF/s at M = 15360, 55.5 / 62.1 / 64.3 / 65.3 TF/s at M = 32256.  Inside the factorisation
// (whole potrf, groups capped at 1 / 2 / 3 / 4): N = 32768: 217.8 / 203.9 / 201.3 / 200.9 ms,
// N = 65536: 1650 / 1512 / 1505 / 1505 ms -- pairs bring most of it.  The group is also kept
// small enough for the operand panel (far x K x 8 bytes) to fit the 256 MB Infinity Cache, which
// caps it at 3 (a group of 4 would need far > 32768 and far <= 32768 at once).
static inline int group_size(int64_t far, int64_t pair_above)
{
    if (far <= pair_above) return 1;
    const int by_benefit = (far > 16384) ? 3 : 2;
    const int64_t by_cache = ((int64_t)1 << 17) / far;       // 2^28 bytes / (8 bytes x far rows x 256 columns)
    const int g = (by_cache < by_benefit) ? (int)by_cache : by_benefit;
    return g < 1 ? 1 : g;
}

struct PanelGroup {
    int64_t g0 = -1;      // first column of the group's first panel (-1: no group open)
    int left = 0;         // panels of the group still to come, this one included
    bool near_pending = false;   // carried rows: the previous panel's update of THIS panel's columns is owed (k_rows_step applies it)
};

// The latency-bound chain of one panel [k0, k0 + w) on stream st: the first 64-column diagonal
// block in a launch of its own, then one k_link per further sub-block (panel solve of sub-block s
// beside the factorisation of diagonal block s + 1), and the panel solve of the last sub-block.
// b (m x .., leading dimension ldb): carried rows solved along (second row set), or nullptr.
// `alone`: nothing heavy runs beside the chain (one-queue sweeps, the tail).  Then the nine-wave
// kernels are used (a few per cent faster by themselves: N = 2048 1.05 against 1.09 ms).  Beside a
// running trailing update a nine-wave workgroup is dispatched only to a compute unit BOTH of whose
// update workgroups have retired -- in effect after the update's last generation (first link of a
// panel 230-300 us at N = 8192) -- while a four-wave workgroup fits beside one update workgroup and
// gets, by queue priority, the first slot that falls free: there the four-wave forms run
// (N = 8192: period of the update-bound panels 437 / 391 / 371 -> 405 / 363 / 355 us).
// (CIMRGP_CHAIN overrides the choice for measurements: see Tuning.)
// `riders`: nullptr, or the update tiles riding in the chain's launches -- riders[0] in the first diagonal
// block's launch, riders[1..3] in the links, riders[4] in the last sub-block's panel solve (fused_sweep).
// `left64`: the first diagonal block takes, as its left-looking prologue, the 64 columns just left of the
// panel (the previous panel's last sub-block, whose contribution the riders could not apply before it was final).
template <typename T>
static int panel_chain(T* kmat, int64_t n, int64_t ld, T* ws, int32_t* info, int64_t k0, int64_t w,
                       T* b, int64_t m, int64_t ldb, PotrfBatch bt, hipStream_t st, const char* fn, bool alone,
                       bool first_done = false, const Riders<T>* riders = nullptr, bool left64 = false)
{
    const int chain_mode = knobs().chain_mode;
    const bool split_links = (chain_mode == 1);
    // (a batch of factorisations in one launch is its own crowd: many link workgroups compete for the
    // compute units, and the four-wave form packs twice as many of them)
    const bool waves4 = (chain_mode == 3) || (chain_mode == 0 && (!alone || bt.count > 1));
    const bool rows = (b != nullptr && m > 0);
    const unsigned nbatch = (unsigned)bt.count;
    const int64_t k1 = k0 + w;
    // rows per panel-solve workgroup: groups of TRSM_GROUP tiles in batched launches (k_linkq / k_trsm64 take it as an argument)
    const int trg = (bt.count > 1 && waves4 && knobs().trsm_group) ? TR * TRSM_GROUP : TR;
    const int nb2 = rows ? (int)((m + trg - 1) / trg) : 0;
    const Riders<T> none = no_riders<T>();
    int launch = 0;                                   // 0: first diagonal block, 1..3: links, 4: last panel solve
    for (int64_t c0 = k0; c0 < k1; c0 += SB) {
        const int sw = (int)((k1 - c0 < SB) ? (k1 - c0) : SB);
        const int kprev = (int)(c0 - k0);
        const int64_t pc = c0 + sw;            // first row after this sub-block
        T* inv = ws + (c0 / SB) * (SB * SB);
        const T* lrow = kmat + c0 * ld + k0;   // rows of the diagonal block, earlier panel columns
        if ((split_links || c0 == k0) && !(first_done && c0 == k0)) {
            const Riders<T>& rd = (riders && c0 == k0) ? riders[0] : none;
            const bool l64 = left64 && c0 == k0;
            const T* lr = l64 ? lrow - SB : lrow;
            const int kp = l64 ? SB : kprev;
            if (waves4)
                hipLaunchKernelGGL((k_diag64q<T>), dim3((unsigned)(1 + rd.total), nbatch), dim3(Q_NT), 0, st,
                                   kmat + c0 * ld + c0, ld, sw, lr, kp, inv, info, (int)c0, bt.sk, bt.sws, bt.sb, rd);
            else
                hipLaunchKernelGGL((k_diag64<T>), dim3((unsigned)(1 + rd.total), nbatch), dim3(DG_NT), 0, st,
                                   kmat + c0 * ld + c0, ld, sw, lr, kp, inv, info, (int)c0, bt.sk, bt.sws, bt.sb, rd);
            CIMRGP_LAUNCH_CHECK(fn);
        }
        if (!split_links && pc < k1) {
            const int wn = (int)((k1 - pc < SB) ? (k1 - pc) : SB);         // next diagonal block of this panel
            const int64_t m1 = n - (pc + wn);
            const int nb1 = (int)((m1 + (waves4 ? trg : TR) - 1) / (waves4 ? trg : TR));
            ++launch;
            const Riders<T>& rd = (riders && launch <= 3) ? riders[launch] : none;
            const int nchain = 1 + nb1 + nb2;
            if (waves4)
                hipLaunchKernelGGL((k_linkq<T>), dim3((unsigned)(nchain + rd.total), nbatch), dim3(Q_NT), 0, st,
                                   kmat, ld, (int)n, (int)c0, (int)k0, wn, rows ? b + c0 : (T*)nullptr, ldb, rows ? (int)m : 0,
                                   ws, info, bt.sk, bt.sws, bt.sb, nchain, rd, trg);
            else
                hipLaunchKernelGGL((k_link<T>), dim3((unsigned)(nchain + rd.total), nbatch), dim3(DG_NT), 0, st,
                                   kmat, ld, (int)n, (int)c0, (int)k0, wn, rows ? b + c0 : (T*)nullptr, ldb, rows ? (int)m : 0,
                                   ws, info, bt.sk, bt.sws, bt.sb, nchain, rd);
            CIMRGP_LAUNCH_CHECK(fn);
            continue;
        }
        const int64_t m1 = n - pc;
        const int nb1 = (int)((m1 + trg - 1) / trg);
        const Riders<T>& rd = (riders && !split_links && pc == k1) ? riders[4] : none;
        if (nb1 + nb2 + rd.total > 0) {
            hipLaunchKernelGGL((k_trsm64<T>), dim3((unsigned)(nb1 + nb2 + rd.total), nbatch), dim3(256), 0, st,
                               kmat + pc * ld + c0, ld, (int)m1, nb1,
                               rows ? b + c0 : (T*)nullptr, ldb, rows ? (int)m : 0,
                               sw, kprev, lrow, ld, (const T*)inv, bt.sk, bt.sws, bt.sb, nb1 + nb2, rd, trg);
            CIMRGP_LAUNCH_CHECK(fn);
        }
    }
    return 0;
}

// One panel of a row-wise solve  B <- B L^-T : the 256-wide solve of the panel's columns, then
// the update of the columns right of it.  While more than `pair_above` columns lie beyond the
// next panel, the far columns are updated once per GROUP of panels with K = 256 x group size
// (fewer passes over B): every panel of a group but the last only updates the next panel's
// columns (with all the group's panels so far), the last one everything right of itself
// (adjacent panels are adjacent columns of B and of L).
// The rows' solve of one full panel [r0, r0 + 256): k_rows_step, with the previous panel's update of these columns
// fused in (`with_prev`: the caller left it out of its updates) or as the solve alone.
template <typename T>
static int rows_step_launch(T* b, int64_t ldb, int64_t m, const T* lmat, int64_t ld, const T* ws, int64_t r0, bool with_prev,
                            hipStream_t st, const char* fn, PotrfBatch bt = PotrfBatch())
{
    // (a batch: blockIdx.y = matrix, strides of the rows' arena, the matrices and the workspaces)
    const dim3 grid((unsigned)((m + RowsStep<T>::R - 1) / RowsStep<T>::R), (unsigned)bt.count);
    if (with_prev)
        hipLaunchKernelGGL((k_rows_step<T, true>), grid, dim3(256), 0, st, b + r0, ldb, (int)m,
                           (const T*)(lmat + r0 * ld + (r0 - CIMRGP_NB)), ld, (const T*)(ws + (r0 / SB) * (SB * SB)),
                           (const T*)nullptr, (int64_t)0, 0, 1, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0, bt.sb, bt.sk, bt.sws);
    else
        hipLaunchKernelGGL((k_rows_step<T, false>), grid, dim3(256), 0, st, b + r0, ldb, (int)m,
                           (const T*)(lmat + r0 * ld + r0), ld, (const T*)(ws + (r0 / SB) * (SB * SB)),
                           (const T*)nullptr, (int64_t)0, 0, 1, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0, bt.sb, bt.sk, bt.sws);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template <typename T>
static int rows_panel_step(T* b, int64_t ldb, int64_t m, const T* lmat, int64_t ld, int64_t n, const T* ws,
                           int64_t r0, PanelGroup& grp, int64_t pair_above, hipStream_t st, const char* fn,
                           PotrfBatch bt = PotrfBatch())
{
    const int64_t rw = (n - r0 < CIMRGP_NB) ? (n - r0) : CIMRGP_NB;
    const int64_t r1 = r0 + rw;
    GemmBatch gb; gb.count = bt.count; gb.sc = gb.sa = bt.sb; gb.sb = bt.sk;
    const bool step_ok = knobs().rows_step != 0 && m > 0;
    const bool near_pending = grp.near_pending;
    grp.near_pending = false;
    if (step_ok && rw == CIMRGP_NB) {
        int rcs = rows_step_launch<T>(b, ldb, m, lmat, ld, ws, r0, near_pending, st, fn, bt);
        if (rcs) return rcs;
    } else {
        if (near_pending) {                      // (cannot happen: the promise below is made for full panels only)
            int rcn = gemm_nt_sub<T>(b + r0, ldb, b + r0 - CIMRGP_NB, ldb, lmat + r0 * ld + r0 - CIMRGP_NB, ld, m, rw, CIMRGP_NB, false, st, gb);
            if (rcn) return rcn;
        }
        hipLaunchKernelGGL((k_trsm256<T>), dim3((unsigned)((m + TR - 1) / TR), (unsigned)bt.count), dim3(256), 0, st,
                           b + r0, ldb, (int)m, (int)rw, (const T*)(lmat + r0 * ld + r0), ld,
                           (const T*)(ws + (r0 / SB) * (SB * SB)), bt.sb, bt.sk, bt.sws);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    if (n <= r1) { grp = PanelGroup(); return 0; }
    const int64_t rn = (n - r1 < CIMRGP_NB) ? (n - r1) : CIMRGP_NB;
    if (grp.g0 < 0) {
        const int g = group_size(n - (r1 + rn), pair_above);
        if (g > 1) { grp.g0 = r0; grp.left = g; }
    }
    if (grp.g0 >= 0 && grp.left > 1 && n > r1 + rn) {
        --grp.left;
        return gemm_nt_sub<T>(b + r1, ldb, b + grp.g0, ldb, lmat + r1 * ld + grp.g0, ld, m, rn, (int)(r1 - grp.g0), false, st, gb);
    }
    const int64_t kk0 = (grp.g0 >= 0) ? grp.g0 : r0;
    grp = PanelGroup();
    if (step_ok && kk0 == r0 && rw == CIMRGP_NB && rn == CIMRGP_NB) {
        // the next panel's columns take this panel's update inside their own solve (k_rows_step)
        grp.near_pending = true;
        if (n <= r1 + rn) return 0;
        return gemm_nt_sub<T>(b + r1 + rn, ldb, b + r0, ldb, lmat + (r1 + rn) * ld + r0, ld, m, n - (r1 + rn), (int)rw, false, st, gb);
    }
    return gemm_nt_sub<T>(b + r1, ldb, b + kk0, ldb, lmat + r1 * ld + kk0, ld, m, n - r1, (int)(r1 - kk0), false, st, gb);
}

template <typename T, bool FACTOR>
static int panel_sweep(T* kmat, int64_t n, int64_t ld, T* ws, int32_t* info,
                       T* b, int64_t m, int64_t ldb, hipStream_t st, PotrfBatch bt = PotrfBatch())
{
    const char* fn = FACTOR ? "cimrgp_potrf" : "cimrgp_trsm_rows";
    const bool rows = (b != nullptr && m > 0);
    PanelGroup rows_grp;
    for (int64_t k0 = 0; k0 < n; k0 += CIMRGP_NB) {
        const int64_t w = (n - k0 < CIMRGP_NB) ? (n - k0) : CIMRGP_NB;
        const int64_t k1 = k0 + w;
        if (!FACTOR && rows) {
            int rc = rows_panel_step<T>(b, ldb, m, kmat, ld, n, ws, k0, rows_grp, ROWS_PAIR_ABOVE_SOLVE, st, fn, bt);
            if (rc) return rc;
            continue;
        }
        if (FACTOR) {
            int rcc = panel_chain<T>(kmat, n, ld, ws, info, k0, w, b, m, ldb, bt, st, fn, true);
            if (rcc) return rcc;
        }
        if (n > k1) {
            if (FACTOR) {
                const double mm = (double)(n - k1);
                hipEvent_t rec = rec_open(st, mm * (mm + 1.0) * (double)w * (double)bt.count, (mm * (mm + 1.0) + mm * (double)w) * (double)sizeof(T) * (double)bt.count);   // lower SYRK: M(M+1)K flop
                GemmBatch gb; gb.count = bt.count; gb.sc = gb.sa = gb.sb = bt.sk;
                int rc = gemm_nt_sub<T>(kmat + k1 * ld + k1, ld, kmat + k1 * ld + k0, ld,
                                        kmat + k1 * ld + k0, ld, n - k1, n - k1, (int)w, true, st, gb);
                if (rec) (void)hipEventRecord(rec, st);
                if (rc) return rc;
            }
            if (rows) {
                GemmBatch gb; gb.count = bt.count; gb.sc = gb.sa = bt.sb; gb.sb = bt.sk;
                int rc = gemm_nt_sub<T>(b + k1, ldb, b + k0, ldb, kmat + k1 * ld + k0, ld,
                                        m, n - k1, (int)w, false, st, gb);
                if (rc) return rc;
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------
// One-queue factorisation with the updates riding in the chain's launches (round 3; "Riders" above).
// Panel p = columns [k0, k1), next panel [k1, k2), previous panel `prev` = [q0, k0).  What the chain of
// panel p needs is that its OWN columns hold every earlier panel's contribution; everything else may lag.
//   launch              chain part                         riders
//   D(p)   diag block 0 (+ K = 64 prologue: prev's last    PH3(prev): prev's last sub-block -> panel p's columns (not its
//          sub-block, if PH3 is pending)                   tile (0,0)); ROWS(prev); NEAR(prev) 1st part; FAR(prev) share
//   L1(p)  solve sub-block 0 | diag block 1                NEAR(prev) rest; FAR(prev) share
//   L2(p)  solve sub-block 1 | diag block 2                PH(p, 0): sub-block 0 of p -> panel p+1's columns (K = 64); FAR(prev)
//   L3(p)  solve sub-block 2 | diag block 3                PH(p, 1); FAR(prev) share
//   T(p)   solve sub-block 3                               PH(p, 2); FAR(prev) rest
// with  NEAR(prev) = panel p+1's columns (all rows from k1) -= prev's panel, K = 256
//       FAR(prev)  = lower triangle from k2 on             -= prev's panel, K = 256
//       ROWS(prev) = carried rows, columns from k0 on      -= their own prev columns x prev's panel.
// Who writes what when: panel p's columns -- PH3(prev) in D(p) only, beside workgroup 0 on a different tile;
// panel p+1's columns -- NEAR(prev) in D, L1, then PH(p, s) from L2 on; beyond -- FAR(prev) only; launches of
// one queue follow one another, so each of these sets is complete before its first reader starts.
// `prev_w` > 0: the sweep starts behind a final panel [k_begin - prev_w, k_begin) whose contribution has been
// applied to panel k_begin's columns ONLY (the tail of a look-ahead factorisation, after its last head update).
// ---------------------------------------------------------------------------
// Optional second queue of a fused sweep: while FAR(prev) is large the riders cannot hide it (the first panels of
// a tail were rider-bound: 225 us per panel at 4352 trailing rows against a chain of ~115), so it runs as a
// persistent launch on `bulk` within `cus` compute units, beside the chain on the others.  FAR(prev) touches
// columns from k2 on only, the chain of panel p and its riders (PH3, NEAR, PH) the columns before k2: nothing
// else to order than "FAR of the panel before last is done" before a chain starts and "prev is final" before a FAR.
struct FarBulk {
    hipStream_t bulk = nullptr;
    std::vector<hipEvent_t>* ev = nullptr;
    size_t* ne = nullptr;
    int cus = 0;
    int64_t min_rows = 1 << 30;          // FAR on the bulk queue while n - k2 >= min_rows
    // called once per panel [k0, k1) after its chain has been enqueued, with an event that says "panel final"
    // (carried rows that follow the factorisation on queues of their own)
    std::function<int(int64_t, int64_t, hipEvent_t)> on_final;
};

template <typename T>
static int fused_sweep(T* kmat, int64_t n, int64_t ld, T* ws, int32_t* info, T* b, int64_t m, int64_t ldb,
                       PotrfBatch bt, hipStream_t st, int64_t k_begin = 0, int64_t prev_w = 0, const FarBulk* fb = nullptr)
{
    const char* fn = "cimrgp_potrf";
    const bool rows = (b != nullptr && m > 0);
    hipEvent_t ev_prev_final = nullptr;                     // (far-bulk mode) panel `prev` is final, on st
    hipEvent_t ev_far_last = nullptr;                       // (far-bulk mode) the last FAR launched on the bulk queue
    auto next_event = [&]() { return (*fb->ev)[(*fb->ne)++]; };
    if (fb && prev_w > 0) {
        ev_prev_final = next_event();
        hipError_t e = hipEventRecord(ev_prev_final, st);
        if (e != hipSuccess) return check_hip(e, fn, "hipEventRecord");
    }
    // the chain parts of the five launches last about this long alone (us)
    // (round 4, knobs().rider_lean: as measured in the tail of an N = 8192 factorisation without riders' help)
    static const double chain_us_r3[5] = {17.0, 22.0, 27.0, 31.0, 12.0};
    static const double chain_us_r4[5] = {15.0, 21.0, 24.0, 28.0, 13.0};
    const bool lean = knobs().rider_lean != 0;
    const double* chain_us = lean ? chain_us_r4 : chain_us_r3;
    int64_t q0 = (prev_w > 0) ? k_begin - prev_w : -1;      // previous panel (-1: none)
    int64_t qw = prev_w;
    bool ph3_pending = false;                                // prev's last sub-block still owed to this panel's columns
    bool rows_pending = false;                               // ROWS(prev) owed (at a tail entry the rows have seen prev already)
    auto tiles64 = [](int64_t v) { return (v + 63) / 64; };
    for (int64_t k0 = k_begin; k0 < n; k0 += CIMRGP_NB) {
        const int64_t w = (n - k0 < CIMRGP_NB) ? (n - k0) : CIMRGP_NB;
        const int64_t k1 = k0 + w;
        const int64_t wn = (k1 < n) ? ((n - k1 < CIMRGP_NB) ? (n - k1) : CIMRGP_NB) : 0;
        const int64_t k2 = k1 + wn;
        Riders<T> rd[5];
        for (int i = 0; i < 5; ++i) rd[i] = no_riders<T>();
        bool far_launched = false;
        auto add = [&](int launch, const RiderJob<T>& jb) {
            if (jb.count <= 0) return;
            Riders<T>& r = rd[launch];
            r.job[r.njobs++] = jb;
            r.total += jb.count;
        };
        auto rect_job = [&](T* c, int64_t ldc, const T* a, int64_t lda, const T* bb, int64_t ldbb, int64_t mm, int64_t nn, int64_t kk) {
            RiderJob<T> jb;
            jb.c = c; jb.a = a; jb.b = bb; jb.ldc = ldc; jb.lda = lda; jb.ldb = ldbb;
            jb.m = (int)mm; jb.n = (int)nn; jb.k = (int)kk; jb.lower = 0; jb.tiles_n = (int)tiles64(nn);
            jb.first = 0; jb.count = (int)(tiles64(mm) * tiles64(nn)); jb.skip00 = 0; jb.rows_job = 0;
            return jb;
        };
        if (q0 >= 0) {
            const T* pa = kmat + q0;                          // prev's panel columns, row r at pa + r * ld
            if (ph3_pending) {
                // prev's last 64 columns -> this panel's columns, every row from k0 (tile (0,0) is workgroup 0's)
                RiderJob<T> jb = rect_job(kmat + k0 * ld + k0, ld, kmat + k0 * ld + (k0 - SB), ld, kmat + k0 * ld + (k0 - SB), ld,
                                          n - k0, w, SB);
                jb.skip00 = 1;
                add(0, jb);
            }
            int64_t ph3_tiles = rd[0].total;                  // K = 64 tiles
            int64_t rows_near_tiles = 0, rows_far_tiles = 0;
            if (rows_pending) {
                // the carried rows' columns of THIS panel: needed by the panel's first link (which solves them)
                RiderJob<T> jb = rect_job(b + k0, ldb, b + q0, ldb, pa + k0 * ld, ld, m, w, qw);
                jb.rows_job = 1;
                add(0, jb);
                rows_near_tiles = jb.count;
                if (wn > 0) rows_far_tiles = tiles64(m) * tiles64(n - k1);      // ... and everything right of it: any launch
            }
            if (wn > 0) {
                const int64_t near_tiles = tiles64(n - k1) * tiles64(wn);
                const int64_t tf = (n > k2) ? tiles64(n - k2) : 0;
                // FAR(prev) on the bulk queue (persistent, fb->cus units) while it is large; as riders otherwise
                const bool far_on_bulk = fb && fb->cus >= 8 && ev_prev_final && bt.count == 1 && n - k2 >= fb->min_rows && (n - k2) % 128 == 0 &&
                                         qw == CIMRGP_NB && gemm_pers_head_tiles(n - k2, (int)qw, (int)sizeof(T)) > 0;
                if (far_on_bulk) {
                    hipError_t e = hipStreamWaitEvent(fb->bulk, ev_prev_final, 0);
                    if (e != hipSuccess) return check_hip(e, fn, "hipStreamWaitEvent");
                    GemmBatch gb; gb.pers = fb->cus; gb.pers_force = 1;
                    const double mm = (double)(n - k2);
                    hipEvent_t rec = rec_open(fb->bulk, mm * (mm + 1.0) * (double)qw, (mm * (mm + 1.0) + mm * (double)qw) * (double)sizeof(T));
                    int rcf = gemm_nt_sub<T>(kmat + k2 * ld + k2, ld, pa + k2 * ld, ld, pa + k2 * ld, ld, n - k2, n - k2, (int)qw, true, fb->bulk, gb);
                    if (rec) (void)hipEventRecord(rec, fb->bulk);
                    if (rcf) return rcf;
                    far_launched = true;
                }
                const int64_t far_tiles = far_on_bulk ? 0 : tf * (tf + 1) / 2;
                // How many of these K = 256 tiles each launch takes.  A launch's riders run in ROUNDS of
                // (2 workgroups per compute unit - the launch's own chain workgroups), ~20 us per round of
                // K = 256 tiles, and a launch lasts max(its chain part, its rounds): whole rounds are given
                // to the launches whose chain part they lengthen least (a launch with 1.3 rounds of riders
                // takes two rounds' time: the first version of this schedule, split by the chain parts'
                // durations, took 221 us per panel at 4352 trailing rows where 7 packed rounds take ~150).
                const int slots = 2 * (far_on_bulk ? 256 - fb->cus : 256);
                const double t_round = lean ? (double)knobs().rider_round_us : 20.0;
                const int64_t rows_below = n - k1;
                int nchain_i[5], fixed_i[5];
                for (int i = 0; i < 5; ++i) {
                    // rows the launch's panel solve covers: below diagonal block i + 1 (links), below the panel (last solve)
                    const int64_t below = (i == 4) ? rows_below : n - k0 - SB * (i + 1);
                    nchain_i[i] = (i == 0) ? 1 : (int)((below + TR - 1) / TR) + (i < 4 ? 1 : 0) + (int)(((rows ? m : 0) + TR - 1) / TR);
                    fixed_i[i] = 0;
                }
                // riders already placed (PH3, ROWS in launch 0; this panel's PH(p, s) go to launches 2..4: K = 64 tiles,
                // about a third of a K = 256 tile each)
                fixed_i[0] = (int)(ph3_tiles / 3 + rows_near_tiles);
                if (w == CIMRGP_NB) for (int i = 2; i < 5; ++i) fixed_i[i] = (int)(tiles64(n - k1) * tiles64(wn)) / 3;
                // Round 4 (lean): a launch starts with NO round of K = 256 riders; rounds go first to the launches whose
                // chain part outlasts a round anyway (the links), and the first diagonal kernel and the last solve --
                // 15 and 13 us alone, 25-29 and 16 with a round of riders -- take riders only when the links are full.
                // (Until round 4 every launch started with one round, and NEAR and FAR filled launch 0 first.)
                int rounds[5] = {1, 1, 1, 1, 1};
                if (lean) for (int i = 0; i < 5; ++i) rounds[i] = 0;
                auto capacity = [&](int i) { const int64_t c = (int64_t)rounds[i] * (slots - nchain_i[i]) - fixed_i[i]; return c > 0 ? c : 0; };
                const int64_t need = near_tiles + far_tiles + rows_far_tiles;
                for (;;) {
                    int64_t cap = 0;
                    for (int i = 0; i < 5; ++i) cap += capacity(i);
                    // NEAR must fit launches 0 and 1
                    const bool near_ok = capacity(0) + capacity(1) >= near_tiles;
                    if (cap >= need && near_ok) break;
                    int best = -1;
                    double best_cost = 1e30;
                    for (int i = (near_ok ? 0 : 0); i < (near_ok ? 5 : 2); ++i) {
                        const double now = (rounds[i] * t_round > chain_us[i]) ? rounds[i] * t_round : chain_us[i];
                        const double then = ((rounds[i] + 1) * t_round > chain_us[i]) ? (rounds[i] + 1) * t_round : chain_us[i];
                        const double cost = (then - now) / (double)(slots - nchain_i[i]);
                        if (cost < best_cost - 1e-12) { best_cost = cost; best = i; }
                    }
                    ++rounds[best];
                    if (rounds[best] > 64) break;           // cannot happen (guards the loop)
                }
                // NEAR(prev): launches 0 and 1 only (from launch 2 on this panel's sub-blocks update the same columns)
                RiderJob<T> nr = rect_job(kmat + k1 * ld + k1, ld, pa + k1 * ld, ld, pa + k1 * ld, ld, n - k1, wn, qw);
                int64_t near0 = capacity(0) < near_tiles ? capacity(0) : near_tiles;
                if (lean) near0 = 0;                                                         // the first link before the diagonal kernel
                if (near_tiles - near0 > capacity(1)) near0 = near_tiles - capacity(1);     // (capacities cover it: near_ok)
                RiderJob<T> n0 = nr; n0.first = 0; n0.count = (int)near0; add(0, n0);
                RiderJob<T> n1 = nr; n1.first = (int)near0; n1.count = (int)(near_tiles - near0); add(1, n1);
                {
                    // FAR(prev), then the carried rows' far update, over the launches' remaining capacities
                    RiderJob<T> fr;
                    fr.c = kmat + k2 * ld + k2; fr.a = pa + k2 * ld; fr.b = pa + k2 * ld; fr.ldc = fr.lda = fr.ldb = ld;
                    fr.m = fr.n = (int)(n - k2); fr.k = (int)qw; fr.lower = 1; fr.tiles_n = (int)tf; fr.skip00 = 0; fr.rows_job = 0;
                    fr.first = 0; fr.count = 0;
                    RiderJob<T> rf = fr;
                    if (rows_far_tiles > 0) {
                        rf = rect_job(b + k1, ldb, b + q0, ldb, pa + k1 * ld, ld, m, n - k1, qw);
                        rf.rows_job = 1;
                    }
                    int64_t done_f = 0, done_r = 0;
                    static const int order_r3[5] = {0, 1, 2, 3, 4}, order_r4[5] = {1, 2, 3, 0, 4};
                    const int* order = lean ? order_r4 : order_r3;
                    for (int oi = 0; oi < 5; ++oi) {
                        const int i = order[oi];
                        int64_t room = capacity(i) - (i == 0 ? near0 : i == 1 ? (near_tiles - near0) : 0);
                        if (room < 0) room = 0;
                        int64_t take_f = far_tiles - done_f;
                        if (oi < 4 && take_f > room) take_f = room;
                        room -= take_f;
                        int64_t take_r = rows_far_tiles - done_r;
                        if (oi < 4 && take_r > room) take_r = (room > 0 ? room : 0);
                        if (take_f > 0) { RiderJob<T> part = fr; part.first = (int)done_f; part.count = (int)take_f; add(i, part); }
                        if (take_r > 0) { RiderJob<T> part = rf; part.first = (int)done_r; part.count = (int)take_r; add(i, part); }
                        done_f += take_f;
                        done_r += take_r;
                    }
                }
            }
        }
        if (wn > 0 && w == CIMRGP_NB) {
            // this panel's sub-blocks 0, 1, 2 -> the next panel's columns, each as soon as it is final
            for (int sblk = 0; sblk < 3; ++sblk) {
                const int64_t cs = k0 + SB * sblk;
                add(2 + sblk, rect_job(kmat + k1 * ld + k1, ld, kmat + k1 * ld + cs, ld, kmat + k1 * ld + cs, ld, n - k1, wn, SB));
            }
        }
        if (fb) {
            // the FAR launched during the previous iteration wrote the columns this chain's riders (and, when FAR
            // rides again, its FAR tiles) are about to touch
            if (ev_far_last) {
                hipError_t e = hipStreamWaitEvent(st, ev_far_last, 0);
                if (e != hipSuccess) return check_hip(e, fn, "hipStreamWaitEvent");
                ev_far_last = nullptr;
            }
            if (far_launched) {
                ev_far_last = next_event();
                hipError_t e = hipEventRecord(ev_far_last, fb->bulk);
                if (e != hipSuccess) return check_hip(e, fn, "hipEventRecord");
            }
        }
        int rc = panel_chain<T>(kmat, n, ld, ws, info, k0, w, b, m, ldb, bt, st, fn, true, false, rd, ph3_pending);
        if (rc) return rc;
        if (fb && (k1 < n || fb->on_final)) {
            ev_prev_final = next_event();
            hipError_t e = hipEventRecord(ev_prev_final, st);
            if (e != hipSuccess) return check_hip(e, fn, "hipEventRecord");
            if (fb->on_final) {
                rc = fb->on_final(k0, k1, ev_prev_final);
                if (rc) return rc;
            }
        }
        q0 = k0;
        qw = w;
        ph3_pending = (wn > 0 && w == CIMRGP_NB);
        rows_pending = rows && k1 < n;
        // a ragged panel that still has columns to its right cannot happen (only the last panel is ragged)
    }
    if (fb && ev_far_last) {
        hipError_t e = hipStreamWaitEvent(st, ev_far_last, 0);
        if (e != hipSuccess) return check_hip(e, fn, "hipStreamWaitEvent");
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
constexpr int POSTED_WORD = 32;
constexpr int MAX_CTX = 8;               // look-ahead contexts per device: one per concurrently factoring caller stream
constexpr int64_t SINGLE_QUEUE_MAX = 5120;   // n at or below this: one queue, no look-ahead (see potrf_run)

struct LookAhead {
    hipStream_t side = nullptr;        // panel chain (high priority, all compute units)
    hipStream_t bulk = nullptr;        // trailing updates (unused: the caller's stream runs them)
    hipStream_t rows = nullptr;        // carried rows: lags behind the factorisation
    hipStream_t rows_far = nullptr;    // carried rows: far part of each panel's update (beside the rows' own panel chain)
    std::vector<hipEvent_t> ev;
    int* flag = nullptr;               // device counter: head tiles stored by the combined update launches (k_gate polls it);
                                       // flag[POSTED_WORD] (a line of its own): panels posted as final by the chain (k_post)
    hipStream_t owner = nullptr;       // the caller stream this context was created for
    bool gate_ok = false;              // this context may hold a kernel that waits for another one (k_gate): the device's first context only
    std::mutex enqueue;                // one factorisation at a time enqueues on this context's queues
    hipEvent_t last_done = nullptr;    // behind the latest look-ahead factorisation on this context (recorded on its caller's stream)
};
std::mutex g_reg_mutex;                // guards g_ctx and context creation
std::vector<LookAhead*> g_ctx[16];     // per device; contexts live as long as the process

// Streams of one context.  All three or none: a context without its rows queue would have to
// order the carried rows on the caller's stream, which the schedule below does not do.
LookAhead* make_ctx(int dev)
{
    LookAhead* la = new LookAhead;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    bool ok = hipStreamCreateWithPriority(&la->side, hipStreamNonBlocking, hi) == hipSuccess;
    // (Round 2 could run the update kernels on CU-masked queues -- hipExtStreamCreateWithCUMask,
    // CIMRGP_RESERVE_CUS -- to keep compute units free for the chain: measured useless three times
    // (HISTORY.md, rounds 1-2, rejected (i)) and removed in round 3: masked streams are BLOCKING streams that
    // synchronise with the legacy default stream, and they were the one kind of object still alive at
    // process exit in the profiler runs that crashed in an exit handler.  The persistent update kernel
    // splits the machine instead: a launch of G workgroups occupies G compute units.)
    if (ok) ok = hipStreamCreateWithPriority(&la->rows, hipStreamNonBlocking, lo) == hipSuccess;
    if (ok) ok = hipMalloc(&la->flag, 256) == hipSuccess;
    // (the carried rows' second queue is created on first use: ensure_rows_far)
    if (!ok) {
        if (la->side) (void)hipStreamDestroy(la->side);
        if (la->bulk) (void)hipStreamDestroy(la->bulk);
        if (la->rows) (void)hipStreamDestroy(la->rows);
        if (la->rows_far) (void)hipStreamDestroy(la->rows_far);
        if (la->flag) (void)hipFree(la->flag);
        delete la;
        return nullptr;
    }
    return la;
}

// The context serving caller stream `st` on the current device (created on first use under the
// registry lock; beyond MAX_CTX contexts callers share one by stream hash, which only serialises
// them on its queues).  Independent blocks of a layer are factored on different caller streams
// and so get different contexts: their panel chains run side by side.
LookAhead* acquire_ctx(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> guard(g_reg_mutex);
    std::vector<LookAhead*>& list = g_ctx[dev];
    for (LookAhead* la : list)
        if (la->owner == st) return la;
    if ((int)list.size() < MAX_CTX) {
        LookAhead* la = make_ctx(dev);
        if (la == nullptr) return list.empty() ? nullptr : list[0];
        la->owner = st;
        // One context per device may use the gate.  Streams share the runtime's four hardware queues; with two
        // gate users a gate of A could sit in front of the update B's gate waits for and vice versa.  Within one
        // context the update is always enqueued ahead of its gate, so a single user cannot block itself.
        la->gate_ok = list.empty();
        list.push_back(la);
        return la;
    }
    return list[(reinterpret_cast<uintptr_t>(st) >> 6) % MAX_CTX];
}

// Destroys every look-ahead context (streams, events) of every device.  The caller guarantees that no
// factorisation is in flight or will be enqueued concurrently (cimrgp_shutdown).
int destroy_contexts()
{
    std::lock_guard<std::mutex> guard(g_reg_mutex);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int dev = 0; dev < 16; ++dev) {
        if (g_ctx[dev].empty()) continue;
        (void)hipSetDevice(dev);
        for (LookAhead* la : g_ctx[dev]) {
            for (hipStream_t q : {la->side, la->bulk, la->rows, la->rows_far})
                if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
            for (hipEvent_t e : la->ev) (void)hipEventDestroy(e);
            if (la->last_done) (void)hipEventDestroy(la->last_done);
            if (la->flag) (void)hipFree(la->flag);
            delete la;
        }
        g_ctx[dev].clear();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return 0;
}

// Event pool of a context; call with la->enqueue held.
bool grow_events(LookAhead* la, size_t nevents)
{
    while (la->ev.size() < nevents) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        la->ev.push_back(e);
    }
    return true;
}

template <typename T>
int factor_panel(T* kmat, int64_t n, int64_t ld, T* ws, int32_t* info, int64_t k0, int64_t w, hipStream_t st, bool alone,
                 bool first_done = false)
{
    return panel_chain<T>(kmat, n, ld, ws, info, k0, w, (T*)nullptr, 0, 0, PotrfBatch(), st, "cimrgp_potrf", alone, first_done);
}
}  // namespace

// The queue of `st`'s look-ahead context that is idle between two factorisations on `st` (the carried rows' own
// queue: the rows of a factorisation start ~2 ms after its first panel at N = 8192): a caller that pipelines independent
// blocks runs the latency-bound solve / prediction of block i there, beside the first panels of block i+1
// (cimrgp_solve_queue).  A fifth stream of the caller's own for that purpose costs more than it hides (measured: 8.1 ->
// 10.2 ms per step; the runtime serves four hardware queues).  `st` itself when it owns no context.
hipStream_t solve_queue_for(hipStream_t st)
{
    LookAhead* la = acquire_ctx(st);
    return (la != nullptr && la->owner == st && la->rows != nullptr) ? la->rows : st;
}


// The queue of `st`'s look-ahead context that falls idle BEFORE a factorisation on `st` ends (the panel chain's queue: the
// last third of a factorisation runs on one queue, potrf_run's tail): the front end of the NEXT independent block can run
// there beside that tail (cimrgp_front_queue).  `st` itself when it owns no context.
hipStream_t front_queue_for(hipStream_t st)
{
    LookAhead* la = acquire_ctx(st);
    return (la != nullptr && la->owner == st && la->side != nullptr) ? la->side : st;
}

int potrf_shutdown()
{
    {
        std::lock_guard<std::mutex> guard(g_profile_mutex);
        for (auto& r : g_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
        for (auto& r : g_free) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
        g_recs.clear();
        g_free.clear();
        g_profile = false;
    }
    return destroy_contexts();
}

// Workspace layout: [ceil(n/64) slabs of 64x64 inverses][ceil(n/256) blocks of 256x256 invT].
template <typename T>
static int build_invT(const T* kmat, int64_t n, int64_t ld, T* ws, hipStream_t st, PotrfBatch bt = PotrfBatch())
{
    const char* fn = "cimrgp_potrf";
    const int64_t nslab = (n + SB - 1) / SB, npan = (n + CIMRGP_NB - 1) / CIMRGP_NB;
    T* invT = ws + nslab * (SB * SB);
    const unsigned nbatch = (unsigned)bt.count;
    // Full panels (round 4): invT_p = I L_pp^-T is the carried rows' panel step applied to the identity -- one launch of
    // k_rows_step for all of them (16 workgroups per panel, ~12 us) instead of k_invT_panel's four dependent sub-steps
    // per 32-row strip with their global round trips (40 us at the end of every factorisation); a ragged last panel
    // keeps k_invT_panel.
    // (single matrices only: in a batch of many small blocks the 16-row workgroups' traffic on the diagonal blocks costs
    // more than the strips' latency -- 128 blocks of 2048: fit 11.6 -> 12.2 ms)
    const int64_t nfull = (bt.count == 1) ? n / CIMRGP_NB : 0;
    const int64_t blk = (int64_t)CIMRGP_NB * CIMRGP_NB;
    if (nfull > 0) {
        hipLaunchKernelGGL((k_rows_step<T, false>), dim3(CIMRGP_NB / RowsStep<T>::R, (unsigned)(nfull * bt.count)), dim3(256), 0, st,
                           invT, (int64_t)CIMRGP_NB, (int)CIMRGP_NB, kmat, ld, (const T*)ws, (const T*)nullptr, (int64_t)0, 2, (int)nfull,
                           blk, CIMRGP_NB * ld + CIMRGP_NB, (int64_t)(CIMRGP_NB / SB) * (SB * SB), (int64_t)0,
                           bt.sws, bt.sk, bt.sws);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    if (npan > nfull) {
        hipLaunchKernelGGL((k_invT_panel<T>), dim3(CIMRGP_NB / TR, (unsigned)(npan - nfull), nbatch), dim3(256), 0, st,
                           invT, kmat, ld, (int)n, (const T*)ws, bt.sk, bt.sws, (int)nfull);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    // Round 4: the backward solve takes TWO panels per step through the 512 x 512 inverse
    //     [A 0; B C]^-T = [A^-T  X; 0  C^-T],   X = -A^-T B^T C^-T  (256 x 256, dense),
    // halving its chain of dependent launches (62 -> 32 at n = 8192: 0.33 -> 0.2 ms).  X is what the carried rows' panel
    // step computes for the rows of A^-T (already there: invT of the pair's first panel) with a zero right-hand side:
    // one launch of k_rows_step for all pairs of full panels, behind the inverses (potrs_run: bwd_pairs).
    // Above the one-queue size only (bwd_pairs(n) in common.hpp: the solve applies the same rule): below it the launch that
    // builds X costs what the shorter chain saves.
    const int64_t npairs = bwd_pairs(n);
    if (npairs > 0) {
        T* xbase = invT + npan * (CIMRGP_NB * CIMRGP_NB);
        hipLaunchKernelGGL((k_rows_step<T, true>), dim3(CIMRGP_NB / RowsStep<T>::R, (unsigned)(npairs * bt.count)), dim3(256), 0, st,
                           xbase, (int64_t)CIMRGP_NB, (int)CIMRGP_NB, (const T*)(kmat + (int64_t)CIMRGP_NB * ld), ld,
                           (const T*)(ws + (CIMRGP_NB / SB) * (SB * SB)), (const T*)invT, (int64_t)CIMRGP_NB, 1, (int)npairs,
                           blk, 2 * CIMRGP_NB * ld + 2 * CIMRGP_NB, (int64_t)(2 * CIMRGP_NB / SB) * (SB * SB), 2 * blk,
                           bt.sws, bt.sk, bt.sws);
        CIMRGP_LAUNCH_CHECK(fn);
    }
    return 0;
}

#define CIMRGP_HIP_TRY(call, what) \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) return check_hip(e__, "cimrgp_potrf", what); } while (0)

template <typename T>
int potrf_run(T* k, int64_t n, int64_t ld, T* ws, int32_t* info, T* b, int64_t m, int64_t ldb, hipStream_t st, hipStream_t ready_on)
{
    const bool rows = (b != nullptr && m > 0);
    const int64_t npanels = (n + CIMRGP_NB - 1) / CIMRGP_NB;
    // Small matrices: one queue.  Measured in round 1 (whole potrf, one queue vs look-ahead): n = 2048:
    // 1.23 vs 1.37 ms, 4096: 2.80 vs 3.04 -- and independent blocks of a layer run concurrently on
    // their callers' streams, which fills the machine better than look-ahead inside each of them.
    LookAhead* la = (n > SINGLE_QUEUE_MAX && npanels > 2) ? acquire_ctx(st) : nullptr;
    // `ready_on`: the queue on which the caller wrote K and B, when that is this context's chain queue (cimrgp_front_queue) and
    // not `st`.  The FIRST panel's chain then follows them there in queue order instead of waiting for `st` -- for a caller that
    // pipelines independent blocks it runs beside the previous factorisation's latency-bound tail, not behind it (~100 us of a
    // nearly idle machine per factorisation).  Everything after it waits for `st` as before.
    const bool early = la != nullptr && ready_on != nullptr && ready_on == la->side && ready_on != st && knobs().early_first_panel != 0;
    CIMRGP_HIP_TRY(hipMemsetAsync(info, 0, sizeof(int32_t), early ? la->side : st), "hipMemsetAsync(info)");
    if (la == nullptr) {
        int rc0 = fused_sweep<T>(k, n, ld, ws, info, b, m, ldb, PotrfBatch(), st);
        return rc0 ? rc0 : build_invT<T>(k, n, ld, ws, st);
    }

    // Host threads whose streams share a context serialise their ENQUEUE (microseconds); distinct
    // caller streams have distinct contexts and enqueue concurrently.
    std::lock_guard<std::mutex> guard(la->enqueue);
    if (!grow_events(la, (size_t)(9 * npanels + 16 + 4))) return fail("cimrgp_potrf", "hipEventCreate failed");
    hipStream_t sp = la->side;
    hipStream_t sb = la->bulk ? la->bulk : st;         // bulk trailing updates
    size_t ne = 0;
    // The device counter of the gate belongs to the CONTEXT, and beyond MAX_CTX contexts a context serves caller
    // streams other than its owner (acquire_ctx, by hash).  Only the owner's factorisations may reset and count on
    // it: a second caller stream would reset the counter under the owner's in-flight gates (which then expire) and
    // have its own gates satisfied by the owner's tiles (a silently wrong factor).  Everybody else keeps the head
    // update on the chain queue (round 2's schedule), which needs no counter.
    const bool may_gate = la->gate_ok && la->owner == st;
    // Everything below enqueues on several queues; an error return in the middle must not leave the caller's
    // stream running ahead of work already queued on them (the caller may free or reuse K / workspace / B):
    // the enqueue proper is `body`, and whatever it returns the queues are joined into `st` behind it.
    auto body = [&]() -> int {
    // The side stream runs the whole latency-bound chain in stream order -- "head" update of the
    // next panel's columns, then that panel's factorisation -- so that no inter-queue signal
    // sits between two links of the chain; the caller's stream runs the bulk of each trailing
    // update (and, off the chain, the carried rows).  Cross-stream edges: "panel final"
    // (side -> main, before the bulk update that reads it) and "bulk update done" (main -> side,
    // before the next head touches columns the bulk update wrote).
    if (may_gate) CIMRGP_HIP_TRY(hipMemsetAsync(la->flag, 0, 256, st), "hipMemsetAsync(flag)");
    int flag_expected = 0;                             // head tiles the chain has been told to wait for so far
    int posted = 0;                                    // la->flag[POSTED_WORD]: panels the chain has posted as final (k_post) so far
    bool final_posted = false;                         // ... the panel this iteration starts from among them
    hipEvent_t ev_start = la->ev[ne++];
    CIMRGP_HIP_TRY(hipEventRecord(ev_start, st), "hipEventRecord");
    if (!early) CIMRGP_HIP_TRY(hipStreamWaitEvent(sp, ev_start, 0), "hipStreamWaitEvent");
    if (sb != st) CIMRGP_HIP_TRY(hipStreamWaitEvent(sb, ev_start, 0), "hipStreamWaitEvent");
    int rc = factor_panel<T>(k, n, ld, ws, info, 0, (n < CIMRGP_NB) ? n : CIMRGP_NB, sp, true);
    if (rc) return rc;
    hipEvent_t ev_panel = la->ev[ne++];
    CIMRGP_HIP_TRY(hipEventRecord(ev_panel, sp), "hipEventRecord");
    hipEvent_t ev_rest = nullptr;                      // bulk update of the previous panel
    PanelGroup grp;                                    // open group of panels whose far update is still owed
    auto grp_open = [&]() { return grp.g0 >= 0; };
    bool tail_done = false;
    const int64_t single_tail_below = knobs().tail_below;
    hipEvent_t ev_bulk_last = nullptr;                 // last thing queued on the bulk stream
    // Bulk updates beside the chain: the persistent update kernel on all compute units but `chain_cus`,
    // which stay free for the chain's kernels (one persistent workgroup fills a unit's registers, so the
    // grid size IS the split).
    GemmBatch bulk_gb;
    if (knobs().gemm_pers > 0 && knobs().chain_cus > 0 && knobs().chain_cus < knobs().gemm_pers)
        bulk_gb.pers = knobs().gemm_pers - knobs().chain_cus;
    int64_t rows_next = 0;                             // first panel the carried rows have not seen yet
    // Panel k0 is final (event ev_final): solve + update the carried rows.  They form their own
    // chain (panel p+1 of the rows needs panel p of the rows) that depends on the factorisation
    // only through "panel k0 final", so it runs on its own queues and lags behind: nothing of it is
    // issued while the trailing updates are still large (that phase is MFMA-bound and the rows
    // would only take compute units away from the critical path); from then on it fills the
    // compute units the latency-bound panel chain leaves idle.
    // (Measured and rejected in round 2: cutting the rows into 2..4 slices on queues of their own so
    // that their latency-bound chains overlap -- 9.5 -> 10.1 / 12.6 / 21 ms at N = 8192 with 2050 rows:
    // more low-priority queues only add contention for the panel chain; 64-tile updates for the rows,
    // whose workgroups retire four times as often: 94.8 -> 92.6 posteriors/s; an earlier or later
    // start than 4608 trailing rows: 3072 / 5632 / 6656 / 8192 -> 89.2 / 94.3 / 91.6 / 89.5.)
    PanelGroup rows_grp;                               // carried rows: open group of panels whose far update is owed
    hipEvent_t ev_rows_far = nullptr;                  // carried rows: last far update queued on the second rows queue
    hipEvent_t ev_rows_far_prev = nullptr;             // ... and the one before it
    // second rows queue: created when first wanted (cimrgp_set_rows_queues(1) before the first
    // factorisation with carried rows means it never exists: a process then holds four streams)
    if (rows && rows_queues() == 2 && la->rows_far == nullptr) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&la->rows_far, hipStreamNonBlocking, lo) != hipSuccess) la->rows_far = nullptr;
    }
    const bool rows_pipeline = (rows_queues() == 2) && la->rows_far != nullptr;
    // Round 3 (knobs().rows_fused_tail): the carried rows do not start inside the look-ahead phase at all; at the
    // switch to the one-queue tail they catch up with the whole machine to themselves (`force`), and from
    // there on they ride in the chain's launches like the factorisation's own updates (fused_sweep).
    const bool rows_fused = rows && knobs().rows_fused_tail != 0;
    // trailing columns below which the carried rows start (set again below when this factorisation starts with early panels:
    // beside the previous factorisation's last panels the rows do better starting two panels later)
    int64_t rows_start = knobs().rows_start_below;
    auto rows_after_panel = [&](int64_t k0, int64_t k1, hipEvent_t ev_final, bool force = false) -> int {
        if (!rows) return 0;
        hipStream_t sq = la->rows;                     // always present (make_ctx: all queues or no context)
        const int64_t rows_start_below = rows_start;
        const bool defer = rows_fused ? (k1 < n) : ((n - k1 > rows_start_below) && (k1 < n));
        if (defer && !force) return 0;
        CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_final, 0), "hipStreamWaitEvent");
        // (pairing the rows' updates below that size was measured neutral-to-worse at N = 8192)
        for (int64_t r0 = rows_next; r0 <= k0; r0 += CIMRGP_NB) {
            const int64_t rw = (n - r0 < CIMRGP_NB) ? (n - r0) : CIMRGP_NB;
            const int64_t r1 = r0 + rw;
            const int64_t rn = (n - r1 < CIMRGP_NB) ? (n - r1) : CIMRGP_NB;
            if (!rows_pipeline || rows_grp.g0 >= 0 || (n > r1 + rn && group_size(n - (r1 + rn), knobs().rows_pair_above) > 1)) {
                // grouped far updates (large matrices): the rows' chain as one queue
                if (ev_rows_far) { CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_rows_far, 0), "hipStreamWaitEvent"); ev_rows_far = nullptr; }
                int rcr = rows_panel_step<T>(b, ldb, m, k, ld, n, ws, r0, rows_grp, knobs().rows_pair_above, sq, "cimrgp_potrf_rows");
                if (rcr) return rcr;
                continue;
            }
            // The rows' own chain on `sq`: panel r0's solve with the previous panel's update of its columns fused in
            // (k_rows_step; until round 4 a 64-tile update and k_trsm256, two latency-bound launches); the update of
            // everything beyond the next panel (the bulk of the flops) on a second queue.  far(p) needs the solved
            // columns of panel p only and writes the columns from panel p+2 on: the step of panel p+2 waits for it,
            // the step of panel p+1 does not.
            const bool step_ok = knobs().rows_step != 0;
            const bool near_pending = rows_grp.near_pending;
            rows_grp.near_pending = false;
            if (step_ok && rw == CIMRGP_NB) {
                if (near_pending && ev_rows_far_prev) CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_rows_far_prev, 0), "hipStreamWaitEvent");
                int rcs = rows_step_launch<T>(b, ldb, m, k, ld, ws, r0, near_pending, sq, "cimrgp_potrf_rows");
                if (rcs) return rcs;
            } else {
                if (near_pending) {              // (a promise is made for full panels only)
                    if (ev_rows_far_prev) CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_rows_far_prev, 0), "hipStreamWaitEvent");
                    int rcn = gemm_nt_sub<T>(b + r0, ldb, b + r0 - CIMRGP_NB, ldb, k + r0 * ld + r0 - CIMRGP_NB, ld, m, rw, CIMRGP_NB, false, sq);
                    if (rcn) return rcn;
                }
                hipLaunchKernelGGL((k_trsm256<T>), dim3((unsigned)((m + TR - 1) / TR)), dim3(256), 0, sq,
                                   b + r0, ldb, (int)m, (int)rw, (const T*)(k + r0 * ld + r0), ld,
                                   (const T*)(ws + (r0 / SB) * (SB * SB)));
                CIMRGP_LAUNCH_CHECK("cimrgp_potrf_rows");
            }
            if (n <= r1) continue;
            hipEvent_t ev_w = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_w, sq), "hipEventRecord");
            int rcr = 0;
            // (not for the last panel of a forced catch-up: whoever continues expects these columns complete)
            if (step_ok && rw == CIMRGP_NB && rn == CIMRGP_NB && !(force && r0 == k0)) {
                rows_grp.near_pending = true;
            } else {
                if (ev_rows_far) CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_rows_far, 0), "hipStreamWaitEvent");
                rcr = gemm_nt_sub<T>(b + r1, ldb, b + r0, ldb, k + r1 * ld + r0, ld, m, rn, (int)rw, false, sq);
                if (rcr) return rcr;
            }
            ev_rows_far_prev = ev_rows_far;
            if (n > r1 + rn) {
                hipStream_t sf = la->rows_far;
                CIMRGP_HIP_TRY(hipStreamWaitEvent(sf, ev_w, 0), "hipStreamWaitEvent");
                // The far update: whole 128-row tiles of the rows on the persistent kernel within a budget of compute
                // units (it then cannot crowd the factorisation out of the machine the way a 1000-workgroup launch of
                // the tile-per-workgroup kernel does), the few rows left over (the q target rows) in a thin launch.
                const int64_t m128 = (knobs().rows_cus >= 8 && knobs().gemm_pers >= 8) ? (m / 128) * 128 : 0;
                const int64_t nfar = n - (r1 + rn);
                if (m128 >= 128 && nfar % 128 == 0 && (m128 / 128) * (nfar / 128) >= 2 * knobs().rows_cus) {
                    GemmBatch gb; gb.pers = knobs().rows_cus; gb.pers_force = 1;
                    rcr = gemm_nt_sub<T>(b + r1 + rn, ldb, b + r0, ldb, k + (r1 + rn) * ld + r0, ld, m128, nfar, (int)rw, false, sf, gb);
                    if (rcr) return rcr;
                    if (m > m128) {
                        // On the rows' CHAIN queue (round 4), not behind the persistent launch: the far updates of the
                        // 128-row tiles run back to back on `sf` and are the critical path of this phase; this thin
                        // launch (240 workgroups for 2 rows, 17-65 us when the units are busy, plus a queue gap)
                        // took a fifth of that queue's time.  The thin rows are independent of the others, and on `sq`
                        // the next panel's step -- their only reader -- follows in queue order.
                        GemmBatch g0; g0.pers = 0;
                        rcr = gemm_nt_sub<T>(b + m128 * ldb + r1 + rn, ldb, b + m128 * ldb + r0, ldb, k + (r1 + rn) * ld + r0, ld,
                                             m - m128, nfar, (int)rw, false, sq, g0);
                    }
                } else {
                    rcr = gemm_nt_sub<T>(b + r1 + rn, ldb, b + r0, ldb, k + (r1 + rn) * ld + r0, ld, m, nfar, (int)rw, false, sf);
                }
                if (rcr) return rcr;
                ev_rows_far = la->ev[ne++];
                CIMRGP_HIP_TRY(hipEventRecord(ev_rows_far, sf), "hipEventRecord");
            }
        }
        if (force && rows_grp.near_pending) {
            // whoever continues (the fused sweep's riders) expects the next panel's columns complete
            const int64_t fw = k1 - k0, fn_ = (n - k1 < CIMRGP_NB) ? (n - k1) : CIMRGP_NB;
            if (ev_rows_far) CIMRGP_HIP_TRY(hipStreamWaitEvent(sq, ev_rows_far, 0), "hipStreamWaitEvent");
            int rcn = gemm_nt_sub<T>(b + k1, ldb, b + k0, ldb, k + k1 * ld + k0, ld, m, fn_, (int)fw, false, sq);
            if (rcn) return rcn;
            rows_grp.near_pending = false;
        }
        rows_next = k1;
        return 0;
    };
    int64_t k_begin = 0;
    if (early) {
        // ... and knobs().early_panels more panels one-queue style (update of everything right of panel p, then panel p+1, in
        // queue order on the chain queue): work of THIS factorisation done while the previous one's tail leaves the machine
        // two-thirds idle.  From panel k_begin on the look-ahead schedule below takes over, behind `st` as always.
        // (only while the context's previous factorisation is still in flight when this one is enqueued: with the machine
        //  to itself a factorisation is better off with look-ahead from the first panel on)
        const bool prev_in_flight = la->last_done != nullptr && hipEventQuery(la->last_done) == hipErrorNotReady;
        (void)hipGetLastError();                     // "not ready" is an answer, not an error for the launch checks below to find
        const int np_early = (sizeof(T) == 8 && prev_in_flight) ? knobs().early_panels : 0;
        GemmBatch early_gb = bulk_gb;
        if (knobs().early_cus >= 8) early_gb.pers = knobs().early_cus;
        for (int p = 0; p < np_early && rc == 0; ++p) {
            const int64_t k0e = (int64_t)p * CIMRGP_NB, k1e = k0e + CIMRGP_NB;
            if (n - k1e < 4 * CIMRGP_NB || (n - k1e) % 128 != 0) break;
            const double mme = (double)(n - k1e);
            hipEvent_t rec = rec_open(sp, mme * (mme + 1.0) * (double)CIMRGP_NB, (mme * (mme + 1.0) + mme * (double)CIMRGP_NB) * (double)sizeof(T));
            rc = gemm_nt_sub<T>(k + k1e * ld + k1e, ld, k + k1e * ld + k0e, ld, k + k1e * ld + k0e, ld, n - k1e, n - k1e, (int)CIMRGP_NB, true, sp, early_gb);
            if (rec) (void)hipEventRecord(rec, sp);
            if (rc) return rc;
            rc = factor_panel<T>(k, n, ld, ws, info, k1e, CIMRGP_NB, sp, true);
            if (rc) return rc;
            k_begin = k1e;
        }
        if (k_begin > 0) {
            rows_start = knobs().rows_start_below_early;
            ev_panel = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_panel, sp), "hipEventRecord");
        }
        CIMRGP_HIP_TRY(hipStreamWaitEvent(sp, ev_start, 0), "hipStreamWaitEvent");
    }
    for (int64_t k0 = k_begin; k0 < n; k0 += CIMRGP_NB) {
        const int64_t w  = (n - k0 < CIMRGP_NB) ? (n - k0) : CIMRGP_NB;
        const int64_t k1 = k0 + w;
        // With carried rows (round 3): the last rows_beside_tail_below columns are factored by the same one-queue fused
        // sweep while the rows keep following on their own queues, one panel behind (the sweep tells them when a
        // panel is final).  Entered later than the tail without rows (2560 against 4864 trailing columns): while the
        // rows' far updates are large the sweep's riders would queue behind them for compute units (113 posteriors/s
        // entered at 4864 and 107 at 6144 against 117.3 without and 119.4 at 2560).
        const bool rows_beside = rows && !rows_fused && knobs().rows_beside_tail_below > 0;
        if (k1 < n && !grp_open() && (rows_beside ? n - k1 <= knobs().rows_beside_tail_below
                                                  : ((!rows || rows_fused) && n - k1 <= single_tail_below))) {
            // ---- single-stream tail.  Once the trailing matrix is small the look-ahead no longer
            // pays: its chain kernels wait for slots beside the update and every panel costs an
            // inter-queue hop, while one queue runs 4 x (diag + solve) = 124 us plus ONE update of
            // everything right of the panel, all at full speed (measured whole potrf, single queue
            // vs look-ahead: n = 2048: 1.23 vs 1.37 ms, 4096: 2.80 vs 3.04, 6144: 4.84 vs 5.05,
            // 8192: 7.95 vs 7.73; hybrid at N = 8192: 7.86 -> 7.52 ms).  With carried rows the rows'
            // own queue fills the tail either way and the hybrid is neutral (9.69 vs 9.73 ms): not
            // used then.  Panel k0 is factored; the region right of it holds all earlier
            // panels' updates once the bulk queue has drained.
            CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_panel, 0), "hipStreamWaitEvent");
            if (sb != st && ev_bulk_last) CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_bulk_last, 0), "hipStreamWaitEvent");
            // Round 3: the tail is the fused one-queue sweep -- the head update of panel k0 (the next panel's
            // columns, all rows) in a launch of its own, everything after it rides in the chains' launches
            // (fused_sweep).  Carried rows first catch up with every panel up to k0 on their own queues
            // (the factorisation would be starved of compute units by their large updates anyway: it stood
            // still for ~2 ms of the round-2 schedule), then ride along.
            if (rows_beside) {
                // the carried rows keep following on their own queues (panel k0 here, the tail's panels from the sweep)
                rc = rows_after_panel(k0, k1, ev_panel);
                if (rc) return rc;
            } else if (rows) {
                rc = rows_after_panel(k0, k1, ev_panel, true);
                if (rc) return rc;
                if (ev_rows_far) { CIMRGP_HIP_TRY(hipStreamWaitEvent(la->rows, ev_rows_far, 0), "hipStreamWaitEvent"); ev_rows_far = nullptr; }
                hipEvent_t ev_caught = la->ev[ne++];
                CIMRGP_HIP_TRY(hipEventRecord(ev_caught, la->rows), "hipEventRecord");
                CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_caught, 0), "hipStreamWaitEvent");
            }
            {
                const int64_t kn = k1 + ((n - k1 < CIMRGP_NB) ? (n - k1) : CIMRGP_NB);
                rc = gemm_nt_sub<T>(k + k1 * ld + k1, ld, k + k1 * ld + k0, ld, k + k1 * ld + k0, ld,
                                    n - k1, kn - k1, (int)w, false, st);
                if (rc) return rc;
                FarBulk fbk;
                fbk.bulk = sp;                              // the chain's queue of the look-ahead phase is free now
                fbk.ev = &la->ev;
                fbk.ne = &ne;
                fbk.cus = knobs().tail_far_cus;
                fbk.min_rows = knobs().tail_far_min_rows;
                if (rows_beside) {
                    fbk.cus = 0;                            // far updates as riders: the rows' far updates hold the persistent units
                    fbk.on_final = [&](int64_t f0, int64_t f1, hipEvent_t evf) { return rows_after_panel(f0, f1, evf); };
                }
                const bool use_fb = rows_beside || (!rows && fbk.cus >= 8 && knobs().gemm_pers >= 8);
                const bool ride_rows = rows && !rows_beside;
                rc = fused_sweep<T>(k, n, ld, ws, info, ride_rows ? b : (T*)nullptr, ride_rows ? m : 0, ride_rows ? ldb : 0, PotrfBatch(), st,
                                    k1, w, use_fb ? &fbk : nullptr);
                if (rc) return rc;
            }
            tail_done = true;
            break;
        }
        const hipEvent_t ev_final = ev_panel;          // panel k0 is final (recorded on the side stream)
        const bool chained = final_posted;             // ... and posted in la->flag[POSTED_WORD] as number `posted`
        final_posted = false;
        hipEvent_t ev_go = ev_panel;                   // what the bulk stream waits for: panel k0 final ...
        const int64_t wn = (k1 < n) ? ((n - k1 < CIMRGP_NB) ? (n - k1) : CIMRGP_NB) : 0;   // next panel
        const int64_t k2 = k1 + wn;
        // Round 3: head and bulk update of panel k0 as ONE persistent launch on the bulk queue.  Its first
        // tiles are the next panel's columns (the old "head": on the chain's queue it ran on the few compute
        // units the persistent bulk update leaves free -- 144 us for 1 Gflop at N = 8192, the longest link of
        // the chain); they are counted as they are stored and the chain waits for the count through a
        // one-workgroup gate kernel.  The next panel's first diagonal block does not wait: it takes its own
        // 64 x 64 update as its prologue (as before) and the rest of the first 128 x 128 tile along as riders.
        // (not while the carried rows are running: their kernels hold compute units the persistent workgroups
        // of the combined launch -- head tiles included -- would have to wait for: 114 -> 109 posteriors/s)
        const bool rows_running = rows && !rows_fused && !knobs().heads_beside_rows && (n - k1 <= rows_start);
        const int heads = (may_gate && knobs().chain_mode == 0 && !rows_running && w == CIMRGP_NB && wn == CIMRGP_NB && n > k2 && !grp_open() &&
                           group_size(n - k2 - ((n - k2 < CIMRGP_NB) ? (n - k2) : CIMRGP_NB), knobs().far_pair_above) == 1)
                              ? gemm_pers_head_tiles(n - k1, (int)w, (int)sizeof(T)) : 0;
        if (heads > 0) {
            // The persistent launch is ENQUEUED before the gate that waits for its head tiles: a tool that runs one
            // kernel at a time in submission order (rocprofv3 --pmc, HIP_LAUNCH_BLOCKING) then finds the count complete
            // when the gate runs, instead of running the gate first and timing it out.
            hipEvent_t ev_rest_prev = ev_rest;
            flag_expected += heads;
            // bulk queue: everything right of panel k0, the next panel's columns first.  "Panel k0 final" reaches it as
            // an event, or -- when the chain posted it (k_post, below) -- through a gate of its own in front of the
            // update: the gate was enqueued after the chain it waits for, like the chain's gate after its update.
            if (chained) {
                hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sb, (const int*)(la->flag + POSTED_WORD), posted, info);
                CIMRGP_LAUNCH_CHECK("cimrgp_potrf");
            } else {
                CIMRGP_HIP_TRY(hipStreamWaitEvent(sb, ev_final, 0), "hipStreamWaitEvent");
            }
            const double mm = (double)(n - k1);
            hipEvent_t rec = rec_open(sb, mm * (mm + 1.0) * (double)w, (mm * (mm + 1.0) + mm * (double)w) * (double)sizeof(T));
            GemmBatch gb = bulk_gb;
            gb.head_first = 1;
            gb.flag = la->flag;
            rc = gemm_nt_sub<T>(k + k1 * ld + k1, ld, k + k1 * ld + k0, ld, k + k1 * ld + k0, ld, n - k1, n - k1, (int)w, true, sb, gb);
            if (rec) (void)hipEventRecord(rec, sb);
            if (rc) return rc;
            ev_rest = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_rest, sb), "hipEventRecord");
            ev_bulk_last = ev_rest;
            // chain queue: first diagonal block (+ the rest of tile (0, 0) of 128 as riders), gate, the other links
            if (ev_rest_prev) CIMRGP_HIP_TRY(hipStreamWaitEvent(sp, ev_rest_prev, 0), "hipStreamWaitEvent");
            Riders<T> r0 = no_riders<T>();
            {
                RiderJob<T>& jb = r0.job[0];
                jb.c = k + k1 * ld + k1; jb.a = k + k1 * ld + k0; jb.b = k + k1 * ld + k0; jb.ldc = jb.lda = jb.ldb = ld;
                jb.m = 128; jb.n = 128; jb.k = (int)w; jb.lower = 0; jb.tiles_n = 2; jb.first = 0; jb.count = 4; jb.skip00 = 1; jb.rows_job = 0;
                r0.njobs = 1; r0.total = 4;
            }
            hipLaunchKernelGGL((k_diag64q<T>), dim3(1 + r0.total), dim3(Q_NT), 0, sp, k + k1 * ld + k1, ld, (int)SB,
                               (const T*)(k + k1 * ld + k0), (int)w, ws + (k1 / SB) * (SB * SB), info, (int)k1, (int64_t)0, (int64_t)0,
                               (int64_t)0, r0);
            CIMRGP_LAUNCH_CHECK("cimrgp_potrf");
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sp, (const int*)la->flag, flag_expected, info);
            CIMRGP_LAUNCH_CHECK("cimrgp_potrf");
            rc = factor_panel<T>(k, n, ld, ws, info, k1, wn, sp, false, true);
            if (rc) return rc;
            if (knobs().post_final) {
                hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, sp, la->flag + POSTED_WORD, ++posted);
                CIMRGP_LAUNCH_CHECK("cimrgp_potrf");
                final_posted = true;
            }
            hipEvent_t ev_next = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_next, sp), "hipEventRecord");
            ev_panel = ev_next;
            rc = rows_after_panel(k0, k1, ev_final);
            if (rc) return rc;
            continue;
        }
        if (k1 < n) {
            // chain: head (columns of the next panel, all rows below), then the next panel
            if (ev_rest) CIMRGP_HIP_TRY(hipStreamWaitEvent(sp, ev_rest, 0), "hipStreamWaitEvent");
            // The next panel's FIRST diagonal block does not wait for the head: one workgroup takes its
            // update by panel k0 as the left-looking prologue of the diagonal kernel (K = 256) and factors it
            // -- launched while the machine is still empty (the bulk update of panel k0 starts at the same
            // moment on the other queue), it does not queue for a slot behind the update's first generation
            // of workgroups (55 us at N = 8192); the head then leaves that 64 x 64 tile alone.  Whole potrf,
            // without / with: N = 8192 6.54 / 6.47 ms, N = 16384 30.05 / 29.75; with carried rows it costs
            // (8.53 -> 8.82 ms with 2050 rows: the rows' queues then see an even busier chain), so not there.
            const bool head0 = knobs().fused_head0 && !rows && knobs().chain_mode != 1 && w == CIMRGP_NB &&
                               gemm_uses_tile64(n - k1, wn, false);
            if (head0) {
                const int sw0 = (int)((wn < SB) ? wn : SB);
                hipLaunchKernelGGL((k_diag64q<T>), dim3(1), dim3(Q_NT), 0, sp, k + k1 * ld + k1, ld, sw0,
                                   (const T*)(k + k1 * ld + k0), (int)w, ws + (k1 / SB) * (SB * SB), info, (int)k1, (int64_t)0, (int64_t)0,
                                   (int64_t)0, no_riders<T>());
                CIMRGP_LAUNCH_CHECK("cimrgp_potrf");
            }
            GemmBatch ghead; ghead.skip_first = head0 ? 1 : 0;
            rc = gemm_nt_sub<T>(k + k1 * ld + k1, ld, k + k1 * ld + k0, ld, k + k1 * ld + k0, ld,
                                n - k1, wn, (int)w, false, sp, ghead);
            if (rc) return rc;
            // (Round 1 made the bulk update wait for the head while the trailing matrix was large: started
            // together, the bulk update took the compute units from the head and stretched it five-fold,
            // and the nine-wave diagonal kernel behind it waited for a whole compute unit.  With the
            // four-wave chain kernels the order no longer pays -- whole potrf, head first above 4608 rows
            // against never: N = 8192 6.72 against 6.56 ms, N = 16384 30.39 against 30.08 -- the switch
            // stays for measurements: CIMRGP_HEAD_FIRST = rows.)
            const int64_t head_first_above = knobs().head_first_above;
            if (!rows && n - k1 > head_first_above) {
                ev_go = la->ev[ne++];
                CIMRGP_HIP_TRY(hipEventRecord(ev_go, sp), "hipEventRecord");
            }
            rc = factor_panel<T>(k, n, ld, ws, info, k1, wn, sp, false, head0);
            if (rc) return rc;
            ev_panel = la->ev[ne++];
            CIMRGP_HIP_TRY(hipEventRecord(ev_panel, sp), "hipEventRecord");
        }
        if (k1 < n) {
            // bulk: lower SYRK beyond the next panel, concurrently with the chain.  While that far
            // region is big, it is updated once per GROUP of 2-4 panels with K = 256 x group size
            // (adjacent panels are adjacent columns, the same kernel applies): fewer passes over C.
            ev_rest = nullptr;
            bool split_far = false;                // the split far update records its own events
            if (n > k2) {
                CIMRGP_HIP_TRY(hipStreamWaitEvent(sb, ev_go, 0), "hipStreamWaitEvent");
                const int64_t wnn = (n - k2 < CIMRGP_NB) ? (n - k2) : CIMRGP_NB;   // panel after next
                const int64_t k3 = k2 + wnn;
                if (grp.g0 < 0) {
                    const int64_t far_pair_above = knobs().far_pair_above;
                    const int g = group_size(n - k3, far_pair_above);
                    if (g > 1) { grp.g0 = k0; grp.left = g; }
                }
                if (grp.g0 >= 0) {
                    // a panel of a group: the columns of the panel after next with all the group's
                    // panels so far (all the chain's next head update needs -- it may start as soon
                    // as they are done) ...
                    const int kk = (int)(k1 - grp.g0);
                    rc = gemm_nt_sub<T>(k + k2 * ld + k2, ld, k + k2 * ld + grp.g0, ld, k + k2 * ld + grp.g0, ld,
                                        n - k2, wnn, kk, false, sb);
                    if (rc) return rc;
                    ev_rest = la->ev[ne++];
                    CIMRGP_HIP_TRY(hipEventRecord(ev_rest, sb), "hipEventRecord");
                    ev_bulk_last = ev_rest;
                    split_far = true;
                    if (--grp.left == 0 || n <= k3) {
                        // ... and, closing the group, the big remainder with K = 256 x group size,
                        // which overlaps the chain's next panels
                        if (n > k3) {
                            const double mm = (double)(n - k3);
                            hipEvent_t rec = rec_open(sb, mm * (mm + 1.0) * (double)kk, (mm * (mm + 1.0) + mm * (double)kk) * (double)sizeof(T));
                            rc = gemm_nt_sub<T>(k + k3 * ld + k3, ld, k + k3 * ld + grp.g0, ld, k + k3 * ld + grp.g0, ld,
                                                n - k3, n - k3, kk, true, sb, bulk_gb);
                            if (rec) (void)hipEventRecord(rec, sb);
                            if (rc) return rc;
                            ev_bulk_last = la->ev[ne++];
                            CIMRGP_HIP_TRY(hipEventRecord(ev_bulk_last, sb), "hipEventRecord");
                        }
                        grp = PanelGroup();
                    }
                } else {
                    const double mm = (double)(n - k2);
                    hipEvent_t rec = rec_open(sb, mm * (mm + 1.0) * (double)w, (mm * (mm + 1.0) + mm * (double)w) * (double)sizeof(T));
                    rc = gemm_nt_sub<T>(k + k2 * ld + k2, ld, k + k2 * ld + k0, ld, k + k2 * ld + k0, ld,
                                        n - k2, n - k2, (int)w, true, sb, bulk_gb);
                    if (rec) (void)hipEventRecord(rec, sb);
                    if (rc) return rc;
                }
                if (!split_far) {
                    ev_rest = la->ev[ne++];
                    CIMRGP_HIP_TRY(hipEventRecord(ev_rest, sb), "hipEventRecord");
                    ev_bulk_last = ev_rest;
                }
            }
        }
        rc = rows_after_panel(k0, k1, ev_final);
        if (rc) return rc;
    }
    if (rows) {
        if (ev_rows_far) CIMRGP_HIP_TRY(hipStreamWaitEvent(la->rows, ev_rows_far, 0), "hipStreamWaitEvent");
        hipEvent_t ev_rows_done = la->ev[ne++];
        CIMRGP_HIP_TRY(hipEventRecord(ev_rows_done, la->rows), "hipEventRecord");
        CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_rows_done, 0), "hipStreamWaitEvent");
    }
    // join: the last panel (side stream); every bulk update precedes it through the chain's waits
    if (!tail_done) {
        CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_panel, 0), "hipStreamWaitEvent");
        if (sb != st && ev_bulk_last) CIMRGP_HIP_TRY(hipStreamWaitEvent(st, ev_bulk_last, 0), "hipStreamWaitEvent");
    }
    return build_invT<T>(k, n, ld, ws, st);
    };   // body
    const int rc_body = body();
    if (rc_body == 0) {
        // (a failed record only costs the next call its early panels)
        if (la->last_done == nullptr && hipEventCreateWithFlags(&la->last_done, hipEventDisableTiming) != hipSuccess) la->last_done = nullptr;
        if (la->last_done != nullptr) (void)hipEventRecord(la->last_done, st);
    }
    if (rc_body != 0) {
        // failed enqueue: the caller's stream waits for everything that did get queued (errors of the join itself
        // cannot improve on the one being reported)
        size_t je = la->ev.size() - 4;
        for (hipStream_t q : {la->side, la->bulk, la->rows, la->rows_far}) {
            if (q == nullptr) continue;
            hipEvent_t ev = la->ev[je++];
            if (hipEventRecord(ev, q) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
        }
    }
    return rc_body;
}

// `bt.count` equal-sized factorisations (the blocks of one layer) in the SAME launches: every
// kernel of the one-queue sweep runs with grid.y = count, so a layer of many small blocks costs the
// host one block's worth of launches and the device sees all the blocks' panel chains side by side.
template <typename T>
int potrf_batched_run(T* k, int64_t n, int64_t ld, T* ws, int32_t* info, T* b, int64_t m, int64_t ldb, PotrfBatch bt,
                      hipStream_t st)
{
    if (bt.count < 1 || bt.count >= 65536) return fail("cimrgp_potrf_batched", "batch count out of range");
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)bt.count, st);
    if (e != hipSuccess) return check_hip(e, "cimrgp_potrf_batched", "hipMemsetAsync(info)");
    // A big batch fills the machine with its panel solves already: the updates then keep launches of their own
    // (128-tiles, more efficient than 64-tile riders); riders pay where the chain's launches leave units idle.
    const bool ride = (int64_t)bt.count * (n / TR) <= knobs().fused_max_chain_wgs;
    if (!ride && bt.count >= knobs().batch_halves_min && n > CIMRGP_NB) {
        // Two halves of the batch on two queues: nothing orders them against each other, so the latency-bound
        // panel chains of one half run beside the MFMA-bound updates of the other (one queue alternates
        // between the two kinds of work and leaves the matrix cores idle for a third of a 128 x 2048 layer).
        LookAhead* la = acquire_ctx(st);
        if (la != nullptr) {
            std::lock_guard<std::mutex> guard(la->enqueue);
            if (!grow_events(la, 2)) return fail("cimrgp_potrf_batched", "hipEventCreate failed");
            PotrfBatch h0 = bt, h1 = bt;
            h0.count = bt.count / 2;
            h1.count = bt.count - h0.count;
            const int64_t c = h0.count;
            CIMRGP_HIP_TRY(hipEventRecord(la->ev[0], st), "hipEventRecord");
            CIMRGP_HIP_TRY(hipStreamWaitEvent(la->side, la->ev[0], 0), "hipStreamWaitEvent");
            int rc = panel_sweep<T, true>(k, n, ld, ws, info, b, m, ldb, st, h0);
            if (!rc) rc = build_invT<T>(k, n, ld, ws, st, h0);
            if (!rc) rc = panel_sweep<T, true>(k + c * bt.sk, n, ld, ws + c * bt.sws, info + c, b ? b + c * bt.sb : b, m, ldb, la->side, h1);
            if (!rc) rc = build_invT<T>(k + c * bt.sk, n, ld, ws + c * bt.sws, la->side, h1);
            // (joined even after a failed enqueue: the caller's stream must not run ahead of the side queue)
            CIMRGP_HIP_TRY(hipEventRecord(la->ev[1], la->side), "hipEventRecord");
            CIMRGP_HIP_TRY(hipStreamWaitEvent(st, la->ev[1], 0), "hipStreamWaitEvent");
            return rc;
        }
    }
    int rc = ride ? fused_sweep<T>(k, n, ld, ws, info, b, m, ldb, bt, st) : panel_sweep<T, true>(k, n, ld, ws, info, b, m, ldb, st, bt);
    return rc ? rc : build_invT<T>(k, n, ld, ws, st, bt);
}

template <typename T>
int solve_rows_run(const T* l, int64_t n, int64_t ld, const T* ws, T* b, int64_t m, int64_t ldb, hipStream_t st, PotrfBatch bt)
{
    if (bt.count < 1 || bt.count >= 65536) return fail("cimrgp_trsm_rows", "batch count out of range");
    return panel_sweep<T, false>(const_cast<T*>(l), n, ld, const_cast<T*>(ws), nullptr, b, m, ldb, st, bt);
}

template int potrf_run<double>(double*, int64_t, int64_t, double*, int32_t*, double*, int64_t, int64_t, hipStream_t, hipStream_t);
template int potrf_run<float>(float*, int64_t, int64_t, float*, int32_t*, float*, int64_t, int64_t, hipStream_t, hipStream_t);
template int potrf_batched_run<double>(double*, int64_t, int64_t, double*, int32_t*, double*, int64_t, int64_t, PotrfBatch, hipStream_t);
template int potrf_batched_run<float>(float*, int64_t, int64_t, float*, int32_t*, float*, int64_t, int64_t, PotrfBatch, hipStream_t);
template int solve_rows_run<double>(const double*, int64_t, int64_t, const double*, double*, int64_t, int64_t, hipStream_t, PotrfBatch);
template int solve_rows_run<float>(const float*, int64_t, int64_t, const float*, float*, int64_t, int64_t, hipStream_t, PotrfBatch);

}  // namespace cimrgp
