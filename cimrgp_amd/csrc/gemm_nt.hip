// Trailing / panel update of the blocked Cholesky and of the row-wise TRSM:
//     C[M x N] -= A[M x K] * B[N x K]^T          (row-major, K contiguous)
// on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32).
//
// Geometry (identical in bytes for f32 and f64):
//   workgroup = 256 threads = 4 waves in a 2 x 2 grid, one 128 x 128 tile of C;
//   each wave owns 64 x 64 = 4 x 4 MFMA tiles (f64: 128 accumulator VGPRs);
//   K is walked in stages of 128 bytes per row (16 doubles / 32 floats);
//   per stage both operand tiles (128 rows x 128 B) are fetched with 16-byte
//   global loads into registers while the previous stage is being multiplied,
//   then written to the other LDS buffer (row stride 144 B: the 8-byte
//   fragment reads of a 32-lane half hit 32 distinct 8-byte bank pairs);
//   one barrier per stage.  2 workgroups per CU (72 KiB LDS each) so that the
//   C read-modify-write of one overlaps the K loop of the other.
// Algorithmic work: 2 M N K flop (M N K for the lower-triangular SYRK form),
// bounded by the MFMA pipe.
#include "common.hpp"
#include "gemm_tile.hpp"

namespace cimrgp {

namespace {

// One workgroup = one tile.  Whether the tile needs bounds handling is decided per workgroup,
// so a ragged matrix (e.g. 2050 carried rows) pays for it only in its last row/column of tiles.
template <typename T, bool LOWER, int W>
__global__ __launch_bounds__(256, 2)
void k_gemm_nt_sub(T* __restrict__ C, int64_t ldc,
                   const T* __restrict__ A, int64_t lda,
                   const T* __restrict__ B, int64_t ldb,
                   int M, int N, int K, int tiles_n, int64_t sc, int64_t sa, int64_t sb, int skip_first)
{
    if (skip_first && blockIdx.x == 0) return;        // tile (0, 0) belongs to the chain (see GemmBatch)
    constexpr int GT = 32 * W;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * GT * LROW];
    // batch of independent products (blocks of one layer): blockIdx.y selects the problem
    C += (int64_t)blockIdx.y * sc;
    A += (int64_t)blockIdx.y * sa;
    B += (int64_t)blockIdx.y * sb;
    int ti, tj;
    if (LOWER) {
        const int id = blockIdx.x;
        ti = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
        while (ti * (ti + 1) / 2 > id) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
        tj = id - ti * (ti + 1) / 2;
    } else {
        ti = blockIdx.x / tiles_n;
        tj = blockIdx.x - ti * tiles_n;
    }
    const bool interior = (ti + 1) * GT <= M && (tj + 1) * GT <= N && (K % (KT_BYTES / (int)sizeof(T))) == 0;
    if (interior) gemm_tile<T, LOWER, false, W>(smem, C, ldc, A, lda, B, ldb, M, N, K, ti, tj);
    else          gemm_tile<T, LOWER, true,  W>(smem, C, ldc, A, lda, B, ldb, M, N, K, ti, tj);
}


// ---------------------------------------------------------------------------
// Persistent form of the 128 x 128 update (round 3): ONE workgroup of 8 waves per compute unit (two
// per SIMD, 64 x 32 of C each) walks a list of C tiles with TWO accumulator sets per wave.  While
// the matrix cores run tile t's K loop on one set, the other set is streamed: the finished tile
// t-1 is stored from it and tile t+1's C is loaded into it, one 16 x 16 sub-tile at a time spread
// over the K stages -- so the C traffic (256 KB per 8.4 Mflop tile at K = 256) flows evenly under
// the K loop instead of arriving as a burst of 512 simultaneous prologues / epilogues per generation
// of workgroups, and no wave ever waits for it.  The operand stream (128 bytes of K per row and
// stage, as in gemm_tile) runs straight across tile boundaries: registers hold stage g+1 while
// stage g is multiplied, the LDS write of stage g+1 sits at the top of stage g, fragments are read
// one k-step ahead (across stages and tiles), so neither a global nor an LDS round trip is exposed
// in steady state.  Results are bit-identical to gemm_tile's (same K order per element).
// Interior tiles only: M, N multiples of 128, K = 16 stages.  A diagonal tile of a lower-triangular update is
// stored WHOLE: the elements above the diagonal inside the 128 x 128 diagonal blocks are overwritten with
// C - A B^T of whatever they held (include/cimrgp.h: the strict upper triangle holds junk after potrf).
// A launch of G <= 256 workgroups occupies G compute units (512 threads x up to 256 registers: one
// workgroup per unit) and leaves the others to whatever else is running -- the factorisation's
// latency-bound panel chain: a spatial split of the machine without CU masks.
// ---------------------------------------------------------------------------
// Tile order of the persistent kernel.  Tiles are numbered strip by strip (a strip = PERS_STRIP rows of
// tiles), column by column inside a strip, so that a run of ~32 consecutive numbers is a patch of
// PERS_STRIP rows x ~8 columns of tiles: 4 row blocks of A and ~8 of B serve 32 tiles.  Workgroup w
// belongs (for speed only: observed placement, never correctness) to XCD w % 8 and is its workgroup
// number w / 8; in round r the workgroups of XCD x take the run of numbers (8 r + x) * per_xcd ...,
// so the tiles that share operand blocks are multiplied at the same time on compute units that share
// an L2.  (With tiles dealt out in plain row-major order every B block was used by one tile per XCD and
// round: TCC hit rate 45 %, 500 MB of the 1 GB of operand reads per launch at M = 7936 came from beyond L2.)
constexpr int PERS_STRIP = 4;

// q / d for 0 <= q < 2^22, 1 <= d: a float quotient with one fix-up either way (scalar code has no integer divide, and
// the generic sequence is ~30 instructions that all eight waves of a workgroup push through the ONE scalar unit)
static __device__ __forceinline__ int pers_div(int q, int d)
{
    int r = (int)((float)q / (float)d);
    r -= (r * d > q);
    r += ((r + 1) * d <= q);
    return r;
}

// Closed form, straight-line (round 5: the strip-by-strip search loop of rounds 3-4 was ~350 scalar instructions per
// tile and wave, and eight waves share one scalar unit: 1800 of a tile's 72 000 clocks, tools/lab/pers_stamps.py).
// LOWER: full strips hold 16 s + 10 tiles, so strip s starts at tile 8 s^2 + 2 s.
template <bool LOWER>
static __device__ __forceinline__ void pers_tile_decode(int q, int tiles_m, int tiles_n, int& ti, int& tj)
{
    if (LOWER) {
        // strip s holds rows [S s, min(S s + S, tiles_m)); column c of a strip holds its rows >= c
        static_assert(PERS_STRIP == 4, "closed form below");
        const int sfull = tiles_m >> 2;
        int s, h;
        if (q < 8 * sfull * sfull + 2 * sfull) {
            s = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.125f);
            s -= (8 * s * s + 2 * s > q);
            s += (8 * (s + 1) * (s + 1) + 2 * (s + 1) <= q);
            h = 4;
        } else {
            s = sfull;
            h = tiles_m - 4 * sfull;
        }
        const int r0 = 4 * s;
        const int off = q - (8 * s * s + 2 * s);
        if (off < h * r0) {
            tj = (h == 4) ? (off >> 2) : (h == 2) ? (off >> 1) : (h == 1) ? off : (int)(((unsigned)off * 43691u) >> 17);   // off / 3: off < 3 * 4 * 512
            ti = r0 + off - tj * h;
        } else {
            int o2 = off - h * r0, c = 0;            // inside the h x h triangle on the diagonal, column by column (h - c tiles each)
            if (o2 >= h) {
                o2 -= h; c = 1;
                if (o2 >= h - 1) {
                    o2 -= h - 1; c = 2;
                    if (o2 >= h - 2) { o2 -= h - 2; c = 3; }
                }
            }
            tj = r0 + c;
            ti = r0 + c + o2;
        }
    } else {
        const int per_strip = PERS_STRIP * tiles_n;
        const int s = pers_div(q, per_strip);
        const int r0 = PERS_STRIP * s;
        const int h = min(PERS_STRIP, tiles_m - r0);
        const int off = q - s * per_strip;
        tj = (h == 4) ? (off >> 2) : (h == 2) ? (off >> 1) : (h == 1) ? off : pers_div(off, 3);
        ti = r0 + off - tj * h;
    }
}

// "Head first" order of a lower-triangular update (the factorisation's look-ahead): the two leftmost tile
// columns -- the next panel's 256 columns, which the panel chain waits for -- come first, (ti, 0) for
// ti = 1 ..., then (ti, 1); tile (0, 0) is NOT part of the launch (the chain's first diagonal block owns its
// top-left 64 x 64 and takes the rest of that tile along); the remaining triangle follows in strip order.
// heads = 2 tiles_m - 2 tiles come first.
static __device__ __forceinline__ void pers_tile_decode_head_first(int q, int tiles_m, int& ti, int& tj)
{
    if (q < tiles_m - 1) { ti = q + 1; tj = 0; return; }
    q -= tiles_m - 1;
    if (q < tiles_m - 1) { ti = q + 1; tj = 1; return; }
    q -= tiles_m - 1;
    pers_tile_decode<true>(q, tiles_m - 2, 0, ti, tj);
    ti += 2;
    tj += 2;
}

// the i-th tile (i = 0, 1, ...) of workgroup w of `grid`: -1 when there is none
static __device__ __forceinline__ int pers_tile_number(int w, int i, int grid, int ntiles)
{
    const int per_xcd = grid >> 3;               // grid is a multiple of 8 (launcher)
    const int x = w & 7, j = w >> 3;
    const int q = (i * 8 + x) * per_xcd + j;
    return q < ntiles ? q : -1;
}

// The C stream of the persistent kernel is cut into 16 "events" per tile and wave, one per K stage
// (K = 16 stages of 128 bytes: 256 doubles / 512 floats): event e covers accumulator tile
// (mi, ni) = (e >> 2, (e >> 1) & 1), values r = 2 (e & 1) + {0, 1} (two rows of 16 lanes x 8 bytes = two
// 128-byte lines per row tile).  An event stores the finished values of the previous tile and requests the
// next tile's straight into the registers just stored from (round 5; rounds 3-4 went through two staging
// registers and four moves per event).
// The 16 stages of a tile are fully unrolled: every register index is a compile-time constant and the
// whole pass is straight-line code, so the compiler counts the memory instructions exactly -- the
// wait for stage g+1's operands at the top of stage g leaves the 4 C accesses issued behind them in
// flight.  (With the events inside a switch over the stage number hipcc waited for vmcnt(0) at the top
// of every stage, and every stage then took a full HBM round trip under load: 40 against 45 TF/s for
// the tile-per-workgroup kernel.)
// Timing-only builds (tools/exp_variants.sh; 1-6 give WRONG results, never the shipped library):
// -DPERS_EXP=1 no second barrier, 2 no barriers at all, 3 no C events, 4 C loads only, 5 C stores only,
// 6 C loads always from the tile's first rows (L2 hits), 7 / 8 the C event issued in the middle / at the
// end of the stage instead of its head (correct results).  Measured at M=7936, K=256 (DESIGN.md section 4,
// round 3 (1); profiles/r03_pers_variants.txt): without the C events the pass is ~10 % shorter, with only the
// loads or only the stores ~4 %, without the barriers 2-4 %, and moving the event within the stage changes
// nothing: the cost of streaming C is the memory pipeline's share of the issue slots, not where in the stage
// the accesses sit and not the barriers.
#ifndef PERS_EXP
#define PERS_EXP 0
#endif
#ifndef PERS_SYNC_FLAGS
#define PERS_SYNC_FLAGS 1
#endif
// Round 5: the two barriers of a stage are fences + s_barrier (common.hpp), no longer inline assembly whose "memory"
// clobber the compiler does not apply to a __shared__ array whose address never escapes (the operand ring here).
#if PERS_EXP == 1
#define PERS_BARRIER_A() lds_barrier()
#define PERS_BARRIER_B() do { } while (0)
#elif PERS_EXP == 2
#define PERS_BARRIER_A() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define PERS_BARRIER_B() do { } while (0)
#else
#define PERS_BARRIER_A() lds_barrier()              /* stage kt+1's LDS writes are done: s_waitcnt lgkmcnt(0); s_barrier */
#define PERS_BARRIER_B() lds_barrier_nowait()       /* every wave has consumed stage kt's fragments: s_barrier alone */
#endif
#ifdef PERS_STAMPS          /* diagnostic build (tools/lab/pers_stamps.py): shader-clock stamps of wave 0 at every stage's first barrier, first 8 tiles of every workgroup */
__device__ long long g_pers_stamp[256][8 * 16 + 4];
#define PERS_STAMP(slot_) do { if (tid == 0 && it < 8) g_pers_stamp[wg & 255][slot_] = __builtin_amdgcn_s_memtime(); } while (0)
#define PERS_STAMP_EXIT() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) { g_pers_stamp[wg & 255][129] = __builtin_amdgcn_s_memtime(); g_pers_stamp[wg & 255][131] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PERS_STAMP(slot_) do { } while (0)
#define PERS_STAMP_EXIT() do { } while (0)
#endif
// A tile of C as a raw buffer: base = the tile's first element (wave-uniform), no stride, the largest extent (a lane's
// offsets stay below 2^31: the launcher checks 128 rows of C against it).
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// the same with NO extent: every access through it is out of range -- stores are dropped, loads return zero
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc_null(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0, 0x00020000);
}
typedef unsigned int cbuf_u2 __attribute__((ext_vector_type(2)));
template <typename T> static __device__ __forceinline__ T cbuf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ double cbuf_load<double>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const cbuf_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
template <> __device__ __forceinline__ float cbuf_load<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <typename T> static __device__ __forceinline__ void cbuf_store(T v, __amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ void cbuf_store<double>(double v, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const cbuf_u2 u = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(u, r, voff, soff, 0);
}
template <> __device__ __forceinline__ void cbuf_store<float>(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

constexpr int PERS_THREADS = 512;
constexpr int PERS_STAGES = 16;

// NKT = K stages of 128 bytes per pass: 16 for K = 256 doubles (or 512 floats), 8 for K = 256 floats (round 4: the
// FP32 form of one panel -- a stage is the same 128 bytes and the same matrix-core time in both precisions, so an
// FP32 pass is half as long and carries the tile's sixteen C events two per stage).
// MULTI: K = nkc chunks of NKT stages (a runtime loop around the unrolled stages); false: one chunk, no loop -- the
// loop costs the one-panel form 1-2 % (profiles/r05_chunk_ab.txt), so the K = 256 launches keep the form without it.
template <typename T, bool LOWER, int NKT, bool MULTI = false>
__global__ __launch_bounds__(PERS_THREADS)
void k_gemm_nt_pers(T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda,
                    const T* __restrict__ B, int64_t ldb, int tiles_m, int tiles_n, int ntiles, int heads, int* __restrict__ flag,
                    int head_direct, int nkc)
{
    static_assert(NKT == 16 || NKT == 8, "sixteen C events per tile: one or two per K stage");
    constexpr int EVS = 16 / NKT;                            // C events per stage
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    constexpr int BKE = KT_BYTES / (int)sizeof(T);
    constexpr int GT = 128;
    constexpr int OP_BYTES = GT * LROW;
    constexpr int RS = (sizeof(T) == 8) ? 4 : 1;             // crow(lane, r) = crow(lane, 0) + RS r
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * OP_BYTES];       // 2 stages x 2 operands

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // 2 x 4 waves: 64 rows x 32 columns each
    const int sc = tid & 7, sr = tid >> 3;                   // staging: 8 threads per 128-byte row segment, 64 rows per pass
    const int grid = (int)gridDim.x, wg = (int)blockIdx.x;

    int a_off_e[2], b_off_e[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a_off_e[p] = (sr + 64 * p) * (int)lda + sc * X::EPC;
        b_off_e[p] = (sr + 64 * p) * (int)ldb + sc * X::EPC;
    }
    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned a_frag = (unsigned)((wr * 64 + frow) * LROW + fslot * 8);
    const unsigned b_frag = (unsigned)((wc * 32 + frow) * LROW + fslot * 8);
    // this lane's first element of its wave's 64 x 32 share, inside a tile
    const int lrow0 = wr * 64 + X::crow(lane, 0), lcol0 = wc * 32 + (lane & 15);
    const int coff = lrow0 * (int)ldc + lcol0;
    // this lane's byte offset from a tile's first element (constant), and the bytes of a row of C
    const int cbyte0 = coff * (int)sizeof(T);
    const int row_bytes = (int)ldc * (int)sizeof(T);

    // heads > 0: head-first order (above); every one of the first `heads` tiles, once STORED, adds 1 to *flag
    auto decode = [&](int q, int& i_, int& j_) {
        if (LOWER && heads > 0) pers_tile_decode_head_first(q, tiles_m, i_, j_);
        else pers_tile_decode<LOWER>(q, tiles_m, tiles_n, i_, j_);
    };
    auto signal_stored = [&]() {
        // every wave's stores have left it, then one release for the workgroup and the count
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
#ifdef PERS_STAMPS
    if (tid == 0) { g_pers_stamp[wg & 255][128] = __builtin_amdgcn_s_memtime(); g_pers_stamp[wg & 255][130] = __builtin_amdgcn_s_memrealtime(); }
#endif
#if PERS_SYNC_FLAGS
    // Stage hand-offs by FLAGS in LDS instead of s_barrier (round 5).  A wave posts its stage number in its own word
    // right behind the LDS accesses the others wait for (the LDS executes a wave's accesses in issue order, so when the
    // post is visible those accesses have been performed) and polls the eight words where it used to stop at a barrier;
    // the poll's read is issued one multiply group ahead of its test.  Measured with the stamped builds
    // (tools/lab/pers_stamps.py): the two barriers of a stage cost ~250 of its ~4530 clocks -- a wave that arrives at
    // s_barrier LAST finds its SIMD partner already waiting, and the matrix core then idles for the barrier's round trip.
    __shared__ int flag_a[8], flag_b[8];           // a: my share of the next stage is written; b: my last fragment reads of this stage are issued
    if (tid < 8) { flag_a[tid] = -1; flag_b[tid] = -1; }
    int gstage = 0;                                // stages since the launch (wave-uniform)
    // (relaxed workgroup-scope atomics on the __shared__ words themselves: plain ds_write_b32 / ds_read_b32, no wait of their
    //  own; the compiler-only fences keep the LDS accesses around them in program order, the LDS keeps them in issue order)
#define PERS_POST(flag_)  do { __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                  \
                               __hip_atomic_store(&flag_[wave], gstage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  \
                               __atomic_signal_fence(__ATOMIC_SEQ_CST); } while (0)
#define PERS_PEEK(flag_)  __hip_atomic_load(&flag_[lane & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define PERS_WAIT(flag_, seen_, g_)                                                                              \
        do {                                                                                                     \
            int v_ = (seen_);                                                                                    \
            while (__builtin_amdgcn_ballot_w64(v_ - (g_) < 0) != 0ull) v_ = PERS_PEEK(flag_);                    \
            __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                             \
        } while (0)
#endif
    int it = 0;                                   // this workgroup's tile counter
    int t = pers_tile_number(wg, it, grid, ntiles);
    if (t < 0) return;
    int ti, tj;
    decode(t, ti, tj);
    // the NEXT tile's number and coordinates, computed one pass ahead in the shadow of the multiplies (scalar code
    // between two matrix-core instructions costs nothing; at the top of a pass it stops all eight waves)
    int t_nx = pers_tile_number(wg, 1, grid, ntiles), ti_nx = ti, tj_nx = tj;
    if (t_nx >= 0) decode(t_nx, ti_nx, tj_nx);
    bool prev_head = false;                       // the tile being stored during this pass is one the chain waits for
    bool prv_stored = true;                       // nothing to store for the "previous tile" of this pass: the first pass, or a head tile stored directly already (head_direct)
    T* c_cur = C + (int64_t)ti * GT * ldc + (int64_t)tj * GT;
    const T* a_cur = A + (int64_t)ti * GT * lda;
    const T* b_cur = B + (int64_t)tj * GT * ldb;
    T* c_prv = c_cur;
    // first-class vectors: arrays of HIP's uint4 struct filled from global memory can stay in scratch
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u ra[2], rb[2];
    const v4u sign4 = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    uint2 fa[2][4], fb[2][2];                                // two fragment sets: k-step s+1 is read while s is multiplied
    acc_t acc0[4][2], acc1[4][2];

#define PERS_GLOAD(ap_, bp_, kt_)                                                  \
    {                                                                              \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                            \
            ra[p] = *reinterpret_cast<const v4u*>((ap_) + (a_off_e[p] + (kt_) * BKE)); \
            rb[p] = *reinterpret_cast<const v4u*>((bp_) + (b_off_e[p] + (kt_) * BKE)); \
        }                                                                          \
    }
#define PERS_SWRITE(buf_)                                                          \
    {                                                                              \
        unsigned char* as_ = smem + (buf_) * 2 * OP_BYTES;                         \
        unsigned char* bs_ = as_ + OP_BYTES;                                       \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                            \
            /* FP32: the sign of C - A B^T goes into the staged A (4 flips per 16 bytes here against 8 per k-step at */ \
            /* the fragments); FP64: the multiply negates its operand itself (Mx<T>::mma_neg)                      */ \
            *reinterpret_cast<v4u*>(as_ + (sr + 64 * p) * LROW + sc * 16) = (sizeof(T) == 4) ? (ra[p] ^ sign4) : ra[p]; \
            *reinterpret_cast<v4u*>(bs_ + (sr + 64 * p) * LROW + sc * 16) = rb[p]; \
        }                                                                          \
    }
#define PERS_FRAGS_NF(set_, buf_, s_)                                              \
    {                                                                              \
        const unsigned char* as_ = smem + (buf_) * 2 * OP_BYTES;                   \
        const unsigned char* bs_ = as_ + OP_BYTES;                                 \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                           \
            fa[set_][mi] = *reinterpret_cast<const uint2*>(as_ + a_frag + mi * 16 * LROW + (s_) * 32); \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                           \
            fb[set_][ni] = *reinterpret_cast<const uint2*>(bs_ + b_frag + ni * 16 * LROW + (s_) * 32); \
    }
#define PERS_FRAGS(set_, buf_, s_)  { PERS_FRAGS_NF(set_, buf_, s_) __builtin_amdgcn_sched_barrier(0); }   /* the reads stay AHEAD of the multiplies that follow */
#define PERS_MMA_NF(cur_, set_)                                                    \
    {                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                         \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                       \
                cur_[mi][ni] = (sizeof(T) == 4) ? X::mma(fa[set_][mi], fb[set_][ni], cur_[mi][ni])            \
                                                : X::mma_neg(fa[set_][mi], fb[set_][ni], cur_[mi][ni]);       \
        }                                                                          \
    }
#define PERS_MMA(cur_, set_)  { PERS_MMA_NF(cur_, set_) __builtin_amdgcn_sched_barrier(0); }
    // one multiply, then up to six of everything else (LDS, global memory, vector and scalar ALU), eight times:
    // the stage's bookkeeping -- LDS write of the next stage, operand and C requests, address arithmetic --
    // is issued in the shadow of the first k-step's multiplies instead of ahead of them (both waves of a SIMD
    // do this part at the same time: issued up front it left the matrix pipe idle for ~10 % of a stage)
#define PERS_INTERLEAVE()                                                          \
    {                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x096, 6, 0);                     \
        }                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                         \
    }

    // ---- prologue: C of the first tile, operand stages 0 (to LDS) and 1 (in registers)
    PERS_GLOAD(a_cur, b_cur, 0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc0[mi][ni][r] = (c_cur + ((int64_t)(mi * 16 + RS * r) * ldc + ni * 16))[coff];
    // The first pass has no finished tile to store, and its events store all the same (no branch around a memory
    // instruction): it addresses the "previous tile" through a descriptor without extent (prv_stored starts true), so
    // those stores drop.  (Rounds 3-4 copied the first tile's C into the other set and stored it back.)
    // The barrier below is LDS-only (round 5): the first tile's C stays in flight across it and the first multiplies wait
    // for their own accumulators only -- __syncthreads() here held every wave until all 128 KB of C had arrived.
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc1[mi][ni] = acc_zero<T>();
    PERS_SWRITE(0);
    PERS_GLOAD(a_cur, b_cur, 1);
    lds_barrier();
    PERS_FRAGS(0, 0, 0);

    // One tile: `cur` holds its C (requested during the previous pass), `oth` the finished previous
    // tile, which is stored and replaced by the next tile's C as the K loop proceeds.  The ring position
    // of a stage is its number's parity (16 stages per tile).
#define PERS_EVENT_ONE(oth_, e_)                                                                         \
        if (PERS_EXP != 3) {                    /* event e_ (compile-time after unrolling) */                   \
                const int mi_ = (e_) >> 2, ne_ = ((e_) >> 1) & 1, r0_ = 2 * ((e_) & 1);                         \
                /* buffer addressing (round 5): the tile's descriptor (SGPRs, rebuilt per tile by scalar code) + this */ \
                /* lane's constant 32-bit byte offset + a wave-uniform scalar offset for the event: NO vector     */ \
                /* instruction per access (the 64-bit flat address took a v_lshl_add_u64 for each of the four)    */ \
                const int so0_ = (mi_ * 16 + RS * r0_) * row_bytes + ne_ * 16 * (int)sizeof(T);                 \
                const int so1_ = so0_ + RS * row_bytes;                                                         \
                if (PERS_EXP != 4) {                                                                            \
                    cbuf_store<T>(oth_[mi_][ne_][r0_], rs_prv, cbyte0, so0_);                                   \
                    cbuf_store<T>(oth_[mi_][ne_][r0_ + 1], rs_prv, cbyte0, so1_);                               \
                }                                                                                               \
                if (PERS_EXP != 5) {            /* straight into the set just stored from: nobody reads it before the next pass */ \
                    oth_[mi_][ne_][r0_] = cbuf_load<T>(rs_nxt, cbyte0, (PERS_EXP == 6) ? 0 : so0_);             \
                    oth_[mi_][ne_][r0_ + 1] = cbuf_load<T>(rs_nxt, cbyte0, (PERS_EXP == 6) ? 0 : so1_);         \
                }                                                                                               \
            }
    /* the stage's events: one (NKT = 16) or two (NKT = 8) */
#define PERS_EVENT_BLOCK(oth_)                                                                                  \
        { _Pragma("unroll") for (int ev_ = 0; ev_ < EVS; ++ev_) { PERS_EVENT_ONE(oth_, kt * EVS + ev_) } }
#define PERS_STAGE_OPERANDS(kt_)                                                                                \
            PERS_SWRITE(((kt_) & 1) ^ 1);           /* stage kt+1, in registers since the previous stage */     \
            if ((kt_) + 2 < NKT) PERS_GLOAD(a_chk, b_chk, (kt_) + 2)    /* stage kt+2 -> registers */            \
            else                 PERS_GLOAD(a_adv, b_adv, (kt_) + 2 - NKT)   /* ... of the next chunk of K, or of the next tile */
#if PERS_SYNC_FLAGS
#define PERS_SYNC_TOP()      PERS_WAIT(flag_b, seen_b, gstage - 1);
#define PERS_SYNC_WRITTEN()  PERS_POST(flag_a);
#define PERS_SYNC_READ()     PERS_POST(flag_b); seen_a = PERS_PEEK(flag_a); __builtin_amdgcn_sched_barrier(0);
#undef PERS_BARRIER_A
#undef PERS_BARRIER_B
#define PERS_BARRIER_A()     PERS_WAIT(flag_a, seen_a, gstage)
#define PERS_BARRIER_B()     do { } while (0)
#define PERS_SYNC_PEEK_B()   seen_b = PERS_PEEK(flag_b); __builtin_amdgcn_sched_barrier(0);
#define PERS_SYNC_NEXT()     ++gstage;
#else
#define PERS_SYNC_TOP()
#define PERS_SYNC_WRITTEN()
#define PERS_SYNC_READ()
#define PERS_SYNC_PEEK_B()
#define PERS_SYNC_NEXT()
#endif
#define PERS_PASS(cur_, oth_)                                                                                   \
    {                                                                                                           \
        const int tn_ = t_nx;                                                                                   \
        const bool has_next = tn_ >= 0;                                                                         \
        const int ni_ = has_next ? ti_nx : ti, nj_ = has_next ? tj_nx : tj;                                     \
        const T* c_nxt = C + (int64_t)ni_ * GT * ldc + (int64_t)nj_ * GT;      /* no next tile: this one */     \
        const __amdgpu_buffer_rsrc_t rs_nxt = tile_rsrc(c_nxt);                                                 \
        const T* a_nxt = A + (int64_t)ni_ * GT * lda;                                                           \
        const T* b_nxt = B + (int64_t)nj_ * GT * ldb;                                                           \
        /* K = nkc chunks of NKT stages (round 5: far updates of two or three panels at once, n > 8192): the same   */ \
        /* 16-stage body once per chunk, the accumulators carried on.  The C stream belongs to the FIRST chunk:    */ \
        /* later chunks address the previous tile through a descriptor without extent (their stores drop) and load */ \
        /* the next tile's values again (the same values into the same registers).                                 */ \
        for (int kc = 0; kc < (MULTI ? nkc : 1); ++kc) {                                                        \
        const bool lastc = !MULTI || (kc + 1 == nkc);                                                           \
        /* a head tile stored directly at the end of its own pass (head_direct) must NOT be stored again by this  */ \
        /* pass's events -- the panel chain may be rewriting it already: no extent, the stores drop              */ \
        const __amdgpu_buffer_rsrc_t rs_prv = (prv_stored || kc > 0) ? tile_rsrc_null(c_prv) : tile_rsrc(c_prv); \
        const T* a_chk = a_cur + kc * (NKT * BKE);                                                              \
        const T* b_chk = b_cur + kc * (NKT * BKE);                                                              \
        const T* a_adv = lastc ? a_nxt : a_chk + NKT * BKE;                                                     \
        const T* b_adv = lastc ? b_nxt : b_chk + NKT * BKE;                                                     \
        _Pragma("unroll") for (int kt = 0; kt < NKT; ++kt) {                                                    \
            if (kt == NKT / 2 && lastc) {            /* the tile after the next one: number and coordinates */   \
                t_nx = pers_tile_number(wg, it + 2, grid, ntiles);                                              \
                if (t_nx >= 0) decode(t_nx, ti_nx, tj_nx);                                                      \
            }                                                                                                   \
            PERS_SYNC_TOP()                          /* flags: everyone's reads of the buffer written below are issued */ \
            PERS_STAGE_OPERANDS(kt)                                                                             \
            PERS_SYNC_WRITTEN()                      /* flags: my share of stage kt+1 is behind me */            \
            if (PERS_EXP != 7 && PERS_EXP != 8) PERS_EVENT_BLOCK(oth_)                                          \
            /* k-step s+1's fragments are requested before k-step s is multiplied (the scheduler is fenced */  \
            /* so that it cannot fold the pairs back into read -> wait -> multiply)                        */  \
            PERS_FRAGS_NF(1, kt & 1, 1);                                                                        \
            PERS_MMA_NF(cur_, 0);                                                                               \
            PERS_INTERLEAVE();                                                                                  \
            PERS_FRAGS(0, kt & 1, 2);                                                                           \
            if (PERS_EXP == 7) { PERS_EVENT_BLOCK(oth_) PERS_MMA_NF(cur_, 1); PERS_INTERLEAVE(); }                  \
            else PERS_MMA(cur_, 1);                                                                             \
            PERS_FRAGS(1, kt & 1, 3);                                                                           \
            PERS_SYNC_READ()                         /* flags: my last reads of stage kt are issued; peek at the others' writes */ \
            PERS_MMA(cur_, 0);                                                                                  \
            PERS_BARRIER_A();                                                    /* stage kt+1 is in LDS */     \
            PERS_STAMP(it * 16 + kt);                                                                                   \
            PERS_FRAGS(0, (kt & 1) ^ 1, 0);         /* first fragments of stage kt+1 */                         \
            PERS_SYNC_PEEK_B()                       /* flags: a look at the others' last reads, tested at the top of the next stage */ \
            if (PERS_EXP == 8) { PERS_EVENT_BLOCK(oth_) PERS_MMA_NF(cur_, 1); PERS_INTERLEAVE(); }                  \
            else PERS_MMA(cur_, 1);                                                                             \
            PERS_BARRIER_B();                        /* everyone has read stage kt: its buffer may be rewritten */ \
            PERS_SYNC_NEXT()                                                                                    \
        }                                                                                                       \
        }                                           /* chunks of K */                                           \
        if (prev_head) signal_stored();             /* the previous tile's last store went out in this pass */  \
        prev_head = (t < heads);                                                                                \
        c_prv = c_cur;                                                                                          \
        if (!has_next) {                                                                                        \
            /* last tile of this workgroup: store it directly (the previous one went out during the pass) */   \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                    \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        (c_cur + ((int64_t)(mi * 16 + RS * r) * ldc + ni * 16))[coff] = cur_[mi][ni][r];        \
            if (prev_head) signal_stored();                                                                     \
            PERS_STAMP_EXIT();                                                                                  \
            return;                                                                                             \
        }                                                                                                       \
        prv_stored = false;                                                                                     \
        if (head_direct && prev_head) {                                                                         \
            /* the chain waits for this tile: out with it now instead of under the next pass (its round trip is   */ \
            /* not hidden: only launches whose chain is the longer path ask for this, gemm_nt_sub)                */ \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                    \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        (c_cur + ((int64_t)(mi * 16 + RS * r) * ldc + ni * 16))[coff] = cur_[mi][ni][r];        \
            signal_stored();                                                                                    \
            prev_head = false;                                                                                  \
            prv_stored = true;                                                                                  \
        }                                                                                                       \
        t = tn_; ti = ni_; tj = nj_; ++it;                                                                      \
        c_cur = const_cast<T*>(c_nxt); a_cur = a_nxt; b_cur = b_nxt;                                            \
    }

#if PERS_SYNC_FLAGS
    int seen_a = -1, seen_b = -1;                  // the flags as last peeked at (re-read in the wait if not there yet)
#endif
    for (;;) {
        PERS_PASS(acc0, acc1)
        PERS_PASS(acc1, acc0)
    }
#undef PERS_PASS
#undef PERS_STAGE_OPERANDS
#undef PERS_SYNC_TOP
#undef PERS_SYNC_WRITTEN
#undef PERS_SYNC_READ
#undef PERS_SYNC_PEEK_B
#undef PERS_SYNC_NEXT
#undef PERS_EVENT_BLOCK
#undef PERS_EVENT_ONE
#undef PERS_INTERLEAVE
#undef PERS_MMA
#undef PERS_MMA_NF
#undef PERS_FRAGS_NF
#undef PERS_FRAGS
#undef PERS_SWRITE
#undef PERS_GLOAD
}

// ---------------------------------------------------------------------------
// C[m x N] -= A[m x K] B[N x K]^T for a FEW rows (m <= 4: the q carried target rows of a fit, the rows left over from
// the 128-row tiles of a far update).  A matrix-core tile for 2 rows wastes the tile and, worse, its latency-bound
// workgroups read B at 0.75 TB/s (a layer of 128 blocks of 2048 spent 4.4 of its 11.8 ms here).  This is a
// matrix-vector product: one wave per row of B (coalesced 32 bytes per lane per 2 KB), the rows of A in registers,
// a fixed-order butterfly reduction; eight rows of B in flight per wave.  Memory-bound on B, read once.
// ---------------------------------------------------------------------------
constexpr int THIN_RPW = 8;        // rows of B per wave
constexpr int THIN_MAXM = 4;       // (5-8 rows: the register budget of eight rows in flight does not hold; those keep the tile kernel)
template <typename T, int M>
__global__ __launch_bounds__(256)
void k_thin_update(T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                   int N, int K, int64_t sc, int64_t sa, int64_t sb)
{
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    constexpr int EPL = 32 / (int)sizeof(T);          // elements per lane and chunk (32 bytes)
    constexpr int CH = 64 * EPL;                      // elements per chunk: 256 doubles / 512 floats
    C += (int64_t)blockIdx.y * sc;
    A += (int64_t)blockIdx.y * sa;
    B += (int64_t)blockIdx.y * sb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = ((int)blockIdx.x * 4 + wave) * THIN_RPW;
    if (n0 >= N) return;
    T acc[THIN_RPW][M];
#pragma unroll
    for (int i = 0; i < THIN_RPW; ++i)
#pragma unroll
        for (int r = 0; r < M; ++r) acc[i][r] = (T)0;
    for (int kc = 0; kc < K; kc += CH) {
        const int ke = kc + lane * EPL;
        const bool live = ke < K;                     // K is a multiple of 16 bytes (the callers' leading dimensions are)
        T av[M][EPL];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            union { v4u v[2]; T t[EPL]; } u;
            u.v[0] = live ? *reinterpret_cast<const v4u*>(A + (int64_t)r * lda + ke) : v4u{0, 0, 0, 0};
            u.v[1] = (live && ke + EPL / 2 < K) ? *reinterpret_cast<const v4u*>(A + (int64_t)r * lda + ke + EPL / 2) : v4u{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < EPL; ++e) av[r][e] = u.t[e];
        }
        union { v4u v[2]; T t[EPL]; } bv[THIN_RPW];
#pragma unroll
        for (int i = 0; i < THIN_RPW; ++i) {
            const int n = min(n0 + i, N - 1);         // rows past the end: a valid row, never stored
            const T* bp = B + (int64_t)n * ldb + ke;
            bv[i].v[0] = live ? *reinterpret_cast<const v4u*>(bp) : v4u{0, 0, 0, 0};
            bv[i].v[1] = (live && ke + EPL / 2 < K) ? *reinterpret_cast<const v4u*>(bp + EPL / 2) : v4u{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < THIN_RPW; ++i)
#pragma unroll
            for (int r = 0; r < M; ++r)
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[i][r] += av[r][e] * bv[i].t[e];
    }
#pragma unroll
    for (int i = 0; i < THIN_RPW; ++i)
#pragma unroll
        for (int r = 0; r < M; ++r) {
            T v = acc[i][r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[i][r] = v;
        }
    // lane (i, r) = i * M + r writes C[r][n0 + i]
    if (lane < THIN_RPW * M) {
        const int i = lane / M, r = lane - i * M;
        T v = (T)0;
#pragma unroll
        for (int ii = 0; ii < THIN_RPW; ++ii)
#pragma unroll
            for (int rr = 0; rr < M; ++rr)
                if (ii == i && rr == r) v = acc[ii][rr];
        if (n0 + i < N) C[(int64_t)r * ldc + n0 + i] -= v;
    }
}

}  // namespace

template <typename T, int W>
static int gemm_launch(T* c, int64_t ldc, const T* a, int64_t lda, const T* b, int64_t ldb,
                       int64_t m, int64_t n, int k, bool lower, hipStream_t st, const GemmBatch& bt)
{
    const char* fn = "gemm_nt_sub";
    constexpr int GT = 32 * W;
    const int64_t tm = (m + GT - 1) / GT, tn = (n + GT - 1) / GT;
    const int64_t tiles = lower ? tm * (tm + 1) / 2 : tm * tn;
    CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
    CIMRGP_REQUIRE(bt.count >= 1 && bt.count < 65536, fn, "batch count out of range");
    const dim3 grid((unsigned)tiles, (unsigned)bt.count);
    if (lower) hipLaunchKernelGGL((k_gemm_nt_sub<T, true, W>), grid, dim3(256), 0, st,
                                  c, ldc, a, lda, b, ldb, (int)m, (int)n, k, (int)tn, bt.sc, bt.sa, bt.sb, bt.skip_first);
    else       hipLaunchKernelGGL((k_gemm_nt_sub<T, false, W>), grid, dim3(256), 0, st,
                                  c, ldc, a, lda, b, ldb, (int)m, (int)n, k, (int)tn, bt.sc, bt.sa, bt.sb, bt.skip_first);
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

// Whether a lower update of this shape can run as ONE head-first persistent launch (potrf_run), and how many
// head tiles it then signals.
int gemm_pers_head_tiles(int64_t m, int k, int elem_bytes)
{
    const int bke = KT_BYTES / elem_bytes;
    const int pers_nkt = (elem_bytes == 8) ? PERS_STAGES : PERS_STAGES / 2;
    if (knobs().gemm_pers < 8 || (elem_bytes != 8 && !knobs().gemm_pers_f32) || m % 128 != 0 || m / 128 < 3 || k % bke != 0 || k / bke != pers_nkt) return 0;
    return (int)(2 * (m / 128) - 2);
}

// The tile-size rule of gemm_nt_sub, for callers that rely on the 64-tile grid (skip_first).
bool gemm_uses_tile64(int64_t m, int64_t n, bool lower, int count)
{
    const int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128) / (lower ? 2 : 1) * count;
    const int64_t t64 = ((m + 63) / 64) * ((n + 63) / 64) / (lower ? 2 : 1) * count;
    const int64_t thin = (m < n) ? m : n;
    return !lower && t64 >= 32 && thin > 32 && (t128 < 768 || thin <= 64);
}

template <typename T>
int gemm_nt_sub(T* c, int64_t ldc, const T* a, int64_t lda, const T* b, int64_t ldb,
                int64_t m, int64_t n, int k, bool lower, hipStream_t st, GemmBatch bt)
{
    const char* fn = "gemm_nt_sub";
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    CIMRGP_REQUIRE(m < (1ll << 30) && n < (1ll << 30), fn, "matrix too large");
    CIMRGP_REQUIRE(aligned16(a) && aligned16(b), fn, "operand base not 16-byte aligned");
    CIMRGP_REQUIRE(lda % Mx<T>::EPC == 0 && ldb % Mx<T>::EPC == 0, fn, "leading dimension not a multiple of 16 bytes");
    CIMRGP_REQUIRE(!lower || m == n, fn, "lower update needs a square C");
    if (!lower && m <= THIN_MAXM && !bt.skip_first && !bt.head_first && k % Mx<T>::EPC == 0 && aligned16(a) && aligned16(b)) {
        const dim3 grid((unsigned)((n + 4 * THIN_RPW - 1) / (4 * THIN_RPW)), (unsigned)bt.count);
#define CIMRGP_THIN(M_) case M_: hipLaunchKernelGGL((k_thin_update<T, M_>), grid, dim3(256), 0, st, c, ldc, a, lda, b, ldb, (int)n, k, bt.sc, bt.sa, bt.sb); break;
        switch ((int)m) { CIMRGP_THIN(1) CIMRGP_THIN(2) CIMRGP_THIN(3) CIMRGP_THIN(4) }
#undef CIMRGP_THIN
        CIMRGP_LAUNCH_CHECK(fn);
        return 0;
    }
    // fewer than ~3 workgroups per CU with 128-tiles: use 64-tiles (4x the workgroups, 1/4 the work
    // each; measured sweep of the switch point inside the factorisation at N = 8192:
    // 256/512/768/1024/1536 tiles -> 91.3/93.5/94.2/91.9/90.6 posteriors/s)
    // (a batch multiplies the number of workgroups: choose the tile for the whole launch)
    CIMRGP_REQUIRE(!bt.skip_first || gemm_uses_tile64(m, n, lower, bt.count), fn, "skip_first needs a 64-tile launch");
    const int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128) / (lower ? 2 : 1) * bt.count;
    // tiny updates on the factorisation's critical path (the 256 x 256 diagonal block): 32-tiles, so
    // that the K loop of a tile is 1/4 as long and ~36 compute units share it instead of 10
    const int64_t t64 = ((m + 63) / 64) * ((n + 63) / 64) / (lower ? 2 : 1) * bt.count;
    // the persistent form: full 128-tiles only, enough of them to give every workgroup several
    {
        constexpr int bke = KT_BYTES / (int)sizeof(T);
        const int want = (bt.pers >= 0) ? bt.pers : knobs().gemm_pers;
        const int nkt = (k % bke) ? 0 : k / bke;
        // one panel in either precision: 16 stages of doubles, 8 of floats.  The FP32 form is off by default
        // (knobs().gemm_pers_f32): stand-alone it equals the tile kernel (0.52-0.58 of the FP32 peak both), inside the
        // factorisation it is slower (3.85 against 3.64 ms at n = 8192): HISTORY.md, round 4.
        constexpr int pers_nkt = (sizeof(T) == 8) ? PERS_STAGES : PERS_STAGES / 2;
        const int nkc = (nkt > 0 && nkt % pers_nkt == 0) ? nkt / pers_nkt : 0;       // chunks of one panel's K (256 columns)
        if (want >= 8 && (sizeof(T) == 8 || knobs().gemm_pers_f32) && nkc >= 1 && nkc <= knobs().pers_max_chunks && (nkc == 1 || !bt.head_first) &&
            bt.count == 1 && !bt.skip_first && m % 128 == 0 && n % 128 == 0 &&
            ldc < (1ll << 20) && lda < (1ll << 23) && ldb < (1ll << 23) && (t128 >= knobs().pers_min_tiles || bt.head_first || bt.pers_force)) {
            // (ldc: 128 rows of C stay below 2^31 bytes -- the C stream addresses a tile through a raw buffer with 32-bit offsets)
            const int64_t tm = m / 128, tn = n / 128;
            // head-first launch (the look-ahead's combined head + bulk update): tile (0, 0) is left to the chain
            const int heads = (bt.head_first && lower && tm >= 3) ? (int)(2 * tm - 2) : 0;
            CIMRGP_REQUIRE(!bt.head_first || heads > 0, fn, "a head-first update needs a lower update of at least 3 x 3 tiles");
            const int64_t tiles = (lower ? tm * (tm + 1) / 2 : tm * tn) - (heads ? 1 : 0);
            const int cus = (want < 256 ? want : 256) & ~7;     // one workgroup per compute unit (gfx950: 256), 8 XCDs
            // A launch lasts a whole number of ROUNDS of tiles (one 128 x 128 tile per workgroup and round), so a few
            // more workgroups can save a whole round: 1829 tiles take 9 rounds on 224 units and 8 on 232 (-11 %).  With
            // knobs().pers_flex_cus > 0 the caller's share may be exceeded by that many units when it removes a round.
            // Round 5 measured it inside the factorisation and left it OFF: what the update gains the panel chain loses.
            int cus_max = cus + (knobs().pers_flex_cus & ~7);
            if (cus_max > 256) cus_max = 256;
            int64_t rounds = (tiles + cus - 1) / cus;
            int use = cus;
            if (rounds >= knobs().pers_flex_min_rounds && (tiles + cus_max - 1) / cus_max < rounds) { rounds = (tiles + cus_max - 1) / cus_max; use = cus_max; }
            int64_t g8 = (tiles + rounds - 1) / rounds;         // every workgroup busy in (nearly) every round ...
            g8 = (g8 + 7) / 8 * 8;                              // ... and the same number of them on every XCD
            if (g8 > use) g8 = use;
            const dim3 grid((unsigned)g8);
            // head tiles straight out at the end of their own pass when the launch is short enough for the panel chain --
            // which waits for them -- to be the longer path of the panel (knobs().head_direct_max_rounds)
            const int head_direct = (heads > 0 && rounds <= knobs().head_direct_max_rounds) ? 1 : 0;
#define CIMRGP_PERS_LAUNCH(LOW_, MULTI_) \
            hipLaunchKernelGGL((k_gemm_nt_pers<T, LOW_, pers_nkt, MULTI_>), grid, dim3(PERS_THREADS), 0, st, c, ldc, a, lda, b, ldb, (int)tm, (int)tn, (int)tiles, heads, bt.flag, head_direct, nkc)
            if (lower) { if (nkc > 1) CIMRGP_PERS_LAUNCH(true, true); else CIMRGP_PERS_LAUNCH(true, false); }
            else       { if (nkc > 1) CIMRGP_PERS_LAUNCH(false, true); else CIMRGP_PERS_LAUNCH(false, false); }
#undef CIMRGP_PERS_LAUNCH
            CIMRGP_LAUNCH_CHECK(fn);
            return 0;
        }
    }
    CIMRGP_REQUIRE(!bt.head_first, fn, "head-first update not eligible for the persistent kernel (gemm_pers_eligible)");
    // a thin C (the q carried target rows of a batched fit: m = 2) must not be padded to 128-row tiles: a batch of
    // 128 such updates used 128 x 128 tiles for 2 rows each and took a quarter of a fine layer's fit time
    const int64_t thin = (m < n) ? m : n;
    if (!bt.skip_first && thin <= 32) return gemm_launch<T, 1>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
    if (t64 < 32) return gemm_launch<T, 1>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
    if (t128 < 768 || thin <= 64) return gemm_launch<T, 2>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
    return gemm_launch<T, 4>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
}

#ifdef PERS_STAMPS
extern "C" int cimrgp_debug_pers_stamps(long long* out_host)
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_pers_stamp), sizeof(g_pers_stamp)) == hipSuccess ? 0 : -1;
}
#endif

template int gemm_nt_sub<double>(double*, int64_t, const double*, int64_t, const double*, int64_t,
                                 int64_t, int64_t, int, bool, hipStream_t, GemmBatch);
template int gemm_nt_sub<float>(float*, int64_t, const float*, int64_t, const float*, int64_t,
                                int64_t, int64_t, int, bool, hipStream_t, GemmBatch);

}  // namespace cimrgp
