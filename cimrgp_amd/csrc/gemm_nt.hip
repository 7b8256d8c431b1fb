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

namespace cimrgp {

namespace {

constexpr int KT_BYTES = 128;            // K bytes per row per stage
constexpr int LROW     = KT_BYTES + 16;  // LDS row stride
// W = MFMA tiles per wave and direction: W = 4 -> 128 x 128 workgroup tile (the trailing
// update: 73,728 B of LDS, 2 workgroups per CU), W = 2 -> 64 x 64 tile (thin updates such as
// the look-ahead "head", where a 128-tile grid would leave half of the CUs empty).

// EDGE = false: M, N multiples of 128 and K a multiple of the stage depth -- no bounds logic
// at all (every select on a prefetched register makes hipcc wait for it right behind the
// load).  EDGE = true: ragged sizes, zero-fill and predicated stores.
template <typename T, bool LOWER, bool EDGE, int W>
static __device__ __forceinline__ void gemm_tile(unsigned char* smem, T* __restrict__ C, int64_t ldc,
                                                  const T* __restrict__ A, int64_t lda,
                                                  const T* __restrict__ B, int64_t ldb,
                                                  int M, int N, int K, int ti, int tj)
{
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    constexpr int BKE = KT_BYTES / (int)sizeof(T);
    constexpr int GT = 32 * W;               // tile edge
    constexpr int NP = GT / 32;              // staging passes (32 rows each)
    constexpr int OP_BYTES = GT * LROW;      // one operand, one stage
    const int row0 = ti * GT, col0 = tj * GT;

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // global -> register staging map: 8 threads cover one 128-byte row segment
    const int sc = tid & 7;
    const int sr = tid >> 3;
    const int nkt = (K + BKE - 1) / BKE;

    // Staging addresses: uniform tile base (SGPRs) + one 32-bit per-thread element offset per
    // operand row group, so the 8 loads of a stage need 8 VGPRs of addressing, not 16.
    bool a_ok[NP], b_ok[NP];
    int a_off_e[NP], b_off_e[NP];
    const T* a_tile = A + (int64_t)row0 * lda;
    const T* b_tile = B + (int64_t)col0 * ldb;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = sr + 32 * p;
        a_ok[p] = (row0 + r) < M;
        b_ok[p] = (col0 + r) < N;
        a_off_e[p] = ((EDGE && !a_ok[p]) ? 0 : r) * (int)lda + sc * X::EPC;
        b_off_e[p] = ((EDGE && !b_ok[p]) ? 0 : r) * (int)ldb + sc * X::EPC;
    }

    uint4 ra[NP], rb[NP];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    // GLOAD only ISSUES the loads (clamped, always valid addresses); every use of the loaded
    // registers (zero-fill of out-of-range rows/columns) is in SWRITE, after the MFMA block,
    // so the s_waitcnt lands there and the loads fly during the multiply.  The sign flip that
    // turns the chain into C - A B^T is applied to the A fragments after the LDS read (one
    // XOR per fragment and 16 MFMAs); flipping the staged registers component-wise makes
    // hipcc shuffle them right behind the loads and wait there.
#define CIMRGP_GLOAD(kt_)                                                        \
    {                                                                            \
        const int kcol = (kt_) * BKE + sc * X::EPC;                              \
        const int koff = (!EDGE || kcol < K) ? (kt_) * BKE : 0;                  \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                         \
            ra[p] = *reinterpret_cast<const uint4*>(a_tile + (a_off_e[p] + koff)); \
            rb[p] = *reinterpret_cast<const uint4*>(b_tile + (b_off_e[p] + koff)); \
        }                                                                        \
    }
#define CIMRGP_SWRITE(buf_, kt_)                                                 \
    {                                                                            \
        unsigned char* as_ = smem + (buf_) * 2 * OP_BYTES;                       \
        unsigned char* bs_ = as_ + OP_BYTES;                                     \
        const int kcol = (kt_) * BKE + sc * X::EPC;                              \
        const bool kin = kcol < K;                                               \
        const bool kfull = kcol + X::EPC <= K;                                   \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                         \
            uint4 va = ra[p], vb = rb[p];                                        \
            if (EDGE) {                                                          \
                if (!(a_ok[p] && kin)) va = zero4;                               \
                if (!(b_ok[p] && kin)) vb = zero4;                               \
                if (!kfull) { va = mask_chunk<T>(va, kcol, K); vb = mask_chunk<T>(vb, kcol, K); } \
            }                                                                    \
            *reinterpret_cast<uint4*>(as_ + (sr + 32 * p) * LROW + sc * 16) = va; \
            *reinterpret_cast<uint4*>(bs_ + (sr + 32 * p) * LROW + sc * 16) = vb; \
        }                                                                        \
    }

    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned a_off = (unsigned)((wr * 16 * W + frow) * LROW + fslot * 8);
    const unsigned b_off = (unsigned)((wc * 16 * W + frow) * LROW + fslot * 8);

    STAMP(16);
    CIMRGP_GLOAD(0);
    // The accumulators start as the C tile and the A fragments are NEGATED, so the MFMA
    // chain itself computes C - A B^T: the 64 C loads per lane are independent and in flight
    // together with the first operand tiles, and the epilogue is stores only.  (A read-modify-
    // write epilogue serialises 64 dependent load->store round trips per lane.)
    const bool diag_tile = LOWER && (ti == tj);
    acc_t acc[W][W];
#pragma unroll
    for (int mi = 0; mi < W; ++mi) {
#pragma unroll
        for (int ni = 0; ni < W; ++ni) {
            const int gc = col0 + wc * 16 * W + ni * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + wr * 16 * W + mi * 16 + X::crow(lane, r);
                // unconditional load from a clamped (always valid) address, then select
                if (EDGE) {
                    const T v = C[(int64_t)min(gr, M - 1) * ldc + min(gc, N - 1)];
                    acc[mi][ni][r] = (gr < M && gc < N && (!diag_tile || gc <= gr)) ? v : (T)0;
                } else {
                    acc[mi][ni][r] = C[(int64_t)gr * ldc + gc];     // junk above the diagonal is never stored
                }
            }
        }
    }
    CIMRGP_SWRITE(0, 0);
    // Make the C loads complete HERE: otherwise hipcc guards the first MFMA of every loop
    // iteration with s_waitcnt vmcnt(0), which also drains the operand prefetch just issued.
#pragma unroll
    for (int mi = 0; mi < W; ++mi)
#pragma unroll
        for (int ni = 0; ni < W; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
    __syncthreads();
    STAMP(17);

    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more) CIMRGP_GLOAD(kt + 1);
        const unsigned char* as = smem + (kt & 1) * 2 * OP_BYTES;
        const unsigned char* bs = as + OP_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint2 a[W], b[W];
#pragma unroll
            for (int mi = 0; mi < W; ++mi)
                a[mi] = X::neg(*reinterpret_cast<const uint2*>(as + a_off + mi * 16 * LROW + s * 32));
#pragma unroll
            for (int ni = 0; ni < W; ++ni)
                b[ni] = *reinterpret_cast<const uint2*>(bs + b_off + ni * 16 * LROW + s * 32);
#pragma unroll
            for (int mi = 0; mi < W; ++mi)
#pragma unroll
                for (int ni = 0; ni < W; ++ni) acc[mi][ni] = X::mma(a[mi], b[ni], acc[mi][ni]);
        }
        if (more) CIMRGP_SWRITE((kt + 1) & 1, kt + 1);
        __syncthreads();
    }
#undef CIMRGP_GLOAD
#undef CIMRGP_SWRITE

    STAMP(18);
    int Mv = M, Nv = N;
    asm volatile("" : "+s"(Mv), "+s"(Nv));      // recompute the store predicates here (not hoisted over the loop)
    // epilogue: store the tile (f64 map: 16 lanes x 8 B = one 128-byte line per row)
#pragma unroll
    for (int mi = 0; mi < W; ++mi) {
#pragma unroll
        for (int ni = 0; ni < W; ++ni) {
            const int gc = col0 + wc * 16 * W + ni * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + wr * 16 * W + mi * 16 + X::crow(lane, r);
                if ((!EDGE || (gr < Mv && gc < Nv)) && (!diag_tile || gc <= gr)) C[(int64_t)gr * ldc + gc] = acc[mi][ni][r];
            }
        }
    }
    STAMP(19);
}

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
template <typename T, bool LOWER>
static __device__ __forceinline__ void pers_tile_of(int id, int tiles_n, int& ti, int& tj)
{
    if (LOWER) {
        ti = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
        while (ti * (ti + 1) / 2 > id) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
        tj = id - ti * (ti + 1) / 2;
    } else {
        ti = id / tiles_n;
        tj = id - ti * tiles_n;
    }
}

// The C stream of the persistent kernel is cut into 16 "events" per tile and wave, one per K stage
// (K = 16 stages of 128 bytes: 256 doubles / 512 floats): event e covers accumulator tile
// (mi, ni) = (e >> 2, (e >> 1) & 1), values r = 2 (e & 1) + {0, 1} (two rows of 16 lanes x 8 bytes = two
// 128-byte lines per row tile).  An event stores the finished values of the previous tile and requests the
// next tile's into two staging registers, which are moved into the accumulator set TWO events later.
// The 16 stages of a tile are fully unrolled: every register index is a compile-time constant and the
// whole pass is straight-line code, so the compiler counts the memory instructions exactly -- the
// wait for stage g+1's operands at the top of stage g leaves the 4 C accesses issued behind them in
// flight.  (With the events inside a switch over the stage number hipcc waited for vmcnt(0) at the top
// of every stage, and every stage then took a full HBM round trip under load: 40 against 45 TF/s for
// the tile-per-workgroup kernel.)
constexpr int PERS_THREADS = 512;
constexpr int PERS_STAGES = 16;

template <typename T, bool LOWER>
__global__ __launch_bounds__(PERS_THREADS)
void k_gemm_nt_pers(T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda,
                    const T* __restrict__ B, int64_t ldb, int tiles_n, int ntiles)
{
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    constexpr int BKE = KT_BYTES / (int)sizeof(T);
    constexpr int GT = 128;
    constexpr int OP_BYTES = GT * LROW;
    constexpr int RS = (sizeof(T) == 8) ? 4 : 1;             // crow(lane, r) = crow(lane, 0) + RS r
    constexpr int NKT = PERS_STAGES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * OP_BYTES];       // 2 stages x 2 operands

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // 2 x 4 waves: 64 rows x 32 columns each
    const int sc = tid & 7, sr = tid >> 3;                   // staging: 8 threads per 128-byte row segment, 64 rows per pass
    const int stride = (int)gridDim.x;

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

    int t = (int)blockIdx.x;
    if (t >= ntiles) return;
    int ti, tj;
    pers_tile_of<T, LOWER>(t, tiles_n, ti, tj);
    T* c_cur = C + (int64_t)ti * GT * ldc + (int64_t)tj * GT;
    const T* a_cur = A + (int64_t)ti * GT * lda;
    const T* b_cur = B + (int64_t)tj * GT * ldb;
    T* c_prv = c_cur;
    // first-class vectors: arrays of HIP's uint4 struct filled from global memory can stay in scratch
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u ra[2], rb[2];
    uint2 fa[2][4], fb[2][2];                                // two fragment sets: k-step s+1 is read while s is multiplied
    acc_t acc0[4][2], acc1[4][2];
    T tld[2][2];                                             // next tile's values in flight (two events deep)

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
            *reinterpret_cast<v4u*>(as_ + (sr + 64 * p) * LROW + sc * 16) = ra[p]; \
            *reinterpret_cast<v4u*>(bs_ + (sr + 64 * p) * LROW + sc * 16) = rb[p]; \
        }                                                                          \
    }
#define PERS_FRAGS(set_, buf_, s_)                                                 \
    {                                                                              \
        const unsigned char* as_ = smem + (buf_) * 2 * OP_BYTES;                   \
        const unsigned char* bs_ = as_ + OP_BYTES;                                 \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                           \
            fa[set_][mi] = *reinterpret_cast<const uint2*>(as_ + a_frag + mi * 16 * LROW + (s_) * 32); \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                           \
            fb[set_][ni] = *reinterpret_cast<const uint2*>(bs_ + b_frag + ni * 16 * LROW + (s_) * 32); \
        __builtin_amdgcn_sched_barrier(0);      /* the reads stay AHEAD of the multiplies that follow */ \
    }
#define PERS_MMA(cur_, set_)                                                       \
    {                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                         \
            const uint2 an_ = X::neg(fa[set_][mi]);                                \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                       \
                cur_[mi][ni] = X::mma(an_, fb[set_][ni], cur_[mi][ni]);            \
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
    // The first pass has no finished tile to store, and its events store all the same (no branch around a
    // memory instruction): the other set starts as a copy of the first tile's C, so those stores put the
    // values just read back where they came from.
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc1[mi][ni] = acc0[mi][ni];
    PERS_SWRITE(0);
    PERS_GLOAD(a_cur, b_cur, 1);
    __syncthreads();
    PERS_FRAGS(0, 0, 0);

    // One tile: `cur` holds its C (requested during the previous pass), `oth` the finished previous
    // tile, which is stored and replaced by the next tile's C as the K loop proceeds.  The ring position
    // of a stage is its number's parity (16 stages per tile).
#define PERS_PASS(cur_, oth_)                                                                                   \
    {                                                                                                           \
        const int tn_ = t + stride;                                                                             \
        const bool has_next = tn_ < ntiles;                                                                     \
        int ni_ = ti, nj_ = tj;                                                                                 \
        if (has_next) pers_tile_of<T, LOWER>(tn_, tiles_n, ni_, nj_);                                           \
        const T* c_nxt = C + (int64_t)ni_ * GT * ldc + (int64_t)nj_ * GT;      /* no next tile: this one */     \
        const T* a_nxt = A + (int64_t)ni_ * GT * lda;                                                           \
        const T* b_nxt = B + (int64_t)nj_ * GT * ldb;                                                           \
        _Pragma("unroll") for (int kt = 0; kt < NKT; ++kt) {                                                    \
            PERS_SWRITE((kt & 1) ^ 1);              /* stage kt+1, in registers since the previous stage */     \
            if (kt + 2 < NKT) PERS_GLOAD(a_cur, b_cur, kt + 2)      /* stage kt+2 -> registers */                \
            else              PERS_GLOAD(a_nxt, b_nxt, kt + 2 - NKT)                                            \
            {                                       /* event kt */                                              \
                constexpr int EM = 0;  (void)EM;                                                                \
                const int mi_ = kt >> 2, ne_ = (kt >> 1) & 1, r0_ = 2 * (kt & 1);                               \
                if (kt >= 2) {                      /* the values requested two events ago have arrived */      \
                    oth_[(kt - 2) >> 2][((kt - 2) >> 1) & 1][2 * ((kt - 2) & 1)] = tld[kt & 1][0];              \
                    oth_[(kt - 2) >> 2][((kt - 2) >> 1) & 1][2 * ((kt - 2) & 1) + 1] = tld[kt & 1][1];          \
                }                                                                                               \
                const int64_t uo_ = (int64_t)(mi_ * 16 + RS * r0_) * ldc + ne_ * 16;                            \
                T* sb_ = c_prv + uo_;                                                                           \
                sb_[coff] = oth_[mi_][ne_][r0_];                                                                \
                (sb_ + (int64_t)RS * ldc)[coff] = oth_[mi_][ne_][r0_ + 1];                                      \
                const T* lb_ = c_nxt + uo_;                                                                     \
                tld[kt & 1][0] = lb_[coff];                                                                     \
                tld[kt & 1][1] = (lb_ + (int64_t)RS * ldc)[coff];                                               \
            }                                                                                                   \
            /* k-step s+1's fragments are requested before k-step s is multiplied (the scheduler is fenced */  \
            /* so that it cannot fold the pairs back into read -> wait -> multiply)                        */  \
            PERS_FRAGS(1, kt & 1, 1);                                                                           \
            PERS_MMA(cur_, 0);                                                                                  \
            PERS_FRAGS(0, kt & 1, 2);                                                                           \
            PERS_MMA(cur_, 1);                                                                                  \
            PERS_FRAGS(1, kt & 1, 3);                                                                           \
            PERS_MMA(cur_, 0);                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      /* stage kt+1 is in LDS */     \
            PERS_FRAGS(0, (kt & 1) ^ 1, 0);         /* first fragments of stage kt+1 */                         \
            PERS_MMA(cur_, 1);                                                                                  \
            asm volatile("s_barrier" ::: "memory");  /* everyone has read stage kt: its buffer may be rewritten */ \
        }                                                                                                       \
        /* the last two events of the pass: the next tile's C is complete in `oth` */                          \
        oth_[3][1][0] = tld[0][0]; oth_[3][1][1] = tld[0][1];                                                   \
        oth_[3][1][2] = tld[1][0]; oth_[3][1][3] = tld[1][1];                                                   \
        c_prv = c_cur;                                                                                          \
        if (!has_next) {                                                                                        \
            /* last tile of this workgroup: store it directly (the previous one went out during the pass) */   \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                    \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
                        (c_cur + ((int64_t)(mi * 16 + RS * r) * ldc + ni * 16))[coff] = cur_[mi][ni][r];        \
            return;                                                                                             \
        }                                                                                                       \
        t = tn_; ti = ni_; tj = nj_;                                                                            \
        c_cur = const_cast<T*>(c_nxt); a_cur = a_nxt; b_cur = b_nxt;                                            \
    }

    for (;;) {
        PERS_PASS(acc0, acc1)
        PERS_PASS(acc1, acc0)
    }
#undef PERS_PASS
#undef PERS_MMA
#undef PERS_FRAGS
#undef PERS_SWRITE
#undef PERS_GLOAD
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

// The tile-size rule of gemm_nt_sub, for callers that rely on the 64-tile grid (skip_first).
bool gemm_uses_tile64(int64_t m, int64_t n, bool lower, int count)
{
    const int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128) / (lower ? 2 : 1) * count;
    const int64_t t64 = ((m + 63) / 64) * ((n + 63) / 64) / (lower ? 2 : 1) * count;
    return !lower && t64 >= 32 && t128 < 768;
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
        if (want > 0 && nkt == PERS_STAGES && bt.count == 1 && !bt.skip_first && m % 128 == 0 && n % 128 == 0 &&
            ldc < (1ll << 23) && lda < (1ll << 23) && ldb < (1ll << 23) && t128 >= knobs().pers_min_tiles) {
            const int64_t tm = m / 128, tn = n / 128;
            const int64_t tiles = lower ? tm * (tm + 1) / 2 : tm * tn;
            const int cus = want < 256 ? want : 256;            // one workgroup per compute unit (gfx950: 256)
            const int64_t rounds = (tiles + cus - 1) / cus;
            const dim3 grid((unsigned)((tiles + rounds - 1) / rounds));          // every workgroup busy in every round
#define CIMRGP_PERS_LAUNCH(LOW_) \
            hipLaunchKernelGGL((k_gemm_nt_pers<T, LOW_>), grid, dim3(PERS_THREADS), 0, st, c, ldc, a, lda, b, ldb, (int)tn, (int)tiles)
            if (lower) CIMRGP_PERS_LAUNCH(true); else CIMRGP_PERS_LAUNCH(false);
#undef CIMRGP_PERS_LAUNCH
            CIMRGP_LAUNCH_CHECK(fn);
            return 0;
        }
    }
    if (t64 < 32) return gemm_launch<T, 1>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
    if (t128 < 768) return gemm_launch<T, 2>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
    return gemm_launch<T, 4>(c, ldc, a, lda, b, ldb, m, n, k, lower, st, bt);
}

template int gemm_nt_sub<double>(double*, int64_t, const double*, int64_t, const double*, int64_t,
                                 int64_t, int64_t, int, bool, hipStream_t, GemmBatch);
template int gemm_nt_sub<float>(float*, int64_t, const float*, int64_t, const float*, int64_t,
                                int64_t, int64_t, int, bool, hipStream_t, GemmBatch);

}  // namespace cimrgp
