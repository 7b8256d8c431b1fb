// The tile body of the update  C[M x N] -= A[M x K] * B[N x K]^T  (gemm_nt.hip has the launchers and the
// geometry notes).  A header so that the panel-chain kernels of the factorisation (potrf.hip) can run update
// tiles as "rider" workgroups of their own launches.
#pragma once
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
    // so the s_waitcnt lands there and the loads fly during the multiply.  The sign that turns the
    // chain into C - A B^T is the multiply's own operand negation (Mx<T>::mma_neg).
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
                a[mi] = *reinterpret_cast<const uint2*>(as + a_off + mi * 16 * LROW + s * 32);
#pragma unroll
            for (int ni = 0; ni < W; ++ni)
                b[ni] = *reinterpret_cast<const uint2*>(bs + b_off + ni * 16 * LROW + s * 32);
#pragma unroll
            for (int mi = 0; mi < W; ++mi)
#pragma unroll
                for (int ni = 0; ni < W; ++ni) acc[mi][ni] = X::mma_neg(a[mi], b[ni], acc[mi][ni]);
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

}  // namespace
}  // namespace cimrgp
