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

constexpr int GT       = 128;            // tile edge (rows of A-tile = rows of B-tile)
constexpr int KT_BYTES = 128;            // K bytes per row per stage
constexpr int LROW     = KT_BYTES + 16;  // LDS row stride
constexpr int OP_BYTES = GT * LROW;      // one operand, one stage
constexpr int SMEM     = 4 * OP_BYTES;   // {A,B} x 2 stages = 73,728 B

template <typename T, bool LOWER>
__global__ __launch_bounds__(256, 2)
void k_gemm_nt_sub(T* __restrict__ C, int64_t ldc,
                   const T* __restrict__ A, int64_t lda,
                   const T* __restrict__ B, int64_t ldb,
                   int M, int N, int K, int tiles_n)
{
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    constexpr int BKE = KT_BYTES / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

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
    const int row0 = ti * GT, col0 = tj * GT;

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // global -> register staging map: 8 threads cover one 128-byte row segment
    const int sc = tid & 7;
    const int sr = tid >> 3;
    const int nkt = (K + BKE - 1) / BKE;

    const T* a_ptr[4];
    const T* b_ptr[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = sr + 32 * p;
        a_ok[p] = (row0 + r) < M;
        b_ok[p] = (col0 + r) < N;
        a_ptr[p] = A + (int64_t)(a_ok[p] ? row0 + r : 0) * lda + sc * X::EPC;
        b_ptr[p] = B + (int64_t)(b_ok[p] ? col0 + r : 0) * ldb + sc * X::EPC;
    }

    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

#define CIMRGP_GLOAD(kt_)                                                        \
    {                                                                            \
        const int kcol = (kt_) * BKE + sc * X::EPC;                              \
        const bool kin = kcol < K;                                               \
        const bool kfull = kcol + X::EPC <= K;                                   \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                          \
            uint4 va = zero4, vb = zero4;                                        \
            if (a_ok[p] && kin) va = *reinterpret_cast<const uint4*>(a_ptr[p] + (kt_) * BKE); \
            if (b_ok[p] && kin) vb = *reinterpret_cast<const uint4*>(b_ptr[p] + (kt_) * BKE); \
            if (!kfull) { va = mask_chunk<T>(va, kcol, K); vb = mask_chunk<T>(vb, kcol, K); } \
            ra[p] = va; rb[p] = vb;                                              \
        }                                                                        \
    }
#define CIMRGP_SWRITE(buf_)                                                      \
    {                                                                            \
        unsigned char* as_ = smem + (buf_) * 2 * OP_BYTES;                       \
        unsigned char* bs_ = as_ + OP_BYTES;                                     \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                          \
            *reinterpret_cast<uint4*>(as_ + (sr + 32 * p) * LROW + sc * 16) = ra[p]; \
            *reinterpret_cast<uint4*>(bs_ + (sr + 32 * p) * LROW + sc * 16) = rb[p]; \
        }                                                                        \
    }

    acc_t acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = acc_zero<T>();

    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned a_off = (unsigned)((wr * 64 + frow) * LROW + fslot * 8);
    const unsigned b_off = (unsigned)((wc * 64 + frow) * LROW + fslot * 8);

    CIMRGP_GLOAD(0);
    CIMRGP_SWRITE(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more) CIMRGP_GLOAD(kt + 1);
        const unsigned char* as = smem + (kt & 1) * 2 * OP_BYTES;
        const unsigned char* bs = as + OP_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint2 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                a[mi] = *reinterpret_cast<const uint2*>(as + a_off + mi * 16 * LROW + s * 32);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                b[ni] = *reinterpret_cast<const uint2*>(bs + b_off + ni * 16 * LROW + s * 32);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = X::mma(a[mi], b[ni], acc[mi][ni]);
        }
        if (more) CIMRGP_SWRITE((kt + 1) & 1);
        __syncthreads();
    }
#undef CIMRGP_GLOAD
#undef CIMRGP_SWRITE

    // epilogue: C -= acc   (f64 map: 16 lanes x 8 B = one 128-byte line per row)
    const bool diag_tile = LOWER && (ti == tj);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int gc = col0 + wc * 64 + ni * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + wr * 64 + mi * 16 + X::crow(lane, r);
                if (gr < M && gc < N && (!diag_tile || gc <= gr)) {
                    T* p = C + (int64_t)gr * ldc + gc;
                    *p = *p - acc[mi][ni][r];
                }
            }
        }
    }
}

}  // namespace

template <typename T>
int gemm_nt_sub(T* c, int64_t ldc, const T* a, int64_t lda, const T* b, int64_t ldb,
                int64_t m, int64_t n, int k, bool lower, hipStream_t st)
{
    const char* fn = "gemm_nt_sub";
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    CIMRGP_REQUIRE(m < (1ll << 30) && n < (1ll << 30), fn, "matrix too large");
    CIMRGP_REQUIRE(aligned16(a) && aligned16(b), fn, "operand base not 16-byte aligned");
    CIMRGP_REQUIRE(lda % Mx<T>::EPC == 0 && ldb % Mx<T>::EPC == 0, fn, "leading dimension not a multiple of 16 bytes");
    const int64_t tm = (m + GT - 1) / GT, tn = (n + GT - 1) / GT;
    if (lower) {
        CIMRGP_REQUIRE(m == n, fn, "lower update needs a square C");
        const int64_t tiles = tm * (tm + 1) / 2;
        CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
        hipLaunchKernelGGL((k_gemm_nt_sub<T, true>), dim3((unsigned)tiles), dim3(256), 0, st,
                           c, ldc, a, lda, b, ldb, (int)m, (int)n, k, (int)tn);
    } else {
        const int64_t tiles = tm * tn;
        CIMRGP_REQUIRE(tiles < (1ll << 31), fn, "grid too large");
        hipLaunchKernelGGL((k_gemm_nt_sub<T, false>), dim3((unsigned)tiles), dim3(256), 0, st,
                           c, ldc, a, lda, b, ldb, (int)m, (int)n, k, (int)tn);
    }
    CIMRGP_LAUNCH_CHECK(fn);
    return 0;
}

template int gemm_nt_sub<double>(double*, int64_t, const double*, int64_t, const double*, int64_t,
                                 int64_t, int64_t, int, bool, hipStream_t);
template int gemm_nt_sub<float>(float*, int64_t, const float*, int64_t, const float*, int64_t,
                                int64_t, int64_t, int, bool, hipStream_t);

}  // namespace cimrgp
