// Blocked right-looking Cholesky (lower, row-major, in place) and the row-wise
// triangular solve that shares its panel kernels.
//
//   for each panel of CIMRGP_NB = 256 columns:
//     for each 64-column sub-block s of the panel:
//        k_diag64   one workgroup: factor the 64x64 diagonal block in LDS and
//                   form its inverse alongside (one barrier per column)
//        k_trsm64   rows below: X = P * inv(L_ss)^T      (MFMA, K = 64, in place)
//        gemm_nt    remaining panel columns -= X * X_panel^T   (MFMA, K = 64)
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
    hipEventRecord(r.start, st);
    g_recs.push_back(r);
    return &g_recs.back();
}

namespace {

constexpr int SB = 64;   // diagonal sub-block

// ---------------------------------------------------------------------------
// 64x64 diagonal block:  D = L L^T in place (lower), inv slab = L^-1 (lower,
// zero above the diagonal and outside w x w).
// Working copies S (unscaled Schur complement) and Mi (unscaled inverse rows)
// live in LDS; column j of L is S[:,j] * r_j and row j of L^-1 is Mi[j,:] * r_j
// with r_j = 1/sqrt(S[j][j]).  During step j nobody writes column j of S or
// row j of Mi, so one barrier per column is enough.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void k_diag64(T* __restrict__ D, int64_t ld, int w, T* __restrict__ inv, int32_t* info, int col_base)
{
    constexpr int LS = SB + 1;
    __shared__ T S[SB * LS];
    __shared__ T Mi[SB * LS];
    const int tid = threadIdx.x;
    const int i  = tid & 63;      // row owned in the update phase
    const int kg = tid >> 6;      // column group 0..3 (wave-uniform)

    for (int e = tid; e < SB * SB; e += 256) {
        const int r = e >> 6, c = e & 63;
        T v = (r == c) ? (T)1 : (T)0;
        if (r < w && c <= r) v = D[(int64_t)r * ld + c];
        S[r * LS + c]  = v;
        Mi[r * LS + c] = (r == c) ? (T)1 : (T)0;
        inv[e] = (T)0;
    }

    for (int j = 0; j < w; ++j) {
        __syncthreads();
        const T d = S[j * LS + j];
        if (!(d > (T)0)) {
            if (tid == 0) atomicCAS(info, 0, col_base + j + 1);
        }
        const T r  = (T)1 / sqrt(d);
        const T li = S[i * LS + j] * r;          // L[i][j] for i >= j
        if (kg == 0) {
            if (i >= j && i < w) D[(int64_t)i * ld + j] = (i == j) ? d * r : li;
        } else if (kg == 1) {
            if (i <= j) inv[j * SB + i] = Mi[j * LS + i] * r;
        }
        if (i > j) {
            for (int k = j + 1 + kg; k <= i; k += 4)
                S[i * LS + k] -= li * (S[k * LS + j] * r);
            for (int c = kg; c <= j; c += 4)
                Mi[i * LS + c] -= li * (Mi[j * LS + c] * r);
        }
    }
}

// ---------------------------------------------------------------------------
// X = P * invL^T for a 64-column strip P (M rows, kw <= 64 valid columns),
// in place.  invL is a 64x64 lower slab.  One workgroup = 64 rows; each wave
// owns 16 rows x 64 columns (4 MFMA tiles); the whole K = 64 is resident in
// LDS, so every row is read completely before it is overwritten.
// Column tile ct only needs k <= 16 ct + 15 (invL is lower triangular).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void k_trsm64(T* __restrict__ P, int64_t ldp, int M, int kw, const T* __restrict__ invL)
{
    using X = Mx<T>;
    using acc_t = typename X::acc_t;
    constexpr int ROWB = SB * (int)sizeof(T);       // bytes of K per row
    constexpr int LROW = ROWB + 16;                 // padded LDS row stride
    constexpr int CPR  = ROWB / 16;                 // 16-byte chunks per row
    constexpr int NSTEP = ROWB / 32;                // slot-steps (4 slots of 8 B each)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SB * LROW];
    unsigned char* ps = smem;
    unsigned char* ls = smem + SB * LROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * SB;

    for (int e = tid; e < SB * CPR; e += 256) {
        const int r = e / CPR, c = e - r * CPR;
        const int kcol = c * X::EPC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < M && kcol < kw) {
            v = *reinterpret_cast<const uint4*>(P + (int64_t)(row0 + r) * ldp + kcol);
            if (kcol + X::EPC > kw) v = mask_chunk<T>(v, kcol, kw);
        }
        *reinterpret_cast<uint4*>(ps + r * LROW + c * 16) = v;
        *reinterpret_cast<uint4*>(ls + r * LROW + c * 16) =
            *reinterpret_cast<const uint4*>(invL + r * SB + kcol);
    }
    __syncthreads();

    acc_t acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = acc_zero<T>();
    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned char* pa = ps + (wave * 16 + frow) * LROW + fslot * 8;
    const unsigned char* pb = ls + frow * LROW + fslot * 8;
    constexpr int KPS = 32 / (int)sizeof(T);        // k values per slot-step
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const uint2 a = *reinterpret_cast<const uint2*>(pa + s * 32);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (s * KPS <= 16 * ct + 15) {   // compile-time after unrolling
                const uint2 b = *reinterpret_cast<const uint2*>(pb + ct * 16 * LROW + s * 32);
                acc[ct] = X::mma(a, b, acc[ct]);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int gc = ct * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + wave * 16 + X::crow(lane, r);
            if (gr < M && gc < kw) P[(int64_t)gr * ldp + gc] = acc[ct][r];
        }
    }
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
    for (int64_t k0 = 0; k0 < n; k0 += CIMRGP_NB) {
        const int64_t w = (n - k0 < CIMRGP_NB) ? (n - k0) : CIMRGP_NB;
        const int64_t k1 = k0 + w;
        for (int64_t c0 = k0; c0 < k1; c0 += SB) {
            const int sw = (int)((k1 - c0 < SB) ? (k1 - c0) : SB);
            const int64_t pc = c0 + sw;            // first column/row after this sub-block
            T* inv = ws + (c0 / SB) * (SB * SB);
            if (FACTOR) {
                hipLaunchKernelGGL((k_diag64<T>), dim3(1), dim3(256), 0, st,
                                   kmat + c0 * ld + c0, ld, sw, inv, info, (int)c0);
                CIMRGP_LAUNCH_CHECK(fn);
                if (n > pc) {
                    hipLaunchKernelGGL((k_trsm64<T>), dim3((unsigned)((n - pc + SB - 1) / SB)), dim3(256), 0, st,
                                       kmat + pc * ld + c0, ld, (int)(n - pc), sw, (const T*)inv);
                    CIMRGP_LAUNCH_CHECK(fn);
                }
            }
            if (b != nullptr && m > 0) {
                hipLaunchKernelGGL((k_trsm64<T>), dim3((unsigned)((m + SB - 1) / SB)), dim3(256), 0, st,
                                   b + c0, ldb, (int)m, sw, (const T*)inv);
                CIMRGP_LAUNCH_CHECK(fn);
            }
            const int64_t pw = k1 - pc;            // panel columns still to update
            if (pw > 0) {
                if (FACTOR) {
                    int rc = gemm_nt_sub<T>(kmat + pc * ld + pc, ld, kmat + pc * ld + c0, ld,
                                            kmat + pc * ld + c0, ld, n - pc, pw, sw, false, st);
                    if (rc) return rc;
                }
                if (b != nullptr && m > 0) {
                    int rc = gemm_nt_sub<T>(b + pc, ldb, b + c0, ldb, kmat + pc * ld + c0, ld,
                                            m, pw, sw, false, st);
                    if (rc) return rc;
                }
            }
        }
        if (n > k1) {
            if (FACTOR) {
                const double mm = (double)(n - k1);
                TrailRec* rec = rec_open(st, mm * (mm + 1.0) * (double)w);   // lower SYRK: M(M+1)K flop
                int rc = gemm_nt_sub<T>(kmat + k1 * ld + k1, ld, kmat + k1 * ld + k0, ld,
                                        kmat + k1 * ld + k0, ld, n - k1, n - k1, (int)w, true, st);
                if (rec) hipEventRecord(rec->stop, st);
                if (rc) return rc;
            }
            if (b != nullptr && m > 0) {
                int rc = gemm_nt_sub<T>(b + k1, ldb, b + k0, ldb, kmat + k1 * ld + k0, ld,
                                        m, n - k1, (int)w, false, st);
                if (rc) return rc;
            }
        }
    }
    return 0;
}

template <typename T>
int potrf_run(T* k, int64_t n, int64_t ld, T* ws, int32_t* info, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return check_hip(e, "cimrgp_potrf", "hipMemsetAsync(info)");
    return panel_sweep<T, true>(k, n, ld, ws, info, nullptr, 0, 0, st);
}

template <typename T>
int solve_rows_run(const T* l, int64_t n, int64_t ld, const T* ws, T* b, int64_t m, int64_t ldb, hipStream_t st)
{
    return panel_sweep<T, false>(const_cast<T*>(l), n, ld, const_cast<T*>(ws), nullptr, b, m, ldb, st);
}

template int potrf_run<double>(double*, int64_t, int64_t, double*, int32_t*, hipStream_t);
template int potrf_run<float>(float*, int64_t, int64_t, float*, int32_t*, hipStream_t);
template int solve_rows_run<double>(const double*, int64_t, int64_t, const double*, double*, int64_t, int64_t, hipStream_t);
template int solve_rows_run<float>(const float*, int64_t, int64_t, const float*, float*, int64_t, int64_t, hipStream_t);

}  // namespace cimrgp
