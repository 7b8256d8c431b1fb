// Lab: where does an FP64 MFMA K loop lose its rate?  One 64x64 wave tile (4 x 4 MFMA tiles of 16x16x4) per wave,
// variants that add the pieces of a GEMM K loop one at a time.  Results are meaningless numbers; only times matter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/mfma_loop.hip -o tools/lab/mfma_loop && tools/lab/mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double acc_t __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LROW = 144;                       // 128-byte chunk row + 16 bytes of padding

static __device__ __forceinline__ double slot(const v4u* v, int s)
{
    const unsigned lo = (s == 0) ? v[0].x : (s == 1) ? v[0].z : (s == 2) ? v[1].x : v[1].z;
    const unsigned hi = (s == 0) ? v[0].y : (s == 1) ? v[0].w : (s == 2) ? v[1].y : v[1].w;
    return __hiloint2double((int)hi, (int)lo);
}

// V: 0 registers only; 1 + LDS fragment reads (b128, both operands); 2 + LDS writes and a barrier per chunk (double buffer);
//    3 + global loads of both operands (the classic loop); 4 A straight from global memory to registers, B through LDS;
//    5 like 3 with 8-byte fragment reads (ds_read_b64, the k order of the shipped kernels); 6 = 2 without the barrier (timing only);
//    7 = 1 with a barrier per chunk and no LDS writes
template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void k_loop(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ out, int chunks, int lda)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * 128 * LROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (WAVES == 4) ? (wave >> 1) : (wave >> 2), wc = (WAVES == 4) ? (wave & 1) : (wave & 3);
    constexpr int MI = 4, NI = (WAVES == 4) ? 4 : 2;         // 64 x 64 or 64 x 32 per wave
    const int fn = lane & 15, fq = lane >> 4;
    acc_t acc[MI][NI];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    // fill LDS once (variants 1+ read it)
    for (int e = tid; e < 2 * 2 * 128 * LROW / 16; e += WAVES * 64) reinterpret_cast<v4u*>(smem)[e] = v4u{0x3ff00000u + e, 1u, 0x3ff00000u, 2u};
    __syncthreads();
    const unsigned char* abase0 = smem + (wr * 64 + fn) * LROW + fq * 32;
    const unsigned char* bbase0 = smem + 128 * LROW + (wc * 16 * NI + fn) * LROW + fq * 32;
    // staging map of the classic loop: 8 threads per 128-byte row, NPASS passes of (threads / 8) rows
    constexpr int RPP = WAVES * 64 / 8, NPASS = 128 / RPP;
    const int sc = tid & 7, sr = tid >> 3;
    const double* ag = A + (size_t)(blockIdx.x % 32) * 128 * lda;      // a few row blocks, cache resident
    const double* bg = B + (size_t)(blockIdx.x % 32) * 128 * lda;
    v4u ra[NPASS], rb[NPASS];
    v4u da[3][MI][2];                                                   // variant 4: A chunks in flight
    if (V == 3 || V == 5) for (int p = 0; p < NPASS; ++p) { ra[p] = *reinterpret_cast<const v4u*>(ag + (size_t)(sr + RPP * p) * lda + sc * 2); rb[p] = *reinterpret_cast<const v4u*>(bg + (size_t)(sr + RPP * p) * lda + sc * 2); }
    if (V == 4) {
        for (int p = 0; p < NPASS; ++p) rb[p] = *reinterpret_cast<const v4u*>(bg + (size_t)(sr + RPP * p) * lda + sc * 2);
        for (int c = 0; c < 2; ++c) for (int i = 0; i < MI; ++i) {
            const double* q = ag + (size_t)(wr * 64 + 16 * i + fn) * lda + c * 16 + fq * 4;
            da[c][i][0] = *reinterpret_cast<const v4u*>(q); da[c][i][1] = *reinterpret_cast<const v4u*>(q + 2);
        }
    }
    v4u rdummy = v4u{1u, 2u, 3u, 4u};
#pragma unroll 1
    for (int c3 = 0; c3 < chunks; c3 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = c3 + u;
            const int buf = c & 1;
            const unsigned char* abase = (V >= 2) ? abase0 + buf * 2 * 128 * LROW : abase0;
            const unsigned char* bbase = (V >= 2) ? bbase0 + buf * 2 * 128 * LROW : bbase0;
            if (V >= 2 && V != 7) {
                // next chunk into the other buffer
                unsigned char* as = smem + (buf ^ 1) * 2 * 128 * LROW, *bs = as + 128 * LROW;
                for (int p = 0; p < NPASS; ++p) {
                    if (V != 4) *reinterpret_cast<v4u*>(as + (sr + RPP * p) * LROW + sc * 16) = (V == 2 || V == 6) ? rdummy : ra[p];
                    *reinterpret_cast<v4u*>(bs + (sr + RPP * p) * LROW + sc * 16) = (V == 2 || V == 6) ? rdummy : rb[p];
                }
            }
            if (V == 3 || V == 5) for (int p = 0; p < NPASS; ++p) {
                ra[p] = *reinterpret_cast<const v4u*>(ag + (size_t)(sr + RPP * p) * lda + ((c + 2) & 15) * 16 + sc * 2);
                rb[p] = *reinterpret_cast<const v4u*>(bg + (size_t)(sr + RPP * p) * lda + ((c + 2) & 15) * 16 + sc * 2);
            }
            if (V == 4) {
                for (int p = 0; p < NPASS; ++p) rb[p] = *reinterpret_cast<const v4u*>(bg + (size_t)(sr + RPP * p) * lda + ((c + 2) & 15) * 16 + sc * 2);
                for (int i = 0; i < MI; ++i) {
                    const double* q = ag + (size_t)(wr * 64 + 16 * i + fn) * lda + ((c + 2) & 15) * 16 + fq * 4;
                    da[(u + 2) % 3][i][0] = *reinterpret_cast<const v4u*>(q); da[(u + 2) % 3][i][1] = *reinterpret_cast<const v4u*>(q + 2);
                }
            }
            if (V == 5) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    double a[MI], b[NI];
                    for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const double*>(abase - fq * 32 + fq * 8 + i * 16 * LROW + s * 32);
                    for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const double*>(bbase - fq * 32 + fq * 8 + j * 16 * LROW + s * 32);
                    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            } else {
                v4u fa[MI][2], fb[NI][2];
                if (V == 0) { for (int i = 0; i < MI; ++i) { fa[i][0] = rdummy; fa[i][1] = rdummy; } for (int j = 0; j < NI; ++j) { fb[j][0] = rdummy; fb[j][1] = rdummy; } }
                else {
                    for (int i = 0; i < MI; ++i) {
                        if (V == 4) { fa[i][0] = da[u][i][0]; fa[i][1] = da[u][i][1]; }
                        else { fa[i][0] = *reinterpret_cast<const v4u*>(abase + i * 16 * LROW); fa[i][1] = *reinterpret_cast<const v4u*>(abase + i * 16 * LROW + 16); }
                    }
                    for (int j = 0; j < NI; ++j) { fb[j][0] = *reinterpret_cast<const v4u*>(bbase + j * 16 * LROW); fb[j][1] = *reinterpret_cast<const v4u*>(bbase + j * 16 * LROW + 16); }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(slot(fa[i], s), slot(fb[j], s), acc[i][j], 0, 0, 0);
            }
            if (V >= 2 && V != 6) __syncthreads();
            if (V == 0) asm volatile("" : "+v"(rdummy));
        }
    }
    double sum = 0;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = sum;
}

template <int V, int WAVES>
static void run(const char* what, const double* A, const double* B, double* out, int grid, int chunks)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loop<V, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, A, B, out, chunks, 256);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_loop<V, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, A, B, out, chunks, 256);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // flop: per chunk and wave MI x NI tiles x 4 k-steps x 16*16*4*2
    const double mi_ni = (WAVES == 4) ? 16.0 : 8.0;
    const double flop = (double)grid * WAVES * chunks * mi_ni * 4 * 2048.0;
    printf("%-78s grid %4d x %d waves: %8.3f ms  %6.2f TF/s\n", what, grid, WAVES, best, flop / best / 1e9);
}

// The classic loop again, parametrised: WAVES x (64 x 16 NI) wave tiles (WR x WC waves), DEPTH chunks of 16 k per barrier
// (LDS rows of 128 DEPTH bytes + 16).  WG tile = 64 WR x 16 NI WC.
template <int WR, int WC, int NI, int DEPTH>
__global__ __launch_bounds__(WR * WC * 64)
void k_classic(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ out, int chunks, int lda)
{
    constexpr int WAVES = WR * WC, NT = WAVES * 64, MI = 4;
    constexpr int ROWS_A = 64 * WR, ROWS_B = 16 * NI * WC, LR = 128 * DEPTH + 16;
    constexpr int BUF = (ROWS_A + ROWS_B) * LR;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int fn = lane & 15, fq = lane >> 4;
    acc_t acc[MI][NI];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    for (int e = tid; e < 2 * BUF / 16; e += NT) reinterpret_cast<v4u*>(smem)[e] = v4u{0x3ff00000u + e, 1u, 0x3ff00000u, 2u};
    __syncthreads();
    // staging: 8 DEPTH threads per row, NT / (8 DEPTH) rows per pass
    constexpr int TPR = 8 * DEPTH, RPP = NT / TPR, PA = ROWS_A / RPP, PB = (ROWS_B + RPP - 1) / RPP;
    const int sc = tid % TPR, sr = tid / TPR;
    const double* ag = A + (size_t)(blockIdx.x % 32) * 128 * lda;
    const double* bg = B + (size_t)(blockIdx.x % 32) * 128 * lda;
    v4u ra[PA], rb[PB];
    for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const v4u*>(ag + (size_t)((sr + RPP * p) % 128) * lda + sc * 2);
    for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const v4u*>(bg + (size_t)((sr + RPP * p) % 128) * lda + sc * 2);
    const int steps = chunks / DEPTH;
#pragma unroll 1
    for (int c = 0; c < steps; ++c) {
        const int buf = c & 1;
        unsigned char* as = smem + (buf ^ 1) * BUF, *bs = as + ROWS_A * LR;
        for (int p = 0; p < PA; ++p) *reinterpret_cast<v4u*>(as + (sr + RPP * p) * LR + sc * 16) = ra[p];
        for (int p = 0; p < PB; ++p) if (ROWS_B % RPP == 0 || sr + RPP * p < ROWS_B) *reinterpret_cast<v4u*>(bs + (sr + RPP * p) * LR + sc * 16) = rb[p];
        const int kc = ((c + 2) * DEPTH) & 15;
        for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const v4u*>(ag + (size_t)((sr + RPP * p) % 128) * lda + (kc * 16 + sc * 2) % 256);
        for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const v4u*>(bg + (size_t)((sr + RPP * p) % 128) * lda + (kc * 16 + sc * 2) % 256);
        const unsigned char* abase = smem + buf * BUF + (wr * 64 + fn) * LR + fq * 32;
        const unsigned char* bbase = smem + buf * BUF + ROWS_A * LR + (wc * 16 * NI + fn) * LR + fq * 32;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v4u fa[MI][2], fb[NI][2];
            for (int i = 0; i < MI; ++i) { fa[i][0] = *reinterpret_cast<const v4u*>(abase + i * 16 * LR + d * 128); fa[i][1] = *reinterpret_cast<const v4u*>(abase + i * 16 * LR + d * 128 + 16); }
            for (int j = 0; j < NI; ++j) { fb[j][0] = *reinterpret_cast<const v4u*>(bbase + j * 16 * LR + d * 128); fb[j][1] = *reinterpret_cast<const v4u*>(bbase + j * 16 * LR + d * 128 + 16); }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(slot(fa[i], s), slot(fb[j], s), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    double sum = 0;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * NT + tid] = sum;
}

template <int WR, int WC, int NI, int DEPTH>
static void run_classic(const char* what, const double* A, const double* B, double* out, int grid, int chunks)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_classic<WR, WC, NI, DEPTH>), dim3(grid), dim3(WR * WC * 64), 0, 0, A, B, out, chunks, 256);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_classic<WR, WC, NI, DEPTH>), dim3(grid), dim3(WR * WC * 64), 0, 0, A, B, out, chunks, 256);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double flop = (double)grid * WR * WC * chunks * (4.0 * NI) * 4 * 2048.0;
    printf("%-78s grid %4d x %d waves: %8.3f ms  %6.2f TF/s\n", what, grid, WR * WC, best, flop / best / 1e9);
}

// FP32 register-only loops: does the 16x16x4 form (32 cycles each) issue back to back from two waves per SIMD?  And the
// 32x32x2 form (64 cycles, the same flops per cycle)?
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int FORM, int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void k_f32(float* __restrict__ out, int iters)
{
    const int tid = threadIdx.x;
    float a = 1.0f + tid * 1e-6f, b = 1.0f - tid * 1e-6f;
    float sum = 0;
    if (FORM == 0) {
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            asm volatile("" : "+v"(a));
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) sum += acc[i][r];
    } else {
        f16v acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            asm volatile("" : "+v"(a));
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    }
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = sum;
}

template <int FORM, int WAVES>
static void run_f32(const char* what, float* out, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_f32<FORM, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_f32<FORM, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // per iteration and wave: FORM 0: 32 x (16*16*4*2) flop, FORM 1: 16 x (32*32*2*2)
    const double flop = 256.0 * WAVES * iters * (FORM == 0 ? 32 * 2048.0 : 16 * 4096.0);
    printf("%-78s grid  256 x %d waves: %8.3f ms  %6.2f TF/s\n", what, WAVES, best, flop / best / 1e9);
}

// The classic eight-wave loop with the operands loaded STRAIGHT INTO LDS (global_load_lds_dwordx4: no staging registers, no
// ds_write): unpadded 128-byte rows, 16-byte pieces XOR-swizzled by (row >> 1) & 7 so that the 8-byte fragment reads stay
// conflict-free; three stage buffers (two stages of loads in flight).
__global__ __launch_bounds__(512)
void k_ldsdma(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ out, int chunks, int lda)
{
    constexpr int STAGE = 256 * 128;                      // [A rows 0..127 ; B rows 0..127] x 128 bytes
    __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fslot = lane >> 4;
    acc_t acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    const double* ag = A + (size_t)(blockIdx.x % 32) * 128 * lda;
    const double* bg = B + (size_t)(blockIdx.x % 32) * 128 * lda;
    // this lane's four loads of a stage: LDS kilobyte (wave * 4 + q), row 8 (wave * 4 + q) + (lane >> 3), physical piece lane & 7
    const double* gsrc[4];
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * (wave * 4 + q) + (lane >> 3), pp = lane & 7;
        const int p = pp ^ ((r >> 1) & 7);
        gsrc[q] = (r < 128 ? ag + (size_t)r * lda : bg + (size_t)(r - 128) * lda) + p * 2;
    }
    auto load_stage = [&](int buf, int kcol) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(gsrc[q] + kcol), (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (wave * 4 + q) * 1024), 16, 0, 0);
    };
    // fragment offsets inside a row for the four k-steps of a stage
    unsigned off[4];
    const int key = (frow >> 1) & 7;
    for (int sx = 0; sx < 4; ++sx) off[sx] = (unsigned)((((2 * sx + (fslot >> 1)) ^ key) * 16) + (fslot & 1) * 8);
    const unsigned arow = (unsigned)((wr * 64 + frow) * 128), brow = (unsigned)((128 + wc * 32 + frow) * 128);
    load_stage(0, 0);
    load_stage(1, 16);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int c3 = 0; c3 < chunks; c3 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = c3 + u;
            load_stage((u + 2) % 3, ((c + 2) & 15) * 16);
            const unsigned char* sb = smem + u * STAGE;
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                double a[4], b[2];
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const double*>(sb + arow + i * 16 * 128 + off[sx]);
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const double*>(sb + brow + j * 16 * 128 + off[sx]);
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // the stage after this one has landed (its successor may still fly)
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double sum = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
}

static void run_ldsdma(const double* A, const double* B, double* out, int chunks)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_ldsdma, dim3(256), dim3(512), 0, 0, A, B, out, chunks, 256);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_ldsdma, dim3(256), dim3(512), 0, 0, A, B, out, chunks, 256);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double flop = 256.0 * 8 * chunks * 8.0 * 4 * 2048.0;
    printf("%-78s grid  256 x 8 waves: %8.3f ms  %6.2f TF/s\n", "8 classic loop, operands straight into LDS (global_load_lds_dwordx4), 3 buffers", best, flop / best / 1e9);
}

int main()
{
    double *A, *B, *out;
    CHECK(hipMalloc(&A, 32 * 128 * 256 * 8)); CHECK(hipMalloc(&B, 32 * 128 * 256 * 8)); CHECK(hipMalloc(&out, 1024 * 512 * 8));
    std::vector<double> h(32 * 128 * 256, 1.0);
    CHECK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const int chunks = 3 * 1024;
    run<0, 4>("0 registers only, 4 waves of 64x64 (1 per SIMD)", A, B, out, 256, chunks);
    run<0, 4>("0 registers only, 2 workgroups per CU", A, B, out, 512, chunks);
    run<0, 8>("0 registers only, 8 waves of 64x32", A, B, out, 256, chunks);
    run<1, 4>("1 + LDS fragment reads (b128)", A, B, out, 256, chunks);
    run<1, 4>("1 ... 2 workgroups per CU", A, B, out, 512, chunks);
    run<1, 8>("1 ... 8 waves of 64x32", A, B, out, 256, chunks);
    run<2, 4>("2 + LDS writes and a barrier per chunk", A, B, out, 256, chunks);
    run<2, 4>("2 ... 2 workgroups per CU", A, B, out, 512, chunks);
    run<2, 8>("2 ... 8 waves of 64x32", A, B, out, 256, chunks);
    run<3, 4>("3 + global loads of both operands (classic loop)", A, B, out, 256, chunks);
    run<3, 4>("3 ... 2 workgroups per CU", A, B, out, 512, chunks);
    run<3, 8>("3 ... 8 waves of 64x32", A, B, out, 256, chunks);
    run<5, 4>("5 classic loop with 8-byte fragment reads (shipped k order)", A, B, out, 256, chunks);
    run<5, 4>("5 ... 2 workgroups per CU", A, B, out, 512, chunks);
    run<5, 8>("5 ... 8 waves of 64x32", A, B, out, 256, chunks);
    run<4, 4>("4 A straight from global memory, B through LDS", A, B, out, 256, chunks);
    run<4, 4>("4 ... 2 workgroups per CU", A, B, out, 512, chunks);
    run<4, 8>("4 ... 8 waves of 64x32: A straight from global memory, B through LDS", A, B, out, 256, chunks);
    run_ldsdma(A, B, out, chunks);
    run<6, 8>("6 LDS writes, NO barrier (timing only), 8 waves of 64x32", A, B, out, 256, chunks);
    run<7, 8>("7 barrier per chunk, NO LDS writes, 8 waves of 64x32", A, B, out, 256, chunks);
    run_classic<2, 4, 2, 1>("classic: 8 waves of 64x32 (128x128 tile), 1 chunk per barrier", A, B, out, 256, chunks);
    run_classic<2, 4, 2, 2>("classic: 8 waves of 64x32 (128x128 tile), 2 chunks per barrier", A, B, out, 256, chunks);
    run_classic<2, 2, 2, 1>("classic: 4 waves of 64x32 (128x64 tile), 2 workgroups per CU", A, B, out, 512, chunks);
    run_classic<2, 2, 2, 2>("classic: 4 waves of 64x32 (128x64 tile), 2 per CU, 2 chunks per barrier", A, B, out, 512, chunks);
    run_classic<1, 2, 2, 1>("classic: 2 waves of 64x32 (64x64 tile), 4 workgroups per CU", A, B, out, 1024, chunks);
    run_classic<2, 2, 4, 1>("classic: 4 waves of 64x64 (128x128 tile)", A, B, out, 256, chunks);
    run_classic<2, 2, 4, 2>("classic: 4 waves of 64x64 (128x128 tile), 2 chunks per barrier", A, B, out, 256, chunks);
    run_f32<0, 4>("FP32 registers only, v_mfma_f32_16x16x4, 4 waves (1 per SIMD)", (float*)out, 20000);
    run_f32<0, 8>("FP32 registers only, v_mfma_f32_16x16x4, 8 waves (2 per SIMD)", (float*)out, 20000);
    run_f32<0, 16>("FP32 registers only, v_mfma_f32_16x16x4, 16 waves (4 per SIMD)", (float*)out, 20000);
    run_f32<1, 4>("FP32 registers only, v_mfma_f32_32x32x2, 4 waves (1 per SIMD)", (float*)out, 20000);
    run_f32<1, 8>("FP32 registers only, v_mfma_f32_32x32x2, 8 waves (2 per SIMD)", (float*)out, 20000);
    return 0;
}
