// Diagnostic: phase stamps (s_memtime) of the panel kernels.  Not part of the product.
#define CIMRGP_STAMP 1
#include "../cimrgp_amd/csrc/potrf.hip"
#include "../cimrgp_amd/csrc/gemm_nt.hip"
#include "../cimrgp_amd/csrc/solve.hip"
#include "../cimrgp_amd/csrc/api.hip"
#include <vector>
#include <cmath>
using namespace cimrgp;
namespace cimrgp {
template <typename T> int rbf_gram_run(const T*, int64_t, const T*, int64_t, int, double, double, double, T*, int64_t, bool, bool, hipStream_t) { return 0; }
template <typename T> int predict_mean_run(const T*, int64_t, int, const T*, int, const T*, int64_t, double, double, const T*, T*, int, hipStream_t) { return 0; }
template <typename T> int misc_block_stats(const T*, const T*, int64_t, int, T*, hipStream_t) { return 0; }
template <typename T> int misc_residual(const T*, const T*, const T*, int64_t, int, T*, hipStream_t) { return 0; }
template <typename T> int misc_train_mean(const T*, const T*, const T*, const T*, int64_t, int, T*, int, hipStream_t) { return 0; }
template <typename T> int misc_add_diag(T*, int64_t, int64_t, const T*, hipStream_t) { return 0; }
template <typename T> int misc_noise_from_stats(const T*, int, double, double, T*, hipStream_t) { return 0; }
template <typename T> int misc_logdet_half(const T*, int64_t, int64_t, double*, hipStream_t) { return 0; }
template <typename T> int laplace_basis_run(const T*, int64_t, int, const double*, int, T*, hipStream_t) { return 0; }
int basis_moments_workgroups(int64_t) { return 0; }
template <typename T> int basis_moments_run(const T*, int64_t, int, const double*, int, const T*, const T*, const T*, const double*, int, double*, double*, hipStream_t) { return 0; }
template <typename T> int basis_apply_run(const T*, int64_t, int, const double*, int, const double*, int, const double*, const double*, double, T*, T*, int, hipStream_t) { return 0; }
template <typename T> int lml_grad_run(const T*, int64_t, int, const T*, int64_t, const T*, int, double, double, double, double*, double*, hipStream_t) { return 0; }
}
int main()
{
    const int n = 2048, ld = 2048;
    std::vector<double> h((size_t)n * ld, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) h[(size_t)i * ld + j] = std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 0.01 : 0.0);
    double *dK, *dws; int32_t* dinfo;
    hipMalloc(&dK, h.size() * 8); hipMalloc(&dws, (n / 64) * 64 * 64 * 8); hipMalloc(&dinfo, 4);
    hipMemcpy(dK, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
    long long st[32];
    for (int kprev = 0; kprev <= 192; kprev += 64) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dK + (size_t)256 * ld + 256, (int64_t)ld, 64, dK + (size_t)256 * ld + 256 - kprev, kprev, dws, dinfo, 0);
            hipDeviceSynchronize();
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
            if (rep && kprev == 0) printf("  pivot wave block 2: read %lld  self-update %lld  pivots %lld  publish %lld  barrier %lld  | iteration %lld ;  tile wave 3: wait %lld  update+gather %lld\n", st[17]-st[16], st[18]-st[17], st[19]-st[18], st[20]-st[19], st[21]-st[20], st[24]-st[16], st[13]-st[12], st[14]-st[13]);
            if (rep) printf("diag64 kprev=%3d: load %lld  mfma %lld  loop %lld  store %lld  total %lld ticks (%.1f us @2.35GHz)\n", kprev, st[1]-st[0], st[2]-st[1], st[3]-st[2], st[4]-st[3], st[4]-st[0], (st[4]-st[0])/2350.0);
        }
    }
    {   // back-to-back diag64 + trsm64 pairs on one stream, nothing else running: wall per pair
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 64; ++it) {
                hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dK + (size_t)256 * ld + 256, (int64_t)ld, 64, dK + (size_t)256 * ld + 256 - 128, 128, dws, dinfo, 0);
                hipLaunchKernelGGL((k_trsm64<double>), dim3(56), dim3(256), 0, 0, dK + (size_t)320 * ld + 256, (int64_t)ld, 1728, 56, (double*)nullptr, (int64_t)0, 0, 64, 128, dK + (size_t)256 * ld + 128, (int64_t)ld, dws);
            }
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("64 x (diag64 kprev=128 + trsm64 kprev=128, 56 WGs): %.1f us per pair\n", ms * 1e3 / 64);
        }
        hipEventRecord(e0);
        for (int it = 0; it < 64; ++it)
            hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dK + (size_t)256 * ld + 256, (int64_t)ld, 64, dK + (size_t)256 * ld + 256, 0, dws, dinfo, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("64 x diag64 kprev=0 alone: %.1f us each\n", ms * 1e3 / 64);
    }
    for (int kprev = 0; kprev <= 192; kprev += 64) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_trsm64<double>), dim3(32), dim3(256), 0, 0, dK + (size_t)512 * ld + 256, (int64_t)ld, 1024, 16, (double*)nullptr, (int64_t)0, 0, 64, kprev, dK + (size_t)256 * ld + 256 - kprev, (int64_t)ld, dws);
            hipDeviceSynchronize();
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
            if (rep) printf("trsm64 kprev=%3d: kloop %lld  stage %lld  mma %lld  store %lld  total %lld ticks (%.1f us)\n", kprev, st[9]-st[8], st[10]-st[9], st[11]-st[10], st[12]-st[11], st[12]-st[8], (st[12]-st[8])/2350.0);
        }
    }
    {   // skinny forward panel step on an 8192 matrix (values irrelevant)
        const int nn = 8192; const int64_t l2 = 8192;
        double *dL, *dws2, *dwork, *dout;
        hipMalloc(&dL, (size_t)nn * l2 * 8); hipMemset(dL, 0, (size_t)nn * l2 * 8);
        hipMalloc(&dws2, (size_t)(128 * 4096 + 32 * 65536) * 8); hipMemset(dws2, 0, (size_t)(128 * 4096 + 32 * 65536) * 8);
        hipMalloc(&dwork, 2 * nn * 8); hipMalloc(&dout, 2 * nn * 8); hipMemset(dwork, 0, 2 * nn * 8);
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_fwd_panel<double, 2>), dim3(124), dim3(1024), 0, 0, dL, l2, nn, dws2 + 128 * 4096, dwork, dout, 2, 0, 256);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
            printf("fwd_panel: load %lld  invapply %lld  reduce %lld  update %lld  total %lld ticks (%.1f us); kernel %.1f us\n",
                   st[21]-st[20], st[22]-st[21], st[23]-st[22], st[24]-st[23], st[24]-st[20], (st[24]-st[20])/2350.0, ms*1e3);
        }
    }
    // trailing-update kernel: one workgroup alone, then a full grid
    {
        const int nn = 8192; const int64_t l2 = 8192;
        double* dA; hipMalloc(&dA, (size_t)nn * l2 * 8); hipMemset(dA, 0, (size_t)nn * l2 * 8);
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int M = (cfg == 0) ? 128 : (cfg == 1 ? 2048 : 7936);
            const bool lower = cfg != 3;
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                gemm_nt_sub<double>(dA + 256 * l2 + 256, l2, dA + 256 * l2, l2, dA + 256 * l2, l2, M, lower ? M : 256, 256, lower, 0);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
                if (rep) printf("gemm M=%d %s: prologue %lld  kloop %lld (%.0f/ktile)  epilogue %lld  total %lld ticks (%.1f us); kernel %.1f us\n", M, lower ? "lower" : "rect N=256",
                                st[17]-st[16], st[18]-st[17], (st[18]-st[17])/16.0, st[19]-st[18], st[19]-st[16], (st[19]-st[16])/2350.0, ms*1e3);
            }
        }
    }
    {   // deeper K per pass: lower update of M = 7424 with K = 256, 512, 768 (C traffic per flop halves / thirds)
        const int nn = 8192; const int64_t l2 = 8192;
        double* dA; hipMalloc(&dA, (size_t)nn * l2 * 8); hipMemset(dA, 0, (size_t)nn * l2 * 8);
        for (int K : {256, 512, 768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                gemm_nt_sub<double>(dA + 768 * l2 + 768, l2, dA + 768 * l2, l2, dA + 768 * l2, l2, 7424, 7424, K, true, 0);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("lower gemm M=7424 K=%d: %.1f us  (%.1f TF/s)\n", K, best * 1e3, 7424.0 * 7425.0 * K / (best * 1e-3) / 1e12);
        }
        hipFree(dA);
    }
    {   // the same at M = 15360 and 32256 (2 GB / 8.5 GB matrices)
        for (int nn : {16384, 33280}) {
            const int64_t l2 = nn + 16;
            double* dA; if (hipMalloc(&dA, (size_t)nn * l2 * 8) != hipSuccess) continue;
            hipMemset(dA, 0, (size_t)nn * l2 * 8);
            const int M = nn - 1024;
            for (int K : {256, 512, 768, 1024}) {
                float best = 1e9f;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0);
                    gemm_nt_sub<double>(dA + 1024 * l2 + 1024, l2, dA + 1024 * l2, l2, dA + 1024 * l2, l2, M, M, K, true, 0);
                    hipEventRecord(e1); hipDeviceSynchronize();
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("lower gemm M=%d K=%d: %.1f us  (%.1f TF/s)\n", M, K, best * 1e3, (double)M * (M + 1.0) * K / (best * 1e-3) / 1e12);
            }
            hipFree(dA);
        }
    }
    {   // rectangular update of carried rows: M = 2048 rows x N columns, K = 256
        const int nn = 8192; const int64_t l2 = 8208;
        double *dA, *dB2; hipMalloc(&dA, (size_t)nn * l2 * 8); hipMemset(dA, 0, (size_t)nn * l2 * 8);
        hipMalloc(&dB2, (size_t)2048 * l2 * 8); hipMemset(dB2, 0, (size_t)2048 * l2 * 8);
        for (int N : {7936, 4096, 2048}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                gemm_nt_sub<double>(dB2 + 256, l2, dB2, l2, dA + 256 * l2, l2, 2048, N, 256, false, 0);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("rect gemm M=2048 N=%d: %.1f us  (%.1f TF/s)\n", N, best * 1e3, 2.0 * 2048 * N * 256 / (best * 1e-3) / 1e12);
        }
    }
    {   // CU-masked streams: does the trailing update keep its rate on a subset of the CUs?
        const int nn = 8192; const int64_t l2 = 8192;
        double* dA; hipMalloc(&dA, (size_t)nn * l2 * 8); hipMemset(dA, 0, (size_t)nn * l2 * 8);
        for (int pat = 0; pat < 5; ++pat) {
            uint32_t mask[8]; for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
            const char* name = "all CUs";
            if (pat == 1) { mask[0] = 0; name = "bits 0-31 off"; }
            if (pat == 2) { for (int i = 0; i < 8; ++i) mask[i] = 0xfffffffeu; name = "bit 0 of every word off"; }
            if (pat == 3) { for (int i = 0; i < 8; ++i) mask[i] = 0xfffefffeu; name = "bits 0,16 of every word off"; }
            if (pat == 4) { mask[0] = 0; mask[1] = 0; name = "bits 0-63 off"; }
            hipStream_t sm;
            if (hipExtStreamCreateWithCUMask(&sm, 8, mask) != hipSuccess) { printf("mask stream failed\n"); continue; }
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, sm);
                gemm_nt_sub<double>(dA + 256 * l2 + 256, l2, dA + 256 * l2, l2, dA + 256 * l2, l2, 7936, 7936, 256, true, sm);
                hipEventRecord(e1, sm); hipStreamSynchronize(sm);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("masked gemm M=7936 [%s]: %.1f us\n", name, best * 1e3);
            hipStreamDestroy(sm);
        }
    }
    {   // diag64 next to a running trailing update: is it the launch or the execution that stretches?
        const int nn = 8192; const int64_t l2 = 8192;
        double* dA; hipMalloc(&dA, (size_t)nn * l2 * 8); hipMemset(dA, 0, (size_t)nn * l2 * 8);
        for (int reserve = 0; reserve <= 2; ++reserve) {
            uint32_t mask[8]; for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
            for (int bit = 0; bit < 8 * reserve; ++bit) mask[bit >> 5] &= ~(1u << (bit & 31));
            hipStream_t sg, sc; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
            hipExtStreamCreateWithCUMask(&sg, 8, mask);
            hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi);
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int gk = 0; gk < 3; ++gk)
                    gemm_nt_sub<double>(dA + 256 * l2 + 256, l2, dA + 256 * l2, l2, dA + 256 * l2, l2, 7936, 7936, 256, true, sg);
                // let the update get going, then the chain kernel
                for (volatile int spin = 0; spin < 200000; ++spin) {}
                hipEventRecord(e0, sc);
                hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, sc, dK + (size_t)256 * ld + 256, (int64_t)ld, 64, dK + (size_t)256 * ld + 256, 0, dws, dinfo, 0);
                hipEventRecord(e1, sc);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
                if (rep) printf("reserve %d CU/XCD: diag64 beside the update: events %.1f us, inside the kernel %.1f us (load %lld mfma %lld loop %lld store %lld)\n",
                                reserve, ms * 1e3, (st[4] - st[0]) / 2350.0, st[1]-st[0], st[2]-st[1], st[3]-st[2], st[4]-st[3]);
            }
            hipStreamDestroy(sg); hipStreamDestroy(sc);
        }
    }
    return 0;
}
