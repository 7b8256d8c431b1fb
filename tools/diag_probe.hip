// Diagnostic: phase stamps (s_memtime: shader clocks, ~2.35 GHz under load) of the diagonal kernel and
// whole one-queue factorisations per chain form.
// Not part of the product.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag_probe.hip -o tools/diag_probe
#ifndef NOSTAMP
#define CIMRGP_STAMP 1
#endif
#include "../cimrgp_amd/csrc/potrf.hip"
#include "../cimrgp_amd/csrc/gemm_nt.hip"
#include <cmath>
#include <cstring>
#include <vector>
using namespace cimrgp;
namespace cimrgp {
static std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const char* fn, const char* what) { fprintf(stderr, "%s: %s\n", fn, what); return -1; }
int check_hip(hipError_t e, const char* fn, const char* what) { fprintf(stderr, "%s: %s: %s\n", fn, what, hipGetErrorString(e)); return -2; }
const Knobs& knobs() { static Knobs k; return k; }
int rows_queues() { return 2; }
}

// which SIMD each wave of a workgroup lands on (HW_REG_HW_ID: SIMD_ID = bits 5:4)
__global__ void k_simd_ids(int* out)
{
    const int hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // 16 bits of HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}

static double tick_us(long long t) { return t / 2350.0; }     // shader clocks at ~2.35 GHz

int main()
{
    const int n = 4096, ld = 4096;
    std::vector<double> h((size_t)n * ld, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) h[(size_t)i * ld + j] = std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 0.01 : 0.0);
    double *dK, *dws;
    int32_t* dinfo;
    const size_t wsn = (size_t)(n / 64) * 64 * 64 + (size_t)(n / 256) * 256 * 256;
    hipMalloc(&dK, h.size() * 8); hipMalloc(&dws, wsn * 8); hipMalloc(&dinfo, 4);
    long long st[32];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // one stamped diagonal kernel (first sub-block of a panel): where its 17 us go
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(dK, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
        hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dK, (int64_t)ld, 64, (const double*)dK, 0, dws, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
        hipDeviceSynchronize();
#ifdef CIMRGP_STAMP
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
#else
        for (auto& v : st) v = 0;
#endif
        if (rep)
            printf("k_diag64 (nine waves): load+gather %.2f us, pivot loop %.2f us, store %.2f us | pivot wave, block 8: read+self-update %.0f, "
                   "4 pivots %.0f, publish %.0f, barrier %.0f clocks; next block starts %.0f after this one; tile wave 3: wait %.0f, update+gather %.0f\n",
                   tick_us(st[2] - st[0]), tick_us(st[3] - st[2]), tick_us(st[4] - st[3]), (double)(st[17] - st[16]), (double)(st[18] - st[17]),
                   (double)(st[19] - st[18]), (double)(st[20] - st[19]), (double)(st[21] - st[16]), (double)(st[13] - st[12]), (double)(st[14] - st[13]));
        if (rep)
            printf("raw stamps relative to pivot-wave iteration start (16): tile wave 3 arrives at barrier %lld, leaves %lld, done update+gather %lld; "
                   "pivot wave: selfupd done %lld, pivots done %lld, published %lld, past barrier %lld, next iteration past barrier %lld\n",
                   st[12] - st[16], st[13] - st[16], st[14] - st[16], st[17] - st[16], st[18] - st[16], st[19] - st[16], st[20] - st[16], st[21] - st[16]);
    }
    {
        int* dsimd; hipMalloc(&dsimd, 4 * 16 * 4);
        int hs[64];
        for (int nt : {576, 256, 768}) {
            hipMemset(dsimd, 0xff, 4 * 16 * 4);
            hipLaunchKernelGGL(k_simd_ids, dim3(4), dim3(nt), 0, 0, dsimd);
            hipMemcpy(hs, dsimd, sizeof(hs), hipMemcpyDeviceToHost);
            for (int b = 0; b < 4; ++b) {
                printf("workgroup %d of %d threads: simd of waves:", b, nt);
                for (int wv = 0; wv < nt / 64; ++wv) printf(" %d", (hs[b * 16 + wv] >> 4) & 3);
                printf("  (cu %d)\n", (hs[b * 16] >> 8) & 15);
            }
        }
    }
    // the two forms of the diagonal kernel must agree (L in place, inverse in the workspace)
    for (int kprev : {0, 192}) {
        std::vector<double> out[2], inv[2];
        for (int form = 0; form < 2; ++form) {
            hipMemcpy(dK, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
            double* blk = dK + (size_t)256 * ld + 256;                 // a block with 256 columns to its left
            const double* lrow = blk - kprev;
            if (form == 0) hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, blk, (int64_t)ld, 64, lrow, kprev, dws, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
            else hipLaunchKernelGGL((k_diag64q<double>), dim3(1), dim3(Q_NT), 0, 0, blk, (int64_t)ld, 64, lrow, kprev, dws, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
            hipDeviceSynchronize();
            out[form].resize(64 * 64); inv[form].resize(64 * 64);
            for (int r = 0; r < 64; ++r) hipMemcpy(&out[form][r * 64], blk + (size_t)r * ld, 64 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(inv[form].data(), dws, 64 * 64 * 8, hipMemcpyDeviceToHost);
            int32_t info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
            printf("kprev %d form %d info %d L[0][0] %.6f L[63][63] %.6f inv[63][0] %.6e\n", kprev, form, info, out[form][0], out[form][63 * 64 + 63], inv[form][63 * 64]);
        }
        double dl = 0, di = 0;
        for (int r = 0; r < 64; ++r) for (int c = 0; c <= r; ++c) {
            dl = std::max(dl, std::fabs(out[0][r * 64 + c] - out[1][r * 64 + c]));
            di = std::max(di, std::fabs(inv[0][r * 64 + c] - inv[1][r * 64 + c]));
        }
        printf("kprev %d: nine-wave vs four-wave max |dL| %.3e max |dinv| %.3e\n", kprev, dl, di);
        if (kprev == 0) {
            // which 4-column blocks of L differ, per 16-row tile (X = differs)
            for (int rt = 0; rt < 4; ++rt) {
                printf("rows %2d-%2d:", rt * 16, rt * 16 + 15);
                for (int cb = 0; cb < 16; ++cb) {
                    double m = 0;
                    for (int r = rt * 16; r < rt * 16 + 16; ++r) for (int c = cb * 4; c < cb * 4 + 4; ++c)
                        if (c <= r) m = std::max(m, std::fabs(out[0][r * 64 + c] - out[1][r * 64 + c]));
                    printf(" %c", m > 1e-9 ? 'X' : '.');
                }
                printf("\n");
            }
            for (int r = 16; r < 26; ++r) printf("row %d col 20: nine %.6f four %.6f | col 16: %.6f %.6f\n", r, out[0][r * 64 + 20], out[1][r * 64 + 20], out[0][r * 64 + 16], out[1][r * 64 + 16]);
        }
    }
    // repeatability: the same block factored 2000 times by each form, FP32 and FP64 (any difference between two
    // runs of the SAME kernel on the SAME input is a race inside it)
    {
        std::vector<float> hf((size_t)64 * 64);
        for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) hf[i * 64 + j] = (float)h[(size_t)(i + 300) * ld + j + 300];
        float *dF, *dwsF, *dref; double* dwsD2;
        hipMalloc(&dF, 64 * 64 * 4 * 2); hipMalloc(&dwsF, 64 * 64 * 4); hipMalloc(&dref, 64 * 64 * 4 * 2); hipMalloc(&dwsD2, 64 * 64 * 8);
        for (int form = 0; form < 2; ++form) {
            int mism = 0;
            std::vector<float> ref(64 * 64 * 2), got(64 * 64 * 2);
            for (int it = 0; it < 2000; ++it) {
                hipMemcpy(dF, hf.data(), 64 * 64 * 4, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
                if (form == 0) hipLaunchKernelGGL((k_diag64<float>), dim3(1), dim3(DG_NT), 0, 0, dF, (int64_t)64, 64, (const float*)dF, 0, dwsF, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<float>());
                else hipLaunchKernelGGL((k_diag64q<float>), dim3(1), dim3(Q_NT), 0, 0, dF, (int64_t)64, 64, (const float*)dF, 0, dwsF, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<float>());
                hipMemcpy(got.data(), dF, 64 * 64 * 4, hipMemcpyDeviceToHost);
                hipMemcpy(got.data() + 64 * 64, dwsF, 64 * 64 * 4, hipMemcpyDeviceToHost);
                if (it == 0) ref = got;
                else {
                    bool same = true;
                    for (int r = 0; r < 64 && same; ++r) for (int c = 0; c <= r; ++c)
                        if (memcmp(&ref[r * 64 + c], &got[r * 64 + c], 4) || memcmp(&ref[4096 + r * 64 + c], &got[4096 + r * 64 + c], 4)) { same = false; break; }
                    if (!same) ++mism;
                }
            }
            printf("FP32 %s: %d of 1999 repetitions differ from the first\n", form == 0 ? "k_diag64 " : "k_diag64q", mism);
        }
        for (int form = 0; form < 2; ++form) {
            int mism = 0;
            std::vector<double> ref(64 * 64), got(64 * 64), blk(64 * 64);
            for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) blk[i * 64 + j] = h[(size_t)(i + 300) * ld + j + 300];
            double* dD; hipMalloc(&dD, 64 * 64 * 8);
            for (int it = 0; it < 2000; ++it) {
                hipMemcpy(dD, blk.data(), 64 * 64 * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
                if (form == 0) hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dD, (int64_t)64, 64, (const double*)dD, 0, dwsD2, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
                else hipLaunchKernelGGL((k_diag64q<double>), dim3(1), dim3(Q_NT), 0, 0, dD, (int64_t)64, 64, (const double*)dD, 0, dwsD2, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
                hipMemcpy(got.data(), dD, 64 * 64 * 8, hipMemcpyDeviceToHost);
                if (it == 0) ref = got;
                else {
                    bool same = true;
                    for (int r = 0; r < 64 && same; ++r) for (int c = 0; c <= r; ++c) if (memcmp(&ref[r * 64 + c], &got[r * 64 + c], 8)) { same = false; break; }
                    if (!same) ++mism;
                }
            }
            printf("FP64 %s: %d of 1999 repetitions differ from the first\n", form == 0 ? "k_diag64 " : "k_diag64q", mism);
        }
    }
    // repeatability of a whole panel chain (first diagonal block, three links, last panel solve) in the four-wave
    // and the nine-wave form, FP32, 1024 rows below: every repetition must reproduce the first bit for bit
    {
        const int nn = 1280, lds = 1296;
        std::vector<float> hf((size_t)nn * lds, 0.f), ref((size_t)nn * 256), got((size_t)nn * 256);
        for (int i = 0; i < nn; ++i) for (int j = 0; j <= i; ++j) hf[(size_t)i * lds + j] = (float)(std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 0.1 : 0.0));
        float *dM, *dW; hipMalloc(&dM, hf.size() * 4); hipMalloc(&dW, ((size_t)(nn / 64) * 4096 + (size_t)(nn / 256) * 65536) * 4);
        for (int alone = 0; alone < 2; ++alone) {
            int mism = 0;
            for (int it = 0; it < 400; ++it) {
                hipMemcpy(dM, hf.data(), hf.size() * 4, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
                panel_chain<float>(dM, nn, lds, dW, dinfo, 0, 256, (float*)nullptr, 0, 0, PotrfBatch(), (hipStream_t)0, "probe", alone != 0);
                hipDeviceSynchronize();
                for (int r = 0; r < nn; ++r) hipMemcpy(&got[(size_t)r * 256], dM + (size_t)r * lds, 256 * 4, hipMemcpyDeviceToHost);
                if (it == 0) ref = got;
                else if (memcmp(ref.data(), got.data(), ref.size() * 4)) ++mism;
            }
            int32_t info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
            printf("FP32 panel chain, %s form: %d of 399 repetitions differ from the first (info %d)\n", alone ? "nine-wave" : "four-wave", mism, info);
        }
    }
    // the same panel chain BESIDE a running FP32 trailing update on another stream (the look-ahead's situation)
    {
        const int nn = 1280, lds = 1296, mu = 6144, ldu = 6160;
        std::vector<float> hf((size_t)nn * lds, 0.f), ref((size_t)nn * 256), got((size_t)nn * 256);
        for (int i = 0; i < nn; ++i) for (int j = 0; j <= i; ++j) hf[(size_t)i * lds + j] = (float)(std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 0.1 : 0.0));
        float *dM, *dW, *dC, *dA; hipMalloc(&dM, hf.size() * 4); hipMalloc(&dW, ((size_t)(nn / 64) * 4096 + (size_t)(nn / 256) * 65536) * 4);
        hipMalloc(&dC, (size_t)mu * ldu * 4); hipMalloc(&dA, (size_t)mu * 256 * 4);
        hipMemset(dC, 0, (size_t)mu * ldu * 4); hipMemset(dA, 0, (size_t)mu * 256 * 4);
        hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        for (int alone = 0; alone < 2; ++alone) {
            int mism = 0;
            for (int it = 0; it < 300; ++it) {
                hipMemcpy(dM, hf.data(), hf.size() * 4, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
                hipDeviceSynchronize();
                if (it) gemm_nt_sub<float>(dC, ldu, dA, 256, dA, 256, mu, mu, 256, true, s2);      // the co-runner (not for the reference run)
                panel_chain<float>(dM, nn, lds, dW, dinfo, 0, 256, (float*)nullptr, 0, 0, PotrfBatch(), s1, "probe", alone != 0);
                hipDeviceSynchronize();
                for (int r = 0; r < nn; ++r) hipMemcpy(&got[(size_t)r * 256], dM + (size_t)r * lds, 256 * 4, hipMemcpyDeviceToHost);
                if (it == 0) ref = got;
                else if (memcmp(ref.data(), got.data(), ref.size() * 4)) ++mism;
            }
            printf("FP32 panel chain beside a running update, %s form: %d of 299 repetitions differ from the undisturbed first\n", alone ? "nine-wave" : "four-wave", mism);
        }
    }
    // un-stamped rate: 200 back-to-back launches of each diagonal kernel on a 64 x 64 block (kprev = 0)
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemcpy(dK, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            for (int it = 0; it < 200; ++it) {
                if (form == 0) hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dK + (size_t)(it % 32) * 64 * (ld + 1), (int64_t)ld, 64, (const double*)dK, 0, dws, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
                else hipLaunchKernelGGL((k_diag64q<double>), dim3(1), dim3(Q_NT), 0, 0, dK + (size_t)(it % 32) * 64 * (ld + 1), (int64_t)ld, 64, (const double*)dK, 0, dws, dinfo, 0, (int64_t)0, (int64_t)0, (int64_t)0, no_riders<double>());
            }
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.2f us per launch (200 back to back)\n", form == 0 ? "k_diag64 " : "k_diag64q", ms * 1000.0 / 200);
        }
    }
    // the chain of one panel, alone on the machine, in its three forms (whole factorisations of n = 4096, one queue)
    // (the form is read once per process: CIMRGP_CHAIN=split|wide|quad tools/diag_probe)
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dK, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        potrf_run<double>(dK, n, ld, dws, dinfo, (double*)nullptr, 0, 0, (hipStream_t)0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int32_t info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
        if (rep) printf("potrf n = %d, chain form %s: %.3f ms (info %d)\n", n, getenv("CIMRGP_CHAIN") ? getenv("CIMRGP_CHAIN") : "default", ms, info);
    }
    return 0;
}
