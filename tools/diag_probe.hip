// Diagnostic: phase stamps (s_memtime: shader clocks, ~2.35 GHz under load) of the diagonal kernel and
// whole one-queue factorisations per chain form.
// Not part of the product.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag_probe.hip -o tools/diag_probe
#define CIMRGP_STAMP 1
#include "../cimrgp_amd/csrc/potrf.hip"
#include "../cimrgp_amd/csrc/gemm_nt.hip"
#include <cmath>
#include <vector>
using namespace cimrgp;
namespace cimrgp {
static std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const char* fn, const char* what) { fprintf(stderr, "%s: %s\n", fn, what); return -1; }
int check_hip(hipError_t e, const char* fn, const char* what) { fprintf(stderr, "%s: %s: %s\n", fn, what, hipGetErrorString(e)); return -2; }
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
        hipLaunchKernelGGL((k_diag64<double>), dim3(1), dim3(DG_NT), 0, 0, dK, (int64_t)ld, 64, dK, 0, dws, dinfo, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
        if (rep)
            printf("k_diag64 (nine waves): load+gather %.2f us, pivot loop %.2f us, store %.2f us | pivot wave, block 8: read+self-update %.0f, "
                   "4 pivots %.0f, publish %.0f, barrier %.0f clocks; next block starts %.0f after this one; tile wave 3: wait %.0f, update+gather %.0f\n",
                   tick_us(st[2] - st[0]), tick_us(st[3] - st[2]), tick_us(st[4] - st[3]), (double)(st[17] - st[16]), (double)(st[18] - st[17]),
                   (double)(st[19] - st[18]), (double)(st[20] - st[19]), (double)(st[21] - st[16]), (double)(st[13] - st[12]), (double)(st[14] - st[13]));
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
