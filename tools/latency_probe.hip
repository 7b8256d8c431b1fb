// Calibration microbenchmarks for the latency-bound panel kernels (diagnostic only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_chase(const int* __restrict__ next, int steps, int* out, long long* cyc)
{
    long long t0 = __builtin_amdgcn_s_memtime();
    int p = threadIdx.x;
    for (int s = 0; s < steps; ++s) p = next[p];
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = p; cyc[0] = t1 - t0; }
}

__global__ void k_lds_barrier(int iters, double* out, long long* cyc, int nwaves)
{
    __shared__ double buf[2][64];
    double a = threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < iters; ++j) {
        if ((int)(threadIdx.x >> 6) == (j & (nwaves - 1))) buf[j & 1][threadIdx.x & 63] = a;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const double d = buf[j & 1][j & 63];
        a = fma(a, 0.999, d * 1e-6);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_rsq_chain(int iters, double* out, long long* cyc)
{
    double d = 1.0 + threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < iters; ++j) {
        double r = __builtin_amdgcn_rsq(d);
        double e0 = fma(-d * r, r, 1.0); r = fma(0.5 * r, e0, r);
        double e1 = fma(-d * r, r, 1.0); r = fma(0.5 * r, e1, r);
        d = d * r + 1.0;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_empty(int* out) { if (threadIdx.x == 999) out[0] = 1; }

int main()
{
    const int N = 1 << 22;
    std::vector<int> h(N);
    // stride permutation so each hop is a different cache line
    for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i + 4099 * 16) % N);
    int* d_next; int* d_out; long long* d_cyc; double* d_dout;
    CK(hipMalloc(&d_next, N * sizeof(int))); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_cyc, 64)); CK(hipMalloc(&d_dout, 8192));
    CK(hipMemcpy(d_next, h.data(), N * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long cyc; float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d_next, 2000, d_out, d_cyc); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("chase rep%d: 2000 hops  %.1f us wall  %lld memtime ticks  -> %.1f ns/hop, %.0f ticks/hop\n", rep, ms * 1e3, cyc, ms * 1e6 / 2000, (double)cyc / 2000);
    }
    for (int nw = 1; nw <= 8; nw *= 2) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_barrier, dim3(1), dim3(64 * nw), 0, 0, 2000, d_dout, d_cyc, nw); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("lds+barrier %d waves: 2000 iters %.1f us wall, %lld ticks -> %.1f ns/iter, %.0f ticks/iter\n", nw, ms * 1e3, cyc, ms * 1e6 / 2000, (double)cyc / 2000);
    }
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rsq_chain, dim3(1), dim3(64), 0, 0, 2000, d_dout, d_cyc); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("rsq+2NR chain: 2000 iters %.1f us wall, %lld ticks -> %.1f ns/iter, %.0f ticks/iter\n", ms * 1e3, cyc, ms * 1e6 / 2000, (double)cyc / 2000);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("100 empty launches: %.1f us -> %.2f us each\n", ms * 1e3, ms * 10);
    }
    return 0;
}
