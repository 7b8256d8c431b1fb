// Diagnostic for the round-4 FP32 wrong-result race of the four-wave chain (HISTORY.md, rounds 4 and 5).
// Built with -DCIMRGP_GATHER_LATE (the round-3 hand-over: the gather right behind a step's multiplies) it runs one
// panel chain (k_diag64q, three k_linkq, k_trsm64) beside a running FP32 trailing update on another stream and
// compares every repetition with an undisturbed first one, bit for bit.  With -DCIMRGP_RACE_PDUMP the pivot wave also
// records, per block of 4 pivots, the 4 columns it READ as gathered: in a wrong run the first differing words are,
// bit for bit, what the same LDS words held two blocks earlier -- the gathering wave's last store instruction(s) had
// not been performed when the pivot wave read behind the barrier.  Variants and results: tools/lab/race_probe.sh.
// Not part of the product.
#include "../../cimrgp_amd/csrc/potrf.hip"
#include "../../cimrgp_amd/csrc/gemm_nt.hip"
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

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    const int nn = 1280, lds = 1296, mu = 6144, ldu = 6160;
    std::vector<float> hf((size_t)nn * lds, 0.f), ref((size_t)nn * 256), got((size_t)nn * 256);
    for (int i = 0; i < nn; ++i) for (int j = 0; j <= i; ++j) hf[(size_t)i * lds + j] = (float)(std::exp(-0.5 * (i - j) * (i - j) / 900.0) + (i == j ? 0.1 : 0.0));
    float *dM, *dW, *dC, *dA; int32_t* dinfo;
    hipMalloc(&dM, hf.size() * 4); hipMalloc(&dW, ((size_t)(nn / 64) * 4096 + (size_t)(nn / 256) * 65536) * 4);
    hipMalloc(&dC, (size_t)mu * ldu * 4); hipMalloc(&dA, (size_t)mu * 256 * 4); hipMalloc(&dinfo, 4);
    hipMemset(dC, 0, (size_t)mu * ldu * 4); hipMemset(dA, 0, (size_t)mu * 256 * 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
#ifdef CIMRGP_RACE_PDUMP
    const int PW = 4 * 16 * 64 * 4;
    float* pbuf; hipMalloc(&pbuf, PW * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_race_pbuf), &pbuf, sizeof(pbuf));
    std::vector<float> pref(PW), pgot(PW);
#endif
    int mism = 0, shown = 0;
    for (int it = 0; it < reps; ++it) {
        hipMemcpy(dM, hf.data(), hf.size() * 4, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
        hipDeviceSynchronize();
        if (it) gemm_nt_sub<float>(dC, ldu, dA, 256, dA, 256, mu, mu, 256, true, s2);      // the co-runner (not for the reference run)
        panel_chain<float>(dM, nn, lds, dW, dinfo, 0, 256, (float*)nullptr, 0, 0, PotrfBatch(), s1, "probe", false);
        hipDeviceSynchronize();
        for (int r = 0; r < nn; ++r) hipMemcpy(&got[(size_t)r * 256], dM + (size_t)r * lds, 256 * 4, hipMemcpyDeviceToHost);
#ifdef CIMRGP_RACE_PDUMP
        hipMemcpy(pgot.data(), pbuf, PW * 4, hipMemcpyDeviceToHost);
        if (it == 0) pref = pgot;
#endif
        if (it == 0) { ref = got; continue; }
        if (!memcmp(ref.data(), got.data(), ref.size() * 4)) continue;
        ++mism;
        if (shown >= 12) continue;
        ++shown;
        // where does L differ (first 64-row block, by 64-column sub-block of the panel)?
        int fr = -1, fc = -1, cnt = 0;
        for (int r = 0; r < nn; ++r) for (int c = 0; c < 256 && c <= r; ++c)
            if (memcmp(&ref[(size_t)r * 256 + c], &got[(size_t)r * 256 + c], 4)) { if (fr < 0) { fr = r; fc = c; } ++cnt; }
        printf("rep %d: %d entries of the panel differ; first (row %d, col %d): ref %.9g got %.9g\n", it, cnt, fr, fc, ref[(size_t)fr * 256 + fc], got[(size_t)fr * 256 + fc]);
        {
            const int b0 = (fr / 64) * 64;                 // rows of the first bad 64-block: which entries differ
            printf("   differing entries within rows %d..%d:", b0, b0 + 63);
            int pr = 0;
            for (int r = b0; r < b0 + 64 && r < nn; ++r) for (int c = 0; c < 256 && c <= r; ++c)
                if (memcmp(&ref[(size_t)r * 256 + c], &got[(size_t)r * 256 + c], 4) && pr++ < 24) printf(" (%d,%d)", r, c);
            printf("\n");
        }
#ifdef CIMRGP_RACE_PDUMP
        {
            // the first block whose gathered columns, AS READ by the pivot wave, differ from the reference run's
            bool found = false;
            for (int sl = 0; sl < 4 && !found; ++sl)
                for (int p = 0; p < 16 && !found; ++p) {
                    int nd = 0;
                    for (int l = 0; l < 64; ++l) for (int t = 0; t < 4; ++t) {
                        const int o = ((sl * 16 + p) * 64 + l) * 4 + t;
                        if (memcmp(&pref[o], &pgot[o], 4)) {
                            const float stale = p >= 2 ? pref[o - 2 * 64 * 4] : 0.f;      // the same LDS word two blocks earlier (same buffer of the pair)
                            if (nd++ < 16)
                                printf("   sub-block %d block %d: row %d column %d+%d: read %.9g, reference run read %.9g; the same LDS word held %.9g two blocks earlier (reference run)%s\n",
                                       sl, p, l, 4 * p, t, pgot[o], pref[o], stale, memcmp(&stale, &pgot[o], 4) ? "" : "  <- EXACTLY the stale word");
                        }
                    }
                    if (nd) { printf("   -> first differing READ: sub-block %d block %d, %d words\n", sl, p, nd); found = true; }
                }
            if (!found) printf("   the pivot wave read the same gathered columns as in the reference run in every block\n");
        }
#endif
    }
    printf("FP32 panel chain (round-3 hand-over) beside a running update: %d of %d repetitions differ from the undisturbed first\n", mism, reps - 1);
    return 0;
}
