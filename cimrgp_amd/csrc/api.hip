// extern "C" entry points of libcimrgp.so (declared in include/cimrgp.h).
#include "common.hpp"
#include <mutex>

#include <string.h>
#include <stdlib.h>
#include <atomic>

namespace cimrgp {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(const char* fn, const char* what)
{
    g_last_error = std::string(fn) + ": " + what;
    return -1;
}

int check_hip(hipError_t e, const char* fn, const char* what)
{
    if (e == hipSuccess) return 0;
    g_last_error = std::string(fn) + ": " + what + ": " + hipGetErrorString(e);
    return -2;
}

// ---- knobs (common.hpp) ----
const Knobs& knobs()
{
    static const Knobs k = [] {
        Knobs v;
#ifdef CIMRGP_TUNING
        auto num = [](const char* name, int64_t dflt) { const char* e = getenv(name); return e ? (int64_t)atoll(e) : dflt; };
        const char* c = getenv("CIMRGP_CHAIN");
        v.chain_mode = !c ? 0 : (c[0] == 's' ? 1 : c[0] == 'w' ? 2 : c[0] == 'q' ? 3 : 0);
        v.tail_below = num("CIMRGP_TAIL_BELOW", v.tail_below);
        v.rows_start_below = num("CIMRGP_ROWS_START", v.rows_start_below);
        v.rows_start_below_early = num("CIMRGP_ROWS_START_EARLY", v.rows_start_below_early);
        v.head_first_above = num("CIMRGP_HEAD_FIRST", v.head_first_above);
        v.far_pair_above = num("CIMRGP_FAR_PAIR", v.far_pair_above);
        v.fused_head0 = (int)num("CIMRGP_HEAD0", v.fused_head0);
        v.gemm_pers = (int)num("CIMRGP_GEMM_PERS", v.gemm_pers);
        v.gemm_pers_f32 = (int)num("CIMRGP_GEMM_PERS_F32", v.gemm_pers_f32);
        v.pers_min_tiles = (int)num("CIMRGP_PERS_MIN_TILES", v.pers_min_tiles);
        v.chain_cus = (int)num("CIMRGP_CHAIN_CUS", v.chain_cus);
        v.pers_max_chunks = (int)num("CIMRGP_PERS_CHUNKS", v.pers_max_chunks);
        v.head_direct_max_rounds = (int)num("CIMRGP_HEAD_DIRECT", v.head_direct_max_rounds);
        v.post_final = (int)num("CIMRGP_POST_FINAL", v.post_final);
        v.heads_beside_rows = (int)num("CIMRGP_HEADS_ROWS", v.heads_beside_rows);
        v.early_first_panel = (int)num("CIMRGP_EARLY_PANEL", v.early_first_panel);
        v.early_panels = (int)num("CIMRGP_EARLY_PANELS", v.early_panels);
        v.early_cus = (int)num("CIMRGP_EARLY_CUS", v.early_cus);
        v.pers_flex_cus = (int)num("CIMRGP_PERS_FLEX", v.pers_flex_cus);
        v.pers_flex_min_rounds = (int)num("CIMRGP_PERS_FLEX_MIN", v.pers_flex_min_rounds);
        v.rows_fused_tail = (int)num("CIMRGP_ROWS_FUSED", v.rows_fused_tail);
        v.rows_pair_above = num("CIMRGP_ROWS_PAIR", v.rows_pair_above);
        v.fused_max_chain_wgs = num("CIMRGP_FUSED_MAX", v.fused_max_chain_wgs);
        v.batch_halves_min = (int)num("CIMRGP_BATCH_HALVES", v.batch_halves_min);
        v.rows_cus = (int)num("CIMRGP_ROWS_CUS", v.rows_cus);
        v.rows_step = (int)num("CIMRGP_ROWS_STEP", v.rows_step);
        v.rider_lean = (int)num("CIMRGP_RIDER_LEAN", v.rider_lean);
        v.rider_round_us = (int)num("CIMRGP_RIDER_ROUND", v.rider_round_us);
        v.trsm_group = (int)num("CIMRGP_TRSM_GROUP", v.trsm_group);
        v.rows_beside_tail_below = num("CIMRGP_ROWS_BESIDE", v.rows_beside_tail_below);
        v.tail_far_cus = (int)num("CIMRGP_TAIL_FAR_CUS", v.tail_far_cus);
        v.tail_far_min_rows = num("CIMRGP_TAIL_FAR_MIN", v.tail_far_min_rows);
#endif
        return v;
    }();
    return k;
}

static std::atomic<int> g_rows_queues{2};
int rows_queues() { return g_rows_queues.load(std::memory_order_relaxed); }

}  // namespace cimrgp

using namespace cimrgp;

#define DISPATCH(dtype, fn, CALL_F32, CALL_F64)                  \
    switch (dtype) {                                             \
        case CIMRGP_F32: return CALL_F32;                        \
        case CIMRGP_F64: return CALL_F64;                        \
        default: return fail(fn, "unknown dtype");               \
    }

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t esize(int dtype) { return dtype == CIMRGP_F64 ? 8 : 4; }
static inline bool ld_ok(int dtype, int64_t ld) { return ld % (dtype == CIMRGP_F64 ? 2 : 4) == 0; }

template <typename T>
static int layer_fit_typed(const void* x, const void* y, const void* fbar, void* train_out, const int64_t* starts, int batch,
                           int64_t n, int d, int q, double ell, double sf2, double noise_fixed, double noise_frac,
                           double noise_floor, const void* shared_bias, const void* shared_noise, void* k, int64_t ldk,
                           int64_t k_stride, void* ws, size_t ws_stride_bytes, int32_t* info, void* rows, int64_t ldr, void* z,
                           void* alpha, void* bias, void* noise, void* scratch, hipStream_t st)
{
    LayerFit<T> a;
    a.x = (const T*)x; a.y = (const T*)y; a.fbar = (const T*)fbar; a.train_out = (T*)train_out; a.starts = starts;
    a.batch = batch; a.n = n; a.d = d; a.q = q;
    a.ell = ell; a.sf2 = sf2; a.noise_fixed = noise_fixed; a.noise_frac = noise_frac; a.noise_floor = noise_floor;
    a.shared_bias = (const T*)shared_bias; a.shared_noise = (const T*)shared_noise;
    a.k = (T*)k; a.ldk = ldk; a.sk = k_stride; a.ws = (T*)ws; a.sws = (int64_t)(ws_stride_bytes / sizeof(T)); a.info = info;
    a.rows = (T*)rows; a.ldr = ldr; a.srows = (int64_t)q * ldr;
    a.z = (T*)z; a.alpha = (T*)alpha; a.bias = (T*)bias; a.noise = (T*)noise; a.scratch = (T*)scratch;
    return layer_fit_run<T>(a, st);
}

template <typename T>
static int layer_predict_typed(const void* x, const int64_t* starts, int64_t n, int d, const void* xs, const int64_t* t_starts,
                               int64_t ns, int batch, double ell, double sf2, const void* l, int64_t ldl, int64_t l_stride,
                               const void* ws, size_t ws_stride_bytes, const void* z, int q, const void* bias, const void* noise,
                               void* w, int64_t ldw, int64_t w_stride, void* mean, void* var, hipStream_t st)
{
    LayerPredict<T> a;
    a.x = (const T*)x; a.starts = starts; a.n = n; a.d = d; a.xs = (const T*)xs; a.t_starts = t_starts; a.ns = ns; a.batch = batch;
    a.ell = ell; a.sf2 = sf2; a.l = (const T*)l; a.ldl = ldl; a.sl = l_stride; a.ws = (const T*)ws;
    a.sws = (int64_t)(ws_stride_bytes / sizeof(T)); a.z = (const T*)z; a.q = q; a.bias = (const T*)bias; a.noise = (const T*)noise;
    a.w = (T*)w; a.ldw = ldw; a.sw = w_stride; a.mean = (T*)mean; a.var = (T*)var;
    return layer_predict_run<T>(a, st);
}

namespace cimrgp {
// (layer.hip) the targets as carried rows, rows[c][j] = y[j][c]; and back: z[j][c] = alpha[j][c] = rows[c][j]
template <typename T> int rhs_rows_run(const T* y, int64_t n, int q, T* rows, int64_t ldr, hipStream_t st);
template <typename T> int rows_to_z_run(const T* rows, int64_t ldr, int64_t n, int q, T* z, T* alpha, hipStream_t st, T* work = nullptr);
}  // namespace cimrgp

namespace {
// cimrgp_block_posterior: the separate entry points' work in one call, in their order.  (Measured and not kept: the
// Gram matrix right of the first panel and the carried rows written BESIDE the first panel's chain -- the chain's
// first diagonal block then took 59-64 us instead of 16-20 under the write traffic and the step got no shorter:
// HISTORY.md.)
// Staged calls in flight, per device (round 5: one record per (stream, stream_solve) PAIR -- round 4 kept one record
// per device, so two caller stream pairs on one device overwrote each other's and lost both safety nets below).
// A record = the latest staged call of its pair: an event (created once per slot) recorded behind its solve stage and
// the call's whole buffer set (k, w, workspace, alpha, z, scratch: the solve stage reads or writes every one of them).
struct StagedRec_ {
    hipStream_t st = nullptr, solve = nullptr;
    hipEvent_t event = nullptr;
    hipEvent_t prev_event = nullptr;           // behind the solve stage of the pair's call BEFORE the latest (has_prev)
    bool has_prev = false;
    bool live = false;
    unsigned long long age = 0;
    const void* buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
constexpr int STAGED_DEVS = 16, STAGED_RECS = 32, STAGED_RING = 64;
struct StagedDev_ {
    std::mutex m;
    StagedRec_ rec[STAGED_RECS];
    unsigned long long clock = 0;
    hipEvent_t ring[STAGED_RING] = {};          // hand-over events between a call's stages
    unsigned ring_next = 0;
};
static StagedDev_* staged_devs_() { static StagedDev_ a[STAGED_DEVS]; return a; }

// cimrgp_shutdown: the events above (idle devices only: the caller has synchronised)
static int staged_shutdown_()
{
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < STAGED_DEVS; ++d) {
        StagedDev_& sd = staged_devs_()[d];
        std::lock_guard<std::mutex> guard(sd.m);
        bool any = false;
        for (auto& r : sd.rec) any = any || r.event != nullptr || r.prev_event != nullptr;
        for (auto& e : sd.ring) any = any || e != nullptr;
        if (!any) continue;
        (void)hipSetDevice(d);
        for (auto& r : sd.rec) { if (r.event) (void)hipEventDestroy(r.event); if (r.prev_event) (void)hipEventDestroy(r.prev_event); r = StagedRec_(); }
        for (auto& e : sd.ring) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    }
    if (have_cur) (void)hipSetDevice(cur);
    return 0;
}

template <typename T>
int block_posterior_typed(const void* x, int64_t n, int d, const void* y, int q, const void* xs, int64_t ns, double ell, double sf2,
                          double noise, void* k, int64_t ldk, void* ws, int32_t* info, void* w, int64_t ldw, void* alpha, void* z,
                          void* scratch, void* mean, void* var, int add_noise, int accumulate,
                          hipStream_t s_front, hipStream_t st, hipStream_t s_solve)
{
    using namespace cimrgp;
    const char* fn = "cimrgp_block_posterior";
    T* wt = (T*)w;
    int dev_id = 0;                                   // the events below belong to the device they were created on
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= STAGED_DEVS) dev_id = 0;
    StagedDev_& sd = staged_devs_()[dev_id];
    // `to` continues where `from` stands now (an event that lives until both have passed it)
    auto hand_over = [&](hipStream_t from, hipStream_t to) -> int {
        if (from == to) return 0;
        // a ring of events per device, created once (a wait captures the record that precedes it: an event may be
        // recorded again while an earlier wait on it is still queued)
        hipEvent_t e = nullptr;
        {
            std::lock_guard<std::mutex> guard(sd.m);
            hipEvent_t& slot = sd.ring[sd.ring_next++ % STAGED_RING];
            if (slot == nullptr) {
                hipError_t e0 = hipEventCreateWithFlags(&slot, hipEventDisableTiming);
                if (e0 != hipSuccess) { slot = nullptr; return check_hip(e0, fn, "hipEventCreate"); }
            }
            e = slot;
        }
        hipError_t e1 = hipEventRecord(e, from);
        hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(to, e, 0) : e1;
        return check_hip(e2, fn, "hipEventRecord / hipStreamWaitEvent");
    };
    const void* mine[6] = {k, w, ws, alpha, z, scratch};
    // Safety net 1: ANY buffer of a staged call whose solve stage may still be running (any pair of streams of this
    // device, a plain call on the same set included) handed in again -- one buffer set where two are needed: the
    // front end waits for that solve stage; correct results, no overlap.
    int rc = 0;
    {
        std::lock_guard<std::mutex> guard(sd.m);
        for (auto& r : sd.rec) {
            if (!r.live || r.solve == s_front || rc) continue;       // same queue: already ordered
            bool shares = false;
            for (const void* a : mine) for (const void* b : r.buf) shares = shares || (a != nullptr && a == b);
            if (shares) rc = check_hip(hipStreamWaitEvent(s_front, r.event, 0), fn, "hipStreamWaitEvent");
        }
    }
    // Safety net 3 (round 5): a front stream of its own (cimrgp_front_queue: the front end beside the PREVIOUS call's
    // factorisation) is not ordered behind anything the earlier calls did on `st`.  The latest call of the pair is covered
    // by net 1 when it shares a buffer; every call before it by one wait for the solve stage of the call before the
    // latest (the solve queue is in order, so that covers all older ones; it finished a factorisation ago: no cost).
    // Without a solve queue of its own there are no records: the front stream then simply follows `st`.
    if (!rc && s_front != st) {
        if (s_solve == st) {
            rc = hand_over(st, s_front);
        } else {
            std::lock_guard<std::mutex> guard(sd.m);
            for (auto& r : sd.rec)
                if (r.live && r.has_prev && r.st == st && r.solve == s_solve && !rc)
                    rc = check_hip(hipStreamWaitEvent(s_front, r.prev_event, 0), fn, "hipStreamWaitEvent");
        }
    }
    // front end: the Gram matrix, the cross-Gram matrix and the targets as carried rows
    if (!rc) rc = rbf_gram_run<T>((const T*)x, n, (const T*)x, n, d, ell, sf2, noise, (T*)k, ldk, true, true, s_front);
    if (!rc && ns > 0) rc = rbf_gram_run<T>((const T*)xs, ns, (const T*)x, n, d, ell, sf2, 0.0, wt, ldw, false, false, s_front);
    if (!rc) rc = rhs_rows_run<T>((const T*)y, n, q, wt + ns * ldw, ldw, s_front);
    if (!rc) rc = hand_over(s_front, st);
    if (!rc) rc = potrf_run<T>((T*)k, n, ldk, (T*)ws, info, wt, ns + q, ldw, st, s_front);
    if (!rc) rc = hand_over(st, s_solve);
    // Safety net 2: two buffer sets in rotation need no ordering by the caller: behind its factorisation `st` waits for
    // the solve stage of the PREVIOUS call on the same pair of streams (finished long ago: it ran beside this
    // factorisation), so whatever the caller enqueues on `st` next -- the front end of the call after this one, on the
    // set that solve stage read -- comes after it.
    if (!rc && s_solve != st) {
        std::lock_guard<std::mutex> guard(sd.m);
        for (auto& r : sd.rec)
            if (r.live && r.st == st && r.solve == s_solve && !rc)
                rc = check_hip(hipStreamWaitEvent(st, r.event, 0), fn, "hipStreamWaitEvent");
    }
    // z = L^-1 y (the last q carried rows), alpha = L^-T z, mean = W z, var = sf2 - sum W^2 (+ noise)
    if (!rc) rc = rows_to_z_run<T>(wt + ns * ldw, ldw, n, q, (T*)z, (T*)alpha, s_solve, (T*)scratch);
    if (!rc) rc = potrs_run<T>((const T*)k, n, ldk, (const T*)ws, (T*)alpha, q, nullptr, (T*)scratch, true, s_solve, PotrfBatch(), true);
    if (!rc && ns > 0) rc = predict_from_w_run<T>((const T*)w, ns, n, ldw, (const T*)z, q, sf2, add_noise ? noise : 0.0, nullptr, nullptr,
                                                   (T*)mean, (T*)var, accumulate, s_solve, 1, nullptr, 0);
    if (!rc && s_solve != st) {
        // this call becomes its pair's record (its slot, or a free one, or the slot of the pair idle longest -- whose
        // solve stage is waited for on the host first, so that no net is lost: 32 pairs per device, rare)
        std::lock_guard<std::mutex> guard(sd.m);
        StagedRec_* slot = nullptr;
        for (auto& r : sd.rec) if (r.live && r.st == st && r.solve == s_solve) slot = &r;
        if (slot) {
            // the pair's own record: the latest call becomes "the one before", its event is kept; the older event is recorded again
            std::swap(slot->event, slot->prev_event);
            slot->has_prev = true;
        } else {
            for (auto& r : sd.rec) if (!r.live && !slot) slot = &r;
            if (!slot) {
                slot = &sd.rec[0];
                for (auto& r : sd.rec) if (r.age < slot->age) slot = &r;
                (void)hipEventSynchronize(slot->event);
            }
            slot->has_prev = false;
        }
        if (slot->event == nullptr && hipEventCreateWithFlags(&slot->event, hipEventDisableTiming) != hipSuccess) {
            slot->event = nullptr;
            slot->live = false;
            return check_hip(hipErrorOutOfMemory, fn, "hipEventCreate");
        }
        rc = check_hip(hipEventRecord(slot->event, s_solve), fn, "hipEventRecord");
        slot->st = st;
        slot->solve = s_solve;
        slot->live = (rc == 0);
        slot->age = ++sd.clock;
        for (int i = 0; i < 6; ++i) slot->buf[i] = mine[i];
    }
    return rc;
}
}  // namespace

extern "C" {

int cimrgp_version(void) { return 100; }

const char* cimrgp_last_error(void) { return g_last_error.c_str(); }

int cimrgp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cimrgp_rbf_gram(int dtype, const void* x_dev, int64_t n, int d, double ell, double sf2, double diag_add,
                    void* k_dev, int64_t ldk, int lower_only, void* stream)
{
    const char* fn = "cimrgp_rbf_gram";
    CIMRGP_REQUIRE(x_dev && k_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0, fn, "negative size");
    DISPATCH(dtype, fn,
             rbf_gram_run<float>((const float*)x_dev, n, (const float*)x_dev, n, d, ell, sf2, diag_add, (float*)k_dev, ldk,
                                 true, lower_only != 0, S(stream)),
             rbf_gram_run<double>((const double*)x_dev, n, (const double*)x_dev, n, d, ell, sf2, diag_add, (double*)k_dev,
                                  ldk, true, lower_only != 0, S(stream)));
}

int cimrgp_rbf_cross(int dtype, const void* xa_dev, int64_t na, const void* xb_dev, int64_t nb, int d, double ell,
                     double sf2, void* kab_dev, int64_t ld, void* stream)
{
    const char* fn = "cimrgp_rbf_cross";
    CIMRGP_REQUIRE(xa_dev && xb_dev && kab_dev, fn, "null pointer");
    CIMRGP_REQUIRE(na >= 0 && nb >= 0, fn, "negative size");
    DISPATCH(dtype, fn,
             rbf_gram_run<float>((const float*)xa_dev, na, (const float*)xb_dev, nb, d, ell, sf2, 0.0, (float*)kab_dev, ld,
                                 false, false, S(stream)),
             rbf_gram_run<double>((const double*)xa_dev, na, (const double*)xb_dev, nb, d, ell, sf2, 0.0, (double*)kab_dev,
                                  ld, false, false, S(stream)));
}

size_t cimrgp_potrf_workspace_bytes(int dtype, int64_t n)
{
    if (n <= 0) return 0;
    const int64_t slabs = (n + 63) / 64, panels = (n + CIMRGP_NB - 1) / CIMRGP_NB, pairs = (n / CIMRGP_NB) / 2;
    // 64 x 64 inverses, 256 x 256 inverses, off-diagonal blocks of the 512 x 512 inverses (pairs of full panels)
    return (size_t)(slabs * 64 * 64 + (panels + pairs) * CIMRGP_NB * CIMRGP_NB) * esize(dtype);
}

int cimrgp_potrf(int dtype, void* k_dev, int64_t n, int64_t ldk, void* workspace_dev, size_t workspace_bytes,
                 int32_t* info_dev, void* stream)
{
    const char* fn = "cimrgp_potrf";
    CIMRGP_REQUIRE(k_dev && workspace_dev && info_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && ldk >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(ld_ok(dtype, ldk), fn, "leading dimension must be a multiple of 16 bytes");
    CIMRGP_REQUIRE(aligned16(k_dev) && aligned16(workspace_dev), fn, "pointers must be 16-byte aligned");
    CIMRGP_REQUIRE(workspace_bytes >= cimrgp_potrf_workspace_bytes(dtype, n), fn, "workspace too small");
    if (n == 0) return check_hip(hipMemsetAsync(info_dev, 0, sizeof(int32_t), S(stream)), fn, "memset");
    DISPATCH(dtype, fn,
             potrf_run<float>((float*)k_dev, n, ldk, (float*)workspace_dev, info_dev, nullptr, 0, 0, S(stream)),
             potrf_run<double>((double*)k_dev, n, ldk, (double*)workspace_dev, info_dev, nullptr, 0, 0, S(stream)));
}

int cimrgp_potrf_rows(int dtype, void* k_dev, int64_t n, int64_t ldk, void* workspace_dev, size_t workspace_bytes,
                      int32_t* info_dev, void* b_dev, int64_t m, int64_t ldb, void* stream)
{
    const char* fn = "cimrgp_potrf_rows";
    CIMRGP_REQUIRE(k_dev && workspace_dev && info_dev && b_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && m >= 0 && ldk >= n && ldb >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(ld_ok(dtype, ldk) && ld_ok(dtype, ldb), fn, "leading dimensions must be multiples of 16 bytes");
    CIMRGP_REQUIRE(aligned16(k_dev) && aligned16(workspace_dev) && aligned16(b_dev), fn, "pointers must be 16-byte aligned");
    CIMRGP_REQUIRE(workspace_bytes >= cimrgp_potrf_workspace_bytes(dtype, n), fn, "workspace too small");
    if (n == 0) return check_hip(hipMemsetAsync(info_dev, 0, sizeof(int32_t), S(stream)), fn, "memset");
    DISPATCH(dtype, fn,
             potrf_run<float>((float*)k_dev, n, ldk, (float*)workspace_dev, info_dev, (float*)b_dev, m, ldb, S(stream)),
             potrf_run<double>((double*)k_dev, n, ldk, (double*)workspace_dev, info_dev, (double*)b_dev, m, ldb, S(stream)));
}

int cimrgp_block_posterior(int dtype, const void* x_dev, int64_t n, int d, const void* y_dev, int q, const void* xs_dev, int64_t ns,
                           double ell, double sf2, double noise, void* k_dev, int64_t ldk, void* workspace_dev,
                           size_t workspace_bytes, int32_t* info_dev, void* w_dev, int64_t ldw, void* alpha_dev, void* z_dev,
                           void* scratch_dev, void* mean_dev, void* var_dev, int add_noise, int accumulate, void* stream)
{
    return cimrgp_block_posterior_staged(dtype, x_dev, n, d, y_dev, q, xs_dev, ns, ell, sf2, noise, k_dev, ldk, workspace_dev, workspace_bytes,
                                         info_dev, w_dev, ldw, alpha_dev, z_dev, scratch_dev, mean_dev, var_dev, add_noise, accumulate,
                                         stream, stream, stream);
}

int cimrgp_solve_queue(void* stream, void** queue_out)
{
    const char* fn = "cimrgp_solve_queue";
    CIMRGP_REQUIRE(queue_out != nullptr, fn, "null pointer");
    *queue_out = (void*)cimrgp::solve_queue_for(S(stream));
    return 0;
}

int cimrgp_front_queue(void* stream, void** queue_out)
{
    const char* fn = "cimrgp_front_queue";
    CIMRGP_REQUIRE(queue_out != nullptr, fn, "null pointer");
    *queue_out = (void*)cimrgp::front_queue_for(S(stream));
    return 0;
}

int cimrgp_block_posterior_staged(int dtype, const void* x_dev, int64_t n, int d, const void* y_dev, int q, const void* xs_dev, int64_t ns,
                                  double ell, double sf2, double noise, void* k_dev, int64_t ldk, void* workspace_dev,
                                  size_t workspace_bytes, int32_t* info_dev, void* w_dev, int64_t ldw, void* alpha_dev, void* z_dev,
                                  void* scratch_dev, void* mean_dev, void* var_dev, int add_noise, int accumulate,
                                  void* stream_front, void* stream, void* stream_solve)
{
    const char* fn = "cimrgp_block_posterior";
    CIMRGP_REQUIRE(x_dev && y_dev && k_dev && workspace_dev && info_dev && w_dev && alpha_dev && z_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(ns == 0 || (xs_dev && mean_dev && var_dev), fn, "null pointer (test points)");
    CIMRGP_REQUIRE(n >= 1 && ns >= 0 && ldk >= n && ldw >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(q >= 1 && q <= 8, fn, "number of outputs must be in [1, 8]");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(ld_ok(dtype, ldk) && ld_ok(dtype, ldw), fn, "leading dimensions must be multiples of 16 bytes");
    CIMRGP_REQUIRE(aligned16(k_dev) && aligned16(workspace_dev) && aligned16(w_dev), fn, "pointers must be 16-byte aligned");
    CIMRGP_REQUIRE(workspace_bytes >= cimrgp_potrf_workspace_bytes(dtype, n), fn, "workspace too small");
    DISPATCH(dtype, fn,
             block_posterior_typed<float>(x_dev, n, d, y_dev, q, xs_dev, ns, ell, sf2, noise, k_dev, ldk, workspace_dev, info_dev, w_dev, ldw,
                                          alpha_dev, z_dev, scratch_dev, mean_dev, var_dev, add_noise, accumulate, S(stream_front), S(stream),
                                          S(stream_solve)),
             block_posterior_typed<double>(x_dev, n, d, y_dev, q, xs_dev, ns, ell, sf2, noise, k_dev, ldk, workspace_dev, info_dev, w_dev, ldw,
                                           alpha_dev, z_dev, scratch_dev, mean_dev, var_dev, add_noise, accumulate, S(stream_front), S(stream),
                                           S(stream_solve)));
}

int cimrgp_potrf_rows_batched(int dtype, void* k_dev, int64_t n, int64_t ldk, int64_t k_stride, void* workspace_dev,
                              size_t workspace_stride_bytes, int32_t* info_dev, void* b_dev, int64_t m, int64_t ldb,
                              int64_t b_stride, int batch, void* stream)
{
    const char* fn = "cimrgp_potrf_rows_batched";
    CIMRGP_REQUIRE(k_dev && workspace_dev && info_dev, fn, "null pointer");
    CIMRGP_REQUIRE(batch >= 1, fn, "batch must be >= 1");
    CIMRGP_REQUIRE(n >= 0 && m >= 0 && ldk >= n && (m == 0 || (b_dev && ldb >= n)), fn, "bad dimensions");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(ld_ok(dtype, ldk) && (m == 0 || ld_ok(dtype, ldb)), fn, "leading dimensions must be multiples of 16 bytes");
    CIMRGP_REQUIRE(aligned16(k_dev) && aligned16(workspace_dev) && aligned16(b_dev), fn, "pointers must be 16-byte aligned");
    CIMRGP_REQUIRE(k_stride >= n * ldk - (ldk - n) && ld_ok(dtype, k_stride), fn, "matrix stride too small or misaligned");
    CIMRGP_REQUIRE(m == 0 || (b_stride >= m * ldb - (ldb - n) && ld_ok(dtype, b_stride)), fn, "row-block stride too small or misaligned");
    CIMRGP_REQUIRE(workspace_stride_bytes >= cimrgp_potrf_workspace_bytes(dtype, n) && workspace_stride_bytes % 16 == 0, fn,
                   "workspace stride too small or misaligned");
    if (n == 0) return check_hip(hipMemsetAsync(info_dev, 0, sizeof(int32_t) * (size_t)batch, S(stream)), fn, "memset");
    PotrfBatch bt;
    bt.count = batch;
    bt.sk = k_stride;
    bt.sws = (int64_t)(workspace_stride_bytes / esize(dtype));
    bt.sb = b_stride;
    DISPATCH(dtype, fn,
             potrf_batched_run<float>((float*)k_dev, n, ldk, (float*)workspace_dev, info_dev, (float*)b_dev, m, ldb, bt, S(stream)),
             potrf_batched_run<double>((double*)k_dev, n, ldk, (double*)workspace_dev, info_dev, (double*)b_dev, m, ldb, bt,
                                       S(stream)));
}

int cimrgp_solve_lt_batched(int dtype, const void* l_dev, int64_t n, int64_t ldl, int64_t l_stride, const void* workspace_dev,
                            size_t workspace_stride_bytes, void* z_dev, int q, void* scratch_dev, int batch, void* stream)
{
    const char* fn = "cimrgp_solve_lt_batched";
    CIMRGP_REQUIRE(l_dev && workspace_dev && z_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(batch >= 1, fn, "batch must be >= 1");
    CIMRGP_REQUIRE(n >= 0 && ldl >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(workspace_stride_bytes >= cimrgp_potrf_workspace_bytes(dtype, n), fn, "workspace stride too small");
    PotrfBatch bt;
    bt.count = batch;
    bt.sk = l_stride;
    bt.sws = (int64_t)(workspace_stride_bytes / esize(dtype));
    DISPATCH(dtype, fn,
             potrs_run<float>((const float*)l_dev, n, ldl, (const float*)workspace_dev, (float*)z_dev, q, nullptr,
                              (float*)scratch_dev, true, S(stream), bt),
             potrs_run<double>((const double*)l_dev, n, ldl, (const double*)workspace_dev, (double*)z_dev, q, nullptr,
                               (double*)scratch_dev, true, S(stream), bt));
}

int cimrgp_potrs(int dtype, const void* l_dev, int64_t n, int64_t ldl, const void* workspace_dev, void* rhs_dev, int q,
                 void* z_dev, void* scratch_dev, void* stream)
{
    const char* fn = "cimrgp_potrs";
    CIMRGP_REQUIRE(l_dev && workspace_dev && rhs_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && ldl >= n, fn, "bad dimensions");
    DISPATCH(dtype, fn,
             potrs_run<float>((const float*)l_dev, n, ldl, (const float*)workspace_dev, (float*)rhs_dev, q, (float*)z_dev,
                              (float*)scratch_dev, false, S(stream)),
             potrs_run<double>((const double*)l_dev, n, ldl, (const double*)workspace_dev, (double*)rhs_dev, q,
                               (double*)z_dev, (double*)scratch_dev, false, S(stream)));
}

int cimrgp_solve_lt(int dtype, const void* l_dev, int64_t n, int64_t ldl, const void* workspace_dev, void* z_dev, int q,
                    void* scratch_dev, void* stream)
{
    const char* fn = "cimrgp_solve_lt";
    CIMRGP_REQUIRE(l_dev && workspace_dev && z_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && ldl >= n, fn, "bad dimensions");
    DISPATCH(dtype, fn,
             potrs_run<float>((const float*)l_dev, n, ldl, (const float*)workspace_dev, (float*)z_dev, q, nullptr,
                              (float*)scratch_dev, true, S(stream)),
             potrs_run<double>((const double*)l_dev, n, ldl, (const double*)workspace_dev, (double*)z_dev, q, nullptr,
                               (double*)scratch_dev, true, S(stream)));
}

int cimrgp_trsm_rows(int dtype, const void* l_dev, int64_t n, int64_t ldl, const void* workspace_dev, void* b_dev,
                     int64_t m, int64_t ldb, void* stream)
{
    const char* fn = "cimrgp_trsm_rows";
    CIMRGP_REQUIRE(l_dev && workspace_dev && b_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && m >= 0 && ldl >= n && ldb >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(ld_ok(dtype, ldl) && ld_ok(dtype, ldb), fn, "leading dimensions must be multiples of 16 bytes");
    CIMRGP_REQUIRE(aligned16(l_dev) && aligned16(b_dev) && aligned16(workspace_dev), fn, "pointers must be 16-byte aligned");
    if (n == 0 || m == 0) return 0;
    DISPATCH(dtype, fn,
             solve_rows_run<float>((const float*)l_dev, n, ldl, (const float*)workspace_dev, (float*)b_dev, m, ldb, S(stream)),
             solve_rows_run<double>((const double*)l_dev, n, ldl, (const double*)workspace_dev, (double*)b_dev, m, ldb,
                                    S(stream)));
}

int cimrgp_predict_mean(int dtype, const void* x_dev, int64_t n, int d, const void* alpha_dev, int q, const void* xs_dev,
                        int64_t ns, double ell, double sf2, const void* bias_dev, void* mean_dev, int accumulate,
                        void* stream)
{
    const char* fn = "cimrgp_predict_mean";
    CIMRGP_REQUIRE(x_dev && alpha_dev && xs_dev && mean_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && ns >= 0, fn, "negative size");
    DISPATCH(dtype, fn,
             predict_mean_run<float>((const float*)x_dev, n, d, (const float*)alpha_dev, q, (const float*)xs_dev, ns, ell,
                                     sf2, (const float*)bias_dev, (float*)mean_dev, accumulate, S(stream)),
             predict_mean_run<double>((const double*)x_dev, n, d, (const double*)alpha_dev, q, (const double*)xs_dev, ns,
                                      ell, sf2, (const double*)bias_dev, (double*)mean_dev, accumulate, S(stream)));
}

int cimrgp_predict_from_w(int dtype, const void* w_dev, int64_t ns, int64_t n, int64_t ldw, const void* z_dev, int q,
                          double sf2, double extra_var, const void* extra_var_dev, const void* bias_dev, void* mean_dev,
                          void* var_dev, int accumulate, void* stream)
{
    const char* fn = "cimrgp_predict_from_w";
    CIMRGP_REQUIRE(w_dev, fn, "null pointer");
    CIMRGP_REQUIRE(ns >= 0 && n >= 0 && ldw >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(mean_dev == nullptr || z_dev != nullptr, fn, "mean requested without z");
    DISPATCH(dtype, fn,
             predict_from_w_run<float>((const float*)w_dev, ns, n, ldw, (const float*)z_dev, q, sf2, extra_var,
                                       (const float*)extra_var_dev, (const float*)bias_dev, (float*)mean_dev,
                                       (float*)var_dev, accumulate, S(stream)),
             predict_from_w_run<double>((const double*)w_dev, ns, n, ldw, (const double*)z_dev, q, sf2, extra_var,
                                        (const double*)extra_var_dev, (const double*)bias_dev, (double*)mean_dev,
                                        (double*)var_dev, accumulate, S(stream)));
}

int cimrgp_block_stats(int dtype, const void* y_dev, const void* fbar_dev, int64_t n, int q, void* stats_dev, void* stream)
{
    const char* fn = "cimrgp_block_stats";
    CIMRGP_REQUIRE(y_dev && stats_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             misc_block_stats<float>((const float*)y_dev, (const float*)fbar_dev, n, q, (float*)stats_dev, S(stream)),
             misc_block_stats<double>((const double*)y_dev, (const double*)fbar_dev, n, q, (double*)stats_dev, S(stream)));
}

int cimrgp_residual(int dtype, const void* y_dev, const void* fbar_dev, const void* bias_dev, int64_t n, int q,
                    void* r_dev, void* stream)
{
    const char* fn = "cimrgp_residual";
    CIMRGP_REQUIRE(y_dev && r_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             misc_residual<float>((const float*)y_dev, (const float*)fbar_dev, (const float*)bias_dev, n, q, (float*)r_dev,
                                  S(stream)),
             misc_residual<double>((const double*)y_dev, (const double*)fbar_dev, (const double*)bias_dev, n, q,
                                   (double*)r_dev, S(stream)));
}

int cimrgp_train_mean(int dtype, const void* r_dev, const void* alpha_dev, const void* bias_dev, const void* noise_dev,
                      int64_t n, int q, void* out_dev, int accumulate, void* stream)
{
    const char* fn = "cimrgp_train_mean";
    CIMRGP_REQUIRE(r_dev && alpha_dev && noise_dev && out_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             misc_train_mean<float>((const float*)r_dev, (const float*)alpha_dev, (const float*)bias_dev,
                                    (const float*)noise_dev, n, q, (float*)out_dev, accumulate, S(stream)),
             misc_train_mean<double>((const double*)r_dev, (const double*)alpha_dev, (const double*)bias_dev,
                                     (const double*)noise_dev, n, q, (double*)out_dev, accumulate, S(stream)));
}

int cimrgp_add_diag(int dtype, void* k_dev, int64_t n, int64_t ldk, const void* noise_dev, void* stream)
{
    const char* fn = "cimrgp_add_diag";
    CIMRGP_REQUIRE(k_dev && noise_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             misc_add_diag<float>((float*)k_dev, n, ldk, (const float*)noise_dev, S(stream)),
             misc_add_diag<double>((double*)k_dev, n, ldk, (const double*)noise_dev, S(stream)));
}

int cimrgp_noise_from_stats(int dtype, const void* stats_dev, int q, double frac, double floor_value, void* noise_dev,
                            void* stream)
{
    const char* fn = "cimrgp_noise_from_stats";
    CIMRGP_REQUIRE(stats_dev && noise_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             misc_noise_from_stats<float>((const float*)stats_dev, q, frac, floor_value, (float*)noise_dev, S(stream)),
             misc_noise_from_stats<double>((const double*)stats_dev, q, frac, floor_value, (double*)noise_dev, S(stream)));
}

int cimrgp_logdet_half(int dtype, const void* l_dev, int64_t n, int64_t ldl, double* out_dev, void* stream)
{
    const char* fn = "cimrgp_logdet_half";
    CIMRGP_REQUIRE(l_dev && out_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             misc_logdet_half<float>((const float*)l_dev, n, ldl, out_dev, S(stream)),
             misc_logdet_half<double>((const double*)l_dev, n, ldl, out_dev, S(stream)));
}

int cimrgp_syrk_lower(int dtype, void* c_dev, int64_t ldc, const void* a_dev, int64_t lda, int64_t n, int64_t k, void* stream)
{
    const char* fn = "cimrgp_syrk_lower";
    CIMRGP_REQUIRE(c_dev && a_dev, fn, "null pointer");
    CIMRGP_REQUIRE(n >= 0 && k >= 0 && ldc >= n && lda >= k && k < (1ll << 31), fn, "bad dimensions");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    DISPATCH(dtype, fn,
             gemm_nt_sub<float>((float*)c_dev, ldc, (const float*)a_dev, lda, (const float*)a_dev, lda, n, n, (int)k, true, S(stream)),
             gemm_nt_sub<double>((double*)c_dev, ldc, (const double*)a_dev, lda, (const double*)a_dev, lda, n, n, (int)k, true,
                                 S(stream)));
}

int cimrgp_lml_grad(int dtype, const void* x_dev, int64_t n, int d, const void* kinv_dev, int64_t ldk, const void* alpha_dev,
                    int q, double ell, double sf2, double noise, double* out_dev, double* scratch_dev, void* stream)
{
    const char* fn = "cimrgp_lml_grad";
    CIMRGP_REQUIRE(x_dev && kinv_dev && alpha_dev && out_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(ldk >= n, fn, "bad dimensions");
    DISPATCH(dtype, fn,
             lml_grad_run<float>((const float*)x_dev, n, d, (const float*)kinv_dev, ldk, (const float*)alpha_dev, q, ell, sf2,
                                 noise, out_dev, scratch_dev, S(stream)),
             lml_grad_run<double>((const double*)x_dev, n, d, (const double*)kinv_dev, ldk, (const double*)alpha_dev, q, ell,
                                  sf2, noise, out_dev, scratch_dev, S(stream)));
}

int cimrgp_lml_grad_ard(int dtype, const void* xs_dev, int64_t n, int d, const void* kinv_dev, int64_t ldk,
                        const void* alpha_dev, int q, double sf2, double noise, double* out_dev, double* scratch_dev,
                        void* stream)
{
    const char* fn = "cimrgp_lml_grad_ard";
    CIMRGP_REQUIRE(xs_dev && kinv_dev && alpha_dev && out_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(ldk >= n, fn, "bad dimensions");
    DISPATCH(dtype, fn,
             lml_grad_run<float>((const float*)xs_dev, n, d, (const float*)kinv_dev, ldk, (const float*)alpha_dev, q, 1.0, sf2,
                                 noise, out_dev, scratch_dev, S(stream), true),
             lml_grad_run<double>((const double*)xs_dev, n, d, (const double*)kinv_dev, ldk, (const double*)alpha_dev, q, 1.0,
                                  sf2, noise, out_dev, scratch_dev, S(stream), true));
}

size_t cimrgp_lml_grad_scratch_bytes(int64_t n)
{
    if (n <= 0) return 0;
    const int64_t tm = (n + 63) / 64;
    return (size_t)(tm * (tm + 1) / 2) * 10 * sizeof(double);      // 10 = the ARD record (sf | 8 dimensions | trace)
}

int cimrgp_laplace_basis(int dtype, const void* x_dev, int64_t n, int d, const double* interval_dev, int m, void* phi_dev,
                         void* stream)
{
    const char* fn = "cimrgp_laplace_basis";
    CIMRGP_REQUIRE(x_dev && interval_dev && phi_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             laplace_basis_run<float>((const float*)x_dev, n, d, interval_dev, m, (float*)phi_dev, S(stream)),
             laplace_basis_run<double>((const double*)x_dev, n, d, interval_dev, m, (double*)phi_dev, S(stream)));
}

size_t cimrgp_basis_moments_scratch_bytes(int64_t n, int m, int q)
{
    if (n <= 0) return 0;
    return (size_t)basis_moments_workgroups(n) * (size_t)(m * q + 2 * m + q + 2) * sizeof(double);
}

int cimrgp_basis_moments(int dtype, const void* x_dev, int64_t n, int d, const double* interval_dev, int m, const void* y_dev,
                         const void* fbar_dev, const void* fvar_dev, const double* eau_dev, int q, double* out_dev,
                         double* scratch_dev, void* stream)
{
    const char* fn = "cimrgp_basis_moments";
    CIMRGP_REQUIRE(x_dev && interval_dev && y_dev && eau_dev && out_dev && scratch_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             basis_moments_run<float>((const float*)x_dev, n, d, interval_dev, m, (const float*)y_dev, (const float*)fbar_dev,
                                      (const float*)fvar_dev, eau_dev, q, out_dev, scratch_dev, S(stream)),
             basis_moments_run<double>((const double*)x_dev, n, d, interval_dev, m, (const double*)y_dev,
                                       (const double*)fbar_dev, (const double*)fvar_dev, eau_dev, q, out_dev, scratch_dev,
                                       S(stream)));
}

int cimrgp_basis_apply(int dtype, const void* x_dev, int64_t n, int d, const double* interval_dev, int m, const double* eau_dev,
                       int q, const double* bias_dev, const double* c2_dev, double bias_var, void* mean_dev, void* var_dev,
                       int accumulate, void* stream)
{
    const char* fn = "cimrgp_basis_apply";
    CIMRGP_REQUIRE(x_dev && interval_dev && eau_dev, fn, "null pointer");
    DISPATCH(dtype, fn,
             basis_apply_run<float>((const float*)x_dev, n, d, interval_dev, m, eau_dev, q, bias_dev, c2_dev, bias_var,
                                    (float*)mean_dev, (float*)var_dev, accumulate, S(stream)),
             basis_apply_run<double>((const double*)x_dev, n, d, interval_dev, m, eau_dev, q, bias_dev, c2_dev, bias_var,
                                     (double*)mean_dev, (double*)var_dev, accumulate, S(stream)));
}

int cimrgp_layer_fit(int dtype, const void* x_dev, const void* y_dev, const void* fbar_dev, void* train_out_dev,
                     const int64_t* starts_dev, int batch, int64_t n, int d, int q, double ell, double sf2, double noise_fixed,
                     double noise_frac, double noise_floor, const void* shared_bias_dev, const void* shared_noise_dev,
                     void* k_arena_dev, int64_t ldk, int64_t k_stride, void* ws_arena_dev, size_t ws_stride_bytes,
                     int32_t* info_dev, void* rows_arena_dev, int64_t ldr, void* z_dev, void* alpha_dev, void* bias_dev,
                     void* noise_dev, void* scratch_dev, void* stream)
{
    const char* fn = "cimrgp_layer_fit";
    CIMRGP_REQUIRE(x_dev && y_dev && train_out_dev && starts_dev && k_arena_dev && ws_arena_dev && info_dev && rows_arena_dev &&
                   z_dev && alpha_dev && bias_dev && noise_dev && scratch_dev, fn, "null pointer");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(batch >= 1 && n >= 1 && ldk >= n && ldr >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(ld_ok(dtype, ldk) && ld_ok(dtype, ldr) && ld_ok(dtype, k_stride), fn, "leading dimensions and strides must be multiples of 16 bytes");
    CIMRGP_REQUIRE(k_stride >= n * ldk - (ldk - n), fn, "matrix stride too small");
    CIMRGP_REQUIRE(aligned16(k_arena_dev) && aligned16(ws_arena_dev) && aligned16(rows_arena_dev), fn, "pointers must be 16-byte aligned");
    CIMRGP_REQUIRE(ws_stride_bytes >= cimrgp_potrf_workspace_bytes(dtype, n) && ws_stride_bytes % 16 == 0, fn,
                   "workspace stride too small or misaligned");
    CIMRGP_REQUIRE(ell > 0.0 && sf2 > 0.0, fn, "kernel parameters must be positive");
    DISPATCH(dtype, fn,
             layer_fit_typed<float>(x_dev, y_dev, fbar_dev, train_out_dev, starts_dev, batch, n, d, q, ell, sf2, noise_fixed,
                                    noise_frac, noise_floor, shared_bias_dev, shared_noise_dev, k_arena_dev, ldk, k_stride,
                                    ws_arena_dev, ws_stride_bytes, info_dev, rows_arena_dev, ldr, z_dev, alpha_dev, bias_dev,
                                    noise_dev, scratch_dev, S(stream)),
             layer_fit_typed<double>(x_dev, y_dev, fbar_dev, train_out_dev, starts_dev, batch, n, d, q, ell, sf2, noise_fixed,
                                     noise_frac, noise_floor, shared_bias_dev, shared_noise_dev, k_arena_dev, ldk, k_stride,
                                     ws_arena_dev, ws_stride_bytes, info_dev, rows_arena_dev, ldr, z_dev, alpha_dev, bias_dev,
                                     noise_dev, scratch_dev, S(stream)));
}

int cimrgp_layer_predict(int dtype, const void* x_dev, const int64_t* starts_dev, int64_t n, int d, const void* xs_dev,
                         const int64_t* t_starts_dev, int64_t ns, int batch, double ell, double sf2, const void* l_arena_dev,
                         int64_t ldl, int64_t l_stride, const void* ws_arena_dev, size_t ws_stride_bytes, const void* z_dev, int q,
                         const void* bias_dev, const void* noise_dev, void* w_arena_dev, int64_t ldw, int64_t w_stride,
                         void* mean_dev, void* var_dev, void* stream)
{
    const char* fn = "cimrgp_layer_predict";
    CIMRGP_REQUIRE(x_dev && starts_dev && xs_dev && t_starts_dev && l_arena_dev && ws_arena_dev && z_dev && w_arena_dev &&
                   mean_dev && var_dev, fn, "null pointer");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(batch >= 1 && n >= 0 && ns >= 0 && ldl >= n && ldw >= n, fn, "bad dimensions");
    CIMRGP_REQUIRE(ld_ok(dtype, ldl) && ld_ok(dtype, ldw) && ld_ok(dtype, l_stride) && ld_ok(dtype, w_stride), fn,
                   "leading dimensions and strides must be multiples of 16 bytes");
    CIMRGP_REQUIRE(w_stride >= ns * ldw - (ldw - n), fn, "W stride too small");
    CIMRGP_REQUIRE(aligned16(l_arena_dev) && aligned16(ws_arena_dev) && aligned16(w_arena_dev), fn, "pointers must be 16-byte aligned");
    CIMRGP_REQUIRE(ws_stride_bytes >= cimrgp_potrf_workspace_bytes(dtype, n), fn, "workspace stride too small");
    DISPATCH(dtype, fn,
             layer_predict_typed<float>(x_dev, starts_dev, n, d, xs_dev, t_starts_dev, ns, batch, ell, sf2, l_arena_dev, ldl,
                                        l_stride, ws_arena_dev, ws_stride_bytes, z_dev, q, bias_dev, noise_dev, w_arena_dev, ldw,
                                        w_stride, mean_dev, var_dev, S(stream)),
             layer_predict_typed<double>(x_dev, starts_dev, n, d, xs_dev, t_starts_dev, ns, batch, ell, sf2, l_arena_dev, ldl,
                                         l_stride, ws_arena_dev, ws_stride_bytes, z_dev, q, bias_dev, noise_dev, w_arena_dev, ldw,
                                         w_stride, mean_dev, var_dev, S(stream)));
}

int cimrgp_set_rows_queues(int queues)
{
    if (queues != 1 && queues != 2) return fail("cimrgp_set_rows_queues", "queues must be 1 or 2");
    g_rows_queues.store(queues, std::memory_order_relaxed);
    return 0;
}

int cimrgp_get_rows_queues(void) { return rows_queues(); }

int cimrgp_shutdown(void) { const int rc = potrf_shutdown(); staged_shutdown_(); return rc; }

int cimrgp_tuning_build(void)
{
#ifdef CIMRGP_TUNING
    return 1;
#else
    return 0;
#endif
}

int cimrgp_profile_begin(void) { return profile_begin(); }

int cimrgp_profile_pause(void) { return cimrgp::profile_pause(); }

int cimrgp_profile_collect(double* total_ms, double* total_flops, int64_t* launches)
{
    return profile_collect(total_ms, total_flops, launches);
}

int cimrgp_profile_collect_bytes(double* total_ms, double* total_flops, double* total_bytes, int64_t* launches)
{
    return profile_collect(total_ms, total_flops, launches, total_bytes);
}

}  // extern "C"
