// Shared device/host helpers for the gfx950 dense-GP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cimrgp.h"

namespace cimrgp {

// ---------------------------------------------------------------- errors ----
void set_error(const std::string& msg);
int  fail(const char* fn, const char* what);
int  check_hip(hipError_t e, const char* fn, const char* what);

#define CIMRGP_REQUIRE(cond, fn, what) \
    do { if (!(cond)) return ::cimrgp::fail(fn, what); } while (0)
#define CIMRGP_LAUNCH_CHECK(fn) \
    do { hipError_t e__ = hipGetLastError(); \
         if (e__ != hipSuccess) return ::cimrgp::check_hip(e__, fn, "kernel launch"); } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#ifdef CIMRGP_STAMP   // diagnostic builds only (tools/diag_probe.hip): phase stamps of workgroup 0
__device__ long long g_stamp[32];
#define STAMP(n) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_stamp[n] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPW(n, wv) do { if (threadIdx.x == 64 * (wv) && blockIdx.x == 0) g_stamp[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(n) do { } while (0)
#define STAMPW(n, wv) do { } while (0)
#endif

// ------------------------------------------------------------ MFMA traits ----
// 16x16x4 matrix-core tiles, one 8-byte "k-slot" per lane per operand:
//   lane l supplies A[row = l & 15][kslot = l >> 4] and B^T[col = l & 15][kslot = l >> 4].
// f64: a slot is one double  -> one v_mfma_f64_16x16x4_f64.
// f32: a slot is two floats  -> two v_mfma_f32_16x16x4_f32 (k order inside the
//      slot is irrelevant because A and B use the same slot->k map).
template <typename T> struct Mx;

template <> struct Mx<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static constexpr int EPC = 2;     // elements per 16-byte chunk
    static constexpr int EPS = 1;     // elements per 8-byte k-slot
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 r
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
    static __device__ __forceinline__ uint2 neg(uint2 a) { a.y ^= 0x80000000u; return a; }
    static __device__ __forceinline__ acc_t mma(uint2 a, uint2 b, acc_t c) {
        double da = __hiloint2double((int)a.y, (int)a.x);
        double db = __hiloint2double((int)b.y, (int)b.x);
        return __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, c, 0, 0, 0);
    }
    // c - a b: the FP64 matrix-core instruction negates an operand itself (on gfx940+ the builtin's last argument is
    // neg:[A, B, C] for v_mfma_f64_*: the assembler prints `neg:[1,0,0]`), so no vector instruction is spent on the sign
    // (round 5: the update kernels flipped the sign bit of every A fragment with a v_xor -- 16 per stage and wave, each
    // taking a vector issue slot the next multiply had to wait for).  Bit-identical: negation is exact.
    static __device__ __forceinline__ acc_t mma_neg(uint2 a, uint2 b, acc_t c) {
        double da = __hiloint2double((int)a.y, (int)a.x);
        double db = __hiloint2double((int)b.y, (int)b.x);
        return __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, c, 0, 0, 1);
    }
};

template <> struct Mx<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static constexpr int EPC = 4;
    static constexpr int EPS = 2;
    // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 (lane >> 4) + r
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
    static __device__ __forceinline__ uint2 neg(uint2 a) { a.x ^= 0x80000000u; a.y ^= 0x80000000u; return a; }
    static __device__ __forceinline__ acc_t mma(uint2 a, uint2 b, acc_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        return c;
    }
    // c - a b (the FP32 instruction has no operand negation: its last field selects lane groups): flip the sign bits
    static __device__ __forceinline__ acc_t mma_neg(uint2 a, uint2 b, acc_t c) { return mma(neg(a), b, c); }
};

template <typename T>
static __device__ __forceinline__ typename Mx<T>::acc_t acc_zero() {
    typename Mx<T>::acc_t z = {0, 0, 0, 0};
    return z;
}

// ------------------------------------------------------- LDS-only barriers ----
// lds_barrier: waits for this wave's LDS traffic, not for global stores in flight (a plain __syncthreads() also
// drains vmcnt, i.e. every store's round trip).  A real fence pair restricted to the LDS address space around
// s_barrier, NOT inline assembly (round 4): the instructions are the same (s_waitcnt lgkmcnt(0); s_barrier), but the
// compiler does not treat an assembly statement's "memory" clobber as an access to __shared__ arrays whose address
// never leaves the kernel -- it kept a value read from LDS two barriers earlier in registers across such a statement
// (potrf.hip, the pivot loop) and may move LDS accesses across one.  Fences are what __syncthreads() is made of and
// do order such accesses.
static __device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// lds_settle: a wave that has just WRITTEN words other waves will read behind the next barrier reads one of them back
// first (the last one it wrote, or any LDS word: the LDS executes one wave's accesses in issue order, so the load's
// data can only return once every earlier store of the wave has been performed).  Round 5: measured on gfx950 with a
// second workgroup on the compute unit (tools/lab/race_probe.hip, HISTORY.md): the LAST one or two ds_write
// instructions a wave issued ahead of `s_waitcnt lgkmcnt(0); s_barrier` were not yet in the LDS array when another
// wave's ds_read, issued right behind the barrier, read their words -- it got, bit for bit, what the words held before
// (1.7-3 % of panel chains beside an FP32 trailing update; 0 of 1999 with the read-back).  The wait for a STORE's
// lgkmcnt does not imply visibility to other waves there; the wait for a LOAD's data does.
template <typename T>
static __device__ __forceinline__ void lds_settle(const T* written)
{
    const volatile T* p = written;
    const T v = *p;
    asm volatile("" :: "v"(v));          // the value is "used": the wait for it stays ahead of whatever follows
}

// lds_barrier_nowait: the barrier WITHOUT the wait for this wave's LDS traffic, for the one place that needs it: a
// buffer every wave has finished READING (each read's value has been consumed by an instruction ahead of the barrier,
// so the read itself is complete) is rewritten behind the barrier, while reads of ANOTHER buffer may stay in flight
// across it.  The compiler-only fences (no instruction) keep the compiler from moving or caching LDS accesses across
// the barrier; the hardware orders the rest: a wave's LDS write issued behind s_barrier cannot overtake a read
// another wave completed ahead of it.
static __device__ __forceinline__ void lds_barrier_nowait()
{
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_barrier();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// Zero the elements of a 16-byte chunk whose k index is >= kvalid.
template <typename T>
static __device__ __forceinline__ uint4 mask_chunk(uint4 v, int kfirst, int kvalid);
template <>
__device__ __forceinline__ uint4 mask_chunk<double>(uint4 v, int kfirst, int kvalid) {
    if (kfirst     >= kvalid) { v.x = 0; v.y = 0; }
    if (kfirst + 1 >= kvalid) { v.z = 0; v.w = 0; }
    return v;
}
template <>
__device__ __forceinline__ uint4 mask_chunk<float>(uint4 v, int kfirst, int kvalid) {
    if (kfirst     >= kvalid) v.x = 0;
    if (kfirst + 1 >= kvalid) v.y = 0;
    if (kfirst + 2 >= kvalid) v.z = 0;
    if (kfirst + 3 >= kvalid) v.w = 0;
    return v;
}

// Flip the sign of every element of a 16-byte chunk.
template <typename T> static __device__ __forceinline__ uint4 neg_chunk(uint4 v);
template <> __device__ __forceinline__ uint4 neg_chunk<double>(uint4 v) { v.y ^= 0x80000000u; v.w ^= 0x80000000u; return v; }
template <> __device__ __forceinline__ uint4 neg_chunk<float>(uint4 v)
{
    v.x ^= 0x80000000u; v.y ^= 0x80000000u; v.z ^= 0x80000000u; v.w ^= 0x80000000u;
    return v;
}

// ---------------------------------------------------------------- knobs ----
// Thresholds of the factorisation's schedule.  In the product build these are the constants below and
// nothing reads the environment.  Built with -DCIMRGP_TUNING (tools/build_tuning.sh: the measurement
// builds behind profiles/ and tools/sweep_*.sh) each CIMRGP_* environment variable named here overrides
// its default once, at the first use.
struct Knobs {
    int chain_mode = 0;                    // CIMRGP_CHAIN = split | wide | quad: 1 round-1 links / 2 nine-wave / 3 four-wave (0: by context)
    int64_t tail_below = 4864;             // CIMRGP_TAIL_BELOW: trailing matrix at or below this: finish on one queue
    int64_t rows_start_below = 6144;       // CIMRGP_ROWS_START: carried rows start once the trailing matrix is smaller
    int64_t rows_start_below_early = 5632; // CIMRGP_ROWS_START_EARLY: ... in a factorisation that started with early panels (round 5, one box: 6144 / 5632 -> 144.1 / 145.4, 144.5 / 145.1, 144.3 / 145.2; a lone factorisation: 136.8 / 134.4 the other way)
    int64_t head_first_above = 1ll << 30;  // CIMRGP_HEAD_FIRST: bulk update waits for the head above this (off)
    int64_t far_pair_above = 8192;         // CIMRGP_FAR_PAIR: far part updated once per group of panels above this
    int fused_head0 = 1;                   // CIMRGP_HEAD0: first diagonal block of a panel takes its head update itself
    int gemm_pers = 256;                   // CIMRGP_GEMM_PERS: persistent trailing update on at most this many compute units (0: off)
    int gemm_pers_f32 = 0;                 // CIMRGP_GEMM_PERS_F32: the persistent update for FP32 too (8-stage passes; built and bit-checked in round 4, not faster than the tile kernel: 3.85 against 3.64 ms for potrf n = 8192, 16.5 against 16.3 at 16 384)
    int pers_min_tiles = 512;              // CIMRGP_PERS_MIN_TILES: 128-tiles below which the tile-per-workgroup kernel is used
    int rows_fused_tail = 0;               // CIMRGP_ROWS_FUSED: carried rows catch up at the tail switch, then ride in the chain's launches
    int64_t rows_pair_above = 8192;        // CIMRGP_ROWS_PAIR: the carried rows' far updates take two panels at a time (K = 512) while more columns remain
    int64_t fused_max_chain_wgs = 768;     // CIMRGP_FUSED_MAX: one-queue sweeps ride their updates in the chain's launches while batch x n / 32 is at most this
    int batch_halves_min = 8;              // CIMRGP_BATCH_HALVES: a batched factorisation of at least this many blocks (that does not ride) runs as two halves on two queues
    int rows_step = 1;                     // CIMRGP_ROWS_STEP: carried rows' chain as one launch per panel (k_rows_step: previous panel's update + 256-wide solve; 0: update and k_trsm256 as two launches)
    int rider_lean = 1;                    // CIMRGP_RIDER_LEAN: one-queue sweeps give rounds of K = 256 riders to the links first (0: every launch starts with one round)
    int rider_round_us = 25;               // CIMRGP_RIDER_ROUND: modelled duration of one round of K = 256 rider tiles (us)
    int rows_cus = 192;                    // CIMRGP_ROWS_CUS: compute units of the carried rows' far updates (persistent kernel; 0: tile-per-workgroup kernel).
                                           // Round 5, one box (profiles/r05_knob_scan.txt): 128 / 160 / 176 / 192 / 208 / 224 / 256 -> 132.0 / 135.8 / 135.8 / 136.8 / 135.4 / 133.4 /
                                           // 130.4 posteriors/s -- two persistent workgroups fill a unit's LDS, and the chain's workgroups need units that hold at most one
    int trsm_group = 1;                    // CIMRGP_TRSM_GROUP: batched launches solve 4 row tiles per workgroup (0: one)
    int64_t rows_beside_tail_below = 2560; // CIMRGP_ROWS_BESIDE: with carried rows, the factorisation's tail (trailing matrix at most this) is the one-queue fused sweep while the rows keep their own queues (0: look-ahead to the end)
    int tail_far_cus = 160;                // CIMRGP_TAIL_FAR_CUS: compute units of FAR(prev) on the second queue of the fused tail (0: always riders)
    int64_t tail_far_min_rows = 2048;      // CIMRGP_TAIL_FAR_MIN: ... while at least this many rows remain beyond the next panel
    int chain_cus = 32;                    // CIMRGP_CHAIN_CUS: compute units the bulk update leaves to the panel chain (look-ahead phase)
    int pers_max_chunks = 2;               // CIMRGP_PERS_CHUNKS: the persistent update takes K up to this many panels (256 columns each) in one pass per tile (1: K = 256 only, as in rounds 3-4;
                                           // round 5, one box: potrf n = 12 288 / 16 384 13.80 / 28.51 -> 13.45 / 27.79 ms with 2-3; n = 32 768, whose groups are K = 768: 191.8 -> 200.1 ms with 3 -- so 2)
    int early_first_panel = 1;             // CIMRGP_EARLY_PANEL: a staged call whose front end ran on the context's chain queue factors its first panel there, in queue order (0: behind the factorisation's stream)
    int early_panels = 8;                  // CIMRGP_EARLY_PANELS: ... and this many further panels (update + next panel) one-queue style on the chain queue before the
                                           // look-ahead schedule takes over.  Round 5, one box (profiles/r05_early_panels.txt): 0 / 2 / 4 / 6 / 8 / 10 / 12 / 14 ->
                                           // 137.7 / 139.8 / 140.5 / 143.1 / 143.6 / 143.8 / 141.5 / 138.8 posteriors/s
    int early_cus = 256;                   // CIMRGP_EARLY_CUS: compute units of those early updates (0: the look-ahead phase's share; no chain of their own factorisation runs beside them: 224 -> 256: 144.4 -> 145.0, 145.1 -> 146.1)
    int heads_beside_rows = 0;             // CIMRGP_HEADS_ROWS: the combined head + bulk launch also while the carried rows are running (round 5, rows on 192 units: 137.2 -> 135.4 / 134.3 posteriors/s: off)
    int post_final = 0;                    // CIMRGP_POST_FINAL: the look-ahead's chain posts "panel final" in a device word and a gate on the update's queue waits for it
                                           // (0: an event between the queues, rounds 1-4).  Round 5: potrf n = 8192 5.43 -> 5.38 ms, the step unchanged -- and OFF, because
                                           // rocprofv3's counter passes serialise kernels in the order their QUEUES become ready, not in submission order: the gate, first in
                                           // its queue the moment the previous update ends, is run before the chain's remaining kernels and k_post three deep in theirs, and
                                           // waits out its 2-s watchdog.  (The chain's own gate never meets that: the update it waits for is ready long before it.)
    int head_direct_max_rounds = 0;        // CIMRGP_HEAD_DIRECT: a head-first persistent update of at most this many rounds stores its head tiles at the end of their own pass (0: always streamed under the next pass)
    int pers_flex_cus = 0;                 // CIMRGP_PERS_FLEX: a persistent update may take up to this many units beyond its share when that saves a whole round of tiles ...
    int pers_flex_min_rounds = 7;          // CIMRGP_PERS_FLEX_MIN: ... of a launch of at least this many rounds.  OFF: measured in round 5 (profiles/r05_flex_scan.txt): the updates
                                           // get 4-8 % shorter (5 of the 83 rounds of a look-ahead phase at n = 8192 go), the step 2-3 % LONGER -- the panel chain needs its 32 units
};
const Knobs& knobs();
// Queues for the carried rows of cimrgp_potrf_rows (1 or 2): cimrgp_set_rows_queues in include/cimrgp.h.
int rows_queues();
// Pairs of full panels the backward solve takes in one step (build_invT builds their off-diagonal inverse blocks,
// potrs_run uses them): a rule of n alone, so that the factorisation and the solve agree without any state.
static inline int64_t bwd_pairs(int64_t n) { return (n > 5120) ? (n / CIMRGP_NB) / 2 : 0; }
// (potrf.hip) the look-ahead context's queue that is idle between two factorisations on st: cimrgp_solve_queue
hipStream_t solve_queue_for(hipStream_t st);
// (potrf.hip) the context's queue that falls idle before a factorisation on st ends: cimrgp_front_queue
hipStream_t front_queue_for(hipStream_t st);

// --------------------------------------------------------- host launchers ----
// potrf.hip
template <typename T> int potrf_run(T* k, int64_t n, int64_t ld, T* ws, int32_t* info, T* b, int64_t m, int64_t ldb,
                                    hipStream_t st, hipStream_t ready_on = nullptr);
// `batch` equal-sized factorisations in the same launches (strides in elements / ints)
struct PotrfBatch { int count = 1; int64_t sk = 0, sws = 0, sb = 0; };
template <typename T> int potrf_batched_run(T* k, int64_t n, int64_t ld, T* ws, int32_t* info, T* b, int64_t m, int64_t ldb,
                                            PotrfBatch bt, hipStream_t st);
template <typename T> int solve_rows_run(const T* l, int64_t n, int64_t ld, const T* ws, T* b, int64_t m,
                                         int64_t ldb, hipStream_t st, PotrfBatch bt = PotrfBatch());
int potrf_shutdown();     // destroys the look-ahead contexts (streams, events): cimrgp_shutdown
int profile_begin();
int profile_pause();
int profile_collect(double* total_ms, double* total_flops, int64_t* launches, double* total_bytes = nullptr);
// gemm_nt.hip
// A batch of independent products in one launch (equal shapes; element strides between problems).
// skip_first: the first 64 x 64 tile of a rectangular update is left alone (the chain has already
// replaced it by its factor; only honoured when the launch uses 64-tiles: gemm_uses_tile64)
// pers: compute units for the persistent form of the update (-1: knobs().gemm_pers, 0: never)
// head_first + flag: the look-ahead's combined head + bulk update as one persistent launch (gemm_nt.hip)
struct GemmBatch { int count = 1; int64_t sc = 0, sa = 0, sb = 0; int skip_first = 0; int pers = -1; int head_first = 0; int* flag = nullptr; int pers_force = 0; };
bool gemm_uses_tile64(int64_t m, int64_t n, bool lower, int count = 1);
int gemm_pers_head_tiles(int64_t m, int k, int elem_bytes);
template <typename T> int gemm_nt_sub(T* c, int64_t ldc, const T* a, int64_t lda, const T* b, int64_t ldb,
                                      int64_t m, int64_t n, int k, bool lower, hipStream_t st, GemmBatch bt = GemmBatch());

// gram.hip
template <typename T> int rbf_gram_run(const T* xa, int64_t na, const T* xb, int64_t nb, int d, double ell, double sf2,
                                       double diag_add, T* k, int64_t ld, bool symm, bool lower_only, hipStream_t st);
template <typename T> int rbf_gram_batched_run(const T* xa, const int64_t* a_starts, int64_t na, const T* xb, const int64_t* b_starts,
                                               int64_t nb, int d, double ell, double sf2, const T* diag_dev, T* k, int64_t ld,
                                               int64_t kstride, int batch, bool symm, hipStream_t st);
template <typename T> int predict_mean_run(const T* x, int64_t n, int d, const T* alpha, int q, const T* xs, int64_t ns,
                                           double ell, double sf2, const T* bias, T* mean, int accumulate, hipStream_t st);
// solve.hip
template <typename T> int potrs_run(const T* l, int64_t n, int64_t ld, const T* ws, T* rhs, int q, T* z_out, T* scratch,
                                    bool backward_only, hipStream_t st, PotrfBatch bt = PotrfBatch(), bool work_ready = false);
template <typename T> int predict_from_w_run(const T* w, int64_t ns, int64_t n, int64_t ldw, const T* z, int q, double sf2,
                                             double extra, const T* extra_dev, const T* bias, T* mean, T* var, int accumulate,
                                             hipStream_t st, int batch = 1, const int64_t* t_starts = nullptr, int64_t sw = 0);
// misc.hip
template <typename T> int misc_block_stats(const T* y, const T* fbar, int64_t n, int q, T* stats, hipStream_t st);
template <typename T> int misc_residual(const T* y, const T* fbar, const T* bias, int64_t n, int q, T* r, hipStream_t st);
template <typename T> int misc_train_mean(const T* r, const T* alpha, const T* bias, const T* noise, int64_t n, int q,
                                          T* out, int accumulate, hipStream_t st);
template <typename T> int misc_add_diag(T* k, int64_t n, int64_t ld, const T* noise, hipStream_t st);
template <typename T> int misc_noise_from_stats(const T* stats, int q, double frac, double floor_value, T* noise, hipStream_t st);
template <typename T> int misc_logdet_half(const T* l, int64_t n, int64_t ld, double* out, hipStream_t st);
template <typename T> int lml_grad_run(const T* x, int64_t n, int d, const T* kinv, int64_t ld, const T* alpha, int q,
                                       double ell, double sf2, double noise, double* out3, double* scratch, hipStream_t st,
                                       bool ard = false);

// layer.hip: one call per layer for a batch of equal-sized blocks (strides in elements)
template <typename T> struct LayerFit {
    const T* x; const T* y; const T* fbar; T* train_out; const int64_t* starts;
    int batch; int64_t n; int d; int q;
    double ell, sf2, noise_fixed, noise_frac, noise_floor;
    const T* shared_bias; const T* shared_noise;
    T* k; int64_t ldk, sk; T* ws; int64_t sws; int32_t* info;
    T* rows; int64_t ldr, srows;
    T* z; T* alpha; T* bias; T* noise; T* scratch;
};
template <typename T> int layer_fit_run(const LayerFit<T>& a, hipStream_t st);
template <typename T> struct LayerPredict {
    const T* x; const int64_t* starts; int64_t n; int d;
    const T* xs; const int64_t* t_starts; int64_t ns; int batch;
    double ell, sf2;
    const T* l; int64_t ldl, sl; const T* ws; int64_t sws;
    const T* z; int q; const T* bias; const T* noise;
    T* w; int64_t ldw, sw;
    T* mean; T* var;
};
template <typename T> int layer_predict_run(const LayerPredict<T>& a, hipStream_t st);

// reduced.hip
template <typename T> int laplace_basis_run(const T* x, int64_t n, int d, const double* interval, int m, T* phi, hipStream_t st);
int basis_moments_workgroups(int64_t n);
template <typename T> int basis_moments_run(const T* x, int64_t n, int d, const double* interval, int m, const T* y, const T* fbar,
                                            const T* fvar, const double* eau, int q, double* out, double* scratch, hipStream_t st);
template <typename T> int basis_apply_run(const T* x, int64_t n, int d, const double* interval, int m, const double* eau, int q,
                                          const double* bias, const double* c2, double bias_var, T* mean, T* var, int accumulate,
                                          hipStream_t st);

}  // namespace cimrgp
