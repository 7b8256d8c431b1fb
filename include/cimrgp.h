/*
 * cimrgp.h -- C ABI of the MI355X (gfx950) dense GP covariance/posterior path.
 *
 * This library replaces, for one (resolution, partition) block, the exact
 * RBF Gaussian-process arithmetic that the reference reaches only through its
 * `RegressionMethod.fit/predict` plugin (src/RegressionInput.py:10-67, class
 * GP_RBF -> third-party GPy) and the per-resolution residual combine of
 * src/Stats.py:126-157 / src/MRGP.py:782-803.  The reference is pure Python;
 * the binding a maintainer adds is a ctypes stub.
 *
 * Conventions
 *  - Every pointer named *_dev is a DEVICE pointer (hipMalloc'd or a
 *    torch.cuda tensor's data_ptr()); `stream` is a hipStream_t passed as
 *    void* (NULL = the null stream).  Calls enqueue work and return; nothing
 *    here synchronises, allocates or frees device memory (graph-capturable).
 *  - Matrices are ROW-MAJOR with an explicit leading dimension in elements.
 *    A symmetric matrix uses its LOWER triangle (row i, column j <= i); the
 *    strictly upper triangle is never read and, after potrf, holds junk.
 *  - dtype: CIMRGP_F32 or CIMRGP_F64 selects the element type of every
 *    matrix/vector argument of that call.  Hyper-parameters are doubles.
 *  - Return value: 0 = enqueued; <0 = argument / runtime error, message via
 *    cimrgp_last_error().  Numerical failure of the factorisation is reported
 *    asynchronously in *info_dev (LAPACK convention: 0 = ok, i > 0 = leading
 *    minor of order i not positive definite; the Python wrapper raises
 *    numpy.linalg.LinAlgError, the exception SanityCheck.py:59-65 keys on).
 */
#ifndef CIMRGP_H
#define CIMRGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CIMRGP_F32 = 0, CIMRGP_F64 = 1 };

/* *info_dev values other than LAPACK's: CIMRGP_INFO_WATCHDOG = the factorisation's internal schedule
 * gave up waiting for one of its own kernels (a bounded device-side wait of 2 s expired: the panel chain
 * waits for tiles of the trailing update through a device counter).  It says nothing about the matrix;
 * the result is undefined.  The Python wrapper raises RuntimeError("schedule watchdog"), NOT the
 * numpy.linalg.LinAlgError that the reference's positive-definiteness guard catches and repairs
 * (src/SanityCheck.py:59-65).  No leading minor can have this order (n < 2^30 is enforced). */
#define CIMRGP_INFO_WATCHDOG 0x7fffffff

/* Outer block size of the factorisation (columns per panel).  The workspace
 * keeps ceil(n/64) inverted 64x64 diagonal blocks followed by ceil(n/256)
 * blocks (L_pp^-1)^T of CIMRGP_NB x CIMRGP_NB. */
#define CIMRGP_NB 256

int         cimrgp_version(void);
const char* cimrgp_last_error(void);
/* Number of visible HIP devices (does not create a context). */
int         cimrgp_device_count(void);

/* ---- D1: RBF Gram builder --------------------------------------------------
 * Replaces GPy.kern.RBF(d, ARD=False).K(X) behind RegressionInput.py:60-61.
 *   K[a][b] = sf2 * exp(-|x_a - x_b|^2 / (2 ell^2)) + (a == b ? diag_add : 0)
 * x_dev: (n x d) row-major.  lower_only != 0 writes only tiles that touch the
 * lower triangle.  d <= 8. */
int cimrgp_rbf_gram(int dtype, const void* x_dev, int64_t n, int d,
                    double ell, double sf2, double diag_add,
                    void* k_dev, int64_t ldk, int lower_only, void* stream);

/* Cross-Gram  Kab[a][b] = k(xa_a, xb_b),  (na x nb) row-major, ld >= nb.
 * Replaces kern.K(X*, X) inside GPy's predict (RegressionInput.py:66-67). */
int cimrgp_rbf_cross(int dtype, const void* xa_dev, int64_t na,
                     const void* xb_dev, int64_t nb, int d,
                     double ell, double sf2,
                     void* kab_dev, int64_t ld, void* stream);

/* ---- D2: blocked in-place Cholesky  K = L L^T (lower) ----------------------
 * Replaces the Cholesky inside GPy's exact-Gaussian inference
 * (RegressionInput.py:61-63).  Right-looking over panels of CIMRGP_NB columns,
 * left-looking over four 64-column sub-blocks inside a panel: diagonal-block
 * factor + inverse (one workgroup), panel solve as a product with the
 * inverted diagonal block (MFMA), trailing SYRK update (MFMA 16x16x4 f64 /
 * f32) with one-panel look-ahead on an internal second stream.
 * workspace: cimrgp_potrf_workspace_bytes(dtype, n) bytes, keeps the inverted
 * diagonal blocks needed by cimrgp_potrs / cimrgp_trsm_rows afterwards (the
 * 64 x 64 and 256 x 256 inverses and, per pair of full panels, the off-diagonal
 * block of the 512 x 512 inverse the backward solve steps through).
 * info_dev: one int32, written asynchronously. */
size_t cimrgp_potrf_workspace_bytes(int dtype, int64_t n);
int cimrgp_potrf(int dtype, void* k_dev, int64_t n, int64_t ldk,
                 void* workspace_dev, size_t workspace_bytes,
                 int32_t* info_dev, void* stream);

/* Factorisation that carries m extra rows through the same panel sweep:
 *   K = L L^T  and  B <- B L^-T   (B: m x n row-major, ldb >= n)
 * in one pass (the row updates fill the compute units the latency-bound panel
 * chain leaves idle).  With B = [K(X*, X); R^T] this yields W = K* L^-T for D5
 * and z^T = (L^-1 R)^T for D3 without a separate forward solve. */
int cimrgp_potrf_rows(int dtype, void* k_dev, int64_t n, int64_t ldk,
                      void* workspace_dev, size_t workspace_bytes,
                      int32_t* info_dev, void* b_dev, int64_t m, int64_t ldb,
                      void* stream);

/* One block's whole posterior in ONE call -- SURVEY section 8b's fused entry: what GPy.models.GPRegression(x, y,
 * RBF) + model.predict(xs) do behind src/RegressionInput.py:60-67 (variance:
 * scripts/tests/GPRBF_vs_ciMRGP_vs_fiMRGP.py:121):
 *   K = sf2 exp(-|x - x'|^2 / (2 ell^2)) + noise I -> L L^T in k_dev (n x ldk, lower), workspace / info_dev as cimrgp_potrf;
 *   W = K(xs, x) L^-T in the first ns rows of w_dev ((ns + q) x ldw; its last q rows hold z^T afterwards);
 *   z_dev (n x q) = L^-1 y, alpha_dev (n x q) = K^-1 y (scratch_dev: 2 q n elements);
 *   mean_dev (ns x q) (+)= W z, var_dev (ns) (+)= sf2 - sum W^2 (+ noise if add_noise).
 * The same kernels as cimrgp_rbf_gram + cimrgp_rbf_cross + cimrgp_potrf_rows + cimrgp_solve_lt +
 * cimrgp_predict_from_w, in that order, with bit-identical results: one call for a ctypes-only caller. */
int cimrgp_block_posterior(int dtype, const void* x_dev, int64_t n, int d, const void* y_dev, int q,
                           const void* xs_dev, int64_t ns, double ell, double sf2, double noise,
                           void* k_dev, int64_t ldk, void* workspace_dev, size_t workspace_bytes,
                           int32_t* info_dev, void* w_dev, int64_t ldw, void* alpha_dev, void* z_dev,
                           void* scratch_dev, void* mean_dev, void* var_dev, int add_noise,
                           int accumulate, void* stream);
/* The same call with its three stages on three streams, for a caller that pipelines INDEPENDENT blocks (the
 * partitions of one layer: the reference's loop over regions, src/Posteriors.py:35-59, has no dependency between
 * its iterations): the front end (Gram matrix, cross-Gram matrix, targets as carried rows) on stream_front, the
 * factorisation with the carried rows on stream (the look-ahead context belongs to this one), backward solve and
 * prediction on stream_solve; each stage starts when the one before it is done (events).  With three buffer sets in
 * rotation the front end of block i+1 and the latency-bound solve of block i-1 run beside the factorisation of block
 * i.  Reuse of a buffer set: behind its factorisation, `stream` waits for the solve stage of the previous call on the
 * same (stream, stream_solve) pair, so with stream_front == stream and TWO sets in rotation the caller orders nothing;
 * in every other arrangement the caller orders the reuse itself (stream_front must wait for the stream_solve work
 * that last read the set).  A BUFFER SET is k_dev, w_dev, workspace_dev, alpha_dev, z_dev and scratch_dev together: the
 * solve stage reads or writes every one of them.  The library keeps one in-flight record per (stream, stream_solve) pair
 * and device; passing ANY buffer of a set whose solve stage may still be running -- one set where two are needed, or a
 * plain cimrgp_block_posterior on the same set right behind a staged call -- is detected: the front end then waits for
 * that solve stage; correct, without overlap.  Results: k_dev is final on stream, alpha / z / mean / var on stream_solve.  Bit-identical to
 * cimrgp_block_posterior; equal streams give exactly that call. */
int cimrgp_block_posterior_staged(int dtype, const void* x_dev, int64_t n, int d, const void* y_dev, int q,
                                  const void* xs_dev, int64_t ns, double ell, double sf2, double noise,
                                  void* k_dev, int64_t ldk, void* workspace_dev, size_t workspace_bytes,
                                  int32_t* info_dev, void* w_dev, int64_t ldw, void* alpha_dev, void* z_dev,
                                  void* scratch_dev, void* mean_dev, void* var_dev, int add_noise,
                                  int accumulate, void* stream_front, void* stream, void* stream_solve);
/* A queue for stream_solve that costs no hardware queue of its own: the queue of `stream`'s look-ahead context that
 * is idle between two factorisations on `stream` (a process is served by four hardware queues and a factorisation with
 * carried rows uses four; a fifth stream for the solve stage made the step 25 % longer, this one 4 % shorter).  The
 * caller orders its own reads of the solve stage's results (an event on *queue_out).  *queue_out = stream when no
 * context can be had for `stream` (then nothing overlaps).  The queue lives until cimrgp_shutdown. */
int cimrgp_solve_queue(void* stream, void** queue_out);
/* A queue for stream_front: the queue of `stream`'s look-ahead context that falls idle before a factorisation on
 * `stream` ends (the last third of a factorisation runs on one queue).  The Gram matrices of the NEXT independent block,
 * enqueued there, run beside that latency-bound tail instead of behind it -- and so does the START of that block's
 * factorisation: a staged call with this queue as stream_front factors its first panel there, in queue order behind its
 * own front end, and, while the previous factorisation on `stream` is still in flight when the call is made, its next
 * eight panels one after the other (update, next panel) before the look-ahead schedule takes over on `stream`.  For
 * consecutive INDEPENDENT blocks only (two buffer sets in rotation); a call that has the machine to itself gains nothing
 * from it.  The caller orders the front end's inputs on *queue_out itself (the staged call orders its own buffer sets).
 * *queue_out = stream when no context can be had. */
int cimrgp_front_queue(void* stream, void** queue_out);
/* `batch` equal-sized factorisations -- the blocks of one layer (independent over regions,
 * Posteriors.py:35-59) -- in the SAME kernel launches: matrix i starts k_stride elements after
 * matrix i-1 (likewise workspace_stride_bytes, b_stride), info_dev holds `batch` int32.  One queue
 * (no look-ahead): a layer of many small blocks is launch- and latency-bound, and batching gives the
 * host one block's worth of launches and the device all the blocks' panel chains side by side.
 * b_dev may be NULL (m = 0). */
int cimrgp_potrf_rows_batched(int dtype, void* k_dev, int64_t n, int64_t ldk, int64_t k_stride,
                              void* workspace_dev, size_t workspace_stride_bytes,
                              int32_t* info_dev, void* b_dev, int64_t m, int64_t ldb,
                              int64_t b_stride, int batch, void* stream);

/* ---- D3: alpha = (L L^T)^-1 R  for q right-hand sides ---------------------
 * Replaces the two triangular solves of GPy's posterior ("woodbury vector").
 * rhs_dev: (n x q) row-major, overwritten with alpha.  z_dev (optional, may
 * be NULL): (n x q) receives z = L^-1 R.  q <= 8.
 * scratch_dev: 2*q*n elements of dtype. */
int cimrgp_potrs(int dtype, const void* l_dev, int64_t n, int64_t ldl,
                 const void* workspace_dev, void* rhs_dev, int q,
                 void* z_dev, void* scratch_dev, void* stream);

/* Backward half only: z_dev (n x q) holds z = L^-1 R and is overwritten with
 * alpha = L^-T z.  scratch_dev: 2*q*n elements. */
int cimrgp_solve_lt(int dtype, const void* l_dev, int64_t n, int64_t ldl,
                    const void* workspace_dev, void* z_dev, int q,
                    void* scratch_dev, void* stream);

/* The backward half for `batch` equal-sized factors in the same launches: z_dev holds batch
 * blocks of (n x q), scratch_dev batch blocks of 2*q*n elements. */
int cimrgp_solve_lt_batched(int dtype, const void* l_dev, int64_t n, int64_t ldl, int64_t l_stride,
                            const void* workspace_dev, size_t workspace_stride_bytes,
                            void* z_dev, int q, void* scratch_dev, int batch, void* stream);

/* Row-wise triangular solve with many right-hand sides (MFMA):
 *   B <- B L^-T      B: (m x n) row-major, ldb >= n,  i.e. row i of B becomes
 * L^-1 b_i.  Used for D5: B = K(X*, X) gives the rows whose squared norms are
 * subtracted from sf2. */
int cimrgp_trsm_rows(int dtype, const void* l_dev, int64_t n, int64_t ldl,
                     const void* workspace_dev, void* b_dev, int64_t m,
                     int64_t ldb, void* stream);

/* ---- D4: fused predictive mean (cross-Gram never stored) ------------------
 *   mean[i][c] (+)= bias[c] + sum_j k(xs_i, x_j) alpha[j][c]
 * Replaces GPy's predict mean (RegressionInput.py:66-67) and the per-region
 * prediction of MRGP.py:794-800.  alpha_dev (n x q), mean_dev (ns x q),
 * bias_dev (q) or NULL.  accumulate != 0 adds to mean_dev (sum over
 * resolutions, MRGP.py:803). */
int cimrgp_predict_mean(int dtype, const void* x_dev, int64_t n, int d,
                        const void* alpha_dev, int q,
                        const void* xs_dev, int64_t ns,
                        double ell, double sf2, const void* bias_dev,
                        void* mean_dev, int accumulate, void* stream);

/* ---- D5 tail: from W = K(X*,X) L^-T  (ns x n, after cimrgp_trsm_rows) ------
 *   var[i]     (+)= sf2 + extra_var + extra_var_dev[0] - sum_j W[i][j]^2
 *   mean[i][c] (+)= bias[c] + sum_j W[i][j] z[j][c]        (if mean_dev)
 * z_dev (n x q) = L^-1 R from cimrgp_potrs.  mean_dev/var_dev may be NULL.
 * extra_var_dev (may be NULL): one device scalar of dtype, e.g. a block's noise
 * variance as left on the device by cimrgp_noise_from_stats -- the y_var term of
 * MRGP.py:907-932 without a host round trip. */
int cimrgp_predict_from_w(int dtype, const void* w_dev, int64_t ns, int64_t n,
                          int64_t ldw, const void* z_dev, int q,
                          double sf2, double extra_var,
                          const void* extra_var_dev, const void* bias_dev,
                          void* mean_dev, void* var_dev, int accumulate,
                          void* stream);

/* ---- D6 / a11: residual chain helpers (Stats.py:126-157, Posteriors.py:68) -
 * Column means of (y - fbar) over rows [0, n):  bias[c], and the population
 * variance of the centred block (all q columns pooled) -> stats_dev[q].
 * stats_dev: q+1 elements. */
int cimrgp_block_stats(int dtype, const void* y_dev, const void* fbar_dev,
                       int64_t n, int q, void* stats_dev, void* stream);
/* r[i][c] = y[i][c] - fbar[i][c] - bias[c]   (targets of one block) */
int cimrgp_residual(int dtype, const void* y_dev, const void* fbar_dev,
                    const void* bias_dev, int64_t n, int q, void* r_dev,
                    void* stream);
/* Training-point prediction without another kernel pass:
 *   K alpha = r - noise * alpha;   out[i][c] (+)= r - noise*alpha + bias[c]
 * noise is read from noise_dev[0] (device scalar of dtype). */
int cimrgp_train_mean(int dtype, const void* r_dev, const void* alpha_dev,
                      const void* bias_dev, const void* noise_dev, int64_t n,
                      int q, void* out_dev, int accumulate, void* stream);
/* Add a device scalar to the diagonal: K[i][i] += noise_dev[0]. */
int cimrgp_add_diag(int dtype, void* k_dev, int64_t n, int64_t ldk,
                    const void* noise_dev, void* stream);
/* noise_dev[0] = max(frac * stats_dev[q], floor)  (RegressionInput.py:62) */
int cimrgp_noise_from_stats(int dtype, const void* stats_dev, int q,
                            double frac, double floor_value, void* noise_dev,
                            void* stream);

/* sum_i log L[i][i]  (for the log marginal likelihood; one element of dtype
 * double written to out_dev regardless of dtype). */
int cimrgp_logdet_half(int dtype, const void* l_dev, int64_t n, int64_t ldl,
                       double* out_dev, void* stream);

/* ---- hyper-parameter optimisation step (RegressionInput.py:63, `.optimize()`) ----
 * C <- C - A A^T on the lower triangle (C: n x n, A: n x k, row-major).  With C = 0 and
 * A = L^-T (cimrgp_trsm_rows applied to the identity) this gives -K^-1. */
int cimrgp_syrk_lower(int dtype, void* c_dev, int64_t ldc, const void* a_dev,
                      int64_t lda, int64_t n, int64_t k, void* stream);
/* Gradient of the log marginal likelihood w.r.t. (log sf, log l, log noise) for
 * K = sf E + noise I:  out_dev[0..2] (doubles) = 1/2 tr((alpha alpha^T - q K^-1) dK/dtheta).
 * kinv_dev: K^-1 in its lower triangle (n x ldk); alpha_dev (n x q); the kernel matrix is
 * re-evaluated from x_dev on the fly.  scratch_dev: cimrgp_lml_grad_scratch_bytes(n). */
size_t cimrgp_lml_grad_scratch_bytes(int64_t n);
int cimrgp_lml_grad(int dtype, const void* x_dev, int64_t n, int d,
                    const void* kinv_dev, int64_t ldk, const void* alpha_dev, int q,
                    double ell, double sf2, double noise, double* out_dev,
                    double* scratch_dev, void* stream);

/* The same for one length-scale PER INPUT DIMENSION (GPy's RBF(ARD=True), the reference's
 * comparison script scripts/tests/GPRBF_vs_ciMRGP_vs_fiMRGP.py:118).  xs_dev holds the inputs
 * already divided by their length-scales (the kernel of scaled inputs has unit length-scale, so
 * every other entry point serves ARD unchanged);  out_dev[0] = d/dlog sf, out_dev[1..d] = d/dlog l_k,
 * out_dev[d+1] = d/dlog noise  (d + 2 doubles). */
int cimrgp_lml_grad_ard(int dtype, const void* xs_dev, int64_t n, int d,
                        const void* kinv_dev, int64_t ldk, const void* alpha_dev, int q,
                        double sf2, double noise, double* out_dev,
                        double* scratch_dev, void* stream);

/* ---- reduced-rank (Laplacian basis) block path of the reference, SURVEY.md 8f rank 2 ----
 * Phi (n x m row-major, ld = m):  phi_i(x) = prod_k L_k^-1/2 sin(pi i (x_k + L_k)/(2 L_k)),
 * i = 1..m  (KernelClass.py:22-37, assembled as in MRGP.py:337-357).  interval_dev: d doubles. */
int cimrgp_laplace_basis(int dtype, const void* x_dev, int64_t n, int d,
                         const double* interval_dev, int m, void* phi_dev, void* stream);
/* The two entry points below never read Phi: they regenerate each row from x and the
 * interval (one sincos per input dimension + a three-term recurrence), so their HBM traffic
 * is x, y, f_bar, f_var only.
 *
 * The N-dependent sums of one block's variational updates (Posteriors.py:298-342,345-372,
 * 396-412), with r0 = y - fbar - Phi E[au]^T (eau_dev: q x m doubles):
 *   out_dev = [ Phi^T r0 (m x q) | colsum Phi (m) | colsum Phi^2 (m) | sum r0 (q) |
 *               sum |r0|^2 (1) | sum fvar (1) ]   (doubles, fixed-order reduction)
 * fbar_dev / fvar_dev may be NULL.  scratch_dev: cimrgp_basis_moments_scratch_bytes(n, m, q). */
size_t cimrgp_basis_moments_scratch_bytes(int64_t n, int m, int q);
int cimrgp_basis_moments(int dtype, const void* x_dev, int64_t n, int d,
                         const double* interval_dev, int m, const void* y_dev,
                         const void* fbar_dev, const void* fvar_dev,
                         const double* eau_dev, int q,
                         double* out_dev, double* scratch_dev, void* stream);
/* mean (n x q) (+)= bias + Phi E[au]^T ;  var (n) (+)= bias_var + Phi^2 c2
 * (Stats.py:316-348, MRGP.py:794-800,847-860).  mean_dev / var_dev / bias_dev / c2_dev may be NULL. */
int cimrgp_basis_apply(int dtype, const void* x_dev, int64_t n, int d,
                       const double* interval_dev, int m,
                       const double* eau_dev, int q, const double* bias_dev,
                       const double* c2_dev, double bias_var, void* mean_dev,
                       void* var_dev, int accumulate, void* stream);

/* ---- one call per layer: a batch of equal-sized blocks --------------------
 * The reference loops over the regions l of a resolution in Python
 * (src/Posteriors.py:35-59 for the fit, src/MRGP.py:782-803 for the
 * prediction); regions are contiguous index ranges (src/Inputs.py:57-60), so
 * block b of the batch is row offset starts_dev[b] (DEVICE array of int64) into
 * the layer's arrays x (N x d), y (N x q), f_bar (N x q or NULL), train_out.
 *
 * cimrgp_layer_fit: for each of `batch` blocks of n points
 *   bias_b  = shared_bias_dev (q values) if given, else the column means of y - f_bar
 *   noise_b = noise_fixed if >= 0, else shared_noise_dev[0] if given, else
 *             max(noise_frac * pooled variance about the column means, noise_floor)
 *             (src/RegressionInput.py:62: labels.var() * 0.01)
 *   K_b = sf2 exp(-|x - x'|^2 / (2 ell^2)) + noise_b I (lower) -> L_b L_b^T in place in
 *   matrix b of k_arena_dev (leading dimension ldk, stride k_stride elements),
 *   with the workspace of cimrgp_potrf at byte stride ws_stride_bytes, info_dev[b]
 *   as cimrgp_potrf; z_b = L_b^-1 r_b (n x q), alpha_b = K_b^-1 r_b (n x q),
 *   r_b = y - f_bar - bias_b; train_out[block rows] += K_noiseless alpha_b + bias_b.
 *   rows_arena_dev (batch x q x ldr) and scratch_dev (batch x 2 q n elements) are
 *   work areas; bias_dev (batch x q) and noise_dev (batch) receive the values used.
 * cimrgp_layer_predict: for each block, at its ns test points (rows
 *   t_starts_dev[b] .. + ns of xs (N* x d), mean (N* x q) and var (N*)):
 *   W_b = K(xs_b, x_b) L_b^-T into matrix b of w_arena_dev (ns x ldw, stride
 *   w_stride), mean += W_b z_b + bias_b, var += sf2 - sum W_b^2 (+ noise_dev[b] if
 *   noise_dev is not NULL).  All blocks of a call share n and ns. */
int cimrgp_layer_fit(int dtype, const void* x_dev, const void* y_dev, const void* fbar_dev,
                     void* train_out_dev, const int64_t* starts_dev, int batch, int64_t n,
                     int d, int q, double ell, double sf2, double noise_fixed,
                     double noise_frac, double noise_floor, const void* shared_bias_dev,
                     const void* shared_noise_dev, void* k_arena_dev, int64_t ldk,
                     int64_t k_stride, void* ws_arena_dev, size_t ws_stride_bytes,
                     int32_t* info_dev, void* rows_arena_dev, int64_t ldr, void* z_dev,
                     void* alpha_dev, void* bias_dev, void* noise_dev, void* scratch_dev,
                     void* stream);
int cimrgp_layer_predict(int dtype, const void* x_dev, const int64_t* starts_dev, int64_t n,
                         int d, const void* xs_dev, const int64_t* t_starts_dev, int64_t ns,
                         int batch, double ell, double sf2, const void* l_arena_dev,
                         int64_t ldl, int64_t l_stride, const void* ws_arena_dev,
                         size_t ws_stride_bytes, const void* z_dev, int q,
                         const void* bias_dev, const void* noise_dev, void* w_arena_dev,
                         int64_t ldw, int64_t w_stride, void* mean_dev, void* var_dev,
                         void* stream);

/* ---- the path's collective ------------------------------------
 * The blocks of a resolution are independent and shard over the GPUs of a node, one process per GPU; what is
 * exchanged is a SUM: at prediction the per-resolution predictions the reference adds up in a Python loop
 * (src/MRGP.py:802-803; variance :902-905) -- every rank accumulates its blocks into a zero-initialised fused
 * [mean | var] buffer and ONE in-place sum over the ranks finishes it -- and during the fit the training-point
 * predictions of the layers coarser than the first one with a region per rank (src/Stats.py:126-157).
 * RCCL over xGMI (ncclAllReduce, sum), loaded at first use (dlopen of librccl.so.1): the library has no link-time
 * dependency on it.  Bootstrap as in RCCL: ONE rank calls cimrgp_comm_unique_id (CIMRGP_COMM_ID_BYTES bytes of HOST
 * memory) and hands the bytes to the others through any channel the caller has (a file, MPI, torch.distributed's
 * store ...); then EVERY rank calls cimrgp_comm_create(world_size, rank, id, &comm) with its HIP device current
 * (collective: returns when all ranks have called it).  cimrgp_allreduce_sum enqueues the in-place sum of `count`
 * elements of `dtype` on `stream` (every rank, same count, same order of calls).  A communicator is used by one
 * thread at a time.  0 = ok; <0 = error (cimrgp_last_error; RCCL's message included). */
#define CIMRGP_COMM_ID_BYTES 128
int cimrgp_comm_unique_id(void* id_out_host);
int cimrgp_comm_create(int world_size, int rank, const void* id_host, void** comm_out);
int cimrgp_comm_destroy(void* comm);
int cimrgp_allreduce_sum(void* comm, int dtype, void* buf_dev, int64_t count, void* stream);

/* ---- process-level policy (no reference counterpart: the reference is one
 * process, src/MRGP.py has no streams) --------------------------------------
 * cimrgp_set_rows_queues: how many low-priority queues cimrgp_potrf_rows may
 * use for the carried rows (2 = default: the rows' own panel chain and their
 * far updates side by side; 1 = one queue, for a process that must stay within
 * four streams -- caller, panel chain, rows, collective -- e.g. one rank of
 * several SHARING a GPU).  Takes effect for
 * factorisations enqueued afterwards; returns 0, or <0 for a value other than
 * 1 or 2.  cimrgp_get_rows_queues returns the current value.
 * cimrgp_tuning_build: 1 when the library was compiled with -DCIMRGP_TUNING
 * (schedule thresholds read from CIMRGP_* environment variables; tools/ only),
 * 0 for the product build, which reads no environment variable.
 * cimrgp_shutdown: destroys the streams and events the factorisations created
 * (one look-ahead context per caller stream and device).  Call it when no call
 * of this library is in flight, before the HIP runtime is torn down -- the
 * Python loader registers it with atexit; later calls re-create what they need. */
int cimrgp_set_rows_queues(int queues);
int cimrgp_get_rows_queues(void);
int cimrgp_tuning_build(void);
int cimrgp_shutdown(void);

/* ---- measurement hooks (bench.py roofline; no reference counterpart) ---------
 * Between begin and collect every lower-triangular trailing-update launch of
 * cimrgp_potrf is bracketed by HIP events on its own stream.  collect waits
 * for them and returns the summed kernel time (ms), the summed ALGORITHMIC
 * flops (M (M+1) K per launch) and the number of launches.
 * HOST pointers.  Not thread-safe; meant for one benchmarking thread. */
int cimrgp_profile_begin(void);
int cimrgp_profile_pause(void);   /* stop recording, keep the records (cimrgp_profile_begin resumes) */
int cimrgp_profile_collect(double* total_ms, double* total_flops, int64_t* launches);
/* the same, plus the summed ALGORITHMIC bytes of those launches: C (lower triangle) read and written
 * once, the K-wide panel read once -- (M (M + 1) + M K) x element size per launch. */
int cimrgp_profile_collect_bytes(double* total_ms, double* total_flops, double* total_bytes,
                                 int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* CIMRGP_H */
