/* blsq.h — C-ABI of the MI355X trust-region step solver (libblsq_hip.so).
 *
 * Drop-in boundary for ONE path of nmayorov/bounded-lsq: the per-iteration
 * linear algebra its `trf` and `dogbox` drivers run between two fun/jac
 * callbacks.  The reference is pure Python; what a maintainer would bind is
 * this library through ctypes.  Each entry point cites
 * the reference code it replaces (paths under the reference repo).
 *
 * Conventions
 *   - all arrays float64, C-contiguous, batch-major: J is B x m x n exactly as
 *     numpy hands it over, vectors are B x n (or B x m for f), masks int64;
 *   - the caller owns every buffer passed in; outputs are caller-allocated;
 *     the library owns device memory and the factor state inside a plan;
 *   - return value: 0 ok; <0 invalid argument (-(index of the argument),
 *     1-based); >0 a hipError_t.  blsq_last_error() gives the text.  Nothing
 *     throws across the ABI.  Per-problem numerical conditions the reference
 *     signals with ValueError (trust_region.py:28-29,34-35) are reported in
 *     status[b] (BLSQ_STATUS_*);
 *   - a ctx is bound to one device and one HIP stream and is NOT thread-safe;
 *     calls taking host pointers are blocking, `_dev` calls are asynchronous
 *     on the ctx stream (blsq_sync to wait).
 */
#ifndef BLSQ_H
#define BLSQ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct blsq_ctx blsq_ctx;
typedef struct blsq_trf_plan blsq_trf_plan;
typedef struct blsq_dogbox_plan blsq_dogbox_plan;

enum {
  BLSQ_STATUS_OK = 0,
  BLSQ_STATUS_ZERO_DIRECTION = 1,  /* intersect_trust_region: "`s` is zero."  (trust_region.py:28-29) */
  BLSQ_STATUS_OUTSIDE_TR = 2       /* "`x` is not within the trust region."  (trust_region.py:34-35) */
};
enum {                     /* how `scale` is treated by *_factor */
  BLSQ_SCALE_GIVEN = 0,    /* numeric `scaling`: scale = 1/scaling, unchanged          */
  BLSQ_SCALE_JAC_INIT = 1, /* scale = 1/||J[:,j]||, zero norm -> 1   (trf.py:216-219, dogbox.py:141-144) */
  BLSQ_SCALE_JAC_UPDATE = 2/* scale = min(scale, 1/||J[:,j]||)        (trf.py:239-242, dogbox.py:165-168) */
};

int blsq_version(void);
int blsq_device_count(void);

int blsq_ctx_create(int device_id, blsq_ctx** out);
int blsq_ctx_destroy(blsq_ctx* ctx);
const char* blsq_last_error(const blsq_ctx* ctx);
int blsq_sync(blsq_ctx* ctx);

/* Run-time switches: ONE table inside the library (csrc/blsq_options.cpp; INTEGRATION.md lists it).  A ctx reads every
 * switch ONCE, at blsq_ctx_create, from the environment variable the table names; afterwards only blsq_ctx_set_option
 * changes it, for that ctx alone.  Route switches (gram, cqr2, csne, optimistic, no_svdfree, svdfree_min_n, gram_k2_max)
 * take effect for plans created afterwards, the others at the next call.  `name` is the table's key or its environment
 * variable's name.  No switch loosens a proven gate (gram_k2_max can only tighten).  blsq_option_info: i in
 * [0, blsq_option_count()). */
int blsq_option_count(void);
int blsq_option_info(int i, const char** name, const char** env, double* default_value, const char** doc);
int blsq_ctx_set_option(blsq_ctx* ctx, const char* name, double value);
int blsq_ctx_get_option(const blsq_ctx* ctx, const char* name, double* value);

/* device memory helpers (so hosts need not bind the HIP runtime themselves) */
int blsq_dev_malloc(blsq_ctx* ctx, size_t bytes, void** dptr);
int blsq_dev_free(blsq_ctx* ctx, void* dptr);
int blsq_memcpy_h2d(blsq_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Page-locked host memory for the buffers handed to the host-pointer calls (blsq_trf_factor, blsq_dogbox_factor):
 * a `jac` callback (least_squares.py:367-371) that writes J into such a buffer has it DMA'd to the GPU at link
 * speed, in sub-batches of problems overlapped with the Gram of the previous sub-batch; pageable buffers work too
 * (the runtime stages them). */
int blsq_host_alloc(blsq_ctx* ctx, size_t bytes, void** hptr);
int blsq_host_free(blsq_ctx* ctx, void* hptr);
int blsq_memcpy_d2h(blsq_ctx* ctx, void* dst, const void* src, size_t bytes);

/* per-kernel device timing (HIP events on the ctx stream, for bench.py).  on: 0 off; 1 every kernel slot (two events
 * around every launch: they cost the step they measure about 2 %); 2 + s: slot s only (its index in blsq_timing_get) */
int blsq_timing_enable(blsq_ctx* ctx, int on);
int blsq_timing_reset(blsq_ctx* ctx);
int blsq_timing_count(const blsq_ctx* ctx);                    /* number of kernel slots */
int blsq_timing_get(blsq_ctx* ctx, int slot, const char** name, double* total_ms,
                    int64_t* launches);

/* ------------------------------------------------------------------ TRF --
 * blsq_trf_factor  replaces trf.py:244-277 (+ :216-219 / :239-242 for 'jac'):
 *   g = J^T f, Coleman-Li v/d/g_h/diag_h, g_norm, theta, and the factorisation
 *   the reference gets from svd(J_augmented) (trf.py:264-274).  Once per outer
 *   iteration.
 * blsq_trf_step    replaces trf.py:284-308 (+ the norm/correction terms of
 *   :318-331 the driver needs): solve_lsq_trust_region, feasibility test,
 *   reflected / gradient steps, model evaluation and selection, x_new.  Once
 *   per inner iteration with a new Delta / alpha, WITHOUT refactorising.
 */
int blsq_trf_plan_create(blsq_ctx* ctx, int B, int m, int n, blsq_trf_plan** out);
int blsq_trf_plan_destroy(blsq_trf_plan* plan);

int blsq_trf_factor(blsq_trf_plan* plan, const double* J, const double* f, const double* x,
                    const double* lb, const double* ub, double* scale_io, int scale_mode,
                    double* g /*B*n*/, double* g_norm /*B*/, double* theta /*B*/);
int blsq_trf_step(blsq_trf_plan* plan, const double* Delta /*B*/, double* alpha_io /*B*/,
                  double active_rtol, double* step_h /*B*n*/, double* step /*B*n*/,
                  double* x_new /*B*n*/, int64_t* hits /*B*n*/, int64_t* active_new /*B*n*/,
                  double* predicted_reduction /*B*/, double* step_h_norm /*B*/,
                  double* correction /*B*/, int32_t* n_iter /*B*/, int32_t* branch /*B*/,
                  int32_t* status /*B*/);

/* device-pointer variants: inputs already resident in HBM, results stay in
 * plan-owned device buffers until fetched.
 * blsq_trf_factor_dev (and blsq_dogbox_factor_dev, with blsq_dogbox_step_dev) is asynchronous all the way: it does not wait for the per-problem verdict of the
 * conditioning certificate (which problems must take the Householder route) but assumes the common one
 * — "none" — and the NEXT call on the plan reads the verdict: blsq_trf_step_dev enqueues its kernels
 * first and, should the guess have been wrong, runs the Householder stage and the step once more
 * (same results either way).  LIFETIME RULE for the caller: dJ / df / dscale_io must stay valid AND
 * unmodified until the verdict has been read — by the next blsq_*_step_dev / blsq_*_fetch_* call on the
 * plan, or by blsq_sync(ctx), which resolves every pending verdict of the ctx (so does blsq_dev_free and
 * blsq_memcpy_h2d: memory handed back to, or overwritten through, the library is never read afterwards).
 * After blsq_sync nothing of the caller's J / f is read again — EXCEPT by blsq_trf_step_dev while the plan holds problems
 * on the CSNE tier (ill-conditioned TRF problems of 80 <= n <= 256; blsq_debug_csne_stats below): their steps are
 * corrected against J itself, so dJ / df must then stay valid and unmodified until the next factor call on the plan
 * (option `csne` = 0 restores the unconditional rule).  A second *_factor_dev on the plan drops
 * the first one's verdict (it is still read for the path statistics).  Option `optimistic` = 0 (environment
 * BLSQ_OPTIMISTIC=0 at blsq_ctx_create): the factor call waits itself and the rule is void. */
int blsq_trf_factor_dev(blsq_trf_plan* plan, const double* dJ, const double* df,
                        const double* dx, const double* dlb, const double* dub,
                        double* dscale_io, int scale_mode);
int blsq_trf_step_dev(blsq_trf_plan* plan, const double* dDelta, const double* dalpha_in,
                      double active_rtol);
int blsq_trf_fetch_factor(blsq_trf_plan* plan, double* g, double* g_norm, double* theta,
                          double* scale, double* sing /*B*n, unsorted, may be NULL*/);
int blsq_trf_fetch_step(blsq_trf_plan* plan, double* alpha_out, double* step_h, double* step,
                        double* x_new, int64_t* hits, int64_t* active_new,
                        double* predicted_reduction, double* step_h_norm, double* correction,
                        int32_t* n_iter, int32_t* branch, int32_t* status,
                        double* p_h_tr /*may be NULL*/, double* to_bound /*may be NULL*/,
                        int32_t* choice /*may be NULL*/);

/* diagnostics: in-kernel phase stamps of the QR kernel (only a -DBLSQ_QR_STAMPS
 * build writes them): 8 doubles per workgroup into the given device buffer */
int blsq_debug_qr_stamps(void* dbuf);

/* diagnostics: 1 where the last factor call chose the SVD-free trust-region path
 * (full-rank gate passed), 0 where it went through the Jacobi SVD */
int blsq_trf_debug_fast(blsq_trf_plan* plan, int32_t* fast /*B*/);

/* diagnostics: the conditioning certificate of the last factor call, per problem: the PROVEN upper
 * bound K2 >= kappa_2 of the equilibrated system the step is solved from on the normal-equations
 * path (0 where no bound was computed: front end off, or a Cholesky pivot failed before).  A
 * problem stays on that path iff K2 <= 2.5e5. */
int blsq_trf_debug_cond(blsq_trf_plan* plan, double* k2 /*B*/);
int blsq_dogbox_debug_cond(blsq_dogbox_plan* plan, double* k2 /*B*/);

/* diagnostics: Jacobi sweeps used by the last factor call, per problem */
int blsq_trf_debug_sweeps(blsq_trf_plan* plan, int32_t* sweeps /*B*/);

/* --------------------------------------------------------------- dogbox --
 * blsq_dogbox_factor replaces dogbox.py:165-199: gradient, active/free split,
 *   gtol quantity, Gauss-Newton step lstsq(J_free,-f) and the Cauchy step.
 * blsq_dogbox_step   replaces dogbox.py:203-220 (+ :235): dogleg_step /
 *   constrained_cauchy_step, predicted reduction (with the reference's
 *   not-recomputed-Js quirk, :216), scatter to x_new, new on_bound.
 */
int blsq_dogbox_plan_create(blsq_ctx* ctx, int B, int m, int n, blsq_dogbox_plan** out);
int blsq_dogbox_plan_destroy(blsq_dogbox_plan* plan);

int blsq_dogbox_factor(blsq_dogbox_plan* plan, const double* J, const double* f,
                       const double* x, const double* lb, const double* ub, double* scale_io,
                       int scale_mode, const int64_t* on_bound, double* g /*B*n*/,
                       uint8_t* active_set /*B*n*/, double* g_norm /*B*/,
                       int32_t* all_active /*B*/);
int blsq_dogbox_step(blsq_dogbox_plan* plan, const double* Delta /*B*/, double* step /*B*n*/,
                     double* x_new /*B*n*/, int64_t* on_bound_new /*B*n*/, uint8_t* tr_hit /*B*/,
                     double* predicted_reduction /*B*/, double* step_scaled_norm /*B*/,
                     uint8_t* fallback /*B*/, int32_t* status /*B*/);

int blsq_dogbox_factor_dev(blsq_dogbox_plan* plan, const double* dJ, const double* df,
                           const double* dx, const double* dlb, const double* dub,
                           double* dscale_io, int scale_mode, const int64_t* don_bound);
int blsq_dogbox_step_dev(blsq_dogbox_plan* plan, const double* dDelta);
int blsq_dogbox_fetch_factor(blsq_dogbox_plan* plan, double* g, uint8_t* active_set,
                             double* g_norm, int32_t* all_active, double* scale,
                             double* newton_full /*B*n, may be NULL*/,
                             double* cauchy_full /*B*n, may be NULL*/);
int blsq_dogbox_fetch_step(blsq_dogbox_plan* plan, double* step, double* x_new,
                           int64_t* on_bound_new, uint8_t* tr_hit,
                           double* predicted_reduction, double* step_scaled_norm,
                           uint8_t* fallback, int32_t* status);

/* ------------------------------------------------------ tall problem / comm --
 * Row-block partition of ONE very tall problem across the GPUs of a node, one process (rank) per
 * GPU (the reference has no counterpart — it would call svd on the whole matrix,
 * trf.py:272).  The collective is the library's own: RCCL over xGMI, bound at run time (dlopen),
 * enqueued on the ctx stream between the local and the replicated kernels.
 *
 *   blsq_comm_get_id   rank 0 makes the rendezvous id (ncclUniqueId, blsq_comm_id_bytes() bytes);
 *                      the HOST distributes it to the other ranks (any channel: a TCP socket,
 *                      MPI, a file, torch.distributed — bounded_lsq._multi has a socket helper)
 *   blsq_comm_init     every rank, collectively: communicator of `nranks` ranks on the ctx device
 *   blsq_tsqr_plan_create / blsq_tsqr_factor_dev / blsq_trf_step_dev / blsq_trf_fetch_step
 *
 * blsq_tsqr_factor_dev takes THIS rank's row block [J_r f_r] and leaves the same factor state on
 * every rank (so every rank computes the same step, redundantly):
 *   - normal-equations front end: local Gram, ONE ncclAllReduce(sum) of the (n+1)^2 Gram (n = 128:
 *     166 KB: "all-reduce of R" in north_star's words), replicated Cholesky + conditioning gate;
 *   - a problem the gate rejects: local Householder TSQR, ONE ncclAllGather of the (n+1)^2
 *     triangles, every rank merges the stack.
 * m_total (all ranks' rows) enters the reference's rank test eps * m * s[0] (trust_region.py:109).
 * blsq_tsqr_local_dev / blsq_tsqr_combine_dev expose the two halves of the Householder route for
 * hosts that exchange the triangles themselves (`tri` buffers: dense row-major
 * blsq_tsqr_tri_ld(n)^2 doubles, stack in rank order).
 * Every rank takes the SAME route: the gate's verdict is compared over the ranks (one 16-byte
 * ncclAllReduce(max) behind the Gram's) and the call fails with BLSQ_ERR_RANKS_DISAGREE on every rank
 * if they differ (x / bounds / scale / scale_mode / BLSQ_* environment must be identical everywhere).
 * Return codes of failed RCCL calls: 10000 + ncclResult_t.
 * The collective library is resolved at the first blsq_comm_* call: $BLSQ_RCCL_PATH if set (a full
 * path to a librccl-compatible shared object; tests/ use it for a socket-based stand-in that lets two
 * processes share ONE GPU, which RCCL itself refuses), otherwise librccl.so.1 next to the HIP runtime
 * the process already uses, then the loader's search path.  blsq_comm_library() reports the choice.
 */
#define BLSQ_ERR_RANKS_DISAGREE 20001
/* path of the collective library in use ("" before the first blsq_comm_* call) and its
 * ncclGetVersion() code (0 if the library has no such entry) */
const char* blsq_comm_library(int* version_out);
int blsq_comm_id_bytes(void);
int blsq_comm_get_id(blsq_ctx* ctx, void* id_out, size_t bytes);
int blsq_comm_init(blsq_ctx* ctx, int nranks, int rank, const void* id, size_t bytes);
int blsq_comm_destroy(blsq_ctx* ctx);
int blsq_comm_size(const blsq_ctx* ctx);
int blsq_comm_rank(const blsq_ctx* ctx);
/* max over the ranks of n <= 64 host doubles, in place; doubles as a barrier (bench timing) */
int blsq_comm_allreduce_max(blsq_ctx* ctx, double* host_io, int n);

int blsq_tsqr_tri_ld(int n);
int blsq_tsqr_plan_create(blsq_ctx* ctx, int m_local, long long m_total, int n, int nranks,
                          blsq_trf_plan** out);
int blsq_tsqr_factor_dev(blsq_trf_plan* plan, const double* dJ_block, const double* df_block,
                         const double* dx, const double* dlb, const double* dub,
                         double* dscale_io, int scale_mode);
int blsq_tsqr_local_dev(blsq_trf_plan* plan, const double* dJ_block, const double* df_block,
                        double* dtri_out);
int blsq_tsqr_combine_dev(blsq_trf_plan* plan, const double* dtri_stack /*nranks tris*/,
                          const double* dx, const double* dlb, const double* dub,
                          double* dscale_io, int scale_mode);

/* diagnostics: number of 16-column QR panels factored by the Cholesky-QR + Householder-
 * reconstruction fast path (out[0]) and by the exact Householder column loop (out[1]: partial
 * last panels and panels whose scaled Gram has a small pivot) since the last reset. */
int blsq_debug_cqr_stats(blsq_ctx* ctx, uint64_t out[2], int reset);
/* Diagnostics: of the problems the gate handed on (out[1] of blsq_debug_gram_stats), how many the CholeskyQR2
 * middle tier factored (second pass over J through the MFMA pipe, proven acceptance test)
 * instead of the Householder tree, since the last reset.  Synchronises the ctx stream. */
int blsq_debug_cqr2_stats(blsq_ctx* ctx, uint64_t* out1, int reset);
/* Diagnostics of the CSNE tier (corrected semi-normal equations, DESIGN.md 3.0d: a rejected problem of 80 <= n <= 256
 * keeps its Gram-Cholesky factor as a preconditioner and every step-solve corrects the trust-region solution against J
 * itself in ONE streaming pass; replaces what the reference gets from svd(J_augmented), trf.py:272-274 +
 * trust_region.py:111-150, for such a problem): out[0] = problems factor calls routed to the tier, out[1] = step-solves
 * it delivered, out[2] = step-solves it declined (measured correction above its bound: the problem went on to
 * CholeskyQR2 / the Householder tree in that step call), since the last reset.
 * LIFETIME RULE of the tier: while a plan holds problems on it (blsq_debug_csne_stats out[0] grows), blsq_trf_step_dev
 * READS the dJ / df of the last blsq_trf_factor_dev: they must stay valid and unmodified until the next factor call on
 * the plan (the reference's drivers keep J for exactly that long, trf.py:283-352).  BLSQ_CSNE=0 switches the tier off. */
int blsq_debug_csne_stats(blsq_ctx* ctx, uint64_t out[3], int reset);
/* ... per problem: on_tier[b] = 1 while problem b is on the tier; eta[b] = the largest first-order correction the last
 * step call measured for it (the quantity its acceptance bounds: <= 1e-7), -1 where the tier declined.  Either may be NULL. */
int blsq_trf_debug_csne(blsq_trf_plan* plan, int32_t* on_tier /*B*/, double* eta /*B*/);
/* Diagnostics of the factorisation front end: out[0] = problems factored by the
 * normal-equations fast path (Gram + equilibrated Cholesky, conditioning-gated), out[1] =
 * problems the gate handed to the Householder TSQR tree, since the last reset. */
int blsq_debug_gram_stats(blsq_ctx* ctx, uint64_t out[2], int reset);

/* Diagnostics: measured peaks of the device the ctx is bound to ("confirm on the
 * box with a copy kernel and an MFMA-f64 probe").
 *   kind 0: FP64 MFMA probe — every SIMD issues independent v_mfma_f64_16x16x4_f64 back to back on
 *           register operands (`arg` waves per SIMD, 1 or 2); out[0] = TFLOP/s, out[1] = number of
 *           MFMA wave-instructions executed, out[2] = milliseconds (HIP events).  The same launch
 *           calibrates the SQ MFMA counters (tools/pmc_mfma.py).
 *   kind 1: streaming copy of `arg` MiB (device to device, 16 B per lane); out[0] = GB/s counting
 *           bytes read + bytes written, out[1] = bytes moved, out[2] = milliseconds. */
int blsq_debug_probe(blsq_ctx* ctx, int kind, int arg, double out[3]);

/* ---- batched outer trust-region drivers, device-resident ------------------
 * Replaces, for B problems of one shape advancing in lock-step, the Python loops around the
 * step path: trf.py:173-237 (initialisation), :238-261 (top of the outer loop: nfev / gtol
 * checks), :309-358 (ratio test, Delta / alpha update, ftol / xtol tests, accept) and
 * dogbox.py:100-163, :164-194, :221-272.  x, f, J and every per-problem scalar stay on the
 * device; only the callbacks' inputs/outputs (device buffers owned by the driver) and ONE
 * integer per call (active / accepted problems) are visible to the host.
 *
 *   create -> buffers -> start(x0,lb,ub,scale,...)
 *   caller:  f  <- fun(x),  J <- jac(x)            (into the driver's buffers)
 *   begin                                          first factorisation, Delta_0
 *   loop:  propose(&n_active)   if n_active == 0: stop
 *          caller: f_trial <- fun(x_trial)
 *          judge(&n_accepted)
 *          caller: J[b] <- jac(x[b]) for the problems with accepted[b] != 0   (if n_accepted)
 *   fetch(...)
 * Per-problem results (x, nfev, njev, status, ...) are what the reference's driver returns for
 * that problem alone; a terminated problem is frozen.  A problem whose step reports a
 * BLSQ_STATUS_* condition (where the reference raises ValueError and aborts the solve,
 * trust_region.py:28-29,34-35) is frozen at its current x with status = -BLSQ_STATUS_*.
 * method: 0 = 'trf', 1 = 'dogbox'. */
typedef struct blsq_outer blsq_outer;
int blsq_outer_create(blsq_ctx* ctx, int method, int B, int m, int n, blsq_outer** out);
int blsq_outer_destroy(blsq_outer* o);
/* Device buffers read / written by the callbacks (any pointer argument may be NULL):
 * x [B][n] current points, x_trial [B][n], f [B][m] residuals at x, f_trial [B][m],
 * J [B][m][n] Jacobians at x, accepted [B] int32 flags set by blsq_outer_judge. */
int blsq_outer_buffers(blsq_outer* o, double** x, double** x_trial, double** f, double** f_trial,
                       double** J, int32_t** accepted);
/* Host inputs, all [B][n]: x0 (as given by the user), x_start (x0 moved strictly inside the
 * bounds for 'trf', trf.py:201; == x0 for 'dogbox'), lb, ub, scale (= 1/scaling, or ones with
 * jac_scaling != 0).  max_nfev > 0.  Uploads and resets the state; afterwards buffer `x` holds
 * x_start. */
int blsq_outer_start(blsq_outer* o, const double* x0, const double* x_start, const double* lb,
                     const double* ub, const double* scale, int jac_scaling, double ftol,
                     double xtol, double gtol, int max_nfev);
int blsq_outer_begin(blsq_outer* o);
int blsq_outer_propose(blsq_outer* o, int32_t* n_active);
int blsq_outer_judge(blsq_outer* o, int32_t* n_accepted);
/* Host outputs (any may be NULL): x [B][n], f [B][m], obj [B], optimality [B], on_bound [B][n]
 * (dogbox's final mask; zeros for 'trf' whose mask the caller derives from x, trf.py:257),
 * nfev, njev, status [B]. */
int blsq_outer_fetch(blsq_outer* o, double* x, double* f, double* obj, double* optimality,
                     int64_t* on_bound, int32_t* nfev, int32_t* njev, int32_t* status);

/* ---- finite-difference Jacobians for the batched drivers, on the device ----
 * jac='2-point' / '3-point' of the reference is the THIRD-PARTY call
 * scipy.optimize._numdiff.approx_derivative(fun, x, rel_step=diff_step, method=jac, f0=f,
 * bounds=bounds) (least_squares.py:357-365; restated against scipy 1.15.3: _compute_absolute_step,
 * _adjust_scheme_to_bounds, _dense_difference).  For B problems at once:
 *   blsq_fd_points_dev    steps h [B][n], one-sided flags [B][n] and the perturbed points
 *                         X [B][P][n] (P = n for method 2, 2n for method 3) which the caller's
 *                         `fun` evaluates in one batched call into F [B][P][m];
 *   blsq_fd_assemble_dev  J [B][m][n] from f0 [B][m] and F (problems with mask[b] == 0 are left
 *                         untouched when mask != NULL).
 * All pointers are device pointers; rel_step is NULL (scipy's default step) or [n].
 * method: 2 = '2-point', 3 = '3-point'. */
int blsq_fd_points_dev(blsq_ctx* ctx, int B, int n, int method, const double* dx,
                       const double* dlb, const double* dub, const double* drel_step, double* dX,
                       double* dh, uint8_t* done_sided);
int blsq_fd_assemble_dev(blsq_ctx* ctx, int B, int m, int n, int method, const double* dx,
                         const double* dh, const uint8_t* done_sided, const double* df0,
                         const double* dF, double* dJ, const int32_t* dmask);

#ifdef __cplusplus
}
#endif
#endif /* BLSQ_H */
