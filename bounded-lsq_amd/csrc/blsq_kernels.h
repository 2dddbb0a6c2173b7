// Host-visible launch interfaces of the HIP kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "blsq_options.h"

namespace blsq {

// ---------------------------------------------------------------- QR ------
struct QrArgs {
  const double* A;        // source, row-major; problem b at A + b*strideA
  long strideA;
  int ldA;                // row stride of the source
  int rowsA;              // rows of the source per problem
  const double* F;        // optional extra (last) column, problem b at F + b*strideF
  long strideF;
  const int* ncols_dev;   // optional per-problem N; nullptr -> N
  const int* batch_list;  // optional compacted problem indices: workgroup y factors problem batch_list[y]
  const int* count_dev;   // optional [1]: the launch is over an upper bound of the list; workgroups y >= *count_dev leave
  const int* require_path;  // optional [B]: problems with require_path[b] == 0 are skipped (they took the Gram path)
  int N;                  // columns to factor (incl. the rhs column if any)
  int rows_per_leaf;      // source rows per leaf (workgroup)
  int RP;                 // padded leaf rows (multiple of 16, >= NPAD)
  int LDP;                // LDS column stride of the panel (set by launch_qr)
  int NPAD;               // leading dimension (and row count) of each R output
  int NPmax;              // panels reserved in the V/T scratch per slot
  double* V;              // scratch [slot][NPmax][RP][16]
  double* T;              // scratch [slot][NPmax][256]
  double* Rout;           // [slot][NPAD][NPAD], slot = b*nleaf + leaf
  // virtual bottom block: source rows >= vdiag_row0 are  vdiag[b] * I  (rhs 0) instead of memory
  int vdiag_row0;         // 0: off
  const double* vdiag;    // [B] per-problem scalar on that diagonal (when vdiag_vec is null)
  const double* vdiag_vec;   // optional [B][stride_vec]: per-column values on that diagonal
  // optional column scaling fused into the panel load: source element (r, c) is A[r][c] *
  // colscale[b][c] for the factored columns (the rhs column is not scaled)
  const double* colscale;    // [B][stride_vec]
  long stride_vec;
  double* dbg;            // diagnostic stamps (nullptr in the product)
  int stack_rows;         // >0: source is a stack of upper-triangular blocks of this many rows
  int cqr;                // 1: Cholesky-QR + Householder-reconstruction panels (set by launch_qr)
  const Options* opt;     // host only: the ctx's switches (nullptr: the table's defaults)
};
// The kernel stages at most QR_MAX_TILES 16-row tiles per workgroup: all RP/16 tiles of a
// dense source, G * ceil((N-1)/16) of a stack of G triangular blocks (qr_panel.hip).
constexpr int QR_MAX_TILES = 64;
inline bool qr_stack_ok(int RP, int stack_rows) {
  return stack_rows > 0 && stack_rows % 16 == 0 && RP % stack_rows == 0 && RP / stack_rows >= 2;
}
int qr_staged_tiles(int RP, int stack_rows, int N);
hipError_t launch_qr(const QrArgs& q, int nleaf, int B, hipStream_t st);
void set_qr_debug_buffer(double* p);
// panels factored by the Cholesky-QR fast path [0] / the Householder column loop [1] so far
hipError_t qr_cqr_stats(unsigned long long out[2], int reset, hipStream_t st);

// ------------------------------------------------- normal-equations fast path ----
// gram_kernels.hip / chol_kernels.hip: G = [J f]^T [J f] by MFMA, equilibrated blocked Cholesky in place in the
// triangle slot, conditioning gate; problems that fail it get fb_mask[b] = n + 1 and are
// factored by the Householder tree instead.
struct GramArgs {
  const double* J; long strideJ; int ldJ;
  const double* F; long strideF;
  int m, n;               // rows / columns of J
  int NPAD;               // row stride (and row count) of each G slot
  double* G;              // [B][chunks][NPAD*NPAD] upper tile blocks
  const int* mask;        // optional: problems with mask[b] <= 1 are skipped
  const int* list;        // optional compacted problem indices: workgroup y works on problem list[y] (the launch
                          // is then over the list's length; a masked launch whose active workgroups alternate
                          // with idle ones lands on a fraction of the XCDs)
  int rows_per_chunk;     // set by launch_gram
  int rhs_valu;           // set by launch_gram: J^T f / f^T f accumulated by the vector ALUs
  double* Gscr;           // set by launch_gram (pair mode): the chunk-partials buffer, used as scratch
  const Options* opt;     // host only: the ctx's switches (nullptr: the table's defaults)
  int src_by_pos;         // 1 (with `list`): J / F are indexed by the LIST POSITION of the workgroup's problem, not by the
                          // problem — the source holds the listed problems only (CholeskyQR2's W, sized by the list)
};
struct GramCholArgs {
  const double* Gsrc;     // [B][NPAD*NPAD] Gram (upper tile blocks); may alias G (in place)
  double* G;              // [B][NPAD*NPAD] out: [R | Q^T f] (upper), zeros elsewhere
  int NPAD, n;
  // optional diagonal modification  H = D Gsrc D + diag(diag_vec^2 + diag_sqrt[b]^2)  on the first
  // n columns (D = diag(colscale)): the trust-region systems of TRF
  const double* colscale;    // [B][stride_vec]
  const double* diag_vec;    // [B][stride_vec]
  const double* diag_sqrt;   // [B]
  long stride_vec;
  const int* batch_list;  // optional compacted problem indices (grid = their number)
  int count;              // problems of this launch (set by launch_gram_chol)
  int skip_zero;          // 1: the part of the slot outside the factor is not written (the caller's buffer holds
                          // zeros there and its readers never look: the Newton-round factors, read by the
                          // triangular solves only) — half of the bytes a factorisation stores are those zeros
  const int* count_dev;   // optional [1]: the launch is over an upper bound; entries beyond *count_dev leave
  int expect;             // > 0: number of problems the launch is expected to work on (kernel choice; default: its size)
  // optional principal sub-matrix (dogbox: free columns ++ rhs): H = Gsrc[idx, idx] with
  // idx = gather[b][0 .. N_b-2] ++ [n], N_b = ncols_dev[b] (0: nothing to do); gather is increasing
  const int* ncols_dev;   // [B]
  const int* gather;      // [B][stride_vec]
  const int* skip_path;   // optional [B]: problems with skip_path[b] != 0 are skipped (they are on the QR path) ...
  const int* qr_mask;     // ... unless qr_mask (optional [B]) says 0: a Newton system of a QR-path problem whose alpha
                          // makes it provably well conditioned is factored here too (LmState::ncols_lm)
  double* hmax;           // optional [B] out: max_j h_jj, the largest diagonal entry of the system factored
  double* lam_out;        // optional [B] out: a proven upper bound on lambda_max of the equilibrated system: n (its
                          // trace) from the factor kernel, min(||R'||_1 ||R'||_inf, ||C||_F, n) once gram_cond_kernel
                          // has seen the problem
  int* path_out;          // optional [B] out (refreshed problems only): n + 1 = Householder tree, 0 = Gram
  const int* mask;        // optional, as above
  int* fb_mask;           // optional [B] out: n + 1 if the problem needs the Householder tree, else 0
  int* fail_count;        // [1] number of such problems (caller zeroes it)
  int* fail_list;         // optional [B] out: their indices, compacted (order of arrival: launch lists only)
  double* dsc;            // optional [B][NPAD] out: column scales 1 / ||J_j||
  double* colinfo;        // optional [B][2] out: min_j sqrt(h_jj), sum_j h_jj over the first n columns
  double* rinv;           // optional [B][NPAD/16][256] out: inverses of the diagonal tiles of R' (row-major)
  // conditioning certificate (launch_gram_gate)
  double* ywork;          // [B][NPAD*NPAD] scratch: Y = R'^-T
  double* k2_out;         // optional [B] out: the proven bound on kappa_2 of the equilibrated system
  double k2_max;          // the gate: a problem stays on the path iff its proven bound is <= k2_max (0: GRAM_K2_MAX);
                          // the host scales it with the accumulation length of the Gram (gram_k2_max)
  double pivot_floor;     // smallest acceptable squared pivot of R' (0: 1 / GRAM_K2_MAX, the early reject of a
                          // gate-bearing factorisation: lambda_min(C) <= min r_jj^2, lambda_max(C) >= 1)
  // Third stage of the certificate (launch_gram_cert_shift): a problem the two norm bounds cannot settle is
  // decided by a Cholesky factorisation of the SHIFTED equilibrated system C - tau I, tau = Lambda / k2_max
  // with Lambda >= lambda_max(C) proven by the norm stage: it runs to completion with pivots above the
  // rounding floor iff lambda_min(C) > tau (up to n eps), i.e. iff kappa_2(C) <= Lambda / tau = k2_max.
  int* cert_flag;         // optional [B]: gram_cond_kernel sets 1 for a problem it leaves undecided (instead of
                          // rejecting it); the shifted launch works on exactly those and clears the flag
  double* cert_tau;       // [B] the shift tau of such a problem (written by gram_cond_kernel)
  int cert_shift;         // 1: THIS launch is the shifted factorisation (G = scratch; verdict -> fb_mask etc.)
  double* cert_ym;        // optional [B] out (N > 80, right-looking factor kernel) / in (stage 0 of the certificate):
  double* cert_r1;        //   max_j (M(R')^-T e)_j and ||R'||_1, accumulated row block by row block while the factor
                          //   is produced (0: not computed — stage 0 then makes its own two passes for them)
  double* cert_open;      // optional [B] (N > 80): what stage 0 leaves for a problem it cannot settle — > 0: Lambda_0 =
                          //   min(||R'||_1 ||R'||_inf, n) >= lambda_max(C): the problem skips the explicit inverse of the norm
                          //   stage and goes to the third stage directly; < 0: hopeless (1 / min r'_jj^2 > k2_max): rejected;
                          //   0: nothing (settled, or stage 0 in its four-pass form)
  int* cert_done;         // optional [B]: 1 = the factor kernel itself proved K2 <= GRAM_K2_MAX (N <= 80: the
                          // register-resident kernel has R' at hand); launch_gram_gate then skips the problem
  // dogbox, N <= 80 (optional; g == nullptr: off): the register-resident kernel also does what
  // dog_gate_solve_kernel does for a problem whose rank the column-norm bound already settles — Cauchy
  // step, Newton step -R_f^-1 c_f, fast flag — and sets done[b]; dog_gate_solve skips those
  struct DogFinish {
    const double* g;      // [B][stride_vec] gradient (full length; gathered through `gather`)
    double* newton;       // [B][stride_vec] out (free order)
    double* cauchy;       // [B][stride_vec] out (free order)
    int* fast;            // [B] out
    int* ncols_jac;       // [B] out
    int* done;            // [B] out: 1 = finished here
    int m, enable;
  } dog;
  // TRF finish (gram_chol_reg_kernel only; fast == nullptr: none): what lm_gate_kernel writes for a
  // problem of this path whose rank the column-norm bound settles (fast flag, singular-value bounds, idle
  // phase)
  struct LmFinish {
    int* fast;            // [B]
    int* ncols_jac;       // [B]
    double* sc;           // [B][16] (SC_SMAX, SC_SMIN)
    int* st;              // [B][4]  (ST_PHASE)
    int m, enable;
  } lmfin;
  // gram_chol_reg_kernel with `dog` or `lmfin`: counter (or null) of the problems of the launch the kernel
  // does NOT settle completely — certified by the first bound AND finished by the block above.  While it
  // stays 0 the certificate / gate / solve launches that follow have nothing to do.  (Counting the
  // exceptions keeps the common case free of atomics: a thousand waves finishing together and adding to
  // one address cost 9 us.)
  int* unsettled;
  double* pmin_out;       // optional [B] out: the smallest squared pivot of R' the factorisation met (the CSNE tier's
                          // own floor is far below the gate's, csne_kernels.hip); not written by a shifted launch
  const Options* opt;     // host only: the ctx's switches (nullptr: the table's defaults)
};
// A problem stays on the normal-equations path only if the PROVEN bound K2 >= kappa_2(R'^T R') of its
// equilibrated system is at most GRAM_K2_MAX (chol_kernels.hip, gram_cond_kernel).  Consequence used
// by the rank gates: lambda_max >= 1 (unit diagonal), so sigma_min(R') >= 1 / sqrt(GRAM_K2_MAX).
#ifndef BLSQ_K2_MAX
#define BLSQ_K2_MAX 2.5e5
#endif
constexpr double GRAM_K2_MAX = BLSQ_K2_MAX;         // (calibration builds: EXTRA_DEFS=-DBLSQ_K2_MAX=...)
constexpr double gram_csqrt(double x, double g = 1.0, int it = 64) {
  return it == 0 ? g : gram_csqrt(x, 0.5 * (g + x / g), it - 1);
}
constexpr double GRAM_SMIN_PROVEN = 1.0 / gram_csqrt(GRAM_K2_MAX);   // = 1 / sqrt(GRAM_K2_MAX): 2e-3
// Rounding-safe floor of the shifted factorisation's squared pivots: a floating-point Cholesky of a unit-
// diagonal matrix of order N <= 272 that completes with every pivot above (N + 1) eps trace = 8e-12 proves
// positive definiteness (Higham, Accuracy and Stability, Thm 10.7 with the a-posteriori bound of Rump 2006)
constexpr double GRAM_CERT_PIVOT_FLOOR = 1.0e-9;
// The gate as a function of the Gram's accumulation length: the entries of G are sums of m products
// accumulated in row chunks (rows_per_chunk terms in the MFMA accumulator, then nchunks partial sums), so
// their rounding error grows like a(m) = sqrt(rows_per_chunk) + sqrt(nchunks); the error constant of
// DESIGN.md 3.0 was calibrated at m = 4096 (a = 46.7) and the gate is tightened by a(4096) / a(m) beyond.
double gram_k2_max(long long m_total, double tighter = 0.0);   // tighter > 0: the gate if it is below the calibrated one
bool gram_supported(int m, int n);
int gram_chunks(int B, int m);
// Gfinal / fused (optional): where the reduced Gram belongs; *fused = true means the launch produced it
// there itself (two chunks summed in the kernel) and launch_gram_reduce must not follow
hipError_t launch_gram(const GramArgs& a, int chunks, int B, hipStream_t s, double* Gfinal = nullptr,
                       bool* fused = nullptr);
hipError_t launch_gram_reduce(const double* Gpart, int chunks, int NPAD, double* Gout,
                              const int* mask, int B, hipStream_t s);
hipError_t launch_gram_chol(const GramCholArgs& a, int B, hipStream_t s);
// (stage0_only: N > 80 and the flag-driven factor kernel's share at hand — ONLY gram_cert0_kernel, which then also finishes
//  the problems it certifies (lmfin) and counts the others (unsettled); BLSQ_CERT0 = 0 is ignored for such a launch)
hipError_t launch_gram_gate(const GramCholArgs& a, int B, hipStream_t s, bool stage0_only = false);
// third stage of the certificate for the problems launch_gram_gate flagged (a.cert_flag / a.cert_tau):
// same source, scalings and gather as the factor call `a` describes, output into a.ywork
hipError_t launch_gram_cert_shift(const GramCholArgs& a, int B, hipStream_t s);

// ------------------------------------------------ CholeskyQR2 middle tier (cqr2_kernels.hip) ----
struct Cqr2Args {
  const double* J; long strideJ; int ldJ;   // [B][m][ldJ] the Jacobians (read in place)
  const double* F; long strideF;            // [B][m]
  int m, n, NPAD;
  const int* list;        // compacted indices of the problems the certificate rejected
  const int* run;         // [B] n + 1: second pass for this problem, 0: not (set by cqr2_prep)
  const double* Y;        // [B][NPAD*NPAD] R1'^-T (lower triangular; gram_cond_kernel's work array)
  const double* dsc;      // [B][NPAD] the equilibration of the first Cholesky: R1' = R1 diag(dsc)
  const double* R1;       // [B][NPAD*NPAD] R1 | c as stored by the first Cholesky
  double* z;              // [B][NPAD] R1^-1 c
  double* Wj; long strideW;   // [list position][m][n]  W = J R1^-1  (sized by the list of rejected problems, not by B)
  double* Wf; long strideWf;  // [list position][m]     f - J R1^-1 c
  int rows_per_wg;        // set by launch_cqr2_apply
};
constexpr double CQR2_K2_MAX = 1.0e12;   // largest proven kappa_2 of the equilibrated J^T J the tier accepts
bool cqr2_supported(int m, int n);
hipError_t launch_cqr2_prep(const Cqr2Args& a, int count, const int* pivot_mask, int* run, hipStream_t s);
hipError_t launch_cqr2_apply(const Cqr2Args& a, int count, hipStream_t s);
// acceptance (Gershgorin on G2 + the second Cholesky's pivots) and R~ = R2 [R c; 0 1] (a.R1: scratch) into
// Rout, the problems' triangle slots; tree_mask[b] = 0 (accepted) / n + 1 (Householder tree); *accepted counts
hipError_t launch_cqr2_combine(const Cqr2Args& a, int count, const int* run, const int* pivot2, const double* G2,
                               const double* R2, double* Rout, int* tree_mask, unsigned long long* accepted,
                               hipStream_t s);

// ------------------------------------- CSNE tier (csne_kernels.hip) ----
// Corrected semi-normal equations for the problems the conditioning certificate keeps off the normal-equations
// path: their Gram-Cholesky factors are kept as PRECONDITIONERS, the cheap Newton iteration on phi(alpha) runs on
// them as for any other problem and records every evaluation; ONE streaming pass over J then yields the residuals
// of all recorded solves, an n-space kernel corrects phi / phi' to first order, replays the scalar iteration and
// corrects the final step.
constexpr int CSNE_MAXE = 8;             // evaluations of phi a problem may record (alpha = 0 and seven Newton rounds)
#ifndef BLSQ_CSNE_K2_MAX
#define BLSQ_CSNE_K2_MAX 2.0e11
#endif
constexpr double CSNE_K2_MAX = BLSQ_CSNE_K2_MAX;   // largest PROVEN kappa_2 of the equilibrated (computed) system the tier takes
constexpr double CSNE_PIVOT_FLOOR = 1.0e-11;   // smallest squared pivot of its factor (> (N + 1) eps trace: proven PD)
#ifndef BLSQ_CSNE_ETA_MAX
#define BLSQ_CSNE_ETA_MAX 1.0e-7
#endif
constexpr double CSNE_ETA_MAX = BLSQ_CSNE_ETA_MAX;   // largest measured first-order correction it accepts (second order
                                                     // ~ 1e2 eta^2; calibration builds: EXTRA_DEFS=-DBLSQ_CSNE_ETA_MAX=...)
struct CsneState {
  int B, m, n, ld;
  const double* J; long strideJ; int ldJ;   // the caller's [J f] of the current factor (read by every step call)
  const double* F; long strideF;
  int* flag;              // [B] 1: on the tier
  int* list;              // [B] their indices, ascending
  int* counts;            // [4]: [0] list length, [1] problems whose acceptance failed in the last step call
  int* fail_list;         // [B] ... and their indices
  int* ne;                // [B]            (LmState::csne_ne)
  double* ralpha;         // [B][CSNE_MAXE] (LmState::csne_alpha)
  double* rvec;           // [B][CSNE_MAXE][3][ld] (LmState::csne_vec)
  double* part;           // [list position][chunks][NE][ld + 8] partial J_h^T (J_h p~ + f) and |J_h w~|^2 of the row chunks
  double* hp;             // [B][ld] out: H p_h (TrfState::csne_hp)
  double* eta;            // [B] out: the largest first-order correction measured (diagnostics / acceptance)
  int rows_per_wg, nchunk, NE;   // set by the host (csne_geometry: functions of m and of the batch's deepest recording)
};
void csne_geometry(int m, int* rows_per_wg, int* nchunk);
int csne_launch_evals(int NE);            // evaluations the pass launch carries for a recording depth of NE (>= NE)
bool csne_supported(int m, int n);
struct LmState;
struct TrfState;
// which of the nfb problems in tree_list (the certificate's rejects; k2 / sel_mask: the bound launch_gram_gate has just
// computed for them with CSNE_K2_MAX) the tier takes: flag, path, rank-gate outputs; tree_list / tree_mask / tree_count
// keep the others; list / counts[0] = every flagged problem of the plan
hipError_t launch_csne_select(const CsneState& cs, const LmState& lm, int nfb, int* tree_list, int* tree_mask,
                              int* tree_count, int* path, const int* sel_mask, const double* k2, const double* pmin,
                              const double* colinfo, hipStream_t s);
// ... the same for a dogbox plan (the free block of problem b has ncols[b] - 1 columns; gelsd's rank test,
// dogbox.py:197, by the proven bound; the rank-gate outputs are the plan's fast / Jacobi-mask arrays)
hipError_t launch_csne_select_dog(const CsneState& cs, int m, const int* ncols, int* fast, int* ncols_jac, int nfb,
                                  int* tree_list, int* tree_mask, int* tree_count, int* path, const int* sel_mask,
                                  const double* k2, const double* pmin, const double* colinfo, hipStream_t s);
// (dvec == nullptr: J_h = J, no column scaling — dogbox)
hipError_t launch_csne_pass(const CsneState& cs, const double* dvec, int count, hipStream_t s);
// the same sums with the dot products on the FP64 MFMA pipe (TRF; cs.NE = CSNE_MAXE: all eight evaluation slots)
hipError_t launch_csne_pass_mfma(const CsneState& cs, const double* dvec, int count, hipStream_t s);
// dogbox on the tier: the Gauss-Newton step of the free block, lstsq(J_free, -f) (dogbox.py:197), is ONE
// solve — corrected at FACTOR time: scatter the cheap step into a full-length vector (the recording of the pass), the
// pass, then  newton += -(X^T X)^-1 J_free^T (J_free newton + f)  with the free block's factor X
struct DogState;
hipError_t launch_dog_csne_scatter(const CsneState& cs, const DogState& st, int count, hipStream_t s);
hipError_t launch_dog_csne_fix(const CsneState& cs, const DogState& st, int count, hipStream_t s);
hipError_t launch_csne_fix(const CsneState& cs, const TrfState& st, const LmState& lm, const double* Delta,
                           const double* alpha_in, int count, hipStream_t s);
// the problems of fail_list leave the tier: flag 0, path n + 1, tree_mask n + 1, tree_list = fail_list
hipError_t launch_csne_reroute(const CsneState& cs, int nfail, int* tree_list, int* tree_mask, int* path, hipStream_t s);

// ------------------------------------------------------------- Jacobi -----
// One-sided Jacobi on the ROWS of the n x (n+1) array [R | c] (row stride ld):
// U^T [R | c] = [S V^T | U^T c].  In place.
struct JacobiArgs {
  double* X;              // [B][ld*ld]
  long strideX;
  int ld;
  const int* ncols_dev;   // optional per-problem N (= n+1); nullptr -> N
  int N;
  double* s;              // [B][ld]  singular values (unsorted)
  double* uf;             // [B][ld]  (U^T c)_i
  double* srange;         // [B][2]   max, min singular value
  int* sweeps;            // [B]
  int max_sweeps;
  int RB;                 // rows per LDS block (set by launch_jacobi)
};
hipError_t launch_jacobi(const JacobiArgs& a, int B, hipStream_t st);

// ---------------------------------------------------------------- TRF -----
struct TrfState {         // all device pointers, batch-major, vector stride ld
  int B, m, n, ld;        // ld == NPAD
  const double* Rt;       // [B][ld*ld]   R~ = [R c; 0 rho] of [J f]   (problems on the Householder path)
  // Problems on the normal-equations path have NO triangle of J: everything the step needs from J
  // comes from the Gram G = [J f]^T [J f] (g = G[:, n], ||J_j||^2 = G[j][j]) and from the factor X of
  // H = D G D + E^2:  (J_h a).(J_h b) + a.diag_h.b = (X a).(X b).
  const double* Gk;       // [B][ld*ld]   Grams (upper tile blocks), or nullptr
  const int* path;        // [B] 0: normal-equations path, != 0: Householder path; nullptr: Householder for all
  double* X;              // [B][ld*ld]   R~_aug, then (Jacobi) rows s_i v_i^T | uf_i
  double *x, *lb, *ub, *scale;            // [B][ld]
  double *scale_in;                       // [B][ld]  `scale` as it was before the last prep (redo after a gate failure)
  double *g, *v, *d, *g_h, *diag_h;       // [B][ld]
  double *ediag;                          // [B][ld]  sqrt(diag_h): diagonal of the Coleman-Li block E
  double *s, *uf;                         // [B][ld]
  double *srange;                         // [B][2]
  double *g_norm, *theta;                 // [B]
  // CSNE tier (csne_kernels.hip): flag[b] = 1 for a problem the certificate rejected whose step is corrected against J
  // itself; hp = H p_h by the normal-equations identity (the step kernel's model products with p_h); nullptr: off
  int* csne;                              // [B]
  const double* csne_hp;                  // [B][ld]
};
// jac_scaling: 0 keep `scale`; 1 scale = 1/||J col|| (0 -> 1)  (trf.py:216-219);
//              2 scale = min(scale, 1/||J col||)               (trf.py:239-242)
// from_gram: g and the column norms come from st.Gk instead of st.Rt.  sel (optional [B]): only
// problems with sel[b] > 1 are processed.  redo: the problem was prepared from its Gram already in
// this factor call (its gate failed since): start again from scale_in.
// up to five caller vectors [B][n] (8-byte elements; nullptr: skipped) -> the [B][ld] state layout, one launch
struct PackVecs { const void* src[5]; void* dst[5]; int* zero; int nzero; };   // zero[0 .. nzero): counters cleared on the way
hipError_t launch_pack_vecs(const PackVecs& pv, int n, int ld, int B, hipStream_t s);
// (pk: the caller's vectors of this factor call, stride n — copied into the state layout by the prep launch itself
//  instead of a pack_vecs launch in front of the Gram; only with sel == nullptr)
hipError_t launch_trf_prep(const TrfState& st, int jac_scaling, int from_gram, const int* sel,
                           int redo, hipStream_t s, const PackVecs* pk = nullptr);

struct TrfStepOut {       // device pointers
  double* step_h;         // [B][ld]
  double* step;           // [B][ld]
  double* x_new;          // [B][ld]
  long long* hits;        // [B][ld]  hits of x + p (bounds.py:24-48)
  long long* active_new;  // [B][ld]  find_active_constraints(x_new, rtol)
  double* p_h_tr;         // [B][ld]  raw trust-region solution
  double* scal;           // [B][8]: predicted_reduction, step_h_norm, correction,
                          //         alpha_out, to_bound, qp0, qp1, qp2
  int* info;              // [B][4]: n_iter, branch, choice, status
};
// E = 0 problems of the Householder path: the triangle of [R D | c; E | 0] is [R D | c] (written here); mask[b] = n + 1
// for the problems that still need the stacked QR, 0 for the rest (path / sel as the QR launch would see them)
hipError_t launch_trf_aug_trivial(const TrfState& st, const int* path, int* mask, const int* sel, hipStream_t s);
struct LmState;
// Counters the host is waiting for, stored by the FIRST lane of a step kernel before anything else: src[0 .. n) into
// the pinned slot dst (blsq_ctx::publish; dst == nullptr: nothing to publish), a system-scope fence, then the sequence
// number as a system-scope RELEASE store — the host's acquire load of dst[3] orders its reads of dst[0 .. 2] behind
// it, whatever width the payload store was given and however the fabric delivers it.
struct PublishArgs { const int* src; int n; int* dst; int seq; };
__device__ __forceinline__ void publish_ints(const PublishArgs& pub) {
  if (!pub.dst) return;
  pub.dst[0] = pub.n > 0 ? pub.src[0] : 0;
  pub.dst[1] = pub.n > 1 ? pub.src[1] : 0;
  pub.dst[2] = pub.n > 2 ? pub.src[2] : 0;
  __threadfence_system();
  __hip_atomic_store(pub.dst + 3, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_trf_step(const TrfState& st, const LmState* lm, const double* Delta,
                           const double* alpha_in, double active_rtol,
                           const TrfStepOut& out, hipStream_t s, const PublishArgs* pub = nullptr);

// ------------------------------------------------ SVD-free TR sub-problem ----
struct LmState {
  int B, m, n, ld;
  const double* Raug;     // [B][ld*ld]  augmented triangle | c_aug (TrfState.X before Jacobi)
  double* sa;             // [B] sqrt(alpha) of the current evaluation (virtual diagonal block)
  double* Xa;             // [B][ld*ld]  its triangle R_alpha | c_alpha
  int* fast;              // [B] 1: problem uses the SVD-free path
  int* ncols_jac;         // [B] N or 0: launch mask of the Jacobi kernel
  int* ncols_lm;          // [B] N or 0: launch mask of the per-iteration QR
  double* sc;             // [B][16] alpha, lo, hi, phi, dphi, Delta, smax_est, smin_est
  int* st;                // [B][4]  it, phase, n_iter
  double* ph;             // [B][ld] p_h of the SVD-free path
  int* active_count;      // [16] active_count[r] = problems that need evaluation r (zeroed per step call);
                          // the kernels of evaluation r leave when their index is beyond it, so the
                          // host may launch them over an upper bound without knowing the count
  int* jac_count;         // optional [1]: lm_gate adds the number of problems it sends to the Jacobi SVD
  int* active_list;       // [2][B] their indices, compacted (list r & 1 feeds evaluation r)
  const double* colinfo;  // optional [B][2] from the augmented Cholesky (Gram-path problems): min / sum of
                          // the squared column norms of R_aug — a cheap sufficient test for the rank gate
  const double* k2;       // optional [B]: the certificate's PROVEN bound on kappa_2 of the equilibrated augmented system
                          // (0: none).  A Householder-path problem with a finite bound needs no estimate of
                          // s_min either: s_min(R_aug)^2 = lambda_min(H) >= min_j h_jj / K2
  const int* path;        // optional [B]: != 0 -> the problem's Newton systems are factored by QR,
                          // 0 -> by Cholesky of the modified Gram (nullptr: QR for all)
  int round;              // evaluation number of this launch (host-set)
  PublishArgs pub{nullptr, 0, nullptr, 0};   // lm_update_kernel: the count of ITS round, sent to the host on the way in (host-set)
  const double* g_h;      // optional [B][ld]: the gradient of the scaled problem, d * (J^T f) (TrfState::g_h).
                          // ||A^T b|| of trust_region.py:119 for the augmented A = [J D; E], b = [f; 0] IS its
                          // norm; without it lm_start forms R_aug^T c_aug — one more pass over the triangle
  // Newton systems of a HOUSEHOLDER-path problem by Cholesky of the modified Gram: H + alpha I is at least
  // alpha I, so with C_alpha its unit-diagonal equilibration  lambda_min(C_alpha) >= alpha / (h_max + alpha)  and
  // lambda_max(C_alpha) <= lambda_max(C) + 1 <= Lambda + 1  (x^T C_alpha x = y^T C y + alpha |D_alpha x|^2 with
  // |y| <= |x|; Lambda: the certificate's bound on lambda_max(C), at worst the trace n):
  //     kappa_2(C_alpha) <= (Lambda + 1) (h_max + alpha) / alpha
  // — a PROVEN bound that costs nothing.  Where it is below the gate, i.e.
  //     alpha >= 1.01 h_max (Lambda + 1) / (k2_max - Lambda - 1)
  // (1 % margin for the re-prepared scalings), the round factors the system from the kept Gram exactly as for a
  // normal-equations-path problem; only below that alpha the stacked QR of [R_aug; sqrt(alpha) I] runs.
  // hmax == nullptr: always the QR.
  const double* hmax;     // optional [B] (GramCholArgs::hmax of the augmented factorisation)
  const double* lam;      // [B] Lambda (GramCholArgs::lam_out)
  double k2_max;          // the gate
  // CSNE tier: evaluations of a flagged problem are RECORDED (p~, w~ = M~^-1 p~, z~ = M~^-1 w~ and alpha of every
  // evaluation) for the correction stage; nullptr: off
  const int* csne;        // [B]
  int* csne_ne;           // [B] evaluations recorded in this step call (CSNE_MAXE + 1: overflow)
  double* csne_alpha;     // [B][CSNE_MAXE]
  double* csne_vec;       // [B][CSNE_MAXE][3][ld]
  int fused_gram;         // 1: problems with path[b] == 0 belong to lm_rounds_reg_kernel (N <= 80) — lm_start
                          // and the round kernels leave them alone (a problem's arithmetic must not depend
                          // on whether its batch also holds Householder-path problems)
};
// rank gate of the SVD-free paths (lm_kernels.hip; the dogbox finish in chol_kernels.hip)
static constexpr double LM_EPS = 2.220446049250313e-16;
static constexpr double LM_GATE_MARGIN = 1.0e3;
// slots of LmState.sc / LmState.st, phases of the iteration (lm_kernels.hip; the fused rounds in chol_kernels.hip)
enum { LM_IDLE = 0, LM_EVAL = 1, LM_FINAL = 2 };
enum { SC_ALPHA = 0, SC_LO, SC_HI, SC_PHI, SC_DPHI, SC_DELTA, SC_SMAX, SC_SMIN };
enum { ST_IT = 0, ST_PHASE, ST_NITER };
// N <= 80, every problem on the normal-equations path: ALL Newton rounds of the problems lm_start
// listed in one launch (one wave per problem; factor, solves and the update of alpha in registers / LDS)
// the launch takes the place of lm_start and of every round for the normal-equations-path problems
// of the batch (c.rinv / c.dsc = what the augmented Cholesky left)
hipError_t launch_lm_rounds_reg(const GramCholArgs& c, const LmState& lm, const double* Delta,
                                const double* alpha_in, hipStream_t s);
hipError_t launch_lm_gate(const LmState& lm, int enable, hipStream_t s);   // enable: bit 0 Householder-path, bit 1 normal-equations-path problems
hipError_t launch_lm_start(const LmState& lm, const double* Delta, const double* alpha_in,
                           hipStream_t s);
hipError_t launch_lm_update(const LmState& lm, int active, hipStream_t s);   // grid = (upper bound of the) active problems of lm.round

// -------------------------------------------------------------- dogbox ----
struct DogState {
  int B, m, n, ld;
  // normal-equations path (as TrfState): g and the column norms from the Gram, the free-column
  // factor X from its gathered principal sub-matrix; (J_free a).(J_free b) = (X a).(X b)
  const double* Gk;       // [B][ld*ld] Grams or nullptr
  const int* path;        // [B] 0: normal-equations path, != 0: Householder; nullptr: Householder for all
  const int* fast;        // [B] 1: X is still the triangle (Newton step solved without the SVD), 0: Jacobi rows
  double* scale_in;       // [B][ld]
  const double* Rt;       // [B][ld*ld]
  double* S;              // [B][ld*ld]   R~[:, free ++ rhs], compacted columns
  double* X;              // [B][ld*ld]   its triangle, then Jacobi rows
  double *x, *lb, *ub, *scale;            // [B][ld]
  long long* on_bound;    // [B][ld]
  double *g;              // [B][ld]
  int* free_idx;          // [B][ld]  free_idx[q] = original column of free var q
  int* ncols;             // [B]      nfree + 1  (0 when every variable is active)
  double *s, *uf, *srange;
  double *newton, *cauchy;                // [B][ld], compact (free order)
  double *g_norm;                         // [B]
  unsigned char* active;                  // [B][ld]
  // CSNE tier: flag[b] = 1 — the Newton step was corrected against J at factor time, the step kernel takes its
  // products with it from the normal equations  J_free^T J_free newton = -g_free  (nullptr: off)
  int* csne;                              // [B]
  const double* csne_k2;                  // [B] the proven bound of the computed free-block system (rank gate of such a problem)
};
// from_gram / sel / redo: as launch_trf_prep (a Gram-path problem gets neither the compacted
// columns S nor — yet — its Cauchy step: that comes from X in dog_gate_solve)
hipError_t launch_dog_prep(const DogState& st, int jac_scaling, int from_gram, const int* sel,
                           int redo, hipStream_t s, const PackVecs* pk = nullptr);
hipError_t launch_dog_solve(const DogState& st, const int* skip, hipStream_t s);
// path / colinfo (optional): Gram-path flags and the column-norm summary of the free block
// done (optional [B]): problems the Cholesky kernel already finished (GramCholArgs::dog) are skipped
hipError_t launch_dog_gate_solve(const DogState& st, int* fast, int* ncols_jac, int enable,
                                 const int* path, const double* colinfo, int* jac_count,
                                 const int* done, hipStream_t s);

struct DogStepOut {
  double* step;           // [B][ld]  full length
  double* x_new;          // [B][ld]
  long long* on_bound_new;  // [B][ld]
  double* scal;           // [B][4]: predicted_reduction, ||step/scale||_inf, -, -
  int* info;              // [B][4]: tr_hit, fallback, all_active, status
};
hipError_t launch_dog_step(const DogState& st, const double* Delta,
                           const DogStepOut& out, hipStream_t s, const PublishArgs* pub = nullptr);


// ------------------------------------------- batched outer drivers (8f-1) ----
struct OuterState {       // device pointers; method 0 = TRF, 1 = dogbox
  int B, m, n, ld, method;
  double ftol, xtol, gtol;
  int max_nfev;
  // views of the step plan's state (vector stride ld)
  double *x, *lb, *ub, *scale;
  const double* g_norm_fac;   // [B]  from the last factorisation
  const double* v;            // [B][ld]  Coleman-Li v (TRF)
  const int* ncols;           // [B]  dogbox: n_free + 1 (0: every variable active)
  long long* on_bound;        // [B][ld]  dogbox
  const double *o_step, *o_xnew, *o_scal;   // step outputs (TrfStepOut / DogStepOut)
  const int* o_info;
  const long long* o_onb;
  // driver-owned
  double* x0;                 // [B][n]  unshifted initial point (Delta_0)
  double *xc, *xt;            // [B][n]  current / trial point, contiguous, for the callbacks
  double *f, *ft;             // [B][m]  residuals at x / at the trial point
  double *Delta, *alpha, *obj, *gnorm, *actual;                    // [B]
  int *nfev, *njev, *pending, *result, *done, *at_top, *accepted;  // [B]
  int* ncols_fac;             // [B]  n + 1 where a fresh Jacobian must be factored, else 0
  int* counts;                // [2]  active problems, accepted problems (of this tick)
};
hipError_t launch_outer_begin(const OuterState& o, hipStream_t s);
hipError_t launch_outer_top(const OuterState& o, hipStream_t s);
hipError_t launch_outer_trial(const OuterState& o, hipStream_t s);
hipError_t launch_outer_judge(const OuterState& o, hipStream_t s);


// ------------------------------------- finite-difference Jacobians (8f-2) ----
// method: 2 = '2-point', 3 = '3-point'.  X [B][P][n] with P = n (2) or 2n (3); F [B][P][m].
hipError_t launch_fd_points(int B, int n, int method, const double* x, const double* lb,
                            const double* ub, const double* rel_step, double* X, double* h,
                            unsigned char* one_sided, hipStream_t s);
hipError_t launch_fd_assemble(int B, int m, int n, int method, const double* x, const double* h,
                              const unsigned char* one_sided, const double* f0, const double* F,
                              double* J, const int* mask, hipStream_t s);

// ---------------------------------------------------------------- probes ----
// probe_kernels.hip: measured peaks / counter calibration (blsq_debug_probe)
hipError_t launch_mfma_probe(int waves_per_simd, int iters, double* sink, long* n_mfma,
                             hipStream_t s);
hipError_t launch_copy_probe(const void* src, void* dst, size_t bytes, hipStream_t s);

}  // namespace blsq
