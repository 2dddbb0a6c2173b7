// Device-side bodies of the safeguarded Newton iteration of the SVD-free trust-region path
// (solve_lsq_trust_region, bounded_lsq/trust_region.py:111-150), shared by the lock-step round kernels of
// lm_kernels.hip and the fused per-problem kernel of chol_kernels.hip.  Templates on NT, the thread count of
// the calling workgroup; one workgroup per problem; `sh` = (3 + 32) * ld doubles of LDS, `red` = 32 doubles.
#pragma once
#include "blsq_device.h"
#include "blsq_kernels.h"
#include "tri_ops.h"

namespace blsq {

// which kernel factors the Newton system of problem b at this alpha: the stacked QR (n + 1) or the
// Cholesky of the modified Gram (0) — LmState::hmax
__device__ __forceinline__ int lm_qr_cols(const LmState& lm, int b, double alpha, int n) {
  if (lm.path && lm.path[b] == 0) return 0;                       // normal-equations path: always the Gram
  if (lm.hmax) {
    const double hm = lm.hmax[b], L1 = lm.lam[b] + 1.0;
    if (is_finite(hm) && hm > 0.0 && L1 >= 2.0 && lm.k2_max > L1 && alpha >= 1.01 * hm * L1 / (lm.k2_max - L1)) return 0;
  }
  return n + 1;
}

// restart rule of trust_region.py:128,134
__device__ __forceinline__ double lm_restart(double lo, double hi) {
  const double gm = sqrt(lo * hi);
  return (0.001 * hi > gm) ? 0.001 * hi : gm;
}

// CSNE tier (csne_kernels.hip): evaluation k of a flagged problem is RECORDED for the correction stage — p~ (in `p`),
// w~ = M~^-1 p~ and z~ = M~^-1 w~ with M~ = R^T R the factor at hand, and alpha.  q: LDS scratch (destroyed).
// last: the iteration ends with this evaluation — only p~ is needed (the correction solves for the final step itself).
template <int NT>
__device__ __forceinline__ void lm_csne_record(const LmState& lm, int b, int k, double alpha, const double* R,
                                               const double* invd, const double* p, double* q, double* pfbuf,
                                               bool last) {
  const int tid = threadIdx.x, n = lm.n, ld = lm.ld;
  if (k >= CSNE_MAXE) {                                   // (uniform) deeper than the tier records: it will decline
    if (tid == 0) lm.csne_ne[b] = CSNE_MAXE + 1;
    return;
  }
  double* rec = lm.csne_vec + ((long)b * CSNE_MAXE + k) * 3 * ld;
  for (int i = tid; i < n; i += NT) { const double v = p[i]; q[i] = v; rec[i] = v; }
  if (last) {                                             // (uniform)
    if (tid == 0) { lm.csne_alpha[(long)b * CSNE_MAXE + k] = alpha; lm.csne_ne[b] = k + 1; }
    __syncthreads();
    return;
  }
  __syncthreads();
  tri_solve_upper_t_pf<NT>(R, n, ld, invd, q, pfbuf);
  tri_solve_upper_pf<NT>(R, n, ld, invd, q, pfbuf);
  for (int i = tid; i < n; i += NT) rec[ld + i] = q[i];
  tri_solve_upper_t_pf<NT>(R, n, ld, invd, q, pfbuf);
  tri_solve_upper_pf<NT>(R, n, ld, invd, q, pfbuf);
  for (int i = tid; i < n; i += NT) rec[2 * ld + i] = q[i];
  if (tid == 0) { lm.csne_alpha[(long)b * CSNE_MAXE + k] = alpha; lm.csne_ne[b] = k + 1; }
  __syncthreads();
}

// Gauss-Newton step (alpha = 0) on R_aug, acceptance test, bracket initialisation (trust_region.py:111-130).
// enqueue: append an iterating problem to evaluation list 0 (the lock-step loop); the fused kernel keeps it.
// -> the problem's phase afterwards (LM_IDLE: nothing to iterate)
template <int NT>
__device__ __forceinline__ int lm_start_body(const LmState& lm, int b, const double* Delta_in,
                                             const double* alpha_in, double* sh, double* red, bool enqueue) {
  const int tid = threadIdx.x;
  if (!lm.fast[b]) {
    if (tid == 0) lm.ncols_lm[b] = 0;
    return LM_IDLE;
  }
  const int n = lm.n, ld = lm.ld;
  const double* R = lm.Raug + (long)b * ld * ld;
  double* p = sh;
  double* q = p + ld;
  double* invd = q + ld;
  double* pfbuf = invd + ld;                 // 2 x 16 x ld doubles: DMA staging of the solves
  double* sc = lm.sc + (long)b * 16;
  int* st = lm.st + (long)b * 4;
  const double Delta = Delta_in[b];
  tri_invdiag<NT>(R, n, ld, invd);
  for (int i = tid; i < n; i += NT) p[i] = R[(long)i * ld + n];     // c_aug
  __syncthreads();
  // alpha_upper = ||A^T b|| / Delta
  double gnorm;
  if (lm.g_h) {                               // A^T b = D J^T f: known since the prep kernel
    for (int i = tid; i < n; i += NT) q[i] = lm.g_h[(long)b * ld + i];
    __syncthreads();
    gnorm = sqrt(tri_dot<NT>(q, q, n, red));
  } else {
    tri_mtv<NT>(R, n, ld, p, q);
    gnorm = sqrt(tri_dot<NT>(q, q, n, red));
  }
  tri_solve_upper_pf<NT>(R, n, ld, invd, p, pfbuf);                           // R^{-1} c
  const double pn = sqrt(tri_dot<NT>(p, p, n, red));
  for (int i = tid; i < n; i += NT) {
    p[i] = -p[i];
    lm.ph[(long)b * ld + i] = p[i];
  }
  __syncthreads();
  if (lm.csne && lm.csne[b]) lm_csne_record<NT>(lm, b, 0, 0.0, R, invd, p, q, pfbuf, pn <= Delta);   // (uniform)
  if (pn <= Delta) {                                                      // trust_region.py:116-117
    if (tid == 0) {
      sc[SC_ALPHA] = 0.0; st[ST_NITER] = 0; st[ST_PHASE] = LM_IDLE; sc[SC_DELTA] = Delta;
      lm.ncols_lm[b] = 0;
    }
    return LM_IDLE;
  }
  // phi(0), phi'(0) -> alpha_lower (trust_region.py:121-123)
  for (int i = tid; i < n; i += NT) q[i] = p[i];
  __syncthreads();
  tri_solve_upper_t_pf<NT>(R, n, ld, invd, q, pfbuf);
  const double qq = tri_dot<NT>(q, q, n, red);
  const double phi = pn - Delta;
  const double dphi = -qq / pn;
  double hi = gnorm / Delta;
  double lo = -phi / dphi;
  double alpha = alpha_in[b];                                             // :127-130 (full rank)
  if (alpha < lo || alpha > hi) alpha = lm_restart(lo, hi);               // :133-134, iteration 0
  if (tid == 0) {
    sc[SC_ALPHA] = alpha; sc[SC_LO] = lo; sc[SC_HI] = hi; sc[SC_PHI] = phi; sc[SC_DPHI] = dphi;
    sc[SC_DELTA] = Delta;
    st[ST_IT] = 0; st[ST_PHASE] = LM_EVAL; st[ST_NITER] = 0;
    lm.sa[b] = sqrt(alpha); lm.ncols_lm[b] = lm_qr_cols(lm, b, alpha, n);                   // QR launch mask
    if (enqueue) lm.active_list[atomicAdd(lm.active_count, 1)] = b;          // list 0 feeds evaluation 0
  }
  return LM_EVAL;
}

// one evaluation of phi / phi' at the current alpha + the Newton update (trust_region.py:132-150), from the
// factor R_alpha | c_alpha in lm.Xa.  next_list: >= 0: append a problem that goes on to evaluation list
// `next_list` (lock-step loop); < 0: no list (fused kernel).  -> the phase afterwards
template <int NT>
__device__ __forceinline__ int lm_update_body(const LmState& lm, int b, double* sh, double* red, int next_list) {
  const int tid = threadIdx.x;
  int* st = lm.st + (long)b * 4;
  const int phase = st[ST_PHASE];
  if (phase == LM_IDLE) return LM_IDLE;
  const int n = lm.n, ld = lm.ld;
  const double* R = lm.Xa + (long)b * ld * ld;            // R_alpha | c_alpha
  double* p = sh;
  double* q = p + ld;
  double* invd = q + ld;
  double* pfbuf = invd + ld;                 // 2 x 16 x ld doubles: DMA staging of the solves
  double* sc = lm.sc + (long)b * 16;
  const double Delta = sc[SC_DELTA];
  tri_invdiag<NT>(R, n, ld, invd);
  for (int i = tid; i < n; i += NT) p[i] = R[(long)i * ld + n];
  __syncthreads();
  tri_solve_upper_pf<NT>(R, n, ld, invd, p, pfbuf);
  const double pn = sqrt(tri_dot<NT>(p, p, n, red));
  for (int i = tid; i < n; i += NT) p[i] = -p[i];
  __syncthreads();
  double alpha = sc[SC_ALPHA], lo = sc[SC_LO], hi = sc[SC_HI];
  double phi = sc[SC_PHI], dphi = sc[SC_DPHI];
  int it = st[ST_IT];
  int next_phase = LM_IDLE;
  int n_iter = st[ST_NITER];
  bool finished = false;
  if (phase == LM_FINAL) {
    // loop exhausted (:132 ran 10 times without break): p at the UPDATED alpha, rescale
    // test on the STALE phi (:149)
    finished = true;
    if (lm.csne && lm.csne[b] && tid == 0) lm.csne_ne[b] = CSNE_MAXE + 1;   // (ten rounds: beyond what the tier records)
  } else {
    for (int i = tid; i < n; i += NT) q[i] = p[i];
    __syncthreads();
    tri_solve_upper_t_pf<NT>(R, n, ld, invd, q, pfbuf);
    const double qq = tri_dot<NT>(q, q, n, red);
    phi = pn - Delta;
    dphi = -qq / pn;
    if (lm.csne && lm.csne[b]) lm_csne_record<NT>(lm, b, it + 1, alpha, R, invd, p, q, pfbuf, fabs(phi) < 0.01 * Delta);   // (uniform)
    if (fabs(phi) < 0.01 * Delta) {                       // :138-139
      finished = true;
      n_iter = it + 1;
    } else {
      if (phi < 0.0) hi = alpha;                          // :141-142
      const double ratio = phi / dphi;
      const double cand = alpha - ratio;
      lo = (cand > lo) ? cand : lo;                       // :145
      alpha -= (phi + Delta) * ratio / Delta;             // :146
      ++it;
      if (it >= 10) {                                     // max_iter reached: final p at new alpha
        n_iter = 10;
        next_phase = LM_FINAL;
      } else {
        if (alpha < lo || alpha > hi) alpha = lm_restart(lo, hi);   // :133-134 of the next pass
        next_phase = LM_EVAL;
      }
    }
  }
  if (finished) {
    const double f = (phi > 0.0) ? Delta / pn : 1.0;      // :149-150
    for (int i = tid; i < n; i += NT) lm.ph[(long)b * ld + i] = p[i] * f;
    next_phase = LM_IDLE;
  }
  __syncthreads();
  if (tid == 0) {
    sc[SC_ALPHA] = alpha; sc[SC_LO] = lo; sc[SC_HI] = hi; sc[SC_PHI] = phi; sc[SC_DPHI] = dphi;
    st[ST_IT] = it; st[ST_PHASE] = next_phase; st[ST_NITER] = n_iter;
    lm.sa[b] = sqrt(alpha);
    lm.ncols_lm[b] = (next_phase != LM_IDLE) ? lm_qr_cols(lm, b, alpha, n) : 0;
    if (next_phase != LM_IDLE && next_list >= 0)
      lm.active_list[(long)(next_list & 1) * lm.B + atomicAdd(lm.active_count + next_list, 1)] = b;
  }
  return next_phase;
}

}  // namespace blsq
