// SVD-free evaluation of the trust-region sub-problem for full-rank factors.
//
// solve_lsq_trust_region (bounded_lsq/trust_region.py:56-152) works in the SVD
// basis of the augmented Jacobian, but every quantity it uses is a function of
// A^T A + alpha I only (A = [J D; E], A^T A = R_aug^T R_aug):
//     p(alpha)    = -(A^T A + alpha I)^{-1} A^T b          (:115, :148)
//     phi(alpha)  = ||p|| - Delta                           (:51)
//     phi'(alpha) = -p^T (A^T A + alpha I)^{-1} p / ||p||   (:52: sum suf^2/denom^3)
//     alpha_upper = ||suf|| / Delta = ||A^T b|| / Delta     (:119)
// With  [R_aug; sqrt(alpha) I] = Q_a [R_a; 0]  (the same structured stacked QR as the
// Coleman-Li augmentation; rhs [c_aug; 0] -> c_a) :  p = -R_a^{-1} c_a  and
// p^T (..)^{-1} p = ||R_a^{-T} p||^2 — two triangular solves.  The safeguarded
// Newton iteration itself (brackets, restart rule, |phi| < 0.01 Delta stop, 10
// iterations, stale-phi rescale) is restated unchanged.
//
// The reference takes this branch only when `full_rank` (s_min > eps m s_max,
// :108-112); here a problem uses the SVD-free path only if a CONSERVATIVE gate
// holds (power / inverse-power estimates of s_max, s_min with a 1e3 margin);
// everything else — rank-deficient, wide (m < n), badly conditioned — goes through
// the Jacobi SVD exactly as before.  Batched problems advance in lock-step rounds
// (one stacked QR per Newton iteration), finished ones drop out.
#include <atomic>

#include "blsq_device.h"
#include "blsq_kernels.h"
#include "tri_ops.h"
#include "lm_body.h"

namespace blsq {

// dynamic LDS above 64 KB has to be granted per kernel once
template <class K>
static hipError_t grant_lds(K kernel, size_t bytes, std::atomic<size_t>* granted_dev) {
  int dev = 0;
  (void)hipGetDevice(&dev);                      // the attribute is per device
  std::atomic<size_t>& granted = granted_dev[dev & 63];
  if (bytes <= granted.load(std::memory_order_acquire)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e == hipSuccess) granted.store(bytes, std::memory_order_release);   // (a racing second grant is harmless)
  return e;
}

// (LM_EPS, LM_GATE_MARGIN: blsq_kernels.h)

// (phases and the sc[] / st[] slots: blsq_kernels.h)

// ------------------------------------------------------------------- gate --
__global__ __launch_bounds__(TRI_NT) void lm_gate_kernel(LmState lm, int enable) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (lm.csne && lm.csne[b]) return;                      // (CSNE tier: gated by its own proven bound, csne_select_kernel)
  const int n = lm.n, ld = lm.ld, N = n + 1;
  const double* R = lm.Raug + (long)b * ld * ld;
  double* v = sh;
  double* u = v + ld;
  double* invd = u + ld;
  double* pfbuf = invd + ld;                 // 2 x 16 x ld doubles: DMA staging of the solves
  // enable: bit 0 for Householder-path problems, bit 1 for normal-equations-path problems (which have no
  // n-band where the SVD is preferred: their Newton rounds are one launch)
  const bool on_gram = lm.path && lm.path[b] == 0;
  int ok = (((enable >> (on_gram ? 1 : 0)) & 1) != 0) && (lm.m >= n);
  // Gram-path problems (chol_kernels.hip): R_aug = R'_aug diag(sqrt h_jj) with sigma_min(R'_aug) >=
  // sigma_min(R') >= GRAM_SMIN_PROVEN (the conditioning certificate), so  s_min >= that x min_j sqrt(h_jj)  and  s_max <= sqrt(sum_j h_jj)
  // (the exact Frobenius norm).  When that already clears the threshold below, the iteration is
  // not needed; otherwise (extreme column scaling near the bounds) the estimate runs as always.
  // Householder-path problems the certificate saw (finite proven K2 > gate): the same bound with 1 / sqrt(K2)
  const bool qr_bound = lm.path && lm.path[b] != 0 && lm.k2 && lm.colinfo && lm.k2[b] > 0.0 && is_finite(lm.k2[b]);
  if (ok && lm.path && lm.colinfo && (lm.path[b] == 0 || qr_bound)) {
    const double mn = lm.colinfo[2 * (long)b], sm = lm.colinfo[2 * (long)b + 1];
    const double smin_lb = (qr_bound ? 1.0 / sqrt(lm.k2[b]) : GRAM_SMIN_PROVEN) * mn, smax_ub = sqrt(sm);
    if (is_finite(sm) && sm > 0.0 && smin_lb > LM_GATE_MARGIN * LM_EPS * lm.m * smax_ub) {
      if (tid == 0) {
        lm.fast[b] = 1;
        lm.ncols_jac[b] = 0;
        lm.sc[(long)b * 16 + SC_SMAX] = smax_ub;
        lm.sc[(long)b * 16 + SC_SMIN] = smin_lb;
        lm.st[(long)b * 4 + ST_PHASE] = LM_IDLE;
      }
      return;
    }
  }
  // diagonal: finite and non-zero
  int bad = 0;
  for (int i = tid; i < n; i += TRI_NT) {
    const double dgi = R[(long)i * ld + i];
    if (!(is_finite(dgi)) || dgi == 0.0) bad = 1;
  }
  if (block_or(bad, red)) ok = 0;
  double smax = 0.0, smin = 0.0;
  if (ok) {                                             // uniform
    tri_invdiag(R, n, ld, invd);
    const double s0 = 1.0 / sqrt((double)n);
    // s_max <= ||R||_F: an UPPER bound makes the gate more conservative, and costs one pass
    {
      // one wave per row, lanes across the columns (coalesced), four rows per pass
      double fro = 0.0;
      const int lane = tid & 63, wv = tid >> 6;
      for (int i0 = wv; i0 < n; i0 += TRI_NW * 4) {
        for (int jj = 0; i0 + lane + jj < n; jj += WAVE) {
          double rv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + r * TRI_NW;
            const int ic = (i < n) ? i : n - 1;
            const int j = ic + lane + jj;
            rv[r] = R[(long)ic * ld + ((j < n) ? j : n - 1)];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + r * TRI_NW;
            if (i < n && i + lane + jj < n) fro = fma(rv[r], rv[r], fro);
          }
        }
      }
      smax = sqrt(block_sum(fro, red));
    }
    // s_min: inverse power iteration on (R^T R)^{-1} (an upper bound on s_min)
    for (int i = tid; i < n; i += TRI_NT) v[i] = (i % 3 == 0) ? s0 : -0.5 * s0;
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
      tri_solve_upper_t_pf(R, n, ld, invd, v, pfbuf);
      tri_solve_upper_pf(R, n, ld, invd, v, pfbuf);
      const double nv = sqrt(tri_dot(v, v, n, red));
      smin = (nv > 0.0 && is_finite(nv)) ? 1.0 / sqrt(nv) : 0.0;   // ||(R^T R)^{-1} v|| ~ 1/s_min^2
      const double inv = (nv > 0.0 && is_finite(nv)) ? 1.0 / nv : 0.0;
      for (int i = tid; i < n; i += TRI_NT) v[i] *= inv;
      __syncthreads();
    }
    if (!(smin > LM_GATE_MARGIN * LM_EPS * lm.m * smax) || !is_finite(smax) || smax == 0.0) ok = 0;
  }
  if (tid == 0) {
    lm.fast[b] = ok;
    if (!ok && lm.jac_count) atomicAdd(lm.jac_count, 1);
    lm.ncols_jac[b] = ok ? 0 : N;
    lm.sc[(long)b * 16 + SC_SMAX] = smax;
    lm.sc[(long)b * 16 + SC_SMIN] = smin;
    lm.st[(long)b * 4 + ST_PHASE] = LM_IDLE;
  }
}

hipError_t launch_lm_gate(const LmState& lm, int enable, hipStream_t s) {
  const size_t lds = sizeof(double) * (3 + 32) * (size_t)lm.ld;
  { static std::atomic<size_t> granted[64]; hipError_t ge = grant_lds(lm_gate_kernel, lds, granted); if (ge != hipSuccess) return ge; }
  hipLaunchKernelGGL(lm_gate_kernel, dim3(lm.B), dim3(TRI_NT), lds, s, lm, enable);
  return hipGetLastError();
}

// ------------------------------------------------------------------ start --
// Gauss-Newton step (alpha = 0) on R_aug, acceptance test, bracket initialisation (lm_body.h).
__global__ __launch_bounds__(TRI_NT) void lm_start_kernel(LmState lm, const double* Delta_in,
                                                          const double* alpha_in) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  const int b = blockIdx.x;
  if (lm.fused_gram && lm.path && lm.path[b] == 0) return;   // (lm_rounds_reg_kernel owns this problem)
  lm_start_body<TRI_NT>(lm, b, Delta_in, alpha_in, sh, red, true);
}

hipError_t launch_lm_start(const LmState& lm, const double* Delta, const double* alpha_in,
                           hipStream_t s) {
  const size_t lds = sizeof(double) * (3 + 32) * (size_t)lm.ld;
  { static std::atomic<size_t> granted[64]; hipError_t ge = grant_lds(lm_start_kernel, lds, granted); if (ge != hipSuccess) return ge; }
  hipLaunchKernelGGL(lm_start_kernel, dim3(lm.B), dim3(TRI_NT), lds, s, lm, Delta, alpha_in);
  return hipGetLastError();
}

// ----------------------------------------------------------------- update --
// one evaluation of phi / phi' at the current alpha + the Newton update (:132-150; lm_body.h)
__global__ __launch_bounds__(TRI_NT) void lm_update_kernel(LmState lm) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  // launched over the compacted list of evaluation lm.round; writes the next one
  if (blockIdx.x == 0 && threadIdx.x == 0) publish_ints(lm.pub);   // (the count of this round: final since the last round ended)
  if ((int)blockIdx.x >= lm.active_count[lm.round]) return;   // (launched over an upper bound)
  const int b = lm.active_list[(long)(lm.round & 1) * lm.B + blockIdx.x];
  if (!lm.fast[b]) return;
  lm_update_body<TRI_NT>(lm, b, sh, red, lm.round + 1);
}

hipError_t launch_lm_update(const LmState& lm, int active, hipStream_t s) {
  const size_t lds = sizeof(double) * (3 + 32) * (size_t)lm.ld;
  { static std::atomic<size_t> granted[64]; hipError_t ge = grant_lds(lm_update_kernel, lds, granted); if (ge != hipSuccess) return ge; }
  hipLaunchKernelGGL(lm_update_kernel, dim3(active), dim3(TRI_NT), lds, s, lm);
  return hipGetLastError();
}


// ---------------------------------------------------------------- dogbox --
// lstsq(J_free, -f)[0] (dogbox.py:197) without an SVD when the free block is clearly of
// full column rank: newton = -R_f^{-1} c_f.  gelsd drops singular values below
// rcond * s_max with rcond = eps * max(m, n_free); the gate requires the inverse-power
// upper bound on s_min to clear that threshold by the same 1e3 margin, otherwise the
// Jacobi SVD computes the truncated min-norm solution as before.
__global__ __launch_bounds__(TRI_NT) void dog_gate_solve_kernel(DogState st, int* fast,
                                                                int* ncols_jac, int enable,
                                                                const int* path,
                                                                const double* colinfo, int* jac_count,
                                                                const int* done) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (done && done[b]) return;                            // (finished inside the Cholesky kernel)
  const int ld = st.ld;
  const int N = st.ncols[b];
  const int nf = N - 1;
  if (N <= 0) {
    if (tid == 0) { fast[b] = 0; ncols_jac[b] = 0; }
    return;
  }
  const double* R = st.X + (long)b * ld * ld;             // triangle of [R[:, free] | c]
  double* v = sh;
  double* u = v + ld;
  double* invd = u + ld;
  double* pfbuf = invd + ld;                 // 2 x 16 x ld doubles: DMA staging of the solves
  if (path && path[b] == 0) {
    // normal-equations path: the Cauchy step (dogbox.py:198-199), -(g.g)/(Jg.Jg) g_free with
    // |J_free g_free|^2 = |X g_free|^2 — the Householder path computed it in dog_prep from R
    const int* fidx = st.free_idx + (long)b * ld;
    for (int q = tid; q < nf; q += TRI_NT) v[q] = st.g[(long)b * ld + fidx[q]];
    __syncthreads();
    tri_mv(R, nf, ld, v, u);
    const double gg = tri_dot(v, v, nf, red);
    const double uu = tri_dot(u, u, nf, red);
    const double fac = -gg / uu;
    for (int q = tid; q < nf; q += TRI_NT) st.cauchy[(long)b * ld + q] = fac * v[q];
    __syncthreads();
  }
  int ok = (enable != 0) && (st.m >= nf);
  int bad = 0;
  for (int i = tid; i < nf; i += TRI_NT) {
    const double dgi = R[(long)i * ld + i];
    if (!(is_finite(dgi)) || dgi == 0.0) bad = 1;
  }
  if (block_or(bad, red)) ok = 0;
  if (ok) {
    tri_invdiag(R, nf, ld, invd);
    const double s0 = 1.0 / sqrt((double)nf);
    const int mx = (st.m > nf) ? st.m : nf;
    // Gram-path problems: s_min >= GRAM_SMIN_PROVEN min_j ||J_free[:, j]||, s_max <= ||J_free||_F (column norms
    // from the Cholesky of the gathered Gram, as in lm_gate_kernel); the estimates below only run
    // when that bound does not already clear the threshold
    bool sure = false;
    if (path && colinfo && path[b] == 0) {
      const double mn = colinfo[2 * (long)b], sm = colinfo[2 * (long)b + 1];
      // (a problem on the CSNE tier: the proven bound of ITS computed system instead of the gate's)
      const double sminp = (st.csne && st.csne[b] && st.csne_k2 && st.csne_k2[b] > 0.0) ? 1.0 / sqrt(st.csne_k2[b])
                                                                                       : GRAM_SMIN_PROVEN;
      sure = is_finite(sm) && sm > 0.0 && (sminp * mn > LM_GATE_MARGIN * LM_EPS * mx * sqrt(sm));
    }
    if (!sure) {
    for (int i = tid; i < nf; i += TRI_NT) v[i] = (i & 1) ? -s0 : s0;
    __syncthreads();
    double smax = 0.0, smin = 0.0;
    for (int it = 0; it < 2; ++it) {
      tri_mv(R, nf, ld, v, u);
      tri_mtv(R, nf, ld, u, v);
      const double nv = sqrt(tri_dot(v, v, nf, red));
      smax = sqrt(nv);
      const double inv = (nv > 0.0) ? 1.0 / nv : 0.0;
      for (int i = tid; i < nf; i += TRI_NT) v[i] *= inv;
      __syncthreads();
    }
    for (int i = tid; i < nf; i += TRI_NT) v[i] = (i % 3 == 0) ? s0 : -0.5 * s0;
    __syncthreads();
    for (int it = 0; it < 3; ++it) {
      tri_solve_upper_t_pf(R, nf, ld, invd, v, pfbuf);
      tri_solve_upper_pf(R, nf, ld, invd, v, pfbuf);
      const double nv = sqrt(tri_dot(v, v, nf, red));
      smin = (nv > 0.0 && is_finite(nv)) ? 1.0 / sqrt(nv) : 0.0;
      const double inv = (nv > 0.0 && is_finite(nv)) ? 1.0 / nv : 0.0;
      for (int i = tid; i < nf; i += TRI_NT) v[i] *= inv;
      __syncthreads();
    }
    if (!(smin > LM_GATE_MARGIN * LM_EPS * mx * smax) || !is_finite(smax) || smax == 0.0) ok = 0;
    }
  }
  if (ok) {
    for (int i = tid; i < nf; i += TRI_NT) v[i] = R[(long)i * ld + nf];      // c_f
    __syncthreads();
    tri_solve_upper_pf(R, nf, ld, invd, v, pfbuf);
    for (int i = tid; i < nf; i += TRI_NT) st.newton[(long)b * ld + i] = -v[i];
  }
  if (tid == 0) {
    fast[b] = ok; ncols_jac[b] = ok ? 0 : N;
    if (!ok && jac_count) atomicAdd(jac_count, 1);
  }
}

hipError_t launch_dog_gate_solve(const DogState& st, int* fast, int* ncols_jac, int enable,
                                 const int* path, const double* colinfo, int* jac_count,
                                 const int* done, hipStream_t s) {
  const size_t lds = sizeof(double) * (3 + 32) * (size_t)st.ld;
  { static std::atomic<size_t> granted[64]; hipError_t ge = grant_lds(dog_gate_solve_kernel, lds, granted); if (ge != hipSuccess) return ge; }
  hipLaunchKernelGGL(dog_gate_solve_kernel, dim3(st.B), dim3(TRI_NT), lds, s, st, fast,
                     ncols_jac, enable, path, colinfo, jac_count, done);
  return hipGetLastError();
}

}  // namespace blsq
