// n-space kernels of the Trust Region Reflective step (one workgroup / problem).
//
// Everything the reference does between the SVD and the next fun() call
// (bounded_lsq/trf.py:244-308) expressed on the triangle R~ = [R c] of [J f]:
//   g = J^T f = R^T c                                  (trf.py:244)
//   ||J[:,j]|| = ||R[:,j]||  for 'jac' scaling          (trf.py:216-219,239-242)
//   J_h = J D  =>  (J_h a).(J_h b) = (R D a).(R D b)    (trf.py:69-73,100)
//   svd([J D; E]) <-> svd of the triangle of [R D; E]  (trf.py:264-274)
// The arithmetic of the bound geometry keeps the reference's elementwise
// operation order (this file is compiled with -ffp-contract=off): masks come
// from exact == on computed minima (bounds.py:47-48).
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int NS_NT = 256;
static constexpr int NS_NW = NS_NT / WAVE;
static constexpr double EPS = 2.220446049250313e-16;

// ------------------------------------------------------------------ prep --
__global__ __launch_bounds__(NS_NT) void trf_prep_kernel(TrfState st, int jac_scaling, int from_gram,
                                                         const int* sel, int redo, PackVecs pk) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b == 0 && pk.zero && tid < pk.nzero) pk.zero[tid] = 0;
  if (sel && sel[b] <= 1) return;
  if (st.csne && tid == 0) st.csne[b] = 0;               // (a problem prepared afresh is on the CSNE tier only if selected again)
  const int n = st.n, ld = st.ld;
  const double* Rt = st.Rt + (long)b * ld * ld;
  const double* Gk = from_gram ? st.Gk + (long)b * ld * ld : nullptr;
  const long vo = (long)b * ld;
  double gmax = 0.0;
  for (int j = tid; j < n; j += NS_NT) {
    double gj = 0.0, nn = 0.0;
    if (from_gram) {
      // g_j = (J^T f)_j and ||J_j||^2 straight from the Gram (trf.py:244; :216-219, :239-242)
      gj = Gk[(long)j * ld + n];
      nn = Gk[(long)j * ld + j];
    } else {
    // g_j = sum_{i<=j} R[i][j] c_i ; column norm of R for 'jac' scaling
    for (int i0 = 0; i0 <= j; i0 += 8) {      // 8 rows per pass, loads unconditional (clamped)
      double rv[8], cv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = (i0 + u <= j) ? i0 + u : j;
        rv[u] = Rt[(long)i * ld + j];
        cv[u] = Rt[(long)i * ld + n];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i0 + u <= j) {
          gj = fma(rv[u], cv[u], gj);
          nn = fma(rv[u], rv[u], nn);
        }
      }
    }
    }
    if (pk.src[0]) {                                      // the caller's vectors (stride n) -> state layout
      const long so = (long)b * n + j;
      st.x[vo + j] = static_cast<const double*>(pk.src[0])[so];
      st.lb[vo + j] = static_cast<const double*>(pk.src[1])[so];
      st.ub[vo + j] = static_cast<const double*>(pk.src[2])[so];
      st.scale[vo + j] = static_cast<const double*>(pk.src[3])[so];
    }
    double sc = redo ? st.scale_in[vo + j] : st.scale[vo + j];
    if (!redo) st.scale_in[vo + j] = sc;
    if (jac_scaling == 1) {
      double jn = sqrt(nn);
      if (jn == 0.0) jn = 1.0;
      sc = 1.0 / jn;
    } else if (jac_scaling == 2) {
      const double inv = 1.0 / sqrt(nn);    // 1/0 = inf keeps the old scale
      sc = (inv < sc) ? inv : sc;
    }
    st.scale[vo + j] = sc;
    // Coleman-Li scaling vector (bounds.py:106-149)
    const double xj = st.x[vo + j], lj = st.lb[vo + j], uj = st.ub[vo + j];
    double v = 1.0, jv = 0.0;
    if (gj < 0.0 && is_finite(uj)) { v = uj - xj; jv = -1.0; }
    if (gj > 0.0 && is_finite(lj)) { v = xj - lj; jv = 1.0; }
    const double d = sqrt(v) * sc;                       // trf.py:248
    st.g[vo + j] = gj;
    st.v[vo + j] = v;
    st.d[vo + j] = d;
    st.g_h[vo + j] = d * gj;                             // trf.py:249
    const double dhj = gj * jv * (sc * sc);              // trf.py:250
    st.diag_h[vo + j] = dhj;
    st.ediag[vo + j] = sqrt(dhj);                        // E = diag(sqrt(diag_h))  (trf.py:264-270)
    gmax = nanmax2(gmax, fabs(gj * v));                  // trf.py:252
  }
  gmax = block_max(gmax, red);
  if (tid == 0) {
    st.g_norm[b] = gmax;
    const double th = 1.0 - gmax;
    st.theta[b] = (th > 0.995) ? th : 0.995;             // trf.py:277
  }
  // The stacked system [R D | c ; E | 0] (trf.py:264-270) is never materialised: the QR reads
  // R in place, scales column j by d_j while loading and takes E from `ediag` (qr_panel.hip).
}

hipError_t launch_trf_prep(const TrfState& st, int jac_scaling, int from_gram, const int* sel,
                           int redo, hipStream_t s, const PackVecs* pk) {
  const PackVecs none{{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, 0};
  hipLaunchKernelGGL(trf_prep_kernel, dim3(st.B), dim3(NS_NT), 0, s, st, jac_scaling, from_gram, sel,
                     redo, (pk && !sel) ? *pk : none);
  return hipGetLastError();
}

// ------------------------------------------------- trivial augmentation --
// [R D | c ; E | 0] with E = 0 (no variable has a finite bound in its descent direction: diag_h = 0,
// trf.py:250 — every unbounded problem) is ALREADY triangular: its triangle is [R D | c] itself, the
// stacked QR (one workgroup per problem through 17 panels: the latency of one problem, whatever the
// batch) has nothing to eliminate.  This kernel writes that triangle for such Householder-path problems
// and builds the launch mask of the QR for the others: mask[b] = n + 1 (QR) or 0 (done here / Gram path).
__global__ __launch_bounds__(NS_NT) void trf_aug_trivial_kernel(TrfState st, const int* path, int* mask,
                                                                const int* sel) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = st.n, ld = st.ld;
  const long vo = (long)b * ld;
  if (path && path[b] == 0) {                            // normal-equations path: no triangle at all
    if (tid == 0) mask[b] = 0;
    return;
  }
  if (sel && sel[b] <= 1) {                              // (masked factor call: this problem keeps its state)
    if (tid == 0) mask[b] = 0;
    return;
  }
  int any = 0;
  for (int j = tid; j < n; j += NS_NT) any |= !(st.ediag[vo + j] == 0.0);   // (NaN counts as non-zero)
  any = block_or(any, red);
  if (any) {
    if (tid == 0) mask[b] = n + 1;
    return;
  }
  if (tid == 0) mask[b] = 0;
  const double* R = st.Rt + (long)b * ld * ld;
  double* X = st.X + (long)b * ld * ld;
  for (int i = tid >> 6; i < ld; i += NS_NW) {           // one wave per row, coalesced
    for (int j = tid & 63; j < ld; j += WAVE) {
      double v = 0.0;
      if (i < n && j >= i) {
        if (j < n) v = R[(long)i * ld + j] * st.d[vo + j];
        else if (j == n) v = R[(long)i * ld + n];
      }
      X[(long)i * ld + j] = v;
    }
  }
}

hipError_t launch_trf_aug_trivial(const TrfState& st, const int* path, int* mask, const int* sel,
                                  hipStream_t s) {
  hipLaunchKernelGGL(trf_aug_trivial_kernel, dim3(st.B), dim3(NS_NT), 0, s, st, path, mask, sel);
  return hipGetLastError();
}

// ------------------------------------------------------------------ step --
struct StepCtx {
  int n, ld;
  const double *x, *lb, *ub, *d;
  double* red;
};

// bounds.py:24-48.  dir[] in LDS; returns min step, writes the per-element
// steps to `steps` (LDS) so the caller can form the hit mask.
__device__ double step_to_bound_dev(const StepCtx& c, const double* xs,
                                    const double* dir, double* steps) {
  double tmin = __builtin_inf();
  for (int j = threadIdx.x; j < c.n; j += NS_NT) {
    const double dj = dir[j];
    double t = __builtin_inf();
    if (dj != 0.0) {
      const double lo = (c.lb[j] - xs[j]) / dj;
      const double hi = (c.ub[j] - xs[j]) / dj;
      t = nanmax2(lo, hi);
    }
    steps[j] = t;
    tmin = nanmin2(tmin, t);
  }
  return block_min(tmin, c.red);
}

// u = R_h s = (R D) s  (R upper triangular, row-major, stride ld; D = diag(dvec)): one wave per row.
__device__ void tri_matvec(const double* R, const double* dvec, int n, int ld, const double* svec,
                           double* u) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // Eight rows per wave pass with unconditional (clamped) loads: R does not fit any cache for a
  // whole batch, so the loads of a pass must be in flight together — one row at a time every
  // iteration would pay a full memory round trip.  Per row the accumulation order is unchanged.
  constexpr int RB = 8;
  for (int i0 = w; i0 < n; i0 += NS_NW * RB) {
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
    for (int jj = 0; i0 + lane + jj < n; jj += WAVE) {       // trip count of the longest row (i0)
      double rv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * NS_NW;
        const int ic = (i < n) ? i : n - 1;
        const int j = ic + lane + jj;
        rv[r] = R[(long)ic * ld + ((j < n) ? j : n - 1)];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * NS_NW;
        const int j = i + lane + jj;
        // (R[i][j] * d[j]) * s[j]: the same roundings as a materialised R_h = R D
        if (i < n && j < n) acc[r] = fma(dvec ? rv[r] * dvec[j] : rv[r], svec[j], acc[r]);
      }
    }
    // (the eight row totals by one transposed butterfly: the tree of wave_sum for each of them, see tri_matvec3)
    static_assert(RB == 8, "one sixteen-value reduction");
    double v[16];
#pragma unroll
    for (int r = 0; r < RB; ++r) { v[r] = acc[r]; v[8 + r] = 0.0; }
    wave_sum16(v);
    const int idx = wave_sum16_index(lane), ri = i0 + (idx & 7) * NS_NW;
    if (lane < 16 && idx < 8 && ri < n) u[ri] = v[0];
  }
  __syncthreads();
}

// u = M s for a DENSE n x n block (row-major, stride ld): the Jacobi rows s_i v_i^T of a problem whose
// factor went through the SVD.  One wave per row.
__device__ void full_matvec(const double* M, int n, int ld, const double* svec, double* u) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int RB = 4;
  for (int i0 = w; i0 < n; i0 += NS_NW * RB) {
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
    for (int jj = lane; jj < n; jj += WAVE) {
      double rv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * NS_NW;
        rv[r] = M[(long)((i < n) ? i : n - 1) * ld + jj];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) acc[r] = fma(rv[r], svec[jj], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * NS_NW;
      const double t = wave_sum(acc[r]);
      if (lane == 0 && i < n) u[i] = t;
    }
  }
  __syncthreads();
}

// Three products with ONE pass over the matrix (the reflective branch needs J_h p_h, J_h r_h and
// J_h (-g_h); the matrix — half a megabyte per problem at n = 256 — does not stay in any cache
// between separate passes):  u1 = M s1,  u2 = M s2 (s2 == nullptr: skipped),  u3 = M (-g3).
// Per row and product exactly the operations of tri_matvec / full_matvec, in the same order.
__device__ void tri_matvec3(const double* R, const double* dvec, int n, int ld, const double* s1,
                            double* u1, const double* s2, double* u2, const double* g3, double* u3) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // Four rows x four 64-column chunks per wave pass: sixteen loads in flight (a pass of eight rows x one chunk
  // was 32 dependent memory round trips for ONE 256-variable problem — 38 of the step kernel's 52 us,
  // tools/step_stamps.py).  Per row the products are accumulated chunk after chunk as before: same bits.
  constexpr int RB = 8, JU = 1;
  for (int i0 = w; i0 < n; i0 += NS_NW * RB) {
    double a1[RB], a2[RB], a3[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { a1[r] = 0.0; a2[r] = 0.0; a3[r] = 0.0; }
    for (int jj = 0; i0 + jj < n; jj += WAVE * JU) {        // (wave-uniform: the longest row, i0)
      double rv[JU][RB];
#pragma unroll
      for (int q = 0; q < JU; ++q) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int i = i0 + r * NS_NW;
          const int ic = (i < n) ? i : n - 1;
          const int j = ic + lane + jj + WAVE * q;
          rv[q][r] = R[(long)ic * ld + ((j < n) ? j : n - 1)];
        }
      }
#pragma unroll
      for (int q = 0; q < JU; ++q) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int i = i0 + r * NS_NW;
          const int j = i + lane + jj + WAVE * q;
          if (i < n && j < n) {
            const double rd = dvec ? rv[q][r] * dvec[j] : rv[q][r];
            a1[r] = fma(rd, s1[j], a1[r]);
            if (s2) a2[r] = fma(rd, s2[j], a2[r]);
            a3[r] = fma(rd, -g3[j], a3[r]);
          }
        }
      }
    }
    // The 24 row totals by two transposed butterflies (wave_sum16: the same tree as wave_sum for every one of them — xor 1,
    // 2, 4, 8 inside the 16-lane rows, then (r0 + r16) + (r32 + r48) — in 15 exchanges per sixteen totals instead of 64).
    static_assert(RB == 8, "two sixteen-value reductions");
    double v[16];
    const int idx = wave_sum16_index(lane), ri = i0 + (idx & 7) * NS_NW;
#pragma unroll
    for (int r = 0; r < RB; ++r) { v[r] = a1[r]; v[8 + r] = s2 ? a2[r] : 0.0; }
    wave_sum16(v);
    if (lane < 16 && ri < n) {
      if (idx < 8) u1[ri] = v[0];
      else if (s2) u2[ri] = v[0];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) { v[r] = a3[r]; v[8 + r] = 0.0; }
    wave_sum16(v);
    if (lane < 16 && idx < 8 && ri < n) u3[ri] = v[0];
  }
  __syncthreads();
}
__device__ void full_matvec3(const double* M, int n, int ld, const double* s1, double* u1,
                             const double* s2, double* u2, const double* g3, double* u3) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int RB = 4;
  for (int i0 = w; i0 < n; i0 += NS_NW * RB) {
    double a1[RB], a2[RB], a3[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { a1[r] = 0.0; a2[r] = 0.0; a3[r] = 0.0; }
    for (int jj = lane; jj < n; jj += WAVE) {
      double rv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * NS_NW;
        rv[r] = M[(long)((i < n) ? i : n - 1) * ld + jj];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        a1[r] = fma(rv[r], s1[jj], a1[r]);
        if (s2) a2[r] = fma(rv[r], s2[jj], a2[r]);
        a3[r] = fma(rv[r], -g3[jj], a3[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * NS_NW;
      const double t1 = wave_sum(a1[r]);
      const double t2 = s2 ? wave_sum(a2[r]) : 0.0;
      const double t3 = wave_sum(a3[r]);
      if (lane == 0 && i < n) { u1[i] = t1; if (s2) u2[i] = t2; u3[i] = t3; }
    }
  }
  __syncthreads();
}

__device__ double dot_dev(const double* a, const double* b, int n, double* red) {
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += NS_NT) acc += a[j] * b[j];
  return block_sum(acc, red);
}
__device__ double dot3_dev(const double* a, const double* w, const double* b, int n,
                           double* red) {      // sum (a*w)*b   (np.dot(a*w, b))
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += NS_NT) acc += (a[j] * w[j]) * b[j];
  return block_sum(acc, red);
}

// trf.py:15-34 (minimize_quadratic)
__device__ double quad_min_dev(double a, double b, double lo, double hi) {
  double tbest = lo, ybest = a * (lo * lo) + b * lo;
  const double yh = a * (hi * hi) + b * hi;
  if (yh < ybest) { ybest = yh; tbest = hi; }
  if (a != 0.0) {
    const double ext = -0.5 * b / a;
    if (lo <= ext && ext <= hi) {
      const double ye = a * (ext * ext) + b * ext;
      if (ye < ybest) { ybest = ye; tbest = ext; }
    }
  }
  return tbest;
}

// MV3: the three model products of the reflective branch in one pass over the matrix (n >= 128, where
// the matrix streams from HBM: 0.118 -> 0.106 ms at 4096 x 256 x 512; for small n the extra
// accumulators cost occupancy — 0.042 -> 0.060 ms at 512 x 64 x 1024 — so those keep three passes).
// Same results bit for bit.
#ifdef BLSQ_CHOL_STAMPS
__device__ long long g_step_st[32];                    // phases of ONE problem's step (diagnostic build; tools/step_stamps.py)
#define SST(i) do { if (b == 0 && tid == 0) g_step_st[i] = (long long)wall_clock64(); } while (0)
int step_debug_stamps(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_step_st), sizeof(g_step_st)); }
#else
#define SST(i) do { } while (0)
#endif
template <bool MV3>
__global__ __launch_bounds__(NS_NT) void trf_step_kernel(TrfState st, const int* lm_fast,
                                                         const double* lm_ph, const double* lm_sc,
                                                         const int* lm_st, const double* Delta_in,
                                                         const double* alpha_in,
                                                         double active_rtol, TrfStepOut out, int* lm_counts,
                                                         PublishArgs pub) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b == 0 && tid == 0) publish_ints(pub);            // (the verdict counters of the factor call: final since it ended)
  // (the Newton-round counters of the NEXT step call: the rounds of this one are over — a fill launch less)
  if (b == 0 && tid < 16 && lm_counts) lm_counts[tid] = 0;
  const int n = st.n, ld = st.ld, m = st.m;
  const long vo = (long)b * ld;
  double* ph = sh;            // p_h
  double* rh = ph + ld;       // reflected direction / step
  double* tmp = rh + ld;      // scratch vector (p, r, steps)
  double* tmp2 = tmp + ld;
  double* up = tmp2 + ld;     // R_h p_h
  double* ur = up + ld;       // R_h r_h
  double* ug = ur + ld;       // R_h g_h
  double* coef = ug + ld;

  const double* sv = st.s + vo;
  const double* uf = st.uf + vo;
  const double* X = st.X + (long)b * ld * ld;
  const double* Rh = st.Rt + (long)b * ld * ld;           // R; R_h = R D is applied on the fly
  // Normal-equations path: there is no R.  X^T X = H = J_h^T J_h + diag_h (X: the triangle R_aug, or
  // after the SVD the rows s_i v_i^T — an orthogonal transformation of it), so every quadratic form
  // of the model, (J_h a).(J_h b) + a.diag_h.b, is (X a).(X b): the diag_h terms below drop out.
  const bool gp = st.path && st.path[b] == 0;
  // CSNE tier (csne_kernels.hip): the factor X of such a problem is a preconditioner only — its products with the
  // trust-region solution p would carry the Gram's error, relative to (X p)^2 of order eps kappa_2.  The corrected p
  // satisfies the normal equations, so every product with it is taken from  H p = -(c g_h + alpha p)  (hp, written by
  // csne_fix_kernel); products of vectors that are NOT in the small singular subspace still go through X.
  const bool cs_ = gp && st.csne && st.csne[b] && lm_fast && lm_fast[b];
  const double* hp = st.csne_hp + vo;
  const double* xg = st.x + vo;
  const double* dg = st.d + vo;
  const double* gh = st.g_h + vo;
  const double* dh = st.diag_h + vo;
  StepCtx cx{n, ld, xg, st.lb + vo, st.ub + vo, dg, red};
  const double Delta = Delta_in[b];
  const double alpha0 = alpha_in[b];
  const double theta = st.theta[b];
  int status = 0;

  SST(0);
  // ---------------- solve_lsq_trust_region (trust_region.py:56-152) --------
  const double smax = st.srange[2 * b], smin = st.srange[2 * b + 1];
  bool full_rank = false;
  if (m >= n) full_rank = smin > EPS * m * smax;
  double alpha = 0.0;
  int n_iter = 0;
  bool have_p = false;
  const bool fast = lm_fast && lm_fast[b];       // SVD-free path already produced p (lm_kernels.hip)
  auto model_mv = [&](const double* svec, double* u) {    // u = J_h s  (resp. X s on the Gram path)
    if (!gp) tri_matvec(Rh, dg, n, ld, svec, u);
    else if (fast) tri_matvec(X, nullptr, n, ld, svec, u);
    else full_matvec(X, n, ld, svec, u);
  };
  if (fast) {
    for (int j = tid; j < n; j += NS_NT) ph[j] = lm_ph[vo + j];
    alpha = lm_sc[(long)b * 16];
    n_iter = lm_st[(long)b * 4 + 2];
    have_p = true;
  } else if (full_rank) {
    // p = -V (uf / s); rows of X are s_i v_i^T
    for (int i = tid; i < n; i += NS_NT) coef[i] = (uf[i] / sv[i]) / sv[i];
    __syncthreads();
    double pn = 0.0;
    for (int j = tid; j < n; j += NS_NT) {
      double acc = 0.0;
      for (int i0 = 0; i0 < n; i0 += 8) {     // 8 rows per pass, loads in flight together
        double xv[8];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) xv[u8] = X[(long)((i0 + u8 < n) ? i0 + u8 : n - 1) * ld + j];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8)
          if (i0 + u8 < n) acc = fma(xv[u8], coef[i0 + u8], acc);
      }
      ph[j] = -acc;
      pn += acc * acc;
    }
    pn = sqrt(block_sum(pn, red));
    if (pn <= Delta) have_p = true;
  }
  if (!have_p) {
    double s2 = 0.0;
    for (int i = tid; i < n; i += NS_NT) { const double t = sv[i] * uf[i]; s2 += t * t; }
    double a_hi = sqrt(block_sum(s2, red)) / Delta;
    double a_lo = 0.0;
    double phi = 0.0, dphi = 0.0;
    auto secular = [&](double al) {
      double q1 = 0.0, q2 = 0.0;
      for (int i = tid; i < n; i += NS_NT) {
        const double suf = sv[i] * uf[i];
        const double den = sv[i] * sv[i] + al;
        const double r = suf / den;
        q1 += r * r;
        q2 += (suf * suf) / (den * den * den);
      }
      q1 = block_sum(q1, red);
      q2 = block_sum(q2, red);
      const double pnorm = sqrt(q1);
      phi = pnorm - Delta;
      dphi = -q2 / pnorm;
    };
    if (full_rank) {
      secular(0.0);
      a_lo = -phi / dphi;
    }
    if (!full_rank && alpha0 == 0.0) {
      const double gm = sqrt(a_lo * a_hi);
      alpha = (0.001 * a_hi > gm) ? 0.001 * a_hi : gm;
    } else {
      alpha = alpha0;
    }
    int it = 0;
    for (; it < 10; ++it) {
      if (alpha < a_lo || alpha > a_hi) {
        const double gm = sqrt(a_lo * a_hi);
        alpha = (0.001 * a_hi > gm) ? 0.001 * a_hi : gm;
      }
      secular(alpha);
      if (fabs(phi) < 0.01 * Delta) break;
      if (phi < 0.0) a_hi = alpha;
      const double ratio = phi / dphi;
      const double cand = alpha - ratio;
      a_lo = (cand > a_lo) ? cand : a_lo;       // max(alpha_lower, alpha - ratio)
      alpha -= (phi + Delta) * ratio / Delta;
    }
    n_iter = (it < 10) ? it + 1 : 10;
    for (int i = tid; i < n; i += NS_NT) coef[i] = uf[i] / (sv[i] * sv[i] + alpha);
    __syncthreads();
    double pn = 0.0;
    for (int j = tid; j < n; j += NS_NT) {
      double acc = 0.0;
      for (int i0 = 0; i0 < n; i0 += 8) {     // 8 rows per pass, loads in flight together
        double xv[8];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) xv[u8] = X[(long)((i0 + u8 < n) ? i0 + u8 : n - 1) * ld + j];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8)
          if (i0 + u8 < n) acc = fma(xv[u8], coef[i0 + u8], acc);
      }
      ph[j] = -acc;
      pn += acc * acc;
    }
    pn = sqrt(block_sum(pn, red));
    if (phi > 0.0) {
      const double f = Delta / pn;
      for (int j = tid; j < n; j += NS_NT) ph[j] *= f;
    }
  }
  __syncthreads();
  for (int j = tid; j < n; j += NS_NT) {
    out.p_h_tr[vo + j] = ph[j];
    tmp[j] = dg[j] * ph[j];                                  // p = d * p_h
  }
  __syncthreads();

  SST(1);
  // ---------------- feasibility of x + p (trf.py:286-292) ------------------
  const double to_bound = step_to_bound_dev(cx, xg, tmp, tmp2);
  for (int j = tid; j < n; j += NS_NT) {
    const long long hit = (tmp2[j] == to_bound) ? (long long)sign_of(tmp[j]) : 0;
    out.hits[vo + j] = hit;
  }
  SST(2);
  int branch, choice = 0;
  double qp[3] = {0.0, 0.0, 0.0};
  const double* step_h;
  __syncthreads();

  if (to_bound >= 1.0) {
    branch = 0;
    const double tt = theta * to_bound;
    const double f = (tt < 1.0) ? tt : 1.0;                  // min(theta*to_bound, 1)
    for (int j = tid; j < n; j += NS_NT) { ph[j] *= f; tmp[j] = dg[j]; }   // (d staged in LDS: tmp = p is done with)
    __syncthreads();
    double q2;
    if (cs_) q2 = f * dot_dev(ph, hp, n, red);            // (f p)^T H (f p) = f (f p) . hp
    else {
      if (!gp) tri_matvec(Rh, tmp, n, ld, ph, up);
      else model_mv(ph, up);
      q2 = dot_dev(up, up, n, red);
    }
    double dq = 0.0, lin = 0.0;
    for (int j = tid; j < n; j += NS_NT) {
      dq += dh[j] * (ph[j] * ph[j]);
      lin += ph[j] * gh[j];
    }
    dq = block_sum(dq, red);
    lin = block_sum(lin, red);
    qp[0] = (gp ? 0.5 * q2 : 0.5 * (q2 + dq)) + lin;
    step_h = ph;
  } else {
    branch = 1;
    // ---- find_reflected_step (trf.py:105-156) -----------------------------
    const double p_stride = to_bound;
    for (int j = tid; j < n; j += NS_NT) {
      const bool hit = (tmp2[j] == to_bound) && (tmp[j] != 0.0);
      double rj = ph[j];
      if (hit) rj *= -1.0;
      rh[j] = rj;                                            // r_h
      if (cs_) ur[j] = hit ? ph[j] : 0.0;                    // the hit components of p_h (CSNE: X r_h = X p_h - 2 X p_hits)
    }
    __syncthreads();
    double cs_pp = 0.0, cs_hp = 0.0, cs_rp = 0.0;            // p.Hp, p_hits.Hp, r.Hp  (unscaled p)
    if (cs_) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int j = tid; j < n; j += NS_NT) { a0 += ph[j] * hp[j]; a1 += ur[j] * hp[j]; a2 += rh[j] * hp[j]; }
      cs_pp = block_sum(a0, red); cs_hp = block_sum(a1, red); cs_rp = block_sum(a2, red);
    }
    for (int j = tid; j < n; j += NS_NT) {
      const double pj = tmp[j] * p_stride;                   // p *= p_stride
      ph[j] = ph[j] * p_stride;                              // p_h *= p_stride
      tmp2[j] = xg[j] + pj;                                  // x_on_bound
      tmp[j] = dg[j] * rh[j];                                // r = d * r_h
    }
    __syncthreads();
    SST(3);
    // intersect_trust_region(p_h, r_h, Delta) (trust_region.py:11-44)
    const double ia = dot_dev(rh, rh, n, red);
    const double ib = dot_dev(ph, rh, n, red);
    const double ic = dot_dev(ph, ph, n, red) - Delta * Delta;
    double to_tr = 0.0;
    if (ia == 0.0) status = 1;
    else if (ic > 0.0) status = 2;
    else {
      const double disc = sqrt(ib * ib - ia * ic);
      const double qq = -(ib + copysign(disc, ib));
      const double t1 = qq / ia, t2 = ic / qq;
      to_tr = (t1 < t2) ? t2 : t1;
    }
    SST(4);
    double* steps2 = coef;
    double to_b2 = step_to_bound_dev(cx, tmp2, tmp, steps2);
    to_b2 *= theta;
    const double r_hi = (to_tr < to_b2) ? to_tr : to_b2;     // min(to_bound, to_tr)
    double r_lo;
    if (r_hi > 0.0) r_lo = (1.0 - theta) * p_stride / r_hi;
    else r_lo = -1.0;
    SST(5);
    // J_h p_h (stride-scaled), J_h r_h (if the reflected step exists) and J_h (-g_h) in one pass
    const bool need_r = (status == 0 && r_lo <= r_hi);
    if constexpr (MV3) {
      // (d and g_h staged in LDS — tmp and tmp2 are free from here to the gradient step: the pass reads them once per
      //  matrix element, and from global memory every one of those reads was a dependent load in front of its fma)
      for (int j = tid; j < n; j += NS_NT) { tmp[j] = dg[j]; tmp2[j] = gh[j]; }
      __syncthreads();
      if (!gp) tri_matvec3(Rh, tmp, n, ld, ph, up, need_r ? rh : nullptr, ur, tmp2, ug);
      else if (cs_) tri_matvec3(X, nullptr, n, ld, ur, up, nullptr, nullptr, tmp2, ug);   // up = X p_hits
      else if (fast) tri_matvec3(X, nullptr, n, ld, ph, up, need_r ? rh : nullptr, ur, tmp2, ug);
      else full_matvec3(X, n, ld, ph, up, need_r ? rh : nullptr, ur, tmp2, ug);
    } else if (cs_) {
      model_mv(ur, up);                                      // up = X p_hits
    } else {
      model_mv(ph, up);
      if (need_r) model_mv(rh, ur);
    }
    // CSNE: the three quadratic forms of the branch from the identities (p_stride = s):
    //   |X (s p)|^2 = s^2 p.Hp,   (X s p).(X r) = s r.Hp,   |X r|^2 = p.Hp - 4 p_hits.Hp + 4 |X p_hits|^2
    double cs_q0 = 0.0, cs_uv = 0.0, cs_vv = 0.0;
    if (cs_) {
      const double hh = dot_dev(up, up, n, red);
      cs_q0 = (p_stride * p_stride) * cs_pp;
      cs_uv = p_stride * cs_rp;
      cs_vv = (cs_pp - 4.0 * cs_hp) + 4.0 * hh;
    }
    SST(6);
    bool have_r = false;
    double r_t = 0.0;
    if (need_r) {
      const double vv = cs_ ? cs_vv : dot_dev(ur, ur, n, red);
      const double sds = dot3_dev(rh, dh, rh, n, red);
      const double qa = gp ? 0.5 * vv : 0.5 * (vv + sds);
      double qb = dot_dev(gh, rh, n, red);
      const double uv = cs_ ? cs_uv : dot_dev(up, ur, n, red);
      const double s0ds = dot3_dev(ph, dh, rh, n, red);
      qb += gp ? uv : uv + s0ds;
      r_t = quad_min_dev(qa, qb, r_lo, r_hi);
      have_r = true;
    }
    SST(7);
    // r_h = p_h + r_h * r_stride ;  p_h *= theta
    for (int j = tid; j < n; j += NS_NT) {
      const double pj = ph[j];
      const double pt = pj * theta;
      rh[j] = have_r ? (pj + rh[j] * r_t) : pt;
      ph[j] = pt;
    }
    __syncthreads();
    // ---- find_gradient_step (trf.py:159-170) ------------------------------
    for (int j = tid; j < n; j += NS_NT) tmp[j] = -gh[j] * dg[j];
    __syncthreads();
    double to_bg = step_to_bound_dev(cx, xg, tmp, steps2);
    to_bg *= theta;
    SST(8);
    const double ghn = sqrt(dot_dev(gh, gh, n, red));
    const double to_trg = Delta / ghn;
    double g_hi = (to_trg < to_bg) ? to_trg : to_bg;
    for (int j = tid; j < n; j += NS_NT) tmp[j] = -gh[j];
    __syncthreads();
    if constexpr (!MV3) model_mv(tmp, ug);                       // J_h (-g_h)  (MV3: the pass above)
    const double gvv = dot_dev(ug, ug, n, red);
    const double gsds = dot3_dev(tmp, dh, tmp, n, red);
    const double ga = gp ? 0.5 * gvv : 0.5 * (gvv + gsds);
    const double gb = dot_dev(gh, tmp, n, red);
    const double g_t = quad_min_dev(ga, gb, 0.0, g_hi);
    for (int j = tid; j < n; j += NS_NT) tmp[j] = -g_t * gh[j];   // c_h
    __syncthreads();
    SST(9);
    // ---- evaluate_quadratic_function on [p_h, r_h, c_h] (trf.py:79-102) ----
    // J_h p_h(final) = theta * up ;  J_h r_h = up + r_t ur ;  J_h c_h = g_t * ug
    double q0 = 0.0, q1 = 0.0, q2 = 0.0;
    for (int j = tid; j < n; j += NS_NT) {
      const double a0 = theta * up[j];
      const double a1 = have_r ? (up[j] + r_t * ur[j]) : a0;
      const double a2 = g_t * ug[j];
      q0 += a0 * a0; q1 += a1 * a1; q2 += a2 * a2;
    }
    q0 = block_sum(q0, red); q1 = block_sum(q1, red); q2 = block_sum(q2, red);
    if (cs_) {                                               // (up / ur hold other things: the forms from the identities)
      q0 = (theta * theta) * cs_q0;
      q1 = have_r ? (cs_q0 + 2.0 * r_t * cs_uv) + (r_t * r_t) * cs_vv : q0;
    }
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, l0 = 0.0, l1 = 0.0, l2 = 0.0;
    for (int j = tid; j < n; j += NS_NT) {
      d0 += dh[j] * (ph[j] * ph[j]);  l0 += ph[j] * gh[j];
      d1 += dh[j] * (rh[j] * rh[j]);  l1 += rh[j] * gh[j];
      d2 += dh[j] * (tmp[j] * tmp[j]); l2 += tmp[j] * gh[j];
    }
    d0 = block_sum(d0, red); d1 = block_sum(d1, red); d2 = block_sum(d2, red);
    l0 = block_sum(l0, red); l1 = block_sum(l1, red); l2 = block_sum(l2, red);
    if (gp) { d0 = 0.0; d1 = 0.0; d2 = 0.0; }                    // (already inside the X products)
    qp[0] = 0.5 * (q0 + d0) + l0;
    qp[1] = have_r ? (0.5 * (q1 + d1) + l1) : qp[0];
    qp[2] = 0.5 * (q2 + d2) + l2;
    choice = 0;
    if (qp[1] < qp[choice]) choice = 1;
    if (qp[2] < qp[choice]) choice = 2;
    step_h = (choice == 0) ? ph : ((choice == 1) ? rh : tmp);
  }

  SST(10);
  // ---------------- step, x_new (trf.py:301-308,318-324) -------------------
  const double pred = -2.0 * qp[choice];
  double sn2 = 0.0, corr = 0.0;
  for (int j = tid; j < n; j += NS_NT) {
    const double sh_ = step_h[j];
    const double stp = dg[j] * sh_;
    const double lj = cx.lb[j], uj = cx.ub[j];
    const double xs = xg[j] + stp;
    double xn = xs;
    if (xs <= lj) xn = next_after(lj, uj);                   // bounds.py:91-94
    if (xs >= uj) xn = next_after(uj, lj);                   // bounds.py:97-99
    out.step_h[vo + j] = sh_;
    out.step[vo + j] = stp;
    out.x_new[vo + j] = xn;
    sn2 += sh_ * sh_;
    corr += (sh_ * dh[j]) * sh_;
    // find_active_constraints(x_new, lb, ub, rtol) (bounds.py:51-76)
    const double lower = xn - lj, upper = uj - xn;
    long long act = 0;
    if (lower < upper) {
      const double thr = active_rtol * ((fabs(lj) > 1.0) ? fabs(lj) : 1.0);
      if (lower < thr) act = -1;
    } else {
      const double thr = active_rtol * ((fabs(uj) > 1.0) ? fabs(uj) : 1.0);
      if (upper < thr) act = 1;
    }
    out.active_new[vo + j] = act;
  }
  SST(11);
  sn2 = block_sum(sn2, red);
  corr = block_sum(corr, red);
  if (tid == 0) {
    double* sc = out.scal + (long)b * 8;
    sc[0] = pred; sc[1] = sqrt(sn2); sc[2] = corr; sc[3] = alpha; sc[4] = to_bound;
    sc[5] = qp[0]; sc[6] = qp[1]; sc[7] = qp[2];
    int* inf = out.info + (long)b * 4;
    inf[0] = n_iter; inf[1] = branch; inf[2] = choice; inf[3] = status;
  }
  SST(12);
}

hipError_t launch_trf_step(const TrfState& st, const LmState* lm, const double* Delta,
                           const double* alpha_in, double active_rtol, const TrfStepOut& out,
                           hipStream_t s, const PublishArgs* pub) {
  const PublishArgs pa = pub ? *pub : PublishArgs{nullptr, 0, nullptr, 0};
  const size_t lds = sizeof(double) * 8 * (size_t)st.ld;
  if (st.n >= 128)
    hipLaunchKernelGGL(trf_step_kernel<true>, dim3(st.B), dim3(NS_NT), lds, s, st,
                       lm ? lm->fast : nullptr, lm ? lm->ph : nullptr, lm ? lm->sc : nullptr,
                       lm ? lm->st : nullptr, Delta, alpha_in, active_rtol, out, lm ? lm->active_count : nullptr, pa);
  else
    hipLaunchKernelGGL(trf_step_kernel<false>, dim3(st.B), dim3(NS_NT), lds, s, st,
                       lm ? lm->fast : nullptr, lm ? lm->ph : nullptr, lm ? lm->sc : nullptr,
                       lm ? lm->st : nullptr, Delta, alpha_in, active_rtol, out, lm ? lm->active_count : nullptr, pa);
  return hipGetLastError();
}

// ---- caller vectors -> state layout (device to device) -----------------------------------------
__global__ void pack_vecs_kernel(PackVecs pv, int n, int ld, int B) {
  const long total = (long)B * n;
  const int v = blockIdx.y;
  if (blockIdx.x == 0 && v == 0 && pv.zero && (int)threadIdx.x < pv.nzero) pv.zero[threadIdx.x] = 0;
  const unsigned long long* src = (const unsigned long long*)pv.src[v];
  unsigned long long* dst = (unsigned long long*)pv.dst[v];
  if (!src || !dst) return;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / n;
    const int i = (int)(e - b * n);
    dst[b * ld + i] = src[e];
  }
}
hipError_t launch_pack_vecs(const PackVecs& pv, int n, int ld, int B, hipStream_t s) {
  const long total = (long)B * n;
  int gx = (int)((total + 255) / 256);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(pack_vecs_kernel, dim3(gx, 5), dim3(256), 0, s, pv, n, ld, B);
  return hipGetLastError();
}

}  // namespace blsq
