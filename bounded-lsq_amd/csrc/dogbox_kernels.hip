// n-space kernels of the dogbox (rectangular trust region dogleg) step.
//
// Reference: bounded_lsq/dogbox.py:170-220 and its helpers :9-97, on the
// triangle R~ = [R c] of [J f]  (J = Q R, c = Q^T f):
//   g = R^T c ; J_free = Q R[:, free]
//   lstsq(J_free, -f)  <->  min-norm solve with the SVD of the triangle of
//                           R[:, free] (gelsd semantics: s_i <= rcond*s_max
//                           dropped, rcond = eps*max(m, n_free))
//   Js = J_free s      =>   Js.Js = |R s_full|^2 ,  Js.f = (R s_full).c
// Compiled with -ffp-contract=off: masks use exact == (dogbox.py:29-33).
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int DG_NT = 256;
static constexpr int DG_NW = DG_NT / WAVE;
static constexpr double EPS = 2.220446049250313e-16;

// u = R s  (R upper triangular n x n, row-major, stride ld): one wave per row, eight rows per wave pass — their
// loads in flight together (a row at a time was a memory round trip per row: 16 in a row for n = 64, most of the step
// kernel's 19 us), their eight totals by one transposed butterfly (wave_sum16: the tree of wave_sum for each).  Per row
// the products are accumulated in the same order as before: same bits.
__device__ static void tri_matvec_d(const double* R, int n, int ld, const double* svec,
                                    double* u) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int RB = 8;
  for (int i0 = w; i0 < n; i0 += DG_NW * RB) {
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
    for (int jj = 0; i0 + jj < n; jj += WAVE) {             // (wave-uniform: the longest row, i0)
      double rv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * DG_NW;
        const int ic = (i < n) ? i : n - 1;
        const int j = ic + lane + jj;
        rv[r] = R[(long)ic * ld + ((j < n) ? j : n - 1)];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * DG_NW;
        const int j = i + lane + jj;
        if (i < n && j < n) acc[r] = fma(rv[r], svec[j], acc[r]);
      }
    }
    double v[16];
#pragma unroll
    for (int r = 0; r < RB; ++r) { v[r] = acc[r]; v[8 + r] = 0.0; }
    wave_sum16(v);
    const int idx = wave_sum16_index(lane), ri = i0 + (idx & 7) * DG_NW;
    if (lane < 16 && idx < 8 && ri < n) u[ri] = v[0];
  }
  __syncthreads();
}

// u = M s for a dense nf x nf block (Jacobi rows of a factor that went through the SVD)
__device__ static void full_matvec_d(const double* M, int n, int ld, const double* svec, double* u) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = w; i < n; i += DG_NW) {
    const double* row = M + (long)i * ld;
    double acc = 0.0;
    for (int j = lane; j < n; j += WAVE) acc = fma(row[j], svec[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) u[i] = acc;
  }
  __syncthreads();
}

// ------------------------------------------------------------------ prep --
__global__ __launch_bounds__(DG_NT) void dog_prep_kernel(DogState st, int jac_scaling, int from_gram,
                                                         const int* sel, int redo, PackVecs pk) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  __shared__ int nfree_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b == 0 && pk.zero && tid < pk.nzero) pk.zero[tid] = 0;
  if (sel && sel[b] <= 1) return;
  if (st.csne && tid == 0) st.csne[b] = 0;               // (a problem prepared afresh is on the CSNE tier only if selected again)
  const int n = st.n, ld = st.ld;
  const long vo = (long)b * ld;
  const double* Rt = st.Rt + (long)b * ld * ld;
  const double* Gk = from_gram ? st.Gk + (long)b * ld * ld : nullptr;
  double* S = st.S + (long)b * ld * ld;
  double* gm = sh;              // g masked to the free set (full length)
  double* u = gm + ld;
  int* actl = (int*)(u + ld);   // [ld] active flags (the ordered free list is built from LDS)

  for (int j = tid; j < n; j += DG_NT) {
    double gj = 0.0, nn = 0.0;
    if (from_gram) {                          // g = J^T f and ||J_j||^2 straight from the Gram
      gj = Gk[(long)j * ld + n];
      nn = Gk[(long)j * ld + j];
    } else {
    for (int i0 = 0; i0 <= j; i0 += 8) {      // 8 rows per pass, loads unconditional (clamped)
      double rv[8], cv[8];
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8) {
        const int i = (i0 + u8 <= j) ? i0 + u8 : j;
        rv[u8] = Rt[(long)i * ld + j];
        cv[u8] = Rt[(long)i * ld + n];
      }
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8) {
        if (i0 + u8 <= j) {
          gj = fma(rv[u8], cv[u8], gj);
          nn = fma(rv[u8], rv[u8], nn);
        }
      }
    }
    }
    if (pk.src[0]) {                                         // the caller's vectors (stride n) -> state layout
      const long so = (long)b * n + j;
      st.x[vo + j] = static_cast<const double*>(pk.src[0])[so];
      st.lb[vo + j] = static_cast<const double*>(pk.src[1])[so];
      st.ub[vo + j] = static_cast<const double*>(pk.src[2])[so];
      st.scale[vo + j] = static_cast<const double*>(pk.src[3])[so];
      st.on_bound[vo + j] = static_cast<const long long*>(pk.src[4])[so];
    }
    double sc = redo ? st.scale_in[vo + j] : st.scale[vo + j];
    if (!redo) st.scale_in[vo + j] = sc;
    if (jac_scaling == 1) {                                  // dogbox.py:141-144
      double jn = sqrt(nn);
      if (jn == 0.0) jn = 1.0;
      sc = 1.0 / jn;
    } else if (jac_scaling == 2) {                           // dogbox.py:165-168
      const double inv = 1.0 / sqrt(nn);
      sc = (inv < sc) ? inv : sc;
    }
    st.scale[vo + j] = sc;
    st.g[vo + j] = gj;
    const bool act = ((double)st.on_bound[vo + j] * gj) < 0.0;   // dogbox.py:172
    st.active[vo + j] = act ? 1 : 0;
    actl[j] = act ? 1 : 0;
    gm[j] = act ? 0.0 : gj;
  }
  __syncthreads();
  if (tid == 0) {                                            // ordered free list
    int q = 0;
    for (int j = 0; j < n; ++j)
      if (!actl[j]) st.free_idx[vo + q++] = j;
    nfree_s = q;
    st.ncols[b] = (q > 0) ? q + 1 : 0;
  }
  __syncthreads();
  const int nf = nfree_s;
  double gmax = 0.0;
  for (int j = tid; j < n; j += DG_NT) gmax = nanmax2(gmax, fabs(gm[j]));
  gmax = block_max(gmax, red);
  if (tid == 0) st.g_norm[b] = (nf > 0) ? gmax : 0.0;        // dogbox.py:182-188
  if (nf == 0) return;
  if (from_gram) return;        // the free-column system comes from the gathered Gram, the Cauchy step from X

  // compacted  [R[:, free] | c]  (dogbox.py:175 on the triangle)
  const int N = nf + 1;
  for (int idx = tid; idx < n * N; idx += DG_NT) {
    const int row = idx / N, q = idx - row * N;
    const int col = (q < nf) ? st.free_idx[vo + q] : n;
    S[(long)row * ld + q] = (col >= row) ? Rt[(long)row * ld + col] : 0.0;
  }
  // Cauchy step (dogbox.py:198-199):  -(g.g)/(Jg.Jg) * g_free
  tri_matvec_d(Rt, n, ld, gm, u);
  double gg = 0.0, uu = 0.0;
  for (int j = tid; j < n; j += DG_NT) { gg += gm[j] * gm[j]; uu += u[j] * u[j]; }
  gg = block_sum(gg, red);
  uu = block_sum(uu, red);
  const double fac = -gg / uu;
  for (int q = tid; q < nf; q += DG_NT) st.cauchy[vo + q] = fac * gm[st.free_idx[vo + q]];
}

hipError_t launch_dog_prep(const DogState& st, int jac_scaling, int from_gram, const int* sel,
                           int redo, hipStream_t s, const PackVecs* pk) {
  const size_t lds = sizeof(double) * 2 * (size_t)st.ld + sizeof(int) * (size_t)st.ld;
  const PackVecs none{{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, 0};
  hipLaunchKernelGGL(dog_prep_kernel, dim3(st.B), dim3(DG_NT), lds, s, st, jac_scaling, from_gram,
                     sel, redo, (pk && !sel) ? *pk : none);
  return hipGetLastError();
}

// ----------------------------------------------------------------- solve --
// newton = lstsq(J_free, -f)[0] from the Jacobi rows  X[i] = s_i v_i^T | uf_i
__global__ __launch_bounds__(DG_NT) void dog_solve_kernel(DogState st, const int* skip) {
  extern __shared__ double sh[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (skip && skip[b]) return;             // Newton step already solved without the SVD
  const int ld = st.ld;
  const long vo = (long)b * ld;
  const int N = st.ncols[b];
  if (N <= 0) return;
  const int nf = N - 1;
  const double* X = st.X + (long)b * ld * ld;
  const double* sv = st.s + vo;
  const double* uf = st.uf + vo;
  double* coef = sh;
  const int mx = (st.m > nf) ? st.m : nf;
  const double cut = (EPS * mx) * st.srange[2 * b];          // rcond * s_max
  for (int i = tid; i < nf; i += DG_NT)
    coef[i] = (sv[i] > cut) ? (uf[i] / sv[i]) / sv[i] : 0.0;
  __syncthreads();
  for (int q = tid; q < nf; q += DG_NT) {
    double acc = 0.0;
    for (int i0 = 0; i0 < nf; i0 += 8) {      // 8 rows per pass, loads in flight together
      double xv[8];
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8) xv[u8] = X[(long)((i0 + u8 < nf) ? i0 + u8 : nf - 1) * ld + q];
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8)
        if (i0 + u8 < nf) acc = fma(xv[u8], coef[i0 + u8], acc);
    }
    st.newton[vo + q] = -acc;
  }
}

hipError_t launch_dog_solve(const DogState& st, const int* skip, hipStream_t s) {
  const size_t lds = sizeof(double) * (size_t)st.ld;
  hipLaunchKernelGGL(dog_solve_kernel, dim3(st.B), dim3(DG_NT), lds, s, st, skip);
  return hipGetLastError();
}

// ------------------------------------------------------------------ step --
struct BoxCtx {
  int nf;
  const double *lt, *ut;      // intersection region (LDS, compact)
  double* red;
};

__device__ static bool in_box(const BoxCtx& c, const double* v) {   // bounds.py:19-21
  int bad = 0;
  for (int q = threadIdx.x; q < c.nf; q += DG_NT)
    if (!((v[q] >= c.lt[q]) && (v[q] <= c.ut[q]))) bad = 1;
  return !block_or(bad, c.red);
}

// step_size_to_bound(x0, dir, lt, ut) with per-element steps kept (bounds.py:24-48)
__device__ static double to_box(const BoxCtx& c, const double* x0, const double* dir,
                                double* steps) {
  double tmin = __builtin_inf();
  for (int q = threadIdx.x; q < c.nf; q += DG_NT) {
    const double dq = dir[q];
    const double xq = x0 ? x0[q] : 0.0;
    double t = __builtin_inf();
    if (dq != 0.0) t = nanmax2((c.lt[q] - xq) / dq, (c.ut[q] - xq) / dq);
    steps[q] = t;
    tmin = nanmin2(tmin, t);
  }
  return block_min(tmin, c.red);
}

#ifdef BLSQ_CHOL_STAMPS
__device__ long long g_dog_st[32];                     // phases of ONE problem's dogbox step (diagnostic build)
#define DST(i) do { if (b == 0 && tid == 0) g_dog_st[i] = (long long)wall_clock64(); } while (0)
int dog_debug_stamps(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dog_st), sizeof(g_dog_st)); }
#else
#define DST(i) do { } while (0)
#endif
__global__ __launch_bounds__(DG_NT) void dog_step_kernel(DogState st, const double* Delta_in,
                                                         DogStepOut out, PublishArgs pub) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  DST(6);
  if (b == 0 && tid == 0) publish_ints(pub);            // (the verdict counters of the factor call: final since it ended)
  DST(7);
  const int n = st.n, ld = st.ld;
  const long vo = (long)b * ld;
  const int N = st.ncols[b];
  int* info = out.info + (long)b * 4;
  double* scal = out.scal + (long)b * 4;
  if (N <= 0) {                                              // every variable active
    for (int j = tid; j < n; j += DG_NT) {
      out.step[vo + j] = 0.0;
      out.x_new[vo + j] = st.x[vo + j];
      out.on_bound_new[vo + j] = st.on_bound[vo + j];
    }
    if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = 1; info[3] = 0; scal[0] = 0.0; scal[1] = 0.0; }
    return;
  }
  const int nf = N - 1;
  const int* fidx = st.free_idx + vo;
  const double* Rt = st.Rt + (long)b * ld * ld;
  double* lt = sh;            double* ut = lt + ld;
  double* lbc = ut + ld;      double* ubc = lbc + ld;
  double* trb = ubc + ld;     double* stp = trb + ld;
  double* cau = stp + ld;     double* dif = cau + ld;
  double* stepsv = dif + ld;  double* full = stepsv + ld;
  double* u = full + ld;
  const double Delta = Delta_in[b];
  const double* newton = st.newton + vo;
  const double* cauchy = st.cauchy + vo;

  DST(0);
  // find_intersection (dogbox.py:9-35)
  for (int q = tid; q < nf; q += DG_NT) {
    const int j = fidx[q];
    const double xq = st.x[vo + j];
    const double tr = Delta * st.scale[vo + j];               // dogbox.py:203
    const double lc = st.lb[vo + j] - xq, uc = st.ub[vo + j] - xq;
    lbc[q] = lc; ubc[q] = uc; trb[q] = tr;
    lt[q] = nanmax2(lc, -tr);
    ut[q] = nanmin2(uc, tr);
  }
  __syncthreads();
  BoxCtx cx{nf, lt, ut, red};

  DST(1);
  // dogleg_step (dogbox.py:38-75); faces: -1 / 0 / +1 in `dif` slots as ints later
  int tr_hit = 0;
  bool newton_inside = in_box(cx, newton);
  // per-element face code kept in registers by owner threads: recomputed below
  double cs_a = 0.0, cs_t = 1.0;                             // (CSNE: the step is cs_a cauchy + cs_t newton)
  if (newton_inside) {
    for (int q = tid; q < nf; q += DG_NT) { stp[q] = newton[q]; stepsv[q] = __builtin_inf(); dif[q] = 0.0; }
    __syncthreads();
  } else {
    const bool c_in = in_box(cx, cauchy);
    double beta = 1.0;
    if (!c_in) beta = to_box(cx, nullptr, cauchy, stepsv);
    for (int q = tid; q < nf; q += DG_NT) {
      const double cq = c_in ? cauchy[q] : beta * cauchy[q];
      cau[q] = cq;
      dif[q] = newton[q] - cq;
    }
    __syncthreads();
    const double t = to_box(cx, cau, dif, stepsv);
    cs_a = (c_in ? 1.0 : beta) * (1.0 - t); cs_t = t;
    int th = 0;
    for (int q = tid; q < nf; q += DG_NT) {
      const double sgn = (stepsv[q] == t) ? sign_of(dif[q]) : 0.0;
      if ((sgn < 0.0 && lt[q] == -trb[q]) || (sgn > 0.0 && ut[q] == trb[q])) th = 1;
      stp[q] = cau[q] + t * dif[q];
      stepsv[q] = (stepsv[q] == t) ? sgn : 0.0;              // keep the hit sign
    }
    tr_hit = block_or(th, red);
  }
  DST(2);
  // bound_hits from hit signs (stepsv holds sign or 0; newton-inside -> none)
  // NB: when newton is inside, stepsv was set to +inf above: treat as no hit.
  auto face_of = [&](int q) -> long long {
    const double sgn = stepsv[q];
    if (sgn == -1.0 && lt[q] == lbc[q]) return -1;
    if (sgn == 1.0 && ut[q] == ubc[q]) return 1;
    return 0;
  };

  DST(3);
  // predicted reduction (dogbox.py:208-209):  Js.Js and Js.f
  const bool gp = st.path && st.path[b] == 0;
  double uu = 0.0, uc = 0.0;
  if (gp && st.csne && st.csne[b] && st.fast && st.fast[b]) {
    // CSNE tier: X is a preconditioner only — its products with the Newton step would carry the Gram's error.  The
    // corrected Newton step satisfies J_free^T J_free newton = -g_free, so with s = a c + t p (c the Cauchy step, a
    // multiple of g: not in the small singular subspace)
    //     |J s|^2 = a^2 |X c|^2 - 2 a t c.g_free - t^2 p.g_free
    const double* Xf = st.X + (long)b * ld * ld;
    tri_matvec_d(Xf, nf, ld, cauchy, u);
    double cc = 0.0, cg = 0.0, pg = 0.0;
    for (int q = tid; q < nf; q += DG_NT) {
      const double gq = st.g[vo + fidx[q]];
      cc += u[q] * u[q]; cg += cauchy[q] * gq; pg += newton[q] * gq; uc += stp[q] * gq;
    }
    cc = block_sum(cc, red); cg = block_sum(cg, red); pg = block_sum(pg, red);
    if (tid == 0) uu = ((cs_a * cs_a) * cc - 2.0 * (cs_a * cs_t) * cg) - (cs_t * cs_t) * pg;   // (block_sum below adds the lanes' shares)
  } else if (gp) {
    // normal-equations path: |J_free s|^2 = |X s|^2 (X: triangle of the free columns, or its Jacobi
    // rows), Js.f = s.g
    const double* Xf = st.X + (long)b * ld * ld;
    if (st.fast && st.fast[b]) tri_matvec_d(Xf, nf, ld, stp, u);
    else full_matvec_d(Xf, nf, ld, stp, u);
    for (int q = tid; q < nf; q += DG_NT) { uu += u[q] * u[q]; uc += stp[q] * st.g[vo + fidx[q]]; }
  } else {
    for (int j = tid; j < n; j += DG_NT) full[j] = 0.0;
    __syncthreads();
    for (int q = tid; q < nf; q += DG_NT) full[fidx[q]] = stp[q];
    __syncthreads();
    tri_matvec_d(Rt, n, ld, full, u);
    for (int i = tid; i < n; i += DG_NT) { uu += u[i] * u[i]; uc += u[i] * Rt[(long)i * ld + n]; }
  }
  uu = block_sum(uu, red);
  uc = block_sum(uc, red);
  const double pred = -uu - 2.0 * uc;
  int fallback = 0;
  if (pred <= 0.0) {                                         // dogbox.py:213-216
    fallback = 1;
    const bool c_in = in_box(cx, cauchy);
    if (c_in) {
      for (int q = tid; q < nf; q += DG_NT) { stp[q] = cauchy[q]; stepsv[q] = 0.0; }
      tr_hit = 0;
      __syncthreads();
    } else {
      const double beta = to_box(cx, nullptr, cauchy, stepsv);
      int th = 0;
      for (int q = tid; q < nf; q += DG_NT) {
        const double sgn = (stepsv[q] == beta) ? sign_of(cauchy[q]) : 0.0;
        if ((sgn < 0.0 && lt[q] == -trb[q]) || (sgn > 0.0 && ut[q] == trb[q])) th = 1;
        stp[q] = beta * cauchy[q];
        stepsv[q] = sgn;
      }
      tr_hit = block_or(th, red);
    }
  }
  __syncthreads();

  DST(4);
  // scatter (dogbox.py:218-220) and the caller-side pieces of :235,253
  for (int j = tid; j < n; j += DG_NT) {
    out.step[vo + j] = 0.0;
    out.x_new[vo + j] = st.x[vo + j] + 0.0;
    out.on_bound_new[vo + j] = st.on_bound[vo + j];
  }
  __syncthreads();
  double smax = 0.0;
  for (int q = tid; q < nf; q += DG_NT) {
    const int j = fidx[q];
    const double sq = stp[q];
    out.step[vo + j] = sq;
    out.x_new[vo + j] = st.x[vo + j] + sq;
    out.on_bound_new[vo + j] = face_of(q);
    smax = nanmax2(smax, fabs(sq / st.scale[vo + j]));
  }
  smax = block_max(smax, red);
  if (tid == 0) {
    scal[0] = pred; scal[1] = smax; scal[2] = 0.0; scal[3] = 0.0;
    info[0] = tr_hit; info[1] = fallback; info[2] = 0; info[3] = 0;
  }
  DST(5);
}

hipError_t launch_dog_step(const DogState& st, const double* Delta, const DogStepOut& out,
                           hipStream_t s, const PublishArgs* pub) {
  const size_t lds = sizeof(double) * 11 * (size_t)st.ld;
  hipLaunchKernelGGL(dog_step_kernel, dim3(st.B), dim3(DG_NT), lds, s, st, Delta, out,
                     pub ? *pub : PublishArgs{nullptr, 0, nullptr, 0});
  return hipGetLastError();
}

}  // namespace blsq
