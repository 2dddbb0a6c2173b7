// Batched outer trust-region drivers, device-resident.
//
// The reference keeps the accept/reject logic, the Delta / alpha updates and the
// termination tests in Python around each problem (trf.py:173-237,309-358;
// dogbox.py:100-163,221-272).  Here B problems of one shape advance in lock-step and
// that logic runs on the GPU, one workgroup (or one thread) per problem, so between two
// user callbacks only ONE integer (how many problems are still active / were accepted)
// crosses the boundary; x, f, J and every per-problem scalar stay in HBM.
//
// Per problem the control flow is exactly the reference's (and `_batch.py`'s):
//   top of the outer loop  -> nfev / gtol / pending-status checks        (outer_top)
//   step (existing kernels) -> x_trial                                   (outer_trial)
//   fun(x_trial) by the caller
//   ratio test, Delta/alpha update, ftol/xtol tests, accept              (outer_judge)
//   jac(x) by the caller for accepted problems, masked re-factorisation.
// A problem that has terminated is frozen: nothing of its state is written again.
//
// Compiled with -ffp-contract=off like the other n-space kernels: the scalar formulas
// follow the reference operation by operation.
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int OUT_NT = 256;
static constexpr double O_SQRT_EPS = 1.4901161193847656e-08;

enum { ST_NONE = -1 };

// ---- after the first factorisation: objective, initial radius, counters -------------------
__global__ __launch_bounds__(OUT_NT) void outer_begin_kernel(OuterState o) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = o.n, m = o.m, ld = o.ld;
  const double* f = o.f + (long)b * m;
  double acc = 0.0;
  for (int i = tid; i < m; i += OUT_NT) acc += f[i] * f[i];
  const double obj = block_sum(acc, red);
  // Delta_0 from the UNSHIFTED x0 (trf.py:223-226; dogbox.py:148-150)
  double dn = 0.0;
  if (o.method == 0) {
    double a2 = 0.0;
    for (int j = tid; j < n; j += OUT_NT) {
      const double t = o.x0[(long)b * n + j] / (o.scale[(long)b * ld + j] * sqrt(o.v[(long)b * ld + j]));
      a2 += t * t;
    }
    dn = sqrt(block_sum(a2, red));
  } else {
    double mx = 0.0;
    for (int j = tid; j < n; j += OUT_NT)
      mx = nanmax2(mx, fabs(o.x0[(long)b * n + j] / o.scale[(long)b * ld + j]));
    dn = block_max(mx, red);
  }
  for (int j = tid; j < n; j += OUT_NT) o.xc[(long)b * n + j] = o.x[(long)b * ld + j];
  if (tid == 0) {
    o.obj[b] = obj;
    o.Delta[b] = (dn == 0.0) ? 1.0 : dn;
    o.alpha[b] = 0.0;
    o.nfev[b] = 1; o.njev[b] = 1;
    o.pending[b] = ST_NONE; o.result[b] = 0; o.done[b] = 0; o.at_top[b] = 1;
    o.accepted[b] = 0; o.ncols_fac[b] = 0;
    o.actual[b] = -1.0; o.gnorm[b] = 0.0;
  }
}

// ---- top of the outer loop (trf.py:238-261 / dogbox.py:164-194): one thread per problem ----
__global__ void outer_top_kernel(OuterState o) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= o.B) return;
  if (o.done[b] || !o.at_top[b]) return;
  if (o.nfev[b] >= o.max_nfev) {                    // `while nfev < max_nfev` failed
    o.done[b] = 1; o.result[b] = 0;
    return;
  }
  int status = o.pending[b];
  double gn;
  if (o.method == 1 && o.ncols[b] == 0) {           // every variable active (dogbox.py:181-184)
    gn = 0.0; status = 1;
  } else {
    gn = o.g_norm_fac[b];
    if (gn < o.gtol) status = 1;
  }
  o.gnorm[b] = gn;
  if (status != ST_NONE) {
    o.done[b] = 1; o.result[b] = status;
    return;
  }
  o.at_top[b] = 0;
  o.actual[b] = -1.0;
}

// ---- trial point for the callback: x_new of active problems, x of frozen ones -------------
__global__ __launch_bounds__(OUT_NT) void outer_trial_kernel(OuterState o) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = o.n, ld = o.ld;
  bool act = !o.done[b];
  // The reference raises ValueError out of the step (trust_region.py:28-29,34-35) and the whole
  // solve aborts.  Here the problem is frozen with result = -status (BLSQ_STATUS_*) before its
  // trial point would be evaluated; the host wrappers raise the reference's exception for it.
  const int err = act ? o.o_info[(long)b * 4 + 3] : 0;
  __syncthreads();                                  // every thread has read done[b]
  if (err != 0) act = false;
  for (int j = tid; j < n; j += OUT_NT)
    o.xt[(long)b * n + j] = act ? o.o_xnew[(long)b * ld + j] : o.xc[(long)b * n + j];
  if (tid == 0) {
    o.accepted[b] = 0; o.ncols_fac[b] = 0;
    if (err != 0) { o.done[b] = 1; o.result[b] = -err; }
    if (act) atomicAdd(&o.counts[0], 1);
  }
}

// ---- after fun(x_trial): ratio test, radius update, termination, accept -------------------
__global__ __launch_bounds__(OUT_NT) void outer_judge_kernel(OuterState o) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (o.done[b]) return;                            // frozen (uniform per workgroup)
  const int n = o.n, m = o.m, ld = o.ld;
  const double* ft = o.ft + (long)b * m;
  double acc = 0.0;
  for (int i = tid; i < m; i += OUT_NT) acc += ft[i] * ft[i];
  const double obj_new = block_sum(acc, red);
  const double obj = o.obj[b];
  const double actual = obj - obj_new;
  double Delta = o.Delta[b], alpha = o.alpha[b];
  double ratio;
  bool xtol_ok;
  if (o.method == 0) {
    const double* sc = o.o_scal + (long)b * 8;      // pred, ||step_h||, correction, alpha_out
    const double pred = sc[0], shn = sc[1], corr = sc[2];
    alpha = sc[3];
    ratio = (pred > 0.0) ? (actual - corr) / pred : 0.0;          // trf.py:316-319
    if (ratio < 0.25) {                                            // trf.py:321-328
      const double Dn = 0.25 * shn;
      alpha *= Delta / Dn;
      Delta = Dn;
    } else if (ratio > 0.75 && shn > 0.95 * Delta) {
      Delta *= 2.0;
      alpha *= 0.5;
    }
    double s2 = 0.0, x2 = 0.0;
    for (int j = tid; j < n; j += OUT_NT) {
      const double sj = o.o_step[(long)b * ld + j], xj = o.x[(long)b * ld + j];
      s2 += sj * sj; x2 += xj * xj;
    }
    const double sn = sqrt(block_sum(s2, red));
    const double xn = sqrt(block_sum(x2, red));
    xtol_ok = sn < o.xtol * ((O_SQRT_EPS > xn) ? O_SQRT_EPS : xn);  // trf.py:335
  } else {
    const double* sc = o.o_scal + (long)b * 4;      // pred, ||step/scale||_inf
    const double pred = sc[0], ssn = sc[1];
    const int tr_hit = o.o_info[(long)b * 4];
    ratio = (pred > 0.0) ? actual / pred : 0.0;                    // dogbox.py:229-232
    if (ratio < 0.25) Delta = 0.25 * ssn;                          // dogbox.py:234-237
    else if (ratio > 0.75 && tr_hit) Delta *= 2.0;
    double mx = 0.0;
    for (int j = tid; j < n; j += OUT_NT)
      mx = nanmax2(mx, fabs(o.x[(long)b * ld + j] / o.scale[(long)b * ld + j]));
    const double xn = block_max(mx, red);
    xtol_ok = Delta < o.xtol * ((O_SQRT_EPS > xn) ? O_SQRT_EPS : xn);   // dogbox.py:241-242
  }
  const bool ftol_ok = (fabs(actual) < o.ftol * obj) && (ratio > 0.25);
  int status = ST_NONE;
  if (ftol_ok && xtol_ok) status = 4;
  else if (ftol_ok) status = 2;
  else if (xtol_ok) status = 3;
  const int nfev = o.nfev[b] + 1;
  const bool accept = actual > 0.0;
  if (accept) {                                     // trf.py:343-349 / dogbox.py:254-264
    for (int j = tid; j < n; j += OUT_NT) {
      double xj = o.o_xnew[(long)b * ld + j];
      if (o.method == 1) {
        const long long ob = o.o_onb[(long)b * ld + j];
        o.on_bound[(long)b * ld + j] = ob;
        if (ob == -1) xj = o.lb[(long)b * ld + j];
        if (ob == 1) xj = o.ub[(long)b * ld + j];
      }
      o.x[(long)b * ld + j] = xj;
      o.xc[(long)b * n + j] = xj;
    }
    double* f = o.f + (long)b * m;
    for (int i = tid; i < m; i += OUT_NT) f[i] = ft[i];
  }
  if (tid == 0) {
    o.nfev[b] = nfev;
    o.actual[b] = actual;
    o.Delta[b] = Delta;
    o.alpha[b] = alpha;
    if (status != ST_NONE) o.pending[b] = status;
    if (accept) {
      o.obj[b] = obj_new;
      o.njev[b] += 1;                               // the caller evaluates jac(x) next
      o.accepted[b] = 1;
      o.ncols_fac[b] = n + 1;
      atomicAdd(&o.counts[1], 1);
    }
    if (status != ST_NONE || accept || nfev >= o.max_nfev) o.at_top[b] = 1;
  }
}

hipError_t launch_outer_begin(const OuterState& o, hipStream_t s) {
  hipLaunchKernelGGL(outer_begin_kernel, dim3(o.B), dim3(OUT_NT), 0, s, o);
  return hipGetLastError();
}
hipError_t launch_outer_top(const OuterState& o, hipStream_t s) {
  hipLaunchKernelGGL(outer_top_kernel, dim3((o.B + 255) / 256), dim3(256), 0, s, o);
  return hipGetLastError();
}
hipError_t launch_outer_trial(const OuterState& o, hipStream_t s) {
  hipLaunchKernelGGL(outer_trial_kernel, dim3(o.B), dim3(OUT_NT), 0, s, o);
  return hipGetLastError();
}
hipError_t launch_outer_judge(const OuterState& o, hipStream_t s) {
  hipLaunchKernelGGL(outer_judge_kernel, dim3(o.B), dim3(OUT_NT), 0, s, o);
  return hipGetLastError();
}

}  // namespace blsq
