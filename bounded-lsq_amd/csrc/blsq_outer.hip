// C-ABI entry points (include/blsq.h): the batched, device-resident outer drivers and the finite-difference Jacobians.
#include "blsq_host.h"

// ==================================================== batched outer drivers ===
struct blsq_outer {
  blsq_ctx* ctx = nullptr;
  int method = 0, B = 0, m = 0, n = 0, ld = 0;
  blsq_trf_plan* trf = nullptr;
  blsq_dogbox_plan* dog = nullptr;
  DevBuf x0, xc, xt, f, ft, J, dvec, ivec, counts;
  OuterState st{};
  int jac_scaling = 0;
  double xtol = 0.0;
  bool started = false, begun = false;
  int last_accepted = 0;
};

extern "C" int blsq_outer_create(blsq_ctx* ctx, int method, int B, int m, int n,
                                 blsq_outer** out) {
  if (!ctx) return -1;
  if (!out) return ctx->bad(6, "out is NULL");
  *out = nullptr;
  if (method != 0 && method != 1) return ctx->bad(2, "method must be 0 (trf) or 1 (dogbox)");
  blsq_outer* o = new blsq_outer();
  o->ctx = ctx; o->method = method; o->B = B; o->m = m; o->n = n;
  int rc = (method == 0) ? blsq_trf_plan_create(ctx, B, m, n, &o->trf)
                         : blsq_dogbox_plan_create(ctx, B, m, n, &o->dog);
  if (rc) { delete o; return rc; }
  o->ld = (method == 0) ? o->trf->ld : o->dog->ld;
  const size_t vn = sizeof(double) * (size_t)B * n, vm = sizeof(double) * (size_t)B * m;
  hipError_t e = hipSuccess;
  auto al = [&](DevBuf& b, size_t bytes) { if (e == hipSuccess) e = b.alloc(bytes); };
  al(o->x0, vn); al(o->xc, vn); al(o->xt, vn); al(o->f, vm); al(o->ft, vm);
  al(o->J, vm * n); al(o->dvec, sizeof(double) * (size_t)B * 5);
  al(o->ivec, sizeof(int) * (size_t)B * 8); al(o->counts, sizeof(int) * 2);
  if (e != hipSuccess) { blsq_outer_destroy(o); return ctx->fail(e, "hipMalloc(outer driver)"); }
  OuterState& st = o->st;
  st.B = B; st.m = m; st.n = n; st.ld = o->ld; st.method = method;
  if (method == 0) {
    blsq_trf_plan* p = o->trf;
    st.x = p->st.x; st.lb = p->st.lb; st.ub = p->st.ub; st.scale = p->st.scale;
    st.g_norm_fac = p->st.g_norm; st.v = p->st.v; st.ncols = nullptr; st.on_bound = nullptr;
    st.o_step = p->out.step; st.o_xnew = p->out.x_new; st.o_scal = p->out.scal;
    st.o_info = p->out.info; st.o_onb = nullptr;
  } else {
    blsq_dogbox_plan* p = o->dog;
    st.x = p->st.x; st.lb = p->st.lb; st.ub = p->st.ub; st.scale = p->st.scale;
    st.g_norm_fac = p->st.g_norm; st.v = nullptr; st.ncols = p->st.ncols;
    st.on_bound = p->st.on_bound;
    st.o_step = p->out.step; st.o_xnew = p->out.x_new; st.o_scal = p->out.scal;
    st.o_info = p->out.info; st.o_onb = p->out.on_bound_new;
  }
  st.x0 = o->x0.as<double>(); st.xc = o->xc.as<double>(); st.xt = o->xt.as<double>();
  st.f = o->f.as<double>(); st.ft = o->ft.as<double>();
  double* dv = o->dvec.as<double>();
  st.Delta = dv; st.alpha = dv + B; st.obj = dv + 2 * (size_t)B; st.gnorm = dv + 3 * (size_t)B;
  st.actual = dv + 4 * (size_t)B;
  int* iv = o->ivec.as<int>();
  st.nfev = iv; st.njev = iv + B; st.pending = iv + 2 * (size_t)B; st.result = iv + 3 * (size_t)B;
  st.done = iv + 4 * (size_t)B; st.at_top = iv + 5 * (size_t)B; st.accepted = iv + 6 * (size_t)B;
  st.ncols_fac = iv + 7 * (size_t)B;
  st.counts = o->counts.as<int>();
  *out = o;
  return 0;
}

extern "C" int blsq_outer_destroy(blsq_outer* o) {
  if (!o) return 0;
  if (o->trf) blsq_trf_plan_destroy(o->trf);
  if (o->dog) blsq_dogbox_plan_destroy(o->dog);
  o->x0.release(); o->xc.release(); o->xt.release(); o->f.release(); o->ft.release();
  o->J.release(); o->dvec.release(); o->ivec.release(); o->counts.release();
  delete o;
  return 0;
}

extern "C" int blsq_outer_buffers(blsq_outer* o, double** x, double** x_trial, double** f,
                                  double** f_trial, double** J, int32_t** accepted) {
  if (!o) return -1;
  if (x) *x = o->st.xc;
  if (x_trial) *x_trial = o->st.xt;
  if (f) *f = o->st.f;
  if (f_trial) *f_trial = o->st.ft;
  if (J) *J = o->J.as<double>();
  if (accepted) *accepted = o->st.accepted;
  return 0;
}

extern "C" int blsq_outer_start(blsq_outer* o, const double* x0, const double* x_start,
                                const double* lb, const double* ub, const double* scale,
                                int jac_scaling, double ftol, double xtol, double gtol,
                                int max_nfev) {
  if (!o) return -1;
  blsq_ctx* ctx = o->ctx;
  if (!x0) return ctx->bad(2, "x0 is NULL");
  if (!x_start) return ctx->bad(3, "x_start is NULL");
  if (!lb || !ub) return ctx->bad(4, "lb/ub is NULL");
  if (!scale) return ctx->bad(6, "scale is NULL");
  if (max_nfev <= 0) return ctx->bad(11, "max_nfev must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int B = o->B, n = o->n, ld = o->ld;
  OuterState& st = o->st;
  int rc;
  if ((rc = put_vec(ctx, st.x, ld, x_start, n, B, hipMemcpyHostToDevice))) return rc;
  if ((rc = put_vec(ctx, st.lb, ld, lb, n, B, hipMemcpyHostToDevice))) return rc;
  if ((rc = put_vec(ctx, st.ub, ld, ub, n, B, hipMemcpyHostToDevice))) return rc;
  if ((rc = put_vec(ctx, st.scale, ld, scale, n, B, hipMemcpyHostToDevice))) return rc;
  const size_t vn = sizeof(double) * (size_t)B * n;
  HIPCHK(ctx, hipMemcpyAsync(st.x0, x0, vn, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(st.xc, x_start, vn, hipMemcpyHostToDevice, ctx->stream));
  if (o->method == 1) {
    // on_bound_0 from x0 == lb / ub exactly (dogbox.py:152-154)
    std::vector<long long> ob((size_t)B * n);
    for (size_t i = 0; i < ob.size(); ++i) ob[i] = (x0[i] == lb[i]) ? -1 : ((x0[i] == ub[i]) ? 1 : 0);
    HIPCHK(ctx, hipMemcpy2DAsync(st.on_bound, sizeof(long long) * ld, ob.data(),
                                 sizeof(long long) * n, sizeof(long long) * n, B,
                                 hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  st.ftol = ftol; st.xtol = xtol; st.gtol = gtol; st.max_nfev = max_nfev;
  o->xtol = xtol; o->jac_scaling = jac_scaling ? 1 : 0;
  o->started = true; o->begun = false; o->last_accepted = 0;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

namespace blsq_host {
// factor the problems selected by `mask` (nullptr: all) from the driver's J / f buffers
int outer_factor(blsq_outer* o, int scale_mode, const int* mask) {
  if (o->method == 0) {
    blsq_trf_plan* p = o->trf;
    return trf_factor_core(p, o->J.as<double>(), o->st.f, p->n, scale_mode, mask);
  }
  blsq_dogbox_plan* p = o->dog;
  return dog_factor_core(p, o->J.as<double>(), o->st.f, p->n, scale_mode, mask);
}
}  // namespace blsq_host

extern "C" int blsq_outer_begin(blsq_outer* o) {
  if (!o) return -1;
  blsq_ctx* ctx = o->ctx;
  if (!o->started) return ctx->bad(1, "blsq_outer_start has not been called");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = outer_factor(o, o->jac_scaling ? BLSQ_SCALE_JAC_INIT : BLSQ_SCALE_GIVEN, nullptr);
  if (rc) return rc;
  hipError_t e = launch_outer_begin(o->st, ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_outer_begin");
  o->begun = true; o->last_accepted = 0;
  return 0;
}

extern "C" int blsq_outer_propose(blsq_outer* o, int32_t* n_active) {
  if (!o) return -1;
  blsq_ctx* ctx = o->ctx;
  if (!o->begun) return ctx->bad(1, "blsq_outer_begin has not been called");
  if (!n_active) return ctx->bad(2, "n_active is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if (o->last_accepted > 0) {            // fresh Jacobians: factor those problems only
    rc = outer_factor(o, o->jac_scaling ? BLSQ_SCALE_JAC_UPDATE : BLSQ_SCALE_GIVEN,
                      o->st.ncols_fac);
    if (rc) return rc;
    o->last_accepted = 0;
  }
  hipError_t e = launch_outer_top(o->st, ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_outer_top");
  rc = (o->method == 0) ? blsq_trf_step_dev(o->trf, o->st.Delta, o->st.alpha, o->xtol)
                        : blsq_dogbox_step_dev(o->dog, o->st.Delta);
  if (rc) return rc;
  HIPCHK(ctx, hipMemsetAsync(o->st.counts, 0, sizeof(int) * 2, ctx->stream));
  e = launch_outer_trial(o->st, ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_outer_trial");
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, o->st.counts, sizeof(int), hipMemcpyDeviceToHost,
                             ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *n_active = ctx->pinned[0];
  return 0;
}

extern "C" int blsq_outer_judge(blsq_outer* o, int32_t* n_accepted) {
  if (!o) return -1;
  blsq_ctx* ctx = o->ctx;
  if (!o->begun) return ctx->bad(1, "blsq_outer_begin has not been called");
  if (!n_accepted) return ctx->bad(2, "n_accepted is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipError_t e = launch_outer_judge(o->st, ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_outer_judge");
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, o->st.counts + 1, sizeof(int), hipMemcpyDeviceToHost,
                             ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *n_accepted = ctx->pinned[0];
  o->last_accepted = ctx->pinned[0];
  return 0;
}

extern "C" int blsq_outer_fetch(blsq_outer* o, double* x, double* f, double* obj,
                                double* optimality, int64_t* on_bound, int32_t* nfev,
                                int32_t* njev, int32_t* status) {
  if (!o) return -1;
  blsq_ctx* ctx = o->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int B = o->B, n = o->n, m = o->m, ld = o->ld;
  const OuterState& st = o->st;
  auto d2h = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!dst) return 0;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
  };
  int rc;
  if ((rc = d2h(x, st.xc, sizeof(double) * (size_t)B * n))) return rc;
  if ((rc = d2h(f, st.f, sizeof(double) * (size_t)B * m))) return rc;
  if ((rc = d2h(obj, st.obj, sizeof(double) * B))) return rc;
  if ((rc = d2h(optimality, st.gnorm, sizeof(double) * B))) return rc;
  if ((rc = d2h(nfev, st.nfev, sizeof(int) * B))) return rc;
  if ((rc = d2h(njev, st.njev, sizeof(int) * B))) return rc;
  if ((rc = d2h(status, st.result, sizeof(int) * B))) return rc;
  if (on_bound) {
    if (o->method == 1) {
      HIPCHK(ctx, hipMemcpy2DAsync(on_bound, sizeof(long long) * n, st.on_bound,
                                   sizeof(long long) * ld, sizeof(long long) * n, B,
                                   hipMemcpyDeviceToHost, ctx->stream));
    } else {
      memset(on_bound, 0, sizeof(int64_t) * (size_t)B * n);
    }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// ============================================= finite-difference Jacobians ===
extern "C" int blsq_fd_points_dev(blsq_ctx* ctx, int B, int n, int method, const double* dx,
                                  const double* dlb, const double* dub, const double* drel_step,
                                  double* dX, double* dh, uint8_t* done_sided) {
  if (!ctx) return -1;
  if (B <= 0) return ctx->bad(2, "B must be positive");
  if (n <= 0) return ctx->bad(3, "n must be positive");
  if (method != 2 && method != 3) return ctx->bad(4, "method must be 2 or 3");
  if (!dx || !dlb || !dub) return ctx->bad(5, "x/lb/ub is NULL");
  if (!dX || !dh || !done_sided) return ctx->bad(9, "output is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipError_t e = launch_fd_points(B, n, method, dx, dlb, dub, drel_step, dX, dh, done_sided,
                                  ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_fd_points");
  return 0;
}

extern "C" int blsq_fd_assemble_dev(blsq_ctx* ctx, int B, int m, int n, int method,
                                    const double* dx, const double* dh,
                                    const uint8_t* done_sided, const double* df0,
                                    const double* dF, double* dJ, const int32_t* dmask) {
  if (!ctx) return -1;
  if (B <= 0 || B > 65535) return ctx->bad(2, "B must be in 1..65535");
  if (m <= 0) return ctx->bad(3, "m must be positive");
  if (n <= 0) return ctx->bad(4, "n must be positive");
  if (method != 2 && method != 3) return ctx->bad(5, "method must be 2 or 3");
  if (!dx || !dh || !done_sided || !df0 || !dF) return ctx->bad(6, "input is NULL");
  if (!dJ) return ctx->bad(11, "J is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipError_t e = launch_fd_assemble(B, m, n, method, dx, dh, done_sided, df0, dF, dJ, dmask,
                                    ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_fd_assemble");
  return 0;
}
