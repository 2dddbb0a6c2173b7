// C-ABI entry points (include/blsq.h): the TRF plans — factor / step / fetch — and the row-split (TSQR) plans.
#include "blsq_host.h"

// ================================================================== TRF ====
namespace blsq_host {


int trf_alloc_state(blsq_trf_plan* p) {
  blsq_ctx* ctx = p->ctx;
  const int B = p->B, ld = p->ld;
  const size_t mat = (size_t)ld * ld;
#define ALLOC(buf, bytes)                                               \
  do {                                                                  \
    hipError_t e__ = (buf).alloc(bytes);                                \
    if (e__ != hipSuccess) return ctx->fail(e__, "hipMalloc(" #buf ")"); \
  } while (0)
  ALLOC(p->X, sizeof(double) * B * mat);
  ALLOC(p->vecs, sizeof(double) * (size_t)B * ld * 13);
  ALLOC(p->scal2, sizeof(double) * (size_t)B * 8);
  ALLOC(p->sweeps, sizeof(int) * (size_t)B);
  ALLOC(p->o_vec, sizeof(double) * (size_t)B * ld * 4);
  ALLOC(p->o_hits, sizeof(long long) * (size_t)B * ld);
  ALLOC(p->o_act, sizeof(long long) * (size_t)B * ld);
  ALLOC(p->o_scal, sizeof(double) * (size_t)B * 8);
  ALLOC(p->o_info, sizeof(int) * (size_t)B * 4);
  ALLOC(p->in_scal, sizeof(double) * (size_t)B * 2);
  HIPCHK(ctx, hipMemsetAsync(p->vecs.p, 0, p->vecs.bytes, ctx->stream));
  double* v = p->vecs.as<double>();
  const size_t vs = (size_t)B * ld;
  TrfState& st = p->st;
  st.B = B; st.m = p->m_total; st.n = p->n; st.ld = ld;
  st.X = p->X.as<double>();
  st.x = v; st.lb = v + vs; st.ub = v + 2 * vs; st.scale = v + 3 * vs;
  st.g = v + 4 * vs; st.v = v + 5 * vs; st.d = v + 6 * vs; st.g_h = v + 7 * vs;
  st.diag_h = v + 8 * vs; st.s = v + 9 * vs; st.uf = v + 10 * vs; st.ediag = v + 11 * vs;
  st.scale_in = v + 12 * vs;
  double* sc = p->scal2.as<double>();
  st.srange = sc; st.g_norm = sc + 2 * (size_t)B; st.theta = sc + 3 * (size_t)B;
  double* ov = p->o_vec.as<double>();
  p->out.step_h = ov; p->out.step = ov + vs; p->out.x_new = ov + 2 * vs;
  p->out.p_h_tr = ov + 3 * vs;
  p->out.hits = p->o_hits.as<long long>();
  p->out.active_new = p->o_act.as<long long>();
  p->out.scal = p->o_scal.as<double>();
  p->out.info = p->o_info.as<int>();
  ALLOC(p->lm_sa, sizeof(double) * (size_t)B);
  ALLOC(p->lm_Xa, sizeof(double) * B * mat);
  ALLOC(p->lm_ints, sizeof(int) * ((size_t)B * 9 + 16));
  ALLOC(p->aug_colinfo, sizeof(double) * (size_t)B * 2);
  ALLOC(p->aug_hmax, sizeof(double) * (size_t)B);
  ALLOC(p->aug_lam, sizeof(double) * (size_t)B);
  ALLOC(p->aug_ym, sizeof(double) * (size_t)B);
  ALLOC(p->aug_r1, sizeof(double) * (size_t)B);
  ALLOC(p->aug_open, sizeof(double) * (size_t)B);
  ALLOC(p->aug_mask, sizeof(int) * (size_t)B);
  ALLOC(p->lm_sc, sizeof(double) * (size_t)B * 16);
  ALLOC(p->lm_ph, sizeof(double) * vs);
  HIPCHK(ctx, hipMemsetAsync(p->lm_sa.p, 0, p->lm_sa.bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(p->aug_ym.p, 0, p->aug_ym.bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(p->aug_open.p, 0, p->aug_open.bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(p->lm_Xa.p, 0, p->lm_Xa.bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(p->lm_ints.p, 0, p->lm_ints.bytes, ctx->stream));
  {
    LmState& lm = p->lm;
    lm.B = B; lm.m = p->m_total; lm.n = p->n; lm.ld = ld;
    lm.Raug = p->X.as<double>(); lm.sa = p->lm_sa.as<double>(); lm.Xa = p->lm_Xa.as<double>();
    lm.g_h = p->st.g_h;
    int* ii = p->lm_ints.as<int>();
    lm.fast = ii; lm.ncols_jac = ii + B; lm.ncols_lm = ii + 2 * (size_t)B; lm.st = ii + 3 * (size_t)B;
    lm.active_count = ii + 7 * (size_t)B;
    lm.active_list = ii + 7 * (size_t)B + 16; lm.round = 0;
    lm.sc = p->lm_sc.as<double>(); lm.ph = p->lm_ph.as<double>();
    // Householder-path problems: the SVD-free Newton iteration costs one small stacked QR per
    // iteration.  Measured in round 1 (8192..16384 problems per launch, Delta mix 10/0.5; SVD-free vs
    // Jacobi-SVD step-solves/s): 64x8 5.3M vs 3.4M, 128x16 6.2M vs 2.9M (one panel: the QR is trivial),
    // 256x32 1.55M vs 1.73M, 512x40 0.90M vs 0.90M, 512x48 0.98M vs 0.89M, 512x64 0.76M vs 0.70M.  So for
    // THEM the Jacobi SVD keeps the band 16 < n < 48, where a round of tiny two-panel QRs costs as much
    // as the whole in-LDS SVD.  Normal-equations-path problems have no such band: their rounds are one
    // launch (lm_rounds_reg_kernel) — 256x32: 6.6M vs 2.4M, 512x40: 5.1M vs 1.3M, 128x24: 9.8M vs 2.9M.
    // BLSQ_SVDFREE_MIN_N overrides the upper edge of the band (0: none), BLSQ_NO_SVDFREE=1 forces the SVD.
    const int min_n = ctx->opt.i(OPT_SVDFREE_MIN_N);
    const bool band = p->n > 16 && p->n < min_n;
    p->lm_enable = ctx->opt.i(OPT_NO_SVDFREE) == 1 ? 0 : 1;
    p->lm_gate_mask = p->lm_enable ? (band ? 2 : 3) : 0;
  }
  {
    // CSNE tier: single-rank plans of its shapes with the normal-equations front end on (BLSQ_CSNE = 0: off)
    p->csne_on = p->tree.gram && p->nranks == 1 && csne_supported(p->m, p->n) && ctx->opt.on(OPT_CSNE);
    if (p->csne_on) {
      ALLOC(p->cs_ints, sizeof(int) * (5 * (size_t)B + 8));
      ALLOC(p->cs_pmin, sizeof(double) * (size_t)B);
      ALLOC(p->cs_eta, sizeof(double) * (size_t)B);
      ALLOC(p->cs_k2, sizeof(double) * (size_t)B);
      HIPCHK(ctx, hipMemsetAsync(p->cs_k2.p, 0, p->cs_k2.bytes, ctx->stream));
      ALLOC(p->cs_alpha, sizeof(double) * (size_t)B * CSNE_MAXE);
      ALLOC(p->cs_hp, sizeof(double) * vs);
      HIPCHK(ctx, hipMemsetAsync(p->cs_ints.p, 0, p->cs_ints.bytes, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(p->cs_pmin.p, 0, p->cs_pmin.bytes, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(p->cs_eta.p, 0, p->cs_eta.bytes, ctx->stream));
      CsneState& cs = p->cs;
      cs.B = B; cs.m = p->m; cs.n = p->n; cs.ld = ld;
      int* ii = p->cs_ints.as<int>();
      cs.flag = ii; cs.list = ii + B; cs.fail_list = ii + 2 * (size_t)B; cs.ne = ii + 3 * (size_t)B;
      cs.counts = ii + 5 * (size_t)B;                     // (sel_mask: ii + 4 B; scratch counter: counts + 4)
      cs.ralpha = p->cs_alpha.as<double>(); cs.hp = p->cs_hp.as<double>(); cs.eta = p->cs_eta.as<double>();
      csne_geometry(p->m, &cs.rows_per_wg, &cs.nchunk);
      cs.NE = 1;
      p->st.csne = cs.flag; p->st.csne_hp = cs.hp;
      p->lm.csne = cs.flag; p->lm.csne_ne = cs.ne; p->lm.csne_alpha = cs.ralpha;
    }
  }
  p->aug_RP = std::max(aug_rows(p->n), ld);
  if (aug_rows(p->n) > RMAX) return ctx->bad(4, "n too large for the augmented system (n <= 512)");
  p->aug_LDP = 0;
  return 0;
#undef ALLOC
}

// ---- after the front end --------------------------------------------------------------------------
// Two ways into the n-space path:
//   trf_after_triangle   a triangle [R c] of [J f] is given for every problem (front end off, TSQR
//                        merge): prep from R, stacked QR of [R D; E]
//   trf_gram_stage ...   the normal-equations path: prep from the Gram, H = D G D + E^2 factored by
//                        Cholesky, and the conditioning gate applied to THAT factor — the system the
//                        step is solved from.  No triangle of J is ever formed for such a problem;
//                        a problem the gate rejects is factored by the Householder tree and prepared
//                        again from its triangle (trf_fallback_stage).
int trf_finish(blsq_trf_plan* p) {
  blsq_ctx* ctx = p->ctx;
  hipError_t e;
  if (p->use_qr || p->njac != 0 || !p->gate_done) p->x_dirty = true;   // (a stacked QR or a Jacobi launch may follow)
  if (p->use_qr) {
    // E = 0 (unbounded problems): [R D | c] is the triangle already — written by a copy, masked out of the QR
    ctx->begin(K_QR_AUG);
    e = launch_trf_aug_trivial(p->st, p->path, p->aug_mask.as<int>(), nullptr, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_trf_aug_trivial");
    QrArgs q = p->tree.base_args();
    q.ncols_dev = p->aug_mask.as<int>();
    // source = R read in place, columns scaled by d on the fly, on top of the VIRTUAL block
    // E = diag(ediag): [R D | c ; E | 0] is never written to memory
    q.A = p->st.Rt; q.strideA = (long)p->ld * p->ld; q.ldA = p->ld;
    q.rowsA = aug_block_rows(p->n) + p->n;
    q.vdiag_row0 = aug_block_rows(p->n); q.vdiag_vec = p->st.ediag;
    q.colscale = p->st.d; q.stride_vec = p->ld;
    q.F = nullptr; q.strideF = 0;
    q.rows_per_leaf = p->aug_RP; q.RP = p->aug_RP; q.LDP = p->aug_LDP;
    q.Rout = p->st.X;
    q.stack_rows = aug_block_rows(p->n); // [R D; E]: two upper-triangular blocks
    ctx->begin(K_QR_AUG);
    e = launch_qr(q, 1, p->B, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_qr(aug)");
  }
  // rank gate: clearly full-rank problems skip the SVD (lm_kernels.hip)
  if (!p->gate_done) {
    p->lm.jac_count = nullptr;
    ctx->begin(K_LM_GATE);
    e = launch_lm_gate(p->lm, p->lm_gate_mask, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_lm_gate");
    p->njac = -1;
  }
  p->gate_done = false;
  if (p->njac == 0) return 0;             // nobody needs the SVD: no launch
  JacobiArgs ja{};
  ja.X = p->st.X; ja.strideX = (long)p->ld * p->ld; ja.ld = p->ld; ja.ncols_dev = p->lm.ncols_jac;
  ja.N = p->n + 1; ja.s = p->st.s; ja.uf = p->st.uf; ja.srange = p->st.srange;
  ja.sweeps = p->sweeps.as<int>(); ja.max_sweeps = 40;
  ctx->begin(K_JACOBI);
  e = launch_jacobi(ja, p->B, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_jacobi");
  return 0;
}

// redo: the problems were prepared from their Gram already in this factor call — start again from the
// scale the caller passed (a 'jac' scaling update is applied once, to the caller's vector)
int trf_after_triangle(blsq_trf_plan* p, const double* Rt, int scale_mode, int redo = 0) {
  blsq_ctx* ctx = p->ctx;
  p->st.Rt = Rt; p->st.Gk = nullptr; p->st.path = nullptr;
  p->path = nullptr; p->use_chol = false; p->use_qr = true;
  p->lm.path = nullptr; p->lm.colinfo = nullptr; p->lm.hmax = nullptr; p->lm.k2 = nullptr; p->gram_valid = false;
  p->tree.path_valid = false; p->tree.any_gram = false; p->tree.any_qr = true;
  p->gate_done = false;
  ctx->begin(K_PREP);
  hipError_t e = launch_trf_prep(p->st, scale_mode, 0, nullptr, redo, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_trf_prep");
  return trf_finish(p);
}

GramCholArgs trf_chol_args(blsq_trf_plan* p, const int* mask) {
  QrTree& t = p->tree;
  GramCholArgs c{};
  c.opt = &p->ctx->opt;
  c.Gsrc = t.gram_keep.as<double>(); c.G = p->st.X; c.NPAD = p->ld; c.n = p->n;
  c.colscale = p->st.d; c.diag_vec = p->st.ediag; c.stride_vec = p->ld;
  c.mask = mask; c.fb_mask = t.fb_mask(); c.fail_count = t.fb_count(); c.path_out = t.path_rw();
  c.fail_list = t.fb_list();
  c.dsc = t.gram_dsc.as<double>();
  c.rinv = t.gram_rinv.as<double>(); c.ywork = t.gram_ywork.as<double>(); c.k2_out = t.gram_k2.as<double>();
  c.cert_done = t.gram_cert.as<int>();
  c.k2_max = t.k2_max; c.pivot_floor = 1.0 / t.k2_max;
  c.cert_flag = t.gram_cflag.as<int>(); c.cert_tau = t.gram_ctau.as<double>();
  c.colinfo = p->aug_colinfo.as<double>();
  c.hmax = p->aug_hmax.as<double>(); c.lam_out = p->aug_lam.as<double>();
  if (p->csne_on) c.pmin_out = p->cs_pmin.as<double>();
  if (p->ld > 80) {
    c.cert_ym = p->aug_ym.as<double>(); c.cert_r1 = p->aug_r1.as<double>();
    // (cert_direct = 0: every open problem through the norm stage, the explicit inverse)
    c.cert_open = p->ctx->opt.on(OPT_CERT_DIRECT) ? p->aug_open.as<double>() : nullptr;
  }
  // (N <= 80: the register-resident factor kernel also does the rank gate's sure case; N > 80: stage 0 of the certificate
  //  does — gram_cert0_kernel — for the problems it certifies.  `unsettled` counts the others.)
  c.lmfin.fast = p->lm.fast; c.lmfin.ncols_jac = p->lm.ncols_jac; c.lmfin.sc = p->lm.sc; c.lmfin.st = p->lm.st;
  c.unsettled = t.fb_count() + 2;
  c.lmfin.m = p->lm.m; c.lmfin.enable = (p->lm_gate_mask >> 1) & 1;
  return c;
}

// the second half of the certificate + the rank gate of the trust-region solver (counters:
// fb_count()[0] problems that leave the path, [1] problems for the SVD)
int trf_gate_tail(blsq_trf_plan* p, const GramCholArgs& c, bool full = true) {
  blsq_ctx* ctx = p->ctx;
  QrTree& t = p->tree;
  ctx->begin(K_GRAM_GATE);
  hipError_t e = launch_gram_gate(c, p->B, ctx->stream);
  if (e == hipSuccess) e = launch_gram_cert_shift(c, p->B, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_gram_gate");
  // The rank gate runs BEFORE the verdict is read back: in the common case (no problem leaves this
  // path) its result stands, and the same read-back tells whether anybody needs the Jacobi SVD at all.
  p->lm.path = t.path_rw();
  p->lm.colinfo = p->aug_colinfo.as<double>();
  p->lm.jac_count = t.fb_count() + 1;
  // (normal-equations-path problems only: a problem the certificate has just rejected gets its triangle first and
  //  is gated by trf_finish afterwards — estimating the rank of its abandoned factor here cost the latency of one
  //  problem's inverse iteration for nothing; such a problem counts as "needs the SVD" until then, which the verdict
  //  logic ignores whenever a problem left the path)
  ctx->begin(K_LM_GATE);
  //  (`full`: every problem is refreshed by this call — a masked call keeps the others' state as it is)
  e = launch_lm_gate(p->lm, full ? (p->lm_gate_mask & 2) : p->lm_gate_mask, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_lm_gate");
  return 0;
}

// call that must go to the Householder tree (their indices are flagged in tree.fb_mask()).
int trf_gram_stage(blsq_trf_plan* p, int scale_mode, const int* mask, int* nfb, bool defer = false) {
  blsq_ctx* ctx = p->ctx;
  QrTree& t = p->tree;
  if (!t.fb_zeroed) HIPCHK(ctx, hipMemsetAsync(t.fb_count(), 0, 3 * sizeof(int), ctx->stream));
  t.fb_zeroed = false;
  p->st.Rt = t.Rfinal(); p->st.Gk = t.gram_keep.as<double>(); p->st.path = t.path_rw();
  const PackVecs* pk = nullptr;
  { int rc_ = take_pack(p, mask, &pk); if (rc_) return rc_; }
  ctx->begin(K_PREP);
  hipError_t e = launch_trf_prep(p->st, scale_mode, 1, mask, 0, ctx->stream, pk);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_trf_prep(gram)");
  GramCholArgs c = trf_chol_args(p, mask);
  c.skip_zero = p->x_dirty ? 0 : 1;
  if (!mask) p->x_dirty = false;                        // (every slot is rewritten, zeros included, by this launch)
  ctx->begin(K_AUG_CHOL);
  e = launch_gram_chol(c, p->B, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_gram_chol(aug)");
  // second guess (N <= 80): the Cholesky kernel settles EVERY problem itself — certificate by its first
  // bound, rank gate by the column-norm bound — as it did in the last call: then the certificate and gate
  // launches would both be empty and are not enqueued (trf_resolve checks the settled counter)
  // N > 80: stage 0 of the certificate is still launched (it IS what settles a problem there) — the norm stage, the
  // shifted factorisation and the rank gate, three launches that would find nothing to do, are not.
  bool skip_tail = defer && p->guess_settled && c.lmfin.fast != nullptr;
  if (skip_tail && p->ld > 80) {
    // (settle0 = 0: the whole gate tail for N > 80, as before)
    if (!c.cert_ym || !ctx->opt.on(OPT_CERT0) || !ctx->opt.on(OPT_SETTLE0)) skip_tail = false;
  }
  int rc;
  if (skip_tail) {
    p->lm.path = t.path_rw();
    p->lm.colinfo = p->aug_colinfo.as<double>();
    if (p->ld > 80) {
      ctx->begin(K_GRAM_GATE);
      e = launch_gram_gate(c, p->B, ctx->stream, /*stage0_only=*/true);
      ctx->end();
      if (e != hipSuccess) return ctx->fail(e, "launch_gram_gate(stage 0)");
    }
  } else if ((rc = trf_gate_tail(p, c, mask == nullptr))) return rc;
  if (defer) {                              // the counters travel; the verdict is read by trf_resolve
    if ((rc = verdict_arm(p, skip_tail, p->pend_dJ, p->pend_df, p->pend_ldJ, p->pend_scale_mode))) return rc;
    *nfb = 0;
  } else {
    // (the synchronous verdict: published and polled for — slot [4 .. 8) of the pinned ints — instead of a blit and a
    //  stream synchronisation)
    int seq = 0;
    HIPCHK(ctx, ctx->publish(t.fb_count(), 3, ctx->pinned + 4, ctx->lm_ev[0], &seq));
    HIPCHK(ctx, ctx->await(ctx->pinned + 4, ctx->lm_ev[0], seq));
    *nfb = ctx->pinned[4];
    p->gate_done = (*nfb == 0);
    p->njac = p->gate_done ? ctx->pinned[5] : -1;
    if (!mask) p->guess_settled = (c.unsettled && ctx->pinned[6] == 0);
  }
  t.note_paths(ctx, *nfb, mask != nullptr);
  p->path = t.path_rw();
  p->use_chol = t.any_gram;
  p->use_qr = t.any_qr;
  p->lm.path = p->path;
  // Newton systems of Householder-path problems from the Gram where alpha makes them provably well
  // conditioned (LmState::hmax; BLSQ_LM_CHOL_QRPATH = 0: always the stacked QR)
  {
    const bool on = ctx->opt.on(OPT_LM_CHOL_QRPATH);
    p->lm.hmax = on ? p->aug_hmax.as<double>() : nullptr;
    p->lm.lam = p->aug_lam.as<double>();
    p->lm.k2_max = t.k2_max;
    p->lm.k2 = t.gram_k2.as<double>();
    p->lm.colinfo = p->aug_colinfo.as<double>();         // (written for every problem the Cholesky kernel looked at)
    p->gram_valid = true;
  }
  return 0;
}

// the problems the gate rejected: Householder tree on [J f], prep again from the triangle
// (a masked factor call has refreshed some problems: the tier's list is rebuilt from the flags)
int trf_csne_relist(blsq_trf_plan* p) {
  blsq_ctx* ctx = p->ctx;
  hipError_t e = launch_csne_reroute(p->cs, -1, nullptr, nullptr, nullptr, ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_csne_reroute(relist)");
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned + 9, p->cs.counts, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  p->ncsne = ctx->pinned[9];
  return 0;
}

// CSNE tier, factor side: which of the nfb problems the certificate has just rejected (tree.fb_list()) keep their
// Gram-Cholesky factor as a preconditioner and have their steps corrected against J (csne_kernels.hip).  The bound on
// kappa_2 of the COMPUTED augmented system comes from the certificate's norm stage run with CSNE_K2_MAX as its gate
// (explicit inverse, as the CholeskyQR2 tier does for the plain system); *ntree = the problems left for the other tiers.
int trf_csne_select(blsq_trf_plan* p, const double* dJ, const double* df, int ldJ, int nfb, int* ntree, bool masked) {
  blsq_ctx* ctx = p->ctx;
  QrTree& t = p->tree;
  const int B = p->B;
  *ntree = nfb;
  if (!p->csne_on || !p->lm_enable) return 0;
  hipError_t e = hipSuccess;
  if (!p->cs_vec.p) {                                     // first use: the recordings (52 KB per problem at n = 256)
    e = p->cs_vec.alloc(sizeof(double) * (size_t)B * CSNE_MAXE * 3 * p->ld);
    if (e != hipSuccess) { (void)hipGetLastError(); p->csne_on = false; return 0; }   // (no room: the other tiers)
    p->cs.rvec = p->cs_vec.as<double>();
    p->lm.csne_vec = p->cs.rvec;
  }
  int* sel = p->cs_ints.as<int>() + 4 * (size_t)B;
  int* scratch = p->cs.counts + 4;
  HIPCHK(ctx, hipMemsetAsync(sel, 0, sizeof(int) * (size_t)B, ctx->stream));
  GramCholArgs cy = trf_chol_args(p, t.fb_mask());        // (mask: the rejected problems only)
  cy.fb_mask = sel; cy.fail_count = scratch; cy.fail_list = nullptr; cy.path_out = nullptr;
  cy.cert_done = nullptr; cy.cert_flag = nullptr; cy.cert_tau = nullptr; cy.cert_open = nullptr;
  cy.cert_ym = nullptr; cy.cert_r1 = nullptr; cy.unsettled = nullptr; cy.lmfin = GramCholArgs::LmFinish{};
  cy.lam_out = nullptr; cy.hmax = nullptr; cy.colinfo = nullptr; cy.pmin_out = nullptr;
  cy.k2_max = CSNE_K2_MAX; cy.k2_out = p->cs_k2.as<double>();
  ctx->begin(K_GRAM_GATE);
  e = launch_gram_gate(cy, B, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_gram_gate(csne bound)");
  e = launch_csne_select(p->cs, p->lm, nfb, t.fb_list(), t.fb_mask(), t.fb_count(), t.path_rw(), sel,
                         p->cs_k2.as<double>(), p->cs_pmin.as<double>(), p->aug_colinfo.as<double>(), ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_csne_select");
  // two counters to the host: the problems left for the tree, the problems on the tier
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned + 8, t.fb_count(), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned + 9, p->cs.counts, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *ntree = ctx->pinned[8];
  p->ncsne = ctx->pinned[9];
  ctx->csne_routed += (unsigned long long)(nfb - *ntree);
  p->cs.J = dJ; p->cs.strideJ = (long)p->m * ldJ; p->cs.ldJ = ldJ; p->cs.F = df; p->cs.strideF = p->m;
  if (!masked) t.any_qr = *ntree > 0;                     // (a masked call keeps the others' paths: any_qr stays)
  t.any_gram = t.any_gram || *ntree < nfb;
  p->use_chol = t.any_gram; p->use_qr = t.any_qr;
  return 0;
}

int trf_fallback_stage(blsq_trf_plan* p, const double* dJ, const double* df, int ldJ, int scale_mode,
                       int nfb, bool masked = false) {
  blsq_ctx* ctx = p->ctx;
  int rc;
  {
    int ntree = nfb;
    if ((rc = trf_csne_select(p, dJ, df, ldJ, nfb, &ntree, masked))) return rc;
    nfb = ntree;
    if (nfb == 0) return 0;
  }
  rc = p->tree.run_fallback(ctx, dJ, df, ldJ, nfb);
  if (rc) return rc;
  ctx->begin(K_PREP);
  hipError_t e = launch_trf_prep(p->st, scale_mode, 0, p->tree.fb_mask(), 1, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_trf_prep(redo)");
  return 0;
}

// the whole factor call from device-resident [J f] (mask: outer driver, fresh Jacobians only)
int trf_factor_core(blsq_trf_plan* p, const double* dJ, const double* df, int ldJ, int scale_mode,
                    const int* mask, bool may_defer, bool gram_done) {
  blsq_ctx* ctx = p->ctx;
  int rc;
  if ((rc = verdict_drop(p))) return rc;
  if (!p->tree.gram) {
    if ((rc = p->tree.run_levels(ctx, dJ, df, ldJ, mask))) return rc;
    return trf_after_triangle(p, p->tree.Rfinal(), scale_mode);
  }
  if (!gram_done && (rc = p->tree.run_gram_only(ctx, dJ, df, ldJ, mask, false))) return rc;
  p->last_scale_mode = scale_mode;
  if (!mask) p->ncsne = 0;                                // (the prep launch clears every flag; trf_csne_select sets them anew)
  int nfb = 0;
  // (never in the n-band that always takes the SVD, nor right after a wrong guess)
  const bool defer = may_defer && !mask && verdict_may_guess(p) && p->lm_enable && p->pend_pin &&
                     p->pend_ev;
  if ((rc = trf_gram_stage(p, scale_mode, mask, &nfb, defer))) return rc;
  if (defer) { p->pend_dJ = dJ; p->pend_df = df; p->pend_ldJ = ldJ; p->pend_scale_mode = scale_mode; }
  else if (!mask) p->guess_ok = (nfb == 0 && p->njac == 0);
  if (nfb > 0 && (rc = trf_fallback_stage(p, dJ, df, ldJ, scale_mode, nfb, mask != nullptr))) return rc;
  if (nfb == 0 && mask && p->ncsne > 0 && (rc = trf_csne_relist(p))) return rc;   // (refreshed problems have left the tier)
  return trf_finish(p);
}

// The verdict of an optimistic factor call.  *redo = false: nothing was pending, or the guess held.
// *redo = true: it did not — the state is now what the synchronous path would have left (fallback
// stage, rank gate, SVD), and whatever was computed from the guessed state must be computed again.
int trf_resolve(blsq_trf_plan* p, bool* redo) {
  return verdict_resolve(
      p, redo, [&]() { return trf_gate_tail(p, trf_chol_args(p, nullptr)); },
      [&](int nfb) {
        QrTree& t = p->tree;
        p->use_chol = t.any_gram;
        p->use_qr = t.any_qr;
        p->lm.colinfo = p->aug_colinfo.as<double>();
        int rc;
        if (nfb > 0 && (rc = trf_fallback_stage(p, p->pend_dJ, p->pend_df, p->pend_ldJ, p->pend_scale_mode, nfb)))
          return rc;
        return trf_finish(p);
      });
}

// Safeguarded Newton iteration of the SVD-free problems: lock-step rounds of
// (factor of the system at the current alpha) + (two triangular solves + update).
//
// The kernels of round r run over the compacted list of the problems still iterating and leave
// when their index is beyond the DEVICE counter of that round, so the host does not have to know
// the count to launch them — only an upper bound (the previous round's count).  When every problem
// is on the normal-equations path the host therefore runs one round AHEAD of what it knows: it
// enqueues round r, then waits for the counter of round r (written by round r - 1, i.e. while
// round r executes).  The GPU does not idle on a host round trip for the rounds that had work in the
// plan's last call; from the first round that was empty then, the counter is read before the round is
// enqueued (no round of empty launches at the end).  Problems on the Householder path (stacked QR per
// round: several launches sized by the count) keep the synchronous loop.
int trf_lm_rounds(blsq_trf_plan* p, const double* dDelta, const double* dalpha_in) {
  blsq_ctx* ctx = p->ctx;
  int* counts = p->lm.active_count;
  hipError_t e;
  p->lm.fused_gram = 0;
  if (p->use_chol && p->lm_enable && p->ld <= 80) {
    // N <= 80: the Gauss-Newton step, the bracket and ALL rounds of every normal-equations-path problem
    // in ONE launch (one wave per problem iterates to the end; chol_kernels.hip).  Householder-path
    // problems of the same batch go through lm_start and the round loop below.
    // BLSQ_LM_FUSED = 0: lm_start + the round-by-round loop for everybody.
    if (ctx->opt.on(OPT_LM_FUSED)) {
      GramCholArgs c{};
      c.opt = &ctx->opt;
      c.Gsrc = p->tree.gram_keep.as<double>(); c.NPAD = p->ld; c.n = p->n;
      c.colscale = p->st.d; c.diag_vec = p->st.ediag; c.stride_vec = p->ld;
      c.rinv = p->tree.gram_rinv.as<double>(); c.dsc = p->tree.gram_dsc.as<double>();
      ctx->begin(K_LM_CHOL);
      e = launch_lm_rounds_reg(c, p->lm, dDelta, dalpha_in, ctx->stream);
      ctx->end();
      if (e != hipSuccess) return ctx->fail(e, "launch_lm_rounds_reg");
      p->lm_rounds_done = 0;
      if (!p->use_qr) return 0;
      p->lm.fused_gram = 1;
    }
  }
  int* pin = ctx->pinned + 32;                           // slot of round r: pin + 4 r
  int pin_seq[16] = {0};
  bool rides[16] = {false};
  int ride_rounds = 0;                                   // rounds [0, ride_rounds) are enqueued before their counter is read
  if (!p->lm_counts_clean) HIPCHK(ctx, hipMemsetAsync(counts, 0, 16 * sizeof(int), ctx->stream));
  p->lm_counts_clean = false;
  ctx->begin(K_LM_SOLVE);
  e = launch_lm_start(p->lm, dDelta, dalpha_in, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_lm_start");
  auto read_back = [&](int r) -> hipError_t {            // counter of round r -> pin[r], event r & 1
    // (a round that is enqueued ahead of its counter takes the counter along: lm_update_kernel of that round stores it)
    if (ride_rounds > r && ctx->pub_direct() && ctx->pub_ride()) { rides[r] = true; return hipSuccess; }
    return ctx->publish(counts + r, 1, pin + 4 * r, ctx->lm_ev[r & 1], &pin_seq[r]);
  };
  auto landed = [&](int r) -> hipError_t { return ctx->await(pin + 4 * r, ctx->lm_ev[r & 1], pin_seq[r]); };
  auto chol_round = [&](int round, int grid, int expect, const int* count_dev) -> hipError_t {
    // R_alpha = chol(D G D + E^2 + alpha I) straight from the Gram, active Gram-path problems
    GramCholArgs c{};
    c.opt = &ctx->opt;
    c.Gsrc = p->tree.gram_keep.as<double>(); c.G = p->lm.Xa; c.NPAD = p->ld; c.n = p->n;
    c.colscale = p->st.d; c.diag_vec = p->st.ediag; c.diag_sqrt = p->lm.sa; c.stride_vec = p->ld;
    c.batch_list = p->lm.active_list + (size_t)(round & 1) * p->B;
    c.skip_path = p->path;
    c.qr_mask = p->lm.hmax ? p->lm.ncols_lm : nullptr;
    c.count_dev = count_dev; c.expect = expect;
    c.skip_zero = 1;                                    // (lm_Xa: zeroed at allocation, read by lm_update's solves only)
    ctx->begin(K_LM_CHOL);
    hipError_t ee = launch_gram_chol(c, grid, ctx->stream);
    ctx->end();
    return ee;
  };
  // the stacked QR of [R_aug; sqrt(alpha) I] for the problems of the round whose mask says so (LmState::ncols_lm)
  auto qr_round = [&](int round, int grid, const int* count_dev) -> hipError_t {
    // source = [R_aug | c_aug] read in place, stacked on a VIRTUAL sqrt(alpha) I block
    QrArgs q = p->tree.base_args();
    q.A = p->lm.Raug; q.strideA = (long)p->ld * p->ld; q.ldA = p->ld;
    q.rowsA = aug_block_rows(p->n) + p->n;
    q.vdiag_row0 = aug_block_rows(p->n); q.vdiag = p->lm.sa;
    q.F = nullptr; q.strideF = 0; q.ncols_dev = p->lm.ncols_lm;
    q.batch_list = p->lm.active_list + (size_t)(round & 1) * p->B;   // only the active problems
    q.count_dev = count_dev;
    q.rows_per_leaf = p->aug_RP; q.RP = p->aug_RP; q.LDP = p->aug_LDP;
    q.Rout = p->lm.Xa;
    q.stack_rows = aug_block_rows(p->n);
    ctx->begin(K_LM_QR);
    hipError_t ee = launch_qr(q, 1, grid, ctx->stream);
    ctx->end();
    return ee;
  };
  const bool chol_any = p->gram_valid && p->lm_enable && (p->use_chol || p->lm.hmax != nullptr) &&
                        !(p->lm.fused_gram && !p->lm.hmax);
  if (chol_any) ride_rounds = p->lm_rounds_last < 12 ? p->lm_rounds_last : 12;
  HIPCHK(ctx, read_back(0));
  // Run-ahead loop: whenever the Grams of the current problems are at hand.  Householder-path problems join
  // the Cholesky launch where their alpha allows it (LmState::hmax); the stacked QR of the round is enqueued
  // only while the batch holds such problems at all, over the same upper bound, and leaves at once for a
  // problem whose mask is 0.  Their triangles dirty the lm_Xa slots outside the factor: the solves never look.
  if (chol_any) {
    int bound = p->B;                                   // upper bound of the count of the round being enqueued
    int expect = p->lm_expect0 > 0 ? p->lm_expect0 : p->B;   // (kernel choice only: last call's first count)
    // Rounds that had work in the LAST call of this plan are enqueued ahead of their counter, as described
    // above; from the first round that was empty last time on, the host looks at the counter first — the
    // GPU idles for one host round trip (~10 us) instead of running a round of three empty launches.
    const int ahead_rounds = p->lm_rounds_last;
    int done_rounds = 0;
    for (int round = 0; round < 12; ++round) {
      const bool ahead = round < ahead_rounds;
      if (!ahead) {
        HIPCHK(ctx, landed(round));
        const int active = pin[4 * round];
        if (round == 0) p->lm_expect0 = active > 0 ? active : -1;
        if (active == 0) break;
        bound = expect = active;
      }
      e = chol_round(round, bound, expect, counts + round);
      if (e != hipSuccess) return ctx->fail(e, "launch_gram_chol(lm)");
      if (p->use_qr) {
        e = qr_round(round, bound, counts + round);
        if (e != hipSuccess) return ctx->fail(e, "launch_qr(lm)");
      }
      ctx->begin(K_LM_SOLVE);
      p->lm.round = round;
      if (rides[round]) {
        pin_seq[round] = ++ctx->pub_seq;
        p->lm.pub = PublishArgs{counts + round, 1, pin + 4 * round, pin_seq[round]};
      }
      e = launch_lm_update(p->lm, bound, ctx->stream);
      p->lm.pub = PublishArgs{nullptr, 0, nullptr, 0};
      ctx->end();
      if (e != hipSuccess) return ctx->fail(e, "launch_lm_update");
      HIPCHK(ctx, read_back(round + 1));
      if (ahead) {
        HIPCHK(ctx, landed(round));
        const int active = pin[4 * round];                  // what round `round` really worked on
        if (round == 0) p->lm_expect0 = active > 0 ? active : -1;
        if (active == 0) break;                         // (the round just enqueued is empty)
        bound = expect = active;
      }
      done_rounds = round + 1;
    }
    p->lm_rounds_last = done_rounds;
    p->lm_rounds_done = done_rounds;
    return 0;
  }
  HIPCHK(ctx, landed(0));
  int active = pin[0];
  p->lm_rounds_done = 0;
  for (int round = 0; round < 12 && active > 0; ++round) {
    p->lm_rounds_done = round + 1;
    if (p->use_chol && !p->lm.fused_gram) {
      e = chol_round(round, active, active, nullptr);
      if (e != hipSuccess) return ctx->fail(e, "launch_gram_chol(lm)");
    }
    if (p->use_qr) {
      e = qr_round(round, active, nullptr);
      if (e != hipSuccess) return ctx->fail(e, "launch_qr(lm)");
    }
    ctx->begin(K_LM_SOLVE);
    p->lm.round = round;
    e = launch_lm_update(p->lm, active, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_lm_update");
    HIPCHK(ctx, read_back(round + 1));
    HIPCHK(ctx, landed(round + 1));
    active = pin[4 * (round + 1)];
  }
  return 0;
}

// CSNE tier, step side (csne_kernels.hip): ONE streaming pass over the caller's J for every recorded evaluation of
// every problem on the tier, then the n-space correction (replayed Newton iteration, corrected final step, H p for
// the step kernel).  Problems whose acceptance fails are listed in cs.fail_list (trf_csne_verdict).
int trf_csne_correct(blsq_trf_plan* p, const double* dDelta, const double* dalpha_in) {
  blsq_ctx* ctx = p->ctx;
  CsneState& cs = p->cs;
  const int ne_max = std::min(CSNE_MAXE, 1 + std::max(0, p->lm_rounds_done));
  const bool mfma = ctx->opt.on(OPT_CSNE_MFMA);           // (all eight evaluation slots; the sums do not depend on the depth)
  const int NE = mfma ? CSNE_MAXE : csne_launch_evals(ne_max);   // (else the launch's split over the waves: NEH x G >= ne_max)
  cs.NE = NE;
  const size_t need = (size_t)p->ncsne * cs.nchunk * ((size_t)NE * p->ld + 16);
  if (need > p->cs_part_cap) {                            // (grows geometrically; hipFree waits for the stream)
    p->cs_part.release();
    const size_t cap = std::max(need, 2 * p->cs_part_cap);
    hipError_t ae = p->cs_part.alloc(sizeof(double) * cap);
    if (ae != hipSuccess) { p->cs_part_cap = 0; return ctx->fail(ae, "hipMalloc(CSNE partial sums)"); }
    p->cs_part_cap = cap;
    cs.part = p->cs_part.as<double>();
  }
  HIPCHK(ctx, hipMemsetAsync(cs.counts + 1, 0, sizeof(int), ctx->stream));
  ctx->begin(K_CSNE_PASS);
  hipError_t e = mfma ? launch_csne_pass_mfma(cs, p->st.d, p->ncsne, ctx->stream)
                      : launch_csne_pass(cs, p->st.d, p->ncsne, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_csne_pass");
  ctx->begin(K_CSNE_FIX);
  e = launch_csne_fix(cs, p->st, p->lm, dDelta, dalpha_in, p->ncsne, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_csne_fix");
  return 0;
}

// ... and what became of them: problems the tier declined in this step call leave it — factored by CholeskyQR2 / the
// Householder tree from the caller's J (still valid: the lifetime rule of the tier), prepared again from the triangle —
// and the step runs once more (*redo).
int trf_csne_verdict(blsq_trf_plan* p, int ncs, bool* redo) {
  blsq_ctx* ctx = p->ctx;
  CsneState& cs = p->cs;
  QrTree& t = p->tree;
  int seq = 0;
  HIPCHK(ctx, ctx->publish(cs.counts + 1, 1, ctx->pinned + 12, ctx->lm_ev[0], &seq));
  HIPCHK(ctx, ctx->await(ctx->pinned + 12, ctx->lm_ev[0], seq));
  const int nfail = ctx->pinned[12];
  ctx->csne_steps += (unsigned long long)(ncs - nfail);
  ctx->csne_declined += (unsigned long long)nfail;
  if (nfail == 0) return 0;
  *redo = true;
  hipError_t e = launch_csne_reroute(cs, nfail, t.fb_list(), t.fb_mask(), t.path_rw(), ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "launch_csne_reroute");
  p->ncsne = ncs - nfail;
  int rc = t.run_fallback(ctx, cs.J, cs.F, cs.ldJ, nfail);
  if (rc) return rc;
  ctx->begin(K_PREP);
  e = launch_trf_prep(p->st, p->last_scale_mode, 0, t.fb_mask(), 1, ctx->stream);
  ctx->end();
  if (e != hipSuccess) return ctx->fail(e, "launch_trf_prep(csne redo)");
  t.any_qr = true; p->use_qr = true;
  p->gate_done = false; p->njac = -1;
  return trf_finish(p);
}

}  // namespace blsq_host

extern "C" int blsq_trf_plan_create(blsq_ctx* ctx, int B, int m, int n, blsq_trf_plan** out) {
  if (!ctx) return -1;
  if (!out) return ctx->bad(5, "out is NULL");
  *out = nullptr;
  if (B <= 0) return ctx->bad(2, "B must be positive");
  if (m <= 0) return ctx->bad(3, "m must be positive");
  if (n <= 0) return ctx->bad(4, "n must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  blsq_trf_plan* p = new blsq_trf_plan();
  p->ctx = ctx; p->B = B; p->m = m; p->n = n; p->m_total = m; p->nranks = 1;
  const int aug_rp = std::max(aug_rows(n), round_up(n + 1, 16));
  int rc = p->tree.build(ctx, B, m, n, (size_t)B * aug_rp);
  if (rc == 0) { p->ld = p->tree.NPAD; rc = trf_alloc_state(p); }
  if (rc == 0) {
    p->optimistic = ctx->opt.on(OPT_OPTIMISTIC);
    hipError_t e = hipHostMalloc((void**)&p->pend_pin, 4 * sizeof(int), hipHostMallocCoherent);
    if (e == hipSuccess) memset(p->pend_pin, 0, 4 * sizeof(int));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->pend_ev, hipEventDisableTiming);
    if (e != hipSuccess) rc = ctx->fail(e, "optimistic-verdict resources");
  }
  if (rc != 0) { blsq_trf_plan_destroy(p); return rc; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->trf_plans.push_back(p);
  *out = p;
  return 0;
}

extern "C" int blsq_trf_plan_destroy(blsq_trf_plan* p) {
  if (!p) return -1;
  hipStreamSynchronize(p->ctx->stream);
  { auto& v = p->ctx->trf_plans; v.erase(std::remove(v.begin(), v.end(), p), v.end()); }
  if (p->pend_pin) hipHostFree(p->pend_pin);
  if (p->pend_ev) hipEventDestroy(p->pend_ev);
  p->tree.release(); p->Rcomb.release(); p->Rstack.release();
  p->X.release(); p->vecs.release(); p->scal2.release(); p->sweeps.release();
  p->o_vec.release(); p->o_hits.release(); p->o_act.release(); p->o_scal.release();
  p->o_info.release(); p->in_J.release(); p->in_f.release(); p->in_vec.release();
  p->in_scal.release();
  p->lm_sa.release(); p->lm_Xa.release(); p->lm_ints.release(); p->lm_sc.release();
  p->cs_k2.release(); p->cs_ints.release(); p->cs_pmin.release(); p->cs_eta.release(); p->cs_alpha.release(); p->cs_hp.release();
  p->cs_vec.release(); p->cs_part.release();
  p->lm_ph.release(); p->aug_colinfo.release(); p->aug_hmax.release(); p->aug_lam.release(); p->aug_ym.release(); p->aug_r1.release(); p->aug_open.release(); p->aug_mask.release();
  delete p;
  return 0;
}

static int trf_put_bounds(blsq_trf_plan* p, const double* x, const double* lb, const double* ub,
                          const double* scale, hipMemcpyKind kind, bool zero_counts = false) {
  blsq_ctx* ctx = p->ctx;
  int rc;
  if (kind == hipMemcpyDeviceToDevice) {                // one launch instead of four strided copies
    // (the two gate counters of the factor call that follows are cleared by the same launch)
    PackVecs pv{{x, lb, ub, scale, nullptr}, {p->st.x, p->st.lb, p->st.ub, p->st.scale, nullptr},
                (zero_counts && p->tree.gram) ? p->tree.fb_count() : nullptr, 3};
    p->pack_pend = false;
    if (zero_counts && p->tree.gram && ctx->fuse_pack()) {   // (the Gram stage's prep launch does it: trf_gram_stage)
      p->pack_pv = pv; p->pack_pend = true;
      p->tree.fb_zeroed = true;
      return 0;
    }
    hipError_t e = launch_pack_vecs(pv, p->n, p->ld, p->B, ctx->stream);
    if (e != hipSuccess) return ctx->fail(e, "launch_pack_vecs");
    p->tree.fb_zeroed = zero_counts && p->tree.gram;
    return 0;
  }
  if ((rc = put_vec(ctx, p->st.x, p->ld, x, p->n, p->B, kind))) return rc;
  if ((rc = put_vec(ctx, p->st.lb, p->ld, lb, p->n, p->B, kind))) return rc;
  if ((rc = put_vec(ctx, p->st.ub, p->ld, ub, p->n, p->B, kind))) return rc;
  if ((rc = put_vec(ctx, p->st.scale, p->ld, scale, p->n, p->B, kind))) return rc;
  return 0;
}

extern "C" int blsq_trf_factor_dev(blsq_trf_plan* p, const double* dJ, const double* df,
                                   const double* dx, const double* dlb, const double* dub,
                                   double* dscale_io, int scale_mode) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!dJ) return ctx->bad(2, "J is NULL");
  if (!df) return ctx->bad(3, "f is NULL");
  if (!dx || !dlb || !dub) return ctx->bad(4, "x/lb/ub is NULL");
  if (!dscale_io) return ctx->bad(7, "scale is NULL");
  if (scale_mode < 0 || scale_mode > 2) return ctx->bad(8, "scale_mode");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = verdict_published(p);               // (a verdict nobody read: its counters leave before they are cleared)
  if (rc) return rc;
  if ((rc = trf_put_bounds(p, dx, dlb, dub, dscale_io, hipMemcpyDeviceToDevice, true))) return rc;
  p->pend_scale_io = dscale_io;
  if ((rc = trf_factor_core(p, dJ, df, p->n, scale_mode, nullptr, true))) return rc;
  if (scale_mode != BLSQ_SCALE_GIVEN) {
    HIPCHK(ctx, hipMemcpy2DAsync(dscale_io, sizeof(double) * p->n, p->st.scale,
                                 sizeof(double) * p->ld, sizeof(double) * p->n, p->B,
                                 hipMemcpyDeviceToDevice, ctx->stream));
  }
  return 0;
}

extern "C" int blsq_trf_step_dev(blsq_trf_plan* p, const double* dDelta, const double* dalpha_in,
                                 double active_rtol) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!dDelta) return ctx->bad(2, "Delta is NULL");
  if (!dalpha_in) return ctx->bad(3, "alpha is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // (pass 0 may run on the guessed state of an optimistic factor call; pass 1 only if the guess was wrong)
  // (... and one more if a problem leaves the CSNE tier in this call: it is factored by the next tier, then the step again)
  for (int pass = 0; pass < 4; ++pass) {
    int rc = trf_lm_rounds(p, dDelta, dalpha_in);
    if (rc) return rc;
    const int ncs = p->ncsne;
    if (ncs > 0 && (rc = trf_csne_correct(p, dDelta, dalpha_in))) return rc;
    ctx->begin(K_STEP);
    const PublishArgs pub = verdict_rides(p);
    hipError_t e = launch_trf_step(p->st, &p->lm, dDelta, dalpha_in, active_rtol, p->out,
                                   ctx->stream, &pub);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_trf_step");
    p->lm_counts_clean = true;              // (the step kernel leaves the round counters zeroed)
    bool redo = false;
    if ((rc = trf_resolve(p, &redo))) return rc;
    if (!redo && ncs > 0 && (rc = trf_csne_verdict(p, ncs, &redo))) return rc;
    if (!redo) break;
  }
  return 0;
}

extern "C" int blsq_trf_fetch_factor(blsq_trf_plan* p, double* g, double* g_norm, double* theta,
                                     double* scale, double* sing) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  int rc;
  if ((rc = trf_resolve(p, nullptr))) return rc;
  if ((rc = get_vec(ctx, g, p->n, p->st.g, p->ld, p->B))) return rc;
  if ((rc = get_vec(ctx, scale, p->n, p->st.scale, p->ld, p->B))) return rc;
  if ((rc = get_vec(ctx, sing, p->n, p->st.s, p->ld, p->B))) return rc;
  if (g_norm) HIPCHK(ctx, hipMemcpyAsync(g_norm, p->st.g_norm, sizeof(double) * p->B,
                                         hipMemcpyDeviceToHost, ctx->stream));
  if (theta) HIPCHK(ctx, hipMemcpyAsync(theta, p->st.theta, sizeof(double) * p->B,
                                        hipMemcpyDeviceToHost, ctx->stream));
  return blsq_sync(ctx);
}

extern "C" int blsq_trf_debug_fast(blsq_trf_plan* p, int32_t* fast) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!fast) return ctx->bad(2, "fast is NULL");
  { int rc_ = trf_resolve(p, nullptr); if (rc_) return rc_; }
  HIPCHK(ctx, hipMemcpyAsync(fast, p->lm.fast, sizeof(int) * p->B, hipMemcpyDeviceToHost,
                             ctx->stream));
  return blsq_sync(ctx);
}

extern "C" int blsq_trf_debug_cond(blsq_trf_plan* p, double* k2) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!k2) return ctx->bad(2, "k2 is NULL");
  if (!p->tree.gram) { for (int b = 0; b < p->B; ++b) k2[b] = 0.0; return 0; }
  { int rc_ = trf_resolve(p, nullptr); if (rc_) return rc_; }
  HIPCHK(ctx, hipMemcpyAsync(k2, p->tree.gram_k2.p, sizeof(double) * p->B, hipMemcpyDeviceToHost,
                             ctx->stream));
  return blsq_sync(ctx);
}

extern "C" int blsq_trf_debug_csne(blsq_trf_plan* p, int32_t* on_tier, double* eta) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  { int rc_ = trf_resolve(p, nullptr); if (rc_) return rc_; }
  if (!p->csne_on) {
    for (int b = 0; b < p->B; ++b) { if (on_tier) on_tier[b] = 0; if (eta) eta[b] = 0.0; }
    return 0;
  }
  if (on_tier) HIPCHK(ctx, hipMemcpyAsync(on_tier, p->cs.flag, sizeof(int) * p->B, hipMemcpyDeviceToHost, ctx->stream));
  if (eta) HIPCHK(ctx, hipMemcpyAsync(eta, p->cs.eta, sizeof(double) * p->B, hipMemcpyDeviceToHost, ctx->stream));
  return blsq_sync(ctx);
}

extern "C" int blsq_trf_debug_sweeps(blsq_trf_plan* p, int32_t* sweeps) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!sweeps) return ctx->bad(2, "sweeps is NULL");
  { int rc_ = trf_resolve(p, nullptr); if (rc_) return rc_; }
  HIPCHK(ctx, hipMemcpyAsync(sweeps, p->sweeps.p, sizeof(int) * p->B, hipMemcpyDeviceToHost,
                             ctx->stream));
  return blsq_sync(ctx);
}

extern "C" int blsq_trf_fetch_step(blsq_trf_plan* p, double* alpha_out, double* step_h,
                                   double* step, double* x_new, int64_t* hits,
                                   int64_t* active_new, double* predicted_reduction,
                                   double* step_h_norm, double* correction, int32_t* n_iter,
                                   int32_t* branch, int32_t* status, double* p_h_tr,
                                   double* to_bound, int32_t* choice) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  const int B = p->B, n = p->n, ld = p->ld;
  int rc;
  if ((rc = get_vec(ctx, step_h, n, p->out.step_h, ld, B))) return rc;
  if ((rc = get_vec(ctx, step, n, p->out.step, ld, B))) return rc;
  if ((rc = get_vec(ctx, x_new, n, p->out.x_new, ld, B))) return rc;
  if ((rc = get_vec(ctx, p_h_tr, n, p->out.p_h_tr, ld, B))) return rc;
  if ((rc = get_vec(ctx, (long long*)hits, n, p->out.hits, ld, B))) return rc;
  if ((rc = get_vec(ctx, (long long*)active_new, n, p->out.active_new, ld, B))) return rc;
  std::vector<double> sc((size_t)B * 8);
  std::vector<int> inf((size_t)B * 4);
  HIPCHK(ctx, hipMemcpyAsync(sc.data(), p->out.scal, sizeof(double) * sc.size(),
                             hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(inf.data(), p->out.info, sizeof(int) * inf.size(),
                             hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = blsq_sync(ctx))) return rc;
  for (int b = 0; b < B; ++b) {
    if (predicted_reduction) predicted_reduction[b] = sc[8 * b + 0];
    if (step_h_norm) step_h_norm[b] = sc[8 * b + 1];
    if (correction) correction[b] = sc[8 * b + 2];
    if (alpha_out) alpha_out[b] = sc[8 * b + 3];
    if (to_bound) to_bound[b] = sc[8 * b + 4];
    if (n_iter) n_iter[b] = inf[4 * b + 0];
    if (branch) branch[b] = inf[4 * b + 1];
    if (choice) choice[b] = inf[4 * b + 2];
    if (status) status[b] = inf[4 * b + 3];
  }
  return 0;
}

extern "C" int blsq_trf_factor(blsq_trf_plan* p, const double* J, const double* f,
                               const double* x, const double* lb, const double* ub,
                               double* scale_io, int scale_mode, double* g, double* g_norm,
                               double* theta) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!J) return ctx->bad(2, "J is NULL");
  if (!f) return ctx->bad(3, "f is NULL");
  if (!x || !lb || !ub) return ctx->bad(4, "x/lb/ub is NULL");
  if (!scale_io) return ctx->bad(7, "scale is NULL");
  if (scale_mode < 0 || scale_mode > 2) return ctx->bad(8, "scale_mode");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t jb = sizeof(double) * (size_t)p->B * p->m * p->n;
  const size_t fb = sizeof(double) * (size_t)p->B * p->m;
  if (!p->in_J.p || !p->in_f.p) {       // lazily, and again if an earlier attempt failed half way
    hipError_t e = p->in_J.p ? hipSuccess : p->in_J.alloc(jb);
    if (e != hipSuccess) return ctx->fail(e, "hipMalloc(J staging)");
    e = p->in_f.p ? hipSuccess : p->in_f.alloc(fb);
    if (e != hipSuccess) return ctx->fail(e, "hipMalloc(f staging)");
  }
  int rc = trf_put_bounds(p, x, lb, ub, scale_io, hipMemcpyHostToDevice);
  if (rc) return rc;
  // [J f] crosses PCIe in sub-batches of problems on a copy stream; the Gram of sub-batch k runs while sub-batch
  // k + 1 is in flight — for caller buffers in page-locked memory (blsq_host_alloc), which are DMA'd straight.
  // BLSQ_H2D_PIPE = 0 / 1: never / always (pageable memory too).
  bool piped = false;
  {
    const int pipe_o = ctx->opt.i(OPT_H2D_PIPE);          // -1: page-locked sources only, 0 never, 1 always
    const size_t per = sizeof(double) * (size_t)p->m * (p->n + 1);
    const int sub = (int)std::max<size_t>(1, std::min<size_t>((size_t)p->B, ((size_t)96 << 20) / std::max<size_t>(per, 1)));
    // (page-locked source only — BLSQ_H2D_PIPE = 1 forces it for pageable memory too: there the runtime's own
    //  pin-on-the-fly path for ONE large copy reached 52-53 GB/s, sub-batches of it as little as 27)
    bool pinned_src = false;
    {
      hipPointerAttribute_t at{};
      if (hipPointerGetAttributes(&at, J) == hipSuccess) pinned_src = (at.type == hipMemoryTypeHost);
      else (void)hipGetLastError();                       // (plain malloc memory: "invalid value", not an error here)
    }
    const bool want = pipe_o < 0 ? pinned_src : pipe_o == 1;
    if (p->tree.gram && p->B >= 2 * sub && want) {
      piped = true;
      const int nsub = (p->B + sub - 1) / sub;
      while ((int)ctx->copy_ev.size() < nsub) {
        hipEvent_t ev = nullptr;
        HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->copy_ev.push_back(ev);
      }
      for (int k = 0, k0 = 0; k0 < p->B; ++k, k0 += sub) {
        const int nb = std::min(sub, p->B - k0);
        const size_t jo = (size_t)k0 * p->m * p->n, fo = (size_t)k0 * p->m;
        HIPCHK(ctx, hipMemcpyAsync(p->in_J.as<double>() + jo, J + jo, sizeof(double) * (size_t)nb * p->m * p->n,
                                   hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipMemcpyAsync(p->in_f.as<double>() + fo, f + fo, sizeof(double) * (size_t)nb * p->m,
                                   hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->copy_ev[k], ctx->copy_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->copy_ev[k], 0));
        if ((rc = p->tree.run_gram_only(ctx, p->in_J.as<double>(), p->in_f.as<double>(), p->n, nullptr, false, k0, nb)))
          return rc;
      }
    }
  }
  if (!piped) {
    HIPCHK(ctx, hipMemcpyAsync(p->in_J.p, J, jb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p->in_f.p, f, fb, hipMemcpyHostToDevice, ctx->stream));
  }
  if ((rc = trf_factor_core(p, p->in_J.as<double>(), p->in_f.as<double>(), p->n, scale_mode, nullptr, false, piped)))
    return rc;
  return blsq_trf_fetch_factor(p, g, g_norm, theta,
                               scale_mode != BLSQ_SCALE_GIVEN ? scale_io : nullptr, nullptr);
}

extern "C" int blsq_trf_step(blsq_trf_plan* p, const double* Delta, double* alpha_io,
                             double active_rtol, double* step_h, double* step, double* x_new,
                             int64_t* hits, int64_t* active_new, double* predicted_reduction,
                             double* step_h_norm, double* correction, int32_t* n_iter,
                             int32_t* branch, int32_t* status) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!Delta) return ctx->bad(2, "Delta is NULL");
  if (!alpha_io) return ctx->bad(3, "alpha is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double* dD = p->in_scal.as<double>();
  double* dA = dD + p->B;
  HIPCHK(ctx, hipMemcpyAsync(dD, Delta, sizeof(double) * p->B, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dA, alpha_io, sizeof(double) * p->B, hipMemcpyHostToDevice, ctx->stream));
  int rc = blsq_trf_step_dev(p, dD, dA, active_rtol);
  if (rc) return rc;
  return blsq_trf_fetch_step(p, alpha_io, step_h, step, x_new, hits, active_new,
                             predicted_reduction, step_h_norm, correction, n_iter, branch,
                             status, nullptr, nullptr, nullptr);
}

// ================================================================= TSQR ====
extern "C" int blsq_tsqr_tri_ld(int n) { return round_up(n + 1, 16); }

extern "C" int blsq_tsqr_plan_create(blsq_ctx* ctx, int m_local, long long m_total, int n,
                                     int nranks, blsq_trf_plan** out) {
  if (!ctx) return -1;
  if (!out) return ctx->bad(6, "out is NULL");
  *out = nullptr;
  if (m_local <= 0) return ctx->bad(2, "m_local must be positive");
  if (m_total < m_local) return ctx->bad(3, "m_total must be >= m_local");
  if (n <= 0) return ctx->bad(4, "n must be positive");
  if (nranks <= 0) return ctx->bad(5, "nranks must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  blsq_trf_plan* p = new blsq_trf_plan();
  p->ctx = ctx; p->B = 1; p->m = m_local; p->n = n; p->nranks = nranks;
  // the GLOBAL row count enters the reference's rank test eps * m * s[0] (trust_region.py:109):
  // it must be the same number on every rank, whatever the sizes of the row blocks
  p->m_total = m_total > 2147483647LL ? 2147483647 : (int)m_total;
  const int NPAD = round_up(n + 1, 16);
  if (nranks > 1 && !merge_fits(n)) {
    delete p;
    return ctx->bad(4, "TSQR needs n <= 512");
  }
  const int aug_rp = std::max(aug_rows(n), NPAD);
  // scratch must also cover the combine merges: nranks triangles, G per workgroup
  const int G = merge_group(n);
  const size_t comb_rows = (size_t)((nranks + G - 1) / G) * (size_t)(G * NPAD);
  int rc = p->tree.build(ctx, 1, m_local, n, std::max((size_t)aug_rp, comb_rows));
  if (rc == 0 && p->tree.gram) p->tree.k2_max = gram_k2_max(m_total, ctx->opt.d(OPT_GRAM_K2_MAX));   // (the Gram sums over ALL ranks' rows)
  if (rc == 0) { p->ld = p->tree.NPAD; rc = trf_alloc_state(p); }
  if (rc == 0) {
    // two ping-pong levels for the combine tree
    hipError_t e = p->Rcomb.alloc(sizeof(double) * 2 * (size_t)((nranks + G - 1) / G + 1) *
                                  NPAD * NPAD);
    if (e != hipSuccess) rc = ctx->fail(e, "hipMalloc(Rcomb)");
  }
  if (rc == 0) {
    hipError_t e = p->Rstack.alloc(sizeof(double) * (size_t)nranks * NPAD * NPAD);
    if (e != hipSuccess) rc = ctx->fail(e, "hipMalloc(Rstack)");
  }
  if (rc != 0) { blsq_trf_plan_destroy(p); return rc; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->trf_plans.push_back(p);
  *out = p;
  return 0;
}

extern "C" int blsq_tsqr_local_dev(blsq_trf_plan* p, const double* dJ_block,
                                   const double* df_block, double* dtri_out) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!dJ_block) return ctx->bad(2, "J block is NULL");
  if (!df_block) return ctx->bad(3, "f block is NULL");
  if (!dtri_out) return ctx->bad(4, "tri_out is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = p->tree.run(ctx, dJ_block, df_block, p->n);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(dtri_out, p->tree.Rfinal(), sizeof(double) * p->ld * p->ld,
                             hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

namespace blsq_host {
// merge the stack of nranks triangles (rank order) and run the n-space path on the result
int tsqr_merge_and_finish(blsq_trf_plan* p, const double* dtri_stack, int scale_mode,
                          double* dscale_io, int redo = 0) {
  blsq_ctx* ctx = p->ctx;
  const int NPAD = p->ld;
  const double* src = dtri_stack;
  int ntri = p->nranks;
  const int G = merge_group(p->n);
  double* pp[2] = {p->Rcomb.as<double>(),
                   p->Rcomb.as<double>() + (size_t)((p->nranks + G - 1) / G + 1) * NPAD * NPAD};
  int flip = 0;
  while (ntri > 1) {
    QrArgs q = p->tree.base_args();
    q.A = src; q.strideA = 0; q.ldA = NPAD; q.rowsA = ntri * NPAD; q.F = nullptr; q.strideF = 0;
    q.stack_rows = NPAD;
    q.rows_per_leaf = G * NPAD;
    const int nleaf = (ntri + G - 1) / G;
    q.RP = std::max(round_up(std::min(q.rows_per_leaf, q.rowsA), 16), NPAD);
    q.Rout = pp[flip];
    ctx->begin(K_QR_MERGE);
    hipError_t e = launch_qr(q, nleaf, 1, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_qr(combine)");
    src = pp[flip];
    flip ^= 1;
    ntri = nleaf;
  }
  int rc = trf_after_triangle(p, src, scale_mode, redo);
  if (rc) return rc;
  if (scale_mode != BLSQ_SCALE_GIVEN) {
    HIPCHK(ctx, hipMemcpyAsync(dscale_io, p->st.scale, sizeof(double) * p->n,
                               hipMemcpyDeviceToDevice, ctx->stream));
  }
  return 0;
}
}  // namespace blsq_host

extern "C" int blsq_tsqr_combine_dev(blsq_trf_plan* p, const double* dtri_stack,
                                     const double* dx, const double* dlb, const double* dub,
                                     double* dscale_io, int scale_mode) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!dtri_stack) return ctx->bad(2, "tri stack is NULL");
  if (!dx || !dlb || !dub) return ctx->bad(3, "x/lb/ub is NULL");
  if (!dscale_io) return ctx->bad(6, "scale is NULL");
  if (scale_mode < 0 || scale_mode > 2) return ctx->bad(7, "scale_mode");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = trf_put_bounds(p, dx, dlb, dub, dscale_io, hipMemcpyDeviceToDevice);
  if (rc) return rc;
  return tsqr_merge_and_finish(p, dtri_stack, scale_mode, dscale_io);
}

// The whole factor call of one tall problem whose rows are split over the ranks of the ctx's
// communicator (blsq_comm_init), this rank's row block in, replicated factor state out:
//   normal-equations front end:  local Gram -> ncclAllReduce(sum) of the (n+1)^2 Gram -> Cholesky + gate
//                                (replicated, bit-identical on every rank)
//   if the gate rejects:         local Householder TSQR -> ncclAllGather of the triangles -> merge
// then the ordinary n-space path.  Everything is enqueued on the ctx stream; the only host wait is
// the read-back of the gate's verdict (one integer).
extern "C" int blsq_tsqr_factor_dev(blsq_trf_plan* p, const double* dJ_block, const double* df_block,
                                    const double* dx, const double* dlb, const double* dub,
                                    double* dscale_io, int scale_mode) {
  if (!p) return -1;
  blsq_ctx* ctx = p->ctx;
  if (!dJ_block) return ctx->bad(2, "J block is NULL");
  if (!df_block) return ctx->bad(3, "f block is NULL");
  if (!dx || !dlb || !dub) return ctx->bad(4, "x/lb/ub is NULL");
  if (!dscale_io) return ctx->bad(7, "scale is NULL");
  if (scale_mode < 0 || scale_mode > 2) return ctx->bad(8, "scale_mode");
  if (p->nranks > 1 && (!ctx->comm || ctx->comm_ranks != p->nranks))
    return ctx->bad(1, "the plan's ranks need a communicator of that size on this ctx (blsq_comm_init)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // (zero_counts: with the normal-equations front end the prep launch of the Gram stage packs the vectors and clears the
  //  gate counters — no pack launch, no fill)
  int rc = trf_put_bounds(p, dx, dlb, dub, dscale_io, hipMemcpyDeviceToDevice, true);
  if (rc) return rc;
  auto put_scale = [&]() -> int {
    if (scale_mode != BLSQ_SCALE_GIVEN)
      HIPCHK(ctx, hipMemcpyAsync(dscale_io, p->st.scale, sizeof(double) * p->n,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  };
  // max AND min over the ranks of a few integers (the ranks' copies must be equal): nv <= 4 values
  auto agree = [&](const int* vals, int nv, const char* what) -> int {
    double* d = p->Rstack.as<double>();                   // (free until the all-gather)
    double h[8];
    for (int i = 0; i < nv; ++i) { h[2 * i] = (double)vals[i]; h[2 * i + 1] = -(double)vals[i]; }
    HIPCHK(ctx, hipMemcpyAsync(d, h, sizeof(double) * 2 * nv, hipMemcpyHostToDevice, ctx->stream));
    RCCLCHK(ctx, g_rccl.AllReduce(d, d, (size_t)(2 * nv), ncclDouble, ncclMax, ctx->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h, d, sizeof(double) * 2 * nv, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nv; ++i)
      if (h[2 * i] != -h[2 * i + 1]) {
        ctx->err = std::string("blsq_tsqr_factor_dev: the ranks disagree on ") + what +
                   " (x, bounds, scale, scale_mode, the plan's n / m_total and the BLSQ_* environment must be "
                   "identical on every rank)";
        return BLSQ_ERR_RANKS_DISAGREE;
      }
    return 0;
  };
  if (p->nranks > 1 && !p->ranks_agreed) {
    // once per plan, BEFORE the first data collective: a rank whose front end is switched off would enter
    // the all-gather while the others sit in the Gram's all-reduce
    const int cfg[3] = {p->tree.gram ? 1 : 0, p->n, p->m_total};
    if ((rc = agree(cfg, 3, "the plan (normal-equations front end on / off, n, m_total)"))) return rc;
    p->ranks_agreed = true;
  }
  int redo = 0;
  if (p->tree.gram) {
    if ((rc = p->tree.run_gram_only(ctx, dJ_block, df_block, p->n, nullptr, /*collective=*/true))) return rc;
    int nfb = 0;
    if ((rc = trf_gram_stage(p, scale_mode, nullptr, &nfb))) return rc;   // replicated: same verdict everywhere
    if (p->nranks > 1) {
      // ... which is checked, not assumed: the route (return here, or enter the all-gather below) is taken
      // from the max AND the min of the verdict over the ranks.  Ranks that disagree (inputs or environment
      // that differ between them) all fail with the same code instead of one of them waiting in a
      // collective the others never enter.
      const int vd[2] = {nfb, scale_mode};
      if ((rc = agree(vd, 2, "the gate's verdict"))) return rc;
    }
    if (nfb == 0) {
      if ((rc = trf_finish(p))) return rc;
      return put_scale();
    }
    redo = 1;                                             // (prepared from the Gram once already)
  }
  // Householder route: this rank's triangle, all-gather, replicated merge
  if ((rc = p->tree.run_levels(ctx, dJ_block, df_block, p->n, nullptr))) return rc;
  if (p->nranks == 1) {
    if ((rc = trf_after_triangle(p, p->tree.Rfinal(), scale_mode, redo))) return rc;
    return put_scale();
  }
  const size_t tri = (size_t)p->ld * p->ld;
  RCCLCHK(ctx, g_rccl.AllGather(p->tree.Rfinal(), p->Rstack.as<double>(), tri, ncclDouble, ctx->comm,
                                ctx->stream));
  return tsqr_merge_and_finish(p, p->Rstack.as<double>(), scale_mode, dscale_io, redo);
}
