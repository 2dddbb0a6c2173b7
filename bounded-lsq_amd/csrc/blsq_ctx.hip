// C-ABI entry points (include/blsq.h): contexts, memory, timing, the communicator, diagnostics.
#include "blsq_host.h"

// Device counters -> host without a blit: ONE lane stores [v0, v1, v2] into a pinned (coherent) slot, fences, and
// RELEASES the sequence number the host polls for with an acquire load (publish_ints, blsq_kernels.h).  (A hipMemcpyAsync of 12 bytes is a blit kernel of 4 us and a signal the
// next dispatch waits 6 - 9 us behind — `tools/trace_gaps.py` — four times per step-solve call of the headline
// batch, once in the middle of a 0.17 ms step of the 512 x 64 batches.)
__global__ void publish_ints_kernel(const int* __restrict__ src, int n, int* dst, int seq) {
  if (threadIdx.x != 0) return;
  publish_ints(PublishArgs{src, n, dst, seq});
}

hipError_t blsq_ctx::publish(const int* src, int n, int* slot, hipEvent_t ev, int* expect) {
    if (!pub_direct()) {
      hipError_t e = hipMemcpyAsync(slot, src, n * sizeof(int), hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipEventRecord(ev, stream);
      return e;
    }
    *expect = ++pub_seq;
    hipLaunchKernelGGL(publish_ints_kernel, dim3(1), dim3(64), 0, stream, src, n, slot, *expect);
    return hipGetLastError();
}

namespace blsq_host {
Rccl g_rccl;
int rccl_fail(blsq_ctx* ctx, ncclResult_t r, const char* where) {
  ctx->err = std::string(where) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return RCCL_ERR_BASE + (int)r;
}

int put_vec(blsq_ctx* ctx, double* dst, int ld, const double* src, int n, int B, hipMemcpyKind kind) {
  HIPCHK(ctx, hipMemcpy2DAsync(dst, sizeof(double) * ld, src, sizeof(double) * n,
                               sizeof(double) * n, B, kind, ctx->stream));
  return 0;
}

int ctx_resolve_pending(blsq_ctx* ctx) {
  for (blsq_trf_plan* p : ctx->trf_plans)
    if (p->pending) { int rc = trf_resolve(p, nullptr); if (rc) return rc; }
  for (blsq_dogbox_plan* p : ctx->dog_plans)
    if (p->pending) { int rc = dog_resolve(p, nullptr); if (rc) return rc; }
  return 0;
}
}  // namespace blsq_host

// ============================================================ ctx / misc ===
extern "C" int blsq_version(void) { return 100; }

extern "C" int blsq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int blsq_ctx_create(int device_id, blsq_ctx** out) {
  if (!out) return -2;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return e != hipSuccess ? (int)e : (int)hipErrorNoDevice;
  if (device_id < 0 || device_id >= ndev) return -1;
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return (int)e;
  blsq_ctx* c = new blsq_ctx();
  c->device = device_id;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return (int)e; }
  e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { hipStreamDestroy(c->stream); delete c; return (int)e; }
  e = hipHostMalloc((void**)&c->pinned, 128 * sizeof(int), hipHostMallocCoherent);
  if (e != hipSuccess) { hipStreamDestroy(c->stream); delete c; return (int)e; }
  memset(c->pinned, 0, 128 * sizeof(int));
  c->opt = options_from_env();
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->lm_ev[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&c->cq_accept_dev, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(c->cq_accept_dev, 0, sizeof(unsigned long long));
  if (e != hipSuccess) { hipHostFree(c->pinned); hipStreamDestroy(c->stream); delete c; return (int)e; }
  *out = c;
  return 0;
}

extern "C" int blsq_ctx_destroy(blsq_ctx* ctx) {
  if (!ctx) return -1;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  ctx->collect();
  if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
  ctx->comm = nullptr;
  for (auto e : ctx->pool) hipEventDestroy(e);
  for (auto e : ctx->lm_ev) if (e) hipEventDestroy(e);
  for (auto e : ctx->copy_ev) hipEventDestroy(e);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  hipStreamDestroy(ctx->stream);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  if (ctx->cq_accept_dev) hipFree(ctx->cq_accept_dev);
  delete ctx;
  return 0;
}

extern "C" const char* blsq_last_error(const blsq_ctx* ctx) {
  return ctx ? ctx->err.c_str() : "null ctx";
}

// ---- the option table (blsq_options.h) ------------------------------------------------------------
extern "C" int blsq_option_count(void) { return OPT_COUNT; }
extern "C" int blsq_option_info(int i, const char** name, const char** env, double* dflt, const char** doc) {
  if (i < 0 || i >= OPT_COUNT) return -1;
  if (name) *name = kOptTable[i].name;
  if (env) *env = kOptTable[i].env;
  if (dflt) *dflt = kOptTable[i].dflt;
  if (doc) *doc = kOptTable[i].doc;
  return 0;
}
extern "C" int blsq_ctx_set_option(blsq_ctx* ctx, const char* name, double value) {
  if (!ctx) return -1;
  const int k = option_index(name);
  if (k < 0) return ctx->bad(2, "unknown option (blsq_option_info lists them)");
  if (!(value == value)) return ctx->bad(3, "option value is NaN");
  ctx->opt.v[k] = value;
  return 0;
}
extern "C" int blsq_ctx_get_option(const blsq_ctx* ctx, const char* name, double* value) {
  if (!ctx) return -1;
  const int k = option_index(name);
  if (k < 0 || !value) return k < 0 ? -2 : -3;
  *value = ctx->opt.v[k];
  return 0;
}

// Every verdict an optimistic factor call left pending on a plan of this ctx is read, and a wrong guess
// repaired (Householder tree on the caller's J, which is why this runs in blsq_sync and before the
// library frees or overwrites device memory: after blsq_sync nothing of the caller's J / f is read again).

extern "C" int blsq_sync(blsq_ctx* ctx) {
  if (!ctx) return -1;
  { int rc_ = ctx_resolve_pending(ctx); if (rc_) return rc_; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->collect();
  return 0;
}

// ================================================================= comm ====
extern "C" int blsq_comm_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int blsq_comm_get_id(blsq_ctx* ctx, void* id_out, size_t bytes) {
  if (!ctx) return -1;
  if (!id_out) return ctx->bad(2, "id_out is NULL");
  if (bytes < (size_t)NCCL_UNIQUE_ID_BYTES) return ctx->bad(3, "id buffer too small (blsq_comm_id_bytes)");
  if (!g_rccl.load()) { ctx->err = g_rccl.err; return RCCL_ERR_BASE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  RCCLCHK(ctx, g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int blsq_comm_init(blsq_ctx* ctx, int nranks, int rank, const void* id, size_t bytes) {
  if (!ctx) return -1;
  if (nranks < 1) return ctx->bad(2, "nranks must be positive");
  if (rank < 0 || rank >= nranks) return ctx->bad(3, "rank out of range");
  if (!id) return ctx->bad(4, "id is NULL");
  if (bytes < (size_t)NCCL_UNIQUE_ID_BYTES) return ctx->bad(5, "id too small (blsq_comm_id_bytes)");
  if (ctx->comm) return ctx->bad(1, "this ctx already has a communicator");
  if (!g_rccl.load()) { ctx->err = g_rccl.err; return RCCL_ERR_BASE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, NCCL_UNIQUE_ID_BYTES);
  RCCLCHK(ctx, g_rccl.CommInitRank(&ctx->comm, nranks, uid, rank));
  ctx->comm_ranks = nranks; ctx->comm_rank = rank;
  return 0;
}

extern "C" int blsq_comm_destroy(blsq_ctx* ctx) {
  if (!ctx) return -1;
  if (!ctx->comm) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  RCCLCHK(ctx, g_rccl.CommDestroy(ctx->comm));
  ctx->comm = nullptr; ctx->comm_ranks = 1; ctx->comm_rank = 0;
  return 0;
}

extern "C" const char* blsq_comm_library(int* version_out) {
  int v = 0;
  if (g_rccl.lib && g_rccl.GetVersion) g_rccl.GetVersion(&v);
  if (version_out) *version_out = v;
  return g_rccl.path.c_str();
}

extern "C" int blsq_comm_size(const blsq_ctx* ctx) { return ctx ? ctx->comm_ranks : 0; }
extern "C" int blsq_comm_rank(const blsq_ctx* ctx) { return ctx ? ctx->comm_rank : -1; }

// max over the ranks of `n` host doubles (n <= 64), which is also a barrier: used by bench.py for
// the max-over-ranks timing; blocks until the collective has finished on the ctx stream
extern "C" int blsq_comm_allreduce_max(blsq_ctx* ctx, double* host_io, int n) {
  if (!ctx) return -1;
  if (!host_io) return ctx->bad(2, "host_io is NULL");
  if (n < 1 || n > 64) return ctx->bad(3, "n must be in 1..64");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (!ctx->comm || ctx->comm_ranks == 1) return 0;
  double* d = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&d, sizeof(double) * 64));
  hipError_t e = hipMemcpyAsync(d, host_io, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = g_rccl.AllReduce(d, d, (size_t)n, ncclDouble, ncclMax, ctx->comm, ctx->stream);
  if (e == hipSuccess && r == ncclSuccess)
    e = hipMemcpyAsync(host_io, d, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d);
  if (r != ncclSuccess) return rccl_fail(ctx, r, "ncclAllReduce(max)");
  if (e != hipSuccess) return ctx->fail(e, "blsq_comm_allreduce_max");
  return 0;
}

extern "C" int blsq_dev_malloc(blsq_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx) return -1;
  if (!dptr) return ctx->bad(3, "dptr is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(dptr, bytes ? bytes : 8));
  return 0;
}
extern "C" int blsq_dev_free(blsq_ctx* ctx, void* dptr) {
  if (!ctx) return -1;
  { int rc_ = ctx_resolve_pending(ctx); if (rc_) return rc_; }   // (it may be the J of a pending verdict)
  HIPCHK(ctx, hipFree(dptr));
  return 0;
}
extern "C" int blsq_host_alloc(blsq_ctx* ctx, size_t bytes, void** hptr) {
  if (!ctx) return -1;
  if (!hptr) return ctx->bad(3, "hptr is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault));
  return 0;
}
extern "C" int blsq_host_free(blsq_ctx* ctx, void* hptr) {
  if (!ctx) return -1;
  { int rc_ = ctx_resolve_pending(ctx); if (rc_) return rc_; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipHostFree(hptr));
  return 0;
}
extern "C" int blsq_memcpy_h2d(blsq_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return -1;
  { int rc_ = ctx_resolve_pending(ctx); if (rc_) return rc_; }   // (dst may be the J of a pending verdict)
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int blsq_memcpy_d2h(blsq_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int blsq_timing_enable(blsq_ctx* ctx, int on) {
  if (!ctx) return -1;
  if (on < 0 || on >= 2 + K_NSLOT) return ctx->bad(2, "timing mode");
  ctx->timing = on;
  return 0;
}
extern "C" int blsq_timing_reset(blsq_ctx* ctx) {
  if (!ctx) return -1;
  for (int i = 0; i < K_NSLOT; ++i) { ctx->t_ms[i] = 0; ctx->t_n[i] = 0; }
  return 0;
}
extern "C" int blsq_timing_count(const blsq_ctx*) { return K_NSLOT; }
extern "C" int blsq_timing_get(blsq_ctx* ctx, int slot, const char** name, double* total_ms,
                               int64_t* launches) {
  if (!ctx) return -1;
  if (slot < 0 || slot >= K_NSLOT) return ctx->bad(2, "slot");
  if (name) *name = kSlotNames[slot];
  if (total_ms) *total_ms = ctx->t_ms[slot];
  if (launches) *launches = ctx->t_n[slot];
  return 0;
}


#ifdef BLSQ_CHOL_STAMPS
namespace blsq { int chol_debug_stamps(long long* host); }
extern "C" int blsq_debug_chol_stamps(long long* host) { return blsq::chol_debug_stamps(host); }
namespace blsq { int gram_debug_stamps(long long* host); }
extern "C" int blsq_debug_gram_stamps(long long* host) { return blsq::gram_debug_stamps(host); }
namespace blsq { int cqr2_debug_stamps(long long* host); }
extern "C" int blsq_debug_cqr2_stamps(long long* host) { return blsq::cqr2_debug_stamps(host); }
namespace blsq { int step_debug_stamps(long long* host); }
extern "C" int blsq_debug_step_stamps(long long* host) { return blsq::step_debug_stamps(host); }
namespace blsq { int dog_debug_stamps(long long* host); }
extern "C" int blsq_debug_dog_stamps(long long* host) { return blsq::dog_debug_stamps(host); }
#endif
extern "C" int blsq_debug_qr_stamps(void* dbuf) {
  set_qr_debug_buffer(reinterpret_cast<double*>(dbuf));
  return 0;
}

extern "C" int blsq_debug_probe(blsq_ctx* ctx, int kind, int arg, double out[3]) {
  if (!ctx) return -1;
  if (!out) return ctx->bad(4, "out is NULL");
  if (kind != 0 && kind != 1) return ctx->bad(2, "kind must be 0 (MFMA f64) or 1 (copy)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipEvent_t e0 = ctx->get_event(), e1 = ctx->get_event();
  float ms = 0.f;
  if (kind == 0) {
    double* sink = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&sink, 64));
    long nm = 0;
    const int iters = 20000;                       // x 8 MFMAs: ~10 ms per launch at the nominal rate
    hipError_t e = launch_mfma_probe(arg, 200, sink, &nm, ctx->stream);   // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) e = launch_mfma_probe(arg, iters, sink, &nm, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipFree(sink);
    ctx->pool.push_back(e0); ctx->pool.push_back(e1);
    if (e != hipSuccess) return ctx->fail(e, "mfma probe");
    out[0] = (double)nm * 2048.0 / ((double)ms * 1e-3) * 1e-12;
    out[1] = (double)nm; out[2] = ms;
    return 0;
  }
  if (arg <= 0) return ctx->bad(3, "copy probe needs a size in MiB");
  const size_t bytes = (size_t)arg << 20;
  void *src = nullptr, *dst = nullptr;
  hipError_t e = hipMalloc(&src, bytes);
  if (e == hipSuccess) e = hipMalloc(&dst, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(src, 1, bytes, ctx->stream);
  if (e == hipSuccess) e = launch_copy_probe(src, dst, bytes, ctx->stream);   // warm-up
  if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
  for (int r = 0; r < 4 && e == hipSuccess; ++r) e = launch_copy_probe(src, dst, bytes, ctx->stream);
  if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (src) hipFree(src);
  if (dst) hipFree(dst);
  ctx->pool.push_back(e0); ctx->pool.push_back(e1);
  if (e != hipSuccess) return ctx->fail(e, "copy probe");
  out[0] = 4.0 * 2.0 * (double)bytes / ((double)ms * 1e-3) * 1e-9;
  out[1] = 4.0 * 2.0 * (double)bytes; out[2] = ms;
  return 0;
}

extern "C" int blsq_debug_gram_stats(blsq_ctx* ctx, uint64_t* out2, int reset) {
  if (!ctx) return -1;
  if (!out2) return ctx->bad(2, "out is NULL");
  out2[0] = (uint64_t)ctx->gram_fast;
  out2[1] = (uint64_t)ctx->gram_fallback;
  if (reset) { ctx->gram_fast = 0; ctx->gram_fallback = 0; }
  return 0;
}

extern "C" int blsq_debug_csne_stats(blsq_ctx* ctx, uint64_t out[3], int reset) {
  if (!ctx) return -1;
  if (out) { out[0] = ctx->csne_routed; out[1] = ctx->csne_steps; out[2] = ctx->csne_declined; }
  if (reset) { ctx->csne_routed = 0; ctx->csne_steps = 0; ctx->csne_declined = 0; }
  return 0;
}

extern "C" int blsq_debug_cqr2_stats(blsq_ctx* ctx, uint64_t* out1, int reset) {
  if (!ctx) return -1;
  if (!out1) return ctx->bad(2, "out is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  unsigned long long v = 0ULL;
  HIPCHK(ctx, hipMemcpyAsync(&v, ctx->cq_accept_dev, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
  if (reset) HIPCHK(ctx, hipMemsetAsync(ctx->cq_accept_dev, 0, sizeof(v), ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  out1[0] = v;
  return 0;
}

extern "C" int blsq_debug_cqr_stats(blsq_ctx* ctx, uint64_t out[2], int reset) {
  if (!ctx) return -1;
  if (!out) return ctx->bad(2, "out is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  unsigned long long v[2] = {0ULL, 0ULL};
  hipError_t e = qr_cqr_stats(v, reset, ctx->stream);
  if (e != hipSuccess) return ctx->fail(e, "qr_cqr_stats");
  out[0] = v[0]; out[1] = v[1];
  return 0;
}

