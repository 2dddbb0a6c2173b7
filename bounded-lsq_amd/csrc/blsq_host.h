// Host side of the C-ABI (include/blsq.h), shared by its translation units: contexts, the RCCL binding, device buffers,
// the factorisation front end of a plan (QrTree: Gram / certificate / CholeskyQR2 / Householder tree) and the plan
// structures.  blsq_ctx.hip: contexts, memory, timing, communicator, diagnostics; blsq_trf.hip: TRF and the row-split
// (TSQR) plans; blsq_dogbox.hip: dogbox plans; blsq_outer.hip: the batched outer drivers and finite differences.
// Internal: nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen'ed on first use
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/blsq.h"
#include "blsq_kernels.h"

using namespace blsq;

namespace {

constexpr int RMAX = QR_MAX_TILES * 16;   // rows a workgroup can stage (qr_panel.hip): 1024

enum Slot { K_QR_LEAF = 0, K_QR_MERGE, K_PREP, K_QR_AUG, K_JACOBI, K_STEP, K_LM_GATE, K_LM_QR, K_LM_SOLVE,
            K_GRAM, K_GRAM_CHOL, K_GRAM_GATE, K_AUG_CHOL, K_LM_CHOL, K_CQR2_APPLY, K_CQR2_COMBINE, K_CSNE_PASS,
            K_CSNE_FIX, K_NSLOT };
static const char* const kSlotNames[K_NSLOT] = {"qr_leaf", "qr_merge", "prep", "qr_aug", "jacobi_svd", "step",
                                   "lm_gate", "lm_qr", "lm_solve", "gram", "gram_chol", "gram_gate",
                                   "aug_chol", "lm_chol", "cqr2_apply", "cqr2_combine", "csne_pass", "csne_fix"};

inline int round_up(int v, int q) { return (v + q - 1) / q * q; }
// rows of the stacked systems [R D; E] / [R_aug; sqrt(alpha) I]: two blocks of
// round_up(n, 16) rows each (the second block starts on a tile boundary, see qr_panel.hip)
inline int aug_block_rows(int n) { return (n + 15) / 16 * 16; }
inline int aug_rows(int n) { return 2 * aug_block_rows(n); }

// triangles merged per workgroup: the kernel stages ceil(n/16) tiles of each (>= 2 must fit)
inline int merge_group(int n) { return std::max(2, QR_MAX_TILES / ((n + 15) / 16)); }
inline bool merge_fits(int n) { return 2 * ((n + 15) / 16) <= QR_MAX_TILES; }

}  // namespace


// Device counters -> host without a blit (blsq_ctx.hip; publish_ints, blsq_kernels.h)
__global__ void publish_ints_kernel(const int* __restrict__ src, int n, int* dst, int seq);

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#endif
}
struct blsq_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;              // host-pointer API: H2D of the next problems while the Gram of the last runs
  std::vector<hipEvent_t> copy_ev;
  std::string err;
  int timing = 0;                   // 0 off, 1 every slot, 2 + slot: that slot only (blsq_timing_enable)
  bool timing_open = false;         // the last begin() recorded an event
  double t_ms[K_NSLOT] = {0};
  int64_t t_n[K_NSLOT] = {0};
  struct Pending { int slot; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  int* pinned = nullptr;            // 128 pinned host ints: device -> host counters without staging
                                    // ([0..3] one-shot read-backs, [32 + 4 r ..] the slot of Newton round r)
  int pub_seq = 0;                  // sequence number of the last publish()
  blsq::Options opt;                // the switches of this ctx (blsq_options.h: environment at creation, blsq_ctx_set_option)
  bool pub_direct() const { return opt.on(OPT_PUBLISH); }        // 0: hipMemcpyAsync + event instead of the publishing kernel
  bool fuse_pack() const { return opt.on(OPT_FUSE_PACK); }       // 0: the caller's vectors are packed by a launch of their own
  bool pub_ride() const { return opt.on(OPT_PUBLISH_RIDE); }     // 0: the verdict's counters get a publishing launch of their own
  hipEvent_t lm_ev[2] = {nullptr, nullptr};   // read-back of the counter of round r has landed (r & 1)
  long long gram_fast = 0, gram_fallback = 0;   // problems factored by the normal equations / handed to the QR tree
  unsigned long long* cq_accept_dev = nullptr;  // device counter: rejected problems the CholeskyQR2 tier factored
  // CSNE tier (csne_kernels.hip): problems routed to it by factor calls, step-solves it delivered, step-solves it
  // declined (acceptance failed at step time: the problem went on to CholeskyQR2 / the tree)
  unsigned long long csne_routed = 0, csne_steps = 0, csne_declined = 0;
  // collective over the ranks of one tall problem (RCCL over xGMI; blsq_comm_*)
  ncclComm_t comm = nullptr;
  int comm_ranks = 1, comm_rank = 0;
  // plans of this ctx (an optimistic factor call leaves a verdict pending on its plan: blsq_sync and
  // the calls that may invalidate the caller's J resolve it, see ctx_resolve_pending)
  std::vector<blsq_trf_plan*> trf_plans;
  std::vector<blsq_dogbox_plan*> dog_plans;

  int fail(hipError_t e, const char* where) {
    err = std::string(where) + ": " + hipGetErrorString(e);
    return (int)e;
  }
  // n <= 3 device ints -> the 16-byte pinned slot `slot` ([3] = sequence number, returned in *expect); `ev` is
  // recorded on the blit route only
  hipError_t publish(const int* src, int n, int* slot, hipEvent_t ev, int* expect);   // (blsq_ctx.hip)
  // ... and the wait for it: polls the slot; looks at the stream now and then so that a failed launch cannot hang it
  hipError_t await(const int* slot, hipEvent_t ev, int expect) {
    if (!pub_direct()) return hipEventSynchronize(ev);
    for (unsigned long it = 1;; ++it) {
      if (__atomic_load_n(slot + 3, __ATOMIC_ACQUIRE) == expect) return hipSuccess;
      if ((it & 0x3fff) == 0) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return __atomic_load_n(slot + 3, __ATOMIC_ACQUIRE) == expect ? hipSuccess : hipErrorUnknown;
        if (q != hipErrorNotReady) return q;
      }
      cpu_relax();
    }
  }
  int bad(int argidx, const char* what) {
    err = std::string("invalid argument: ") + what;
    return -argidx;
  }
  hipEvent_t get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    // (timing only: without the system-scope fence a default event carries — its cache write-back and invalidation
    //  between every two launches slowed the step it measured by 2-3 %)
    hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    return e;
  }
  void begin(int slot) {
    timing_open = timing == 1 || (timing >= 2 && timing - 2 == slot);
    if (!timing_open) return;
    Pending p{slot, get_event(), get_event()};
    hipEventRecord(p.a, stream);
    pending.push_back(p);
  }
  void end() {
    if (!timing_open) return;
    timing_open = false;
    hipEventRecord(pending.back().b, stream);
  }
  void collect() {                  // after a stream sync
    for (auto& p : pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        t_ms[p.slot] += ms;
        t_n[p.slot] += 1;
      }
      pool.push_back(p.a);
      pool.push_back(p.b);
    }
    pending.clear();
  }
};
#define HIPCHK(ctx, call)                                   \
  do {                                                      \
    hipError_t e__ = (call);                                \
    if (e__ != hipSuccess) return (ctx)->fail(e__, #call);  \
  } while (0)
namespace blsq_host {

// ---- RCCL, bound at run time ------------------------------------------------------------------
// Only the tall-problem path needs a collective, so librccl (0.5 GB) is not a link-time dependency:
// it is dlopen'ed by the first blsq_comm_* call, from the directory of the HIP runtime this process
// already uses (a host that imported PyTorch first runs on PyTorch's bundled runtime and must get
// the RCCL built against it; everybody else gets /opt/rocm's).
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;     // optional
  std::string err, path;
  bool load() {
    if (lib) return true;
    std::vector<std::string> cand;
    Dl_info info;
    // BLSQ_RCCL_PATH: this library and no other (a wrong path is an error, not a reason to look elsewhere)
    const char* forced = getenv("BLSQ_RCCL_PATH");
    if (forced && forced[0]) cand.push_back(forced);
    else if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t k = dir.rfind('/');
      if (k != std::string::npos) {
        dir.resize(k);
        cand.push_back(dir + "/librccl.so.1");
        cand.push_back(dir + "/librccl.so");
      }
    }
    if (!(forced && forced[0])) {
      cand.push_back("librccl.so.1");
      cand.push_back("/opt/rocm/lib/librccl.so.1");
    }
    for (const auto& c : cand) {
      lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (lib) { path = c; break; }
    }
    if (!lib) { err = std::string("dlopen(librccl): ") + dlerror(); return false; }
#define BLSQ_RCCL_SYM(name)                                                 \
    name = reinterpret_cast<decltype(name)>(dlsym(lib, "nccl" #name));        \
    if (!name) { err = "librccl lacks nccl" #name; dlclose(lib); lib = nullptr; return false; }
    BLSQ_RCCL_SYM(GetUniqueId) BLSQ_RCCL_SYM(CommInitRank) BLSQ_RCCL_SYM(CommDestroy)
    BLSQ_RCCL_SYM(AllGather) BLSQ_RCCL_SYM(AllReduce) BLSQ_RCCL_SYM(GetErrorString)
#undef BLSQ_RCCL_SYM
    GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(lib, "ncclGetVersion"));
    {                                                  // the resolved file, not the name it was asked by
      Dl_info li;
      if (dladdr((void*)GetUniqueId, &li) && li.dli_fname) path = li.dli_fname;
    }
    return true;
  }
};
extern Rccl g_rccl;                             // (blsq_ctx.hip)
constexpr int RCCL_ERR_BASE = 10000;           // return code of a failed RCCL call: 10000 + ncclResult_t

int rccl_fail(blsq_ctx* ctx, ncclResult_t r, const char* where);   // (blsq_ctx.hip)
#define RCCLCHK(ctx, call)                                           \
  do {                                                               \
    ncclResult_t r__ = (call);                                       \
    if (r__ != ncclSuccess) return rccl_fail((ctx), r__, #call);     \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t alloc(size_t b) {
    bytes = b;
    if (b == 0) return hipSuccess;
    return hipMalloc(&p, b);
  }
  void release() { if (p) hipFree(p); p = nullptr; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// One level of the TSQR tree: nleaf workgroups per problem.
struct Level {
  int rowsA, rows_per_leaf, nleaf, RP, LDP;
  DevBuf R;                         // [B][nleaf][NPAD*NPAD]
};

struct QrTree {
  int B = 0, m = 0, n = 0, N = 0, NPAD = 0, NP = 0;
  const blsq::Options* opt = nullptr;      // the ctx's switches (set by build)
  std::vector<Level> levels;        // levels.back().nleaf == 1
  DevBuf V, T;                      // scratch shared by all QR launches of the plan
  // normal-equations fast path (gram_kernels.hip, chol_kernels.hip); problems that fail its gate use the levels
  bool gram = false;
  int gram_nchunk = 1;
  DevBuf gram_part, gram_dsc, gram_ints;   // partial Grams, column scales, [B] fallback mask + count
  DevBuf gram_keep;                 // [B][NPAD*NPAD] the Grams themselves (kept: the trust-region
                                    // systems are diagonal modifications of them)
  DevBuf gram_rinv, gram_ywork, gram_k2;   // conditioning certificate: inverse diagonal tiles, Y = R'^-T, bound [B]
  DevBuf gram_cert;                        // [B] ints: 1 = proven inside the factor kernel (N <= 80)
  DevBuf gram_cflag, gram_ctau;            // [B] certificate stage 3: problems left to the shifted factorisation, their shifts
  double k2_max = 0.0;                     // the gate for this plan's row count (gram_k2_max)
  // CholeskyQR2 middle tier (cqr2_kernels.hip): buffers allocated on first use
  bool cqr2 = false;
  DevBuf cq_W, cq_Wf, cq_G2, cq_R1, cq_R2, cq_z, cq_ints;
  size_t cq_cap = 0;                       // listed problems cq_W / cq_Wf hold (high-water mark of the rejected list)
  bool fb_zeroed = false;                  // the gate counters were cleared by pack_vecs_kernel of this factor call
  // per-problem path of the CURRENT triangles: gram_path()[b] = n + 1 (Householder tree) or 0 (Gram).
  // any_gram / any_qr: whether a problem of either kind can exist (host-side upper bounds)
  bool any_gram = false, any_qr = true;
  bool path_valid = false;          // gram_path() describes the current triangles
  const int* gram_path() const { return gram ? gram_ints.as<int>() + B + 4 : nullptr; }

  // rows: source rows per problem at level 0
  int build(blsq_ctx* ctx, int B_, int rows, int n_, size_t extra_rp_rows) {
    B = B_; m = rows; n = n_; opt = &ctx->opt;
    N = n + 1; NPAD = round_up(N, 16); NP = NPAD / 16;
    if (NPAD > RMAX) return ctx->bad(4, "n too large (n + 1 must be <= 1024)");
    int cur_rows = rows;
    bool first = true;
    size_t max_slot_rows = extra_rp_rows;   // max over launches of nslot*RP
    size_t max_slots = (size_t)B;
    while (true) {
      Level L;
      L.rowsA = cur_rows;
      if (first) {
        L.nleaf = std::max(1, (cur_rows + RMAX - 1) / RMAX);
        if (L.nleaf > 1 && !merge_fits(n))
          return ctx->bad(4, "m > 1024 needs n <= 512 (TSQR merge capacity)");
        L.rows_per_leaf = round_up((cur_rows + L.nleaf - 1) / L.nleaf, 16);
        if (L.rows_per_leaf < NPAD && L.nleaf > 1) L.rows_per_leaf = NPAD;
        L.nleaf = std::max(1, (cur_rows + L.rows_per_leaf - 1) / L.rows_per_leaf);
      } else {
        const int G = merge_group(n);   // triangles merged per workgroup (>= 2)
        L.rows_per_leaf = G * NPAD;
        L.nleaf = (cur_rows + L.rows_per_leaf - 1) / L.rows_per_leaf;
      }
      L.RP = std::max(round_up(std::min(L.rows_per_leaf, std::max(cur_rows, 1)), 16), NPAD);
      if (qr_staged_tiles(L.RP, first ? 0 : NPAD, N) > QR_MAX_TILES)
        return ctx->bad(3, "leaf does not fit LDS");
      L.LDP = 0;
      hipError_t e = L.R.alloc(sizeof(double) * (size_t)B * L.nleaf * NPAD * NPAD);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(R level)");
      max_slot_rows = std::max(max_slot_rows, (size_t)B * L.nleaf * L.RP);
      max_slots = std::max(max_slots, (size_t)B * L.nleaf);
      levels.push_back(L);
      if (L.nleaf == 1) break;
      cur_rows = L.nleaf * NPAD;
      first = false;
    }
    hipError_t e = V.alloc(sizeof(double) * max_slot_rows * NP * 16);
    if (e != hipSuccess) return ctx->fail(e, "hipMalloc(V scratch)");
    e = T.alloc(sizeof(double) * max_slots * NP * 256);
    if (e != hipSuccess) return ctx->fail(e, "hipMalloc(T scratch)");
    gram = gram_supported(rows, n) && ctx->opt.on(OPT_GRAM);
    if (gram) {
      gram_nchunk = gram_chunks(B, rows);
      if (gram_nchunk > 1) {
        e = gram_part.alloc(sizeof(double) * (size_t)B * gram_nchunk * NPAD * NPAD);
        if (e != hipSuccess) return ctx->fail(e, "hipMalloc(partial Grams)");
      }
      e = gram_dsc.alloc(sizeof(double) * (size_t)B * NPAD);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Gram scales)");
      e = gram_keep.alloc(sizeof(double) * (size_t)B * NPAD * NPAD);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Grams)");
      e = gram_rinv.alloc(sizeof(double) * (size_t)B * NP * 256);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Gram tile inverses)");
      e = gram_ywork.alloc(sizeof(double) * (size_t)B * NPAD * NPAD);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Gram gate work)");
      e = gram_k2.alloc(sizeof(double) * (size_t)B);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Gram gate bound)");
      e = gram_cert.alloc(sizeof(int) * (size_t)B);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Gram certificate flags)");
      e = gram_cflag.alloc(sizeof(int) * (size_t)B);
      if (e == hipSuccess) e = gram_ctau.alloc(sizeof(double) * (size_t)B);
      if (e == hipSuccess) e = hipMemsetAsync(gram_cflag.p, 0, gram_cflag.bytes, ctx->stream);
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(certificate stage 3)");
      k2_max = gram_k2_max(rows, ctx->opt.d(OPT_GRAM_K2_MAX));
      cqr2 = cqr2_supported(rows, n) && ctx->opt.on(OPT_CQR2);
      e = hipMemsetAsync(gram_cert.p, 0, gram_cert.bytes, ctx->stream);
      if (e != hipSuccess) return ctx->fail(e, "hipMemsetAsync(Gram certificate flags)");
      e = hipMemsetAsync(gram_k2.p, 0, gram_k2.bytes, ctx->stream);
      if (e != hipSuccess) return ctx->fail(e, "hipMemsetAsync(Gram gate bound)");
      e = gram_ints.alloc(sizeof(int) * (3 * (size_t)B + 4));     // launch mask, count, path, fallback list
      if (e != hipSuccess) return ctx->fail(e, "hipMalloc(Gram mask)");
      e = hipMemsetAsync(gram_keep.p, 0, gram_keep.bytes, ctx->stream);      // (lower tiles are never written)
      if (e != hipSuccess) return ctx->fail(e, "hipMemsetAsync(Grams)");
      e = hipMemsetAsync(gram_ints.p, 0xFF, gram_ints.bytes, ctx->stream);   // path: all QR until factored
      if (e != hipSuccess) return ctx->fail(e, "hipMemsetAsync(Gram mask)");
    }
    return 0;
  }
  void release() {
    for (auto& L : levels) L.R.release();
    V.release(); T.release();
    gram_part.release(); gram_dsc.release(); gram_ints.release(); gram_keep.release();
    gram_rinv.release(); gram_ywork.release(); gram_k2.release(); gram_cert.release();
    gram_cflag.release(); gram_ctau.release();
    cq_W.release(); cq_Wf.release(); cq_G2.release(); cq_R1.release(); cq_R2.release(); cq_z.release(); cq_ints.release();
  }
  // [J f] -> triangle by the normal equations where the conditioning gate allows it.
  // Returns the number of problems left for the Householder tree in *nfallback; their indices
  // are flagged in the fallback mask (n + 1 / 0 per problem).
  // `collective`: the rows of the problem are split over the ranks of ctx->comm — the local Grams
  // are summed over the ranks (ONE ncclAllReduce on the ctx stream) before the factorisation, which
  // is then replicated: every rank holds the same bits, so every rank takes the same gate decision.
  int run_gram(blsq_ctx* ctx, const double* dJ, const double* df, int ldJ, const int* mask,
               int* nfallback, bool collective) {
    int* fb = gram_ints.as<int>();
    int* cnt = fb + B;
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, sizeof(int), ctx->stream));
    double* Rf = levels.back().R.as<double>();
    GramArgs g{};
    g.opt = opt;
    g.J = dJ; g.strideJ = (long)m * ldJ; g.ldJ = ldJ; g.F = df; g.strideF = m;
    g.m = m; g.n = n; g.NPAD = NPAD; g.mask = mask;
    double* Gk = gram_keep.as<double>();
    g.G = gram_nchunk > 1 ? gram_part.as<double>() : Gk;
    ctx->begin(K_GRAM);
    bool fused = false;
    hipError_t e = launch_gram(g, gram_nchunk, B, ctx->stream, Gk, &fused);
    if (e == hipSuccess && gram_nchunk > 1 && !fused)
      e = launch_gram_reduce(gram_part.as<double>(), gram_nchunk, NPAD, Gk, mask, B, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram");
    if (collective && ctx->comm && ctx->comm_ranks > 1)
      RCCLCHK(ctx, g_rccl.AllReduce(Gk, Gk, (size_t)B * NPAD * NPAD, ncclDouble, ncclSum, ctx->comm,
                                    ctx->stream));
    GramCholArgs c{};
    c.opt = opt;
    c.Gsrc = Gk; c.G = Rf; c.NPAD = NPAD; c.n = n; c.mask = mask; c.fb_mask = fb; c.fail_count = cnt;
    c.path_out = fb + B + 4;
    c.dsc = gram_dsc.as<double>();
    c.rinv = gram_rinv.as<double>(); c.ywork = gram_ywork.as<double>(); c.k2_out = gram_k2.as<double>();
    c.k2_max = k2_max; c.pivot_floor = 1.0 / k2_max;
    c.cert_flag = gram_cflag.as<int>(); c.cert_tau = gram_ctau.as<double>();
    ctx->begin(K_GRAM_CHOL);
    e = launch_gram_chol(c, B, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram_chol");
    ctx->begin(K_GRAM_GATE);
    e = launch_gram_gate(c, B, ctx->stream);
    if (e == hipSuccess) e = launch_gram_cert_shift(c, B, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram_gate");
    HIPCHK(ctx, hipMemcpyAsync(ctx->pinned + 1, cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nfallback = ctx->pinned[1];
    return 0;
  }
  // Gram front end ONLY: G = [J f]^T [J f] into gram_keep (+ the cross-rank sum); nothing is factored.
  // (k0, nb): problems k0 .. k0 + nb - 1 only (the host-pointer API feeds the Grams in sub-batches behind
  // the copies; the result does not depend on the split — a problem's chunks and their order are functions of m)
  int run_gram_only(blsq_ctx* ctx, const double* dJ, const double* df, int ldJ, const int* mask,
                    bool collective, int k0 = 0, int nb = -1) {
    if (nb < 0) nb = B;
    const size_t tri = (size_t)NPAD * NPAD;
    GramArgs g{};
    g.opt = opt;
    g.J = dJ + (size_t)k0 * m * ldJ; g.strideJ = (long)m * ldJ; g.ldJ = ldJ; g.F = df + (size_t)k0 * m; g.strideF = m;
    g.m = m; g.n = n; g.NPAD = NPAD; g.mask = mask ? mask + k0 : nullptr;
    double* Gk = gram_keep.as<double>() + (size_t)k0 * tri;
    double* Gp = gram_nchunk > 1 ? gram_part.as<double>() + (size_t)k0 * gram_nchunk * tri : nullptr;
    g.G = gram_nchunk > 1 ? Gp : Gk;
    ctx->begin(K_GRAM);
    bool fused = false;
    hipError_t e = launch_gram(g, gram_nchunk, nb, ctx->stream, Gk, &fused);
    if (e == hipSuccess && gram_nchunk > 1 && !fused)
      e = launch_gram_reduce(Gp, gram_nchunk, NPAD, Gk, g.mask, nb, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram");
    if (collective && ctx->comm && ctx->comm_ranks > 1)
      RCCLCHK(ctx, g_rccl.AllReduce(Gk, Gk, (size_t)nb * NPAD * NPAD, ncclDouble, ncclSum, ctx->comm,
                                    ctx->stream));
    return 0;
  }
  // Householder TSQR tree only (problems selected by ncols_mask; nullptr: all)
  // list / count (optional): compacted indices of the selected problems — a masked launch whose
  // active workgroups alternate with idle ones lands on a fraction of the XCDs
  int run_levels(blsq_ctx* ctx, const double* dJ, const double* df, int ldJ, const int* ncols_mask,
                 const int* list = nullptr, int count = 0) {
    for (size_t l = 0; l < levels.size(); ++l) {
      const Level& L = levels[l];
      QrArgs q = base_args();
      q.ncols_dev = ncols_mask;
      q.batch_list = list;
      if (l == 0) {
        q.A = dJ; q.strideA = (long)m * ldJ; q.ldA = ldJ; q.rowsA = m;
        q.F = df; q.strideF = m;
      } else {
        const Level& Pv = levels[l - 1];
        q.A = Pv.R.as<double>(); q.strideA = (long)Pv.nleaf * NPAD * NPAD;
        q.ldA = NPAD; q.rowsA = Pv.nleaf * NPAD; q.F = nullptr; q.strideF = 0;
        q.stack_rows = NPAD;
      }
      q.rows_per_leaf = L.rows_per_leaf; q.RP = L.RP; q.LDP = L.LDP;
      q.Rout = L.R.as<double>();
      ctx->begin(l == 0 ? K_QR_LEAF : K_QR_MERGE);
      hipError_t e = launch_qr(q, L.nleaf, list ? count : B, ctx->stream);
      ctx->end();
      if (e != hipSuccess) return ctx->fail(e, "launch_qr");
    }
    return 0;
  }
  // The problems the certificate rejected (fb_list(), nfb of them; fb_mask() = n + 1 for each): a triangle of
  // [J f] of Householder quality into their Rfinal slots — by CholeskyQR2 where its acceptance test passes
  // (second pass over J through the MFMA pipe, cqr2_kernels.hip), by the Householder TSQR tree for the rest.
  int run_fallback(blsq_ctx* ctx, const double* dJ, const double* df, int ldJ, int nfb) {
    if (!cqr2 || !gram) return run_levels(ctx, dJ, df, ldJ, fb_mask(), fb_list(), nfb);
    hipError_t e = hipSuccess;
    // W = J R1^-1 and w_f for the LISTED problems only (list position, not problem index): sized by the high-water mark
    // of the list, grown geometrically — a single rejected problem of a 512-problem batch costs 8 MB, not 4.3 GB.
    if ((size_t)nfb > cq_cap) {
      const size_t cap = std::min<size_t>((size_t)B, std::max<size_t>((size_t)nfb, 2 * cq_cap));
      cq_W.release(); cq_Wf.release();
      e = cq_W.alloc(sizeof(double) * cap * m * n);
      if (e == hipSuccess) e = cq_Wf.alloc(sizeof(double) * cap * m);
      if (e != hipSuccess) {                              // no room for the second pass: the tree does it all
        cq_W.release(); cq_Wf.release(); cq_cap = 0;
        (void)hipGetLastError();
        return run_levels(ctx, dJ, df, ldJ, fb_mask(), fb_list(), nfb);
      }
      cq_cap = cap;
    }
    if (!cq_G2.p) {
      e = cq_G2.alloc(sizeof(double) * (size_t)B * NPAD * NPAD);
      if (e == hipSuccess) e = cq_R2.alloc(sizeof(double) * (size_t)B * NPAD * NPAD);
      if (e == hipSuccess) e = cq_R1.alloc(sizeof(double) * (size_t)B * NPAD * NPAD);
      if (e == hipSuccess) e = hipMemsetAsync(cq_R1.p, 0, cq_R1.bytes, ctx->stream);
      if (e == hipSuccess) e = cq_z.alloc(sizeof(double) * (size_t)B * NPAD);
      if (e == hipSuccess) e = cq_ints.alloc(sizeof(int) * (4 * (size_t)B + 4));
      if (e == hipSuccess) e = hipMemsetAsync(cq_G2.p, 0, cq_G2.bytes, ctx->stream);   // (lower tiles are never written)
      if (e == hipSuccess) e = hipMemsetAsync(cq_R2.p, 0, cq_R2.bytes, ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(cq_ints.p, 0, cq_ints.bytes, ctx->stream);
      if (e != hipSuccess) {                              // no room for the second pass: the tree does it all
        cq_G2.release(); cq_R1.release(); cq_R2.release(); cq_z.release(); cq_ints.release();
        cqr2 = false;
        (void)hipGetLastError();
        return run_levels(ctx, dJ, df, ldJ, fb_mask(), fb_list(), nfb);
      }
    }
    int* piv1 = cq_ints.as<int>();
    int* runm = piv1 + B;
    int* piv2 = piv1 + 2 * (size_t)B;
    int* tmask = piv1 + 3 * (size_t)B;
    int* cnt = piv1 + 4 * (size_t)B;
    double* Rf = levels.back().R.as<double>();
    double* R1 = cq_R1.as<double>();
    // 1. R1 | c = chol of the plain Gram (listed problems) into scratch, its tile inverses and scales
    GramCholArgs c{};
    c.opt = opt;
    c.Gsrc = gram_keep.as<double>(); c.G = R1; c.NPAD = NPAD; c.n = n; c.skip_zero = 1;
    c.batch_list = fb_list(); c.fb_mask = piv1; c.fail_count = cnt;
    c.dsc = gram_dsc.as<double>(); c.rinv = gram_rinv.as<double>(); c.ywork = gram_ywork.as<double>();
    c.k2_max = 1e300; c.pivot_floor = 1e-14;
    ctx->begin(K_GRAM_CHOL);
    e = launch_gram_chol(c, nfb, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram_chol(cqr2 first factor)");
    // 2. Y = R1'^-T by the certificate's kernel, which also bounds kappa_2 of the equilibrated plain Gram: the
    //    second pass multiplies by the EXPLICIT inverse, whose error enters the triangle as eps kappa(J) (measured:
    //    step error 2e-18 kappa, tools/cqr2_check.py), so the tier takes a problem only if that PROVEN bound is
    //    below CQR2_K2_MAX = 1e12 (kappa(J D) <= 1e6: error <= 2e-12); beyond, the Householder tree.
    GramCholArgs cy = c;
    cy.batch_list = nullptr; cy.mask = fb_mask(); cy.k2_max = CQR2_K2_MAX;
    // (the bound on the PLAIN equilibrated Gram also bounds the augmented system's — its spectrum lies inside,
    //  chol_kernels.hip — so it replaces the missing / larger bound of a rejected problem: the rank gate uses it)
    cy.k2_out = gram_k2.as<double>();
    ctx->begin(K_GRAM_GATE);
    e = launch_gram_gate(cy, B, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram_gate(cqr2 inverse)");
    // 3. z = R^-1 c, launch mask;  4. W = J R^-1, w_f = f - J z
    Cqr2Args q{};
    q.J = dJ; q.strideJ = (long)m * ldJ; q.ldJ = ldJ; q.F = df; q.strideF = m;
    q.m = m; q.n = n; q.NPAD = NPAD; q.list = fb_list(); q.run = runm;
    q.Y = gram_ywork.as<double>(); q.dsc = gram_dsc.as<double>(); q.R1 = R1; q.z = cq_z.as<double>();
    q.Wj = cq_W.as<double>(); q.strideW = (long)m * n; q.Wf = cq_Wf.as<double>(); q.strideWf = m;
    ctx->begin(K_CQR2_APPLY);
    e = launch_cqr2_prep(q, nfb, piv1, runm, ctx->stream);
    if (e == hipSuccess) e = launch_cqr2_apply(q, nfb, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_cqr2_apply");
    // 5. G2 = [W w_f]^T [W w_f]
    GramArgs g{};
    g.opt = opt;
    g.J = q.Wj; g.strideJ = q.strideW; g.ldJ = n; g.F = q.Wf; g.strideF = m;
    g.m = m; g.n = n; g.NPAD = NPAD; g.mask = runm; g.list = fb_list();   // (compacted: all XCDs)
    g.src_by_pos = 1;                                   // (W holds the listed problems only)
    double* G2 = cq_G2.as<double>();
    g.G = gram_nchunk > 1 ? gram_part.as<double>() : G2;
    ctx->begin(K_GRAM);
    bool fused = false;
    e = launch_gram(g, gram_nchunk, nfb, ctx->stream, G2, &fused);
    if (e == hipSuccess && gram_nchunk > 1 && !fused)
      e = launch_gram_reduce(gram_part.as<double>(), gram_nchunk, NPAD, G2, runm, B, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram(cqr2 second pass)");
    // 6. R2 | c2 = chol(G2)
    GramCholArgs c2{};
    c2.opt = opt;
    c2.Gsrc = G2; c2.G = cq_R2.as<double>(); c2.NPAD = NPAD; c2.n = n;
    c2.batch_list = fb_list(); c2.mask = runm; c2.fb_mask = piv2; c2.fail_count = cnt + 1;
    c2.k2_max = 1e300; c2.pivot_floor = 0.25;           // (G2 ~ I: a pivot below 1/2 means the first pass failed)
    ctx->begin(K_GRAM_CHOL);
    e = launch_gram_chol(c2, nfb, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_gram_chol(cqr2 second factor)");
    // 7. acceptance + R~ = R2 [R c; 0 1] into the triangle slot;  8. the tree for what is left
    ctx->begin(K_CQR2_COMBINE);
    e = launch_cqr2_combine(q, nfb, runm, piv2, G2, cq_R2.as<double>(), Rf, tmask, ctx->cq_accept_dev, ctx->stream);
    ctx->end();
    if (e != hipSuccess) return ctx->fail(e, "launch_cqr2_combine");
    return run_levels(ctx, dJ, df, ldJ, tmask, fb_list(), nfb);
  }
  int* fb_mask() const { return gram_ints.as<int>(); }
  int* fb_count() const { return gram_ints.as<int>() + B; }
  int* path_rw() const { return gram_ints.as<int>() + B + 4; }
  int* fb_list() const { return gram_ints.as<int>() + 2 * (size_t)B + 4; }
  // host bookkeeping after a gate verdict: nfb of the problems refreshed by this call failed
  void note_paths(blsq_ctx* ctx, int nfb, bool masked) {
    ctx->gram_fallback += nfb;
    ctx->gram_fast += B - nfb;              // (masked problems count as fast: diagnostics only)
    // a masked call refreshes some problems only: the others keep their earlier path
    if (!masked || !path_valid) { any_qr = nfb > 0 || masked; any_gram = nfb < B; }
    else { any_qr = any_qr || nfb > 0; any_gram = true; }
    path_valid = true;
  }
  const double* Rfinal() const { return levels.back().R.as<double>(); }

  QrArgs base_args() const {
    QrArgs q{};
    q.opt = opt;
    q.N = N; q.NPAD = NPAD; q.NPmax = NP;
    q.V = V.as<double>(); q.T = T.as<double>();
    return q;
  }
  // [J f] -> R~  (levels 0..end); first_level lets TSQR-combine skip level 0
  // ncols_mask (optional, device [B]): problems with an entry <= 1 are skipped — their
  // triangles of the previous run stay in place (outer driver: only fresh Jacobians are factored)
  int run(blsq_ctx* ctx, const double* dJ, const double* df, int ldJ,
          const int* ncols_mask = nullptr, bool collective = false) {
    if (gram && df != nullptr) {
      int nfb = 0;
      int rc = run_gram(ctx, dJ, df, ldJ, ncols_mask, &nfb, collective);
      if (rc) return rc;
      ctx->gram_fallback += nfb;
      ctx->gram_fast += B - nfb;            // (masked problems count as fast: diagnostics only)
      // a masked call refreshes some problems only: the others keep their earlier path
      if (ncols_mask == nullptr || !path_valid) { any_qr = nfb > 0 || ncols_mask != nullptr; any_gram = nfb < B; }
      else { any_qr = any_qr || nfb > 0; any_gram = true; }
      path_valid = true;
      if (nfb == 0) return 0;
      ncols_mask = gram_ints.as<int>();     // only the problems the gate rejected
    }
    else { any_gram = false; any_qr = true; path_valid = false; }
    for (size_t l = 0; l < levels.size(); ++l) {
      const Level& L = levels[l];
      QrArgs q = base_args();
      q.ncols_dev = ncols_mask;
      if (l == 0) {
        q.A = dJ; q.strideA = (long)m * ldJ; q.ldA = ldJ; q.rowsA = m;
        q.F = df; q.strideF = m;
      } else {
        const Level& Pv = levels[l - 1];
        q.A = Pv.R.as<double>(); q.strideA = (long)Pv.nleaf * NPAD * NPAD;
        q.ldA = NPAD; q.rowsA = Pv.nleaf * NPAD; q.F = nullptr; q.strideF = 0;
        q.stack_rows = NPAD;
      }
      q.rows_per_leaf = L.rows_per_leaf; q.RP = L.RP; q.LDP = L.LDP;
      q.Rout = L.R.as<double>();
      ctx->begin(l == 0 ? K_QR_LEAF : K_QR_MERGE);
      hipError_t e = launch_qr(q, L.nleaf, B, ctx->stream);
      ctx->end();
      if (e != hipSuccess) return ctx->fail(e, "launch_qr");
    }
    return 0;
  }
};

}  // namespace blsq_host
using namespace blsq_host;

// The optimistic verdict of a device-resident factor call — ONE state machine for the TRF and the dogbox plans
// (verdict_drop / verdict_arm / verdict_resolve below).  blsq_*_factor_dev does not wait for the gate's counters
// (problems that leave the normal-equations path, problems that need the SVD, problems the factor kernel did not
// settle itself): it assumes the common verdict — "none" — and the NEXT call on the plan checks, by which time the
// counters have long arrived.  blsq_*_step_dev enqueues its kernels first and checks afterwards; a wrong guess runs
// the repair (the next tier's factorisation, from the caller's J) and the step once more.
struct VerdictState {
  bool optimistic = true;           // option `optimistic` = 0 switches it off
  bool guess_ok = true;             // the last verdict of this plan was "all fast": only then is the next one guessed
  // Back-off: a plan whose guess keeps failing (inputs that change their conditioning class from call to call) stops
  // guessing for 2, 4, ... 32 factor calls after each wrong guess — a wrong guess costs the Newton rounds and the step
  // once more (about half a step-solve), a synchronous verdict ten microseconds of idle stream; a right guess takes
  // one level off again.
  int guess_streak = 0, guess_pause = 0;
  bool pending = false;
  // Second guess: every problem is settled inside the factor kernel / stage 0 of the certificate, so the certificate
  // and gate launches are not even enqueued; checked with the same read-back.
  bool guess_settled = false, pend_tail = false;
  int* pend_pin = nullptr;          // 4 pinned ints of this plan ([3]: sequence number of the publish)
  hipEvent_t pend_ev = nullptr;
  int pend_seq = 0;
  bool pend_unpub = false;          // the verdict's counters have not been sent yet: the step kernel of the next
                                    // step call stores them on its way in (or verdict_published() sends them now)
  // the caller's vectors of a device-resident factor call, copied into the state layout by the prep launch
  // (option `fuse_pack` = 0: by a pack_vecs launch in front of the Gram)
  bool pack_pend = false;
  PackVecs pack_pv{};
  const double* pend_dJ = nullptr; const double* pend_df = nullptr;
  int pend_ldJ = 0, pend_scale_mode = 0;
  double* pend_scale_io = nullptr;
};

struct blsq_trf_plan : VerdictState {
  blsq_ctx* ctx = nullptr;
  int B = 0, m = 0, n = 0, ld = 0;
  QrTree tree;
  // which kernels factor the augmented / Newton systems of the current triangles (per problem:
  // `path`, see trf_after_triangle)
  const int* path = nullptr;
  bool use_chol = false, use_qr = true;
  DevBuf aug_colinfo;               // [B][2] column-norm summary of R_aug (Gram-path problems)
  DevBuf aug_mask;                  // [B] launch mask of the stacked QR of [R D; E] (trf_aug_trivial_kernel)
  DevBuf aug_lam;                   // [B] proven bound on lambda_max of the equilibrated H (LmState::lam)
  DevBuf aug_ym, aug_r1;            // [B] the factor kernel's share of the certificate's stage 0 (GramCholArgs::cert_ym)
  DevBuf aug_open;                  // [B] stage 0's note for the problems it leaves open (GramCholArgs::cert_open)
  DevBuf aug_hmax;                  // [B] largest diagonal entry of H (LmState::hmax: which Newton systems of a
                                    // Householder-path problem may be factored from the Gram)
  bool gram_valid = false;          // tree.gram_keep holds the Grams of the current factor call's problems
  // CSNE tier (csne_kernels.hip): rejected problems whose steps are corrected against J in one streaming pass
  int last_scale_mode = 0;          // scale_mode of the last factor call (a problem that leaves the tier at step time is prepared again)
  int lm_rounds_done = 0;           // Newton rounds the last trf_lm_rounds call ran (the deepest recording: 1 + that)
  bool csne_on = false;             // the shape is supported and BLSQ_CSNE != 0
  int ncsne = 0;                    // problems on the tier now (host copy of cs.counts[0])
  DevBuf cs_ints;                   // flag [B], list [B], fail_list [B], ne [B], sel_mask [B], counts [4], scratch [4]
  DevBuf cs_pmin, cs_eta, cs_alpha, cs_hp, cs_vec, cs_part;
  DevBuf cs_k2;                     // [B] the bound on kappa_2 of the COMPUTED system (the certificate's own output, gram_k2, keeps its meaning)
  size_t cs_part_cap = 0;           // (list positions x chunks x NE) the partial-sum buffer holds
  CsneState cs{};
  // TSQR (multi-rank) extras
  int nranks = 1, m_total = 0;
  bool ranks_agreed = false;        // the ranks have compared their plan configuration (first factor call)
  DevBuf Rcomb;                     // [1][NPAD*NPAD] merged triangle
  DevBuf Rstack;                    // [nranks][NPAD*NPAD] gathered triangles (blsq_tsqr_factor_dev)
  // n-space state
  DevBuf X, vecs, scal2, sweeps;
  DevBuf o_vec, o_hits, o_act, o_scal, o_info;
  DevBuf in_J, in_f, in_vec, in_scal;   // staging for the host-pointer API
  TrfState st{};
  TrfStepOut out{};
  double* d_alpha_in = nullptr;
  int aug_RP = 0, aug_LDP = 0;
  // SVD-free trust-region path (lm_kernels.hip)
  DevBuf lm_Xa, lm_ints, lm_sc, lm_ph, lm_sa;
  LmState lm{};
  int lm_enable = 1;                // SVD-free trust-region path allowed at all (BLSQ_NO_SVDFREE)
  int lm_gate_mask = 3;             // launch_lm_gate: bit 0 Householder-path problems, bit 1 normal-equations-path problems
  bool gate_done = false;           // lm_gate already ran in this factor call (no problem left the normal-equations path)
  bool lm_counts_clean = false;     // the Newton-round counters are zero (left so by the last step kernel)
  // The triangle slots st.X hold zeros outside the factors as long as only the Cholesky kernels have
  // written them (zeroed at allocation); the stacked QR and the Jacobi SVD write there.  While clean, the
  // Cholesky of the augmented system does not store those zeros again (half of its bytes).
  bool x_dirty = true;
  int lm_expect0 = 0;               // problems the first Newton round of the last step call worked on (kernel choice hint)
  int lm_rounds_last = 12;          // Newton rounds that had work in the last step call (run-ahead only over those)
  int njac = -1;                    // problems it sent to the Jacobi SVD (-1: unknown)
};

struct blsq_dogbox_plan : VerdictState {
  blsq_ctx* ctx = nullptr;
  int B = 0, m = 0, n = 0, ld = 0;
  QrTree tree;
  DevBuf S, X, vecs, ivecs, scal2, sweeps, active, onb;
  DevBuf o_vec, o_onb, o_scal, o_info;
  DevBuf in_J, in_f, in_vec, in_scal;
  bool gate_done = false;           // as blsq_trf_plan
  int njac = -1;
  DevBuf gate_ints;                 // [3B] fast flags, Jacobi launch mask, finished-in-the-Cholesky-kernel flags
  DevBuf colinfo;                   // [B][2] column-norm summary of the free block (Gram-path problems)
  int svdfree_enable = 1;
  // CSNE tier (csne_kernels.hip): the Newton step of a rejected problem's free block corrected against J at factor time
  bool csne_on = false;
  int ncsne = 0;
  DevBuf cs_ints, cs_pmin, cs_eta, cs_alpha, cs_k2, cs_vec, cs_part;
  size_t cs_part_cap = 0;
  CsneState cs{};
  DogState st{};
  DogStepOut out{};
};

// ---- shared between the translation units -------------------------------------------------------
namespace blsq_host {
// copy a [B][n] caller vector into the [B][ld] state layout (device to device or host to device, by `kind`)
int put_vec(blsq_ctx* ctx, double* dst, int ld, const double* src, int n, int B, hipMemcpyKind kind);
template <class T>
int get_vec(blsq_ctx* ctx, T* dst, int n, const T* src, int ld, int B) {
  if (!dst) return 0;
  HIPCHK(ctx, hipMemcpy2DAsync(dst, sizeof(T) * n, src, sizeof(T) * ld, sizeof(T) * n, B,
                               hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}

// the deferred vectors of this factor call: handed to the prep launch (returns them), or — a masked call keeps the
// other problems' state, so its prep launch cannot do the copy — packed by the stand-alone launch right here
template <class Plan>
int take_pack(Plan* p, const int* mask, const PackVecs** pk) {
  *pk = nullptr;
  if (!p->pack_pend) return 0;
  p->pack_pend = false;
  if (!mask) { *pk = &p->pack_pv; return 0; }
  hipError_t e = launch_pack_vecs(p->pack_pv, p->n, p->ld, p->B, p->ctx->stream);
  if (e != hipSuccess) return p->ctx->fail(e, "launch_pack_vecs");
  return 0;
}

// The counters of a pending verdict are on their way to the host (a stand-alone publish unless a step kernel has
// taken them along) — to be called before anything waits for them or overwrites them.
template <class Plan>
int verdict_published(Plan* p) {
  if (!p->pend_unpub) return 0;
  p->pend_unpub = false;
  blsq_ctx* ctx = p->ctx;
  HIPCHK(ctx, ctx->publish(p->tree.fb_count(), 3, p->pend_pin, p->pend_ev, &p->pend_seq));
  return 0;
}
// ... and the arguments with which the step kernel of this call takes them along (dst == nullptr: nothing to do)
template <class Plan>
PublishArgs verdict_rides(Plan* p) {
  if (!p->pend_unpub) return PublishArgs{nullptr, 0, nullptr, 0};
  p->pend_unpub = false;
  p->pend_seq = ++p->ctx->pub_seq;
  return PublishArgs{p->tree.fb_count(), 3, p->pend_pin, p->pend_seq};
}

// prep from the Gram, Cholesky of H with the pivot gate, conditioning gate; *nfb = problems of this

// the back-off of a plan's guessing (VerdictState::guess_pause)
inline void verdict_wrong(VerdictState* p) {
  if (p->guess_streak < 5) ++p->guess_streak;
  p->guess_pause = 1 << p->guess_streak;
}
inline void verdict_right(VerdictState* p) { if (p->guess_streak > 0) --p->guess_streak; }
// may this factor call guess?  (consumes one call of a pause)
inline bool verdict_may_guess(VerdictState* p) {
  if (p->guess_pause > 0) { --p->guess_pause; return false; }
  return p->optimistic && p->guess_ok;
}

// has the factor kernel (N <= 80) / stage 0 of the certificate (TRF, N > 80) settled every problem of the call?
inline bool verdict_settled(const blsq_trf_plan* p) { return p->pend_pin[2] == 0; }
inline bool verdict_settled(const blsq_dogbox_plan* p) { return p->ld <= 80 && p->pend_pin[2] == 0; }

// A verdict nobody asked for belongs to a factor that is being overwritten (top of a factor call): no repair, but it is
// still read — the path statistics and the decision whether to guess again depend on it.
template <class Plan>
int verdict_drop(Plan* p) {
  if (!p->pending) return 0;
  blsq_ctx* ctx = p->ctx;
  p->pending = false;
  { int rc_ = verdict_published(p); if (rc_) return rc_; }
  HIPCHK(ctx, ctx->await(p->pend_pin, p->pend_ev, p->pend_seq));
  const int nfb_ = p->pend_pin[0], njac_ = p->pend_pin[1];
  if (p->pend_tail) { if (!verdict_settled(p)) p->guess_settled = false; }
  else if (nfb_ > 0 || njac_ > 0) {
    p->guess_ok = false;
    verdict_wrong(p);
    ctx->gram_fast -= nfb_; ctx->gram_fallback += nfb_;
  } else verdict_right(p);
  return 0;
}

// The counters of this factor call travel (or ride on the next step kernel); the verdict is read by verdict_resolve.
// skip_tail: the second guess — the gate launches were not enqueued.
template <class Plan>
int verdict_arm(Plan* p, bool skip_tail, const double* dJ, const double* df, int ldJ, int scale_mode) {
  blsq_ctx* ctx = p->ctx;
  p->pend_unpub = ctx->pub_direct() && ctx->pub_ride();
  if (!p->pend_unpub) HIPCHK(ctx, ctx->publish(p->tree.fb_count(), 3, p->pend_pin, p->pend_ev, &p->pend_seq));
  p->pending = true; p->pend_tail = skip_tail;
  p->pend_dJ = dJ; p->pend_df = df; p->pend_ldJ = ldJ; p->pend_scale_mode = scale_mode;
  p->gate_done = true;
  p->njac = 0;
  return 0;
}

// The verdict of an optimistic factor call.  *redo = false: nothing was pending, or the guess held.  *redo = true: it
// did not — after `repair` the state is what the synchronous path would have left, and whatever was computed from the
// guessed state must be computed again.  gate_tail(): the launches the second guess left out (certificate, rank gate);
// repair(nfb): the plan's own way from "nfb problems left the normal-equations path" to a finished factor state.
template <class Plan, class GateTail, class Repair>
int verdict_resolve(Plan* p, bool* redo, GateTail gate_tail, Repair repair) {
  if (redo) *redo = false;
  if (!p->pending) return 0;
  blsq_ctx* ctx = p->ctx;
  QrTree& t = p->tree;
  p->pending = false;
  { int rc_ = verdict_published(p); if (rc_) return rc_; }
  HIPCHK(ctx, ctx->await(p->pend_pin, p->pend_ev, p->pend_seq));
  int nfb = p->pend_pin[0], njac = p->pend_pin[1];
  const bool settled = verdict_settled(p);
  if (p->pend_tail) {
    if (settled) return 0;                  // (settled: certified and finished in the factor kernel — nfb = njac = 0)
    // wrong second guess: the launches that were left out, then the verdict as a synchronous call reads it
    p->guess_settled = false;
    if (redo) *redo = true;
    { int rc_ = gate_tail(); if (rc_) return rc_; }
    HIPCHK(ctx, hipMemcpyAsync(p->pend_pin, t.fb_count(), 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    nfb = p->pend_pin[0]; njac = p->pend_pin[1];
    if (nfb > 0 || njac > 0) { p->guess_ok = false; verdict_wrong(p); }
  } else {
    p->guess_settled = settled;
    if (nfb == 0 && njac == 0) { verdict_right(p); return 0; }
    p->guess_ok = false;
    verdict_wrong(p);
    if (redo) *redo = true;
  }
  ctx->gram_fast -= nfb; ctx->gram_fallback += nfb;      // (note_paths counted everybody as fast)
  t.any_qr = nfb > 0; t.any_gram = nfb < p->B; t.path_valid = true;
  p->gate_done = (nfb == 0);
  p->njac = p->gate_done ? njac : -1;
  { int rc_ = repair(nfb); if (rc_) return rc_; }
  if (p->pend_scale_mode != BLSQ_SCALE_GIVEN && p->pend_scale_io)
    HIPCHK(ctx, hipMemcpy2DAsync(p->pend_scale_io, sizeof(double) * p->n, p->st.scale, sizeof(double) * p->ld,
                                 sizeof(double) * p->n, p->B, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

// the whole factor call of a plan from device-resident [J f] (mask: outer driver, fresh Jacobians only), and the verdict
// of an optimistic one (blsq_trf.hip / blsq_dogbox.hip)
int trf_factor_core(blsq_trf_plan* p, const double* dJ, const double* df, int ldJ, int scale_mode,
                    const int* mask, bool may_defer = false, bool gram_done = false);
int trf_resolve(blsq_trf_plan* p, bool* redo);
int dog_factor_core(blsq_dogbox_plan* p, const double* dJ, const double* df, int ldJ, int scale_mode,
                    const int* mask, bool may_defer = false);
int dog_resolve(blsq_dogbox_plan* p, bool* redo);
// every verdict an optimistic factor call left pending on a plan of this ctx is read, and a wrong guess repaired
int ctx_resolve_pending(blsq_ctx* ctx);
}  // namespace blsq_host

