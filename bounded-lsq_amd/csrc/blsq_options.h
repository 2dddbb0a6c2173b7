// The library's run-time switches: ONE table (blsq_options.cpp), read ONCE per ctx — at blsq_ctx_create, from the
// environment variable named in the table — and afterwards changed only through blsq_ctx_set_option, per ctx
// (include/blsq.h; INTEGRATION.md lists them).  Route switches are consulted when a PLAN is created, launch switches at
// every call of a plan; none of them changes a result beyond the tolerance of the path it selects (tests compare the
// routes, most of them bit for bit).  BLSQ_RCCL_PATH is not in the table: which collective library the PROCESS loads is
// resolved once, by the first blsq_comm_* call.
#pragma once

namespace blsq {

enum Opt {
  // ---- routes (read at plan creation)
  OPT_GRAM = 0,            // normal-equations front end (Gram + Cholesky + certificate); 0: Householder tree for all
  OPT_CQR2,                // CholeskyQR2 tier for rejected problems; 0: the tree
  OPT_CSNE,                // CSNE tier for rejected problems (TRF, 80 <= n <= 256); 0: CholeskyQR2 / the tree
  OPT_OPTIMISTIC,          // *_factor_dev does not wait for the certificate's verdict; 0: it does
  OPT_NO_SVDFREE,          // 1: every trust-region sub-problem through the Jacobi SVD
  OPT_SVDFREE_MIN_N,       // Householder-path problems with 16 < n < this use the Jacobi SVD (0: no band)
  OPT_GRAM_K2_MAX,         // > 0: a TIGHTER gate than gram_k2_max(m) (it can only tighten)
  // ---- per call
  OPT_PUBLISH,             // device counters to the host by a publishing store; 0: hipMemcpyAsync + event
  OPT_FUSE_PACK,           // the caller's vectors packed by the prep launch; 0: a pack launch of their own
  OPT_PUBLISH_RIDE,        // counters ride on a kernel that runs anyway; 0: a publishing launch of their own
  OPT_CERT0,               // stage 0 of the certificate (comparison-matrix bound); 0: the norm stage for every problem
  OPT_CERT_DIRECT,         // open pure-Jacobian systems go to the shifted factorisation directly; 0: through the norm stage
  OPT_SETTLE0,             // second guess for N > 80 (empty gate launches left out); 0: always enqueued
  OPT_LM_CHOL_QRPATH,      // Newton systems of Householder-path problems by Cholesky where alpha allows; 0: stacked QR
  OPT_LM_FUSED,            // N <= 80: all Newton rounds of a problem in one launch; 0: lock-step rounds
  OPT_H2D_PIPE,            // host-pointer API: -1 sub-batched copies for page-locked sources only, 0 never, 1 always
  // ---- kernel selection (same bits whatever the choice unless noted)
  OPT_CHOL_REG,            // N <= 80: register-resident right-looking Cholesky; 0: left-looking one-wave kernel
  OPT_CHOL_RL,             // N > 80: -1 right-looking (default), 0 left-looking, 1 right-looking
  OPT_CHOL_RL2,            // N > 80 right-looking: flag-driven kernel; 0: barrier-synchronous one
  OPT_GRAM16,              // 16 column tiles: static tile rows per wave; 0: the generic Gram kernel
  OPT_GRAM8,               // 8 column tiles: k-split static-tile kernel; 0: generic (ANOTHER summation order)
  OPT_GRAM_PAIR,           // two row chunks summed inside the kernel; 0: reduction pass
  OPT_GRAM_TILE_GROUPS,    // > 0: split the tiles of a row chunk over this many workgroups (0: by launch size)
  OPT_GRAM_DIRECT_NW,      // narrow problems: waves per workgroup of the direct kernel (0: by the row count)
  OPT_GRAM_DIRECT_MAX_NT,  // narrow problems: the direct kernel up to this many column tiles (0: never)
  OPT_QR_CQR,              // Householder panels by Cholesky-QR + reconstruction where the pivot test allows; 0: column loop
  OPT_GRAM1,               // a handful of problems: one Gram tile per wave straight from global memory; 0: tile groups
  OPT_CSNE_MFMA,           // CSNE pass of TRF plans with its dot products on the FP64 MFMA pipe; 0: the vector-ALU kernel
  OPT_COUNT
};

struct OptInfo {
  const char* name;        // blsq_ctx_set_option's key
  const char* env;         // environment variable read at blsq_ctx_create
  double dflt;
  const char* doc;
};
extern const OptInfo kOptTable[OPT_COUNT];

struct Options {
  double v[OPT_COUNT];
  int i(Opt k) const { return (int)v[k]; }
  bool on(Opt k) const { return v[k] != 0.0; }
  double d(Opt k) const { return v[k]; }
};
// the defaults of the table / the defaults overridden by the environment
Options options_default();
Options options_from_env();
// the switches of a launch: `opt` of its argument structure, or the defaults when the caller gave none
const Options& options_or_default(const Options* opt);
int option_index(const char* name);     // -1: unknown

}  // namespace blsq
