// The option table (blsq_options.h).
#include "blsq_options.h"

#include <cstdlib>
#include <cstring>

namespace blsq {

const OptInfo kOptTable[OPT_COUNT] = {
  {"gram", "BLSQ_GRAM", 1, "normal-equations front end (MFMA Gram, equilibrated Cholesky, conditioning certificate); 0: Householder TSQR tree for every problem"},
  {"cqr2", "BLSQ_CQR2", 1, "CholeskyQR2 tier for problems the certificate rejects (80 <= n <= 256); 0: Householder tree"},
  {"csne", "BLSQ_CSNE", 1, "CSNE tier for rejected TRF problems (80 <= n <= 256, single rank): steps corrected against J in one streaming pass; 0: CholeskyQR2 / tree"},
  {"optimistic", "BLSQ_OPTIMISTIC", 1, "*_factor_dev returns without waiting for the certificate's verdict (read by the next call); 0: synchronous verdict"},
  {"no_svdfree", "BLSQ_NO_SVDFREE", 0, "1: every trust-region sub-problem through the Jacobi SVD (no SVD-free Newton rounds)"},
  {"svdfree_min_n", "BLSQ_SVDFREE_MIN_N", 48, "Householder-path problems with 16 < n < value take the Jacobi SVD (measured crossover); 0: no band"},
  {"gram_k2_max", "BLSQ_GRAM_K2_MAX", 0, "> 0: the certificate's gate, if TIGHTER than gram_k2_max(m) = 2.5e5 min(1, a(4096) / a(m)); a larger value is ignored"},
  {"publish", "BLSQ_PUBLISH", 1, "device counters reach the host by one fenced store into a pinned slot the host polls; 0: hipMemcpyAsync + event"},
  {"fuse_pack", "BLSQ_FUSE_PACK", 1, "the caller's x / lb / ub / scale are copied into the state layout by the prep launch; 0: a pack launch of their own"},
  {"publish_ride", "BLSQ_PUBLISH_RIDE", 1, "the verdict's counters ride on the next step kernel, a round's count on its update kernel; 0: publishing launches"},
  {"cert0", "BLSQ_CERT0", 1, "stage 0 of the certificate (comparison-matrix bound, two triangular solves); 0: explicit inverse for every problem"},
  {"cert_direct", "BLSQ_CERT_DIRECT", 1, "a pure-Jacobian system stage 0 leaves open goes to the shifted factorisation directly; 0: through the norm stage"},
  {"settle0", "BLSQ_SETTLE0", 1, "N > 80: after a call in which stage 0 settled every problem the empty gate launches are left out; 0: always enqueued"},
  {"lm_chol_qrpath", "BLSQ_LM_CHOL_QRPATH", 1, "Newton systems of Householder-path problems factored from the Gram where alpha makes them provably well conditioned; 0: stacked QR"},
  {"lm_fused", "BLSQ_LM_FUSED", 1, "N <= 80: Gauss-Newton step, bracket and all Newton rounds of a problem in one launch; 0: lock-step rounds"},
  {"h2d_pipe", "BLSQ_H2D_PIPE", -1, "host-pointer API: [J f] copied in sub-batches under the Grams: -1 for page-locked sources only, 0 never, 1 always"},
  {"chol_reg", "BLSQ_CHOL_REG", 1, "N <= 80: register-resident right-looking Cholesky (one wave per problem); 0: the left-looking one-wave kernel"},
  {"chol_rl", "BLSQ_CHOL_RL", -1, "N > 80: -1 / 1 right-looking Cholesky, 0 left-looking (bit-identical)"},
  {"chol_rl2", "BLSQ_CHOL_RL2", 1, "N > 80, right-looking: the flag-driven kernel; 0: the barrier-synchronous one (bit-identical)"},
  {"gram16", "BLSQ_GRAM16", 1, "n = 241 .. 256: Gram kernel with static tile rows per wave; 0: the generic kernel (bit-identical)"},
  {"gram8", "BLSQ_GRAM8", 1, "n = 113 .. 128: k-split static-tile Gram kernel; 0: the generic kernel (a different summation order)"},
  {"gram_pair", "BLSQ_GRAM_PAIR", 1, "two row chunks of a problem summed by one workgroup; 0: partial Grams + reduction pass (bit-identical)"},
  {"gram_tile_groups", "BLSQ_GRAM_TILE_GROUPS", 0, "> 0: the tiles of a row chunk split over this many workgroups (bit-identical); 0: by the launch size"},
  {"gram_direct_nw", "BLSQ_GRAM_DIRECT_NW", 0, "narrow problems: 2 | 4 | 8 waves per workgroup of the direct Gram kernel; 0: by the row count"},
  {"gram_direct_max_nt", "BLSQ_GRAM_DIRECT_MAX_NT", 4, "narrow problems: the direct (no LDS) Gram kernel up to this many column tiles; 0: never"},
  {"qr_cqr", "BLSQ_QR_CQR", 1, "Householder tree: panels by Cholesky-QR + Householder reconstruction where the pivot test allows; 0: column loop"},
  {"gram1", "BLSQ_GRAM1", 1, "at most four (row chunk, problem) pairs, n a multiple of 16 in 80 .. 256: Gram with one tile per wave, operands straight from global memory; 0: tile groups through LDS (bit-identical)"},
  {"csne_mfma", "BLSQ_CSNE_MFMA", 1, "CSNE tier, TRF: the pass over J computes its dot products as 16-row MFMA tiles; 0: the vector-ALU kernel (a different summation order)"},
};

Options options_default() {
  Options o;
  for (int k = 0; k < OPT_COUNT; ++k) o.v[k] = kOptTable[k].dflt;
  return o;
}

Options options_from_env() {
  Options o = options_default();
  for (int k = 0; k < OPT_COUNT; ++k) {
    const char* e = getenv(kOptTable[k].env);
    if (e && e[0]) o.v[k] = atof(e);
  }
  return o;
}

const Options& options_or_default(const Options* opt) {
  static const Options dflt = options_default();
  return opt ? *opt : dflt;
}

int option_index(const char* name) {
  if (!name) return -1;
  for (int k = 0; k < OPT_COUNT; ++k)
    if (strcmp(name, kOptTable[k].name) == 0 || strcmp(name, kOptTable[k].env) == 0) return k;
  return -1;
}

}  // namespace blsq
