// Factorisation side of the normal-equations path (gram_kernels.hip computes the Grams):
//
//   gram_chol_kernel<NWP>, gram_chol_rl_kernel, gram_chol_reg_kernel
//                           equilibrated blocked Cholesky of D G D + E^2 (+ alpha I) or of a gathered
//                           principal sub-matrix, straight from the kept Gram (chol16.h: the 16 x 16 chain)
//   lm_rounds_reg_kernel    N <= 80: the whole trust-region sub-problem after the factor in one launch
//   gram_cond_kernel<NWP>   the conditioning certificate (a PROVEN bound on kappa_2)
#include "gram_common.h"
#include "tri_ops.h"
#include "lm_body.h"
#include "chol16.h"

namespace blsq {

// Diagnostic build only (-DBLSQ_CHOL_STAMPS): wall-clock stamps (100 MHz) of the phases of every row block, taken
// by lane 0 of two waves of ONE problem (launch index CHOL_STAMP_PIDX).  Never enabled in the product.
#ifdef BLSQ_CHOL_STAMPS
__device__ long long g_chol_st[4][20][8];
#define CST(cond, p, kb, i) do { if ((cond) && lane == 0) g_chol_st[p][kb][i] = (long long)wall_clock64(); } while (0)
#else
#define CST(cond, p, kb, i) do { } while (0)
#endif

// column-norm summary of the first n columns by ONE wave: min / max of sqrt(h_jj) and the sum of h_jj in a FIXED
// order (lane-strided partial sums, then a butterfly) — the same bits in every kernel that factors N > 80
__device__ __forceinline__ void colinfo_wave(const double* sq, int n, int lane, double& mn, double& mx, double& sm) {
  mn = __builtin_inf(); mx = 0.0; sm = 0.0;
  for (int j = lane; j < n; j += WAVE) {
    const double v = sq[j];
    mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm = fma(v, v, sm);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double omn = __shfl_xor(mn, o, WAVE), omx = __shfl_xor(mx, o, WAVE), osm = __shfl_xor(sm, o, WAVE);
    mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx; sm = sm + osm;
  }
}

// ---- equilibrated blocked Cholesky, in place in the triangle slot -------------------------------
// Row block kb of R' :  S_j = C_{kb,j} - sum_{k<kb} R'_{k,kb}^T R'_{k,j}   (MFMA, operands from the
// rows already written),  R'_{kb,kb} = chol(S_kb) and its inverse on wave 0 (lane j owns column
// j, broadcasts by v_readlane),  R'_{kb,j} = R'_{kb,kb}^-T S_j  (MFMA; the accumulator layout of S is
// the B-operand layout).  What is stored is R = R' D^-1; operands are re-scaled on the fly.
//
// The same kernel factors the diagonally modified Grams of the trust-region systems (TRF):
//     H = D G D + diag(e^2) (+ alpha I on the first n columns),    D = diag(colscale, 1)
// whose Cholesky factor is the triangle of [R D | c; E | 0] (and of [R_aug; sqrt(alpha) I]).  With
// C = equil(G):  equil(H) = Theta^1/2 C Theta^1/2 + (I - Theta),  0 < Theta <= I diagonal, so its
// extreme eigenvalues lie inside those of C: a problem that passed the gate on C needs no new one.
// NWP waves work on one problem: 8 (a whole workgroup; any size) or 1 (N <= 80: eight problems per
// workgroup, no workgroup barrier at all — the 16x16 chain of one problem overlaps the MFMAs of
// the others on the same SIMD, and sixteen instead of two problems are resident per CU).
template <int NWP>
__global__ __launch_bounds__(GR_NT, 4) void gram_chol_kernel(GramCholArgs a) {
  constexpr int PT = WAVE * NWP;                        // threads per problem
  constexpr int UMAX = (NWP == 8) ? 3 : 5;              // tiles of a row block per wave
  constexpr int PPW = GR_NW / NWP;                      // problems per workgroup
  extern __shared__ double sh_all[];
  __shared__ double red[32];
  __shared__ double pmin_all[GR_NW];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int pslot = wv / NWP;                           // problem slot inside the workgroup
  const int pidx = (int)blockIdx.x * PPW + pslot;
  if (pidx >= a.count) return;                          // (NWP == 1 only: wave-uniform)
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  const int tid = (int)threadIdx.x % PT, lane = tid & 63;
  const int w = wv % NWP;
  double* sh = sh_all + (size_t)pslot * (4 * (size_t)a.NPAD + 512);
  double& pminsh = pmin_all[pslot];
  // synchronisation among the threads of one problem
  auto psync = [&]() {
    if (NWP == 8) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // one wave: program order
  };
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0 && !(a.qr_mask && a.qr_mask[b] == 0)) return;
  double tau = 0.0;                                     // certificate stage 3: factor C - tau I
  if (a.cert_shift) {
    if (!a.cert_flag[b]) return;                        // (uniform: only the problems the norm stage left open)
    tau = a.cert_tau[b];
  }
  // columns of this problem: all n (+ rhs), or the gathered free columns (+ rhs)
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  // source row / column of H's index i  (the rhs is the source's column a.n)
  auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;    // source Gram (may alias the output)
  double* Gb = a.G + (long)b * NPAD * NPAD;             // output triangle
  double* dl = sh;                 // [NPAD] equilibration 1 / sqrt(h_jj)
  double* sq = dl + NPAD;          // [NPAD] sqrt(h_jj)
  double* sc = sq + NPAD;          // [NPAD] colscale_j * dl_j  (scale applied to source entries)
  double* Dt = sc + NPAD;          // [256]  diagonal tile (row-major)
  double* Ri = Dt + 256;           // [256]  its inverse
  double* td = Ri + 256;           // [NPAD] (e_j^2 + alpha) * dl_j^2  (added to the diagonal of C)
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  // 0. column scales from the diagonal of H
  int bad = 0;
  for (int j = tid; j < NPAD; j += PT) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s = g * d;
    }
    dl[j] = d; sq[j] = s; sc[j] = cs * d; td[j] = add * d * d - ((j < n) ? tau : 0.0);
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
  }
  if (a.colinfo) {                                      // (uniform) column-norm summary for the rank gate
    psync();
    double mn = __builtin_inf(), sm = 0.0, mx = 0.0;
    if (NWP == 8) {
      if (w == 0) colinfo_wave(sq, n, lane, mn, mx, sm);
    } else if (tid == 0) {
      for (int j = 0; j < n; ++j) { const double v = sq[j]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm = fma(v, v, sm); }
    }
    if (tid == 0) {
      a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
      if (a.hmax) a.hmax[b] = mx * mx;
      if (a.lam_out) a.lam_out[b] = (double)n;
    }
  }
  // strictly lower tiles are part of the triangle's image: zero
  for (int r = 16 + w; r < (a.skip_zero ? 0 : NPAD); r += NWP) {
    const int cend = r & ~15;
    for (int c = lane; c < cend; c += WAVE) Gb[(long)r * NPAD + c] = 0.0;
  }
  if (NWP == 8) bad = block_or(bad, red); else bad = __any(bad);
  if (tid == 0) pminsh = 1.0;
  psync();
  if (bad) {                                            // uniform: hand the problem to the QR tree
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = 0.0;
      if (a.path_out) a.path_out[b] = a.n + 1;
      if (a.k2_out && !a.cert_shift) a.k2_out[b] = 0.0;
    }
    return;
  }

  const bool stp = (NWP == 8) && pidx == (a.count > 300 ? 300 : 0) && !a.cert_shift;   // (diagnostic stamps)
  (void)stp;
  for (int kb = 0; kb < NT; ++kb) {
    CST(stp && w == 0, 0, kb, 0); CST(stp && w == 3, 1, kb, 0);
    // ---- A. Schur complements of this row block (tile j = kb + w + 8 u) ----
    v4d S[UMAX];
    const double dk = dl[16 * kb + lc];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = kb + w + NWP * u;
      S[u] = v4d{0.0, 0.0, 0.0, 0.0};
      if (j < NT) {
        const double dj = dl[16 * j + lc];
        const double scj = sc[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * kb + lr + 4 * g;
          const int col = 16 * j + lc;
          double v = 0.0;
          if (row < N && col < N) {
            int sr_ = src(row), sc_ = src(col);
            if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }   // symmetric: stay in the upper tiles
            v = Gs[(long)sr_ * NPAD + sc_] * sc[row] * scj;
          }
          if (j == kb && lr + 4 * g == lc) v += td[row];
          S[u][g] = v;
        }
        for (int k = 0; k < kb; ++k) {
          double av[4], bv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const long ro = (long)(16 * k + 4 * s + lr) * NPAD;
            av[s] = Gb[ro + 16 * kb + lc];
            bv[s] = Gb[ro + 16 * j + lc];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) S[u] = gmfma(-(av[s] * dk), bv[s] * dj, S[u]);
        }
      }
    }
    // ---- B. wave 0: Cholesky of the diagonal tile and its inverse (chol16.h), straight from the
    // accumulators of its Schur complement ----
    CST(stp && w == 0, 0, kb, 1); CST(stp && w == 3, 1, kb, 1);
    if (w == 0) {
      const double pm = chol16_blocked3(S[0], Dt, Ri, n - 16 * kb, pminsh);
      if (lane == 0) pminsh = pm;
    }
    CST(stp && w == 0, 0, kb, 2);
    psync();
    CST(stp && w == 0, 0, kb, 3); CST(stp && w == 3, 1, kb, 3);
    if (a.rinv && w == NWP - 1) {                       // kept for the conditioning certificate (off the chain)
      double* ro = a.rinv + ((long)b * (NPAD / 16) + kb) * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = Ri[q * 64 + lane];
    }
    // ---- C. R'_{kb,j} = R'_{kb,kb}^-T S_j, stored as R = R' D^-1 ----
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = kb + w + NWP * u;
      if (j < NT) {
        v4d X = {0.0, 0.0, 0.0, 0.0};
        if (j == kb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) X[g] = Dt[(lr + 4 * g) * 16 + lc];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) X = gmfma(Ri[(4 * s + lr) * 16 + lc], S[u][s], X);
        }
        const double sj = sq[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * kb + lr + 4 * g;
          const int colg = 16 * j + lc;
          double val = X[g] * sj;
          if (row >= n || row > colg || colg > n) val = 0.0;
          Gb[(long)row * NPAD + colg] = val;
        }
      }
    }
    CST(stp && w == 0, 0, kb, 4); CST(stp && w == 3, 1, kb, 4);
    psync();
    CST(stp && w == 0, 0, kb, 5); CST(stp && w == 3, 1, kb, 5);
  }
  if (16 * NT < NPAD && !a.skip_zero) {                  // sub-matrix: the rest of the slot is zero
    for (int r = w; r < NPAD; r += NWP) {
      const int c0 = (r < 16 * NT) ? 16 * NT : (r & ~15);
      for (int c = c0 + lane; c < NPAD; c += WAVE) Gb[(long)r * NPAD + c] = 0.0;
    }
  }
  if (tid == 0 && a.fb_mask) {
    const double floor_ = a.pivot_floor > 0.0 ? a.pivot_floor : 1.0 / GRAM_K2_MAX;
    const bool fail = !(pminsh >= floor_);
    if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = pminsh;
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
    if (fail && !a.cert_shift && a.k2_out) a.k2_out[b] = 0.0;      // (no bound for this factorisation)
    if (a.cert_shift) {
      a.cert_flag[b] = 0;
      if (!fail && a.k2_out) a.k2_out[b] = a.k2_max;    // proven: kappa_2 <= Lambda / tau
    }
  }
}

// ---- N <= 80, one wave per problem, the whole matrix in registers -------------------------------
// At most 5 x 5 tiles: the 15 upper tiles of the equilibrated matrix are loaded ONCE into
// accumulators and never leave the wave until their row block is final.  Per row block: the chain of
// the diagonal tile straight from its accumulator (chol16.h), R'_{kb,j} = R'_kk^-T S_j by MFMA, and
// the right-looking update of the remaining tiles — whose MFMA operands are the rows just solved,
// already in the right layout (register s of a tile in the accumulator layout holds rows 4 s + lr:
// the operand fragment of k-step s).  No L2 round trip inside the factorisation (the left-looking
// kernel above pays one per tile and finished row block: 20 exposed latencies at N = 65), no barrier.
// Same arguments, outputs and gate bookkeeping as gram_chol_kernel<1>; eight problems per workgroup.
__global__ __launch_bounds__(REG_NT, 1) void gram_chol_reg_kernel(GramCholArgs a) {
  constexpr int MT = 5;                                 // tile rows at most (N <= 80)
  extern __shared__ double sh_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int pidx = reg_problem((int)blockIdx.x, wv);
  if (pidx >= a.count) return;                          // (wave-uniform)
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  const int lane = threadIdx.x & 63, tid = lane;
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  double* sh = sh_all + (size_t)wv * (8 * (size_t)NPAD + 256 + MT * 256 + 64 + 16);
  auto wsync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };
  // LDS hand-over between the lanes of this one wave: its DS instructions execute in order, the compiler only has to
  // keep them in order too.  (A `vmcnt(0)` here also waits for every store of the factor issued so far — a round trip
  // to memory per row block on the critical path of a kernel that is one problem's latency.)
  auto lsync = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto unsettle = [&]() { if (tid == 0 && a.unsettled) atomicAdd(a.unsettled, 1); };
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    unsettle();
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0 && !(a.qr_mask && a.qr_mask[b] == 0)) return;
  double tau = 0.0;                                     // certificate stage 3: factor C - tau I
  if (a.cert_shift) {
    if (!a.cert_flag[b]) return;                        // (wave-uniform)
    tau = a.cert_tau[b];
  }
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    unsettle();
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;    // source Gram (may alias the output)
  double* Gb = a.G + (long)b * NPAD * NPAD;             // output triangle
  double* dl = sh;                 // [NPAD] equilibration 1 / sqrt(h_jj)
  double* sq = dl + NPAD;          // [NPAD] sqrt(h_jj)
  double* sc = sq + NPAD;          // [NPAD] colscale_j * dl_j
  double* Dt = sc + NPAD;          // [256]  diagonal tile (row-major)
  double* Ria = Dt + 256;          // [MT][256] the inverses of the diagonal tiles (kept: certificate below)
  double* td = Ria + MT * 256;     // [NPAD] (e_j^2 + alpha) * dl_j^2
  double* xs = td + NPAD;          // [64]   four-row sums
  double* cv = xs + 64;            // [NPAD] c' = R'[:, n]            (dogbox finish below)
  double* yv = cv + NPAD;          // [NPAD] R'^-1 c'
  double* vv = yv + NPAD;          // [NPAD] g of the free variables
  double* wq = vv + NPAD;          // [NPAD] sq . g
  double* tv = wq + NPAD;          // [16]
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  int* sidx = (int*)Ria;           // [NPAD] source indices — only until the tiles are loaded (Ria is free till then)
  const bool stpr = pidx == 500; (void)stpr;
  CST(stpr, 0, 19, 0);
  // 0. column scales from the diagonal of H
  int bad = 0;
  for (int j = tid; j < NPAD; j += WAVE) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s_ = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s_ = g * d;
    }
    dl[j] = d; sq[j] = s_; sc[j] = cs * d; td[j] = add * d * d - ((j < n) ? tau : 0.0);
    sidx[j] = sj_ < NPAD ? sj_ : 0;                     // source row / column of index j (always a valid one)
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
  }
  wsync();
  if (a.colinfo && tid == 0) {                          // column-norm summary for the rank gate
    double mn = __builtin_inf(), sm = 0.0;
    double mx = 0.0;
    for (int j = 0; j < n; ++j) { const double v = sq[j]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm = fma(v, v, sm); }
    a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
    if (a.hmax) a.hmax[b] = mx * mx;
    if (a.lam_out) a.lam_out[b] = (double)n;
    tv[0] = mn; tv[1] = sm;                             // (kept for the dogbox finish)
  }
  bad = __any(bad);
  CST(stpr, 0, 19, 1);
  // 1. the scaled source tiles -> accumulators (the source may alias the output: every read comes
  //    before any write)
  v4d acc[MT * (MT + 1) / 2];
  auto tix = [](int i, int j) { return i * MT - i * (i - 1) / 2 + (j - i); };   // upper tile (i, j) of a 5 x 5 grid
  if (!bad) {
    // Every tile's loads FIRST, from clamped indices and without a branch between them, then the scaling: a uniform
    // `if (j < NT)` around each tile made fifteen basic blocks, i.e. fifteen memory round trips in a row (17 of this
    // kernel's 57 us, tools/reg_stamps.py).
    int scol[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) { const int col = 16 * j + lc; scol[j] = sidx[col < N ? col : N - 1]; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int srow[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) { const int row = 16 * i + lr + 4 * g; srow[g] = sidx[row < N ? row : N - 1]; }
#pragma unroll
      for (int j = i; j < MT; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int lo_ = srow[g] < scol[j] ? srow[g] : scol[j], hi_ = srow[g] < scol[j] ? scol[j] : srow[g];   // symmetric: upper tiles
          acc[tix(i, j)][g] = Gs[(long)lo_ * NPAD + hi_];
        }
      }
    }
    // (the scales of a lane's rows and columns and the diagonal shifts read FIRST, no branch around a tile: what is
    //  outside the N x N block scales to zero anyway)
    double scj_[MT], tdl_[MT], scr_[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int col = 16 * i + lc;
      scj_[i] = sc[col < NPAD ? col : NPAD - 1];
      tdl_[i] = td[col < NPAD ? col : NPAD - 1];
#pragma unroll
      for (int g = 0; g < 4; ++g) { const int row = 16 * i + lr + 4 * g; scr_[i][g] = sc[row < NPAD ? row : NPAD - 1]; }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = i; j < MT; ++j) {
        v4d v4;
        const int col = 16 * j + lc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g;
          double v = acc[tix(i, j)][g];
          v = (row < N && col < N) ? v * scr_[i][g] * scj_[j] : 0.0;
          if (j == i && lr + 4 * g == lc && row < 16 * NT) v += tdl_[i];
          v4[g] = v;
        }
        acc[tix(i, j)] = v4;
      }
    }
  }
  wsync();
  if (bad) {                                            // hand the problem to the QR tree
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = 0.0;
      if (a.path_out) a.path_out[b] = a.n + 1;
      if (a.k2_out && !a.cert_shift) a.k2_out[b] = 0.0;
    }
    unsettle();
    return;
  }
  // strictly lower tiles and everything beyond 16 NT are part of the triangle's image: zero
  // (tile by tile in the accumulators' lane layout: 4 stores per tile, none of them waited for)
#pragma unroll
  for (int ti = 0; ti < MT; ++ti) {
#pragma unroll
    for (int tj = 0; tj < MT; ++tj) {
      if (16 * ti < NPAD && 16 * tj < NPAD && (ti >= NT || tj < ti || tj >= NT)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * ti + lr + 4 * g, col = 16 * tj + lc;
          if (row < NPAD && col < NPAD) Gb[(long)row * NPAD + col] = 0.0;
        }
      }
    }
  }
  double pmin = 1.0;
  CST(stpr, 0, 19, 2);
#pragma unroll
  for (int kb = 0; kb < MT; ++kb) {
    if (kb < NT) {
      // 2. chain of the diagonal tile: R'_kk -> Dt, its inverse -> Ri
      double* Ri = Ria + kb * 256;
      pmin = chol16_blocked3(acc[tix(kb, kb)], Dt, Ri, n - 16 * kb, pmin);
      if (kb == (n >> 4) && lane < 16) cv[16 * kb + lane] = Dt[lane * 16 + (n & 15)];   // rhs column through the diagonal tile
      if (a.rinv) {                                     // kept for the conditioning certificate
        double* ro = a.rinv + ((long)b * (NPAD / 16) + kb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = Ri[q * 64 + lane];
      }
      // 3. the row block: R'_{kb,j} = R'_kk^-T S_j (kept in the accumulators of row kb), stored as R = R' D^-1
#pragma unroll
      for (int j = kb; j < MT; ++j) {
        if (j < NT) {
          v4d X = {0.0, 0.0, 0.0, 0.0};
          if (j == kb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) X[g] = Dt[(lr + 4 * g) * 16 + lc];
            acc[tix(kb, kb)] = X;                       // (all 15 tiles of R' stay in registers: certificate)
          } else {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) X = gmfma(Ri[(4 * s_ + lr) * 16 + lc], acc[tix(kb, j)][s_], X);
            acc[tix(kb, j)] = X;
            if (j == (n >> 4) && lc == (n & 15)) {
#pragma unroll
              for (int g = 0; g < 4; ++g) cv[16 * kb + lr + 4 * g] = X[g];
            }
          }
          const double sj = sq[16 * j + lc];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * kb + lr + 4 * g;
            const int colg = 16 * j + lc;
            double val = X[g] * sj;
            if (row >= n || row > colg || colg > n) val = 0.0;
            Gb[(long)row * NPAD + colg] = val;
          }
        }
      }
      // 4. right-looking update of the tiles below: (i, j) -= R'_{kb,i}^T R'_{kb,j}
#pragma unroll
      for (int i = kb + 1; i < MT; ++i) {
#pragma unroll
        for (int j = i; j < MT; ++j) {
          if (j < NT) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
              acc[tix(i, j)] = gmfma(-acc[tix(kb, i)][s_], acc[tix(kb, j)][s_], acc[tix(i, j)]);
          }
        }
      }
      lsync();                                          // (Dt / Ri are rewritten by the next chain)
    }
  }
  const double kmax = a.k2_max > 0.0 ? a.k2_max : GRAM_K2_MAX;
  const bool fail = !(pmin >= (a.pivot_floor > 0.0 ? a.pivot_floor : 1.0 / GRAM_K2_MAX));
  if (tid == 0 && a.fb_mask) {
    if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = pmin;
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
    if (fail && !a.cert_shift && a.k2_out) a.k2_out[b] = 0.0;      // (no bound for this factorisation)
    if (a.cert_shift) {
      a.cert_flag[b] = 0;
      if (!fail && a.k2_out) a.k2_out[b] = kmax;        // proven: kappa_2 <= Lambda / tau
    }
  }
  CST(stpr, 0, 19, 3);
  // 5. The first bound of the conditioning certificate (gram_cond_kernel below: same quantities, same
  //    definition) while R' and the inverse diagonal tiles are still at hand:
  //        K2 = ||R'||_1 ||R'||_inf ||Y||_1 ||Y||_inf ,   Y = R'^-T  column block by column block.
  //    K2 <= GRAM_K2_MAX settles the problem here; otherwise the separate kernel decides (it also
  //    has the tighter Frobenius bound).
  bool passed = false;
  if (a.cert_done) {
    double k2 = 0.0;
    if (!fail) {
      const int NTn = (n + 15) / 16;
      double r1 = 0.0, rinf = 0.0, y1 = 0.0, yinf = 0.0;
      double colp[MT];
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) colp[jj] = 0.0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < NTn) {
          double rp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int jj = i; jj < MT; ++jj) {
            if (jj < NTn) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int row = 16 * i + lr + 4 * g, col = 16 * jj + lc;
                const double v = (row < n && col < n) ? fabs(acc[tix(i, jj)][g]) : 0.0;
                rp[g] += v; colp[jj] += v;
              }
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) rinf = fmax(rinf, row16_sum(rp[g]));
        }
      }
      rinf = wave_max(rinf);
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) {
        if (jj < NTn) {
          xs[lane] = colp[jj];
          lsync();
          r1 = fmax(r1, (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]));
          lsync();
        }
      }
      r1 = wave_max(r1);
      double rsY[MT][4];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) rsY[i][g] = 0.0;
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) {
        if (jj < NTn) {
          v4d Yc[MT];
          double cY = 0.0;
#pragma unroll
          for (int i = jj; i < MT; ++i) {
            if (i < NTn) {
              v4d Yt = {0.0, 0.0, 0.0, 0.0};
              if (i == jj) {
#pragma unroll
                for (int g = 0; g < 4; ++g) Yt[g] = Ria[jj * 256 + lc * 16 + lr + 4 * g];   // (R'_jj^-1)^T
              } else {
                v4d av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = jj; kk < i; ++kk) {
#pragma unroll
                  for (int s_ = 0; s_ < 4; ++s_) av = gmfma(acc[tix(kk, i)][s_], Yc[kk][s_], av);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) Yt = gmfma(-Ria[i * 256 + (4 * s_ + lr) * 16 + lc], av[s_], Yt);
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int row = 16 * i + lr + 4 * g, col = 16 * jj + lc;
                const double v = (row < n && col < n) ? Yt[g] : 0.0;
                Yt[g] = v;
                const double av_ = fabs(v);
                rsY[i][g] += row16_sum(av_);
                cY += av_;
              }
              Yc[i] = Yt;
            }
          }
          xs[lane] = cY;
          lsync();
          y1 = fmax(y1, (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]));
          lsync();
        }
      }
      y1 = wave_max(y1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) yinf = fmax(yinf, rsY[i][g]);
      yinf = wave_max(yinf);
      k2 = (r1 * rinf) * (y1 * yinf);
      passed = k2 <= kmax;                              // (NaN fails)
    }
    if (tid == 0) {
      a.cert_done[b] = passed ? 1 : 0;
      if (passed && a.k2_out) a.k2_out[b] = k2;
    }
  }
  CST(stpr, 0, 19, 4);
  // 5b. TRF finish (GramCholArgs::lmfin): the `sure` branch of lm_gate_kernel, same expressions
  if (a.lmfin.fast && tid == 0) {
    bool finished = false;
    if (!fail && passed && a.colinfo && a.lmfin.enable != 0 && a.lmfin.m >= n) {
      const double mn = tv[0], sm = tv[1];
      const double smin_lb = GRAM_SMIN_PROVEN * mn, smax_ub = sqrt(sm);
      if (is_finite(sm) && sm > 0.0 && smin_lb > LM_GATE_MARGIN * LM_EPS * a.lmfin.m * smax_ub) {
        a.lmfin.fast[b] = 1;
        a.lmfin.ncols_jac[b] = 0;
        a.lmfin.sc[(long)b * 16 + SC_SMAX] = smax_ub;
        a.lmfin.sc[(long)b * 16 + SC_SMIN] = smin_lb;
        a.lmfin.st[(long)b * 4 + ST_PHASE] = LM_IDLE;
        finished = true;
      }
    }
    if (!finished) unsettle();
  }
  CST(stpr, 0, 19, 5);
  // 6. dogbox finish (GramCholArgs::dog): what dog_gate_solve_kernel computes for a problem on this path —
  //    Cauchy step -(g.g)/(J_f g . J_f g) g_f with |J_f g_f| = |R g_f|, and, when the column-norm bound
  //    already proves the free block full rank (the `sure` case there), the Newton step -R_f^-1 c_f —
  //    from the register tiles:  R = R' diag(sq),  c = c' sq_n.
  if (a.dog.g) {
    bool finished = false;
    if (!fail && a.colinfo) {
      const int NTn = (n + 15) / 16;
      const double mn = tv[0], sm = tv[1];
      lsync();
      const double* gb = a.dog.g + (long)b * a.stride_vec;
      double gg = 0.0;
      for (int q = lane; q < NPAD; q += WAVE) {
        const double gq = (q < n) ? gb[gidx ? gidx[q] : q] : 0.0;
        vv[q] = gq; wq[q] = gq * sq[q]; yv[q] = 0.0;
        gg = fma(gq, gq, gg);
      }
      gg = wave_sum(gg);
      lsync();
      double uu = 0.0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < NTn) {
          double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int jj = i; jj < MT; ++jj) {
            if (jj < NTn) {
              const double wj = wq[16 * jj + lc];
#pragma unroll
              for (int g = 0; g < 4; ++g) part[g] = fma(acc[tix(i, jj)][g], wj, part[g]);
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const double u = row16_sum(part[g]);
            if (lc == 0 && 16 * i + lr + 4 * g < n) uu = fma(u, u, uu);
          }
        }
      }
      uu = wave_sum(uu);
      const double fac = -gg / uu;
      for (int q = lane; q < n; q += WAVE) a.dog.cauchy[(long)b * a.stride_vec + q] = fac * vv[q];
      const int mx = a.dog.m > n ? a.dog.m : n;
      const bool sure = is_finite(sm) && sm > 0.0 && (GRAM_SMIN_PROVEN * mn > LM_GATE_MARGIN * LM_EPS * mx * sqrt(sm));
      if (a.dog.enable != 0 && a.dog.m >= n && sure) {
        // y = R'^-1 c' (block rows from the bottom), newton = -sq_n dl . y
#pragma unroll
        for (int kk = MT - 1; kk >= 0; --kk) {
          if (kk < NTn) {
            double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int jj = kk + 1; jj < MT; ++jj) {
              if (jj < NTn) {
                const double yj = yv[16 * jj + lc];
#pragma unroll
                for (int g = 0; g < 4; ++g) part[g] = fma(acc[tix(kk, jj)][g], yj, part[g]);
              }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) part[g] = row16_sum(part[g]);
            if (lc == 0) {
#pragma unroll
              for (int g = 0; g < 4; ++g) tv[lr + 4 * g] = cv[16 * kk + lr + 4 * g] - part[g];
            }
            lsync();
            const int nb = (n - 16 * kk < 16) ? n - 16 * kk : 16;
            const double* Rk = Ria + kk * 256;
            double yi = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) yi = fma(Rk[lc * 16 + c], (c < nb) ? tv[c] : 0.0, yi);
            if (lc >= nb) yi = 0.0;
            if (lr == 0) yv[16 * kk + lc] = yi;
            lsync();
          }
        }
        const double sqn = sq[n];
        for (int q = lane; q < n; q += WAVE) a.dog.newton[(long)b * a.stride_vec + q] = -(sqn * dl[q] * yv[q]);
        finished = true;
      }
    }
    if (tid == 0) {
      a.dog.done[b] = finished ? 1 : 0;
      if (finished) { a.dog.fast[b] = 1; a.dog.ncols_jac[b] = 0; }
      if (!(finished && passed)) unsettle();
    }
  }
  CST(stpr, 0, 19, 6);
}

// ---- N <= 80: ALL Newton rounds of a problem in one launch ----------------------------------------
// The safeguarded Newton iteration on alpha (trust_region.py:126-150) factors H + alpha I once per
// round.  For N <= 80 one wave owns a problem for the whole iteration: per round the factor of
// gram_chol_reg_kernel (tiles in registers, nothing stored), p = -R^-1 c by block back substitution
// and q = R^-T p by block forward substitution straight from the register tiles (tile x vector: four
// FMAs per lane and tile + a 16-lane DPP sum, or a four-row sum through LDS for the transposed
// product; the 16 x 16 diagonal solves are matvecs with the inverse tiles the chain produces anyway),
// then the scalar update of lm_update_kernel, verbatim.  No launch, no counter read-back and no
// triangle written between rounds (six stream operations per round otherwise, each with its dispatch
// gap).  Everything happens in the equilibrated system:  R = R' diag(sq),  c = c' sq_n  =>
//     p_j = -sq_n dl_j (R'^-1 c')_j ,      q = R'^-T (dl . p) .
__device__ __forceinline__ double lm_restart_reg(double lo, double hi) {     // trust_region.py:128,134
  const double gm = sqrt(lo * hi);
  return (0.001 * hi > gm) ? 0.001 * hi : gm;
}
// The launch first does what lm_start_kernel does — the Gauss-Newton step from the AUGMENTED factor
// (its stored triangle, column scales and inverse diagonal tiles are re-loaded: R' = R diag(dl)), the
// acceptance test |p| <= Delta and the bracket (trust_region.py:116-130) — for every
// normal-equations-path problem of the batch (the others are left to lm_start and the round loop,
// LmState.fused_gram): no list, no counter, and the same arithmetic for a problem whatever else its
// batch holds.
__global__ __launch_bounds__(REG_NT, 1) void lm_rounds_reg_kernel(GramCholArgs a, LmState lm,
                                                                 const double* Delta_in,
                                                                 const double* alpha_in) {
  constexpr int MT = 5;
  constexpr bool start = true;
  extern __shared__ double sh_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int b = reg_problem((int)blockIdx.x, wv);
  if (b >= lm.B) return;
  if (lm.path && lm.path[b] != 0) return;               // (Householder-path problem: lm_start and the round loop)
  const int lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
  if (!lm.fast[b]) {
    if (lane == 0) lm.ncols_lm[b] = 0;
    return;
  }
  double* scv = lm.sc + (long)b * 16;
  int* stv = lm.st + (long)b * 4;
  int phase = LM_EVAL;
  const int NPAD = a.NPAD, n = a.n, N = n + 1;
  const int NT = (N + 15) / 16, NTn = (n + 15) / 16;
  const int jn = n >> 4, cn = n & 15;                   // tile column / column inside it of the rhs
  double* sh = sh_all + (size_t)wv * (8 * (size_t)NPAD + 256 + MT * 256 + 16 + 64);
  double* dl = sh;                 // [NPAD] 1 / sqrt(h_jj)
  double* sq = dl + NPAD;          // [NPAD] sqrt(h_jj)
  double* sc = sq + NPAD;          // [NPAD] colscale_j dl_j
  double* td = sc + NPAD;          // [NPAD] (e_j^2 + alpha) dl_j^2
  double* cv = td + NPAD;          // [NPAD] c' = R'[:, n]
  double* yv = cv + NPAD;          // [NPAD] R'^-1 c', then dl . p
  double* pv = yv + NPAD;          // [NPAD] p
  double* zv = pv + NPAD;          // [NPAD] R'^-T (dl . p)
  double* Dt = zv + NPAD;          // [256]
  double* Ria = Dt + 256;          // [MT][256] inverse diagonal tiles
  double* tv = Ria + MT * 256;     // [16]
  double* xs = tv + 16;            // [64]
  auto wsync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  auto tix = [](int i, int j) { return i * MT - i * (i - 1) / 2 + (j - i); };
  v4d acc[MT * (MT + 1) / 2];
  double sqn = 1.0;

  // y = R'^-1 c' (rows and columns below n only; c' in cv), block rows from the bottom -> yv
  auto back_solve = [&]() {
#pragma unroll
    for (int kk = MT - 1; kk >= 0; --kk) {
      if (kk < NTn) {
        double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = kk + 1; j < MT; ++j) {
          if (j < NTn) {
            const double yj = yv[16 * j + lc];
#pragma unroll
            for (int g = 0; g < 4; ++g) part[g] = fma(acc[tix(kk, j)][g], yj, part[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) part[g] = row16_sum(part[g]);
        if (lc == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) tv[lr + 4 * g] = cv[16 * kk + lr + 4 * g] - part[g];
        }
        wsync();
        const int nb = (n - 16 * kk < 16) ? n - 16 * kk : 16;
        const double* Rk = Ria + kk * 256;
        double yi = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) yi = fma(Rk[lc * 16 + c], (c < nb) ? tv[c] : 0.0, yi);
        if (lc >= nb) yi = 0.0;
        if (lr == 0) yv[16 * kk + lc] = yi;
        wsync();
      }
    }
  };
  // p = -sq_n dl . y -> pv,  w = dl . p -> yv;  returns |p|
  auto form_p = [&]() -> double {
    double pp = 0.0;
    for (int j = lane; j < NPAD; j += WAVE) {
      const double pj = (j < n) ? -(sqn * dl[j] * yv[j]) : 0.0;
      pv[j] = pj;
      pp = fma(pj, pj, pp);
    }
    wsync();
    for (int j = lane; j < NPAD; j += WAVE) yv[j] = (j < n) ? dl[j] * pv[j] : 0.0;
    const double pn_ = sqrt(wave_sum(pp));
    wsync();
    return pn_;
  };
  // z = R'^-T w (w in yv), block rows from the top -> zv;  returns |z|^2
  auto fwd_solve = [&]() -> double {
#pragma unroll
    for (int kk = 0; kk < MT; ++kk) {
      if (kk < NTn) {
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < kk; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) part = fma(acc[tix(j, kk)][g], zv[16 * j + lr + 4 * g], part);
        }
        xs[lane] = part;
        wsync();
        const double tot = (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]);
        if (lr == 0) tv[lc] = yv[16 * kk + lc] - tot;
        wsync();
        const int nb = (n - 16 * kk < 16) ? n - 16 * kk : 16;
        const double* Rk = Ria + kk * 256;
        double zi = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) zi = fma(Rk[c * 16 + lc], (c <= lc) ? tv[c] : 0.0, zi);
        if (lc >= nb) zi = 0.0;
        if (lr == 0) zv[16 * kk + lc] = zi;
        wsync();
      }
    }
    double qq = 0.0;
    for (int j = lane; j < NPAD; j += WAVE) { const double zj = (j < n) ? zv[j] : 0.0; qq = fma(zj, zj, qq); }
    return wave_sum(qq);
  };

  double alpha, lo, hi, phi, dphi, Delta;
  int it, n_iter;
  if (start) {
    // ---- the augmented factor back into registers:  R' = R diag(dl),  Ri from the factor kernel ----
    Delta = Delta_in[b];
    const double* Ra = lm.Raug + (long)b * NPAD * NPAD;
    const double* dsc = a.dsc + (long)b * NPAD;
    const double* rinv = a.rinv + (long)b * (NPAD / 16) * 256;
    for (int j = lane; j < NPAD; j += WAVE) { dl[j] = dsc[j]; yv[j] = 0.0; pv[j] = 0.0; zv[j] = 0.0; }
    for (int e = lane; e < NT * 256; e += WAVE) Ria[e] = rinv[e];
    wsync();
    sqn = 1.0 / dl[n];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = i; j < MT; ++j) {
        v4d v4 = {0.0, 0.0, 0.0, 0.0};
        if (j < NT) {
          const int col = 16 * j + lc;
          const double dj = dl[col];
#pragma unroll
          for (int g = 0; g < 4; ++g) v4[g] = Ra[(long)(16 * i + lr + 4 * g) * NPAD + col] * dj;
        }
        acc[tix(i, j)] = v4;
      }
    }
    for (int r = lane; r < NPAD; r += WAVE) cv[r] = (r < n) ? Ra[(long)r * NPAD + n] * dl[n] : 0.0;
    wsync();
    // |R^T c| = sq_n |sq . (R'^T c')|  (alpha_upper = |A^T b| / Delta, trust_region.py:111-113)
    double gg = 0.0;
#pragma unroll
    for (int kk = 0; kk < MT; ++kk) {
      if (kk < NTn) {
        double part = 0.0;
#pragma unroll
        for (int j = 0; j <= kk; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) part = fma(acc[tix(j, kk)][g], cv[16 * j + lr + 4 * g], part);
        }
        xs[lane] = part;
        wsync();
        const double tot = (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]);
        const int col = 16 * kk + lc;
        const double gj = (col < n) ? tot / dl[col] : 0.0;
        if (lr == 0) gg = fma(gj, gj, gg);
        wsync();
      }
    }
    const double gnorm = sqn * sqrt(wave_sum(gg));
    back_solve();
    const double pn = form_p();
    for (int j = lane; j < n; j += WAVE) lm.ph[(long)b * lm.ld + j] = pv[j];
    if (pn <= Delta) {                                  // trust_region.py:116-117
      if (lane == 0) {
        scv[SC_ALPHA] = 0.0; stv[ST_NITER] = 0; stv[ST_PHASE] = LM_IDLE; scv[SC_DELTA] = Delta;
        lm.ncols_lm[b] = 0;
      }
      return;
    }
    const double qq = fwd_solve();                      // phi(0), phi'(0) -> alpha_lower (:121-123)
    phi = pn - Delta;
    dphi = -qq / pn;
    hi = gnorm / Delta;
    lo = -phi / dphi;
    alpha = alpha_in[b];                                // :127-130 (full rank)
    if (alpha < lo || alpha > hi) alpha = lm_restart_reg(lo, hi);   // :133-134, iteration 0
    it = 0; n_iter = 0;
    if (lane == 0) scv[SC_DELTA] = Delta;
  }
  for (int guard = 0; guard < 12; ++guard) {
    const double sa = sqrt(alpha);
    // ---- factor of H + alpha I (as gram_chol_reg_kernel; no gather, nothing stored) ----
    for (int j = lane; j < NPAD; j += WAVE) {
      const double cs = (csv && j < n) ? csv[j] : 1.0;
      const double ej = (edv && j < n) ? edv[j] : 0.0;
      const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
      const double g = (j < N) ? fma(Gs[(long)j * NPAD + j] * cs, cs, add) : 0.0;
      const bool okc = (g > 0.0) && is_finite(g);
      double d = 1.0, s_ = 1.0;
      if (j < N && okc) {
        d = __builtin_amdgcn_rsq(g);
        d = d * fma(-0.5 * g * d, d, 1.5);
        d = d * fma(-0.5 * g * d, d, 1.5);
        s_ = g * d;
      }
      dl[j] = d; sq[j] = s_; sc[j] = cs * d; td[j] = add * d * d;
      yv[j] = 0.0; pv[j] = 0.0; zv[j] = 0.0;
    }
    wsync();
    sqn = sq[n];
    // (every tile's loads FIRST, from clamped indices and without a branch between them — a uniform `if (j < NT)` around
    //  each tile made fifteen basic blocks, i.e. fifteen memory round trips in a row: 15 of this kernel's us per round,
    //  tools/reg_stamps.py — then the scaling)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = i; j < MT; ++j) {
        const int col = 16 * j + lc;
        const int ccl = col < N ? col : N - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g;
          const int rcl = row < N ? row : N - 1;
          const int lo_ = rcl < ccl ? rcl : ccl, hi_ = rcl < ccl ? ccl : rcl;
          acc[tix(i, j)][g] = Gs[(long)lo_ * NPAD + hi_];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      double scr_[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) scr_[g] = sc[(16 * i + lr + 4 * g) < NPAD ? 16 * i + lr + 4 * g : NPAD - 1];
#pragma unroll
      for (int j = i; j < MT; ++j) {
        v4d v4 = {0.0, 0.0, 0.0, 0.0};
        if (j < NT) {
          const int col = 16 * j + lc;
          const double scj = sc[col];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * i + lr + 4 * g;
            double v = acc[tix(i, j)][g];
            v = (row < N && col < N) ? v * scr_[g] * scj : 0.0;
            if (j == i && lr + 4 * g == lc) v += td[row];
            v4[g] = v;
          }
        }
        acc[tix(i, j)] = v4;
      }
    }
    wsync();
    double pmin = 1.0;
#pragma unroll
    for (int kb = 0; kb < MT; ++kb) {
      if (kb < NT) {
        double* Ri = Ria + kb * 256;
        pmin = chol16_blocked3(acc[tix(kb, kb)], Dt, Ri, n - 16 * kb, pmin);
        if (kb == jn) {                                 // the rhs column runs through this diagonal tile
          if (lane < 16) cv[16 * kb + lane] = Dt[lane * 16 + cn];
        }
#pragma unroll
        for (int j = kb + 1; j < MT; ++j) {
          if (j < NT) {
            v4d X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) X = gmfma(Ri[(4 * s_ + lr) * 16 + lc], acc[tix(kb, j)][s_], X);
            acc[tix(kb, j)] = X;
            if (j == jn && lc == cn) {
#pragma unroll
              for (int g = 0; g < 4; ++g) cv[16 * kb + lr + 4 * g] = X[g];
            }
          }
        }
#pragma unroll
        for (int i = kb + 1; i < MT; ++i) {
#pragma unroll
          for (int j = i; j < MT; ++j) {
            if (j < NT) {
#pragma unroll
              for (int s_ = 0; s_ < 4; ++s_)
                acc[tix(i, j)] = gmfma(-acc[tix(kb, i)][s_], acc[tix(kb, j)][s_], acc[tix(i, j)]);
            }
          }
        }
        wsync();
      }
    }
    back_solve();
    const double pn = form_p();
    bool finished = false;
    if (phase == LM_FINAL) {
      finished = true;                                  // p at the updated alpha, rescale test on the STALE phi (:149)
    } else {
      const double qq = fwd_solve();
      // ---- the update of lm_update_kernel (trust_region.py:136-146) ----
      phi = pn - Delta;
      dphi = -qq / pn;
      if (fabs(phi) < 0.01 * Delta) {                   // :138-139
        finished = true;
        n_iter = it + 1;
      } else {
        if (phi < 0.0) hi = alpha;                      // :141-142
        const double ratio = phi / dphi;
        const double cand = alpha - ratio;
        lo = (cand > lo) ? cand : lo;                   // :145
        alpha -= (phi + Delta) * ratio / Delta;         // :146
        ++it;
        if (it >= 10) {                                 // max_iter reached: final p at the new alpha
          n_iter = 10;
          phase = LM_FINAL;
        } else {
          if (alpha < lo || alpha > hi) alpha = lm_restart_reg(lo, hi);   // :133-134 of the next pass
          phase = LM_EVAL;
        }
      }
    }
    if (finished) {
      const double f = (phi > 0.0) ? Delta / pn : 1.0;  // :149-150
      for (int j = lane; j < n; j += WAVE) lm.ph[(long)b * lm.ld + j] = pv[j] * f;
      break;
    }
  }
  if (lane == 0) {
    scv[SC_ALPHA] = alpha; scv[SC_LO] = lo; scv[SC_HI] = hi; scv[SC_PHI] = phi; scv[SC_DPHI] = dphi;
    stv[ST_IT] = it; stv[ST_PHASE] = LM_IDLE; stv[ST_NITER] = n_iter;
    lm.sa[b] = sqrt(alpha);
    lm.ncols_lm[b] = 0;
  }
}

hipError_t launch_lm_rounds_reg(const GramCholArgs& c, const LmState& lm, const double* Delta,
                                const double* alpha_in, hipStream_t s) {
  const size_t per = sizeof(double) * (8 * (size_t)c.NPAD + 256 + 5 * 256 + 16 + 64);
  static std::atomic<size_t> granted[64];
  hipError_t ge = gram_grant_lds(lm_rounds_reg_kernel, per * REG_NW, granted);
  if (ge != hipSuccess) return ge;
  hipLaunchKernelGGL(lm_rounds_reg_kernel, dim3(reg_grid(lm.B)), dim3(REG_NT), per * REG_NW, s, c, lm,
                     Delta, alpha_in);
  return hipGetLastError();
}

// ---- right-looking variant: the whole (scaled) matrix lives in accumulators ----------------------
// The NT (NT + 1) / 2 <= 153 upper tiles are dealt CYCLICALLY (row-major tile q -> wave q % 8, slot
// q / 8) so that the shrinking trailing matrix stays balanced, and never leave the registers until
// their row block is final.  Per row block kb: the diagonal tile goes through LDS to wave 0 for the
// 16x16 Cholesky + inverse, the owners of the tiles (kb, j) solve them by MFMA and publish them in an
// LDS row buffer, and every wave updates its own trailing tiles from that buffer — no global-memory
// round trip inside the factorisation (the left-looking kernel above re-reads finished rows from L2).
// Same arguments, same outputs, same gate bookkeeping as gram_chol_kernel.
// SL tile slots per worker wave, the first KL of them kept in LDS instead of registers: the cyclic
// tile table gives slot t the tiles 7 t .. 7 t + 6 in row-major order, so the first slots hold the top
// rows — solved after at most three trailing updates and dead afterwards.  With all 22 slots in
// registers (176 VGPRs) the compiler spilled 76 VGPRs to scratch and reloaded / stored them in
// every row-block step; with eight slots in LDS (112 KB) 28 remain (two Newton rounds of 4096 x 256,
// 512 problems: 0.404 -> 0.376 ms; same operations in the same order, same bits).
// The factorisation of ONE problem by the calling workgroup (GR_NT threads): the body of gram_chol_rl_kernel.  sh: the dynamic LDS (launch_gram_chol sizes it),
// red: 32 doubles, pminsh / flagsh: one double / int of static LDS.  All exits are uniform over the workgroup.
template <int SL, int KL>
__device__ __forceinline__ void chol_rl_body(const GramCholArgs& a, const int b, double* sh, double* red,
                                             double& pminsh, int& flagsh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  const bool stp0 = (int)blockIdx.x == 100 && !a.cert_shift; (void)stp0;
  CST(stp0 && (w == 0 || w == 2), w == 0 ? 2 : 3, 17, 0);
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0 && !(a.qr_mask && a.qr_mask[b] == 0)) return;
  double tau = 0.0;                                     // certificate stage 3: factor C - tau I
  if (a.cert_shift) {
    if (!a.cert_flag[b]) return;                        // (uniform)
    tau = a.cert_tau[b];
  }
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
  double* Gb = a.G + (long)b * NPAD * NPAD;
  double* dl = sh;                 // [NPAD]
  double* sq = dl + NPAD;          // [NPAD]
  double* sc = sq + NPAD;          // [NPAD]
  double* td = sc + NPAD;          // [NPAD]
  double* Dt = td + NPAD;          // [256]
  double* Ri = Dt + 256;           // [256]
  double* Rrow = Ri + 256;         // [NT][256] finished tiles of the current row block
  double* accL = Rrow + (NPAD / 16) * 256;   // [KL][7][256] the LDS-resident tile slots
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  int bad = 0;
  for (int j = tid; j < NPAD; j += GR_NT) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s = g * d;
    }
    dl[j] = d; sq[j] = s; sc[j] = cs * d; td[j] = add * d * d - ((j < n) ? tau : 0.0);
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
  }
  if (a.colinfo) {
    __syncthreads();
    double mn = __builtin_inf(), sm = 0.0, mx = 0.0;
    if (w == 0) colinfo_wave(sq, n, lane, mn, mx, sm);
    if (tid == 0) {
      a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
      if (a.hmax) a.hmax[b] = mx * mx;
      if (a.lam_out) a.lam_out[b] = (double)n;
    }
  }
  bad = block_or(bad, red);
  if (tid == 0) { pminsh = 1.0; flagsh = 0; }
  __syncthreads();
  if (bad) {
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = 0.0;
      if (a.path_out) a.path_out[b] = a.n + 1;
      if (a.k2_out && !a.cert_shift) a.k2_out[b] = 0.0;
    }
    return;
  }
  CST(stp0 && (w == 0 || w == 2), w == 0 ? 2 : 3, 17, 1);
  // ROLES: wave 0 only runs the 16x16 chains (its registers hold the column / inverse vectors, no
  // tiles); waves 1..7 own the tiles.  LOOKAHEAD: in the trailing update of row block kb the owner
  // of the next diagonal tile updates it first, puts it into LDS and raises a flag; wave 0 starts
  // the chain of block kb + 1 on that flag while the other tiles are still being updated.  Both
  // loops pass the same two barriers per row block.
  constexpr int NWK = GR_NW - 1;                        // worker waves
  const int ntile = NT * (NT + 1) / 2;
  const bool stp = (int)blockIdx.x == 100 && !a.cert_shift;   // (diagnostic stamps)
  (void)stp;
  // zeros outside the factor: strictly lower tiles, and everything beyond 16 NT (sub-matrix use)
  auto zero_fill = [&]() {
    for (int r = w; r < (a.skip_zero ? 0 : NPAD); r += GR_NW) {
      const int cend = (r < 16 * NT) ? (r & ~15) : NPAD;
      for (int c = lane; c < cend; c += WAVE) Gb[(unsigned)(r * NPAD + c)] = 0.0;
      if (r < 16 * NT)
        for (int c = 16 * NT + lane; c < NPAD; c += WAVE) Gb[(unsigned)(r * NPAD + c)] = 0.0;
    }
  };
  if (w == 0) {
    __syncthreads();                                    // X: workers have read the source, Dt holds tile (0, 0)
    CST(stp, 2, 17, 3);
    zero_fill();
    CST(stp, 2, 17, 4);
    double pmin = 1.0;
    for (int kb = 0; kb < NT; ++kb) {
      if (kb > 0) {                                     // wait for the updated diagonal tile kb
        while (__hip_atomic_load(&flagsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kb)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      CST(stp, 2, kb, 0);
      pmin = chol16_blocked3(Dt, Ri, n - 16 * kb, pmin);   // (chol16.h)
      CST(stp, 2, kb, 1);
      if (a.rinv) {                                     // kept for the conditioning certificate
        double* ro = a.rinv + ((long)b * (NPAD / 16) + kb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = Ri[q * 64 + lane];
      }
      __syncthreads();                                  // B: R'_kk and its inverse are in LDS
      CST(stp, 2, kb, 2);
      __syncthreads();                                  // C: (workers published the row block)
      CST(stp, 2, kb, 3);
    }
    if (lane == 0) pminsh = pmin;
  } else {
    // tile table (cyclic over the worker waves: the shrinking trailing matrix stays balanced) and
    // the scaled source tiles -> accumulators; the source may alias the output, so everything is
    // read before anything is written
    const int ww = w - 1;
    int ti[SL], tj[SL];
    v4d accR[SL - KL];                                  // slots KL .. SL-1 (registers)
    auto slot = [&](int t) -> double* { return accL + ((size_t)t * NWK + ww) * 256 + lane; };   // [g * 64]
#define RL_ACC_GET(t, dst)                                                        \
    do {                                                                          \
      if ((t) < KL) { const double* p_ = slot(t);                                 \
        dst = v4d{p_[0], p_[64], p_[128], p_[192]}; }                             \
      else dst = accR[(t) < KL ? 0 : (t) - KL];                                   \
    } while (0)
#define RL_ACC_PUT(t, src)                                                        \
    do {                                                                          \
      if ((t) < KL) { double* p_ = slot(t);                                       \
        p_[0] = (src)[0]; p_[64] = (src)[1]; p_[128] = (src)[2]; p_[192] = (src)[3]; } \
      else accR[(t) < KL ? 0 : (t) - KL] = src;                                   \
    } while (0)
#pragma unroll
    for (int t = 0; t < SL; ++t) {
      int q = ww + NWK * t;
      const bool valid = q < ntile;
      int i = 0;
      while (valid && q >= NT - i) { q -= NT - i; ++i; }
      ti[t] = valid ? i : -1;
      tj[t] = valid ? i + q : -1;
      v4d a0 = v4d{0.0, 0.0, 0.0, 0.0};
      if (valid) {
        const int j = tj[t];
        const double scj = sc[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          double v = 0.0;
          if (row < N && col < N) {
            int sr_ = src(row), sc_ = src(col);
            if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }
            v = Gs[(unsigned)(sr_ * NPAD + sc_)] * sc[row] * scj;
          }
          if (j == i && lr + 4 * g == lc) v += td[row];
          a0[g] = v;
        }
        if (i == 0 && j == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) Dt[(lr + 4 * g) * 16 + lc] = a0[g];
        }
      }
      RL_ACC_PUT(t, a0);
    }
    CST(stp && w == 2, 3, 17, 2);
    __syncthreads();                                    // X: all source reads done before the first store
    CST(stp && w == 2, 3, 17, 3);
    zero_fill();
    CST(stp && w == 2, 3, 17, 4);
    for (int kb = 0; kb < NT; ++kb) {
      CST(stp && w == 2, 3, kb, 5);
      __syncthreads();                                  // B: wave 0 finished the chain of block kb
      CST(stp && w == 2, 3, kb, 0);
      // c. the row block: R'_{kb,j} = R'_{kb,kb}^-T S_j -> LDS row buffer and (unscaled) to memory
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (ti[t] == kb) {
          const int j = tj[t];
          v4d X = {0.0, 0.0, 0.0, 0.0};
          if (j == kb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) X[g] = Dt[(lr + 4 * g) * 16 + lc];
          } else {
            v4d S;
            RL_ACC_GET(t, S);
#pragma unroll
            for (int s = 0; s < 4; ++s) X = gmfma(Ri[(4 * s + lr) * 16 + lc], S[s], X);
          }
          const double sj = sq[16 * j + lc];
          const double dj = dl[16 * j + lc];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * kb + lr + 4 * g;
            const int colg = 16 * j + lc;
            double val = X[g] * sj;
            if (row >= n || row > colg || colg > n) val = 0.0;
            Gb[(unsigned)(row * NPAD + colg)] = val;
            // the operand of the trailing updates is what the left-looking kernel reads back: the
            // STORED entry times its column's equilibration — the two kernels agree bit for bit
            if (j != kb) Rrow[j * 256 + (lr + 4 * g) * 16 + lc] = val * dj;
          }
        }
      }
      CST(stp && w == 2, 3, kb, 1);
      __syncthreads();                                  // C: the row block is in the LDS buffer
      CST(stp && w == 2, 3, kb, 2);
      // d. trailing update: the next diagonal tile first (-> LDS, flag for wave 0), then the rest
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (ti[t] == kb + 1 && tj[t] == kb + 1) {
          const double* Ra = Rrow + ti[t] * 256 + lr * 16 + lc;
          v4d S;
          RL_ACC_GET(t, S);
#pragma unroll
          for (int s = 0; s < 4; ++s) S = gmfma(-Ra[64 * s], Ra[64 * s], S);
          RL_ACC_PUT(t, S);
#pragma unroll
          for (int g = 0; g < 4; ++g) Dt[(lr + 4 * g) * 16 + lc] = S[g];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(&flagsh, kb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      CST(stp && w == 2, 3, kb, 3);
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (ti[t] > kb && !(ti[t] == kb + 1 && tj[t] == kb + 1)) {
          const double* Ra = Rrow + ti[t] * 256 + lr * 16 + lc;
          const double* Rb = Rrow + tj[t] * 256 + lr * 16 + lc;
          v4d S;
          RL_ACC_GET(t, S);
#pragma unroll
          for (int s = 0; s < 4; ++s) S = gmfma(-Ra[64 * s], Rb[64 * s], S);
          RL_ACC_PUT(t, S);
        }
      }
      CST(stp && w == 2, 3, kb, 4);
    }
#undef RL_ACC_GET
#undef RL_ACC_PUT
  }
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 18, 0);
  __syncthreads();
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 18, 1);
  if (tid == 0 && a.fb_mask) {
    const bool fail = !(pminsh >= (a.pivot_floor > 0.0 ? a.pivot_floor : 1.0 / GRAM_K2_MAX));
    if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = pminsh;
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
    if (fail && !a.cert_shift && a.k2_out) a.k2_out[b] = 0.0;      // (no bound for this factorisation)
    if (a.cert_shift) {
      a.cert_flag[b] = 0;
      if (!fail && a.k2_out) a.k2_out[b] = a.k2_max;    // proven: kappa_2 <= Lambda / tau
    }
  }
}

template <int SL, int KL>
__global__ __launch_bounds__(GR_NT, 2) void gram_chol_rl_kernel(GramCholArgs a) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  __shared__ double pminsh;
  __shared__ int flagsh;                                // last diagonal tile handed to wave 0
  const int pidx = (int)blockIdx.x;
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  chol_rl_body<SL, KL>(a, b, sh, red, pminsh, flagsh);
}

// ---- right-looking, flag-driven: no workgroup barrier inside the factorisation -------------------
// The arithmetic of gram_chol_rl_kernel (same tiles, same operands, same order: the same bits), scheduled along its
// critical path  chain(kb) -> R'_{kb,kb+1} -> S_{kb+1,kb+1} -> chain(kb + 1):
//   * wave 0 only runs the 16x16 chains: it waits for the flag "diagonal tile kb is in Dt", factors, raises "R'_kk
//     and its inverse are in LDS" — it never meets a barrier, nor the stores of the workers;
//   * the owner of tile (kb, kb+1) solves it first and raises a flag; the owner of (kb+1, kb+1) waits for exactly
//     that tile, updates the diagonal tile and hands it to wave 0; only then come the other tiles of the row;
//   * the trailing update of a worker starts when a COUNTER says that all seven workers have published their
//     tiles of the row block — LDS flags and lgkmcnt waits only, so nobody waits for the acknowledgement of the
//     global stores of the factor (a __syncthreads does: vmcnt counts stores on gfx9);
//   * Dt, Ri and the row buffer are double-buffered by the parity of kb.  Buffer kb & 1 is written again at row block
//     kb + 2, whose chain needs S_{kb+2,kb+2}, i.e. its owner's trailing update with row kb, which waited for the
//     counter of row kb — every worker had then read Ri / Dt of kb and finished its trailing update of kb - 1.
//   * the source tiles are requested ALL AT ONCE at kernel entry (the old preamble paid one memory round trip per
//     tile slot: 37 us) and the column summary is a wave reduction (14 -> 3 us).
// SL tile slots per worker wave, the first KL in LDS (top rows: dead after three row blocks).
__device__ __forceinline__ void spin_ge(const int* f, int v) {
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void raise_flag(int* f, int v, int lane) {      // (after this wave's LDS writes)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
enum { FL_DIAG = 0, FL_RINV = 1, FL_ROW1 = 2, FL_PUB = 3, FL_BAD = 4, FL_YB = 5, FL_YPUB = 6 };

template <int SL, int KL, bool CERT>
__device__ __forceinline__ void chol_rl2_body(const GramCholArgs& a, const int b, double* sh, int* fl, double& pminsh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  const bool stp = (int)blockIdx.x == 100 && !a.cert_shift; (void)stp;
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 17, 0);
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0 && !(a.qr_mask && a.qr_mask[b] == 0)) return;
  double tau = 0.0;                                     // certificate stage 3: factor C - tau I
  if (a.cert_shift) {
    if (!a.cert_flag[b]) return;                        // (uniform)
    tau = a.cert_tau[b];
  }
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int NTP = NPAD / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
  double* Gb = a.G + (long)b * NPAD * NPAD;
  double* dl = sh;                 // [NPAD]
  double* sq = dl + NPAD;          // [NPAD]
  double* sc = sq + NPAD;          // [NPAD]
  double* td = sc + NPAD;          // [NPAD]
  double* Dt = td + NPAD;          // [2][256]
  double* Ri = Dt + 512;           // [2][256]
  double* Rrow = Ri + 512;         // [2][NTP][256] finished tiles of a row block (operands of the trailing updates)
  double* accL = Rrow + 2 * (size_t)NTP * 256;          // [KL][7][256] the LDS-resident tile slots
  int* gl = reinterpret_cast<int*>(accL + (size_t)KL * (GR_NW - 1) * 256);   // [NPAD] gathered source indices
  // The factor kernel's share of the certificate's stage 0 (GramCholArgs::cert_ym): the solve M(R')^T y = e and
  // the column sums of |R'| advance row block by row block as R' is produced — stage 0 then needs ONE pass over
  // the factor (the backward solve, with the row sums on the way) instead of four.  Fixed order: reproducible.
  double* yv = reinterpret_cast<double*>(gl + NPAD);     // [NPAD] y (final for the finished row blocks)
  double* csum = yv + NPAD;                              // [NPAD] sum_i |R'_ij| over the finished row blocks
  constexpr bool cert = CERT;                            // (a launch with cert_ym set, never the shifted one)
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  if (tid < 8) fl[tid] = 0;
  if (tid == 0) pminsh = 1.0;
  if (gidx) {                                           // (uniform) the index map of a gathered sub-matrix -> LDS
    for (int j = tid; j < NPAD; j += GR_NT) gl[j] = j < n ? gidx[j] : a.n;
  }
  __syncthreads();
  auto src = [&](int i) -> int { return gidx ? gl[i] : i; };     // (i < N)

  constexpr int NWK = GR_NW - 1;                        // worker waves
  const int ww = w - 1;
  // tile q (row-major over the upper tiles) -> worker q % 7, slot q / 7; packed (i | j << 8) per slot
  int tij[SL];
  v4d acc[SL];                                          // (slots < KL leave for LDS after the scaling)
  if (w > 0) {
    int i = 0, off = ww;                                // slot 0: q = ww
#pragma unroll
    for (int t = 0; t < SL; ++t) {
      while (i < NT && off >= NT - i) { off -= NT - i; ++i; }
      const bool valid = i < NT;
      tij[t] = valid ? (i | ((i + off) << 8)) : -1;
      acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
      if (valid) {
        const int j = i + off;
        const int col = 16 * j + lc;
        const int scol = col < N ? src(col) : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g;
          int sr_ = row < N ? src(row) : 0, sc_ = scol;
          if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }       // symmetric: stay in the upper tiles
          acc[t][g] = Gs[(unsigned)(sr_ * NPAD + sc_)];                     // (raw; selected and scaled below)
        }
      }
      off += NWK;
    }
  }
  // column scales from the diagonal of H
  int bad = 0;
  for (int j = tid; j < NPAD; j += GR_NT) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s_ = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s_ = g * d;
    }
    dl[j] = d; sq[j] = s_; sc[j] = cs * d; td[j] = add * d * d - ((j < n) ? tau : 0.0);
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
    if (cert) { yv[j] = 1.0; csum[j] = 0.0; }
  }
  if (__any(bad) && lane == 0) __hip_atomic_store(&fl[FL_BAD], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 17, 1);
  __syncthreads();                                      // S: the scales are in LDS
  auto slot = [&](int t) -> double* { return accL + ((size_t)t * NWK + ww) * 256 + lane; };   // [g * 64]
#define RL2_GET(t, dst)                                                           \
    do {                                                                          \
      if ((t) < KL) { const double* p_ = slot(t);                                 \
        dst = v4d{p_[0], p_[64], p_[128], p_[192]}; }                             \
      else dst = acc[t];                                                          \
    } while (0)
#define RL2_PUT(t, srcv)                                                          \
    do {                                                                          \
      if ((t) < KL) { double* p_ = slot(t);                                       \
        p_[0] = (srcv)[0]; p_[64] = (srcv)[1]; p_[128] = (srcv)[2]; p_[192] = (srcv)[3]; } \
      else acc[t] = srcv;                                                         \
    } while (0)
  if (w == 0) {
    if (a.colinfo) {                                    // column-norm summary for the rank gate
      double mn, mx, sm;
      colinfo_wave(sq, n, lane, mn, mx, sm);
      if (lane == 0) {
        a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
        if (a.hmax) a.hmax[b] = mx * mx;
        if (a.lam_out) a.lam_out[b] = (double)n;
      }
    }
  } else {
    // the scaled source tiles (the source may alias the output: everything is read before anything is written)
#pragma unroll
    for (int t = 0; t < SL; ++t) {
      if (tij[t] >= 0) {
        const int i = tij[t] & 255, j = tij[t] >> 8;
        const double scj = sc[16 * j + lc];
        v4d a0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          double v = 0.0;
          if (row < N && col < N) v = acc[t][g] * sc[row] * scj;
          if (j == i && lr + 4 * g == lc) v += td[row];
          a0[g] = v;
        }
        if (i == 0 && j == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) Dt[(lr + 4 * g) * 16 + lc] = a0[g];
        }
        RL2_PUT(t, a0);
      }
    }
  }
  CST(stp && w == 2, 3, 17, 2);
  __syncthreads();                                      // X: all source reads done before the first store
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 17, 3);
  if (__hip_atomic_load(&fl[FL_BAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {   // uniform: to the QR tree
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = 0.0;
      if (a.path_out) a.path_out[b] = a.n + 1;
      if (a.k2_out && !a.cert_shift) a.k2_out[b] = 0.0;
    }
    return;
  }
  // zeros outside the factor: strictly lower tiles, and everything beyond 16 NT (sub-matrix use)
  for (int r = w; r < (a.skip_zero ? 0 : NPAD); r += GR_NW) {
    const int cend = (r < 16 * NT) ? (r & ~15) : NPAD;
    for (int c = lane; c < cend; c += WAVE) Gb[(unsigned)(r * NPAD + c)] = 0.0;
    if (r < 16 * NT)
      for (int c = 16 * NT + lane; c < NPAD; c += WAVE) Gb[(unsigned)(r * NPAD + c)] = 0.0;
  }
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 17, 4);
  if (w == 0) {
    double pmin = 1.0;
    for (int kb = 0; kb < NT; ++kb) {
      double* DtC = Dt + (kb & 1) * 256;
      double* RiC = Ri + (kb & 1) * 256;
      if (kb > 0) spin_ge(&fl[FL_DIAG], kb);            // the updated diagonal tile kb is in DtC
      CST(stp, 2, kb, 0);
      pmin = chol16_blocked3(DtC, RiC, n - 16 * kb, pmin);   // (chol16.h; ends with lgkmcnt(0))
      if (lane == 0) __hip_atomic_store(&fl[FL_RINV], kb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      CST(stp, 2, kb, 1);
      if (a.rinv) {                                     // kept for the conditioning certificate
        double* ro = a.rinv + ((long)b * NTP + kb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = RiC[q * 64 + lane];
      }
    }
    if (lane == 0) pminsh = pmin;
  } else {
    for (int kb = 0; kb < NT; ++kb) {
      const double* DtC = Dt + (kb & 1) * 256;
      const double* RiC = Ri + (kb & 1) * 256;
      double* RrowC = Rrow + (size_t)(kb & 1) * NTP * 256;
      // this wave's slots of row block kb: tiles q0 .. q0 + NT - kb - 1
      const int q0 = kb * NT - kb * (kb - 1) / 2;
      const int t_lo = (q0 - ww + NWK - 1 + NWK) / NWK - 1;          // ceil((q0 - ww) / 7), q0 - ww >= -6
      const int t_hi = (q0 + NT - kb - 1 - ww + NWK) / NWK - 1;      // floor(.. / 7)
      const int t_one = ((q0 + 1 - ww) % NWK == 0 && kb + 1 < NT) ? (q0 + 1 - ww) / NWK : -1;   // slot of (kb, kb+1)
      const int q1 = q0 + NT - kb;                                     // tile (kb+1, kb+1)
      const int t_dia = ((q1 - ww) % NWK == 0 && kb + 1 < NT) ? (q1 - ww) / NWK : -1;
      CST(stp && w == 2, 3, kb, 5);
      spin_ge(&fl[FL_RINV], kb + 1);                    // R'_kk and its inverse are in LDS
      CST(stp && w == 2, 3, kb, 0);
      double rf[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) rf[s_] = RiC[(4 * s_ + lr) * 16 + lc];
      // R'_{kb,j} = R'_{kb,kb}^-T S_j -> memory (as R = R' D^-1) and the LDS row buffer
      auto solve_tile = [&](int t, const v4d& S) {
        const int j = kb + (ww + NWK * t - q0);
        v4d X = {0.0, 0.0, 0.0, 0.0};
        if (j == kb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) X[g] = DtC[(lr + 4 * g) * 16 + lc];
        } else {
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) X = gmfma(rf[s_], S[s_], X);
        }
        const double sj = sq[16 * j + lc];
        const double dj = dl[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * kb + lr + 4 * g;
          const int colg = 16 * j + lc;
          double val = X[g] * sj;
          if (row >= n || row > colg || colg > n) val = 0.0;
          Gb[(unsigned)(row * NPAD + colg)] = val;
          // the operand of the trailing updates is what the left-looking kernel reads back: the STORED
          // entry times its column's equilibration — all factor kernels agree bit for bit
          if (j != kb) RrowC[j * 256 + (lr + 4 * g) * 16 + lc] = val * dj;
        }
      };
      // 1. the critical tile (kb, kb + 1) first
      if (t_one >= 0) {
        CST(stp, 1, kb, 0);
#pragma unroll
        for (int t = 0; t < SL; ++t) {
          if (t == t_one) { v4d S; RL2_GET(t, S); solve_tile(t, S); }
        }
        raise_flag(&fl[FL_ROW1], kb + 1, lane);
        CST(stp, 1, kb, 1);
      }
      // 2. the next diagonal tile: S_{kb+1,kb+1} -= R'_{kb,kb+1}^T R'_{kb,kb+1} -> Dt of the other parity
      if (t_dia >= 0) {
        spin_ge(&fl[FL_ROW1], kb + 1);
        CST(stp, 1, kb, 2);
        const double* Ra = RrowC + (kb + 1) * 256 + lr * 16 + lc;
        double* DtN = Dt + ((kb + 1) & 1) * 256;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
          if (t == t_dia) {
            v4d S;
            RL2_GET(t, S);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) S = gmfma(-Ra[64 * s_], Ra[64 * s_], S);
            RL2_PUT(t, S);
#pragma unroll
            for (int g = 0; g < 4; ++g) DtN[(lr + 4 * g) * 16 + lc] = S[g];
          }
        }
        raise_flag(&fl[FL_DIAG], kb + 1, lane);
        CST(stp, 1, kb, 3);
      }
      if (cert && (q0 - ww) % NWK == 0) {
        // this wave owns the diagonal tile: y of the block by forward substitution with |R'_kk| — once every worker
        // has added its share of the row block before (FL_YPUB) — and the tile's column sums
        spin_ge(&fl[FL_YPUB], NWK * kb);
        const int i_ = lane & 15, gi = 16 * kb + i_;
        double Dc[16], cs_ = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
          const double v_ = fabs(DtC[s_ * 16 + i_]);
          Dc[s_] = s_ < i_ ? v_ : 0.0;
          if (s_ <= i_) cs_ += v_;
        }
        const double dg_ = DtC[i_ * 16 + i_];
        const bool live_ = gi < n && dg_ > 0.0;
        const double iv_ = live_ ? 1.0 / dg_ : 0.0;
        double r_ = live_ ? yv[gi] : 0.0;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
          const double ys_ = read_lane(r_ * iv_, s_);
          if (i_ > s_) r_ = fma(Dc[s_], ys_, r_);
        }
        if (lane < 16 && gi < n) { yv[gi] = r_ * iv_; csum[gi] += cs_; }
        raise_flag(&fl[FL_YB], kb + 1, lane);
      }
      CST(stp && w == 2, 3, kb, 1);
      // 3. the other tiles of the row block
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (t >= t_lo && t <= t_hi && t != t_one) { v4d S; RL2_GET(t, S); solve_tile(t, S); }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(&fl[FL_PUB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (cert) {                                       // (behind the counter: nobody's trailing update waits for this)
        // y_j += sum_s |R'_{kb,j}[s][.]| y_s and the column sums, for this wave's tiles of the row block (from the
        // row buffer; lane (lr, lc) holds rows lr + 4 g of column lc, the four lane rows are added in a fixed tree)
        if (kb + 1 < NT) spin_ge(&fl[FL_YB], kb + 1);
        for (int t = t_lo; t <= t_hi && kb + 1 < NT; ++t) {
          const int j = kb + (ww + NWK * t - q0);
          if (j == kb) continue;
          double p_ = 0.0, q_ = 0.0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const double a_ = fabs(RrowC[j * 256 + (lr + 4 * g) * 16 + lc]);
            p_ = fma(a_, yv[16 * kb + lr + 4 * g], p_);
            q_ += a_;
          }
          p_ += __shfl_xor(p_, 16, WAVE); q_ += __shfl_xor(q_, 16, WAVE);
          p_ += __shfl_xor(p_, 32, WAVE); q_ += __shfl_xor(q_, 32, WAVE);
          if (lr == 0 && 16 * j + lc < n) { yv[16 * j + lc] += p_; csum[16 * j + lc] += q_; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&fl[FL_YPUB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      CST(stp && w == 2, 3, kb, 2);
      if (kb + 1 >= NT) break;
      spin_ge(&fl[FL_PUB], NWK * (kb + 1));             // every worker's tiles of the row block are in RrowC
      CST(stp && w == 2, 3, kb, 3);
      // 4. trailing update of this wave's tiles (ascending rows; the next diagonal tile is done)
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (t > t_hi && t != t_dia && tij[t] >= 0) {
          const int i = tij[t] & 255, j = tij[t] >> 8;
          const double* Ra = RrowC + i * 256 + lr * 16 + lc;
          const double* Rb = RrowC + j * 256 + lr * 16 + lc;
          v4d S;
          RL2_GET(t, S);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) S = gmfma(-Ra[64 * s_], Rb[64 * s_], S);
          RL2_PUT(t, S);
        }
      }
      CST(stp && w == 2, 3, kb, 4);
    }
  }
#undef RL2_GET
#undef RL2_PUT
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 18, 0);
  __syncthreads();
  CST(stp && (w == 0 || w == 2), w == 0 ? 2 : 3, 18, 1);
  if (cert && w == 0) {
    double ym_ = 0.0, r1_ = 0.0;
    for (int j = lane; j < n; j += WAVE) {
      ym_ = yv[j] > ym_ ? yv[j] : ym_;
      r1_ = csum[j] > r1_ ? csum[j] : r1_;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double oy = __shfl_xor(ym_, o, WAVE), orr = __shfl_xor(r1_, o, WAVE);
      ym_ = oy > ym_ ? oy : ym_; r1_ = orr > r1_ ? orr : r1_;
    }
    if (lane == 0) { a.cert_ym[b] = ym_; a.cert_r1[b] = r1_; }
  }
  if (tid == 0 && a.fb_mask) {
    const bool fail = !(pminsh >= (a.pivot_floor > 0.0 ? a.pivot_floor : 1.0 / GRAM_K2_MAX));
    if (a.pmin_out && !a.cert_shift) a.pmin_out[b] = pminsh;
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
    if (fail && !a.cert_shift && a.k2_out) a.k2_out[b] = 0.0;      // (no bound for this factorisation)
    if (a.cert_shift) {
      a.cert_flag[b] = 0;
      if (!fail && a.k2_out) a.k2_out[b] = a.k2_max;    // proven: kappa_2 <= Lambda / tau
    }
  }
}

template <int SL, int KL, bool CERT>
__global__ __launch_bounds__(GR_NT, 2) void gram_chol_rl2_kernel(GramCholArgs a) {
  extern __shared__ double sh[];
  __shared__ double pminsh;
  __shared__ int fl[8];                                 // FL_*: hand-over flags and the publication counter
  const int pidx = (int)blockIdx.x;
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  chol_rl2_body<SL, KL, CERT>(a, b, sh, fl, pminsh);
}

// ---- certificate, stage 0: the comparison-matrix bound (two triangular solves instead of an inverse) ----
// For triangular T, |T^-1| <= M(T)^-1 entrywise, M(T) the comparison matrix (diagonal |t_ii|, off-diagonal
// -|t_ij|; Higham, ASNA 8.2), so  ||R'^-1||_inf <= max_i (M(R')^-1 e)_i  and  ||R'^-1||_1 <= max_j (M(R')^-T e)_j :
// an O(n^2) PROVEN bound on the pair ||Y||_1 ||Y||_inf of the stage below, which needs the O(n^3) explicit
// inverse.  It grows like exp(sum of the off-diagonal mass), so it settles what is well conditioned by a margin
// (Gaussian 4096 x 256: 1600 against the inverse's 430; the bench's bounded problems) and leaves everything else
// to gram_cond_kernel, which then finds the problem flagged (cert_done) and leaves at once.  With R' = T diag(dl):
// M(R') z = e  <=>  M(T) w = e, z = w / dl;   M(R')^T y = e  <=>  M(T)^T y = 1 / dl.  All terms are non-negative
// (no cancellation; the result is inflated by 1e-9 for the rounding of <= 2 n additions per entry).
__global__ __launch_bounds__(TRI_NT) void gram_cert0_kernel(GramCholArgs a) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // (GramCholArgs::unsettled: problems this launch does NOT finish — certified here AND through the rank gate's sure
  //  case, as the N <= 80 factor kernel counts them: zero means the rest of the gate has nothing to do)
  auto unsettle = [&]() { if (tid == 0 && a.unsettled) atomicAdd(a.unsettled, 1); };
  if (a.mask && a.mask[b] <= 1) { unsettle(); return; }
  if (a.fb_mask[b] != 0) { unsettle(); return; }        // already failed on a pivot
  const int NPAD = a.NPAD;
  const int n = a.ncols_dev ? a.ncols_dev[b] - 1 : a.n;
  if (n <= 0) { unsettle(); return; }
  const double* T = a.G + (long)b * NPAD * NPAD;
  double* x = sh;                        // [NPAD] M(T)^-1 e
  double* y = x + NPAD;                  // [NPAD] M(T)^-T (1 / dl)
  double* invd = y + NPAD;               // [NPAD]
  double* dl = invd + NPAD;              // [NPAD]
  double* pfbuf = dl + NPAD;             // [2 * 16 * NPAD] DMA staging of the solves
  double* rowsR = pfbuf;                 // [NPAD] row sums (after the solves: two workgroups per CU need <= 80 KB each)
  double* rowsB = pfbuf + 32 * NPAD;     // [NPAD] row sums gathered DURING the backward solve
  for (int j = tid; j < NPAD; j += TRI_NT) dl[j] = a.dsc[(long)b * NPAD + j];
  __syncthreads();
  const bool stp = b == 0; (void)stp;
  CST(stp && w == 0, 0, 18, 0);
  if (a.cert_ym && a.cert_ym[b] > 0.0) {
    // The factor kernel has done the transposed solve and the column sums on its way (GramCholArgs::cert_ym):
    // what is left is ONE pass over the factor — the backward solve M(T) x = e by column panels, each thread adding
    // the panel's share of its row's |T_ij| / ||J_j|| sum while the panel is in LDS.
    const double ym = a.cert_ym[b], r1 = a.cert_r1[b];
    CST(stp && w == 0, 0, 18, 1);
    tri_invdiag(T, n, NPAD, invd);
    for (int i = tid; i < NPAD; i += TRI_NT) { x[i] = 1.0; rowsB[i] = 0.0; }
    __syncthreads();
    CST(stp && w == 0, 0, 18, 2);
    {
      const int nblk = (n + 15) / 16;
      const int bsz = 16 * NPAD;
      int cur = 0;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      tri_pf_issue_upper<TRI_NT>(T, NPAD, (nblk - 1) * 16, pfbuf);
      for (int kb = nblk - 1; kb >= 0; --kb) {
        const int c0 = kb * 16;
        const int bs = (n - c0 < 16) ? n - c0 : 16;
        const double* bq = pfbuf + cur * bsz;
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 2);
        lds_barrier();                                   // every wave's pieces have landed
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 3);
        if (kb > 0) tri_pf_issue_upper<TRI_NT>(T, NPAD, c0 - 16, pfbuf + (cur ^ 1) * bsz);
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 4);
        // (all LDS operands of a phase are requested before the first is used: read -> wait -> fma sixteen times in a row
        //  cost 0.9 us per phase, tools/cert0_stamps.py)
        const unsigned xb_ = lds_addr(x) + 8u * (unsigned)c0, db_ = lds_addr(dl) + 8u * (unsigned)c0;
        if (tid < 64) {                                  // wave 0 (lanes >= 16 are idle copies)
          const int i = tid & 15;
          double bv[16], dv[16], D[16], rs = 0.0;
          tri_v2d bt[8];
          tri_pf_upper_issue(bt, bq, c0 + i);
          static_for<0, 16>([&](auto is) { constexpr int s_ = decltype(is)::value; lds_read64_off<8 * s_>(dv[s_], db_); });
          double r = (i < bs) ? x[c0 + i] : 0.0;
          const double iv = (i < bs) ? invd[c0 + i] : 0.0;
          tri_pf_upper_wait(bt, bv);
          asm volatile("" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dv[3]), "+v"(dv[4]), "+v"(dv[5]), "+v"(dv[6]), "+v"(dv[7]),
                            "+v"(dv[8]), "+v"(dv[9]), "+v"(dv[10]), "+v"(dv[11]), "+v"(dv[12]), "+v"(dv[13]), "+v"(dv[14]),
                            "+v"(dv[15]));
#pragma unroll
          for (int s_ = 0; s_ < 16; ++s_) {
            const double av = fabs(bv[s_]);
            D[s_] = (i < bs && s_ < bs && s_ > i) ? -av : 0.0;
            if (i < bs && s_ < bs && s_ >= i) rs = fma(av, dv[s_], rs);
          }
          CST(stp && w == 0, 3, kb, 0);
#pragma unroll
          for (int s_ = 15; s_ >= 0; --s_) {
            const double xs = read_lane(r * iv, s_);
            if (i < s_) r = fma(-D[s_], xs, r);
          }
          asm volatile("" : "+v"(r));
          CST(stp && w == 0, 3, kb, 1);
          if (tid < bs) { x[c0 + tid] = r * iv; rowsB[c0 + tid] += rs; }
        }
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 5);
        lds_barrier();
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 6);
        if (tid < c0) {                                   // rows above the block (c0 <= 256 = TRI_NT: one row per thread)
          double xv[16], dv[16];
          static_for<0, 16>([&](auto is) { constexpr int s_ = decltype(is)::value; lds_read64_off<8 * s_>(xv[s_], xb_); });
          static_for<0, 16>([&](auto is) { constexpr int s_ = decltype(is)::value; lds_read64_off<8 * s_>(dv[s_], db_); });
          for (int i = tid; i < c0; i += TRI_NT) {
            double rv[16];
            tri_v2d rt[8];
            tri_pf_upper_issue(rt, bq, i);
            tri_pf_upper_wait(rt, rv);
            asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7]),
                              "+v"(xv[8]), "+v"(xv[9]), "+v"(xv[10]), "+v"(xv[11]), "+v"(xv[12]), "+v"(xv[13]), "+v"(xv[14]),
                              "+v"(xv[15]), "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dv[3]), "+v"(dv[4]), "+v"(dv[5]),
                              "+v"(dv[6]), "+v"(dv[7]), "+v"(dv[8]), "+v"(dv[9]), "+v"(dv[10]), "+v"(dv[11]), "+v"(dv[12]),
                              "+v"(dv[13]), "+v"(dv[14]), "+v"(dv[15]));
            double acc = 0.0, rs = 0.0;
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
              const double av = fabs(rv[s_]);
              acc = fma(-av, (s_ < bs) ? xv[s_] : 0.0, acc);
              if (s_ < bs) rs = fma(av, dv[s_], rs);
            }
            x[i] -= acc;
            rowsB[i] += rs;
          }
        }
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 7);
        lds_barrier();
        cur ^= 1;
        CST(stp && (w == 0 || w == 2), w == 0 ? 1 : 2, kb, 0);
      }
    }
    CST(stp && w == 0, 0, 18, 3);
    double zm = 0.0, rinf = 0.0;
    for (int i = tid; i < n; i += TRI_NT) { zm = nanmax2(zm, x[i] / dl[i]); rinf = fmax(rinf, rowsB[i]); }
    zm = block_max(zm, red);
    rinf = block_max(rinf, red);
    const double kmax = a.k2_max > 0.0 ? a.k2_max : GRAM_K2_MAX;
    const double k2 = (r1 * rinf) * (zm * ym) * (1.0 + 1.0e-9);
    const bool passed = k2 <= kmax;                     // (NaN / inf fail)
    // for a problem left open: Lambda_0 for the third stage, or "hopeless" — lambda_min(C) <= min_j r'_jj^2 and
    // lambda_max(C) >= 1, so kappa_2(C) >= 1 / min_j r'_jj^2 (r'_jj = T_jj / ||J_j||: 1 / (invd_j / dl_j))
    double pinv = 0.0, emax = 0.0;
    if (a.cert_open && !passed) {
      const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
      for (int i = tid; i < n; i += TRI_NT) {
        const double t = invd[i] / dl[i];
        pinv = nanmax2(pinv, t * t);
        if (edv) emax = nanmax2(emax, fabs(edv[i]));
      }
      pinv = block_max(pinv, red);
      emax = block_max(emax, red);
    }
    if (tid == 0) {
      a.cert_done[b] = passed ? 1 : 0;
      if (passed) {
        if (a.k2_out) a.k2_out[b] = k2;
        if (a.lam_out) a.lam_out[b] = fmin(r1 * rinf, (double)n);
      }
      // TRF finish (GramCholArgs::lmfin): the `sure` branch of lm_gate_kernel, same expressions
      bool finished = false;
      if (a.lmfin.fast && passed && a.colinfo && a.lmfin.enable != 0 && a.lmfin.m >= n) {
        const double mn = a.colinfo[2 * (long)b], sm = a.colinfo[2 * (long)b + 1];
        const double smin_lb = GRAM_SMIN_PROVEN * mn, smax_ub = sqrt(sm);
        if (is_finite(sm) && sm > 0.0 && smin_lb > LM_GATE_MARGIN * LM_EPS * a.lmfin.m * smax_ub) {
          a.lmfin.fast[b] = 1;
          a.lmfin.ncols_jac[b] = 0;
          a.lmfin.sc[(long)b * 16 + SC_SMAX] = smax_ub;
          a.lmfin.sc[(long)b * 16 + SC_SMIN] = smin_lb;
          a.lmfin.st[(long)b * 4 + ST_PHASE] = LM_IDLE;
          finished = true;
        }
      }
      if (!finished && a.unsettled) atomicAdd(a.unsettled, 1);
      CST(stp, 0, 18, 4);
      if (a.cert_open) {
        const double lam0 = fmin(r1 * rinf, (double)n);
        // kappa_2 >= pinv: beyond the gate the problem is hopeless.  Otherwise the note depends on what the system IS:
        // with a Coleman-Li block (E != 0: a variable near a bound in its descent direction) the comparison-matrix
        // bound fails long before the system is ill conditioned, and the norm stage settles such a problem more cheaply
        // than a factorisation (bench, bounded mix: certificate 0.26 against 0.36 ms); a pure Jacobian system (E = 0)
        // that fails it is usually near or beyond the gate, where the norm stage — 3.5 ... 13 of overestimate — cannot
        // decide and the third stage is where the problem ends up anyway (unbounded mix: 0.84 -> 0.56 ms).
        double note = 0.0;
        if (!passed && is_finite(lam0) && lam0 >= 1.0) {
          if (!(pinv <= kmax)) note = -1.0;
          else if (emax == 0.0) note = lam0;
        }
        a.cert_open[b] = note;
      }
    }
    return;
  }
  if (a.cert_open && tid == 0) a.cert_open[b] = 0.0;    // (four-pass form: the norm stage below keeps its own counsel)
  unsettle();                                           // (... and the rank gate's launch finishes what passes here)
  // the two comparison solves FIRST: ||R'||_1 ||R'||_inf >= lambda_max(C) >= 1, so a product of the two maxima
  // beyond the gate already decides "not settled here" and the norm passes are skipped
  tri_invdiag(T, n, NPAD, invd);
  for (int i = tid; i < n; i += TRI_NT) { x[i] = 1.0; y[i] = 1.0 / dl[i]; }
  __syncthreads();
  tri_solve_upper_pf<TRI_NT, true>(T, n, NPAD, invd, x, pfbuf);
  tri_solve_upper_t_pf<TRI_NT, true>(T, n, NPAD, invd, y, pfbuf);
  double zm = 0.0, ym = 0.0;
  for (int i = tid; i < n; i += TRI_NT) { zm = nanmax2(zm, x[i] / dl[i]); ym = nanmax2(ym, y[i]); }
  zm = block_max(zm, red);
  ym = block_max(ym, red);
  const double kmax = a.k2_max > 0.0 ? a.k2_max : GRAM_K2_MAX;
  if (!(zm * ym <= kmax)) {                             // (uniform; NaN / inf included)
    if (tid == 0) a.cert_done[b] = 0;
    return;
  }
  __syncthreads();                                      // (rowsR aliases the solves' staging)
  // ||R'||_1 (thread per column) and ||R'||_inf (wave per row), as gram_cond_kernel
  double r1 = 0.0;
  for (int j = tid; j < n; j += TRI_NT) {
    double sum = 0.0;
    for (int i0 = 0; i0 <= j; i0 += 32) {
      double rv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) rv[u] = T[(long)((i0 + u <= j) ? i0 + u : j) * NPAD + j];
#pragma unroll
      for (int u = 0; u < 32; ++u) if (i0 + u <= j) sum += fabs(rv[u]);
    }
    r1 = fmax(r1, sum * dl[j]);
  }
  r1 = block_max(r1, red);
  for (int i0 = w; i0 < n; i0 += 4 * TRI_NW) {
    double rv[4][4], dv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = (i0 + TRI_NW * q < n) ? i0 + TRI_NW * q : n - 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = i + lane + WAVE * c;
        const int jc = j < n ? j : n - 1;
        rv[q][c] = T[(long)i * NPAD + jc];
        dv[q][c] = dl[jc];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + TRI_NW * q;
      double sum = 0.0;
      for (int c = 0; c < 4; ++c)
        if (i < n && i + lane + WAVE * c < n) sum += fabs(rv[q][c]) * dv[q][c];
      // (n <= 256 per pass of four 64-lane chunks; wider rows: the remaining chunks)
      for (int j = i + lane + WAVE * 4; i < n && j < n; j += WAVE) sum += fabs(T[(long)i * NPAD + j]) * dl[j];
      sum = wave_sum(sum);
      if (lane == 0 && i < n) rowsR[i] = sum;
    }
  }
  __syncthreads();
  double rinf = 0.0;
  for (int i = tid; i < n; i += TRI_NT) rinf = fmax(rinf, rowsR[i]);
  rinf = block_max(rinf, red);
  const double k2 = (r1 * rinf) * (zm * ym) * (1.0 + 1.0e-9);
  const bool passed = k2 <= kmax;                       // (NaN / inf fail)
  if (tid == 0) {
    a.cert_done[b] = passed ? 1 : 0;
    if (passed) {
      if (a.k2_out) a.k2_out[b] = k2;
      if (a.lam_out) a.lam_out[b] = fmin(r1 * rinf, (double)n);
    }
  }
}

// ---- conditioning gate: a PROVEN bound on kappa_2 of the equilibrated system -----------------------
// The normal-equations path loses kappa_2(C) eps where C = R'^T R' is the equilibrated system matrix
// (unit diagonal) the step is solved from.  An estimate of sigma_min(R') by inverse iteration is a
// LOWER bound on ||R'^-1||, i.e. it can only err on the unsafe side.  This kernel computes an UPPER
// bound instead, from the explicit inverse:
//     Y = R'^-T   (lower triangular; 16 x 16 tiles by FP64 MFMA, the inverses of the diagonal tiles
//                  come from the Cholesky kernel:  Y_ii = R'_ii^-T,
//                  Y_ij = -R'_ii^-T sum_{k=j}^{i-1} R'_ki^T Y_kj   for j < i)
//     1 / lambda_min(C) = ||Y||_2^2 <= ||Y||_1 ||Y||_inf ,   lambda_max(C) = ||R'||_2^2 <= ||R'||_1 ||R'||_inf
//     K2 = ||R'||_1 ||R'||_inf ||Y||_1 ||Y||_inf  >=  kappa_2(C)
// (all four norms are exact sums of absolute values, accumulated in a fixed order) and keeps the
// problem on the normal-equations path only if K2 <= GRAM_K2_MAX.  DESIGN.md 3.0 has the error bound
// this gives for the step.  NWP waves per problem as in gram_chol_kernel.
template <int NWP>
__global__ __launch_bounds__(GR_NT, 4) void gram_cond_kernel(GramCholArgs a) {
  constexpr int PT = WAVE * NWP;
  constexpr int PPW = GR_NW / NWP;
  constexpr int UMAX = (NWP == 8) ? 3 : 5;              // column tiles of a row block per wave
  extern __shared__ double sh_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int pslot = wv / NWP;
  const int pidx = (int)blockIdx.x * PPW + pslot;
  if (pidx >= a.count) return;                          // (NWP == 1 only: wave-uniform)
  const int b = pidx;
  const int tid = (int)threadIdx.x % PT, lane = tid & 63;
  const int w = wv % NWP;
  const int lr = lane >> 4, lc = lane & 15;
  if (a.mask && a.mask[b] <= 1) return;
  if (a.fb_mask[b] != 0) return;                        // already failed on a pivot
  if (a.cert_done && a.cert_done[b]) return;            // already proven inside the factor kernel (N <= 80)
  const int NPAD = a.NPAD;
  const int n = a.ncols_dev ? a.ncols_dev[b] - 1 : a.n;
  if (n <= 0) return;
  const int NTn = (n + 15) / 16;
  auto psync = [&]() {
    if (NWP == 8) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };
  double* sh = sh_all + (size_t)pslot * (6 * (size_t)NPAD + 16 * NWP + 64);
  double* dl = sh;                      // [NPAD] column scales: R'[i][j] = T[i][j] dl[j]
  double* cs4 = dl + NPAD;              // [NPAD][4] column sums of |Y|, one slot per lane row
  double* rs = cs4 + 4 * NPAD;          // [NWP][16] row-sum partials of the current block row
  double* vals = rs + 16 * NWP;         // [64] reduction scratch
  double* rowsR = vals + 64;            // [NPAD] row sums of |R'|
  const double* T = a.G + (long)b * NPAD * NPAD;
  double* Y = a.ywork + (long)b * NPAD * NPAD;
  const double* Rinv = a.rinv + (long)b * (NPAD / 16) * 256;
  for (int j = tid; j < NPAD; j += PT) dl[j] = a.dsc[(long)b * NPAD + j];
  psync();
  auto reduce_max = [&](double v) -> double {           // max over the threads of this problem
    v = wave_max(v);
    if (NWP == 1) return v;
    psync();
    if (lane == 0) vals[w] = v;
    psync();
    double t = vals[0];
    for (int q = 1; q < NWP; ++q) t = fmax(t, vals[q]);
    return t;
  };
  // Stage 0 has been here (N > 80, GramCholArgs::cert_open): it could not settle the problem, but it left
  // Lambda_0 = min(||R'||_1 ||R'||_inf, n) >= lambda_max(C) — or the verdict "hopeless" from the smallest pivot.  The
  // explicit inverse below would only produce looser bounds than the shifted factorisation of the third stage proves
  // anyway: the problem goes there directly, with tau from Lambda = min(Lambda_0, ||C||_F) (one pass over the Gram).
  if constexpr (NWP == 8) {
    const double open0 = a.cert_open ? a.cert_open[b] : 0.0;
    if (open0 != 0.0) {
      const double kmax = a.k2_max > 0.0 ? a.k2_max : GRAM_K2_MAX;
      double lam = open0;
      if (open0 > 0.0 && a.cert_flag) {
        const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
        auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
        const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
        const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
        const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
        double* scl = cs4;                  // [NPAD] cs_j dl_j
        double* tdl = cs4 + NPAD;           // [NPAD] e_j^2 dl_j^2
        for (int j = tid; j < NPAD; j += PT) {
          const double cs = (csv && j < n) ? csv[j] : 1.0;
          const double ej = (edv && j < n) ? edv[j] : 0.0;
          scl[j] = cs * dl[j];
          tdl[j] = (ej * ej) * dl[j] * dl[j];
        }
        psync();
        double cf = 0.0;                    // this wave's share of ||C||_F^2 (fixed order)
        int q = 0;
        for (int j = 0; j < NTn; ++j) {
          for (int i = 0; i <= j; ++i, ++q) {
            if (q % NWP != w) continue;     // (wave-uniform)
            double c2 = 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
              double v = 0.0;
              if (row < n && col < n) {
                int sr_ = src(row), sc_ = src(col);
                if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }
                v = Gs[(long)sr_ * NPAD + sc_] * scl[row] * scl[col];
                if (row == col) v += tdl[row];
              }
              c2 = fma(v, v, c2);
            }
            cf = fma((i == j) ? 1.0 : 2.0, wave_sum(c2), cf);
          }
        }
        psync();
        if (lane == 0) vals[w] = cf;
        psync();
        cf = 0.0;
        for (int qq = 0; qq < NWP; ++qq) cf += vals[qq];
        lam = fmin(lam, sqrt(cf));                        // lambda_max(C) <= ||C||_F
      }
      if (tid == 0) {
        if (a.k2_out) a.k2_out[b] = __builtin_inf();     // (no proven bound from here; the third stage writes k2_max)
        if (open0 > 0.0 && a.cert_flag && lam >= 1.0 && is_finite(lam)) {
          if (a.lam_out) a.lam_out[b] = lam;
          a.cert_tau[b] = lam / kmax;
          a.cert_flag[b] = 1;
        } else {
          a.fb_mask[b] = a.n + 1;
          if (a.path_out) a.path_out[b] = a.n + 1;
          { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
        }
      }
      return;
    }
  }
  // ---- ||R'||_1 (thread per column) and ||R'||_inf (wave per row) ----
  double r1 = 0.0;
  for (int j = tid; j < n; j += PT) {
    // (32 rows of loads in flight per pass: the passes are serialised by their waits, and the longest
    //  column has n rows; same order of additions as a plain loop over the rows)
    double sum = 0.0;
    for (int i0 = 0; i0 <= j; i0 += 32) {
      double rv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) rv[u] = T[(long)((i0 + u <= j) ? i0 + u : j) * NPAD + j];
#pragma unroll
      for (int u = 0; u < 32; ++u) if (i0 + u <= j) sum += fabs(rv[u]);
    }
    r1 = fmax(r1, sum * dl[j]);
  }
  r1 = reduce_max(r1);
  if constexpr (NWP == 1) {
    // one wave per problem: four rows at a time, one per 16-lane group (64 sequential wave
    // reductions at n = 64 were 20 of this kernel's 46 us)
    for (int i0 = 0; i0 < n; i0 += 4) {
      const int i = i0 + lr;
      double sum = 0.0;
      if (i < n)
        for (int j = i + lc; j < n; j += 16) sum += fabs(T[(long)i * NPAD + j]) * dl[j];
      sum = row16_sum(sum);
      if (lc == 0 && i < n) rowsR[i] = sum;
    }
  } else {
    // four rows of a wave per pass, their loads issued together (clamped, unconditional; n <= 256: at
    // most four 64-lane chunks per row); per lane the same additions in the same order as row by row
    for (int i0 = w; i0 < n; i0 += 4 * NWP) {
      double rv[4][4], dv[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (i0 + NWP * q < n) ? i0 + NWP * q : n - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int j = i + lane + WAVE * c;
          const int jc = j < n ? j : n - 1;
          rv[q][c] = T[(long)i * NPAD + jc];
          dv[q][c] = dl[jc];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + NWP * q;
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (i < n && i + lane + WAVE * c < n) sum += fabs(rv[q][c]) * dv[q][c];
        sum = wave_sum(sum);
        if (lane == 0 && i < n) rowsR[i] = sum;
      }
    }
  }
  psync();
  double rinf = 0.0;
  for (int i = tid; i < n; i += PT) rinf = fmax(rinf, rowsR[i]);
  rinf = reduce_max(rinf);
  // ---- Y = R'^-T by block rows; row and column sums of |Y| on the way ----
  double csum[UMAX];
#pragma unroll
  for (int u = 0; u < UMAX; ++u) csum[u] = 0.0;
  double rmax = 0.0;                                    // threads 0..15: max over block rows of "their" row
  for (int i = 0; i < NTn; ++i) {
    double rsum[4] = {0.0, 0.0, 0.0, 0.0};
    const double* Ri = Rinv + (long)i * 256;
    const double dli = dl[16 * i + lc];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = w + NWP * u;
      if (j <= i) {
        v4d Yt = {0.0, 0.0, 0.0, 0.0};
        if (j == i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) Yt[g] = Ri[lc * 16 + lr + 4 * g];         // (R'_ii^-1)^T
        } else {
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          for (int k = j; k < i; ++k) {
            double av[4], bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const long ro = (long)(16 * k + 4 * s + lr) * NPAD;
              av[s] = T[ro + 16 * i + lc];
              bv[s] = Y[ro + 16 * j + lc];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = gmfma(av[s] * dli, bv[s], acc);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) Yt = gmfma(-Ri[(4 * s + lr) * 16 + lc], acc[s], Yt);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          const double v = (row < n && col < n) ? Yt[g] : 0.0;
          Y[(long)row * NPAD + col] = v;
          const double av_ = fabs(v);
          rsum[g] += row16_sum(av_);
          csum[u] += av_;
        }
      }
    }
    if (lc == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) rs[w * 16 + lr + 4 * g] = rsum[g];
    }
    psync();                                            // row sums in LDS; Y row block i visible
    if (tid < 16) {
      double t = 0.0;
      for (int q = 0; q < NWP; ++q) t += rs[q * 16 + tid];
      rmax = fmax(rmax, t);
    }
    psync();
  }
#pragma unroll
  for (int u = 0; u < UMAX; ++u) {
    const int j = w + NWP * u;
    if (j < NTn) cs4[(16 * j + lc) * 4 + lr] = csum[u];
  }
  psync();
  double y1 = 0.0;
  for (int c = tid; c < n; c += PT)
    y1 = fmax(y1, (cs4[4 * c] + cs4[4 * c + 1]) + (cs4[4 * c + 2] + cs4[4 * c + 3]));
  y1 = reduce_max(y1);
  const double yinf = reduce_max(tid < 16 ? rmax : 0.0);
  double k2 = (r1 * rinf) * (y1 * yinf);
  const double kmax = a.k2_max > 0.0 ? a.k2_max : GRAM_K2_MAX;
  double lam = fmin(r1 * rinf, (double)n);              // lambda_max(C) <= ||R'||_1 ||R'||_inf, <= trace(C) = n
  if (!(k2 <= kmax)) {                                   // (uniform over the problem's threads)
    // The 1- / inf-norm products overestimate kappa_2 by 10 ... 1000 (profiles/r02p_gate_calibration.txt).
    // Second, tighter proven bound for a problem they reject:  lambda_max(C) <= ||C||_F  and
    // 1 / lambda_min(C) = ||C^-1||_2 <= ||C^-1||_F  with  C^-1 = Y^T Y  formed tile by tile (MFMA; only
    // its sum of squares is kept) and C rebuilt from the source Gram with the Cholesky's own scalings.
    // Measured overestimate 4 ... 30 on the ill-conditioned families.  Sums in a fixed order.
    const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
    auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
    const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
    const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
    const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
    double* scl = cs4;                  // [NPAD] cs_j dl_j   (cs4 is free now)
    double* tdl = cs4 + NPAD;           // [NPAD] e_j^2 dl_j^2
    psync();
    for (int j = tid; j < NPAD; j += PT) {
      const double cs = (csv && j < n) ? csv[j] : 1.0;
      const double ej = (edv && j < n) ? edv[j] : 0.0;
      scl[j] = cs * dl[j];
      tdl[j] = (ej * ej) * dl[j] * dl[j];
    }
    psync();
    double cf = 0.0, zf = 0.0;          // this wave's share of ||C||_F^2 / ||C^-1||_F^2
    int q = 0;
    for (int j = 0; j < NTn; ++j) {
      for (int i = 0; i <= j; ++i, ++q) {
        if (q % NWP != w) continue;     // (wave-uniform)
        const double wgt = (i == j) ? 1.0 : 2.0;
        double c2 = 0.0;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          double v = 0.0;
          if (row < n && col < n) {
            int sr_ = src(row), sc_ = src(col);
            if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }
            v = Gs[(long)sr_ * NPAD + sc_] * scl[row] * scl[col];
            if (row == col) v += tdl[row];
          }
          c2 = fma(v, v, c2);
        }
        for (int k = j; k < NTn; ++k) {
          double av[4], bv[4];
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const long ro = (long)(16 * k + 4 * s2 + lr) * NPAD;
            av[s2] = Y[ro + 16 * i + lc];
            bv[s2] = Y[ro + 16 * j + lc];
          }
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc = gmfma(av[s2], bv[s2], acc);
        }
        const double z2 = (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
        cf = fma(wgt, wave_sum(c2), cf);
        zf = fma(wgt, wave_sum(z2), zf);
      }
    }
    if (NWP > 1) {
      psync();
      if (lane == 0) { vals[w] = cf; vals[8 + w] = zf; }
      psync();
      cf = 0.0; zf = 0.0;
      for (int qq = 0; qq < NWP; ++qq) { cf += vals[qq]; zf += vals[8 + qq]; }
    }
    const double k2f = sqrt(cf) * sqrt(zf);
    if (k2f < k2) k2 = k2f;
    lam = fmin(lam, sqrt(cf));                          // lambda_max(C) <= ||C||_F
  }
  if (tid == 0) {
    if (a.k2_out) a.k2_out[b] = k2;
    if (a.lam_out && lam >= 1.0) a.lam_out[b] = lam;
    if (!(k2 <= kmax)) {                                 // (NaN fails)
      // (a bound 64x above the gate is beyond what its overestimate — 3.5 ... 13 measured, 30 at the worst — can
      //  explain: such a problem is rejected here, without the third stage's factorisation)
      if (a.cert_flag && is_finite(k2) && lam >= 1.0 && k2 <= 64.0 * kmax) {
        // The norm bounds overestimate kappa_2 by 3.5 ... 13 where they decide: leave the verdict to the third
        // stage, a Cholesky factorisation of C - tau I with tau = Lambda / k2_max (launch_gram_cert_shift) —
        // it succeeds iff lambda_min(C) > tau, which proves kappa_2(C) <= Lambda / tau = k2_max.
        a.cert_tau[b] = lam / kmax;
        a.cert_flag[b] = 1;
      } else {
        a.fb_mask[b] = a.n + 1;
        if (a.path_out) a.path_out[b] = a.n + 1;
        { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
      }
    }
  }
}

double gram_k2_max(long long m_total, double tighter) {
  auto acc = [](double m) {
    const double chunk = m > 131072.0 ? 1024.0 : 2048.0;           // (gram_chunks: a function of m alone)
    const double rows = m < chunk ? m : chunk;
    return sqrt(rows) + sqrt(ceil(m / chunk));
  };
  const double f = acc(4096.0) / acc((double)(m_total > 1 ? m_total : 1));
  double k = GRAM_K2_MAX * (f < 1.0 ? f : 1.0);
  if (tighter > 0.0 && tighter < k) k = tighter;                   // (option gram_k2_max: may only tighten the gate)
  return k;
}

bool gram_supported(int m, int n) {
  const int NT = (n + 1 + 15) / 16;
  return NT <= 17 && m >= n && n >= 1;
}
// Row chunks whose size is a function of m ALONE (never of the batch size): the summation order of
// a problem's Gram (and so every bit of its result) does not depend on how many problems share the
// launch.  2048 rows; 1024 for very tall problems (one 250 000 x 128 row block of BASELINE config 5:
// 245 workgroups fill the 256 CUs, 123 leave half of them idle).
hipError_t launch_gram_chol(const GramCholArgs& a_in, int B, hipStream_t s) {
  GramCholArgs a = a_in;
  a.count = B;
  const size_t per = sizeof(double) * (4 * (size_t)a.NPAD + 512);
  if (a.NPAD <= 80) {                                   // one wave per problem, eight per workgroup
    // (register-resident right-looking kernel; BLSQ_CHOL_REG = 0: the left-looking one-wave kernel)
    if (!options_or_default(a.opt).on(OPT_CHOL_REG)) {
      // (this kernel has no finish blocks: nothing is settled, nobody is done)
      hipError_t me = hipSuccess;
      if (a.unsettled) me = hipMemsetD32Async((hipDeviceptr_t)a.unsettled, 1, 1, s);
      if (me == hipSuccess && a.dog.done) me = hipMemsetAsync(a.dog.done, 0, sizeof(int) * (size_t)B, s);
      if (me != hipSuccess) return me;
      hipLaunchKernelGGL(gram_chol_kernel<1>, dim3((B + GR_NW - 1) / GR_NW), dim3(GR_NT), per * GR_NW,
                         s, a);
    } else {
      const size_t per_reg = sizeof(double) * (8 * (size_t)a.NPAD + 256 + 5 * 256 + 64 + 16);
      static std::atomic<size_t> granted[64];
      hipError_t ge = gram_grant_lds(gram_chol_reg_kernel, per_reg * REG_NW, granted);
      if (ge != hipSuccess) return ge;
      hipLaunchKernelGGL(gram_chol_reg_kernel, dim3(reg_grid(B)), dim3(REG_NT), per_reg * REG_NW,
                         s, a);
    }
  } else {
    // Right-looking register kernel, flag-driven: 0.12 ms per problem on a CU of its own (one workgroup per
    // CU) against 0.27 ms for a PAIR of problems on a CU through the left-looking kernel (its Schur-complement
    // phase waits for L2: 190 of its 275 us, tools/chol_stamps.py) — the right-looking one serves every launch
    // (512 problems: two generations, 0.241 against 0.262 ms).  BLSQ_CHOL_RL = 0 / 1 forces either;
    // BLSQ_CHOL_RL2 = 0 selects the barrier-synchronous right-looking kernel (<= 256 problems: 0.187 ms).
    // All three agree bit for bit (same operands, same order), so the choice is speed only.
    const Options& opt = options_or_default(a.opt);    // (per launch: tests compare the kernels)
    const bool rl = opt.i(OPT_CHOL_RL) != 0;           // (-1 / 1: right-looking)
    const bool rl2 = rl && opt.on(OPT_CHOL_RL2);       // 0: the barrier-synchronous right-looking kernel
    if (a.cert_ym && !rl2) {                            // (only the flag-driven kernel has a share in stage 0)
      hipError_t me = hipMemsetAsync(a.cert_ym, 0, sizeof(double) * (size_t)B, s);
      if (me != hipSuccess) return me;
    }
    if (rl2) {
      constexpr int R2_KL = 5;
      const size_t lds = sizeof(double) * (4 * (size_t)a.NPAD + 1024 + 2 * (size_t)(a.NPAD / 16) * 256 +
                                           (size_t)R2_KL * (GR_NW - 1) * 256 + 2 * (size_t)a.NPAD) +
                         sizeof(int) * (size_t)a.NPAD;
      if (a.cert_ym && !a.cert_shift) {
        static std::atomic<size_t> granted[64];
        hipError_t ge = gram_grant_lds(gram_chol_rl2_kernel<22, R2_KL, true>, lds, granted);
        if (ge != hipSuccess) return ge;
        hipLaunchKernelGGL((gram_chol_rl2_kernel<22, R2_KL, true>), dim3(B), dim3(GR_NT), lds, s, a);
      } else {
        static std::atomic<size_t> granted[64];
        hipError_t ge = gram_grant_lds(gram_chol_rl2_kernel<22, R2_KL, false>, lds, granted);
        if (ge != hipSuccess) return ge;
        hipLaunchKernelGGL((gram_chol_rl2_kernel<22, R2_KL, false>), dim3(B), dim3(GR_NT), lds, s, a);
      }
    } else if (rl) {
      constexpr int RL_KL = 8;                          // tile slots per worker wave kept in LDS
      const size_t lds = per + sizeof(double) * 256 * ((size_t)(a.NPAD / 16) + (size_t)RL_KL * (GR_NW - 1));
      static std::atomic<size_t> granted[64];
      hipError_t ge = gram_grant_lds(gram_chol_rl_kernel<22, RL_KL>, lds, granted);
      if (ge != hipSuccess) return ge;
      hipLaunchKernelGGL((gram_chol_rl_kernel<22, RL_KL>), dim3(B), dim3(GR_NT), lds, s, a);
    } else {
      hipLaunchKernelGGL(gram_chol_kernel<8>, dim3(B), dim3(GR_NT), per, s, a);
    }
  }
  return hipGetLastError();
}
hipError_t launch_gram_cert_shift(const GramCholArgs& a_in, int B, hipStream_t s) {
  if (!a_in.cert_flag || !a_in.cert_tau) return hipSuccess;
  GramCholArgs a = a_in;
  a.cert_shift = 1;
  a.G = a.ywork;                                        // (Y of the norm stage is dead by now)
  a.skip_zero = 1;
  a.pivot_floor = GRAM_CERT_PIVOT_FLOOR;
  a.dsc = nullptr; a.colinfo = nullptr; a.rinv = nullptr; a.cert_done = nullptr; a.unsettled = nullptr;
  a.dog = GramCholArgs::DogFinish{}; a.lmfin = GramCholArgs::LmFinish{};
  a.batch_list = nullptr; a.count_dev = nullptr; a.skip_path = nullptr; a.diag_sqrt = nullptr;
  a.qr_mask = nullptr; a.hmax = nullptr; a.lam_out = nullptr; a.pmin_out = nullptr;
  a.expect = B;                                         // (left-looking kernel: most workgroups leave at once)
  return launch_gram_chol(a, B, s);
}

hipError_t launch_gram_gate(const GramCholArgs& a_in, int B, hipStream_t s, bool stage0_only) {
  GramCholArgs a = a_in;
  a.count = B;
  if (stage0_only && !(a.NPAD > 80 && a.cert_done && a.dsc && a.cert_ym)) return hipErrorInvalidValue;
  // stage 0 (N > 80; the register-resident factor kernel of the small shapes carries its own first bound):
  // BLSQ_CERT0 = 0 switches it off
  if (a.NPAD > 80 && a.cert_done && a.dsc) {
    if (options_or_default(a.opt).on(OPT_CERT0)) {
      const size_t lds0 = sizeof(double) * (4 + 32 + 1) * (size_t)a.NPAD;
      static std::atomic<size_t> granted[64];
      hipError_t ge = gram_grant_lds(gram_cert0_kernel, lds0, granted);
      if (ge != hipSuccess) return ge;
      hipLaunchKernelGGL(gram_cert0_kernel, dim3(B), dim3(TRI_NT), lds0, s, a);
    } else {
      hipError_t me = hipMemsetAsync(a.cert_done, 0, sizeof(int) * (size_t)B, s);
      if (me != hipSuccess) return me;
      a.cert_open = nullptr;                            // (no stage 0: nothing for the norm stage to go by)
    }
  } else {
    a.cert_open = nullptr;
  }
  if (stage0_only) return hipGetLastError();
  const size_t per1 = sizeof(double) * (6 * (size_t)a.NPAD + 16 * 1 + 64);
  const size_t per8 = sizeof(double) * (6 * (size_t)a.NPAD + 16 * 8 + 64);
  if (a.NPAD <= 80) {                                   // one wave per problem, eight per workgroup
    hipLaunchKernelGGL(gram_cond_kernel<1>, dim3((B + GR_NW - 1) / GR_NW), dim3(GR_NT), per1 * GR_NW, s, a);
  } else {
    hipLaunchKernelGGL(gram_cond_kernel<8>, dim3(B), dim3(GR_NT), per8, s, a);
  }
  return hipGetLastError();
}

#ifdef BLSQ_CHOL_STAMPS
int chol_debug_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chol_st), sizeof(g_chol_st));
}
#endif
}  // namespace blsq
