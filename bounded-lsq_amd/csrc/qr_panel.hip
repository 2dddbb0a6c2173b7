// Blocked Householder QR of tall-skinny f64 matrices, R-factor only.
//
// Replaces, for the trust-region step path, everything the reference gets
// from LAPACK on the m x n Jacobian: the thin SVD of the augmented Jacobian
// (bounded_lsq/trf.py:264-274 -> scipy.linalg.svd/gesdd) and the least-squares
// solve (bounded_lsq/dogbox.py:197 -> numpy.linalg.lstsq/gelsd) both start
// here from  [J f] = Q [R c; 0 rho]  — one pass over J.
//
// One workgroup factors one (problem, row-leaf):
//   * left-looking over 16-column panels.  Panel k of the SOURCE is read from HBM
//     exactly once, straight into REGISTERS in the f64 MFMA accumulator layout
//     (lane (lr, lc) holds rows lr + 4 g, column lc of its wave's row tiles), and
//     stays there while all previous block reflectors are applied:
//         P -= V_j (T_j^T (V_j^T P)).
//     The accumulator layout is exactly the B-operand layout of k-step g, so the
//     panel feeds V_j^T P directly and V_j W accumulates into it in place: the
//     panel never touches LDS during the update;
//   * the V_j tiles stream HBM -> LDS by LDS-DMA (global_load_lds, 16 B / lane, no
//     VGPR staging).  Each wave stages only ITS tiles (tile t belongs to wave
//     t mod 8), so the staging needs no workgroup barrier: as soon as a tile slot has
//     fed its last MFMA of reflector j it is refilled with the tile of V_{j+1}, and the
//     wave later waits with a counted s_waitcnt vmcnt(N) for exactly that slot.  Both
//     operand shapes (V^T for the first GEMM, V for the third) are read from the same
//     LDS image — V is stored in HBM pre-swizzled (col ^ row within a tile) so both
//     reads are bank-conflict-free — and no on-chip transposes are needed;
//   * panels go in PAIRS (both in registers) so each V_j tile read serves two panels;
//   * the updated panel is then written to LDS (column-major) and factored: normally
//     from its 16x16 Gram by Cholesky-QR + Householder reconstruction (panel_cqr below:
//     two MFMA passes + a 16x16 chain on two waves, no per-column barriers); when the
//     Gram's pivots say the panel is not well conditioned, exactly, column by column,
//     with each thread's rows in registers (one fused wave-shuffle + LDS reduction per
//     column for the norm and the 15 dot products);
//   * cross-wave reduction of the 16x16 W goes through LDS in a fixed order
//     (deterministic results).
// Leaves produce (N x N) triangles; the same kernel merges stacked triangles
// (TSQR tree), factors the Coleman-Li augmented system [R D; E], the Newton systems
// [R_aug; sqrt(alpha) I] and the dogbox free-column block R[:, free].
#include <algorithm>
#include <cstdlib>

#include <atomic>

#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

// Diagnostic build only (-DBLSQ_QR_STAMPS): thread 0 of each workgroup sums the
// cycles of each phase into q.dbg[slot*8 + phase].  Never enabled in the product.
#ifdef BLSQ_QR_STAMPS
#define STAMP_DECL unsigned long long st_acc[8] = {0,0,0,0,0,0,0,0}; unsigned long long st_t = __builtin_readcyclecounter();
#define STAMP(i) { unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_t; st_t = now_; }
#define STAMP_OUT if (tid == 0 && q.dbg) { for (int i_ = 0; i_ < 8; ++i_) q.dbg[slot * 8 + i_] = (double)st_acc[i_]; }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_OUT
#endif

// panels factored by the Cholesky-QR fast path / by the Householder column loop (diagnostics)
__device__ unsigned long long g_cqr_stats[2];

static constexpr int QR_NT = 512;
static constexpr int QR_NW = QR_NT / WAVE;

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// copy one 2 KB tile (two DMA instructions); lane16 = 16 * lane id (bytes)
__device__ __forceinline__ void glds_tile(const double* src_tile_uniform, double* lds_tile,
                                          unsigned lane16) {
  glds16(src_tile_uniform, lane16, lds_tile);
  glds16(src_tile_uniform, lane16 + 1024u, lds_tile + 128);
}
// wait until at most 2*n of this wave's DMA instructions are outstanding (n wave-uniform)
__device__ __forceinline__ void wait_tiles_outstanding(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Structured sources (ST): TSQR merges, the Coleman-Li system [R D; E] and the
// Newton systems [R_aug; sqrt(alpha) I] are stacks of G upper-triangular blocks of
// `sr` rows (sr % 16 == 0, RP % sr == 0).  Row tile tl of block qb holds zeros in
// every column panel < tl, and the reflectors of panel j only ever touch (and fill)
// tiles with tl <= j.  QR does not depend on the row order, so the workgroup stages
// the stack INTERLEAVED: LDS tile tau = tl * G + qb.  The rows a reflector j touches
// are then the contiguous tile range [j, G (j+1)): V_j is stored, loaded and
// multiplied only there, the cyclic tile->wave map stays balanced, and tiles with
// tl >= NP (padding rows of each block) never enter at all.
//
// MAXT = tile slots per wave (ceil(ntile/8)); slot i of wave w is tile w + 8 i.
template <int MAXT, bool ST>
__global__ __launch_bounds__(QR_NT, (MAXT <= 2 ? 4 : 2)) void qr_panel_kernel(QrArgs q) {
  constexpr int NR = (MAXT * QR_NW * TILE + QR_NT - 1) / QR_NT;   // rows per thread
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id, in an SGPR
  const int leaf = blockIdx.x;
  // Masked launches over a sparse set of problems go through a compacted index list: a grid
  // whose active workgroups alternate with idle ones lands them on half of the XCDs only
  // (workgroups are dealt round-robin to the 8 XCDs).
  if (q.count_dev && (int)blockIdx.y >= *q.count_dev) return;   // (launched over an upper bound of the list)
  const int b = q.batch_list ? q.batch_list[blockIdx.y] : (int)blockIdx.y;
  if (q.require_path && q.require_path[b] == 0) return;   // uniform per workgroup
  const int N = q.ncols_dev ? q.ncols_dev[b] : q.N;       // columns to factor
  const int RP = q.RP;
  const int LDP = q.LDP;
  const int NPAD = q.NPAD;

  // ONE LDS array (a second object de-pipelines the DMA waits).  Region 0 is the
  // column-major panel [16][LDP] while a panel is factored, and the V staging slots
  // [NW][MAXT][256] while block reflectors are applied.
  const int r0sz = (16 * LDP > QR_NW * MAXT * 256) ? 16 * LDP : QR_NW * MAXT * 256;
  double* P = lds;
  double* Vs = lds + w * MAXT * 256;        // this wave's staging slots
  const unsigned lane16_ = 16u * (unsigned)lane;
  double* Wred = lds + r0sz;                // [NW][256] partial W tiles
  double* Gs = Wred + QR_NW * 256;          // [256]
  double* xch = Gs + 256;                   // [2][NW*16 + 16] per-column exchange
  double* taus = xch + 2 * (QR_NW * 16 + 16);  // [16]
  double* Tst = taus + 16;                  // [2][256] staged T_j (parity j & 1)
  double* Wred2 = Tst + 512;                // [NW][256] second panel's partial W tiles (MAXT <= 6 only)

  const long slot = (long)b * gridDim.x + leaf;
  double* Rout = q.Rout + slot * (long)NPAD * NPAD;
  if (N <= 1) return;                       // uniform per workgroup
  // The LAST logical column is always the right-hand side (f, or column N-1 of a
  // stacked source).  It is not factored: every thread keeps its rows of it in
  // registers and applies each block reflector to it right after the panel is
  // done (O(16 m) VALU work per panel instead of a 16-column MFMA panel pass).
  const int nc = N - 1;                     // columns to factor
  const int NP = (nc + TILE - 1) / TILE;
  const int nA = nc;                        // columns taken from A
  const int sr = ST ? q.stack_rows : 0;
  const int G = ST ? RP / sr : 1;           // blocks in the stack
  const unsigned Gm = ST ? 65536u / (unsigned)G + 1u : 0u;   // tau / G == (tau * Gm) >> 16  (tau * G < 65536)
  const int ntile = (ST && G * NP < RP / TILE) ? G * NP : RP / TILE;   // tiles that can ever be live
  const int RPe = ntile * TILE;
  // LDS row -> source row
  auto src_row = [&](int row) -> int {
    if (!ST) return row;
    const int tau = row >> 4;
    const int tl = (int)(((unsigned)tau * Gm) >> 16);
    return (tau - tl * G) * sr + tl * TILE + (row & 15);
  };
  const int r0 = leaf * q.rows_per_leaf;
  int nrows = q.rowsA - r0;
  if (nrows > q.rows_per_leaf) nrows = q.rows_per_leaf;
  if (nrows < 0) nrows = 0;
  const double* A = q.A + (long)b * q.strideA + (long)r0 * q.ldA;
  const double* F = q.F ? q.F + (long)b * q.strideF + r0 : nullptr;
  const int vrow0 = q.vdiag_row0;                       // single-leaf launches only
  const double vdiag = (vrow0 > 0 && !q.vdiag_vec) ? q.vdiag[b] : 0.0;
  const double* vdvec = q.vdiag_vec ? q.vdiag_vec + (long)b * q.stride_vec : nullptr;
  const double* cscale = q.colscale ? q.colscale + (long)b * q.stride_vec : nullptr;
  double* V = q.V + slot * (long)q.NPmax * RP * 16;     // tiles [t][row][col]
  double* T = q.T + slot * (long)q.NPmax * 256;

  const int lr = lane >> 4;                 // 0..3
  const int lc = lane & 15;                 // 0..15

  // right-hand side rows owned by this thread (rows tid + 512 r)
  double fr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = tid + r * QR_NT;
    const int srow = src_row(row);
    const int rlim = (vrow0 > 0 && vrow0 < nrows) ? vrow0 : nrows;   // rows backed by memory
    const int rc = (rlim > 0) ? (srow < rlim ? srow : rlim - 1) : 0;
    double val = 0.0;
    if (rlim > 0) val = F ? F[rc] : A[(long)rc * q.ldA + nc];
    fr[r] = (row < RPe && srow < rlim) ? val : 0.0;
  }

  // Panels are processed in PAIRS (k even, k+1): both live in registers while the
  // reflectors 0..k-1 are applied, so every V_j tile read from HBM (and from LDS)
  // serves two panels — half the V traffic, which is what bounds the update phase.
  // Panel k is then factored (panel k+1 stays parked in registers), V_k — still
  // explicit in LDS — is applied to panel k+1 straight from LDS, and panel k+1 is
  // factored without touching HBM for V at all.
  // (Small stacks — MAXT <= 4: the latency-bound [R D; E] / Newton systems — keep one
  // panel per pass: their V tiles are L2-resident and the pair costs two extra barriers.)
  constexpr bool PAIR = MAXT >= 4;
  // a second reduction buffer (one barrier per reflector for BOTH panels) where LDS has room
  constexpr bool DBUF = PAIR && MAXT <= 6;
  v4d pd[MAXT];                             // the pair's second panel
#pragma unroll
  for (int i = 0; i < MAXT; ++i) pd[i] = v4d{0.0, 0.0, 0.0, 0.0};

  // slots of this wave that reflector jj touches: tiles [jj, hi_jj) -> slots [lo, hi)
  auto slot_lo = [&](int jj) { const int d = jj - w; return d > 0 ? (d + QR_NW - 1) / QR_NW : 0; };
  auto slot_hi = [&](int jj) {
    const int hi = (ST && G * (jj + 1) < ntile) ? G * (jj + 1) : ntile;
    const int d = hi - w;
    const int n = d > 0 ? (d + QR_NW - 1) / QR_NW : 0;
    return n < MAXT ? n : MAXT;
  };
  // panel kk of the source -> registers, accumulator layout (single HBM read):
  // x[i][g] = element (row 16 t + lr + 4 g, column 16 kk + lc) of tile t = w + 8 i.
  // Loads are unconditional (clamped addresses, select afterwards).
  // In two halves so that both panels of a pair have their loads in flight together:
  // issue (raw values, clamped unconditional loads) ... finish (masks, scaling).
  auto issue_panel = [&](v4d* x, int kk) {
    const int rmem = (vrow0 > 0 && vrow0 < nrows) ? vrow0 : nrows;   // rows backed by memory
    const int rmax = rmem > 0 ? rmem - 1 : 0;
    const int col = kk * TILE + lc;
    const int cc = col < nA ? col : (nA > 0 ? nA - 1 : 0);
    const bool have = nrows > 0 && nA > 0;
    int lrk = lr;
    asm volatile("" : "+v"(lrk));           // per-panel recomputation instead of 32 hoisted addresses
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = src_row((w + QR_NW * i) * TILE + lrk + 4 * g);
        const int rc = row < rmax ? row : rmax;
        x[i][g] = have ? __builtin_nontemporal_load(A + (unsigned)(rc * q.ldA + cc)) : 0.0;   // read once: stream
      }
    }
  };
  auto finish_panel = [&](v4d* x, int kk) {
    const int hi = (ST && G * (kk + 1) < ntile) ? G * (kk + 1) : ntile;
    const int col = kk * TILE + lc;
    const int cc = col < nA ? col : (nA > 0 ? nA - 1 : 0);
    const double cs = (cscale && nA > 0) ? cscale[cc] : 1.0;          // fused column scaling
    const double vd = (vdvec && nA > 0) ? vdvec[cc] : vdiag;          // this column's diagonal value
    int lrk = lr;
    asm volatile("" : "+v"(lrk));
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = src_row((w + QR_NW * i) * TILE + lrk + 4 * g);
        double val = (row < nrows && col < nA) ? x[i][g] * cs : 0.0;
        if (vrow0 > 0 && row >= vrow0) val = (col == row - vrow0 && col < nA) ? vd : 0.0;
        if (w + QR_NW * i >= hi) val = 0.0;
        x[i][g] = val;
      }
    }
  };
  // cross-wave sum of the 16x16 partial products, then  -(T^T W)  (fixed order: deterministic)
  auto reduce_put = [&](const v4d& part, double* Wb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) Wb[w * 256 + g * 64 + lane] = part[g];
  };
  auto reduce_get = [&](const double* Wb, const double* Tl) -> v4d {
    const double* Wred = Wb;
    v4d W = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < QR_NW; h += 2) {      // two partials (8 values) in flight at a time
      double wp[2][4];
#pragma unroll
      for (int ww = 0; ww < 2; ++ww) {
#pragma unroll
        for (int g = 0; g < 4; ++g) wp[ww][g] = Wred[(h + ww) * 256 + g * 64 + lane];
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int ww = 0; ww < 2; ++ww) {        // fixed order: deterministic
#pragma unroll
        for (int g = 0; g < 4; ++g) W[g] += wp[ww][g];
      }
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    }
    v4d W2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) W2 = mfma_f64(Tl[64 * s + lane], W[s], W2);
#pragma unroll
    for (int g = 0; g < 4; ++g) W2[g] = -W2[g];
    return W2;
  };
  auto reduce_w = [&](const v4d& part, const double* Tl) -> v4d {
    reduce_put(part, Wred);
    lds_barrier();
    return reduce_get(Wred, Tl);
  };

  // ---- Cholesky-QR + Householder reconstruction of one panel --------------------------------
  // The column-by-column Householder loop below is a latency chain (16 column steps, each a
  // wave butterfly + an LDS exchange + a barrier + a sqrt/div chain) during which the MFMA pipe
  // idles.  When the panel A (live rows x 16) is well conditioned the same factorisation comes
  // from ONE Gram matrix:  G = A^T A (MFMA, straight from the panel registers),  column-scaled
  // Cholesky G = R^T R (16x16),  and the Householder representation of Q = A R^-1 is
  // RECONSTRUCTED (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik 2014): LU without
  // pivoting of  Q - [S; 0]  with  S_jj = -sgn(pivot)  gives  Y_1 (unit lower), U,
  // Y_2 = Q_2 U^-1,  T = -U S Y_1^-T,  R_hh = S R,  so that (I - Y T Y^T)^T A = [R_hh; 0].
  // Multiplying through by R, the same L comes from the LU of  A_top - S R  (pivot j is the Q-form
  // pivot times R_jj > 0, so the signs agree), whose upper factor is U' = U R; then
  //     Y_2 = A_2 U'^-1,        T = -U' W^-1  with  W = Y_1^T S R  (upper triangular),
  // and neither Q nor R^-1 is ever formed.  Only Y_2 = A_2 M needs the tall data again (one MFMA
  // pass by all waves).  The 16x16 work runs on TWO waves that repeat the (deterministic) Cholesky
  // and LU and then split: wave 0 inverts U' (-> M), wave 1 forms W, inverts it and multiplies
  // (-> T); lane j (mod 16) owns column j in registers, broadcasts are v_readlane, the 16x16x16
  // products go to the MFMA pipe through small LDS scratch tiles.  The Cholesky pivots of the
  // unit-diagonal Gram bound the conditioning: if the smallest one is below CQR_PMIN = 0.1 (or a column
  // is zero / not finite) nothing has been modified and the exact Householder column loop runs
  // instead.  Returns true on success with P = [R_hh upper | Y_1 strictly lower] in the pivot tile
  // and Y_2 below it, Gs = T_k (also in the global T scratch) — exactly what the column loop and
  // the T recurrence leave behind.
  constexpr double CQR_PMIN = 0.1;
  auto panel_cqr = [&](v4d* pa, int k, int hik) -> bool {
    const int base = k * TILE;
    int lcq = lc, lrq = lr, laneq = lane;
    asm volatile("" : "+v"(lcq), "+v"(lrq), "+v"(laneq));   // per-call address arithmetic (no long-lived registers)
    const int j = lcq;                                   // column owned by this lane (waves 0, 1)
    double* Mbuf = Tst;                                  // M = U'^-1 for every wave (wave 0's scratch)
    double* flag = xch;                                  // [0]: 1.0 success
    // (a) Gram of the live rows, straight from the registers
    v4d gacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = w + QR_NW * i;
      if (t >= k && t < hik) {
#pragma unroll
        for (int s = 0; s < 4; ++s) gacc = mfma_f64(pa[i][s], pa[i][s], gacc);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) Wred[w * 256 + g * 64 + laneq] = gacc[g];
    double ac[TILE];                                     // column j of A_top (pivot tile), read
    if (w < 2) {                                         // BEFORE the barrier: the tile is rewritten later
#pragma unroll
      for (int i = 0; i < TILE; ++i) ac[i] = P[j * LDP + base + i];
    }
    lds_barrier();
    if (w < 2) {
      double* SA = Tst + w * 256;                        // per-wave 16x16 scratch tiles (row-major)
      double* SB = xch + 16;                             // wave 1 only
      v4d G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ww = 0; ww < QR_NW; ++ww) {
#pragma unroll
        for (int g = 0; g < 4; ++g) G[g] += Wred[ww * 256 + g * 64 + laneq];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) SA[(lrq + 4 * g) * 16 + lcq] = G[g];
      double col[TILE];
#pragma unroll
      for (int i = 0; i < TILE; ++i) col[i] = SA[i * 16 + j];
      const double gjj = SA[j * 17];
      const bool bad = !(gjj > 0.0) || !is_finite(gjj);
      double dj = __builtin_amdgcn_rsq(bad ? 1.0 : gjj);
      dj = dj * fma(-0.5 * gjj * dj, dj, 1.5);
      dj = dj * fma(-0.5 * gjj * dj, dj, 1.5);
#pragma unroll
      for (int i = 0; i < TILE; ++i) col[i] *= read_lane(dj, i) * dj;     // unit-diagonal Gram
      // Cholesky (right-looking): afterwards col[i] = R'[i][j] for i <= j, 0 below
      double pmin = 1.0;
#pragma unroll
      for (int kk = 0; kk < TILE; ++kk) {
        const double d = read_lane(col[kk], kk);
        if (!(d >= pmin)) pmin = d;                      // also catches NaN
        const double ds = (d > 1e-300) ? d : 1.0;
        double ri = __builtin_amdgcn_rsq(ds);
        ri = ri * fma(-0.5 * ds * ri, ri, 1.5);
        ri = ri * fma(-0.5 * ds * ri, ri, 1.5);
        const double rkj = col[kk] * ri;
        col[kk] = (j >= kk) ? rkj : 0.0;
#pragma unroll
        for (int i = kk + 1; i < TILE; ++i) col[i] = fma(-read_lane(rkj, i), rkj, col[i]);
      }
      const bool ok = !__any(bad) && (pmin >= CQR_PMIN);
      if (ok) {                                          // wave-uniform, identical in both waves
        const double inv_dj = gjj * dj;                  // sqrt(g_jj) = 1 / d_j
        // R = R' D^-1 goes to this wave's LDS scratch (row-major) and leaves the registers:
        // the LU below needs one row of it per step, the products need it as an operand
        double* SR = (w == 0) ? SA : SB;
#pragma unroll
        for (int i = 0; i < TILE; ++i) SR[i * 16 + j] = col[i] * inv_dj;
        // LU of A_top - S R without pivoting, S_k = -sgn(pivot): ac -> [U' upper | Y_1 lower]
        double sgv = 1.0, pnv = 0.0;                     // lane k: S_k and 1 / U'[k][k]
#pragma unroll
        for (int kk = 0; kk < TILE; ++kk) {
          double piv = read_lane(ac[kk], kk);
          const double sk = (piv >= 0.0) ? -1.0 : 1.0;
          const double rk = SR[kk * 16 + j];             // R[k][j]
          ac[kk] = fma(-sk, rk, ac[kk]);                 // row k -= S_k R[k][:]
          piv = fma(-sk, read_lane(rk, kk), piv);        // |piv| >= R[k][k] > 0
          double pi = __builtin_amdgcn_rcp(piv);
          pi = pi * fma(-piv, pi, 2.0);
          pi = pi * fma(-piv, pi, 2.0);
          sgv = (j == kk) ? sk : sgv;
          pnv = (j == kk) ? pi : pnv;
          // columns right of kk take the update; column kk keeps its raw entries (scaled by its
          // own 1 / pivot after the loop), columns left of it already hold Y_1
          const double ukp = (j > kk) ? ac[kk] * pi : 0.0;
#pragma unroll
          for (int i = kk + 1; i < TILE; ++i) ac[i] = fma(-read_lane(ac[i], kk), ukp, ac[i]);
        }
#pragma unroll
        for (int i = 1; i < TILE; ++i) ac[i] = (i > j) ? ac[i] * pnv : ac[i];
        if (w == 0) {
          // pivot tile of the panel: R_hh = S R on and above the diagonal, Y_1 below it
          // (first: M below reuses the scratch that holds R)
#pragma unroll
          for (int i = 0; i < TILE; ++i) {
            const double si = read_lane(sgv, i);
            const double rij = SR[i * 16 + j];
            if (laneq < TILE) P[j * LDP + base + i] = (i <= j) ? si * rij : ac[i];
          }
          // M = U'^-1, column j (back substitution)
          double u[TILE];
#pragma unroll
          for (int i = TILE - 1; i >= 0; --i) {
            double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int l = i + 1; l < TILE; ++l) acc = fma(-read_lane(ac[i], l), u[l], acc);
            u[i] = acc * read_lane(pnv, i);
          }
#pragma unroll
          for (int i = 0; i < TILE; ++i) Mbuf[i * 16 + j] = u[i];
        } else {
          // W = Y_1^T S R (MFMA): A operand E[m][k] = Y_1[k][m] S_k, B operand R
#pragma unroll
          for (int i = 0; i < TILE; ++i) {
            const double si = read_lane(sgv, i);
            SA[j * 16 + i] = (i > j) ? ac[i] * si : ((i == j) ? si : 0.0);
          }                                              // (SB already holds R)
          v4d wacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s)
            wacc = mfma_f64(SA[lcq * 16 + 4 * s + lrq], SB[(4 * s + lrq) * 16 + lcq], wacc);
#pragma unroll
          for (int g = 0; g < 4; ++g) SA[(lrq + 4 * g) * 16 + lcq] = wacc[g];
          double wc[TILE];
#pragma unroll
          for (int i = 0; i < TILE; ++i) wc[i] = SA[i * 16 + j];
          const double wjj = SA[j * 17];
          double wdi = __builtin_amdgcn_rcp(wjj);
          wdi = wdi * fma(-wjj, wdi, 2.0);
          wdi = wdi * fma(-wjj, wdi, 2.0);
          // W^-1, column j
          double z[TILE];
#pragma unroll
          for (int i = TILE - 1; i >= 0; --i) {
            double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int l = i + 1; l < TILE; ++l) acc = fma(-read_lane(wc[i], l), z[l], acc);
            z[i] = acc * read_lane(wdi, i);
          }
          // T = -U' W^-1 (MFMA): A operand U' (upper part of ac), B operand W^-1
#pragma unroll
          for (int i = 0; i < TILE; ++i) {
            SB[i * 16 + j] = (i <= j) ? ac[i] : 0.0;
            SA[i * 16 + j] = z[i];
          }
          v4d tacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s)
            tacc = mfma_f64(SB[lcq * 16 + 4 * s + lrq], SA[(4 * s + lrq) * 16 + lcq], tacc);
          double* Tk = T + k * 256;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int e = (lrq + 4 * g) * 16 + lcq;
            Gs[e] = -tacc[g];
            Tk[e] = -tacc[g];
          }
        }
      }
      if (w == 0 && laneq == 0) {
        flag[0] = ok ? 1.0 : 0.0;
        atomicAdd(&g_cqr_stats[ok ? 0 : 1], 1ULL);
      }
    }
    lds_barrier();
    const bool ok = flag[0] != 0.0;
    if (ok) {
      // Y_2 = A_2 M for the tiles below the pivot tile (A operand read back from the LDS panel)
      double mb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) mb[s] = Mbuf[(4 * s + lrq) * 16 + lcq];
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        const int t = w + QR_NW * i;
        if (t > k && t < hik) {
          const double* pv = P + lrq * LDP + t * TILE + lcq;
          v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s) y = mfma_f64(pv[4 * s * LDP], mb[s], y);
          double* pt = P + lcq * LDP + t * TILE + lrq;
#pragma unroll
          for (int g = 0; g < 4; ++g) pt[4 * g] = y[g];
          if (k < NP - 1) {                  // spill the tile image for later panels right away
            double* vt = V + ((long)k * RP + (long)t * TILE) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int r = lrq + 4 * g;
              vt[16 * r + (lcq ^ r)] = y[g];
            }
          }
        }
      }
    }
    lds_barrier();
    return ok;
  };

  STAMP_DECL
  for (int k = 0; k < NP; ++k) {
    // live tiles of this panel: [k, hik); rows past them are never read or written
    const int hik = (ST && G * (k + 1) < ntile) ? G * (k + 1) : ntile;
    const int rows_k = hik * TILE;
    const bool second = PAIR && (k & 1);      // panel k arrives in pd (reflectors 0..k-2 applied)
    const bool lead = PAIR && !second && (k + 1 < NP);   // panel k+1 rides along in pd
    v4d pc[MAXT];
    if (!second) {
      unsigned lane16 = lane16_;
      asm volatile("" : "+v"(lane16));        // keeps per-slot DMA addresses out of long-lived registers
      // ---- 0. start the DMA of V_0 / T_0 (overlaps the panel loads) ---------
      if (k > 0) {
        if (w == 0) glds_tile(T, Tst, lane16);
        const int lo = slot_lo(0), hi = slot_hi(0);
#pragma unroll
        for (int i = 0; i < MAXT; ++i)
          if (i >= lo && i < hi) glds_tile(V + (w + QR_NW * i) * 256, Vs + i * 256, lane16);
      }
      // ---- 1. panels k (and k+1) of the source -> registers ------------------
      issue_panel(pc, k);
      if (lead) issue_panel(pd, k + 1);
      finish_panel(pc, k);
      if (lead) finish_panel(pd, k + 1);
      STAMP(0)

      // ---- 2. apply block reflectors 0..k-1:  P -= V_j (T_j^T (V_j^T P)) ----
      // HBM traffic bounds this phase (the V panels of a leaf do not fit L2/MALL): every
      // V_j tile is read ONCE per panel pair, by DMA into its wave's LDS slot, prefetched
      // one reflector ahead.  Tile image (2 KB): element (r, c) at 16 r + (c ^ r).  First
      // GEMM operand A = V^T: lane reads (r = 4 s + lr, c = lc) — a permuted full row per
      // 16 lanes; third GEMM operand A = V: lane reads (r = lc, c = 4 s + lr) — 2 lanes
      // per bank pair.  Both conflict-free.
      for (int j = 0; j < k; ++j) {
        const int lo = slot_lo(j), hi = slot_hi(j);
        const int lon = slot_lo(j + 1), hin = (j + 1 < k) ? slot_hi(j + 1) : 0;
        const double* Tj = Tst + (j & 1) * 256;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        v4d acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          if (i >= lo && i < hi) {
            wait_tiles_outstanding(hi - 1 - i);   // slot i has landed (later slots may still fly)
            const double* vt = Vs + i * 256;
            double a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int r = 4 * s + lr;
              a[s] = vt[16 * r + (lc ^ r)];
            }
            __builtin_amdgcn_s_setprio(1);                 // MFMA bursts win the issue arbitration
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc = mfma_f64(a[s], pc[i][s], acc);
              if (PAIR) acc2 = mfma_f64(a[s], pd[i][s], acc2);   // (all-zero pd when there is no second panel)
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 LDS reads, then the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, PAIR ? 8 : 4, 0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (wave 0: T_j has landed too)
        v4d W2, W2b;
        if (DBUF) {                             // both panels' partials, ONE barrier
          reduce_put(acc, Wred);
          reduce_put(acc2, Wred2);
          lds_barrier();
          W2 = reduce_get(Wred, Tj);
          W2b = reduce_get(Wred2, Tj);
        } else {
          W2 = reduce_w(acc, Tj);
          W2b = W2;
          if (PAIR) {
            lds_barrier();                      // Wred is rewritten for the second panel
            W2b = reduce_w(acc2, Tj);
          }
        }
        asm volatile("" ::: "memory");
        if (w == 0 && j + 1 < k) glds_tile(T + (j + 1) * 256, Tst + ((j + 1) & 1) * 256, lane16);
        const double* Vn = V + (long)(j + 1) * RP * 16;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          if (i >= lo && i < hi) {
            const double* vt = Vs + i * 256;
            double a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int c = 4 * s + lr;
              a[s] = vt[16 * lc + (c ^ lc)];
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              pc[i] = mfma_f64(a[s], W2[s], pc[i]);
              if (PAIR) pd[i] = mfma_f64(a[s], W2b[s], pd[i]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, PAIR ? 8 : 4, 0);
          }
          if (i >= lon && i < hin) {            // refill the slot with its tile of V_{j+1}
            asm volatile("" ::: "memory");
            glds_tile(Vn + (w + QR_NW * i) * 256, Vs + i * 256, lane16);
          }
        }
        lds_barrier();                          // Wred is reused by the next j
      }
      lds_barrier();                            // every wave is done with the staging slots
    } else {
      // ---- 1'+2'. panel k arrives in pd; V_{k-1} is still explicit in LDS (panel
      // layout) and T_{k-1} in Gs: apply it from LDS.
      const int j = k - 1;
      const int lo = slot_lo(j), hi = slot_hi(j);
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        if (i >= lo && i < hi) {
          const double* pt = P + lc * LDP + (w + QR_NW * i) * TILE + lr;   // V^T operand
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma_f64(pt[4 * s], pd[i][s], acc);
        }
      }
      const v4d W2 = reduce_w(acc, Gs);
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        if (i >= lo && i < hi) {
          const double* pv = P + lr * LDP + (w + QR_NW * i) * TILE + lc;   // V[row 16 t + lc][col 4 s + lr]
#pragma unroll
          for (int s = 0; s < 4; ++s) pd[i] = mfma_f64(pv[4 * s * LDP], W2[s], pd[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < MAXT; ++i) pc[i] = pd[i];
      lds_barrier();                            // everyone is done reading V_{k-1} from the panel
      STAMP(0)
    }
    // ---- 2b. hand the updated panel to LDS (column-major) for the factorisation
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = w + QR_NW * i;
      if (t < hik) {
        double* pt = P + lc * LDP + t * TILE + lr;
#pragma unroll
        for (int g = 0; g < 4; ++g) pt[4 * g] = pc[i][g];
      }
    }
    if (PAIR && !lead) {
#pragma unroll
      for (int i = 0; i < MAXT; ++i) pd[i] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    __syncthreads();

    STAMP(1)
    // ---- 3. Householder-factor rows >= 16k of the panel --------------------
    // Each thread owns rows tid + 512 i, i < NR (RP <= 512 NR) and keeps ITS rows of
    // all 16 panel columns in registers for the whole factorisation (one LDS read
    // before, one LDS write after; the column loop is fully unrolled so the
    // register file is indexed statically).  Per column: one fused reduction (sum
    // x^2 and the <=15 dot products x.P[c']) through a transposed DPP butterfly +
    // one LDS exchange + ONE barrier; totals are summed lane-parallel and broadcast
    // with v_readlane.
    const int base = k * TILE;
    int tidk = tid;
    if (PAIR) asm volatile("" : "+v"(tidk));  // LDS addresses are recomputed per panel, not kept live
    int rowi[NR], rci[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      rowi[r] = tidk + r * QR_NT;
      rci[r] = rowi[r] < rows_k ? rowi[r] : rows_k - 1;
    }
    // ---- 3C. fast path: Cholesky-QR + Householder reconstruction ----------
    bool cqr_ok = false;
    if (q.cqr && base + TILE <= nc) cqr_ok = panel_cqr(pc, k, hik);
    if (!cqr_ok) {
    double pr[NR][TILE];
#pragma unroll
    for (int c = 0; c < TILE; ++c) {
#pragma unroll
      for (int r = 0; r < NR; ++r) pr[r][c] = P[c * LDP + rci[r]];
    }
#pragma unroll
    for (int c = 0; c < TILE; ++c) {
      const int p = base + c;               // pivot row == global column
      if (!(p < nc && p < RPe)) {           // padding column: H = I (uniform)
        if (tid == 0) taus[c] = 0.0;
      } else {
        bool inr[NR], own[NR];
        double x[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          inr[r] = rowi[r] > p && rowi[r] < rows_k;
          own[r] = rowi[r] == p;
          x[r] = inr[r] ? pr[r][c] : 0.0;
        }
        double* ex = xch + (c & 1) * (QR_NW * 16 + 16);
        {
          double part[TILE];
#pragma unroll
          for (int i = 0; i < TILE; ++i) {
            double acc = 0.0;
            if (c + i < TILE) {
#pragma unroll
              for (int r = 0; r < NR; ++r)
                acc = fma(x[r], (i == 0) ? x[r] : pr[r][(c + i < TILE) ? c + i : 0], acc);
            }
            part[i] = acc;
          }
          wave_sum16(part);                  // all 16 totals in one transposed butterfly
          if (lane < TILE) ex[w * 16 + wave_sum16_index(lane)] = part[0];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (own[r]) {                      // the pivot row's owner publishes it
#pragma unroll
            for (int i = 0; i < TILE; ++i)
              ex[QR_NW * 16 + i] = pr[r][(c + i < TILE) ? c + i : TILE - 1];
          }
        }
        STAMP(6)
        __syncthreads();
        STAMP(7)
        double totl = 0.0;                   // lane i (mod 16): total of value i
        double exv[QR_NW];
#pragma unroll
        for (int ww = 0; ww < QR_NW; ++ww) exv[ww] = ex[ww * 16 + lc];
        const double pvl = ex[QR_NW * 16 + lc];
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);   // all exchange reads in flight together
#pragma unroll
        for (int ww = 0; ww < QR_NW; ++ww) totl += exv[ww];
        const double xn2 = read_lane(totl, 0);
        const double alpha = read_lane(pvl, 0);
        double beta, tau, scal;
        if (xn2 == 0.0) {
          beta = alpha; tau = 0.0; scal = 0.0;
        } else {
          // The column step is a latency chain every wave runs: one v_rsq + one v_rcp (two Newton
          // steps each, full precision) instead of a sqrt and two divisions:
          //   norm = s rsqrt(s),  tau = (beta - alpha)/beta = 1 + |alpha|/norm,
          //   1/(alpha - beta) = sign(alpha)/(|alpha| + norm).
          const double ssq = fma(alpha, alpha, xn2);
          double y = __builtin_amdgcn_rsq(ssq);
          y = y * fma(-0.5 * ssq * y, y, 1.5);
          y = y * fma(-0.5 * ssq * y, y, 1.5);
          const double nrm = ssq * y;
          const double aa = fabs(alpha);
          beta = -copysign(nrm, alpha);
          tau = fma(aa, y, 1.0);
          const double den = aa + nrm;
          double rc = __builtin_amdgcn_rcp(den);
          rc = rc * fma(-den, rc, 2.0);
          rc = rc * fma(-den, rc, 2.0);
          scal = copysign(rc, alpha);
        }
        const double wvl = tau * (pvl + scal * totl);      // lane i holds w_i
        double vv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          vv[r] = scal * x[r];
          if (inr[r]) pr[r][c] = vv[r];
          if (own[r]) pr[r][c] = beta;
        }
#pragma unroll
        for (int i = 1; i < TILE; ++i) {
          if (c + i < TILE) {
            const double wv = read_lane(wvl, i);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              if (inr[r]) pr[r][c + i] = fma(-vv[r], wv, pr[r][c + i]);
              if (own[r]) pr[r][c + i] -= wv;
            }
          }
        }
        if (tid == 0) taus[c] = tau;
      }
    }
#pragma unroll
    for (int c = 0; c < TILE; ++c) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (rowi[r] < rows_k) P[c * LDP + rowi[r]] = pr[r][c];
    }
    __syncthreads();
    }   // !cqr_ok

    STAMP(2)
    // ---- 4. emit the R block column (rows 0..NPAD-1 of these 16 columns) --
    for (int idx = tid; idx < NPAD * 16; idx += QR_NT) {
      const int row = idx >> 4, c = idx & 15;
      const int col = base + c;
      if (col < NPAD) {
        double val = 0.0;
        if (row <= col && row < RPe) val = P[c * LDP + row];
        Rout[(long)row * NPAD + col] = val;
      }
    }
    STAMP(3)
    __syncthreads();

    // ---- 5. make V_k explicit (unit lower trapezoid, zeros above) ---------
    for (int idx = tid; idx < (base + TILE) * 16; idx += QR_NT) {
      const int row = idx >> 4, c = idx & 15;
      if (row < RPe) {
        if (row < base + c) P[c * LDP + row] = 0.0;
        else if (row == base + c) P[c * LDP + row] = 1.0;
      }
    }
    __syncthreads();

    // ---- 6. T_k from G = V_k^T V_k (MFMA) + 16-step row recurrence --------
    if (!cqr_ok) {
      int t0 = k + ((w - k) % QR_NW + QR_NW) % QR_NW;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int t = t0; t < hik; t += QR_NW) {
        const double* pt = P + lc * LDP + t * TILE + lr;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double e = pt[4 * s];
          acc = mfma_f64(e, e, acc);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) Wred[w * 256 + g * 64 + lane] = acc[g];
      __syncthreads();
      if (w == 0) {
        v4d G = {0.0, 0.0, 0.0, 0.0};
        for (int ww = 0; ww < QR_NW; ++ww) {
#pragma unroll
          for (int g = 0; g < 4; ++g) G[g] += Wred[ww * 256 + g * 64 + lane];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) Gs[(lr + 4 * g) * 16 + lc] = G[g];
      }
      __syncthreads();
      if (tid < TILE) {
        // thread i owns row i of T:  T[i][c] = -tau_c * sum_{q=i}^{c-1} T[i][q] G[q][c].
        // Fully unrolled: the row lives in registers and the (static) G / tau reads are
        // independent of the recurrence, so they pipeline instead of serialising.
        const int i = tid;
        double Trow[TILE];
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
          const double tc = taus[c];
          double gc[TILE];                     // column c of G: read up front, off the dependency chain
#pragma unroll
          for (int qq = 0; qq < TILE; ++qq) gc[qq] = (qq < c) ? Gs[qq * 16 + c] : 0.0;
          double sa[4] = {0.0, 0.0, 0.0, 0.0}; // four interleaved partial sums: a 4x shorter chain
#pragma unroll
          for (int qq = 0; qq < TILE; ++qq)
            if (qq < c) sa[qq & 3] = fma(Trow[qq], gc[qq], sa[qq & 3]);
          const double sacc = (sa[0] + sa[1]) + (sa[2] + sa[3]);
          Trow[c] = (i == c) ? tc : ((i < c) ? -tc * sacc : 0.0);
        }
        double* Ts = Gs + i * TILE;           // G is consumed (one wave, program order): T_k replaces it
        double* Tk = T + k * 256;
#pragma unroll
        for (int c = 0; c < TILE; ++c) { Ts[c] = Trow[c]; Tk[i * 16 + c] = Trow[c]; }
      }
    }
    STAMP(4)
    // ---- 7. spill V_k for later panels: 2 KB tile images, element (r, c) at 16 r + (c ^ r)
    // (tiles above the pivot tile are never read back; after the Cholesky-QR path the tiles
    // below it have been written by that path already)
    if (k < NP - 1) {
      double* Vk = V + (long)k * RP * 16;
      const int hi7 = cqr_ok ? (base + TILE) * 16 : rows_k * 16;
      for (int idx = base * 16 + tid; idx < hi7; idx += QR_NT) {
        const int row = idx >> 4, c = (idx & 15) ^ (row & 15);   // image position idx holds column c
        Vk[idx] = P[c * LDP + row];
      }
    }
    __threadfence_block();
    __syncthreads();
    STAMP(5)

    // ---- 8. right-hand side:  f -= V_k (T_k^T (V_k^T f))  ------------------
    {
      double vk[NR][TILE];
#pragma unroll
      for (int c = 0; c < TILE; ++c) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const double val = P[c * LDP + rci[r]];
          vk[r][c] = (rowi[r] < rows_k) ? val : 0.0;  // rows past the live range must stay inert
        }
      }
      double part[TILE];
#pragma unroll
      for (int c = 0; c < TILE; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) acc = fma(vk[r][c], fr[r], acc);
        part[c] = acc;
      }
      wave_sum16(part);
      double* ex = xch;
      if (lane < TILE) ex[w * 16 + wave_sum16_index(lane)] = part[0];
      __syncthreads();
      double wl = 0.0;                       // lane i (mod 16): (V_k^T f)_i
#pragma unroll
      for (int ww = 0; ww < QR_NW; ++ww) wl += ex[ww * 16 + lc];
      const double* Ts = Gs;                 // T_k rows, left there by step 6
      double zl = 0.0;                       // lane c: (T_k^T w)_c = sum_i T[i][c] w_i
#pragma unroll
      for (int i = 0; i < TILE; ++i) zl = fma(Ts[i * TILE + lc], read_lane(wl, i), zl);
#pragma unroll
      for (int c = 0; c < TILE; ++c) {
        const double zc = read_lane(zl, c);
#pragma unroll
        for (int r = 0; r < NR; ++r) fr[r] = fma(-vk[r][c], zc, fr[r]);
      }
      __syncthreads();                       // ex / Wred / P are rewritten by the next panel
    }
    STAMP(5)
  }
  // rhs column of the triangle: c = (Q^T f)[0:nc]; rows >= nc are not needed downstream
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = tid + r * QR_NT;
    if (row < NPAD) Rout[(long)row * NPAD + nc] = (row < nc && row < RPe) ? fr[r] : 0.0;
  }
  STAMP_OUT
}

size_t qr_lds_bytes(int LDP, int maxt) {
  const size_t r0 = std::max((size_t)16 * LDP, (size_t)QR_NW * maxt * 256);
  const size_t second = (maxt >= 4 && maxt <= 6) ? (size_t)QR_NW * 256 : 0;   // Wred2
  return sizeof(double) * (r0 + QR_NW * 256 + 256 + 2 * (QR_NW * 16 + 16) + 16 + 512 + second);
}

template <int MAXT, bool ST>
static hipError_t launch_qr_t(const QrArgs& q, int nleaf, int B, hipStream_t st) {
  const size_t lds = qr_lds_bytes(q.LDP, MAXT);
  // the attribute is per device; one ctx per host thread: launches may race (a repeated set is harmless)
  static std::atomic<size_t> configured_dev[64];
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  std::atomic<size_t>& configured = configured_dev[dev_ & 63];
  if (lds > configured.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)qr_panel_kernel<MAXT, ST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured.store(lds, std::memory_order_release);
  }
  hipLaunchKernelGGL((qr_panel_kernel<MAXT, ST>), dim3(nleaf, B), dim3(QR_NT), lds, st, q);
  return hipGetLastError();
}

template <bool ST>
static hipError_t launch_qr_s(const QrArgs& q, int nleaf, int B, int slots, hipStream_t st) {
  if (slots <= 2) return launch_qr_t<2, ST>(q, nleaf, B, st);
  if (slots <= 4) return launch_qr_t<4, ST>(q, nleaf, B, st);
  if (slots <= 6) return launch_qr_t<6, ST>(q, nleaf, B, st);
  return launch_qr_t<8, ST>(q, nleaf, B, st);
}

hipError_t qr_cqr_stats(unsigned long long out[2], int reset, hipStream_t st) {
  hipError_t e = hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(g_cqr_stats), 2 * sizeof(unsigned long long),
                                          0, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess || !reset) return e;
  const unsigned long long z[2] = {0ULL, 0ULL};
  e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_cqr_stats), z, sizeof(z), 0, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

static double* g_qr_dbg = nullptr;
void set_qr_debug_buffer(double* p) { g_qr_dbg = p; }

int qr_staged_tiles(int RP, int stack_rows, int N) {
  int ntile = RP / TILE;
  if (qr_stack_ok(RP, stack_rows)) {
    const int G = RP / stack_rows;
    const int NP = (N - 1 + TILE - 1) / TILE;
    if (G * NP < ntile) ntile = G * NP;
  }
  return ntile;
}

hipError_t launch_qr(const QrArgs& q_in, int nleaf, int B, hipStream_t st) {
  QrArgs q = q_in;
  q.dbg = g_qr_dbg;
  {
    q.cqr = options_or_default(q.opt).on(OPT_QR_CQR) ? 1 : 0;
  }
  if (g_qr_dbg) g_qr_dbg += (size_t)nleaf * B * 8;   // successive launches append
  // a stack the interleaved staging cannot express is factored as a dense source
  const bool st_ok = qr_stack_ok(q.RP, q.stack_rows);
  if (!st_ok) q.stack_rows = 0;
  const int ntile = qr_staged_tiles(q.RP, q.stack_rows, q.N);   // q.N bounds every per-problem column count
  if (ntile > QR_MAX_TILES) return hipErrorInvalidValue;
  // LDS column stride of the staged panel: smallest value >= rows with LDP == 2 (mod 32)
  q.LDP = (ntile * TILE + 29) / 32 * 32 + 2;
  const int slots = (ntile + QR_NW - 1) / QR_NW;
  return st_ok ? launch_qr_s<true>(q, nleaf, B, slots, st) : launch_qr_s<false>(q, nleaf, B, slots, st);
}

}  // namespace blsq
