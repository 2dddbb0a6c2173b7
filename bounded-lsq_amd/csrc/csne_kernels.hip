// CSNE tier: corrected semi-normal equations for the problems the conditioning certificate keeps off the
// normal-equations path (kappa_2 of the equilibrated system above the gate), in MEMORY-bound passes over J
// instead of the MFMA-bound second factorisations of the CholeskyQR2 tier / the Householder tree.
//
// What the reference computes for such a problem (trf.py:264-308, trust_region.py:111-150) is a chain of
// solves with M(alpha) = J_h^T J_h + diag_h + alpha I:  p(alpha) = -M^-1 g_h,  phi = ||p|| - Delta,
// phi' = -p^T M^-1 p / ||p||,  a safeguarded Newton iteration on alpha, the final p.  The Gram-Cholesky factor
// the problem already has (X^T X = H~ = H + E, |E| ~ eps ||H||) solves a NEARBY problem: every cheap solve is
// off by T = M~^-1 E, ||T|| ~ c eps kappa_2 =: rho << 1 up to kappa_2 ~ 1e10.  Instead of a better factor
// (two more m n^2 passes) the tier corrects the solves (Bjorck 1987: corrected semi-normal equations):
//
//   1. the cheap iteration runs as for any normal-equations-path problem and RECORDS every evaluation k:
//      alpha~_k, p~_k = -M~_k^-1 g_h, w~_k = M~_k^-1 p~_k, z~_k = M~_k^-1 w~_k          (lm_body.h)
//   2. ONE streaming pass over J (csne_pass_kernel) forms, for all recorded evaluations at once,
//      y_k = J_h^T (J_h p~_k + f)  and  b_k = ||J_h w~_k||^2 — the residuals of the recorded solves against J
//      ITSELF, f inside the product so that a small residual J p + f is not lost against g = J^T f
//   3. csne_fix_kernel (n-space): res_k = y_k + (diag_h + alpha~_k) p~_k, and to first order in rho
//         ||p_k||^2      = ||p~_k||^2 - 2 w~_k . res_k
//         p_k^T M^-1 p_k = 2 p~_k . w~_k - (b_k + w~_k (diag_h + alpha) w~_k) - 2 z~_k . res_k
//      (errors O(rho eta), eta the measured size of these corrections), phi'' from the cheap vectors; the scalar
//      Newton iteration of the reference is REPLAYED on the corrected phi, phi' (Taylor models around alpha~_k:
//      the replayed alpha_k differ from them by O(rho)), same brackets, restarts and stop rule, so alpha and n_iter
//      come out as the reference's; the final step is p~_K - M~_K^-1 res_K - w~_K eps + z~_K eps^2
//      (eps = alpha_K - alpha~_K), one corrected solve with the factor of the last evaluation
//   4. the step kernel takes every model product with p from the normal equations the corrected p satisfies,
//      H p = -(c g_h + alpha p)  (c: the final rescaling), instead of from the factor (trf_kernels.hip).
//
// Acceptance: PROVEN — the factor ran to completion with squared pivots >= CSNE_PIVOT_FLOOR and the certificate's
// bound on kappa_2 of the COMPUTED system is <= CSNE_K2_MAX (M~ is positive definite, lambda_min >= 1 / K2), the
// rank gate of the reference holds by that bound; MEASURED on the problem itself — every first-order correction
// applied (eta) must be <= CSNE_ETA_MAX: the neglected second-order terms are rho eta with rho ~ 1e1..1e2 eta.
// The replay must end at the evaluation the cheap iteration ended at.  Anything else leaves the tier (CholeskyQR2 /
// Householder tree, as before) — per problem, by the problem's own numbers.
#include "gram_common.h"
#include "tri_ops.h"
#include "lm_body.h"

namespace blsq {

// row chunks of the pass: a function of m ALONE (a problem's partial sums, and so every bit of its result, do not
// depend on the batch it shares a launch with)
void csne_geometry(int m, int* rows_per_wg, int* nchunk) {
  int rows = 512;
  while ((m + rows - 1) / rows > 64) rows *= 2;
  *rows_per_wg = rows;
  *nchunk = (m + rows - 1) / rows;
}
bool csne_supported(int m, int n) { return n + 1 > 80 && n <= 256 && m >= n; }

static constexpr int CS_NT = 256;
static constexpr int CS_NW = CS_NT / WAVE;
#ifndef BLSQ_CS_RB
#define BLSQ_CS_RB 4
#endif
static constexpr int CS_RB = BLSQ_CS_RB; // rows per wave and batch

__host__ __device__ inline long csne_part_stride(int NE, int ld) { return (long)NE * ld + 16; }
__device__ __forceinline__ constexpr int cs_bitrev4(int i) {
  return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
}

// ---- the pass over J -------------------------------------------------------------------------------
// grid (row chunks, listed problems) x 256 threads.  Lane l owns columns l + 64 cb.  The rows of a chunk go in batches
// of four; batch bi belongs to row class bi mod 4, and a problem's sums are DEFINED as
//     ((S_0 + S_1) + S_2) + S_3,   S_c = the batches of class c in increasing order, rows 0 .. 3 of a batch in order
// — whatever the launch looks like.  The launch splits the NE recorded evaluations over G groups of waves (G = 1, 2, 4;
// NEH evaluations each): the four waves are G groups of 4 / G members, a member takes the classes c = j mod (4 / G) —
// G of them, each in its own accumulators — for its group's evaluations.  So the deeper the batch's recordings the
// fewer evaluations a wave carries (vectors, accumulators and the butterflies of its totals scale with NEH, not NE: the
// kernel is VALU-issue-bound, not memory-bound, at NEH = 6), each row is then loaded by G waves (the second to fourth
// from L1 / L2), and a problem's bits do not depend on the depth its batch forces on the launch.
// Per row and evaluation e: the two dot products  u = J_h[r] . p~_e + f_r,  t = J_h[r] . w~_e  (J_h = J D: the vectors
// are pre-multiplied by d), their 64-lane totals by transposed butterflies (wave_sum16), then  y_e += u J[r]  (per lane:
// its columns) and  b_e += t^2  (uniform: broadcast first, so that its order is the rows' order).  J streams straight
// into registers (the next batch is requested before the current one is consumed); no LDS on the way.
#ifndef BLSQ_CS_WGS_PER_CU
#define BLSQ_CS_WGS_PER_CU 1
#endif
template <int NCB, int NEH, int G, int NS>
__global__ __launch_bounds__(CS_NT, BLSQ_CS_WGS_PER_CU) void csne_pass_kernel(CsneState cs, const double* __restrict__ dvec) {
  static_assert(NS >= 2, "register slots of the row batches: one consumed, NS - 1 in flight");
  static_assert(G == 1 || G == 2 || G == 4, "groups of waves");
  constexpr int MEM = CS_NW / G;                          // members (waves) per group
  constexpr int NV = 2 * NEH;                             // totals per row and wave
  constexpr int NTOT = CS_RB * NV;                        // ... per batch
  constexpr int NBF = (NTOT + 15) / 16;                   // butterflies per batch
  constexpr int NE = NEH * G;                             // evaluations the launch carries (the partial sums' stride)
  extern __shared__ double ysh[];                         // [group][class][NEH][NCB][64] + [group][class][NEH]
  const int li = blockIdx.y, chunk = blockIdx.x;
  const int b = cs.list[li];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w / MEM, mem = w % MEM;                 // (wave-uniform)
  const int n = cs.n, ld = cs.ld;
  const int r0 = chunk * cs.rows_per_wg;
  const int r1 = (r0 + cs.rows_per_wg < cs.m) ? r0 + cs.rows_per_wg : cs.m;
  const double* __restrict__ Jb = cs.J + (long)b * cs.strideJ;
  const double* __restrict__ Fb = cs.F + (long)b * cs.strideF;
  int ne = cs.ne[b];
  if (ne > NE) ne = NE;                                   // (deeper recordings are declined by csne_fix_kernel)
  const int vidx = wave_sum16_index(lane);                // the total this lane receives from a butterfly
  const int e0 = grp * NEH;                               // this wave's evaluations: e0 .. e0 + NEH - 1

  // the recorded vectors of the lane's columns, pre-multiplied by d (J_h = J D); zero beyond n and beyond the
  // problem's recording: columns n .. 64 NCB of J are loaded clamped and count for nothing
  double V[NEH][2][NCB];
#pragma unroll
  for (int e = 0; e < NEH; ++e)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        double v = 0.0;
        if (e0 + e < ne && col < n)
          v = (dvec ? dvec[(long)b * ld + col] : 1.0) * cs.rvec[(((long)b * CSNE_MAXE + e0 + e) * 3 + c) * ld + col];
        V[e][c][cb] = v;
      }
  double y[G][NEH][NCB], accB[G][NEH];                    // per class of this wave
#pragma unroll
  for (int ci = 0; ci < G; ++ci)
#pragma unroll
    for (int e = 0; e < NEH; ++e) {
      accB[ci][e] = 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) y[ci][e][cb] = 0.0;
    }

  const int nbatch = (r1 - r0 + CS_RB - 1) / CS_RB;
  double jr[NS][CS_RB][NCB], fr[NS][NBF];
  auto issue = [&](int bi, auto slotc) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::value;
    const int rb = r0 + bi * CS_RB;
#pragma unroll
    for (int r = 0; r < CS_RB; ++r) {
      const int row = rb + r;
      const int rc = row < r1 ? row : r1 - 1;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        jr[slot][r][cb] = __builtin_nontemporal_load(Jb + (long)rc * cs.ldJ + (col < n ? col : n - 1));
      }
    }
#pragma unroll
    for (int g = 0; g < NBF; ++g) {
      const int row = rb + (16 * g + vidx) / NV;
      fr[slot][g] = Fb[row < r1 ? row : r1 - 1];
    }
  };
  auto consume = [&](int bi, auto slotc, auto cic) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::value;
    constexpr int ci = decltype(cic)::value;
    const int rb = r0 + bi * CS_RB;
    if (rb + CS_RB > r1) {                                 // (uniform; the last batch of a chunk only) rows beyond it: zero
#pragma unroll
      for (int r = 0; r < CS_RB; ++r)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          if (rb + r >= r1) jr[slot][r][cb] = 0.0;
#pragma unroll
      for (int g = 0; g < NBF; ++g)
        if (rb + (16 * g + vidx) / NV >= r1) fr[slot][g] = 0.0;
    }
    // one butterfly's sixteen totals at a time: their dot products, the butterfly, what the totals feed
    // (compile-time recursion instead of unrolled loops: every register-array index must be a constant)
    static_for<0, NBF>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      double v[16];
      static_for<0, 16>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int idx = 16 * g + i;
        double acc = 0.0;
        if constexpr (idx < NTOT) {
          constexpr int r = idx / NV, e = (idx % NV) / 2, c = idx % 2;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc = fma(jr[slot][r][cb], V[e][c][cb], acc);
        }
        v[i] = acc;
      });
      wave_sum16(v);
      const double tot = v[0];
      const double uu = tot + fr[slot][g];                // (meaningful on the lanes of the u totals: bit 3 clear)
      static_for<0, 8>([&](auto hc) __attribute__((always_inline)) {   // the (row, evaluation) pairs of this butterfly, in row order
        constexpr int i = 2 * decltype(hc)::value;
        constexpr int idx = 16 * g + i;
        if constexpr (idx < NTOT) {
          constexpr int r = idx / NV, e = (idx % NV) / 2;
          const double us = read_lane(uu, cs_bitrev4(i));
          const double ts = read_lane(tot, cs_bitrev4(i + 1));
          accB[ci][e] = fma(ts, ts, accB[ci][e]);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) y[ci][e][cb] = fma(us, jr[slot][r][cb], y[ci][e][cb]);
        }
      });
    });
  };
  // member `mem` takes batches mem, mem + MEM, ...: batch mem + MEM k is of class mem + MEM (k mod G), its local class
  // k mod G.  NS G batches per trip (every register slot and every local class at a compile-time index); NS - 1
  // batches in flight behind the one being consumed — the bytes a CU keeps in flight are what bounds this kernel.
  {
    int k = 0;
    auto bat = [&](int kk) { return mem + MEM * kk; };
    static_for<0, NS - 1>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      if (bat(q) < nbatch) issue(bat(q), std::integral_constant<int, q>{});
    });
    for (; bat(k) < nbatch; k += NS * G) {
      static_for<0, NS * G>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        const int bi = bat(k + q);
        if (bi < nbatch) {                                 // (uniform)
          if (bat(k + q + NS - 1) < nbatch) issue(bat(k + q + NS - 1), std::integral_constant<int, ((q + NS - 1) % NS)>{});
          consume(bi, std::integral_constant<int, (q % NS)>{}, std::integral_constant<int, q % G>{});
        }
      });
    }
  }
  // every wave's class sums to LDS; the problem's sums in class order
  double* bsh = ysh + (size_t)G * 4 * NEH * NCB * WAVE;
#pragma unroll
  for (int ci = 0; ci < G; ++ci) {
    const int cls = mem + MEM * ci;
#pragma unroll
    for (int e = 0; e < NEH; ++e) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) ysh[((((size_t)grp * 4 + cls) * NEH + e) * NCB + cb) * WAVE + lane] = y[ci][e][cb];
      if (lane == 0) bsh[((size_t)grp * 4 + cls) * NEH + e] = accB[ci][e];
    }
  }
  __syncthreads();
  double* out = cs.part + ((long)li * cs.nchunk + chunk) * csne_part_stride(NE, ld);
  for (int idx = tid; idx < NE * NCB * WAVE; idx += CS_NT) {
    const int l = idx % WAVE, cb = (idx / WAVE) % NCB, eg = idx / (WAVE * NCB);
    const int g = eg / NEH, e = eg % NEH;
    const double* src = ysh + ((((size_t)g * 4) * NEH + e) * NCB + cb) * WAVE + l;
    const size_t cs_ = (size_t)NEH * NCB * WAVE;           // class stride
    const double t = ((src[0] + src[cs_]) + src[2 * cs_]) + src[3 * cs_];
    const int col = l + 64 * cb;
    if (col < n) out[(long)eg * ld + col] = t;
  }
  if (tid < NE) {
    const int g = tid / NEH, e = tid % NEH;
    const double* src = bsh + ((size_t)g * 4) * NEH + e;
    out[(long)NE * ld + tid] = ((src[0] + src[NEH]) + src[2 * NEH]) + src[3 * NEH];
  }
}

// ---- the pass over J on the matrix pipe (TRF) ------------------------------------------------------------
// The kernel above spends most of its issue slots on the 64-lane totals of its dot products (VALU-issue-bound at six
// evaluations: 1.37 ms against the 0.63 ms the same kernel needs with ONE vector, i.e. the streaming rate).  Here
// both products of a 16-row tile are small GEMMs on the FP64 MFMA pipe (v_mfma_f64_16x16x4):
//   T = J_tile V          V = [d p~_0, d w~_0, d p~_1, ...]: sixteen columns = eight evaluations; the reduction over
//                         the columns happens inside the instruction;
//   Y += U^T J_tile       U = T (+ f on the even columns): T lands in the accumulator layout (lane (lr, lc), element g:
//                         row 4 g + lr, vector lc), which IS the A-operand layout of U^T for k-step g — no shuffle;
//                         b_e += t^2 on the odd columns, lane-local.  (The rows of Y that belong to the w~ are computed
//                         and dropped: the pipe has the room.)
// The four waves of a workgroup work on the SAME tile, each on its own slice of 16 NST columns: a wave stages its slice
// in its own LDS region (the next tile's rows already requested into registers), holds its part of V in registers as B operands, computes the
// partial T of its columns, and the four partial sums meet in LDS — one barrier per tile, the exchange area double
// buffered — before every wave multiplies the full U into its own columns of Y.  Small per-wave state (164 registers,
// 8 KB of LDS), so three workgroups share a CU and one wave's waits hide under the others' products.
// Sums are defined by the tiles: T = ((P_0 + P_1) + P_2) + P_3 over the waves' column slices (each: k-steps of even
// index in one accumulator, odd in the other, a0 + a1), Y and b summed over a chunk's tiles in order — nothing depends
// on how many evaluations the launch carries.
static constexpr int CSM_TR = 16;                         // rows per tile
// row stride of a wave's slice: = 18 mod 32 doubles, so that the A fragments J[row lc][4 s + lr] of the first product
// hit disjoint banks and the B fragments J[row 4 g + lr][16 c + lc] of the second nearly so
// Measured at 512 x (4096 x 256), six evaluations (the vector-ALU kernel: 1.37 ms): 2 workgroups per CU and 2 tiles
// ahead 0.89 ms, 2 and 1: 0.91 ms, 3 and 1 (164 registers, 50 KB of LDS): 0.81 ms.
#ifndef BLSQ_CSM_OCC
#define BLSQ_CSM_OCC 3                                    // workgroups per CU the kernel is built for
#endif
#ifndef BLSQ_CSM_PF
#define BLSQ_CSM_PF 1                                     // tiles requested ahead (register sets)
#endif
__host__ __device__ constexpr int csm_stride(int NST) { return NST <= 3 ? 50 : (BLSQ_CSM_OCC >= 3 ? 66 : 82); }
template <int NST>
__global__ __launch_bounds__(CS_NT, BLSQ_CSM_OCC) void csne_pass_mfma_kernel(CsneState cs, const double* __restrict__ dvec) {
  constexpr int PF = BLSQ_CSM_PF;
  constexpr int NK = 4 * NST;                             // k-steps of four columns in a wave's slice
  constexpr int SW = 16 * NST;                            // columns of a slice
  constexpr int S = csm_stride(NST);
  constexpr int NE = CSNE_MAXE;                           // vector slots: 2 NE = 16 = the MFMA's columns
  extern __shared__ double lds[];
  const int li = blockIdx.y, chunk = blockIdx.x;
  const int b = cs.list[li];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = cs.n, ld = cs.ld;
  const int r0 = chunk * cs.rows_per_wg;
  const int r1 = (r0 + cs.rows_per_wg < cs.m) ? r0 + cs.rows_per_wg : cs.m;
  const int c0 = w * SW;                                  // first column of this wave's slice
  const double* __restrict__ Jb = cs.J + (long)b * cs.strideJ;
  const double* __restrict__ Fb = cs.F + (long)b * cs.strideF;
  int ne = cs.ne[b];
  if (ne > NE) ne = NE;                                   // (deeper recordings are declined by csne_fix_kernel)
  double* X = lds + (size_t)w * (CSM_TR * S);             // this wave's slice of the tile [16][S]
  double* E = lds + (size_t)CS_NW * (CSM_TR * S);         // partial T: [parity][wave][g][lane]
  // B operands of the first product: V[col = c0 + 4 s + lr][vec = lc]; vec 2 e = d p~_e, vec 2 e + 1 = d w~_e (the
  // LAST evaluation recorded p~ only); zero beyond column n
  double Vb[NK];
  {
    const int e = lc >> 1, c = lc & 1;
    const bool have = e < ne && !(c == 1 && e == ne - 1);
    const double* rv = cs.rvec + (((long)b * CSNE_MAXE + e) * 3 + c) * ld;
    const double* dv = dvec + (long)b * ld;
#pragma unroll
    for (int s_ = 0; s_ < NK; ++s_) {
      const int col = c0 + 4 * s_ + lr;
      const int cc = col < n ? col : n - 1;
      const double v = dv[cc] * rv[cc];
      Vb[s_] = (have && col < n) ? v : 0.0;
    }
  }
  for (int i = lane; i < CSM_TR * S; i += WAVE) X[i] = 0.0;   // (the padding columns are read by nobody; keep them finite)
  v4d yacc[NST];
#pragma unroll
  for (int c = 0; c < NST; ++c) yacc[c] = v4d{0.0, 0.0, 0.0, 0.0};
  double accB = 0.0;
  const int ntile = (r1 - r0 + CSM_TR - 1) / CSM_TR;
  // staging: lane l < SW owns column c0 + l; PF tiles ahead
  const int mycol = c0 + lane;
  const bool colin = lane < SW && mycol < n;
  const bool allin = c0 + SW <= n;                        // (uniform: the whole slice lies inside the matrix)
  const double* __restrict__ Jc = Jb + (colin ? mycol : (n - 1));
  const long ldj = cs.ldJ;
  double jr[PF][CSM_TR], fr[PF][4];
  auto request = [&](int tl, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    const int rb = r0 + tl * CSM_TR;
    if (rb + CSM_TR <= r1) {                              // (uniform) a full tile: no clamping
      const double* rp = Jc + (long)rb * ldj;
#pragma unroll
      for (int r = 0; r < CSM_TR; ++r) jr[set][r] = __builtin_nontemporal_load(rp + r * ldj);
#pragma unroll
      for (int g = 0; g < 4; ++g) fr[set][g] = Fb[rb + 4 * g + lr];
    } else {
#pragma unroll
      for (int r = 0; r < CSM_TR; ++r) {
        const int row = rb + r;
        jr[set][r] = __builtin_nontemporal_load(Jc + (long)(row < r1 ? row : r1 - 1) * ldj);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = rb + 4 * g + lr;
        fr[set][g] = Fb[row < r1 ? row : r1 - 1];
      }
    }
  };
  auto tile = [&](int ti, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    const int rb = r0 + ti * CSM_TR;
    // commit the slice (rows beyond the chunk and columns beyond n: zero), request the tile PF ahead
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (this wave's reads of the last tile are done)
    double fv[4];
    if (allin && rb + CSM_TR <= r1) {                     // (uniform)
      if (SW == WAVE || lane < SW) {
#pragma unroll
        for (int r = 0; r < CSM_TR; ++r) X[r * S + lane] = jr[set][r];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) fv[g] = fr[set][g];
    } else {
      if (lane < SW) {
#pragma unroll
        for (int r = 0; r < CSM_TR; ++r) X[r * S + lane] = (colin && rb + r < r1) ? jr[set][r] : 0.0;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) fv[g] = (rb + 4 * g + lr < r1) ? fr[set][g] : 0.0;
    }
    if (ti + PF < ntile) request(ti + PF, setc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // partial T of this wave's columns
    v4d acc[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
    const double* Xa = X + lc * S + lr;                   // A[row = lc][k = lr] of k-step 0
#pragma unroll
    for (int s_ = 0; s_ < NK; ++s_) acc[s_ & 1] = gmfma(Xa[4 * s_], Vb[s_], acc[s_ & 1]);
    double* Ew = E + ((size_t)(ti & 1) * CS_NW + w) * (4 * WAVE);
#pragma unroll
    for (int g = 0; g < 4; ++g) Ew[g * WAVE + lane] = acc[0][g] + acc[1][g];
    __syncthreads();
    // u = T + f on the even vectors, b += t^2 on the odd ones (wave 0 keeps b)
    const double* Er = E + (size_t)(ti & 1) * CS_NW * (4 * WAVE) + lane;
    double u[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double t = ((Er[g * WAVE] + Er[(4 + g) * WAVE]) + Er[(8 + g) * WAVE]) + Er[(12 + g) * WAVE];
      u[g] = (lc & 1) ? t : t + fv[g];
      accB = (lc & 1) ? fma(t, t, accB) : accB;
    }
    // Y += U^T J_tile for this wave's columns: k-step g covers rows 4 g .. 4 g + 3
    const double* Xb = X + lr * S + lc;                   // B[k = lr][j = lc] of k-step 0, column tile 0
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < NST; ++c) yacc[c] = gmfma(u[g], Xb[4 * g * S + 16 * c], yacc[c]);
  };
  static_for<0, PF>([&](auto pc) __attribute__((always_inline)) {
    if (decltype(pc)::value < ntile) request(decltype(pc)::value, pc);
  });
  for (int ti = 0; ti < ntile; ti += PF) {                // (every wave runs every tile: the barriers match)
    static_for<0, PF>([&](auto pc) __attribute__((always_inline)) {
      if (ti + decltype(pc)::value < ntile) tile(ti + decltype(pc)::value, pc);
    });
  }
  // yacc[c], lane (lr, lc), element g: Y[vec 4 g + lr][col c0 + 16 c + lc]; vec 2 e: lr in {0, 2}, e = 2 g + lr / 2
  double* out = cs.part + ((long)li * cs.nchunk + chunk) * csne_part_stride(NE, ld);
  if ((lr & 1) == 0) {
#pragma unroll
    for (int c = 0; c < NST; ++c) {
      const int col = c0 + 16 * c + lc;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (col < n) out[(long)(2 * g + (lr >> 1)) * ld + col] = yacc[c][g];
    }
  }
  __syncthreads();                                        // (the exchange area is done with)
  if (w == 0) E[lane] = accB;
  __syncthreads();
  if (tid < NE) {                                         // b_e: lanes (lr, lc = 2 e + 1) of wave 0, lr in order
    const double* s4 = E + 2 * tid + 1;
    out[(long)NE * ld + tid] = ((s4[0] + s4[16]) + s4[32]) + s4[48];
  }
}

template <int NST>
static hipError_t csne_pass_mfma_launch(const CsneState& cs, const double* dvec, int count, hipStream_t s) {
  const size_t lds = sizeof(double) * ((size_t)CS_NW * CSM_TR * csm_stride(NST) + 2 * CS_NW * 4 * WAVE);
  static std::atomic<size_t> granted[64];
  hipError_t ge = gram_grant_lds(csne_pass_mfma_kernel<NST>, lds, granted);
  if (ge != hipSuccess) return ge;
  hipLaunchKernelGGL((csne_pass_mfma_kernel<NST>), dim3(cs.nchunk, count), dim3(CS_NT), lds, s, cs, dvec);
  return hipGetLastError();
}
// cs.NE must be CSNE_MAXE (the partial sums carry all eight evaluation slots, whatever the batch's depth)
hipError_t launch_csne_pass_mfma(const CsneState& cs, const double* dvec, int count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  if (cs.NE != CSNE_MAXE || !dvec) return hipErrorInvalidValue;
  const int nst = (cs.n + 63) / 64;                       // column tiles of sixteen per wave
  if (nst <= 2) return csne_pass_mfma_launch<2>(cs, dvec, count, s);
  if (nst == 3) return csne_pass_mfma_launch<3>(cs, dvec, count, s);
  if (nst == 4) return csne_pass_mfma_launch<4>(cs, dvec, count, s);
  return hipErrorInvalidValue;
}

// the split of a launch that carries NE evaluations: (NEH, G) with NEH G >= NE.  Measured (512 problems of 4096 x 256,
// six evaluations): one group 1.51 ms, two groups of three evaluations 1.66 ms — the kernel is bound by the bytes a CU
// keeps in flight (four waves x one batch of 8 KB: what the loaded latency lets through), and a row requested by two
// waves halves the UNIQUE bytes in flight; the split stays in the kernel (G > 1 compiles and is tested through
// BLSQ_CS_GROUPS builds) but the launch takes G = 1.
#ifndef BLSQ_CS_GROUPS
#define BLSQ_CS_GROUPS 1
#endif
static void csne_split(int NE, int* neh, int* g) {
  if (BLSQ_CS_GROUPS == 1 || NE <= 3) { *neh = NE < 1 ? 1 : NE; *g = 1; }
  else if (NE == 4) { *neh = 2; *g = 2; }
  else if (NE <= 6) { *neh = 3; *g = 2; }
  else { *neh = 2; *g = 4; }
}
int csne_launch_evals(int NE) { int a, b; csne_split(NE, &a, &b); return a * b; }

#ifndef BLSQ_CS_SLOTS_DEEP
#define BLSQ_CS_SLOTS_DEEP 4           // evaluations up to which THREE register slots (two batches in flight) fit the registers
#endif
template <int NCB, int NEH, int G>
static hipError_t csne_pass_launch1(const CsneState& cs, const double* dvec, int count, hipStream_t s) {
  constexpr int NS = (NEH <= BLSQ_CS_SLOTS_DEEP) ? 3 : 2;
  const size_t lds = sizeof(double) * ((size_t)G * 4 * NEH * NCB * WAVE + (size_t)G * 4 * NEH);
  static std::atomic<size_t> granted[64];
  hipError_t ge = gram_grant_lds(csne_pass_kernel<NCB, NEH, G, NS>, lds, granted);
  if (ge != hipSuccess) return ge;
  hipLaunchKernelGGL((csne_pass_kernel<NCB, NEH, G, NS>), dim3(cs.nchunk, count), dim3(CS_NT), lds, s, cs, dvec);
  return hipGetLastError();
}
template <int NCB>
static hipError_t csne_pass_launch(const CsneState& cs, const double* dvec, int count, hipStream_t s) {
  int neh = 1, g = 1;
  csne_split(cs.NE, &neh, &g);
  if (neh * g != cs.NE) return hipErrorInvalidValue;       // (the host sizes the partial sums with csne_launch_evals)
  if (g == 1) {
    switch (neh) {
      case 1: return csne_pass_launch1<NCB, 1, 1>(cs, dvec, count, s);
      case 2: return csne_pass_launch1<NCB, 2, 1>(cs, dvec, count, s);
      case 3: return csne_pass_launch1<NCB, 3, 1>(cs, dvec, count, s);
      case 4: return csne_pass_launch1<NCB, 4, 1>(cs, dvec, count, s);
      case 5: return csne_pass_launch1<NCB, 5, 1>(cs, dvec, count, s);
      case 6: return csne_pass_launch1<NCB, 6, 1>(cs, dvec, count, s);
      case 7: return csne_pass_launch1<NCB, 7, 1>(cs, dvec, count, s);
      case 8: return csne_pass_launch1<NCB, 8, 1>(cs, dvec, count, s);
    }
  }
#if BLSQ_CS_GROUPS != 1
  if (g == 2 && neh == 2) return csne_pass_launch1<NCB, 2, 2>(cs, dvec, count, s);
  if (g == 2 && neh == 3) return csne_pass_launch1<NCB, 3, 2>(cs, dvec, count, s);
  if (g == 4 && neh == 2) return csne_pass_launch1<NCB, 2, 4>(cs, dvec, count, s);
#endif
  return hipErrorInvalidValue;
}
hipError_t launch_csne_pass(const CsneState& cs, const double* dvec, int count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  const int ncb = (cs.n + 63) / 64;
  if (ncb <= 2) return csne_pass_launch<2>(cs, dvec, count, s);
  if (ncb == 3) return csne_pass_launch<3>(cs, dvec, count, s);
  if (ncb == 4) return csne_pass_launch<4>(cs, dvec, count, s);
  return hipErrorInvalidValue;
}

// ---- the correction, in n-space -------------------------------------------------------------------
// NSUM block totals at once: per-thread partials -> one transposed butterfly per wave -> the waves in order.
// scr: CS_NW * 16 doubles of LDS.  Every thread gets all totals.  Fixed order.
template <int NSUM>
__device__ __forceinline__ void cs_block_sums(double (&part)[NSUM], double* scr) {
  static_assert(NSUM <= 16, "one butterfly");
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = i < NSUM ? part[i] : 0.0;
  wave_sum16(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                                        // (scr may still be read from an earlier call)
  if (lane < 16) scr[w * 16 + wave_sum16_index(lane)] = v[0];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NSUM; ++i) {
    double t = 0.0;
    for (int q = 0; q < TRI_NW; ++q) t += scr[q * 16 + i];
    part[i] = t;
  }
}

__global__ __launch_bounds__(TRI_NT) void csne_fix_kernel(CsneState cs, TrfState st, LmState lm,
                                                          const double* Delta_in, const double* alpha_in) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  __shared__ double scr[TRI_NW * 16];
  __shared__ double kphi[CSNE_MAXE], kdphi[CSNE_MAXE], kddphi[CSNE_MAXE], keta[CSNE_MAXE];
  const int li = blockIdx.x;
  const int b = cs.list[li];
  const int tid = threadIdx.x;
  const int n = cs.n, ld = cs.ld;
  const long vo = (long)b * ld;
  double* res = sh;
  double* pv = res + ld;
  double* wv = pv + ld;
  double* zv = wv + ld;
  double* invd = zv + ld;
  double* q = invd + ld;
  double* pfbuf = q + ld;                                 // 2 x 16 x ld doubles: DMA staging of the solves
  const double* dg = st.d + vo;
  const double* gh = st.g_h + vo;
  const double* dh = st.diag_h + vo;
  const double Delta = Delta_in[b];
  const int ne = cs.ne[b];
  const int NE = cs.NE;
  const long PS = csne_part_stride(NE, ld);
  const double* part = cs.part + (long)li * cs.nchunk * PS;
  bool fail = (ne < 1 || ne > CSNE_MAXE || ne > NE);      // (uniform)
  // ||g_h|| with the operations of lm_start_body: the bracket's upper end is the cheap run's, bit for bit
  for (int i = tid; i < n; i += TRI_NT) q[i] = gh[i];
  __syncthreads();
  const double gnorm = sqrt(tri_dot<TRI_NT>(q, q, n, red));
  // residual of recorded evaluation k into `res`, its vectors into pv / wv / zv; the eight sums
  auto load_eval = [&](int k, double (&sums)[8]) {
    const double ak = cs.ralpha[(long)b * CSNE_MAXE + k];
    const double* rec = cs.rvec + ((long)b * CSNE_MAXE + k) * 3 * ld;
    const bool last = k == ne - 1;                        // (the last evaluation recorded p~ only)
    double pp = 0.0, pw = 0.0, ww = 0.0, wr = 0.0, zr = 0.0, wdw = 0.0, rr = 0.0;
    __syncthreads();
    for (int j = tid; j < n; j += TRI_NT) {
      double yj = 0.0;
      for (int c = 0; c < cs.nchunk; ++c) yj += part[(long)c * PS + (long)k * ld + j];
      const double pj = rec[j], wj = last ? 0.0 : rec[ld + j], zj = last ? 0.0 : rec[2 * ld + j];
      const double sh_ = dh[j] + ak;
      const double rj = fma(dg[j], yj, sh_ * pj);        // d_j (J^T u)_j + (diag_h + alpha) p~_j
      res[j] = rj; pv[j] = pj; wv[j] = wj; zv[j] = zj;
      pp = fma(pj, pj, pp); pw = fma(pj, wj, pw); ww = fma(wj, wj, ww);
      wr = fma(wj, rj, wr); zr = fma(zj, rj, zr); wdw = fma(sh_ * wj, wj, wdw); rr = fma(rj, rj, rr);
    }
    sums[0] = pp; sums[1] = pw; sums[2] = ww; sums[3] = 0.0; sums[4] = wr; sums[5] = zr; sums[6] = wdw; sums[7] = rr;
    cs_block_sums<8>(sums, scr);
  };
  // ---- phi, phi', phi'' of the evaluations the iteration went on from, corrected to first order ----
  if (!fail) {
    for (int k = 0; k + 1 < ne; ++k) {
      double s8[8];
      load_eval(k, s8);
      const double pp = s8[0], pw = s8[1], ww = s8[2], wr = s8[4], zr = s8[5], wdw = s8[6];
      double bk = 0.0;
      for (int c = 0; c < cs.nchunk; ++c) bk += part[(long)c * PS + (long)NE * ld + k];
      const double pn2 = pp - 2.0 * wr;                   // ||p||^2 to first order
      const double pn = sqrt(pn2);
      const double pMp = (2.0 * pw - (bk + wdw)) - 2.0 * zr;   // p^T M^-1 p to first order
      const double e1 = fabs(2.0 * wr) / pp, e2 = fabs(pMp - pw) / fabs(pw);
      if (tid == 0) {
        kphi[k] = pn - Delta;
        kdphi[k] = -pMp / pn;
        // phi'' = (w.w + 2 p.z) / |p| - (p.w)^2 / |p|^3 with p.z = p^T M^-1 w = w.w (M symmetric): from the cheap vectors
        kddphi[k] = 3.0 * ww / pn - (pw * pw) / (pn * pn * pn);
        keta[k] = e1 > e2 ? e1 : e2;
      }
      if (!(pn2 > 0.0) || !(pMp > 0.0) || !is_finite(pn2) || !is_finite(pMp)) fail = true;
    }
  }
  __syncthreads();
  // ---- replay of trust_region.py:111-150 on the corrected phi, phi': it must arrive at the LAST recorded evaluation
  //      (index ne - 1) exactly as the cheap run did, and stop there ----
  const int kf = ne - 1;
  int n_iter = 0;
  double alpha = 0.0, eps = 0.0, eta = 0.0;
  if (!fail && ne > 1) {
    if (kphi[0] <= 0.0) fail = true;                      // (the corrected Gauss-Newton step is inside: the cheap run went on)
    double lo = -kphi[0] / kdphi[0];                      // :121-123
    double hi = gnorm / Delta;                            // :119
    eta = keta[0];
    alpha = alpha_in[b];                                  // :127-130 (full rank)
    if (alpha < lo || alpha > hi) alpha = lm_restart(lo, hi);
    int it = 0;
    for (int k = 1; !fail; ++k) {
      const double e = alpha - cs.ralpha[(long)b * CSNE_MAXE + k];
      if (!(fabs(e) <= 1.0e-3 * fabs(cs.ralpha[(long)b * CSNE_MAXE + k]))) { fail = true; break; }   // (not the iterate the cheap run evaluated)
      if (k == kf) { eps = e; n_iter = it + 1; break; }   // the evaluation the cheap run ended with: judged on the corrected step below
      const double phi = kphi[k] + (kdphi[k] + 0.5 * kddphi[k] * e) * e;
      const double dphi = kdphi[k] + kddphi[k] * e;
      eta = keta[k] > eta ? keta[k] : eta;
      if (fabs(phi) < 0.01 * Delta) { fail = true; break; }   // :138-139 — the replay ends where the cheap run went on
      if (phi < 0.0) hi = alpha;                          // :141-142
      const double ratio = phi / dphi;
      const double cand = alpha - ratio;
      lo = (cand > lo) ? cand : lo;                       // :145
      alpha -= (phi + Delta) * ratio / Delta;             // :146
      ++it;
      if (it >= 10) { fail = true; break; }               // (ten rounds: beyond what the tier records)
      if (alpha < lo || alpha > hi) alpha = lm_restart(lo, hi);   // :133-134 of the next pass
    }
  }
  if (!fail) {
    // ---- the final step: p(alpha) = p~ + M~^-1 (-res - eps p~), one corrected solve with the factor of the last
    //      evaluation (dp/dalpha = -M^-1 p: the shift to the replayed alpha rides on the same solve) ----
    double s8[8];
    load_eval(kf, s8);
    const double pp = s8[0];
    const double* R = (kf == 0 ? lm.Raug : lm.Xa) + (long)b * ld * ld;
    tri_invdiag<TRI_NT>(R, n, ld, invd);
    for (int j = tid; j < n; j += TRI_NT) q[j] = -(res[j] + eps * pv[j]);
    __syncthreads();
    tri_solve_upper_t_pf<TRI_NT>(R, n, ld, invd, q, pfbuf);
    tri_solve_upper_pf<TRI_NT>(R, n, ld, invd, q, pfbuf);
    double s2[2] = {0.0, 0.0};
    for (int j = tid; j < n; j += TRI_NT) {
      const double dj = q[j];
      const double pj = pv[j] + dj;
      res[j] = pj;
      s2[0] = fma(pj, pj, s2[0]); s2[1] = fma(dj, dj, s2[1]);
    }
    cs_block_sums<2>(s2, scr);
    const double pn = sqrt(s2[0]);
    const double dpn = sqrt(s2[1] / pp);
    eta = dpn > eta ? dpn : eta;
    const double phi_f = pn - Delta;
    if (!(eta <= CSNE_ETA_MAX) || !is_finite(pn) || !(pn > 0.0)) fail = true;
    if (ne == 1) { if (!(pn <= Delta)) fail = true; }     // :116-117 on the corrected Gauss-Newton step
    else if (!(fabs(phi_f) < 0.01 * Delta)) fail = true;  // :138-139 on the corrected step
    if (!fail) {
      const double c = (ne > 1 && phi_f > 0.0) ? Delta / pn : 1.0;   // :149-150
      for (int j = tid; j < n; j += TRI_NT) {
        const double pj = res[j] * c;
        lm.ph[vo + j] = pj;
        cs.hp[vo + j] = -(c * gh[j] + alpha * pj);        // H p_h by the normal equations the corrected p satisfies
      }
      if (tid == 0) {
        lm.sc[(long)b * 16 + SC_ALPHA] = alpha;
        lm.st[(long)b * 4 + ST_NITER] = n_iter;
      }
    }
  }
  if (tid == 0) {
    cs.eta[b] = fail ? -1.0 : eta;
    if (fail) cs.fail_list[atomicAdd(cs.counts + 1, 1)] = b;
  }
}

hipError_t launch_csne_fix(const CsneState& cs, const TrfState& st, const LmState& lm, const double* Delta,
                           const double* alpha_in, int count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  const size_t lds = sizeof(double) * (6 + 32) * (size_t)cs.ld;
  static std::atomic<size_t> granted[64];
  hipError_t ge = gram_grant_lds(csne_fix_kernel, lds, granted);
  if (ge != hipSuccess) return ge;
  hipLaunchKernelGGL(csne_fix_kernel, dim3(count), dim3(TRI_NT), lds, s, cs, st, lm, Delta, alpha_in);
  return hipGetLastError();
}

// ---- who is on the tier ---------------------------------------------------------------------------
// order-preserving compaction by ONE workgroup of 256 threads: out[0 .. return) = the entries e of
// get(0 .. count) with keep(e); in place allowed (the writes of a tile never pass its reads)
template <class Get, class Keep>
__device__ __forceinline__ int cs_compact(int count, int* out, Get get, Keep keep, int* scr) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int base = 0;
  for (int t0 = 0; t0 < count; t0 += 256) {
    const int i = t0 + tid;
    const int e = i < count ? get(i) : 0;
    const bool k = i < count && keep(e);
    const unsigned long long bal = __ballot(k);
    __syncthreads();
    if (lane == 0) scr[w] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int q = 0; q < w; ++q) off += scr[q];
    if (k) out[off + __popcll(bal & ((1ull << lane) - 1ull))] = e;
    base += scr[0] + scr[1] + scr[2] + scr[3];
  }
  __syncthreads();
  return base;
}

// rank-gate outputs of a selected problem: TRF (LmState) or dogbox (fast / Jacobi-mask arrays); mrank = the row count
// of the reference's rank test, ncols (optional): per-problem column count + 1 of the system factored (dogbox's free block)
struct CsneGateOut {
  LmState lm; int use_lm;
  int* fast; int* ncols_jac;
  const int* ncols; int mrank;
};
__global__ __launch_bounds__(256) void csne_select_kernel(CsneState cs, CsneGateOut go, int nfb, int* tree_list, int* tree_mask,
                                                          int* tree_count, int* path, const int* sel_mask,
                                                          const double* k2, const double* pmin, const double* colinfo) {
  __shared__ int scr[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < nfb; i += 256) {
    const int b = tree_list[i];
    const int n = go.ncols ? go.ncols[b] - 1 : cs.n;
    const double kb = k2[b], mn = colinfo[2 * (long)b], sm = colinfo[2 * (long)b + 1];
    bool ok = sel_mask[b] == 0 && kb > 0.0 && kb <= CSNE_K2_MAX && pmin[b] >= CSNE_PIVOT_FLOOR && go.mrank >= n && n >= 1;
    // the reference's rank test (trust_region.py:108-110; gelsd's rcond = eps max(m, n), dogbox.py:197) by the proven
    // bound: s_min(R)^2 >= min_j h_jj / K2
    const int mx = go.mrank > n ? go.mrank : n;
    const double smin_lb = mn / sqrt(kb), smax_ub = sqrt(sm);
    ok = ok && is_finite(sm) && sm > 0.0 && smin_lb > LM_GATE_MARGIN * LM_EPS * mx * smax_ub;
    if (ok) {
      cs.flag[b] = 1; path[b] = 0; tree_mask[b] = 0;
      if (go.use_lm) {
        go.lm.fast[b] = 1; go.lm.ncols_jac[b] = 0;
        go.lm.sc[(long)b * 16 + SC_SMAX] = smax_ub;
        go.lm.sc[(long)b * 16 + SC_SMIN] = smin_lb;
        go.lm.st[(long)b * 4 + ST_PHASE] = LM_IDLE;
      } else {
        go.fast[b] = 1; go.ncols_jac[b] = 0;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  const int ntree = cs_compact(nfb, tree_list, [&](int i) { return tree_list[i]; },
                               [&](int b) { return cs.flag[b] == 0; }, scr);
  const int ncs = cs_compact(cs.B, cs.list, [&](int i) { return i; }, [&](int b) { return cs.flag[b] != 0; }, scr);
  if (tid == 0) { *tree_count = ntree; cs.counts[0] = ncs; }
}

hipError_t launch_csne_select(const CsneState& cs, const LmState& lm, int nfb, int* tree_list, int* tree_mask,
                              int* tree_count, int* path, const int* sel_mask, const double* k2, const double* pmin,
                              const double* colinfo, hipStream_t s) {
  CsneGateOut go{lm, 1, nullptr, nullptr, nullptr, lm.m};
  hipLaunchKernelGGL(csne_select_kernel, dim3(1), dim3(256), 0, s, cs, go, nfb, tree_list, tree_mask, tree_count, path,
                     sel_mask, k2, pmin, colinfo);
  return hipGetLastError();
}
hipError_t launch_csne_select_dog(const CsneState& cs, int m, const int* ncols, int* fast, int* ncols_jac, int nfb,
                                  int* tree_list, int* tree_mask, int* tree_count, int* path, const int* sel_mask,
                                  const double* k2, const double* pmin, const double* colinfo, hipStream_t s) {
  CsneGateOut go{LmState{}, 0, fast, ncols_jac, ncols, m};
  hipLaunchKernelGGL(csne_select_kernel, dim3(1), dim3(256), 0, s, cs, go, nfb, tree_list, tree_mask, tree_count, path,
                     sel_mask, k2, pmin, colinfo);
  return hipGetLastError();
}

// ---- dogbox on the tier ------------------------------------------------------------------------------
// the cheap Newton step of the free block (free order) -> the full-length vector the pass multiplies J with (zeros at
// the active variables): recording 0 of the problem, depth 1
__global__ __launch_bounds__(256) void dog_csne_scatter_kernel(CsneState cs, DogState st) {
  const int b = cs.list[blockIdx.x];
  const int tid = threadIdx.x, n = cs.n, ld = cs.ld;
  const long vo = (long)b * ld;
  double* rec = cs.rvec + ((long)b * CSNE_MAXE) * 3 * ld;
  const int nf = st.ncols[b] - 1;
  for (int j = tid; j < n; j += 256) { rec[j] = 0.0; rec[ld + j] = 0.0; }
  __syncthreads();
  for (int q = tid; q < nf; q += 256) rec[st.free_idx[vo + q]] = st.newton[vo + q];
  if (tid == 0) { cs.ne[b] = 1; cs.ralpha[(long)b * CSNE_MAXE] = 0.0; }
}
hipError_t launch_dog_csne_scatter(const CsneState& cs, const DogState& st, int count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(dog_csne_scatter_kernel, dim3(count), dim3(256), 0, s, cs, st);
  return hipGetLastError();
}

// newton += -(X^T X)^-1 J_free^T (J_free newton + f): the residual of the cheap solve against J itself, one corrected
// solve with the free block's factor; eta = |correction| / |newton| must stay below CSNE_ETA_MAX
__global__ __launch_bounds__(TRI_NT) void dog_csne_fix_kernel(CsneState cs, DogState st) {
  extern __shared__ double sh[];
  __shared__ double scr[TRI_NW * 16];
  const int li = blockIdx.x;
  const int b = cs.list[li];
  const int tid = threadIdx.x, ld = cs.ld;
  const long vo = (long)b * ld;
  const int nf = st.ncols[b] - 1;
  double* q = sh;
  double* invd = q + ld;
  double* pfbuf = invd + ld;                              // 2 x 16 x ld doubles: DMA staging of the solves
  const long PS = csne_part_stride(cs.NE, ld);
  const double* part = cs.part + (long)li * cs.nchunk * PS;
  const int* fidx = st.free_idx + vo;
  const double* R = st.X + (long)b * ld * ld;
  for (int k = tid; k < nf; k += TRI_NT) {
    const int j = fidx[k];
    double yj = 0.0;
    for (int c = 0; c < cs.nchunk; ++c) yj += part[(long)c * PS + j];
    q[k] = -yj;
  }
  tri_invdiag<TRI_NT>(R, nf, ld, invd);
  __syncthreads();
  tri_solve_upper_t_pf<TRI_NT>(R, nf, ld, invd, q, pfbuf);
  tri_solve_upper_pf<TRI_NT>(R, nf, ld, invd, q, pfbuf);
  double s2[2] = {0.0, 0.0};
  for (int k = tid; k < nf; k += TRI_NT) {
    const double dk = q[k], pk = st.newton[vo + k];
    s2[0] = fma(pk, pk, s2[0]); s2[1] = fma(dk, dk, s2[1]);
  }
  cs_block_sums<2>(s2, scr);
  const double eta = sqrt(s2[1] / s2[0]);
  const bool fail = !(eta <= CSNE_ETA_MAX) || !is_finite(s2[0]) || !(s2[0] > 0.0);
  if (!fail)
    for (int k = tid; k < nf; k += TRI_NT) st.newton[vo + k] += q[k];
  if (tid == 0) {
    cs.eta[b] = fail ? -1.0 : eta;
    if (fail) cs.fail_list[atomicAdd(cs.counts + 1, 1)] = b;
  }
}
hipError_t launch_dog_csne_fix(const CsneState& cs, const DogState& st, int count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  const size_t lds = sizeof(double) * (2 + 32) * (size_t)cs.ld;
  static std::atomic<size_t> granted[64];
  hipError_t ge = gram_grant_lds(dog_csne_fix_kernel, lds, granted);
  if (ge != hipSuccess) return ge;
  hipLaunchKernelGGL(dog_csne_fix_kernel, dim3(count), dim3(TRI_NT), lds, s, cs, st);
  return hipGetLastError();
}

// the problems whose acceptance failed in the last step call leave the tier for good (until the next factor call)
__global__ __launch_bounds__(256) void csne_reroute_kernel(CsneState cs, int nfail, int* tree_list, int* tree_mask, int* path) {
  __shared__ int scr[4];
  const int tid = threadIdx.x;
  // (nfail < 0: only the list is rebuilt from the flags — a masked factor call has refreshed some problems)
  for (int b = tid; b < cs.B && nfail >= 0; b += 256) tree_mask[b] = 0;
  __syncthreads();
  for (int i = tid; i < nfail; i += 256) {
    const int b = cs.fail_list[i];
    cs.flag[b] = 0; path[b] = cs.n + 1; tree_mask[b] = cs.n + 1; tree_list[i] = b;
  }
  __threadfence_block();
  __syncthreads();
  const int ncs = cs_compact(cs.B, cs.list, [&](int i) { return i; }, [&](int b) { return cs.flag[b] != 0; }, scr);
  if (tid == 0) cs.counts[0] = ncs;
}
hipError_t launch_csne_reroute(const CsneState& cs, int nfail, int* tree_list, int* tree_mask, int* path, hipStream_t s) {
  hipLaunchKernelGGL(csne_reroute_kernel, dim3(1), dim3(256), 0, s, cs, nfail, tree_list, tree_mask, path);
  return hipGetLastError();
}

}  // namespace blsq
