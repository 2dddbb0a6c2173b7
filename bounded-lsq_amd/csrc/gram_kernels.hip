// Normal-equations fast path of the Jacobian factorisation (north_star: "MFMA ... for the
// tall-skinny J^T J"; reference call sites: trf.py:244-252 / dogbox.py:197 need R and Q^T f of
// [J f] only through R^T R and R^T (Q^T f)).
//
//   G = [J f]^T [J f]            one streaming pass over J, FP64 MFMA, upper tile blocks only
//   C = D G D,  D = diag(1/||J_j||)                    (unit diagonal: column equilibration)
//   C = R'^T R'                  blocked Cholesky, 16x16 tiles, MFMA Schur updates
//   [R z; 0 rho] = R' D^-1       the same triangle the Householder TSQR tree produces (up to row
//                                signs, which nothing downstream depends on)
//
// Half the flops of Householder QR, no reflector traffic, no per-panel latency chain — but the
// error of R grows with kappa(J D)^2 instead of kappa(J D).  The path is therefore GATED per
// problem by a PROVEN upper bound on the condition number of the equilibrated system that is
// actually solved (gram_cond_kernel below: explicit inverse of the Cholesky factor, exact 1- and
// inf-norms); a problem whose bound exceeds GRAM_K2_MAX is handed to the Householder tree through a
// launch mask.
// Non-finite input, zero columns, rank deficiency and m < n all fail the gate by construction.
#include "gram_common.h"

namespace blsq {

// ---- G = [J f]^T [J f] ---------------------------------------------------------------------
// One workgroup per (row chunk, problem).  The chunk streams through LDS 32 rows at a time
// (double-buffered, next rows prefetched into registers during the MFMA burst); the NT (NT+1)/2
// upper 16x16 output tiles are dealt to the 8 waves in contiguous runs of the row-major tile
// order and stay in accumulators for the whole pass.  Both MFMA operands of tile (i, j) are the
// same fragment pattern X[4 s + lr][16 c + lc] (c = i for A, c = j for B), read from LDS; the LDS
// row stride is an odd multiple of 16 doubles so the two rows a half-wave reads hit disjoint banks.
template <int SLOTS, int NCB>
__global__ __launch_bounds__(GR_NT, (SLOTS <= 8 ? 4 : 2)) void gram_kernel(GramArgs a) {
  extern __shared__ double lds[];
  const int b = a.list ? a.list[blockIdx.y] : (int)blockIdx.y;
  if (a.mask && a.mask[b] <= 1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, N = n + 1, NT = (N + 15) / 16;
  const int LDX = gram_ldx(NT);
  // The MFMA tiles cover J^T J only.  The rhs column (J^T f, f^T f) would cost a whole tile column
  // — 17 of 153 tiles at n = 256 — for one useful column per tile; it is accumulated beside the
  // MFMA stream by the vector ALUs from the same staged rows instead (32 FMAs per thread and chunk).
  // (Only where that saves a tile slot per wave — a.rhs_valu, decided by launch_gram.)
  const int NTJ = a.rhs_valu ? (n + 15) / 16 : NT;
  const int ntile = NTJ * (NTJ + 1) / 2;
  // Small batches: the tiles of a problem are split over gridDim.z workgroups (tile groups), each
  // streaming the same rows for its share of the tiles — a tile still sees the same k-steps in
  // the same order, so the result does not depend on the split.
  const int tpg = (ntile + (int)gridDim.z - 1) / (int)gridDim.z;   // tiles per group
  const int q_lo = (int)blockIdx.z * tpg;
  const int q_hi = (q_lo + tpg < ntile) ? q_lo + tpg : ntile;
  const int per = (tpg + GR_NW - 1) / GR_NW;
  const bool rhs_here = a.rhs_valu && blockIdx.z == 0;  // one group accumulates the rhs column
  // rows of this chunk
  const int r_lo = blockIdx.x * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;                                   // exclusive row limit
  const long bsrc = a.src_by_pos ? (long)blockIdx.y : (long)b;   // (the source indexed by list position: CholeskyQR2's W)
  const double* Jb = a.J + bsrc * a.strideJ;
  const double* Fb = a.F + bsrc * a.strideF;

  // tile table (wave-uniform): slot t -> (ti, tj); idle slots recompute tile (0, 0) and are
  // not stored (no branch inside the MFMA loop)
  int ti[SLOTS], tj[SLOTS];
  bool tv[SLOTS];
  {
    int i = 0, rem = q_lo + w * per;
    while (i < NTJ && rem >= NTJ - i) { rem -= NTJ - i; ++i; }
    int j = i + rem;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
      const bool valid = (t < per) && (i < NTJ) && (q_lo + w * per + t < q_hi);
      tv[t] = valid;
      ti[t] = valid ? 16 * i : 0;
      tj[t] = valid ? 16 * j : 0;
      ++j;
      if (j >= NTJ) { ++i; j = i; }
    }
  }

  // Staging: wave w owns rows w, w + 8, w + 16, w + 24 of a chunk, lanes run along the row (NCB
  // blocks of 64 columns of J; the rhs f is column n, written by lane 0).  The rows of the NEXT
  // chunk are prefetched into registers in two halves, each behind half of the MFMA burst.
  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];
  auto issue = [&](int row0, int h) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int row = row0 + w + GR_NW * (HR * h + rr);
      const int rc = row < m ? row : m - 1;
      fpre[rr] = Fb[rc];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        const int cc = col < n ? col : n - 1;
        pre[rr][cb] = __builtin_nontemporal_load(Jb + (long)rc * a.ldJ + cc);
      }
    }
  };
  // rhs column in registers (a.rhs_valu): a wave has the rows it stages in registers, one column
  // per lane and 64-column block, so it adds its rows' share of J^T f / f^T f before committing
  // them (rows w, w + 8, ... in order); the eight per-wave partials are added in wave order at
  // the end.  gram16_wave below does exactly the same, so the two kernels agree bit for bit.
  double gf[NCB], gff = 0.0;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) gf[cb] = 0.0;
  auto commit = [&](int row0, int h, double* X) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int lrow = w + GR_NW * (HR * h + rr);
      const bool in = row0 + lrow < m;
      const double fv = in ? fpre[rr] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
        if (rhs_here) gf[cb] = fma(pre[rr][cb], fv, gf[cb]);   // (columns >= n: clamped loads, never stored)
      }
      if (rhs_here) gff = fma(fv, fv, gff);
      if (lane == 0) X[lrow * LDX + n] = fv;
    }
  };

  v4d acc[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  // padding columns (n, LDX) of both buffers are zero for the whole pass
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    const unsigned xb = lds_addr(X) + 8u * (unsigned)(lr * LDX + lc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (more) issue(row0 + GR_RC, h);
      // Operand fragments are fetched TWO MFMAs ahead of their use: the LDS round trip under
      // load is longer than one MFMA of each of the SIMD's two waves.  The reads and the counted
      // waits are explicit (the compiler would otherwise fold the stages back into one register
      // set and wait for every read right before its MFMA).
      constexpr int KS = GR_RC / 4 / 2;                 // k-steps per half
      constexpr int TOT = KS * SLOTS;
      double fa[3], fb[3];
      auto fetch = [&](int q) {
        const unsigned so = 8u * (unsigned)((KS * h + q / SLOTS) * 4 * LDX);
        lds_read64(fa[q % 3], xb + so + 8u * (unsigned)ti[q % SLOTS]);
        lds_read64(fb[q % 3], xb + so + 8u * (unsigned)tj[q % SLOTS]);
      };
      fetch(0);
      fetch(1);
#pragma unroll
      for (int q = 0; q < TOT; ++q) {
        if (q + 2 < TOT) {
          fetch(q + 2);
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[q % 3]), "+v"(fb[q % 3]));
        } else if (q + 1 < TOT) {
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[q % 3]), "+v"(fb[q % 3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[q % 3]), "+v"(fb[q % 3]));
        }
        acc[q % SLOTS] = gmfma(fa[q % 3], fb[q % 3], acc[q % SLOTS]);
      }
      if (more) commit(row0 + GR_RC, h, Xn);
    }
    __syncthreads();
  }

  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    if (tv[t]) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        G[(long)(ti[t] + lr + 4 * g) * a.NPAD + tj[t] + lc] = acc[t][g];
    }
  }
  if (!rhs_here) return;
  // rhs column: the eight per-wave partials, added in wave order (the last barrier is behind us)
  constexpr int GW = 64 * NCB;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) lds[w * GW + 64 * cb + lane] = gf[cb];
  if (lane == 0) lds[GR_NW * GW + w] = gff;
  __syncthreads();
  for (int c = tid; c < n; c += GR_NT) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[ww * GW + c];
    G[(long)c * a.NPAD + n] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[GR_NW * GW + ww];
    G[(long)n * a.NPAD + n] = sum;
  }
  // the rest of the rhs tile column of the slot: padding columns (n, 16 NT) stay zero
  for (int e = tid; e < a.NPAD * (16 * NT - N); e += GR_NT) {
    const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

// ---- 16 column tiles (n = 241 .. 256): static tile rows per wave -------------------------------
// The generic kernel above keeps its tile table in registers, so every MFMA fetches both operand
// fragments from LDS (2 ds_read_b64 per MFMA) — nothing tells the compiler that two tiles share one.
// Here the assignment is a compile-time function of the wave: wave W owns tile rows W and 15 - W of
// the upper triangle, (16 - W) + (W + 1) = 17 tiles for every wave.  Both operands of tile (i, j) are
// the fragment pattern X[4 s + lr][16 c + lc] with c = i resp. c = j, and all of a wave's tiles have
// i, j >= W: a k-step needs the 16 - W fragments of column tiles W .. 15 ONCE (9 .. 16 reads for 17
// MFMAs instead of 34), held in two register sets so that the reads of k-step s + 1 fly behind the
// MFMAs of k-step s.  Each wave runs its own specialisation of the row loop (wave-uniform switch);
// the workgroup barriers are the same instruction in every specialisation.
// The rhs column (J^T f, f^T f) at n = 256 costs no LDS pass either: a wave has the rows it stages in
// registers (one column per lane and 64-column block), so it accumulates its rows' contribution
// there before committing them; the eight per-wave partials are added in a fixed order at the end.
// PAIR: one workgroup takes BOTH row chunks of a problem of two chunks, one after the other: at the
// chunk boundary the accumulators go to the output slot and restart from zero, at the end the first
// chunk's tiles are read back and added — (0 + P0) + P1, exactly what gram_reduce_kernel computes
// from two partial Grams, without writing the second one, reading both and a launch in between.
#ifdef BLSQ_CHOL_STAMPS
__device__ long long g_gram_st[8][130][4];             // [wave][chunk][phase] of ONE workgroup (diagnostic build)
#define GST(c, i) do { if (stp && lane == 0 && (c) < 130) g_gram_st[W][c][i] = (long long)wall_clock64(); } while (0)
#else
#define GST(c, i) do { } while (0)
#endif
template <int W, bool RHS, bool PAIR>
__device__ __forceinline__ void gram16_wave(const GramArgs& a, double* lds) {
  constexpr int LDX = 272, NCB = 4;
  constexpr int NF = 16 - W;                            // fragments per k-step: column tiles W .. 15
  constexpr int N0 = 16 - W, N1 = W + 1;                // tiles of tile row W / of tile row 15 - W
  constexpr int A1 = 15 - 2 * W;                        // fragment index of tile row 15 - W
  const int b = a.list ? a.list[blockIdx.y] : (int)blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  const bool stp = blockIdx.y == 300 && blockIdx.x == 0; (void)stp;
  const int n = a.n, N = n + 1;
  const int r_lo = PAIR ? 0 : blockIdx.x * a.rows_per_chunk;
  int r_hi = PAIR ? a.m : r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;
  const int boundary = a.rows_per_chunk;                // (PAIR) first row of the second chunk
  const long bsrc = a.src_by_pos ? (long)blockIdx.y : (long)b;   // (the source indexed by list position: CholeskyQR2's W)
  const double* Jb = a.J + bsrc * a.strideJ;
  const double* Fb = a.F + bsrc * a.strideF;

  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];
  double gf[NCB] = {0.0, 0.0, 0.0, 0.0}, gff = 0.0;     // this wave's rows of J^T f / f^T f
  // (PAIR) at the chunk boundary the first chunk's sums are parked in the scratch slot of the problem
  double* scr = PAIR ? a.Gscr + (long)b * 2 * a.NPAD * a.NPAD : nullptr;
  auto issue = [&](int row0, int h) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int row = row0 + W + GR_NW * (HR * h + rr);
      const int rc = row < m ? row : m - 1;
      fpre[rr] = Fb[rc];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        const int cc = col < n ? col : n - 1;
        pre[rr][cb] = __builtin_nontemporal_load(Jb + (long)rc * a.ldJ + cc);
      }
    }
  };
  auto commit = [&](int row0, int h, double* X) {
    if (PAIR && RHS && h == 0 && row0 == boundary) {    // (uniform) the rows of the second chunk start here
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) { scr[W * 256 + 64 * cb + lane] = gf[cb]; gf[cb] = 0.0; }
      if (lane == 0) scr[GR_NW * 256 + W] = gff;
      gff = 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int lrow = W + GR_NW * (HR * h + rr);
      const bool in = row0 + lrow < m;
      const double fv = in ? fpre[rr] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
        if (RHS) gf[cb] = fma(pre[rr][cb], fv, gf[cb]);   // (columns >= n: clamped loads, never stored)
      }
      if (RHS) gff = fma(fv, fv, gff);
      else if (lane == 0) X[lrow * LDX + n] = fv;
    }
  };

  v4d acc0[N0], acc1[N1];
#pragma unroll
  for (int t = 0; t < N0; ++t) acc0[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < N1; ++t) acc1[t] = v4d{0.0, 0.0, 0.0, 0.0};

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {   // padding columns stay zero
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  if (RHS) {                                            // the f column is not staged: keep it defined
    for (int r = tid; r < 2 * GR_RC; r += GR_NT) lds[r * LDX + n] = 0.0;
  }
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();

  double fr[2][NF];
  // one asm per k-step ties every fragment register of a set to the counted wait
  auto wait_set = [&](double (&f)[NF]) {
    if constexpr (NF == 16)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
    else if constexpr (NF == 15)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]), "+v"(f[13]), "+v"(f[14]));
    else if constexpr (NF == 14)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]), "+v"(f[13]));
    else if constexpr (NF == 13)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]));
    else if constexpr (NF == 12)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
    else if constexpr (NF == 11)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]));
    else if constexpr (NF == 10)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]));
  };

  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
  auto tile_ptr0 = [&](int t, int g) { return G + (long)(16 * W + lr + 4 * g) * a.NPAD + 16 * (W + t) + lc; };
  auto tile_ptr1 = [&](int t, int g) { return G + (long)(16 * (15 - W) + lr + 4 * g) * a.NPAD + 16 * (15 - W + t) + lc; };
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    if (PAIR && row0 == boundary) {                     // first chunk done: park its tiles in the output slot
#pragma unroll
      for (int t = 0; t < N0; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *tile_ptr0(t, g) = acc0[t][g];
        acc0[t] = v4d{0.0, 0.0, 0.0, 0.0};
      }
#pragma unroll
      for (int t = 0; t < N1; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *tile_ptr1(t, g) = acc1[t][g];
        acc1[t] = v4d{0.0, 0.0, 0.0, 0.0};
      }
    }
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    const unsigned xb = lds_addr(X) + 8u * (unsigned)(lr * LDX + lc);
    GST(cidx, 0);
    // fragments of the chunk's first k-step (its latency is exposed once per chunk)
    static_for<0, NF>([&](auto ic) {
      constexpr int c = decltype(ic)::value;
      lds_read64_off<8 * 16 * (W + c)>(fr[0][c], xb);
    });
    static_for<0, 8>([&](auto is) {
      constexpr int s = decltype(is)::value;            // k-step of the chunk
      constexpr int cur = s & 1, nxt = cur ^ 1;
      if constexpr (s == 0 || s == 4) {
        if (more) issue(row0 + GR_RC, s / 4);
      }
      wait_set(fr[cur]);
      // 17 MFMAs; the next k-step's fragments are requested two per MFMA behind the first ones
      static_for<0, N0 + N1>([&](auto it) {
        constexpr int t = decltype(it)::value;
        if constexpr (t < N0) acc0[t] = gmfma(fr[cur][0], fr[cur][t], acc0[t]);
        else acc1[t - N0] = gmfma(fr[cur][A1], fr[cur][A1 + (t - N0)], acc1[t - N0]);
        if constexpr (s < 7) {
          if constexpr (2 * t < NF)
            lds_read64_off<8 * (4 * (s + 1) * LDX + 16 * (W + 2 * t))>(fr[nxt][2 * t], xb);
          if constexpr (2 * t + 1 < NF)
            lds_read64_off<8 * (4 * (s + 1) * LDX + 16 * (W + 2 * t + 1))>(fr[nxt][2 * t + 1], xb);
        }
      });
      if constexpr (s == 3) GST(cidx, 1);
      if constexpr (s == 3 || s == 7) {
        if (more) commit(row0 + GR_RC, s / 4, Xn);
      }
    });
    GST(cidx, 2);
    __syncthreads();
    GST(cidx, 3);
  }

  const bool two = PAIR && m > boundary;                // (a second chunk was accumulated)
  if (two) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this lane's own stores of the first chunk)
#pragma unroll
  for (int t = 0; t < N0; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double* q = tile_ptr0(t, g);
      *q = two ? (0.0 + *q) + acc0[t][g] : acc0[t][g];
    }
#pragma unroll
  for (int t = 0; t < N1; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double* q = tile_ptr1(t, g);
      *q = two ? (0.0 + *q) + acc1[t][g] : acc1[t][g];
    }
  if (!RHS) return;
  // rhs column: the eight per-wave partials, added in wave order (the last barrier is behind us);
  // PAIR: the two chunks' columns one after the other, then (0 + first) + second
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) lds[W * 256 + 64 * cb + lane] = two ? scr[W * 256 + 64 * cb + lane] : gf[cb];
  if (lane == 0) lds[GR_NW * 256 + W] = two ? scr[GR_NW * 256 + W] : gff;
  __syncthreads();
  double sum = 0.0, sumf = 0.0;
  if (tid < n) {
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[ww * 256 + tid];
  }
  if (tid == 0) {
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sumf += lds[GR_NW * 256 + ww];
  }
  if (two) {
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) lds[W * 256 + 64 * cb + lane] = gf[cb];
    if (lane == 0) lds[GR_NW * 256 + W] = gff;
    __syncthreads();
    double s2 = 0.0, s2f = 0.0;
    if (tid < n) {
#pragma unroll
      for (int ww = 0; ww < GR_NW; ++ww) s2 += lds[ww * 256 + tid];
    }
    if (tid == 0) {
#pragma unroll
      for (int ww = 0; ww < GR_NW; ++ww) s2f += lds[GR_NW * 256 + ww];
    }
    sum = (0.0 + sum) + s2; sumf = (0.0 + sumf) + s2f;
  }
  if (tid < n) G[(long)tid * a.NPAD + n] = sum;
  if (tid == 0) G[(long)n * a.NPAD + n] = sumf;
  const int NT = (N + 15) / 16;
  for (int e = tid; e < a.NPAD * (16 * NT - N); e += GR_NT) {   // padding columns of the rhs tile column
    const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

template <bool RHS, bool PAIR>
__global__ __launch_bounds__(GR_NT, 2) void gram16_kernel(GramArgs a) {
  extern __shared__ double lds[];
  if (a.mask && a.mask[a.list ? a.list[blockIdx.y] : (int)blockIdx.y] <= 1) return;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  switch (w) {
    case 0: gram16_wave<0, RHS, PAIR>(a, lds); break;
    case 1: gram16_wave<1, RHS, PAIR>(a, lds); break;
    case 2: gram16_wave<2, RHS, PAIR>(a, lds); break;
    case 3: gram16_wave<3, RHS, PAIR>(a, lds); break;
    case 4: gram16_wave<4, RHS, PAIR>(a, lds); break;
    case 5: gram16_wave<5, RHS, PAIR>(a, lds); break;
    case 6: gram16_wave<6, RHS, PAIR>(a, lds); break;
    default: gram16_wave<7, RHS, PAIR>(a, lds); break;
  }
}

// ---- 8 column tiles (n = 113 .. 128): static tile rows per wave, k-steps split over two groups ----
// The scheme of gram16_kernel for half the width: the 36 upper tiles of an 8 x 8 tile grid are four
// pairs of tile rows (P, 7 - P) of 9 tiles each — four waves' worth.  The other four waves take the
// same tiles for the OTHER k-steps: wave W = P + 4 K works on k-steps s with s % 2 == K (K = 0, 1), so
// every wave has 9 MFMAs per k-step it owns and 8 - P operand fragments to fetch for them (the
// generic kernel runs these widths as 8 slots per wave and k-step with two reads per MFMA and idle
// slots recomputing a tile).  A tile is therefore the sum of two partial accumulations — even and odd
// k-steps — added in that order at the end (through LDS, once per row chunk).  This order is the
// DEFINITION of the result for these widths: the kernel serves every batch size (no tile groups), so
// a problem's bits still do not depend on its batch.  Row chunks, staging and the rhs column are
// those of the generic kernel.
template <int W, bool RHS>
__device__ __forceinline__ void gram8_wave(const GramArgs& a, double* lds) {
  constexpr int NTJ = 8, NCB = 2;
  constexpr int P = W % 4, K = W / 4;
  constexpr int NF = NTJ - P;                           // fragments per k-step: column tiles P .. 7
  constexpr int N0 = NTJ - P, N1 = P + 1;               // tiles of tile row P / of tile row 7 - P
  constexpr int A1 = NTJ - 1 - 2 * P;                   // fragment index of tile row 7 - P
  const int b = a.list ? a.list[blockIdx.y] : (int)blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, N = n + 1;
  const int NT = (N + 15) / 16;                        // 8 or 9
  constexpr int LDX = 144;                              // gram_ldx(8) == gram_ldx(9)
  const int r_lo = blockIdx.x * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;
  const long bsrc = a.src_by_pos ? (long)blockIdx.y : (long)b;   // (the source indexed by list position: CholeskyQR2's W)
  const double* Jb = a.J + bsrc * a.strideJ;
  const double* Fb = a.F + bsrc * a.strideF;

  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];
  double gf[NCB] = {0.0, 0.0}, gff = 0.0;
  auto issue = [&](int row0, int h) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int row = row0 + W + GR_NW * (HR * h + rr);
      const int rc = row < m ? row : m - 1;
      fpre[rr] = Fb[rc];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        const int cc = col < n ? col : n - 1;
        pre[rr][cb] = __builtin_nontemporal_load(Jb + (long)rc * a.ldJ + cc);
      }
    }
  };
  auto commit = [&](int row0, int h, double* X) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int lrow = W + GR_NW * (HR * h + rr);
      const bool in = row0 + lrow < m;
      const double fv = in ? fpre[rr] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
        if (RHS) gf[cb] = fma(pre[rr][cb], fv, gf[cb]);
      }
      if (RHS) gff = fma(fv, fv, gff);
      else if (lane == 0) X[lrow * LDX + n] = fv;
    }
  };

  v4d acc0[N0], acc1[N1];
#pragma unroll
  for (int t = 0; t < N0; ++t) acc0[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < N1; ++t) acc1[t] = v4d{0.0, 0.0, 0.0, 0.0};

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {   // padding columns stay zero
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  if (RHS) {
    for (int r = tid; r < 2 * GR_RC; r += GR_NT) lds[r * LDX + n] = 0.0;
  }
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();

  double fr[2][NF];
  auto wait_set = [&](double (&f)[NF]) {
    if constexpr (NF == 8)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    else if constexpr (NF == 7)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]));
    else if constexpr (NF == 6)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]));
  };
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    const unsigned xb = lds_addr(X) + 8u * (unsigned)(lr * LDX + lc);
    static_for<0, NF>([&](auto ic) {
      constexpr int c = decltype(ic)::value;
      lds_read64_off<8 * (4 * K * LDX + 16 * (P + c))>(fr[0][c], xb);
    });
    static_for<0, 4>([&](auto iq) {
      constexpr int q = decltype(iq)::value;            // this wave's q-th k-step of the chunk: s = 2 q + K
      constexpr int cur = q & 1, nxt = cur ^ 1;
      if constexpr (q == 0 || q == 2) {
        if (more) issue(row0 + GR_RC, q / 2);
      }
      wait_set(fr[cur]);
      static_for<0, N0 + N1>([&](auto it) {
        constexpr int t = decltype(it)::value;
        if constexpr (t < N0) acc0[t] = gmfma(fr[cur][0], fr[cur][t], acc0[t]);
        else acc1[t - N0] = gmfma(fr[cur][A1], fr[cur][A1 + (t - N0)], acc1[t - N0]);
        if constexpr (q < 3 && t < NF)
          lds_read64_off<8 * (4 * (2 * (q + 1) + K) * LDX + 16 * (P + t))>(fr[nxt][t], xb);
      });
      if constexpr (q == 1 || q == 3) {
        if (more) commit(row0 + GR_RC, q / 2, Xn);
      }
    });
    __syncthreads();
  }

  // the odd k-steps' partial tiles -> LDS, the even group adds them (even first) and stores
  double* part = lds + (size_t)P * (NTJ + 1) * 256;     // [pair][tile][256]
  if (K == 1) {
#pragma unroll
    for (int t = 0; t < N0; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) part[t * 256 + g * 64 + lane] = acc0[t][g];
#pragma unroll
    for (int t = 0; t < N1; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) part[(N0 + t) * 256 + g * 64 + lane] = acc1[t][g];
  }
  __syncthreads();
  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
  if (K == 0) {
#pragma unroll
    for (int t = 0; t < N0; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        G[(long)(16 * P + lr + 4 * g) * a.NPAD + 16 * (P + t) + lc] = acc0[t][g] + part[t * 256 + g * 64 + lane];
#pragma unroll
    for (int t = 0; t < N1; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        G[(long)(16 * (NTJ - 1 - P) + lr + 4 * g) * a.NPAD + 16 * (NTJ - 1 - P + t) + lc] =
            acc1[t][g] + part[(N0 + t) * 256 + g * 64 + lane];
  }
  if (!RHS) return;
  __syncthreads();                                      // (the partial tiles have been consumed)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) lds[W * 128 + 64 * cb + lane] = gf[cb];
  if (lane == 0) lds[GR_NW * 128 + W] = gff;
  __syncthreads();
  if (tid < n) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[ww * 128 + tid];
    G[(long)tid * a.NPAD + n] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[GR_NW * 128 + ww];
    G[(long)n * a.NPAD + n] = sum;
  }
  for (int e = tid; e < a.NPAD * (16 * NT - N); e += GR_NT) {   // padding columns of the rhs tile column
    const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

template <bool RHS>
__global__ __launch_bounds__(GR_NT, 2) void gram8_kernel(GramArgs a) {
  extern __shared__ double lds[];
  if (a.mask && a.mask[a.list ? a.list[blockIdx.y] : (int)blockIdx.y] <= 1) return;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  switch (w) {
    case 0: gram8_wave<0, RHS>(a, lds); break;
    case 1: gram8_wave<1, RHS>(a, lds); break;
    case 2: gram8_wave<2, RHS>(a, lds); break;
    case 3: gram8_wave<3, RHS>(a, lds); break;
    case 4: gram8_wave<4, RHS>(a, lds); break;
    case 5: gram8_wave<5, RHS>(a, lds); break;
    case 6: gram8_wave<6, RHS>(a, lds); break;
    default: gram8_wave<7, RHS>(a, lds); break;
  }
}

// ---- narrow problems (NT <= 5 column tiles): no LDS staging ------------------------------------
// The MFMA operand fragment of column tile c at k-step s is X[4 s + lr][16 c + lc]: 16 lanes read
// 128 contiguous bytes of a row — a coalesced global load straight into the operand register.
// So with few column tiles every wave takes its own k-steps (4 rows each, wave w of NWD: k-steps w,
// w + NWD, ...) for ALL output tiles, keeps two rounds of four k-steps of loads in flight (the loads of round
// i + 1 are issued before round i is consumed: 512 x 64, 1024 problems 181 -> 95 us), and never meets
// the other waves until the final, fixed-order reduction of the NWD partial Grams through LDS.
// (NWD = 2, 4 or 8 waves per workgroup by the row count: launch_gram.)
template <int NTT, bool RHS, int NWD>
__global__ __launch_bounds__(64 * NWD, (NTT <= 2 ? 4 : 2)) void gram_direct_kernel(GramArgs a) {
  constexpr int NTD = 64 * NWD;                         // NWD waves per workgroup
  constexpr int NTILE = NTT * (NTT + 1) / 2;
  constexpr int KU = 4;                                 // k-steps per round (two rounds in flight per wave)
  extern __shared__ double lds[];                       // [NWD][256]
  const int b = a.list ? a.list[blockIdx.y] : (int)blockIdx.y;
  if (a.mask && a.mask[b] <= 1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n;
  const int r_lo = blockIdx.x * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const long bsrc = a.src_by_pos ? (long)blockIdx.y : (long)b;   // (the source indexed by list position: CholeskyQR2's W)
  const double* Jb = a.J + bsrc * a.strideJ;
  const double* Fb = a.F + bsrc * a.strideF;
  v4d acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  // RHS (n = 16 NTT exactly): the tiles cover J^T J only; the rhs column (J^T f, f^T f) is accumulated
  // from the operand fragments a lane holds anyway (column 16 c + lc, rows lr mod 4 of this wave's
  // k-steps) and reduced over the lane rows and the waves at the end, in a fixed order
  double gf[NTT], gff = 0.0;
#pragma unroll
  for (int c = 0; c < NTT; ++c) gf[c] = 0.0;
  const int last = r_hi > r_lo ? r_hi - 1 : r_lo;
  constexpr int RSTEP = 4 * NWD * KU;                 // rows the workgroup consumes per round
  // raw loads of one round (clamped, unconditional — nothing here waits for the data) ...
  auto load_round = [&](int r0, double (&fr)[KU][NTT], double (&fv)[KU]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int row = r0 + 4 * NWD * u + lr;
      const int rc = row < r_hi ? row : last;
      if (RHS) fv[u] = Fb[rc];
#pragma unroll
      for (int c = 0; c < NTT; ++c) {
        const int col = 16 * c + lc;
        // (without RHS the rhs f is column n)
        const double* ptr = (RHS || col < n) ? Jb + (long)rc * a.ldJ + (col < n ? col : n - 1) : Fb + rc;
        fr[u][c] = __builtin_nontemporal_load(ptr);
      }
    }
  };
  // ... and their use: rows / columns outside the problem count as zeros
  auto use_round = [&](int r0, double (&fr)[KU][NTT], double (&fv)[KU]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const bool rin = r0 + 4 * NWD * u + lr < r_hi;
      if (RHS) fv[u] = rin ? fv[u] : 0.0;
#pragma unroll
      for (int c = 0; c < NTT; ++c) {
        const int col = 16 * c + lc;
        fr[u][c] = (rin && (RHS ? col < n : col <= n)) ? fr[u][c] : 0.0;
      }
      int t = 0;
#pragma unroll
      for (int i = 0; i < NTT; ++i)
#pragma unroll
        for (int j = i; j < NTT; ++j, ++t) acc[t] = gmfma(fr[u][i], fr[u][j], acc[t]);
      if (RHS) {
#pragma unroll
        for (int c = 0; c < NTT; ++c) gf[c] = fma(fr[u][c], fv[u], gf[c]);
        gff = fma(fv[u], fv[u], gff);
      }
    }
  };
  // two rounds in flight: the loads of round i + 1 are issued before round i is consumed
  int r_start = r_lo + 4 * w;
  if constexpr (RHS) {
    // FULL rounds (every row inside the chunk; n = 16 NTT: every column inside the matrix) — no clamps, no masks, one
    // pointer per k-step of the round advanced by a scalar, the column tiles and f as immediate offsets; and the next
    // round's requests UNCONDITIONAL (after the last full round they repeat it: the step is zero): behind a branch the
    // compiler's wait counts at the join assumed the path without new requests and the fourth k-step of every round
    // drained the next round's loads with vmcnt(0) — one round in flight instead of two.  The same sums as below.
    const int span = 4 * NWD * (KU - 1) + 4;              // rows from a round's first to its last, + 1
    const int nfull = r_hi - span >= r_start ? (r_hi - span - r_start) / RSTEP + 1 : 0;
    if (nfull > 0) {
      const double* pj[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) pj[u] = Jb + (long)(r_start + 4 * NWD * u + lr) * a.ldJ + lc;
      const double* pf = Fb + r_start + lr;
      const long jstep = (long)RSTEP * a.ldJ;
      auto load_full = [&](double (&fr)[KU][NTT], double (&fv)[KU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          fv[u] = pf[4 * NWD * u];
#pragma unroll
          for (int c = 0; c < NTT; ++c) fr[u][c] = __builtin_nontemporal_load(pj[u] + 16 * c);
        }
      };
      auto advance = [&](bool more) __attribute__((always_inline)) {
        const long dj = more ? jstep : 0;
        const int df_ = more ? RSTEP : 0;
#pragma unroll
        for (int u = 0; u < KU; ++u) pj[u] += dj;
        pf += df_;
      };
      auto use_full = [&](double (&fr)[KU][NTT], double (&fv)[KU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          int t = 0;
#pragma unroll
          for (int i = 0; i < NTT; ++i)
#pragma unroll
            for (int j = i; j < NTT; ++j, ++t) acc[t] = gmfma(fr[u][i], fr[u][j], acc[t]);
#pragma unroll
          for (int c = 0; c < NTT; ++c) gf[c] = fma(fr[u][c], fv[u], gf[c]);
          gff = fma(fv[u], fv[u], gff);
        }
      };
      double frA[KU][NTT], fvA[KU], frB[KU][NTT], fvB[KU];
      load_full(frA, fvA);
      for (int k = 0; k < nfull; k += 2) {
        advance(k + 1 < nfull);
        load_full(frB, fvB);
        __builtin_amdgcn_sched_barrier(0);
        use_full(frA, fvA);
        if (k + 1 >= nfull) break;
        advance(k + 2 < nfull);
        load_full(frA, fvA);
        __builtin_amdgcn_sched_barrier(0);
        use_full(frB, fvB);
      }
      r_start += nfull * RSTEP;
    }
  }
  {
    double frA[KU][NTT], fvA[KU], frB[KU][NTT], fvB[KU];
    int r0 = r_start;
    if (r0 < r_hi) load_round(r0, frA, fvA);
    while (r0 < r_hi) {
      if (r0 + RSTEP < r_hi) load_round(r0 + RSTEP, frB, fvB);
      use_round(r0, frA, fvA);
      r0 += RSTEP;
      if (r0 >= r_hi) break;
      if (r0 + RSTEP < r_hi) load_round(r0 + RSTEP, frA, fvA);
      use_round(r0, frB, fvB);
      r0 += RSTEP;
    }
  }
  // cross-wave reduction through LDS (fixed order: deterministic)
  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
  const int NT = (n + 1 + 15) / 16;
  // (PH tiles per pass through LDS — [tile][wave][256] — and two barriers per pass)
  constexpr int PH = NTILE < 5 ? NTILE : 5;
#pragma unroll
  for (int t0 = 0; t0 < NTILE; t0 += PH) {
#pragma unroll
    for (int tt = 0; tt < PH; ++tt) {
      if (t0 + tt < NTILE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) lds[(tt * NWD + w) * 256 + g * 64 + lane] = acc[t0 + tt][g];
      }
    }
    __syncthreads();
    for (int e = tid; e < PH * 256; e += NTD) {
      const int tt = e >> 8, el = e & 255;
      int q = t0 + tt, i = 0;
      if (q < NTILE) {
        while (q >= NTT - i) { q -= NTT - i; ++i; }
        const int j = i + q;
        if (j < NT) {
          double sum = 0.0;
#pragma unroll
          for (int ww = 0; ww < NWD; ++ww) sum += lds[(tt * NWD + ww) * 256 + el];
          const int g = el >> 6, ln = el & 63;
          G[(long)(16 * i + (ln >> 4) + 4 * g) * a.NPAD + 16 * j + (ln & 15)] = sum;
        }
      }
    }
    __syncthreads();
  }
  if (!RHS) return;
  // rhs column: [wave][lane row][column] partials -> column totals (wave-major, then lane row)
#pragma unroll
  for (int c = 0; c < NTT; ++c) lds[(w * 4 + lr) * 64 + 16 * c + lc] = gf[c];
  if (lc == 0) lds[NWD * 256 + w * 4 + lr] = gff;
  __syncthreads();
  if (tid < n) {
    double sum = 0.0;
    for (int q = 0; q < 4 * NWD; ++q) sum += lds[q * 64 + tid];
    G[(long)tid * a.NPAD + n] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
    for (int q = 0; q < 4 * NWD; ++q) sum += lds[NWD * 256 + q];
    G[(long)n * a.NPAD + n] = sum;
  }
  for (int e = tid; e < a.NPAD * (16 * NT - (n + 1)); e += NTD) {   // padding of the rhs tile column
    const int r = e / (16 * NT - (n + 1)), c = n + 1 + e % (16 * NT - (n + 1));
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

// ---- a handful of problems (chunks x B <= 4), n a multiple of 16: one tile per wave, no LDS staging ---------------
// One 4096 x 256 problem gives gram_kernel two row chunks: even split over 32 tile groups that is 64 workgroups whose
// waves each run a chain of 512 k-steps through the LDS staging loop at ~800 cycles a step (0.17 ms, whatever the number
// of groups).  Here a wave owns ONE output tile (i, j) of one row chunk and takes its two operand fragments
// X[4 s + lr][16 i + lc], X[4 s + lr][16 j + lc] straight from global memory (L2 / MALL: every fragment is read by 16
// tile jobs), BLSQ_G1_DEPTH k-steps ahead in registers; the rhs column is the job of four more workgroups per chunk whose
// waves reproduce the generic kernel's sums — "wave" v adds the rows v, v + 8, ... of the chunk in order, the eight
// partials are added in wave order.  A tile sees the same k-steps in the same order with the same instruction: the
// result equals gram_kernel's bit for bit (tests/test_gram_gpu.py).
// Measured, ONE problem of 4096 x 256 (tile groups through LDS: 171 us): 146 us with a branch per k-step in the loop,
// 75 without, 61 with the row chunks dealt to the XCDs; by the depth of the prefetch 4 / 6 / 8 / 10 / 12 / 16 / 24 / 30
// k-steps: 62 / 50 / 49 / 45 / 43 / 42 / 61 / 68 us (more requests in flight than the fabric takes queue up).
#ifndef BLSQ_G1_DEPTH
#define BLSQ_G1_DEPTH 16
#endif
static constexpr int G1_NW = 4, G1_NT = 64 * G1_NW;
__global__ __launch_bounds__(G1_NT) void gram1_kernel(GramArgs a, int chunks, int nprob, int tile_wgs) {
  constexpr int D = BLSQ_G1_DEPTH;
  __shared__ double red[2 * G1_NW * 64 + 2 * G1_NW];
  // Workgroups go to the eight XCDs round robin: the (row chunk, problem) pairs are dealt to the XCDs, so that an XCD's
  // L2 sees the rows of ONE chunk (its waves walk them at the same pace: every line is fetched once per XCD).
  const int groups = chunks * nprob;
  const int xcd = (int)blockIdx.x % 8, slot = (int)blockIdx.x / 8;
  const int group = xcd % groups, mg = (8 - group + groups - 1) / groups;
  const int job = slot * mg + xcd / groups;
  if (job >= tile_wgs + (a.n + 63) / 64) return;
  const int pz = group / chunks;
  const int b = a.list ? a.list[pz] : pz;
  if (a.mask && a.mask[b] <= 1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, NTJ = n / 16;
  const int chunk = group % chunks;
  const int r_lo = chunk * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const long bsrc = a.src_by_pos ? (long)pz : (long)b;
  const double* Jb = a.J + bsrc * a.strideJ;
  const double* Fb = a.F + bsrc * a.strideF;
  const long ldJ = a.ldJ;
  double* G = a.G + ((long)b * chunks + chunk) * (long)a.NPAD * a.NPAD;
  if (job < tile_wgs) {
    // ---- tile job: workgroup `job` owns four consecutive tiles (i, j0 .. j0 + 3) of ONE tile row — the four waves share the
    // fragment of column tile i (one of them fetches a line, the others find it in the L1) and their own four are
    // neighbours: rows 4 s .. 4 s + 3 are read as 512 contiguous bytes each
    int i = 0, rem = job;
    while (i < NTJ && rem >= (NTJ - i + G1_NW - 1) / G1_NW) { rem -= (NTJ - i + G1_NW - 1) / G1_NW; ++i; }
    const int j = i + G1_NW * rem + w;
    if (i >= NTJ || j >= NTJ) return;
    const int nk = (r_hi - r_lo + 3) / 4;
    const double* pa = Jb + 16 * i + lc;
    const double* pb = Jb + 16 * j + lc;
    const int last = r_hi - 1;
    auto rowof = [&](int s_) { const int r = r_lo + 4 * s_ + lr; return r < r_hi ? r : last; };
    double fa[D], fb[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const long ro = (long)rowof(u < nk ? u : 0) * ldJ;
      fa[u] = pa[ro];
      fb[u] = pb[ro];
    }
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < nk; s0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        // (no branch in here: the compiler counts the outstanding loads only along straight-line code — with a branch
        //  per k-step it waited for ALL of them at every join, one memory round trip per k-step: 0.146 ms.  A k-step
        //  beyond the chunk multiplies zeros, its request repeats the last row.)
        const int s_ = s0 + u;
        const bool in = r_lo + 4 * s_ + lr < r_hi;        // (rows beyond the chunk count as zeros)
        const double av = in ? fa[u] : 0.0;
        const double bv = in ? fb[u] : 0.0;
        const int sn = s_ + D < nk ? s_ + D : nk - 1;
        const long ro = (long)rowof(sn) * ldJ;
        fa[u] = pa[ro];
        fb[u] = pb[ro];
        acc = gmfma(av, bv, acc);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) G[(long)(16 * i + lr + 4 * g) * a.NPAD + 16 * j + lc] = acc[g];
    return;
  }
  // ---- rhs job: columns 64 cb + lane; wave w plays the generic kernel's waves w and w + 4 (rows v, v + 8, ... in order)
  const int cb = job - tile_wgs;
  const int col = 64 * cb + lane;
  const int cc = col < n ? col : n - 1;
  double gf0 = 0.0, gf1 = 0.0, gff0 = 0.0, gff1 = 0.0;
  {
    const double* pc = Jb + cc;
    constexpr int U = 8;                                  // rows of each of the two sums in flight
    for (int r0 = r_lo; r0 < r_hi; r0 += 8 * U) {
      double jv0[U], jv1[U], fv0[U], fv1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ra = r0 + 8 * u + w, rb = ra + G1_NW;
        const int rac = ra < r_hi ? ra : r_hi - 1, rbc = rb < r_hi ? rb : r_hi - 1;
        jv0[u] = pc[(long)rac * ldJ]; fv0[u] = Fb[rac];
        jv1[u] = pc[(long)rbc * ldJ]; fv1[u] = Fb[rbc];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ra = r0 + 8 * u + w, rb = ra + G1_NW;
        // (the generic kernel stages whole 32-row blocks: a row beyond the chunk enters its sums with f = 0)
        const double f0 = ra < r_hi ? fv0[u] : 0.0, f1 = rb < r_hi ? fv1[u] : 0.0;
        const bool live0 = ra < r_lo + ((r_hi - r_lo + GR_RC - 1) / GR_RC) * GR_RC;
        const bool live1 = rb < r_lo + ((r_hi - r_lo + GR_RC - 1) / GR_RC) * GR_RC;
        if (live0) { gf0 = fma(jv0[u], f0, gf0); gff0 = fma(f0, f0, gff0); }
        if (live1) { gf1 = fma(jv1[u], f1, gf1); gff1 = fma(f1, f1, gff1); }
      }
    }
  }
  red[w * 64 + lane] = gf0;
  red[(w + G1_NW) * 64 + lane] = gf1;
  if (lane == 0) { red[2 * G1_NW * 64 + w] = gff0; red[2 * G1_NW * 64 + G1_NW + w] = gff1; }
  __syncthreads();
  if (w == 0) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < 2 * G1_NW; ++ww) sum += red[ww * 64 + lane];
    if (col < n) G[(long)col * a.NPAD + n] = sum;
  }
  if (cb == 0) {
    if (tid == 0) {
      double sum = 0.0;
#pragma unroll
      for (int ww = 0; ww < 2 * G1_NW; ++ww) sum += red[2 * G1_NW * 64 + ww];
      G[(long)n * a.NPAD + n] = sum;
    }
    // the rest of the rhs tile column of the slot: padding columns (n, 16 NT) stay zero
    const int N = n + 1, NT = (N + 15) / 16;
    for (int e = tid; e < a.NPAD * (16 * NT - N); e += G1_NT) {
      const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
      if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
    }
  }
}

// partial Grams of the row chunks -> the triangle slot (fixed order: deterministic)
__global__ void gram_reduce_kernel(const double* Gpart, int chunks, int NPAD, double* Gout,
                                   const int* mask) {
  const int b = blockIdx.y;
  if (mask && mask[b] <= 1) return;
  const long sz = (long)NPAD * NPAD;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= sz) return;
  const int r = (int)(idx / NPAD), c = (int)(idx % NPAD);
  if ((r >> 4) > (c >> 4)) return;                      // lower tiles are never written
  const double* src = Gpart + (long)b * chunks * sz + idx;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += src[(long)k * sz];
  Gout[(long)b * sz + idx] = s;
}
// Many row chunks (one very tall problem): eight thread groups sum eight contiguous ranges of the
// chunks, the eight partial sums are added in range order.  (The order depends on the chunk count
// only, i.e. on m — never on the batch.)
constexpr int GRED_G = 8, GRED_W = 32;
__global__ __launch_bounds__(GRED_G * GRED_W) void gram_reduce_wide_kernel(
    const double* Gpart, int chunks, int NPAD, double* Gout, const int* mask) {
  __shared__ double part[GRED_G][GRED_W];
  const int b = blockIdx.y;
  if (mask && mask[b] <= 1) return;
  const long sz = (long)NPAD * NPAD;
  const int e = threadIdx.x % GRED_W, g = threadIdx.x / GRED_W;
  const long idx = (long)blockIdx.x * GRED_W + e;
  const int r = (int)(idx / NPAD), c = (int)(idx % NPAD);
  const bool live = idx < sz && (r >> 4) <= (c >> 4);
  const int per = (chunks + GRED_G - 1) / GRED_G;
  const int k0 = g * per, k1 = min(chunks, k0 + per);
  double s0 = 0.0;
  if (live) {
    const double* src = Gpart + (long)b * chunks * sz + idx;
    for (int k = k0; k < k1; ++k) s0 += __builtin_nontemporal_load(src + (long)k * sz);
  }
  part[g][e] = s0;
  __syncthreads();
  if (g == 0 && live) {
    double s = part[0][e];
#pragma unroll
    for (int q = 1; q < GRED_G; ++q) s += part[q][e];
    Gout[(long)b * sz + idx] = s;
  }
}

static int gram_chunk_rows(int m) { return m > 131072 ? 1024 : 2048; }
int gram_chunks(int B, int m) {
  (void)B;
  const int r = gram_chunk_rows(m);
  const int c = (m + r - 1) / r;
  return c < 1 ? 1 : c;
}

hipError_t launch_gram(const GramArgs& a_in, int chunks, int B, hipStream_t s, double* Gfinal, bool* fused) {
  GramArgs a = a_in;
  if (fused) *fused = false;
  const int NT = (a.n + 1 + 15) / 16;
  a.rows_per_chunk = chunks > 1 ? gram_chunk_rows(a.m) : a.m;
  const size_t lds = sizeof(double) * 2 * GR_RC * (size_t)gram_ldx(NT);
  // slot variant a wave needs for `nt` column tiles
  auto slots_for = [](int nt) {
    const int per_ = (nt * (nt + 1) / 2 + GR_NW - 1) / GR_NW;
    return per_ <= 4 ? 4 : per_ <= 8 ? 8 : per_ <= 12 ? 12 : per_ <= 17 ? 17 : 20;
  };
  // n % 16 == 0: the rhs column would be a tile column of its own (NT tiles for one useful column
  // each) — it is accumulated in registers beside the MFMA stream instead (n = 256: 153 -> 136 tiles)
  a.rhs_valu = ((a.n + 15) / 16 < NT) ? 1 : 0;
  (void)slots_for;
  const int NTJ = a.rhs_valu ? (a.n + 15) / 16 : NT;
  const int ntile = NTJ * (NTJ + 1) / 2;
#define BLSQ_GRAM_LAUNCH(SL, CB)                                                              \
  do {                                                                                        \
    static std::atomic<size_t> granted[64];                                                   \
    hipError_t ge = gram_grant_lds(gram_kernel<SL, CB>, lds, granted);                        \
    if (ge != hipSuccess) return ge;                                                          \
    hipLaunchKernelGGL((gram_kernel<SL, CB>), dim3(chunks, B, tg), dim3(GR_NT), lds, s, a);   \
  } while (0)
#define BLSQ_GRAM_DIRECT_NW(NTT, RHS, NWD)                                                    \
  do {                                                                                        \
    constexpr int nt_ = (NTT) * ((NTT) + 1) / 2;                                              \
    const size_t dl_ = sizeof(double) * ((nt_ < 5 ? nt_ : 5) * (NWD) * 256 + 64);            \
    static std::atomic<size_t> granted[64];                                                   \
    hipError_t ge = gram_grant_lds(gram_direct_kernel<NTT, RHS, NWD>, dl_, granted);          \
    if (ge != hipSuccess) return ge;                                                          \
    hipLaunchKernelGGL((gram_direct_kernel<NTT, RHS, NWD>), dim3(chunks, B), dim3(64 * (NWD)), dl_, s, a); \
    return hipGetLastError();                                                                 \
  } while (0)
#define BLSQ_GRAM_DIRECT(NTT, RHS)                                                            \
  do {                                                                                        \
    if (direct_nw == 4) BLSQ_GRAM_DIRECT_NW(NTT, RHS, 4);                                     \
    else if (direct_nw == 2) BLSQ_GRAM_DIRECT_NW(NTT, RHS, 2);                                \
    else BLSQ_GRAM_DIRECT_NW(NTT, RHS, 8);                                                    \
  } while (0)
  // waves per workgroup of the direct kernel: about 256 rows per wave (a function of m only, so a
  // problem's bits do not depend on its batch).  A short problem on eight waves is four rounds of loads
  // per wave between a cold start and an eight-way reduction, and one workgroup fills the CU: 1024
  // problems of 512 x 64 pass in four lock-step generations.  On two waves, four workgroups share the CU
  // and drift apart: 96 -> 78 us.  BLSQ_GRAM_DIRECT_NW = 2 | 4 | 8 forces one.
  const Options& opt = options_or_default(a.opt);
  const int dnw_o = opt.i(OPT_GRAM_DIRECT_NW);
  const int rows_wg = a.m < a.rows_per_chunk ? a.m : a.rows_per_chunk;
  // (up to 128 rows eight waves need a single round of loads: nothing to drift, lowest latency)
  const int direct_nw = dnw_o > 0 ? dnw_o : (rows_wg <= 128 ? 8 : rows_wg <= 512 ? 2 : rows_wg <= 1024 ? 4 : 8);
  {
    const int dmax = opt.i(OPT_GRAM_DIRECT_MAX_NT);            // tuning / tests: 0 disables; measured: direct wins up to 4 column tiles
    // n a multiple of 16: the tiles cover J^T J only (n / 16 column tiles), the rhs column is
    // accumulated from the same fragments (n = 64: 10 tiles instead of 15, and still no LDS staging)
    if (a.rhs_valu && NTJ <= dmax) {
      if (NTJ == 1) BLSQ_GRAM_DIRECT(1, true);
      else if (NTJ == 2) BLSQ_GRAM_DIRECT(2, true);
      else if (NTJ == 3) BLSQ_GRAM_DIRECT(3, true);
      else if (NTJ == 4) BLSQ_GRAM_DIRECT(4, true);
    }
    if (!a.rhs_valu && NT <= dmax) {
      if (NT <= 1) BLSQ_GRAM_DIRECT(1, false);
      else if (NT == 2) BLSQ_GRAM_DIRECT(2, false);
      else if (NT == 3) BLSQ_GRAM_DIRECT(3, false);
      else if (NT == 4) BLSQ_GRAM_DIRECT(4, false);
    }
  }
#undef BLSQ_GRAM_DIRECT
#undef BLSQ_GRAM_DIRECT_NW
  {
    // 8 column tiles of J^T J (n = 113 .. 128): the k-split static-tile kernel, for EVERY batch size
    // (its summation order defines the result for these widths).  BLSQ_GRAM8 = 0: the generic kernel.
    if (NTJ == 8 && opt.on(OPT_GRAM8)) {
      const size_t need = sizeof(double) * 4 * 9 * 256;          // partial tiles of the odd k-steps
      const size_t l8 = lds > need ? lds : need;
      if (a.rhs_valu) {
        static std::atomic<size_t> granted[64];
        hipError_t ge = gram_grant_lds(gram8_kernel<true>, l8, granted);
        if (ge != hipSuccess) return ge;
        hipLaunchKernelGGL((gram8_kernel<true>), dim3(chunks, B, 1), dim3(GR_NT), l8, s, a);
      } else {
        static std::atomic<size_t> granted[64];
        hipError_t ge = gram_grant_lds(gram8_kernel<false>, l8, granted);
        if (ge != hipSuccess) return ge;
        hipLaunchKernelGGL((gram8_kernel<false>), dim3(chunks, B, 1), dim3(GR_NT), l8, s, a);
      }
      return hipGetLastError();
    }
  }
  // a handful of problems, rhs column outside the tiles, widths whose sums the tile-table kernel defines: one tile per wave
  // straight from global memory (bit-identical; option gram1 = 0: the tile groups below)
  if (a.rhs_valu && (long)chunks * B <= 4 && NTJ > opt.i(OPT_GRAM_DIRECT_MAX_NT) && NTJ > 4 && !(NTJ == 8 && opt.on(OPT_GRAM8)) &&
      NTJ <= 16 && opt.on(OPT_GRAM1)) {
    int tile_wgs = 0;
    for (int i = 0; i < NTJ; ++i) tile_wgs += (NTJ - i + G1_NW - 1) / G1_NW;
    const int rhs_wgs = (a.n + 63) / 64;
    const int groups = chunks * B, mg_min = 8 / groups;             // (XCDs per (row chunk, problem) pair, at least)
    const int slots = (tile_wgs + rhs_wgs + mg_min - 1) / mg_min;
    hipLaunchKernelGGL(gram1_kernel, dim3(8 * slots), dim3(G1_NT), 0, s, a, chunks, B, tile_wgs);
    return hipGetLastError();
  }
  const int ncb = (a.n + 63) / 64;
  // tile groups: enough workgroups to occupy the CUs when the batch is small (results identical)
  int tg = 1;
  {
    const int tenv = opt.i(OPT_GRAM_TILE_GROUPS);        // (tests compare splits bit for bit)
    const long wgs = (long)chunks * B;
    // (only when the row chunks alone leave most CUs idle: every group re-reads the rows)
    tg = tenv > 0 ? tenv : (wgs <= 64 ? (int)((256 + wgs - 1) / wgs) : 1);
    // (Every group re-reads the rows: 8 groups at most — but a launch of a handful of workgroups is bound by the
    //  MFMAs of its few CUs: ONE problem of 4096 x 256 took 0.27 ms in 16 workgroups of four tile slots per wave,
    //  0.17 ms in 64 of one slot.  A tile sees the same k-steps in the same order whatever the split: same bits.
    //  Measured and not kept: both halves of the next rows requested a whole chunk ahead — slower, the loads are
    //  not what such a launch waits for.)
    const int tcap = (tenv > 0 || wgs <= 4) ? 32 : 8;
    if (tg > tcap) tg = tcap;
    if (tg > ntile) tg = ntile;
    if (tg < 1) tg = 1;
  }
  {
    // 16 column tiles of J^T J (n = 241 .. 256) and one workgroup per row chunk: the kernel with
    // static tile rows per wave (BLSQ_GRAM16 = 0 keeps the generic one: tests compare the two)
    const int g16_env = opt.i(OPT_GRAM16);
    if (NTJ == 16 && tg == 1 && g16_env != 0 && a.m >= 1) {
      // two row chunks and enough problems to fill the device with one workgroup each: both chunks by
      // the same workgroup, summed in the kernel straight into the final slot (bit-identical to the
      // reduction pass: BLSQ_GRAM_PAIR = 0 keeps that)
      const bool pair = Gfinal && chunks == 2 && B >= 256 && opt.on(OPT_GRAM_PAIR);
      if (pair) { a.Gscr = a.G; a.G = Gfinal; if (fused) *fused = true; }
      const dim3 grid(pair ? 1 : chunks, B, 1);
#define BLSQ_GRAM16(RHS_, PAIR_)                                                              \
  do {                                                                                        \
    static std::atomic<size_t> granted[64];                                                   \
    hipError_t ge = gram_grant_lds(gram16_kernel<RHS_, PAIR_>, lds, granted);                 \
    if (ge != hipSuccess) return ge;                                                          \
    hipLaunchKernelGGL((gram16_kernel<RHS_, PAIR_>), grid, dim3(GR_NT), lds, s, a);           \
  } while (0)
      if (a.rhs_valu) { if (pair) BLSQ_GRAM16(true, true); else BLSQ_GRAM16(true, false); }
      else { if (pair) BLSQ_GRAM16(false, true); else BLSQ_GRAM16(false, false); }
#undef BLSQ_GRAM16
      return hipGetLastError();
    }
  }
  const int per = (((ntile + tg - 1) / tg) + GR_NW - 1) / GR_NW;   // tile slots a wave needs
  if (per <= 1 && tg > 8) {                             // (many tile groups of a tiny launch: one tile per wave)
    if (ncb <= 1) BLSQ_GRAM_LAUNCH(1, 1);
    else if (ncb <= 2) BLSQ_GRAM_LAUNCH(1, 2);
    else BLSQ_GRAM_LAUNCH(1, 5);
  } else if (per <= 2 && tg > 8) {
    if (ncb <= 1) BLSQ_GRAM_LAUNCH(2, 1);
    else if (ncb <= 2) BLSQ_GRAM_LAUNCH(2, 2);
    else BLSQ_GRAM_LAUNCH(2, 5);
  } else if (per <= 4) {                                // n <= 111, or tile groups
    if (ncb <= 1) BLSQ_GRAM_LAUNCH(4, 1);
    else if (ncb <= 2) BLSQ_GRAM_LAUNCH(4, 2);
    else BLSQ_GRAM_LAUNCH(4, 5);
  } else if (per <= 8) {                                // n <= 159, or tile groups
    if (ncb <= 2) BLSQ_GRAM_LAUNCH(8, 2);
    else if (ncb <= 3) BLSQ_GRAM_LAUNCH(8, 3);
    else BLSQ_GRAM_LAUNCH(8, 5);
  } else if (per <= 12) {                               // n <= 207, or tile groups
    if (ncb <= 3) BLSQ_GRAM_LAUNCH(12, 3);
    else if (ncb <= 4) BLSQ_GRAM_LAUNCH(12, 4);
    else BLSQ_GRAM_LAUNCH(12, 5);
  } else if (per <= 17) {                               // n <= 256
    if (ncb <= 4) BLSQ_GRAM_LAUNCH(17, 4); else BLSQ_GRAM_LAUNCH(17, 5);
  } else {
    if (ncb <= 4) BLSQ_GRAM_LAUNCH(20, 4); else BLSQ_GRAM_LAUNCH(20, 5);
  }
#undef BLSQ_GRAM_LAUNCH
  return hipGetLastError();
}
hipError_t launch_gram_reduce(const double* Gpart, int chunks, int NPAD, double* Gout,
                              const int* mask, int B, hipStream_t s) {
  const long sz = (long)NPAD * NPAD;
  if (chunks > 32)
    hipLaunchKernelGGL(gram_reduce_wide_kernel, dim3((unsigned)((sz + GRED_W - 1) / GRED_W), B),
                       dim3(GRED_G * GRED_W), 0, s, Gpart, chunks, NPAD, Gout, mask);
  else
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((sz + 255) / 256), B), dim3(256), 0, s,
                       Gpart, chunks, NPAD, Gout, mask);
  return hipGetLastError();
}
#ifdef BLSQ_CHOL_STAMPS
int gram_debug_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gram_st), sizeof(g_gram_st));
}
#endif
}  // namespace blsq
