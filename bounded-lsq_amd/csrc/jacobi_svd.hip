// One-sided (Hestenes) Jacobi SVD of the small triangular factor.
//
// The reference takes  U, s, V^T = svd(J_augmented)  and  uf = U^T f_augmented
// (bounded_lsq/trf.py:272-274) and, for dogbox, an SVD-based min-norm solve
// (dogbox.py:197, gelsd).  After the QR pass only the n x n triangle R and
// c = Q^T f are left, and rotating the ROWS of [R | c] until the rows of R are
// mutually orthogonal gives  U^T [R | c] = [S V^T | U^T c]:
//     row i  ->  s_i * v_i^T   (length n)   and   uf_i   (entry n)
// which is everything solve_lsq_trust_region (trust_region.py:56-152) consumes
// (it only needs s, s*uf and products V*(...); the ordering of s is irrelevant
// except for max/min, which are returned separately).  No U or V is ever
// accumulated; the rotations are computed from the first n entries of the two
// rows and applied to all n+1.
//
// Layout / schedule (one workgroup of 256 threads per problem):
//   * rows are staged through LDS in pairs of row blocks (2*RB rows x (n+1)
//     doubles <= ~150 KB; RB = 32 for n <= 256): block A stays resident while
//     its partners B = A+1.. are streamed through the second half (the next
//     partner is prefetched into registers while the current pair is rotated);
//   * work unit = a 2x2 ROW TILE PAIR owned by 16 lanes (a DPP row): two rows
//     of A and two rows of B live in REGISTERS (17 doubles per lane per row at
//     n = 256) and receive their 4 cross rotations there, so LDS sees one
//     load/store of a B tile per 4 rotations, and the A tile is loaded/stored
//     once per block pair (LDS write bandwidth was the limiter of the
//     one-pair-per-visit version).  16 tile slots per workgroup; B tiles rotate
//     over the slots round-robin, so every row pair is visited once per sweep;
//   * the dot product is reduced with a 4-step DPP xor-butterfly inside the
//     16-lane row (bit-identical in all 16 lanes, no LDS crossbar);
//   * squared row norms are cached (registers / LDS) and updated by
//     a -= t g, b += t g (refreshed from the data at every block load), so a
//     pair costs ONE dot product; tan/cos/sin come from v_rcp_f64 / v_rsq_f64
//     + Newton steps — any t is a valid rotation, only cos^2+sin^2 = 1 needs
//     full precision.
#include <atomic>

#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int JAC_NT = 256;
static constexpr int JAC_LPR = 16;                // lanes per row (4 row pairs per wave-instruction)
static constexpr int JAC_SLOTS = JAC_NT / JAC_LPR; // 16 concurrent tile slots
static constexpr int JAC_NW = JAC_NT / 64;

// sum over the 32 lanes {0-31} / {32-63}: 16-lane butterfly + one xor-16 swizzle
__device__ __forceinline__ double row32_sum(double v) {
  v = row16_sum(v);
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);   // bit mode: lane ^ 16
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  return v + __hiloint2double(hi, lo);
}

// sum over the JAC_LPR lanes that share a row
__device__ __forceinline__ double row_sum(double v) {
  return (JAC_LPR == 16) ? row16_sum(v) : row32_sum(v);
}

__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// One FAST (self-scaled) rotation between two rows held in registers.
// The true rows are  U = du*u,  V = dv*v  (idu = 1/du, idv = 1/dv); a, b are the
// TRUE squared norms.  With t = tan(theta):  U' = cs (U - t V),  V' = cs (V + t U)
//   <=>  u' = u - (t dv/du) v,  v' = v + (t du/dv) u,  du' = cs du,  dv' = cs dv
// i.e. 2 FMAs per element pair instead of 4 (the scale is applied once, when the
// row leaves the chip).  tan comes from an f32 evaluation (any tan is a valid
// rotation; it only has to annihilate the pair to ~1e-7, quadratic convergence
// does the rest), cos = rsqrt(1 + tan^2) in full f64 so norms are preserved.
// Executed by the 32 lanes of a row group; lane l holds elements l + 32k.  Rows
// are zero-padded to 32*EPL entries, so nothing here is guarded; the carried
// right-hand-side entries (cu, cv) ride along as scalars.
struct RowSt { double a, d, id, c; };     // true norm^2, scale, 1/scale, carried entry

template <int EPL>
__device__ __forceinline__ int rot_regs(double (&u)[EPL], double (&v)[EPL], RowSt& U, RowSt& V,
                                        double tol2) {
  double g0 = 0.0, g1 = 0.0;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    if (k & 1) g1 = fma(u[k], v[k], g1); else g0 = fma(u[k], v[k], g0);
  }
  const double g = row_sum(g0 + g1) * (U.d * V.d);
  const double a = U.a, b = V.a;
  if (!(a > 0.0 && b > 0.0 && g * g > tol2 * a * b)) return 0;
  // cos^2 > 1e-16 (|cos| > 1e-8): after this rotation the pair is NOT yet guaranteed to be
  // below tolerance at the next visit, so another sweep is needed (bit 1 of the result)
  const int big_cos = (g * g > 1e-16 * a * b) ? 2 : 0;
  // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)),  zeta = (b - a) / (2 g)
  const double h = b - a, ag = 2.0 * g;
  const double big = fmax(fabs(h), fabs(ag));
  const int ex = __builtin_amdgcn_frexp_exp(big);          // scale into f32 range
  const float hf = (float)__builtin_amdgcn_ldexp(h, -ex);
  const float gf = (float)__builtin_amdgcn_ldexp(ag, -ex);
  const float rf = __builtin_amdgcn_sqrtf(fmaf(hf, hf, gf * gf));
  float tf = fabsf(gf) * __builtin_amdgcn_rcpf(fabsf(hf) + rf);
  if ((h < 0.0) != (ag < 0.0)) tf = -tf;
  const double t = (double)tf;
  const double w = fma(t, t, 1.0);
  const double cs = fast_rsqrt(w);
  const double ics = w * cs;                               // 1 / cs
  const double t1 = t * V.d * U.id, t2 = t * U.d * V.id;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const double uu = u[k];
    u[k] = fma(-t1, v[k], uu);
    v[k] = fma(t2, uu, v[k]);
  }
  const double cu = U.c;
  U.c = fma(-t1, V.c, cu);
  V.c = fma(t2, cu, V.c);
  U.d *= cs; V.d *= cs; U.id *= ics; V.id *= ics;
  U.a = a - t * g;
  V.a = b + t * g;
  return 1 | big_cos;
}

template <int EPL>
__device__ __forceinline__ void row_load(double (&u)[EPL], const double* row, int l) {
#pragma unroll
  for (int k = 0; k < EPL; ++k) u[k] = row[l + JAC_LPR * k];
}
template <int EPL>
__device__ __forceinline__ void row_store(const double (&u)[EPL], double* row, int l) {
#pragma unroll
  for (int k = 0; k < EPL; ++k) row[l + JAC_LPR * k] = u[k];
}

// round-robin (chess tournament) pairing of `np` (even) players, round r,
// pair index i in [0, np/2)
__device__ __forceinline__ void rr_pair(int np, int r, int i, int& p, int& q) {
  const int m1 = np - 1;
  if (i == 0) { p = m1; q = r; }
  else { p = (r + i) % m1; q = (r - i + m1) % m1; }
}

// ---- block transfers -------------------------------------------------------
// LDS row layout: stride LDX = 32*EPL doubles, entries [0,n) = the row of R,
// [n, LDX) = 0; the carried entry (global column n) lives in cz[row].
// A block is RB (<= 32) rows; wave w moves rows w, w+8, w+16, w+24 of it, lanes
// stride the row.  All loads of a block are issued before the first use.
template <int EPL>
struct BlockRegs {
  static constexpr int CH = (JAC_LPR * EPL + 63) / 64;
  static constexpr int RPW = 32 / JAC_NW;      // rows of a 32-row block per wave
  double t[RPW][CH];
  double c[RPW];
};

template <int EPL>
__device__ __forceinline__ void block_fetch(BlockRegs<EPL>& R, const double* X, int ld, int n,
                                            int g0, int RB, int w, int lane) {
#pragma unroll
  for (int j = 0; j < BlockRegs<EPL>::RPW; ++j) {
    const int r = w + JAC_NW * j;
    const int gr = g0 + r;
    const bool rok = (r < RB) && (gr < n);
    const double* src = X + (long)(rok ? gr : 0) * ld;   // clamped: loads are unconditional
#pragma unroll
    for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
      const int e = lane + 64 * c;
      const double val = src[(e < n) ? e : 0];
      R.t[j][c] = (rok && e < n) ? val : 0.0;
    }
    const double cval = src[n];
    R.c[j] = rok ? cval : 0.0;
  }
}
template <int EPL>
__device__ __forceinline__ void block_commit(const BlockRegs<EPL>& R, double* Xs, double* cz,
                                             int s0, int RB, int w, int lane) {
  constexpr int LDX = JAC_LPR * EPL;
#pragma unroll
  for (int j = 0; j < BlockRegs<EPL>::RPW; ++j) {
    const int r = w + JAC_NW * j;
    if (r < RB) {
      double* dst = Xs + (s0 + r) * LDX;
#pragma unroll
      for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
        const int e = lane + 64 * c;
        if (e < LDX) dst[e] = R.t[j][c];
      }
      if (lane == 0) cz[s0 + r] = R.c[j];
    }
  }
}
template <int EPL>
__device__ __forceinline__ void block_store(double* X, int ld, int n, int g0, const double* Xs,
                                            const double* cz, const double* dsc, int s0, int RB,
                                            int w, int lane) {
  constexpr int LDX = JAC_LPR * EPL;
  double t[BlockRegs<EPL>::RPW][BlockRegs<EPL>::CH];
  double tc[BlockRegs<EPL>::RPW];
#pragma unroll
  for (int j = 0; j < BlockRegs<EPL>::RPW; ++j) {
    const int r = w + JAC_NW * j;
    const int rr = s0 + (r < RB ? r : 0);
    const double* src = Xs + rr * LDX;
    const double sc = dsc[rr];                     // fold the fast-rotation scale in
#pragma unroll
    for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
      const int e = lane + 64 * c;
      t[j][c] = src[(e < LDX) ? e : 0] * sc;
    }
    tc[j] = cz[rr] * sc;
  }
#pragma unroll
  for (int j = 0; j < BlockRegs<EPL>::RPW; ++j) {
    const int r = w + JAC_NW * j;
    const int gr = g0 + r;
    if (r < RB && gr < n) {
      double* dst = X + (long)gr * ld;
#pragma unroll
      for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
        const int e = lane + 64 * c;
        if (e < n) dst[e] = t[j][c];
      }
      if (lane == 0) dst[n] = tc[j];
    }
  }
}
// squared norms of LDS rows [s0, s0+cnt), one row per 32 lanes; resets the
// fast-rotation scales to 1 (after optionally folding them into the data)
template <int EPL>
__device__ __forceinline__ void block_norms(double* Xs, double* cz, double* sq, double* dsc,
                                            double* idsc, int s0, int cnt, int slot, int l,
                                            bool apply) {
  constexpr int LDX = JAC_LPR * EPL;
  for (int r0 = 0; r0 < cnt; r0 += JAC_SLOTS) {
    const int r = r0 + slot;
    const int rr = s0 + (r < cnt ? r : 0);
    double* src = Xs + rr * LDX;
    const double sc = apply ? dsc[rr] : 1.0;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const double v = src[l + JAC_LPR * k] * sc;
      if (apply && r < cnt) src[l + JAC_LPR * k] = v;
      acc = fma(v, v, acc);
    }
    acc = row_sum(acc);
    if (r < cnt && l == 0) {
      sq[rr] = acc; dsc[rr] = 1.0; idsc[rr] = 1.0;
      if (apply) cz[rr] *= sc;
    }
  }
}

template <int EPL>
__global__ __launch_bounds__(JAC_NT) void jacobi_rows_kernel(JacobiArgs a) {
  extern __shared__ double lds[];
  __shared__ double red[32];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int slot = tid / JAC_LPR, l = tid % JAC_LPR;
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.N;
  const int n = N - 1;
  double* X = a.X + (long)b * a.strideX;
  const int ld = a.ld;
  if (n <= 0) {
    if (tid == 0) { a.srange[2 * b] = 0.0; a.srange[2 * b + 1] = 0.0; a.sweeps[b] = 0; }
    return;
  }
  const int RB = a.RB;                       // rows per block (even, <= 32)
  constexpr int LDX = JAC_LPR * EPL;         // LDS row stride (zero padded)
  const int NROW = 2 * RB + 2;               // + a dummy zero tile for idle slots
  double* Xs = lds;                          // [NROW][LDX]: [0,RB) = A, [RB,2RB) = B
  double* sq = Xs + NROW * LDX;              // [NROW] true squared norms
  double* dsc = sq + NROW;                   // [NROW] fast-rotation scale of each LDS row
  double* idsc = dsc + NROW;                 // [NROW] its inverse
  double* cz = idsc + NROW;                  // [NROW] carried entry (column n of the row)
  const double tol = sqrt((double)n) * 2.220446049250313e-16;
  const double tol2 = tol * tol;
  const int nb = (n + RB - 1) / RB;          // row blocks (last may be partial: zero rows)
  const int NT2 = RB / 2;                    // 2-row tiles per block (<= 16 = JAC_SLOTS)
  const int DUMMY = 2 * RB;                  // rows 2RB, 2RB+1: zeros, norm 0 -> never rotated
  BlockRegs<EPL> pre;
  for (int e = tid; e < 2 * LDX; e += JAC_NT) Xs[DUMMY * LDX + e] = 0.0;
  if (tid < 2) { sq[DUMMY + tid] = 0.0; dsc[DUMMY + tid] = 1.0; idsc[DUMMY + tid] = 1.0; cz[DUMMY + tid] = 0.0; }

  auto st_load = [&](int row) -> RowSt {
    RowSt r;
    r.a = sq[row]; r.d = dsc[row]; r.id = idsc[row]; r.c = cz[row];
    return r;
  };
  auto st_store = [&](int row, const RowSt& r) {
    sq[row] = r.a; dsc[row] = r.d; idsc[row] = r.id; cz[row] = r.c;
  };

  // all pairs between the two 2-row tiles at LDS rows (ra, ra+1) and (rb, rb+1);
  // `within` adds the pair inside each tile (done once per sweep per tile).
  // Idle slots pass the dummy tile for both (zero norms: nothing happens).
  auto tile_pair = [&](int ra, int rb, bool within) -> int {
    double u0[EPL], u1[EPL], v0[EPL], v1[EPL];
    row_load<EPL>(u0, Xs + ra * LDX, l);
    row_load<EPL>(u1, Xs + (ra + 1) * LDX, l);
    row_load<EPL>(v0, Xs + rb * LDX, l);
    row_load<EPL>(v1, Xs + (rb + 1) * LDX, l);
    RowSt A0 = st_load(ra), A1 = st_load(ra + 1), B0 = st_load(rb), B1 = st_load(rb + 1);
    int rot = 0;
    if (within) {
      rot |= rot_regs<EPL>(u0, u1, A0, A1, tol2);
      rot |= rot_regs<EPL>(v0, v1, B0, B1, tol2);
    }
    rot |= rot_regs<EPL>(u0, v0, A0, B0, tol2);
    rot |= rot_regs<EPL>(u1, v1, A1, B1, tol2);
    rot |= rot_regs<EPL>(u0, v1, A0, B1, tol2);
    rot |= rot_regs<EPL>(u1, v0, A1, B0, tol2);
    if (rot) {
      row_store<EPL>(u0, Xs + ra * LDX, l);
      row_store<EPL>(u1, Xs + (ra + 1) * LDX, l);
      row_store<EPL>(v0, Xs + rb * LDX, l);
      row_store<EPL>(v1, Xs + (rb + 1) * LDX, l);
      if (l == 0) { st_store(ra, A0); st_store(ra + 1, A1); st_store(rb, B0); st_store(rb + 1, B1); }
    }
    return rot;
  };

  // intra-block sweep of the blocks at LDS row offsets 0 (A) and RB (B):
  // round-robin over the NT2 tiles of each block; the first half of the slots
  // serves A, the second half B.  Round 0 also rotates the pair inside every tile.
  auto intra = [&](bool doA, bool doB) -> int {
    int rot = 0;
    if (NT2 < 2) {                           // RB == 2: a block is a single tile
      const bool valid = (slot == 0 && doA) || (slot == 1 && doB);
      double u0[EPL], u1[EPL];
      const int r0 = valid ? ((slot == 1) ? RB : 0) : DUMMY;
      row_load<EPL>(u0, Xs + r0 * LDX, l);
      row_load<EPL>(u1, Xs + (r0 + 1) * LDX, l);
      RowSt A0 = st_load(r0), A1 = st_load(r0 + 1);
      rot = rot_regs<EPL>(u0, u1, A0, A1, tol2);
      if (rot) {
        row_store<EPL>(u0, Xs + r0 * LDX, l);
        row_store<EPL>(u1, Xs + (r0 + 1) * LDX, l);
        if (l == 0) { st_store(r0, A0); st_store(r0 + 1, A1); }
      }
      __syncthreads();
      return rot;
    }
    const int np2 = NT2 + (NT2 & 1);         // even number of players
    for (int r = 0; r < np2 - 1; ++r) {
      const bool inA = slot < np2 / 2;
      const int li = inA ? slot : slot - np2 / 2;
      bool valid = (li < np2 / 2) && (slot < np2) && (inA ? doA : doB);
      int p = 0, q = 0;
      rr_pair(np2, r, li < np2 / 2 ? li : 0, p, q);
      if (p >= NT2 || q >= NT2) valid = false;     // dummy tile (odd NT2)
      const int off = inA ? 0 : RB;
      rot |= tile_pair(valid ? off + 2 * p : DUMMY, valid ? off + 2 * q : DUMMY, r == 0);
      __syncthreads();
    }
    return rot;
  };

  // cross phase between the block at rows [0,RB) and the block at [RB,2RB):
  // slot s keeps A tile s in registers; B tile (s + r) mod NT2 visits it in round r.
  auto cross = [&]() -> int {
    const bool valid = slot < NT2;
    const int ra = valid ? 2 * slot : DUMMY;
    double u0[EPL], u1[EPL];
    row_load<EPL>(u0, Xs + ra * LDX, l);
    row_load<EPL>(u1, Xs + (ra + 1) * LDX, l);
    RowSt A0 = st_load(ra), A1 = st_load(ra + 1);
    int rotA = 0;
    int tb = slot;                           // B tile index, advances by one per round
    for (int r = 0; r < NT2; ++r) {
      const int rb = valid ? RB + 2 * tb : DUMMY;
      tb = (tb + 1 == NT2) ? 0 : tb + 1;
      double v0[EPL], v1[EPL];
      row_load<EPL>(v0, Xs + rb * LDX, l);
      row_load<EPL>(v1, Xs + (rb + 1) * LDX, l);
      RowSt B0 = st_load(rb), B1 = st_load(rb + 1);
      int rr = 0;
      rr |= rot_regs<EPL>(u0, v0, A0, B0, tol2);
      rr |= rot_regs<EPL>(u1, v1, A1, B1, tol2);
      rr |= rot_regs<EPL>(u0, v1, A0, B1, tol2);
      rr |= rot_regs<EPL>(u1, v0, A1, B0, tol2);
      if (rr) {
        row_store<EPL>(v0, Xs + rb * LDX, l);
        row_store<EPL>(v1, Xs + (rb + 1) * LDX, l);
        if (l == 0) { st_store(rb, B0); st_store(rb + 1, B1); }
      }
      rotA |= rr;
      __syncthreads();
    }
    if (rotA) {
      row_store<EPL>(u0, Xs + ra * LDX, l);
      row_store<EPL>(u1, Xs + (ra + 1) * LDX, l);
      if (l == 0) { st_store(ra, A0); st_store(ra + 1, A1); }
    }
    __syncthreads();
    return rotA;
  };

  int sweep = 0;
  if (nb <= 2) {
    // ---- everything fits: load once, full sweeps in LDS ---------------------
    block_fetch<EPL>(pre, X, ld, n, 0, RB, w, lane);
    block_commit<EPL>(pre, Xs, cz, 0, RB, w, lane);
    block_fetch<EPL>(pre, X, ld, n, RB, RB, w, lane);
    block_commit<EPL>(pre, Xs, cz, RB, RB, w, lane);
    __syncthreads();
    for (; sweep < a.max_sweeps;) {
      // (re)compute the true norms; from the 2nd sweep on first fold the scales in
      block_norms<EPL>(Xs, cz, sq, dsc, idsc, 0, 2 * RB, slot, l, sweep > 0);
      __syncthreads();
      int rotated = intra(true, nb == 2);
      if (nb == 2) rotated |= cross();
      ++sweep;
      if (!block_or(rotated & 2, red)) break;   // quadratic convergence: what was rotated is done
    }
    block_store<EPL>(X, ld, n, 0, Xs, cz, dsc, 0, RB, w, lane);
    block_store<EPL>(X, ld, n, RB, Xs, cz, dsc, RB, RB, w, lane);
  } else {
    // ---- block-cyclic sweeps: A resident, partners B streamed ---------------
    for (; sweep < a.max_sweeps;) {
      int rotated = 0;
      for (int A = 0; A + 1 < nb; ++A) {
        block_fetch<EPL>(pre, X, ld, n, A * RB, RB, w, lane);
        block_commit<EPL>(pre, Xs, cz, 0, RB, w, lane);
        for (int Bk = A + 1; Bk < nb; ++Bk) {
          block_fetch<EPL>(pre, X, ld, n, Bk * RB, RB, w, lane);
          block_commit<EPL>(pre, Xs, cz, RB, RB, w, lane);
          __syncthreads();
          if (Bk == A + 1) block_norms<EPL>(Xs, cz, sq, dsc, idsc, 0, RB, slot, l, false);
          block_norms<EPL>(Xs, cz, sq, dsc, idsc, RB, RB, slot, l, false);
          __syncthreads();
          if (Bk == A + 1) rotated |= intra(A == 0, true);  // each block once per sweep
          rotated |= cross();
          block_store<EPL>(X, ld, n, Bk * RB, Xs, cz, dsc, RB, RB, w, lane);
          __threadfence_block();
          __syncthreads();
        }
        block_store<EPL>(X, ld, n, A * RB, Xs, cz, dsc, 0, RB, w, lane);
        __threadfence_block();
        __syncthreads();
      }
      ++sweep;
      if (!block_or(rotated & 2, red)) break;
    }
  }
  __threadfence_block();
  __syncthreads();

  double smax = 0.0, smin = __builtin_inf();
  for (int i = w; i < n; i += JAC_NW) {
    const double* xi = X + (long)i * ld;
    double aa = 0.0;
    for (int e = lane; e < n; e += WAVE) { const double u = xi[e]; aa = fma(u, u, aa); }
    aa = wave_sum(aa);
    const double si = sqrt(aa);
    if (lane == 0) {
      a.s[(long)b * ld + i] = si;
      a.uf[(long)b * ld + i] = xi[n];
    }
    smax = fmax(smax, si); smin = fmin(smin, si);
  }
  smax = block_max(smax, red);
  smin = block_min(smin, red);
  if (tid == 0) {
    a.srange[2 * b] = smax; a.srange[2 * b + 1] = smin; a.sweeps[b] = sweep;
  }
}

// rows per LDS block for LDS row stride ldx doubles (even, <= 32; 2*RB+2 rows <= ~156 KB)
int jacobi_block_rows(int ldx) {
  int rb = (int)(((156 * 1024) / (8 * (size_t)ldx) - 2) / 2);
  if (rb > 32) rb = 32;
  rb &= ~1;
  if (rb < 2) rb = 2;
  return rb;
}

template <int EPL>
static hipError_t launch_jacobi_t(const JacobiArgs& a, int B, size_t lds, hipStream_t st) {
  // the attribute is per device; one ctx per host thread: launches may race (a repeated set is harmless)
  static std::atomic<size_t> configured_dev[64];
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  std::atomic<size_t>& configured = configured_dev[dev_ & 63];
  if (lds > configured.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)jacobi_rows_kernel<EPL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured.store(lds, std::memory_order_release);
  }
  hipLaunchKernelGGL(jacobi_rows_kernel<EPL>, dim3(B), dim3(JAC_NT), lds, st, a);
  return hipGetLastError();
}

template <int EPL>
static hipError_t launch_jacobi_e(JacobiArgs a, int B, hipStream_t st) {
  const int ldx = JAC_LPR * EPL;
  a.RB = jacobi_block_rows(ldx);
  const size_t nrow = 2 * (size_t)a.RB + 2;
  const size_t lds = sizeof(double) * (nrow * ldx + 4 * nrow);
  return launch_jacobi_t<EPL>(a, B, lds, st);
}

hipError_t launch_jacobi(const JacobiArgs& a, int B, hipStream_t st) {
  const int epl = (a.N - 1 + JAC_LPR - 1) / JAC_LPR;      // row length n = N - 1
  if (epl <= 1) return launch_jacobi_e<1>(a, B, st);
  if (epl <= 2) return launch_jacobi_e<2>(a, B, st);
  if (epl <= 4) return launch_jacobi_e<4>(a, B, st);
  if (epl <= 8) return launch_jacobi_e<8>(a, B, st);
  if (epl <= 16) return launch_jacobi_e<16>(a, B, st);
  if (epl <= 34) return launch_jacobi_e<34>(a, B, st);
  return launch_jacobi_e<68>(a, B, st);
}

}  // namespace blsq
