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
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int JAC_NT = 512;
static constexpr int JAC_LPR = 32;                // lanes per row
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
// Executed by the 32 lanes of a row group; lane l holds elements l + 32k.
template <int EPL>
__device__ __forceinline__ int rot_regs(double (&u)[EPL], double (&v)[EPL], double& a, double& b,
                                        double& du, double& idu, double& dv, double& idv, int n,
                                        double tol2, int l) {
  double g0 = 0.0, g1 = 0.0;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    if (l + JAC_LPR * k < n) {
      if (k & 1) g1 = fma(u[k], v[k], g1); else g0 = fma(u[k], v[k], g0);
    }
  }
  const double g = row32_sum(g0 + g1) * (du * dv);
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
  const double t1 = t * dv * idu, t2 = t * du * idv;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const double uu = u[k];
    u[k] = fma(-t1, v[k], uu);
    v[k] = fma(t2, uu, v[k]);
  }
  du *= cs; dv *= cs; idu *= ics; idv *= ics;
  a -= t * g;
  b += t * g;
  return 1 | big_cos;
}

template <int EPL>
__device__ __forceinline__ void row_load(double (&u)[EPL], const double* row, int N, int l) {
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = l + JAC_LPR * k;
    const double val = row[(e < N) ? e : 0];         // unconditional load, select after
    u[k] = (e < N) ? val : 0.0;
  }
}
template <int EPL>
__device__ __forceinline__ void row_store(const double (&u)[EPL], double* row, int N, int l) {
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = l + JAC_LPR * k;
    if (e < N) row[e] = u[k];
  }
}

// round-robin (chess tournament) pairing of `np` (even) players, round r,
// pair index i in [0, np/2)
__device__ __forceinline__ void rr_pair(int np, int r, int i, int& p, int& q) {
  const int m1 = np - 1;
  if (i == 0) { p = m1; q = r; }
  else { p = (r + i) % m1; q = (r - i + m1) % m1; }
}

// ---- block transfers -------------------------------------------------------
// A block is RB (<= 32) rows; wave w moves rows w, w+8, w+16, w+24 of it, lanes
// stride the row.  All loads of a block are issued before the first use, so a
// block costs one memory round trip (and can be prefetched into registers
// while the previous block pair is being rotated).
template <int EPL>
struct BlockRegs {
  static constexpr int CH = (JAC_LPR * EPL + 63) / 64;
  double t[4][CH];
};

template <int EPL>
__device__ __forceinline__ void block_fetch(BlockRegs<EPL>& R, const double* X, int ld, int n,
                                            int N, int g0, int RB, int w, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + JAC_NW * j;
    const int gr = g0 + r;
    const bool rok = (r < RB) && (gr < n);
    const double* src = X + (long)(rok ? gr : 0) * ld;   // clamped: loads are unconditional
#pragma unroll
    for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
      const int e = lane + 64 * c;
      const double val = src[(e < N) ? e : 0];
      R.t[j][c] = (rok && e < N) ? val : 0.0;
    }
  }
}
template <int EPL>
__device__ __forceinline__ void block_commit(const BlockRegs<EPL>& R, double* Xs, int ldx, int N,
                                             int s0, int RB, int w, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + JAC_NW * j;
    if (r < RB) {
      double* dst = Xs + (s0 + r) * ldx;
#pragma unroll
      for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
        const int e = lane + 64 * c;
        if (e < N) dst[e] = R.t[j][c];
      }
    }
  }
}
template <int EPL>
__device__ __forceinline__ void block_store(double* X, int ld, int n, int N, int g0,
                                            const double* Xs, const double* dsc, int ldx, int s0,
                                            int RB, int w, int lane) {
  double t[4][BlockRegs<EPL>::CH];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + JAC_NW * j;
    const double* src = Xs + (s0 + (r < RB ? r : 0)) * ldx;
    const double sc = dsc[s0 + (r < RB ? r : 0)];          // fold the fast-rotation scale in
#pragma unroll
    for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
      const int e = lane + 64 * c;
      t[j][c] = src[(e < N) ? e : 0] * sc;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + JAC_NW * j;
    const int gr = g0 + r;
    if (r < RB && gr < n) {
      double* dst = X + (long)gr * ld;
#pragma unroll
      for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
        const int e = lane + 64 * c;
        if (e < N) dst[e] = t[j][c];
      }
    }
  }
}
// squared norms (first n entries) of LDS rows [s0, s0+cnt), one row per 32 lanes;
// also resets the fast-rotation scales of those rows to 1
template <int EPL>
__device__ __forceinline__ void block_norms(double* Xs, double* sq, double* dsc, double* idsc,
                                            int ldx, int n, int N, int s0, int cnt, int slot,
                                            int l, bool apply) {
  for (int r0 = 0; r0 < cnt; r0 += JAC_SLOTS) {
    const int r = r0 + slot;
    double* src = Xs + (s0 + (r < cnt ? r : 0)) * ldx;
    const double sc = apply ? dsc[s0 + (r < cnt ? r : 0)] : 1.0;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = l + JAC_LPR * k;
      const double v = src[(e < N) ? e : 0] * sc;
      if (apply && r < cnt && e < N) src[e] = v;       // fold the scale into the data
      if (e < n) acc = fma(v, v, acc);
    }
    acc = row32_sum(acc);
    if (r < cnt && l == 0) { sq[s0 + r] = acc; dsc[s0 + r] = 1.0; idsc[s0 + r] = 1.0; }
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
  const int ldx = N;                         // LDS row stride
  double* Xs = lds;                          // [2*RB][ldx]: slots [0,RB) = A, [RB,2RB) = B
  double* sq = Xs + 2 * RB * ldx;            // [2*RB] true squared norms
  double* dsc = sq + 2 * RB;                 // [2*RB] fast-rotation scale of each LDS row
  double* idsc = dsc + 2 * RB;               // [2*RB] its inverse
  const double tol = sqrt((double)n) * 2.220446049250313e-16;
  const double tol2 = tol * tol;
  const int nb = (n + RB - 1) / RB;          // row blocks (last may be partial: zero rows)
  const int NT2 = RB / 2;                    // 2-row tiles per block (<= 16 = JAC_SLOTS)
  BlockRegs<EPL> pre;

  struct RowSt { double a, d, id; };
  auto st_load = [&](int row, bool valid) -> RowSt {
    RowSt r;
    r.a = valid ? sq[row] : 0.0;
    r.d = valid ? dsc[row] : 1.0;
    r.id = valid ? idsc[row] : 1.0;
    return r;
  };
  auto st_store = [&](int row, const RowSt& r) { sq[row] = r.a; dsc[row] = r.d; idsc[row] = r.id; };

  // all pairs between the two 2-row tiles at LDS rows (ta, ta+1) and (tb, tb+1);
  // `within` adds the pair inside each tile (done once per sweep per tile).
  auto tile_pair = [&](int ta, int tb, bool valid, bool within) -> int {
    double u0[EPL], u1[EPL], v0[EPL], v1[EPL];
    const int ra = valid ? ta : 0, rb = valid ? tb : 0;
    row_load<EPL>(u0, Xs + ra * ldx, N, l);
    row_load<EPL>(u1, Xs + (ra + 1) * ldx, N, l);
    row_load<EPL>(v0, Xs + rb * ldx, N, l);
    row_load<EPL>(v1, Xs + (rb + 1) * ldx, N, l);
    RowSt A0 = st_load(ra, valid), A1 = st_load(ra + 1, valid);
    RowSt B0 = st_load(rb, valid), B1 = st_load(rb + 1, valid);
    int rot = 0;
    if (within) {
      rot |= rot_regs<EPL>(u0, u1, A0.a, A1.a, A0.d, A0.id, A1.d, A1.id, n, tol2, l);
      rot |= rot_regs<EPL>(v0, v1, B0.a, B1.a, B0.d, B0.id, B1.d, B1.id, n, tol2, l);
    }
    rot |= rot_regs<EPL>(u0, v0, A0.a, B0.a, A0.d, A0.id, B0.d, B0.id, n, tol2, l);
    rot |= rot_regs<EPL>(u1, v1, A1.a, B1.a, A1.d, A1.id, B1.d, B1.id, n, tol2, l);
    rot |= rot_regs<EPL>(u0, v1, A0.a, B1.a, A0.d, A0.id, B1.d, B1.id, n, tol2, l);
    rot |= rot_regs<EPL>(u1, v0, A1.a, B0.a, A1.d, A1.id, B0.d, B0.id, n, tol2, l);
    if (valid && rot) {
      row_store<EPL>(u0, Xs + ra * ldx, N, l);
      row_store<EPL>(u1, Xs + (ra + 1) * ldx, N, l);
      row_store<EPL>(v0, Xs + rb * ldx, N, l);
      row_store<EPL>(v1, Xs + (rb + 1) * ldx, N, l);
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
      const int r0 = (slot == 1) ? RB : 0;
      row_load<EPL>(u0, Xs + r0 * ldx, N, l);
      row_load<EPL>(u1, Xs + (r0 + 1) * ldx, N, l);
      RowSt A0 = st_load(r0, valid), A1 = st_load(r0 + 1, valid);
      rot = rot_regs<EPL>(u0, u1, A0.a, A1.a, A0.d, A0.id, A1.d, A1.id, n, tol2, l);
      if (valid && rot) {
        row_store<EPL>(u0, Xs + r0 * ldx, N, l);
        row_store<EPL>(u1, Xs + (r0 + 1) * ldx, N, l);
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
      rot |= tile_pair(off + 2 * p, off + 2 * q, valid, r == 0);
      __syncthreads();
    }
    return rot;
  };

  // cross phase between the block at rows [0,RB) and the block at [RB,2RB):
  // slot s keeps A tile s in registers; B tile (s + r) mod NT2 visits it in round r.
  auto cross = [&]() -> int {
    const bool valid = slot < NT2;
    const int ra = valid ? 2 * slot : 0;
    double u0[EPL], u1[EPL];
    row_load<EPL>(u0, Xs + ra * ldx, N, l);
    row_load<EPL>(u1, Xs + (ra + 1) * ldx, N, l);
    RowSt A0 = st_load(ra, valid), A1 = st_load(ra + 1, valid);
    int rotA = 0;
    for (int r = 0; r < NT2; ++r) {
      const int rb = valid ? RB + 2 * ((slot + r) % NT2) : RB;
      double v0[EPL], v1[EPL];
      row_load<EPL>(v0, Xs + rb * ldx, N, l);
      row_load<EPL>(v1, Xs + (rb + 1) * ldx, N, l);
      RowSt B0 = st_load(rb, valid), B1 = st_load(rb + 1, valid);
      int rr = 0;
      rr |= rot_regs<EPL>(u0, v0, A0.a, B0.a, A0.d, A0.id, B0.d, B0.id, n, tol2, l);
      rr |= rot_regs<EPL>(u1, v1, A1.a, B1.a, A1.d, A1.id, B1.d, B1.id, n, tol2, l);
      rr |= rot_regs<EPL>(u0, v1, A0.a, B1.a, A0.d, A0.id, B1.d, B1.id, n, tol2, l);
      rr |= rot_regs<EPL>(u1, v0, A1.a, B0.a, A1.d, A1.id, B0.d, B0.id, n, tol2, l);
      if (valid && rr) {
        row_store<EPL>(v0, Xs + rb * ldx, N, l);
        row_store<EPL>(v1, Xs + (rb + 1) * ldx, N, l);
        if (l == 0) { st_store(rb, B0); st_store(rb + 1, B1); }
      }
      rotA |= rr;
      __syncthreads();
    }
    if (valid && rotA) {
      row_store<EPL>(u0, Xs + ra * ldx, N, l);
      row_store<EPL>(u1, Xs + (ra + 1) * ldx, N, l);
      if (l == 0) { st_store(ra, A0); st_store(ra + 1, A1); }
    }
    __syncthreads();
    return rotA;
  };

  int sweep = 0;
  if (nb <= 2) {
    // ---- everything fits: load once, full sweeps in LDS ---------------------
    block_fetch<EPL>(pre, X, ld, n, N, 0, RB, w, lane);
    block_commit<EPL>(pre, Xs, ldx, N, 0, RB, w, lane);
    block_fetch<EPL>(pre, X, ld, n, N, RB, RB, w, lane);
    block_commit<EPL>(pre, Xs, ldx, N, RB, RB, w, lane);
    __syncthreads();
    for (; sweep < a.max_sweeps;) {
      // (re)compute the true norms; from the 2nd sweep on first fold the scales in
      block_norms<EPL>(Xs, sq, dsc, idsc, ldx, n, N, 0, 2 * RB, slot, l, sweep > 0);
      __syncthreads();
      int rotated = intra(true, nb == 2);
      if (nb == 2) rotated |= cross();
      ++sweep;
      if (!block_or(rotated & 2, red)) break;   // quadratic convergence: what was rotated is done
    }
    block_store<EPL>(X, ld, n, N, 0, Xs, dsc, ldx, 0, RB, w, lane);
    block_store<EPL>(X, ld, n, N, RB, Xs, dsc, ldx, RB, RB, w, lane);
  } else {
    // ---- block-cyclic sweeps: A resident, partners B streamed (prefetched) --
    for (; sweep < a.max_sweeps;) {
      int rotated = 0;
      for (int A = 0; A + 1 < nb; ++A) {
        block_fetch<EPL>(pre, X, ld, n, N, A * RB, RB, w, lane);
        block_commit<EPL>(pre, Xs, ldx, N, 0, RB, w, lane);
        for (int Bk = A + 1; Bk < nb; ++Bk) {
          block_fetch<EPL>(pre, X, ld, n, N, Bk * RB, RB, w, lane);
          block_commit<EPL>(pre, Xs, ldx, N, RB, RB, w, lane);
          __syncthreads();
          if (Bk == A + 1) block_norms<EPL>(Xs, sq, dsc, idsc, ldx, n, N, 0, RB, slot, l, false);
          block_norms<EPL>(Xs, sq, dsc, idsc, ldx, n, N, RB, RB, slot, l, false);
          __syncthreads();
          if (Bk == A + 1) rotated |= intra(A == 0, true);  // each block once per sweep
          rotated |= cross();
          block_store<EPL>(X, ld, n, N, Bk * RB, Xs, dsc, ldx, RB, RB, w, lane);
          __threadfence_block();
          __syncthreads();
        }
        block_store<EPL>(X, ld, n, N, A * RB, Xs, dsc, ldx, 0, RB, w, lane);
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

// rows per LDS block for row length N (even, <= 32, 2*RB*N doubles <= ~150 KB)
int jacobi_block_rows(int N) {
  int rb = (int)((150 * 1024) / (16 * (size_t)N));
  if (rb > 32) rb = 32;
  rb &= ~1;
  if (rb < 2) rb = 2;
  return rb;
}

template <int EPL>
static hipError_t launch_jacobi_t(const JacobiArgs& a, int B, size_t lds, hipStream_t st) {
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute((const void*)jacobi_rows_kernel<EPL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL(jacobi_rows_kernel<EPL>, dim3(B), dim3(JAC_NT), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_jacobi(const JacobiArgs& a_in, int B, hipStream_t st) {
  JacobiArgs a = a_in;
  a.RB = jacobi_block_rows(a.N);
  const size_t lds = sizeof(double) * ((size_t)2 * a.RB * a.N + 6 * a.RB);
  const int epl = (a.N + JAC_LPR - 1) / JAC_LPR;
  if (epl <= 1) return launch_jacobi_t<1>(a, B, lds, st);
  if (epl <= 3) return launch_jacobi_t<3>(a, B, lds, st);
  if (epl <= 5) return launch_jacobi_t<5>(a, B, lds, st);
  if (epl <= 9) return launch_jacobi_t<9>(a, B, lds, st);
  if (epl <= 17) return launch_jacobi_t<17>(a, B, lds, st);
  return launch_jacobi_t<34>(a, B, lds, st);
}

}  // namespace blsq
