// Cholesky factor AND inverse of one 16 x 16 diagonal tile, by one wave (all 64 lanes active).
//
// This is the sequential core of every blocked Cholesky in chol_kernels.hip: row block kb of the
// factor cannot start before the chain of its diagonal tile has finished, so the time of one chain
// times the number of row blocks is a floor under the whole factorisation.
//
//   chol16_columns   the column-per-lane chain: lane j holds column j, 16 pivots in sequence, each
//                    followed by up to 15 rank-1 column updates and the forward substitution of the
//                    inverse — about 30 v_readlane pairs per pivot.
//   chol16_blocked   4 x 4 blocking.  The tile lives in the MFMA accumulator layout (lane (lr, lc),
//                    register g: A[lr + 4 g][lc]); per block q = 0..3
//                      - its 4 x 16 row panel goes through 64 doubles of LDS into the column layout
//                        (every lane: the 4 panel entries of column lc),
//                      - 4 pivots with at most 3 + 2 + 1 rank-1 updates (10 v_readlane pairs),
//                      - ONE v_mfma_f64_16x16x4 applies the panel to the rest of the tile,
//                      - the inverse rides along by blocks:  Y = R'^-T,
//                            Y_q = T_q^T (E_q - sum_{p<q} R'_pq^T Y_p),   T_q = R'_qq^-1  (4 x 4, closed form)
//                        with one MFMA per product (the accumulator layout of a 4 x 16 panel is the
//                        B-operand layout of the next product).
//
// Both leave in LDS:  Dt = R' (row-major, zeros below the diagonal),  Ri = R'^-1 (row-major, upper).
// `nlive` = number of leading columns of the tile that take part in the pivot gate; returns the
// running minimum pivot (squared diagonal of R'), NaN-catching.  Pivots <= 1e-300 give a zero row.
#pragma once
#include "blsq_device.h"

namespace blsq {

__device__ __forceinline__ double chol16_rsqrt(double ds) {
  double ri = __builtin_amdgcn_rsq(ds);
  ri = ri * fma(-0.5 * ds * ri, ri, 1.5);
  ri = ri * fma(-0.5 * ds * ri, ri, 1.5);
  return ri;
}

__device__ __forceinline__ double chol16_columns(double* Dt, double* Ri, int nlive, double pmin) {
  const int lane = threadIdx.x & 63, jc = lane & 15;
  double col[TILE], yy[TILE];
#pragma unroll
  for (int i = 0; i < TILE; ++i) col[i] = Dt[i * 16 + jc];
#pragma unroll
  for (int kk = 0; kk < TILE; ++kk) {
    const double d = read_lane(col[kk], kk);
    if (kk < nlive && !(d >= pmin)) pmin = d;           // also catches NaN
    const bool pos = d > 1e-300;
    double ri = chol16_rsqrt(pos ? d : 1.0);
    if (!pos) ri = 0.0;                                 // zero row (rho = 0, padding)
    const double rkj = col[kk] * ri;
    col[kk] = (jc >= kk) ? rkj : 0.0;
#pragma unroll
    for (int i = kk + 1; i < TILE; ++i) col[i] = fma(-read_lane(rkj, i), rkj, col[i]);
    double ay0 = (jc == kk) ? 1.0 : 0.0, ay1 = 0.0;
#pragma unroll
    for (int p = 0; p < kk; p += 2) {
      ay0 = fma(-read_lane(col[p], kk), yy[p], ay0);
      if (p + 1 < kk) ay1 = fma(-read_lane(col[p + 1], kk), yy[p + 1], ay1);
    }
    yy[kk] = (jc <= kk) ? (ay0 + ay1) * ri : 0.0;
  }
  if (lane < TILE) {
#pragma unroll
    for (int i = 0; i < TILE; ++i) { Dt[i * 16 + jc] = col[i]; Ri[jc * 16 + i] = yy[i]; }
  }
  return pmin;
}

// scr: 64 doubles of LDS for this wave alone.  All LDS traffic is this wave's own, in program order.
__device__ __forceinline__ double chol16_blocked(double* Dt, double* Ri, double* scr, int nlive,
                                                 double pmin) {
  const int lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
  auto wave_lds = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto mf = [](double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); };
  v4d T;
#pragma unroll
  for (int g = 0; g < 4; ++g) T[g] = Dt[(lr + 4 * g) * 16 + lc];
  double Y[4];                                          // Y[q], lane (lr, lc): (R'^-T)[4 q + lr][lc]
  wave_lds();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // a. row panel q -> column layout
    scr[lane] = T[q];
    wave_lds();
    double P[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) P[p] = scr[p * 16 + lc];
    wave_lds();
    // b. four pivots
    double ri[4], rr[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = 4 * q + p;
      const double d = read_lane(P[p], k);
      if (k < nlive && !(d >= pmin)) pmin = d;
      const bool pos = d > 1e-300;
      double r = chol16_rsqrt(pos ? d : 1.0);
      if (!pos) r = 0.0;
      ri[p] = r;
      const double rk = P[p] * r;
      P[p] = (lc >= k) ? rk : 0.0;
#pragma unroll
      for (int p2 = p + 1; p2 < 4; ++p2) {
        rr[p][p2] = read_lane(rk, 4 * q + p2);
        P[p2] = fma(-rr[p][p2], rk, P[p2]);
      }
    }
    // c. the panel as MFMA operand / as rows of R'
    const double Rq = lr == 0 ? P[0] : lr == 1 ? P[1] : lr == 2 ? P[2] : P[3];
    Dt[(4 * q + lr) * 16 + lc] = Rq;
    // d. the rest of the tile
    if (q < 3) T = mf(-Rq, Rq, T);
    // e. inverse: T_q = R'_qq^-1 in closed form (uniform values), Y_q = T_q^T (E_q - W)
    const double t00 = ri[0], t11 = ri[1], t22 = ri[2], t33 = ri[3];
    const double t01 = -ri[0] * (rr[0][1] * t11);
    const double t12 = -ri[1] * (rr[1][2] * t22);
    const double t23 = -ri[2] * (rr[2][3] * t33);
    const double t02 = -ri[0] * fma(rr[0][1], t12, rr[0][2] * t22);
    const double t13 = -ri[1] * fma(rr[1][2], t23, rr[1][3] * t33);
    const double t03 = -ri[0] * fma(rr[0][1], t13, fma(rr[0][2], t23, rr[0][3] * t33));
    const double row0 = lc == 0 ? t00 : lc == 1 ? t01 : lc == 2 ? t02 : t03;
    const double row1 = lc == 1 ? t11 : lc == 2 ? t12 : lc == 3 ? t13 : 0.0;
    const double row2 = lc == 2 ? t22 : lc == 3 ? t23 : 0.0;
    const double row3 = lc == 3 ? t33 : 0.0;
    double Top = lr == 0 ? row0 : lr == 1 ? row1 : lr == 2 ? row2 : row3;
    if (lc >= 4) Top = 0.0;
    double W = 0.0;
    if (q > 0) {
      wave_lds();                                       // (the rows of R' written above are read back)
      v4d Wa[3];
#pragma unroll
      for (int p = 0; p < q; ++p) {
        double a = Dt[(4 * p + lr) * 16 + 4 * q + (lc & 3)];
        if (lc >= 4) a = 0.0;
        Wa[p] = mf(a, Y[p], v4d{0.0, 0.0, 0.0, 0.0});
      }
      W = Wa[0][0];
      if (q > 1) W += Wa[1][0];
      if (q > 2) W += Wa[2][0];
    }
    const double Z = ((lc == 4 * q + lr) ? 1.0 : 0.0) - W;
    const v4d Ya = mf(Top, Z, v4d{0.0, 0.0, 0.0, 0.0});
    Y[q] = Ya[0];
    Ri[lc * 16 + 4 * q + lr] = Y[q];
  }
  wave_lds();
  return pmin;
}

// 1 / sqrt(d) to rounding error from the 24-bit v_rsq_f64 in ONE third-order step (4 dependent
// operations instead of the 6 of two Newton steps):  y (1 + e/2 + 3 e^2 / 8),  e = 1 - d y^2.
__device__ __forceinline__ double chol16_rsqrt3(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}

// row r of a 16-lane-row register to all four rows (gfx950 lane swaps; no LDS)
__device__ __forceinline__ void rows_to_all(double v, double (&P)[4]) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // (r0 r1 r0 r1), (r2 r3 r2 r3)
  const auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const auto l01 = __builtin_amdgcn_permlane16_swap(l32[0], l32[0], false, false);   // (r0 x4), (r1 x4)
  const auto h01 = __builtin_amdgcn_permlane16_swap(h32[0], h32[0], false, false);
  const auto l23 = __builtin_amdgcn_permlane16_swap(l32[1], l32[1], false, false);
  const auto h23 = __builtin_amdgcn_permlane16_swap(h32[1], h32[1], false, false);
  P[0] = __hiloint2double((int)h01[0], (int)l01[0]); P[1] = __hiloint2double((int)h01[1], (int)l01[1]);
  P[2] = __hiloint2double((int)h23[0], (int)l23[0]); P[3] = __hiloint2double((int)h23[1], (int)l23[1]);
}

// chol16_blocked with the panel transposition in registers, the one-step reciprocal square root and
// the triangle mask applied once per panel
// (T: the tile in the accumulator layout, lane (lr, lc) register g = A[lr + 4 g][lc])
__device__ __forceinline__ double chol16_blocked3(v4d T, double* Dt, double* Ri, int nlive, double pmin) {
  const int lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
  auto wave_lds = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto mf = [](double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); };
  double Y[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // The LAST diagonal tile of a factor holds nlive = n mod 16 live variables and the rhs at index nlive; what lies
    // behind it is padding (exact zeros).  A four-row block of nothing but padding leaves zero rows, a zero block of the
    // inverse and an unchanged tile — written here without the pivots' dependent chain (n = 64 and n = 256: the whole
    // last tile but its first block).  Wave-uniform.
    if (4 * q > nlive) {
      Dt[(4 * q + lr) * 16 + lc] = 0.0;
      Ri[lc * 16 + 4 * q + lr] = 0.0;
      Y[q] = 0.0;
      continue;
    }
    double P[4];
    rows_to_all(T[q], P);
    double ri[4], rr[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = 4 * q + p;
      const double d = read_lane(P[p], k);
      if (k < nlive && !(d >= pmin)) pmin = d;
      double r = chol16_rsqrt3(d);
      if (!(d > 1e-300)) r = 0.0;                       // zero row (rho = 0, padding); NaN pivots too
      ri[p] = r;
      P[p] *= r;
#pragma unroll
      for (int p2 = p + 1; p2 < 4; ++p2) {
        rr[p][p2] = read_lane(P[p], 4 * q + p2);
        P[p2] = fma(-rr[p][p2], P[p], P[p2]);
      }
    }
    double Rq = lr == 0 ? P[0] : lr == 1 ? P[1] : lr == 2 ? P[2] : P[3];
    if (lc < 4 * q + lr) Rq = 0.0;
    Dt[(4 * q + lr) * 16 + lc] = Rq;
    if (q < 3) T = mf(-Rq, Rq, T);
    const double t00 = ri[0], t11 = ri[1], t22 = ri[2], t33 = ri[3];
    const double t01 = -ri[0] * (rr[0][1] * t11);
    const double t12 = -ri[1] * (rr[1][2] * t22);
    const double t23 = -ri[2] * (rr[2][3] * t33);
    const double t02 = -ri[0] * fma(rr[0][1], t12, rr[0][2] * t22);
    const double t13 = -ri[1] * fma(rr[1][2], t23, rr[1][3] * t33);
    const double t03 = -ri[0] * fma(rr[0][1], t13, fma(rr[0][2], t23, rr[0][3] * t33));
    const double row0 = lc == 0 ? t00 : lc == 1 ? t01 : lc == 2 ? t02 : t03;
    const double row1 = lc == 1 ? t11 : lc == 2 ? t12 : lc == 3 ? t13 : 0.0;
    const double row2 = lc == 2 ? t22 : lc == 3 ? t23 : 0.0;
    const double row3 = lc == 3 ? t33 : 0.0;
    double Top = lr == 0 ? row0 : lr == 1 ? row1 : lr == 2 ? row2 : row3;
    if (lc >= 4) Top = 0.0;
    double W = 0.0;
    if (q > 0) {
      wave_lds();
      v4d Wa[3];
#pragma unroll
      for (int p = 0; p < q; ++p) {
        double a = Dt[(4 * p + lr) * 16 + 4 * q + (lc & 3)];
        if (lc >= 4) a = 0.0;
        Wa[p] = mf(a, Y[p], v4d{0.0, 0.0, 0.0, 0.0});
      }
      W = Wa[0][0];
      if (q > 1) W += Wa[1][0];
      if (q > 2) W += Wa[2][0];
    }
    const double Z = ((lc == 4 * q + lr) ? 1.0 : 0.0) - W;
    const v4d Ya = mf(Top, Z, v4d{0.0, 0.0, 0.0, 0.0});
    Y[q] = Ya[0];
    Ri[lc * 16 + 4 * q + lr] = Y[q];
  }
  wave_lds();
  return pmin;
}

__device__ __forceinline__ double chol16_blocked3(double* Dt, double* Ri, int nlive, double pmin) {
  const int lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
  v4d T;
#pragma unroll
  for (int g = 0; g < 4; ++g) T[g] = Dt[(lr + 4 * g) * 16 + lc];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return chol16_blocked3(T, Dt, Ri, nlive, pmin);
}

}  // namespace blsq
