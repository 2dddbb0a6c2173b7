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
// Layout / schedule (one workgroup of 512 threads per problem):
//   * rows are staged through LDS in pairs of row blocks (2*RB rows x (n+1)
//     doubles <= ~150 KB; RB = 32 for n <= 256): block A stays resident while
//     its partners B = A+1.. are streamed through the second half (the next
//     partner is prefetched into registers while the current pair is rotated);
//     inside a block pair RB independent row pairs per inner round (cross
//     pairs A_i x B_(i+r); each block's intra pairs once per sweep), so every
//     row pair is visited exactly once per sweep;
//   * a row pair is processed by 16 lanes (4 pairs per wave, 32 per workgroup):
//     the dot product is reduced with a 4-step DPP xor-butterfly inside the
//     16-lane row (bit-identical in all 16 lanes, no LDS crossbar);
//   * squared row norms are cached in LDS and updated by  a -= t g, b += t g
//     (refreshed from the data at every block load), so a pair costs ONE dot
//     product; tan/cos/sin come from v_rcp_f64 / v_rsq_f64 + Newton steps —
//     any t is a valid rotation, only cos^2+sin^2 = 1 needs full precision.
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int JAC_NT = 512;
static constexpr int JAC_SLOTS = JAC_NT / 16;     // 32 concurrent row pairs

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row; every lane gets the bit-identical total
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]   (xor 1)
  v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]   (xor 2)
  v += dpp_mov<0x141>(v);     // row_half_mirror       (other quad of the 8)
  v += dpp_mov<0x140>(v);     // row_mirror            (other half of the 16)
  return v;
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

// Rotate LDS rows p, q (stride ldx) if they are not yet orthogonal.
// Executed by all 16 lanes of a DPP row with identical (p, q, valid).
// EPL = elements per lane (N <= 16*EPL): both rows are held in registers, so
// all 2*EPL LDS reads are in flight together and the rotation re-reads nothing.
template <int EPL>
__device__ __forceinline__ int rotate_pair(double* Xs, double* sq, int ldx, int n, int N, int p,
                                           int q, bool valid, double tol2, int l16) {
  double* xp = Xs + p * ldx;
  double* xq = Xs + q * ldx;
  double a = 0.0, b = 0.0;
  if (valid) { a = sq[p]; b = sq[q]; }
  const bool act = valid && a > 0.0 && b > 0.0;
  double u[EPL], v[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = l16 + 16 * k;
    const bool ok = act && e < N;
    u[k] = ok ? xp[e] : 0.0;
    v[k] = ok ? xq[e] : 0.0;
  }
  double g = 0.0;
#pragma unroll
  for (int k = 0; k < EPL; ++k)
    if (l16 + 16 * k < n) g = fma(u[k], v[k], g);
  g = row16_sum(g);
  if (!(act && g * g > tol2 * a * b)) return 0;
  // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)),  zeta = (b - a) / (2 g)
  const double h = b - a, ag = 2.0 * g;
  const double den = fabs(h) + sqrt(fma(h, h, ag * ag));
  double t = fabs(ag) * fast_rcp(den);
  if ((h < 0.0) != (ag < 0.0)) t = -t;
  const double cs = fast_rsqrt(fma(t, t, 1.0));
  const double sn = cs * t;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = l16 + 16 * k;
    if (e < N) {
      xp[e] = cs * u[k] - sn * v[k];
      xq[e] = sn * u[k] + cs * v[k];
    }
  }
  if (l16 == 0) { sq[p] = a - t * g; sq[q] = b + t * g; }
  return 1;
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
  static constexpr int CH = (16 * EPL + 63) / 64;
  double t[4][CH];
};

template <int EPL>
__device__ __forceinline__ void block_fetch(BlockRegs<EPL>& R, const double* X, int ld, int n,
                                            int N, int g0, int RB, int w, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + 8 * j;
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
    const int r = w + 8 * j;
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
                                            const double* Xs, int ldx, int s0, int RB, int w,
                                            int lane) {
  double t[4][BlockRegs<EPL>::CH];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + 8 * j;
    const double* src = Xs + (s0 + (r < RB ? r : 0)) * ldx;
#pragma unroll
    for (int c = 0; c < BlockRegs<EPL>::CH; ++c) {
      const int e = lane + 64 * c;
      t[j][c] = src[(e < N) ? e : 0];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = w + 8 * j;
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
// squared norms (first n entries) of LDS rows [s0, s0+cnt): one row per 16 lanes
template <int EPL>
__device__ __forceinline__ void block_norms(const double* Xs, double* sq, int ldx, int n, int s0,
                                            int cnt, int slot, int l16) {
  for (int r0 = 0; r0 < cnt; r0 += JAC_SLOTS) {
    const int r = r0 + slot;
    const double* src = Xs + (s0 + (r < cnt ? r : 0)) * ldx;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = l16 + 16 * k;
      const double v = src[(e < n) ? e : 0];
      if (e < n) acc = fma(v, v, acc);
    }
    acc = row16_sum(acc);
    if (r < cnt && l16 == 0) sq[s0 + r] = acc;
  }
}

template <int EPL>
__global__ __launch_bounds__(JAC_NT) void jacobi_rows_kernel(JacobiArgs a) {
  extern __shared__ double lds[];
  __shared__ double red[32];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int slot = tid >> 4, l16 = tid & 15;
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
  double* sq = Xs + 2 * RB * ldx;            // [2*RB]
  const double tol = sqrt((double)n) * 2.220446049250313e-16;
  const double tol2 = tol * tol;
  const int nb = (n + RB - 1) / RB;          // row blocks (last may be partial: zero rows)
  const int half = RB / 2;
  BlockRegs<EPL> pre;

  // intra-block round-robin; blocks in LDS at offsets offA (slots < half) and offB
  auto intra = [&](bool doA, bool doB) -> int {
    int rot = 0;
    for (int r = 0; r < RB - 1; ++r) {
      const bool inA = slot < half;
      const bool valid = (slot < RB) && (inA ? doA : doB);
      int p = 0, q = 0;
      if (valid) { rr_pair(RB, r, inA ? slot : slot - half, p, q); if (!inA) { p += RB; q += RB; } }
      rot |= rotate_pair<EPL>(Xs, sq, ldx, n, N, p, q, valid, tol2, l16);
      __syncthreads();
    }
    return rot;
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
      block_norms<EPL>(Xs, sq, ldx, n, 0, 2 * RB, slot, l16);
      __syncthreads();
      int rotated = intra(true, nb == 2);
      if (nb == 2) {
        for (int r = 0; r < RB; ++r) {
          const bool valid = slot < RB;
          rotated |= rotate_pair<EPL>(Xs, sq, ldx, n, N, valid ? slot : 0,
                                      valid ? RB + ((slot + r) % RB) : 0, valid, tol2, l16);
          __syncthreads();
        }
      }
      ++sweep;
      if (!block_or(rotated, red)) break;
    }
    block_store<EPL>(X, ld, n, N, 0, Xs, ldx, 0, RB, w, lane);
    block_store<EPL>(X, ld, n, N, RB, Xs, ldx, RB, RB, w, lane);
  } else {
    // ---- block-cyclic sweeps: A resident, partners B streamed (prefetched) --
    for (; sweep < a.max_sweeps;) {
      int rotated = 0;
      for (int A = 0; A + 1 < nb; ++A) {
        block_fetch<EPL>(pre, X, ld, n, N, A * RB, RB, w, lane);
        block_commit<EPL>(pre, Xs, ldx, N, 0, RB, w, lane);
        block_fetch<EPL>(pre, X, ld, n, N, (A + 1) * RB, RB, w, lane);   // first partner
        for (int Bk = A + 1; Bk < nb; ++Bk) {
          block_commit<EPL>(pre, Xs, ldx, N, RB, RB, w, lane);
          __syncthreads();
          if (Bk == A + 1) block_norms<EPL>(Xs, sq, ldx, n, 0, RB, slot, l16);
          block_norms<EPL>(Xs, sq, ldx, n, RB, RB, slot, l16);
          __syncthreads();
          if (Bk + 1 < nb)                                 // prefetch the next partner
            block_fetch<EPL>(pre, X, ld, n, N, (Bk + 1) * RB, RB, w, lane);
          if (Bk == A + 1) rotated |= intra(A == 0, true);  // each block once per sweep
          for (int r = 0; r < RB; ++r) {                    // cross pairs A_i x B_(i+r)
            const bool valid = slot < RB;
            rotated |= rotate_pair<EPL>(Xs, sq, ldx, n, N, valid ? slot : 0,
                                        valid ? RB + ((slot + r) % RB) : 0, valid, tol2, l16);
            __syncthreads();
          }
          block_store<EPL>(X, ld, n, N, Bk * RB, Xs, ldx, RB, RB, w, lane);
          __threadfence_block();
          __syncthreads();
        }
        block_store<EPL>(X, ld, n, N, A * RB, Xs, ldx, 0, RB, w, lane);
        __threadfence_block();
        __syncthreads();
      }
      ++sweep;
      if (!block_or(rotated, red)) break;
    }
  }
  __threadfence_block();
  __syncthreads();

  double smax = 0.0, smin = __builtin_inf();
  for (int i = w; i < n; i += JAC_NT / WAVE) {
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
  const size_t lds = sizeof(double) * ((size_t)2 * a.RB * a.N + 2 * a.RB);
  const int epl = (a.N + 15) / 16;
  if (epl <= 2) return launch_jacobi_t<2>(a, B, lds, st);
  if (epl <= 5) return launch_jacobi_t<5>(a, B, lds, st);
  if (epl <= 9) return launch_jacobi_t<9>(a, B, lds, st);
  if (epl <= 17) return launch_jacobi_t<17>(a, B, lds, st);
  if (epl <= 34) return launch_jacobi_t<34>(a, B, lds, st);
  return launch_jacobi_t<68>(a, B, lds, st);
}

}  // namespace blsq
