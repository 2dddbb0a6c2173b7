// Triangular kernels on the n x n factor R (upper triangular, row-major, stride ld,
// resident in global memory / L2), executed by ONE workgroup of TRI_NT threads.
// Vectors live in LDS.  Used by the SVD-free trust-region path (lm_kernels.hip).
#pragma once
#include <type_traits>

#include "blsq_device.h"

namespace blsq {

// Sixteen LDS operands requested together and waited for ONCE: written as a plain loop the compiler emitted
// read -> wait -> fma sixteen times in a row (0.9 us per phase of a block step, tools/cert0_stamps.py).
// base: LDS byte address (per lane); element k is at base + 8 * STRIDE * k.
template <int STRIDE>
__device__ __forceinline__ void tri_lds16_issue(double (&v)[16], unsigned base) {
#define BLSQ_TRI_RD(K) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[K]) : "v"(base), "n"(8 * STRIDE * (K)));
  if constexpr (8 * STRIDE * 15 < 65536) {
    BLSQ_TRI_RD(0) BLSQ_TRI_RD(1) BLSQ_TRI_RD(2) BLSQ_TRI_RD(3) BLSQ_TRI_RD(4) BLSQ_TRI_RD(5) BLSQ_TRI_RD(6) BLSQ_TRI_RD(7)
    BLSQ_TRI_RD(8) BLSQ_TRI_RD(9) BLSQ_TRI_RD(10) BLSQ_TRI_RD(11) BLSQ_TRI_RD(12) BLSQ_TRI_RD(13) BLSQ_TRI_RD(14) BLSQ_TRI_RD(15)
  }
#undef BLSQ_TRI_RD
}
// (run-time element stride, in doubles)
__device__ __forceinline__ void tri_lds16_issue_rt(double (&v)[16], unsigned base, unsigned stride_bytes) {
#pragma unroll
  for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(v[k]) : "v"(base + stride_bytes * (unsigned)k));
}
__device__ __forceinline__ void tri_lds16_wait(double (&v)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}
__device__ __forceinline__ void tri_lds16_tie(double (&v)[16]) {       // (already waited for: pin behind that wait)
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                    "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}
__device__ __forceinline__ unsigned tri_lds_addr(const double* p) {
  return (unsigned)(unsigned long)(lptr_t*)p;
}

static constexpr int TRI_NT = 256;
static constexpr int TRI_NW = TRI_NT / WAVE;
// Every routine is a template on NT, the thread count of the calling workgroup (default TRI_NT): the fused
// Newton-round kernel of chol_kernels.hip runs them with its 512 threads.

// u = R s   (one wave per row, lanes stride the columns)
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_mv(const double* R, int n, int ld, const double* s, double* u) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int RB = 4;                     // rows per wave pass: their loads fly together
  for (int i0 = w; i0 < n; i0 += (NT / WAVE) * RB) {
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
    for (int jj = 0; i0 + lane + jj < n; jj += WAVE) {
      double rv[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * (NT / WAVE);
        const int ic = (i < n) ? i : n - 1;
        const int j = ic + lane + jj;
        rv[r] = R[(long)ic * ld + ((j < n) ? j : n - 1)];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int i = i0 + r * (NT / WAVE);
        const int j = i + lane + jj;
        if (i < n && j < n) acc[r] = fma(rv[r], s[j], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = i0 + r * (NT / WAVE);
      const double t = wave_sum(acc[r]);
      if (lane == 0 && i < n) u[i] = t;
    }
  }
  __syncthreads();
}

// u = R^T s  (thread per column j: sum_{i<=j} R[i][j] s_i; coalesced across threads)
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_mtv(const double* R, int n, int ld, const double* s,
                                        double* u) {
  for (int j = threadIdx.x; j < n; j += NT) {
    double acc = 0.0;
    // 32 rows per pass, unconditional (clamped) loads in flight together: the passes are serialised by
    // their waits, and the longest column has n rows (8 per pass: 32 round trips at n = 256)
    for (int i0 = 0; i0 <= j; i0 += 32) {
      double rv[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) rv[k] = R[(long)((i0 + k <= j) ? i0 + k : j) * ld + j];
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (i0 + k <= j) acc = fma(rv[k], s[i0 + k], acc);
    }
    u[j] = acc;
  }
  __syncthreads();
}

// invd[i] = 1 / R[i][i]
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_invdiag(const double* R, int n, int ld, double* invd) {
  for (int i = threadIdx.x; i < n; i += NT) invd[i] = 1.0 / R[(long)i * ld + i];
  __syncthreads();
}

// In place: x <- R^{-1} x.  Blocked back substitution, 16-wide blocks: the diagonal
// block is solved by lanes 0..15 of wave 0 (lane i owns row i, x_s broadcast with
// v_readlane), the part above it is updated by all threads (one row each).
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_solve_upper(const double* R, int n, int ld,
                                                const double* invd, double* x) {
  const int tid = threadIdx.x;
  const int nblk = (n + 15) / 16;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int c0 = kb * 16;
    const int bs = (n - c0 < 16) ? n - c0 : 16;
    if (tid < 64) {                       // wave 0 (all 64 lanes run; lanes >= 16 are idle copies)
      const int i = tid & 15;
      double D[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {        // clamped unconditional loads, select afterwards
        const double val = R[(long)(c0 + ((i < bs) ? i : bs - 1)) * ld + c0 + ((s < bs) ? s : bs - 1)];
        D[s] = (i < bs && s < bs && s > i) ? val : 0.0;
      }
      double r = (i < bs) ? x[c0 + i] : 0.0;
      const double iv = (i < bs) ? invd[c0 + i] : 0.0;
#pragma unroll
      for (int s = 15; s >= 0; --s) {
        const double xs = read_lane(r * iv, s);       // x_s (0 for s >= bs)
        if (i < s) r = fma(-D[s], xs, r);
      }
      if (tid < bs) x[c0 + tid] = r * iv;
    }
    __syncthreads();
    for (int i = tid; i < c0; i += NT) {           // rows above the block
      const double* row = R + (long)i * ld + c0;
      double rv[16], acc = 0.0;             // unconditional (clamped) loads: a guarded load would
#pragma unroll                              // serialise into branch + load + wait per element
      for (int s = 0; s < 16; ++s) rv[s] = row[(s < bs) ? s : bs - 1];
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = fma(rv[s], (s < bs) ? x[c0 + s] : 0.0, acc);
      x[i] -= acc;
    }
    __syncthreads();
  }
}

// In place: y <- R^{-T} y.  Blocked forward substitution.
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_solve_upper_t(const double* R, int n, int ld,
                                                  const double* invd, double* y) {
  const int tid = threadIdx.x;
  const int nblk = (n + 15) / 16;
  for (int kb = 0; kb < nblk; ++kb) {
    const int c0 = kb * 16;
    const int bs = (n - c0 < 16) ? n - c0 : 16;
    if (tid < 64) {
      const int i = tid & 15;               // row i of the lower-triangular block = column i of R's block
      double D[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double val = R[(long)(c0 + ((s < bs) ? s : bs - 1)) * ld + c0 + ((i < bs) ? i : bs - 1)];
        D[s] = (i < bs && s < bs && s < i) ? val : 0.0;
      }
      double r = (i < bs) ? y[c0 + i] : 0.0;
      const double iv = (i < bs) ? invd[c0 + i] : 0.0;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double ys = read_lane(r * iv, s);
        if (i > s) r = fma(-D[s], ys, r);
      }
      if (tid < bs) y[c0 + tid] = r * iv;
    }
    __syncthreads();
    for (int j = c0 + 16 + tid; j < n; j += NT) {   // columns to the right of the block
      double rv[16], acc = 0.0;             // 16 rows = 16 cache lines: all loads in flight together
#pragma unroll
      for (int s = 0; s < 16; ++s) rv[s] = R[(long)(c0 + ((s < bs) ? s : bs - 1)) * ld + j];
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = fma(rv[s], (s < bs) ? y[c0 + s] : 0.0, acc);
      y[j] -= acc;
    }
    __syncthreads();
  }
}

// ---- the same two solves with the operands prefetched by LDS-DMA ---------------------------
// R does not fit any cache level for a whole batch, so each 16-wide block step of the solves
// above pays HBM / MALL round trips for its diagonal block and its panel.  Here the operands of
// block step kb-1 (kb+1) stream global -> LDS (global_load_lds, no VGPR staging) while step kb
// computes from LDS; the barriers inside the loop are LDS-only so the DMA stays in flight.
// buf: 2 * 16 * ld doubles of LDS.  R rows must be 16-byte aligned (ld % 2 == 0).

// rows 0 .. c0+15, columns c0 .. c0+15  ->  dst[row * 16 + col - c0]
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_pf_issue_upper(const double* R, int ld, int c0, double* dst) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nrow = c0 + 16;                              // multiple of 16: whole 8-row DMA pieces
  // A panel row is 128 bytes: one row per thread, every thread at the same column, is a 32-way bank conflict in the
  // plain row-major image (1.2 us of a block step at 240 rows).  The DMA writes 1 KB of LDS per instruction in lane
  // order, but WHICH 16-byte piece a lane fetches is free: piece (row r, column pair c) of the 8-row block g goes to
  // slot ((r ^ (g & 1)) * 8 + (c ^ r)) of the block — sixteen consecutive rows then hit sixteen different 4-bank
  // groups at every column pair (tri_pf_upper_read16 below undoes it).
  for (int r0 = w * 8; r0 < nrow; r0 += (NT / WAVE) * 8) {
    const int g1 = (r0 >> 3) & 1;
    const int r = (lane >> 3) ^ g1, c = (lane & 7) ^ r;
    glds16(R + c0, (unsigned)((r0 + r) * ld + 2 * c) * 8u, dst + r0 * 16);
  }
}
// the sixteen entries of panel row `row` (staged by tri_pf_issue_upper) as eight 16-byte reads: issue, then
// tri_pf_upper_wait (the registers hold nothing before it), which also unpacks into v
typedef double tri_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tri_pf_upper_issue(tri_v2d (&t)[8], const double* b, int row) {
  const int r = row & 7, g1 = (row >> 3) & 1;
  const unsigned blk = tri_lds_addr(b) + 8u * (unsigned)((row & ~7) * 16) + 128u * (unsigned)(r ^ g1);
#define BLSQ_TRI_RD2(C) asm volatile("ds_read_b128 %0, %1" : "=v"(t[C]) : "v"(blk + 16u * (unsigned)((C) ^ r)));
  BLSQ_TRI_RD2(0) BLSQ_TRI_RD2(1) BLSQ_TRI_RD2(2) BLSQ_TRI_RD2(3) BLSQ_TRI_RD2(4) BLSQ_TRI_RD2(5) BLSQ_TRI_RD2(6) BLSQ_TRI_RD2(7)
#undef BLSQ_TRI_RD2
}
__device__ __forceinline__ void tri_pf_upper_wait(tri_v2d (&t)[8], double (&v)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
  for (int c = 0; c < 8; ++c) { v[2 * c] = t[c][0]; v[2 * c + 1] = t[c][1]; }
}
// rows c0 .. c0+15, columns c0 .. ld-1  ->  dst[s * (ld - c0) + col - c0]
template <int NT = TRI_NT>
__device__ __forceinline__ void tri_pf_issue_lower(const double* R, int ld, int c0, double* dst) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int L = ld - c0;                                 // multiple of 16
  const int half = L >> 1;                               // 16-byte pieces per row
  const int total = 16 * half;                           // multiple of 64
  for (int i0 = w * 64; i0 < total; i0 += (NT / WAVE) * 64) {
    const int idx = i0 + lane;
    const int srow = idx / half, off = idx - srow * half;
    glds16(R + c0, (unsigned)((c0 + srow) * ld + 2 * off) * 8u, dst + i0 * 2);
  }
}

// CMP: the COMPARISON matrix M(R) instead of R (diagonal as it is, off-diagonal entries -|r_ij|): for a non-negative
// right-hand side the solution is entrywise >= |R^-1| rhs (Higham, ASNA 8.2) — the certificate's stage 0.
template <int NT = TRI_NT, bool CMP = false>
__device__ __forceinline__ void tri_solve_upper_pf(const double* R, int n, int ld,
                                                   const double* invd, double* x, double* buf) {
  const int tid = threadIdx.x;
  const int nblk = (n + 15) / 16;
  const int bsz = 16 * ld;
  int cur = 0;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  tri_pf_issue_upper<NT>(R, ld, (nblk - 1) * 16, buf);
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int c0 = kb * 16;
    const int bs = (n - c0 < 16) ? n - c0 : 16;
    const double* b = buf + cur * bsz;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                       // every wave's pieces have landed
    if (kb > 0) tri_pf_issue_upper<NT>(R, ld, c0 - 16, buf + (cur ^ 1) * bsz);
    if (tid < 64) {                                      // wave 0 (lanes >= 16 are idle copies)
      const int i = tid & 15;
      double D[16], bv[16];
      tri_v2d bt[8];
      tri_pf_upper_issue(bt, b, c0 + i);
      double r = (i < bs) ? x[c0 + i] : 0.0;
      const double iv = (i < bs) ? invd[c0 + i] : 0.0;
      tri_pf_upper_wait(bt, bv);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double val = CMP ? -fabs(bv[s]) : bv[s];
        D[s] = (i < bs && s < bs && s > i) ? val : 0.0;
      }
#pragma unroll
      for (int s = 15; s >= 0; --s) {
        const double xs = read_lane(r * iv, s);
        if (i < s) r = fma(-D[s], xs, r);
      }
      if (tid < bs) x[c0 + tid] = r * iv;
    }
    lds_barrier();
    if (tid < c0) {                                      // rows above the block
      double xv[16];
      tri_lds16_issue<1>(xv, tri_lds_addr(x) + 8u * (unsigned)c0);
      for (int i = tid; i < c0; i += NT) {
        double rv[16];
        tri_v2d rt[8];
        tri_pf_upper_issue(rt, b, i);
        tri_pf_upper_wait(rt, rv);
        tri_lds16_tie(xv);
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = fma(CMP ? -fabs(rv[s]) : rv[s], (s < bs) ? xv[s] : 0.0, acc);
        x[i] -= acc;
      }
    }
    lds_barrier();
    cur ^= 1;
  }
}

template <int NT = TRI_NT, bool CMP = false>
__device__ __forceinline__ void tri_solve_upper_t_pf(const double* R, int n, int ld,
                                                     const double* invd, double* y, double* buf) {
  const int tid = threadIdx.x;
  const int nblk = (n + 15) / 16;
  const int bsz = 16 * ld;
  int cur = 0;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  tri_pf_issue_lower<NT>(R, ld, 0, buf);
  for (int kb = 0; kb < nblk; ++kb) {
    const int c0 = kb * 16;
    const int bs = (n - c0 < 16) ? n - c0 : 16;
    const int L = ld - c0;
    const double* b = buf + cur * bsz;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (kb + 1 < nblk) tri_pf_issue_lower<NT>(R, ld, c0 + 16, buf + (cur ^ 1) * bsz);
    if (tid < 64) {
      const int i = tid & 15;               // row i of the lower-triangular block = column i of R's block
      double D[16], bv[16];
      tri_lds16_issue_rt(bv, tri_lds_addr(b) + 8u * (unsigned)i, 8u * (unsigned)L);
      double r = (i < bs) ? y[c0 + i] : 0.0;
      const double iv = (i < bs) ? invd[c0 + i] : 0.0;
      tri_lds16_wait(bv);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double val = CMP ? -fabs(bv[s]) : bv[s];
        D[s] = (i < bs && s < bs && s < i) ? val : 0.0;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double ys = read_lane(r * iv, s);
        if (i > s) r = fma(-D[s], ys, r);
      }
      if (tid < bs) y[c0 + tid] = r * iv;
    }
    lds_barrier();
    if (c0 + 16 + tid < n) {                             // columns to the right of the block
      double yv[16];
      tri_lds16_issue<1>(yv, tri_lds_addr(y) + 8u * (unsigned)c0);
      for (int j = c0 + 16 + tid; j < n; j += NT) {
        double cv[16];
        tri_lds16_issue_rt(cv, tri_lds_addr(b) + 8u * (unsigned)(j - c0), 8u * (unsigned)L);
        tri_lds16_wait(cv);
        tri_lds16_tie(yv);
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = fma(CMP ? -fabs(cv[s]) : cv[s], (s < bs) ? yv[s] : 0.0, acc);
        y[j] -= acc;
      }
    }
    lds_barrier();
    cur ^= 1;
  }
}

template <int NT = TRI_NT>
__device__ __forceinline__ double tri_dot(const double* a, const double* b, int n, double* red) {
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += NT) acc = fma(a[j], b[j], acc);
  return block_sum(acc, red);
}

}  // namespace blsq
