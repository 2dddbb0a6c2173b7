// Triangular kernels on the n x n factor R (upper triangular, row-major, stride ld,
// resident in global memory / L2), executed by ONE workgroup of TRI_NT threads.
// Vectors live in LDS.  Used by the SVD-free trust-region path (lm_kernels.hip).
#pragma once
#include "blsq_device.h"

namespace blsq {

static constexpr int TRI_NT = 256;
static constexpr int TRI_NW = TRI_NT / WAVE;

// u = R s   (one wave per row, lanes stride the columns)
__device__ __forceinline__ void tri_mv(const double* R, int n, int ld, const double* s, double* u) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = w; i < n; i += TRI_NW) {
    const double* row = R + (long)i * ld;
    double acc = 0.0;
    for (int j = i + lane; j < n; j += WAVE) acc = fma(row[j], s[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) u[i] = acc;
  }
  __syncthreads();
}

// u = R^T s  (thread per column j: sum_{i<=j} R[i][j] s_i; coalesced across threads)
__device__ __forceinline__ void tri_mtv(const double* R, int n, int ld, const double* s,
                                        double* u) {
  for (int j = threadIdx.x; j < n; j += TRI_NT) {
    double acc = 0.0;
    for (int i = 0; i <= j; ++i) acc = fma(R[(long)i * ld + j], s[i], acc);
    u[j] = acc;
  }
  __syncthreads();
}

// invd[i] = 1 / R[i][i]
__device__ __forceinline__ void tri_invdiag(const double* R, int n, int ld, double* invd) {
  for (int i = threadIdx.x; i < n; i += TRI_NT) invd[i] = 1.0 / R[(long)i * ld + i];
  __syncthreads();
}

// In place: x <- R^{-1} x.  Blocked back substitution, 16-wide blocks: the diagonal
// block is solved by lanes 0..15 of wave 0 (lane i owns row i, x_s broadcast with
// v_readlane), the part above it is updated by all threads (one row each).
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
    for (int i = tid; i < c0; i += TRI_NT) {           // rows above the block
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
    for (int j = c0 + 16 + tid; j < n; j += TRI_NT) {   // columns to the right of the block
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

__device__ __forceinline__ double tri_dot(const double* a, const double* b, int n, double* red) {
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += TRI_NT) acc = fma(a[j], b[j], acc);
  return block_sum(acc, red);
}

}  // namespace blsq
