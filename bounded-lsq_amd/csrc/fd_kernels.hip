// Finite-difference Jacobians on the device for the batched drivers.
//
// The reference obtains jac='2-point' / '3-point' from a THIRD-PARTY routine:
// scipy.optimize._numdiff.approx_derivative (least_squares.py:357-365; pinned here against
// scipy 1.15.3, dense path): `_compute_absolute_step`, `_adjust_scheme_to_bounds` and
// `_dense_difference`.  This file restates those three for B problems at once:
//   fd_points   : step h_j per variable (sign-aware, bounds-aware, one-sided switching) and the
//                 perturbed points  X[b][p][:]  (P = n for '2-point', 2n for '3-point') that the
//                 caller's `fun` evaluates in ONE batched call;
//   fd_assemble : J[b][:, j] = df / dx with dx recomputed from the perturbed coordinate, so it is
//                 an exactly representable number as in `_dense_difference`.
// Compiled with -ffp-contract=off: the step formulas follow scipy operation by operation.
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

// one thread per (b, j): h and the one-sided flag
__global__ void fd_steps_kernel(int B, int n, int method, const double* x, const double* lb,
                                const double* ub, const double* rel_step, double* h,
                                unsigned char* one_sided) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * n) return;
  const int j = (int)(idx % n);
  const double x0 = x[idx], l = lb[idx], u = ub[idx];
  const double sgn = (x0 >= 0.0) ? 1.0 : -1.0;                 // 1 at x0 == 0
  // _eps_for_method (float64): EPS**0.5 and EPS**(1/3) exactly as numpy evaluates them
  const double rstep = (method == 2) ? 0x1.0000000000000p-26 : 0x1.965fea53d6e41p-18;
  const double ax = fabs(x0);
  const double dflt = rstep * sgn * ((1.0 > ax) ? 1.0 : ax);
  double hs = dflt;
  if (rel_step) {                                              // user-requested relative step
    hs = rel_step[j] * sgn * ax;
    const double dx = (x0 + hs) - x0;
    if (dx == 0.0) hs = dflt;
  }
  // _adjust_scheme_to_bounds(x0, h, num_steps = 1, '1-sided' | '2-sided', lb, ub)
  bool os;
  double ha;
  const double lower = x0 - l, upper = u - x0;
  if (method == 2) {
    os = true;
    ha = hs;
    const double xp = x0 + hs;
    const bool violated = (xp < l) || (xp > u);
    const double md = (lower > upper) ? lower : upper;
    const bool fitting = fabs(hs) <= md;
    if (violated && fitting) ha = -ha;
    if (!fitting) ha = (upper >= lower) ? upper : -lower;
  } else {
    hs = fabs(hs);
    os = false;
    ha = hs;
    const bool central = (lower >= hs) && (upper >= hs);
    if (!central) {
      if (upper >= lower) {
        const double c = 0.5 * upper;
        ha = (hs < c) ? hs : c;
      } else {
        const double c = 0.5 * lower;
        ha = -((hs < c) ? hs : c);
      }
      os = true;
      const double mind = (upper < lower) ? upper : lower;
      if (fabs(ha) <= mind) { ha = mind; os = false; }
    }
  }
  h[idx] = ha;
  one_sided[idx] = os ? 1 : 0;
}

// perturbed points: X[b][p][i] = x[b][i] (+ the step on coordinate j(p))
__global__ void fd_points_kernel(int B, int n, int method, const double* x, const double* h,
                                 const unsigned char* one_sided, double* X) {
  const int P = (method == 2) ? n : 2 * n;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * P * n) return;
  const int i = (int)(idx % n);
  const long bp = idx / n;
  const int p = (int)(bp % P);
  const long b = bp / P;
  double v = x[b * n + i];
  const int j = (method == 2) ? p : (p >> 1);
  if (i == j) {
    const double hj = h[b * n + j];
    if (method == 2) {
      v += hj;
    } else if (one_sided[b * n + j]) {
      v += (p & 1) ? 2 * hj : hj;                 // x1 = x0 + h, x2 = x0 + 2 h
    } else {
      v += (p & 1) ? hj : -hj;                    // x1 = x0 - h, x2 = x0 + h
    }
  }
  X[idx] = v;
}

// J[b][i][j] = df / dx : 32 x 32 tiles transposed through LDS (F is [b][p][i], J is [b][i][j])
__global__ __launch_bounds__(256) void fd_assemble_kernel(int B, int m, int n, int method,
                                                          const double* x, const double* h,
                                                          const unsigned char* one_sided,
                                                          const double* f0, const double* F,
                                                          double* J, const int* mask) {
  __shared__ double tile[32][33];
  const int b = blockIdx.z;
  if (mask && !mask[b]) return;
  const int P = (method == 2) ? n : 2 * n;
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  for (int jj = ty; jj < 32; jj += 8) {
    const int j = j0 + jj, i = i0 + tx;
    double val = 0.0;
    if (j < n && i < m) {
      const double x0 = x[(long)b * n + j], hj = h[(long)b * n + j];
      const double fz = f0[(long)b * m + i];
      if (method == 2) {
        const double x1 = x0 + hj;
        const double dx = x1 - x0;                 // exactly representable
        const double df = F[((long)b * P + j) * m + i] - fz;
        val = df / dx;
      } else {
        const double f1 = F[((long)b * P + 2 * j) * m + i];
        const double f2 = F[((long)b * P + 2 * j + 1) * m + i];
        if (one_sided[(long)b * n + j]) {
          const double x2 = x0 + 2 * hj;
          const double dx = x2 - x0;
          const double df = -3.0 * fz + 4 * f1 - f2;
          val = df / dx;
        } else {
          const double x1 = x0 - hj, x2 = x0 + hj;
          const double dx = x2 - x1;
          val = (f2 - f1) / dx;
        }
      }
    }
    tile[jj][tx] = val;
  }
  __syncthreads();
  for (int ii = ty; ii < 32; ii += 8) {
    const int i = i0 + ii, j = j0 + tx;
    if (i < m && j < n) J[((long)b * m + i) * n + j] = tile[tx][ii];
  }
}

hipError_t launch_fd_points(int B, int n, int method, const double* x, const double* lb,
                            const double* ub, const double* rel_step, double* X, double* h,
                            unsigned char* one_sided, hipStream_t s) {
  const long nv = (long)B * n;
  hipLaunchKernelGGL(fd_steps_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, B, n,
                     method, x, lb, ub, rel_step, h, one_sided);
  const long np = nv * ((method == 2) ? n : 2 * n);
  hipLaunchKernelGGL(fd_points_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, B, n,
                     method, x, h, one_sided, X);
  return hipGetLastError();
}

hipError_t launch_fd_assemble(int B, int m, int n, int method, const double* x, const double* h,
                              const unsigned char* one_sided, const double* f0, const double* F,
                              double* J, const int* mask, hipStream_t s) {
  hipLaunchKernelGGL(fd_assemble_kernel, dim3((m + 31) / 32, (n + 31) / 32, B), dim3(256), 0, s, B,
                     m, n, method, x, h, one_sided, f0, F, J, mask);
  return hipGetLastError();
}

}  // namespace blsq
