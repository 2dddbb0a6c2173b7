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
// One workgroup per problem; round-robin (chess tournament) pair schedule:
// n/2 independent row pairs per round, one wave per pair, the three dot
// products by wave shuffles.  Rows live in global memory (L2-resident: the
// array is <= 0.6 MB); each pair touches 2 contiguous rows (coalesced).
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int JAC_NT = 1024;
static constexpr int JAC_NW = JAC_NT / WAVE;

__global__ __launch_bounds__(JAC_NT) void jacobi_rows_kernel(JacobiArgs a) {
  __shared__ double red[32];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.N;
  const int n = N - 1;
  double* X = a.X + (long)b * a.strideX;
  const int ld = a.ld;
  if (n <= 0) {
    if (tid == 0) { a.srange[2 * b] = 0.0; a.srange[2 * b + 1] = 0.0; a.sweeps[b] = 0; }
    return;
  }
  const int np = n + (n & 1);               // even number of players
  const int rounds = np - 1;
  const int half = np / 2;
  const double tol = sqrt((double)n) * 2.220446049250313e-16;

  int sweep = 0;
  for (; sweep < a.max_sweeps; ++sweep) {
    int rotated = 0;
    for (int r = 0; r < rounds; ++r) {
      for (int pi = w; pi < half; pi += JAC_NW) {
        int p, qv;
        if (pi == 0) { p = np - 1; qv = r; }
        else { p = (r + pi) % rounds; qv = (r - pi + rounds) % rounds; }
        if (p >= n || qv >= n) continue;    // dummy player (odd n)
        double* xp = X + (long)p * ld;
        double* xq = X + (long)qv * ld;
        double aa = 0.0, bb = 0.0, gg = 0.0;
        for (int e = lane; e < n; e += WAVE) {
          const double u = xp[e], v = xq[e];
          aa += u * u; bb += v * v; gg += u * v;
        }
        aa = wave_sum(aa); bb = wave_sum(bb); gg = wave_sum(gg);
        if (aa > 0.0 && bb > 0.0 && fabs(gg) > tol * sqrt(aa * bb)) {
          const double zeta = (bb - aa) / (2.0 * gg);
          const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t);
          const double sn = cs * t;
          for (int e = lane; e < N; e += WAVE) {
            const double u = xp[e], v = xq[e];
            xp[e] = cs * u - sn * v;
            xq[e] = sn * u + cs * v;
          }
          rotated = 1;
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    if (!block_or(rotated, red)) { ++sweep; break; }
  }

  double smax = 0.0, smin = __builtin_inf();
  for (int i = w; i < n; i += JAC_NW) {
    const double* xi = X + (long)i * ld;
    double aa = 0.0;
    for (int e = lane; e < n; e += WAVE) { const double u = xi[e]; aa += u * u; }
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

hipError_t launch_jacobi(const JacobiArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(jacobi_rows_kernel, dim3(B), dim3(JAC_NT), 0, st, a);
  return hipGetLastError();
}

}  // namespace blsq
