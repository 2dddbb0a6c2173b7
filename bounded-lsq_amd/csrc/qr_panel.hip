// Blocked Householder QR of tall-skinny f64 matrices, R-factor only.
//
// Replaces, for the trust-region step path, everything the reference gets
// from LAPACK on the m x n Jacobian: the thin SVD of the augmented Jacobian
// (bounded_lsq/trf.py:264-274 -> scipy.linalg.svd/gesdd) and the least-squares
// solve (bounded_lsq/dogbox.py:197 -> numpy.linalg.lstsq/gelsd) both start
// here from  [J f] = Q [R c; 0 rho]  — one pass over J.
//
// One workgroup factors one (problem, row-leaf):
//   * left-looking over 16-column panels: panel k of the SOURCE is read from
//     HBM exactly once (coalesced rows), held column-major in LDS, updated by
//     all previous block reflectors  P -= V_j (T_j^T (V_j^T P)),  then
//     factored in LDS (Householder, one fused wave-shuffle + LDS reduction per
//     column for the norm and the 15 dot products);
//   * the three GEMMs of each block-reflector application run on
//     v_mfma_f64_16x16x4_f64; the f64 accumulator layout (row = lane/16 + 4*reg)
//     is exactly the B-operand layout of k-step `reg`, so W = V^T P feeds
//     T^T W and V W straight from registers;
//   * wave w owns row tiles t = w (mod NW) (cyclic, so the shrinking active
//     row range stays balanced); cross-wave reduction of the 16x16 W goes
//     through LDS in a fixed order (deterministic results).
// Leaves produce (N x N) triangles; the same kernel merges stacked triangles
// (TSQR tree), factors the Coleman-Li augmented system [R D; E] and the
// dogbox free-column block R[:, free].
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int QR_NT = 512;
static constexpr int QR_NW = QR_NT / WAVE;

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(QR_NT) void qr_panel_kernel(QrArgs q) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int leaf = blockIdx.x;
  const int b = blockIdx.y;
  const int N = q.ncols_dev ? q.ncols_dev[b] : q.N;       // columns to factor
  const int RP = q.RP;
  const int LDP = q.LDP;
  const int NPAD = q.NPAD;
  const int ntile = RP / TILE;

  double* P = lds;                          // [16][LDP], column-major panel
  double* Wred = P + 16 * LDP;              // [NW][256] partial W tiles
  double* Gs = Wred + QR_NW * 256;          // [256]
  double* xch = Gs + 256;                   // [2][NW*16 + 16] per-column exchange
  double* taus = xch + 2 * (QR_NW * 16 + 16);  // [16]

  const long slot = (long)b * gridDim.x + leaf;
  double* Rout = q.Rout + slot * (long)NPAD * NPAD;
  if (N <= 0) return;                       // uniform per workgroup
  const int NP = (N + TILE - 1) / TILE;
  const int nA = q.F ? N - 1 : N;           // columns taken from A
  const int r0 = leaf * q.rows_per_leaf;
  int nrows = q.rowsA - r0;
  if (nrows > q.rows_per_leaf) nrows = q.rows_per_leaf;
  if (nrows < 0) nrows = 0;
  const double* A = q.A + (long)b * q.strideA + (long)r0 * q.ldA;
  const double* F = q.F ? q.F + (long)b * q.strideF + r0 : nullptr;
  double* V = q.V + slot * (long)q.NPmax * RP * 16;
  double* T = q.T + slot * (long)q.NPmax * 256;

  const int lr = lane >> 4;                 // 0..3
  const int lc = lane & 15;                 // 0..15

  for (int k = 0; k < NP; ++k) {
    // ---- 1. stage panel k of the source into LDS (single HBM read) --------
    for (int idx = tid; idx < RP * 16; idx += QR_NT) {
      const int row = idx >> 4, c = idx & 15;
      const int col = k * TILE + c;
      double val = 0.0;
      if (row < nrows) {
        if (col < nA) val = A[(long)row * q.ldA + col];
        else if (col == nA && F) val = F[row];
      }
      P[c * LDP + row] = val;
    }
    __syncthreads();

    // ---- 2. apply block reflectors 0..k-1:  P -= V_j (T_j^T (V_j^T P)) ----
    for (int j = 0; j < k; ++j) {
      const double* Vj = V + (long)j * RP * 16;
      const double* Tj = T + j * 256;
      int t0 = j + ((w - j) % QR_NW + QR_NW) % QR_NW;   // first own tile >= j
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int t = t0; t < ntile; t += QR_NW) {
        const double* vt = Vj + (long)t * 256;
        const double* pt = P + lc * LDP + t * TILE + lr;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc = mfma_f64(vt[64 * s + lane], pt[4 * s], acc);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) Wred[w * 256 + g * 64 + lane] = acc[g];
      __syncthreads();
      v4d W = {0.0, 0.0, 0.0, 0.0};
      for (int ww = 0; ww < QR_NW; ++ww) {
#pragma unroll
        for (int g = 0; g < 4; ++g) W[g] += Wred[ww * 256 + g * 64 + lane];
      }
      v4d W2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) W2 = mfma_f64(Tj[64 * s + lane], W[s], W2);
      for (int t = t0; t < ntile; t += QR_NW) {
        const double* vt = Vj + (long)t * 256 + lc * 16 + lr;
        double* pt = P + lc * LDP + t * TILE + lr;
        v4d C;
#pragma unroll
        for (int g = 0; g < 4; ++g) C[g] = pt[4 * g];
#pragma unroll
        for (int s = 0; s < 4; ++s) C = mfma_f64(-vt[4 * s], W2[s], C);
#pragma unroll
        for (int g = 0; g < 4; ++g) pt[4 * g] = C[g];
      }
      __syncthreads();                      // Wred is reused by the next j
    }

    // ---- 3. Householder-factor rows >= 16k of the panel in LDS ------------
    // Each thread owns rows tid, tid+512, tid+1024 (RP <= 1088).  Per column:
    // one fused reduction (sum x^2 and the <=15 dot products x.P[c']) through
    // wave shuffles + one LDS exchange + ONE barrier.
    const int base = k * TILE;
    const int row0 = tid, row1 = tid + QR_NT, row2 = tid + 2 * QR_NT;
#pragma unroll 1
    for (int c = 0; c < TILE; ++c) {
      const int p = base + c;               // pivot row == global column
      if (!(p < N && p < RP)) {             // padding column: H = I (uniform)
        if (tid == 0) taus[c] = 0.0;
        continue;
      }
      const double* pc = P + c * LDP;
      const bool in0 = row0 > p && row0 < RP, in1 = row1 > p && row1 < RP,
                 in2 = row2 > p && row2 < RP;
      const double x0 = in0 ? pc[row0] : 0.0;
      const double x1 = in1 ? pc[row1] : 0.0;
      const double x2 = in2 ? pc[row2] : 0.0;
      double* ex = xch + (c & 1) * (QR_NW * 16 + 16);
      {
        const double sred = wave_sum(x0 * x0 + x1 * x1 + x2 * x2);
        if (lane == 0) ex[w * 16] = sred;
      }
      for (int i = 1; c + i < TILE; ++i) {
        const double* pi = P + (c + i) * LDP;
        double acc = 0.0;
        if (in0) acc += x0 * pi[row0];
        if (in1) acc += x1 * pi[row1];
        if (in2) acc += x2 * pi[row2];
        acc = wave_sum(acc);
        if (lane == 0) ex[w * 16 + i] = acc;
      }
      const bool owner = (tid == (p % QR_NT));
      if (owner) {                          // publish the (old) pivot row
        for (int i = 0; c + i < TILE; ++i) ex[QR_NW * 16 + i] = P[(c + i) * LDP + p];
      }
      __syncthreads();
      double xn2 = 0.0;
      for (int ww = 0; ww < QR_NW; ++ww) xn2 += ex[ww * 16];
      const double alpha = ex[QR_NW * 16];
      double beta, tau, scal;
      if (xn2 == 0.0) {
        beta = alpha; tau = 0.0; scal = 0.0;
      } else {
        beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
      }
      const double v0 = scal * x0, v1 = scal * x1, v2 = scal * x2;
      double* pcw = P + c * LDP;
      if (in0) pcw[row0] = v0;
      if (in1) pcw[row1] = v1;
      if (in2) pcw[row2] = v2;
      if (owner) pcw[p] = beta;
      for (int i = 1; c + i < TILE; ++i) {
        double tot = 0.0;
        for (int ww = 0; ww < QR_NW; ++ww) tot += ex[ww * 16 + i];
        const double wv = tau * (ex[QR_NW * 16 + i] + scal * tot);
        double* pi = P + (c + i) * LDP;
        if (in0) pi[row0] -= v0 * wv;
        if (in1) pi[row1] -= v1 * wv;
        if (in2) pi[row2] -= v2 * wv;
        if (owner) pi[p] -= wv;
      }
      if (tid == 0) taus[c] = tau;
    }
    __syncthreads();

    // ---- 4. emit the R block column (rows 0..NPAD-1 of these 16 columns) --
    for (int idx = tid; idx < NPAD * 16; idx += QR_NT) {
      const int row = idx >> 4, c = idx & 15;
      const int col = base + c;
      if (col < NPAD) {
        double val = 0.0;
        if (row <= col && row < RP) val = P[c * LDP + row];
        Rout[(long)row * NPAD + col] = val;
      }
    }
    if (k == NP - 1) break;                 // no later panel needs V_k / T_k
    __syncthreads();

    // ---- 5. make V_k explicit (unit lower trapezoid, zeros above) ---------
    for (int idx = tid; idx < (base + TILE) * 16; idx += QR_NT) {
      const int row = idx >> 4, c = idx & 15;
      if (row < RP) {
        if (row < base + c) P[c * LDP + row] = 0.0;
        else if (row == base + c) P[c * LDP + row] = 1.0;
      }
    }
    __syncthreads();

    // ---- 6. T_k from G = V_k^T V_k (MFMA) + 16-step row recurrence --------
    {
      int t0 = k + ((w - k) % QR_NW + QR_NW) % QR_NW;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int t = t0; t < ntile; t += QR_NW) {
        const double* pt = P + lc * LDP + t * TILE + lr;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double e = pt[4 * s];
          acc = mfma_f64(e, e, acc);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) Wred[w * 256 + g * 64 + lane] = acc[g];
      __syncthreads();
      if (w == 0) {
        v4d G = {0.0, 0.0, 0.0, 0.0};
        for (int ww = 0; ww < QR_NW; ++ww) {
#pragma unroll
          for (int g = 0; g < 4; ++g) G[g] += Wred[ww * 256 + g * 64 + lane];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) Gs[(lr + 4 * g) * 16 + lc] = G[g];
      }
      __syncthreads();
      if (tid < TILE) {
        // thread i owns row i of T:  T[i][c] = -tau_c * sum_{q=i}^{c-1} T[i][q] G[q][c]
        const int i = tid;
        double* Ts = Wred + i * TILE;         // Wred is idle until the next panel
        for (int c = 0; c < TILE; ++c) {
          const double tc = taus[c];
          double sacc = 0.0;
          for (int qq = i; qq < c; ++qq) sacc += Ts[qq] * Gs[qq * 16 + c];
          Ts[c] = (i == c) ? tc : ((i < c) ? -tc * sacc : 0.0);
        }
        double* Tk = T + k * 256;
        for (int c = 0; c < TILE; ++c) Tk[i * 16 + c] = Ts[c];
      }
    }
    // ---- 7. spill V_k (tile-contiguous: [row][16]) for later panels -------
    {
      double* Vk = V + (long)k * RP * 16;
      for (int idx = tid; idx < RP * 16; idx += QR_NT) {
        const int row = idx >> 4, c = idx & 15;
        Vk[idx] = P[c * LDP + row];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

size_t qr_lds_bytes(int LDP) {
  return sizeof(double) * (size_t)(16 * LDP + QR_NW * 256 + 256 +
                                   2 * (QR_NW * 16 + 16) + 16);
}

hipError_t launch_qr(const QrArgs& q, int nleaf, int B, hipStream_t st) {
  const size_t lds = qr_lds_bytes(q.LDP);
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute((const void*)qr_panel_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL(qr_panel_kernel, dim3(nleaf, B), dim3(QR_NT), lds, st, q);
  return hipGetLastError();
}

}  // namespace blsq
