// Middle tier of the factorisation front end: CholeskyQR2 for the problems the conditioning certificate
// keeps off the normal-equations path (kappa_2 of the equilibrated J^T J above the gate, yet far below
// 1 / eps).  Replaces the Householder TSQR tree of [J f] for them (what the reference gets from LAPACK's
// gesdd on the whole matrix, trf.py:264-274, dogbox.py:197):
//
//   first pass (already done for the certificate):  G = [J f]^T [J f],  G = R1^T R1,  Y = R1'^-T
//   second pass, this file + the Gram kernel:       W = J R^-1 = (J D) Y^T,   w_f = f - J R^-1 c
//                                                   G2 = [W w_f]^T [W w_f]   (~ identity),  G2 = R2^T R2
//   combine:                                         [J f] = Q (R2 [R c; 0 1])
//
// R1 carries the rounding of the first Gram — relative error eps kappa_2(C) — but only as a PRECONDITIONER:
// the columns of W are orthonormal to within that error, so G2 is the Gram of a matrix of condition number
// ~1 and its Cholesky factor R2 is accurate to O(eps); R2 R1 is then a triangle of [J f] of Householder
// quality (Yamamoto, Nakatsukasa, Yanagisawa, Fukaya 2015: CholeskyQR2).  The acceptance test is proven,
// not estimated: Gershgorin on G2 — every off-diagonal absolute row sum plus |g_ii - 1| at most 1/2 bounds
// kappa_2(G2) by 3 — and a problem that fails it (kappa_2(C) near 1 / eps) goes to the Householder tree.
// Cost per problem: one GEMM with the triangular n x n inverse + one more Gram = 2 m n^2 flops through
// the MFMA pipe, J read twice — against the tree's 2 m n^2 at 0.41 of the MFMA peak and 5.5x its bytes.
#include "gram_common.h"

namespace blsq {

// LDS row stride of the staged rows for the GEMM: the A operand of W = X P is X[row0 + lc][16 k + 4 s + lr]
// — sixteen ROWS per half-wave — so the stride must step the banks by 4 dwords per row: 2 mod 32 doubles.
__host__ __device__ inline int cqr2_ldx(int N) { return ((N - 2 + 31) / 32) * 32 + 2; }

// ---- z = R^-1 c and the launch mask of the second pass ---------------------------------------------
// grid = listed problems.  run[b] = n + 1 if the first Cholesky of the plain Gram went through (pivot mask 0),
// else 0 (such a problem goes to the tree).  z_k = dl_k sum_{i >= k} Y[i][k] c_i  (R^-1 = D Y^T).
__global__ __launch_bounds__(256) void cqr2_prep_kernel(Cqr2Args a, const int* pivot_mask, int* run) {
  const int b = a.list[blockIdx.x];
  const int tid = threadIdx.x;
  const int n = a.n, NPAD = a.NPAD;
  const bool ok = pivot_mask[b] == 0;
  if (tid == 0) run[b] = ok ? n + 1 : 0;
  if (!ok) return;
  const double* Y = a.Y + (long)b * NPAD * NPAD;
  const double* R = a.R1 + (long)b * NPAD * NPAD;
  const double* dl = a.dsc + (long)b * NPAD;
  for (int k = tid; k < n; k += 256) {
    // c' = R'[:, n] = c dl_n ... in stored form: R^-1 c with R = R' D^-1:  R^-1 = D R'^-1 = D Y^T, c as stored
    double acc = 0.0;
    for (int i = k; i < n; ++i) acc = fma(Y[(long)i * NPAD + k], R[(long)i * NPAD + n], acc);
    a.z[(long)b * NPAD + k] = dl[k] * acc;
  }
}

// first k-tile of the second half of a wave's MFMA stream: the smallest ks with at least half of the 8 (T0 + T1)
// MFMAs of a chunk's row tile pair in the k-tiles below it
constexpr int cqr2_split(int T0, int T1) {
  int sum = 0, ks = 0;
  while (ks < T1 && 2 * sum < 8 * (T0 + T1)) { sum += (ks < T0) ? 16 : 8; ++ks; }
  return ks;
}

// ---- W = (J D) Y^T, w_f = f - J z -------------------------------------------------------------------
// One workgroup per (row block, listed problem): the rows stream through LDS 32 at a time exactly as in the
// Gram kernels (double buffered, the next rows prefetched into registers behind the MFMA stream).  Wave W
// owns the column tiles W and NTJ - 1 - W of W: its B operands — the tiles (k, j), k <= j, of the upper
// triangular P = D Y^T, at most 17 of them — are loaded ONCE and stay in registers for all rows of the
// block; the A operands X[row][16 k + 4 s + lr] come from LDS, each fragment feeding both column tiles.
// FULL: sixteen column tiles (n = 241 .. 256) — every wave has two column tiles with W + 1 and 16 - W B tiles, no
// guard is left in the MFMA stream (the generic form spent ~2500 instructions per chunk around its 136 MFMAs).
#ifdef BLSQ_CHOL_STAMPS
__device__ long long g_cq_st[8][130][8];               // [wave][chunk][phase] of ONE workgroup (diagnostic build)
#define QST(c, i) do { if (qstp && lane == 0 && (c) < 130) g_cq_st[W][c][i] = (long long)wall_clock64(); } while (0)
int cqr2_debug_stamps(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cq_st), sizeof(g_cq_st)); }
#else
#define QST(c, i) do { } while (0)
#endif
template <int W, int NCB, bool FULL>
__device__ __forceinline__ void cqr2_wave(const Cqr2Args& a, double* lds, int b) {
  constexpr int T0 = W + 1, T1 = 16 - W;                // B tiles of column tile W / of column tile NTJ - 1 - W (at most)
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  const bool qstp = blockIdx.y == 100 && blockIdx.x == 0; (void)qstp;
  const int n = a.n, N = n + 1, NPAD = a.NPAD;
  const int NTJ = FULL ? 16 : (n + 15) / 16;
  const int LDX = cqr2_ldx(N);
  const int c0 = W, c1 = NTJ - 1 - W;                   // (wave-uniform) c0 <= c1: two tiles; c0 == c1: one; else idle
  const bool has0 = FULL || c0 < c1, has1 = FULL || c0 <= c1;   // tile c0 only when distinct from c1
  const int r_lo = blockIdx.x * a.rows_per_wg;
  int r_hi = r_lo + a.rows_per_wg;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;
  const double* Jb = a.J + (long)b * a.strideJ;
  const double* Fb = a.F + (long)b * a.strideF;
  const double* Y = a.Y + (long)b * NPAD * NPAD;
  const double* dl = a.dsc + (long)b * NPAD;

  // B tiles: B[kk = 4 s + lr][nn = lc] = P[16 k + kk][16 j + nn] = dl[16 k + kk] Y[16 j + nn][16 k + kk]
  double B0[T0][4], B1[T1][4];
  auto load_tile = [&](int k, int j, double (&dst)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = 16 * k + 4 * s + lr, col = 16 * j + lc;
      double v = 0.0;
      if (row < n && col < n && row <= col) v = dl[row] * Y[(long)col * NPAD + row];
      // (opaque to the compiler from here on: it otherwise REMATERIALISES the tile — load, scale and guards — in
      //  every chunk of the row loop instead of keeping it in its register: 177 global loads per 136 MFMAs)
      asm volatile("" : "+v"(v));
      dst[s] = v;
    }
  };
#pragma unroll
  for (int k = 0; k < T0; ++k) { if (has0) load_tile(k, c0, B0[k]); else { B0[k][0] = B0[k][1] = B0[k][2] = B0[k][3] = 0.0; } }
#pragma unroll
  for (int k = 0; k < T1; ++k) { if (has1 && k <= c1) load_tile(k, c1, B1[k]); else { B1[k][0] = B1[k][1] = B1[k][2] = B1[k][3] = 0.0; } }

  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];                        // one half of the next chunk in flight behind half of the MFMAs
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
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
      }
      if (lane == 0) X[lrow * LDX + n] = in ? fpre[rr] : 0.0;
    }
  };

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  double* zs = lds + 2 * GR_RC * LDX;                   // [n] z = R^-1 c
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {   // padding columns stay zero
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  for (int k = tid; k < n; k += GR_NT) zs[k] = a.z[(long)b * NPAD + k];
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();
  double* Wjb = a.Wj + (long)blockIdx.y * a.strideW;    // (by list position: the buffers hold the listed problems only)
  double* Wfb = a.Wf + (long)blockIdx.y * a.strideWf;
  constexpr bool LATE = W >= 4;
  v4d acc[2][2];
  // lane holds W[row0 + 16 rt + lr + 4 g][16 j + lc]
  auto store_tiles = [&](int row0_) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = row0_ + 16 * rt + lr + 4 * g;
        if (row < m) {
          if (has0 && 16 * c0 + lc < n) Wjb[(long)row * n + 16 * c0 + lc] = acc[rt][0][g];
          if (has1 && 16 * c1 + lc < n) Wjb[(long)row * n + 16 * c1 + lc] = acc[rt][1][g];
        }
      }
    }
  };
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    QST(cidx, 0);
    if (more) issue(row0 + GR_RC, 0);
    // Waves 4 .. 7 (the SIMDs' second waves) store the tiles of the PREVIOUS chunk first and compute after, waves
    // 0 .. 3 compute first and store after: the two waves of a SIMD are out of phase, the stores (and w_f) of one
    // run under the MFMAs of the other instead of all eight waves leaving the pipe idle together.
    if (LATE && cidx > 0) store_tiles(row0 - GR_RC);
    QST(cidx, 1);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { acc[rt][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[rt][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
    const double* Xa = X + lc * LDX + lr;               // A[m = lc][kk = lr] of row tile 0
    // The A fragments of k-tile k + 1 are fetched while the MFMAs of k-tile k issue (two register sets; the
    // scheduling barriers keep the compiler from hoisting every read of the chunk to the top — it spilled
    // 866 registers doing so).  Each fragment feeds both column tiles of the wave.
    double fa[2][8];
    auto fetch = [&](int k, double (&f)[8]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f[2 * s] = Xa[16 * k + 4 * s];
        f[2 * s + 1] = Xa[16 * LDX + 16 * k + 4 * s];
      }
    };
    auto ktile = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (FULL || (has1 && k <= c1)) {                  // (wave-uniform)
        if (k + 1 < T1 && (FULL || k + 1 <= c1)) fetch(k + 1, fa[(k + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double a0 = fa[k & 1][2 * s], a1 = fa[k & 1][2 * s + 1];
          if constexpr (k < T0) {
            if (FULL || has0) {
              acc[0][0] = gmfma(a0, B0[k][s], acc[0][0]);
              acc[1][0] = gmfma(a1, B0[k][s], acc[1][0]);
            }
          }
          acc[0][1] = gmfma(a0, B1[k][s], acc[0][1]);
          acc[1][1] = gmfma(a1, B1[k][s], acc[1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (has1) fetch(0, fa[0]);
    // the second half of the next rows is requested in the MIDDLE of this wave's MFMAs (k-tiles below T0 carry
    // 16 of them, the others 8): with the split at k = 8 wave 7 had 128 MFMAs before and 8 behind it, and waited
    // for those loads at the end of every chunk
    constexpr int KS = cqr2_split(T0, T1);
    static_for<0, KS>(ktile);
    QST(cidx, 2);
    if (more) { commit(row0 + GR_RC, 0, Xn); issue(row0 + GR_RC, 1); }
    QST(cidx, 3);
    if constexpr (T1 > KS) static_for<KS, T1>(ktile);
    QST(cidx, 4);
    if (!LATE) store_tiles(row0);
    QST(cidx, 5);
    // w_f = f - x^T z for the 32 rows of the chunk: waves 0 .. 3, eight rows each, 16 lanes per row
    if (!LATE) {
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int row = W * 8 + pp * 4 + lr, cg = lc;
        double sum = 0.0;
        for (int c = cg; c < n; c += 16) sum = fma(X[row * LDX + c], zs[c], sum);
        sum = row16_sum(sum);
        if (cg == 0 && row0 + row < m) Wfb[row0 + row] = X[row * LDX + n] - sum;
      }
    }
    QST(cidx, 6);
    if (more) commit(row0 + GR_RC, 1, Xn);
    __syncthreads();
    QST(cidx, 7);
  }
  if (LATE && cidx > 0) store_tiles(r_lo + (cidx - 1) * GR_RC);
}

template <int NCB, bool FULL>
__global__ __launch_bounds__(GR_NT, 2) void cqr2_apply_kernel(Cqr2Args a) {
  extern __shared__ double lds[];
  const int b = a.list[blockIdx.y];
  if (a.run[b] <= 1) return;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  switch (w) {
    case 0: cqr2_wave<0, NCB, FULL>(a, lds, b); break;
    case 1: cqr2_wave<1, NCB, FULL>(a, lds, b); break;
    case 2: cqr2_wave<2, NCB, FULL>(a, lds, b); break;
    case 3: cqr2_wave<3, NCB, FULL>(a, lds, b); break;
    case 4: cqr2_wave<4, NCB, FULL>(a, lds, b); break;
    case 5: cqr2_wave<5, NCB, FULL>(a, lds, b); break;
    case 6: cqr2_wave<6, NCB, FULL>(a, lds, b); break;
    default: cqr2_wave<7, NCB, FULL>(a, lds, b); break;
  }
}

// ---- acceptance ----------------------------------------------------------------------------------------
// grid = listed problems.  A problem is ACCEPTED iff its second Cholesky went through and Gershgorin on the
// n x n block of G2 proves kappa_2(G2) <= 3.  tree_mask[b] = 0 (accepted) or n + 1 (the Householder tree factors
// the problem).  Sums in a fixed order (the verdict decides a problem's path, hence its bits).
__global__ __launch_bounds__(256) void cqr2_accept_kernel(Cqr2Args a, const int* run, const int* pivot2,
                                                          const double* G2, int* tree_mask,
                                                          unsigned long long* accepted) {
  __shared__ double red[32];
  const int b = a.list[blockIdx.x];
  const int tid = threadIdx.x;
  const int n = a.n, NPAD = a.NPAD;
  bool ok = run[b] > 1 && pivot2[b] == 0;
  const double* G = G2 + (long)b * NPAD * NPAD;         // upper tile blocks of a symmetric matrix (diagonal tiles full)
  if (ok) {                                             // (uniform)
    double worst = 0.0;
    for (int i = tid; i < n; i += 256) {
      double sum = 0.0;
      // (16 loads in flight per pass; clamped addresses, the same additions in the same order as a plain loop)
      for (int r0 = 0; r0 < i; r0 += 16) {              // column i above the diagonal (coalesced over i)
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = G[(long)((r0 + u < i) ? r0 + u : i) * NPAD + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (r0 + u < i) sum += fabs(v[u]);
      }
      sum += fabs(G[(long)i * NPAD + i] - 1.0);
      for (int c0 = i + 1; c0 < n; c0 += 16) {          // row i to the right of it
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = G[(long)i * NPAD + ((c0 + u < n) ? c0 + u : i)];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (c0 + u < n) sum += fabs(v[u]);
      }
      worst = nanmax2(worst, sum);
    }
    worst = block_max(worst, red);
    ok = worst <= 0.5;                                  // (NaN fails) eigenvalues of G2 in [1/2, 3/2]
  }
  if (tid == 0) {
    tree_mask[b] = ok ? 0 : n + 1;
    if (ok && accepted) atomicAdd(accepted, 1ULL);
  }
}

// ---- combine: R~ | c~ = R2 [R c; 0 1] ------------------------------------------------------------------
// grid = (tile groups, listed problems), ONE WAVE PER OUTPUT TILE (i, j) of the full NT x NT image (tiles left of
// the diagonal: zeros): sum_{k = i..j} R2(i, k) R1'(k, j) by MFMA, R1' = [R c; 0 1] read from the scratch the
// first Cholesky wrote, the result into the problem's triangle slot (out of place: no ordering between tiles;
// the kernel is all latency, so the tiles are spread over as many waves as there are).
constexpr int CQ_CW = 4;                                // waves (tiles) per workgroup
__global__ __launch_bounds__(64 * CQ_CW) void cqr2_combine_kernel(Cqr2Args a, const int* tree_mask, const double* R2,
                                                                 double* Rout) {
  const int b = a.list[blockIdx.y];
  if (tree_mask[b] != 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, N = n + 1, NPAD = a.NPAD;
  const int NT = (N + 15) / 16;
  const int q = (int)blockIdx.x * CQ_CW + w;            // tile index, row-major over the full image
  if (q >= NT * NT) return;
  const int ti = q / NT, tj = q - ti * NT;
  const double* S = R2 + (long)b * NPAD * NPAD;         // R2 | c2 (rows < n)
  const double* T = a.R1 + (long)b * NPAD * NPAD;       // R | c (rows < n)
  double* O = Rout + (long)b * NPAD * NPAD;
  v4d out = v4d{0.0, 0.0, 0.0, 0.0};
  for (int k = ti; k <= tj; ++k) {
    double av[4], bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kk = 16 * k + 4 * s + lr;               // inner index
      const int ar = 16 * ti + lc;                      // A[m = lc][kk]: R2[row ar][col kk]
      av[s] = (ar < n && kk < N) ? S[(long)ar * NPAD + kk] : 0.0;
      const int bc = 16 * tj + lc;                      // B[kk][nn = lc]: R1'[row kk][col bc]
      double t = 0.0;
      if (kk < n && bc < N) t = T[(long)kk * NPAD + bc];
      else if (kk == n && bc == n) t = 1.0;
      bv[s] = t;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) out = gmfma(av[s], bv[s], out);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = 16 * ti + lr + 4 * g, col = 16 * tj + lc;
    O[(long)row * NPAD + col] = (row < n && col < N && col >= row) ? out[g] : 0.0;
  }
}

hipError_t launch_cqr2_prep(const Cqr2Args& a, int count, const int* pivot_mask, int* run, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(cqr2_prep_kernel, dim3(count), dim3(256), 0, s, a, pivot_mask, run);
  return hipGetLastError();
}

bool cqr2_supported(int m, int n) { return n > 79 && n <= 256 && m >= n; }

hipError_t launch_cqr2_apply(const Cqr2Args& a_in, int count, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  Cqr2Args a = a_in;
  a.rows_per_wg = 512;                                  // (finer than the Gram's chunks: fewer idle CUs in the last round)
  const int N = a.n + 1;
  const size_t lds = sizeof(double) * (2 * GR_RC * (size_t)cqr2_ldx(N) + (size_t)a.NPAD);
  const dim3 grid((a.m + a.rows_per_wg - 1) / a.rows_per_wg, count, 1);
  const int ncb = (a.n + 63) / 64;
#define BLSQ_CQR2_LAUNCH(CB, FL)                                                         \
  do {                                                                                   \
    static std::atomic<size_t> granted[64];                                              \
    hipError_t ge = gram_grant_lds(cqr2_apply_kernel<CB, FL>, lds, granted);             \
    if (ge != hipSuccess) return ge;                                                     \
    hipLaunchKernelGGL((cqr2_apply_kernel<CB, FL>), grid, dim3(GR_NT), lds, s, a);       \
  } while (0)
  if (ncb <= 2) BLSQ_CQR2_LAUNCH(2, false);
  else if (ncb <= 3) BLSQ_CQR2_LAUNCH(3, false);
  else if ((a.n + 15) / 16 == 16) BLSQ_CQR2_LAUNCH(4, true);
  else BLSQ_CQR2_LAUNCH(4, false);
#undef BLSQ_CQR2_LAUNCH
  return hipGetLastError();
}

hipError_t launch_cqr2_combine(const Cqr2Args& a, int count, const int* run, const int* pivot2, const double* G2,
                               const double* R2, double* Rout, int* tree_mask, unsigned long long* accepted,
                               hipStream_t s) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(cqr2_accept_kernel, dim3(count), dim3(256), 0, s, a, run, pivot2, G2, tree_mask, accepted);
  const int NT = (a.n + 1 + 15) / 16;
  hipLaunchKernelGGL(cqr2_combine_kernel, dim3((NT * NT + CQ_CW - 1) / CQ_CW, count), dim3(64 * CQ_CW), 0, s, a,
                     tree_mask, R2, Rout);
  return hipGetLastError();
}

}  // namespace blsq
