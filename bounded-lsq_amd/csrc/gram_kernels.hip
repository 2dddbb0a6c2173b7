// Normal-equations fast path of the Jacobian factorisation (north_star: "MFMA ... for the
// tall-skinny J^T J"; reference call sites: trf.py:244-252 / dogbox.py:197 need R and Q^T f of
// [J f] only through R^T R and R^T (Q^T f)).
//
//   G = [J f]^T [J f]            one streaming pass over J, FP64 MFMA, upper tile blocks only
//   C = D G D,  D = diag(1/||J_j||)                    (unit diagonal: column equilibration)
//   C = R'^T R'                  blocked Cholesky, 16x16 tiles, MFMA Schur updates
//   [R z; 0 rho] = R' D^-1       the same triangle the Householder TSQR tree produces (up to row
//                                signs, which nothing downstream depends on)
//
// Half the flops of Householder QR, no reflector traffic, no per-panel latency chain — but the
// error of R grows with kappa(J D)^2 instead of kappa(J D).  The path is therefore GATED per
// problem by a PROVEN upper bound on the condition number of the equilibrated system that is
// actually solved (gram_cond_kernel below: explicit inverse of the Cholesky factor, exact 1- and
// inf-norms); a problem whose bound exceeds GRAM_K2_MAX is handed to the Householder tree through a
// launch mask.
// Non-finite input, zero columns, rank deficiency and m < n all fail the gate by construction.
#include <atomic>
#include <stdlib.h>
#include <type_traits>

#include "blsq_device.h"
#include "blsq_kernels.h"
#include "tri_ops.h"
#include "chol16.h"

namespace blsq {

static constexpr int GR_NT = 512;
static constexpr int GR_NW = GR_NT / WAVE;
static constexpr int GR_RC = 32;          // rows per staged chunk (8 MFMA k-steps)
static constexpr int REG_NW = 4;          // one-wave-per-problem kernels (N <= 80): problems (waves) per workgroup —
static constexpr int REG_NT = REG_NW * WAVE;   // 1024 problems spread over 256 workgroups instead of 128
static constexpr double GRAM_SMIN = GRAM_SMIN_PROVEN;   // early reject: a pivot of R' below what the certificate could accept

template <class K>
static hipError_t gram_grant_lds(K kernel, size_t bytes, std::atomic<size_t>* granted_dev) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t>& granted = granted_dev[dev & 63];
  if (bytes <= granted.load(std::memory_order_acquire)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e == hipSuccess) granted.store(bytes, std::memory_order_release);
  return e;
}

__device__ __forceinline__ v4d gmfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// LDS byte address of a pointer into the dynamic LDS array, and an explicit 8-byte LDS read whose
// completion the CALLER waits for (counted s_waitcnt lgkmcnt)
__device__ __forceinline__ unsigned lds_addr(const double* p) {
  return (unsigned)(unsigned long)(lptr_t*)p;
}
__device__ __forceinline__ void lds_read64(double& dst, unsigned byte_addr) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(byte_addr));
}

__host__ __device__ inline int gram_ldx(int NT) { return NT * 16 + ((NT & 1) ? 0 : 16); }

// ---- G = [J f]^T [J f] ---------------------------------------------------------------------
// One workgroup per (row chunk, problem).  The chunk streams through LDS 32 rows at a time
// (double-buffered, next rows prefetched into registers during the MFMA burst); the NT (NT+1)/2
// upper 16x16 output tiles are dealt to the 8 waves in contiguous runs of the row-major tile
// order and stay in accumulators for the whole pass.  Both MFMA operands of tile (i, j) are the
// same fragment pattern X[4 s + lr][16 c + lc] (c = i for A, c = j for B), read from LDS; the LDS
// row stride is an odd multiple of 16 doubles so the two rows a half-wave reads hit disjoint banks.
template <int SLOTS, int NCB>
__global__ __launch_bounds__(GR_NT, (SLOTS <= 8 ? 4 : 2)) void gram_kernel(GramArgs a) {
  extern __shared__ double lds[];
  const int b = blockIdx.y;
  if (a.mask && a.mask[b] <= 1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, N = n + 1, NT = (N + 15) / 16;
  const int LDX = gram_ldx(NT);
  // The MFMA tiles cover J^T J only.  The rhs column (J^T f, f^T f) would cost a whole tile column
  // — 17 of 153 tiles at n = 256 — for one useful column per tile; it is accumulated beside the
  // MFMA stream by the vector ALUs from the same staged rows instead (32 FMAs per thread and chunk).
  // (Only where that saves a tile slot per wave — a.rhs_valu, decided by launch_gram.)
  const int NTJ = a.rhs_valu ? (n + 15) / 16 : NT;
  const int ntile = NTJ * (NTJ + 1) / 2;
  // Small batches: the tiles of a problem are split over gridDim.z workgroups (tile groups), each
  // streaming the same rows for its share of the tiles — a tile still sees the same k-steps in
  // the same order, so the result does not depend on the split.
  const int tpg = (ntile + (int)gridDim.z - 1) / (int)gridDim.z;   // tiles per group
  const int q_lo = (int)blockIdx.z * tpg;
  const int q_hi = (q_lo + tpg < ntile) ? q_lo + tpg : ntile;
  const int per = (tpg + GR_NW - 1) / GR_NW;
  const bool rhs_here = a.rhs_valu && blockIdx.z == 0;  // one group accumulates the rhs column
  // rows of this chunk
  const int r_lo = blockIdx.x * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;                                   // exclusive row limit
  const double* Jb = a.J + (long)b * a.strideJ;
  const double* Fb = a.F + (long)b * a.strideF;

  // tile table (wave-uniform): slot t -> (ti, tj); idle slots recompute tile (0, 0) and are
  // not stored (no branch inside the MFMA loop)
  int ti[SLOTS], tj[SLOTS];
  bool tv[SLOTS];
  {
    int i = 0, rem = q_lo + w * per;
    while (i < NTJ && rem >= NTJ - i) { rem -= NTJ - i; ++i; }
    int j = i + rem;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
      const bool valid = (t < per) && (i < NTJ) && (q_lo + w * per + t < q_hi);
      tv[t] = valid;
      ti[t] = valid ? 16 * i : 0;
      tj[t] = valid ? 16 * j : 0;
      ++j;
      if (j >= NTJ) { ++i; j = i; }
    }
  }

  // Staging: wave w owns rows w, w + 8, w + 16, w + 24 of a chunk, lanes run along the row (NCB
  // blocks of 64 columns of J; the rhs f is column n, written by lane 0).  The rows of the NEXT
  // chunk are prefetched into registers in two halves, each behind half of the MFMA burst.
  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];
  auto issue = [&](int row0, int h) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int row = row0 + w + GR_NW * (HR * h + rr);
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
  // rhs column in registers (a.rhs_valu): a wave has the rows it stages in registers, one column
  // per lane and 64-column block, so it adds its rows' share of J^T f / f^T f before committing
  // them (rows w, w + 8, ... in order); the eight per-wave partials are added in wave order at
  // the end.  gram16_wave below does exactly the same, so the two kernels agree bit for bit.
  double gf[NCB], gff = 0.0;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) gf[cb] = 0.0;
  auto commit = [&](int row0, int h, double* X) {
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int lrow = w + GR_NW * (HR * h + rr);
      const bool in = row0 + lrow < m;
      const double fv = in ? fpre[rr] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
        if (rhs_here) gf[cb] = fma(pre[rr][cb], fv, gf[cb]);   // (columns >= n: clamped loads, never stored)
      }
      if (rhs_here) gff = fma(fv, fv, gff);
      if (lane == 0) X[lrow * LDX + n] = fv;
    }
  };

  v4d acc[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  // padding columns (n, LDX) of both buffers are zero for the whole pass
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    const unsigned xb = lds_addr(X) + 8u * (unsigned)(lr * LDX + lc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (more) issue(row0 + GR_RC, h);
      // Operand fragments are fetched TWO MFMAs ahead of their use: the LDS round trip under
      // load is longer than one MFMA of each of the SIMD's two waves.  The reads and the counted
      // waits are explicit (the compiler would otherwise fold the stages back into one register
      // set and wait for every read right before its MFMA).
      constexpr int KS = GR_RC / 4 / 2;                 // k-steps per half
      constexpr int TOT = KS * SLOTS;
      double fa[3], fb[3];
      auto fetch = [&](int q) {
        const unsigned so = 8u * (unsigned)((KS * h + q / SLOTS) * 4 * LDX);
        lds_read64(fa[q % 3], xb + so + 8u * (unsigned)ti[q % SLOTS]);
        lds_read64(fb[q % 3], xb + so + 8u * (unsigned)tj[q % SLOTS]);
      };
      fetch(0);
      fetch(1);
#pragma unroll
      for (int q = 0; q < TOT; ++q) {
        if (q + 2 < TOT) {
          fetch(q + 2);
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[q % 3]), "+v"(fb[q % 3]));
        } else if (q + 1 < TOT) {
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[q % 3]), "+v"(fb[q % 3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[q % 3]), "+v"(fb[q % 3]));
        }
        acc[q % SLOTS] = gmfma(fa[q % 3], fb[q % 3], acc[q % SLOTS]);
      }
      if (more) commit(row0 + GR_RC, h, Xn);
    }
    __syncthreads();
  }

  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    if (tv[t]) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        G[(long)(ti[t] + lr + 4 * g) * a.NPAD + tj[t] + lc] = acc[t][g];
    }
  }
  if (!rhs_here) return;
  // rhs column: the eight per-wave partials, added in wave order (the last barrier is behind us)
  constexpr int GW = 64 * NCB;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) lds[w * GW + 64 * cb + lane] = gf[cb];
  if (lane == 0) lds[GR_NW * GW + w] = gff;
  __syncthreads();
  for (int c = tid; c < n; c += GR_NT) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[ww * GW + c];
    G[(long)c * a.NPAD + n] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[GR_NW * GW + ww];
    G[(long)n * a.NPAD + n] = sum;
  }
  // the rest of the rhs tile column of the slot: padding columns (n, 16 NT) stay zero
  for (int e = tid; e < a.NPAD * (16 * NT - N); e += GR_NT) {
    const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

// ---- 16 column tiles (n = 241 .. 256): static tile rows per wave -------------------------------
// The generic kernel above keeps its tile table in registers, so every MFMA fetches both operand
// fragments from LDS (2 ds_read_b64 per MFMA) — nothing tells the compiler that two tiles share one.
// Here the assignment is a compile-time function of the wave: wave W owns tile rows W and 15 - W of
// the upper triangle, (16 - W) + (W + 1) = 17 tiles for every wave.  Both operands of tile (i, j) are
// the fragment pattern X[4 s + lr][16 c + lc] with c = i resp. c = j, and all of a wave's tiles have
// i, j >= W: a k-step needs the 16 - W fragments of column tiles W .. 15 ONCE (9 .. 16 reads for 17
// MFMAs instead of 34), held in two register sets so that the reads of k-step s + 1 fly behind the
// MFMAs of k-step s.  Each wave runs its own specialisation of the row loop (wave-uniform switch);
// the workgroup barriers are the same instruction in every specialisation.
// The rhs column (J^T f, f^T f) at n = 256 costs no LDS pass either: a wave has the rows it stages in
// registers (one column per lane and 64-column block), so it accumulates its rows' contribution
// there before committing them; the eight per-wave partials are added in a fixed order at the end.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int OFF>
__device__ __forceinline__ void lds_read64_off(double& dst, unsigned byte_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(byte_addr), "n"(OFF));
}

// PAIR: one workgroup takes BOTH row chunks of a problem of two chunks, one after the other: at the
// chunk boundary the accumulators go to the output slot and restart from zero, at the end the first
// chunk's tiles are read back and added — (0 + P0) + P1, exactly what gram_reduce_kernel computes
// from two partial Grams, without writing the second one, reading both and a launch in between.
template <int W, bool RHS, bool PAIR>
__device__ __forceinline__ void gram16_wave(const GramArgs& a, double* lds) {
  constexpr int LDX = 272, NCB = 4;
  constexpr int NF = 16 - W;                            // fragments per k-step: column tiles W .. 15
  constexpr int N0 = 16 - W, N1 = W + 1;                // tiles of tile row W / of tile row 15 - W
  constexpr int A1 = 15 - 2 * W;                        // fragment index of tile row 15 - W
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, N = n + 1;
  const int r_lo = PAIR ? 0 : blockIdx.x * a.rows_per_chunk;
  int r_hi = PAIR ? a.m : r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;
  const int boundary = a.rows_per_chunk;                // (PAIR) first row of the second chunk
  const double* Jb = a.J + (long)b * a.strideJ;
  const double* Fb = a.F + (long)b * a.strideF;

  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];
  double gf[NCB] = {0.0, 0.0, 0.0, 0.0}, gff = 0.0;     // this wave's rows of J^T f / f^T f
  // (PAIR) at the chunk boundary the first chunk's sums are parked in the scratch slot of the problem
  double* scr = PAIR ? a.Gscr + (long)b * 2 * a.NPAD * a.NPAD : nullptr;
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
    if (PAIR && RHS && h == 0 && row0 == boundary) {    // (uniform) the rows of the second chunk start here
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) { scr[W * 256 + 64 * cb + lane] = gf[cb]; gf[cb] = 0.0; }
      if (lane == 0) scr[GR_NW * 256 + W] = gff;
      gff = 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < HR; ++rr) {
      const int lrow = W + GR_NW * (HR * h + rr);
      const bool in = row0 + lrow < m;
      const double fv = in ? fpre[rr] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
        if (RHS) gf[cb] = fma(pre[rr][cb], fv, gf[cb]);   // (columns >= n: clamped loads, never stored)
      }
      if (RHS) gff = fma(fv, fv, gff);
      else if (lane == 0) X[lrow * LDX + n] = fv;
    }
  };

  v4d acc0[N0], acc1[N1];
#pragma unroll
  for (int t = 0; t < N0; ++t) acc0[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < N1; ++t) acc1[t] = v4d{0.0, 0.0, 0.0, 0.0};

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {   // padding columns stay zero
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  if (RHS) {                                            // the f column is not staged: keep it defined
    for (int r = tid; r < 2 * GR_RC; r += GR_NT) lds[r * LDX + n] = 0.0;
  }
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();

  double fr[2][NF];
  // one asm per k-step ties every fragment register of a set to the counted wait
  auto wait_set = [&](double (&f)[NF]) {
    if constexpr (NF == 16)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
    else if constexpr (NF == 15)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]), "+v"(f[13]), "+v"(f[14]));
    else if constexpr (NF == 14)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]), "+v"(f[13]));
    else if constexpr (NF == 13)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]),
                   "+v"(f[12]));
    else if constexpr (NF == 12)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
    else if constexpr (NF == 11)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]));
    else if constexpr (NF == 10)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]));
  };

  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
  auto tile_ptr0 = [&](int t, int g) { return G + (long)(16 * W + lr + 4 * g) * a.NPAD + 16 * (W + t) + lc; };
  auto tile_ptr1 = [&](int t, int g) { return G + (long)(16 * (15 - W) + lr + 4 * g) * a.NPAD + 16 * (15 - W + t) + lc; };
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    if (PAIR && row0 == boundary) {                     // first chunk done: park its tiles in the output slot
#pragma unroll
      for (int t = 0; t < N0; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *tile_ptr0(t, g) = acc0[t][g];
        acc0[t] = v4d{0.0, 0.0, 0.0, 0.0};
      }
#pragma unroll
      for (int t = 0; t < N1; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *tile_ptr1(t, g) = acc1[t][g];
        acc1[t] = v4d{0.0, 0.0, 0.0, 0.0};
      }
    }
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    const unsigned xb = lds_addr(X) + 8u * (unsigned)(lr * LDX + lc);
    // fragments of the chunk's first k-step (its latency is exposed once per chunk)
    static_for<0, NF>([&](auto ic) {
      constexpr int c = decltype(ic)::value;
      lds_read64_off<8 * 16 * (W + c)>(fr[0][c], xb);
    });
    static_for<0, 8>([&](auto is) {
      constexpr int s = decltype(is)::value;            // k-step of the chunk
      constexpr int cur = s & 1, nxt = cur ^ 1;
      if constexpr (s == 0 || s == 4) {
        if (more) issue(row0 + GR_RC, s / 4);
      }
      wait_set(fr[cur]);
      // 17 MFMAs; the next k-step's fragments are requested two per MFMA behind the first ones
      static_for<0, N0 + N1>([&](auto it) {
        constexpr int t = decltype(it)::value;
        if constexpr (t < N0) acc0[t] = gmfma(fr[cur][0], fr[cur][t], acc0[t]);
        else acc1[t - N0] = gmfma(fr[cur][A1], fr[cur][A1 + (t - N0)], acc1[t - N0]);
        if constexpr (s < 7) {
          if constexpr (2 * t < NF)
            lds_read64_off<8 * (4 * (s + 1) * LDX + 16 * (W + 2 * t))>(fr[nxt][2 * t], xb);
          if constexpr (2 * t + 1 < NF)
            lds_read64_off<8 * (4 * (s + 1) * LDX + 16 * (W + 2 * t + 1))>(fr[nxt][2 * t + 1], xb);
        }
      });
      if constexpr (s == 3 || s == 7) {
        if (more) commit(row0 + GR_RC, s / 4, Xn);
      }
    });
    __syncthreads();
  }

  const bool two = PAIR && m > boundary;                // (a second chunk was accumulated)
  if (two) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this lane's own stores of the first chunk)
#pragma unroll
  for (int t = 0; t < N0; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double* q = tile_ptr0(t, g);
      *q = two ? (0.0 + *q) + acc0[t][g] : acc0[t][g];
    }
#pragma unroll
  for (int t = 0; t < N1; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double* q = tile_ptr1(t, g);
      *q = two ? (0.0 + *q) + acc1[t][g] : acc1[t][g];
    }
  if (!RHS) return;
  // rhs column: the eight per-wave partials, added in wave order (the last barrier is behind us);
  // PAIR: the two chunks' columns one after the other, then (0 + first) + second
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) lds[W * 256 + 64 * cb + lane] = two ? scr[W * 256 + 64 * cb + lane] : gf[cb];
  if (lane == 0) lds[GR_NW * 256 + W] = two ? scr[GR_NW * 256 + W] : gff;
  __syncthreads();
  double sum = 0.0, sumf = 0.0;
  if (tid < n) {
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[ww * 256 + tid];
  }
  if (tid == 0) {
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sumf += lds[GR_NW * 256 + ww];
  }
  if (two) {
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) lds[W * 256 + 64 * cb + lane] = gf[cb];
    if (lane == 0) lds[GR_NW * 256 + W] = gff;
    __syncthreads();
    double s2 = 0.0, s2f = 0.0;
    if (tid < n) {
#pragma unroll
      for (int ww = 0; ww < GR_NW; ++ww) s2 += lds[ww * 256 + tid];
    }
    if (tid == 0) {
#pragma unroll
      for (int ww = 0; ww < GR_NW; ++ww) s2f += lds[GR_NW * 256 + ww];
    }
    sum = (0.0 + sum) + s2; sumf = (0.0 + sumf) + s2f;
  }
  if (tid < n) G[(long)tid * a.NPAD + n] = sum;
  if (tid == 0) G[(long)n * a.NPAD + n] = sumf;
  const int NT = (N + 15) / 16;
  for (int e = tid; e < a.NPAD * (16 * NT - N); e += GR_NT) {   // padding columns of the rhs tile column
    const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

template <bool RHS, bool PAIR>
__global__ __launch_bounds__(GR_NT, 2) void gram16_kernel(GramArgs a) {
  extern __shared__ double lds[];
  if (a.mask && a.mask[blockIdx.y] <= 1) return;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  switch (w) {
    case 0: gram16_wave<0, RHS, PAIR>(a, lds); break;
    case 1: gram16_wave<1, RHS, PAIR>(a, lds); break;
    case 2: gram16_wave<2, RHS, PAIR>(a, lds); break;
    case 3: gram16_wave<3, RHS, PAIR>(a, lds); break;
    case 4: gram16_wave<4, RHS, PAIR>(a, lds); break;
    case 5: gram16_wave<5, RHS, PAIR>(a, lds); break;
    case 6: gram16_wave<6, RHS, PAIR>(a, lds); break;
    default: gram16_wave<7, RHS, PAIR>(a, lds); break;
  }
}

// ---- 8 column tiles (n = 113 .. 128): static tile rows per wave, k-steps split over two groups ----
// The scheme of gram16_kernel for half the width: the 36 upper tiles of an 8 x 8 tile grid are four
// pairs of tile rows (P, 7 - P) of 9 tiles each — four waves' worth.  The other four waves take the
// same tiles for the OTHER k-steps: wave W = P + 4 K works on k-steps s with s % 2 == K (K = 0, 1), so
// every wave has 9 MFMAs per k-step it owns and 8 - P operand fragments to fetch for them (the
// generic kernel runs these widths as 8 slots per wave and k-step with two reads per MFMA and idle
// slots recomputing a tile).  A tile is therefore the sum of two partial accumulations — even and odd
// k-steps — added in that order at the end (through LDS, once per row chunk).  This order is the
// DEFINITION of the result for these widths: the kernel serves every batch size (no tile groups), so
// a problem's bits still do not depend on its batch.  Row chunks, staging and the rhs column are
// those of the generic kernel.
template <int W, bool RHS>
__device__ __forceinline__ void gram8_wave(const GramArgs& a, double* lds) {
  constexpr int NTJ = 8, NCB = 2;
  constexpr int P = W % 4, K = W / 4;
  constexpr int NF = NTJ - P;                           // fragments per k-step: column tiles P .. 7
  constexpr int N0 = NTJ - P, N1 = P + 1;               // tiles of tile row P / of tile row 7 - P
  constexpr int A1 = NTJ - 1 - 2 * P;                   // fragment index of tile row 7 - P
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n, N = n + 1;
  const int NT = (N + 15) / 16;                        // 8 or 9
  constexpr int LDX = 144;                              // gram_ldx(8) == gram_ldx(9)
  const int r_lo = blockIdx.x * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const int m = r_hi;
  const double* Jb = a.J + (long)b * a.strideJ;
  const double* Fb = a.F + (long)b * a.strideF;

  constexpr int HR = GR_RC / GR_NW / 2;                 // rows per wave per half chunk
  double pre[HR][NCB], fpre[HR];
  double gf[NCB] = {0.0, 0.0}, gff = 0.0;
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
      const double fv = in ? fpre[rr] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int col = lane + 64 * cb;
        if (col < n) X[lrow * LDX + col] = in ? pre[rr][cb] : 0.0;
        if (RHS) gf[cb] = fma(pre[rr][cb], fv, gf[cb]);
      }
      if (RHS) gff = fma(fv, fv, gff);
      else if (lane == 0) X[lrow * LDX + n] = fv;
    }
  };

  v4d acc0[N0], acc1[N1];
#pragma unroll
  for (int t = 0; t < N0; ++t) acc0[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < N1; ++t) acc1[t] = v4d{0.0, 0.0, 0.0, 0.0};

  double* X0 = lds;
  double* X1 = lds + GR_RC * LDX;
  for (int idx = tid; idx < 2 * GR_RC * (LDX - N); idx += GR_NT) {   // padding columns stay zero
    const int r = idx / (LDX - N), c = idx - r * (LDX - N);
    lds[r * LDX + N + c] = 0.0;
  }
  if (RHS) {
    for (int r = tid; r < 2 * GR_RC; r += GR_NT) lds[r * LDX + n] = 0.0;
  }
  if (r_lo < m) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { issue(r_lo, h); commit(r_lo, h, X0); }
  }
  __syncthreads();

  double fr[2][NF];
  auto wait_set = [&](double (&f)[NF]) {
    if constexpr (NF == 8)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    else if constexpr (NF == 7)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]), "+v"(f[6]));
    else if constexpr (NF == 6)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]),
                   "+v"(f[5]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]));
  };
  int cidx = 0;
  for (int row0 = r_lo; row0 < m; row0 += GR_RC, ++cidx) {
    const bool more = row0 + GR_RC < m;
    const double* X = (cidx & 1) ? X1 : X0;
    double* Xn = (cidx & 1) ? X0 : X1;
    const unsigned xb = lds_addr(X) + 8u * (unsigned)(lr * LDX + lc);
    static_for<0, NF>([&](auto ic) {
      constexpr int c = decltype(ic)::value;
      lds_read64_off<8 * (4 * K * LDX + 16 * (P + c))>(fr[0][c], xb);
    });
    static_for<0, 4>([&](auto iq) {
      constexpr int q = decltype(iq)::value;            // this wave's q-th k-step of the chunk: s = 2 q + K
      constexpr int cur = q & 1, nxt = cur ^ 1;
      if constexpr (q == 0 || q == 2) {
        if (more) issue(row0 + GR_RC, q / 2);
      }
      wait_set(fr[cur]);
      static_for<0, N0 + N1>([&](auto it) {
        constexpr int t = decltype(it)::value;
        if constexpr (t < N0) acc0[t] = gmfma(fr[cur][0], fr[cur][t], acc0[t]);
        else acc1[t - N0] = gmfma(fr[cur][A1], fr[cur][A1 + (t - N0)], acc1[t - N0]);
        if constexpr (q < 3 && t < NF)
          lds_read64_off<8 * (4 * (2 * (q + 1) + K) * LDX + 16 * (P + t))>(fr[nxt][t], xb);
      });
      if constexpr (q == 1 || q == 3) {
        if (more) commit(row0 + GR_RC, q / 2, Xn);
      }
    });
    __syncthreads();
  }

  // the odd k-steps' partial tiles -> LDS, the even group adds them (even first) and stores
  double* part = lds + (size_t)P * (NTJ + 1) * 256;     // [pair][tile][256]
  if (K == 1) {
#pragma unroll
    for (int t = 0; t < N0; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) part[t * 256 + g * 64 + lane] = acc0[t][g];
#pragma unroll
    for (int t = 0; t < N1; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) part[(N0 + t) * 256 + g * 64 + lane] = acc1[t][g];
  }
  __syncthreads();
  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
  if (K == 0) {
#pragma unroll
    for (int t = 0; t < N0; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        G[(long)(16 * P + lr + 4 * g) * a.NPAD + 16 * (P + t) + lc] = acc0[t][g] + part[t * 256 + g * 64 + lane];
#pragma unroll
    for (int t = 0; t < N1; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        G[(long)(16 * (NTJ - 1 - P) + lr + 4 * g) * a.NPAD + 16 * (NTJ - 1 - P + t) + lc] =
            acc1[t][g] + part[(N0 + t) * 256 + g * 64 + lane];
  }
  if (!RHS) return;
  __syncthreads();                                      // (the partial tiles have been consumed)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) lds[W * 128 + 64 * cb + lane] = gf[cb];
  if (lane == 0) lds[GR_NW * 128 + W] = gff;
  __syncthreads();
  if (tid < n) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[ww * 128 + tid];
    G[(long)tid * a.NPAD + n] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
#pragma unroll
    for (int ww = 0; ww < GR_NW; ++ww) sum += lds[GR_NW * 128 + ww];
    G[(long)n * a.NPAD + n] = sum;
  }
  for (int e = tid; e < a.NPAD * (16 * NT - N); e += GR_NT) {   // padding columns of the rhs tile column
    const int r = e / (16 * NT - N), c = N + e % (16 * NT - N);
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

template <bool RHS>
__global__ __launch_bounds__(GR_NT, 2) void gram8_kernel(GramArgs a) {
  extern __shared__ double lds[];
  if (a.mask && a.mask[blockIdx.y] <= 1) return;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  switch (w) {
    case 0: gram8_wave<0, RHS>(a, lds); break;
    case 1: gram8_wave<1, RHS>(a, lds); break;
    case 2: gram8_wave<2, RHS>(a, lds); break;
    case 3: gram8_wave<3, RHS>(a, lds); break;
    case 4: gram8_wave<4, RHS>(a, lds); break;
    case 5: gram8_wave<5, RHS>(a, lds); break;
    case 6: gram8_wave<6, RHS>(a, lds); break;
    default: gram8_wave<7, RHS>(a, lds); break;
  }
}

// ---- narrow problems (NT <= 5 column tiles): no LDS staging ------------------------------------
// The MFMA operand fragment of column tile c at k-step s is X[4 s + lr][16 c + lc]: 16 lanes read
// 128 contiguous bytes of a row — a coalesced global load straight into the operand register.
// So with few column tiles every wave takes its own k-steps (4 rows each, wave w: k-steps w, w + 8,
// ...) for ALL output tiles, keeps two rounds of four k-steps of loads in flight (the loads of round
// i + 1 are issued before round i is consumed: 512 x 64, 1024 problems 181 -> 95 us), and never meets
// the other waves until the final, fixed-order reduction of the eight partial Grams through LDS.
template <int NTT, bool RHS>
__global__ __launch_bounds__(GR_NT, (NTT <= 2 ? 4 : 2)) void gram_direct_kernel(GramArgs a) {
  constexpr int NTILE = NTT * (NTT + 1) / 2;
  constexpr int KU = 4;                                 // k-steps per round (two rounds in flight per wave)
  extern __shared__ double lds[];                       // [GR_NW][256]
  const int b = blockIdx.y;
  if (a.mask && a.mask[b] <= 1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int n = a.n;
  const int r_lo = blockIdx.x * a.rows_per_chunk;
  int r_hi = r_lo + a.rows_per_chunk;
  if (r_hi > a.m) r_hi = a.m;
  const double* Jb = a.J + (long)b * a.strideJ;
  const double* Fb = a.F + (long)b * a.strideF;
  v4d acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  // RHS (n = 16 NTT exactly): the tiles cover J^T J only; the rhs column (J^T f, f^T f) is accumulated
  // from the operand fragments a lane holds anyway (column 16 c + lc, rows lr mod 4 of this wave's
  // k-steps) and reduced over the lane rows and the waves at the end, in a fixed order
  double gf[NTT], gff = 0.0;
#pragma unroll
  for (int c = 0; c < NTT; ++c) gf[c] = 0.0;
  const int last = r_hi > r_lo ? r_hi - 1 : r_lo;
  constexpr int RSTEP = 4 * GR_NW * KU;                 // rows the workgroup consumes per round
  // raw loads of one round (clamped, unconditional — nothing here waits for the data) ...
  auto load_round = [&](int r0, double (&fr)[KU][NTT], double (&fv)[KU]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int row = r0 + 4 * GR_NW * u + lr;
      const int rc = row < r_hi ? row : last;
      if (RHS) fv[u] = Fb[rc];
#pragma unroll
      for (int c = 0; c < NTT; ++c) {
        const int col = 16 * c + lc;
        // (without RHS the rhs f is column n)
        const double* ptr = (RHS || col < n) ? Jb + (long)rc * a.ldJ + (col < n ? col : n - 1) : Fb + rc;
        fr[u][c] = __builtin_nontemporal_load(ptr);
      }
    }
  };
  // ... and their use: rows / columns outside the problem count as zeros
  auto use_round = [&](int r0, double (&fr)[KU][NTT], double (&fv)[KU]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const bool rin = r0 + 4 * GR_NW * u + lr < r_hi;
      if (RHS) fv[u] = rin ? fv[u] : 0.0;
#pragma unroll
      for (int c = 0; c < NTT; ++c) {
        const int col = 16 * c + lc;
        fr[u][c] = (rin && (RHS ? col < n : col <= n)) ? fr[u][c] : 0.0;
      }
      int t = 0;
#pragma unroll
      for (int i = 0; i < NTT; ++i)
#pragma unroll
        for (int j = i; j < NTT; ++j, ++t) acc[t] = gmfma(fr[u][i], fr[u][j], acc[t]);
      if (RHS) {
#pragma unroll
        for (int c = 0; c < NTT; ++c) gf[c] = fma(fr[u][c], fv[u], gf[c]);
        gff = fma(fv[u], fv[u], gff);
      }
    }
  };
  // two rounds in flight: the loads of round i + 1 are issued before round i is consumed
  {
    double frA[KU][NTT], fvA[KU], frB[KU][NTT], fvB[KU];
    int r0 = r_lo + 4 * w;
    if (r0 < r_hi) load_round(r0, frA, fvA);
    while (r0 < r_hi) {
      if (r0 + RSTEP < r_hi) load_round(r0 + RSTEP, frB, fvB);
      use_round(r0, frA, fvA);
      r0 += RSTEP;
      if (r0 >= r_hi) break;
      if (r0 + RSTEP < r_hi) load_round(r0 + RSTEP, frA, fvA);
      use_round(r0, frB, fvB);
      r0 += RSTEP;
    }
  }
  // cross-wave reduction through LDS (fixed order: deterministic)
  double* G = a.G + ((long)b * gridDim.x + blockIdx.x) * (long)a.NPAD * a.NPAD;
  const int NT = (n + 1 + 15) / 16;
  // (PH tiles per pass through LDS — [tile][wave][256] — and two barriers per pass)
  constexpr int PH = NTILE < 5 ? NTILE : 5;
#pragma unroll
  for (int t0 = 0; t0 < NTILE; t0 += PH) {
#pragma unroll
    for (int tt = 0; tt < PH; ++tt) {
      if (t0 + tt < NTILE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) lds[(tt * GR_NW + w) * 256 + g * 64 + lane] = acc[t0 + tt][g];
      }
    }
    __syncthreads();
    for (int e = tid; e < PH * 256; e += GR_NT) {
      const int tt = e >> 8, el = e & 255;
      int q = t0 + tt, i = 0;
      if (q < NTILE) {
        while (q >= NTT - i) { q -= NTT - i; ++i; }
        const int j = i + q;
        if (j < NT) {
          double sum = 0.0;
#pragma unroll
          for (int ww = 0; ww < GR_NW; ++ww) sum += lds[(tt * GR_NW + ww) * 256 + el];
          const int g = el >> 6, ln = el & 63;
          G[(long)(16 * i + (ln >> 4) + 4 * g) * a.NPAD + 16 * j + (ln & 15)] = sum;
        }
      }
    }
    __syncthreads();
  }
  if (!RHS) return;
  // rhs column: [wave][lane row][column] partials -> column totals (wave-major, then lane row)
#pragma unroll
  for (int c = 0; c < NTT; ++c) lds[(w * 4 + lr) * 64 + 16 * c + lc] = gf[c];
  if (lc == 0) lds[2048 + w * 4 + lr] = gff;
  __syncthreads();
  if (tid < n) {
    double sum = 0.0;
    for (int q = 0; q < 4 * GR_NW; ++q) sum += lds[q * 64 + tid];
    G[(long)tid * a.NPAD + n] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
    for (int q = 0; q < 4 * GR_NW; ++q) sum += lds[2048 + q];
    G[(long)n * a.NPAD + n] = sum;
  }
  for (int e = tid; e < a.NPAD * (16 * NT - (n + 1)); e += GR_NT) {   // padding of the rhs tile column
    const int r = e / (16 * NT - (n + 1)), c = n + 1 + e % (16 * NT - (n + 1));
    if ((r >> 4) <= (c >> 4)) G[(long)r * a.NPAD + c] = 0.0;
  }
}

// partial Grams of the row chunks -> the triangle slot (fixed order: deterministic)
__global__ void gram_reduce_kernel(const double* Gpart, int chunks, int NPAD, double* Gout,
                                   const int* mask) {
  const int b = blockIdx.y;
  if (mask && mask[b] <= 1) return;
  const long sz = (long)NPAD * NPAD;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= sz) return;
  const int r = (int)(idx / NPAD), c = (int)(idx % NPAD);
  if ((r >> 4) > (c >> 4)) return;                      // lower tiles are never written
  const double* src = Gpart + (long)b * chunks * sz + idx;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += src[(long)k * sz];
  Gout[(long)b * sz + idx] = s;
}
// Many row chunks (one very tall problem): eight thread groups sum eight contiguous ranges of the
// chunks, the eight partial sums are added in range order.  (The order depends on the chunk count
// only, i.e. on m — never on the batch.)
constexpr int GRED_G = 8, GRED_W = 32;
__global__ __launch_bounds__(GRED_G * GRED_W) void gram_reduce_wide_kernel(
    const double* Gpart, int chunks, int NPAD, double* Gout, const int* mask) {
  __shared__ double part[GRED_G][GRED_W];
  const int b = blockIdx.y;
  if (mask && mask[b] <= 1) return;
  const long sz = (long)NPAD * NPAD;
  const int e = threadIdx.x % GRED_W, g = threadIdx.x / GRED_W;
  const long idx = (long)blockIdx.x * GRED_W + e;
  const int r = (int)(idx / NPAD), c = (int)(idx % NPAD);
  const bool live = idx < sz && (r >> 4) <= (c >> 4);
  const int per = (chunks + GRED_G - 1) / GRED_G;
  const int k0 = g * per, k1 = min(chunks, k0 + per);
  double s0 = 0.0;
  if (live) {
    const double* src = Gpart + (long)b * chunks * sz + idx;
    for (int k = k0; k < k1; ++k) s0 += __builtin_nontemporal_load(src + (long)k * sz);
  }
  part[g][e] = s0;
  __syncthreads();
  if (g == 0 && live) {
    double s = part[0][e];
#pragma unroll
    for (int q = 1; q < GRED_G; ++q) s += part[q][e];
    Gout[(long)b * sz + idx] = s;
  }
}

// ---- equilibrated blocked Cholesky, in place in the triangle slot -------------------------------
// Row block kb of R' :  S_j = C_{kb,j} - sum_{k<kb} R'_{k,kb}^T R'_{k,j}   (MFMA, operands from the
// rows already written),  R'_{kb,kb} = chol(S_kb) and its inverse on wave 0 (lane j owns column
// j, broadcasts by v_readlane),  R'_{kb,j} = R'_{kb,kb}^-T S_j  (MFMA; the accumulator layout of S is
// the B-operand layout).  What is stored is R = R' D^-1; operands are re-scaled on the fly.
//
// The same kernel factors the diagonally modified Grams of the trust-region systems (TRF):
//     H = D G D + diag(e^2) (+ alpha I on the first n columns),    D = diag(colscale, 1)
// whose Cholesky factor is the triangle of [R D | c; E | 0] (and of [R_aug; sqrt(alpha) I]).  With
// C = equil(G):  equil(H) = Theta^1/2 C Theta^1/2 + (I - Theta),  0 < Theta <= I diagonal, so its
// extreme eigenvalues lie inside those of C: a problem that passed the gate on C needs no new one.
// NWP waves work on one problem: 8 (a whole workgroup; any size) or 1 (N <= 80: eight problems per
// workgroup, no workgroup barrier at all — the 16x16 chain of one problem overlaps the MFMAs of
// the others on the same SIMD, and sixteen instead of two problems are resident per CU).
template <int NWP>
__global__ __launch_bounds__(GR_NT, 4) void gram_chol_kernel(GramCholArgs a) {
  constexpr int PT = WAVE * NWP;                        // threads per problem
  constexpr int UMAX = (NWP == 8) ? 3 : 5;              // tiles of a row block per wave
  constexpr int PPW = GR_NW / NWP;                      // problems per workgroup
  extern __shared__ double sh_all[];
  __shared__ double red[32];
  __shared__ double pmin_all[GR_NW];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int pslot = wv / NWP;                           // problem slot inside the workgroup
  const int pidx = (int)blockIdx.x * PPW + pslot;
  if (pidx >= a.count) return;                          // (NWP == 1 only: wave-uniform)
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  const int tid = (int)threadIdx.x % PT, lane = tid & 63;
  const int w = wv % NWP;
  double* sh = sh_all + (size_t)pslot * (4 * (size_t)a.NPAD + 512);
  double& pminsh = pmin_all[pslot];
  // synchronisation among the threads of one problem
  auto psync = [&]() {
    if (NWP == 8) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // one wave: program order
  };
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0) return;
  // columns of this problem: all n (+ rhs), or the gathered free columns (+ rhs)
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  // source row / column of H's index i  (the rhs is the source's column a.n)
  auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;    // source Gram (may alias the output)
  double* Gb = a.G + (long)b * NPAD * NPAD;             // output triangle
  double* dl = sh;                 // [NPAD] equilibration 1 / sqrt(h_jj)
  double* sq = dl + NPAD;          // [NPAD] sqrt(h_jj)
  double* sc = sq + NPAD;          // [NPAD] colscale_j * dl_j  (scale applied to source entries)
  double* Dt = sc + NPAD;          // [256]  diagonal tile (row-major)
  double* Ri = Dt + 256;           // [256]  its inverse
  double* td = Ri + 256;           // [NPAD] (e_j^2 + alpha) * dl_j^2  (added to the diagonal of C)
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  // 0. column scales from the diagonal of H
  int bad = 0;
  for (int j = tid; j < NPAD; j += PT) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s = g * d;
    }
    dl[j] = d; sq[j] = s; sc[j] = cs * d; td[j] = add * d * d;
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
  }
  if (a.colinfo) {                                      // (uniform) column-norm summary for the rank gate
    psync();
    if (tid == 0) {
      double mn = __builtin_inf(), sm = 0.0;
      for (int j = 0; j < n; ++j) { const double v = sq[j]; mn = v < mn ? v : mn; sm = fma(v, v, sm); }
      a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
    }
  }
  // strictly lower tiles are part of the triangle's image: zero
  for (int r = 16 + w; r < NPAD; r += NWP) {
    const int cend = r & ~15;
    for (int c = lane; c < cend; c += WAVE) Gb[(long)r * NPAD + c] = 0.0;
  }
  if (NWP == 8) bad = block_or(bad, red); else bad = __any(bad);
  if (tid == 0) pminsh = 1.0;
  psync();
  if (bad) {                                            // uniform: hand the problem to the QR tree
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.path_out) a.path_out[b] = a.n + 1;
    }
    return;
  }

  for (int kb = 0; kb < NT; ++kb) {
    // ---- A. Schur complements of this row block (tile j = kb + w + 8 u) ----
    v4d S[UMAX];
    const double dk = dl[16 * kb + lc];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = kb + w + NWP * u;
      S[u] = v4d{0.0, 0.0, 0.0, 0.0};
      if (j < NT) {
        const double dj = dl[16 * j + lc];
        const double scj = sc[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * kb + lr + 4 * g;
          const int col = 16 * j + lc;
          double v = 0.0;
          if (row < N && col < N) {
            int sr_ = src(row), sc_ = src(col);
            if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }   // symmetric: stay in the upper tiles
            v = Gs[(long)sr_ * NPAD + sc_] * sc[row] * scj;
          }
          if (j == kb && lr + 4 * g == lc) v += td[row];
          S[u][g] = v;
        }
        for (int k = 0; k < kb; ++k) {
          double av[4], bv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const long ro = (long)(16 * k + 4 * s + lr) * NPAD;
            av[s] = Gb[ro + 16 * kb + lc];
            bv[s] = Gb[ro + 16 * j + lc];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) S[u] = gmfma(-(av[s] * dk), bv[s] * dj, S[u]);
        }
      }
    }
    // ---- B. wave 0: Cholesky of the diagonal tile and its inverse (chol16.h), straight from the
    // accumulators of its Schur complement ----
    if (w == 0) {
      const double pm = chol16_blocked3(S[0], Dt, Ri, n - 16 * kb, pminsh);
      if (lane == 0) pminsh = pm;
    }
    psync();
    if (a.rinv && w == NWP - 1) {                       // kept for the conditioning certificate (off the chain)
      double* ro = a.rinv + ((long)b * (NPAD / 16) + kb) * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = Ri[q * 64 + lane];
    }
    // ---- C. R'_{kb,j} = R'_{kb,kb}^-T S_j, stored as R = R' D^-1 ----
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = kb + w + NWP * u;
      if (j < NT) {
        v4d X = {0.0, 0.0, 0.0, 0.0};
        if (j == kb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) X[g] = Dt[(lr + 4 * g) * 16 + lc];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) X = gmfma(Ri[(4 * s + lr) * 16 + lc], S[u][s], X);
        }
        const double sj = sq[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * kb + lr + 4 * g;
          const int colg = 16 * j + lc;
          double val = X[g] * sj;
          if (row >= n || row > colg || colg > n) val = 0.0;
          Gb[(long)row * NPAD + colg] = val;
        }
      }
    }
    psync();
  }
  if (16 * NT < NPAD) {                                  // sub-matrix: the rest of the slot is zero
    for (int r = w; r < NPAD; r += NWP) {
      const int c0 = (r < 16 * NT) ? 16 * NT : (r & ~15);
      for (int c = c0 + lane; c < NPAD; c += WAVE) Gb[(long)r * NPAD + c] = 0.0;
    }
  }
  if (tid == 0 && a.fb_mask) {
    const bool fail = !(pminsh >= GRAM_SMIN * GRAM_SMIN);
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
  }
}

// ---- N <= 80, one wave per problem, the whole matrix in registers -------------------------------
// At most 5 x 5 tiles: the 15 upper tiles of the equilibrated matrix are loaded ONCE into
// accumulators and never leave the wave until their row block is final.  Per row block: the chain of
// the diagonal tile straight from its accumulator (chol16.h), R'_{kb,j} = R'_kk^-T S_j by MFMA, and
// the right-looking update of the remaining tiles — whose MFMA operands are the rows just solved,
// already in the right layout (register s of a tile in the accumulator layout holds rows 4 s + lr:
// the operand fragment of k-step s).  No L2 round trip inside the factorisation (the left-looking
// kernel above pays one per tile and finished row block: 20 exposed latencies at N = 65), no barrier.
// Same arguments, outputs and gate bookkeeping as gram_chol_kernel<1>; eight problems per workgroup.
__global__ __launch_bounds__(REG_NT, 1) void gram_chol_reg_kernel(GramCholArgs a) {
  constexpr int MT = 5;                                 // tile rows at most (N <= 80)
  extern __shared__ double sh_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int pidx = (int)blockIdx.x * REG_NW + wv;
  if (pidx >= a.count) return;                          // (wave-uniform)
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  const int lane = threadIdx.x & 63, tid = lane;
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  double* sh = sh_all + (size_t)wv * (4 * (size_t)NPAD + 256 + MT * 256 + 64);
  auto wsync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0) return;
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;    // source Gram (may alias the output)
  double* Gb = a.G + (long)b * NPAD * NPAD;             // output triangle
  double* dl = sh;                 // [NPAD] equilibration 1 / sqrt(h_jj)
  double* sq = dl + NPAD;          // [NPAD] sqrt(h_jj)
  double* sc = sq + NPAD;          // [NPAD] colscale_j * dl_j
  double* Dt = sc + NPAD;          // [256]  diagonal tile (row-major)
  double* Ria = Dt + 256;          // [MT][256] the inverses of the diagonal tiles (kept: certificate below)
  double* td = Ria + MT * 256;     // [NPAD] (e_j^2 + alpha) * dl_j^2
  double* xs = td + NPAD;          // [64]   four-row sums
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  int* sidx = (int*)Ria;           // [NPAD] source indices — only until the tiles are loaded (Ria is free till then)
  // 0. column scales from the diagonal of H
  int bad = 0;
  for (int j = tid; j < NPAD; j += WAVE) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s_ = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s_ = g * d;
    }
    dl[j] = d; sq[j] = s_; sc[j] = cs * d; td[j] = add * d * d;
    sidx[j] = sj_ < NPAD ? sj_ : 0;                     // source row / column of index j (always a valid one)
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
  }
  wsync();
  if (a.colinfo && tid == 0) {                          // column-norm summary for the rank gate
    double mn = __builtin_inf(), sm = 0.0;
    for (int j = 0; j < n; ++j) { const double v = sq[j]; mn = v < mn ? v : mn; sm = fma(v, v, sm); }
    a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
  }
  bad = __any(bad);
  // 1. the scaled source tiles -> accumulators (the source may alias the output: every read comes
  //    before any write)
  v4d acc[MT * (MT + 1) / 2];
  auto tix = [](int i, int j) { return i * MT - i * (i - 1) / 2 + (j - i); };   // upper tile (i, j) of a 5 x 5 grid
  if (!bad) {
    // (unconditional loads from clamped indices, all in flight together; selected afterwards)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int srow[4];
      double scr_[4];
      if (i < NT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g;
          srow[g] = sidx[row < N ? row : N - 1];
          scr_[g] = sc[row];
        }
      }
#pragma unroll
      for (int j = i; j < MT; ++j) {
        v4d v4 = {0.0, 0.0, 0.0, 0.0};
        if (j < NT) {
          const int col = 16 * j + lc;
          const int scol = sidx[col < N ? col : N - 1];
          const double scj = sc[col];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * i + lr + 4 * g;
            const int lo_ = srow[g] < scol ? srow[g] : scol, hi_ = srow[g] < scol ? scol : srow[g];   // symmetric: upper tiles
            double v = Gs[(long)lo_ * NPAD + hi_];
            v = (row < N && col < N) ? v * scr_[g] * scj : 0.0;
            if (j == i && lr + 4 * g == lc) v += td[row];
            v4[g] = v;
          }
        }
        acc[tix(i, j)] = v4;
      }
    }
  }
  wsync();
  if (bad) {                                            // hand the problem to the QR tree
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.path_out) a.path_out[b] = a.n + 1;
    }
    return;
  }
  // strictly lower tiles and everything beyond 16 NT are part of the triangle's image: zero
  for (int r = 0; r < NPAD; ++r) {
    const int cend = (r < 16 * NT) ? (r & ~15) : NPAD;
    for (int c = lane; c < cend; c += WAVE) Gb[(long)r * NPAD + c] = 0.0;
    if (r < 16 * NT)
      for (int c = 16 * NT + lane; c < NPAD; c += WAVE) Gb[(long)r * NPAD + c] = 0.0;
  }
  double pmin = 1.0;
#pragma unroll
  for (int kb = 0; kb < MT; ++kb) {
    if (kb < NT) {
      // 2. chain of the diagonal tile: R'_kk -> Dt, its inverse -> Ri
      double* Ri = Ria + kb * 256;
      pmin = chol16_blocked3(acc[tix(kb, kb)], Dt, Ri, n - 16 * kb, pmin);
      if (a.rinv) {                                     // kept for the conditioning certificate
        double* ro = a.rinv + ((long)b * (NPAD / 16) + kb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = Ri[q * 64 + lane];
      }
      // 3. the row block: R'_{kb,j} = R'_kk^-T S_j (kept in the accumulators of row kb), stored as R = R' D^-1
#pragma unroll
      for (int j = kb; j < MT; ++j) {
        if (j < NT) {
          v4d X = {0.0, 0.0, 0.0, 0.0};
          if (j == kb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) X[g] = Dt[(lr + 4 * g) * 16 + lc];
            acc[tix(kb, kb)] = X;                       // (all 15 tiles of R' stay in registers: certificate)
          } else {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) X = gmfma(Ri[(4 * s_ + lr) * 16 + lc], acc[tix(kb, j)][s_], X);
            acc[tix(kb, j)] = X;
          }
          const double sj = sq[16 * j + lc];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * kb + lr + 4 * g;
            const int colg = 16 * j + lc;
            double val = X[g] * sj;
            if (row >= n || row > colg || colg > n) val = 0.0;
            Gb[(long)row * NPAD + colg] = val;
          }
        }
      }
      // 4. right-looking update of the tiles below: (i, j) -= R'_{kb,i}^T R'_{kb,j}
#pragma unroll
      for (int i = kb + 1; i < MT; ++i) {
#pragma unroll
        for (int j = i; j < MT; ++j) {
          if (j < NT) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
              acc[tix(i, j)] = gmfma(-acc[tix(kb, i)][s_], acc[tix(kb, j)][s_], acc[tix(i, j)]);
          }
        }
      }
      wsync();                                          // (Dt / Ri are rewritten by the next chain)
    }
  }
  const bool fail = !(pmin >= GRAM_SMIN * GRAM_SMIN);
  if (tid == 0 && a.fb_mask) {
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
  }
  // 5. The first bound of the conditioning certificate (gram_cond_kernel below: same quantities, same
  //    definition) while R' and the inverse diagonal tiles are still at hand:
  //        K2 = ||R'||_1 ||R'||_inf ||Y||_1 ||Y||_inf ,   Y = R'^-T  column block by column block.
  //    K2 <= GRAM_K2_MAX settles the problem here; otherwise the separate kernel decides (it also
  //    has the tighter Frobenius bound).
  if (a.cert_done) {
    bool passed = false;
    double k2 = 0.0;
    if (!fail) {
      const int NTn = (n + 15) / 16;
      double r1 = 0.0, rinf = 0.0, y1 = 0.0, yinf = 0.0;
      double colp[MT];
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) colp[jj] = 0.0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < NTn) {
          double rp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int jj = i; jj < MT; ++jj) {
            if (jj < NTn) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int row = 16 * i + lr + 4 * g, col = 16 * jj + lc;
                const double v = (row < n && col < n) ? fabs(acc[tix(i, jj)][g]) : 0.0;
                rp[g] += v; colp[jj] += v;
              }
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) rinf = fmax(rinf, row16_sum(rp[g]));
        }
      }
      rinf = wave_max(rinf);
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) {
        if (jj < NTn) {
          xs[lane] = colp[jj];
          wsync();
          r1 = fmax(r1, (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]));
          wsync();
        }
      }
      r1 = wave_max(r1);
      double rsY[MT][4];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) rsY[i][g] = 0.0;
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) {
        if (jj < NTn) {
          v4d Yc[MT];
          double cY = 0.0;
#pragma unroll
          for (int i = jj; i < MT; ++i) {
            if (i < NTn) {
              v4d Yt = {0.0, 0.0, 0.0, 0.0};
              if (i == jj) {
#pragma unroll
                for (int g = 0; g < 4; ++g) Yt[g] = Ria[jj * 256 + lc * 16 + lr + 4 * g];   // (R'_jj^-1)^T
              } else {
                v4d av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = jj; kk < i; ++kk) {
#pragma unroll
                  for (int s_ = 0; s_ < 4; ++s_) av = gmfma(acc[tix(kk, i)][s_], Yc[kk][s_], av);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) Yt = gmfma(-Ria[i * 256 + (4 * s_ + lr) * 16 + lc], av[s_], Yt);
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int row = 16 * i + lr + 4 * g, col = 16 * jj + lc;
                const double v = (row < n && col < n) ? Yt[g] : 0.0;
                Yt[g] = v;
                const double av_ = fabs(v);
                rsY[i][g] += row16_sum(av_);
                cY += av_;
              }
              Yc[i] = Yt;
            }
          }
          xs[lane] = cY;
          wsync();
          y1 = fmax(y1, (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]));
          wsync();
        }
      }
      y1 = wave_max(y1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) yinf = fmax(yinf, rsY[i][g]);
      yinf = wave_max(yinf);
      k2 = (r1 * rinf) * (y1 * yinf);
      passed = k2 <= GRAM_K2_MAX;                       // (NaN fails)
    }
    if (tid == 0) {
      a.cert_done[b] = passed ? 1 : 0;
      if (passed && a.k2_out) a.k2_out[b] = k2;
    }
  }
}

// ---- N <= 80: ALL Newton rounds of a problem in one launch ----------------------------------------
// The safeguarded Newton iteration on alpha (trust_region.py:126-150) factors H + alpha I once per
// round.  For N <= 80 one wave owns a problem for the whole iteration: per round the factor of
// gram_chol_reg_kernel (tiles in registers, nothing stored), p = -R^-1 c by block back substitution
// and q = R^-T p by block forward substitution straight from the register tiles (tile x vector: four
// FMAs per lane and tile + a 16-lane DPP sum, or a four-row sum through LDS for the transposed
// product; the 16 x 16 diagonal solves are matvecs with the inverse tiles the chain produces anyway),
// then the scalar update of lm_update_kernel, verbatim.  No launch, no counter read-back and no
// triangle written between rounds (six stream operations per round otherwise, each with its dispatch
// gap).  Everything happens in the equilibrated system:  R = R' diag(sq),  c = c' sq_n  =>
//     p_j = -sq_n dl_j (R'^-1 c')_j ,      q = R'^-T (dl . p) .
__device__ __forceinline__ double lm_restart_reg(double lo, double hi) {     // trust_region.py:128,134
  const double gm = sqrt(lo * hi);
  return (0.001 * hi > gm) ? 0.001 * hi : gm;
}
// The launch first does what lm_start_kernel does — the Gauss-Newton step from the AUGMENTED factor
// (its stored triangle, column scales and inverse diagonal tiles are re-loaded: R' = R diag(dl)), the
// acceptance test |p| <= Delta and the bracket (trust_region.py:116-130) — for every
// normal-equations-path problem of the batch (the others are left to lm_start and the round loop,
// LmState.fused_gram): no list, no counter, and the same arithmetic for a problem whatever else its
// batch holds.
__global__ __launch_bounds__(REG_NT, 1) void lm_rounds_reg_kernel(GramCholArgs a, LmState lm,
                                                                 const double* Delta_in,
                                                                 const double* alpha_in) {
  constexpr int MT = 5;
  constexpr bool start = true;
  extern __shared__ double sh_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int b = (int)blockIdx.x * REG_NW + wv;
  if (b >= lm.B) return;
  if (lm.path && lm.path[b] != 0) return;               // (Householder-path problem: lm_start and the round loop)
  const int lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
  if (!lm.fast[b]) {
    if (lane == 0) lm.ncols_lm[b] = 0;
    return;
  }
  double* scv = lm.sc + (long)b * 16;
  int* stv = lm.st + (long)b * 4;
  int phase = LM_EVAL;
  const int NPAD = a.NPAD, n = a.n, N = n + 1;
  const int NT = (N + 15) / 16, NTn = (n + 15) / 16;
  const int jn = n >> 4, cn = n & 15;                   // tile column / column inside it of the rhs
  double* sh = sh_all + (size_t)wv * (8 * (size_t)NPAD + 256 + MT * 256 + 16 + 64);
  double* dl = sh;                 // [NPAD] 1 / sqrt(h_jj)
  double* sq = dl + NPAD;          // [NPAD] sqrt(h_jj)
  double* sc = sq + NPAD;          // [NPAD] colscale_j dl_j
  double* td = sc + NPAD;          // [NPAD] (e_j^2 + alpha) dl_j^2
  double* cv = td + NPAD;          // [NPAD] c' = R'[:, n]
  double* yv = cv + NPAD;          // [NPAD] R'^-1 c', then dl . p
  double* pv = yv + NPAD;          // [NPAD] p
  double* zv = pv + NPAD;          // [NPAD] R'^-T (dl . p)
  double* Dt = zv + NPAD;          // [256]
  double* Ria = Dt + 256;          // [MT][256] inverse diagonal tiles
  double* tv = Ria + MT * 256;     // [16]
  double* xs = tv + 16;            // [64]
  auto wsync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  auto tix = [](int i, int j) { return i * MT - i * (i - 1) / 2 + (j - i); };
  v4d acc[MT * (MT + 1) / 2];
  double sqn = 1.0;

  // y = R'^-1 c' (rows and columns below n only; c' in cv), block rows from the bottom -> yv
  auto back_solve = [&]() {
#pragma unroll
    for (int kk = MT - 1; kk >= 0; --kk) {
      if (kk < NTn) {
        double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = kk + 1; j < MT; ++j) {
          if (j < NTn) {
            const double yj = yv[16 * j + lc];
#pragma unroll
            for (int g = 0; g < 4; ++g) part[g] = fma(acc[tix(kk, j)][g], yj, part[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) part[g] = row16_sum(part[g]);
        if (lc == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) tv[lr + 4 * g] = cv[16 * kk + lr + 4 * g] - part[g];
        }
        wsync();
        const int nb = (n - 16 * kk < 16) ? n - 16 * kk : 16;
        const double* Rk = Ria + kk * 256;
        double yi = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) yi = fma(Rk[lc * 16 + c], (c < nb) ? tv[c] : 0.0, yi);
        if (lc >= nb) yi = 0.0;
        if (lr == 0) yv[16 * kk + lc] = yi;
        wsync();
      }
    }
  };
  // p = -sq_n dl . y -> pv,  w = dl . p -> yv;  returns |p|
  auto form_p = [&]() -> double {
    double pp = 0.0;
    for (int j = lane; j < NPAD; j += WAVE) {
      const double pj = (j < n) ? -(sqn * dl[j] * yv[j]) : 0.0;
      pv[j] = pj;
      pp = fma(pj, pj, pp);
    }
    wsync();
    for (int j = lane; j < NPAD; j += WAVE) yv[j] = (j < n) ? dl[j] * pv[j] : 0.0;
    const double pn_ = sqrt(wave_sum(pp));
    wsync();
    return pn_;
  };
  // z = R'^-T w (w in yv), block rows from the top -> zv;  returns |z|^2
  auto fwd_solve = [&]() -> double {
#pragma unroll
    for (int kk = 0; kk < MT; ++kk) {
      if (kk < NTn) {
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < kk; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) part = fma(acc[tix(j, kk)][g], zv[16 * j + lr + 4 * g], part);
        }
        xs[lane] = part;
        wsync();
        const double tot = (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]);
        if (lr == 0) tv[lc] = yv[16 * kk + lc] - tot;
        wsync();
        const int nb = (n - 16 * kk < 16) ? n - 16 * kk : 16;
        const double* Rk = Ria + kk * 256;
        double zi = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) zi = fma(Rk[c * 16 + lc], (c <= lc) ? tv[c] : 0.0, zi);
        if (lc >= nb) zi = 0.0;
        if (lr == 0) zv[16 * kk + lc] = zi;
        wsync();
      }
    }
    double qq = 0.0;
    for (int j = lane; j < NPAD; j += WAVE) { const double zj = (j < n) ? zv[j] : 0.0; qq = fma(zj, zj, qq); }
    return wave_sum(qq);
  };

  double alpha, lo, hi, phi, dphi, Delta;
  int it, n_iter;
  if (start) {
    // ---- the augmented factor back into registers:  R' = R diag(dl),  Ri from the factor kernel ----
    Delta = Delta_in[b];
    const double* Ra = lm.Raug + (long)b * NPAD * NPAD;
    const double* dsc = a.dsc + (long)b * NPAD;
    const double* rinv = a.rinv + (long)b * (NPAD / 16) * 256;
    for (int j = lane; j < NPAD; j += WAVE) { dl[j] = dsc[j]; yv[j] = 0.0; pv[j] = 0.0; zv[j] = 0.0; }
    for (int e = lane; e < NT * 256; e += WAVE) Ria[e] = rinv[e];
    wsync();
    sqn = 1.0 / dl[n];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = i; j < MT; ++j) {
        v4d v4 = {0.0, 0.0, 0.0, 0.0};
        if (j < NT) {
          const int col = 16 * j + lc;
          const double dj = dl[col];
#pragma unroll
          for (int g = 0; g < 4; ++g) v4[g] = Ra[(long)(16 * i + lr + 4 * g) * NPAD + col] * dj;
        }
        acc[tix(i, j)] = v4;
      }
    }
    for (int r = lane; r < NPAD; r += WAVE) cv[r] = (r < n) ? Ra[(long)r * NPAD + n] * dl[n] : 0.0;
    wsync();
    // |R^T c| = sq_n |sq . (R'^T c')|  (alpha_upper = |A^T b| / Delta, trust_region.py:111-113)
    double gg = 0.0;
#pragma unroll
    for (int kk = 0; kk < MT; ++kk) {
      if (kk < NTn) {
        double part = 0.0;
#pragma unroll
        for (int j = 0; j <= kk; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) part = fma(acc[tix(j, kk)][g], cv[16 * j + lr + 4 * g], part);
        }
        xs[lane] = part;
        wsync();
        const double tot = (xs[lc] + xs[16 + lc]) + (xs[32 + lc] + xs[48 + lc]);
        const int col = 16 * kk + lc;
        const double gj = (col < n) ? tot / dl[col] : 0.0;
        if (lr == 0) gg = fma(gj, gj, gg);
        wsync();
      }
    }
    const double gnorm = sqn * sqrt(wave_sum(gg));
    back_solve();
    const double pn = form_p();
    for (int j = lane; j < n; j += WAVE) lm.ph[(long)b * lm.ld + j] = pv[j];
    if (pn <= Delta) {                                  // trust_region.py:116-117
      if (lane == 0) {
        scv[SC_ALPHA] = 0.0; stv[ST_NITER] = 0; stv[ST_PHASE] = LM_IDLE; scv[SC_DELTA] = Delta;
        lm.ncols_lm[b] = 0;
      }
      return;
    }
    const double qq = fwd_solve();                      // phi(0), phi'(0) -> alpha_lower (:121-123)
    phi = pn - Delta;
    dphi = -qq / pn;
    hi = gnorm / Delta;
    lo = -phi / dphi;
    alpha = alpha_in[b];                                // :127-130 (full rank)
    if (alpha < lo || alpha > hi) alpha = lm_restart_reg(lo, hi);   // :133-134, iteration 0
    it = 0; n_iter = 0;
    if (lane == 0) scv[SC_DELTA] = Delta;
  }
  for (int guard = 0; guard < 12; ++guard) {
    const double sa = sqrt(alpha);
    // ---- factor of H + alpha I (as gram_chol_reg_kernel; no gather, nothing stored) ----
    for (int j = lane; j < NPAD; j += WAVE) {
      const double cs = (csv && j < n) ? csv[j] : 1.0;
      const double ej = (edv && j < n) ? edv[j] : 0.0;
      const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
      const double g = (j < N) ? fma(Gs[(long)j * NPAD + j] * cs, cs, add) : 0.0;
      const bool okc = (g > 0.0) && is_finite(g);
      double d = 1.0, s_ = 1.0;
      if (j < N && okc) {
        d = __builtin_amdgcn_rsq(g);
        d = d * fma(-0.5 * g * d, d, 1.5);
        d = d * fma(-0.5 * g * d, d, 1.5);
        s_ = g * d;
      }
      dl[j] = d; sq[j] = s_; sc[j] = cs * d; td[j] = add * d * d;
      yv[j] = 0.0; pv[j] = 0.0; zv[j] = 0.0;
    }
    wsync();
    sqn = sq[n];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      double scr_[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) scr_[g] = sc[(16 * i + lr + 4 * g) < NPAD ? 16 * i + lr + 4 * g : NPAD - 1];
#pragma unroll
      for (int j = i; j < MT; ++j) {
        v4d v4 = {0.0, 0.0, 0.0, 0.0};
        if (j < NT) {
          const int col = 16 * j + lc;
          const int ccl = col < N ? col : N - 1;
          const double scj = sc[col];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * i + lr + 4 * g;
            const int rcl = row < N ? row : N - 1;
            const int lo_ = rcl < ccl ? rcl : ccl, hi_ = rcl < ccl ? ccl : rcl;
            double v = Gs[(long)lo_ * NPAD + hi_];
            v = (row < N && col < N) ? v * scr_[g] * scj : 0.0;
            if (j == i && lr + 4 * g == lc) v += td[row];
            v4[g] = v;
          }
        }
        acc[tix(i, j)] = v4;
      }
    }
    wsync();
    double pmin = 1.0;
#pragma unroll
    for (int kb = 0; kb < MT; ++kb) {
      if (kb < NT) {
        double* Ri = Ria + kb * 256;
        pmin = chol16_blocked3(acc[tix(kb, kb)], Dt, Ri, n - 16 * kb, pmin);
        if (kb == jn) {                                 // the rhs column runs through this diagonal tile
          if (lane < 16) cv[16 * kb + lane] = Dt[lane * 16 + cn];
        }
#pragma unroll
        for (int j = kb + 1; j < MT; ++j) {
          if (j < NT) {
            v4d X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) X = gmfma(Ri[(4 * s_ + lr) * 16 + lc], acc[tix(kb, j)][s_], X);
            acc[tix(kb, j)] = X;
            if (j == jn && lc == cn) {
#pragma unroll
              for (int g = 0; g < 4; ++g) cv[16 * kb + lr + 4 * g] = X[g];
            }
          }
        }
#pragma unroll
        for (int i = kb + 1; i < MT; ++i) {
#pragma unroll
          for (int j = i; j < MT; ++j) {
            if (j < NT) {
#pragma unroll
              for (int s_ = 0; s_ < 4; ++s_)
                acc[tix(i, j)] = gmfma(-acc[tix(kb, i)][s_], acc[tix(kb, j)][s_], acc[tix(i, j)]);
            }
          }
        }
        wsync();
      }
    }
    back_solve();
    const double pn = form_p();
    bool finished = false;
    if (phase == LM_FINAL) {
      finished = true;                                  // p at the updated alpha, rescale test on the STALE phi (:149)
    } else {
      const double qq = fwd_solve();
      // ---- the update of lm_update_kernel (trust_region.py:136-146) ----
      phi = pn - Delta;
      dphi = -qq / pn;
      if (fabs(phi) < 0.01 * Delta) {                   // :138-139
        finished = true;
        n_iter = it + 1;
      } else {
        if (phi < 0.0) hi = alpha;                      // :141-142
        const double ratio = phi / dphi;
        const double cand = alpha - ratio;
        lo = (cand > lo) ? cand : lo;                   // :145
        alpha -= (phi + Delta) * ratio / Delta;         // :146
        ++it;
        if (it >= 10) {                                 // max_iter reached: final p at the new alpha
          n_iter = 10;
          phase = LM_FINAL;
        } else {
          if (alpha < lo || alpha > hi) alpha = lm_restart_reg(lo, hi);   // :133-134 of the next pass
          phase = LM_EVAL;
        }
      }
    }
    if (finished) {
      const double f = (phi > 0.0) ? Delta / pn : 1.0;  // :149-150
      for (int j = lane; j < n; j += WAVE) lm.ph[(long)b * lm.ld + j] = pv[j] * f;
      break;
    }
  }
  if (lane == 0) {
    scv[SC_ALPHA] = alpha; scv[SC_LO] = lo; scv[SC_HI] = hi; scv[SC_PHI] = phi; scv[SC_DPHI] = dphi;
    stv[ST_IT] = it; stv[ST_PHASE] = LM_IDLE; stv[ST_NITER] = n_iter;
    lm.sa[b] = sqrt(alpha);
    lm.ncols_lm[b] = 0;
  }
}

hipError_t launch_lm_rounds_reg(const GramCholArgs& c, const LmState& lm, const double* Delta,
                                const double* alpha_in, hipStream_t s) {
  const size_t per = sizeof(double) * (8 * (size_t)c.NPAD + 256 + 5 * 256 + 16 + 64);
  static std::atomic<size_t> granted[64];
  hipError_t ge = gram_grant_lds(lm_rounds_reg_kernel, per * REG_NW, granted);
  if (ge != hipSuccess) return ge;
  hipLaunchKernelGGL(lm_rounds_reg_kernel, dim3((lm.B + REG_NW - 1) / REG_NW), dim3(REG_NT), per * REG_NW, s, c, lm,
                     Delta, alpha_in);
  return hipGetLastError();
}

// ---- right-looking variant: the whole (scaled) matrix lives in accumulators ----------------------
// The NT (NT + 1) / 2 <= 153 upper tiles are dealt CYCLICALLY (row-major tile q -> wave q % 8, slot
// q / 8) so that the shrinking trailing matrix stays balanced, and never leave the registers until
// their row block is final.  Per row block kb: the diagonal tile goes through LDS to wave 0 for the
// 16x16 Cholesky + inverse, the owners of the tiles (kb, j) solve them by MFMA and publish them in an
// LDS row buffer, and every wave updates its own trailing tiles from that buffer — no global-memory
// round trip inside the factorisation (the left-looking kernel above re-reads finished rows from L2).
// Same arguments, same outputs, same gate bookkeeping as gram_chol_kernel.
template <int SL>
__global__ __launch_bounds__(GR_NT, 2) void gram_chol_rl_kernel(GramCholArgs a) {
  extern __shared__ double sh[];
  __shared__ double red[32];
  __shared__ double pminsh;
  __shared__ int flagsh;                                // last diagonal tile handed to wave 0
  const int pidx = (int)blockIdx.x;
  if (a.count_dev && pidx >= *a.count_dev) return;
  const int b = a.batch_list ? a.batch_list[pidx] : pidx;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int NPAD = a.NPAD;
  if (a.mask && a.mask[b] <= 1) {
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  if (a.skip_path && a.skip_path[b] != 0) return;
  const int N = a.ncols_dev ? a.ncols_dev[b] : a.n + 1;
  if (N <= 1) {                                         // (dogbox: every variable active — nothing to factor)
    if (tid == 0 && a.fb_mask) a.fb_mask[b] = 0;
    return;
  }
  const int n = N - 1;
  const int NT = (N + 15) / 16;
  const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
  auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
  const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
  double* Gb = a.G + (long)b * NPAD * NPAD;
  double* dl = sh;                 // [NPAD]
  double* sq = dl + NPAD;          // [NPAD]
  double* sc = sq + NPAD;          // [NPAD]
  double* td = sc + NPAD;          // [NPAD]
  double* Dt = td + NPAD;          // [256]
  double* Ri = Dt + 256;           // [256]
  double* Rrow = Ri + 256;         // [NT][256] finished tiles of the current row block
  const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
  const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
  const double sa = a.diag_sqrt ? a.diag_sqrt[b] : 0.0;
  int bad = 0;
  for (int j = tid; j < NPAD; j += GR_NT) {
    const double cs = (csv && j < n) ? csv[j] : 1.0;
    const double ej = (edv && j < n) ? edv[j] : 0.0;
    const double add = (j < n) ? fma(ej, ej, sa * sa) : 0.0;
    const int sj_ = (j < N) ? src(j) : j;
    const double g = (j < N) ? fma(Gs[(long)sj_ * NPAD + sj_] * cs, cs, add) : 0.0;
    const bool okc = (g > 0.0) && is_finite(g);
    if (j < n && !okc) bad = 1;
    double d = 1.0, s = 1.0;
    if (j < N && okc) {
      d = __builtin_amdgcn_rsq(g);
      d = d * fma(-0.5 * g * d, d, 1.5);
      d = d * fma(-0.5 * g * d, d, 1.5);
      s = g * d;
    }
    dl[j] = d; sq[j] = s; sc[j] = cs * d; td[j] = add * d * d;
    if (a.dsc) a.dsc[(long)b * NPAD + j] = d;
  }
  if (a.colinfo) {
    __syncthreads();
    if (tid == 0) {
      double mn = __builtin_inf(), sm = 0.0;
      for (int j = 0; j < n; ++j) { const double v = sq[j]; mn = v < mn ? v : mn; sm = fma(v, v, sm); }
      a.colinfo[2 * (long)b] = mn; a.colinfo[2 * (long)b + 1] = sm;
    }
  }
  bad = block_or(bad, red);
  if (tid == 0) { pminsh = 1.0; flagsh = 0; }
  __syncthreads();
  if (bad) {
    if (tid == 0 && a.fb_mask) {
      a.fb_mask[b] = a.n + 1; { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }   // (the tree factors ALL n + 1 columns)
      if (a.path_out) a.path_out[b] = a.n + 1;
    }
    return;
  }
  // ROLES: wave 0 only runs the 16x16 chains (its registers hold the column / inverse vectors, no
  // tiles); waves 1..7 own the tiles.  LOOKAHEAD: in the trailing update of row block kb the owner
  // of the next diagonal tile updates it first, puts it into LDS and raises a flag; wave 0 starts
  // the chain of block kb + 1 on that flag while the other tiles are still being updated.  Both
  // loops pass the same two barriers per row block.
  constexpr int NWK = GR_NW - 1;                        // worker waves
  const int ntile = NT * (NT + 1) / 2;
  // zeros outside the factor: strictly lower tiles, and everything beyond 16 NT (sub-matrix use)
  auto zero_fill = [&]() {
    for (int r = w; r < NPAD; r += GR_NW) {
      const int cend = (r < 16 * NT) ? (r & ~15) : NPAD;
      for (int c = lane; c < cend; c += WAVE) Gb[(unsigned)(r * NPAD + c)] = 0.0;
      if (r < 16 * NT)
        for (int c = 16 * NT + lane; c < NPAD; c += WAVE) Gb[(unsigned)(r * NPAD + c)] = 0.0;
    }
  };
  if (w == 0) {
    __syncthreads();                                    // X: workers have read the source, Dt holds tile (0, 0)
    zero_fill();
    double pmin = 1.0;
    for (int kb = 0; kb < NT; ++kb) {
      if (kb > 0) {                                     // wait for the updated diagonal tile kb
        while (__hip_atomic_load(&flagsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kb)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      pmin = chol16_blocked3(Dt, Ri, n - 16 * kb, pmin);   // (chol16.h)
      if (a.rinv) {                                     // kept for the conditioning certificate
        double* ro = a.rinv + ((long)b * (NPAD / 16) + kb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[q * 64 + lane] = Ri[q * 64 + lane];
      }
      __syncthreads();                                  // B: R'_kk and its inverse are in LDS
      __syncthreads();                                  // C: (workers published the row block)
    }
    if (lane == 0) pminsh = pmin;
  } else {
    // tile table (cyclic over the worker waves: the shrinking trailing matrix stays balanced) and
    // the scaled source tiles -> accumulators; the source may alias the output, so everything is
    // read before anything is written
    const int ww = w - 1;
    int ti[SL], tj[SL];
    v4d acc[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
      int q = ww + NWK * t;
      const bool valid = q < ntile;
      int i = 0;
      while (valid && q >= NT - i) { q -= NT - i; ++i; }
      ti[t] = valid ? i : -1;
      tj[t] = valid ? i + q : -1;
      acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
      if (valid) {
        const int j = tj[t];
        const double scj = sc[16 * j + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          double v = 0.0;
          if (row < N && col < N) {
            int sr_ = src(row), sc_ = src(col);
            if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }
            v = Gs[(unsigned)(sr_ * NPAD + sc_)] * sc[row] * scj;
          }
          if (j == i && lr + 4 * g == lc) v += td[row];
          acc[t][g] = v;
        }
        if (i == 0 && j == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) Dt[(lr + 4 * g) * 16 + lc] = acc[t][g];
        }
      }
    }
    __syncthreads();                                    // X: all source reads done before the first store
    zero_fill();
    for (int kb = 0; kb < NT; ++kb) {
      __syncthreads();                                  // B: wave 0 finished the chain of block kb
      // c. the row block: R'_{kb,j} = R'_{kb,kb}^-T S_j -> LDS row buffer and (unscaled) to memory
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (ti[t] == kb) {
          const int j = tj[t];
          v4d X = {0.0, 0.0, 0.0, 0.0};
          if (j == kb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) X[g] = Dt[(lr + 4 * g) * 16 + lc];
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) X = gmfma(Ri[(4 * s + lr) * 16 + lc], acc[t][s], X);
          }
          const double sj = sq[16 * j + lc];
          const double dj = dl[16 * j + lc];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * kb + lr + 4 * g;
            const int colg = 16 * j + lc;
            double val = X[g] * sj;
            if (row >= n || row > colg || colg > n) val = 0.0;
            Gb[(unsigned)(row * NPAD + colg)] = val;
            // the operand of the trailing updates is what the left-looking kernel reads back: the
            // STORED entry times its column's equilibration — the two kernels agree bit for bit
            if (j != kb) Rrow[j * 256 + (lr + 4 * g) * 16 + lc] = val * dj;
          }
        }
      }
      __syncthreads();                                  // C: the row block is in the LDS buffer
      // d. trailing update: the next diagonal tile first (-> LDS, flag for wave 0), then the rest
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (ti[t] == kb + 1 && tj[t] == kb + 1) {
          const double* Ra = Rrow + ti[t] * 256 + lr * 16 + lc;
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[t] = gmfma(-Ra[64 * s], Ra[64 * s], acc[t]);
#pragma unroll
          for (int g = 0; g < 4; ++g) Dt[(lr + 4 * g) * 16 + lc] = acc[t][g];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(&flagsh, kb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
#pragma unroll
      for (int t = 0; t < SL; ++t) {
        if (ti[t] > kb && !(ti[t] == kb + 1 && tj[t] == kb + 1)) {
          const double* Ra = Rrow + ti[t] * 256 + lr * 16 + lc;
          const double* Rb = Rrow + tj[t] * 256 + lr * 16 + lc;
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[t] = gmfma(-Ra[64 * s], Rb[64 * s], acc[t]);
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0 && a.fb_mask) {
    const bool fail = !(pminsh >= GRAM_SMIN * GRAM_SMIN);
    a.fb_mask[b] = fail ? a.n + 1 : 0;
    if (a.path_out) a.path_out[b] = fail ? a.n + 1 : 0;
    if (fail) { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
  }
}

// ---- conditioning gate: a PROVEN bound on kappa_2 of the equilibrated system -----------------------
// The normal-equations path loses kappa_2(C) eps where C = R'^T R' is the equilibrated system matrix
// (unit diagonal) the step is solved from.  An estimate of sigma_min(R') by inverse iteration is a
// LOWER bound on ||R'^-1||, i.e. it can only err on the unsafe side.  This kernel computes an UPPER
// bound instead, from the explicit inverse:
//     Y = R'^-T   (lower triangular; 16 x 16 tiles by FP64 MFMA, the inverses of the diagonal tiles
//                  come from the Cholesky kernel:  Y_ii = R'_ii^-T,
//                  Y_ij = -R'_ii^-T sum_{k=j}^{i-1} R'_ki^T Y_kj   for j < i)
//     1 / lambda_min(C) = ||Y||_2^2 <= ||Y||_1 ||Y||_inf ,   lambda_max(C) = ||R'||_2^2 <= ||R'||_1 ||R'||_inf
//     K2 = ||R'||_1 ||R'||_inf ||Y||_1 ||Y||_inf  >=  kappa_2(C)
// (all four norms are exact sums of absolute values, accumulated in a fixed order) and keeps the
// problem on the normal-equations path only if K2 <= GRAM_K2_MAX.  DESIGN.md 3.0 has the error bound
// this gives for the step.  NWP waves per problem as in gram_chol_kernel.
template <int NWP>
__global__ __launch_bounds__(GR_NT, 4) void gram_cond_kernel(GramCholArgs a) {
  constexpr int PT = WAVE * NWP;
  constexpr int PPW = GR_NW / NWP;
  constexpr int UMAX = (NWP == 8) ? 3 : 5;              // column tiles of a row block per wave
  extern __shared__ double sh_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int pslot = wv / NWP;
  const int pidx = (int)blockIdx.x * PPW + pslot;
  if (pidx >= a.count) return;                          // (NWP == 1 only: wave-uniform)
  const int b = pidx;
  const int tid = (int)threadIdx.x % PT, lane = tid & 63;
  const int w = wv % NWP;
  const int lr = lane >> 4, lc = lane & 15;
  if (a.mask && a.mask[b] <= 1) return;
  if (a.fb_mask[b] != 0) return;                        // already failed on a pivot
  if (a.cert_done && a.cert_done[b]) return;            // already proven inside the factor kernel (N <= 80)
  const int NPAD = a.NPAD;
  const int n = a.ncols_dev ? a.ncols_dev[b] - 1 : a.n;
  if (n <= 0) return;
  const int NTn = (n + 15) / 16;
  auto psync = [&]() {
    if (NWP == 8) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };
  double* sh = sh_all + (size_t)pslot * (6 * (size_t)NPAD + 16 * NWP + 64);
  double* dl = sh;                      // [NPAD] column scales: R'[i][j] = T[i][j] dl[j]
  double* cs4 = dl + NPAD;              // [NPAD][4] column sums of |Y|, one slot per lane row
  double* rs = cs4 + 4 * NPAD;          // [NWP][16] row-sum partials of the current block row
  double* vals = rs + 16 * NWP;         // [64] reduction scratch
  double* rowsR = vals + 64;            // [NPAD] row sums of |R'|
  const double* T = a.G + (long)b * NPAD * NPAD;
  double* Y = a.ywork + (long)b * NPAD * NPAD;
  const double* Rinv = a.rinv + (long)b * (NPAD / 16) * 256;
  for (int j = tid; j < NPAD; j += PT) dl[j] = a.dsc[(long)b * NPAD + j];
  psync();
  auto reduce_max = [&](double v) -> double {           // max over the threads of this problem
    v = wave_max(v);
    if (NWP == 1) return v;
    psync();
    if (lane == 0) vals[w] = v;
    psync();
    double t = vals[0];
    for (int q = 1; q < NWP; ++q) t = fmax(t, vals[q]);
    return t;
  };
  // ---- ||R'||_1 (thread per column) and ||R'||_inf (wave per row) ----
  double r1 = 0.0;
  for (int j = tid; j < n; j += PT) {
    double sum = 0.0;
    for (int i0 = 0; i0 <= j; i0 += 8) {
      double rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rv[u] = T[(long)((i0 + u <= j) ? i0 + u : j) * NPAD + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (i0 + u <= j) sum += fabs(rv[u]);
    }
    r1 = fmax(r1, sum * dl[j]);
  }
  r1 = reduce_max(r1);
  if constexpr (NWP == 1) {
    // one wave per problem: four rows at a time, one per 16-lane group (64 sequential wave
    // reductions at n = 64 were 20 of this kernel's 46 us)
    for (int i0 = 0; i0 < n; i0 += 4) {
      const int i = i0 + lr;
      double sum = 0.0;
      if (i < n)
        for (int j = i + lc; j < n; j += 16) sum += fabs(T[(long)i * NPAD + j]) * dl[j];
      sum = row16_sum(sum);
      if (lc == 0 && i < n) rowsR[i] = sum;
    }
  } else {
    for (int i = w; i < n; i += NWP) {
      double sum = 0.0;
      for (int j = i + lane; j < n; j += WAVE) sum += fabs(T[(long)i * NPAD + j]) * dl[j];
      sum = wave_sum(sum);
      if (lane == 0) rowsR[i] = sum;
    }
  }
  psync();
  double rinf = 0.0;
  for (int i = tid; i < n; i += PT) rinf = fmax(rinf, rowsR[i]);
  rinf = reduce_max(rinf);
  // ---- Y = R'^-T by block rows; row and column sums of |Y| on the way ----
  double csum[UMAX];
#pragma unroll
  for (int u = 0; u < UMAX; ++u) csum[u] = 0.0;
  double rmax = 0.0;                                    // threads 0..15: max over block rows of "their" row
  for (int i = 0; i < NTn; ++i) {
    double rsum[4] = {0.0, 0.0, 0.0, 0.0};
    const double* Ri = Rinv + (long)i * 256;
    const double dli = dl[16 * i + lc];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = w + NWP * u;
      if (j <= i) {
        v4d Yt = {0.0, 0.0, 0.0, 0.0};
        if (j == i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) Yt[g] = Ri[lc * 16 + lr + 4 * g];         // (R'_ii^-1)^T
        } else {
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          for (int k = j; k < i; ++k) {
            double av[4], bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const long ro = (long)(16 * k + 4 * s + lr) * NPAD;
              av[s] = T[ro + 16 * i + lc];
              bv[s] = Y[ro + 16 * j + lc];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = gmfma(av[s] * dli, bv[s], acc);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) Yt = gmfma(-Ri[(4 * s + lr) * 16 + lc], acc[s], Yt);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          const double v = (row < n && col < n) ? Yt[g] : 0.0;
          Y[(long)row * NPAD + col] = v;
          const double av_ = fabs(v);
          rsum[g] += row16_sum(av_);
          csum[u] += av_;
        }
      }
    }
    if (lc == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) rs[w * 16 + lr + 4 * g] = rsum[g];
    }
    psync();                                            // row sums in LDS; Y row block i visible
    if (tid < 16) {
      double t = 0.0;
      for (int q = 0; q < NWP; ++q) t += rs[q * 16 + tid];
      rmax = fmax(rmax, t);
    }
    psync();
  }
#pragma unroll
  for (int u = 0; u < UMAX; ++u) {
    const int j = w + NWP * u;
    if (j < NTn) cs4[(16 * j + lc) * 4 + lr] = csum[u];
  }
  psync();
  double y1 = 0.0;
  for (int c = tid; c < n; c += PT)
    y1 = fmax(y1, (cs4[4 * c] + cs4[4 * c + 1]) + (cs4[4 * c + 2] + cs4[4 * c + 3]));
  y1 = reduce_max(y1);
  const double yinf = reduce_max(tid < 16 ? rmax : 0.0);
  double k2 = (r1 * rinf) * (y1 * yinf);
  if (!(k2 <= GRAM_K2_MAX)) {                            // (uniform over the problem's threads)
    // The 1- / inf-norm products overestimate kappa_2 by 10 ... 1000 (profiles/r02p_gate_calibration.txt).
    // Second, tighter proven bound for a problem they reject:  lambda_max(C) <= ||C||_F  and
    // 1 / lambda_min(C) = ||C^-1||_2 <= ||C^-1||_F  with  C^-1 = Y^T Y  formed tile by tile (MFMA; only
    // its sum of squares is kept) and C rebuilt from the source Gram with the Cholesky's own scalings.
    // Measured overestimate 4 ... 30 on the ill-conditioned families.  Sums in a fixed order.
    const int* gidx = a.gather ? a.gather + (long)b * a.stride_vec : nullptr;
    auto src = [&](int i) -> int { return gidx ? (i < n ? gidx[i] : a.n) : i; };
    const double* Gs = a.Gsrc + (long)b * NPAD * NPAD;
    const double* csv = a.colscale ? a.colscale + (long)b * a.stride_vec : nullptr;
    const double* edv = a.diag_vec ? a.diag_vec + (long)b * a.stride_vec : nullptr;
    double* scl = cs4;                  // [NPAD] cs_j dl_j   (cs4 is free now)
    double* tdl = cs4 + NPAD;           // [NPAD] e_j^2 dl_j^2
    psync();
    for (int j = tid; j < NPAD; j += PT) {
      const double cs = (csv && j < n) ? csv[j] : 1.0;
      const double ej = (edv && j < n) ? edv[j] : 0.0;
      scl[j] = cs * dl[j];
      tdl[j] = (ej * ej) * dl[j] * dl[j];
    }
    psync();
    double cf = 0.0, zf = 0.0;          // this wave's share of ||C||_F^2 / ||C^-1||_F^2
    int q = 0;
    for (int j = 0; j < NTn; ++j) {
      for (int i = 0; i <= j; ++i, ++q) {
        if (q % NWP != w) continue;     // (wave-uniform)
        const double wgt = (i == j) ? 1.0 : 2.0;
        double c2 = 0.0;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * i + lr + 4 * g, col = 16 * j + lc;
          double v = 0.0;
          if (row < n && col < n) {
            int sr_ = src(row), sc_ = src(col);
            if (sr_ > sc_) { const int t_ = sr_; sr_ = sc_; sc_ = t_; }
            v = Gs[(long)sr_ * NPAD + sc_] * scl[row] * scl[col];
            if (row == col) v += tdl[row];
          }
          c2 = fma(v, v, c2);
        }
        for (int k = j; k < NTn; ++k) {
          double av[4], bv[4];
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const long ro = (long)(16 * k + 4 * s2 + lr) * NPAD;
            av[s2] = Y[ro + 16 * i + lc];
            bv[s2] = Y[ro + 16 * j + lc];
          }
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc = gmfma(av[s2], bv[s2], acc);
        }
        const double z2 = (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
        cf = fma(wgt, wave_sum(c2), cf);
        zf = fma(wgt, wave_sum(z2), zf);
      }
    }
    if (NWP > 1) {
      psync();
      if (lane == 0) { vals[w] = cf; vals[8 + w] = zf; }
      psync();
      cf = 0.0; zf = 0.0;
      for (int qq = 0; qq < NWP; ++qq) { cf += vals[qq]; zf += vals[8 + qq]; }
    }
    const double k2f = sqrt(cf) * sqrt(zf);
    if (k2f < k2) k2 = k2f;
  }
  if (tid == 0) {
    if (a.k2_out) a.k2_out[b] = k2;
    if (!(k2 <= GRAM_K2_MAX)) {                          // (NaN fails)
      a.fb_mask[b] = a.n + 1;
      if (a.path_out) a.path_out[b] = a.n + 1;
      { const int fi_ = atomicAdd(a.fail_count, 1); if (a.fail_list) a.fail_list[fi_] = b; }
    }
  }
}

bool gram_supported(int m, int n) {
  const int NT = (n + 1 + 15) / 16;
  return NT <= 17 && m >= n && n >= 1;
}
// Row chunks whose size is a function of m ALONE (never of the batch size): the summation order of
// a problem's Gram (and so every bit of its result) does not depend on how many problems share the
// launch.  2048 rows; 1024 for very tall problems (one 250 000 x 128 row block of BASELINE config 5:
// 245 workgroups fill the 256 CUs, 123 leave half of them idle).
static int gram_chunk_rows(int m) { return m > 131072 ? 1024 : 2048; }
int gram_chunks(int B, int m) {
  (void)B;
  const int r = gram_chunk_rows(m);
  const int c = (m + r - 1) / r;
  return c < 1 ? 1 : c;
}

hipError_t launch_gram(const GramArgs& a_in, int chunks, int B, hipStream_t s, double* Gfinal, bool* fused) {
  GramArgs a = a_in;
  if (fused) *fused = false;
  const int NT = (a.n + 1 + 15) / 16;
  a.rows_per_chunk = chunks > 1 ? gram_chunk_rows(a.m) : a.m;
  const size_t lds = sizeof(double) * 2 * GR_RC * (size_t)gram_ldx(NT);
  // slot variant a wave needs for `nt` column tiles
  auto slots_for = [](int nt) {
    const int per_ = (nt * (nt + 1) / 2 + GR_NW - 1) / GR_NW;
    return per_ <= 4 ? 4 : per_ <= 8 ? 8 : per_ <= 12 ? 12 : per_ <= 17 ? 17 : 20;
  };
  // n % 16 == 0: the rhs column would be a tile column of its own (NT tiles for one useful column
  // each) — it is accumulated in registers beside the MFMA stream instead (n = 256: 153 -> 136 tiles)
  a.rhs_valu = ((a.n + 15) / 16 < NT) ? 1 : 0;
  (void)slots_for;
  const int NTJ = a.rhs_valu ? (a.n + 15) / 16 : NT;
  const int ntile = NTJ * (NTJ + 1) / 2;
#define BLSQ_GRAM_LAUNCH(SL, CB)                                                              \
  do {                                                                                        \
    static std::atomic<size_t> granted[64];                                                   \
    hipError_t ge = gram_grant_lds(gram_kernel<SL, CB>, lds, granted);                        \
    if (ge != hipSuccess) return ge;                                                          \
    hipLaunchKernelGGL((gram_kernel<SL, CB>), dim3(chunks, B, tg), dim3(GR_NT), lds, s, a);   \
  } while (0)
#define BLSQ_GRAM_DIRECT(NTT, RHS)                                                            \
  do {                                                                                        \
    constexpr int nt_ = (NTT) * ((NTT) + 1) / 2;                                              \
    const size_t dl_ = sizeof(double) * ((nt_ < 5 ? nt_ : 5) * GR_NW * 256 + 64);            \
    static std::atomic<size_t> granted[64];                                                   \
    hipError_t ge = gram_grant_lds(gram_direct_kernel<NTT, RHS>, dl_, granted);               \
    if (ge != hipSuccess) return ge;                                                          \
    hipLaunchKernelGGL((gram_direct_kernel<NTT, RHS>), dim3(chunks, B), dim3(GR_NT), dl_, s, a); \
    return hipGetLastError();                                                                 \
  } while (0)
  {
    const char* env = getenv("BLSQ_GRAM_DIRECT_MAX_NT");     // tuning / tests: 0 disables
    const int dmax = env ? atoi(env) : 4;                      // measured: direct wins up to 4 column tiles
    // n a multiple of 16: the tiles cover J^T J only (n / 16 column tiles), the rhs column is
    // accumulated from the same fragments (n = 64: 10 tiles instead of 15, and still no LDS staging)
    if (a.rhs_valu && NTJ <= dmax) {
      if (NTJ == 1) BLSQ_GRAM_DIRECT(1, true);
      else if (NTJ == 2) BLSQ_GRAM_DIRECT(2, true);
      else if (NTJ == 3) BLSQ_GRAM_DIRECT(3, true);
      else if (NTJ == 4) BLSQ_GRAM_DIRECT(4, true);
    }
    if (!a.rhs_valu && NT <= dmax) {
      if (NT <= 1) BLSQ_GRAM_DIRECT(1, false);
      else if (NT == 2) BLSQ_GRAM_DIRECT(2, false);
      else if (NT == 3) BLSQ_GRAM_DIRECT(3, false);
      else if (NT == 4) BLSQ_GRAM_DIRECT(4, false);
    }
  }
#undef BLSQ_GRAM_DIRECT
  {
    // 8 column tiles of J^T J (n = 113 .. 128): the k-split static-tile kernel, for EVERY batch size
    // (its summation order defines the result for these widths).  BLSQ_GRAM8 = 0: the generic kernel.
    const char* g8e = getenv("BLSQ_GRAM8");
    if (NTJ == 8 && !(g8e && g8e[0] == '0')) {
      const size_t need = sizeof(double) * 4 * 9 * 256;          // partial tiles of the odd k-steps
      const size_t l8 = lds > need ? lds : need;
      if (a.rhs_valu) {
        static std::atomic<size_t> granted[64];
        hipError_t ge = gram_grant_lds(gram8_kernel<true>, l8, granted);
        if (ge != hipSuccess) return ge;
        hipLaunchKernelGGL((gram8_kernel<true>), dim3(chunks, B, 1), dim3(GR_NT), l8, s, a);
      } else {
        static std::atomic<size_t> granted[64];
        hipError_t ge = gram_grant_lds(gram8_kernel<false>, l8, granted);
        if (ge != hipSuccess) return ge;
        hipLaunchKernelGGL((gram8_kernel<false>), dim3(chunks, B, 1), dim3(GR_NT), l8, s, a);
      }
      return hipGetLastError();
    }
  }
  const int ncb = (a.n + 63) / 64;
  // tile groups: enough workgroups to occupy the CUs when the batch is small (results identical)
  int tg = 1;
  {
    const char* tge = getenv("BLSQ_GRAM_TILE_GROUPS");    // (tests compare splits bit for bit)
    const int tenv = tge ? atoi(tge) : 0;
    const long wgs = (long)chunks * B;
    // (only when the row chunks alone leave most CUs idle: every group re-reads the rows)
    tg = tenv > 0 ? tenv : (wgs <= 64 ? (int)((256 + wgs - 1) / wgs) : 1);
    if (tg > 8) tg = 8;
    if (tg > ntile) tg = ntile;
    if (tg < 1) tg = 1;
  }
  {
    // 16 column tiles of J^T J (n = 241 .. 256) and one workgroup per row chunk: the kernel with
    // static tile rows per wave (BLSQ_GRAM16 = 0 keeps the generic one: tests compare the two)
    const char* g16e = getenv("BLSQ_GRAM16");
    const int g16_env = g16e ? atoi(g16e) : 1;
    if (NTJ == 16 && tg == 1 && g16_env != 0 && a.m >= 1) {
      // two row chunks and enough problems to fill the device with one workgroup each: both chunks by
      // the same workgroup, summed in the kernel straight into the final slot (bit-identical to the
      // reduction pass: BLSQ_GRAM_PAIR = 0 keeps that)
      const char* pe = getenv("BLSQ_GRAM_PAIR");
      const bool pair = Gfinal && chunks == 2 && B >= 256 && !(pe && pe[0] == '0');
      if (pair) { a.Gscr = a.G; a.G = Gfinal; if (fused) *fused = true; }
      const dim3 grid(pair ? 1 : chunks, B, 1);
#define BLSQ_GRAM16(RHS_, PAIR_)                                                              \
  do {                                                                                        \
    static std::atomic<size_t> granted[64];                                                   \
    hipError_t ge = gram_grant_lds(gram16_kernel<RHS_, PAIR_>, lds, granted);                 \
    if (ge != hipSuccess) return ge;                                                          \
    hipLaunchKernelGGL((gram16_kernel<RHS_, PAIR_>), grid, dim3(GR_NT), lds, s, a);           \
  } while (0)
      if (a.rhs_valu) { if (pair) BLSQ_GRAM16(true, true); else BLSQ_GRAM16(true, false); }
      else { if (pair) BLSQ_GRAM16(false, true); else BLSQ_GRAM16(false, false); }
#undef BLSQ_GRAM16
      return hipGetLastError();
    }
  }
  const int per = (((ntile + tg - 1) / tg) + GR_NW - 1) / GR_NW;   // tile slots a wave needs
  if (per <= 4) {                                       // n <= 111, or tile groups
    if (ncb <= 1) BLSQ_GRAM_LAUNCH(4, 1);
    else if (ncb <= 2) BLSQ_GRAM_LAUNCH(4, 2);
    else BLSQ_GRAM_LAUNCH(4, 5);
  } else if (per <= 8) {                                // n <= 159, or tile groups
    if (ncb <= 2) BLSQ_GRAM_LAUNCH(8, 2);
    else if (ncb <= 3) BLSQ_GRAM_LAUNCH(8, 3);
    else BLSQ_GRAM_LAUNCH(8, 5);
  } else if (per <= 12) {                               // n <= 207, or tile groups
    if (ncb <= 3) BLSQ_GRAM_LAUNCH(12, 3);
    else if (ncb <= 4) BLSQ_GRAM_LAUNCH(12, 4);
    else BLSQ_GRAM_LAUNCH(12, 5);
  } else if (per <= 17) {                               // n <= 256
    if (ncb <= 4) BLSQ_GRAM_LAUNCH(17, 4); else BLSQ_GRAM_LAUNCH(17, 5);
  } else {
    if (ncb <= 4) BLSQ_GRAM_LAUNCH(20, 4); else BLSQ_GRAM_LAUNCH(20, 5);
  }
#undef BLSQ_GRAM_LAUNCH
  return hipGetLastError();
}
hipError_t launch_gram_reduce(const double* Gpart, int chunks, int NPAD, double* Gout,
                              const int* mask, int B, hipStream_t s) {
  const long sz = (long)NPAD * NPAD;
  if (chunks > 32)
    hipLaunchKernelGGL(gram_reduce_wide_kernel, dim3((unsigned)((sz + GRED_W - 1) / GRED_W), B),
                       dim3(GRED_G * GRED_W), 0, s, Gpart, chunks, NPAD, Gout, mask);
  else
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((sz + 255) / 256), B), dim3(256), 0, s,
                       Gpart, chunks, NPAD, Gout, mask);
  return hipGetLastError();
}
hipError_t launch_gram_chol(const GramCholArgs& a_in, int B, hipStream_t s) {
  GramCholArgs a = a_in;
  a.count = B;
  const size_t per = sizeof(double) * (4 * (size_t)a.NPAD + 512);
  if (a.NPAD <= 80) {                                   // one wave per problem, eight per workgroup
    // (register-resident right-looking kernel; BLSQ_CHOL_REG = 0: the left-looking one-wave kernel)
    const char* rge = getenv("BLSQ_CHOL_REG");
    if (rge && rge[0] == '0')
      hipLaunchKernelGGL(gram_chol_kernel<1>, dim3((B + GR_NW - 1) / GR_NW), dim3(GR_NT), per * GR_NW,
                         s, a);
    else {
      const size_t per_reg = sizeof(double) * (4 * (size_t)a.NPAD + 256 + 5 * 256 + 64);
      static std::atomic<size_t> granted[64];
      hipError_t ge = gram_grant_lds(gram_chol_reg_kernel, per_reg * REG_NW, granted);
      if (ge != hipSuccess) return ge;
      hipLaunchKernelGGL(gram_chol_reg_kernel, dim3((B + REG_NW - 1) / REG_NW), dim3(REG_NT), per_reg * REG_NW,
                         s, a);
    }
  } else {
    // Right-looking register variant: 0.22 ms per problem on a CU of its own against 0.27 ms for
    // the left-looking kernel, but one workgroup per CU instead of two — so it serves the launches
    // that cannot fill the CUs twice anyway (the Newton rounds), the left-looking one the rest.
    // BLSQ_CHOL_RL = 0 / 1 forces either.
    // The two kernels agree bit for bit (same operands, same order), so the choice is speed only.
    const char* rle = getenv("BLSQ_CHOL_RL");          // (read per launch: tests compare the two)
    const int rl_env = rle ? (rle[0] == '0' ? 0 : 1) : -1;
    constexpr int ncu = 256;                           // MI355X: 8 XCDs x 32 CUs
    const bool rl = rl_env >= 0 ? rl_env != 0 : (a.expect > 0 ? a.expect : B) <= ncu;
    if (rl) {
      const size_t lds = per + sizeof(double) * 256 * (size_t)(a.NPAD / 16);
      static std::atomic<size_t> granted[64];
      hipError_t ge = gram_grant_lds(gram_chol_rl_kernel<22>, lds, granted);
      if (ge != hipSuccess) return ge;
      hipLaunchKernelGGL(gram_chol_rl_kernel<22>, dim3(B), dim3(GR_NT), lds, s, a);
    } else {
      hipLaunchKernelGGL(gram_chol_kernel<8>, dim3(B), dim3(GR_NT), per, s, a);
    }
  }
  return hipGetLastError();
}
hipError_t launch_gram_gate(const GramCholArgs& a_in, int B, hipStream_t s) {
  GramCholArgs a = a_in;
  a.count = B;
  const size_t per1 = sizeof(double) * (6 * (size_t)a.NPAD + 16 * 1 + 64);
  const size_t per8 = sizeof(double) * (6 * (size_t)a.NPAD + 16 * 8 + 64);
  if (a.NPAD <= 80) {                                   // one wave per problem, eight per workgroup
    hipLaunchKernelGGL(gram_cond_kernel<1>, dim3((B + GR_NW - 1) / GR_NW), dim3(GR_NT), per1 * GR_NW, s, a);
  } else {
    hipLaunchKernelGGL(gram_cond_kernel<8>, dim3(B), dim3(GR_NT), per8, s, a);
  }
  return hipGetLastError();
}

}  // namespace blsq
