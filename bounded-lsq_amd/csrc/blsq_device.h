// Device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blsq {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int TILE = 16;          // f64 MFMA tile edge (v_mfma_f64_16x16x4_f64)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- DPP helpers ----------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // all rows/banks enabled and every source lane valid: no `old` value needed
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
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
__device__ __forceinline__ double read_lane(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// ---- wave reductions (all 64 lanes end with the result) ------------------
// DPP butterfly inside each 16-lane row, then the 4 row totals through SGPRs:
// no LDS crossbar (ds_bpermute), fixed order (deterministic).
__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
// Reduce SIXTEEN per-lane values across the wave at once (transposed butterfly:
// at every step a lane hands the half it does not keep to its partner, so the
// 16 totals cost 8+4+2+1 exchanges instead of 16 x 4).  On return v[0] of lane l
// holds the wave total of the value with index wave_sum16_index(l); all 64 lanes
// are valid.  Must be called in wave-uniform control flow.  Fixed order.
__device__ __forceinline__ int wave_sum16_index(int lane) {
  return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
}
__device__ __forceinline__ void wave_sum16(double (&v)[16]) {
  const int lane = threadIdx.x & 63;
  const bool k0 = (lane & 1) == 0, k1 = (lane & 2) == 0, k2 = (lane & 4) == 0,
             k3 = (lane & 8) == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {                       // partner = lane ^ 1
    const double send = k0 ? v[8 + i] : v[i];
    const double keep = k0 ? v[i] : v[8 + i];
    v[i] = keep + dpp_mov<0xB1>(send);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {                       // partner = lane ^ 2
    const double send = k1 ? v[4 + i] : v[i];
    const double keep = k1 ? v[i] : v[4 + i];
    v[i] = keep + dpp_mov<0x4E>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {                       // partner = lane ^ 4
    const double send = k2 ? v[2 + i] : v[i];
    const double keep = k2 ? v[i] : v[2 + i];
    const double up = dpp_mov<0x104>(send);           // row_shl:4  (from lane + 4)
    const double dn = dpp_mov<0x114>(send);           // row_shr:4  (from lane - 4)
    v[i] = keep + (k2 ? up : dn);
  }
  {                                                   // partner = lane ^ 8
    const double send = k3 ? v[1] : v[0];
    const double keep = k3 ? v[0] : v[1];
    const double up = dpp_mov<0x108>(send);           // row_shl:8
    const double dn = dpp_mov<0x118>(send);           // row_shr:8
    v[0] = keep + (k3 ? up : dn);
  }
  {                                                   // rows: lane ^ 16, then lane ^ 32
    double t = v[0];
    int lo = __builtin_amdgcn_ds_swizzle(__double2loint(t), 0x401F);
    int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(t), 0x401F);
    t += __hiloint2double(hi, lo);
    const int addr = (lane ^ 32) << 2;
    lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(t));
    hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(t));
    v[0] = t + __hiloint2double(hi, lo);
  }
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
  return v;
}

// ---- block reductions through LDS ---------------------------------------
// `red` must hold >= 32 doubles.  Every thread gets the result.  The sum
// order is fixed (lane tree, then wave 0..W-1) so results are deterministic.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int nw = (blockDim.x + WAVE - 1) / WAVE;
  __syncthreads();                      // red[] may still be read by a previous call
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}
// NaN-propagating max/min (numpy's np.max / np.min / norm(inf) semantics).
__device__ __forceinline__ double nanmax2(double a, double b) {
  return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b);
}
__device__ __forceinline__ double nanmin2(double a, double b) {
  return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b);
}
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nanmax2(v, __shfl_xor(v, o, WAVE));
  const int nw = (blockDim.x + WAVE - 1) / WAVE;
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  double t = red[0];
  for (int w = 1; w < nw; ++w) t = nanmax2(t, red[w]);
  return t;
}
__device__ __forceinline__ double block_min(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nanmin2(v, __shfl_xor(v, o, WAVE));
  const int nw = (blockDim.x + WAVE - 1) / WAVE;
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  double t = red[0];
  for (int w = 1; w < nw; ++w) t = nanmin2(t, red[w]);
  return t;
}
__device__ __forceinline__ int block_or(int v, double* red) {
  int any = __any(v) ? 1 : 0;
  const int nw = (blockDim.x + WAVE - 1) / WAVE;
  int* r = reinterpret_cast<int*>(red);
  __syncthreads();
  if (lane_id() == 0) r[wave_id()] = any;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < nw; ++w) t |= r[w];
  return t;
}

// ---- IEEE helpers ---------------------------------------------------------
// nextafter(a, b) exactly as C99 / numpy (bounds.py:94,99 use np.nextafter).
__device__ __forceinline__ double next_after(double a, double b) {
  if (a != a || b != b) return a + b;
  if (a == b) return b;
  if (a == 0.0) {
    const uint64_t tiny = 1ull | (b < 0.0 ? 0x8000000000000000ull : 0ull);
    return __longlong_as_double((long long)tiny);
  }
  long long ia = __double_as_longlong(a);
  if ((a < b) == (a > 0.0)) ia += 1; else ia -= 1;
  return __longlong_as_double(ia);
}
__device__ __forceinline__ double sign_of(double v) {   // np.sign for finite / inf
  return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0);
}
__device__ __forceinline__ bool is_finite(double v) {
  return (v - v) == 0.0;
}

// ---- LDS-DMA (global -> LDS without VGPR staging) ------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// One LDS-DMA instruction: 64 lanes x 16 B, global (per-lane address) -> LDS
// (wave-uniform base + 16 lane).  Completion is tracked by vmcnt, in issue order.
// The source is (wave-uniform base) + (per-lane 32-bit byte offset): the SGPR-base
// addressing form, so a slot costs no 64-bit address VGPRs.
__device__ __forceinline__ void glds16(const double* base_uniform, unsigned byte_off_lane,
                                       double* lds_uniform) {
  __builtin_amdgcn_global_load_lds((gptr_t*)((const char*)base_uniform + byte_off_lane),
                                   (lptr_t*)lds_uniform, 16, 0, 0);
}

// workgroup barrier that orders LDS traffic only: DMA loads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace blsq
