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

// ---- wave reductions (all 64 lanes end with the result) ------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
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

}  // namespace blsq
