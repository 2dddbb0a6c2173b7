// Shared by gram_kernels.hip (the Gram kernels) and chol_kernels.hip (Cholesky, certificate, fused
// Newton rounds): launch geometry, the LDS grant helper, the FP64 MFMA wrapper and the explicit LDS
// read helpers with counted waits.
#pragma once
#include <atomic>
#include <stdlib.h>
#include <type_traits>

#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

static constexpr int GR_NT = 512;
static constexpr int GR_NW = GR_NT / WAVE;
static constexpr int GR_RC = 32;          // rows per staged chunk (8 MFMA k-steps)
static constexpr int REG_NW = 4;          // one-wave-per-problem kernels (N <= 80): problems (waves) per workgroup —
static constexpr int REG_NT = REG_NW * WAVE;   // 1024 problems spread over 256 workgroups instead of 128
// Problem of wave `wv` of workgroup `w` in the one-wave-per-problem kernels.  Workgroups are dealt round-robin over
// the eight XCDs and the Gram kernel that has just written the problem's matrix ran as workgroup b (one per problem):
// b and the workgroup that reads it back agree mod 8, so the read finds the matrix in the L2 it was written through.
__device__ __forceinline__ int reg_problem(int w, int wv) { return (((w >> 3) * REG_NW + wv) << 3) | (w & 7); }
static inline unsigned reg_grid(int B) { return (unsigned)(((B + 8 * REG_NW - 1) / (8 * REG_NW)) * 8); }
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

}  // namespace blsq
