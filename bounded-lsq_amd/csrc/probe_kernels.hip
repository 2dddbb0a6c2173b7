// Diagnostic probes (blsq_debug_probe): the MEASURED peaks the roofline fractions are quoted
// beside (SURVEY.md 8d asks for "an MFMA-f64 probe" and "a copy kernel" on the box), and the
// calibration kernel for the SQ MFMA counters (tools/pmc_mfma.py): a launch that executes a
// known number of v_mfma_f64_16x16x4_f64 and nothing else.
#include "blsq_device.h"
#include "blsq_kernels.h"

namespace blsq {

// Every wave: `iters` rounds of NACC independent FP64 MFMAs on operands that stay in registers
// (pseudo-random, non-zero: the clock the chip holds depends on the data).
template <int NACC>
__global__ __launch_bounds__(512) void mfma_f64_probe_kernel(int iters, double* sink) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = tid * 2654435761u + 12345u;
  double a[NACC], b[NACC];
  v4d acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    h = h * 1664525u + 1013904223u;
    a[t] = 0.5 + (double)(h >> 8) * (1.0 / 16777216.0);
    h = h * 1664525u + 1013904223u;
    b[t] = (0.5 + (double)(h >> 8) * (1.0 / 16777216.0)) * 1e-3;
    acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc[t], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 123.456) sink[0] = s;                       // never true: keeps the loop alive
}

// streaming copy, 16 B per lane per access, grid-stride
typedef double v2d_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void copy_probe_kernel(const v2d_t* __restrict__ src,
                                                         v2d_t* __restrict__ dst, long n2) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride)
    dst[i] = __builtin_nontemporal_load(src + i);
}

hipError_t launch_mfma_probe(int waves_per_simd, int iters, double* sink, long* n_mfma,
                             hipStream_t s) {
  constexpr int NACC = 8;
  const int threads = 256 * (waves_per_simd < 1 ? 1 : (waves_per_simd > 2 ? 2 : waves_per_simd));
  const int grid = 256;                                // one workgroup per CU
  hipLaunchKernelGGL(mfma_f64_probe_kernel<NACC>, dim3(grid), dim3(threads), 0, s, iters, sink);
  *n_mfma = (long)grid * (threads / 64) * (long)iters * NACC;
  return hipGetLastError();
}

hipError_t launch_copy_probe(const void* src, void* dst, size_t bytes, hipStream_t s) {
  hipLaunchKernelGGL(copy_probe_kernel, dim3(256 * 16), dim3(256), 0, s, (const v2d_t*)src,
                     (v2d_t*)dst, (long)(bytes / 16));
  return hipGetLastError();
}

}  // namespace blsq
