// Development probe: semantics of v_permlane16_swap / v_permlane32_swap on gfx950 and the raw
// accuracy of v_rsq_f64 (how many Newton steps the Cholesky pivots need).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void perm_kernel(unsigned* out) {
  const unsigned a = threadIdx.x, b = threadIdx.x + 1000;
  auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r32[0]; out[64 + threadIdx.x] = r32[1];
  out[128 + threadIdx.x] = r16[0]; out[192 + threadIdx.x] = r16[1];
}
__global__ void rsq_kernel(const double* x, double* y0, double* y1, int n) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= n) return;
  const double d = x[i];
  const double r = __builtin_amdgcn_rsq(d);
  y0[i] = r;
  y1[i] = r * fma(-0.5 * d * r, r, 1.5);
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(perm_kernel, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane32_swap [0]", "permlane32_swap [1]", "permlane16_swap [0]", "permlane16_swap [1]"};
  for (int k = 0; k < 4; ++k) { printf("%s: rows", nm[k]); for (int r = 0; r < 4; ++r) printf("  %u..", h[64 * k + 16 * r]); printf("\n"); }
  const int n = 1 << 20;
  std::vector<double> x(n), y0(n), y1(n);
  for (int i = 0; i < n; ++i) x[i] = std::exp(std::log(1e-6) + (std::log(1e6) - std::log(1e-6)) * (i + 0.37) / n);
  double *dx, *d0, *d1; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(rsq_kernel, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
  hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / sqrtl((long double)x[i]);
    e0 = std::fmax(e0, (double)fabsl((y0[i] - ex) / ex)); e1 = std::fmax(e1, (double)fabsl((y1[i] - ex) / ex));
  }
  printf("v_rsq_f64 max rel error %.3e (2^%.1f); after one Newton step %.3e (%.2f ulp of 2^-53)\n", e0, std::log2(e0), e1, e1 / 1.11e-16);
  return 0;
}
