// Stand-alone check and timing of the two 16 x 16 Cholesky + inverse chains of csrc/chol16.h
//   hipcc --offload-arch=gfx950 -O3 -I bounded-lsq_amd/csrc tools/chol16_probe.hip -o gpurun_out/chol16_probe
#include "chol16.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace blsq;

template <int V>
__global__ __launch_bounds__(64) void chain_kernel(int iters, int nlive, const double* A, double* outD,
                                                   double* outR, double* outP) {
  __shared__ double Dt[256], Ri[256], scr[64];
  const int lane = threadIdx.x;
  double pmin = 1.0;
  for (int it = 0; it < iters; ++it) {
    for (int i = lane; i < 256; i += 64) Dt[i] = A[blockIdx.x * 256 + i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (V == 1) pmin = chol16_columns(Dt, Ri, nlive, pmin);
    else if (V == 2) pmin = chol16_blocked(Dt, Ri, scr, nlive, pmin);
    else if (V == 3) pmin = chol16_blocked3(Dt, Ri, nlive, pmin);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int i = lane; i < 256; i += 64) { outD[blockIdx.x * 256 + i] = Dt[i]; outR[blockIdx.x * 256 + i] = Ri[i]; }
  if (lane == 0) outP[blockIdx.x] = pmin;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int G = 256;
  std::vector<double> A(G * 256);
  srand(7);
  for (int b = 0; b < G; ++b) {
    double Bm[16][24];
    for (auto& r : Bm) for (auto& v : r) v = rand() / (double)RAND_MAX - 0.5;
    if (b % 5 == 1) for (int k = 0; k < 24; ++k) Bm[7][k] = Bm[3][k] * (1 + 1e-9 * k);   // nearly dependent
    if (b % 7 == 2) for (int k = 0; k < 24; ++k) Bm[15][k] = 0.0;                          // zero column
    double S[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 24; ++k) s += Bm[i][k] * Bm[j][k]; S[i][j] = s; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      const double di = S[i][i] > 0 ? S[i][i] : 1.0, dj = S[j][j] > 0 ? S[j][j] : 1.0;
      A[b * 256 + i * 16 + j] = S[i][j] / std::sqrt(di * dj);
    }
  }
  constexpr int NV = 4; double *dA, *dD[NV], *dR[NV], *dP[NV];
  CK(hipMalloc(&dA, sizeof(double) * G * 256));
  CK(hipMemcpy(dA, A.data(), sizeof(double) * G * 256, hipMemcpyHostToDevice));
  for (int v = 0; v < NV; ++v) { CK(hipMalloc(&dD[v], sizeof(double) * G * 256)); CK(hipMalloc(&dR[v], sizeof(double) * G * 256)); CK(hipMalloc(&dP[v], sizeof(double) * G)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> D[NV], R[NV], P[NV];
  for (int v = 0; v < NV; ++v) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (v == 0) hipLaunchKernelGGL(chain_kernel<1>, dim3(G), dim3(64), 0, 0, rep ? iters : 10, 13, dA, dD[v], dR[v], dP[v]);
      else if (v == 1) hipLaunchKernelGGL(chain_kernel<2>, dim3(G), dim3(64), 0, 0, rep ? iters : 10, 13, dA, dD[v], dR[v], dP[v]);
      else if (v == 2) hipLaunchKernelGGL(chain_kernel<3>, dim3(G), dim3(64), 0, 0, rep ? iters : 10, 13, dA, dD[v], dR[v], dP[v]);
      else hipLaunchKernelGGL(chain_kernel<0>, dim3(G), dim3(64), 0, 0, rep ? iters : 10, 13, dA, dD[v], dR[v], dP[v]);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("variant %d: %.3f us per chain (one wave per CU)%s\n", v + 1, ms * 1e3 / iters, v == NV - 1 ? "  <- empty loop (tile load only)" : "");
    D[v].resize(G * 256); R[v].resize(G * 256); P[v].resize(G);
    CK(hipMemcpy(D[v].data(), dD[v], sizeof(double) * G * 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(R[v].data(), dR[v], sizeof(double) * G * 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(P[v].data(), dP[v], sizeof(double) * G, hipMemcpyDeviceToHost));
  }
  // checks: R^T R = A, Ri R = I (well-conditioned tiles), variants against each other, structure
  double worst_fact[3] = {0, 0, 0}, worst_inv[3] = {0, 0, 0}, worst_diff = 0, worst_pd = 0;
  int bad_struct = 0;
  for (int b = 0; b < G; ++b) {
    const bool regular = (b % 5 != 1) && (b % 7 != 2);
    for (int v = 0; v < 3; ++v) {
      const double* Rm = &D[v][b * 256]; const double* Im = &R[v][b * 256];
      for (int i = 0; i < 16; ++i) for (int j = 0; j < i; ++j) if (Rm[i * 16 + j] != 0.0 || Im[i * 16 + j] != 0.0) ++bad_struct;
      if (!regular) continue;
      for (int i = 0; i < 16; ++i) for (int j = i; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 16; ++k) s += Rm[k * 16 + i] * Rm[k * 16 + j];
        worst_fact[v] = std::fmax(worst_fact[v], std::fabs(s - A[b * 256 + i * 16 + j]));
        double t = 0; for (int k = 0; k < 16; ++k) t += Im[i * 16 + k] * Rm[k * 16 + j];
        worst_inv[v] = std::fmax(worst_inv[v], std::fabs(t - (i == j ? 1.0 : 0.0)));
      }
    }
    if (regular) for (int i = 0; i < 256; ++i) {
      worst_diff = std::fmax(worst_diff, std::fabs(D[0][b * 256 + i] - D[2][b * 256 + i]));
      worst_diff = std::fmax(worst_diff, std::fabs(R[0][b * 256 + i] - R[2][b * 256 + i]) / (1.0 + std::fabs(R[0][b * 256 + i])));
    }
    const double pa = P[0][b], pb = P[2][b];
    if (regular) worst_pd = std::fmax(worst_pd, std::fabs(pa - pb) / std::fabs(pa));
    if (!regular && b < 16) printf("  tile %d (irregular): pmin %.3e / %.3e\n", b, pa, pb);
  }
  printf("R^T R - A: %.2e / %.2e / %.2e   Ri R - I: %.2e / %.2e / %.2e   v1-v3: %.2e   pmin rel diff: %.2e   structure violations: %d\n",
         worst_fact[0], worst_fact[1], worst_fact[2], worst_inv[0], worst_inv[1], worst_inv[2], worst_diff, worst_pd, bad_struct);
  return 0;
}
