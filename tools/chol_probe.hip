// Stand-alone timing / check of the blocked Cholesky kernels of csrc/chol_kernels.hip (development aid)
//   hipcc --offload-arch=gfx950 -O3 -I bounded-lsq_amd/csrc tools/chol_probe.hip -o tools/_build/chol_probe
//   ./chol_probe n B [reps]
#include "chol_kernels.hip"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace blsq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, B = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 10;
  const int N = n + 1, NPAD = (N + 15) / 16 * 16;
  const int ND = 4;                                     // distinct problems
  const size_t sz = (size_t)NPAD * NPAD;
  std::vector<double> G(ND * sz, 0.0);
  srand(3);
  for (int d = 0; d < ND; ++d) {
    const int m = 2 * N + 7;
    std::vector<double> M((size_t)m * N);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < N; ++i)
      for (int j = i; j < N; ++j) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += M[(size_t)k * N + i] * M[(size_t)k * N + j];
        G[d * sz + (size_t)i * NPAD + j] = s;
        if ((i >> 4) == (j >> 4)) G[d * sz + (size_t)j * NPAD + i] = s;   // diagonal tiles are full
      }
  }
  double *dG, *dR, *dinv;
  int *dfb, *dcnt;
  CK(hipMalloc(&dG, sizeof(double) * B * sz));
  CK(hipMalloc(&dR, sizeof(double) * B * sz));
  CK(hipMalloc(&dinv, sizeof(double) * B * (NPAD / 16) * 256));
  CK(hipMalloc(&dfb, sizeof(int) * (B + 4)));
  CK(hipMalloc(&dcnt, sizeof(int) * 4));
  for (int b = 0; b < B; ++b) CK(hipMemcpy(dG + b * sz, G.data() + (b % ND) * sz, sizeof(double) * sz, hipMemcpyHostToDevice));
  CK(hipMemset(dcnt, 0, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  GramCholArgs a{};
  a.Gsrc = dG; a.G = dR; a.NPAD = NPAD; a.n = n; a.stride_vec = NPAD;
  a.fb_mask = dfb; a.fail_count = dcnt; a.rinv = dinv;
  for (int variant = 0; variant < 2; ++variant) {
    if (NPAD > 80) setenv("BLSQ_CHOL_RL", variant ? "1" : "0", 1);
    else if (variant) break;
    CK(launch_gram_chol(a, B, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) CK(launch_gram_chol(a, B, 0));
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> R(sz);
    double worst = 0;
    for (int b : {0, B - 1}) {
      CK(hipMemcpy(R.data(), dR + (size_t)b * sz, sizeof(double) * sz, hipMemcpyDeviceToHost));
      const double* Gd = G.data() + (b % ND) * sz;
      for (int i = 0; i < N; ++i)
        for (int j = i; j < N; ++j) {
          if (i == n && j == n) continue;               // (rho is not stored)
          double s = 0;
          for (int k = 0; k <= i && k < n; ++k) s += R[(size_t)k * NPAD + i] * R[(size_t)k * NPAD + j];
          const double ref = Gd[(size_t)i * NPAD + j];
          worst = std::fmax(worst, std::fabs(s - ref) / std::sqrt(Gd[(size_t)i * NPAD + i] * Gd[(size_t)j * NPAD + j]));
        }
    }
    int cnt = 0; CK(hipMemcpy(&cnt, dcnt, 4, hipMemcpyDeviceToHost));
    printf("n %d B %d %s: %.1f us per launch   max |R^T R - G| (scaled) %.2e   gate failures %d\n", n, B,
           NPAD <= 80 ? "one wave per problem" : (variant ? "right-looking" : "left-looking"), ms * 1e3 / reps, worst, cnt);
  }
  return 0;
}
