// Probe (GPU box): host -> device copy of a large PAGEABLE buffer — the runtime's own path against a ring of page-locked
// staging slots filled by T host threads while the previous slot is DMA'd.
//   hipcc -O2 tools/h2d_stage_probe.hip -o tools/_build/h2d_stage_probe -lpthread && tools/_build/h2d_stage_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char* dst, const char* src, size_t n, int T) {
  if (T <= 1) { memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = ((n + T - 1) / T + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= n) break;
    th.emplace_back([=] { memcpy(dst + o, src + o, (o + per <= n) ? per : n - o); });
  }
  for (auto& t : th) t.join();
}
int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atof(argv[1]) : 2048) << 20;
  char* src = (char*)malloc(total);
  for (size_t i = 0; i < total; i += 4096) src[i] = (char)i;
  char* dev = nullptr; CK(hipMalloc((void**)&dev, total));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); CK(hipMemcpyAsync(dev, src, total, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    printf("runtime pageable copy: %.1f GB/s\n", total / (now() - t0) / 1e9);
  }
  for (size_t slot_mb : {16, 32, 64}) {
    const size_t slot = slot_mb << 20;
    const int NS = 3;
    char* pin[NS]; hipEvent_t ev[NS];
    for (int k = 0; k < NS; ++k) { CK(hipHostMalloc((void**)&pin[k], slot, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
    for (int T : {1, 2, 4, 8, 12, 16}) {
      double best = 0;
      for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        int k = 0;
        for (size_t o = 0; o < total; o += slot, k = (k + 1) % NS) {
          const size_t nb = (o + slot <= total) ? slot : total - o;
          CK(hipEventSynchronize(ev[k]));
          par_copy(pin[k], src + o, nb, T);
          CK(hipMemcpyAsync(dev + o, pin[k], nb, hipMemcpyHostToDevice, s));
          CK(hipEventRecord(ev[k], s));
        }
        CK(hipStreamSynchronize(s));
        const double r = total / (now() - t0) / 1e9;
        if (r > best) best = r;
      }
      printf("slots of %3zu MB, %2d threads: %.1f GB/s\n", slot_mb, T, best);
    }
    for (int k = 0; k < NS; ++k) { hipHostFree(pin[k]); hipEventDestroy(ev[k]); }
  }
  // host memcpy alone
  char* dst = (char*)malloc(total);
  for (int T : {1, 4, 8, 16}) { par_copy(dst, src, total, T); double t0 = now(); par_copy(dst, src, total, T); printf("host memcpy %2d threads: %.1f GB/s\n", T, total / (now() - t0) / 1e9); }
  return 0;
}
