// pcie_probe — what the host link of this box gives, so that the host-pointer path of the C ABI (run_pipelined in
// jubjub_amd/csrc/jj_engine.hip) can be priced against it (profiles/r4_pcie_probe.txt).
//   build: hipcc -O2 -o tools/pcie_probe tools/pcie_probe.cpp -lpthread
// Measures: page-locked H2D / D2H rate by transfer size, both directions at once, hipHostRegister / hipHostUnregister cost by
// size (touched pages), hipMemcpy from / to pageable memory, and the rate at which host threads copy into a page-locked bounce buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main() {
  CK(hipSetDevice(0));
  const size_t MAX = (size_t)1 << 30;
  void *dev, *dev2, *pin, *pin2;
  CK(hipMalloc(&dev, MAX)); CK(hipMalloc(&dev2, MAX));
  CK(hipHostMalloc(&pin, MAX, hipHostMallocDefault)); CK(hipHostMalloc(&pin2, MAX, hipHostMallocDefault));
  memset(pin, 1, MAX); memset(pin2, 2, MAX);
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  printf("# page-locked transfers (hipHostMalloc), one stream\n");
  for (size_t mb : {1, 8, 32, 128, 512, 1024}) {
    const size_t b = mb << 20; const int reps = mb >= 512 ? 3 : 10;
    CK(hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
    double t0 = now(); for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double h2d = (now() - t0) / reps;
    t0 = now(); for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(pin2, dev2, b, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); double d2h = (now() - t0) / reps;
    t0 = now();
    for (int r = 0; r < reps; r++) { CK(hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin2, dev2, b, hipMemcpyDeviceToHost, s2)); }
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double both = (now() - t0) / reps;
    printf("%5zu MB: H2D %6.1f GB/s  D2H %6.1f GB/s  both at once: %6.1f GB/s each way (%.2f ms)\n", mb, b / h2d / 1e9, b / d2h / 1e9, b / both / 1e9, both * 1e3);
  }
  printf("# hipHostRegister / hipHostUnregister of touched malloc memory; transfers from the registered range\n");
  for (size_t mb : {32, 64, 128, 512, 1024}) {
    const size_t b = mb << 20;
    void* p = nullptr; if (posix_memalign(&p, 4096, b)) return 1; memset(p, 3, b);
    double t0 = now(); CK(hipHostRegister(p, b, hipHostRegisterDefault)); double reg = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(dev, p, b, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double h2d = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(p, dev, b, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double d2h = now() - t0;
    t0 = now(); CK(hipHostUnregister(p)); double unreg = now() - t0;
    // pageable copies (the runtime stages them itself)
    t0 = now(); CK(hipMemcpy(dev, p, b, hipMemcpyHostToDevice)); double ph2d = now() - t0;
    t0 = now(); CK(hipMemcpy(p, dev, b, hipMemcpyDeviceToHost)); double pd2h = now() - t0;
    // untouched (fresh) output pages: what a caller's newly allocated result vector looks like
    void* q = nullptr; if (posix_memalign(&q, 4096, b)) return 1;
    t0 = now(); CK(hipHostRegister(q, b, hipHostRegisterDefault)); double regf = now() - t0; CK(hipHostUnregister(q));
    free(q); q = nullptr; if (posix_memalign(&q, 4096, b)) return 1;
    t0 = now(); CK(hipMemcpy(q, dev, b, hipMemcpyDeviceToHost)); double pd2hf = now() - t0;
    printf("%5zu MB: register %7.2f ms (%.2f ms/100MB)  unregister %6.2f ms | registered H2D %5.1f D2H %5.1f GB/s | pageable hipMemcpy H2D %5.1f D2H %5.1f GB/s | "
           "fresh pages: register %7.2f ms, pageable D2H %5.1f GB/s\n", mb, reg * 1e3, reg * 1e3 / mb * 100, unreg * 1e3, b / h2d / 1e9, b / d2h / 1e9, b / ph2d / 1e9, b / pd2h / 1e9,
           regf * 1e3, b / pd2hf / 1e9);
    free(p); free(q);
  }
  printf("# host threads copying pageable -> page-locked bounce buffer (512 MB)\n");
  {
    const size_t b = (size_t)512 << 20;
    void* p = nullptr; if (posix_memalign(&p, 4096, b)) return 1; memset(p, 5, b);
    for (int T : {1, 2, 4, 8, 16}) {
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back([=]() { const size_t lo = b / T * t; memcpy((char*)pin + lo, (char*)p + lo, b / T); });
      for (auto& x : th) x.join();
      double dt = now() - t0;
      printf("%2d thread(s): %6.1f GB/s\n", T, b / dt / 1e9);
    }
    free(p);
  }
  return 0;
}
