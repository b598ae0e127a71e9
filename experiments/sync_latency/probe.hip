// What does waiting for a kernel cost on the host?  (round 6: the synchronous MSM call spends ~15 us between the last kernel's end and the host tail.)
//   A  hipEventRecord + hipEventSynchronize      (what jj_msm_finish does)
//   B  the kernel's last act is a system-scope store of a flag into page-locked host memory; the host spins on the flag
//   C  hipStreamSynchronize
// for a kernel that spins ~5 us and ~300 us.  Median wall time of launch -> host knows, minus the kernel's own spin.
//   hipcc --offload-arch=gfx950 -O2 -o probe probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(unsigned* flag, unsigned val, long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned* flag; hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocPortable);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (double us : {5.0, 300.0}) {
    const long long cyc = (long long)(us * 100.0);            // wall_clock64 ticks at 100 MHz
    for (int mode = 0; mode < 3; mode++) {
      std::vector<double> ts;
      for (int it = 0; it < 300; it++) {
        *flag = 0;
        const unsigned val = (unsigned)it + 1;
        const double t0 = now_us();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, flag, val, cyc);
        if (mode == 0) { hipEventRecord(ev, st); hipEventSynchronize(ev); }
        else if (mode == 1) { hipEventRecord(ev, st); while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != val) {} }
        else hipStreamSynchronize(st);
        ts.push_back(now_us() - t0);
        hipStreamSynchronize(st);
      }
      std::sort(ts.begin(), ts.end());
      printf("kernel spins %5.0f us  %-42s median %7.1f us  (- spin = %5.1f us)  p10 %7.1f  p90 %7.1f\n", us,
             mode == 0 ? "A hipEventRecord + hipEventSynchronize" : mode == 1 ? "B flag in host memory, host spins" : "C hipStreamSynchronize", ts[ts.size() / 2], ts[ts.size() / 2] - us, ts[ts.size() / 10], ts[ts.size() * 9 / 10]);
    }
  }
  return 0;
}
