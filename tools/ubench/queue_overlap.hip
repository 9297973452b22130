// How many streams make progress at once?  N streams each run a chain of `len` dependent launches of a kernel that spins for `us`
// microseconds in `wgs` workgroups of 256 threads holding `lds` bytes of LDS.  If the chains overlap, wall time = len * us.
// hipcc -O2 --offload-arch=gfx950 tools/ubench/queue_overlap.hip -o /tmp/queue_overlap && /tmp/queue_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks, int* sink) {
  extern __shared__ int sm[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { if (threadIdx.x == 0) sm[0] += 1; }
  if (sm[0] == -1) *sink = 1;
}
int main() {
  int* sink; hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  const long long tick_per_us = 100;   // wall_clock64: 100 MHz
  for (int lds : {1024, 100 * 1024})
    for (int wgs : {1, 128, 256})
      for (int n : {1, 2, 3, 4, 6, 8}) {
        std::vector<hipStream_t> st(n);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const int len = 8, us = 250;
        for (int rep = 0; rep < 2; rep++) {
          hipDeviceSynchronize();
          auto t0 = std::chrono::steady_clock::now();
          for (int k = 0; k < len; k++)
            for (auto& s : st) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), lds, s, us * tick_per_us, sink);
          hipDeviceSynchronize();
          double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          if (rep) printf("lds %6d  wgs %3d  streams %d : %.2f ms (one chain alone = %.2f ms)\n", lds, wgs, n, ms, len * us * 1e-3);
        }
        for (auto& s : st) hipStreamDestroy(s);
      }
  return 0;
}
