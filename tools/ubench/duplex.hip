// Host-link duplex probe: pinned hipMemcpyAsync against copy KERNELS that read / write host-mapped pinned memory, alone and in pairs.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/duplex tools/ubench/duplex.hip ; run on the GPU box: tools/ubench/duplex [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
  for (int i = 0; i < iters; i++) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
  if (a + b == 12345.f) out[0] = a;
}
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 256, bytes = mb << 20, n16 = bytes / 16;
  const int wgs = argc > 2 ? atoi(argv[2]) : 512;
  void *hIn, *hOut, *dIn, *dOut, *hInDev, *hOutDev;
  CK(hipHostMalloc(&hIn, bytes, hipHostMallocMapped)); CK(hipHostMalloc(&hOut, bytes, hipHostMallocMapped));
  CK(hipMalloc(&dIn, bytes)); CK(hipMalloc(&dOut, bytes));
  CK(hipHostGetDevicePointer(&hInDev, hIn, 0)); CK(hipHostGetDevicePointer(&hOutDev, hOut, 0));
  memset(hIn, 1, bytes); memset(hOut, 0, bytes); CK(hipMemset(dOut, 2, bytes));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  auto up_dma = [&]() { return hipMemcpyAsync(dIn, hIn, bytes, hipMemcpyHostToDevice, sa); };
  auto down_dma = [&]() { return hipMemcpyAsync(hOut, dOut, bytes, hipMemcpyDeviceToHost, sb); };
  auto up_krn = [&]() { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, sa, (const uint4*)hInDev, (uint4*)dIn, n16); return hipGetLastError(); };
  auto down_krn = [&]() { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, sb, (const uint4*)dOut, (uint4*)hOutDev, n16); return hipGetLastError(); };
  struct Case { const char* name; int up, down; };   // 0 none, 1 dma, 2 kernel
  const Case cases[] = {{"h2d dma", 1, 0}, {"d2h dma", 0, 1}, {"h2d kernel", 2, 0}, {"d2h kernel", 0, 2}, {"h2d dma + d2h dma", 1, 1},
                        {"h2d dma + d2h kernel", 1, 2}, {"h2d kernel + d2h dma", 2, 1}, {"h2d kernel + d2h kernel", 2, 2}};
  for (const Case& c : cases) {
    auto issue = [&]() -> hipError_t {
      hipError_t e = hipSuccess;
      if (c.up == 1) e = up_dma(); else if (c.up == 2) e = up_krn();
      if (e != hipSuccess) return e;
      if (c.down == 1) e = down_dma(); else if (c.down == 2) e = down_krn();
      return e;
    };
    CK(issue()); CK(hipDeviceSynchronize());
    const int reps = 8;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) CK(issue());
    CK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%-26s %6.1f ms  up %5.1f GB/s  down %5.1f GB/s  (both streams finish together: sum %5.1f)\n", c.name, dt * 1e3, c.up ? bytes / dt / 1e9 : 0.0,
           c.down ? bytes / dt / 1e9 : 0.0, ((c.up ? 1 : 0) + (c.down ? 1 : 0)) * bytes / dt / 1e9);
  }
  // unequal sizes: a full upload with a quarter-size read-back beside it (the pipeline's ratio), kernel read-back against dma read-back
  for (int kd = 1; kd <= 2; kd++) {
    auto issue = [&]() -> hipError_t {
      hipError_t e = up_dma();
      if (e != hipSuccess) return e;
      if (kd == 1) return hipMemcpyAsync(hOut, dOut, bytes / 4, hipMemcpyDeviceToHost, sb);
      hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, sb, (const uint4*)dOut, (uint4*)hOutDev, n16 / 4);
      return hipGetLastError();
    };
    CK(issue()); CK(hipDeviceSynchronize());
    const int reps = 8;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) CK(issue());
    CK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("h2d dma + quarter d2h %-6s %6.1f ms  up %5.1f GB/s  down %5.1f GB/s\n", kd == 1 ? "dma" : "kernel", dt * 1e3, bytes / dt / 1e9, bytes / 4 / dt / 1e9);
  }
  // does work on the GPU slow an upload?  a VALU-bound and an HBM-bound kernel on a third stream, re-launched while 8 uploads (+ quarter read-backs) run
  {
    hipStream_t sc; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    void *dA, *dB; CK(hipMalloc(&dA, 1ull << 30)); CK(hipMalloc(&dB, 1ull << 30));
    for (int kind = 0; kind < 2; kind++) {
      const int reps = 8;
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; r++) {
        CK(up_dma());
        CK(hipMemcpyAsync(hOut, dOut, bytes / 4, hipMemcpyDeviceToHost, sb));
        for (int q = 0; q < 4; q++) {
          if (kind == 0) hipLaunchKernelGGL(k_valu, dim3(256 * 32), dim3(256), 0, sc, (float*)dA, 20000);
          else hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, sc, (const uint4*)dA, (uint4*)dB, (size_t)(1ull << 30) / 16);
        }
      }
      CK(hipStreamSynchronize(sa));
      const double dtUp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
      CK(hipDeviceSynchronize());
      const double dtAll = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
      printf("h2d dma + quarter d2h dma + %s kernels: uploads %5.1f GB/s (kernels done after %.1f ms per round)\n", kind == 0 ? "VALU-bound" : "HBM-bound ", bytes / dtUp / 1e9, dtAll * 1e3);
    }
  }
  return 0;
}
