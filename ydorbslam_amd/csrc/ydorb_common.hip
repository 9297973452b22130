// Error reporting + device guard shared by every entry point of the C ABI.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/ydorb/c_api.h"
#include "ydorb_host.h"

namespace {
thread_local char g_err[512] = "";
}

namespace ydorb {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int require_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible: the ydorb hot path runs on gfx950 only (no CPU fallback)");
    return YDORB_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (0..%d)", device, n - 1);
    return YDORB_ERR_INVALID_ARG;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) {
    set_error("hipGetDeviceProperties(%d) failed", device);
    return YDORB_ERR_NO_DEVICE;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library carries gfx950 code objects only", device, p.gcnArchName);
    return YDORB_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed", device);
    return YDORB_ERR_HIP;
  }
  return YDORB_OK;
}
}  // namespace ydorb

extern "C" {
const char* ydorb_last_error(void) { return g_err; }
const char* ydorb_version(void) { return "ydorb-mi355x 0.1 (gfx950)"; }
int ydorb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}
}
