// Shared host helpers of the C-ABI library (error text, device guard).
#pragma once
namespace ydorb {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// YDORB_OK if `device` is a usable gfx950 HIP device and is now current; YDORB_ERR_NO_DEVICE otherwise.
// The product has no CPU path: callers propagate the error.
int require_device(int device);
}  // namespace ydorb
