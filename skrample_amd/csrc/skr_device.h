// Device selection for the C-ABI entry points: kernels launch on the device that owns the caller's buffers, whatever the
// caller's current device is (a null stream handle names "the current device's default stream", so launching with the
// wrong device current would run on the wrong GPU).  Single-device processes -- the deployment model, one process per
// GPU -- skip the pointer query entirely.
#pragma once
#include <hip/hip_runtime.h>

namespace skr {

struct DeviceGuard {
  int prev = -1, dev = -1;
  bool switched = false;
  static int device_count() {
    static const int n = [] { int c = 0; if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 1; } return c; }();
    return n;
  }
  explicit DeviceGuard(const void* ptr) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    dev = prev;
    if (device_count() <= 1 || !ptr) return;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) == hipSuccess) dev = attr.device;
    else (void)hipGetLastError();
    if (dev != prev && dev >= 0) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace skr
