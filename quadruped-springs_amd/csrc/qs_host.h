// Host-side helpers shared by the C-ABI translation units (qs_hip.hip, qs_norm.hip).
#pragma once
#include <hip/hip_runtime.h>

// Entry points run on the handle's device whatever the calling thread's current device is, and leave that as they found it.
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define QS_ON_DEVICE(h) DeviceGuard qs_guard_((h)->device)
