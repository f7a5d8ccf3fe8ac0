// Shared helpers for the libvc_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "vc_hip.h"

#define VC_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return VC_ELAUNCH;           \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Kernels that need more than the default 64 KiB of dynamic LDS opt in per kernel instance AND device (the attribute
// belongs to the current device's copy of the function).  `raised` is the instance's own table of the largest size
// requested so far on each device: an instance whose LDS size depends on the layer (the 4-channel configuration keeps
// the layer's weights behind its tile image) raises the limit again when a later layer needs more.  Two threads racing
// on a first launch at worst set the attribute twice, which is harmless.
constexpr int VC_MAX_DEVICES = 64;
struct vc_lds_raised {
    std::atomic<uint32_t> bytes[VC_MAX_DEVICES];
};
static inline bool vc_raise_lds_limit(const void *kern, size_t lds_bytes, vc_lds_raised &raised)
{
    if (lds_bytes <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VC_MAX_DEVICES) return false;
    if (raised.bytes[dev].load(std::memory_order_acquire) >= lds_bytes) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return false;
    uint32_t cur = raised.bytes[dev].load(std::memory_order_relaxed);
    while (cur < lds_bytes && !raised.bytes[dev].compare_exchange_weak(cur, (uint32_t)lds_bytes, std::memory_order_release)) {}
    return true;
}

static inline hipStream_t as_stream(vc_stream s) { return reinterpret_cast<hipStream_t>(s); }

// Grid size for memory-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8, capped.
static inline int ew_grid(long long work_items, int block)
{
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

__device__ __forceinline__ long long view_off(const vc_view &v, int n, int y, int x)
{
    return (long long)n * v.sn + (long long)y * v.sh + (long long)x * v.sw;
}
