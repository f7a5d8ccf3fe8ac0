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

// Kernels that need more than the default 64 KiB of dynamic LDS opt in once per kernel instance AND device (the
// attribute belongs to the current device's copy of the function).  `raised` is the instance's own flag word, one bit
// per device; two threads racing on a first launch at worst set the attribute twice, which is harmless.
static inline bool vc_raise_lds_limit(const void *kern, size_t lds_bytes, std::atomic<uint64_t> &raised)
{
    if (lds_bytes <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (raised.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return false;
    raised.fetch_or(bit, std::memory_order_release);
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
