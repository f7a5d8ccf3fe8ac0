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

// ---- split tensors ([n][c/8][h][w][3 pieces][8 channels] bf16; csrc/conv_split.h) ----
// bf16 pieces of an fp32 value by truncation: hi = top 16 bits, the remainder x - hi is exact, and so on; lo is exact (<= 8 bits left)
__device__ __forceinline__ void vc_split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    const unsigned ux = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, ux);
    const unsigned um = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, um);
    h = ux;
    m = um;
    l = __builtin_bit_cast(unsigned, r2);
}
// 4 consecutive channels (c % 4 == 0) of one pixel into its split record: 3 x 8 bytes.  `rec`: the 48-byte record of the pixel's
// group of 8 channels; `half`: 0 / 1 = channels 0-3 / 4-7 of the group
__device__ __forceinline__ void vc_store_split4(unsigned char *rec, int half, const f32x4 &v)
{
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vc_split3(v[e], h[e], m[e], l[e]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 ph = {(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
    const u32x2 pm = {(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
    const u32x2 pl = {(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
    unsigned char *d = rec + 8 * half;
    *reinterpret_cast<u32x2 *>(d) = ph;
    *reinterpret_cast<u32x2 *>(d + 16) = pm;
    *reinterpret_cast<u32x2 *>(d + 32) = pl;
}
// 8 consecutive channels of one pixel as the three 16-byte pieces of its split record
typedef unsigned vc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vc_split_record(const f32x4 &v0, const f32x4 &v1, vc_u32x4 &ph, vc_u32x4 &pm, vc_u32x4 &pl)
{
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vc_split3(v0[e], h[e], m[e], l[e]);
        vc_split3(v1[e], h[4 + e], m[4 + e], l[4 + e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ph[e] = (h[2 * e] >> 16) | h[2 * e + 1];
        pm[e] = (m[2 * e] >> 16) | m[2 * e + 1];
        pl[e] = (l[2 * e] >> 16) | (l[2 * e + 1] & 0xffff0000u);
    }
}
// Records of a 256-thread workgroup -- thread tid holds the record of plane tid % PB, pixel tid / PB of a run of 256 / PB consecutive
// pixels -- to memory as whole lines.  A lane storing its own record writes 16 bytes of every 48 per instruction (24+ lines touched per
// wave-instruction, three instructions to complete each); through LDS every wave-instruction writes 1 KiB contiguous.
// `sm`: VC_RECORDS_LDS(PB) bytes; `dst0`: the run's first record in plane 0, `plane_bytes` apart per plane; `npx`: valid pixels of the run
#define VC_RECORDS_LDS(PB) (256 * 48 + 16 * (PB))
template <int PB>
__device__ __forceinline__ void vc_store_records_256(unsigned char *sm, int tid, bool valid, const vc_u32x4 &ph, const vc_u32x4 &pm,
                                                     const vc_u32x4 &pl, unsigned char *dst0, long long plane_bytes, int npx)
{
    constexpr int PPS = 256 / PB, PLB = PPS * 48 + 16;     // + 16: the PB lanes of a pixel land on distinct banks
    if (valid) {
        unsigned char *d = sm + (tid % PB) * PLB + (tid / PB) * 48;
        *reinterpret_cast<vc_u32x4 *>(d) = ph;
        *reinterpret_cast<vc_u32x4 *>(d + 16) = pm;
        *reinterpret_cast<vc_u32x4 *>(d + 32) = pl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int slot = k * 256 + tid, p = slot / (3 * PPS), wi = slot - p * (3 * PPS);
        if (wi < 3 * npx) *reinterpret_cast<vc_u32x4 *>(dst0 + p * plane_bytes + wi * 16) = *reinterpret_cast<const vc_u32x4 *>(sm + p * PLB + wi * 16);
    }
}
// the exact fp32 values back out of a split record (hi + mid + lo with lo + mid first: both partial sums are representable)
__device__ __forceinline__ f32x4 vc_load_split4(const unsigned char *rec, int half)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned char *s = rec + 8 * half;
    const u32x2 ph = *reinterpret_cast<const u32x2 *>(s), pm = *reinterpret_cast<const u32x2 *>(s + 16), pl = *reinterpret_cast<const u32x2 *>(s + 32);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // (element 2 i sits in the low half of word i, element 2 i + 1 in the high half)
        const unsigned wh = ph[e >> 1], wm = pm[e >> 1], wl = pl[e >> 1];
        const float hi = __builtin_bit_cast(float, (e & 1) ? (wh & 0xffff0000u) : (wh << 16));
        const float mi = __builtin_bit_cast(float, (e & 1) ? (wm & 0xffff0000u) : (wm << 16));
        const float lo = __builtin_bit_cast(float, (e & 1) ? (wl & 0xffff0000u) : (wl << 16));
        r[e] = (lo + mi) + hi;
    }
    return r;
}

__device__ __forceinline__ long long view_off(const vc_view &v, int n, int y, int x)
{
    return (long long)n * v.sn + (long long)y * v.sh + (long long)x * v.sw;
}
