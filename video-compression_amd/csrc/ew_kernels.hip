// HBM-bound kernels of the hot path: layout conversion, pooling, bilinear resampling, warping,
// blending.  All tensors are fp32 channels-last views (vc_view); channel is the fastest thread index
// so that a wave's accesses are contiguous.  Arithmetic follows the PyTorch operators the reference
// calls (same formulas, same association where it is documented) -- see include/vc_hip.h for the
// reference call sites each entry point replaces.
#include "common.h"

#define EW_BLOCK 256

// ------------------------------------------------------------------------------------------------
// Row-mapped indexing (round 6).  The grid-stride form below every kernel of this file started with -- a 64-bit linear index
// taken apart by three 64-bit divisions per element -- costs ~300 vector instructions per element, more than the element's own
// work (k_axpby, one float per thread, was 4.8 % of the configs[4] frame).  ROWS: blockIdx.z = image, blockIdx.y (+ a grid stride) = row,
// the threads of blockIdx.x walk the w * per_px work items of the row; item -> (x, c) by ONE multiply-high with a host-made reciprocal
// (exact while item * per_px < 2^32; vc_rowmap_make checks it).  The linear form stays as the fallback for shapes outside that.
// The body of a kernel is the same lambda under both forms: same arithmetic per element, same bits.
// ------------------------------------------------------------------------------------------------
struct vc_rowmap {
    unsigned items, per_px, magic;
};

static inline bool vc_rowmap_make(vc_rowmap &m, int n, int h, int w, int per_px)
{
    if (n < 1 || h < 1 || w < 1 || per_px < 1 || n > 65535 || h > 65535) return false;
    const unsigned long long items = (unsigned long long)w * (unsigned long long)per_px;
    if (items >= (1ull << 24) || items * (unsigned long long)per_px >= (1ull << 32)) return false;
    m.items = (unsigned)items;
    m.per_px = (unsigned)per_px;
    m.magic = per_px == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned long long)per_px) + 1u;
    return true;
}

template <bool ROWS, class F>
__device__ __forceinline__ void ew_for_each(const vc_rowmap &m, int N, int H, int W, int PP, F f)
{
    if constexpr (ROWS) {
        const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= m.items) return;
        const unsigned x = m.per_px == 1 ? j : __umulhi(j, m.magic);
        const int c = (int)(j - __umul24(x, m.per_px));
        for (int y = blockIdx.y; y < H; y += gridDim.y) f((int)blockIdx.z, y, (int)x, c);      // (x, c) once per thread, rows in a grid stride
    } else {
        const long long total = (long long)N * H * W * PP;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % PP);
            long long t = i / PP;
            const int x = (int)(t % W); t /= W;
            const int y = (int)(t % H);
            const int n = (int)(t / H);
            f(n, y, x, c);
        }
    }
}

// grid of the row form: enough rows per grid step that ~8192 workgroups exist (a workgroup per row of a big tensor would be 65 K
// workgroups of one element per thread: measured slower than the grid-stride form on the pooling kernel), the rest in the row loop
static inline dim3 vc_rowmap_grid(const vc_rowmap &m, int n, int h)
{
    const long long bx = (m.items + EW_BLOCK - 1) / EW_BLOCK;
    long long gy = (8192 + bx * n - 1) / (bx * n);
    if (gy < 1) gy = 1;
    if (gy > h) gy = h;
    return dim3((unsigned)bx, (unsigned)gy, (unsigned)n);
}

// launch `rows` (the ROWS = true instance) on the row grid when the shape allows it, `lin` (ROWS = false) on the grid-stride grid otherwise
#define VC_EW_LAUNCH(st, KERN, N, H, W, PP, ...)                                                                                   \
    do {                                                                                                                           \
        vc_rowmap m_;                                                                                                              \
        if (vc_rowmap_make(m_, (N), (H), (W), (PP)))                                                                               \
            KERN<true><<<vc_rowmap_grid(m_, (N), (H)), dim3(EW_BLOCK), 0, (st)>>>(__VA_ARGS__, m_);                                \
        else                                                                                                                       \
            KERN<false><<<dim3(ew_grid((long long)(N) * (H) * (W) * (PP), EW_BLOCK)), dim3(EW_BLOCK), 0, (st)>>>(__VA_ARGS__, vc_rowmap{0, 0, 0}); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------------
__global__ void k_nchw_to_nhwc(const float *__restrict__ src, vc_view d)
{
    const long long total = (long long)d.n * d.h * d.w * d.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d.c);
        long long t = i / d.c;
        const int x = (int)(t % d.w); t /= d.w;
        const int y = (int)(t % d.h);
        const int n = (int)(t / d.h);
        d.p[view_off(d, n, y, x) + c] = src[(((long long)n * d.c + c) * d.h + y) * d.w + x];
    }
}

__global__ void k_nhwc_to_nchw(vc_view s, float *__restrict__ dst)
{
    const long long total = (long long)s.n * s.h * s.w * s.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % s.w);
        long long t = i / s.w;
        const int y = (int)(t % s.h); t /= s.h;
        const int c = (int)(t % s.c);
        const int n = (int)(t / s.c);
        dst[i] = s.p[view_off(s, n, y, x) + c];
    }
}

// ---- frame ingest / output (the data loader's and the CLI's pixel work) ----
// uint8 RGB [h][w][3] (what a PNG decoder delivers) -> fp32 NCHW [3][hp][wp], /255, reflection-padded on the bottom and
// right to (hp, wp): normalize + pad of LHBDC/encode_B.py:39-64 and test/utils.py:190-203 in one pass.
__global__ void k_u8hwc_to_f32nchw_pad(const uint8_t *__restrict__ src, int h, int w, float *__restrict__ dst, int hp, int wp)
{
    const long long total = 3ll * hp * wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % wp);
        const long long t = i / wp;
        const int y = (int)(t % hp), c = (int)(t / hp);
        const int sy = y < h ? y : 2 * (h - 1) - y, sx = x < w ? x : 2 * (w - 1) - x;      // nn.ReflectionPad2d
        dst[i] = (float)src[((long long)sy * w + sx) * 3 + c] / 255.0f;
    }
}

// fp32 NCHW [3][hp][wp] -> uint8 RGB [h][w][3] of the top-left h x w window: clip to [0,1], x255, round half to even
// (np.round), as float_to_uint8 of LHBDC/decode_B.py:35-38,122-123.
__global__ void k_f32nchw_to_u8hwc(const float *__restrict__ src, int hp, int wp, uint8_t *__restrict__ dst, int h, int w)
{
    const long long total = (long long)h * w * 3;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const long long t = i / 3;
        const int x = (int)(t % w), y = (int)(t / w);
        const float v = fminf(fmaxf(src[((long long)c * hp + y) * wp + x], 0.0f), 1.0f) * 255.0f;
        dst[i] = (uint8_t)rintf(v);
    }
}

extern "C" int vc_u8hwc_to_f32nchw_pad(vc_stream s, const uint8_t *src_hwc, int h, int w, float *dst_nchw, int hp, int wp)
{
    if (!src_hwc || !dst_nchw || h < 1 || w < 1 || hp < h || wp < w || hp - h >= h || wp - w >= w) return VC_EINVAL;
    hipLaunchKernelGGL(k_u8hwc_to_f32nchw_pad, dim3(ew_grid(3ll * hp * wp, 256)), dim3(256), 0, as_stream(s), src_hwc, h, w, dst_nchw, hp, wp);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_f32nchw_to_u8hwc(vc_stream s, const float *src_nchw, int hp, int wp, uint8_t *dst_hwc, int h, int w)
{
    if (!src_nchw || !dst_hwc || h < 1 || w < 1 || hp < h || wp < w) return VC_EINVAL;
    hipLaunchKernelGGL(k_f32nchw_to_u8hwc, dim3(ew_grid((long long)h * w * 3, 256)), dim3(256), 0, as_stream(s), src_nchw, hp, wp, dst_hwc, h, w);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_nchw_to_nhwc(vc_stream s, const float *src, vc_view dst)
{
    if (!src || !dst.p) return VC_EINVAL;
    const long long total = (long long)dst.n * dst.h * dst.w * dst.c;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), src, dst);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_nhwc_to_nchw(vc_stream s, vc_view src, float *dst)
{
    if (!src.p || !dst) return VC_EINVAL;
    const long long total = (long long)src.n * src.h * src.w * src.c;
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), src, dst);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// pooling
// ------------------------------------------------------------------------------------------------
template <bool ROWS> __global__ void k_avgpool_reflectpad(vc_view in, vc_view out, int k, float scale, vc_rowmap m)
{
    const int hp = in.h / k, wp = in.w / k;  // pooled size before padding
    const float inv = 1.0f / (float)(k * k);
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        const int py = y < hp ? y : 2 * (hp - 1) - y;  // ReflectionPad2d bottom/right
        const int px = x < wp ? x : 2 * (wp - 1) - x;
        float sum = 0.0f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) sum += in.p[view_off(in, n, py * k + dy, px * k + dx) + c];
        out.p[view_off(out, n, y, x) + c] = sum * inv * scale;
    });
}

extern "C" int vc_avgpool_reflectpad(vc_stream s, vc_view in, vc_view out, int k, float scale)
{
    if (!in.p || !out.p || k < 1 || in.c != out.c || in.n != out.n) return VC_EINVAL;
    const int hp = in.h / k, wp = in.w / k;
    if (out.h < hp || out.w < wp || out.h - hp >= hp || out.w - wp >= wp) return VC_EINVAL;  // reflect needs pad < size
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_avgpool_reflectpad, out.n, out.h, out.w, out.c, in, out, k, scale);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

template <bool ROWS> __global__ void k_maxpool2(vc_view in, vc_view out, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        const float *p = in.p + view_off(in, n, 2 * y, 2 * x) + c;
        const float a = fmaxf(p[0], p[in.sw]);
        const float b = fmaxf(p[in.sh], p[in.sh + in.sw]);
        out.p[view_off(out, n, y, x) + c] = fmaxf(a, b);
    });
}

// nn.MaxPool2d(2, 2) on a SPLIT tensor, split result (between the split-operand encoder layers of the mask U-Net, LHBDC/model/
// layers.py:200,224-230): the pieces of a record sum to the exact fp32 value, the maximum is split again -- bit for bit what the fp32
// kernel followed by vc_split3 gives.
template <bool ROWS> __global__ void k_maxpool2_sp3(const unsigned char *__restrict__ in, long long in_img_bytes, int n_img, int h, int w, int cg,
                                                    unsigned char *__restrict__ out, long long out_img_bytes, vc_rowmap m)
{
    const int oh = h >> 1, ow = w >> 1;
    // "image" of the row map = (image, plane): its split is one division of a wave-uniform value per thread
    ew_for_each<ROWS>(m, n_img * cg, oh, ow, 2, [&](int ng, int y, int x, int half) {
        const int n = ng / cg, g = ng - n * cg;
        const unsigned char *b = in + n * in_img_bytes + (((long long)g * h + 2 * y) * w + 2 * x) * 48;
        const f32x4 a0 = vc_load_split4(b, half), a1 = vc_load_split4(b + 48, half);
        const f32x4 a2 = vc_load_split4(b + (long long)w * 48, half), a3 = vc_load_split4(b + (long long)w * 48 + 48, half);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(a0[e], a1[e]), fmaxf(a2[e], a3[e]));
        vc_store_split4(out + n * out_img_bytes + (((long long)g * oh + y) * ow + x) * 48, half, v);
    });
}

// F.avg_pool2d(x, 2) on a SPLIT tensor (the skip tensor of a Flex-Rate U-Net level lives inside a split concat buffer and is pooled
// from there: Flex.../b_model/unet.py:62-68), result split or fp32.  Same sum order and scaling as k_avgpool_reflectpad (k = 2).
template <bool OSP, bool ROWS> __global__ void k_avgpool2_sp3(const unsigned char *__restrict__ in, long long in_img_bytes, int n_img, int h, int w, int cg,
                                                              float scale, unsigned char *__restrict__ out, long long out_img_bytes, vc_view of, vc_rowmap m)
{
    const int oh = h >> 1, ow = w >> 1;
    ew_for_each<ROWS>(m, n_img * cg, oh, ow, 2, [&](int ng, int y, int x, int half) {
        const int n = ng / cg, g = ng - n * cg;
        const unsigned char *b = in + n * in_img_bytes + (((long long)g * h + 2 * y) * w + 2 * x) * 48;
        const f32x4 a0 = vc_load_split4(b, half), a1 = vc_load_split4(b + 48, half);
        const f32x4 a2 = vc_load_split4(b + (long long)w * 48, half), a3 = vc_load_split4(b + (long long)w * 48 + 48, half);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = 0.0f;
            sum += a0[e]; sum += a1[e]; sum += a2[e]; sum += a3[e];
            v[e] = sum * 0.25f * scale;
        }
        if (OSP) vc_store_split4(out + n * out_img_bytes + (((long long)g * oh + y) * ow + x) * 48, half, v);
        else *reinterpret_cast<f32x4 *>(of.p + view_off(of, n, y, x) + 8 * g + 4 * half) = v;
    });
}

extern "C" int vc_avgpool2_sp3(vc_stream s, const void *in_split, long long in_image_bytes, int n, int h, int w, int c, float scale,
                               void *out_split, long long out_image_bytes, vc_view out_f32)
{
    if (!in_split || n < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || (c % 8) || ((uintptr_t)in_split % 8) || (in_image_bytes % 8)) return VC_EINVAL;
    if ((out_split != nullptr) == (out_f32.p != nullptr)) return VC_EINVAL;            // exactly one result
    const long long ii = in_image_bytes ? in_image_bytes : (long long)(c / 8) * h * w * 48;
    vc_rowmap m = {0, 0, 0};
    const bool rows = vc_rowmap_make(m, n * (c / 8), h / 2, w / 2, 2);
    const dim3 grid = rows ? vc_rowmap_grid(m, n * (c / 8), h / 2)
                           : dim3(ew_grid((long long)n * (c / 8) * (h / 2) * (w / 2) * 2, EW_BLOCK));
    const unsigned char *src = static_cast<const unsigned char *>(in_split);
    if (out_split) {
        if (((uintptr_t)out_split % 8) || (out_image_bytes % 8)) return VC_EINVAL;
        const long long oi = out_image_bytes ? out_image_bytes : (long long)(c / 8) * (h / 2) * (w / 2) * 48;
        unsigned char *dst = static_cast<unsigned char *>(out_split);
        if (rows) k_avgpool2_sp3<true, true><<<grid, dim3(EW_BLOCK), 0, as_stream(s)>>>(src, ii, n, h, w, c / 8, scale, dst, oi, out_f32, m);
        else k_avgpool2_sp3<true, false><<<grid, dim3(EW_BLOCK), 0, as_stream(s)>>>(src, ii, n, h, w, c / 8, scale, dst, oi, out_f32, m);
    } else {
        if (out_f32.n != n || out_f32.h != h / 2 || out_f32.w != w / 2 || out_f32.c != c || (out_f32.sw % 4) || (out_f32.sh % 4) || (out_f32.sn % 4) ||
            ((uintptr_t)out_f32.p % 16))
            return VC_EINVAL;
        if (rows) k_avgpool2_sp3<false, true><<<grid, dim3(EW_BLOCK), 0, as_stream(s)>>>(src, ii, n, h, w, c / 8, scale, nullptr, 0, out_f32, m);
        else k_avgpool2_sp3<false, false><<<grid, dim3(EW_BLOCK), 0, as_stream(s)>>>(src, ii, n, h, w, c / 8, scale, nullptr, 0, out_f32, m);
    }
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_maxpool2_sp3(vc_stream s, const void *in_split, long long in_image_bytes, int n, int h, int w, int c, void *out_split,
                               long long out_image_bytes)
{
    if (!in_split || !out_split || n < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || (c % 8) || ((uintptr_t)in_split % 8) || ((uintptr_t)out_split % 8) ||
        (in_image_bytes % 8) || (out_image_bytes % 8))
        return VC_EINVAL;
    const long long ii = in_image_bytes ? in_image_bytes : (long long)(c / 8) * h * w * 48;
    const long long oi = out_image_bytes ? out_image_bytes : (long long)(c / 8) * (h / 2) * (w / 2) * 48;
    VC_EW_LAUNCH(as_stream(s), k_maxpool2_sp3, n * (c / 8), h / 2, w / 2, 2, static_cast<const unsigned char *>(in_split), ii,
                 n, h, w, c / 8, static_cast<unsigned char *>(out_split), oi);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__global__ void k_maxpool2_v4(vc_view in, vc_view out)
{
    const int c4n = out.c >> 2;
    const long long total = (long long)out.n * out.h * out.w * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        long long t = i / c4n;
        const int x = (int)(t % out.w); t /= out.w;
        const int y = (int)(t % out.h);
        const int n = (int)(t / out.h);
        const float *p = in.p + view_off(in, n, 2 * y, 2 * x) + c;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + in.sw);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(p + in.sh), e = *reinterpret_cast<const f32x4 *>(p + in.sh + in.sw);
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(d[k], e[k]));
        *reinterpret_cast<f32x4 *>(out.p + view_off(out, n, y, x) + c) = r;
    }
}

extern "C" int vc_maxpool2(vc_stream s, vc_view in, vc_view out)
{
    if (!in.p || !out.p || in.c != out.c || in.n != out.n || out.h != in.h / 2 || out.w != in.w / 2) return VC_EINVAL;
    if (((in.c % 4) == 0 && (in.sw % 4) == 0 && (in.sh % 4) == 0 && (in.sn % 4) == 0 && ((uintptr_t)in.p % 16) == 0) &&
        ((out.sw % 4) == 0 && (out.sh % 4) == 0 && (out.sn % 4) == 0 && ((uintptr_t)out.p % 16) == 0)) {
        const long long total4 = (long long)out.n * out.h * out.w * (out.c / 4);
        hipLaunchKernelGGL(k_maxpool2_v4, dim3(ew_grid(total4, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), in, out);
        VC_LAUNCH_CHECK();
        return VC_OK;
    }
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_maxpool2, out.n, out.h, out.w, out.c, in, out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// bilinear up-sampling (ATen upsample_bilinear2d source-index rule)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_src(int dst, int in_size, int out_size, int factor, int align_corners,
                                             int &i0, int &i1, float &l0, float &l1)
{
    float src;
    if (align_corners) {
        const float sc = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
        src = sc * (float)dst;
    } else {
        const float sc = 1.0f / (float)factor;
        src = sc * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.0f) src = 0.0f;
    }
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

template <bool ROWS> __global__ void k_upsample_bilinear(vc_view in, vc_view out, int factor, int align_corners, float scale, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilinear_src(y, in.h, out.h, factor, align_corners, y0, y1, ly0, ly1);
        bilinear_src(x, in.w, out.w, factor, align_corners, x0, x1, lx0, lx1);
        const float *b = in.p + (long long)n * in.sn + c;
        const float v00 = b[(long long)y0 * in.sh + (long long)x0 * in.sw], v01 = b[(long long)y0 * in.sh + (long long)x1 * in.sw];
        const float v10 = b[(long long)y1 * in.sh + (long long)x0 * in.sw], v11 = b[(long long)y1 * in.sh + (long long)x1 * in.sw];
        const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        out.p[view_off(out, n, y, x) + c] = v * scale;
    });
}

// one bilinear sample of 4 channels: ONE definition (contraction pinned off) shared by the fp32 and the split-tensor kernel, so that
// the two give the same bits whatever the compiler makes of the code around it
__device__ __forceinline__ f32x4 bilinear4(const f32x4 &v00, const f32x4 &v01, const f32x4 &v10, const f32x4 &v11, float lx0, float lx1,
                                           float ly0, float ly1, float scale)
{
#pragma clang fp contract(off)
    const f32x4 top = lx0 * v00 + lx1 * v01, bot = lx0 * v10 + lx1 * v11;
    const f32x4 v = ly0 * top + ly1 * bot;
    return v * scale;
}

// 4 channels per lane (16-byte loads/stores) when the views allow it: the U-Net up-sampling layers move
// hundreds of MB per call and are purely HBM-bound
template <bool ROWS> __global__ void k_upsample_bilinear_v4(vc_view in, vc_view out, int factor, int align_corners, float scale, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c >> 2, [&](int n, int y, int x, int c4) {
        const int c = 4 * c4;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilinear_src(y, in.h, out.h, factor, align_corners, y0, y1, ly0, ly1);
        bilinear_src(x, in.w, out.w, factor, align_corners, x0, x1, lx0, lx1);
        const float *b = in.p + (long long)n * in.sn + c;
        const f32x4 v00 = *reinterpret_cast<const f32x4 *>(b + (long long)y0 * in.sh + (long long)x0 * in.sw);
        const f32x4 v01 = *reinterpret_cast<const f32x4 *>(b + (long long)y0 * in.sh + (long long)x1 * in.sw);
        const f32x4 v10 = *reinterpret_cast<const f32x4 *>(b + (long long)y1 * in.sh + (long long)x0 * in.sw);
        const f32x4 v11 = *reinterpret_cast<const f32x4 *>(b + (long long)y1 * in.sh + (long long)x1 * in.sw);
        *reinterpret_cast<f32x4 *>(out.p + view_off(out, n, y, x) + c) = bilinear4(v00, v01, v10, v11, lx0, lx1, ly0, ly1, scale);
    });
}

// The same into a SPLIT tensor (the up-sampled half of a concat buffer a split-operand convolution reads: LHBDC/model/layers.py:
// 232-246): one lane per 48-byte record (8 channels of a pixel), consecutive lanes consecutive pixels: three 16-byte stores per lane
// that together cover a contiguous run.
// (PB consecutive lanes take PB consecutive planes of ONE output pixel: their source reads are one contiguous 32 PB-byte run per corner;
//  lanes PB apart take consecutive pixels: each plane receives 64 / PB consecutive records per wave)
template <int PB>
__global__ void __launch_bounds__(EW_BLOCK) k_upsample_bilinear_sp3(vc_view in, unsigned char *__restrict__ out, long long out_img_bytes, int factor, int align_corners, float scale)
{
    // grid: x = runs of EW_BLOCK / PB pixels of a row, y = output row, z = (image, block of PB planes): no per-lane index division,
    // the row's source rows and weights are wave-uniform
    static_assert(EW_BLOCK == 256, "vc_store_records_256");
    __shared__ __attribute__((aligned(16))) unsigned char sm[VC_RECORDS_LDS(PB)];
    const int cg = in.c >> 3, nb = cg / PB, oh = in.h * factor, ow = in.w * factor;
    const int x_run = (int)blockIdx.x * (EW_BLOCK / PB);
    const int gi = threadIdx.x % PB, x = x_run + (int)threadIdx.x / PB, y = blockIdx.y;
    const int n = blockIdx.z / nb, g0 = (blockIdx.z - n * nb) * PB;
    vc_u32x4 ph = {0, 0, 0, 0}, pm = ph, pl = ph;
    if (x < ow) {
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilinear_src(y, in.h, oh, factor, align_corners, y0, y1, ly0, ly1);
        bilinear_src(x, in.w, ow, factor, align_corners, x0, x1, lx0, lx1);
        const float *b = in.p + (long long)n * in.sn + 8 * (g0 + gi);
        const float *r0 = b + (long long)y0 * in.sh, *r1 = b + (long long)y1 * in.sh;
        const int o0 = x0 * in.sw, o1 = x1 * in.sw;
        f32x4 v[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x4 v00 = *reinterpret_cast<const f32x4 *>(r0 + o0 + 4 * hf);
            const f32x4 v01 = *reinterpret_cast<const f32x4 *>(r0 + o1 + 4 * hf);
            const f32x4 v10 = *reinterpret_cast<const f32x4 *>(r1 + o0 + 4 * hf);
            const f32x4 v11 = *reinterpret_cast<const f32x4 *>(r1 + o1 + 4 * hf);
            v[hf] = bilinear4(v00, v01, v10, v11, lx0, lx1, ly0, ly1, scale);
        }
        vc_split_record(v[0], v[1], ph, pm, pl);
    }
    const long long plane_bytes = (long long)oh * ow * 48;
    vc_store_records_256<PB>(sm, threadIdx.x, x < ow, ph, pm, pl, out + n * out_img_bytes + g0 * plane_bytes + ((long long)y * ow + x_run) * 48,
                             plane_bytes, min(EW_BLOCK / PB, ow - x_run));
}

static inline bool view_vec4(const vc_view &v)
{
    return (v.c % 4) == 0 && (v.sw % 4) == 0 && (v.sh % 4) == 0 && (v.sn % 4) == 0 && ((uintptr_t)v.p % 16) == 0;
}

extern "C" int vc_upsample_bilinear_sp3(vc_stream s, vc_view in, void *out_split, long long out_image_bytes, int factor, int align_corners,
                                        float scale)
{
    if (!in.p || !out_split || factor < 1 || (in.c % 8) || !view_vec4(in) || ((uintptr_t)out_split % 16) || (out_image_bytes % 16)) return VC_EINVAL;
    const long long img = out_image_bytes ? out_image_bytes : (long long)(in.c / 8) * in.h * factor * in.w * factor * 48;
    const long long total = (long long)in.n * (in.c / 8) * in.h * factor * in.w * factor;
    if (total <= 0) return VC_OK;
    unsigned char *o = static_cast<unsigned char *>(out_split);
    const int cg = in.c / 8, oh = in.h * factor, ow = in.w * factor;
    const int pb = cg % 8 == 0 ? 8 : (cg % 4 == 0 ? 4 : 1);
    const long long gz = (long long)in.n * (cg / pb);
    if (oh > 65535 || gz > 65535 || (long long)in.w * in.sw > 0x7fffffffll) return VC_EINVAL;
    const dim3 grid((unsigned)((ow + EW_BLOCK / pb - 1) / (EW_BLOCK / pb)), (unsigned)oh, (unsigned)gz);
    if (pb == 8)
        hipLaunchKernelGGL(k_upsample_bilinear_sp3<8>, grid, dim3(EW_BLOCK), 0, as_stream(s), in, o, img, factor, align_corners, scale);
    else if (pb == 4)
        hipLaunchKernelGGL(k_upsample_bilinear_sp3<4>, grid, dim3(EW_BLOCK), 0, as_stream(s), in, o, img, factor, align_corners, scale);
    else
        hipLaunchKernelGGL(k_upsample_bilinear_sp3<1>, grid, dim3(EW_BLOCK), 0, as_stream(s), in, o, img, factor, align_corners, scale);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_upsample_bilinear(vc_stream s, vc_view in, vc_view out, int factor, int align_corners, float scale)
{
    if (!in.p || !out.p || factor < 1 || in.c != out.c || in.n != out.n) return VC_EINVAL;
    if (out.h != in.h * factor || out.w != in.w * factor) return VC_EINVAL;
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    if (out.c % 4 == 0 && view_vec4(in) && view_vec4(out))
        VC_EW_LAUNCH(as_stream(s), k_upsample_bilinear_v4, out.n, out.h, out.w, out.c / 4, in, out, factor, align_corners, scale);
    else
        VC_EW_LAUNCH(as_stream(s), k_upsample_bilinear, out.n, out.h, out.w, out.c, in, out, factor, align_corners, scale);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// out = alpha*a + beta*b
// ------------------------------------------------------------------------------------------------
template <bool ROWS> __global__ void k_axpby(vc_view a, vc_view b, vc_view out, float alpha, float beta, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        float v = alpha * a.p[view_off(a, n, y, x) + c];
        if (b.p) v += beta * b.p[view_off(b, n, y, x) + c];
        out.p[view_off(out, n, y, x) + c] = v;
    });
}

// 4 channels per thread (16-byte loads / stores) when the three views allow it; the same expression per value
template <bool ROWS> __global__ void k_axpby_v4(vc_view a, vc_view b, vc_view out, float alpha, float beta, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c >> 2, [&](int n, int y, int x, int c4) {
        const int c = 4 * c4;
        f32x4 v = alpha * *reinterpret_cast<const f32x4 *>(a.p + view_off(a, n, y, x) + c);
        if (b.p) v += beta * *reinterpret_cast<const f32x4 *>(b.p + view_off(b, n, y, x) + c);
        *reinterpret_cast<f32x4 *>(out.p + view_off(out, n, y, x) + c) = v;
    });
}

extern "C" int vc_axpby(vc_stream s, vc_view a, vc_view b, vc_view out, float alpha, float beta)
{
    if (!a.p || !out.p) return VC_EINVAL;
    if (a.h < out.h || a.w < out.w || a.c < out.c || a.n != out.n) return VC_EINVAL;
    if (b.p && (b.h < out.h || b.w < out.w || b.c < out.c || b.n != out.n)) return VC_EINVAL;
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    // rows that are dense in all three views (pixel stride == channel count: 3-channel frames, whole tensors) are runs of w * c floats:
    // the same values through the 16-byte form, whatever c is
    auto dense_row = [&](const vc_view &v) {
        return v.sw == out.c && (v.sh % 4) == 0 && (v.sn % 4) == 0 && ((uintptr_t)v.p % 16) == 0;
    };
    if (((long long)out.w * out.c) % 4 == 0 && dense_row(a) && dense_row(out) && (!b.p || dense_row(b))) {
        auto as_rows = [&](vc_view v) { v.w = out.w * out.c / 4, v.c = 4, v.sw = 4; return v; };
        const vc_view a4 = as_rows(a), o4 = as_rows(out);
        vc_view b4 = b;
        if (b.p) b4 = as_rows(b);
        VC_EW_LAUNCH(as_stream(s), k_axpby_v4, o4.n, o4.h, o4.w, 1, a4, b4, o4, alpha, beta);
    } else if (out.c % 4 == 0 && view_vec4(a) && view_vec4(out) && (!b.p || view_vec4(b)))
        VC_EW_LAUNCH(as_stream(s), k_axpby_v4, out.n, out.h, out.w, out.c / 4, a, b, out, alpha, beta);
    else
        VC_EW_LAUNCH(as_stream(s), k_axpby, out.n, out.h, out.w, out.c, a, b, out, alpha, beta);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

template <bool ROWS> __global__ void k_clamp01(vc_view a, vc_view out, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        out.p[view_off(out, n, y, x) + c] = fminf(fmaxf(a.p[view_off(a, n, y, x) + c], 0.0f), 1.0f);
    });
}

// out = clamp(a, 0, 1): decoded frames before they serve as references (ICIP2024/src/test.py:94 `torch.clamp(x_hat, 0, 1)`)
extern "C" int vc_clamp01(vc_stream s, vc_view a, vc_view out)
{
    if (!a.p || !out.p) return VC_EINVAL;
    if (a.h < out.h || a.w < out.w || a.c < out.c || a.n != out.n) return VC_EINVAL;
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_clamp01, out.n, out.h, out.w, out.c, a, out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// 4 consecutive channels per thread: 16 bytes in, 8 bytes out
template <bool ROWS> __global__ void k_to_half(vc_view a, _Float16 *__restrict__ out, vc_rowmap m)
{
    const int c4n = a.c >> 2;
    ew_for_each<ROWS>(m, a.n, a.h, a.w, c4n, [&](int n, int y, int x, int c4) {
        const float4 v = *reinterpret_cast<const float4 *>(a.p + view_off(a, n, y, x) + 4 * c4);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<h4 *>(out + 4 * ((((long long)n * a.h + y) * a.w + x) * c4n + c4)) = hv;
    });
}

// dense half-precision copy of a channels-last window (round to nearest even, what an fp16-path layer does to its input
// while staging): the features the fp16-path deformable fusion gathers from (vc_offset_diversity_hx)
extern "C" int vc_to_half(vc_stream s, vc_view a, void *out_half)
{
    if (!a.p || !out_half || (a.c % 4) || (a.sw % 4) || (a.sh % 4) || (a.sn % 4) || ((uintptr_t)a.p % 16) || ((uintptr_t)out_half % 8)) return VC_EINVAL;
    if ((long long)a.n * a.h * a.w * a.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_to_half, a.n, a.h, a.w, a.c / 4, a, static_cast<_Float16 *>(out_half));
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// group-planar half copy: one thread per (pixel, group) in that order (group fastest): a wave reads whole pixels and writes
// 64 / G pixels x cg halves contiguously into each of the G planes
template <int CG, bool ROWS> __global__ void k_to_half_planar(vc_view a, _Float16 *__restrict__ out, vc_rowmap m)
{
    const int G = a.c / CG;
    ew_for_each<ROWS>(m, a.n, a.h, a.w, G, [&](int n, int y, int x, int g) {
        const float *src = a.p + view_off(a, n, y, x) + g * CG;
        _Float16 *dst = out + ((((long long)n * G + g) * a.h + y) * a.w + x) * CG;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < CG; c += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + c);
            *reinterpret_cast<h4 *>(dst + c) = h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        }
    });
}

// group-planar half-precision copy [n][c / cg][h][w][cg] of a channels-last window: the layout vc_offset_diversity_hxp gathers from
extern "C" int vc_to_half_planar(vc_stream s, vc_view a, int cg, void *out_half)
{
    if (!a.p || !out_half || cg < 4 || (cg % 4) || (a.c % cg) || (a.sw % 4) || (a.sh % 4) || (a.sn % 4) || ((uintptr_t)a.p % 16) || ((uintptr_t)out_half % 8)) return VC_EINVAL;
    if (cg != 4 && cg != 8 && cg != 12 && cg != 16) return VC_EINVAL;
    if ((long long)a.n * a.h * a.w * a.c <= 0) return VC_OK;
    _Float16 *o = static_cast<_Float16 *>(out_half);
    vc_rowmap m = {0, 0, 0};
    const bool rows = vc_rowmap_make(m, a.n, a.h, a.w, a.c / cg);
    const dim3 grid = rows ? vc_rowmap_grid(m, a.n, a.h)
                           : dim3(ew_grid((long long)a.n * a.h * a.w * (a.c / cg), EW_BLOCK));
#define VC_PLANAR_CASE(CG)                                                                       \
    case CG:                                                                                     \
        if (rows) k_to_half_planar<CG, true><<<grid, dim3(EW_BLOCK), 0, as_stream(s)>>>(a, o, m); \
        else k_to_half_planar<CG, false><<<grid, dim3(EW_BLOCK), 0, as_stream(s)>>>(a, o, m);     \
        break;
    switch (cg) {
        VC_PLANAR_CASE(4)
        VC_PLANAR_CASE(8)
        VC_PLANAR_CASE(12)
        VC_PLANAR_CASE(16)
    }
#undef VC_PLANAR_CASE
    VC_LAUNCH_CHECK();
    return VC_OK;
}

template <bool ROWS> __global__ void k_channel_scale(vc_view a, const float *__restrict__ gain, vc_view out, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        out.p[view_off(out, n, y, x) + c] = gain[c] * a.p[view_off(a, n, y, x) + c];
    });
}

extern "C" int vc_channel_scale(vc_stream s, vc_view a, const float *gain, vc_view out)
{
    if (!a.p || !gain || !out.p || a.c < out.c || a.h < out.h || a.w < out.w || a.n != out.n) return VC_EINVAL;
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_channel_scale, out.n, out.h, out.w, out.c, a, gain, out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// compressai.layers.AttentionBlock gate: out = a * sigmoid(b) + identity (ICIP2024 ELIC intra codec, elic.py:97-121)
// ------------------------------------------------------------------------------------------------
template <bool ROWS> __global__ void k_attention_gate(vc_view a, vc_view b, vc_view identity, vc_view out, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        const float g = 1.0f / (1.0f + expf(-b.p[view_off(b, n, y, x) + c]));
        out.p[view_off(out, n, y, x) + c] = a.p[view_off(a, n, y, x) + c] * g + identity.p[view_off(identity, n, y, x) + c];
    });
}

extern "C" int vc_attention_gate(vc_stream s, vc_view a, vc_view b, vc_view identity, vc_view out)
{
    if (!a.p || !b.p || !identity.p || !out.p) return VC_EINVAL;
    const vc_view *vs[] = {&a, &b, &identity};
    for (const vc_view *v : vs)
        if (v->n != out.n || v->h != out.h || v->w != out.w || v->c < out.c) return VC_EINVAL;
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_attention_gate, out.n, out.h, out.w, out.c, a, b, identity, out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// quantise + checkerboard mask (ICIP2024 compression_bottlenecks.py:237-246,268-269)
// ------------------------------------------------------------------------------------------------
template <bool ROWS> __global__ void k_quantize_mask(vc_view in, vc_view out, const float *__restrict__ gain, int keep_parity, int do_round, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, out.c, [&](int n, int y, int x, int c) {
        float v = 0.0f;
        if (keep_parity < 0 || ((x + y) & 1) == keep_parity) {
            v = in.p[view_off(in, n, y, x) + c];
            if (do_round) v = rintf(v);
            if (gain) v *= gain[c];
        }
        out.p[view_off(out, n, y, x) + c] = v;
    });
}

extern "C" int vc_quantize_mask(vc_stream s, vc_view in, vc_view out, const float *gain, int keep_parity, int do_round)
{
    if (!in.p || !out.p || in.c < out.c || in.h != out.h || in.w != out.w || in.n != out.n) return VC_EINVAL;
    if (keep_parity < -1 || keep_parity > 1) return VC_EINVAL;
    if ((long long)out.n * out.h * out.w * out.c <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_quantize_mask, out.n, out.h, out.w, out.c, in, out, gain, keep_parity, do_round);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// warping: torch.grid_sample(bilinear, align_corners=False) behind the reference's two grid recipes
// ------------------------------------------------------------------------------------------------
// torch.linspace(start, end, steps)[i] in fp32 (symmetric evaluation from both ends)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i)
{
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// normalised grid coordinate of output pixel `i` displaced by `d` pixels
__device__ __forceinline__ float grid_coord(int convention, int i, float d, int size_flow, int size_img)
{
    if (convention == VC_WARP_W1) {
        const float inv = 1.0f / (float)size_flow;
        const float g = linspace_at(-1.0f + inv, 1.0f - inv, size_flow, i);
        return g + d / (((float)size_img - 1.0f) / 2.0f);
    }
    if (convention == VC_WARP_W3) {                    // ICIP2024 m.py:262-282: linspace(-1,1) grid
        const float g = linspace_at(-1.0f, 1.0f, size_flow, i);
        return g + d / (((float)size_flow - 1.0f) / 2.0f);
    }
    const float x = (float)i + d;                      // W2: b_model.py:104-109
    return 2.0f * (x / (float)size_img - 0.5f);
}

__device__ __forceinline__ float sample_bilinear(const float *img, long long sh, long long sw, int H, int W,
                                                 float gx, float gy, bool border, bool align_corners = false)
{
    // grid_sampler_unnormalize: align_corners=False -> ((g+1)*size-1)/2 ; True -> (g+1)/2*(size-1)
    float ix = align_corners ? ((gx + 1.0f) / 2.0f) * (float)(W - 1) : ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
    float iy = align_corners ? ((gy + 1.0f) / 2.0f) * (float)(H - 1) : ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    if (border) {
        ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
        iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
    } else {
        // zeros padding: anything beyond one pixel outside samples nothing; clamping there keeps the
        // float->int conversion defined for huge / infinite / NaN displacements (NaN -> -2 -> zeros)
        ix = fminf(fmaxf(ix, -2.0f), (float)W + 1.0f);
        iy = fminf(fmaxf(iy, -2.0f), (float)H + 1.0f);
    }
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.0f - w, n = iy - yn, s_ = 1.0f - n;
    const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
    const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
    const float nw = (x0ok && y0ok) ? img[(long long)y0 * sh + (long long)x0 * sw] : 0.0f;
    const float ne = (x1ok && y0ok) ? img[(long long)y0 * sh + (long long)x1 * sw] : 0.0f;
    const float sw_ = (x0ok && y1ok) ? img[(long long)y1 * sh + (long long)x0 * sw] : 0.0f;
    const float se = (x1ok && y1ok) ? img[(long long)y1 * sh + (long long)x1 * sw] : 0.0f;
    return nw * (s_ * e) + ne * (s_ * w) + sw_ * (n * e) + se * (n * w);
}

__global__ void k_warp(int convention, vc_view img, vc_view flow, vc_view out)
{
    const long long total = (long long)out.n * out.h * out.w * out.c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % out.c);
        long long t = i / out.c;
        const int x = (int)(t % out.w); t /= out.w;
        const int y = (int)(t % out.h);
        const int n = (int)(t / out.h);
        const float *f = flow.p + view_off(flow, n, y, x);
        const float gx = grid_coord(convention, x, f[0], flow.w, img.w);
        const float gy = grid_coord(convention, y, f[1], flow.h, img.h);
        out.p[view_off(out, n, y, x) + c] =
            sample_bilinear(img.p + (long long)n * img.sn + c, img.sh, img.sw, img.h, img.w, gx, gy,
                            convention != VC_WARP_W2, convention == VC_WARP_W3);
    }
}

// frames (3 channels, or any count not a multiple of 4 up to 4): one thread per PIXEL -- the flow vector is read and the grid
// coordinate computed once per pixel instead of once per channel, neighbouring lanes read neighbouring pixels.  Same
// arithmetic per value (sample_bilinear per channel).
template <bool ROWS> __global__ void k_warp_px(int convention, vc_view img, vc_view flow, vc_view out, vc_rowmap m)
{
    ew_for_each<ROWS>(m, out.n, out.h, out.w, 1, [&](int n, int y, int x, int) {
        const float *f = flow.p + view_off(flow, n, y, x);
        const float gx = grid_coord(convention, x, f[0], flow.w, img.w);
        const float gy = grid_coord(convention, y, f[1], flow.h, img.h);
        float *o = out.p + view_off(out, n, y, x);
        for (int c = 0; c < out.c; ++c)
            o[c] = sample_bilinear(img.p + (long long)n * img.sn + c, img.sh, img.sw, img.h, img.w, gx, gy,
                                   convention != VC_WARP_W2, convention == VC_WARP_W3);
    });
}

// 3-channel frames (the flow-resolution search of ICIP2024 warps 2176x3840 frames five times per B-frame; k_warp_px ran them at
// 0.3-0.5 TB/s: twelve 4-byte loads per pixel, each under its own bounds test, so the compiler waited for them one by one).  Here ONE
// 12-byte load per corner from an address clamped into the image, all four in flight together, and the value replaced by zero
// afterwards where sample_bilinear would not have loaded it; the weights and the sum are sample_bilinear's expressions, per channel.
struct vc_px3 {
    float v[3];
};
template <bool ROWS> __global__ void k_warp_px3(int convention, vc_view img, vc_view flow, vc_view out, vc_rowmap m)
{
    const bool border = convention != VC_WARP_W2, align_corners = convention == VC_WARP_W3;
    const int H = img.h, W = img.w;
    ew_for_each<ROWS>(m, out.n, out.h, out.w, 1, [&](int n_img, int y, int x, int) {
        const float *f = flow.p + view_off(flow, n_img, y, x);
        const float gx = grid_coord(convention, x, f[0], flow.w, img.w);
        const float gy = grid_coord(convention, y, f[1], flow.h, img.h);
        float ix = align_corners ? ((gx + 1.0f) / 2.0f) * (float)(W - 1) : ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        float iy = align_corners ? ((gy + 1.0f) / 2.0f) * (float)(H - 1) : ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
        if (border) {
            ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
            iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
        } else {
            ix = fminf(fmaxf(ix, -2.0f), (float)W + 1.0f);
            iy = fminf(fmaxf(iy, -2.0f), (float)H + 1.0f);
        }
        const float xw = floorf(ix), yn = floorf(iy);
        const float w = ix - xw, e = 1.0f - w, n = iy - yn, s_ = 1.0f - n;
        const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
        const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
        const float *base = img.p + (long long)n_img * img.sn;
        const long long r0 = (long long)min(max(y0, 0), H - 1) * img.sh, r1 = (long long)min(max(y1, 0), H - 1) * img.sh;
        const long long c0 = (long long)min(max(x0, 0), W - 1) * img.sw, c1 = (long long)min(max(x1, 0), W - 1) * img.sw;
        const vc_px3 a = *reinterpret_cast<const vc_px3 *>(base + r0 + c0), b = *reinterpret_cast<const vc_px3 *>(base + r0 + c1);
        const vc_px3 c = *reinterpret_cast<const vc_px3 *>(base + r1 + c0), d = *reinterpret_cast<const vc_px3 *>(base + r1 + c1);
        vc_px3 o;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float nw = (x0ok && y0ok) ? a.v[k] : 0.0f, ne = (x1ok && y0ok) ? b.v[k] : 0.0f;
            const float sw_ = (x0ok && y1ok) ? c.v[k] : 0.0f, se = (x1ok && y1ok) ? d.v[k] : 0.0f;
            o.v[k] = nw * (s_ * e) + ne * (s_ * w) + sw_ * (n * e) + se * (n * w);
        }
        *reinterpret_cast<vc_px3 *>(out.p + view_off(out, n_img, y, x)) = o;
    });
}

// 4 channels per thread: the sample position and the bilinear weights are computed once per 16 bytes (feature maps
// of 64-128 channels in ICIP2024 made the scalar version recompute them per element).  Same arithmetic per value.
template <bool ROWS> __global__ void k_warp_v4(int convention, vc_view img, vc_view flow, vc_view out, vc_rowmap m)
{
    const int c4n = out.c >> 2;
    const bool border = convention != VC_WARP_W2, ac = convention == VC_WARP_W3;
    const int H = img.h, W = img.w;
    ew_for_each<ROWS>(m, out.n, out.h, out.w, c4n, [&](int n, int y, int x, int c4) {
        const int c = 4 * c4;
        const float *f = flow.p + view_off(flow, n, y, x);
        const float gx = grid_coord(convention, x, f[0], flow.w, img.w);
        const float gy = grid_coord(convention, y, f[1], flow.h, img.h);
        float ix = ac ? ((gx + 1.0f) / 2.0f) * (float)(W - 1) : ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        float iy = ac ? ((gy + 1.0f) / 2.0f) * (float)(H - 1) : ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
        if (border) {
            ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
            iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
        } else {
            ix = fminf(fmaxf(ix, -2.0f), (float)W + 1.0f);
            iy = fminf(fmaxf(iy, -2.0f), (float)H + 1.0f);
        }
        const float xw = floorf(ix), yn = floorf(iy);
        const float w = ix - xw, e = 1.0f - w, nn = iy - yn, s_ = 1.0f - nn;
        const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
        const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
        const float *base = img.p + (long long)n * img.sn + c;
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 nw = (x0ok && y0ok) ? *reinterpret_cast<const f32x4 *>(base + (long long)y0 * img.sh + (long long)x0 * img.sw) : z;
        const f32x4 ne = (x1ok && y0ok) ? *reinterpret_cast<const f32x4 *>(base + (long long)y0 * img.sh + (long long)x1 * img.sw) : z;
        const f32x4 sw_ = (x0ok && y1ok) ? *reinterpret_cast<const f32x4 *>(base + (long long)y1 * img.sh + (long long)x0 * img.sw) : z;
        const f32x4 se = (x1ok && y1ok) ? *reinterpret_cast<const f32x4 *>(base + (long long)y1 * img.sh + (long long)x1 * img.sw) : z;
        const f32x4 r = nw * (s_ * e) + ne * (s_ * w) + sw_ * (nn * e) + se * (nn * w);
        *reinterpret_cast<f32x4 *>(out.p + view_off(out, n, y, x) + c) = r;
    });
}

extern "C" int vc_warp(vc_stream s, int convention, vc_view img, vc_view flow, vc_view out)
{
    if (!img.p || !flow.p || !out.p) return VC_EINVAL;
    if (convention != VC_WARP_W1 && convention != VC_WARP_W2 && convention != VC_WARP_W3) return VC_EINVAL;
    if (flow.c < 2 || out.c < 1 || out.c > img.c || out.h != flow.h || out.w != flow.w || img.n != out.n || flow.n != out.n) return VC_EINVAL;
    const long long total = (long long)out.n * out.h * out.w * out.c;
    if (total <= 0) return VC_OK;            // (an empty tensor: nothing to launch)
    if (out.c % 4 == 0 && view_vec4(img) && view_vec4(out))
        VC_EW_LAUNCH(as_stream(s), k_warp_v4, out.n, out.h, out.w, out.c / 4, convention, img, flow, out);
    else if (out.c == 3)
        VC_EW_LAUNCH(as_stream(s), k_warp_px3, out.n, out.h, out.w, 1, convention, img, flow, out);
    else if (out.c <= 4)
        VC_EW_LAUNCH(as_stream(s), k_warp_px, out.n, out.h, out.w, 1, convention, img, flow, out);
    else
        hipLaunchKernelGGL(k_warp, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), convention, img, flow, out);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// SPyNet glue
// ------------------------------------------------------------------------------------------------
__global__ void k_spynet_preprocess(const float *__restrict__ src, vc_view d)
{
    const long long total = (long long)d.n * d.h * d.w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % d.w);
        long long t = i / d.w;
        const int y = (int)(t % d.h);
        const int n = (int)(t / d.h);
        const long long plane = (long long)d.h * d.w;
        const float *s0 = src + (long long)n * 3 * plane + (long long)y * d.w + x;
        // flow.py:40-44: input channel 0 takes the "blue" statistics and is emitted LAST
        const float c0 = (s0[0] - 0.406f) / 0.225f;
        const float c1 = (s0[plane] - 0.456f) / 0.224f;
        const float c2 = (s0[2 * plane] - 0.485f) / 0.229f;
        float *o = d.p + view_off(d, n, y, x);
        o[0] = c2; o[1] = c1; o[2] = c0;
    }
}

extern "C" int vc_spynet_preprocess(vc_stream s, const float *src, vc_view dst)
{
    if (!src || !dst.p || dst.c != 3) return VC_EINVAL;
    const long long total = (long long)dst.n * dst.h * dst.w;
    hipLaunchKernelGGL(k_spynet_preprocess, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), src, dst);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// SP3: the 8-channel level input as ONE split record per pixel (`feat.p` = the dense split tensor [n][1][h][w][3][8] bf16): the
// first Basic-block layer runs on the split-operand pipeline and reads it as it lies (no fp32 copy, no conversion pass)
template <bool VEC, bool SP3 = false> __global__ void k_spynet_level_input(vc_view first, vc_view second, vc_view fc, vc_view feat, vc_view up)
{
    const long long total = (long long)feat.n * feat.h * feat.w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % feat.w);
        long long t = i / feat.w;
        const int y = (int)(t % feat.h);
        const int n = (int)(t / feat.h);
        float u = 0.0f, v = 0.0f;
        if (fc.p) {
            // F.interpolate(x2, bilinear, align_corners=True) * 2, then replicate-pad one row/col when
            // the level is odd-sized (flow.py:93-96): clamp the destination index into the x2 grid.
            const int uh = 2 * fc.h, uw = 2 * fc.w;
            const int yy = y < uh ? y : uh - 1, xx = x < uw ? x : uw - 1;
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            bilinear_src(yy, fc.h, uh, 2, 1, y0, y1, ly0, ly1);
            bilinear_src(xx, fc.w, uw, 2, 1, x0, x1, lx0, lx1);
            const float *b = fc.p + (long long)n * fc.sn;
            const float *p00 = b + (long long)y0 * fc.sh + (long long)x0 * fc.sw, *p01 = b + (long long)y0 * fc.sh + (long long)x1 * fc.sw;
            const float *p10 = b + (long long)y1 * fc.sh + (long long)x0 * fc.sw, *p11 = b + (long long)y1 * fc.sh + (long long)x1 * fc.sw;
            u = (ly0 * (lx0 * p00[0] + lx1 * p01[0]) + ly1 * (lx0 * p10[0] + lx1 * p11[0])) * 2.0f;
            v = (ly0 * (lx0 * p00[1] + lx1 * p01[1]) + ly1 * (lx0 * p10[1] + lx1 * p11[1])) * 2.0f;
        }
        const float gx = grid_coord(VC_WARP_W1, x, u, feat.w, second.w);
        const float gy = grid_coord(VC_WARP_W1, y, v, feat.h, second.h);
        const float *f1 = first.p + view_off(first, n, y, x);
        const float *s2 = second.p + (long long)n * second.sn;
        float *o = feat.p + view_off(feat, n, y, x);
        const float w0 = sample_bilinear(s2 + 0, second.sh, second.sw, second.h, second.w, gx, gy, true);
        const float w1 = sample_bilinear(s2 + 1, second.sh, second.sw, second.h, second.w, gx, gy, true);
        const float w2 = sample_bilinear(s2 + 2, second.sh, second.sw, second.h, second.w, gx, gy, true);
        float *q = up.p + view_off(up, n, y, x);
        if (SP3) {
            const f32x4 lo = {f1[0], f1[1], f1[2], w0}, hi = {w1, w2, u, v};
            unsigned char *rec = reinterpret_cast<unsigned char *>(feat.p) + (((long long)n * feat.h + y) * feat.w + x) * 48;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 ph, pm, pl;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 vv = hf ? hi : lo;
                unsigned h4[4], m4[4], l4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) vc_split3(vv[e], h4[e], m4[e], l4[e]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ph[2 * hf + e] = (h4[2 * e] >> 16) | h4[2 * e + 1];
                    pm[2 * hf + e] = (m4[2 * e] >> 16) | m4[2 * e + 1];
                    pl[2 * hf + e] = (l4[2 * e] >> 16) | (l4[2 * e + 1] & 0xffff0000u);
                }
            }
            *reinterpret_cast<u32x4 *>(rec) = ph;
            *reinterpret_cast<u32x4 *>(rec + 16) = pm;
            *reinterpret_cast<u32x4 *>(rec + 32) = pl;
            const f32x2 uv = {u, v};
            *reinterpret_cast<f32x2 *>(q) = uv;
        } else if (VEC) {      // the 8 channels of a pixel as two 16-byte stores (eight 4-byte stores touched every line eight times)
            const f32x4 lo = {f1[0], f1[1], f1[2], w0}, hi = {w1, w2, u, v};
            *reinterpret_cast<f32x4 *>(o) = lo;
            *reinterpret_cast<f32x4 *>(o + 4) = hi;
            const f32x2 uv = {u, v};
            *reinterpret_cast<f32x2 *>(q) = uv;
        } else {
            o[0] = f1[0]; o[1] = f1[1]; o[2] = f1[2];
            o[3] = w0; o[4] = w1; o[5] = w2;
            o[6] = u; o[7] = v;
            q[0] = u; q[1] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6 form of the level input (split record output): same values as k_spynet_level_input<true, true> up to fp32 contraction,
//   * work items are (image, row, strip of EW_BLOCK pixels), walked grid-stride by whole workgroups: the row / image of an item come
//     out of two SCALAR divisions per workgroup-iteration (no per-lane 64-bit index division), lanes are consecutive pixels of a row;
//   * each bilinear corner of the warped frame is ONE 12-byte load (three channels) instead of three 4-byte loads, the first frame's
//     pixel likewise: 9 gathers per pixel instead of 19;
//   * the row's records leave through LDS as whole lines (vc_store_records_256);
//   * the two flow components are interpolated in separate dependency chains behind opaque barriers (the un-paired forms of the
//     3-D-grid diagnostic kernels never failed: DESIGN section 5f).
// Opt-in only (VC_LI_FORM=rows): two processes on one device still bring out wrong lanes 48-63 in it -- see the launch site.
// ------------------------------------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) vc_f3 { float a, b, c; };
__global__ void __launch_bounds__(EW_BLOCK) k_spynet_level_input_rows(vc_view first, vc_view second, vc_view fc, vc_view feat, vc_view up, int strips)
{
    __shared__ __attribute__((aligned(16))) unsigned char sm[VC_RECORDS_LDS(1)];
    const int items = feat.n * feat.h * strips;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int strip = item % strips, row = item / strips;
        int y = row % feat.h;
        const int n = row / feat.h;
        const int x_run = strip * EW_BLOCK, x = x_run + (int)threadIdx.x;
        vc_u32x4 ph = {0, 0, 0, 0}, pm = ph, pl = ph;
        if (x < feat.w) {
            float u = 0.0f, v = 0.0f;
            if (fc.p) {
                const int uh = 2 * fc.h, uw = 2 * fc.w;
                const int yy = y < uh ? y : uh - 1, xx = x < uw ? x : uw - 1;
                int y0, y1, x0, x1;
                float ly0, ly1, lx0, lx1;
                bilinear_src(yy, fc.h, uh, 2, 1, y0, y1, ly0, ly1);
                bilinear_src(xx, fc.w, uw, 2, 1, x0, x1, lx0, lx1);
                const float *b = fc.p + (long long)n * fc.sn;
                const float *p00 = b + (long long)y0 * fc.sh + (long long)x0 * fc.sw, *p01 = b + (long long)y0 * fc.sh + (long long)x1 * fc.sw;
                const float *p10 = b + (long long)y1 * fc.sh + (long long)x0 * fc.sw, *p11 = b + (long long)y1 * fc.sh + (long long)x1 * fc.sw;
                float a0 = p00[0], a1 = p01[0], a2 = p10[0], a3 = p11[0], b0 = p00[1], b1 = p01[1], b2 = p10[1], b3 = p11[1];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                u = (ly0 * (lx0 * a0 + lx1 * a1) + ly1 * (lx0 * a2 + lx1 * a3)) * 2.0f;
                asm volatile("" : "+v"(u), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                v = (ly0 * (lx0 * b0 + lx1 * b1) + ly1 * (lx0 * b2 + lx1 * b3)) * 2.0f;
                asm volatile("" : "+v"(v));
            }
            const float gx = grid_coord(VC_WARP_W1, x, u, feat.w, second.w);
            const float gy = grid_coord(VC_WARP_W1, y, v, feat.h, second.h);
            // sample_bilinear(border, align_corners = false) of the three channels, the corner pixels as whole 12-byte loads
            const int W = second.w, H = second.h;
            float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
            ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
            iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
            const float xw = floorf(ix), yn = floorf(iy);
            const float w_ = ix - xw, e = 1.0f - w_, nn = iy - yn, s_ = 1.0f - nn;
            const int sx0 = (int)xw, sy0 = (int)yn, sx1 = sx0 + 1, sy1 = sy0 + 1;
            const bool x1ok = sx1 < W, y1ok = sy1 < H;            // (after the border clamp x0 / y0 are always inside)
            const float *s2 = second.p + (long long)n * second.sn;
            const int cx1 = x1ok ? sx1 : sx0, cy1 = y1ok ? sy1 : sy0;
            const vc_f3 nw = *reinterpret_cast<const vc_f3 *>(s2 + (long long)sy0 * second.sh + (long long)sx0 * second.sw);
            vc_f3 ne = *reinterpret_cast<const vc_f3 *>(s2 + (long long)sy0 * second.sh + (long long)cx1 * second.sw);
            vc_f3 sw = *reinterpret_cast<const vc_f3 *>(s2 + (long long)cy1 * second.sh + (long long)sx0 * second.sw);
            vc_f3 se = *reinterpret_cast<const vc_f3 *>(s2 + (long long)cy1 * second.sh + (long long)cx1 * second.sw);
            if (!x1ok) ne = vc_f3{0.0f, 0.0f, 0.0f};
            if (!y1ok) sw = vc_f3{0.0f, 0.0f, 0.0f};
            if (!(x1ok && y1ok)) se = vc_f3{0.0f, 0.0f, 0.0f};
            const float w0 = nw.a * (s_ * e) + ne.a * (s_ * w_) + sw.a * (nn * e) + se.a * (nn * w_);
            const float w1 = nw.b * (s_ * e) + ne.b * (s_ * w_) + sw.b * (nn * e) + se.b * (nn * w_);
            const float w2 = nw.c * (s_ * e) + ne.c * (s_ * w_) + sw.c * (nn * e) + se.c * (nn * w_);
            const vc_f3 f1 = *reinterpret_cast<const vc_f3 *>(first.p + view_off(first, n, y, x));
            const f32x4 lo = {f1.a, f1.b, f1.c, w0}, hi = {w1, w2, u, v};
            vc_split_record(lo, hi, ph, pm, pl);
            const f32x2 uv = {u, v};
            *reinterpret_cast<f32x2 *>(up.p + view_off(up, n, y, x)) = uv;
        }
        vc_store_records_256<1>(sm, threadIdx.x, x < feat.w, ph, pm, pl,
                                reinterpret_cast<unsigned char *>(feat.p) + (((long long)n * feat.h + y) * feat.w + x_run) * 48, 0, min(EW_BLOCK, feat.w - x_run));
        __syncthreads();               // (the LDS image is re-used by the next item)
    }
}

#ifdef VC_LI_DIAG
// ------------------------------------------------------------------------------------------------------------------------------
// Diagnostic build only (make li_diag -> libvc_hip_lidiag.so; tools/r06.sh li-diag): the 3-D-grid form of the level-input kernel that gave
// intermittent wrong pixels when two processes shared the GPU, with one suspect removed per variant
// (VC_LI_VARIANT): 1 = as it was; 2 = coarse flow through non-temporal loads; 3 = coarse flow through system-scope (sc0 sc1) loads;
// 4 = ~2 us of s_sleep before the first load; 5 = row index kept in a VGPR (no wave-uniform hoisting of the row weights through
// v_readfirstlane); 6 = variant 1 behind a hipStreamSynchronize on the host; 7 = variant 1 with every store system-scope; 8 = no packed
// fp32 instructions pairing u with v (opaque asm barriers: no v_pk_mov_b32 swizzles); 9 = the interpolation in scalar inline asm.
// ------------------------------------------------------------------------------------------------------------------------------
template <int VAR> __device__ __forceinline__ float li_load(const float *p)
{
    if (VAR == 2) return __builtin_nontemporal_load(p);
    if (VAR == 3) {
        float r;
        asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
        return r;
    }
    return *p;
}
// where each wave ran: HW_ID (wave / SIMD / CU / SE ...) and XCC_ID per (image, row, strip, wave) of the launch, one slot per pyramid level
#define LI_DBG_SLOT (1 << 17)
__device__ unsigned li_dbg[6 * LI_DBG_SLOT * 4];
extern "C" int vc_li_diag_read(unsigned *host_dst)      // 6 slots x LI_DBG_SLOT x {HW_ID, XCC_ID, wave lifetime in 100 MHz ticks, HW_ID at the end}
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(li_dbg), sizeof(unsigned) * 6 * LI_DBG_SLOT * 4) == hipSuccess ? VC_OK : VC_ELAUNCH;
}
template <bool SP3, int VAR> __global__ void k_spynet_level_input_3d(vc_view first, vc_view second, vc_view fc, vc_view feat, vc_view up, int slot)
{
    // (s_memrealtime: the constant 100 MHz counter -- it keeps running while a wave is saved and swapped out, so a wave that was
    //  preempted shows a lifetime of milliseconds instead of microseconds)
    const unsigned li_idx = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
    unsigned long long li_t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(li_t0)::"memory");
    if ((threadIdx.x & 63) == 0) {
        if (li_idx < LI_DBG_SLOT) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
            li_dbg[(slot * LI_DBG_SLOT + li_idx) * 4] = hw;
            li_dbg[(slot * LI_DBG_SLOT + li_idx) * 4 + 1] = xcc;
        }
    }
    if (VAR == 4) {
#pragma unroll 1
        for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(100);
    }
    int x = (int)(blockIdx.x * EW_BLOCK + threadIdx.x), y = blockIdx.y, n = blockIdx.z;
    if (VAR == 5) asm volatile("" : "+v"(y));
    vc_u32x4 ph = {0, 0, 0, 0}, pm = ph, pl = ph;
    if (x < feat.w) {
        float u = 0.0f, v = 0.0f;
        if (fc.p) {
            const int uh = 2 * fc.h, uw = 2 * fc.w;
            const int yy = y < uh ? y : uh - 1, xx = x < uw ? x : uw - 1;
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            bilinear_src(yy, fc.h, uh, 2, 1, y0, y1, ly0, ly1);
            bilinear_src(xx, fc.w, uw, 2, 1, x0, x1, lx0, lx1);
            const float *b = fc.p + (long long)n * fc.sn;
            const float *p00 = b + (long long)y0 * fc.sh + (long long)x0 * fc.sw, *p01 = b + (long long)y0 * fc.sh + (long long)x1 * fc.sw;
            const float *p10 = b + (long long)y1 * fc.sh + (long long)x0 * fc.sw, *p11 = b + (long long)y1 * fc.sh + (long long)x1 * fc.sw;
            if (VAR == 9) {      // scalar VALU instructions only, written out
                auto lerp4 = [&](float a0, float a1, float a2, float a3) {
                    float t0, t1, r;
                    asm volatile("v_mul_f32 %0, %3, %5\n\tv_fma_f32 %0, %2, %4, %0\n\tv_mul_f32 %1, %3, %7\n\tv_fma_f32 %1, %2, %6, %1"
                                 : "=&v"(t0), "=&v"(t1) : "v"(lx0), "v"(lx1), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
                    asm volatile("v_mul_f32 %0, %2, %4\n\tv_fma_f32 %0, %1, %3, %0\n\tv_add_f32 %0, %0, %0" : "=&v"(r) : "v"(ly0), "v"(ly1), "v"(t0), "v"(t1));
                    return r;
                };
                u = lerp4(p00[0], p01[0], p10[0], p11[0]);
                v = lerp4(p00[1], p01[1], p10[1], p11[1]);
            } else if (VAR == 8) {
                float a0 = p00[0], a1 = p01[0], a2 = p10[0], a3 = p11[0], b0 = p00[1], b1 = p01[1], b2 = p10[1], b3 = p11[1];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                u = (ly0 * (lx0 * a0 + lx1 * a1) + ly1 * (lx0 * a2 + lx1 * a3)) * 2.0f;
                asm volatile("" : "+v"(u), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                v = (ly0 * (lx0 * b0 + lx1 * b1) + ly1 * (lx0 * b2 + lx1 * b3)) * 2.0f;
                asm volatile("" : "+v"(v));
            } else {
            u = (ly0 * (lx0 * li_load<VAR>(p00) + lx1 * li_load<VAR>(p01)) + ly1 * (lx0 * li_load<VAR>(p10) + lx1 * li_load<VAR>(p11))) * 2.0f;
            v = (ly0 * (lx0 * li_load<VAR>(p00 + 1) + lx1 * li_load<VAR>(p01 + 1)) + ly1 * (lx0 * li_load<VAR>(p10 + 1) + lx1 * li_load<VAR>(p11 + 1))) * 2.0f;
            }
        }
        const float gx = grid_coord(VC_WARP_W1, x, u, feat.w, second.w);
        const float gy = grid_coord(VC_WARP_W1, y, v, feat.h, second.h);
        const float *f1 = first.p + view_off(first, n, y, x);
        const float *s2 = second.p + (long long)n * second.sn;
        float *o = feat.p + view_off(feat, n, y, x);
        const float w0 = sample_bilinear(s2 + 0, second.sh, second.sw, second.h, second.w, gx, gy, true);
        const float w1 = sample_bilinear(s2 + 1, second.sh, second.sw, second.h, second.w, gx, gy, true);
        const float w2 = sample_bilinear(s2 + 2, second.sh, second.sw, second.h, second.w, gx, gy, true);
        float *q = up.p + view_off(up, n, y, x);
        const f32x4 lo = {f1[0], f1[1], f1[2], w0}, hi = {w1, w2, u, v};
        const f32x2 uv = {u, v};
        if (SP3) {
            vc_split_record(lo, hi, ph, pm, pl);
            if (VAR == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(q), "v"(uv) : "memory");
            else *reinterpret_cast<f32x2 *>(q) = uv;
        } else if (VAR == 7) {
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1\n\t"
                         "global_store_dwordx2 %3, %4, off sc0 sc1" :: "v"(o), "v"(lo), "v"(hi), "v"(q), "v"(uv) : "memory");
        } else {
            *reinterpret_cast<f32x4 *>(o) = lo;
            *reinterpret_cast<f32x4 *>(o + 4) = hi;
            *reinterpret_cast<f32x2 *>(q) = uv;
        }
    }
    if constexpr (SP3) {
        __shared__ __attribute__((aligned(16))) unsigned char sm[VC_RECORDS_LDS(1)];
        const int x_run = (int)blockIdx.x * EW_BLOCK;
        vc_store_records_256<1>(sm, threadIdx.x, x < feat.w, ph, pm, pl,
                                reinterpret_cast<unsigned char *>(feat.p) + (((long long)n * feat.h + y) * feat.w + x_run) * 48, 0, min(EW_BLOCK, feat.w - x_run));
    }
    unsigned long long li_t1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(li_t1)::"memory");
    if ((threadIdx.x & 63) == 0 && li_idx < LI_DBG_SLOT) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        li_dbg[(slot * LI_DBG_SLOT + li_idx) * 4 + 2] = (unsigned)(li_t1 - li_t0);
        li_dbg[(slot * LI_DBG_SLOT + li_idx) * 4 + 3] = hw;
    }
}
static int li_variant()
{
    const char *e = getenv("VC_LI_VARIANT");
    return e ? atoi(e) : 0;
}
template <bool SP3> static bool li_launch_3d(hipStream_t st, vc_view first, vc_view second, vc_view fc, vc_view feat, vc_view up)
{
    const int var = li_variant();
    if (var < 1 || var > 9) return false;
    const dim3 grid((unsigned)((feat.w + EW_BLOCK - 1) / EW_BLOCK), (unsigned)feat.h, (unsigned)feat.n);
    int slot = 0;
    for (int hh = feat.h; hh < 1000 && slot < 5; hh *= 2) ++slot;       // 1088 -> 0, 544 -> 1, ... 34 -> 5
    if (var == 6) (void)hipStreamSynchronize(st);
    switch (var) {
    case 2: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 2>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    case 3: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 3>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    case 4: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 4>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    case 5: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 5>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    case 7: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 7>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    case 8: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 8>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    case 9: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 9>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    default: hipLaunchKernelGGL((k_spynet_level_input_3d<SP3, 1>), grid, dim3(EW_BLOCK), 0, st, first, second, fc, feat, up, slot); break;
    }
    return true;
}
#endif

extern "C" int vc_spynet_level_input(vc_stream s, vc_view first, vc_view second, vc_view fc, vc_view feat, vc_view up)
{
    if (!first.p || !second.p || !feat.p || !up.p) return VC_EINVAL;
    if (first.c != 3 || second.c != 3 || feat.c != 8 || up.c != 2) return VC_EINVAL;
    if (first.h != feat.h || first.w != feat.w || second.h != feat.h || second.w != feat.w) return VC_EINVAL;
    if (up.h != feat.h || up.w != feat.w) return VC_EINVAL;
    if (fc.p && (fc.c != 2 || (2 * fc.h != feat.h && 2 * fc.h + 1 != feat.h) || (2 * fc.w != feat.w && 2 * fc.w + 1 != feat.w)))
        return VC_EINVAL;
    const long long total = (long long)feat.n * feat.h * feat.w;
    const bool vec = view_vec4(feat) && reinterpret_cast<uintptr_t>(up.p) % 8 == 0 && up.sn % 2 == 0 && up.sh % 2 == 0 && up.sw % 2 == 0;
#ifdef VC_LI_DIAG
    if (vec && li_launch_3d<false>(as_stream(s), first, second, fc, feat, up)) return VC_OK;
#endif
    if (vec)
        hipLaunchKernelGGL(k_spynet_level_input<true>, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), first, second,
                           fc, feat, up);
    else
        hipLaunchKernelGGL(k_spynet_level_input<false>, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), first, second,
                           fc, feat, up);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

extern "C" int vc_spynet_level_input_sp3(vc_stream s, vc_view first, vc_view second, vc_view fc, void *feat_split, vc_view up)
{
    if (!first.p || !second.p || !feat_split || !up.p || ((uintptr_t)feat_split % 16)) return VC_EINVAL;
    if (first.c != 3 || second.c != 3 || up.c != 2) return VC_EINVAL;
    if (second.h != first.h || second.w != first.w || up.h != first.h || up.w != first.w || up.n != first.n) return VC_EINVAL;
    if (fc.p && (fc.c != 2 || (2 * fc.h != first.h && 2 * fc.h + 1 != first.h) || (2 * fc.w != first.w && 2 * fc.w + 1 != first.w)))
        return VC_EINVAL;
    if (reinterpret_cast<uintptr_t>(up.p) % 8 || up.sn % 2 || up.sh % 2 || up.sw % 2) return VC_EINVAL;
    vc_view feat = first;                     // (shape only; p = the split tensor)
    feat.p = static_cast<float *>(feat_split);
    feat.c = 8;
    const long long total = (long long)feat.n * feat.h * feat.w;
#ifdef VC_LI_DIAG
    if (li_launch_3d<true>(as_stream(s), first, second, fc, feat, up)) return VC_OK;
#endif
    {   // VC_LI_FORM=rows (read once) selects the round-6 row form: 1.7x faster (160 against 278 us at 4 x 1088 x 1920) but NOT the default --
        // with a second process on the device it shows the lane-48..63 fault of DESIGN section 5f (17 of 400 runs, this time in the
        // warped channels), the grid-stride form over pixels never has (0 of 480)
        static const bool rows_form = [] { const char *e = getenv("VC_LI_FORM"); return e && e[0] == 'r'; }();
        const long long items = (long long)feat.n * feat.h * ((feat.w + EW_BLOCK - 1) / EW_BLOCK);
        if (rows_form && first.sw == 3 && second.sw == 3 && items < (1ll << 30)) {
            const int strips = (feat.w + EW_BLOCK - 1) / EW_BLOCK;
            hipLaunchKernelGGL(k_spynet_level_input_rows, dim3((unsigned)(items < 256 * 16 ? items : 256 * 16)), dim3(EW_BLOCK), 0, as_stream(s), first,
                               second, fc, feat, up, strips);
            VC_LAUNCH_CHECK();
            return VC_OK;
        }
    }
    hipLaunchKernelGGL((k_spynet_level_input<true, true>), dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), first, second, fc, feat, up);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// fp32 window with ANY channel count -> split tensor of c_out channels (a multiple of 8, >= c), channels past c zero: the input of a
// split-operand layer whose channel count is no multiple of the chunk (first layers: 6 = two warped frames, LHBDC/model/layers.py:202,
// Flex.../b_model/unet.py:43) -- the layer's weights are packed with zero rows for the padding channels (vcamd/hip.py: PackedConv).
// One lane per (pixel, group of 8 channels), scalar loads (no alignment asked of the window), three 16-byte stores.
// ------------------------------------------------------------------------------------------------
__global__ void k_split3_pad(vc_view a, unsigned char *__restrict__ out, long long out_img_bytes, int planes)
{
    const long long per_plane = (long long)a.h * a.w;
    const long long total = (long long)a.n * planes * per_plane;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pos = i % per_plane;
        long long t = i / per_plane;
        const int g = (int)(t % planes), n = (int)(t / planes);
        const int y = (int)(pos / a.w), x = (int)(pos - (long long)y * a.w);
        const float *src = a.p + view_off(a, n, y, x) + 8 * g;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (8 * g + e < a.c) ? src[e] : 0.0f;
        const f32x4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
        vc_u32x4 ph, pm, pl;
        vc_split_record(v0, v1, ph, pm, pl);
        unsigned char *dst = out + n * out_img_bytes + ((long long)g * per_plane + pos) * 48;
        *reinterpret_cast<vc_u32x4 *>(dst) = ph;
        *reinterpret_cast<vc_u32x4 *>(dst + 16) = pm;
        *reinterpret_cast<vc_u32x4 *>(dst + 32) = pl;
    }
}

extern "C" int vc_split3_pad(vc_stream s, vc_view a, void *out_split, long long out_image_bytes, int c_out)
{
    if (!a.p || !out_split || a.c < 1 || c_out < a.c || (c_out % 8) || ((uintptr_t)out_split % 16) || (out_image_bytes % 16)) return VC_EINVAL;
    const int planes = c_out / 8;
    const long long total = (long long)a.n * planes * a.h * a.w;
    if (total <= 0) return VC_OK;
    const long long img = out_image_bytes ? out_image_bytes : (long long)planes * a.h * a.w * 48;
    hipLaunchKernelGGL(k_split3_pad, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), a, static_cast<unsigned char *>(out_split), img, planes);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------
// blending
// ------------------------------------------------------------------------------------------------
template <bool ROWS> __global__ void k_lhbdc_blend(vc_view fwbw, vc_view mask, vc_view cur, vc_view pred, vc_view resid, vc_rowmap m)
{
    ew_for_each<ROWS>(m, pred.n, pred.h, pred.w, 3, [&](int n, int y, int x, int c) {
        const float m = mask.p[view_off(mask, n, y, x)];
        const float *f = fwbw.p + view_off(fwbw, n, y, x);
        const float pv = m * f[c] + (1.0f - m) * f[3 + c];   // m.py:65
        pred.p[view_off(pred, n, y, x) + c] = pv;
        if (resid.p) resid.p[view_off(resid, n, y, x) + c] = cur.p[view_off(cur, n, y, x) + c] - pv;
    });
}

extern "C" int vc_lhbdc_blend(vc_stream s, vc_view fwbw, vc_view mask, vc_view cur, vc_view pred, vc_view resid)
{
    if (!fwbw.p || !mask.p || !pred.p || fwbw.c < 6 || pred.c != 3) return VC_EINVAL;
    if (resid.p && !cur.p) return VC_EINVAL;
    if ((long long)pred.n * pred.h * pred.w <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_lhbdc_blend, pred.n, pred.h, pred.w, 3, fwbw, mask, cur, pred, resid);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

template <bool ROWS> __global__ void k_flex_blend(vc_view xb, vc_view xa, vc_view mask, vc_view cur, vc_view pred, vc_view resid, vc_rowmap m)
{
    ew_for_each<ROWS>(m, pred.n, pred.h, pred.w, 3, [&](int n, int y, int x, int c) {
        const float *m = mask.p + view_off(mask, n, y, x);
        const float w1 = 0.5f * m[0], w2 = 0.5f * m[1];    // b_model.py:70-71
        const float pv = (w1 * xb.p[view_off(xb, n, y, x) + c] + w2 * xa.p[view_off(xa, n, y, x) + c]) / (w1 + w2 + 1e-8f);
        pred.p[view_off(pred, n, y, x) + c] = pv;
        if (resid.p) resid.p[view_off(resid, n, y, x) + c] = cur.p[view_off(cur, n, y, x) + c] - pv;
    });
}

extern "C" int vc_flex_blend(vc_stream s, vc_view xb, vc_view xa, vc_view mask, vc_view cur, vc_view pred, vc_view resid)
{
    if (!xb.p || !xa.p || !mask.p || !pred.p || mask.c < 2 || pred.c != 3) return VC_EINVAL;
    if (resid.p && !cur.p) return VC_EINVAL;
    if ((long long)pred.n * pred.h * pred.w <= 0) return VC_OK;
    VC_EW_LAUNCH(as_stream(s), k_flex_blend, pred.n, pred.h, pred.w, 3, xb, xa, mask, cur, pred, resid);
    VC_LAUNCH_CHECK();
    return VC_OK;
}

__global__ void k_flex_motion_split(vc_view f4, vc_view ft0, vc_view ft1, float t)
{
    const long long total = (long long)f4.n * f4.h * f4.w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % f4.w);
        long long q = i / f4.w;
        const int y = (int)(q % f4.h);
        const int n = (int)(q / f4.h);
        const float *f = f4.p + view_off(f4, n, y, x);
        float *a = ft0.p + view_off(ft0, n, y, x), *b = ft1.p + view_off(ft1, n, y, x);
        // b_model.py:39-40 with the same association:  -(1-t)*t*F01 + t*t*F10 ;  (1-t)*(1-t)*F01 - t*(1-t)*F10
        const float k0 = -(1.0f - t) * t, k1 = t * t, k2 = (1.0f - t) * (1.0f - t), k3 = t * (1.0f - t);
        a[0] = k0 * f[0] + k1 * f[2]; a[1] = k0 * f[1] + k1 * f[3];
        b[0] = k2 * f[0] - k3 * f[2]; b[1] = k2 * f[1] - k3 * f[3];
    }
}

extern "C" int vc_flex_motion_split(vc_stream s, vc_view f4, vc_view ft0, vc_view ft1, float t)
{
    if (!f4.p || !ft0.p || !ft1.p || f4.c < 4) return VC_EINVAL;
    const long long total = (long long)f4.n * f4.h * f4.w;
    hipLaunchKernelGGL(k_flex_motion_split, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0, as_stream(s), f4, ft0, ft1, t);
    VC_LAUNCH_CHECK();
    return VC_OK;
}
