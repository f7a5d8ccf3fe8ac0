// Modulated deformable 3x3 convolution (torchvision.ops.deform_conv2d semantics) for the ICIP2024 OffsetDiversity
// fusion (ICIP2024/src/model/helpers.py:35-58).  Gather-bound, not matrix-bound: per output pixel and group the
// kernel samples 9 taps x 4 corners x cg channels (cg = 8..16 contiguous floats in NHWC) and contracts them with
// 9*cg*og weights held in LDS; one lane owns one (pixel, group) pair and the group index is uniform per wave
// (broadcast LDS reads of the weights).  The offset/mask preparation of OffsetDiversity.prep (tanh *
// magnitude + flipped flow, sigmoid) is folded into the offset fetch in the fused entry point.
#include "common.h"

namespace {

struct DeformArgs {
    vc_view x1, x2;        // inputs of the first / second half of the groups (generic mode: x2 == x1 shifted)
    vc_view off1, off2;    // generic: offset tensor halves;   fused: raw offset-net outputs (27 * G/2 channels)
    vc_view msk1, msk2;    // generic: mask halves (p may be 0); fused: unused
    vc_view flow1, flow2;  // fused only
    vc_view out;
    const float *wpk, *bias;
    float magnitude;
    int groups;
};

// One workgroup = one 8x8 pixel tile x all G/2 groups of ONE reference (wave w <-> group w of that half): the
// waves of a workgroup then consume every channel of the 128-byte lines they pull in (a group alone uses only
// cg of the C channels of a pixel), and the per-pixel offset/mask record of that reference is read completely
// by the workgroup.  Lanes of a wave are the 64 pixels of the tile, so the group (and its weights) is wave-uniform.
template <int CG, int OG, bool FUSED, bool VEC>
__global__ void __launch_bounds__(1024) k_deform(DeformArgs a)
{
    extern __shared__ float wsm_all[];
    const int half = a.groups / 2;
    const bool second = blockIdx.y != 0;
    const int n = blockIdx.z;
    const int nthreads = 64 * half;
    const float *wsrc = a.wpk + (long long)(second ? half : 0) * 9 * CG * OG;
    for (int i = threadIdx.x; i < half * 9 * CG * OG; i += nthreads) wsm_all[i] = wsrc[i];
    const int gl = threadIdx.x >> 6;                  // group inside its half == wave index
    const int g = second ? gl + half : gl;
    const int lane = threadIdx.x & 63;
    const float *wsm = wsm_all + gl * 9 * CG * OG;
    const vc_view &X = second ? a.x2 : a.x1;
    const vc_view &O = second ? a.off2 : a.off1;
    const vc_view &M = second ? a.msk2 : a.msk1;
    const vc_view &FL = second ? a.flow2 : a.flow1;
    const int H = a.out.h, W = a.out.w;
    const int tiles_x = (W + 7) >> 3;
    const int ty0 = (blockIdx.x / tiles_x) * 8, tx0 = (blockIdx.x % tiles_x) * 8;
    const int y = ty0 + (lane >> 3), x = tx0 + (lane & 7);
    // FUSED: the 27*half-float offset/mask record of each of the 64 pixels is fetched ONCE by the workgroup with
    // coalesced loads (a lane reading its own 18+9 values straight from global touches 27 separate 4-byte words
    // 1.7 KB apart from its neighbours': 1.4 of 3.2 ms at 544x960) and parked in LDS at an odd pixel stride.
    const int RS = 27 * half + 1;                      // record stride in floats (odd for half = 8: 217)
    float *rec = wsm_all + half * 9 * CG * OG;
    if (FUSED) {
        const int per_px = 27 * half;
        for (int i = threadIdx.x; i < 64 * per_px; i += nthreads) {
            const int px = i / per_px, c = i - px * per_px;
            const int yy = min(ty0 + (px >> 3), H - 1), xx = min(tx0 + (px & 7), W - 1);
            rec[px * RS + c] = O.p[view_off(O, n, yy, xx) + c];
        }
    }
    __syncthreads();
    if (y >= H || x >= W) return;

    float acc[OG];
#pragma unroll
    for (int o = 0; o < OG; ++o) acc[o] = a.bias ? a.bias[g * OG + o] : 0.0f;

    const float *op = FUSED ? rec + lane * RS + gl * 18 : O.p + view_off(O, n, y, x) + gl * 18;
    const float *mp = FUSED ? rec + lane * RS + half * 18 + gl * 9
                            : (M.p ? M.p + view_off(M, n, y, x) + gl * 9 : nullptr);
    float fu = 0.0f, fv = 0.0f;
    if (FUSED) {
        const float *fp = FL.p + view_off(FL, n, y, x);
        fu = fp[0];
        fv = fp[1];
    }
    const float *xbase = X.p + (long long)n * X.sn + gl * CG;

#pragma unroll 3
    for (int k = 0; k < 9; ++k) {
        float dy = op[2 * k], dx = op[2 * k + 1];
        float m = 1.0f;
        if (FUSED) {
            dy = tanhf(dy) * a.magnitude + fv;        // flow.flip(1): (v, u) pairs with (dy, dx)
            dx = tanhf(dx) * a.magnitude + fu;
            m = 1.0f / (1.0f + expf(-mp[k]));
        } else if (mp) {
            m = mp[k];
        }
        const float py = (float)(y - 1 + k / 3) + dy;
        const float px = (float)(x - 1 + k % 3) + dx;
        if (!(py > -1.0f && py < (float)H && px > -1.0f && px < (float)W)) continue;   // also rejects NaN
        const float fy = floorf(py), fx = floorf(px);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        const float lh = py - fy, lw = px - fx, hh = 1.0f - lh, hw = 1.0f - lw;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const bool t = y0 >= 0, b = y1 <= H - 1, l = x0 >= 0, r = x1 <= W - 1;
        const float *p1 = xbase + (long long)y0 * X.sh + (long long)x0 * X.sw;
        const float *p2 = p1 + X.sw, *p3 = p1 + X.sh, *p4 = p3 + X.sw;
        float val[CG];
        if (VEC) {
#pragma unroll
            for (int c = 0; c < CG; c += 4) {
                const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 v1 = (t && l) ? *reinterpret_cast<const f32x4 *>(p1 + c) : z;
                const f32x4 v2 = (t && r) ? *reinterpret_cast<const f32x4 *>(p2 + c) : z;
                const f32x4 v3 = (b && l) ? *reinterpret_cast<const f32x4 *>(p3 + c) : z;
                const f32x4 v4 = (b && r) ? *reinterpret_cast<const f32x4 *>(p4 + c) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) val[c + e] = (w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e]) * m;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float v1 = (t && l) ? p1[c] : 0.0f, v2 = (t && r) ? p2[c] : 0.0f;
                const float v3 = (b && l) ? p3[c] : 0.0f, v4 = (b && r) ? p4[c] : 0.0f;
                val[c] = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * m;
            }
        }
        const float *wk = wsm + k * CG * OG;
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int o = 0; o < OG; ++o) acc[o] = fmaf(wk[c * OG + o], val[c], acc[o]);
    }
    float *outp = a.out.p + view_off(a.out, n, y, x) + g * OG;
#pragma unroll
    for (int o = 0; o < OG; ++o) outp[o] = acc[o];
}

inline bool aligned16(const vc_view &v)
{
    return (reinterpret_cast<uintptr_t>(v.p) % 16 == 0) && v.sn % 4 == 0 && v.sh % 4 == 0 && v.sw % 4 == 0;
}

template <int CG, int OG, bool FUSED> int launch(hipStream_t st, const DeformArgs &a, bool vec)
{
    const int half = a.groups / 2;
    if (half > 16) return VC_EINVAL;                       // one wave per group of a half, 1024 threads at most
    const unsigned tiles = (unsigned)(((a.out.h + 7) / 8) * ((a.out.w + 7) / 8));
    const dim3 grid(tiles, 2u, (unsigned)a.out.n), block((unsigned)(64 * half));
    const size_t lds = ((size_t)half * 9 * CG * OG + (FUSED ? 64 * (27 * half + 1) : 0)) * sizeof(float);
    auto launch_one = [&](auto kern) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VC_ELAUNCH;
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
        return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
    };
    return vec ? launch_one(k_deform<CG, OG, FUSED, true>) : launch_one(k_deform<CG, OG, FUSED, false>);
}

template <bool FUSED> int dispatch(hipStream_t st, const DeformArgs &a, int cg, int og)
{
    const bool vec = cg % 4 == 0 && aligned16(a.x1) && aligned16(a.x2);
    switch (cg * 16 + og) {
    case 4 * 16 + 2: return launch<4, 2, FUSED>(st, a, vec);
    case 4 * 16 + 4: return launch<4, 4, FUSED>(st, a, vec);
    case 8 * 16 + 4: return launch<8, 4, FUSED>(st, a, vec);      // C = 64  (level 1)
    case 12 * 16 + 6: return launch<12, 6, FUSED>(st, a, vec);    // C = 96  (level 2)
    case 16 * 16 + 8: return launch<16, 8, FUSED>(st, a, vec);    // C = 128 (level 3)
    case 8 * 16 + 8: return launch<8, 8, FUSED>(st, a, vec);
    case 16 * 16 + 4: return launch<16, 4, FUSED>(st, a, vec);
    }
    return VC_EINVAL;
}

}  // namespace

extern "C" int vc_deform_pack_weights(const float *w, int cout, int cg, int groups, float *dst)
{
    if (!w || !dst || groups < 1 || cout % groups || cg < 1) return VC_EINVAL;
    const int og = cout / groups;
    for (int g = 0; g < groups; ++g)
        for (int k = 0; k < 9; ++k)
            for (int c = 0; c < cg; ++c)
                for (int o = 0; o < og; ++o)
                    dst[(((long long)g * 9 + k) * cg + c) * og + o] = w[(((long long)(g * og + o)) * cg + c) * 9 + k];
    return VC_OK;
}

extern "C" int vc_deform_conv2d(vc_stream s, vc_view in, vc_view offset, vc_view mask, const float *wpk, const float *bias,
                                int groups, vc_view out)
{
    if (!in.p || !offset.p || !wpk || !out.p || groups < 2 || groups % 2) return VC_EINVAL;
    if (in.c % groups || out.c % groups || offset.c != groups * 18 || (mask.p && mask.c != groups * 9)) return VC_EINVAL;
    if (in.h != out.h || in.w != out.w || offset.h != out.h || offset.w != out.w || in.n != out.n || offset.n != out.n)
        return VC_EINVAL;
    if (mask.p && (mask.h != out.h || mask.w != out.w || mask.n != out.n)) return VC_EINVAL;
    const int cg = in.c / groups, og = out.c / groups, half = groups / 2;
    DeformArgs a = {};
    a.x1 = in;
    a.x2 = in;
    a.x2.p = in.p + half * cg;
    a.off1 = offset;
    a.off2 = offset;
    a.off2.p = offset.p + half * 18;
    a.msk1 = mask;
    a.msk2 = mask;
    if (mask.p) a.msk2.p = mask.p + half * 9;
    a.out = out;
    a.wpk = wpk;
    a.bias = bias;
    a.groups = groups;
    return dispatch<false>(as_stream(s), a, cg, og);
}

extern "C" int vc_offset_diversity(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2,
                                   vc_view flow2, float magnitude, const float *wpk, const float *bias, int groups,
                                   vc_view out)
{
    if (!x1.p || !x2.p || !raw1.p || !raw2.p || !flow1.p || !flow2.p || !wpk || !out.p) return VC_EINVAL;
    if (groups < 2 || groups % 2) return VC_EINVAL;
    const int half = groups / 2;
    if (x1.c != x2.c || x1.c % half || out.c % groups || raw1.c != 27 * half || raw2.c != 27 * half) return VC_EINVAL;
    if (flow1.c < 2 || flow2.c < 2) return VC_EINVAL;
    const vc_view *vs[] = {&x1, &x2, &raw1, &raw2, &flow1, &flow2};
    for (const vc_view *v : vs)
        if (v->h != out.h || v->w != out.w || v->n != out.n) return VC_EINVAL;
    DeformArgs a = {};
    a.x1 = x1;
    a.x2 = x2;
    a.off1 = raw1;
    a.off2 = raw2;
    a.flow1 = flow1;
    a.flow2 = flow2;
    a.out = out;
    a.wpk = wpk;
    a.bias = bias;
    a.magnitude = magnitude;
    a.groups = groups;
    return dispatch<true>(as_stream(s), a, x1.c / half, out.c / groups);
}
