// Modulated deformable 3x3 convolution (torchvision.ops.deform_conv2d semantics) for the ICIP2024 OffsetDiversity
// fusion (ICIP2024/src/model/helpers.py:35-58).  Gather-bound, not matrix-bound: per output pixel and group the
// kernel samples 9 taps x 4 corners x cg channels (cg = 8..16 contiguous values) and contracts them with
// 9*cg*og weights; one lane owns one (pixel, group) pair and the group index is uniform per wave
// (the vector path reads the weights through the scalar cache, the scalar path from LDS).  The offset/mask preparation of OffsetDiversity.prep (tanh *
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
    int x_half;            // fp16 path: x1 / x2 point at half-precision features (strides in elements); 2: group-planar ones
    int twl;               // log2 of the pixel tile's width: a wave's 64 lanes are a (64 >> twl) x (1 << twl) tile
    unsigned gstride;      // vector path: bytes from one group's channels to the next one's (cg * element size; planar: h * w * cg * 2)
};

// Branch-free tanh / logistic for the half-precision vector path: the library's tanhf spelled out (its polynomial below 0.625,
// 1 - 2 / (e^2|x| + 1) with the library's expf and the hardware reciprocal above) as ONE straight-line sequence with a select
// instead of two branch sides that a divergent wave both runs -- the same bits as tanhf, which the fp32 instances call; the
// logistic is their expression too.  (tests/test_icip2024_gpu.py holds the two instances together at 1e-6.)
__device__ __forceinline__ float tanh_bf(float x)
{
    const float ax = fabsf(x), x2 = x * x;
    const float big = fmaf(__builtin_amdgcn_rcpf(expf(2.0f * ax) + 1.0f), -2.0f, 1.0f);   // e^(2|x|) = +inf for large |x| -> 1
    float p = fmaf(x2, -0x1.758e7ap-8f, 0x1.521192p-6f);
    p = fmaf(x2, p, -0x1.b8389cp-5f);
    p = fmaf(x2, p, 0x1.110704p-3f);
    p = fmaf(x2, p, -0x1.555532p-2f);
    const float small = fmaf(x2, ax * p, ax);
    return copysignf(ax < 0.625f ? small : big, x);                            // (NaN: the comparison is false, `big` is NaN)
}

__device__ __forceinline__ float sigmoid_bf(float x)
{
    return 1.0f / (1.0f + expf(-x));
}

// One workgroup = one 4 x 16 pixel tile (args.twl) x all G/2 groups of ONE reference (wave w <-> group w of that half): the
// waves of a workgroup then consume every channel of the 128-byte lines they pull in (a group alone uses only
// cg of the C channels of a pixel), and the per-pixel offset/mask record of that reference is read completely
// by the workgroup.  Lanes of a wave are the 64 pixels of the tile, so the group (and its weights) is wave-uniform.
// With pixel-interleaved features every lane of a gather meets its own 128-byte line (a pixel is C * 2 or C * 4 bytes);
// the fp16 path therefore gathers from GROUP-PLANAR half features (vc_to_half_planar, vc_offset_diversity_hxp: one group's
// cg halves of neighbouring pixels are neighbours in memory), 8-16 lines per gather when neighbouring pixels sample
// neighbouring positions: 3.21 -> 2.10 ms on the 128 -> 64 fusion @1088x1920 of the configs[4] frame, bit-identical results.
// MAXT: 512 when the half has at most 8 groups (the ICIP2024 model: 16 groups) -- leaves the register file for the batched
// gathers; 1024 (up to 16 groups per half) keeps 128 registers and gathers one tap at a time.
template <int CG, int OG, bool FUSED, bool VEC, int RV = 1, int MAXT = 1024, bool XH = false>
__global__ void __launch_bounds__(MAXT) k_deform(DeformArgs a)
{
    extern __shared__ float wsm_all[];
    const int half = a.groups / 2;
    const bool second = blockIdx.y != 0;
    const int n = blockIdx.z;
    const int nthreads = 64 * half;
    const float *wsrc = a.wpk + (long long)(second ? half : 0) * 9 * CG * OG;
    constexpr bool WLDS = !(VEC && XH);                // (the half-precision vector path reads its group's weights through the scalar cache)
    if constexpr (WLDS)
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
    const int twl = a.twl, twm = (1 << twl) - 1;
    const int tiles_x = (W + twm) >> twl;
    const int ty0 = (blockIdx.x / tiles_x) * (64 >> twl), tx0 = (blockIdx.x % tiles_x) << twl;
    const int y = ty0 + (lane >> twl), x = tx0 + (lane & twm);
    // FUSED: the 27*half-float offset/mask record of each of the 64 pixels is fetched ONCE by the workgroup with
    // coalesced loads (a lane reading its own 18+9 values straight from global touches 27 separate 4-byte words
    // 1.7 KB apart from its neighbours': 1.4 of 3.2 ms at 544x960) and parked in LDS at an odd pixel stride.
    const int RS = 27 * half + 1;                      // record stride in floats (odd for half = 8: 217)
    float *rec = wsm_all + (WLDS ? half * 9 * CG * OG : 0);
    if (FUSED) {
        // `half` threads per pixel, thread s of a pixel takes the RV-float pieces s, s + half, ... of its record: a wave
        // instruction reads whole contiguous stretches (128 bytes per pixel for RV = 4, half = 8), and the (27 + RV - 1) / RV
        // loads of a thread are all issued before the first one is waited for.  (The first version walked the record
        // with one 4-byte load per thread and iteration, 27 dependent global round trips per workgroup: 54 us of its 60.)
        constexpr int IT = (27 + RV - 1) / RV;
        const int per_px = 27 * half, pieces = per_px / RV;           // per_px % RV == 0 (checked by the launcher)
        const int px = threadIdx.x / half, s = threadIdx.x - px * half;
        const int yy = min(ty0 + (px >> twl), H - 1), xx = min(tx0 + (px & twm), W - 1);
        const float *src = O.p + view_off(O, n, yy, xx);
        float v[IT][RV];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int pc = min(s + j * half, pieces - 1);
            if constexpr (RV == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(src + 4 * pc);
                v[j][0] = t.x, v[j][1] = t.y, v[j][2] = t.z, v[j][3] = t.w;
            } else if constexpr (RV == 2) {
                const float2 t = *reinterpret_cast<const float2 *>(src + 2 * pc);
                v[j][0] = t.x, v[j][1] = t.y;
            } else {
                v[j][0] = src[pc];
            }
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int pc = s + j * half;
            if (pc < pieces) {
#pragma unroll
                for (int e = 0; e < RV; ++e) rec[px * RS + RV * pc + e] = v[j][e];     // (odd record stride: scalar LDS writes)
            }
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
    // (XH: half-precision features -- a group's CG channels of a pixel are CG / 8 gathers of 16 bytes instead of CG / 4; the
    //  kernel is bound by the gathers, every lane of an instruction in a different 128-byte line, so that halves it)
    constexpr int ESZ = XH ? 2 : 4;
    const unsigned char *xbase = reinterpret_cast<const unsigned char *>(X.p) + ((long long)n * X.sn + gl * CG) * ESZ;

    // One tap's sampling geometry.  Nothing below sits under a branch: a corner (or a whole tap) outside the image gets
    // its address clamped into the image and its VALUE replaced by zero after the load, so the 4 * CG / 4 gathers of all
    // taps of a batch are issued back to back and their L2 round trips overlap.  (With the loads under `if`s the compiler
    // waited for each of them in turn: ~45 dependent round trips per wave, 53 us per workgroup at 1088x1920.)
    struct Tap {
        const unsigned char *p1, *p2, *p3, *p4;
        float w1, w2, w3, w4, m;
        bool ok, tl, tr, bl, br;
    };
    auto tap = [&](int k) {
        Tap q;
        float dy = op[2 * k], dx = op[2 * k + 1];
        q.m = 1.0f;
        if (FUSED) {
            dy = tanhf(dy) * a.magnitude + fv;        // flow.flip(1): (v, u) pairs with (dy, dx)
            dx = tanhf(dx) * a.magnitude + fu;
            q.m = 1.0f / (1.0f + expf(-mp[k]));
        } else if (mp) {
            q.m = mp[k];
        }
        const float py = (float)(y - 1 + k / 3) + dy;
        const float px = (float)(x - 1 + k % 3) + dx;
        q.ok = py > -1.0f && py < (float)H && px > -1.0f && px < (float)W;      // also rejects NaN
        const float fy = floorf(py), fx = floorf(px);
        const int y0 = q.ok ? (int)fy : 0, x0 = q.ok ? (int)fx : 0, y1 = y0 + 1, x1 = x0 + 1;
        const float lh = py - fy, lw = px - fx, hh = 1.0f - lh, hw = 1.0f - lw;
        q.w1 = hh * hw, q.w2 = hh * lw, q.w3 = lh * hw, q.w4 = lh * lw;
        const bool t = y0 >= 0, b = y1 <= H - 1, l = x0 >= 0, r = x1 <= W - 1;
        q.tl = q.ok && t && l, q.tr = q.ok && t && r, q.bl = q.ok && b && l, q.br = q.ok && b && r;
        const long long r0 = (long long)max(y0, 0) * X.sh * ESZ, r1 = (long long)min(y1, H - 1) * X.sh * ESZ;
        const long long c0 = (long long)max(x0, 0) * X.sw * ESZ, c1 = (long long)min(x1, W - 1) * X.sw * ESZ;
        q.p1 = xbase + r0 + c0, q.p2 = xbase + r0 + c1, q.p3 = xbase + r1 + c0, q.p4 = xbase + r1 + c1;
        return q;
    };
    if constexpr (VEC && XH) {
        // Half-precision features (round 6).  The round-5 form of this path issued ~3 200 vector instructions per (pixel, group):
        // 12 of every 356 per tap quarter-rate 64-bit multiplies of the address arithmetic, ~95 the two library tanhf calls (both
        // sides of their branch run in a divergent wave).  Here ~1 700: 32-bit byte offsets from a wave-uniform base (24-bit
        // multiplies; the launcher refuses images above 4 GiB), branch-free tanh (the library's arithmetic, both sides computed,
        // one select), the out-of-image zeroing on the four bilinear weights instead of the gathered values, two channels per
        // packed-fp32 instruction, the group's weights through the scalar cache (wave-uniform: no LDS, no vector registers).
        // That alone bought 3 % (3.34 -> 3.22 ms, C = 64 @1088x1920): the kernel is bound by its gathers, by how many cache lines
        // each one meets -- what pays is the group-planar layout of the features (a.gstride; 3.22 -> 2.36 ms), and NOT more gathers
        // in flight: a fence that issues a whole batch before its first use costs 30 %.
#ifndef VC_DEFORM_TB
#define VC_DEFORM_TB 3
#endif
        constexpr int TB = (MAXT > 512 || CG > 8) ? 1 : VC_DEFORM_TB;       // taps per batch (divides 9; one tap per batch measured 5 % slower)
        constexpr int V = (CG + 7) / 8;                          // gathers per corner: 16 bytes each; CG % 8 == 4: the last one 8 bytes
        constexpr int EPV = 8;                                   // channels per gather
        constexpr bool TAIL8 = (CG % 8) == 4;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const unsigned char *xb = reinterpret_cast<const unsigned char *>(X.p) + (long long)n * X.sn * ESZ;      // wave-uniform
        const unsigned rowB = (unsigned)(X.sh * ESZ), pixB = (unsigned)(X.sw * ESZ), goff = (unsigned)gl * a.gstride;
        const int gu = __builtin_amdgcn_readfirstlane(g);
        const float *wg = a.wpk + (long long)gu * 9 * CG * OG;
        struct TapV {
            unsigned o1, o2, o3, o4;
            float w1, w2, w3, w4, m;
        };
        auto tapv = [&](int k) {
            TapV q;
            float dy = op[2 * k], dx = op[2 * k + 1], m = 1.0f;
            if (FUSED) {
                dy = fmaf(tanh_bf(dy), a.magnitude, fv);          // flow.flip(1): (v, u) pairs with (dy, dx)
                dx = fmaf(tanh_bf(dx), a.magnitude, fu);
                m = sigmoid_bf(mp[k]);
            } else if (mp) {
                m = mp[k];
            }
            const float py = (float)(y - 1 + k / 3) + dy;
            const float px = (float)(x - 1 + k % 3) + dx;
            const bool ok = py > -1.0f && py < (float)H && px > -1.0f && px < (float)W;      // also rejects NaN
            const float fy = floorf(py), fx = floorf(px);
            const int y0 = ok ? (int)fy : 0, x0 = ok ? (int)fx : 0, y1 = y0 + 1, x1 = x0 + 1;
            const float lh = py - fy, lw = px - fx, hh = 1.0f - lh, hw = 1.0f - lw;
            const bool t = y0 >= 0, b = y1 <= H - 1, l = x0 >= 0, r = x1 <= W - 1;
            // (a corner -- or a whole tap -- outside the image: address clamped into the image, WEIGHT zero; a NaN position ends here too)
            q.w1 = (ok && t && l) ? hh * hw : 0.0f, q.w2 = (ok && t && r) ? hh * lw : 0.0f;
            q.w3 = (ok && b && l) ? lh * hw : 0.0f, q.w4 = (ok && b && r) ? lh * lw : 0.0f;
            q.m = m;
            const unsigned r0 = __umul24((unsigned)max(y0, 0), rowB) + goff, r1 = __umul24((unsigned)min(y1, H - 1), rowB) + goff;
            const unsigned c0 = __umul24((unsigned)max(x0, 0), pixB), c1 = __umul24((unsigned)min(x1, W - 1), pixB);
            q.o1 = r0 + c0, q.o2 = r0 + c1, q.o3 = r1 + c0, q.o4 = r1 + c1;
            return q;
        };
        f32x2 acc2[OG / 2];
#pragma unroll
        for (int o = 0; o < OG / 2; ++o) acc2[o] = f32x2{acc[2 * o], acc[2 * o + 1]};
#pragma unroll 1
        for (int k0 = 0; k0 < 9; k0 += TB) {     // (rolled: unrolled, the compiler hoists all 9 * CG * OG weight reads and spills)
            TapV q[TB];
            f32x4 v[TB][4][V];
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                q[j] = tapv(k0 + j);
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    if (TAIL8 && c == V - 1) {
                        const float2 t1 = *reinterpret_cast<const float2 *>(xb + (size_t)(q[j].o1 + 16 * c)), t2 = *reinterpret_cast<const float2 *>(xb + (size_t)(q[j].o2 + 16 * c));
                        const float2 t3 = *reinterpret_cast<const float2 *>(xb + (size_t)(q[j].o3 + 16 * c)), t4 = *reinterpret_cast<const float2 *>(xb + (size_t)(q[j].o4 + 16 * c));
                        v[j][0][c] = f32x4{t1.x, t1.y, 0.f, 0.f}, v[j][1][c] = f32x4{t2.x, t2.y, 0.f, 0.f};
                        v[j][2][c] = f32x4{t3.x, t3.y, 0.f, 0.f}, v[j][3][c] = f32x4{t4.x, t4.y, 0.f, 0.f};
                    } else {
                        v[j][0][c] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(q[j].o1 + 16 * c));
                        v[j][1][c] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(q[j].o2 + 16 * c));
                        v[j][2][c] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(q[j].o3 + 16 * c));
                        v[j][3][c] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(q[j].o4 + 16 * c));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const float *wk = wg + (k0 + j) * CG * OG;
                const f32x2 w1 = {q[j].w1, q[j].w1}, w2 = {q[j].w2, q[j].w2}, w3 = {q[j].w3, q[j].w3}, w4 = {q[j].w4, q[j].w4}, mm = {q[j].m, q[j].m};
#pragma unroll
                for (int c = 0; c < V; ++c) {
#pragma unroll
                    for (int e = 0; e < ((TAIL8 && c == V - 1) ? 4 : EPV); e += 2) {
                        const h8 h1 = __builtin_bit_cast(h8, v[j][0][c]), h2 = __builtin_bit_cast(h8, v[j][1][c]);
                        const h8 h3 = __builtin_bit_cast(h8, v[j][2][c]), h4 = __builtin_bit_cast(h8, v[j][3][c]);
                        const f32x2 e1 = {(float)h1[e], (float)h1[e + 1]}, e2 = {(float)h2[e], (float)h2[e + 1]};
                        const f32x2 e3 = {(float)h3[e], (float)h3[e + 1]}, e4 = {(float)h4[e], (float)h4[e + 1]};
                        f32x2 val = e1 * w1;
                        val = __builtin_elementwise_fma(e2, w2, val);
                        val = __builtin_elementwise_fma(e3, w3, val);
                        val = __builtin_elementwise_fma(e4, w4, val) * mm;
                        const float *wc = wk + (EPV * c + e) * OG;
#pragma unroll
                        for (int o = 0; o < OG / 2; ++o) {
                            acc2[o] = __builtin_elementwise_fma(f32x2{wc[2 * o], wc[2 * o + 1]}, f32x2{val.x, val.x}, acc2[o]);
                            acc2[o] = __builtin_elementwise_fma(f32x2{wc[OG + 2 * o], wc[OG + 2 * o + 1]}, f32x2{val.y, val.y}, acc2[o]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OG / 2; ++o) acc[2 * o] = acc2[o].x, acc[2 * o + 1] = acc2[o].y;
    } else if constexpr (VEC) {
        // fp32 features: the round-2 form (64-bit addresses, library tanhf / expf, weights from LDS).  The half-precision form above
        // measured 7-30 % SLOWER here on the same box (1.47 / 0.63 / 0.19 ms against 1.37 / 0.48 / 0.15 at the three ICIP2024 levels):
        // its shorter instruction stream puts more gathers in flight at once, and this kernel loses L1 hits when it does.
        constexpr int TB = (MAXT > 512 || CG > 8) ? 1 : 3;       // taps per batch (divides 9): 12-24 16-byte gathers in flight per lane
        constexpr int V = XH ? (CG + 7) / 8 : CG / 4;            // gathers per corner: 16 bytes each; XH with CG % 8 == 4: the last one 8 bytes
        constexpr int EPV = XH ? 8 : 4;                          // channels per gather
        constexpr bool TAIL8 = XH && (CG % 8) == 4;
#pragma unroll 1
        for (int k0 = 0; k0 < 9; k0 += TB) {     // (rolled: unrolled, the compiler hoists all 9 * CG * OG weight reads and spills)
            Tap q[TB];
            f32x4 v[TB][4][V];
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                q[j] = tap(k0 + j);
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    if (TAIL8 && c == V - 1) {
                        const float2 t1 = *reinterpret_cast<const float2 *>(q[j].p1 + 16 * c), t2 = *reinterpret_cast<const float2 *>(q[j].p2 + 16 * c);
                        const float2 t3 = *reinterpret_cast<const float2 *>(q[j].p3 + 16 * c), t4 = *reinterpret_cast<const float2 *>(q[j].p4 + 16 * c);
                        v[j][0][c] = f32x4{t1.x, t1.y, 0.f, 0.f}, v[j][1][c] = f32x4{t2.x, t2.y, 0.f, 0.f};
                        v[j][2][c] = f32x4{t3.x, t3.y, 0.f, 0.f}, v[j][3][c] = f32x4{t4.x, t4.y, 0.f, 0.f};
                    } else {
                        v[j][0][c] = *reinterpret_cast<const f32x4 *>(q[j].p1 + 16 * c);
                        v[j][1][c] = *reinterpret_cast<const f32x4 *>(q[j].p2 + 16 * c);
                        v[j][2][c] = *reinterpret_cast<const f32x4 *>(q[j].p3 + 16 * c);
                        v[j][3][c] = *reinterpret_cast<const f32x4 *>(q[j].p4 + 16 * c);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const float *wk = wsm + (k0 + j) * CG * OG;
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                    const f32x4 v1 = q[j].tl ? v[j][0][c] : z, v2 = q[j].tr ? v[j][1][c] : z;      // (half +0.0 == fp32 +0.0 bit pattern 0)
                    const f32x4 v3 = q[j].bl ? v[j][2][c] : z, v4 = q[j].br ? v[j][3][c] : z;
                    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
                    for (int e = 0; e < ((TAIL8 && c == V - 1) ? 4 : EPV); ++e) {
                        float e1, e2, e3, e4;
                        if constexpr (XH) {
                            e1 = (float)__builtin_bit_cast(h8, v1)[e], e2 = (float)__builtin_bit_cast(h8, v2)[e];
                            e3 = (float)__builtin_bit_cast(h8, v3)[e], e4 = (float)__builtin_bit_cast(h8, v4)[e];
                        } else {
                            e1 = v1[e < 4 ? e : 0], e2 = v2[e < 4 ? e : 0], e3 = v3[e < 4 ? e : 0], e4 = v4[e < 4 ? e : 0];
                        }
                        // (a tap outside the image contributes nothing, whatever its modulation value)
                        const float val = q[j].ok ? (q[j].w1 * e1 + q[j].w2 * e2 + q[j].w3 * e3 + q[j].w4 * e4) * q[j].m : 0.0f;
#pragma unroll
                        for (int o = 0; o < OG; ++o) acc[o] = fmaf(wk[(EPV * c + e) * OG + o], val, acc[o]);
                    }
                }
            }
        }
    } else {
#pragma unroll 3
        for (int k = 0; k < 9; ++k) {
            const Tap q = tap(k);
            if (!q.ok) continue;
            const float *wk = wsm + k * CG * OG;
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float *f1 = reinterpret_cast<const float *>(q.p1), *f2 = reinterpret_cast<const float *>(q.p2);
                const float *f3 = reinterpret_cast<const float *>(q.p3), *f4 = reinterpret_cast<const float *>(q.p4);
                const float v1 = q.tl ? f1[c] : 0.0f, v2 = q.tr ? f2[c] : 0.0f;
                const float v3 = q.bl ? f3[c] : 0.0f, v4 = q.br ? f4[c] : 0.0f;
                const float val = (q.w1 * v1 + q.w2 * v2 + q.w3 * v3 + q.w4 * v4) * q.m;
#pragma unroll
                for (int o = 0; o < OG; ++o) acc[o] = fmaf(wk[c * OG + o], val, acc[o]);
            }
        }
    }
    float *outp = a.out.p + view_off(a.out, n, y, x) + g * OG;
#pragma unroll
    for (int o = 0; o < OG; ++o) outp[o] = acc[o];
}

inline bool aligned16(const vc_view &v)
{
    return (reinterpret_cast<uintptr_t>(v.p) % 16 == 0) && v.sn % 4 == 0 && v.sh % 4 == 0 && v.sw % 4 == 0;
}

template <int CG, int OG, bool FUSED> int launch(hipStream_t st, const DeformArgs &a_in, bool vec)
{
    DeformArgs a = a_in;
    a.gstride = a.x_half == 2 ? (unsigned)((long long)a.x1.h * a.x1.sh * 2) : (unsigned)(CG * (a.x_half ? 2 : 4));
    const int half = a.groups / 2;
    if (half > 16) return VC_EINVAL;                       // one wave per group of a half, 1024 threads at most
    // pixels per wave: 4 x 16 on group-planar features, 8 x 8 on pixel-interleaved ones (measured against each other and 2 x 32, 1 x 64)
    a.twl = a.x_half == 2 ? 4 : 3;
    const int tw = 1 << a.twl, th = 64 >> a.twl;
    const unsigned tiles = (unsigned)(((a.out.h + th - 1) / th) * ((a.out.w + tw - 1) / tw));
    const dim3 grid(tiles, 2u, (unsigned)a.out.n), block((unsigned)(64 * half));
    // half-precision features: 32-bit byte offsets inside one image of the features (24-bit row / pixel multiplies)
    auto fits32 = [&](const vc_view &v) {
        return (long long)v.h * v.sh * 2 * (a.x_half == 2 ? a.groups / 2 : 1) < (1ll << 32) && v.sh * 2 < (1ll << 24) && v.sw * 2 < (1ll << 24) && v.h < (1 << 24) && v.w < (1 << 24);
    };
    if (a.x_half && !(fits32(a.x1) && fits32(a.x2))) return VC_EINVAL;
    const size_t lds = (((vec && a.x_half) ? (size_t)0 : (size_t)half * 9 * CG * OG) + (FUSED ? 64 * (27 * half + 1) : 0)) * sizeof(float);
    auto launch_one = [&](auto kern) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VC_ELAUNCH;
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
        return hipGetLastError() == hipSuccess ? VC_OK : VC_ELAUNCH;
    };
    const bool small = half <= 8;
    if constexpr (FUSED) {
        // width of the record pieces: what the record length and the alignment of the raw offset tensors allow
        auto ok = [&](const vc_view &v, int w) {
            return (27 * half) % w == 0 && reinterpret_cast<uintptr_t>(v.p) % (4 * w) == 0 && v.sn % w == 0 && v.sh % w == 0 && v.sw % w == 0;
        };
        const int rv = (ok(a.off1, 4) && ok(a.off2, 4)) ? 4 : ((ok(a.off1, 2) && ok(a.off2, 2)) ? 2 : 1);
        if (a.x_half) {             // half-precision features: the fast fused instance or nothing
            if constexpr (CG % 4 == 0) {
                if (vec && small && rv == 4) return launch_one(k_deform<CG, OG, true, true, 4, 512, true>);
            }
            return VC_EINVAL;
        }
        if (vec && small && rv == 4) return launch_one(k_deform<CG, OG, true, true, 4, 512>);
        if (rv == 4) return vec ? launch_one(k_deform<CG, OG, true, true, 4>) : launch_one(k_deform<CG, OG, true, false, 4>);
        if (rv == 2) return vec ? launch_one(k_deform<CG, OG, true, true, 2>) : launch_one(k_deform<CG, OG, true, false, 2>);
    }
    if (vec && small) return launch_one(k_deform<CG, OG, FUSED, true, 1, 512>);
    return vec ? launch_one(k_deform<CG, OG, FUSED, true>) : launch_one(k_deform<CG, OG, FUSED, false>);
}

template <bool FUSED> int dispatch(hipStream_t st, const DeformArgs &a, int cg, int og)
{
    // (half features: a group's 2 * cg bytes start 8-byte aligned; a 16-byte gather that is only 8-byte aligned is legal)
    auto aligned8h = [](const vc_view &v) { return (reinterpret_cast<uintptr_t>(v.p) % 16 == 0) && v.sn % 4 == 0 && v.sh % 4 == 0 && v.sw % 4 == 0; };
    const bool vec = a.x_half ? (cg % 4 == 0 && aligned8h(a.x1) && aligned8h(a.x2)) : (cg % 4 == 0 && aligned16(a.x1) && aligned16(a.x2));
    if (a.x_half && !FUSED) return VC_EINVAL;
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

static int offset_diversity(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2, vc_view flow2, float magnitude,
                            const float *wpk, const float *bias, int groups, vc_view out, int x_half);

extern "C" int vc_offset_diversity(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2,
                                   vc_view flow2, float magnitude, const float *wpk, const float *bias, int groups,
                                   vc_view out)
{
    return offset_diversity(s, x1, raw1, flow1, x2, raw2, flow2, magnitude, wpk, bias, groups, out, 0);
}

// The same with HALF-precision features: x1.p / x2.p point at _Float16 tensors (strides in elements; vc_to_half makes one).
// fp16 path only -- the gathered values are the features rounded to half, everything else (offsets, modulation, bilinear
// weights, accumulation) stays fp32; 8 or 16 channels per group, offsets / output as in vc_offset_diversity.
extern "C" int vc_offset_diversity_hx(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2,
                                      vc_view flow2, float magnitude, const float *wpk, const float *bias, int groups,
                                      vc_view out)
{
    return offset_diversity(s, x1, raw1, flow1, x2, raw2, flow2, magnitude, wpk, bias, groups, out, 1);
}

// The same on GROUP-PLANAR half features ([n][G/2][h][w][cg], vc_to_half_planar): x1 / x2 describe ONE group's plane (c = cg, sw = cg,
// sh = w * cg, sn = (G/2) * h * w * cg; the planes of an image follow one another).  Neighbouring pixels of a group are then 2 * cg bytes
// apart instead of a whole pixel: the 64 lanes of a gather (a 4 x 16 pixel tile of one group) meet 8-16 cache lines instead of 64.
extern "C" int vc_offset_diversity_hxp(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2,
                                       vc_view flow2, float magnitude, const float *wpk, const float *bias, int groups,
                                       vc_view out)
{
    return offset_diversity(s, x1, raw1, flow1, x2, raw2, flow2, magnitude, wpk, bias, groups, out, 2);
}

static int offset_diversity(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2, vc_view flow2, float magnitude,
                            const float *wpk, const float *bias, int groups, vc_view out, int x_half)
{
    if (!x1.p || !x2.p || !raw1.p || !raw2.p || !flow1.p || !flow2.p || !wpk || !out.p) return VC_EINVAL;
    if (groups < 2 || groups % 2) return VC_EINVAL;
    const int half = groups / 2;
    if (x_half == 2) {          // planar: the views describe one group's plane
        if (x1.c != x2.c || x1.sw != x1.c || x2.sw != x2.c || x1.sh != (long long)x1.w * x1.c || x2.sh != (long long)x2.w * x2.c) return VC_EINVAL;
        if (x1.sn != (long long)half * x1.h * x1.sh || x2.sn != (long long)half * x2.h * x2.sh) return VC_EINVAL;
        x1.c *= half, x2.c *= half;
    }
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
    a.x_half = x_half;
    return dispatch<true>(as_stream(s), a, x1.c / half, out.c / groups);
}
