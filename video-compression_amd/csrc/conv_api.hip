// C-ABI front end of the convolution engine: tile-config selection, weight packing, launch.
#include <stdlib.h>
#include <string.h>
#include "conv_mfma.h"

// N-tile width (output channels per MFMA tile), k-step (channels per packed fragment) and lanes per k-group
static inline int cfg_nt(int cfg) { return cfg == VC_CFG_N4 ? 4 : (cfg == VC_CFG_N16 ? 16 : 32); }
static inline int cfg_ks(int cfg) { return cfg == VC_CFG_N4 ? 4 : (cfg == VC_CFG_N16 ? 16 : 8); }
static inline int cfg_bn(int cfg)
{
    switch (cfg) {
    case VC_CFG_N128:
    case VC_CFG_N128B: return 128;
    case VC_CFG_N64: return 64;
    case VC_CFG_N32:
    case VC_CFG_N32T16: return 32;
    case VC_CFG_N4: return 4;
    default: return 16;
    }
}
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

extern "C" int vc_conv_select_cfg(int cout, int cin, int k, int stride)
{
    (void)cin;
    int cfg;
    if (cout <= 4 && stride == 1 && (k == 3 || k == 5 || k == 7)) cfg = VC_CFG_N4;
    else if (cout <= 16) cfg = VC_CFG_N16;
    else if (cout <= 32) cfg = VC_CFG_N32;
    else if (cout <= 64 || (cout % 128 != 0 && cout % 64 == 0)) cfg = VC_CFG_N64;
    else cfg = VC_CFG_N128;
    if (cfg == VC_CFG_N16 && stride == 2) cfg = VC_CFG_N32;  // no 16-wide stride-2 instance
    (void)k;
    return cfg;
}

extern "C" int vc_conv_chunk(int cfg, int k, int stride, int cin)
{
    if (cfg == VC_CFG_SPLIT) return stride != 1 ? -1 : ((k == 5 || k == 7) ? 8 : (k == 3 ? 16 : -1));   // planes of 8 channels of a split tensor per chunk
    if (cfg == VC_CFG_N4) {   // 64-pixel-wide tiles: smaller channel chunks keep the footprint in LDS
        if (stride != 1) return -1;
        return k == 3 ? 16 : ((k == 5 || k == 7) ? 8 : -1);
    }
    switch (k) {
    case 1: return 32;
    case 3: return stride == 2 ? 8 : 32;
    case 5: return stride == 2 ? 8 : 16;
    case 7: return ((cfg == VC_CFG_N32 || cfg == VC_CFG_N32T16) && cin <= 8) ? 8 : 16;   // (SPyNet's first layer: 8 input channels)
    }
    return -1;
}

extern "C" size_t vc_conv_packed_weight_floats(int cfg, int cout, int cin, int kh, int kw, int stride)
{
    const int ck = vc_conv_chunk(cfg, kh, stride, cin);
    if (ck <= 0) return 0;
    const int cin_pad = round_up(cin, ck), cout_pad = round_up(cout, cfg_bn(cfg));
    // one 64-lane x 4-float fragment per (N-tile, tap, k-step); N4 replicates its 16 weights over the 16 blocks
    return (size_t)(cout_pad / cfg_nt(cfg)) * kh * kw * (cin_pad / cfg_ks(cfg)) * 256;
}

extern "C" size_t vc_conv_packed_bias_floats(int cfg, int cout) { return (size_t)round_up(cout, cfg_bn(cfg)); }

extern "C" int vc_conv_pack_weights(const float *w, const float *bias, int cout, int cin, int kh, int kw,
                                    int stride, int cfg, int pixelshuffle, float *wpk, float *bpk)
{
    if (kh != kw) return VC_EINVAL;
    const int ck = vc_conv_chunk(cfg, kh, stride, cin);
    if (ck <= 0) return VC_EINVAL;
    if (pixelshuffle && (cout % 4)) return VC_EINVAL;
    const int mt = cfg_nt(cfg), ks = cfg_ks(cfg);
    const int cin_pad = round_up(cin, ck), cout_pad = round_up(cout, cfg_bn(cfg));
    const int taps = kh * kw, ksteps = cin_pad / ks, ntiles = cout_pad / mt;
    const int cps = cout / 4;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int tap = 0; tap < taps; ++tap)
            for (int kst = 0; kst < ksteps; ++kst)
                for (int lane = 0; lane < 64; ++lane) {
                    // N4: every 4-lane block carries the same 4 channels (k = 1 per MFMA, no k-groups)
                    const int j = lane % mt, kk = (cfg == VC_CFG_N4) ? 0 : lane / mt;
                    const int cop = nt * mt + j;  // packed (possibly permuted) output channel
                    int co = cop;
                    if (pixelshuffle && cop < cout) {
                        const int pos = cop / cps, c = cop % cps;
                        co = c * 4 + pos;
                    }
                    float *dst = wpk + ((((size_t)nt * taps + tap) * ksteps + kst) * 64 + lane) * 4;
                    for (int e = 0; e < 4; ++e) {
                        const int ci = kst * ks + kk * 4 + e;
                        float v = 0.0f;
                        if (cop < cout && ci < cin) v = w[(((size_t)co * cin + ci) * kh + tap / kw) * kw + tap % kw];
                        dst[e] = v;
                    }
                }
    for (int cop = 0; cop < cout_pad; ++cop) {
        float v = 0.0f;
        if (cop < cout && bias) {
            int co = cop;
            if (pixelshuffle) co = (cop % cps) * 4 + cop / cps;
            v = bias[co];
        }
        bpk[cop] = v;
    }
    return VC_OK;
}

// ---- fp16 path: same fragment order with 8 halves (16 B) per lane = 16 input channels per k-step ----
static inline bool cfg_f16_ok(int cfg, int cin)
{
    return (cfg == VC_CFG_N128 || cfg == VC_CFG_N64 || cfg == VC_CFG_N32 || cfg == VC_CFG_N128B || cfg == VC_CFG_N16 ||
            cfg == VC_CFG_PW || cfg == VC_CFG_PWS || cfg == VC_CFG_N32T16 || cfg == VC_CFG_DMA) && (cin % 8) == 0;
}

extern "C" size_t vc_conv_packed_weight_bytes_f16(int cfg, int cout, int cin, int kh, int kw, int stride)
{
    const int ck = vc_conv_chunk(cfg, kh, stride, cin);
    if (ck <= 0 || !cfg_f16_ok(cfg, cin)) return 0;
    const int cin_pad = round_up(cin, 2 * ck), cout_pad = round_up(cout, cfg_bn(cfg));
    return (size_t)(cout_pad / cfg_nt(cfg)) * kh * kw * (cin_pad / (2 * cfg_ks(cfg))) * 1024;
}

extern "C" int vc_conv_pack_weights_f16(const float *w, const float *bias, int cout, int cin, int kh, int kw, int stride,
                                        int cfg, int pixelshuffle, void *wpk_out, float *bpk)
{
    if (kh != kw || !cfg_f16_ok(cfg, cin)) return VC_EINVAL;
    const int ck = vc_conv_chunk(cfg, kh, stride, cin);
    if (ck <= 0) return VC_EINVAL;
    if (pixelshuffle && (cout % 4)) return VC_EINVAL;
    _Float16 *wpk = static_cast<_Float16 *>(wpk_out);
    const int cin_pad = round_up(cin, 2 * ck), cout_pad = round_up(cout, cfg_bn(cfg));
    const int mt = cfg_nt(cfg), kch = 2 * cfg_ks(cfg);   // 32 x (2 groups of 8) or 16 x (4 groups of 8) channels per k-step
    const int taps = kh * kw, ksteps = cin_pad / kch, ntiles = cout_pad / mt;
    const int cps = cout / 4;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int tap = 0; tap < taps; ++tap)
            for (int kst = 0; kst < ksteps; ++kst)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane % mt, h = lane / mt;
                    const int cop = nt * mt + j;
                    int co = cop;
                    if (pixelshuffle && cop < cout) co = (cop % cps) * 4 + cop / cps;
                    _Float16 *dst = wpk + ((((size_t)nt * taps + tap) * ksteps + kst) * 64 + lane) * 8;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = kst * kch + h * 8 + e;
                        float v = 0.0f;
                        if (cop < cout && ci < cin) v = w[(((size_t)co * cin + ci) * kh + tap / kw) * kw + tap % kw];
                        dst[e] = (_Float16)v;
                    }
                }
    for (int cop = 0; cop < cout_pad; ++cop) {
        float v = 0.0f;
        if (cop < cout && bias) {
            int co = cop;
            if (pixelshuffle) co = (cop % cps) * 4 + cop / cps;
            v = bias[co];
        }
        bpk[cop] = v;
    }
    return VC_OK;
}

// Weights of the 1x1 layer fused behind a VC_CFG_DMA 3x3 layer (vc_conv_desc.tail_wpk): half-precision MFMA fragments
// [n-tile o][k-step j][lane (m, h)][8] with the k order of the 3x3 layer's ACCUMULATOR layout: value i of lane (m, h) of fragment
// (o, j) is w[32 o + m][16 j + 4 h + (i & 3) + 8 (i >> 2)].  cout = cin = 128 or 64.
extern "C" int vc_conv_pack_tail_f16(const float *w, const float *bias, int cout, int cin, void *wpk_half_out, float *bias_out)
{
    if (!w || !wpk_half_out || !bias_out || cout != cin || (cout != 128 && cout != 64)) return VC_EINVAL;
    _Float16 *dst = static_cast<_Float16 *>(wpk_half_out);
    for (int o = 0; o < cout / 32; ++o)
        for (int j = 0; j < cin / 16; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 31, h = lane >> 5;
                for (int i = 0; i < 8; ++i)
                    dst[(((size_t)o * (cin / 16) + j) * 64 + lane) * 8 + i] =
                        (_Float16)w[(size_t)(32 * o + m) * cin + 16 * j + 4 * h + (i & 3) + 8 * (i >> 2)];
            }
    for (int c = 0; c < cout; ++c) bias_out[c] = bias ? bias[c] : 0.0f;
    return VC_OK;
}

extern "C" int vc_conv2d_nhwc(vc_stream s, const vc_conv_desc *d)
{
    if (!d || !d->in.p || !d->out.p || !d->wpk || !d->bias) return VC_EINVAL;
    if (d->kh != d->kw) return VC_EINVAL;
    const int k = d->kh, st = d->stride;
    const int ck = vc_conv_chunk(d->cfg & 0xff, k, st, d->in.c);
    if (ck <= 0) return VC_EINVAL;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d->in.p; a.in_sn = d->in.sn; a.in_sh = d->in.sh; a.in_sw = d->in.sw;
    a.N = d->in.n; a.H = d->in.h; a.W = d->in.w; a.Cin = d->in.c;
    a.Ho = (d->in.h + 2 * (k / 2) - k) / st + 1;
    a.Wo = (d->in.w + 2 * (k / 2) - k) / st + 1;
    a.out = d->out.p; a.out_sn = d->out.sn; a.out_sh = d->out.sh; a.out_sw = d->out.sw;
    if (d->out_mode == VC_OUT_PIXELSHUFFLE2) {
        a.Cout = d->out.c * 4;
        if (d->out.h != 2 * a.Ho || d->out.w != 2 * a.Wo) return VC_EINVAL;
    } else {
        a.Cout = d->out.c;
        if (d->out.h != a.Ho || d->out.w != a.Wo) return VC_EINVAL;
    }
    if (d->out.n != d->in.n) return VC_EINVAL;
    if (d->epi != VC_EPI_NONE && !d->mul) return VC_EINVAL;
    if (d->epi != VC_EPI_NONE && d->act != VC_ACT_NONE) return VC_EINVAL;   // GDN/IGDN are never followed by an activation
    a.wpk = d->wpk; a.bias = d->bias;
    a.res = d->res; a.res_sn = d->res_sn; a.res_sh = d->res_sh; a.res_sw = d->res_sw;
    a.mul = d->mul; a.mul_sn = d->mul_sn; a.mul_sh = d->mul_sh; a.mul_sw = d->mul_sw;
    a.chscale = d->chscale;
    const bool f16 = (d->cfg & VC_CFG_F16) != 0;
    if (f16 && (!cfg_f16_ok(d->cfg & 0xff, a.Cin) || d->in_xform != VC_IN_NONE)) return VC_EINVAL;
    a.in_f16 = (d->cfg & VC_CFG_IN_F16) ? 1 : 0;
    a.out_f16 = (d->cfg & VC_CFG_OUT_F16) ? 1 : 0;
    // half-precision tensors exist on the fp16 path only -- one exception: the fp32 GDN / IGDN instance of the streaming 1x1 kernel
    // may STORE its result as half (the input of a residual block that runs on the fp16 path, identity included)
    if ((a.in_f16 || a.out_f16) && !f16 &&
        !(a.out_f16 && !a.in_f16 && (d->cfg & 0xff) == VC_CFG_PWS && (d->epi == VC_EPI_GDN || d->epi == VC_EPI_IGDN)))
        return VC_EINVAL;
    a.res_f16 = (d->cfg & VC_CFG_RES_F16) ? 1 : 0;
    a.pack128 = (d->cfg & VC_CFG_PACK128) ? 1 : 0;
    a.tail_wpk = d->tail_wpk;
    a.tail_bias = d->tail_bias;
    a.in_sp3 = (d->cfg & VC_CFG_IN_SP3) ? 1 : 0;
    a.out_sp3 = (d->cfg & VC_CFG_OUT_SP3) ? 1 : 0;
    a.res_sp3 = (d->cfg & VC_CFG_RES_SP3) ? 1 : 0;
    {
        // split tensors: read (input, residual) by the split pipeline only; WRITTEN by it and by the classic fp32 instances (the
        // layer in front of a split consumer: stride-2 / 1x1 / GDN layers), plain or pixel-shuffled, 16-byte epilogue accesses
        const int c0 = d->cfg & 0xff;
        const bool classic = c0 == VC_CFG_N128 || c0 == VC_CFG_N64 || c0 == VC_CFG_N32 || c0 == VC_CFG_N16 || c0 == VC_CFG_N128B || c0 == VC_CFG_N32T16;
        if ((a.in_sp3 || a.res_sp3) && c0 != VC_CFG_SPLIT) return VC_EINVAL;
        if (a.out_sp3 && c0 != VC_CFG_SPLIT && ((!classic && c0 != VC_CFG_PWS) || f16 || (d->out.c % 8))) return VC_EINVAL;
        if (a.out_sp3 && c0 != VC_CFG_SPLIT && !a.out_sn) a.out_sn = (long long)(d->out.c / 8) * d->out.h * d->out.w * 48;
    }
    if (a.tail_wpk && (!f16 || (d->cfg & 0xff) != VC_CFG_DMA)) return VC_EINVAL;             // the fused tail lives in the LDS-DMA kernel
    // a half-precision residual: the streaming 1x1 kernel, or the LDS-DMA kernel's fused-tail epilogue
    if (a.res_f16 && (!f16 || !d->res || !((d->cfg & 0xff) == VC_CFG_PWS || (d->cfg & 0xff) == VC_CFG_DMA))) return VC_EINVAL;
    a.res_first = (d->cfg & VC_CFG_RES_FIRST) ? 1 : 0;
    if (a.res_first && (!d->res || d->epi != VC_EPI_NONE || d->act == VC_ACT_SIGMOID || d->act == VC_ACT_CLAMP01)) return VC_EINVAL;
    a.cin_pad = round_up(a.Cin, f16 ? 2 * ck : ck);
    const int cfg0 = d->cfg & 0xff;
    const int th = (cfg0 == VC_CFG_N32T16) ? 16 : 8, tw = (cfg0 == VC_CFG_N4) ? 64 : 32;
    a.tiles_x = (a.Wo + tw - 1) / tw;
    a.tiles_y = (a.Ho + th - 1) / th;
    // Small feature maps (hyper-networks, MV codec, coarse pyramid levels): a 128-channel block would
    // leave most of the 256 CUs idle while each block walks the whole K loop alone.  The 32-wide MFMA
    // configurations share one packed-weight layout, so drop to a narrower channel block (more blocks,
    // shorter serial chain) until the launch can fill the chip twice over.
    int cfg = d->cfg & 0xff;
    const bool exact = (d->cfg & VC_CFG_EXACT) != 0;   // caller (autotuner) pinned the tile configuration
    while (!exact && (cfg == VC_CFG_N128 || cfg == VC_CFG_N64) &&
           (long long)a.tiles_x * a.tiles_y * a.N * (round_up(a.Cout, cfg_bn(cfg)) / cfg_bn(cfg)) < 512 &&
           vc_conv_chunk(cfg + 1, k, st, d->in.c) == ck)
        ++cfg;
    const int bn = cfg_bn(cfg);
    a.nblks = round_up(a.Cout, bn) / bn;
    a.total_blocks = a.tiles_x * a.tiles_y * a.nblks * a.N;
    {   // rows of tiles per band of the 2-D tile order (VC_TILE_BAND overrides for experiments; 1 = plain row-major)
        static const int band = [] { const char *e = getenv("VC_TILE_BAND"); const int v = e ? atoi(e) : 8; return v >= 1 ? v : 8; }();
        a.tile_band = band;
    }
    a.act = d->act; a.slope = d->slope;
    a.epi = d->epi; a.in_xform = d->in_xform; a.out_mode = d->out_mode;
    a.vec4 = ((a.Cin % 4) == 0 && (a.in_sw % 4) == 0 && (a.in_sh % 4) == 0 && (a.in_sn % 4) == 0 &&
              ((uintptr_t)a.in % 16) == 0) ? 1 : 0;
    if (a.in_f16)   // one 16-byte load = 8 halves
        a.vec4 = ((a.Cin % 8) == 0 && (a.in_sw % 8) == 0 && (a.in_sh % 8) == 0 && (a.in_sn % 8) == 0 &&
                  ((uintptr_t)a.in % 16) == 0) ? 1 : 0;
    {   // 16-byte epilogue accesses: every view that is touched must keep groups of 4 channels aligned
        auto ok = [](const void *ptr, long long sn, long long sh, long long sw) {
            return !ptr || (((uintptr_t)ptr % 16) == 0 && (sn % 4) == 0 && (sh % 4) == 0 && (sw % 4) == 0);
        };
        const int cgrp = (d->out_mode == VC_OUT_PIXELSHUFFLE2) ? (a.Cout >> 2) : a.Cout;   // channels per output pixel
        a.vec_out = (cgrp % 4) == 0 && (a.out_sp3 ? ((uintptr_t)a.out % 16) == 0 : ok(a.out, a.out_sn, a.out_sh, a.out_sw)) &&
                    (a.res_sp3 ? ((uintptr_t)a.res % 8) == 0 : ok(a.res, a.res_sn, a.res_sh, a.res_sw)) &&
                    ok(a.mul, a.mul_sn, a.mul_sh, a.mul_sw) && ok(a.chscale, 0, 0, 0);
    }
    if (a.total_blocks <= 0) return VC_EINVAL;
    if (a.out_sp3 && !a.vec_out) return VC_EINVAL;     // the split store is a 16-byte-group epilogue
    if (f16 && !a.vec4) return VC_EINVAL;   // the fp16 staging path reads 2 x 16 bytes per item
    hipStream_t stream = as_stream(s);
    if (cfg == VC_CFG_PW) {                 // streaming 1x1 kernel: only ever chosen explicitly (autotuner)
        if (!conv_pw_eligible(a, k, st, f16)) return VC_EINVAL;
        return conv_dispatch_pw(stream, a, f16);
    }
    if (cfg == VC_CFG_PWS) {                // LDS-DMA streaming 1x1 kernel: only ever chosen explicitly (autotuner)
        if (!conv_pws_eligible(a, k, st, f16)) return VC_EINVAL;
        return conv_dispatch_pws(stream, a, f16);
    }
    if (cfg == VC_CFG_SPLIT)                // split-operand fp32 pipeline: only ever chosen explicitly (its own packing and input format)
        return conv_dispatch_split(stream, a, k, st);
    if (cfg == VC_CFG_DMA)                  // LDS-DMA pipeline (fp16 path; fp32: the two big 7x7 layers): only ever chosen explicitly (autotuner)
        return conv_dispatch_dma(stream, a, k, st, f16);
    switch (k) {
    case 1: return f16 ? conv_dispatch_k1_f16(stream, a, st, cfg, ck) : conv_dispatch_k1_f32(stream, a, st, cfg, ck);
    case 3: return f16 ? conv_dispatch_k3_f16(stream, a, st, cfg, ck) : conv_dispatch_k3_f32(stream, a, st, cfg, ck);
    case 5: return f16 ? conv_dispatch_k5_f16(stream, a, st, cfg, ck) : conv_dispatch_k5_f32(stream, a, st, cfg, ck);
    case 7: return f16 ? conv_dispatch_k7_f16(stream, a, st, cfg, ck) : conv_dispatch_k7_f32(stream, a, st, cfg, ck);
    }
    return VC_EINVAL;
}

#ifdef VC_STAMPS
extern "C" int vc_debug_read_stamps(unsigned long long *out8)
{
    if (hipDeviceSynchronize() != hipSuccess) return VC_ELAUNCH;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_vc_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return VC_ELAUNCH;
    unsigned long long zero[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_vc_stamps), zero, sizeof(zero)) != hipSuccess) return VC_ELAUNCH;
    return VC_OK;
}
extern "C" int vc_debug_set_skip(int mask)
{
    if (hipDeviceSynchronize() != hipSuccess) return VC_ELAUNCH;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vc_skip), &mask, sizeof(mask)) == hipSuccess ? VC_OK : VC_ELAUNCH;
}
#endif
