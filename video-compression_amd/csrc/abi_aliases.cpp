// The operator names SURVEY.md section 8(b) gives for the C-ABI boundary, forwarded to the library's own entry points.
#include <string.h>
#include "vc_hip.h"

extern "C" int vc_gdn(vc_stream s, vc_view x, const float *gamma_packed, const float *beta_packed, int inverse, vc_view res, vc_view out)
{
    if (!x.p || !gamma_packed || !beta_packed || !out.p || x.c != out.c) return VC_EINVAL;
    vc_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.in = x;
    d.out = out;
    d.wpk = gamma_packed;
    d.bias = beta_packed;
    if (res.p) {
        d.res = res.p;
        d.res_sn = res.sn;
        d.res_sh = res.sh;
        d.res_sw = res.sw;
    }
    d.mul = x.p;
    d.mul_sn = x.sn;
    d.mul_sh = x.sh;
    d.mul_sw = x.sw;
    d.kh = d.kw = 1;
    d.stride = 1;
    d.act = VC_ACT_NONE;
    d.epi = inverse ? VC_EPI_IGDN : VC_EPI_GDN;
    d.in_xform = VC_IN_SQUARE;
    d.out_mode = VC_OUT_PLAIN;
    d.cfg = vc_conv_select_cfg(x.c, x.c, 1, 1);
    return vc_conv2d_nhwc(s, &d);
}

extern "C" int vc_spynet_level(vc_stream s, vc_view first, vc_view second, vc_view flow_coarse, vc_view feat8, vc_view up2)
{
    return vc_spynet_level_input(s, first, second, flow_coarse, feat8, up2);
}

extern "C" int vc_pool(vc_stream s, vc_view in, vc_view out, int k, float scale) { return vc_avgpool_reflectpad(s, in, out, k, scale); }

extern "C" int vc_upsample(vc_stream s, vc_view in, vc_view out, int factor, int align_corners, float scale)
{
    return vc_upsample_bilinear(s, in, out, factor, align_corners, scale);
}

extern "C" int vc_pad(vc_stream s, vc_view in, vc_view out) { return vc_avgpool_reflectpad(s, in, out, 1, 1.0f); }

extern "C" int vc_blend(vc_stream s, vc_view fwbw, vc_view mask, vc_view cur, vc_view pred, vc_view resid)
{
    return vc_lhbdc_blend(s, fwbw, mask, cur, pred, resid);
}

extern "C" int vc_factorized_bits(vc_stream s, vc_view z, const float *params, const float *in_gain, const float *out_gain,
                                  vc_view z_hat, int32_t *symbols, double *bits_partial, int bits_slots, float *likelihoods)
{
    return vc_eb_forward(s, z, params, in_gain, out_gain, z_hat, symbols, bits_partial, bits_slots, likelihoods);
}

extern "C" int vc_gaussian_symbols(vc_stream s, vc_view y, vc_view scales, vc_view means, const float *in_gain,
                                   const float *out_gain, vc_view y_hat, double *bits_partial, int bits_slots,
                                   const float *sym_src_p, int32_t *symbols, int32_t *indexes, const float *scale_table,
                                   int n_scales, float *likelihoods)
{
    return vc_gc_forward(s, y, scales, means, in_gain, out_gain, y_hat, bits_partial, bits_slots, sym_src_p, symbols, indexes,
                         scale_table, n_scales, likelihoods);
}
