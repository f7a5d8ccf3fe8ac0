/*
 * vc_hip.h -- C ABI of libvc_hip.so, the MI355X (gfx950) implementation of the per-B-frame codec
 * hot path  flow -> warp -> residual analysis -> hyperprior -> arithmetic code -> synthesis.
 *
 * The reference (KUIS-AI-Tekalp-Research-Group/video-compression) is pure Python on PyTorch +
 * CompressAI: it has NO FFI of its own.  Each entry point below therefore replaces a PyTorch /
 * CompressAI operator *call site* of the reference, cited as file:line under /root/reference.
 * The Python host layer (video-compression_amd/vcamd) binds these with ctypes and mirrors the
 * reference's module surface (m.Model, b_model.BidirFlowRef, encode_B, decode_B).
 *
 * Conventions
 *   - every device entry point takes a hipStream_t (passed as void*), enqueues asynchronously and
 *     returns 0 on success or a negative VC_E* code; no exceptions, no allocation, no sync inside
 *     (graph-capturable); all buffers are caller-owned device memory;
 *   - activations are fp32, channels-last ("NHWC"): a vc_view addresses element (n,y,x,c) at
 *     p[n*sn + y*sh + x*sw + c] (strides in floats), which lets callers express channel slices of a
 *     concat buffer and spatial crops without copies;
 *   - host entry points (vc_pmf_to_quantized_cdf, vc_rans_*, vc_conv_pack_*) touch host memory only.
 */
#ifndef VC_HIP_H
#define VC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_OK 0
#define VC_EINVAL (-1)   /* bad argument / unsupported shape */
#define VC_ELAUNCH (-2)  /* HIP launch failure */
#define VC_ENOMEM (-3)
#define VC_EDATA (-4)    /* corrupt bitstream / degenerate pmf */

typedef void *vc_stream; /* hipStream_t */

typedef struct {
    float *p;
    int n, h, w, c;
    long long sn, sh, sw; /* strides in floats; channel stride is 1 */
} vc_view;

/* ------------------------------------------------------------------------------------------
 * Library info
 * ---------------------------------------------------------------------------------------- */
const char *vc_version(void);
/* ABI number of this header: bumped whenever a struct below changes size or layout (vc_conv_desc grew trailing fields in
 * rounds 4 and 5).  A caller compiled against another number must not call the library (the Python binding checks it on
 * load); descriptors must be zero-initialised (memset) so that fields a caller does not know stay NULL / 0. */
#define VC_ABI_VERSION 6
int vc_abi_version(void);
/* Name of the code-object target compiled in ("gfx950"). */
const char *vc_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * Convolution engine (im2col-free implicit GEMM on fp32 MFMA, LDS-staged NHWC input tiles).
 * Replaces: torch.nn.Conv2d call sites  LHBDC/model/flow.py:52-62 (7x7), layers.py:202-209 (5x5/3x3),
 *           compressai conv3x3/conv1x1/subpel_conv3x3 inside layers.py:48-91,123-166 and
 *           Flex.../b_model/unet.py:27-31,62-74, layers.py:79-123;
 *           compressai GDN (1x1 contraction on x^2) -- SURVEY.md A.2.
 * ---------------------------------------------------------------------------------------- */
enum { VC_ACT_NONE = 0, VC_ACT_RELU = 1, VC_ACT_LRELU = 2, VC_ACT_SIGMOID = 3, VC_ACT_CLAMP01 = 4 /* Flex decompress clamp_(0,1) */ };
enum { VC_EPI_NONE = 0, VC_EPI_GDN = 1, VC_EPI_IGDN = 2 };   /* out = mul * rsqrt(acc) / mul * sqrt(acc) */
enum { VC_IN_NONE = 0, VC_IN_SQUARE = 1 };                   /* transform applied to the staged input */
enum { VC_OUT_PLAIN = 0, VC_OUT_PIXELSHUFFLE2 = 1 };         /* nn.PixelShuffle(2) fused into the store */

/* Tile configurations (output-channel block / MFMA shape).  Chosen by vc_conv_select_cfg. */
enum { VC_CFG_N128 = 0, VC_CFG_N64 = 1, VC_CFG_N32 = 2, VC_CFG_N16 = 3, VC_CFG_N4 = 4 /* 64 px x 4 ch, 4x4x1 MFMA */,
       VC_CFG_N128B = 5 /* 128 channels, waves 2x2 (3x3 stride-1 only; same packed weights as N128) */,
       VC_CFG_PW = 6 /* streaming 1x1 stride-1 kernel: weights resident in LDS, activations global -> registers with
                        next-tile prefetch; 32 <= cin <= 128, cout <= 128, plain/ReLU/LeakyReLU epilogue (+ gain,
                        residual); reads the packed weights of N128/N64/N32 and gives bit-identical results */,
       VC_CFG_N32T16 = 7 /* 7x7 stride 1: 16-row tiles of 32 channels (22 x 38 input footprint per 16 x 32 outputs
                        instead of 14 x 38 per 8 x 32); same packed weights as N32, bit-identical results */,
       VC_CFG_DMA = 8 /* fp16 path only (VC_CFG_F16 | VC_CFG_IN_F16), 3x3 / 7x7 stride 1, cin a multiple of 32, cout 32, 64
                        or a multiple of 128 -- with VC_CFG_PACK128 also any multiple of 4 from 96 up (plain output only:
                        blocks of 128 whose last one is partly padding) --, plain / ReLU / LeakyReLU / sigmoid / clamp epilogue (+ gain, residual, pixel
                        shuffle): one persistent 512-thread workgroup per CU, both operands streamed into LDS by
                        global_load_lds with counted vmcnt across raw barriers (csrc/conv_dma.h); reads the packed weights
                        of the 32-wide configurations and gives bit-identical results */,
       VC_CFG_PWS = 9 /* streaming 1x1 stride-1 kernel, second generation (csrc/conv_pws.hip): one persistent 512-thread
                        workgroup per CU, weights resident in LDS, activations streamed into per-wave LDS rings by
                        global_load_lds in whole 128-byte lines with counted vmcnt waits (no barriers), the residual
                        requested a tile ahead; cin 32/64/96/128, cout <= 128, plain/ReLU/LeakyReLU epilogue (+ gain,
                        residual), fp32 and both fp16-path input types; reads the packed weights of N128/N64/N32 and
                        gives bit-identical results */,
       VC_CFG_SPLIT = 10 /* fp32 layer on the bf16 matrix pipe with SPLIT operands (csrc/conv_split.h): every fp32 operand is the
                        exact sum of three bf16 pieces, all nine piece products (each exact in fp32) are accumulated in fp32 by
                        v_mfma_f32_16x16x32_bf16 -- 9/16 of the native fp32 matrix time; same precision class as the native
                        instances (errors against fp64 measured no larger), NOT bit-identical to them (summation order).
                        5x5 / 7x7 stride 1 (cin a multiple of 8, cout of 32; 7x7 also of 16) and 3x3 stride 1 (cin a multiple of 16, cout of 32),
                        plain / ReLU / LeakyReLU epilogue (+ gain, residual, pixel shuffle); `in` must be a split tensor (VC_CFG_IN_SP3), `wpk` from
                        vc_conv_pack_weights_split */ };
/* OR into vc_conv_desc.cfg to launch exactly that configuration (a narrower 32-wide configuration reads the
 * same packed weights and produces bit-identical results); without it the library narrows the block for
 * small feature maps by itself. */
#define VC_CFG_EXACT 0x100
/* OR into vc_conv_desc.cfg: `wpk` holds half-precision fragments from vc_conv_pack_weights_f16 and the layer runs on
 * v_mfma_f32_32x32x16_f16 (activations converted to half while staged, fp32 accumulate) -- the "fp16 MFMA conv
 * path" of BASELINE.json configs[4].  32-wide tile configurations with cin % 8 == 0 only; not bit-comparable with
 * the fp32 reference (judged on PSNR/bpp tolerance). */
#define VC_CFG_F16 0x200
/* With VC_CFG_F16 only: `in` (VC_CFG_IN_F16) / `out` (VC_CFG_OUT_F16) are HALF-precision tensors (pointer to
 * _Float16, strides counted in elements).  An fp16-path layer rounds its input to half while staging anyway, so an
 * activation whose only consumers are such layers can live in HBM as half without changing a single result bit:
 * the producer's epilogue performs the identical round-to-nearest conversion.  Halves the traffic of those tensors.
 * `res`, `mul`, `chscale` and `bias` stay fp32. */
#define VC_CFG_IN_F16 0x400
#define VC_CFG_OUT_F16 0x800
/* OR into vc_conv_desc.cfg: add `res` BEFORE the (plain / ReLU / LeakyReLU) activation instead of after it --
 * out = relu(conv(x) + res), the ResidualUnit of compressai.layers.AttentionBlock (ICIP2024/src/model/elic.py:97-121). */
#define VC_CFG_RES_FIRST 0x1000
/* With VC_CFG_F16, VC_CFG_PWS only: `res` is a HALF-precision tensor (pointer to _Float16, strides in elements) -- the
 * identity path of a chain of bottleneck blocks kept as half on the fp16 path (ICIP2024/src/model/elic.py:69-83: out =
 * x + f(x) per block).  Unlike VC_CFG_IN_F16 / OUT_F16 this CHANGES results (one more rounding of the identity per block):
 * part of the fp16 mode's stated tolerance, never of the fp32 path.  Every other configuration returns VC_EINVAL. */
#define VC_CFG_RES_F16 0x2000
/* With VC_CFG_DMA only: the caller states that `wpk` and `bias` were packed by vc_conv_pack_weights[_f16] with
 * cfg = VC_CFG_N128 (or N128B), i.e. zero-padded to whole blocks of 128 output channels.  Required for output-channel counts
 * above 64 that are no multiple of 128 (the kernel reads whole blocks); without it such a call returns VC_EINVAL instead of
 * reading past a narrower packing. */
#define VC_CFG_PACK128 0x4000
/* With VC_CFG_SPLIT: `in` (VC_CFG_IN_SP3, required) / `out` (VC_CFG_OUT_SP3, optional) are SPLIT tensors: dense
 * [n][c/8][h][w][3 pieces][8 channels] bf16 -- per pixel and group of 8 channels one 48-byte record hi | mid | lo with
 * hi + mid + lo == the fp32 value exactly.  The view's p is the buffer, n/h/w/c the logical shape, strides are ignored.
 * Produced by vc_split3 or by the VC_CFG_OUT_SP3 epilogue of the layer in front (a chain of split layers never converts). */
#define VC_CFG_IN_SP3 0x8000
#define VC_CFG_OUT_SP3 0x10000
/* VC_CFG_OUT_SP3 is also accepted by the classic fp32 configurations (N128 / N64 / N32 / N16 / N128B / N32T16, channels per output
 * pixel a multiple of 8): the layer in front of a split consumer writes the three pieces itself (stride-2 / 1x1 / GDN layers).
 * VC_CFG_RES_SP3 (VC_CFG_SPLIT only): `res` is a split tensor laid out like the output -- the identity of a residual block whose
 * input is a split tensor; its pieces sum to the exact fp32 value.  The image strides of split tensors (in.sn / out.sn / res_sn)
 * count BYTES, 0 = dense: a window of planes inside a wider split tensor passes its parent's image size. */
#define VC_CFG_RES_SP3 0x20000
typedef struct {
    vc_view in;            /* [n,h,w,cin] */
    vc_view out;           /* [n,ho,wo,cout]  (PIXELSHUFFLE2: [n,2ho,2wo,cout/4]) */
    const float *wpk;      /* packed weights from vc_conv_pack_weights (device) */
    const float *bias;     /* packed bias, vc_conv_packed_bias_floats() entries (device) */
    const float *res;      /* optional residual added after the activation, laid out like `out` */
    long long res_sn, res_sh, res_sw;
    const float *mul;      /* GDN/IGDN: the un-squared input, laid out like `out` */
    long long mul_sn, mul_sh, mul_sw;
    const float *chscale;  /* optional per-output-channel gain applied before the residual add */
    int kh, kw, stride;    /* square "same" padding kh/2, kw/2 */
    int act; float slope;
    int epi, in_xform, out_mode;
    int cfg;
    /* Optional fused 1x1 layer behind a VC_CFG_DMA 3x3 layer (fp16 path; ICIP2024/src/model/elic.py:69-83: the tail of a
     * ResidualBottleneckBlock: ... -> conv3x3 -> ReLU -> conv1x1, + identity).  With tail_wpk set the launch computes
     *   out = W_tail . act(conv3x3(in)) + b_tail (+ res)
     * -- `act` / `slope` belong to the 3x3 layer (none / ReLU / LeakyReLU with 0 <= slope <= 1), its result is rounded to half
     * exactly as the unfused layer would store it and never leaves the CU; the residual (fp32, or half with VC_CFG_RES_F16)
     * is added behind the 1x1 layer.  cin = cout = 128 or 64, plain output, no channel gain.  tail_wpk / tail_bias come from
     * vc_conv_pack_tail_f16 (device copies).  Anything else returns VC_EINVAL.  NULL = no fused layer. */
    const void *tail_wpk;
    const float *tail_bias;
} vc_conv_desc;

int vc_conv_select_cfg(int cout, int cin, int k, int stride);
/* channel chunk (K-granule) of a (cfg,k,stride) instance; weights are zero-padded to a multiple of it */
int vc_conv_chunk(int cfg, int k, int stride, int cin);
size_t vc_conv_packed_weight_floats(int cfg, int cout, int cin, int kh, int kw, int stride);
size_t vc_conv_packed_bias_floats(int cfg, int cout);
/* Host: OIHW fp32 -> fragment-ordered packed weights/bias.  `pixelshuffle` applies the
 * (c*4+dy*2+dx) -> ((dy*2+dx)*cout/4 + c) output-channel permutation the fused store expects. */
int vc_conv_pack_weights(const float *w_oihw, const float *bias, int cout, int cin, int kh, int kw,
                         int stride, int cfg, int pixelshuffle, float *wpk_out, float *bias_out);
size_t vc_conv_packed_weight_bytes_f16(int cfg, int cout, int cin, int kh, int kw, int stride); /* 0 = not eligible */
/* 1x1 weights [cout][cin] (fp32, host) -> the fused-tail fragments of vc_conv_desc.tail_wpk (cout * cin halves) and its bias
 * (cout floats); cout = cin = 128 or 64. */
int vc_conv_pack_tail_f16(const float *w, const float *bias, int cout, int cin, void *wpk_half_out, float *bias_out);
int vc_conv_pack_weights_f16(const float *w_oihw, const float *bias, int cout, int cin, int kh, int kw,
                             int stride, int cfg, int pixelshuffle, void *wpk_half_out, float *bias_out);
/* Split-operand path: packed weights = bf16 piece fragments [n-block][chunk of 8 channels][tap unit][piece][n-tile][lane][8]
 * (bytes incl. the slack the last DMA round may over-read; 0 = shape not served); bias_out: cout floats. */
size_t vc_conv_packed_weight_bytes_split(int cout, int cin, int k);
int vc_conv_pack_weights_split(const float *w_oihw, const float *bias, int cout, int cin, int k, int pixelshuffle, void *wpk_out,
                               float *bias_out);
/* fp32 channels-last window (c % 8 == 0, 16-byte aligned rows) -> dense split tensor, 6 bytes per element */
int vc_split3(vc_stream s, vc_view in, void *out_split, long long out_image_bytes /* 0 = dense */);
/* The same for a window whose channel count is no multiple of 8 (or of the layer's chunk): c_out (multiple of 8, >= in.c) channels
 * are written, those past in.c as zeros -- the input of a split layer packed with zero weights for the padding channels (first layers
 * with 6 input channels: LHBDC/model/layers.py:202, Flex-Rate.../b_model/unet.py:43). */
int vc_split3_pad(vc_stream s, vc_view in, void *out_split, long long out_image_bytes /* 0 = dense */, int c_out);
int vc_conv2d_nhwc(vc_stream s, const vc_conv_desc *d);

/* ------------------------------------------------------------------------------------------
 * Layout conversion at the module boundary (reference tensors are NCHW: m.py:32, b_model.py:49)
 * ---------------------------------------------------------------------------------------- */
int vc_nchw_to_nhwc(vc_stream s, const float *src_nchw, vc_view dst);
int vc_nhwc_to_nchw(vc_stream s, vc_view src, float *dst_nchw);
/* Frame ingest / output of the test loop and the CLI scripts.
 * vc_u8hwc_to_f32nchw_pad: 8-bit RGB [h][w][3] (a decoded PNG, device memory) -> fp32 NCHW [3][hp][wp] = x/255 with
 *   nn.ReflectionPad2d((0, wp-w, 0, hp-h)): `normalize` + `pad` of LHBDC/encode_B.py:39-64, test/utils.py:190-203.
 * vc_f32nchw_to_u8hwc: the top-left h x w window of an fp32 NCHW [3][hp][wp] frame -> 8-bit RGB [h][w][3] with
 *   np.round(np.clip(x,0,1)*255): `float_to_uint8` + crop of LHBDC/decode_B.py:35-38,122-123. */
int vc_u8hwc_to_f32nchw_pad(vc_stream s, const uint8_t *src_hwc, int h, int w, float *dst_nchw, int hp, int wp);
int vc_f32nchw_to_u8hwc(vc_stream s, const float *src_nchw, int hp, int wp, uint8_t *dst_hwc, int h, int w);

/* ------------------------------------------------------------------------------------------
 * Resampling / pooling / padding
 * ---------------------------------------------------------------------------------------- */
/* F.avg_pool2d(x*scale, k) followed by ReflectionPad2d bottom/right to out.h x out.w
 * (flow.py:86-87 with k=2; m.py:38-50 with k=4, scale .5 or 1, pad to x64). */
int vc_avgpool_reflectpad(vc_stream s, vc_view in, vc_view out, int k, float scale);
/* nn.MaxPool2d(2,2) (layers.py:200) */
int vc_maxpool2(vc_stream s, vc_view in, vc_view out);
/* The same on a split tensor ([n][c/8][h][w][3][8] bf16, see VC_CFG_SPLIT; image strides in bytes, 0 = dense), split result:
 * between the split-operand encoder layers of the mask U-Net. */
/* F.avg_pool2d(x * scale, 2) of a split tensor (no padding: even h, w), result a split tensor (out_split) or an fp32 view (out_f32):
 * exactly one of the two. */
int vc_avgpool2_sp3(vc_stream s, const void *in_split, long long in_image_bytes, int n, int h, int w, int c, float scale, void *out_split,
                    long long out_image_bytes, vc_view out_f32);
int vc_maxpool2_sp3(vc_stream s, const void *in_split, long long in_image_bytes, int n, int h, int w, int c, void *out_split,
                    long long out_image_bytes);
/* F.interpolate / nn.Upsample bilinear by an integer factor (layers.py:232,238,244; m.py:30;
 * unet.py:63), out = scale * bilinear(in); out.h/out.w = factor * in.h/in.w. */
int vc_upsample_bilinear(vc_stream s, vc_view in, vc_view out, int factor, int align_corners, float scale);
/* The same with a split tensor as the result (the up-sampled part of a concat buffer a split-operand convolution reads). */
int vc_upsample_bilinear_sp3(vc_stream s, vc_view in, void *out_split, long long out_image_bytes, int factor, int align_corners, float scale);
/* out = alpha*a + beta*b (b may be NULL-pointer view with p==0) -- m.py:52,56-59,71 */
int vc_axpby(vc_stream s, vc_view a, vc_view b, vc_view out, float alpha, float beta);
/* out = clamp(a, 0, 1) on views: a decoded frame before it serves as a reference (ICIP2024/src/test.py:94, src/utils.py:194) */
int vc_clamp01(vc_stream s, vc_view a, vc_view out);
/* out[...,c] = gain[c] * a[...,c] -- Flex Gain_Module.forward (Flex.../b_model/layers.py:54-73) */
int vc_channel_scale(vc_stream s, vc_view a, const float *gain, vc_view out);

/* ------------------------------------------------------------------------------------------
 * Warping (torch grid_sample call sites)
 *   VC_WARP_W1: LHBDC flow.py:15-25, m.py:111-126 -- sample at (x+u*W/(W-1), y+v*H/(H-1)), border clamp
 *   VC_WARP_W2: Flex b_model.py:99-112           -- sample at (x+u-.5, y+v-.5), zeros outside
 *   VC_WARP_W3: ICIP2024 src/model/m.py:262-282  -- align_corners=True: sample exactly at (x+u, y+v), border clamp
 * ---------------------------------------------------------------------------------------- */
enum { VC_WARP_W1 = 1, VC_WARP_W2 = 2, VC_WARP_W3 = 3 };
int vc_warp(vc_stream s, int convention, vc_view img, vc_view flow, vc_view out);

/* SPyNet pre-processing (flow.py:39-45): NCHW frame -> normalised, channel-flipped NHWC level-0 image */
int vc_spynet_preprocess(vc_stream s, const float *src_nchw, vc_view dst);
/* One pyramid level's network input (flow.py:93-98): up = 2*bilinear_x2(flow_coarse, align_corners=True)
 * (replicate-padded when the level is odd-sized; zeros when flow_coarse.p==NULL), feat = [first,
 * warp_W1(second, up), up] (8 ch), and `up` alone (2 ch) for the residual add after the last conv. */
int vc_spynet_level_input(vc_stream s, vc_view first, vc_view second, vc_view flow_coarse,
                          vc_view feat8, vc_view up2);
/* The same with the 8-channel level input written as a dense split tensor (one 48-byte record per pixel, see VC_CFG_SPLIT): the
 * first 7x7 layer then reads it as it lies. */
int vc_spynet_level_input_sp3(vc_stream s, vc_view first, vc_view second, vc_view flow_coarse, void *feat_split, vc_view up);

/* LHBDC mask blend + residual (m.py:63-67): pred = m*fw + (1-m)*bw ; resid = cur - pred.
 * fwbw holds fw in channels 0..2 and bw in 3..5; mask has 1 channel. resid.p may be NULL. */
int vc_lhbdc_blend(vc_stream s, vc_view fwbw, vc_view mask, vc_view cur, vc_view pred, vc_view resid);
/* Flex blend (b_model.py:68-73): mask has 2 channels (already through sigmoid),
 * pred = (.5m0*xb + .5m1*xa)/(.5m0+.5m1+1e-8) ; resid = cur - pred. */
int vc_flex_blend(vc_stream s, vc_view xb, vc_view xa, vc_view mask, vc_view cur, vc_view pred, vc_view resid);
/* Flex linear-motion split (b_model.py:38-40): flow4=[F01,F10] -> ft0, ft1 (2 ch each) */
int vc_flex_motion_split(vc_stream s, vc_view flow4, vc_view ft0, vc_view ft1, float t);

/* ------------------------------------------------------------------------------------------
 * ICIP2024 flow-guided deformable compensation
 * ---------------------------------------------------------------------------------------- */
/* Quantise-and-mask used by the checkerboard / channel-context entropy model
 * (ICIP2024/src/model/compression_bottlenecks.py:237-246,268-269):
 *   out = keep(y,x) ? (do_round ? round_half_even(in) : in) * (gain ? gain[c] : 1) : 0
 * keep_parity: -1 keeps every position, 0 keeps (y+x) even, 1 keeps (y+x) odd ("anchors":
 * y_half[:, :, 0::2, 0::2] = 0; y_half[:, :, 1::2, 1::2] = 0  <=>  keep_parity 1;
 * ctx_params[:, :, 0::2, 1::2] = 0; ctx_params[:, :, 1::2, 0::2] = 0  <=>  keep_parity 0).  in may alias out. */
int vc_quantize_mask(vc_stream s, vc_view in, vc_view out, const float *gain, int keep_parity, int do_round);

/* torchvision.ops.deform_conv2d, modulated (DCNv2), kernel 3x3, stride 1, padding 1, dilation 1, as called at
 * ICIP2024/src/model/helpers.py:56 (DeformConv2d(2C, C, 3, padding=1, groups=16)):
 *   in [n,h,w,G*cg], offset [n,h,w,G*9*2] (per group and tap: dy then dx), mask [n,h,w,G*9] (p==0: all ones),
 *   out [n,h,w,G*og]; conv groups == offset groups == G, cg in {4,8,12,16}, og <= 8.
 * wpk: weights re-laid out by vc_deform_pack_weights ([G][9][cg][og] floats), bias [G*og] or NULL. */
int vc_deform_pack_weights(const float *w_oihw, int cout, int cin_per_group, int groups, float *dst);
int vc_deform_conv2d(vc_stream s, vc_view in, vc_view offset, vc_view mask, const float *wpk, const float *bias,
                     int groups, vc_view out);
/* OffsetDiversity.forward (helpers.py:43-58) in one launch: for each reference r in {1,2}
 *   o1, o2, m = chunk(raw_r, 3); offset_r = tanh(cat(o1,o2))*magnitude + flow_r.flip(1).repeat(..); mask_r = sigmoid(m)
 * then deform_conv2d(cat(x1,x2), cat(offset1,offset2), cat(mask1,mask2)) with groups 0..G/2-1 reading x1 and
 * the rest x2.  raw_r [n,h,w,27*G/2], flow_r [n,h,w,2] = (u,v), x_r [n,h,w,(G/2)*cg], out [n,h,w,G*og]. */
int vc_offset_diversity(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2, vc_view flow2,
                        float magnitude, const float *wpk, const float *bias, int groups, vc_view out);
/* fp16 path: the same with HALF-precision features -- x1.p / x2.p point at _Float16 tensors (strides in elements), as
 * vc_to_half writes them.  The deformable fusion (helpers.py:35-58) is bound by its gathers; half features halve them.
 * 8 or 16 channels per group; offsets, modulation, bilinear weights and accumulation stay fp32.  One image of the features
 * stays below 4 GiB (the gathers use 32-bit byte offsets from the image's base; VC_EINVAL above -- the fp32 entry points have
 * no such limit). */
int vc_offset_diversity_hx(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2, vc_view flow2,
                        float magnitude, const float *wpk, const float *bias, int groups, vc_view out);
/* Dense half-precision copy [n,h,w,c] of a channels-last window (c % 4 == 0), round to nearest even. */
int vc_to_half(vc_stream s, vc_view a, void *out_half);
/* The same on GROUP-PLANAR half features, the layout the fp16 path uses (same values, same arithmetic, same result bit for bit;
 * the gathers of neighbouring pixels then share cache lines).  vc_to_half_planar writes [n][c/cg][h][w][cg] halves (cg in
 * {4, 8, 12, 16}); x1 / x2 of vc_offset_diversity_hxp describe ONE group's plane of such a tensor: c = cg, sw = cg, sh = w * cg,
 * sn = (G/2) * h * w * cg (strides in elements).  One image of the features stays below 4 GiB (32-bit byte offsets). */
int vc_offset_diversity_hxp(vc_stream s, vc_view x1, vc_view raw1, vc_view flow1, vc_view x2, vc_view raw2, vc_view flow2,
                            float magnitude, const float *wpk, const float *bias, int groups, vc_view out);
int vc_to_half_planar(vc_stream s, vc_view a, int cg, void *out_half);

/* Gate of compressai.layers.AttentionBlock (ELIC intra codec of ICIP2024, src/model/elic.py:97-121):
 * out = a * sigmoid(b) + identity. */
int vc_attention_gate(vc_stream s, vc_view a, vc_view b, vc_view identity, vc_view out);

/* Motion-adaptive flow resolution (ICIP2024/src/opt_helpers.py:41-51) without leaving the device:
 * vc_sse_clamp01 writes workgroup partial sums of (clamp(pred,0,1) - cur)^2 (fold them with vc_bits_reduce);
 * vc_select_flow evaluates psnr_i = 10*log10(1/(sse[i]/n_elems)) in fp32 and copies the candidate with the
 * first strictly greatest positive PSNR (the reference's `if psnr > best` loop) into `out`, its index into
 * *choice (candidate 0 when no PSNR is positive).  count <= 8; all candidates have out's shape. */
int vc_sse_clamp01(vc_stream s, vc_view pred, vc_view cur, double *sse_partial, int slots);
int vc_select_flow(vc_stream s, const double *sse, int count, double n_elems, const vc_view *candidates, vc_view out,
                   int32_t *choice);

/* ------------------------------------------------------------------------------------------
 * Entropy models (CompressAI EntropyBottleneck / GaussianConditional)
 * ---------------------------------------------------------------------------------------- */
/* Factorised prior parameters for C channels, pre-resolved on the host at load time:
 *   per channel 60 floats: softplus(M0)[3], b0[3], tanh(f0)[3], softplus(M1)[9], b1[3], tanh(f1)[3],
 *   softplus(M2)[9], b2[3], tanh(f2)[3], softplus(M3)[9], b3[3], tanh(f3)[3], softplus(M4)[3], b4[1],
 *   median[1], pad[1]  (matrices row-major [out][in]) */
#define VC_EB_PARAMS_PER_CHANNEL 60
/* EntropyBottleneck.forward (eval): z_hat = round(z*gain - med) + med, likelihood through the
 * logistic-CDF MLP, lower-bounded 1e-9; bits += sum(-log2 p) into bits_partial (see vc_bits_reduce).
 * `in_gain` (nullable, per channel) is Flex's hyper_gain_unit, `out_gain` (nullable) its
 * hyper_inv_gain_unit applied to the stored z_hat (layers.py:139-141).
 * symbols (nullable, int32 NCHW order n,c,y,x) receives round(z*gain - med) for the range coder.
 * likelihoods (nullable, fp32, same dense NCHW order) receives the per-element likelihood tensor the reference's
 * callers read as out["likelihoods"]["z"] (LHBDC/model/m.py:73-91, layers.py:72-91). */
int vc_eb_forward(vc_stream s, vc_view z, const float *params, const float *in_gain, const float *out_gain,
                  vc_view z_hat, int32_t *symbols, double *bits_partial, int bits_slots, float *likelihoods);
/* inverse for the decoder: z_hat = (sym + med) * out_gain */
int vc_eb_dequant(vc_stream s, const int32_t *symbols, const float *params, const float *out_gain, vc_view z_hat);

/* GaussianConditional.forward (eval) on y with (scales, means) = chunk(h_s output, 2):
 *   y_hat = (round(y*gain - mu) + mu) * out_gain ; p = Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), s>=0.11, p>=1e-9.
 * sym_src (nullable) lets Flex's compress() quantise the UN-gained y (layers.py:167): symbols are
 * round(sym_src - mu) when given, else round(y*gain - mu).  indexes (nullable, needs scale_table) =
 * build_indexes(scales) against scale_table[n_scales].  likelihoods (nullable, fp32, dense NCHW order like
 * symbols) receives out["likelihoods"]["y"]. */
int vc_gc_forward(vc_stream s, vc_view y, vc_view scales, vc_view means, const float *in_gain,
                  const float *out_gain, vc_view y_hat, double *bits_partial, int bits_slots,
                  const float *sym_src_p, int32_t *symbols, int32_t *indexes, const float *scale_table,
                  int n_scales, float *likelihoods);
/* decoder side: indexes from scales; y_hat = (sym + mu) * out_gain */
int vc_gc_indexes(vc_stream s, vc_view scales, const float *scale_table, int n_scales, int32_t *indexes);
/* Scale refinement: for the elements of `scales` (the first N output channels of the hyper-synthesis transform's last layer,
 * conv3x3(cin -> 2N, stride 1, padding 1; LHBDC/model/layers.py:82-91) whose fp32 value lies within rel_eps (relative) of an
 * entry of the scale table, recompute that layer's output from its input `in` and its ORIGINAL weights [2N][cin][3][3] / bias in
 * fp64 and store the correctly rounded fp32 value -- the scale-table index then no longer depends on this platform's summation
 * order (the CompressAI format carries no indexes: INTEGRATION.md, cross-platform limits).  counter (optional, device): +1 per
 * refined element. */
int vc_refine_scales(vc_stream s, vc_view scales, vc_view in, const float *w_oihw, const float *bias, const float *scale_table,
                     int n_scales, float rel_eps, int *counter);
int vc_gc_dequant(vc_stream s, const int32_t *symbols, vc_view means, const float *out_gain, vc_view y_hat);
/* Symbol refinement (ABI 6): the twin of vc_refine_scales for the coded integers.  `vc_refine_layer` names a convolution
 * (k x k, k odd <= 7, stride 1 or 2, padding k / 2, nn.Conv2d weights [cout][cin][k][k] as stored in the checkpoint, nullable
 * bias) whose output channels c0 .. c0 + C - 1 are the tensor under refinement, and `in` its channels-last fp32 input.
 *   vc_refine_y_symbols: elements of y (the analysis transform's last layer, LHBDC/model/layers.py:48-56) with
 *     | frac(y - mu) - 1/2 | <= eps: y from `y_layer` and mu from `mu_layer` (the hyper-synthesis transform's last layer,
 *     c0 = M: the means half, layers.py:70-80) in fp64, rounded once to fp32; symbols[n,c,y,x] = round(y - mu) of THOSE values
 *     (GaussianConditional.compress, layers.py:103).  y / means in memory stay as they are (the decoder re-derives the same mu
 *     without knowing which elements were refined); y_hat (p nullable) receives (symbol + means) * out_gain for the refined
 *     elements -- the closed-loop encoder's reconstruction stays what the decoder rebuilds.
 *   vc_refine_z_symbols: elements of z (the hyper-analysis transform's last layer, layers.py:58-68) with
 *     | frac(z * in_gain - median) - 1/2 | <= eps (eb_params: the layout of vc_eb_forward): z from `z_layer` in fp64;
 *     symbols = round(z * in_gain - median), z_hat (p nullable) = (symbol + median) * out_gain (EntropyBottleneck.compress /
 *     decompress, layers.py:97-98).  Call it BEFORE the hyper-synthesis transform reads z_hat.
 * counter (optional, device): +1 per refined element.  0 < eps <= 1e-2. */
typedef struct vc_refine_layer {
    vc_view in;
    const float *w_oihw;
    const float *bias;
    int k, stride, c0;
} vc_refine_layer;
int vc_refine_y_symbols(vc_stream s, vc_view y, vc_refine_layer y_layer, vc_view means, vc_refine_layer mu_layer, float eps,
                        int32_t *symbols, vc_view y_hat, const float *out_gain, int *counter);
int vc_refine_z_symbols(vc_stream s, vc_view z, vc_refine_layer z_layer, const float *eb_params, const float *in_gain, float eps,
                        int32_t *symbols, vc_view z_hat, const float *out_gain, int *counter);
/* Deterministic two-stage reduction of the per-workgroup partial sums written by the kernels above:
 * out[i] = sum_j partial[i*slots + j], i < count. */
int vc_bits_reduce(vc_stream s, const double *partial, int slots, int count, double *out);
/* number of partial-sum slots each entropy launch uses (size bits_partial accordingly) */
int vc_bits_slots(void);
/* PSNR as the evaluation loops measure it (LHBDC/test/testing.py:176-182, test/utils.py:32-51): clamp both CHW fp32 images
 * (row pitch W, plane pitch H*W) to [0,1], x255, round half to even, mean squared error over the [:h,:w] crop of all
 * channels in double, 10*log10(255^2/mse) -> *psnr_out (device).  scratch: vc_bits_slots() doubles (device). */
int vc_psnr_uint8(vc_stream s, const float *a_chw, const float *b_chw, int channels, int H, int W, int h, int w,
                  double *scratch, int slots, double *psnr_out);

/* ------------------------------------------------------------------------------------------
 * Range coder (host).  Replaces compressai._CXX.pmf_to_quantized_cdf and
 * compressai.ans.RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes as called from
 * EntropyModel.compress/decompress (reached from LHBDC/model/layers.py:97-98,103,108,112).
 * cdfs: dense int32 [n_tables][cdf_stride]; cdf_sizes / offsets: [n_tables].  Every index is checked against
 * n_tables and every cdf_size against cdf_stride (VC_EINVAL) -- a corrupt index never reads outside the tables.
 * ---------------------------------------------------------------------------------------- */
int vc_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf_out /* n+1 */);
/* returns bytes written (>=0) or a negative VC_E* code; out_cap >= 4*(count*? ) -- use vc_rans_bound */
size_t vc_rans_bound(size_t count);
long long vc_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, size_t count,
                                      const int32_t *cdfs, int n_tables, int cdf_stride, const int32_t *cdf_sizes,
                                      const int32_t *offsets, uint8_t *out, size_t out_cap);
int vc_rans_decode_with_indexes(const uint8_t *data, size_t nbytes, const int32_t *indexes, size_t count,
                                const int32_t *cdfs, int n_tables, int cdf_stride, const int32_t *cdf_sizes,
                                const int32_t *offsets, int32_t *symbols_out);
/* compressai.ans.RansDecoder.set_stream + decode_stream (ICIP2024/src/model/elic.py:428-429,566,584: the checkerboard
 * codec decodes one string in two calls, the second call's indexes depending on the first call's symbols).
 * state[2] = {coder state, next 32-bit word}; {0, 0} starts a stream; updated in place. */
int vc_rans_decode_stream(const uint8_t *data, size_t nbytes, uint64_t *state, const int32_t *indexes, size_t count,
                          const int32_t *cdfs, int n_tables, int cdf_stride, const int32_t *cdf_sizes,
                          const int32_t *offsets, int32_t *symbols_out);

/* ------------------------------------------------------------------------------------------
 * Entry points under the operator names SURVEY.md section 8(b) lists for the boundary.  Thin forwards to the
 * functions above (same kernels, same argument rules); a binding may use either spelling.
 * ---------------------------------------------------------------------------------------- */
/* compressai.layers.GDN(.inverse): y = x * rsqrt(beta + gamma @ x^2) (inverse: * sqrt), optional residual added to the
 * result.  gamma_packed / beta_packed: the resolved (non-negative) gamma [C,C,1,1] and beta [C] packed with
 * vc_conv_pack_weights for cfg = vc_conv_select_cfg(C, C, 1, 1). */
int vc_gdn(vc_stream s, vc_view x, const float *gamma_packed, const float *beta_packed, int inverse, vc_view res /* p may be NULL */,
           vc_view out);
/* = vc_spynet_level_input (flow.py:93-98) */
int vc_spynet_level(vc_stream s, vc_view first, vc_view second, vc_view flow_coarse, vc_view feat8, vc_view up2);
/* = vc_avgpool_reflectpad / vc_upsample_bilinear / reflection padding alone (k = 1) / vc_lhbdc_blend */
int vc_pool(vc_stream s, vc_view in, vc_view out, int k, float scale);
int vc_upsample(vc_stream s, vc_view in, vc_view out, int factor, int align_corners, float scale);
int vc_pad(vc_stream s, vc_view in, vc_view out);
int vc_blend(vc_stream s, vc_view fwbw, vc_view mask, vc_view cur, vc_view pred, vc_view resid);
/* = vc_eb_forward / vc_gc_forward */
int vc_factorized_bits(vc_stream s, vc_view z, const float *params, const float *in_gain, const float *out_gain,
                       vc_view z_hat, int32_t *symbols, double *bits_partial, int bits_slots, float *likelihoods);
int vc_gaussian_symbols(vc_stream s, vc_view y, vc_view scales, vc_view means, const float *in_gain,
                        const float *out_gain, vc_view y_hat, double *bits_partial, int bits_slots,
                        const float *sym_src_p, int32_t *symbols, int32_t *indexes, const float *scale_table,
                        int n_scales, float *likelihoods);

#ifdef __cplusplus
}
#endif
#endif /* VC_HIP_H */
