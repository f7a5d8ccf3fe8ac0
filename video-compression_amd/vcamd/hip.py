"""ctypes binding of libvc_hip.so (C ABI declared in include/vc_hip.h).

This is the only door to the compute path: there is NO CPU fallback.  If the library is missing the
import of any model module fails loudly with instructions to build it.
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# VC_HIP_LIB lets a developer A/B two builds of the same ABI on one device; it must still be a HIP build.
LIB_PATH = os.environ.get("VC_HIP_LIB") or os.path.join(_PKG_DIR, "libvc_hip.so")

VC_OK = 0
ABI_VERSION = 6          # VC_ABI_VERSION of include/vc_hip.h (struct layouts below)
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_CLAMP01 = 0, 1, 2, 3, 4
EPI_NONE, EPI_GDN, EPI_IGDN = 0, 1, 2
IN_NONE, IN_SQUARE = 0, 1
OUT_PLAIN, OUT_PIXELSHUFFLE2 = 0, 1
WARP_W1, WARP_W2, WARP_W3 = 1, 2, 3
EB_PARAMS_PER_CHANNEL = 60

_ERR = {-1: "VC_EINVAL (bad argument / unsupported shape)", -2: "VC_ELAUNCH (HIP launch failure)",
        -3: "VC_ENOMEM", -4: "VC_EDATA (corrupt bitstream / degenerate pmf)"}


class VcError(RuntimeError):
    pass


class View(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("n", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int),
                ("c", ctypes.c_int), ("sn", ctypes.c_longlong), ("sh", ctypes.c_longlong),
                ("sw", ctypes.c_longlong)]


class ConvDesc(ctypes.Structure):
    _fields_ = [("inp", View), ("out", View), ("wpk", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("res", ctypes.c_void_p), ("res_sn", ctypes.c_longlong), ("res_sh", ctypes.c_longlong),
                ("res_sw", ctypes.c_longlong),
                ("mul", ctypes.c_void_p), ("mul_sn", ctypes.c_longlong), ("mul_sh", ctypes.c_longlong),
                ("mul_sw", ctypes.c_longlong),
                ("chscale", ctypes.c_void_p),
                ("kh", ctypes.c_int), ("kw", ctypes.c_int), ("stride", ctypes.c_int),
                ("act", ctypes.c_int), ("slope", ctypes.c_float),
                ("epi", ctypes.c_int), ("in_xform", ctypes.c_int), ("out_mode", ctypes.c_int),
                ("cfg", ctypes.c_int),
                ("tail_wpk", ctypes.c_void_p), ("tail_bias", ctypes.c_void_p)]      # fused 1x1 tail (VC_CFG_DMA), include/vc_hip.h


class RefineLayer(ctypes.Structure):
    """vc_refine_layer of include/vc_hip.h: the convolution whose output is recomputed in fp64 for boundary-case symbols"""
    _fields_ = [("inp", View), ("w_oihw", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("k", ctypes.c_int), ("stride", ctypes.c_int),
                ("c0", ctypes.c_int)]


_lib = None


class KernelTimer:
    """Opt-in per-launch timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  Used by bench.py for the roofline line; never active in production."""

    def __init__(self):
        self.items = []

    def bracket(self, key, flops, launch, nbytes=0.0):
        """``flops`` / ``nbytes``: algorithmic work of the launch (bytes = each operand read once + the result
        written once, in the dtypes actually stored)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = launch()
        e1.record()
        self.items.append((key, flops, e0, e1, nbytes))
        return out

    def table(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, e0, e1, nbytes in self.items:
            a = agg.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


timer = None  # set to a KernelTimer() to collect
split_keys = set()   # (while a KernelTimer collects) the timing keys of launches that ran on the split-operand pipeline
kernel_symbols = {}  # timing key -> a prefix of the kernel symbol the launch ran (what a rocprofv3 trace names it)


def timed_hbm(key, nbytes, launch):
    """Run ``launch`` -- bracketed by HIP events under the key "hbm <key>" while a KernelTimer collects (bench.py's
    ``hbm_kernels`` block: achieved GB/s of the memory-bound kernels against the HBM peak); ``nbytes`` = algorithmic
    traffic (every operand read once, the result written once)."""
    if timer is None:
        return launch()
    return timer.bracket("hbm " + key, 0.0, launch, nbytes)


def lib():
    """Load libvc_hip.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VcError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C video-compression_amd/csrc`.")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, cll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
    sz = ctypes.c_size_t

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("vc_version", ctypes.c_char_p)
    sig("vc_abi_version", ci)
    if L.vc_abi_version() != ABI_VERSION:
        raise VcError(f"{LIB_PATH} has ABI {L.vc_abi_version()}, this binding was written against {ABI_VERSION} (include/vc_hip.h: "
                      "VC_ABI_VERSION): rebuild with `make -C video-compression_amd/csrc`")
    sig("vc_target_arch", ctypes.c_char_p)
    sig("vc_conv_select_cfg", ci, ci, ci, ci, ci)
    sig("vc_conv_chunk", ci, ci, ci, ci, ci)
    sig("vc_conv_packed_weight_floats", sz, ci, ci, ci, ci, ci, ci)
    sig("vc_conv_packed_bias_floats", sz, ci, ci)
    sig("vc_conv_pack_weights", ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp)
    sig("vc_conv_packed_weight_bytes_f16", sz, ci, ci, ci, ci, ci, ci)
    sig("vc_conv_pack_weights_f16", ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp)
    sig("vc_conv_pack_tail_f16", ci, vp, vp, ci, ci, vp, vp)
    sig("vc_conv2d_nhwc", ci, vp, ctypes.POINTER(ConvDesc))
    sig("vc_conv_packed_weight_bytes_split", sz, ci, ci, ci)
    sig("vc_conv_pack_weights_split", ci, vp, vp, ci, ci, ci, ci, vp, vp)
    sig("vc_split3", ci, vp, View, vp, cll)
    sig("vc_split3_pad", ci, vp, View, vp, cll, ci)
    sig("vc_nchw_to_nhwc", ci, vp, vp, View)
    sig("vc_nhwc_to_nchw", ci, vp, View, vp)
    sig("vc_u8hwc_to_f32nchw_pad", ci, vp, vp, ci, ci, vp, ci, ci)
    sig("vc_f32nchw_to_u8hwc", ci, vp, vp, ci, ci, vp, ci, ci)
    sig("vc_avgpool_reflectpad", ci, vp, View, View, ci, cf)
    sig("vc_maxpool2", ci, vp, View, View)
    sig("vc_maxpool2_sp3", ci, vp, vp, cll, ci, ci, ci, ci, vp, cll)
    sig("vc_avgpool2_sp3", ci, vp, vp, cll, ci, ci, ci, ci, cf, vp, cll, View)
    sig("vc_upsample_bilinear_sp3", ci, vp, View, vp, cll, ci, ci, cf)
    sig("vc_upsample_bilinear", ci, vp, View, View, ci, ci, cf)
    sig("vc_axpby", ci, vp, View, View, View, cf, cf)
    sig("vc_clamp01", ci, vp, View, View)
    sig("vc_channel_scale", ci, vp, View, vp, View)
    sig("vc_warp", ci, vp, ci, View, View, View)
    sig("vc_spynet_preprocess", ci, vp, vp, View)
    sig("vc_spynet_level_input", ci, vp, View, View, View, View, View)
    sig("vc_spynet_level_input_sp3", ci, vp, View, View, View, vp, View)
    sig("vc_lhbdc_blend", ci, vp, View, View, View, View, View)
    sig("vc_flex_blend", ci, vp, View, View, View, View, View, View)
    sig("vc_flex_motion_split", ci, vp, View, View, View, cf)
    sig("vc_quantize_mask", ci, vp, View, View, vp, ci, ci)
    sig("vc_deform_pack_weights", ci, vp, ci, ci, ci, vp)
    sig("vc_deform_conv2d", ci, vp, View, View, View, vp, vp, ci, View)
    sig("vc_offset_diversity", ci, vp, View, View, View, View, View, View, cf, vp, vp, ci, View)
    sig("vc_offset_diversity_hx", ci, vp, View, View, View, View, View, View, cf, vp, vp, ci, View)
    sig("vc_to_half", ci, vp, View, vp)
    sig("vc_offset_diversity_hxp", ci, vp, View, View, View, View, View, View, cf, vp, vp, ci, View)
    sig("vc_to_half_planar", ci, vp, View, ci, vp)
    sig("vc_attention_gate", ci, vp, View, View, View, View)
    sig("vc_sse_clamp01", ci, vp, View, View, vp, ci)
    sig("vc_select_flow", ci, vp, vp, ci, ctypes.c_double, ctypes.POINTER(View), View, vp)
    sig("vc_eb_forward", ci, vp, View, vp, vp, vp, View, vp, vp, ci, vp)
    sig("vc_eb_dequant", ci, vp, vp, vp, vp, View)
    sig("vc_gc_forward", ci, vp, View, View, View, vp, vp, View, vp, ci, vp, vp, vp, vp, ci, vp)
    sig("vc_gc_indexes", ci, vp, View, vp, ci, vp)
    sig("vc_refine_scales", ci, vp, View, View, vp, vp, vp, ci, cf, vp)
    sig("vc_gc_dequant", ci, vp, vp, View, vp, View)
    sig("vc_refine_y_symbols", ci, vp, View, RefineLayer, View, RefineLayer, cf, vp, View, vp, vp)
    sig("vc_refine_z_symbols", ci, vp, View, RefineLayer, vp, vp, cf, vp, View, vp, vp)
    sig("vc_bits_reduce", ci, vp, vp, ci, ci, vp)
    sig("vc_bits_slots", ci)
    sig("vc_psnr_uint8", ci, vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp)
    sig("vc_gdn", ci, vp, View, vp, vp, ci, View, View)
    sig("vc_pad", ci, vp, View, View)
    sig("vc_pmf_to_quantized_cdf", ci, vp, ci, ci, vp)
    sig("vc_rans_bound", sz, sz)
    sig("vc_rans_encode_with_indexes", cll, vp, vp, sz, vp, ci, ci, vp, vp, vp, sz)
    sig("vc_rans_decode_with_indexes", ci, vp, sz, vp, sz, vp, ci, ci, vp, vp, vp)
    sig("vc_rans_decode_stream", ci, vp, sz, vp, vp, sz, vp, ci, ci, vp, vp, vp)
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "vc_version", "vc_abi_version", "vc_target_arch", "vc_conv_select_cfg", "vc_conv_chunk", "vc_conv_packed_weight_floats",
    "vc_conv_packed_bias_floats", "vc_conv_pack_weights", "vc_conv_packed_weight_bytes_f16",
    "vc_conv_pack_weights_f16", "vc_conv_pack_tail_f16", "vc_conv2d_nhwc", "vc_conv_packed_weight_bytes_split",
    "vc_conv_pack_weights_split", "vc_split3", "vc_split3_pad", "vc_nchw_to_nhwc",
    "vc_nhwc_to_nchw", "vc_u8hwc_to_f32nchw_pad", "vc_f32nchw_to_u8hwc", "vc_avgpool_reflectpad", "vc_maxpool2", "vc_maxpool2_sp3", "vc_avgpool2_sp3", "vc_upsample_bilinear", "vc_upsample_bilinear_sp3", "vc_axpby", "vc_clamp01", "vc_channel_scale", "vc_warp",
    "vc_spynet_preprocess", "vc_spynet_level_input", "vc_spynet_level_input_sp3", "vc_lhbdc_blend", "vc_flex_blend",
    "vc_flex_motion_split", "vc_quantize_mask", "vc_deform_pack_weights", "vc_deform_conv2d", "vc_offset_diversity", "vc_offset_diversity_hx", "vc_to_half", "vc_offset_diversity_hxp", "vc_to_half_planar",
    "vc_attention_gate", "vc_sse_clamp01", "vc_select_flow", "vc_eb_forward", "vc_eb_dequant", "vc_gc_forward", "vc_gc_indexes", "vc_refine_scales",
    "vc_refine_y_symbols", "vc_refine_z_symbols", "vc_gc_dequant", "vc_bits_reduce", "vc_bits_slots", "vc_psnr_uint8", "vc_pmf_to_quantized_cdf", "vc_rans_bound",
    "vc_rans_encode_with_indexes", "vc_rans_decode_with_indexes", "vc_rans_decode_stream",
    # the operator spellings of SURVEY.md 8(b), thin forwards (csrc/abi_aliases.cpp)
    "vc_gdn", "vc_spynet_level", "vc_pool", "vc_upsample", "vc_pad", "vc_blend", "vc_factorized_bits", "vc_gaussian_symbols",
]


DEBUG_SYNC = bool(int(os.environ.get("VC_DEBUG_SYNC", "0")))   # synchronise after every launch to localise a fault


def check(rc, what):
    if rc != VC_OK:
        raise VcError(f"{what} failed: {_ERR.get(rc, rc)}")
    if DEBUG_SYNC:
        try:
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            raise VcError(f"device fault during {what}: {e}") from e


def stream():
    """The HIP stream torch is currently enqueuing on (so torch allocations and our kernels order)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------
# device tensors: channels-last fp32 buffers addressed through views
# ------------------------------------------------------------------------------------------------
class T:
    """A channels-last window into a flat device buffer: shape (n,h,w,c), strides in ELEMENTS, element offset.
    ``dtype`` is "f32" everywhere except for activations that only feed fp16-path convolutions ("f16", see
    VC_CFG_IN_F16 / VC_CFG_OUT_F16): those are produced and consumed by vc_conv2d_nhwc alone."""
    __slots__ = ("buf", "n", "h", "w", "c", "sn", "sh", "sw", "off", "dtype")

    def __init__(self, buf, n, h, w, c, sn, sh, sw, off=0, dtype="f32"):
        self.buf, self.n, self.h, self.w, self.c = buf, n, h, w, c
        self.sn, self.sh, self.sw, self.off, self.dtype = sn, sh, sw, off, dtype

    @staticmethod
    def empty(n, h, w, c, device, dtype="f32"):
        if dtype == "sp3":     # split tensor of the split-operand fp32 path: dense [n][c/8][h][w][3][8] bf16 (csrc/conv_split.h)
            if c % 8:
                raise VcError("a split tensor holds groups of 8 channels")
            # sn = planes of 8 channels per image of the underlying buffer, off = first plane of this window
            return T(torch.empty(n * h * w * c * 3, dtype=torch.int16, device=device), n, h, w, c, c // 8, 0, 0, 0, dtype)
        buf = torch.empty(n * h * w * c, dtype=torch.float16 if dtype == "f16" else torch.float32, device=device)
        return T(buf, n, h, w, c, h * w * c, w * c, c, 0, dtype)

    def channels(self, c0, c1):
        if self.dtype == "sp3":          # a window of planes: the split form of a channel slice of a concat buffer
            if c0 % 8 or c1 % 8:
                raise VcError("a split tensor is sliced in groups of 8 channels")
            return T(self.buf, self.n, self.h, self.w, c1 - c0, self.sn, 0, 0, self.off + c0 // 8, self.dtype)
        return T(self.buf, self.n, self.h, self.w, c1 - c0, self.sn, self.sh, self.sw, self.off + c0, self.dtype)

    def crop(self, h, w):
        if self.dtype == "sp3":
            raise VcError("a split tensor is plane-major: it has no crop window")
        return T(self.buf, self.n, h, w, self.c, self.sn, self.sh, self.sw, self.off, self.dtype)

    def images(self, n0, n1):
        if self.dtype == "sp3":
            return T(self.buf, n1 - n0, self.h, self.w, self.c, self.sn, 0, 0, self.off + n0 * self.sn, self.dtype)
        return T(self.buf, n1 - n0, self.h, self.w, self.c, self.sn, self.sh, self.sw, self.off + n0 * self.sn, self.dtype)

    @property
    def ptr(self):
        if self.dtype == "sp3":
            return self.buf.data_ptr() + self.off * self.h * self.w * 48
        return self.buf.data_ptr() + (2 if self.dtype == "f16" else 4) * self.off

    def view(self, allow_half=False):
        if self.dtype == "sp3":
            raise VcError("split tensors are private to the convolution engine (split_view)")
        if self.dtype != "f32" and not allow_half:
            raise VcError("half-precision activations are private to the fp16 convolution path")
        return View(self.ptr, self.n, self.h, self.w, self.c, self.sn, self.sh, self.sw)

    @property
    def image_bytes(self):
        """split tensors: distance between images in bytes"""
        return self.sn * self.h * self.w * 48

    def split_view(self):
        return View(self.ptr, self.n, self.h, self.w, self.c, self.image_bytes, 0, 0)

    def to_nchw(self):
        """Debug/inspection helper (torch indexing, not on the hot path)."""
        if self.dtype == "sp3":
            raise VcError("to_nchw reads fp32 / half windows (a split tensor holds three bf16 pieces per value)")
        full = self.buf[self.off:]
        return torch.as_strided(full, (self.n, self.c, self.h, self.w), (self.sn, 1, self.sh, self.sw)).contiguous().float()


NULL_VIEW = View(None, 0, 0, 0, 0, 0, 0, 0)


def nchw_to_nhwc(x):
    x = x.contiguous().float()
    n, c, h, w = x.shape
    out = T.empty(n, h, w, c, x.device)
    check(lib().vc_nchw_to_nhwc(stream(), x.data_ptr(), out.view()), "vc_nchw_to_nhwc")
    return out


def nhwc_to_nchw(t):
    out = torch.empty((t.n, t.c, t.h, t.w), dtype=torch.float32, device=t.buf.device)
    check(lib().vc_nhwc_to_nchw(stream(), t.view(), out.data_ptr()), "vc_nhwc_to_nchw")
    return out


def frame_from_uint8(u8_hwc, hp=None, wp=None, out=None):
    """uint8 RGB [h,w,3] CUDA tensor -> padded fp32 NCHW [1,3,hp,wp] (x/255, reflection pad to multiples of 64):
    ``process_frame`` of LHBDC/encode_B.py:58-64 without a torch operator."""
    h, w, c = u8_hwc.shape
    if c != 3 or u8_hwc.dtype != torch.uint8 or not u8_hwc.is_cuda or not u8_hwc.is_contiguous():
        raise VcError("frame_from_uint8 takes a contiguous uint8 [h,w,3] CUDA tensor")
    hp = hp or h + (64 - h % 64) % 64
    wp = wp or w + (64 - w % 64) % 64
    if out is None:
        out = torch.empty((1, 3, hp, wp), dtype=torch.float32, device=u8_hwc.device)
    check(lib().vc_u8hwc_to_f32nchw_pad(stream(), u8_hwc.data_ptr(), h, w, out.data_ptr(), hp, wp), "vc_u8hwc_to_f32nchw_pad")
    return out


def frame_to_uint8(x_nchw, h, w):
    """fp32 NCHW [1,3,hp,wp] CUDA frame -> uint8 RGB [h,w,3] of its top-left window (``float_to_uint8`` + crop of
    LHBDC/decode_B.py:35-38,122-123)."""
    _, c, hp, wp = x_nchw.shape
    if c != 3 or x_nchw.dtype != torch.float32 or not x_nchw.is_cuda:
        raise VcError("frame_to_uint8 takes an fp32 [1,3,hp,wp] CUDA tensor")
    x_nchw = x_nchw.contiguous()
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=x_nchw.device)
    check(lib().vc_f32nchw_to_u8hwc(stream(), x_nchw.data_ptr(), hp, wp, out.data_ptr(), h, w), "vc_f32nchw_to_u8hwc")
    return out


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
CFG_EXACT = 0x100
CFG_F16 = 0x200
CFG_IN_F16 = 0x400
CFG_OUT_F16 = 0x800
CFG_RES_FIRST = 0x1000
CFG_RES_F16 = 0x2000        # fp16 path, VC_CFG_PWS only: `res` is a half-precision tensor (the identity path of a bottleneck chain)
CFG_PACK128 = 0x4000        # with CFG_DMA: weights / bias packed with the 128-channel configuration (padded to blocks of 128)
CFG_SPLIT = 10            # fp32 on the bf16 matrix pipe with split operands (csrc/conv_split.h); input: a split tensor (dtype "sp3")
CFG_IN_SP3 = 0x8000
CFG_OUT_SP3 = 0x10000
CFG_RES_SP3 = 0x20000
# "native": v_mfma_f32_* instances everywhere (rounds 1-4).  "split" (default since round 5, after the reference-parity tests
# came out equal on it: tests/test_reference_1080p_gpu.py): the layers the split-operand pipeline serves (5x5 / 7x7
# stride 1, cin % 8 == 0, cout % 32 == 0) run on it -- exact bf16 x 3 pieces, nine exact products, fp32 accumulate; same precision
# class, different summation order.  VC_FP32_MODE / set_fp32_mode().
_FP32_MODE = os.environ.get("VC_FP32_MODE", "split")
CFG_DMA = 8               # fp16 path: LDS-DMA pipeline, one persistent workgroup per CU (csrc/conv_dma.h); half-precision input only
CFG_PWS = 9               # streaming 1x1 kernel with LDS-DMA activation rings (csrc/conv_pws.hip)
AUTOTUNE = bool(int(os.environ.get("VC_AUTOTUNE", "1")))
# "fp32" (default, exact fp32 FMA chains like the reference) or "fp16" (BASELINE.json configs[4]: half-precision MFMA
# with fp32 accumulate for every eligible layer; judged on PSNR/bpp tolerance, never the headline number).
_PRECISION = os.environ.get("VC_CONV_PRECISION", "fp32")


# fp16 path only: keep activations whose sole consumers are fp16-path convolutions as half in HBM (bit-identical
# results, see VC_CFG_OUT_F16); VC_HALF_ACTIVATIONS=0 stores them as fp32 like every other tensor (A/B, tests).
HALF_ACTIVATIONS = bool(int(os.environ.get("VC_HALF_ACTIVATIONS", "1")))
# fp16 path only: keep the identity path of a chain of bottleneck blocks (x + f(x) per block, ICIP2024/src/model/elic.py:69-83)
# as half in HBM too.  NOT bit-neutral (one more rounding of the identity per block; VC_CFG_RES_F16): part of the fp16 mode's
# stated tolerance.  VC_HALF_RESIDUAL=0 keeps the identity fp32 (A/B, tests).
HALF_RESIDUAL = bool(int(os.environ.get("VC_HALF_RESIDUAL", "1")))
# fp16 path: the trailing 1x1 layer of a bottleneck block (ICIP2024/src/model/elic.py:69-83) rides in the epilogue of the block's
# 3x3 layer (LDS-DMA kernel, vc_conv_desc.tail_wpk): the 3x3 layer's output never reaches HBM.  VC_FUSE_TAIL=0 = three launches.
FUSE_TAIL = bool(int(os.environ.get("VC_FUSE_TAIL", "1")))
# fp16 path only: the deformable fusion of ICIP2024 gathers from HALF-precision copies of its feature maps (8 / 16 channels per
# group: one 16-byte gather per corner instead of two; the kernel is bound by its gathers).  Offsets, modulation, bilinear
# weights and accumulation stay fp32.  VC_HALF_DEFORM=0 gathers fp32 features (A/B, tests).
HALF_DEFORM = bool(int(os.environ.get("VC_HALF_DEFORM", "1")))
HALF_DEFORM_PLANAR = bool(int(os.environ.get("VC_HALF_DEFORM_PLANAR", "1")))   # group-planar half features under the deformable gathers
# Bitstream paths (compress / decompress): scales within SCALE_REFINE_EPS (relative) of a scale-table entry are recomputed in fp64
# from the last hyper-synthesis layer's input (vc_refine_scales): this side's table indexes stop depending on its summation order.
# VC_SCALE_REFINE=0 keeps the plain fp32 scales (A/B, tests).
SCALE_REFINE = bool(int(os.environ.get("VC_SCALE_REFINE", "1")))
SCALE_REFINE_EPS = 2e-5
# Encoder side of the bitstream paths: latents whose difference with their centre (mu / the channel's median) lies within
# SYMBOL_REFINE_EPS of a half-integer are recomputed in fp64 from the inputs of the layers that produced them
# (vc_refine_y_symbols / vc_refine_z_symbols): the coded integer stops depending on this side's summation order in those layers.
# VC_SYMBOL_REFINE=0 keeps the plain fp32 rounding (A/B, tests).
SYMBOL_REFINE = bool(int(os.environ.get("VC_SYMBOL_REFINE", "1")))
SYMBOL_REFINE_EPS = 2e-5


def set_conv_precision(mode):
    """Applies to PackedConv objects created afterwards (models pack lazily: set it before the first forward, or
    call ``model.float()`` to drop the caches)."""
    global _PRECISION
    if mode not in ("fp32", "fp16"):
        raise ValueError("precision must be 'fp32' or 'fp16'")
    _PRECISION = mode


def conv_precision():
    return _PRECISION


def set_fp32_mode(mode):
    """"native" or "split" (see CFG_SPLIT); applies to calls made afterwards (the split packing is made on first use)."""
    global _FP32_MODE
    if mode not in ("native", "split"):
        raise ValueError("fp32 mode must be 'native' or 'split'")
    _FP32_MODE = mode


def fp32_mode():
    return _FP32_MODE


# The hyper-synthesis transform of the BITSTREAM paths always runs on this pipeline, whatever VC_FP32_MODE says: the scale-table
# indexes a decoder derives from it are part of the stream's meaning (the CompressAI format carries none), so a stream written
# under one fp32 mode must decode under the other.  (The transform is five small layers: its cost is nothing.)
BITSTREAM_HS_MODE = "native"


@contextlib.contextmanager
def fp32_mode_pinned(mode):
    global _FP32_MODE
    keep = _FP32_MODE
    _FP32_MODE = mode
    try:
        yield
    finally:
        _FP32_MODE = keep


def wants_split_at(pc, n, h, w):
    return _FP32_MODE == "split" and pc is not None and pc.split_ok and pc.cin_split == pc.cin and pc.split_pays(n, h, w)


def wants_split(pc, x, h=None, w=None):
    """Will the layer ``pc`` run on the split-operand pipeline for an input of x's batch at h x w (default: x's own size)?  The hint
    a producer needs to leave its result as a split tensor (``out_sp3``)."""
    return (_FP32_MODE == "split" and pc is not None and pc.split_ok and pc.cin_split == pc.cin      # (a producer cannot write padding channels)
            and pc.split_pays(x.n, x.h if h is None else h, x.w if w is None else w))


def split3(x, out=None, c_out=None):
    """fp32 channels-last window -> split tensor (three bf16 pieces per value, exact): the input format of CFG_SPLIT layers.
    ``c_out`` (a multiple of 8, >= x.c): pad with zero channels (a layer whose weights are packed with zero rows for them)."""
    if c_out is not None and c_out != x.c or x.c % 8:
        c_out = c_out or (x.c + 7) // 8 * 8
        if out is None:
            out = T.empty(x.n, x.h, x.w, c_out, x.buf.device, "sp3")
        timed_hbm(f"k_split3_pad c{x.c}->{c_out} @{x.n}x{x.h}x{x.w}", x.n * x.h * x.w * (4.0 * x.c + 6.0 * c_out),
                  lambda: check(lib().vc_split3_pad(stream(), x.view(), out.ptr, out.image_bytes, c_out), "vc_split3_pad"))
        return out
    if out is None:
        out = T.empty(x.n, x.h, x.w, x.c, x.buf.device, "sp3")
    timed_hbm(f"k_split3 c{x.c} @{x.n}x{x.h}x{x.w}", 10.0 * x.n * x.h * x.w * x.c,
              lambda: check(lib().vc_split3(stream(), x.view(), out.ptr, out.image_bytes), "vc_split3"))
    return out


class PackedConv:
    """Weights of one nn.Conv2d re-laid out for the MFMA kernel (done once per model load).

    The three 32-wide tile configurations (128/64/32 output channels per workgroup) read the same packed
    weights and give bit-identical results; which one is fastest depends on how the launch quantises over
    256 CUs (tail effects), so the first eager call per input shape times the candidates on the device and
    keeps the winner (never during graph capture)."""

    def __init__(self, weight, bias, stride=1, pixelshuffle=False, device=None):
        L = lib()
        w = weight.detach().to("cpu", torch.float32).contiguous()
        cout, cin, kh, kw = w.shape
        self.cout, self.cin, self.k, self.stride, self.ps = cout, cin, kh, stride, bool(pixelshuffle)
        self.cfg = L.vc_conv_select_cfg(cout, cin, kh, stride)
        nw = L.vc_conv_packed_weight_floats(self.cfg, cout, cin, kh, kw, stride)
        nb = L.vc_conv_packed_bias_floats(self.cfg, cout)
        if nw == 0:
            raise VcError(f"unsupported convolution {cout}x{cin}x{kh}x{kw} stride {stride}")
        wpk = np.empty(nw, dtype=np.float32)
        bpk = np.empty(nb, dtype=np.float32)
        wnp = w.numpy()
        bnp = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous().numpy()
        check(L.vc_conv_pack_weights(wnp.ctypes.data, None if bnp is None else bnp.ctypes.data, cout, cin, kh, kw,
                                     stride, self.cfg, int(self.ps), wpk.ctypes.data, bpk.ctypes.data),
              "vc_conv_pack_weights")
        self.wpk = torch.from_numpy(wpk).to(device)
        self.bias = torch.from_numpy(bpk).to(device)
        self.wpk16 = None
        if _PRECISION == "fp16":
            nbytes = L.vc_conv_packed_weight_bytes_f16(self.cfg, cout, cin, kh, kw, stride)
            if nbytes:
                w16 = np.empty(nbytes // 2, dtype=np.float16)
                check(L.vc_conv_pack_weights_f16(wnp.ctypes.data, None if bnp is None else bnp.ctypes.data, cout, cin, kh,
                                                 kw, stride, self.cfg, int(self.ps), w16.ctypes.data, bpk.ctypes.data),
                      "vc_conv_pack_weights_f16")
                self.wpk16 = torch.from_numpy(w16).to(device)
        self.tuned = {}
        self._tail = None
        self._wsplit = None
        # input channels are padded to the chunk of the split instance (8 for 5x5 / 7x7, 16 for 3x3) with zero weights: first layers
        # with 6 channels (LHBDC mask U-Net, Flex-Rate U-Net) run on the split pipeline behind vc_split3_pad
        chunk = 16 if kh == 3 else 8
        self.cin_split = (cin + chunk - 1) // chunk * chunk
        split_shape = stride == 1 and ((kh in (5, 7) and (cout % 32 == 0 or (kh == 7 and cout % 16 == 0)) and not pixelshuffle) or
                                       (kh == 3 and cout % 32 == 0))
        if split_shape and self.cin_split != cin:
            wpad = np.zeros((cout, self.cin_split, kh, kw), dtype=np.float32)
            wpad[:, :cin] = wnp.reshape(cout, cin, kh, kw)
            self._raw32 = (wpad, bnp)
        else:
            self._raw32 = (wnp, bnp) if split_shape else None
        self._raw = (wnp, bnp) if (_PRECISION == "fp16" and kh == 1 and stride == 1 and cout == cin and cin in (64, 128) and not pixelshuffle) else None
        self._device = device
        ck = L.vc_conv_chunk(self.cfg, kh, stride, cin)
        self.candidates = [c for c in range(self.cfg, 3) if L.vc_conv_chunk(c, kh, stride, cin) == ck] if self.cfg <= 2 else []
        if self.cfg == 0 and kh == 3 and stride == 1:
            self.candidates.append(5)          # VC_CFG_N128B: 128-channel block with the waves arranged 2x2
        if self.candidates and kh == 1 and stride == 1 and not self.ps and 32 <= cin <= 128 and cout <= 128 and cin % 8 == 0:
            self.candidates.append(6)          # VC_CFG_PW: streaming 1x1 kernel (skipped by the tuner when the call is not eligible)
            if cin % 32 == 0 and cout % 32 == 0 and os.environ.get("VC_PWS_KERNELS", "1") != "0":
                self.candidates.append(CFG_PWS)  # VC_CFG_PWS: the same through per-wave LDS-DMA rings
        if self.candidates and kh == 7 and stride == 1:
            self.candidates.append(7)          # VC_CFG_N32T16: 16-row tiles (less halo per output)
        # VC_CFG_DMA (fp16 path, half-precision input; the library refuses shapes it has no instance for): 32 / 64 output channels,
        # blocks of 128, or -- plain output on the N128 packing, which pads weights and bias to 128s -- a partly padded last block
        dma_cout = cout in (32, 64) or cout % 128 == 0 or (cout >= 96 and cout % 4 == 0 and self.cfg == 0 and not self.ps)
        if (self.candidates and kh in (3, 7) and stride == 1 and cin % 32 == 0 and dma_cout
                and os.environ.get("VC_DMA_KERNELS", "1") != "0"):
            self.candidates.append(CFG_DMA)
        # the exact fp32 instances of the same pipeline (csrc/conv_dma.h, DmaCfg::F32): SPyNet's two big 7x7 layers
        # (7x7: 64 -> 32, 32 -> 64; 3x3: 64 / 128 / 256 input channels to 64 or a multiple of 128 output channels, pixel shuffle
        #  included -- the library declines what it has no instance for and the tuner skips it)
        f32_shape = ((kh == 7 and (cin, cout) in ((64, 32), (32, 64)) and not self.ps) or
                     (kh == 3 and cin in (64, 128, 256) and (cout == 64 or cout % 128 == 0) and self.cfg in (0, 1)) or
                     (kh == 3 and (cin, cout) == (32, 64) and not self.ps) or
                     (kh == 5 and (cin, cout) in ((96, 32), (192, 64), (32, 64)) and not self.ps))
        self.dma_f32 = (bool(self.candidates) and stride == 1 and f32_shape
                        and os.environ.get("VC_DMA_KERNELS", "1") != "0" and os.environ.get("VC_DMA_F32", "1") != "0")
        if self.dma_f32 and CFG_DMA not in self.candidates:
            self.candidates.append(CFG_DMA)
        # every alternative must read THIS packing: same channel chunk (the zero padding of cin depends on it)
        self.candidates = [c for c in self.candidates if L.vc_conv_chunk(c, kh, stride, cin) == ck]

    def _pick_cfg(self, d, key, flags=0):
        if key in self.tuned:
            return self.tuned[key]
        if flags & CFG_RES_F16:                       # a half-precision residual: the streaming 1x1 kernel / the LDS-DMA 3x3 kernel
            return (CFG_PWS if self.k == 1 else CFG_DMA) | CFG_EXACT | flags
        cands = self.candidates
        if flags & CFG_F16 and 5 in cands:
            cands = [c for c in cands if c != 0]     # the 4x1 128-channel fp16 instance spills registers
        if not (flags & CFG_F16 and flags & CFG_IN_F16) and not (self.dma_f32 and not flags & CFG_F16):
            cands = [c for c in cands if c != CFG_DMA]   # the LDS-DMA pipeline copies pixels as they are: half tensors, or fp32 on its fp32 instances
        if flags & CFG_OUT_SP3:
            cands = [c for c in cands if c in (0, 1, 2, 3, 5, 7, CFG_PWS)]    # split output: the classic instances' and the streaming 1x1 kernel's epilogue
        if not AUTOTUNE or len(cands) < 2 or torch.cuda.is_current_stream_capturing():
            return (cands[0] if cands else self.cfg) | flags
        best, best_ms = self.cfg, float("inf")
        for c in cands:
            d.cfg = c | CFG_EXACT | flags
            if lib().vc_conv2d_nhwc(stream(), ctypes.byref(d)) != VC_OK:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                lib().vc_conv2d_nhwc(stream(), ctypes.byref(d))
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            if ms < best_ms:
                best, best_ms = c, ms
        self.tuned[key] = best | CFG_EXACT | flags
        return self.tuned[key]

    @property
    def split_ok(self):
        """True when the split-operand fp32 pipeline (CFG_SPLIT) serves this layer."""
        return self._raw32 is not None and self.wpk16 is None

    def split_pays(self, n, h, w):
        """The split pipeline walks 16 x 32-pixel tiles with one workgroup per CU: on coarse pyramid levels (few tiles) the native
        instances with their 8-row tiles and several workgroups per CU are faster.  Decided per IMAGE, never by the batch: a
        frame must get the same bits whether it is coded alone or in a level-batched pass."""
        bn = 64 if self.cout % 64 == 0 else (32 if self.cout % 32 == 0 else 16)
        th = 12 if self.k == 3 else (24 if bn == 16 else 16)
        return ((h + th - 1) // th) * ((w + 31) // 32) * (self.cout // bn) >= 48

    def split_pack(self):
        if self._wsplit is None:
            if self._raw32 is None:
                raise VcError("no split-operand instance for this layer")
            wnp, bnp = self._raw32
            nbytes = lib().vc_conv_packed_weight_bytes_split(self.cout, self.cin_split, self.k)
            if not nbytes:
                raise VcError("no split-operand instance for this layer")
            w = np.zeros(nbytes // 2, dtype=np.int16)
            b = np.empty(self.cout, dtype=np.float32)
            check(lib().vc_conv_pack_weights_split(wnp.ctypes.data, None if bnp is None else bnp.ctypes.data, self.cout, self.cin_split, self.k,
                                                   int(self.ps), w.ctypes.data, b.ctypes.data), "vc_conv_pack_weights_split")
            self._wsplit = (torch.from_numpy(w).to(self._device), torch.from_numpy(b).to(self._device))
        return self._wsplit

    def tail_pack(self):
        """This 1x1 C -> C layer (C = 64 or 128) as the fused tail of a 3x3 layer (vc_conv_pack_tail_f16): (weights, bias) on the device."""
        if self._tail is None:
            if self._raw is None:
                raise VcError("only a 1x1 64 -> 64 / 128 -> 128 layer of the fp16 path can be fused behind a 3x3 layer")
            wnp, bnp = self._raw
            c = self.cout
            w16 = np.empty(c * c, dtype=np.float16)
            b = np.empty(c, dtype=np.float32)
            check(lib().vc_conv_pack_tail_f16(np.ascontiguousarray(wnp.reshape(c, c)).ctypes.data, None if bnp is None else bnp.ctypes.data,
                                              c, c, w16.ctypes.data, b.ctypes.data), "vc_conv_pack_tail_f16")
            self._tail = (torch.from_numpy(w16).to(self._device), torch.from_numpy(b).to(self._device))
        return self._tail

    def can_fuse_tail(self, tail):
        """True when ``tail`` (a 1x1 PackedConv) can ride in this 3x3 layer's epilogue: fp16 path, C -> C -> C with C = 64 / 128."""
        return (FUSE_TAIL and self.wpk16 is not None and self.k == 3 and self.stride == 1 and not self.ps and self.cin == self.cout
                and self.cin in (64, 128) and CFG_DMA in self.candidates and tail._raw is not None and tail.cin == self.cout
                and tail.wpk16 is not None)

    def out_shape(self, h, w):
        k, s = self.k, self.stride
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        return (2 * ho, 2 * wo, self.cout // 4) if self.ps else (ho, wo, self.cout)

    @property
    def half_ok(self):
        """True when this layer runs on the fp16 path and can therefore read a half-precision activation."""
        return self.wpk16 is not None

    @property
    def half_res_ok(self):
        """True when this layer can add a HALF-precision residual (fp16 path; streaming 1x1 kernel, or the LDS-DMA kernel for a 3x3
        layer with a half-precision input): the last layer of a bottleneck / residual block whose identity path is kept as half
        (see VC_CFG_RES_F16 in include/vc_hip.h)."""
        return self.wpk16 is not None and (CFG_PWS in self.candidates or (self.k == 3 and self.stride == 1 and CFG_DMA in self.candidates))

    def _call_split(self, x, out, act, slope, res, chscale, out_sp3, res_first, epi=EPI_NONE, mul=None, in_xform=IN_NONE, tail=None,
                    out_f16=False):
        """The layer on the split-operand pipeline.  ``x``: an fp32 window (converted here) or a split tensor left by the layer in
        front; ``out_sp3``: leave the result as a split tensor for a split consumer."""
        if epi != EPI_NONE or mul is not None or in_xform != IN_NONE or tail is not None or act >= ACT_SIGMOID:
            raise VcError("the split-operand pipeline has plain / (Leaky)ReLU epilogues only: no GDN, multiplier, input transform, "
                          "fused tail, sigmoid or clamp (a split tensor reached a layer that needs one)")
        if out_f16 and out is None:
            out_f16 = False            # (a hint, like on the native path: the result of a split layer stays fp32 / split)
        if out is not None and out.dtype == "f16":
            raise VcError("the split-operand pipeline stores fp32 or split tensors")
        if not self.split_ok:
            raise VcError("a split tensor reached a layer the split-operand pipeline does not serve")
        if x.dtype == "sp3" and x.c != self.cin_split:
            raise VcError("a split tensor must bring the layer's (padded) channel count")
        xs = x if x.dtype == "sp3" else split3(x, c_out=self.cin_split)
        ho, wo, co = self.out_shape(x.h, x.w)
        if out is None:
            out = T.empty(x.n, ho, wo, co, x.buf.device, "sp3" if out_sp3 else "f32")
        wsp, bsp = self.split_pack()
        d = ConvDesc()
        d.inp = xs.split_view()
        d.out = out.split_view() if out.dtype == "sp3" else out.view()
        d.wpk, d.bias = wsp.data_ptr(), bsp.data_ptr()
        res_sp3 = res is not None and res.dtype == "sp3"
        if res is not None:
            if res.dtype == "f16":
                raise VcError("the split-operand pipeline adds fp32 or split residuals")
            d.res, d.res_sn, d.res_sh, d.res_sw = (res.ptr, res.image_bytes, 0, 0) if res_sp3 else (res.ptr, res.sn, res.sh, res.sw)
        if chscale is not None:
            d.chscale = chscale.data_ptr()
        d.kh = d.kw = self.k
        d.stride = 1
        d.act, d.slope = act, slope
        d.out_mode = OUT_PIXELSHUFFLE2 if self.ps else OUT_PLAIN
        d.cfg = (CFG_SPLIT | CFG_EXACT | CFG_IN_SP3 | (CFG_OUT_SP3 if out.dtype == "sp3" else 0) | (CFG_RES_FIRST if res_first else 0)
                 | (CFG_RES_SP3 if res_sp3 else 0))
        what = f"vc_conv2d_nhwc(split k={self.k},{self.cin}->{self.cout})"
        if timer is None:
            check(lib().vc_conv2d_nhwc(stream(), ctypes.byref(d)), what)
        else:
            flops = 2.0 * x.n * x.h * x.w * self.cout * self.cin * self.k * self.k       # (stride 1: one output position per input pixel)
            key = f"conv k{self.k} s1 {self.cin}->{self.cout} @{x.n}x{x.h}x{x.w}"
            nbytes = (x.n * x.h * x.w * self.cin * 6 + x.n * ho * wo * co * (6 if out.dtype == "sp3" else 4) + self.cout * self.cin * self.k * self.k * 6
                      + (x.n * ho * wo * co * (6 if res_sp3 else 4) if res is not None else 0))
            split_keys.add(key)
            ntw = 4 if self.cout % 64 == 0 else (2 if self.cout % 32 == 0 else 1)
            cpl, th = (2, 12) if self.k == 3 else (1, 24 if ntw == 1 else 16)
            kernel_symbols[key] = f"conv_split_kernel<SplitCfg<{self.k}, {ntw}, {cpl}, {th},"
            if (ntw > 1 and self.cin_split % (32 if self.k == 3 else 16) == 0 and os.environ.get("VC_SPLIT_PADDED", "0") in ("", "0")
                    and os.environ.get("VC_SPLIT3_PADDED", "0") in ("", "0")):
                kernel_symbols[key] = f"conv_split_period_kernel<SplitPeriodCfg<{self.k}, {ntw}, {th},"    # two chunks per period: less tap padding
            timer.bracket(key, flops, lambda: check(lib().vc_conv2d_nhwc(stream(), ctypes.byref(d)), what), nbytes)
        return out

    def __call__(self, x, out=None, act=ACT_NONE, slope=0.01, res=None, epi=EPI_NONE, mul=None,
                 in_xform=IN_NONE, chscale=None, out_f16=False, res_first=False, tail=None, out_sp3=False):
        """``out_f16``: a hint that every consumer of the result is an fp16-path convolution (``half_ok``), so the
        result may be stored as half (bit-identical downstream, half the traffic).  Honoured only when this layer
        itself runs on the fp16 path and allocates its own output; otherwise the result stays fp32."""
        ho, wo, co = self.out_shape(x.h, x.w)

        def vec16(t):
            """an fp32 window the split epilogue can touch in 16-byte groups (split windows are always aligned)"""
            return t is None or t.dtype == "sp3" or (t.dtype == "f32" and t.ptr % 16 == 0 and t.sn % 4 == 0 and t.sh % 4 == 0 and t.sw % 4 == 0)
        if x.dtype == "sp3" or (_FP32_MODE == "split" and self.split_ok and self.split_pays(x.n, x.h, x.w) and epi == EPI_NONE
                                and in_xform == IN_NONE and act < ACT_SIGMOID
                                and mul is None and tail is None and (out is None or out.dtype != "f16")
                                and vec16(out) and vec16(res) and co % 4 == 0
                                and x.dtype == "f32" and x.c == self.cin
                                and (self.cin_split != self.cin or (x.c % 8 == 0 and x.sw % 4 == 0 and x.sh % 4 == 0 and x.sn % 4 == 0 and x.ptr % 16 == 0))):
            return self._call_split(x, out, act, slope, res, chscale, out_sp3, res_first, epi, mul, in_xform, tail, out_f16)
        half_in = x.dtype == "f16"
        esz = 8 if half_in else 4
        use16 = (self.wpk16 is not None and in_xform == IN_NONE and x.sw % esz == 0 and x.sh % esz == 0 and x.sn % esz == 0
                 and x.ptr % 16 == 0)
        if half_in and not use16:
            raise VcError("a half-precision activation reached a layer that is not on the fp16 path")
        # a split consumer behind a NATIVE fp32 layer (stride-2 / 1x1 / GDN / small-cin layers): the classic instances write the three
        # bf16 pieces themselves (CFG_OUT_SP3); the streaming / LDS-DMA configurations do not, so the tuner is held to the classic ones
        sp_out = (out.dtype == "sp3") if out is not None else bool(out_sp3 and _FP32_MODE == "split" and not use16 and co % 8 == 0
                                                                    and self.cfg in (0, 1, 2, 3) and (res is None or res.dtype == "f32"))
        # fp16 mode: the fp32 GDN / IGDN instance of the streaming 1x1 kernel may STORE half (the input -- operand and identity -- of a
        # residual block that runs on the fp16 path; part of that mode's tolerance like every VC_HALF_RESIDUAL tensor)
        gdn_half_ok = (_PRECISION == "fp16" and HALF_ACTIVATIONS and HALF_RESIDUAL and epi != EPI_NONE and self.k == 1 and self.cin == 128
                       and self.cout == 128 and CFG_PWS in self.candidates and mul is x and x.dtype == "f32" and not res_first and chscale is None
                       and (res is None or res.dtype == "f32"))
        if out is None:
            out = T.empty(x.n, ho, wo, co, x.buf.device, "sp3" if sp_out else
                          ("f16" if (out_f16 and HALF_ACTIVATIONS and (use16 or gdn_half_ok) and co % 4 == 0) else "f32"))
        half_out = out.dtype == "f16"
        gdn_half = half_out and not use16 and gdn_half_ok
        if half_out and not use16 and not gdn_half:
            raise VcError("a half-precision output needs the fp16 path")
        d = ConvDesc()
        d.inp, d.out = x.view(True), (out.split_view() if sp_out else out.view(True))
        d.wpk, d.bias = self.wpk.data_ptr(), self.bias.data_ptr()
        res_half = res is not None and res.dtype == "f16"
        if tail is not None and not (self.can_fuse_tail(tail) and use16 and half_in and epi == EPI_NONE and act < ACT_SIGMOID
                                     and chscale is None and not res_first):
            raise VcError("fused tail: a half-precision activation through a 3x3 C -> C layer of the fp16 path (PackedConv.can_fuse_tail)")
        if res_half and tail is None and not (use16 and self.half_res_ok and epi == EPI_NONE and act < ACT_SIGMOID):
            raise VcError("a half-precision residual needs the fp16 path's streaming 1x1 kernel (PackedConv.half_res_ok)")
        if res is not None:
            d.res, d.res_sn, d.res_sh, d.res_sw = res.ptr, res.sn, res.sh, res.sw
        if mul is not None:
            d.mul, d.mul_sn, d.mul_sh, d.mul_sw = mul.ptr, mul.sn, mul.sh, mul.sw
        if chscale is not None:
            d.chscale = chscale.data_ptr()
        d.kh = d.kw = self.k
        d.stride = self.stride
        d.act, d.slope = act, slope
        d.epi, d.in_xform = epi, in_xform
        d.out_mode = OUT_PIXELSHUFFLE2 if self.ps else OUT_PLAIN
        d.cfg = self.cfg
        flags = (CFG_RES_FIRST if res_first else 0) | (CFG_OUT_SP3 if sp_out else 0)      # RES_FIRST: out = act(conv + res) instead of act(conv) + res
        if use16:
            d.wpk = self.wpk16.data_ptr()
            flags |= CFG_F16 | (CFG_IN_F16 if half_in else 0) | (CFG_OUT_F16 if half_out else 0) | (CFG_RES_F16 if res_half else 0)
        # (the tuned choice is per shape AND per epilogue class: the streaming 1x1 kernel, for one, takes plain / ReLU /
        #  GDN epilogues but not sigmoid or clamp, so a layer called both ways must not share one entry)
        key = (x.n, x.h, x.w, flags) if (act < ACT_SIGMOID and epi == EPI_NONE) else (x.n, x.h, x.w, flags, act, epi)
        # (stated on every call of a layer packed with the 128-channel configuration; only the LDS-DMA kernel reads it: its
        #  blocks of 128 may run into the padding of 96 / 160 / 432 ... output channels.  Not part of the tuning key.)
        pack = CFG_PACK128 if self.cfg == 0 else 0
        if gdn_half:                  # the fp32 GDN instance of the streaming kernel, half store
            flags |= CFG_OUT_F16
            d.cfg = CFG_PWS | CFG_EXACT | flags
        elif tail is not None:        # out = tail(act(conv3x3(x))) + res in ONE launch of the LDS-DMA kernel
            tw, tb = tail.tail_pack()
            d.tail_wpk, d.tail_bias = tw.data_ptr(), tb.data_ptr()
            d.cfg = CFG_DMA | CFG_EXACT | flags | pack
        else:
            d.cfg = self._pick_cfg(d, key, flags | pack) | pack
        what = f"vc_conv2d_nhwc(k={self.k},s={self.stride},{self.cin}->{self.cout}{'+1x1 tail' if tail is not None else ''})"

        def launch_once():
            # A tuned choice is keyed by shape and flags, not by the alignment class of the views: a later call of the same shape
            # on a sliced / unaligned view may be refused by a configuration that needs 16-byte accesses (LDS-DMA, streaming 1x1).
            # The general configuration of this packing takes any view: retry on it instead of failing the layer.
            rc = lib().vc_conv2d_nhwc(stream(), ctypes.byref(d))
            if rc == -1 and tail is None and (d.cfg & 0xff) != self.cfg:
                d.cfg = self.cfg | (d.cfg & ~0x1ff)
                rc = lib().vc_conv2d_nhwc(stream(), ctypes.byref(d))
            return rc
        if timer is None:
            check(launch_once(), what)
        else:
            hq, wq = (ho // 2, wo // 2) if self.ps else (ho, wo)
            flops = 2.0 * x.n * hq * wq * self.cout * (self.cin * self.k * self.k + (tail.cin if tail is not None else 0))
            key = f"conv k{self.k}{'+k1' if tail is not None else ''} s{self.stride} {self.cin}->{self.cout} @{x.n}x{x.h}x{x.w}"
            nbytes = (x.n * x.h * x.w * self.cin * (2 if half_in else 4) + x.n * ho * wo * co * (2 if half_out else 4)
                      + self.cout * self.cin * self.k * self.k * (2 if use16 else 4)
                      + (x.n * ho * wo * co * (2 if res_half else 4) if res is not None else 0) + (x.n * ho * wo * co * 4 if mul is not None else 0))
            c0 = d.cfg & 0xff
            if c0 == CFG_DMA:
                kernel_symbols[key] = f"conv_dma_kernel<DmaCfg<{self.k}, {self.k}, {self.cin // (32 if use16 else 16)},"
            elif c0 == CFG_PWS:
                kernel_symbols[key] = "conv_pws_kernel<"
            elif c0 == 7:
                kernel_symbols[key] = f"conv_mfma_kernel<{self.k}, {self.k}, 1, 16, TileCfg<32, 16"
            else:
                kernel_symbols[key] = f"conv_mfma_kernel<{self.k}, {self.k}, {self.stride},"
            timer.bracket(key, flops, lambda: check(launch_once(), what), nbytes)
        return out


# ------------------------------------------------------------------------------------------------
# thin wrappers
# ------------------------------------------------------------------------------------------------
def avgpool_reflectpad(x, k, scale=1.0, out_h=None, out_w=None):
    hp, wp = x.h // k, x.w // k
    out = T.empty(x.n, out_h or hp, out_w or wp, x.c, x.buf.device)
    check(lib().vc_avgpool_reflectpad(stream(), x.view(), out.view(), k, scale), "vc_avgpool_reflectpad")
    return out


def avgpool2_split(x, out_sp3):
    """F.avg_pool2d(x, 2) of a split tensor (window): split or fp32 result."""
    out = T.empty(x.n, x.h // 2, x.w // 2, x.c, x.buf.device, "sp3" if out_sp3 else "f32")
    timed_hbm(f"k_avgpool2_sp3 c{x.c} @{x.n}x{x.h}x{x.w}", x.n * x.c * (6.0 * x.h * x.w + (6.0 if out_sp3 else 4.0) * out.h * out.w),
              lambda: check(lib().vc_avgpool2_sp3(stream(), x.ptr, x.image_bytes, x.n, x.h, x.w, x.c, 1.0, out.ptr if out_sp3 else None,
                                                  out.image_bytes if out_sp3 else 0, NULL_VIEW if out_sp3 else out.view()), "vc_avgpool2_sp3"))
    return out


def maxpool2(x):
    if x.dtype == "sp3":          # between split-operand layers: split in, split out (exactly maxpool + vc_split3)
        out = T.empty(x.n, x.h // 2, x.w // 2, x.c, x.buf.device, "sp3")
        timed_hbm(f"k_maxpool2_sp3 c{x.c} @{x.n}x{x.h}x{x.w}", 6.0 * x.n * x.c * (x.h * x.w + out.h * out.w),
                  lambda: check(lib().vc_maxpool2_sp3(stream(), x.ptr, x.image_bytes, x.n, x.h, x.w, x.c, out.ptr, out.image_bytes), "vc_maxpool2_sp3"))
        return out
    out = T.empty(x.n, x.h // 2, x.w // 2, x.c, x.buf.device)
    check(lib().vc_maxpool2(stream(), x.view(), out.view()), "vc_maxpool2")
    return out


def upsample_bilinear(x, factor, align_corners=False, scale=1.0, out=None):
    if out is None:
        out = T.empty(x.n, x.h * factor, x.w * factor, x.c, x.buf.device)
    if out.dtype == "sp3":        # the up-sampled part of a concat buffer a split-operand convolution reads
        timed_hbm(f"k_upsample_bilinear_sp3 x{factor} c{x.c} @{x.n}x{x.h}x{x.w}", x.n * x.c * (4.0 * x.h * x.w + 6.0 * out.h * out.w),
                  lambda: check(lib().vc_upsample_bilinear_sp3(stream(), x.view(), out.ptr, out.image_bytes, factor, int(align_corners), scale),
                                "vc_upsample_bilinear_sp3"))
        return out
    timed_hbm(f"k_upsample_bilinear x{factor} c{x.c} @{x.n}x{x.h}x{x.w}", 4.0 * x.n * x.c * (x.h * x.w + out.h * out.w),
              lambda: check(lib().vc_upsample_bilinear(stream(), x.view(), out.view(), factor, int(align_corners), scale),
                            "vc_upsample_bilinear"))
    return out


def axpby(a, b, alpha=1.0, beta=1.0, out=None):
    if out is None:
        out = T.empty(a.n, a.h, a.w, a.c, a.buf.device)
    check(lib().vc_axpby(stream(), a.view(), b.view() if b is not None else NULL_VIEW, out.view(), alpha, beta),
          "vc_axpby")
    return out


def to_half(x):
    """Dense half-precision copy of a channels-last fp32 window (round to nearest even): the features the fp16-path
    deformable fusion gathers from."""
    out = T.empty(x.n, x.h, x.w, x.c, x.buf.device, "f16")
    check(lib().vc_to_half(stream(), x.view(), out.ptr), "vc_to_half")
    return out


def to_half_planar(x, cg):
    """Group-planar half-precision copy [n][c / cg][h][w][cg] of a channels-last fp32 window (vc_to_half_planar)."""
    out = T.empty(x.n, x.h, x.w, x.c, x.buf.device, "f16")
    check(lib().vc_to_half_planar(stream(), x.view(), int(cg), out.ptr), "vc_to_half_planar")
    return out


def clamp01(x, out=None):
    """clamp(x, 0, 1) of a channels-last window (``torch.clamp(x_hat, 0, 1)`` of ICIP2024/src/test.py:94 on the device path)."""
    if out is None:
        out = T.empty(x.n, x.h, x.w, x.c, x.buf.device)
    check(lib().vc_clamp01(stream(), x.view(), out.view()), "vc_clamp01")
    return out


def clamp01_nchw(x):
    """The same for a contiguous fp32 CUDA tensor of any shape (a decoded NCHW frame entering the reference buffer)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    n = x.numel()
    check(lib().vc_clamp01(stream(), View(x.data_ptr(), 1, 1, n, 1, n, n, 1), View(out.data_ptr(), 1, 1, n, 1, n, n, 1)), "vc_clamp01")
    return out


def stack_images(items, out=None):
    """Windows of k images each -> ONE batched window, image by image (the batch of a level pass assembled by the
    layout kernel: no torch.cat of the frames)."""
    if len(items) == 1 and out is None:
        return items[0]
    total = sum(t.n for t in items)
    t0 = items[0]
    if out is None:
        out = T.empty(total, t0.h, t0.w, t0.c, t0.buf.device)
    i = 0
    for t in items:
        check(lib().vc_axpby(stream(), t.view(), NULL_VIEW, out.images(i, i + t.n).view(), 1.0, 0.0), "vc_axpby")
        i += t.n
    return out


def warp(convention, img, flow, out=None):
    if out is None:
        out = T.empty(img.n, flow.h, flow.w, img.c, img.buf.device)
    timed_hbm(f"k_warp W{convention} c{img.c} @{img.n}x{img.h}x{img.w}", 4.0 * img.n * img.h * img.w * (2 * img.c + 2),
              lambda: check(lib().vc_warp(stream(), convention, img.view(), flow.view(), out.view()), "vc_warp"))
    return out


def channel_scale(x, gain, out=None):
    if out is None:
        out = T.empty(x.n, x.h, x.w, x.c, x.buf.device)
    check(lib().vc_channel_scale(stream(), x.view(), gain.data_ptr(), out.view()), "vc_channel_scale")
    return out


def psnr_uint8(x_hat, x, h, w, out=None):
    """PSNR of the first image of two NCHW fp32 CUDA tensors on the uint8-rounded [:h,:w] crop (vc_psnr_uint8); returns a
    0-dim float64 device tensor (``out``: a 0-dim float64 view to write into) -- no host synchronisation."""
    if x_hat.dtype != torch.float32 or x.dtype != torch.float32 or not x_hat.is_cuda or not x.is_cuda:
        raise VcError("psnr_uint8 takes fp32 CUDA tensors")
    if x_hat.dim() != 4 or x.shape[1:] != x_hat.shape[1:]:
        raise VcError("psnr_uint8: NCHW tensors of the same frame size")
    a, b = x_hat[0], x[0]
    if not a.is_contiguous() or not b.is_contiguous():
        a, b = a.contiguous(), b.contiguous()
    h, w = min(int(h), a.shape[1]), min(int(w), a.shape[2])      # the loops slice [:h, :w]: a crop larger than the frame is the frame
    slots = lib().vc_bits_slots()
    scratch = torch.empty(slots, dtype=torch.float64, device=x.device)
    if out is None:
        out = torch.empty((), dtype=torch.float64, device=x.device)
    check(lib().vc_psnr_uint8(stream(), a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], a.shape[2], int(h), int(w),
                              scratch.data_ptr(), slots, out.data_ptr()), "vc_psnr_uint8")
    return out


def nchw_frames_to_nhwc(frames, out=None):
    """A list of [k,C,H,W] fp32 CUDA tensors -> ONE batched channels-last window, image by image: the batch of a level pass
    is assembled by the layout kernel itself (no torch.cat copy of the frames)."""
    total = sum(int(f.shape[0]) for f in frames)
    _, c, h, w = frames[0].shape
    if out is None:
        out = T.empty(total, h, w, c, frames[0].device)
    i = 0
    for f in frames:
        f = f.contiguous().float()
        if tuple(f.shape[1:]) != (c, h, w):
            raise VcError("frames of one pass must have the same shape")
        check(lib().vc_nchw_to_nhwc(stream(), f.data_ptr(), out.images(i, i + f.shape[0]).view()), "vc_nchw_to_nhwc")
        i += f.shape[0]
    return out


def attention_gate(a, b, identity, out=None):
    """out = a * sigmoid(b) + identity (compressai AttentionBlock)."""
    if out is None:
        out = T.empty(a.n, a.h, a.w, a.c, a.buf.device)
    check(lib().vc_attention_gate(stream(), a.view(), b.view(), identity.view(), out.view()), "vc_attention_gate")
    return out


def quantize_mask(x, out=None, gain=None, keep_parity=-1, do_round=True):
    """ICIP2024 entropy-model glue: out = keep ? round(x) * gain : 0 (see vc_quantize_mask)."""
    if out is None:
        out = T.empty(x.n, x.h, x.w, x.c, x.buf.device)
    check(lib().vc_quantize_mask(stream(), x.view(), out.view(), None if gain is None else gain.data_ptr(),
                                 keep_parity, int(do_round)), "vc_quantize_mask")
    return out


class PackedDeform:
    """Weights of a torchvision DeformConv2d(k=3, padding=1, groups=G) re-laid out per group for vc_deform_conv2d."""

    def __init__(self, weight, bias, groups, device):
        w = weight.detach().to("cpu", torch.float32).contiguous().numpy()
        cout, cg, kh, kw = w.shape
        if (kh, kw) != (3, 3) or cout % groups:
            raise VcError("only 3x3 grouped deformable convolutions are supported")
        dst = np.empty(groups * 9 * cg * (cout // groups), dtype=np.float32)
        check(lib().vc_deform_pack_weights(w.ctypes.data, cout, cg, groups, dst.ctypes.data), "vc_deform_pack_weights")
        self.wpk = torch.from_numpy(dst).to(device)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()
        self.groups, self.cout, self.cin = groups, cout, cg * groups

    def _bias_ptr(self):
        return None if self.bias is None else self.bias.data_ptr()

    def conv(self, x, offset, mask=None, out=None):
        if out is None:
            out = T.empty(x.n, x.h, x.w, self.cout, x.buf.device)
        check(lib().vc_deform_conv2d(stream(), x.view(), offset.view(), NULL_VIEW if mask is None else mask.view(),
                                     self.wpk.data_ptr(), self._bias_ptr(), self.groups, out.view()), "vc_deform_conv2d")
        return out

    def offset_diversity(self, x1, raw1, flow1, x2, raw2, flow2, magnitude, out=None):
        if out is None:
            out = T.empty(x1.n, x1.h, x1.w, self.cout, x1.buf.device)
        cg = self.cin // self.groups
        # the half-precision-feature entry has ONE instance (csrc/deform.hip: vector gathers, <= 8 groups per reference, offset
        # records fetched in 16-byte pieces): features AND both raw offset tensors 16-byte aligned with strides in whole float4s
        half_x = (_PRECISION == "fp16" and HALF_DEFORM and cg % 4 == 0 and cg >= 8 and self.groups <= 16 and raw1.c % 4 == 0
                  and raw2.c % 4 == 0 and x1.dtype == "f32" and x2.dtype == "f32"
                  and all(t.ptr % 16 == 0 and t.sw % 4 == 0 and t.sh % 4 == 0 and t.sn % 4 == 0 and t.c % 4 == 0 for t in (x1, x2))
                  and all(t.ptr % 16 == 0 and t.sw % 4 == 0 and t.sh % 4 == 0 and t.sn % 4 == 0 for t in (raw1, raw2)))
        planar = half_x and HALF_DEFORM_PLANAR
        if planar:
            n_, h_, w_ = x1.n, x1.h, x1.w
            p1, p2 = to_half_planar(x1, cg), to_half_planar(x2, cg)
            hg = self.groups // 2
            v1 = View(p1.ptr, n_, h_, w_, cg, hg * h_ * w_ * cg, w_ * cg, cg)
            v2 = View(p2.ptr, n_, h_, w_, cg, hg * h_ * w_ * cg, w_ * cg, cg)
        elif half_x:
            x1, x2 = to_half(x1), to_half(x2)
        fn = lib().vc_offset_diversity_hxp if planar else (lib().vc_offset_diversity_hx if half_x else lib().vc_offset_diversity)

        def launch():
            check(fn(stream(), v1 if planar else x1.view(True), raw1.view(), flow1.view(), v2 if planar else x2.view(True), raw2.view(),
                     flow2.view(), float(magnitude), self.wpk.data_ptr(), self._bias_ptr(),
                     self.groups, out.view()), "vc_offset_diversity_hxp" if planar else ("vc_offset_diversity_hx" if half_x else "vc_offset_diversity"))
        if timer is None:
            launch()
        else:
            flops = 2.0 * x1.n * x1.h * x1.w * self.cout * (self.cin // self.groups) * 9
            nbytes = x1.n * x1.h * x1.w * ((2.0 if half_x else 4.0) * self.cin + 4.0 * (raw1.c + raw2.c + 4 + self.cout))
            timer.bracket(f"deform k3 {self.cin}->{self.cout} g{self.groups} @{x1.n}x{x1.h}x{x1.w}", flops, launch, nbytes)
        return out


# ------------------------------------------------------------------------------------------------
# host range coder
# ------------------------------------------------------------------------------------------------
def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    cdf = np.empty(p.size + 1, dtype=np.uint32)
    check(lib().vc_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, cdf.ctypes.data), "vc_pmf_to_quantized_cdf")
    return cdf


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _tables(cdfs, cdf_sizes, offsets):
    cdfs = _i32(cdfs)
    sizes, offs = _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
    if cdfs.ndim != 2 or sizes.size != cdfs.shape[0] or offs.size != cdfs.shape[0]:
        raise VcError("range-coder tables: cdfs must be [n_tables, stride] with one size and one offset per table")
    return cdfs, sizes, offs


def rans_encode(symbols, indexes, cdfs, cdf_sizes, offsets):
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    if sym.size != idx.size:
        raise VcError("range coder: one table index per symbol")
    cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
    cap = lib().vc_rans_bound(sym.size)
    out = np.empty(cap // 4, dtype=np.uint32)
    n = lib().vc_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, sym.size, cdfs.ctypes.data, cdfs.shape[0],
                                          cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data, out.ctypes.data, cap)
    if n < 0:
        check(int(n), "vc_rans_encode_with_indexes")
    return out.view(np.uint8)[:n].tobytes()


def rans_decode(data, indexes, cdfs, cdf_sizes, offsets):
    idx = _i32(indexes).reshape(-1)
    cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(idx.size, dtype=np.int32)
    check(lib().vc_rans_decode_with_indexes(buf.ctypes.data, buf.size, idx.ctypes.data, idx.size, cdfs.ctypes.data,
                                            cdfs.shape[0], cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data, out.ctypes.data),
          "vc_rans_decode_with_indexes")
    return out


class RansStreamDecoder:
    """compressai.ans.RansDecoder with set_stream / decode_stream (ICIP2024/src/model/elic.py:428-429): several decode calls
    over ONE string, each continuing where the previous stopped."""

    def __init__(self, data):
        self.buf = np.frombuffer(bytes(data), dtype=np.uint8)
        self.state = np.zeros(2, dtype=np.uint64)

    def decode_stream(self, indexes, cdfs, cdf_sizes, offsets):
        idx = _i32(indexes).reshape(-1)
        cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
        out = np.empty(idx.size, dtype=np.int32)
        check(lib().vc_rans_decode_stream(self.buf.ctypes.data, self.buf.size, self.state.ctypes.data, idx.ctypes.data, idx.size,
                                          cdfs.ctypes.data, cdfs.shape[0], cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data,
                                          out.ctypes.data), "vc_rans_decode_stream")
        return out
