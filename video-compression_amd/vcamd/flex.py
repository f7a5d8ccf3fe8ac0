"""Flex-Rate hierarchical bi-directional codec on MI355X -- host-side mirror of the reference surface.

Drop-in for (under /root/reference/Flex-Rate-Hier-Bidir-Video-Compression):
  b_model/unet.py:9-91       UNet / UNetConvBlock / UNetUpBlock     -> same names here
  b_model/layers.py:40-73    Gain_Module                            -> :class:`Gain_Module`
  b_model/layers.py:76-305   FlowCompressor / ResidualCompressor    -> same names
  b_model/b_model.py:21-111  BidirFlowRef                           -> :class:`BidirFlowRef`
  test/encode_B.py:74-109, test/decode_B.py:74-95                  -> :func:`encode_B`, :func:`decode_B`
Attribute names reproduce the reference's state_dict keys (checkpoints load child by child like
test/utils.py:253-270).  Reference quirks are reproduced deliberately.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip
from .hip import T
from .layers import (BitCounter, MeanScaleHyperprior, ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride,
                     _Prepared, conv3x3, pack_conv, subpel_conv3x3)
from .lhbdc import _count, _require_cuda, _require_frames, frame_list


# ------------------------------------------------------------------------------------------------
# U-Net (unet.py)
# ------------------------------------------------------------------------------------------------
class UNetConvBlock(nn.Module):
    def __init__(self, in_size, out_size, padding=True):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(in_size, out_size, kernel_size=3, padding=int(padding)), nn.LeakyReLU(0.1),
                                   nn.Conv2d(out_size, out_size, kernel_size=3, padding=int(padding)), nn.LeakyReLU(0.1))


class UNetUpBlock(nn.Module):
    def __init__(self, in_size, out_size, padding=True):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2),
                                nn.Conv2d(in_size, out_size, kernel_size=3, padding=1))
        self.conv_block = UNetConvBlock(in_size, out_size, padding)


class UNet(_Prepared):
    def __init__(self, in_channels=1, n_classes=2, depth=5, wf=5, padding=True):
        super().__init__()
        if not padding:
            raise hip.VcError("only padding=True (as used by the reference) is supported")
        self.padding, self.depth = padding, depth
        prev = in_channels
        self.down_path = nn.ModuleList()
        for i in range(depth):
            self.down_path.append(UNetConvBlock(prev, 2 ** (wf + i), padding))
            prev = 2 ** (wf + i)
        self.midconv = nn.Conv2d(prev, prev, kernel_size=3, padding=1)
        self.up_path = nn.ModuleList()
        for i in reversed(range(depth - 1)):
            self.up_path.append(UNetUpBlock(prev, 2 ** (wf + i), padding))
            prev = 2 ** (wf + i)
        self.last = nn.Conv2d(prev, n_classes, kernel_size=3, padding=1)

    def run(self, x, final_act=hip.ACT_NONE):
        """x: T [n,H,W,Cin] -> T [n,H,W,n_classes].  Each skip tensor is written by its producer straight
        into the upper half of the concat buffer its up-block reads (no torch.cat)."""
        if self._packed is None:
            p = {"down": [(pack_conv(b.block[0]), pack_conv(b.block[2])) for b in self.down_path],
                 "mid": pack_conv(self.midconv),
                 "up": [(pack_conv(u.up[1]), pack_conv(u.conv_block.block[0]), pack_conv(u.conv_block.block[2]))
                        for u in self.up_path],
                 "last": pack_conv(self.last)}
            self._packed = p
        p, dev = self._packed, x.buf.device
        if x.h % (1 << (self.depth - 1)) or x.w % (1 << (self.depth - 1)):
            raise hip.VcError("UNet input must be divisible by 2^(depth-1) (the reference pads frames to x64)")
        # fp32 mode "split" (hip.set_fp32_mode): a level whose up-block convolution runs on the split-operand pipeline keeps its concat
        # buffer -- and everything the split layers hand each other -- as SPLIT tensors: the skip convolution, the up-path convolution
        # and the up-sampling kernel write the three bf16 pieces directly, the skip tensor is pooled from its window of the buffer
        cats = []
        nd = len(p["down"])
        for i, (c1, c2) in enumerate(p["down"]):
            t = c1(x, act=hip.ACT_LRELU, slope=0.1, out_f16=c2.half_ok, out_sp3=hip.wants_split(c2, x))
            if i != nd - 1:
                c = c2.cout
                up_c1 = p["up"][nd - 2 - i][1]                       # the convolution that reads this level's concat buffer
                sp_cat = hip.wants_split_at(up_c1, x.n, t.h, t.w) and c % 8 == 0
                cat = T.empty(x.n, t.h, t.w, 2 * c, dev, "sp3" if sp_cat else "f32")     # [up(c) | skip(c)]
                skip = c2(t, out=cat.channels(c, 2 * c), act=hip.ACT_LRELU, slope=0.1)
                cats.append(cat)
                nxt_c1 = p["down"][i + 1][0]
                if sp_cat:
                    x = hip.avgpool2_split(skip, out_sp3=hip.wants_split_at(nxt_c1, x.n, t.h // 2, t.w // 2))
                else:
                    x = hip.avgpool_reflectpad(skip, 2)
            else:
                x = c2(t, act=hip.ACT_LRELU, slope=0.1, out_sp3=hip.wants_split(p["mid"], t))
        x = p["mid"](x, act=hip.ACT_LRELU, slope=0.1)
        for i, (cu, c1, c2) in enumerate(p["up"]):
            cat = cats[-i - 1]
            if hip.wants_split_at(cu, x.n, cat.h, cat.w) and x.c % 8 == 0 and x.dtype == "f32":
                ups = hip.upsample_bilinear(x, 2, out=T.empty(x.n, cat.h, cat.w, x.c, dev, "sp3"))
            else:
                ups = hip.upsample_bilinear(x, 2)
            cu(ups, out=cat.channels(0, cu.cout))
            x = c2(c1(cat, act=hip.ACT_LRELU, slope=0.1, out_f16=c2.half_ok, out_sp3=hip.wants_split(c2, cat)), act=hip.ACT_LRELU, slope=0.1)
        return p["last"](x, act=final_act)

    def forward(self, x):
        _require_cuda(x)
        return hip.nhwc_to_nchw(self.run(hip.nchw_to_nhwc(x)))


# ------------------------------------------------------------------------------------------------
# gained hyperprior codecs (layers.py)
# ------------------------------------------------------------------------------------------------
class Gain_Module(nn.Module):
    def __init__(self, n=6, N=128, bias=False, inv=False):
        super().__init__()
        self.gain_matrix = nn.Parameter(torch.ones(n, N))
        if bias:
            raise hip.VcError("Gain_Module(bias=True) is never used by the reference (b_model.py:31-32)")
        self.bias = False
        self._cache = {}

    def _load_from_state_dict(self, *a, **k):
        self._cache = {}
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._cache = {}
        return super()._apply(fn, *a, **k)

    def vector(self, n, l):
        """per-channel gain on the device (layers.py:54-66); ``n`` is the reference's list ``[int]``"""
        key = (int(n[0]), float(l))
        if key not in self._cache:
            with torch.no_grad():
                g = self.gain_matrix.detach().cpu()
                if l != 1:
                    v = torch.abs(g[n]) ** l * torch.abs(g[[n[0] + 1]]) ** (1 - l)
                else:
                    v = torch.abs(g[n])
            self._cache[key] = v.reshape(-1).float().contiguous().to(self.gain_matrix.device)
        return self._cache[key]


def _require_inference(codec):
    """Noise quantisation (entropy models in training mode) is outside the inference hot path: refuse it loudly."""
    if codec.entropy_bottleneck.training or codec.gaussian_conditional.training:
        raise NotImplementedError("the entropy models are in training mode (noise quantisation): call .eval() -- "
                                  "only the inference path is built")


class _GainedCodec(MeanScaleHyperprior):
    def __init__(self, n, in_ch, out_ch, N=128, bias=False, zero_last=False, **kwargs):
        super().__init__(N=N, M=N, **kwargs)
        self.g_a = nn.Sequential(
            ResidualBlockWithStride(in_ch, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            conv3x3(N, N, stride=2))
        self.h_a = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2))
        self.h_s = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), subpel_conv3x3(N, N, 2), nn.LeakyReLU(inplace=True),
            conv3x3(N, N * 3 // 2), nn.LeakyReLU(inplace=True), subpel_conv3x3(N * 3 // 2, N * 3 // 2, 2),
            nn.LeakyReLU(inplace=True), conv3x3(N * 3 // 2, N * 2))
        self.g_s = nn.Sequential(
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2), ResidualBlock(N, N),
            ResidualBlockUpsample(N, N, 2), ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), subpel_conv3x3(N, out_ch, 2))
        if zero_last:   # layers.py:125-126
            self.g_s[-1][0].weight.data.fill_(0.0)
            self.g_s[-1][0].bias.data.fill_(0.0)
        self.gain_unit = Gain_Module(n=n, N=N, bias=bias, inv=False)
        self.inv_gain_unit = Gain_Module(n=n, N=N, bias=bias, inv=True)
        self.hyper_gain_unit = Gain_Module(n=n, N=N, bias=bias, inv=False)
        self.hyper_inv_gain_unit = Gain_Module(n=n, N=N, bias=bias, inv=True)

    def gains(self, n, l):
        return (self.gain_unit.vector(n, l), self.inv_gain_unit.vector(n, l),
                self.hyper_gain_unit.vector(n, l), self.hyper_inv_gain_unit.vector(n, l))

    def forward(self, x, n=None, l=None, train=False):
        """{"x_hat", "likelihoods": {"y","z"}} like layers.py:133-151 (+ "bits": the -log2 sums, reduced on the device).
        ``train`` only sets ``self.training`` there (layers.py:134): whether the entropy models quantise with noise
        is decided by THEIR training flags, so on an ``.eval()`` model train=True computes exactly the same thing."""
        self.training = train
        _require_inference(self)
        _require_cuda(x)
        bits = BitCounter(x.device, max_rows=2 * x.shape[0])
        lik = {}
        x_hat = self.forward_t(hip.nchw_to_nhwc(x), bits, self.gains(n, l), likelihoods=lik)
        tot = bits.totals()
        tot = tot.view(-1, 2)       # rows are (y, z) per image; the reference sums over the whole batch
        return {"x_hat": hip.nhwc_to_nchw(x_hat), "likelihoods": lik,
                "bits": {"y": tot[:, 0].sum(), "z": tot[:, 1].sum()}}

    def compress(self, x, n, l):
        _require_cuda(x)
        # layers.py:167 codes the UN-gained y while sigma/mu come from the gained path (quirk B.6)
        strings, (hz, wz) = self.compress_t(hip.nchw_to_nhwc(x), self.gains(n, l), code_ungained_y=True)
        return {"strings": strings, "shape": torch.Size([hz, wz])}

    def decompress(self, strings, shape, n, l):
        dev = self.entropy_bottleneck.quantiles.device
        # layers.py:185 -- .clamp_(0, 1), applied to the flow refinement too
        x_hat = self.decompress_t(strings, shape, dev, self.gains(n, l), final_act=hip.ACT_CLAMP01)
        return {"x_hat": hip.nhwc_to_nchw(x_hat)}


class FlowCompressor(_GainedCodec):
    def __init__(self, n=6, in_ch=19, out_ch=5, N=128, bias=False, **kwargs):
        super().__init__(n, in_ch, out_ch, N=N, bias=bias, zero_last=True, **kwargs)


class ResidualCompressor(_GainedCodec):
    def __init__(self, n=6, in_ch=3, N=128, bias=False, **kwargs):
        super().__init__(n, in_ch, in_ch, N=N, bias=bias, **kwargs)


# ------------------------------------------------------------------------------------------------
# the B-frame model (b_model.py)
# ------------------------------------------------------------------------------------------------
class BidirFlowRef(nn.Module):
    """Bidirectional compression with flow refinement.  ``forward`` returns {"x_hat","size","rate"}."""

    def __init__(self, n=6, N=128):
        super().__init__()
        self.flow_predictor = UNet(6, 4, 5)
        self.Mask = UNet(16, 2, 4)
        self.flow_compressor = FlowCompressor(n=n, in_ch=19, out_ch=4, N=N, bias=False)
        self.residual_compressor = ResidualCompressor(n=n, in_ch=3, N=N, bias=False)

    def backwarp(self, img, flow):
        _require_cuda(img)
        return hip.nhwc_to_nchw(hip.warp(hip.WARP_W2, hip.nchw_to_nhwc(img), hip.nchw_to_nhwc(flow)))

    # -- channels-last stages ----------------------------------------------------------------------
    def _process_t(self, xb_, xa_, xc_=None, t=0.5):
        """b_model.py:35-45 into one 19-channel buffer [Ft0 | Ft1 | x0 | x1 | warp(x0) | warp(x1) | x_cur]."""
        xb_, xa_ = frame_list(xb_), frame_list(xa_)         # tensors or lists of frames (no torch.cat of a level's frames)
        n, (_, _, h, w), dev = _count(xb_), xb_[0].shape, xb_[0].device
        L = hip.lib()
        buf = T.empty(n, h, w, 19, dev)
        hip.nchw_frames_to_nhwc(xb_, out=buf.channels(4, 7))
        hip.nchw_frames_to_nhwc(xa_, out=buf.channels(7, 10))
        if xc_ is not None:
            hip.nchw_frames_to_nhwc(frame_list(xc_), out=buf.channels(16, 19))
        flow = self.flow_predictor.run(buf.channels(4, 10))
        ft0, ft1 = buf.channels(0, 2), buf.channels(2, 4)
        hip.check(L.vc_flex_motion_split(hip.stream(), flow.view(), ft0.view(), ft1.view(), t), "vc_flex_motion_split")
        hip.warp(hip.WARP_W2, buf.channels(4, 7), ft0, out=buf.channels(10, 13))
        hip.warp(hip.WARP_W2, buf.channels(7, 10), ft1, out=buf.channels(13, 16))
        return buf

    def _compensate_t(self, buf, flow_hat, cur=None, trace=None):
        """b_model.py:61-73: refine the motion, warp, 2-channel mask, normalised blend (+ residual)."""
        n, h, w, dev = buf.n, buf.h, buf.w, buf.buf.device
        L = hip.lib()
        mbuf = T.empty(n, h, w, 16, dev)     # [mv_b' | mv_a' | x0 | x1 | x_b | x_a]
        hip.axpby(buf.channels(0, 2), flow_hat.channels(0, 2), out=mbuf.channels(0, 2))
        hip.axpby(buf.channels(2, 4), flow_hat.channels(2, 4), out=mbuf.channels(2, 4))
        hip.axpby(buf.channels(4, 10), None, out=mbuf.channels(4, 10))
        hip.warp(hip.WARP_W2, buf.channels(4, 7), mbuf.channels(0, 2), out=mbuf.channels(10, 13))
        hip.warp(hip.WARP_W2, buf.channels(7, 10), mbuf.channels(2, 4), out=mbuf.channels(13, 16))
        mask = self.Mask.run(mbuf, final_act=hip.ACT_SIGMOID)
        if trace is not None:
            trace.update({"mbuf": mbuf, "mask": mask})
        pred = T.empty(n, h, w, 3, dev)
        resid = T.empty(n, h, w, 3, dev) if cur is not None else None
        hip.check(L.vc_flex_blend(hip.stream(), mbuf.channels(10, 13).view(), mbuf.channels(13, 16).view(), mask.view(),
                                  cur.view() if cur is not None else hip.NULL_VIEW, pred.view(),
                                  resid.view() if resid is not None else hip.NULL_VIEW), "vc_flex_blend")
        return pred, resid

    def process(self, x0, x1, t=0.5):
        _require_cuda(x0)
        buf = self._process_t(x0.contiguous().float(), x1.contiguous().float(), None, t)
        return (hip.nhwc_to_nchw(buf.channels(0, 2)), hip.nhwc_to_nchw(buf.channels(2, 4)),
                hip.nhwc_to_nchw(buf.channels(0, 16)))

    def forward_device(self, x_before, x_current, x_after, n=None, l=1, trace=None):
        """B-frame path with no host synchronisation (graph-capturable): (x_hat, bits[B,4] float64 device tensor =
        flow.y, flow.z, res.y, res.z per frame).  A batch codes B independent frames at the SAME rate point (n, l) --
        the frames of one hierarchy level of a GOP (gop.code_gop_flex)."""
        _require_frames(x_before, x_current, x_after)
        xb_, xc_, xa_ = (frame_list(t) for t in (x_before, x_current, x_after))
        dev, b = xc_[0].device, _count(xc_)
        buf = self._process_t(xb_, xa_, xc_)
        bits = BitCounter(dev, max_rows=4 * b)
        t_mv, t_res = ({}, {}) if trace is not None else (None, None)     # parity instrumentation (tests / bench.py)
        flow_hat = self.flow_compressor.forward_t(buf, bits, self.flow_compressor.gains(n, l), trace=t_mv)
        pred, resid = self._compensate_t(buf, flow_hat, cur=buf.channels(16, 19), trace=trace)
        res_hat = self.residual_compressor.forward_t(resid, bits, self.residual_compressor.gains(n, l), trace=t_res)
        if trace is not None:
            trace.update({"buf": buf, "flow_hat": flow_hat, "pred": pred, "resid": resid, "flow": t_mv, "res": t_res})
        # rows: flow (y, z) per image, then residual (y, z) per image
        return hip.nhwc_to_nchw(hip.axpby(pred, res_hat)), bits.totals().view(2, b, 2).permute(1, 0, 2).reshape(b, 4)

    def forward(self, x_before, x_current, x_after, n=None, l=1, train=False):
        # b_model.py:49-96: ``train`` is handed to the compressors, where it only sets their own .training attribute
        for comp in (self.flow_compressor, self.residual_compressor):
            comp.training = train
            _require_inference(comp)
        if x_current.shape[0] != 1:   # per-item sizes: run items one by one (the reference harness uses batch 1)
            outs = [self.forward(x_before[i:i + 1], x_current[i:i + 1], x_after[i:i + 1], n, l, train)
                    for i in range(x_current.shape[0])]
            return {k: torch.cat([o[k] for o in outs], 0) for k in ("x_hat", "size", "rate")}
        x_hat, tot = self.forward_device(x_before, x_current, x_after, n, l)
        num_pixels = x_current.shape[2] * x_current.shape[3]
        size = tot.sum().reshape(1)
        return {"x_hat": x_hat, "size": size.to(torch.float32), "rate": (size / num_pixels).to(torch.float32)}


# ------------------------------------------------------------------------------------------------
# CLI functions (test/encode_B.py, test/decode_B.py)
# ------------------------------------------------------------------------------------------------
def encode_B(model, x_before, x_current, x_after, n=None, l=1.0, train=False, trace=None):
    """test/encode_B.py:72-122.  ``trace``: a dict that receives {"flow": {...}, "res": {...}} = the coder's integers."""
    for t in (x_before, x_current, x_after):
        _require_cuda(t)
    xb_, xc_, xa_ = (t.contiguous().float() for t in (x_before, x_current, x_after))
    dev = xc_.device
    buf = model._process_t(xb_, xa_, xc_)
    fc, rc = model.flow_compressor, model.residual_compressor
    t_mv, t_res = ({}, {}) if trace is not None else (None, None)
    strings, shape = fc.compress_t(buf, fc.gains([n], l), code_ungained_y=True, trace=t_mv)
    mv_bits = {"strings": strings, "shape": torch.Size(shape)}
    flow_hat = fc.forward_t(buf, BitCounter(dev, max_rows=2 * buf.n), fc.gains([n], l))    # un-clamped (encode_B.py:92-93)
    _, resid = model._compensate_t(buf, flow_hat, cur=buf.channels(16, 19))
    strings, shape = rc.compress_t(resid, rc.gains([n], l), code_ungained_y=True, trace=t_res)
    if trace is not None:
        trace.update({"flow": t_mv, "res": t_res, "resid": resid})
    return mv_bits, {"strings": strings, "shape": torch.Size(shape)}


def decode_B(model, x_before, x_after, string_flow, string_res, shape_flow, shape_res, n, l, trace=None):
    for t in (x_before, x_after):
        _require_cuda(t)
    xb_, xa_ = x_before.contiguous().float(), x_after.contiguous().float()
    dev = xb_.device
    buf = model._process_t(xb_, xa_, None)
    fc, rc = model.flow_compressor, model.residual_compressor
    t_mv, t_res = ({}, {}) if trace is not None else (None, None)
    flow_hat = fc.decompress_t(string_flow, shape_flow, dev, fc.gains([n], l), final_act=hip.ACT_CLAMP01, trace=t_mv)
    pred, _ = model._compensate_t(buf, flow_hat)
    res_hat = rc.decompress_t(string_res, shape_res, dev, rc.gains([n], l), final_act=hip.ACT_CLAMP01, trace=t_res)
    if trace is not None:
        trace.update({"flow": t_mv, "res": t_res})
    return hip.nhwc_to_nchw(hip.axpby(res_hat, pred))


def write_container(path_or_none, l, mv_bits, res_bits):
    """test/encode_B.py:124-145: same 24-byte layout as LHBDC; the first field is np.array(l, uint32), i.e.
    the interpolation factor TRUNCATED to an integer (0.33 -> 0); n is not stored (quirk B.8)."""
    from .lhbdc import write_container as _w
    return _w(path_or_none, np.array(l).astype(np.uint32), mv_bits, res_bits)


def read_container(blob):
    from .lhbdc import read_container as _r
    return _r(blob)
