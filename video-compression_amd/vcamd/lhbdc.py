"""LHBDC B-frame codec on MI355X -- host-side mirror of the reference module surface.

Drop-in for (file:line under /root/reference/LHBDC):
  model/flow.py:30-101   ``Network`` (SPyNet)            -> :class:`Network`
  model/layers.py:43-116 ``MVCompressor``                -> :class:`MVCompressor`
  model/layers.py:118-191 ``ResidualCompressor``         -> :class:`ResidualCompressor`
  model/layers.py:194-249 ``Mask``                       -> :class:`Mask`
  model/m.py:20-126      ``Model``                       -> :class:`Model`
  encode_B.py:39-105 / decode_B.py:31-86                 -> module functions
Same constructor arguments, attribute names (= state_dict keys), method names, argument order and return
structure.  Tensors at the boundary are the reference's NCHW fp32 CUDA tensors; inside, everything is a
channels-last window (``hip.T``) driven through libvc_hip.so -- there is no torch-operator fallback.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip
from .hip import T
from .layers import (BitCounter, MeanScaleHyperprior, ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride,
                     _Prepared, conv3x3, pack_conv, subpel_conv3x3)


def _require_cuda(x):
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise hip.VcError("inputs must be CUDA (HIP) tensors: this implementation has no CPU path")


def frame_list(x):
    """A batch of frames as a list of contiguous fp32 [k,3,H,W] tensors.  ``x`` is one NCHW tensor or a list / tuple of
    them: the level passes of a GOP hand over the frames they batch as a LIST, and the layout kernels assemble the
    batch image by image (no torch.cat copy)."""
    items = list(x) if isinstance(x, (list, tuple)) else [x]
    for f in items:
        _require_cuda(f)
    return [f.contiguous().float() for f in items]


def _count(frames):
    return sum(int(f.shape[0]) for f in frames)


def _require_frames(*groups):
    """The caller pads frames to a multiple of 64 (encode_B.py:49-56, test/utils.py `pad`); the reference fails with
    shape mismatches deep inside otherwise -- fail early and clearly instead.  Each argument: a tensor or a list of them."""
    shape, count = None, None
    for g in groups:
        items = list(g) if isinstance(g, (list, tuple)) else [g]
        for f in items:
            _require_cuda(f)
            if f.dim() != 4 or f.shape[1] != 3:
                raise hip.VcError(f"expected [N,3,H,W] frames, got {tuple(f.shape)}")
            if f.shape[2] % 64 or f.shape[3] % 64:
                raise hip.VcError(f"frame size {f.shape[2]}x{f.shape[3]} is not a multiple of 64: pad it first (pad/process_frame)")
            if shape is not None and tuple(f.shape[1:]) != shape:
                raise hip.VcError("all frames of a triple must have the same shape")
            shape = tuple(f.shape[1:])
        n = sum(int(f.shape[0]) for f in items)
        if count is not None and n != count:
            raise hip.VcError("all frames of a triple must have the same shape")
        count = n


# ------------------------------------------------------------------------------------------------
# SPyNet
# ------------------------------------------------------------------------------------------------
class _Basic(nn.Module):
    def __init__(self, intLevel):
        super().__init__()
        chans = (8, 32, 64, 32, 16, 2)
        layers = []
        for i in range(5):
            layers.append(nn.Conv2d(chans[i], chans[i + 1], kernel_size=7, stride=1, padding=3))
            if i < 4:
                layers.append(nn.ReLU(inplace=False))
        self.netBasic = nn.Sequential(*layers)


class Network(_Prepared):
    """SPyNet, 6 levels.  ``forward(tenFirst, tenSecond) -> flow [B,2,H,W]`` like flow.py:77."""

    def __init__(self):
        super().__init__()
        self.netBasic = nn.ModuleList([_Basic(i) for i in range(6)])

    def _convs(self, level):
        if self._packed is None:
            self._packed = {}
        if level not in self._packed:
            self._packed[level] = [pack_conv(self.netBasic[level].netBasic[j]) for j in (0, 2, 4, 6, 8)]
        return self._packed[level]

    @staticmethod
    def pyramid(level0):
        """flow.py:83-88: halve (avg-pool 2) up to five times while a side is > 32."""
        pyr = [level0]
        for _ in range(5):
            if pyr[0].h > 32 or pyr[0].w > 32:
                pyr.insert(0, hip.avgpool_reflectpad(pyr[0], 2))
        return pyr

    def flow_t(self, pyr_first, pyr_second):
        """Coarse-to-fine estimation on pre-built (batched) pyramids; returns T [n,H,W,2]."""
        L = hip.lib()
        flow = None
        for lvl in range(len(pyr_first)):
            f1, f2 = pyr_first[lvl], pyr_second[lvl]
            dev = f1.buf.device
            c = self._convs(lvl)
            # (fp32 mode "split", hip.set_fp32_mode: the 8 -> 32 -> 64 -> 32 layers on the split-operand pipeline; the level input and
            #  the activations between them are split tensors -- every producer writes the three bf16 pieces itself -- where the level
            #  is large enough for that pipeline to pay)
            sp = hip.fp32_mode() == "split" and c[0].split_ok and c[1].split_ok and c[1].split_pays(f1.n, f1.h, f1.w)
            feat = T.empty(f1.n, f1.h, f1.w, 8, dev, "sp3" if sp else "f32")
            up = T.empty(f1.n, f1.h, f1.w, 2, dev)
            fv = flow.view() if flow is not None else _zero_flow_view(f1)
            # algorithmic traffic: both frames read (3 ch each) + coarse flow read (2 ch at 1/4 of the pixels) + 8-ch level input
            # and 2-ch upsampled flow written
            if sp:
                hip.timed_hbm(f"k_spynet_level_input @{f1.n}x{f1.h}x{f1.w}", f1.n * f1.h * f1.w * (4.0 * (3 + 3 + 0.5 + 2) + 48.0),
                              lambda: hip.check(L.vc_spynet_level_input_sp3(hip.stream(), f1.view(), f2.view(), fv, feat.ptr, up.view()),
                                                "vc_spynet_level_input_sp3"))
            else:
                hip.timed_hbm(f"k_spynet_level_input @{f1.n}x{f1.h}x{f1.w}", 4.0 * f1.n * f1.h * f1.w * (3 + 3 + 0.5 + 8 + 2),
                              lambda: hip.check(L.vc_spynet_level_input(hip.stream(), f1.view(), f2.view(), fv, feat.view(), up.view()),
                                                "vc_spynet_level_input"))
            # (each intermediate feeds exactly one convolution: on the fp16 path it is kept as half in HBM)
            x = c[0](feat, act=hip.ACT_RELU, out_f16=c[1].half_ok, out_sp3=sp)
            x = c[1](x, act=hip.ACT_RELU, out_f16=c[2].half_ok, out_sp3=sp)
            x = c[2](x, act=hip.ACT_RELU, out_f16=c[3].half_ok, out_sp3=sp and hip.wants_split(c[3], f1))
            x = c[3](x, act=hip.ACT_RELU, out_f16=c[4].half_ok)
            flow = c[4](x, res=up)
        return flow

    def preprocess_into(self, frame_nchw, dst):
        hip.check(hip.lib().vc_spynet_preprocess(hip.stream(), frame_nchw.data_ptr(), dst.view()), "vc_spynet_preprocess")

    def forward(self, tenFirst, tenSecond):
        _require_cuda(tenFirst)
        a, b = tenFirst.contiguous().float(), tenSecond.contiguous().float()
        n, _, h, w = a.shape
        p1, p2 = T.empty(n, h, w, 3, a.device), T.empty(n, h, w, 3, a.device)
        self.preprocess_into(a, p1)
        self.preprocess_into(b, p2)
        return hip.nhwc_to_nchw(self.flow_t(self.pyramid(p1), self.pyramid(p2)))


def _zero_flow_view(f1):
    """First level: the initial flow is all zeros at half the coarsest size (flow.py:90) -> NULL view
    (the kernel then uses zeros), but the size check still wants consistent dims."""
    v = hip.View(None, f1.n, f1.h // 2, f1.w // 2, 2, 0, 0, 0)
    return v


# ------------------------------------------------------------------------------------------------
# hyperprior codecs
# ------------------------------------------------------------------------------------------------
class _LhbdcCodec(MeanScaleHyperprior):
    def __init__(self, io_channels, N=128, **kwargs):
        super().__init__(N=N, M=N, **kwargs)
        self.g_a = nn.Sequential(
            ResidualBlockWithStride(io_channels, N, stride=2), ResidualBlock(N, N),
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
            ResidualBlock(N, N), subpel_conv3x3(N, io_channels, 2))

    def forward(self, x):
        """{"x_hat": NCHW tensor, "likelihoods": {"y","z"}} like layers.py:72-91, plus "bits": {"y","z"} = the
        -log2 sums of the same likelihoods reduced on the device (what m.py:73-91 computes from the tensors)."""
        _require_cuda(x)
        bits = BitCounter(x.device, max_rows=2 * x.shape[0])
        lik = {}
        x_hat = self.forward_t(hip.nchw_to_nhwc(x), bits, likelihoods=lik)
        tot = bits.totals()
        tot = tot.view(-1, 2)       # rows are (y, z) per image; the reference sums over the whole batch
        return {"x_hat": hip.nhwc_to_nchw(x_hat), "likelihoods": lik,
                "bits": {"y": tot[:, 0].sum(), "z": tot[:, 1].sum()}}

    def compress(self, x):
        _require_cuda(x)
        strings, (hz, wz) = self.compress_t(hip.nchw_to_nhwc(x))
        return {"strings": strings, "shape": torch.Size([hz, wz])}

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 2
        dev = self.entropy_bottleneck.quantiles.device
        return {"x_hat": hip.nhwc_to_nchw(self.decompress_t(strings, shape, dev))}


class MVCompressor(_LhbdcCodec):
    def __init__(self, N=128, **kwargs):
        super().__init__(4, N=N, **kwargs)


class ResidualCompressor(_LhbdcCodec):
    def __init__(self, N=128, **kwargs):
        super().__init__(3, N=N, **kwargs)


# ------------------------------------------------------------------------------------------------
# Mask U-Net
# ------------------------------------------------------------------------------------------------
def _conv(in_ch, out_ch, k, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=k, stride=stride, padding=k // 2)


class Mask(_Prepared):
    def __init__(self, ch=32):
        super().__init__()
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv1 = _conv(6, ch, 5)
        self.conv2 = _conv(ch, ch * 2, 5)
        self.conv3 = _conv(ch * 2, ch * 4, 3)
        self.bottleneck = _conv(ch * 4, ch * 4, 3)
        self.deconv1 = _conv(ch * 8, ch * 4, 3)
        self.deconv2 = _conv(ch * 4 + ch * 2, ch * 2, 5)
        self.deconv3 = _conv(ch * 2 + ch, ch, 5)
        self.conv4 = _conv(ch, 1, 5)
        self.ch = ch

    def run(self, x):
        """x: T [n,H,W,6] -> mask T [n,H,W,1].  Skip tensors are produced directly inside the channel
        slice of the concat buffer the decoder convolution reads (torch.cat never materialises)."""
        if self._packed is None:
            self._packed = {k: pack_conv(getattr(self, k)) for k in
                            ("conv1", "conv2", "conv3", "bottleneck", "deconv1", "deconv2", "deconv3", "conv4")}
        p, ch, dev = self._packed, self.ch, x.buf.device
        n, h, w = x.n, x.h, x.w
        # fp32 mode "split" (frames large enough for every layer below to run on the split-operand pipeline): the concat buffers and
        # everything between the encoder layers are SPLIT tensors -- conv1's classic epilogue, the split layers' epilogues, the
        # pooling and the up-sampling kernels write the three bf16 pieces directly; no conversion pass
        if all(hip.wants_split_at(p[k], n, h >> s, w >> s) for k, s in (("conv2", 1), ("conv3", 2), ("bottleneck", 3), ("deconv1", 2),
                                                                           ("deconv2", 1), ("deconv3", 0))) and x.dtype == "f32":
            cat3 = T.empty(n, h, w, ch * 3, dev, "sp3")              # [up(64) | conv1(32)]
            cat2 = T.empty(n, h // 2, w // 2, ch * 6, dev, "sp3")    # [up(128) | conv2(64)]
            cat1 = T.empty(n, h // 4, w // 4, ch * 8, dev, "sp3")    # [up(128) | conv3(128)]
            s1 = p["conv1"](x, out=cat3.channels(ch * 2, ch * 3), act=hip.ACT_RELU)
            s2 = p["conv2"](hip.maxpool2(s1), out=cat2.channels(ch * 4, ch * 6), act=hip.ACT_RELU)
            s3 = p["conv3"](hip.maxpool2(s2), out=cat1.channels(ch * 4, ch * 8), act=hip.ACT_RELU)
            b = p["bottleneck"](hip.maxpool2(s3), act=hip.ACT_RELU)
            hip.upsample_bilinear(b, 2, out=cat1.channels(0, ch * 4))
            d1 = p["deconv1"](cat1, act=hip.ACT_RELU)
            hip.upsample_bilinear(d1, 2, out=cat2.channels(0, ch * 4))
            d2 = p["deconv2"](cat2, act=hip.ACT_RELU)
            hip.upsample_bilinear(d2, 2, out=cat3.channels(0, ch * 2))
            d3 = p["deconv3"](cat3, act=hip.ACT_RELU)
            return p["conv4"](d3, act=hip.ACT_SIGMOID)
        cat3 = T.empty(n, h, w, ch * 2 + ch, dev)              # [up(64) | conv1(32)]
        cat2 = T.empty(n, h // 2, w // 2, ch * 4 + ch * 2, dev)  # [up(128) | conv2(64)]
        cat1 = T.empty(n, h // 4, w // 4, ch * 8, dev)           # [up(128) | conv3(128)]
        s1 = p["conv1"](x, out=cat3.channels(ch * 2, ch * 3), act=hip.ACT_RELU)
        s2 = p["conv2"](hip.maxpool2(s1), out=cat2.channels(ch * 4, ch * 6), act=hip.ACT_RELU)
        s3 = p["conv3"](hip.maxpool2(s2), out=cat1.channels(ch * 4, ch * 8), act=hip.ACT_RELU)
        b = p["bottleneck"](hip.maxpool2(s3), act=hip.ACT_RELU)
        hip.upsample_bilinear(b, 2, out=cat1.channels(0, ch * 4))
        d1 = p["deconv1"](cat1, act=hip.ACT_RELU)
        hip.upsample_bilinear(d1, 2, out=cat2.channels(0, ch * 4))
        d2 = p["deconv2"](cat2, act=hip.ACT_RELU)
        hip.upsample_bilinear(d2, 2, out=cat3.channels(0, ch * 2))
        d3 = p["deconv3"](cat3, act=hip.ACT_RELU)
        return p["conv4"](d3, act=hip.ACT_SIGMOID)

    def forward(self, x):
        _require_cuda(x)
        return hip.nhwc_to_nchw(self.run(hip.nchw_to_nhwc(x)))


# ------------------------------------------------------------------------------------------------
# the B-frame model
# ------------------------------------------------------------------------------------------------
def _pad64(v):
    return (64 - (v % 64)) % 64


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.FlowNet = Network()
        self.mv_compressor = MVCompressor()
        self.residual_compressor = ResidualCompressor()
        self.masknet = Mask()
        self.upsample_flow = nn.Upsample(scale_factor=4, mode="bilinear")

    # -- helpers kept for API parity (m.py:101-126) ---------------------------------------------
    def pad(self, im):
        _require_cuda(im)
        t = hip.nchw_to_nhwc(im)
        out = hip.avgpool_reflectpad(t, 1, 1.0, t.h + _pad64(t.h), t.w + _pad64(t.w))
        return hip.nhwc_to_nchw(out)

    def backwarp(self, tenInput, tenFlow):
        _require_cuda(tenInput)
        return hip.nhwc_to_nchw(hip.warp(hip.WARP_W1, hip.nchw_to_nhwc(tenInput), hip.nchw_to_nhwc(tenFlow)))

    # -- stages (channels-last) --------------------------------------------------------------------
    def _flows(self, frames, pairs):
        """Batched SPyNet: ``pairs`` = list of (first, second) keys into ``frames`` (NCHW tensors).
        Each distinct frame is pre-processed once; all pairs run as one batch per pyramid level."""
        frames = {k_: frame_list(v) for k_, v in frames.items()}
        some = next(iter(frames.values()))
        n, (_, _, h, w), dev = _count(some), some[0].shape, some[0].device
        k = len(pairs)
        first, second = T.empty(k * n, h, w, 3, dev), T.empty(k * n, h, w, 3, dev)
        for i, (a, b) in enumerate(pairs):
            for dst, key in ((first, a), (second, b)):
                j = i * n
                for f in frames[key]:
                    self.FlowNet.preprocess_into(f, dst.images(j, j + f.shape[0]))
                    j += f.shape[0]
        return self.FlowNet.flow_t(Network.pyramid(first), Network.pyramid(second))

    @staticmethod
    def _pool_pad(flow, scale):
        hp, wp = flow.h // 4, flow.w // 4
        return hip.avgpool_reflectpad(flow, 4, scale, hp + _pad64(hp), wp + _pad64(wp)), hp, wp

    def _predict(self, xb, xa, mv_hat, flow_ab, flow_ba, hh, ww, cur=None, trace=None):
        """m.py:55-67: add the predictors back, crop, x4 bilinear, warp both references, mask, blend.
        Returns (pred, resid or None)."""
        dev = xb.buf.device
        n = xb.n
        cb = hip.axpby(mv_hat.channels(0, 2).crop(hh, ww), flow_ab.crop(hh, ww), out=T.empty(n, hh, ww, 2, dev))
        ca = hip.axpby(mv_hat.channels(2, 4).crop(hh, ww), flow_ba.crop(hh, ww), out=T.empty(n, hh, ww, 2, dev))
        cb_up = hip.upsample_bilinear(cb, 4)
        ca_up = hip.upsample_bilinear(ca, 4)
        fwbw = T.empty(n, xb.h, xb.w, 6, dev)
        hip.warp(hip.WARP_W1, xb, cb_up, out=fwbw.channels(0, 3))
        hip.warp(hip.WARP_W1, xa, ca_up, out=fwbw.channels(3, 6))
        mask = self.masknet.run(fwbw)
        if trace is not None:
            trace.update({"fwbw": fwbw, "mask": mask})
        pred = T.empty(n, xb.h, xb.w, 3, dev)
        resid = T.empty(n, xb.h, xb.w, 3, dev) if cur is not None else None
        hip.check(hip.lib().vc_lhbdc_blend(hip.stream(), fwbw.view(), mask.view(),
                                           cur.view() if cur is not None else hip.NULL_VIEW, pred.view(),
                                           resid.view() if resid is not None else hip.NULL_VIEW), "vc_lhbdc_blend")
        return pred, resid

    def forward_device(self, x_before, x_current, x_after, trace=None):
        """The whole B-frame path with NO host synchronisation (graph-capturable) for a batch of n
        independent frames: returns (x_hat NCHW [n,3,H,W], bits float64 device tensor [n, 4] =
        per frame (mv.y, mv.z, res.y, res.z)).
        ``trace`` (parity instrumentation of the tests / bench.py): a dict that receives the stage outputs -- "flows"
        (the four SPyNet fields ba, ab, cb, ca), "mv_hat", "mask", "pred", "resid" as T windows and "mv" / "res" = the
        compressors' traces (MeanScaleHyperprior.forward_t)."""
        _require_frames(x_before, x_current, x_after)
        xb_, xc_, xa_ = (frame_list(t) for t in (x_before, x_current, x_after))     # tensors or lists of frames
        n = _count(xc_)
        dev = xc_[0].device
        frames = {"b": xb_, "c": xc_, "a": xa_}
        # m.py:38-47 -- four SPyNet calls as one batch: ba, ab, cb, ca
        flows = self._flows(frames, [("b", "a"), ("a", "b"), ("c", "b"), ("c", "a")])
        pred_flows, hh, ww = self._pool_pad(flows.images(0, 2 * n), 0.5)     # [ba | ab]
        cur_flows, _, _ = self._pool_pad(flows.images(2 * n, 4 * n), 1.0)    # [cb | ca]
        flow_ba, flow_ab = pred_flows.images(0, n), pred_flows.images(n, 2 * n)
        flow_cb, flow_ca = cur_flows.images(0, n), cur_flows.images(n, 2 * n)
        diff = T.empty(n, flow_ba.h, flow_ba.w, 4, dev)
        hip.axpby(flow_cb, flow_ab, 1.0, -1.0, out=diff.channels(0, 2))      # m.py:52
        hip.axpby(flow_ca, flow_ba, 1.0, -1.0, out=diff.channels(2, 4))
        bits = BitCounter(dev, max_rows=4 * n)
        t_mv, t_res = ({}, {}) if trace is not None else (None, None)
        mv_hat = self.mv_compressor.forward_t(diff, bits, trace=t_mv)
        xb, xc, xa = hip.nchw_frames_to_nhwc(xb_), hip.nchw_frames_to_nhwc(xc_), hip.nchw_frames_to_nhwc(xa_)
        pred, resid = self._predict(xb, xa, mv_hat, flow_ab, flow_ba, hh, ww, cur=xc, trace=trace)
        res_hat = self.residual_compressor.forward_t(resid, bits, trace=t_res)
        if trace is not None:
            trace.update({"flows": flows, "diff": diff, "mv_hat": mv_hat, "pred": pred, "resid": resid, "mv": t_mv, "res": t_res})
        x_hat = hip.nhwc_to_nchw(hip.axpby(res_hat, pred))                   # m.py:71
        # counter rows were appended as mv:(y,z) per image, then res:(y,z) per image
        tot = bits.totals().view(2, n, 2).permute(1, 0, 2).reshape(n, 4)
        return x_hat, tot

    def forward(self, x_before, x_current, x_after, train=False):
        x_hat, tot = self.forward_device(x_before, x_current, x_after)
        n, _, h, w = x_hat.shape
        num_pixels = n * h * w
        size = tot.sum()
        rate = (size / num_pixels / 2.0).to(torch.float32)                   # m.py:96,98 (halved)
        if train:
            return x_hat, rate
        return x_hat, rate, float(size.item())


# ------------------------------------------------------------------------------------------------
# CLI functions (encode_B.py / decode_B.py)
# ------------------------------------------------------------------------------------------------
def normalize(tensor):
    return tensor / 255.0


def float_to_uint8(image):
    clip = np.clip(image, 0, 1) * 255.0
    return np.round(clip).astype(np.uint8).transpose(1, 2, 0)


def pad(im):
    """encode_B.py:49-56 (reflection pad bottom/right to a multiple of 64) on the device."""
    _require_cuda(im)
    t = hip.nchw_to_nhwc(im)
    return hip.nhwc_to_nchw(hip.avgpool_reflectpad(t, 1, 1.0, t.h + _pad64(t.h), t.w + _pad64(t.w)))


def ups(flow):
    """encode_B.py:66-68 / decode_B.py:57-59: bilinear x4 of an NCHW CUDA flow field (nn.Upsample(scale_factor=4))."""
    _require_cuda(flow)
    return hip.nhwc_to_nchw(hip.upsample_bilinear(hip.nchw_to_nhwc(flow), 4))


def process_frame(img, device="cuda"):
    x = np.ascontiguousarray(img.transpose(2, 0, 1))[None]
    x = normalize(torch.from_numpy(x).to(device).float())
    return pad(x)


def _cli_predictors(model, frames, n):
    """encode_B.py:74-79 / decode_B.py:65-70 INCLUDING the swapped assignment: both padded predictors
    end up equal to pad(flow_ab)."""
    flows = model._flows(frames, [("a", "b")])
    flow_ab, hh, ww = Model._pool_pad(flows, 0.5)
    return flow_ab, flow_ab, hh, ww


def encode_B(model, x_after, x_current, x_before, trace=None):
    """(mv_bits, res_bits) like encode_B.py:71-105 -- note the argument order.
    ``trace``: a dict that receives {"mv": {...}, "res": {...}} = the integers handed to the range coder."""
    for t in (x_after, x_current, x_before):
        _require_cuda(t)
    xb_, xc_, xa_ = (t.contiguous().float() for t in (x_before, x_current, x_after))
    n = xc_.shape[0]
    dev = xc_.device
    frames = {"b": xb_, "c": xc_, "a": xa_}
    flow_ba, flow_ab, hh, ww = _cli_predictors(model, frames, n)
    cur = model._flows(frames, [("c", "b"), ("c", "a")])
    cur_flows, _, _ = Model._pool_pad(cur, 1.0)
    flow_cb, flow_ca = cur_flows.images(0, n), cur_flows.images(n, 2 * n)
    diff = T.empty(n, flow_ab.h, flow_ab.w, 4, dev)
    hip.axpby(flow_cb, flow_ab, 1.0, -1.0, out=diff.channels(0, 2))
    hip.axpby(flow_ca, flow_ba, 1.0, -1.0, out=diff.channels(2, 4))
    mv_hat = model.mv_compressor.forward_t(diff, BitCounter(dev, max_rows=2 * n))
    t_mv, t_res = ({}, {}) if trace is not None else (None, None)
    strings, shape = model.mv_compressor.compress_t(diff, trace=t_mv)
    mv_bits = {"strings": strings, "shape": torch.Size(shape)}
    xb, xc, xa = hip.nchw_to_nhwc(xb_), hip.nchw_to_nhwc(xc_), hip.nchw_to_nhwc(xa_)
    _, resid = model._predict(xb, xa, mv_hat, flow_ab, flow_ba, hh, ww, cur=xc)
    strings, shape = model.residual_compressor.compress_t(resid, trace=t_res)
    if trace is not None:
        trace.update({"mv": t_mv, "res": t_res})
    return mv_bits, {"strings": strings, "shape": torch.Size(shape)}


def expected_latent_shapes(h, w):
    """Hyper-latent (z) shapes of the motion and the residual codec for an h x w frame (both already multiples of 64,
    as every caller pads): the motion codec sees the flows pooled by 4 and re-padded to a multiple of 64 (m.py:38-47)."""
    hp, wp = h // 4, w // 4
    return ((hp + _pad64(hp)) // 64, (wp + _pad64(wp)) // 64), (h // 64, w // 64)


def check_container_shapes(shape_flow, shape_res, h, w):
    """The uint16 shapes of a bits_B header size every buffer of the decoder (index tensors, latents) BEFORE the range
    decoder can notice a corrupt string: a hostile 65535 x 65535 would ask for tens of GB.  The shapes are a function of
    the frame size alone, so anything else is refused here."""
    want_mv, want_res = expected_latent_shapes(h, w)
    got_mv, got_res = tuple(int(v) for v in shape_flow), tuple(int(v) for v in shape_res)
    if got_mv != want_mv or got_res != want_res:
        raise hip.VcError(f"bits_B container: latent shapes {got_mv} / {got_res} do not belong to a {h}x{w} frame "
                          f"(expected {want_mv} / {want_res})")


def decode_B(x_before, x_after, model, string_flow, string_res, shape_flow, shape_res, trace=None):
    """decode_B.py:63-86.  ``trace``: a dict that receives {"mv": {...}, "res": {...}} = the decoder's integers."""
    for t in (x_before, x_after):
        _require_cuda(t)
    check_container_shapes(shape_flow, shape_res, x_before.shape[-2], x_before.shape[-1])
    xb_, xa_ = x_before.contiguous().float(), x_after.contiguous().float()
    n = xb_.shape[0]
    dev = xb_.device
    flow_ba, flow_ab, hh, ww = _cli_predictors(model, {"b": xb_, "a": xa_}, n)
    t_mv, t_res = ({}, {}) if trace is not None else (None, None)
    if trace is not None:      # (cross-platform diagnostic, see MeanScaleHyperprior.decompress_t)
        t_mv["y_idx_override"] = trace.get("y_idx_override", {}).get("mv")
        t_res["y_idx_override"] = trace.get("y_idx_override", {}).get("res")
        trace.update({"mv": t_mv, "res": t_res})          # (filled as decoding proceeds: still there if a string is refused)
    mv_hat = model.mv_compressor.decompress_t(string_flow, shape_flow, dev, trace=t_mv)
    xb, xa = hip.nchw_to_nhwc(xb_), hip.nchw_to_nhwc(xa_)
    pred, _ = model._predict(xb, xa, mv_hat, flow_ab, flow_ba, hh, ww)
    res_hat = model.residual_compressor.decompress_t(string_res, shape_res, dev, trace=t_res)
    if trace is not None:
        trace.update({"mv": t_mv, "res": t_res})
    return hip.nhwc_to_nchw(hip.axpby(res_hat, pred))


def write_container(path_or_none, lmbda, mv_bits, res_bits):
    """bits_B.bin (encode_B.py:114-126): u32 lambda | u16x2 mv z-shape | u32 len(mv_y) | u32 len(mv_z) |
    u16x2 res z-shape | u32 len(res_y) | mv_y | mv_z | res_y | res_z (to EOF); little-endian."""
    out = bytearray()
    out += np.array(lmbda, dtype=np.uint32).tobytes()
    out += np.array(tuple(mv_bits["shape"]), dtype=np.uint16).tobytes()
    out += np.array(len(mv_bits["strings"][0][0]), dtype=np.uint32).tobytes()
    out += np.array(len(mv_bits["strings"][1][0]), dtype=np.uint32).tobytes()
    out += np.array(tuple(res_bits["shape"]), dtype=np.uint16).tobytes()
    out += np.array(len(res_bits["strings"][0][0]), dtype=np.uint32).tobytes()
    for s in (mv_bits["strings"][0][0], mv_bits["strings"][1][0], res_bits["strings"][0][0], res_bits["strings"][1][0]):
        out += s
    blob = bytes(out)
    if path_or_none is not None:
        with open(path_or_none, "wb") as f:
            f.write(blob)
    return blob


def read_container(blob):
    """decode_B.py:88-104."""
    if not isinstance(blob, (bytes, bytearray)):
        with open(blob, "rb") as f:
            blob = f.read()
    if len(blob) < 24:
        raise hip.VcError("truncated bits_B container")
    u32 = lambda o: int(np.frombuffer(blob[o:o + 4], dtype=np.uint32)[0])  # noqa: E731
    lmbda = u32(0)
    shape_mv = torch.Size(np.frombuffer(blob[4:8], dtype=np.uint16).astype(int).tolist())
    len0_mv, len1_mv = u32(8), u32(12)
    shape_res = torch.Size(np.frombuffer(blob[16:20], dtype=np.uint16).astype(int).tolist())
    len0_res = u32(20)
    p = 24
    if p + len0_mv + len1_mv + len0_res > len(blob):      # (the last string runs to the end of the file: decode_B.py:99)
        raise hip.VcError("bits_B container: the declared string lengths exceed the file")
    s0 = bytes(blob[p:p + len0_mv]); p += len0_mv
    s1 = bytes(blob[p:p + len1_mv]); p += len1_mv
    s2 = bytes(blob[p:p + len0_res]); p += len0_res
    s3 = bytes(blob[p:])
    return lmbda, [[s0], [s1]], [[s2], [s3]], shape_mv, shape_res
