"""ICIP2024 flow-guided deformable B-frame codec on MI355X -- host-side mirror of the reference surface.

Drop-in for (under /root/reference/ICIP2024/src):
  model/m.py:31-282                        FlowGuidedB                      -> :class:`FlowGuidedB`
  model/helpers.py:35-259                  OffsetDiversity, MS_Feature, FlowNET, OffsetTemproalEnc,
                                           ResidualTemproalEnc, Reconstuctor -> same names
  model/elic.py:69-83                      ResidualBottleneckBlock          -> same name
  model/layers.py:6-29                     CheckerboardContext              -> same name
  model/compression_bottlenecks.py:72-551  Offset_ELIC / Res_ELIC           -> same names
  model/elic.py:85-260                     ELIC intra codec (forward / rate estimate, via utils.image_compress) -> :class:`ELIC`
  opt_helpers.py:23-51                     prediction_flowonly, get_best_down_ratio_prediction
  utils.py:153-250                         select_references, update_buffer, get_order_typ_list, get_scales
Attribute names reproduce the reference's state_dict keys (1126 entries, tests/golden/icip2024_state_schema.txt),
including the never-executed g_a / g_s / context_prediction members the compressors inherit from CompressAI's
JointAutoregressiveHierarchicalPriors.  The modules only hold parameters; compute goes through libvc_hip.so.

torch.cat of the reference is never materialised at 1/2 and 1/4 resolution: producers write straight into channel
slices of the concat buffers, and a convolution over ``cat([a, b])`` whose operands live in different buffers runs
as two launches (weights split along the input channels, the second accumulating through the residual input).
The reference has no compress() for this model (rate = likelihood estimate only).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .hip import T
from .layers import (GDN, BitCounter, EntropyBottleneck, GaussianConditional, _Prepared, conv1x1, conv3x3, pack_conv,
                     run_sequential, subpel_conv3x3)
from .lhbdc import _require_cuda, _require_frames


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              output_padding=stride - 1, padding=kernel_size // 2)


class ResidualBottleneckBlock(_Prepared):
    """1x1 -> ReLU -> 3x3 -> ReLU -> 1x1, plus identity (elic.py:69-83); the add rides in the last epilogue."""
    vc_block = True

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.BottleneckBlock = nn.Sequential(conv1x1(in_ch, out_ch), nn.ReLU(inplace=True), conv3x3(out_ch, out_ch),
                                             nn.ReLU(inplace=True), conv1x1(out_ch, out_ch))

    def _pack(self):
        if self._packed is None:
            b = self.BottleneckBlock
            self._packed = (pack_conv(b[0]), pack_conv(b[2]), pack_conv(b[4]))
        return self._packed

    def half_stream_ok(self):
        """fp16 path: the block can take its input -- first layer's operand AND the identity -- as a half-precision tensor
        and hand its result on as one (hip.HALF_RESIDUAL; the streaming 1x1 kernel adds the half identity)."""
        c1, c2, c3 = self._pack()
        return hip.HALF_RESIDUAL and hip.HALF_ACTIVATIONS and c1.half_ok and c2.half_ok and c3.half_res_ok and c3.cout % 8 == 0

    def run(self, x, out=None, out_f16=False):
        c1, c2, c3 = self._pack()
        if x.dtype == "f16" and not self.half_stream_ok():
            raise hip.VcError("a half-precision tensor reached a bottleneck block that keeps its identity path in fp32")
        t = c1(x, act=hip.ACT_RELU, out_f16=c2.half_ok, out_sp3=hip.wants_split(c2, x))        # both intermediates feed one convolution each:
        if t.dtype == "f16" and c2.can_fuse_tail(c3) and (x.dtype == "f32" or self.half_stream_ok()):
            # fp16 path, 128 channels: the trailing 1x1 + identity in the 3x3 layer's epilogue (hip.FUSE_TAIL) -- the 3x3
            # layer's output (rounded to half exactly as it would be stored) never leaves the CU
            try:
                return c2(t, act=hip.ACT_RELU, tail=c3, res=x, out=out, out_f16=bool(out_f16 and out is None and self.half_stream_ok()))
            except hip.VcError as e:           # (views the fused launch cannot take -- unaligned slices: the two launches below can)
                if "VC_EINVAL" not in str(e):
                    raise
        t = c2(t, act=hip.ACT_RELU, out_f16=c3.half_ok)        # half-precision storage on the fp16 path
        return c3(t, res=x, out=out, out_f16=bool(out_f16 and out is None and self.half_stream_ok()))


def _rbb(c, n=3):
    return [ResidualBottleneckBlock(c, c) for _ in range(n)]


class CheckerboardContext(nn.Conv2d):
    """layers.py:6-29: 5x5 convolution with the weights masked to the (row+col) odd taps (mask is a buffer)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("mask", torch.zeros_like(self.weight.data))
        self.mask[:, :, 0::2, 1::2] = 1
        self.mask[:, :, 1::2, 0::2] = 1


class MaskedConv2d(nn.Conv2d):
    """compressai.layers.MaskedConv2d: parameter/buffer holder only (dead member of the parent class)."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.size()
        self.mask[:, :, h // 2, w // 2 + (mask_type == "B"):] = 0
        self.mask[:, :, h // 2 + 1:] = 0


class _Seq(_Prepared):
    """Owner of the packed-weight caches of its nn.Sequential members."""

    def __init__(self):
        super().__init__()
        self._caches = {}

    def _load_from_state_dict(self, *args, **kwargs):
        self._caches = {}
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._caches = {}
        return super()._apply(fn, *a, **k)

    def seq(self, name, x, **kw):
        return run_sequential(getattr(self, name), x, self._caches.setdefault(name, {}), **kw)

    def seq_cat(self, name, parts, **kw):
        """``getattr(self, name)(torch.cat(parts, 1))`` without the cat: the first layer (a Conv2d not followed
        by an activation in every use of the reference) runs once per part on its slice of the weights."""
        mods = list(getattr(self, name))
        first = mods[0]
        cache = self._caches.setdefault(name + ":split", {})
        key = tuple(p.c for p in parts)
        if key not in cache:
            if sum(key) != first.in_channels:
                raise hip.VcError(f"{name}: concat of {key} channels does not match the layer ({first.in_channels})")
            packs, c0 = [], 0
            for i, c in enumerate(key):
                packs.append(hip.PackedConv(first.weight[:, c0:c0 + c], first.bias if i == 0 else None,
                                            stride=first.stride[0], device=first.weight.device))
                c0 += c
            cache[key] = packs
        # (fp16 path: when a bottleneck chain follows, the LAST partial sum is the chain's input and leaves as half)
        half_tail = len(mods) > 1 and hasattr(mods[1], "half_stream_ok") and mods[1].half_stream_ok()
        acc = None
        for i, (pk, part) in enumerate(zip(cache[key], parts)):
            acc = pk(part, res=acc, out_f16=bool(half_tail and i == len(parts) - 1))
        return run_sequential(nn.Sequential(*mods[1:]), acc, self._caches.setdefault(name + ":tail", {}), **kw)


# ------------------------------------------------------------------------------------------------
# helpers.py
# ------------------------------------------------------------------------------------------------
class DeformConv2d(_Prepared):
    """Parameter holder with torchvision's names; executed by vc_offset_diversity / vc_deform_conv2d."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, groups=1):
        super().__init__()
        if kernel_size != 3 or padding != 1:
            raise hip.VcError("only DeformConv2d(kernel_size=3, padding=1) is on the path")
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -0.1, 0.1)

    def packed(self):
        if self._packed is None:
            self._packed = hip.PackedDeform(self.weight, self.bias, self.groups, self.weight.device)
        return self._packed


class OffsetDiversity(nn.Module):
    def __init__(self, in_channel, magnitude):
        super().__init__()
        self.in_channel, self.magnitude = in_channel, magnitude
        self.fusion = DeformConv2d(in_channel * 2, in_channel, kernel_size=3, padding=1, groups=2 * 8)

    def run(self, x1, offset1, flow1, x2, offset2, flow2, out=None):
        """helpers.py:43-58 in one launch (offset preparation fused into the gather)."""
        return self.fusion.packed().offset_diversity(x1, offset1, flow1, x2, offset2, flow2, self.magnitude, out=out)


class MS_Feature(_Seq):
    def __init__(self):
        super().__init__()
        self.layer1 = nn.Sequential(conv(3, 64, kernel_size=3, stride=2), *_rbb(64))
        self.layer2 = nn.Sequential(conv(64, 96, kernel_size=3, stride=2), *_rbb(96))
        self.layer3 = nn.Sequential(conv(96, 128, kernel_size=3, stride=2), *_rbb(128))

    def run(self, x, outs=(None, None, None)):
        l1 = self.seq("layer1", x, out=outs[0])
        l2 = self.seq("layer2", l1, out=outs[1])
        return l1, l2, self.seq("layer3", l2, out=outs[2])


class FlowNET(_Seq):
    def __init__(self):
        super().__init__()
        self.down0 = nn.Sequential(conv(6, 32, kernel_size=3, stride=2), *_rbb(32, 2))
        self.down1 = nn.Sequential(conv(32, 64, kernel_size=3, stride=2), *_rbb(64, 2))
        self.down2 = nn.Sequential(conv(64, 128, kernel_size=3, stride=2), *_rbb(128, 2))
        self.down3 = nn.Sequential(conv(128, 192, kernel_size=3, stride=2), *_rbb(192, 2))
        self.up0 = nn.Sequential(*_rbb(192, 2), subpel_conv3x3(192, 128, 2))
        self.up1 = nn.Sequential(conv(256, 128, 1, 1), *_rbb(128, 2), subpel_conv3x3(128, 64, 2))
        self.up2 = nn.Sequential(conv(128, 64, 1, 1), *_rbb(64, 2), subpel_conv3x3(64, 32, 2))
        self.up3 = nn.Sequential(conv(64, 32, 1, 1), *_rbb(32, 2), subpel_conv3x3(32, 4, 2))

    def run(self, inp):
        """helpers.py:163-172; every skip tensor is written next to the slot the up-path fills (no cat)."""
        dev, n = inp.buf.device, inp.n
        if inp.h % 16 or inp.w % 16:
            raise hip.VcError("FlowNET input must be a multiple of 16 (FlowGuidedB.pad_flow)")
        cat3 = T.empty(n, inp.h // 2, inp.w // 2, 64, dev)       # [up2 out | s0]
        cat2 = T.empty(n, inp.h // 4, inp.w // 4, 128, dev)      # [up1 out | s1]
        cat1 = T.empty(n, inp.h // 8, inp.w // 8, 256, dev)      # [up0 out | s2]
        s0 = self.seq("down0", inp, out=cat3.channels(32, 64))
        s1 = self.seq("down1", s0, out=cat2.channels(64, 128))
        s2 = self.seq("down2", s1, out=cat1.channels(128, 256))
        s3 = self.seq("down3", s2)
        self.seq("up0", s3, out=cat1.channels(0, 128))
        self.seq("up1", cat1, out=cat2.channels(0, 64))
        self.seq("up2", cat2, out=cat3.channels(0, 32))
        return self.seq("up3", cat3)


class _TemporalEnc(_Seq):
    def __init__(self, mult, N=128, M=128):
        super().__init__()
        self.g_a1 = nn.Sequential(conv(64 * mult, N, kernel_size=5, stride=2), *_rbb(N))
        self.g_a2 = nn.Sequential(conv(N + 96 * mult, N, kernel_size=5, stride=2), *_rbb(N))
        self.g_a3 = nn.Sequential(conv(N + 128 * mult, M, kernel_size=5, stride=2), *_rbb(M))

    def run(self, l1, l2, l3, out=None):
        y = self.seq("g_a1", l1)
        y = self.seq_cat("g_a2", [y, l2])
        return self.seq_cat("g_a3", [y, l3], out=out)


class OffsetTemproalEnc(_TemporalEnc):
    def __init__(self, N=128, M=128):
        super().__init__(4, N, M)


class ResidualTemproalEnc(_TemporalEnc):
    def __init__(self, N=128, M=128):
        super().__init__(1, N, M)


class Reconstuctor(_Seq):
    def __init__(self):
        super().__init__()
        self.layer3 = nn.Sequential(*_rbb(128), subpel_conv3x3(128, 128, 2))
        self.layer2 = nn.Sequential(conv(128 + 96, 96, 1, 1), *_rbb(96), subpel_conv3x3(96, 96, 2))
        self.layer1 = nn.Sequential(conv(96 + 64, 64, 1, 1), *_rbb(64), subpel_conv3x3(64, 3, 2))

    def run(self, x_comp_l1, x_comp_l2, x_comp_l3):
        l3 = self.seq("layer3", x_comp_l3)
        l2 = self.seq_cat("layer2", [x_comp_l2, l3])
        return self.seq_cat("layer1", [x_comp_l1, l2])


# ------------------------------------------------------------------------------------------------
# compression_bottlenecks.py
# ------------------------------------------------------------------------------------------------
GROUPS = (0, 6, 12, 24, 48)


class JointAutoregressiveHierarchicalPriors(_Seq):
    """Members (and therefore state_dict keys) of compressai.models.JointAutoregressiveHierarchicalPriors(N, M);
    the ICIP2024 subclasses replace h_a / h_s / entropy_parameters and never run g_a / g_s / context_prediction."""

    def __init__(self, N=192, M=192):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = nn.Sequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = nn.Sequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True),
                                 deconv(N, N), GDN(N, inverse=True), deconv(N, 3))
        self.h_a = nn.Sequential()
        self.h_s = nn.Sequential()
        self.entropy_parameters = nn.Sequential()
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = int(N), int(M)

    def update(self, scale_table=None, force=False):
        """compressai JointAutoregressiveHierarchicalPriors.update: range-coder tables of both entropy models."""
        from .layers import get_scale_table
        updated = self.gaussian_conditional.update_scale_table(get_scale_table() if scale_table is None else scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        return updated


class _Elic(JointAutoregressiveHierarchicalPriors):
    """Body shared by Offset_ELIC (enc_mult 5, dec_mult 4, three 432-channel offset heads) and Res_ELIC
    (enc_mult 2, dec_mult 1, residual heads of 64/96/128 channels)."""

    def __init__(self, enc_mult, dec_mult, outs, N=128, M=128):
        super().__init__(N, M)
        self.g_a1 = nn.Sequential(conv(64 * enc_mult, N, kernel_size=5, stride=2), *_rbb(N))
        self.g_a2 = nn.Sequential(conv(N + 96 * enc_mult, N, kernel_size=5, stride=2), *_rbb(N))
        self.g_a3 = nn.Sequential(conv(N + 128 * enc_mult, M, kernel_size=5, stride=2), *_rbb(M))
        self.g_s3 = nn.Sequential(*_rbb(M), deconv(M, N, kernel_size=5, stride=2))
        self.g_o3 = nn.Sequential(conv(N + 128 * dec_mult, N, kernel_size=3, stride=1), *_rbb(N),
                                  conv(N, outs[2], kernel_size=3, stride=1))
        self.g_s2 = nn.Sequential(conv(N + 128 * dec_mult, N, kernel_size=1, stride=1), *_rbb(N),
                                  deconv(N, N, kernel_size=5, stride=2))
        self.g_o2 = nn.Sequential(conv(N + 96 * dec_mult, N, kernel_size=3, stride=1), *_rbb(N),
                                  conv(N, outs[1], kernel_size=3, stride=1))
        self.g_s1 = nn.Sequential(conv(N + 96 * dec_mult, N, kernel_size=1, stride=1), *_rbb(N),
                                  deconv(N, N, kernel_size=5, stride=2))
        self.g_o1 = nn.Sequential(conv(N + 64 * dec_mult, N, kernel_size=3, stride=1), *_rbb(N),
                                  conv(N, outs[0], kernel_size=3, stride=1))
        self.h_a = nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.ReLU(inplace=True),
                                 conv(N, N, stride=2, kernel_size=5), nn.ReLU(inplace=True),
                                 conv(N, N, stride=2, kernel_size=5))
        self.h_s = nn.Sequential(deconv(N, M, stride=2, kernel_size=5), nn.ReLU(inplace=True),
                                 deconv(M, M, stride=2, kernel_size=5), nn.ReLU(inplace=True),
                                 conv(M, M, stride=1, kernel_size=3))
        self.prior_fusion = nn.Sequential(conv(2 * M, 2 * M, stride=1, kernel_size=3), *_rbb(2 * M),
                                          conv(M * 2, M * 2, stride=1, kernel_size=3))
        self.entropy_parameters = nn.ModuleList(
            nn.Sequential(nn.Conv2d(cin, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 8 // 3, cout * 6 // 3, 1))
            for cin, cout in [(M * 4, 6), (M * 6, 6), (M * 6, 12), (M * 6, 24), (M * 6, M - 48)])
        self.channel_context_models = nn.ModuleList(
            nn.Sequential(conv(cin, N, kernel_size=5, stride=1), nn.ReLU(inplace=True),
                          conv(N, N, kernel_size=5, stride=1), nn.ReLU(inplace=True),
                          conv(N, M * 2, kernel_size=5, stride=1)) for cin in [6, 12, 24, 48])
        self.context_prediction_models = nn.ModuleList(
            CheckerboardContext(in_channels=cin, out_channels=M * 2, kernel_size=5, stride=1, padding=2)
            for cin in [6, 6, 12, 24, M - 48])
        self.levels = 5
        self.Gain = nn.Parameter(torch.ones(self.levels, M))
        self.InverseGain = nn.Parameter(torch.ones(self.levels, M))
        self.HyperGain = nn.Parameter(torch.ones(self.levels, N))
        self.InverseHyperGain = nn.Parameter(torch.ones(self.levels, N))
        self._gain_cache = {}

    def _load_from_state_dict(self, *args, **kwargs):
        self._gain_cache = {}
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._gain_cache = {}
        return super()._apply(fn, *a, **k)

    def interpolate_gain(self, s):
        """compression_bottlenecks.py:296-318, evaluated on the host once per quality level (cached), fp32 like
        the reference; returns device vectors (gain, hypergain, invhypergain, invgain)."""
        s = max(min(s, self.levels - 1), 0)
        if s not in self._gain_cache:
            upper = int(min(math.ceil(s), self.levels - 1))
            lower = int(max(math.floor(s), 0))
            out = []
            with torch.no_grad():
                for m in (self.Gain, self.HyperGain, self.InverseHyperGain, self.InverseGain):
                    m = m.detach().to("cpu", torch.float32)
                    if upper == lower:
                        v = torch.abs(m[int(s)])
                    else:
                        l = upper - s
                        v = torch.abs(m[upper]) ** (1 - l) * torch.abs(m[lower]) ** l
                    out.append(v.contiguous().to(self.Gain.device))
            self._gain_cache[s] = tuple(out)
        return self._gain_cache[s]

    def _sub(self, name, i, x, **kw):
        seq = getattr(self, name)[i]
        return run_sequential(seq, x, self._caches.setdefault(f"{name}.{i}", {}), **kw)

    def _ctx_conv(self, i):
        key = f"ctx.{i}"
        if key not in self._caches:
            self._caches[key] = pack_conv(self.context_prediction_models[i])
        return self._caches[key]

    def code(self, enc1, enc2, enc3, f1d, f2d, f3d, temporal_into, s, bits, res=(None, None, None), trace=None):
        """Encoder, entropy model and the three decoder heads.

        enc1/enc2/enc3: lists of views whose concatenation the reference feeds to g_a1 / (after y) g_a2 / g_a3;
        f*d: decoder-side conditioning views; temporal_into(view): callback that makes the temporal conditioner
        write its output into the given slice of the prior-fusion input; res: optional views added to the three
        heads' outputs (Res_ELIC: x_comp + res fused into the last convolution).  Returns (head1, head2, head3).
        ``trace``: a dict that receives the gained latents "y" / "z" (what the quantiser rounds) and "heads".
        Appends 6 x n rows to ``bits`` (n = batch size): z of every image, then y_0 .. y_4 of every image."""
        L = hip.lib()
        M = self.M
        gain, hypergain, invhypergain, invgain = self.interpolate_gain(s)
        y = self.seq_cat("g_a1", enc1) if len(enc1) > 1 else self.seq("g_a1", enc1[0])
        y = self.seq_cat("g_a2", [y] + enc2)
        y = self.seq_cat("g_a3", [y] + enc3)
        dev, n, h, w = y.buf.device, y.n, y.h, y.w
        y = hip.channel_scale(y, gain, out=y)
        z = self.seq("h_a", y, final_chscale=hypergain)
        for i in range(n):                                           # one counter row per image and tensor
            hip.check(L.vc_eb_forward(hip.stream(), z.images(i, i + 1).view(), self.entropy_bottleneck.device_params().data_ptr(),
                                      None, None, hip.NULL_VIEW, None, bits.next_row_ptr(), bits.slots, None), "vc_eb_forward")
        z_hat = hip.quantize_mask(z, gain=invhypergain)
        fusion_in = T.empty(n, h, w, 2 * M, dev)                       # [h_s(z_hat) | temporal condition]
        self.seq("h_s", z_hat, out=fusion_in.channels(0, M))
        temporal_into(fusion_in.channels(M, 2 * M))
        params_in = T.empty(n, h, w, 6 * M, dev)                       # [ctx | channel ctx | hyper]
        hyper = self.seq("prior_fusion", fusion_in, out=params_in.channels(4 * M, 6 * M))
        params_in0 = T.empty(n, h, w, 4 * M, dev)                      # group 0 has no channel context: [ctx | hyper]
        hip.axpby(hyper, None, out=params_in0.channels(2 * M, 4 * M))
        if trace is not None:
            trace.update({"y": y, "z": z})
        y_round = hip.quantize_mask(y)                                 # ste_round(y)
        y_half = hip.quantize_mask(y, keep_parity=1)                   # anchors only (non-anchors zeroed)
        bounds = GROUPS + (M,)
        for i in range(5):
            c0, c1 = bounds[i], bounds[i + 1]
            pin = params_in0 if i == 0 else params_in
            ctx = self._ctx_conv(i)(y_half.channels(c0, c1), out=pin.channels(0, 2 * M))
            hip.quantize_mask(ctx, out=ctx, keep_parity=0, do_round=False)
            if i > 0:
                self._sub("channel_context_models", i - 1, y_round.channels(0, c0), out=pin.channels(2 * M, 4 * M))
            gp = self._sub("entropy_parameters", i, pin)
            half = (c1 - c0)
            for j in range(n):
                hip.check(L.vc_gc_forward(hip.stream(), y.channels(c0, c1).images(j, j + 1).view(),
                                          gp.channels(0, half).images(j, j + 1).view(),
                                          gp.channels(half, 2 * half).images(j, j + 1).view(), None, None, hip.NULL_VIEW,
                                          bits.next_row_ptr(), bits.slots, None, None, None, None, 0, None), "vc_gc_forward")
        y_hat = hip.quantize_mask(y, gain=invgain)
        xhat3 = self.seq("g_s3", y_hat)
        head3 = self._head("g_o3", [xhat3, f3d], res[2])
        xhat2 = self.seq_cat("g_s2", [xhat3, f3d])
        head2 = self._head("g_o2", [xhat2, f2d], res[1])
        xhat1 = self.seq_cat("g_s1", [xhat2, f2d])
        head1 = self._head("g_o1", [xhat1, f1d], res[0])
        if trace is not None:
            trace["heads"] = (head1, head2, head3)
        return head1, head2, head3

    def _head(self, name, parts, res):
        t = self.seq_cat(name, parts)
        if res is None:
            return t
        return hip.axpby(t, res, out=t)


class Offset_ELIC(_Elic):
    def __init__(self, N=128, M=128, **kwargs):
        super().__init__(5, 4, (27 * 8 * 2,) * 3, N, M)


class Res_ELIC(_Elic):
    def __init__(self, N=128, M=128, **kwargs):
        super().__init__(2, 1, (64, 96, 128), N, M)


# ------------------------------------------------------------------------------------------------
# elic.py: the intra codec of the ICIP2024 test loop
# ------------------------------------------------------------------------------------------------
class _ResidualUnit(_Prepared):
    """compressai AttentionBlock.ResidualUnit: relu(conv1x1 -> ReLU -> conv3x3 -> ReLU -> conv1x1 + x)."""

    def __init__(self, N):
        super().__init__()
        self.conv = nn.Sequential(conv1x1(N, N // 2), nn.ReLU(inplace=True), conv3x3(N // 2, N // 2), nn.ReLU(inplace=True),
                                  conv1x1(N // 2, N))
        self.relu = nn.ReLU(inplace=True)

    def run(self, x):
        if self._packed is None:
            self._packed = (pack_conv(self.conv[0]), pack_conv(self.conv[2]), pack_conv(self.conv[4]))
        c1, c2, c3 = self._packed
        t = c1(x, act=hip.ACT_RELU, out_f16=c2.half_ok, out_sp3=hip.wants_split(c2, x))
        t = c2(t, act=hip.ACT_RELU, out_f16=c3.half_ok)
        return c3(t, act=hip.ACT_RELU, res=x, res_first=True)        # the ReLU comes AFTER the skip addition


class AttentionBlock(_Prepared):
    """compressai.layers.AttentionBlock: conv_a(x) * sigmoid(conv_b(x)) + x (elic.py:97-121 uses it four times)."""
    vc_block = True

    def __init__(self, N):
        super().__init__()
        self.conv_a = nn.Sequential(_ResidualUnit(N), _ResidualUnit(N), _ResidualUnit(N))
        self.conv_b = nn.Sequential(_ResidualUnit(N), _ResidualUnit(N), _ResidualUnit(N), conv1x1(N, N))

    def run(self, x, out=None):
        if self._packed is None:
            self._packed = pack_conv(self.conv_b[3])
        a = b = x
        for u in self.conv_a:
            a = u.run(a)
        for u in list(self.conv_b)[:3]:
            b = u.run(b)
        return hip.attention_gate(a, self._packed(b), x, out=out)


ELIC_GROUPS = (0, 16, 32, 64, 128)


class ELIC(JointAutoregressiveHierarchicalPriors):
    """elic.py:85-596: forward (rate estimate, what utils.image_compress calls for the I-frames), forward_stage2, and the
    two-pass checkerboard bitstream codec compress / decompress."""

    def __init__(self, N=192, M=320, **kwargs):
        super().__init__(N, M)
        self.g_a = nn.Sequential(conv(3, N), *_rbb(N), conv(N, N), *_rbb(N), AttentionBlock(N), conv(N, N), *_rbb(N),
                                 conv(N, M), AttentionBlock(M))
        self.g_s = nn.Sequential(AttentionBlock(M), deconv(M, N), *_rbb(N), deconv(N, N), AttentionBlock(N), *_rbb(N),
                                 deconv(N, N), *_rbb(N), deconv(N, 3))
        self.h_a = nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.ReLU(inplace=True), conv(N, N), nn.ReLU(inplace=True),
                                 conv(N, N))
        self.h_s = nn.Sequential(deconv(N, M), nn.ReLU(inplace=True), deconv(M, M * 3 // 2), nn.ReLU(inplace=True),
                                 conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))
        self.entropy_parameters = nn.ModuleList(
            nn.Sequential(nn.Conv2d(cin, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 8 // 3, cout * 6 // 3, 1))
            for cin, cout in [(M * 4, 16), (M * 6, 16), (M * 6, 32), (M * 6, 64), (M * 6, M - 128)])
        self.channel_context_models = nn.ModuleList(
            nn.Sequential(conv(cin, N, kernel_size=5, stride=1), nn.ReLU(inplace=True),
                          conv(N, N, kernel_size=5, stride=1), nn.ReLU(inplace=True),
                          conv(N, M * 2, kernel_size=5, stride=1)) for cin in [16, 32, 64, 128])
        self.context_prediction_models = nn.ModuleList(
            CheckerboardContext(in_channels=cin, out_channels=M * 2, kernel_size=5, stride=1, padding=2)
            for cin in [16, 16, 32, 64, M - 128])

    def _sub(self, name, i, x, **kw):
        return run_sequential(getattr(self, name)[i], x, self._caches.setdefault(f"{name}.{i}", {}), **kw)

    def forward_device(self, x, bits, stage2=False):
        """x: T [n,H,W,3] (H, W multiples of 64) -> x_hat T; appends 6 x n rows to ``bits`` (z, y_0..y_4 per image).
        ``stage2`` = forward_stage2 (elic.py:247-305): the channel context sees round(round(y - mu) + mu) of the groups
        already processed and the synthesis their round(y - mu) + mu, instead of round(y)."""
        L, M = hip.lib(), self.M
        y = self.seq("g_a", x)
        z = self.seq("h_a", y)
        dev, n, h, w = y.buf.device, y.n, y.h, y.w
        for i in range(n):
            hip.check(L.vc_eb_forward(hip.stream(), z.images(i, i + 1).view(), self.entropy_bottleneck.device_params().data_ptr(),
                                      None, None, hip.NULL_VIEW, None, bits.next_row_ptr(), bits.slots, None), "vc_eb_forward")
        params_in = T.empty(n, h, w, 6 * M, dev)                       # [ctx | channel ctx | hyper]
        hyper = self.seq("h_s", hip.quantize_mask(z), out=params_in.channels(4 * M, 6 * M))
        params_in0 = T.empty(n, h, w, 4 * M, dev)                      # group 0: [ctx | hyper]
        hip.axpby(hyper, None, out=params_in0.channels(2 * M, 4 * M))
        y_round = hip.quantize_mask(y)
        y_half = hip.quantize_mask(y, keep_parity=1)
        y_hat = T.empty(n, h, w, M, dev) if stage2 else None            # round(y - mu) + mu, group by group
        bounds = ELIC_GROUPS + (M,)
        for i in range(5):
            c0, c1 = bounds[i], bounds[i + 1]
            pin = params_in0 if i == 0 else params_in
            ctx = self._ctx_conv(i)(y_half.channels(c0, c1), out=pin.channels(0, 2 * M))
            hip.quantize_mask(ctx, out=ctx, keep_parity=0, do_round=False)
            if i > 0:
                prev = hip.quantize_mask(y_hat.channels(0, c0)) if stage2 else y_round.channels(0, c0)
                self._sub("channel_context_models", i - 1, prev, out=pin.channels(2 * M, 4 * M))
            gp = self._sub("entropy_parameters", i, pin)
            half = c1 - c0
            for j in range(n):
                hip.check(L.vc_gc_forward(hip.stream(), y.channels(c0, c1).images(j, j + 1).view(),
                                          gp.channels(0, half).images(j, j + 1).view(),
                                          gp.channels(half, 2 * half).images(j, j + 1).view(), None, None,
                                          y_hat.channels(c0, c1).images(j, j + 1).view() if stage2 else hip.NULL_VIEW,
                                          bits.next_row_ptr(), bits.slots, None, None, None, None, 0, None), "vc_gc_forward")
        return self.seq("g_s", y_hat if stage2 else y_round)

    def _ctx_conv(self, i):
        key = f"ctx.{i}"
        if key not in self._caches:
            self._caches[key] = pack_conv(self.context_prediction_models[i])
        return self._caches[key]

    def forward_stage2(self, x):
        """elic.py:247-305 -- same return structure as forward."""
        _require_frames(x)
        bits = BitCounter(x.device, max_rows=6 * x.shape[0])
        x_hat = hip.nhwc_to_nchw(self.forward_device(hip.nchw_to_nhwc(x), bits, stage2=True))
        return {"x_hat": x_hat, "size": bits.totals().sum().float()}

    # ---- real bitstream (elic.py:307-496) --------------------------------------------------------------------------
    # One string for the hyper-latents + ONE string per channel group: the anchor symbols (checkerboard positions with
    # (row + col) odd, coded against the hyper / channel context only) followed by the non-anchor symbols (coded with the
    # checkerboard context of the just-decoded anchors).  The networks run on the device; between the two passes of a
    # group the integers cross to the host, where the squeezed symbol order of the format is assembled and range-coded.
    @staticmethod
    def _squeeze_np(t, anchor):
        """ckbd_anchor_sequeeze / ckbd_nonanchor_sequeeze (elic.py:498-512) on a [n,c,h,w] array."""
        out = np.empty(t.shape[:3] + (t.shape[3] // 2,), dtype=t.dtype)
        if anchor:
            out[:, :, 0::2, :] = t[:, :, 0::2, 1::2]
            out[:, :, 1::2, :] = t[:, :, 1::2, 0::2]
        else:
            out[:, :, 0::2, :] = t[:, :, 0::2, 0::2]
            out[:, :, 1::2, :] = t[:, :, 1::2, 1::2]
        return out

    @staticmethod
    def _unsqueeze_np(t, anchor):
        out = np.zeros(t.shape[:3] + (t.shape[3] * 2,), dtype=t.dtype)
        if anchor:
            out[:, :, 0::2, 1::2] = t[:, :, 0::2, :]
            out[:, :, 1::2, 0::2] = t[:, :, 1::2, :]
        else:
            out[:, :, 0::2, 0::2] = t[:, :, 0::2, :]
            out[:, :, 1::2, 1::2] = t[:, :, 1::2, :]
        return out

    def _param_buffers(self, hyper, n, h, w, dev):
        """Inputs of the entropy-parameter networks, [ctx | (channel ctx) | hyper]: one pair with the checkerboard context
        held at zero (anchor pass) and one that receives it (non-anchor pass); group 0 has no channel context."""
        M = self.M
        bufs = {}
        for tag in ("a0", "n0"):
            t = T.empty(n, h, w, 4 * M, dev)
            hip.axpby(hyper, None, out=t.channels(2 * M, 4 * M))
            bufs[tag] = t
        for tag in ("a", "n"):
            t = T.empty(n, h, w, 6 * M, dev)
            hip.axpby(hyper, None, out=t.channels(4 * M, 6 * M))
            bufs[tag] = t
        for tag in ("a0", "a"):
            zero = bufs[tag].channels(0, 2 * M)
            hip.axpby(hyper, None, alpha=0.0, out=zero)          # hyper is finite: 0 * hyper = 0
        return bufs

    def _scale_table_dev(self):
        gc = self.gaussian_conditional
        if gc._packed is None:
            if gc.scale_table.numel() == 0:
                raise hip.VcError("scale table is empty: call update(force=True) after loading weights")
            gc._packed = gc.scale_table.detach().float().contiguous().to(gc.scale_bound.device)
        return gc._packed

    def compress(self, x):
        """{"strings": [[[g0], [g1], [g2], [g3], [g4]], [z]], "shape", "y_hat": 5 NCHW tensors} like elic.py:307-414."""
        _require_frames(x)
        L, M = hip.lib(), self.M
        xt = hip.nchw_to_nhwc(x.contiguous().float())
        y = self.seq("g_a", xt)
        z = self.seq("h_a", y)
        dev, n, h, w = y.buf.device, y.n, y.h, y.w
        eb_cdf, eb_len, eb_off = self.entropy_bottleneck.tables()
        gc_tables = self.gaussian_conditional.tables()
        table = self._scale_table_dev()
        z_hat = T.empty(z.n, z.h, z.w, z.c, dev)
        z_sym = torch.empty((z.n, z.c * z.h * z.w), dtype=torch.int32, device=dev)
        hip.check(L.vc_eb_forward(hip.stream(), z.view(), self.entropy_bottleneck.device_params().data_ptr(), None, None,
                                  z_hat.view(), z_sym.data_ptr(), None, 0, None), "vc_eb_forward")
        z_index = np.repeat(np.arange(z.c, dtype=np.int32), z.h * z.w)
        z_strings = [hip.rans_encode(row, z_index, eb_cdf, eb_len, eb_off) for row in z_sym.cpu().numpy()]
        hyper = self.seq("h_s", z_hat)
        bufs = self._param_buffers(hyper, n, h, w, dev)
        y_hat = T.empty(n, h, w, M, dev)
        bounds = ELIC_GROUPS + (M,)
        strings = []
        for i in range(5):
            c0, c1 = bounds[i], bounds[i + 1]
            cg = c1 - c0
            pa, pn = (bufs["a0"], bufs["n0"]) if i == 0 else (bufs["a"], bufs["n"])
            if i > 0:
                self._sub("channel_context_models", i - 1, y_hat.channels(0, c0), out=pn.channels(2 * M, 4 * M))
                hip.axpby(pn.channels(2 * M, 4 * M), None, out=pa.channels(2 * M, 4 * M))
            syms, idxs, parts = [], [], []
            for anchor in (True, False):
                pin = pa if anchor else pn
                if not anchor:
                    self._ctx_conv(i)(parts[0], out=pin.channels(0, 2 * M))
                gp = self._sub("entropy_parameters", i, pin)
                full = T.empty(n, h, w, cg, dev)
                sym = torch.empty((n, cg, h, w), dtype=torch.int32, device=dev)
                idx = torch.empty_like(sym)
                hip.check(L.vc_gc_forward(hip.stream(), y.channels(c0, c1).view(), gp.channels(0, cg).view(), gp.channels(cg, 2 * cg).view(),
                                          None, None, full.view(), None, 0, None, sym.data_ptr(), idx.data_ptr(), table.data_ptr(),
                                          table.numel(), None), "vc_gc_forward")
                parts.append(hip.quantize_mask(full, keep_parity=1 if anchor else 0, do_round=False))
                syms.append(self._squeeze_np(sym.cpu().numpy(), anchor).reshape(-1))
                idxs.append(self._squeeze_np(idx.cpu().numpy(), anchor).reshape(-1))
            hip.axpby(parts[0], parts[1], out=y_hat.channels(c0, c1))
            strings.append([hip.rans_encode(np.concatenate(syms), np.concatenate(idxs), *gc_tables)])
        groups = [hip.nhwc_to_nchw(y_hat.channels(bounds[i], bounds[i + 1])) for i in range(5)]
        return {"strings": [strings, z_strings], "shape": torch.Size([z.h, z.w]), "y_hat": groups}

    def decompress(self, strings, shape):
        """{"x_hat", "cost_time", "y_hat"} like elic.py:416-496 (two decode_stream calls per group string)."""
        import time
        t0 = time.process_time()
        L, M = hip.lib(), self.M
        dev = self.entropy_bottleneck.quantiles.device
        hz, wz = int(shape[0]), int(shape[1])
        n = len(strings[1])
        eb_cdf, eb_len, eb_off = self.entropy_bottleneck.tables()
        gc_tables = self.gaussian_conditional.tables()
        table = self._scale_table_dev()
        z_index = np.repeat(np.arange(self.N, dtype=np.int32), hz * wz)
        z_sym = np.stack([hip.rans_decode(strings[1][k], z_index, eb_cdf, eb_len, eb_off) for k in range(n)])
        z_hat = T.empty(n, hz, wz, self.N, dev)
        z_sym_d = torch.from_numpy(z_sym).to(dev)       # (bound to a local: must outlive the launch that reads it)
        hip.check(L.vc_eb_dequant(hip.stream(), z_sym_d.data_ptr(), self.entropy_bottleneck.device_params().data_ptr(),
                                  None, z_hat.view()), "vc_eb_dequant")
        hyper = self.seq("h_s", z_hat)
        h, w = hyper.h, hyper.w
        bufs = self._param_buffers(hyper, n, h, w, dev)
        y_hat = T.empty(n, h, w, M, dev)
        bounds = ELIC_GROUPS + (M,)
        for i in range(5):
            c0, c1 = bounds[i], bounds[i + 1]
            cg = c1 - c0
            pa, pn = (bufs["a0"], bufs["n0"]) if i == 0 else (bufs["a"], bufs["n"])
            if i > 0:
                self._sub("channel_context_models", i - 1, y_hat.channels(0, c0), out=pn.channels(2 * M, 4 * M))
                hip.axpby(pn.channels(2 * M, 4 * M), None, out=pa.channels(2 * M, 4 * M))
            dec = hip.RansStreamDecoder(strings[0][i][0])
            parts = []
            for anchor in (True, False):
                pin = pa if anchor else pn
                if not anchor:
                    self._ctx_conv(i)(parts[0], out=pin.channels(0, 2 * M))
                gp = self._sub("entropy_parameters", i, pin)
                idx = torch.empty((n, cg, h, w), dtype=torch.int32, device=dev)
                hip.check(L.vc_gc_indexes(hip.stream(), gp.channels(0, cg).view(), table.data_ptr(), table.numel(), idx.data_ptr()),
                          "vc_gc_indexes")
                sq_idx = self._squeeze_np(idx.cpu().numpy(), anchor)
                sym = dec.decode_stream(sq_idx.reshape(-1), *gc_tables).reshape(sq_idx.shape)
                sym_full = torch.from_numpy(self._unsqueeze_np(sym, anchor)).to(dev)
                full = T.empty(n, h, w, cg, dev)
                hip.check(L.vc_gc_dequant(hip.stream(), sym_full.data_ptr(), gp.channels(cg, 2 * cg).view(), None, full.view()),
                          "vc_gc_dequant")
                parts.append(hip.quantize_mask(full, keep_parity=1 if anchor else 0, do_round=False))
            hip.axpby(parts[0], parts[1], out=y_hat.channels(c0, c1))
        x_hat = hip.nhwc_to_nchw(self.seq("g_s", y_hat))
        torch.cuda.synchronize()
        groups = [hip.nhwc_to_nchw(y_hat.channels(bounds[i], bounds[i + 1])) for i in range(5)]
        return {"x_hat": x_hat, "cost_time": time.process_time() - t0, "y_hat": groups}

    def forward(self, x):
        """NCHW CUDA tensor in; ``{"x_hat", "size"}`` out (``size`` = the -log2 likelihood sum the reference's
        image_compress derives from the likelihood tensors, which are not materialised here)."""
        _require_frames(x)
        bits = BitCounter(x.device, max_rows=6 * x.shape[0])
        x_hat = hip.nhwc_to_nchw(self.forward_device(hip.nchw_to_nhwc(x), bits))
        return {"x_hat": x_hat, "size": bits.totals().sum().float()}


def image_compress(im, compressors, n):
    """utils.py:306-316: (reconstruction, estimated size in bits) of an intra-coded frame at quality index n."""
    out = compressors[n](im)
    return out["x_hat"], out["size"]


# ------------------------------------------------------------------------------------------------
# m.py
# ------------------------------------------------------------------------------------------------
def _f32_round2(v):
    """convert_scales (m.py:71-82): fp32 value rounded to two decimals, as a python float."""
    t = torch.tensor([v], dtype=torch.float32) if not torch.is_tensor(v) else v.reshape(-1)[:1].to("cpu", torch.float32)
    return float((torch.round(t * 10 ** 2) / (10 ** 2)).item())


class FlowGuidedB(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = MS_Feature()
        self.flow_estimator = FlowNET()
        self.offset_temporal_conditioner = OffsetTemproalEnc()
        self.offset_compressor = Offset_ELIC()
        self.offset_diversity_l3 = OffsetDiversity(in_channel=128, magnitude=10)
        self.offset_diversity_l2 = OffsetDiversity(in_channel=96, magnitude=20)
        self.offset_diversity_l1 = OffsetDiversity(in_channel=64, magnitude=40)
        self.residue_temporal_conditioner = ResidualTemproalEnc()
        self.residual_compressor = Res_ELIC()
        self.reconstructor = Reconstuctor()

    # -- reference helpers with tensor arguments -----------------------------------------------------------
    def convert_scales(self, scale1, scale2, x=None):
        return _f32_round2(scale1), _f32_round2(scale2)

    def warp(self, img, flow):
        """m.py:262-282 on NCHW CUDA tensors."""
        _require_cuda(img)
        _require_cuda(flow)
        return hip.nhwc_to_nchw(hip.warp(hip.WARP_W3, hip.nchw_to_nhwc(img), hip.nchw_to_nhwc(flow)))

    def estimate_flow(self, xref1, xref2, down_ratio):
        _require_cuda(xref1)
        _require_cuda(xref2)
        return hip.nhwc_to_nchw(self.estimate_flow_t(hip.nchw_to_nhwc(xref1), hip.nchw_to_nhwc(xref2), down_ratio))

    # -- device path -----------------------------------------------------------------------------------------
    def estimate_flow_t(self, r1, r2, down_ratio):
        """m.py:84-102: both references pooled by 2*down_ratio into one zero-padded (x16) 6-channel buffer,
        FlowNET, crop, bilinear x down_ratio with the vectors scaled by down_ratio."""
        k = 2 * int(down_ratio)
        if r1.h % k or r1.w % k:
            raise hip.VcError(f"frame {r1.h}x{r1.w} is not divisible by 2*down_ratio={k}")
        h, w = r1.h // k, r1.w // k
        hp, wp = -(-h // 16) * 16, -(-w // 16) * 16
        buf = torch.zeros(r1.n * hp * wp * 6, dtype=torch.float32, device=r1.buf.device)
        pair = T(buf, r1.n, hp, wp, 6, hp * wp * 6, wp * 6, 6)
        L = hip.lib()
        for r, c0 in ((r1, 0), (r2, 3)):
            hip.check(L.vc_avgpool_reflectpad(hip.stream(), r.view(), pair.crop(h, w).channels(c0, c0 + 3).view(), k, 1.0),
                      "vc_avgpool_reflectpad")
        flow = self.flow_estimator.run(pair).crop(h, w)
        if down_ratio == 1:
            return flow
        return hip.upsample_bilinear(flow, int(down_ratio), align_corners=False, scale=float(down_ratio))

    def forward_device(self, xref1, xref2, scale1, scale2, xcur, s, down_ratio, bits, flow=None, feats1=None, feats2=None,
                       trace=None):
        """Synchronisation-free body of :meth:`forward` on channels-last views; returns x_hat (T) and appends
        12 x n rows to ``bits`` (offset: z, y_0..y_4; residual: z, y_0..y_4; each for the n images of the batch --
        independent frames of one hierarchy level can share a pass, see gop.code_gop_icip2024).

        ``flow`` (optional): the 4-channel half-resolution flow already chosen by :meth:`search_flow_t` (the
        reference recomputes estimate_flow(best down_ratio), the same function of the same inputs).
        ``feats1`` / ``feats2`` (optional): per image of the batch the ``feature_extractor`` outputs (l1, l2, l3) of
        its reference computed earlier -- a decoded frame serves several B-frames of a GOP as reference and its
        features do not change.
        ``trace``: a dict that receives the stages of the pass ("flow", "cond" = per level [wref1|wref2|fref1|fref2],
        "fcur", "offsets", "aligned", "offset" / "residual" = the codecs' own traces) for stage-wise parity checks."""
        dev, n = xcur.buf.device, xcur.n
        if xcur.h % 64 or xcur.w % 64:
            raise hip.VcError("frame size must be a multiple of 64 (the reference pads with utils.pad)")
        s1, s2 = self.convert_scales(scale1, scale2)
        if flow is None:
            flow = self.estimate_flow_t(xref1, xref2, down_ratio)
        if trace is not None:
            trace["flow"] = flow
        t_off, t_res = ({}, {}) if trace is not None else (None, None)
        chans = (64, 96, 128)
        # per level one buffer [wref1 | wref2 | fref1 | fref2 | fcur]: f_cond_inp = first four, f_inp = all five
        F_ = [T.empty(n, xcur.h >> (l + 1), xcur.w >> (l + 1), 5 * c, dev) for l, c in enumerate(chans)]
        sl = lambda l, j: F_[l].channels(j * chans[l], (j + 1) * chans[l])  # noqa: E731
        fref = []
        for j, (x, feats) in enumerate(((xref1, feats1), (xref2, feats2))):
            if feats is None:
                fref.append(self.feature_extractor.run(x, outs=[sl(l, 2 + j) for l in range(3)]))
            else:                                  # feats: per image a triple of single-image views
                for i, triple in enumerate(feats):
                    for l in range(3):
                        hip.axpby(triple[l], None, out=sl(l, 2 + j).images(i, i + 1))
                fref.append([sl(l, 2 + j) for l in range(3)])
        fref1, fref2 = fref
        fcur = self.feature_extractor.run(xcur, outs=[sl(l, 4) for l in range(3)])
        flows = []
        for l in range(3):                                   # get_warpedrefs_at_layer (m.py:104-119)
            fc1 = hip.axpby(flow.channels(0, 2), None, alpha=s1)
            fc2 = hip.axpby(flow.channels(2, 4), None, alpha=s2)
            flows.append((fc1, fc2))
            hip.warp(hip.WARP_W3, fref1[l], fc1, out=sl(l, 0))
            hip.warp(hip.WARP_W3, fref2[l], fc2, out=sl(l, 1))
            if l < 2:                                        # F.interpolate(scale_factor=0.5, bilinear) * 0.5
                flow = hip.avgpool_reflectpad(flow, 2, scale=0.5)
        cond = [F_[l].channels(0, 4 * chans[l]) for l in range(3)]
        oc = self.offset_compressor
        o1, o2, o3 = oc.code([F_[0]], [F_[1]], [F_[2]], cond[0], cond[1], cond[2],
                             lambda dst: self.offset_temporal_conditioner.run(cond[0], cond[1], cond[2], out=dst), s, bits,
                             trace=t_off)
        comp = []
        for l, (off, div) in enumerate(((o1, self.offset_diversity_l1), (o2, self.offset_diversity_l2),
                                        (o3, self.offset_diversity_l3))):
            hc = off.c // 2
            comp.append(div.run(fref1[l], off.channels(0, hc), flows[l][0], fref2[l], off.channels(hc, 2 * hc), flows[l][1]))
        rc = self.residual_compressor
        x1, x2, x3 = rc.code([fcur[0], comp[0]], [fcur[1], comp[1]], [fcur[2], comp[2]], comp[0], comp[1], comp[2],
                             lambda dst: self.residue_temporal_conditioner.run(comp[0], comp[1], comp[2], out=dst), s, bits,
                             res=(comp[0], comp[1], comp[2]), trace=t_res)
        if trace is not None:
            trace.update({"cond": cond, "fcur": fcur, "offsets": (o1, o2, o3), "aligned": comp, "offset": t_off, "residual": t_res})
        return self.reconstructor.run(x1, x2, x3)

    def forward(self, xref1, xref2, scale1, scale2, xcur, s, down_ratio):
        """m.py:181-260: NCHW CUDA tensors in, ``{"x_hat", "size", "rate"}`` out."""
        _require_frames(xref1, xref2, xcur)
        b, _, h, w = xcur.shape
        bits = BitCounter(xcur.device, max_rows=12 * b)
        x_hat = self.forward_device(hip.nchw_to_nhwc(xref1), hip.nchw_to_nhwc(xref2), scale1, scale2,
                                    hip.nchw_to_nhwc(xcur), s, down_ratio, bits)
        rows = bits.totals()
        size_offset, size_res = rows[:6 * b].sum(), rows[6 * b:].sum()
        num_pixels = h * w * b
        return {"x_hat": hip.nhwc_to_nchw(x_hat), "size": (size_offset + size_res).float(),
                "rate": (size_offset / num_pixels + size_res / num_pixels).float(),
                "size_offset": size_offset, "size_residual": size_res}

    # -- motion-adaptive flow resolution (opt_helpers.py:23-51) --------------------------------------------
    def _predict_from_flow(self, flow, xref1, xref2, s1, s2):
        f21 = hip.upsample_bilinear(flow.channels(0, 2), 2, align_corners=False, scale=2.0)
        f12 = hip.upsample_bilinear(flow.channels(2, 4), 2, align_corners=False, scale=2.0)
        f21 = hip.axpby(f21, None, alpha=s1, out=f21)
        f12 = hip.axpby(f12, None, alpha=s2, out=f12)
        w1 = hip.warp(hip.WARP_W3, xref1, f21)
        w2 = hip.warp(hip.WARP_W3, xref2, f12)
        return hip.axpby(w1, w2, alpha=0.5, beta=0.5, out=w1)

    def prediction_flowonly_t(self, xcur, xref1, xref2, scale1, scale2, down_ratio):
        s1, s2 = self.convert_scales(scale1, scale2)
        return self._predict_from_flow(self.estimate_flow_t(xref1, xref2, down_ratio), xref1, xref2, s1, s2)

    def search_flow_t(self, xcur, xref1, xref2, scale1, scale2, ratios=(1, 2, 4, 8, 16)):
        """get_best_down_ratio_prediction (opt_helpers.py:41-51) without a host round trip: every candidate
        flow is estimated and scored (MSE of the clamped warped-average prediction) on the device, and
        vc_select_flow copies the winner (per image of the batch).  Returns (flow T [n,H/2,W/2,4], choice int32[n] =
        index into ``ratios``, sse float64[n, len(ratios)]); the flow feeds forward_device(flow=...)."""
        s1, s2 = self.convert_scales(scale1, scale2)
        L, dev = hip.lib(), xcur.buf.device
        slots = L.vc_bits_slots()
        partial = torch.empty(len(ratios) * slots, dtype=torch.float64, device=dev)
        sse = torch.empty(len(ratios), dtype=torch.float64, device=dev)
        n, nr = xcur.n, len(ratios)
        partial = torch.empty(n * nr * slots, dtype=torch.float64, device=dev)
        sse = torch.empty(n * nr, dtype=torch.float64, device=dev)         # [image][ratio]
        flows = []
        for i, dr in enumerate(ratios):
            flow = self.estimate_flow_t(xref1, xref2, dr)
            pred = self._predict_from_flow(flow, xref1, xref2, s1, s2)
            for j in range(n):                                                # every frame of the batch decides for itself
                hip.check(L.vc_sse_clamp01(hip.stream(), pred.images(j, j + 1).view(), xcur.images(j, j + 1).view(),
                                           partial.data_ptr() + 8 * (j * nr + i) * slots, slots), "vc_sse_clamp01")
            flows.append(flow)
        hip.check(L.vc_bits_reduce(hip.stream(), partial.data_ptr(), slots, n * nr, sse.data_ptr()), "vc_bits_reduce")
        out = T.empty(n, flows[0].h, flows[0].w, 4, dev)
        choice = torch.empty(n, dtype=torch.int32, device=dev)
        for j in range(n):
            views = (hip.View * nr)(*[f.images(j, j + 1).view() for f in flows])
            hip.check(L.vc_select_flow(hip.stream(), sse.data_ptr() + 8 * j * nr, nr, float(xcur.h * xcur.w * xcur.c), views,
                                       out.images(j, j + 1).view(), choice.data_ptr() + 4 * j), "vc_select_flow")
        return out, choice, sse.view(n, nr)


def prediction_flowonly(model, xcur, xref1, xref2, scale1, scale2, down_ratio):
    """opt_helpers.py:23-38 on NCHW CUDA tensors."""
    _require_frames(xref1, xref2, xcur)
    t = model.prediction_flowonly_t(hip.nchw_to_nhwc(xcur), hip.nchw_to_nhwc(xref1), hip.nchw_to_nhwc(xref2),
                                    scale1, scale2, down_ratio)
    return hip.nhwc_to_nchw(t)


def get_best_down_ratio_prediction(model, xref1, xref2, scale1, scale2, xcur, level=None, beta=None):
    """opt_helpers.py:41-51: the flow resolution whose warped-average prediction has the best PSNR.
    The five candidate predictions are computed on the device; the five MSE scalars are compared on the host."""
    _require_frames(xref1, xref2, xcur)
    c, r1, r2 = hip.nchw_to_nhwc(xcur), hip.nchw_to_nhwc(xref1), hip.nchw_to_nhwc(xref2)
    L, ratios = hip.lib(), (1, 2, 4, 8, 16)
    slots = L.vc_bits_slots()
    partial = torch.empty(len(ratios) * slots, dtype=torch.float64, device=xcur.device)
    sse = torch.empty(len(ratios), dtype=torch.float64, device=xcur.device)
    for i, down_ratio in enumerate(ratios):       # squared error of the clamped prediction: vc_sse_clamp01, no torch operator on pixels
        pred = model.prediction_flowonly_t(c, r1, r2, scale1, scale2, down_ratio)
        hip.check(L.vc_sse_clamp01(hip.stream(), pred.view(), c.view(), partial.data_ptr() + 8 * i * slots, slots), "vc_sse_clamp01")
    hip.check(L.vc_bits_reduce(hip.stream(), partial.data_ptr(), slots, len(ratios), sse.data_ptr()), "vc_bits_reduce")
    count = float(xcur.numel())
    best, best_ratio = 0.0, None
    for down_ratio, e in zip(ratios, sse.cpu().tolist()):
        psnr = float(np.float32(10.0) * np.log10(np.float32(1.0) / np.float32(e / count))) if e > 0 else float("inf")
        if psnr > best:
            best, best_ratio = psnr, down_ratio
    return best_ratio, torch.tensor(best, dtype=torch.float32)        # (a tensor, as opt_helpers.py:51 returns one)


# ------------------------------------------------------------------------------------------------
# utils.py: GOP-16 bookkeeping of the test loop (src/test.py:37-101)
# ------------------------------------------------------------------------------------------------
def get_scales(order, order1, order2):
    if order2 - order1 == 0:
        return 0, 0
    return (order - order1) / (order2 - order1), (order - order2) / (order1 - order2)


def select_references(xcur, order, buffer, buffer_order):
    """utils.py:153-177: the two buffered frames closest in display order (ties broken like torch.topk on the
    |distance| list), returned as (past-side ref, future-side ref, their orders)."""
    d = [abs(i - order) for i in buffer_order]
    k = 1 if len(buffer) == 1 else 2
    ind = list(torch.topk(torch.from_numpy(np.array(d)), k, largest=False).indices.numpy())
    if k == 1:
        return buffer[ind[0]], buffer[ind[0]], buffer_order[ind[0]], buffer_order[ind[0]]
    lo, hi = ind[1], ind[0]
    if buffer_order[ind[0]] < buffer_order[ind[1]]:
        lo, hi = ind[0], ind[1]
    return buffer[lo], buffer[hi], buffer_order[lo], buffer_order[hi]


def update_buffer(buffer, buffer_order, new_frame, order, l=32):
    nb, no = buffer + [new_frame], buffer_order + [order]
    return (nb, no) if len(buffer) < l else (nb[1:], no[1:])


def get_order_typ_list(intra_size, frame_number):
    """utils.py:188-221 including the hard-coded tails for 300- and 600-frame UVG sequences."""
    order = [16, 8, 4, 12, 2, 14, 6, 10, 1, 15, 3, 13, 5, 11, 7, 9]
    o = [0] + [order[i % 16] + (i // 16) * 16 for i in range(frame_number - 1)]
    ff = (frame_number - 1) % intra_size
    if ff != 0:
        m = max(o[:-ff])
        o[-ff:] = [m + ff - i for i in range(ff)]
    typ = ["I" if i % intra_size == 0 else "B" for i in range(frame_number)]
    typ[-1] = "I"
    if frame_number == 300:
        o[-11:] = [299, 293, 290, 296, 289, 291, 292, 294, 295, 297, 298]
    if frame_number == 600:
        o[-7:] = [599, 595, 593, 597, 594, 596, 598]
    return o, typ
