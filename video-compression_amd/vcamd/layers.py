"""Building blocks of the hyperprior codecs, executed on MI355X through libvc_hip.so.

Host-side mirror of the CompressAI 1.1.8 module surface the reference builds on
(``compressai.layers`` / ``entropy_models`` / ``models.MeanScaleHyperprior``; call sites
LHBDC/model/layers.py:6-17,43-191 and Flex-Rate.../b_model/layers.py:76-305).  Every class keeps the
attribute names of the original so that ``state_dict`` keys are identical and reference checkpoints load
with ``strict=True``; the modules only HOLD parameters -- compute goes through the HIP kernels
(``.run(T) -> T``), never through torch operators.
"""
import math

import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .hip import T


# ------------------------------------------------------------------------------------------------
# parameter holders with CompressAI-compatible names
# ------------------------------------------------------------------------------------------------
class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        pedestal = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((float(minimum) + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def resolve(self, p):
        """effective (non-negative) parameter: max(p, bound)^2 - pedestal"""
        return torch.max(p, self.lower_bound.bound) ** 2 - self.pedestal


class _Prepared(nn.Module):
    """Caches device-side packed weights; dropped whenever parameters are (re)loaded or moved."""

    def __init__(self):
        super().__init__()
        self._packed = None

    def _load_from_state_dict(self, *args, **kwargs):
        self._packed = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)


def _dev(module):
    return next(module.parameters()).device


def pack_conv(conv, pixelshuffle=False):
    weight = conv.weight
    if getattr(conv, "mask", None) is not None:       # masked convolutions multiply their weights by the mask buffer
        weight = weight.detach() * conv.mask
    return hip.PackedConv(weight, conv.bias, stride=conv.stride[0], pixelshuffle=pixelshuffle,
                          device=conv.weight.device)


class GDN(_Prepared):
    """y = x * rsqrt(beta + gamma @ x^2)  (inverse: * sqrt) as ONE 1x1 MFMA contraction on x^2 with the
    normalisation, the multiply and the block's residual add fused into the epilogue."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma_init * torch.eye(in_channels)))

    def run(self, x, res=None, out=None, out_sp3=False, out_f16=False):
        """``out_sp3``: the consumer is a split-operand block (fp32 mode "split"): the classic 1x1 instance writes the three bf16 pieces.
        ``out_f16``: the consumer is a residual block on the fp16 path that keeps its identity as half (hip.HALF_RESIDUAL)."""
        if self._packed is None:
            with torch.no_grad():
                c = self.beta.numel()
                beta = self.beta_reparam.resolve(self.beta)
                gamma = self.gamma_reparam.resolve(self.gamma).reshape(c, c, 1, 1)
            self._packed = hip.PackedConv(gamma, beta, device=self.beta.device)
        return self._packed(x, out=out, epi=hip.EPI_IGDN if self.inverse else hip.EPI_GDN, mul=x,
                            in_xform=hip.IN_SQUARE, res=res, out_sp3=out_sp3, out_f16=out_f16)


def conv3x3(in_ch, out_ch, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def conv1x1(in_ch, out_ch, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=1, stride=stride)


def subpel_conv3x3(in_ch, out_ch, r=1):
    return nn.Sequential(nn.Conv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), nn.PixelShuffle(r))


class ResidualBlockWithStride(_Prepared):
    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def out_hw(self, h, w):
        s = self.conv1.stride[0]
        return (h - 1) // s + 1, (w - 1) // s + 1

    def _pack(self):
        if self._packed is None:
            self._packed = (pack_conv(self.conv1), pack_conv(self.conv2),
                            pack_conv(self.skip) if self.skip is not None else None)
        return self._packed

    def half_stream_ok(self):
        """fp16 path: both layers that read the block's input (the strided 3x3 and the skip projection) take a half-precision tensor."""
        c1, _, sk = self._pack()
        return hip.HALF_RESIDUAL and hip.HALF_ACTIVATIONS and c1.half_ok and sk is not None and sk.half_ok

    def run(self, x, out=None, out_sp3=False, out_f16=False):
        c1, c2, sk = self._pack()
        if x.dtype == "f16" and not self.half_stream_ok():
            raise hip.VcError("a half-precision tensor reached a residual block that reads its input in fp32")
        ho, wo = self.out_hw(x.h, x.w)
        # (fp32 mode "split": the stride-2 layer is a native instance whose epilogue writes the split tensor conv2 reads)
        t = c1(x, act=hip.ACT_LRELU, slope=0.01, out_f16=c2.half_ok, out_sp3=hip.wants_split(c2, x, ho, wo))
        u = c2(t)
        identity = x if sk is None else sk(x)
        return self.gdn.run(u, res=identity, out=out, out_sp3=out_sp3, out_f16=out_f16)


class ResidualBlockUpsample(_Prepared):
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def _pack(self):
        if self._packed is None:
            self._packed = (pack_conv(self.subpel_conv[0], pixelshuffle=True), pack_conv(self.conv),
                            pack_conv(self.upsample[0], pixelshuffle=True))
        return self._packed

    def out_hw(self, h, w):
        return 2 * h, 2 * w

    def split_in_ok(self, n, h, w):
        """fp32 mode "split": both layers that read the block's input run on the split-operand pipeline at this size, so the
        producer may hand the input over as a split tensor."""
        sp, _, up = self._pack()
        return hip.wants_split_at(sp, n, h, w) and hip.wants_split_at(up, n, h, w)

    def half_stream_ok(self):
        """fp16 path: both layers that read the block's input take a half-precision tensor."""
        sp, _, up = self._pack()
        return hip.HALF_RESIDUAL and hip.HALF_ACTIVATIONS and sp.half_ok and up.half_ok

    def run(self, x, out=None, out_sp3=False, out_f16=False):
        sp, cv, up = self._pack()
        if x.dtype == "f16" and not self.half_stream_ok():
            raise hip.VcError("a half-precision tensor reached a residual block that reads its input in fp32")
        # fp32 mode "split": both branches read ONE split copy of x; the sub-pixel layer hands its result on as a split tensor
        xs = x
        if x.dtype == "f32" and self.split_in_ok(x.n, x.h, x.w):
            xs = hip.split3(x)
        t = sp(xs, act=hip.ACT_LRELU, slope=0.01, out_f16=cv.half_ok,   # LeakyReLU commutes with the pixel shuffle
               out_sp3=hip.wants_split(cv, x, 2 * x.h, 2 * x.w))
        u = cv(t)
        identity = up(xs)
        return self.igdn.run(u, res=identity, out=out, out_sp3=out_sp3, out_f16=out_f16)


class ResidualBlock(_Prepared):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def _pack(self):
        if self._packed is None:
            self._packed = (pack_conv(self.conv1), pack_conv(self.conv2),
                            pack_conv(self.skip) if self.skip is not None else None)
        return self._packed

    def out_hw(self, h, w):
        return h, w

    def split_in_ok(self, n, h, w):
        """fp32 mode "split": conv1 reads the input as a split tensor and conv2 adds it back from the same tensor (its three pieces
        sum to the exact fp32 value): the producer may hand the input over as a split tensor."""
        c1, c2, sk = self._pack()
        return sk is None and hip.wants_split_at(c1, n, h, w) and hip.wants_split_at(c2, n, h, w)

    def half_stream_ok(self):
        """fp16 path (LHBDC/model/layers.py:48-56,82-91; hip.HALF_RESIDUAL): the block takes its input -- conv1's operand AND the
        identity -- as a half-precision tensor: both 3x3 layers then run on the LDS-DMA kernel, conv2 adds the half identity
        (VC_CFG_RES_F16).  One more rounding of the identity per block: the fp16 mode's tolerance, never the fp32 path."""
        c1, c2, sk = self._pack()
        return hip.HALF_RESIDUAL and hip.HALF_ACTIVATIONS and sk is None and c1.half_ok and c2.half_ok and c2.half_res_ok

    def run(self, x, out=None, out_sp3=False, out_f16=False):
        c1, c2, sk = self._pack()
        if x.dtype == "f16" and not self.half_stream_ok():
            raise hip.VcError("a half-precision tensor reached a residual block that keeps its identity path in fp32")
        t = c1(x, act=hip.ACT_LRELU, slope=0.01, out_f16=c2.half_ok, out_sp3=hip.wants_split(c2, x))
        identity = x if sk is None else sk(x)
        return c2(t, act=hip.ACT_LRELU, slope=0.01, res=identity, out=out, out_sp3=out_sp3,
                  out_f16=bool(out_f16 and out is None and self.half_stream_ok()))


def deconv_as_subpel_weights(deconv):
    """ConvTranspose2d(k=5, stride=2, padding=2, output_padding=1) == conv3x3(pad 1) to 4*Cout channels +
    PixelShuffle(2): output phase (dy,dx) only ever meets taps of matching parity, at most 3x3 of them.
    With w_T[ci, co, ky, kx]:  W3[co*4 + dy*2 + dx, ci, jy, jx] = w_T[ci, co, 4-2*jy+dy, 4-2*jx+dx] (0 if > 4).
    Lets the transposed convolutions of the I-frame codec run on the same MFMA kernel with the
    pixel-shuffle store fused (36 tap slots for 25 taps)."""
    if tuple(deconv.kernel_size) != (5, 5) or tuple(deconv.stride) != (2, 2) or tuple(deconv.padding) != (2, 2) \
            or tuple(deconv.output_padding) != (1, 1):
        raise hip.VcError("only ConvTranspose2d(k=5, s=2, p=2, output_padding=1) is supported")
    wt = deconv.weight.detach().to("cpu", torch.float32)          # [cin, cout, 5, 5]
    cin, cout = wt.shape[0], wt.shape[1]
    w3 = torch.zeros(cout, 2, 2, cin, 3, 3)
    for dy in range(2):
        for dx in range(2):
            for jy in range(3):
                ky = 4 - 2 * jy + dy
                if ky > 4:
                    continue
                for jx in range(3):
                    kx = 4 - 2 * jx + dx
                    if kx > 4:
                        continue
                    w3[:, dy, dx, :, jy, jx] = wt[:, :, ky, kx].t()
    bias = None if deconv.bias is None else deconv.bias.detach().to("cpu", torch.float32).repeat_interleave(4)
    return w3.reshape(cout * 4, cin, 3, 3), bias


def run_sequential(seq, x, cache, final_chscale=None, final_act=None, out=None):
    """Execute an nn.Sequential of {Conv2d, ConvTranspose2d, subpel Sequential, GDN, (Leaky)ReLU, Residual*}
    on the HIP path.  A (Leaky)ReLU following a convolution is fused into that convolution's epilogue.
    ``out`` (optional view, e.g. a channel slice of a concat buffer) receives the result of the last layer."""
    mods = list(seq)
    if cache.get("seq") is None:
        cache["seq"] = {}
    packed = cache["seq"]

    # Chains of 1x1 convolutions with only (Leaky)ReLU between them (the entropy-parameter networks of ICIP2024:
    # 768 -> 426 -> 341 -> 2c, compression_bottlenecks.py:72-551): an intermediate channel count like 426 makes every pixel row
    # of the tensor between two layers unaligned (1704 bytes) and keeps the consumer off the 16-byte staging path and the
    # fp16 path.  The intermediate is padded to a multiple of 16 channels with zero weights / zero bias on the producer
    # and zero weights on the consumer: (Leaky)ReLU(0) = 0 and x + 0 * w = x, so no result bit changes.
    if "pads" not in cache:
        def pointwise(m):
            return (isinstance(m, nn.Conv2d) and tuple(m.kernel_size) == (1, 1) and tuple(m.stride) == (1, 1) and m.groups == 1
                    and getattr(m, "mask", None) is None)
        pads, prev = {}, None
        for idx, m in enumerate(mods):
            if pointwise(m):
                if prev is not None and (-mods[prev].out_channels) % 16:
                    pad = (-mods[prev].out_channels) % 16
                    pads[id(mods[prev])] = (pads.get(id(mods[prev]), (0, 0))[0], pad)
                    pads[id(m)] = (pad, 0)
                prev = idx
            elif not isinstance(m, (nn.ReLU, nn.LeakyReLU)):
                prev = None
        cache["pads"] = pads
    pads = cache["pads"]

    def pack_of(m):
        """PackedConv of a Conv2d / ConvTranspose2d / subpel Sequential member (None for anything else)."""
        is_subpel = isinstance(m, nn.Sequential)
        conv = m[0] if is_subpel else m
        key = id(conv)
        if key not in packed:
            if isinstance(conv, nn.ConvTranspose2d):
                w3, b3 = deconv_as_subpel_weights(conv)
                packed[key] = hip.PackedConv(w3, b3, stride=1, pixelshuffle=True, device=conv.weight.device)
            elif isinstance(conv, nn.Conv2d) and key in pads:
                pi, po = pads[key]
                w = torch.zeros(conv.out_channels + po, conv.in_channels + pi, 1, 1)
                w[:conv.out_channels, :conv.in_channels] = conv.weight.detach().to("cpu", torch.float32)
                b = torch.zeros(conv.out_channels + po)
                if conv.bias is not None:
                    b[:conv.out_channels] = conv.bias.detach().to("cpu", torch.float32)
                packed[key] = hip.PackedConv(w, b, stride=1, device=conv.weight.device)
            elif isinstance(conv, nn.Conv2d):
                packed[key] = pack_conv(conv, pixelshuffle=is_subpel)
            else:
                return None
        return packed[key]

    def takes_half(m):
        """May the member's input live in HBM as half (fp16 path)?"""
        if hasattr(m, "half_stream_ok"):
            return m.half_stream_ok()
        if isinstance(m, (ResidualBlock, ResidualBlockWithStride, ResidualBlockUpsample, GDN)) or getattr(m, "vc_block", False):
            return False
        pk = pack_of(m)
        return pk is not None and pk.half_ok

    def takes_split(m, n, h, w):
        """fp32 mode "split": may the member's input arrive as a split tensor (three bf16 pieces per value)?"""
        if hasattr(m, "split_in_ok"):
            return m.split_in_ok(n, h, w)
        if isinstance(m, (ResidualBlockWithStride, GDN)) or getattr(m, "vc_block", False):
            return False
        pk = pack_of(m)
        return pk is not None and hip.wants_split_at(pk, n, h, w)

    i = 0
    while i < len(mods):
        m = mods[i]
        last = i == len(mods) - 1
        if isinstance(m, (ResidualBlock, ResidualBlockWithStride, ResidualBlockUpsample, GDN)) or getattr(m, "vc_block", False):
            if last and (final_chscale is not None or final_act is not None):
                raise hip.VcError("a block cannot take the sequence's final gain/activation")
            if last and out is not None:
                x = m.run(x, out=out)
            elif hasattr(m, "half_stream_ok") and not last and takes_half(mods[i + 1]):
                x = m.run(x, out_f16=True)                # the next member reads (and, a block, adds) it as half
            elif hasattr(m, "out_hw") and not last and takes_split(mods[i + 1], x.n, *m.out_hw(x.h, x.w)):
                x = m.run(x, out_sp3=True)                # fp32 mode "split": the next member reads it as a split tensor
            else:
                x = m.run(x)
            i += 1
            continue
        pk = pack_of(m)
        if pk is None:
            raise hip.VcError(f"unsupported layer in sequential: {type(m).__name__}")
        act, slope = hip.ACT_NONE, 0.0
        if i + 1 < len(mods) and isinstance(mods[i + 1], (nn.LeakyReLU, nn.ReLU)):
            nxt = mods[i + 1]
            act, slope = (hip.ACT_LRELU, nxt.negative_slope) if isinstance(nxt, nn.LeakyReLU) else (hip.ACT_RELU, 0.0)
            i += 1
            last = i == len(mods) - 1
        if last and final_act is not None:
            act = final_act
        # a result consumed by the next convolution alone (or by a bottleneck block that keeps its identity as half) may be
        # kept as half on the fp16 path
        ho, wo, _ = pk.out_shape(x.h, x.w)
        x = pk(x, act=act, slope=slope, chscale=final_chscale if last else None, out=out if last else None,
               out_f16=bool(not last and takes_half(mods[i + 1])),
               out_sp3=bool(not last and takes_split(mods[i + 1], x.n, ho, wo)))
        i += 1
    return x


# ------------------------------------------------------------------------------------------------
# entropy models
# ------------------------------------------------------------------------------------------------
class EntropyModel(_Prepared):
    def __init__(self, likelihood_bound=1e-9, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            q = hip.pmf_to_quantized_cdf(prob.numpy(), self.entropy_coder_precision)
            cdf[i, : q.size] = torch.from_numpy(q.astype(np.int32))
        return cdf

    def tables(self):
        """host copies (numpy int32) of cdf / cdf_length / offset for the range coder"""
        if self._offset.numel() == 0:
            raise hip.VcError("entropy tables are empty: call update(force=True) after loading weights")
        return (self._quantized_cdf.cpu().numpy().astype(np.int32), self._cdf_length.cpu().numpy().astype(np.int32),
                self._offset.cpu().numpy().astype(np.int32))


class EntropyBottleneck(EntropyModel):
    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(**kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        if self.filters != (3, 3, 3, 3):
            raise hip.VcError("the fused factorised-prior kernel is specialised for filters=(3,3,3,3)")
        widths = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / widths[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, widths[i + 1], widths[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(channels, widths[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, widths[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    # -- host-side table construction (once per model load; CompressAI EntropyBottleneck.update) ----
    def _logits_cumulative_host(self, inputs):
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(getattr(self, f"_matrix{i:d}").detach().cpu()), logits)
            logits = logits + getattr(self, f"_bias{i:d}").detach().cpu()
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, f"_factor{i:d}").detach().cpu()) * torch.tanh(logits)
        return logits

    @torch.no_grad()
    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        q = self.quantiles.detach().cpu()
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative_host(samples - 0.5)
        upper = self._logits_cumulative_host(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._offset = (-minima).int().to(dev)
        return True

    def device_params(self):
        """[C,60] fp32: softplus/tanh pre-resolved MLP + median, layout of VC_EB_PARAMS_PER_CHANNEL."""
        if self._packed is None:
            with torch.no_grad():
                c = self.channels
                parts = []
                for i in range(5):
                    parts.append(F.softplus(getattr(self, f"_matrix{i:d}").detach().cpu()).reshape(c, -1))
                    parts.append(getattr(self, f"_bias{i:d}").detach().cpu().reshape(c, -1))
                    if i < 4:
                        parts.append(torch.tanh(getattr(self, f"_factor{i:d}").detach().cpu()).reshape(c, -1))
                parts.append(self.quantiles.detach().cpu()[:, 0, 1].reshape(c, 1))
                parts.append(torch.zeros(c, 1))
                p = torch.cat(parts, dim=1).contiguous().float()
                assert p.shape[1] == hip.EB_PARAMS_PER_CHANNEL
            self._packed = p.to(self.quantiles.device)
        return self._packed


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(**kwargs)
        self.register_buffer("scale_table", torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))
        self.tail_mass = float(tail_mass)
        self.lower_bound_scale = LowerBound(scale_bound)

    @torch.no_grad()
    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.scale_bound.device
        self.scale_table = torch.Tensor(tuple(float(s) for s in scale_table)).to(dev)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        dev = self.scale_bound.device
        table = self.scale_table.detach().cpu()
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(table * float(multiplier)).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        scale = table.unsqueeze(1).float()
        const = float(-(2 ** -0.5))
        upper = 0.5 * torch.erfc(const * ((0.5 - samples) / scale))
        lower = 0.5 * torch.erfc(const * ((-0.5 - samples) / scale))
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._packed = None


def get_scale_table(lo=0.11, hi=256, levels=64):
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


_CDF_BUFFERS = ("_quantized_cdf", "_offset", "_cdf_length")


def _resize_registered_buffers(module, prefix, names, state_dict):
    """CompressAI's load_state_dict override: size the CDF buffers like the checkpoint's (needed for
    checkpoints saved after update(), as Flex loads them child by child -- test/utils.py:253-270)."""
    for name in names:
        key = f"{prefix}.{name}"
        if key not in state_dict:
            raise RuntimeError(f'missing key "{key}" in state_dict')
        src = state_dict[key]
        buf = getattr(module, name)
        if buf.shape != src.shape:
            setattr(module, name, torch.empty(src.shape, dtype=src.dtype, device=buf.device))


# ------------------------------------------------------------------------------------------------
# mean-scale hyperprior codec
# ------------------------------------------------------------------------------------------------
class BitCounter:
    """Device-side accumulation of -log2(likelihood) sums: one row of workgroup partials per entropy
    launch, folded by vc_bits_reduce; a single D2H copy at the very end of the frame."""

    def __init__(self, device, max_rows=16):
        self.slots = hip.lib().vc_bits_slots()
        self.max_rows = int(max_rows)
        self.partial = torch.zeros(self.max_rows * self.slots, dtype=torch.float64, device=device)
        self.out = torch.zeros(self.max_rows, dtype=torch.float64, device=device)
        self.rows = 0

    def next_row_ptr(self):
        if self.rows >= self.max_rows:      # a row past the end would be written into a neighbouring allocation
            raise hip.VcError(f"BitCounter: all {self.max_rows} rows are in use (size it from the batch: rows per image x images)")
        ptr = self.partial.data_ptr() + 8 * self.rows * self.slots
        self.rows += 1
        return ptr

    def totals(self):
        if self.rows == 0:
            raise hip.VcError("BitCounter: no rows were written")
        hip.check(hip.lib().vc_bits_reduce(hip.stream(), self.partial.data_ptr(), self.slots, self.rows,
                                           self.out.data_ptr()), "vc_bits_reduce")
        return self.out[: self.rows]


class MeanScaleHyperprior(_Prepared):
    """Holder + executor with the attribute names of compressai.models.MeanScaleHyperprior.
    Sub-classes assign g_a, h_a, h_s, g_s (LHBDC/model/layers.py:48-91)."""

    def __init__(self, N, M, **kwargs):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = int(N), int(M)
        self._cache = {"g_a": {}, "h_a": {}, "h_s": {}, "g_s": {}}

    def _load_from_state_dict(self, *args, **kwargs):
        self._cache = {"g_a": {}, "h_a": {}, "h_s": {}, "g_s": {}}
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._cache = {"g_a": {}, "h_a": {}, "h_s": {}, "g_s": {}}
        return super()._apply(fn, *a, **k)

    # -- checkpoint behaviour of CompressAI ------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        _resize_registered_buffers(self.entropy_bottleneck, "entropy_bottleneck", _CDF_BUFFERS, state_dict)
        _resize_registered_buffers(self.gaussian_conditional, "gaussian_conditional",
                                   _CDF_BUFFERS + ("scale_table",), state_dict)
        return super().load_state_dict(state_dict, strict=strict)

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        return updated

    def aux_loss(self):
        raise NotImplementedError("training is outside the inference hot path")

    # -- HIP execution -----------------------------------------------------------------------------
    def _gains(self, n, l):
        """(gain, inv_gain, hyper_gain, hyper_inv_gain) device vectors or Nones (Flex overrides)."""
        return None, None, None, None

    @staticmethod
    def _refinable(last):
        """a plain k x k convolution (k odd <= 7, stride 1 / 2, padding k // 2) the fp64 refinement kernels can recompute"""
        return (isinstance(last, nn.Conv2d) and last.kernel_size[0] == last.kernel_size[1] and last.kernel_size[0] % 2 == 1
                and last.kernel_size[0] <= 7 and last.stride[0] == last.stride[1] and last.stride[0] in (1, 2)
                and tuple(last.padding) == (last.kernel_size[0] // 2,) * 2 and last.groups == 1 and tuple(last.dilation) == (1, 1)
                and getattr(last, "mask", None) is None)

    def _run_keeping_last_input(self, name, x, final_chscale=None):
        """Run transform ``name`` as head + last layer; returns (result, t, last) with ``t`` the last layer's fp32 input -- what the
        refinement kernels recompute that layer from -- or (result, None, None) when the transform does not end in a plain
        convolution."""
        seq = getattr(self, name)
        cache = self._cache[name]
        mods = list(seq)
        last = mods[-1]
        if len(mods) < 2 or not self._refinable(last):
            return run_sequential(seq, x, cache, final_chscale=final_chscale), None, None
        if cache.get("seq") is None:
            cache["seq"] = {}
        if "head" not in cache:        # (the parts share the transform's packed weights)
            cache["head"] = (nn.Sequential(*mods[:-1]), {"seq": cache["seq"]})
            cache["tail"] = (nn.Sequential(last), {"seq": cache["seq"]})
        t = run_sequential(cache["head"][0], x, cache["head"][1])
        if t.dtype != "f32":
            raise hip.VcError("the refinement kernels read the last layer's input in fp32")
        return run_sequential(cache["tail"][0], t, cache["tail"][1], final_chscale=final_chscale), t, last

    @staticmethod
    def _refine_layer(t, conv, c0=0):
        w = conv.weight.detach()
        if not w.is_contiguous() or w.dtype != torch.float32:
            raise hip.VcError("the refinement kernels read the checkpoint's fp32 weights in place")
        rl = hip.RefineLayer(t.view(), w.data_ptr(), None if conv.bias is None else conv.bias.detach().data_ptr(),
                             conv.kernel_size[0], conv.stride[0], c0)
        rl.keep = (t, w)       # the record holds raw pointers: the input tensor must outlive it (its block would be re-used otherwise)
        return rl

    def _hs_for_bitstream(self, z_hat):
        """Hyper-synthesis for the BITSTREAM paths (compress / decompress): the scales whose table index could go either way -- within
        hip.SCALE_REFINE_EPS of a table entry -- are recomputed in fp64 from the last layer's input (vc_refine_scales), so this
        side's indexes do not depend on its fp32 summation order.  Encoder and decoder call the same function: identical indexes.
        Returns (gaussian parameters, mu_layer): mu_layer = the vc_refine_layer of the MEANS half, for the encoder's symbol
        refinement (None when the transform does not end in a plain convolution)."""
        last = list(self.h_s)[-1]
        if not (isinstance(last, nn.Conv2d) and last.out_channels == 2 * self.M):
            with hip.fp32_mode_pinned(hip.BITSTREAM_HS_MODE):
                return run_sequential(self.h_s, z_hat, self._cache["h_s"]), None
        with hip.fp32_mode_pinned(hip.BITSTREAM_HS_MODE):         # (one pipeline for every stream: see hip.BITSTREAM_HS_MODE)
            gp, t, last = self._run_keeping_last_input("h_s", z_hat)
        if t is None:
            return gp, None
        if hip.SCALE_REFINE and last.kernel_size[0] == 3 and last.stride[0] == 1:
            table = self._scale_table_dev()
            hip.check(hip.lib().vc_refine_scales(hip.stream(), gp.channels(0, self.M).view(), t.view(), last.weight.detach().data_ptr(),
                                                 None if last.bias is None else last.bias.detach().data_ptr(), table.data_ptr(), table.numel(),
                                                 hip.SCALE_REFINE_EPS, None), "vc_refine_scales")
        return gp, self._refine_layer(t, last, self.M)

    def _analysis_for_bitstream(self, x, gains, code_ungained_y):
        """g_a and h_a of the encoder's bitstream paths.  Returns (y, y_raw, z, y_layer, z_layer): ``y`` feeds h_a (gained when a gain is
        set), ``y_raw`` (or None) is the un-gained latent Flex-Rate's compress() codes (b_model/layers.py:167), ``y_layer`` /
        ``z_layer`` the vc_refine_layer records of the two transforms' last convolutions (None = no symbol refinement)."""
        g = gains[0]
        L = hip.lib()
        refine = hip.SYMBOL_REFINE
        y_layer = z_layer = None
        if g is not None and code_ungained_y:
            if refine:
                y_raw, t, last = self._run_keeping_last_input("g_a", x)
                y_layer = None if t is None else self._refine_layer(t, last)
            else:
                y_raw = run_sequential(self.g_a, x, self._cache["g_a"])
            y = T.empty(y_raw.n, y_raw.h, y_raw.w, y_raw.c, x.buf.device)
            # scaled_y = gain * y  (Gain_Module.forward) -- needed by h_a; the un-gained y is what gets coded
            hip.check(L.vc_channel_scale(hip.stream(), y_raw.view(), g.data_ptr(), y.view()), "vc_channel_scale")
        else:
            y_raw = None
            if refine and g is None:
                y, t, last = self._run_keeping_last_input("g_a", x)
                y_layer = None if t is None else self._refine_layer(t, last)
            else:                              # (a gained latent that is itself coded: not a reference path -- plain rounding)
                y = run_sequential(self.g_a, x, self._cache["g_a"], final_chscale=g)
        if refine:
            z, t, last = self._run_keeping_last_input("h_a", y)
            z_layer = None if t is None else self._refine_layer(t, last)
        else:
            z = run_sequential(self.h_a, y, self._cache["h_a"])
        return y, y_raw, z, y_layer, z_layer

    def _quantise_for_bitstream(self, y, y_raw, z, y_layer, z_layer, gains, want_y_hat, counters=None):
        """Hyper-latent symbols -> z_hat -> (scales, means) -> latent symbols and scale-table indexes, with the boundary cases of both
        roundings decided in fp64 (hip.SYMBOL_REFINE).  Returns (z_sym [n, count], y_sym, y_idx, y_hat or None, (hz, wz)); device
        int32 tensors.  ``counters``: an int32 device tensor [2] that receives the number of refined (y, z) elements."""
        g, ig, hg, hig = gains
        L = hip.lib()
        dev = y.buf.device
        z_hat = T.empty(z.n, z.h, z.w, z.c, dev)
        z_sym = torch.empty((z.n, z.c * z.h * z.w), dtype=torch.int32, device=dev)
        eb = self.entropy_bottleneck.device_params()
        hip.check(L.vc_eb_forward(hip.stream(), z.view(), eb.data_ptr(), None if hg is None else hg.data_ptr(),
                                  None if hig is None else hig.data_ptr(), z_hat.view(), z_sym.data_ptr(), None, 0, None), "vc_eb_forward")
        if z_layer is not None:
            hip.check(L.vc_refine_z_symbols(hip.stream(), z.view(), z_layer, eb.data_ptr(), None if hg is None else hg.data_ptr(),
                                            hip.SYMBOL_REFINE_EPS, z_sym.data_ptr(), z_hat.view(), None if hig is None else hig.data_ptr(),
                                            None if counters is None else counters.data_ptr() + 4), "vc_refine_z_symbols")
        gp, mu_layer = self._hs_for_bitstream(z_hat)
        m = self.M
        scales, means = gp.channels(0, m), gp.channels(m, 2 * m)
        y_hat = T.empty(y.n, y.h, y.w, y.c, dev) if want_y_hat else None
        y_sym = torch.empty((y.n, y.c * y.h * y.w), dtype=torch.int32, device=dev)
        y_idx = torch.empty_like(y_sym)
        table = self._scale_table_dev()
        hip.check(L.vc_gc_forward(hip.stream(), y.view(), scales.view(), means.view(), None, None if ig is None else ig.data_ptr(),
                                  hip.NULL_VIEW if y_hat is None else y_hat.view(), None, 0, None if y_raw is None else y_raw.ptr,
                                  y_sym.data_ptr(), y_idx.data_ptr(), table.data_ptr(), table.numel(), None), "vc_gc_forward")
        if y_layer is not None and mu_layer is not None:
            # (Flex-Rate codes the un-gained latent while y_hat comes from the gained one -- quirk B.6: y_hat is then left alone)
            fix_hat = y_hat is not None and y_raw is None
            hip.check(L.vc_refine_y_symbols(hip.stream(), (y if y_raw is None else y_raw).view(), y_layer, means.view(), mu_layer,
                                            hip.SYMBOL_REFINE_EPS, y_sym.data_ptr(), y_hat.view() if fix_hat else hip.NULL_VIEW,
                                            None if (ig is None or not fix_hat) else ig.data_ptr(),
                                            None if counters is None else counters.data_ptr()), "vc_refine_y_symbols")
        return z_sym, y_sym, y_idx, y_hat, (z.h, z.w)

    def forward_t(self, x, bits, gains=(None, None, None, None), likelihoods=None, trace=None):
        """x: T [n,h,w,c_in] -> x_hat T; appends two rows (y then z) PER IMAGE to the BitCounter.
        ``likelihoods``: a dict that receives the per-element likelihood tensors "y" [n,M,h/16,w/16] and "z"
        [n,N,h/64,w/64] (NCHW fp32 CUDA tensors, what CompressAI's forward returns under "likelihoods").
        ``trace``: a dict that receives the analysis outputs and the quantised integers (parity instrumentation of the
        tests and bench.py: "y", "z", "scales", "means" as T windows, "y_sym" / "z_sym" int32 NCHW device tensors)."""
        g, ig, hg, hig = gains
        L = hip.lib()
        y = run_sequential(self.g_a, x, self._cache["g_a"], final_chscale=g)      # gained y when g is set
        z = run_sequential(self.h_a, y, self._cache["h_a"])
        z_hat = T.empty(z.n, z.h, z.w, z.c, z.buf.device)
        # one (y, z) pair of counter rows PER IMAGE, so a batch of independent frames keeps per-frame sizes
        rows = [(bits.next_row_ptr(), bits.next_row_ptr()) for _ in range(z.n)]
        lik_y = lik_z = None
        if likelihoods is not None:
            lik_z = torch.empty((z.n, z.c, z.h, z.w), dtype=torch.float32, device=z.buf.device)
            lik_y = torch.empty((y.n, y.c, y.h, y.w), dtype=torch.float32, device=y.buf.device)
            likelihoods["y"], likelihoods["z"] = lik_y, lik_z
        sym_y = sym_z = None
        if trace is not None:
            sym_z = torch.empty((z.n, z.c, z.h, z.w), dtype=torch.int32, device=z.buf.device)
            sym_y = torch.empty((y.n, y.c, y.h, y.w), dtype=torch.int32, device=y.buf.device)
            trace.update({"y": y, "z": z, "y_sym": sym_y, "z_sym": sym_z})
        for i in range(z.n):
            hip.check(L.vc_eb_forward(hip.stream(), z.images(i, i + 1).view(), self.entropy_bottleneck.device_params().data_ptr(),
                                      None if hg is None else hg.data_ptr(), None if hig is None else hig.data_ptr(),
                                      z_hat.images(i, i + 1).view(), None if sym_z is None else sym_z[i].data_ptr(),
                                      rows[i][1], bits.slots, None if lik_z is None else lik_z[i].data_ptr()), "vc_eb_forward")
        gp = run_sequential(self.h_s, z_hat, self._cache["h_s"])
        m = self.M
        scales, means = gp.channels(0, m), gp.channels(m, 2 * m)
        if trace is not None:
            trace.update({"scales": scales, "means": means})
        y_hat = T.empty(y.n, y.h, y.w, y.c, y.buf.device)
        for i in range(y.n):
            # (y, scales, means read + y_hat written; + symbols / likelihoods when requested)
            hip.timed_hbm(f"k_gc_forward c{y.c} @1x{y.h}x{y.w}",
                          4.0 * y.h * y.w * y.c * (4 + (sym_y is not None) + (lik_y is not None)),
                          lambda i=i: hip.check(L.vc_gc_forward(
                              hip.stream(), y.images(i, i + 1).view(), scales.images(i, i + 1).view(),
                              means.images(i, i + 1).view(), None, None if ig is None else ig.data_ptr(),
                              y_hat.images(i, i + 1).view(), rows[i][0], bits.slots,
                              None, None if sym_y is None else sym_y[i].data_ptr(), None, None, 0,
                              None if lik_y is None else lik_y[i].data_ptr()), "vc_gc_forward"))
        return run_sequential(self.g_s, y_hat, self._cache["g_s"])

    def _scale_table_dev(self):
        gc = self.gaussian_conditional
        if gc._packed is None:
            if gc.scale_table.numel() == 0:
                raise hip.VcError("scale table is empty: call update(force=True) after loading weights")
            gc._packed = gc.scale_table.detach().float().contiguous().to(gc.scale_bound.device)
        return gc._packed

    def compress_t(self, x, gains=(None, None, None, None), code_ungained_y=False, trace=None):
        """Analysis + symbolisation on the GPU, range coding on the host.  Returns (strings, (hz,wz)).
        ``trace``: a dict that receives the integers handed to the range coder ("y_sym", "y_idx", "z_sym": host int32
        arrays [n, count]) -- parity instrumentation."""
        y, y_raw, z, y_layer, z_layer = self._analysis_for_bitstream(x, gains, code_ungained_y)
        counters = None
        if trace is not None:
            counters = torch.zeros(2, dtype=torch.int32, device=x.buf.device)
        z_sym, y_sym, y_idx, _, _ = self._quantise_for_bitstream(y, y_raw, z, y_layer, z_layer, gains, False, counters)
        if trace is not None:
            trace["refined"] = tuple(int(v) for v in counters.cpu())      # (y, z) elements decided in fp64
        # single D2H of the integer symbols, then the serial coder on the host
        z_sym_h = z_sym.cpu().numpy()
        y_sym_h = y_sym.cpu().numpy()
        y_idx_h = y_idx.cpu().numpy()
        if trace is not None:
            trace.update({"y_sym": y_sym_h, "y_idx": y_idx_h, "z_sym": z_sym_h})
        eb_cdf, eb_len, eb_off = self.entropy_bottleneck.tables()
        gc_cdf, gc_len, gc_off = self.gaussian_conditional.tables()
        z_index = np.repeat(np.arange(z.c, dtype=np.int32), z.h * z.w)
        z_strings = [hip.rans_encode(z_sym_h[i], z_index, eb_cdf, eb_len, eb_off) for i in range(z.n)]
        y_strings = [hip.rans_encode(y_sym_h[i], y_idx_h[i], gc_cdf, gc_len, gc_off) for i in range(y.n)]
        return [y_strings, z_strings], (z.h, z.w)

    # -- building blocks of the pipelined bitstream codec (vcamd/bitstream.py) -------------------------------
    def code_t(self, x, gains=(None, None, None, None), code_ungained_y=False):
        """ONE analysis pass that yields both what an encoder needs: the reconstruction x_hat (exactly what the decoder
        will rebuild from the coded integers) and the integers themselves, left on the device for an asynchronous copy:
        returns (x_hat T, {"y_sym", "y_idx", "z_sym": int32 [n, count], "shape": (hz, wz)}).  Equivalent to
        forward_t + compress_t without running g_a / h_a / h_s twice."""
        y, y_raw, z, y_layer, z_layer = self._analysis_for_bitstream(x, gains, code_ungained_y)
        z_sym, y_sym, y_idx, y_hat, _ = self._quantise_for_bitstream(y, y_raw, z, y_layer, z_layer, gains, True)
        x_hat = run_sequential(self.g_s, y_hat, self._cache["g_s"])
        return x_hat, {"y_sym": y_sym, "y_idx": y_idx, "z_sym": z_sym, "shape": (z.h, z.w)}

    def hyper_decode_t(self, z_sym_d, n, shape, gains=(None, None, None, None)):
        """Decoder, first half (device): hyper-latent symbols [n, C*hz*wz] -> (means T, scale-table indexes int32 device
        tensor [n, M*hy*wy]) -- everything the host needs to decode the y strings."""
        g, ig, hg, hig = gains
        L = hip.lib()
        device = z_sym_d.device
        hz, wz = int(shape[0]), int(shape[1])
        z_hat = T.empty(n, hz, wz, self.N, device)
        hip.check(L.vc_eb_dequant(hip.stream(), z_sym_d.data_ptr(), self.entropy_bottleneck.device_params().data_ptr(),
                                  None if hig is None else hig.data_ptr(), z_hat.view()), "vc_eb_dequant")
        gp, _ = self._hs_for_bitstream(z_hat)
        m = self.M
        scales, means = gp.channels(0, m), gp.channels(m, 2 * m)
        idx_d = torch.empty((n, m * gp.h * gp.w), dtype=torch.int32, device=device)
        table = self._scale_table_dev()
        hip.check(L.vc_gc_indexes(hip.stream(), scales.view(), table.data_ptr(), table.numel(), idx_d.data_ptr()), "vc_gc_indexes")
        return means, idx_d

    def synth_decode_t(self, y_sym_d, means, gains=(None, None, None, None), final_act=None):
        """Decoder, second half (device): y symbols [n, M*hy*wy] + the means of hyper_decode_t -> x_hat T."""
        g, ig, hg, hig = gains
        y_hat = T.empty(means.n, means.h, means.w, self.M, y_sym_d.device)
        hip.check(hip.lib().vc_gc_dequant(hip.stream(), y_sym_d.data_ptr(), means.view(), None if ig is None else ig.data_ptr(),
                                          y_hat.view()), "vc_gc_dequant")
        return run_sequential(self.g_s, y_hat, self._cache["g_s"], final_act=final_act)

    def decompress_t(self, strings, shape, device, gains=(None, None, None, None), final_act=None, trace=None):
        """``trace``: a dict that receives the decoder's integers ("z_sym", "y_idx", "y_sym": host int32 [n, count]).
        A ``trace["y_idx_override"]`` entry (int32 [n, count], put there by the caller) replaces the scale-table indexes
        this decoder derived -- a cross-platform diagnostic: the CompressAI format carries no indexes, every decoder re-derives
        them from ITS hyper-synthesis output, and a scale within fp32 noise of a table entry lands in the neighbouring bin
        on another platform (the stream is then undecodable there).  The parity tests use it to show that such boundary
        cases are the ONLY thing between this decoder and a stream written by the reference on a CPU."""
        assert isinstance(strings, list) and len(strings) == 2
        g, ig, hg, hig = gains
        L = hip.lib()
        n = len(strings[1])
        hz, wz = int(shape[0]), int(shape[1])
        c = self.N
        eb_cdf, eb_len, eb_off = self.entropy_bottleneck.tables()
        gc_cdf, gc_len, gc_off = self.gaussian_conditional.tables()
        z_index = np.repeat(np.arange(c, dtype=np.int32), hz * wz)
        z_sym = np.stack([hip.rans_decode(strings[1][i], z_index, eb_cdf, eb_len, eb_off) for i in range(n)])
        z_sym_d = torch.from_numpy(z_sym).to(device)
        z_hat = T.empty(n, hz, wz, c, device)
        hip.check(L.vc_eb_dequant(hip.stream(), z_sym_d.data_ptr(), self.entropy_bottleneck.device_params().data_ptr(),
                                  None if hig is None else hig.data_ptr(), z_hat.view()), "vc_eb_dequant")
        gp, _ = self._hs_for_bitstream(z_hat)
        m = self.M
        scales, means = gp.channels(0, m), gp.channels(m, 2 * m)
        idx_d = torch.empty(n * m * gp.h * gp.w, dtype=torch.int32, device=device)
        table = self._scale_table_dev()
        hip.check(L.vc_gc_indexes(hip.stream(), scales.view(), table.data_ptr(), table.numel(), idx_d.data_ptr()),
                  "vc_gc_indexes")
        idx_h = idx_d.cpu().numpy().reshape(n, -1)
        if trace is not None:
            trace.update({"z_sym": z_sym, "y_idx": idx_h})
            if trace.get("y_idx_override") is not None:
                idx_h = np.ascontiguousarray(np.asarray(trace["y_idx_override"], dtype=np.int32).reshape(n, -1))
        y_sym = np.stack([hip.rans_decode(strings[0][i], idx_h[i], gc_cdf, gc_len, gc_off) for i in range(n)])
        if trace is not None:
            trace["y_sym"] = y_sym
        y_sym_d = torch.from_numpy(y_sym).to(device)
        y_hat = T.empty(n, gp.h, gp.w, m, device)
        hip.check(L.vc_gc_dequant(hip.stream(), y_sym_d.data_ptr(), means.view(), None if ig is None else ig.data_ptr(),
                                  y_hat.view()), "vc_gc_dequant")
        return run_sequential(self.g_s, y_hat, self._cache["g_s"], final_act=final_act)
