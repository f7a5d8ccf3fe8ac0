"""CPU restatement of the ICIP2024 flow-guided deformable B-frame codec -- TEST INFRASTRUCTURE ONLY.

Follows (relative to /root/reference/ICIP2024/src):
  model/m.py:31-282                      FlowGuidedB (forward :181-260, warp :262-282)   -> :class:`FlowGuidedB`
  model/helpers.py:35-61                 OffsetDiversity                                 -> :class:`OffsetDiversity`
  model/helpers.py:74-259                MS_Feature, FlowNET, *TemproalEnc, Reconstuctor -> same names
  model/elic.py:69-83                    ResidualBottleneckBlock                         -> :class:`ResidualBottleneckBlock`
  model/layers.py:6-29                   CheckerboardContext                             -> :class:`CheckerboardContext`
  model/compression_bottlenecks.py:72-551  Offset_ELIC / Res_ELIC                        -> :class:`_Elic` (one body, two configs)
  model/elic.py:85-260                   ELIC intra codec (forward only; src/test.py:60 via utils.image_compress) -> :class:`ELIC`
  opt_helpers.py:23-51                   prediction_flowonly, get_best_down_ratio_prediction
  utils.py:153-250                       select_references, update_buffer, get_order_typ_list, get_scales

PINNED against the real reference modules by oracle/gen_golden.py (fixtures tests/golden/icip2024_*.npz): the
reference is imported with ``compressai`` -> ``oracle.cai`` and ``torchvision.ops.DeformConv2d`` ->
``oracle.deform.DeformConv2d`` -- PARITY UNPINNED at those two third-party boundaries.  Module and parameter names
equal the reference's so that one state_dict loads into both (strict).  The reference has no compress() for this
model: rate is the likelihood estimate only.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .cai.layers import AttentionBlock, conv1x1, conv3x3, subpel_conv3x3
from .cai.models import JointAutoregressiveHierarchicalPriors
from .deform import DeformConv2d


def _conv(i, o, kernel_size=5, stride=2):
    return nn.Conv2d(i, o, kernel_size, stride, kernel_size // 2)


def _deconv(i, o, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(i, o, kernel_size, stride, kernel_size // 2, output_padding=stride - 1)


class ResidualBottleneckBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.BottleneckBlock = nn.Sequential(conv1x1(in_ch, out_ch), nn.ReLU(inplace=True), conv3x3(out_ch, out_ch),
                                             nn.ReLU(inplace=True), conv1x1(out_ch, out_ch))

    def forward(self, x):
        return self.BottleneckBlock(x) + x


def _rbb(c, n=3):
    return [ResidualBottleneckBlock(c, c) for _ in range(n)]


class CheckerboardContext(nn.Conv2d):
    """5x5 convolution whose weights are kept only where (row + col) is odd (layers.py:20-28)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("mask", torch.zeros_like(self.weight.data))
        self.mask[:, :, 0::2, 1::2] = 1
        self.mask[:, :, 1::2, 0::2] = 1

    def forward(self, x):
        self.weight.data *= self.mask
        return super().forward(x)


def warp_w3(img, flow):
    """m.py:262-282: linspace(-1,1) grid + flow/((size-1)/2), align_corners=True, border padding ==
    bilinear sampling exactly at pixel (x+u, y+v) with the coordinate clamped to the image."""
    b, _, h, w = flow.shape
    xx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, -1, h, -1)
    yy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, -1, -1, w)
    grid = torch.cat([xx, yy], 1).to(img)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    return F.grid_sample(img, (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=True)


class OffsetDiversity(nn.Module):
    def __init__(self, in_channel, magnitude):
        super().__init__()
        self.in_channel, self.magnitude = in_channel, magnitude
        self.fusion = DeformConv2d(in_channel * 2, in_channel, kernel_size=3, padding=1, groups=2 * 8)

    def prep(self, out, flow):
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.tanh(torch.cat((o1, o2), dim=1)) * self.magnitude
        offset = offset + flow.flip(1).repeat(1, offset.size(1) // 2, 1, 1)
        return offset, torch.sigmoid(mask)

    def forward(self, x1, offset1, flow1, x2, offset2, flow2):
        offset1, mask1 = self.prep(offset1, flow1)
        offset2, mask2 = self.prep(offset2, flow2)
        return self.fusion(torch.cat((x1, x2), 1), torch.cat((offset1, offset2), 1), torch.cat((mask1, mask2), 1))


class MS_Feature(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer1 = nn.Sequential(_conv(3, 64, 3, 2), *_rbb(64))
        self.layer2 = nn.Sequential(_conv(64, 96, 3, 2), *_rbb(96))
        self.layer3 = nn.Sequential(_conv(96, 128, 3, 2), *_rbb(128))

    def forward(self, x):
        l1 = self.layer1(x)
        l2 = self.layer2(l1)
        return l1, l2, self.layer3(l2)


class FlowNET(nn.Module):
    def __init__(self):
        super().__init__()
        self.down0 = nn.Sequential(_conv(6, 32, 3, 2), *_rbb(32, 2))
        self.down1 = nn.Sequential(_conv(32, 64, 3, 2), *_rbb(64, 2))
        self.down2 = nn.Sequential(_conv(64, 128, 3, 2), *_rbb(128, 2))
        self.down3 = nn.Sequential(_conv(128, 192, 3, 2), *_rbb(192, 2))
        self.up0 = nn.Sequential(*_rbb(192, 2), subpel_conv3x3(192, 128, 2))
        self.up1 = nn.Sequential(_conv(256, 128, 1, 1), *_rbb(128, 2), subpel_conv3x3(128, 64, 2))
        self.up2 = nn.Sequential(_conv(128, 64, 1, 1), *_rbb(64, 2), subpel_conv3x3(64, 32, 2))
        self.up3 = nn.Sequential(_conv(64, 32, 1, 1), *_rbb(32, 2), subpel_conv3x3(32, 4, 2))

    def forward(self, inp):
        s0 = self.down0(inp)
        s1 = self.down1(s0)
        s2 = self.down2(s1)
        x = self.up0(self.down3(s2))
        x = self.up1(torch.cat((x, s2), 1))
        x = self.up2(torch.cat((x, s1), 1))
        return self.up3(torch.cat((x, s0), 1))


class _TemporalEnc(nn.Module):
    def __init__(self, mult, N=128, M=128):
        super().__init__()
        self.g_a1 = nn.Sequential(_conv(64 * mult, N), *_rbb(N))
        self.g_a2 = nn.Sequential(_conv(N + 96 * mult, N), *_rbb(N))
        self.g_a3 = nn.Sequential(_conv(N + 128 * mult, M), *_rbb(M))

    def forward(self, l1, l2, l3):
        y = self.g_a1(l1)
        y = self.g_a2(torch.cat([y, l2], 1))
        return self.g_a3(torch.cat([y, l3], 1))


class OffsetTemproalEnc(_TemporalEnc):
    def __init__(self):
        super().__init__(4)


class ResidualTemproalEnc(_TemporalEnc):
    def __init__(self):
        super().__init__(1)


class Reconstuctor(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer3 = nn.Sequential(*_rbb(128), subpel_conv3x3(128, 128, 2))
        self.layer2 = nn.Sequential(_conv(128 + 96, 96, 1, 1), *_rbb(96), subpel_conv3x3(96, 96, 2))
        self.layer1 = nn.Sequential(_conv(96 + 64, 64, 1, 1), *_rbb(64), subpel_conv3x3(64, 3, 2))

    def forward(self, l1, l2, l3):
        x = self.layer3(l3)
        x = self.layer2(torch.cat([l2, x], 1))
        return self.layer1(torch.cat([l1, x], 1))


GROUPS = (0, 6, 12, 24, 48)     # uneven channel groups of y (compression_bottlenecks.py:229-235)


class _Elic(JointAutoregressiveHierarchicalPriors):
    """Offset_ELIC (``enc_mult=5, dec_mult=4, out=(432,432,432)``) and Res_ELIC (``enc_mult=2, dec_mult=1,
    out=(64,96,128)``): identical bodies apart from the widths of the concatenated conditioning features."""

    def __init__(self, enc_mult, dec_mult, outs, N=128, M=128):
        super().__init__(N, M)
        self.g_a1 = nn.Sequential(_conv(64 * enc_mult, N), *_rbb(N))
        self.g_a2 = nn.Sequential(_conv(N + 96 * enc_mult, N), *_rbb(N))
        self.g_a3 = nn.Sequential(_conv(N + 128 * enc_mult, M), *_rbb(M))
        self.g_s3 = nn.Sequential(*_rbb(M), _deconv(M, N))
        self.g_o3 = nn.Sequential(_conv(N + 128 * dec_mult, N, 3, 1), *_rbb(N), _conv(N, outs[2], 3, 1))
        self.g_s2 = nn.Sequential(_conv(N + 128 * dec_mult, N, 1, 1), *_rbb(N), _deconv(N, N))
        self.g_o2 = nn.Sequential(_conv(N + 96 * dec_mult, N, 3, 1), *_rbb(N), _conv(N, outs[1], 3, 1))
        self.g_s1 = nn.Sequential(_conv(N + 96 * dec_mult, N, 1, 1), *_rbb(N), _deconv(N, N))
        self.g_o1 = nn.Sequential(_conv(N + 64 * dec_mult, N, 3, 1), *_rbb(N), _conv(N, outs[0], 3, 1))
        self.h_a = nn.Sequential(_conv(M, N, 3, 1), nn.ReLU(inplace=True), _conv(N, N), nn.ReLU(inplace=True), _conv(N, N))
        self.h_s = nn.Sequential(_deconv(N, M), nn.ReLU(inplace=True), _deconv(M, M), nn.ReLU(inplace=True), _conv(M, M, 3, 1))
        self.prior_fusion = nn.Sequential(_conv(2 * M, 2 * M, 3, 1), *_rbb(2 * M), _conv(2 * M, 2 * M, 3, 1))
        self.entropy_parameters = nn.ModuleList(
            nn.Sequential(nn.Conv2d(cin, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 8 // 3, cout * 6 // 3, 1))
            for cin, cout in [(M * 4, 6), (M * 6, 6), (M * 6, 12), (M * 6, 24), (M * 6, M - 48)])
        self.channel_context_models = nn.ModuleList(
            nn.Sequential(_conv(cin, N, 5, 1), nn.ReLU(inplace=True), _conv(N, N, 5, 1), nn.ReLU(inplace=True),
                          _conv(N, M * 2, 5, 1)) for cin in [6, 12, 24, 48])
        self.context_prediction_models = nn.ModuleList(
            CheckerboardContext(in_channels=cin, out_channels=M * 2, kernel_size=5, stride=1, padding=2)
            for cin in [6, 6, 12, 24, M - 48])
        self.levels = 5
        self.Gain = nn.Parameter(torch.ones(self.levels, M))
        self.InverseGain = nn.Parameter(torch.ones(self.levels, M))
        self.HyperGain = nn.Parameter(torch.ones(self.levels, N))
        self.InverseHyperGain = nn.Parameter(torch.ones(self.levels, N))

    def interpolate_gain(self, s):
        """compression_bottlenecks.py:296-318: geometric interpolation between the two neighbouring integer levels."""
        s = max(min(s, self.levels - 1), 0)
        upper = int(min(math.ceil(s), self.levels - 1))
        lower = int(max(math.floor(s), 0))
        mats = (self.Gain, self.HyperGain, self.InverseHyperGain, self.InverseGain)
        if upper == lower:
            return tuple(torch.abs(m[int(s)]) for m in mats)
        l = upper - s
        return tuple(torch.abs(m[upper]) ** (1 - l) * torch.abs(m[lower]) ** l for m in mats)

    def code(self, y_in1, y_in2, y_in3, f1d, f2d, f3d, temporal, s):
        gain, hypergain, invhypergain, invgain = (g.view(1, -1, 1, 1) for g in self.interpolate_gain(s))
        y = self.g_a1(y_in1)
        y = self.g_a2(torch.cat([y] + y_in2, 1))
        y = self.g_a3(torch.cat([y] + y_in3, 1))
        y = y * gain
        z = self.h_a(y) * hypergain
        lik = {}
        _, lik["z"] = self.entropy_bottleneck(z)
        z_hat = torch.round(z) * invhypergain
        hyper = self.prior_fusion(torch.cat([self.h_s(z_hat), temporal], 1))
        bounds = GROUPS + (y.shape[1],)
        for i in range(5):
            cur = y[:, bounds[i]:bounds[i + 1]]
            half = torch.round(cur).clone()
            half[:, :, 0::2, 0::2] = 0
            half[:, :, 1::2, 1::2] = 0
            ctx = self.context_prediction_models[i](half)
            ctx[:, :, 0::2, 1::2] = 0
            ctx[:, :, 1::2, 0::2] = 0
            parts = [ctx, hyper] if i == 0 else [ctx, self.channel_context_models[i - 1](torch.round(y[:, :bounds[i]])), hyper]
            scales, means = self.entropy_parameters[i](torch.cat(parts, 1)).chunk(2, 1)
            _, lik[f"y_{i}"] = self.gaussian_conditional(cur, scales, means=means)
        y_hat = torch.round(y) * invgain
        inp3 = torch.cat([self.g_s3(y_hat), f3d], 1)
        o3 = self.g_o3(inp3)
        inp2 = torch.cat([self.g_s2(inp3), f2d], 1)
        o2 = self.g_o2(inp2)
        inp1 = torch.cat([self.g_s1(inp2), f1d], 1)
        return self.g_o1(inp1), o2, o3, lik


class Offset_ELIC(_Elic):
    def __init__(self, N=128, M=128):
        super().__init__(5, 4, (27 * 8 * 2,) * 3, N, M)

    def forward(self, f1, f2, f3, f1d, f2d, f3d, offset_temp, s):
        o1, o2, o3, lik = self.code(f1, [f2], [f3], f1d, f2d, f3d, offset_temp, s)
        return {"offset3": o3, "offset2": o2, "offset1": o1, "likelihoods": lik}


class Res_ELIC(_Elic):
    def __init__(self, N=128, M=128):
        super().__init__(2, 1, (64, 96, 128), N, M)

    def forward(self, f1, f2, f3, f1d, f2d, f3d, residual_temp, s):
        r1, r2, r3, lik = self.code(torch.cat([f1, f1d], 1), [f2, f2d], [f3, f3d], f1d, f2d, f3d, residual_temp, s)
        return {"res3": r3, "res2": r2, "res1": r1, "likelihoods": lik}


ELIC_GROUPS = (0, 16, 32, 64, 128)


class ELIC(JointAutoregressiveHierarchicalPriors):
    """The I-frame codec of the ICIP2024 test loop (elic.py:85-260): 5x5 stride-2 transforms with bottleneck blocks and
    attention, hyperprior + checkerboard + channel-context entropy model over the uneven groups 16/16/32/64/M-128."""

    def __init__(self, N=192, M=320):
        super().__init__(N, M)
        self.g_a = nn.Sequential(_conv(3, N), *_rbb(N), _conv(N, N), *_rbb(N), AttentionBlock(N), _conv(N, N), *_rbb(N),
                                 _conv(N, M), AttentionBlock(M))
        self.g_s = nn.Sequential(AttentionBlock(M), _deconv(M, N), *_rbb(N), _deconv(N, N), AttentionBlock(N), *_rbb(N),
                                 _deconv(N, N), *_rbb(N), _deconv(N, 3))
        self.h_a = nn.Sequential(_conv(M, N, 3, 1), nn.ReLU(inplace=True), _conv(N, N), nn.ReLU(inplace=True), _conv(N, N))
        self.h_s = nn.Sequential(_deconv(N, M), nn.ReLU(inplace=True), _deconv(M, M * 3 // 2), nn.ReLU(inplace=True),
                                 _conv(M * 3 // 2, M * 2, 3, 1))
        self.entropy_parameters = nn.ModuleList(
            nn.Sequential(nn.Conv2d(cin, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                          nn.Conv2d(M * 8 // 3, cout * 6 // 3, 1))
            for cin, cout in [(M * 4, 16), (M * 6, 16), (M * 6, 32), (M * 6, 64), (M * 6, M - 128)])
        self.channel_context_models = nn.ModuleList(
            nn.Sequential(_conv(cin, N, 5, 1), nn.ReLU(inplace=True), _conv(N, N, 5, 1), nn.ReLU(inplace=True),
                          _conv(N, M * 2, 5, 1)) for cin in [16, 32, 64, 128])
        self.context_prediction_models = nn.ModuleList(
            CheckerboardContext(in_channels=cin, out_channels=M * 2, kernel_size=5, stride=1, padding=2)
            for cin in [16, 16, 32, 64, M - 128])

    def forward(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        lik = {}
        _, lik["z"] = self.entropy_bottleneck(z)
        hyper = self.h_s(torch.round(z))
        bounds = ELIC_GROUPS + (y.shape[1],)
        for i in range(5):
            cur = y[:, bounds[i]:bounds[i + 1]]
            half = torch.round(cur).clone()
            half[:, :, 0::2, 0::2] = 0
            half[:, :, 1::2, 1::2] = 0
            ctx = self.context_prediction_models[i](half)
            ctx[:, :, 0::2, 1::2] = 0
            ctx[:, :, 1::2, 0::2] = 0
            parts = [ctx, hyper] if i == 0 else [ctx, self.channel_context_models[i - 1](torch.round(y[:, :bounds[i]])), hyper]
            scales, means = self.entropy_parameters[i](torch.cat(parts, 1)).chunk(2, 1)
            _, lik[f"y_{i}"] = self.gaussian_conditional(cur, scales, means=means)
        return {"x_hat": self.g_s(torch.round(y)), "likelihoods": lik}


    # ---- forward_stage2 (elic.py:247-305): like forward, but the channel context and the reconstruction use the
    # mean-shifted quantisation round(y - mu) + mu of the groups already processed ----
    def forward_stage2(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        lik = {}
        _, lik["z"] = self.entropy_bottleneck(z)
        hyper = self.h_s(torch.round(z))
        bounds = ELIC_GROUPS + (y.shape[1],)
        groups = []
        for i in range(5):
            cur = y[:, bounds[i]:bounds[i + 1]]
            half = torch.round(cur).clone()
            half[:, :, 0::2, 0::2] = 0
            half[:, :, 1::2, 1::2] = 0
            ctx = self.context_prediction_models[i](half)
            ctx[:, :, 0::2, 1::2] = 0
            ctx[:, :, 1::2, 0::2] = 0
            parts = [ctx, hyper] if i == 0 else [ctx, self.channel_context_models[i - 1](torch.round(torch.cat(groups, 1))), hyper]
            scales, means = self.entropy_parameters[i](torch.cat(parts, 1)).chunk(2, 1)
            _, lik[f"y_{i}"] = self.gaussian_conditional(cur, scales, means=means)
            groups.append(torch.round(cur - means) + means)
        return {"x_hat": self.g_s(torch.cat(groups, 1)), "likelihoods": lik}

    # ---- real bitstream (elic.py:307-496): hyper string + ONE string per channel group holding the anchor symbols
    # (checkerboard positions (even row, odd col) / (odd row, even col), coded against hyper (+ channel) context only)
    # followed by the non-anchor symbols (coded with the checkerboard context of the decoded anchors) ----
    @staticmethod
    def _squeeze(t, anchor):
        b, c, h, w = t.shape
        out = torch.zeros([b, c, h, w // 2], dtype=t.dtype)
        if anchor:
            out[:, :, 0::2, :] = t[:, :, 0::2, 1::2]
            out[:, :, 1::2, :] = t[:, :, 1::2, 0::2]
        else:
            out[:, :, 0::2, :] = t[:, :, 0::2, 0::2]
            out[:, :, 1::2, :] = t[:, :, 1::2, 1::2]
        return out

    @staticmethod
    def _unsqueeze(t, anchor):
        b, c, h, w = t.shape
        out = torch.zeros([b, c, h, w * 2], dtype=t.dtype)
        if anchor:
            out[:, :, 0::2, 1::2] = t[:, :, 0::2, :]
            out[:, :, 1::2, 0::2] = t[:, :, 1::2, :]
        else:
            out[:, :, 0::2, 0::2] = t[:, :, 0::2, :]
            out[:, :, 1::2, 1::2] = t[:, :, 1::2, :]
        return out

    def _group_params(self, i, ctx, groups, hyper):
        parts = [ctx, hyper] if i == 0 else [ctx, self.channel_context_models[i - 1](torch.cat(groups, 1)), hyper]
        return self.entropy_parameters[i](torch.cat(parts, 1)).chunk(2, 1)

    def compress(self, x):
        from .cai import ans
        gc = self.gaussian_conditional
        tables = (gc._quantized_cdf.numpy(), gc._cdf_length.reshape(-1).int().numpy(), gc._offset.reshape(-1).int().numpy())
        y = self.g_a(x)
        z = self.h_a(y)
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        hyper = self.h_s(z_hat)
        bounds = ELIC_GROUPS + (y.shape[1],)
        groups, strings = [], []
        for i in range(5):
            cur = y[:, bounds[i]:bounds[i + 1]]
            zero_ctx = torch.zeros([cur.size(0), self.M * 2, cur.size(2), cur.size(3)])
            syms, idxs, parts = [], [], []
            for anchor in (True, False):
                ctx = zero_ctx if anchor else self.context_prediction_models[i](parts[0])
                scales, means = self._group_params(i, ctx, groups, hyper)
                sq, sq_s, sq_m = self._squeeze(cur, anchor), self._squeeze(scales, anchor), self._squeeze(means, anchor)
                sym = gc.quantize(sq, "symbols", sq_m)
                syms += sym.reshape(-1).tolist()
                idxs += gc.build_indexes(sq_s).reshape(-1).tolist()
                parts.append(self._unsqueeze(sym + sq_m, anchor))
            groups.append(parts[0] + parts[1])
            strings.append([ans.encode_with_indexes(syms, idxs, *tables)])
        return {"strings": [strings, z_strings], "shape": z.size()[-2:], "y_hat": groups}

    def decompress(self, strings, shape):
        from .cai import ans
        gc = self.gaussian_conditional
        tables = (gc._quantized_cdf.numpy(), gc._cdf_length.reshape(-1).int().numpy(), gc._offset.reshape(-1).int().numpy())
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        hyper = self.h_s(z_hat)
        groups = []
        for i in range(5):
            dec = ans.RansDecoder()
            dec.set_stream(strings[0][i][0])
            zero_ctx = torch.zeros([z_hat.size(0), self.M * 2, z_hat.size(2) * 4, z_hat.size(3) * 4])
            parts = []
            for anchor in (True, False):
                ctx = zero_ctx if anchor else self.context_prediction_models[i](parts[0])
                scales, means = self._group_params(i, ctx, groups, hyper)
                sq_s, sq_m = self._squeeze(scales, anchor), self._squeeze(means, anchor)
                sym = dec.decode_stream(gc.build_indexes(sq_s).reshape(-1).tolist(), *tables)
                parts.append(self._unsqueeze(torch.Tensor(sym).reshape(sq_s.shape) + sq_m, anchor))
            groups.append(parts[0] + parts[1])
        return {"x_hat": self.g_s(torch.cat(groups, 1)), "y_hat": groups}


def image_compress(im, compressors, n):
    """utils.py:306-316: intra-code a frame at quality index n -> (reconstruction, estimated size in bits)."""
    out = compressors[n].eval()(im)
    return out["x_hat"], _bits(out["likelihoods"])


def _bits(likelihoods):
    return sum(torch.log(l).sum() / (-math.log(2)) for l in likelihoods.values())


class FlowGuidedB(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = MS_Feature()
        self.flow_estimator = FlowNET()
        self.offset_temporal_conditioner = OffsetTemproalEnc()
        self.offset_compressor = Offset_ELIC()
        self.offset_diversity_l3 = OffsetDiversity(128, 10)
        self.offset_diversity_l2 = OffsetDiversity(96, 20)
        self.offset_diversity_l1 = OffsetDiversity(64, 40)
        self.residue_temporal_conditioner = ResidualTemproalEnc()
        self.residual_compressor = Res_ELIC()
        self.reconstructor = Reconstuctor()

    warp = staticmethod(warp_w3)

    @staticmethod
    def convert_scales(scale1, scale2, x=None):
        """m.py:71-82: python numbers -> fp32 [1,1,1,1], rounded to two decimals."""
        out = []
        for s in (scale1, scale2):
            s = s if torch.is_tensor(s) else torch.tensor([s])
            s = s.view(-1, 1, 1, 1).float()
            out.append(torch.round(s * 10 ** 2) / (10 ** 2))
        return out

    def estimate_flow(self, xref1, xref2, down_ratio):
        """m.py:84-102: flow on 1/(2*down_ratio) resolution frames (zero-padded to x16), brought to 1/2 resolution."""
        d1 = F.avg_pool2d(xref1, down_ratio * 2)
        d2 = F.avg_pool2d(xref2, down_ratio * 2)
        h, w = d1.shape[2:]
        pad = (0, (16 - w % 16) % 16, 0, (16 - h % 16) % 16)
        flow = self.flow_estimator(torch.cat((F.pad(d1, pad), F.pad(d2, pad)), 1))[:, :, :h, :w]
        return F.interpolate(flow, scale_factor=down_ratio, mode="bilinear", align_corners=False) * down_ratio

    def forward(self, xref1, xref2, scale1, scale2, xcur, s, down_ratio):
        b, _, h, w = xcur.shape
        num_pixels = h * w * b
        scale1, scale2 = self.convert_scales(scale1, scale2)
        flow = self.estimate_flow(xref1, xref2, down_ratio)
        fref1, fref2, fcur = (self.feature_extractor(x) for x in (xref1, xref2, xcur))
        flows, wrefs = [], []
        for lvl in range(3):                      # get_warpedrefs_at_layer, m.py:104-119
            f21, f12 = torch.chunk(flow, 2, 1)
            fc1, fc2 = f21 * scale1, f12 * scale2
            flows.append((fc1, fc2))
            wrefs.append((warp_w3(fref1[lvl], fc1), warp_w3(fref2[lvl], fc2)))
            flow = F.interpolate(flow, scale_factor=0.5, mode="bilinear", align_corners=False) * 0.5
        cond = [torch.cat((wrefs[l][0], wrefs[l][1], fref1[l], fref2[l]), 1) for l in range(3)]
        inp = [torch.cat((cond[l], fcur[l]), 1) for l in range(3)]
        off = self.offset_compressor(inp[0], inp[1], inp[2], cond[0], cond[1], cond[2],
                                     self.offset_temporal_conditioner(*cond), s)
        comp = []
        for lvl, div in ((0, self.offset_diversity_l1), (1, self.offset_diversity_l2), (2, self.offset_diversity_l3)):
            o1, o2 = torch.chunk(off[f"offset{lvl + 1}"], 2, 1)
            comp.append(div(fref1[lvl], o1, flows[lvl][0], fref2[lvl], o2, flows[lvl][1]))
        res = self.residual_compressor(fcur[0], fcur[1], fcur[2], comp[0], comp[1], comp[2],
                                       self.residue_temporal_conditioner(*comp), s)
        x_hat = self.reconstructor(comp[0] + res["res1"], comp[1] + res["res2"], comp[2] + res["res3"])
        size_offset, size_res = _bits(off["likelihoods"]), _bits(res["likelihoods"])
        return {"x_hat": x_hat, "size": size_offset + size_res, "rate": size_offset / num_pixels + size_res / num_pixels,
                "size_offset": size_offset, "size_residual": size_res}


# ---------------------------------------------------------------------------------------------------------------
# motion-adaptive flow resolution (opt_helpers.py:23-51) and the GOP-16 bookkeeping of the test loop (utils.py)
# ---------------------------------------------------------------------------------------------------------------
def prediction_flowonly(model, xcur, xref1, xref2, scale1, scale2, down_ratio):
    scale1, scale2 = model.convert_scales(scale1, scale2, xref1)
    f21, f12 = model.estimate_flow(xref1, xref2, down_ratio).chunk(2, 1)
    f21 = F.interpolate(f21, scale_factor=2, mode="bilinear", align_corners=False) * 2 * scale1
    f12 = F.interpolate(f12, scale_factor=2, mode="bilinear", align_corners=False) * 2 * scale2
    return 0.5 * warp_w3(xref1, f21) + (1 - 0.5) * warp_w3(xref2, f12)


def psnr_unit(a, b):
    return 10 * torch.log10(1.0 / torch.mean((a - b) ** 2))


def get_best_down_ratio_prediction(model, xref1, xref2, scale1, scale2, xcur, level=None, beta=None):
    best, best_ratio = 0, None
    for down_ratio in [1, 2, 4, 8, 16]:
        psnr = psnr_unit(torch.clamp(prediction_flowonly(model, xcur, xref1, xref2, scale1, scale2, down_ratio), 0, 1), xcur)
        if psnr > best:
            best, best_ratio = psnr, down_ratio
    return best_ratio, best


def get_scales(order, order1, order2):
    if order2 - order1 == 0:
        return 0, 0
    return (order - order1) / (order2 - order1), (order - order2) / (order1 - order2)


def select_references(order, buffer_order):
    """utils.py:153-177 on the order list alone: indices (into the buffer) of the past/future reference."""
    d = torch.from_numpy(np.array([abs(i - order) for i in buffer_order]))
    k = 1 if len(buffer_order) == 1 else 2
    ind = list(torch.topk(d, k, largest=False).indices.numpy())
    if k == 1:
        return int(ind[0]), int(ind[0])
    lo, hi = ind[1], ind[0]
    if buffer_order[ind[0]] < buffer_order[ind[1]]:
        lo, hi = ind[0], ind[1]
    return int(lo), int(hi)


def get_order_typ_list(intra_size, frame_number):
    """utils.py:188-221 (including its hard-coded tails for 300- and 600-frame sequences)."""
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


def val_sequence_level(frames, im_models, model, order_list, typ_list, level, h=1080, w=1920):
    """src/test.py:37-101 on in-memory frames (list of [1,3,H,W] tensors, already padded): I-frames through
    image_compress, B-frames through the flow-resolution search + FlowGuidedB.forward, references = the two buffered
    decoded frames (clamped to [0,1]) closest in display order.  Returns (psnr_list, size_list) indexed by display
    order; PSNR on uint8-rounded [:h,:w] crops, sizes in bits / (h*w) with the reference's hard-coded 1080x1920."""
    psnr_list = [0.0] * len(frames)
    size_list = [0.0] * len(frames)
    buffer, buffer_order = [], []
    with torch.no_grad():
        for order in order_list:
            frame = frames[order]
            if typ_list[order] == "I":
                dec, size = image_compress(frame, im_models, level)
            else:
                lo, hi = select_references(order, buffer_order)
                o1, o2 = buffer_order[lo], buffer_order[hi]
                s1, s2 = get_scales(order, o1, o2)
                dr, _ = get_best_down_ratio_prediction(model, buffer[lo], buffer[hi], s1, s2, frame)
                out = model(buffer[lo], buffer[hi], s1, s2, frame, level, dr)
                dec, size = out["x_hat"], out["size"]
            a = torch.round(torch.clamp(frame[0, :, :h, :w], 0, 1) * 255.0).to(torch.uint8).float()
            b = torch.round(torch.clamp(dec[0, :, :h, :w], 0, 1) * 255.0).to(torch.uint8).float()
            psnr_list[order] = (10 * torch.log10((255 ** 2) / torch.mean((b - a) ** 2))).item()
            size_list[order] = size.item() / (h * w)
            buffer, buffer_order = (buffer + [torch.clamp(dec, 0, 1)])[-32:], (buffer_order + [order])[-32:]
    return psnr_list, size_list
