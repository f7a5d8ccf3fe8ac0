"""CPU restatement of the Flex-Rate B-frame codec wiring -- TEST INFRASTRUCTURE ONLY.

Follows (relative to /root/reference/Flex-Rate-Hier-Bidir-Video-Compression):
  b_model/unet.py:9-91          UNet / UNetConvBlock / UNetUpBlock   -> :class:`UNet`
  b_model/layers.py:40-73       Gain_Module                          -> :class:`Gain`
  b_model/layers.py:76-305      FlowCompressor / ResidualCompressor  -> :class:`GainedCodec`
  b_model/b_model.py:21-111     BidirFlowRef                         -> :class:`FlexModel`
  test/encode_B.py:74-109, test/decode_B.py:74-95                   -> :func:`encode_B`, :func:`decode_B`

PINNED against the real reference modules by oracle/gen_golden.py (fixtures in tests/golden/);
CompressAI blocks come from ``oracle.cai`` (parity unpinned there).  Reference quirks reproduced on
purpose: half-pixel-shifted zero-padded warp, un-gained ``y`` in
``compress``, ``clamp_(0,1)`` in ``decompress``, ``inv`` flag ignored by the gain unit.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lhbdc import HyperpriorCodec


def warp_w2(img, flow):
    """b_model.py:99-112: normalised coordinate 2*((x+u)/W - 0.5) with grid_sample defaults
    (bilinear, zeros padding, align_corners=False) == sampling at pixel (x+u-0.5, y+v-0.5)."""
    _, _, h, w = img.shape
    gx = torch.arange(w, dtype=torch.float32).view(1, 1, w).expand(1, h, w)
    gy = torch.arange(h, dtype=torch.float32).view(1, h, 1).expand(1, h, w)
    x = gx + flow[:, 0]
    y = gy + flow[:, 1]
    grid = torch.stack((2 * (x / w - 0.5), 2 * (y / h - 0.5)), dim=3)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.LeakyReLU(0.1),
                                   nn.Conv2d(cout, cout, 3, padding=1), nn.LeakyReLU(0.1))

    def forward(self, x):
        return self.block(x)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2), nn.Conv2d(cin, cout, 3, padding=1))
        self.conv_block = _ConvBlock(cin, cout)

    def forward(self, x, bridge):
        up = self.up(x)
        dy = (bridge.shape[2] - up.shape[2]) // 2
        dx = (bridge.shape[3] - up.shape[3]) // 2
        bridge = bridge[:, :, dy:dy + up.shape[2], dx:dx + up.shape[3]]
        return self.conv_block(torch.cat((up, bridge), 1))


class UNet(nn.Module):
    def __init__(self, in_channels, n_classes, depth, wf=5):
        super().__init__()
        self.down_path = nn.ModuleList()
        prev = in_channels
        for i in range(depth):
            self.down_path.append(_ConvBlock(prev, 2 ** (wf + i)))
            prev = 2 ** (wf + i)
        self.midconv = nn.Conv2d(prev, prev, 3, padding=1)
        self.up_path = nn.ModuleList()
        for i in reversed(range(depth - 1)):
            self.up_path.append(_UpBlock(prev, 2 ** (wf + i)))
            prev = 2 ** (wf + i)
        self.last = nn.Conv2d(prev, n_classes, 3, padding=1)

    def forward(self, x):
        skips = []
        for i, down in enumerate(self.down_path):
            x = down(x)
            if i != len(self.down_path) - 1:
                skips.append(x)
                x = F.avg_pool2d(x, 2)
        x = F.leaky_relu(self.midconv(x), negative_slope=0.1)
        for i, up in enumerate(self.up_path):
            x = up(x, skips[-i - 1])
        return self.last(x)


class Gain(nn.Module):
    """layers.py:40-73.  ``n`` is a list ``[int]``; ``l != 1`` interpolates |G[n]|^l * |G[n+1]|^(1-l)."""

    def __init__(self, n, N):
        super().__init__()
        self.gain_matrix = nn.Parameter(torch.ones(n, N))

    def forward(self, x, n=None, l=1):
        if l != 1:
            g = torch.abs(self.gain_matrix[n]) ** l * torch.abs(self.gain_matrix[[n[0] + 1]]) ** (1 - l)
        else:
            g = torch.abs(self.gain_matrix[n])
        return g.unsqueeze(2).unsqueeze(3) * x


class GainedCodec(HyperpriorCodec):
    """Same g_a/h_a/h_s/g_s topology as LHBDC (layers.py:79-123 vs LHBDC layers.py:48-91) with
    ``in_ch`` analysis inputs, ``out_ch`` synthesis outputs and four gain units."""

    def __init__(self, n, in_ch, out_ch, N=128, zero_last=False):
        super().__init__(in_ch, N=N)
        if out_ch != in_ch:
            from .cai.layers import subpel_conv3x3
            self.g_s[-1] = subpel_conv3x3(N, out_ch, 2)
        if zero_last:  # layers.py:125-126
            self.g_s[-1][0].weight.data.fill_(0.0)
            self.g_s[-1][0].bias.data.fill_(0.0)
        self.gain_unit = Gain(n, N)
        self.inv_gain_unit = Gain(n, N)
        self.hyper_gain_unit = Gain(n, N)
        self.hyper_inv_gain_unit = Gain(n, N)

    def forward(self, x, n=None, l=None, train=False):
        y = self.g_a(x)
        ys = self.gain_unit(y, n, l)
        z = self.h_a(ys)
        zs = self.hyper_gain_unit(z, n, l)
        z_hat, z_lik = self.entropy_bottleneck(zs)
        scales, means = self.h_s(self.hyper_inv_gain_unit(z_hat, n, l)).chunk(2, 1)
        y_hat, y_lik = self.gaussian_conditional(ys, scales, means=means)
        x_hat = self.g_s(self.inv_gain_unit(y_hat, n, l))
        return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}

    def compress(self, x, n, l):  # layers.py:154-176
        y = self.g_a(x)
        ys = self.gain_unit(y, n, l)
        z = self.h_a(ys)
        zs = self.hyper_gain_unit(z, n, l)
        z_strings = self.entropy_bottleneck.compress(zs)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        scales, means = self.h_s(self.hyper_inv_gain_unit(z_hat, n, l)).chunk(2, 1)
        idx = self.gaussian_conditional.build_indexes(scales)
        y_strings = self.gaussian_conditional.compress(y, idx, means=means)  # un-gained y (quirk B.6)
        return {"strings": [y_strings, z_strings], "shape": z.size()[-2:]}

    def decompress(self, strings, shape, n, l):  # layers.py:177-189
        assert isinstance(strings, list) and len(strings) == 2
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        scales, means = self.h_s(self.hyper_inv_gain_unit(z_hat, n, l)).chunk(2, 1)
        idx = self.gaussian_conditional.build_indexes(scales)
        y_hat = self.gaussian_conditional.decompress(strings[0], idx, means=means)
        return {"x_hat": self.g_s(self.inv_gain_unit(y_hat, n, l)).clamp_(0, 1)}


def _size(likelihoods):
    return sum(torch.log(l).sum(dim=(1, 2, 3)) / (-math.log(2)) for l in likelihoods.values())


class FlexModel(nn.Module):
    def __init__(self, n=6, N=128):
        super().__init__()
        self.flow_predictor = UNet(6, 4, 5)
        self.Mask = UNet(16, 2, 4)
        self.flow_compressor = GainedCodec(n, 19, 4, N, zero_last=True)
        self.residual_compressor = GainedCodec(n, 3, 3, N)

    backwarp = staticmethod(warp_w2)

    def process(self, x0, x1, t=0.5):
        x = torch.cat((x0, x1), 1)
        flow = self.flow_predictor(x)
        f01, f10 = flow[:, :2], flow[:, 2:4]
        ft0 = -(1 - t) * t * f01 + t * t * f10
        ft1 = (1 - t) * (1 - t) * f01 - t * (1 - t) * f10
        return ft0, ft1, torch.cat((ft0, ft1, x, warp_w2(x0, ft0), warp_w2(x1, ft1)), 1)

    def compensate(self, x_before, x_after, mv_before, mv_after, flow_hat):
        """b_model.py:61-71."""
        mvb = mv_before + flow_hat[:, :2]
        mva = mv_after + flow_hat[:, 2:4]
        xb, xa = warp_w2(x_before, mvb), warp_w2(x_after, mva)
        mask = torch.sigmoid(self.Mask(torch.cat((mvb, mva, x_before, x_after, xb, xa), 1)))
        w1, w2 = 0.5 * mask[:, 0:1], 0.5 * mask[:, 1:2]
        return (w1 * xb + w2 * xa) / (w1 + w2 + 1e-8)

    def forward(self, x_before, x_current, x_after, n=None, l=1, train=False):
        num_pixels = x_current.shape[2] * x_current.shape[3]
        mvb, mva, x_conc = self.process(x_before, x_after)
        fr = self.flow_compressor(torch.cat((x_conc, x_current), 1), n, l, train)
        x_comp = self.compensate(x_before, x_after, mvb, mva, fr["x_hat"])
        rr = self.residual_compressor(x_current - x_comp, n, l, train)
        size_flow, size_res = _size(fr["likelihoods"]), _size(rr["likelihoods"])
        return {"x_hat": x_comp + rr["x_hat"], "size": size_flow + size_res,
                "rate": size_flow / num_pixels + size_res / num_pixels}


def encode_B(model, x_before, x_current, x_after, n=None, l=1.0, train=False):
    """test/encode_B.py:74-109 -- takes a scalar ``n`` and wraps it in a list."""
    mvb, mva, x_conc = model.process(x_before, x_after)
    x_input = torch.cat((x_conc, x_current), 1)
    mv_bits = model.flow_compressor.compress(x_input, [n], l)
    flow_hat = model.flow_compressor(x_input, [n], l, train)["x_hat"]
    x_comp = model.compensate(x_before, x_after, mvb, mva, flow_hat)
    res_bits = model.residual_compressor.compress(x_current - x_comp, [n], l)
    return mv_bits, res_bits


def decode_B(model, x_before, x_after, string_flow, string_res, shape_flow, shape_res, n, l):
    """test/decode_B.py:74-95 -- scalar ``n``, wrapped in a list like the encoder."""
    mvb, mva, _ = model.process(x_before, x_after)
    flow_hat = model.flow_compressor.decompress(string_flow, shape_flow, [n], l)["x_hat"]
    x_comp = model.compensate(x_before, x_after, mvb, mva, flow_hat)
    return model.residual_compressor.decompress(string_res, shape_res, [n], l)["x_hat"] + x_comp


# ---------------------------------------------------------------------------------------------------
# the evaluation loop of Flex-Rate (test/testing.py:124-224), one video, one operating point, i_interval == gop
# ---------------------------------------------------------------------------------------------------
GOP16_ORDER = [0, 16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10, 9, 11, 14, 13, 15]
GOP16_REFS = {8: (0, 16), 4: (0, 8), 2: (0, 4), 1: (0, 2), 3: (2, 4), 6: (4, 8), 5: (4, 6), 7: (6, 8),
              12: (8, 16), 10: (8, 12), 9: (8, 10), 11: (10, 12), 14: (12, 16), 13: (12, 14), 15: (14, 16)}
GOP16_LEVELS = {8: 0, 4: 1, 2: 2, 1: 3, 3: 3, 6: 2, 5: 3, 7: 3, 12: 1, 10: 2, 9: 3, 11: 3, 14: 2, 13: 3, 15: 3}


def test_video(b_model, i_models, frames_u8, quality, gop_size=16):
    """Rows (frame_type, frame_num, psnr, size) in the order the reference's TestInfographic receives them for
    ``quality`` = (i_qual, {hierarchy level: (n, l)}): intra frames through ``i_models[i_qual]``, every B-frame through
    BidirFlowRef.forward with the (n, l) of its level.  Frames are HWC uint8 arrays."""
    import numpy as np
    from .lhbdc import _bits, pad64
    i_model, table = i_models[quality[0]], quality[1]
    h, w = frames_u8[0].shape[:2]
    x = [pad64(torch.from_numpy(f.astype(np.float32).transpose(2, 0, 1))[None] / 255.0) for f in frames_u8]

    def psnr(dec, src):
        a = np.round(np.clip(dec[0, :, :h, :w].numpy(), 0, 1) * 255.0).astype(np.uint8).astype(np.float64)
        b = np.round(np.clip(src[0, :, :h, :w].numpy(), 0, 1) * 255.0).astype(np.uint8).astype(np.float64)
        return 10 * np.log10(255.0 ** 2 / np.mean((a - b) ** 2))

    def intra(im):
        out = i_model(im)
        return out["x_hat"], _bits(out["likelihoods"]).item()

    rows = []
    with torch.no_grad():
        dec0, size0 = intra(x[0])
        rows.append(("I", 0, psnr(dec0, x[0]), size0))
        decoded = {0: dec0}
        for g in range((len(x) - 1) // gop_size):
            gop = x[g * gop_size:(g + 1) * gop_size + 1]
            dec_last, size_last = intra(gop[-1])
            decoded[16] = dec_last
            rows.append(("I", 0, psnr(dec_last, gop[-1]), size_last))
            for order in GOP16_ORDER[2:]:
                r0, r1 = GOP16_REFS[order]
                n, l = table[GOP16_LEVELS[order]]
                out = b_model(decoded[r0], gop[order], decoded[r1], n=[n], l=l, train=False)
                decoded[order] = out["x_hat"]
                rows.append(("B", order, psnr(out["x_hat"], gop[order]), out["size"].squeeze(0).item()))
            decoded = {0: dec_last}
    return rows
