"""CPU restatement of the LHBDC B-frame codec wiring -- TEST INFRASTRUCTURE ONLY.

Follows (file:line relative to /root/reference):
  LHBDC/model/flow.py:15-25    backwarp (convention W1)            -> :func:`warp_w1`
  LHBDC/model/flow.py:30-101   SPyNet ``Network``                   -> :class:`SpyNet`
  LHBDC/model/layers.py:43-191 MVCompressor / ResidualCompressor    -> :class:`HyperpriorCodec`
  LHBDC/model/layers.py:194-249 Mask U-Net                          -> :class:`MaskNet`
  LHBDC/model/m.py:20-126      ``Model``                            -> :class:`LhbdcModel`
  LHBDC/encode_B.py:39-105, decode_B.py:31-86  CLI helper functions -> module functions below
  LHBDC/encode_B.py:114-126, decode_B.py:88-104  bits_B.bin layout  -> :func:`write_container`/:func:`read_container`

PINNED against the real reference modules by oracle/gen_golden.py (fixtures in tests/golden/).
The CompressAI building blocks come from ``oracle.cai`` (parity unpinned there).
Module/attribute names reproduce the reference's state_dict keys so one state dict loads in both.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .cai.layers import (ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride, conv3x3,
                         subpel_conv3x3)
from .cai.models import MeanScaleHyperprior


# ------------------------------------------------------------------------------------------
# warping / padding primitives
# ------------------------------------------------------------------------------------------
def warp_w1(img, flow):
    """flow.py:15-25 and m.py:111-126.  Identity grid at pixel centres in [-1,1], flow scaled by
    2/(W-1), 2/(H-1), bilinear, border clamp, align_corners=False.  Net effect: output(x,y) samples
    input at (x + u*W/(W-1), y + v*H/(H-1))."""
    h, w = flow.shape[2], flow.shape[3]
    gx = torch.linspace(-1.0 + 1.0 / w, 1.0 - 1.0 / w, w).view(1, 1, 1, w).expand(-1, -1, h, -1)
    gy = torch.linspace(-1.0 + 1.0 / h, 1.0 - 1.0 / h, h).view(1, 1, h, 1).expand(-1, -1, -1, w)
    grid = torch.cat([gx, gy], 1).to(flow)
    scaled = torch.cat([flow[:, 0:1] / ((img.shape[3] - 1.0) / 2.0),
                        flow[:, 1:2] / ((img.shape[2] - 1.0) / 2.0)], 1)
    return F.grid_sample(img, (grid + scaled).permute(0, 2, 3, 1), mode="bilinear",
                         padding_mode="border", align_corners=False)


def pad64(im):
    """m.py:101-108 / encode_B.py:49-56: reflection pad bottom/right to a multiple of 64."""
    h, w = im.shape[2], im.shape[3]
    return F.pad(im, (0, (64 - w % 64) % 64, 0, (64 - h % 64) % 64), mode="reflect")


# ------------------------------------------------------------------------------------------
# SPyNet  (flow.py:30-101)
# ------------------------------------------------------------------------------------------
class _SpyBasic(nn.Module):
    def __init__(self):
        super().__init__()
        chans = (8, 32, 64, 32, 16, 2)
        layers = []
        for i in range(5):
            layers.append(nn.Conv2d(chans[i], chans[i + 1], kernel_size=7, stride=1, padding=3))
            if i < 4:
                layers.append(nn.ReLU(inplace=False))
        self.netBasic = nn.Sequential(*layers)  # conv at 0,2,4,6,8 like flow.py:52-62

    def forward(self, x):
        return self.netBasic(x)


class SpyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.netBasic = nn.ModuleList([_SpyBasic() for _ in range(6)])

    @staticmethod
    def preprocess(x):
        # flow.py:39-45 -- literal: channel 0 gets the "blue" statistics and lands last.
        c0 = (x[:, 0:1] - 0.406) / 0.225
        c1 = (x[:, 1:2] - 0.456) / 0.224
        c2 = (x[:, 2:3] - 0.485) / 0.229
        return torch.cat([c2, c1, c0], 1)

    def forward(self, first, second):
        pyr1, pyr2 = [self.preprocess(first)], [self.preprocess(second)]
        for _ in range(5):  # flow.py:83-88
            if pyr1[0].shape[2] > 32 or pyr1[0].shape[3] > 32:
                pyr1.insert(0, F.avg_pool2d(pyr1[0], 2, 2, count_include_pad=False))
                pyr2.insert(0, F.avg_pool2d(pyr2[0], 2, 2, count_include_pad=False))
        flow = pyr1[0].new_zeros([pyr1[0].shape[0], 2, pyr1[0].shape[2] // 2, pyr1[0].shape[3] // 2])
        for lvl in range(len(pyr1)):  # flow.py:92-99
            up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
            if up.shape[2] != pyr1[lvl].shape[2]:
                up = F.pad(up, [0, 0, 0, 1], mode="replicate")
            if up.shape[3] != pyr1[lvl].shape[3]:
                up = F.pad(up, [0, 1, 0, 0], mode="replicate")
            feat = torch.cat([pyr1[lvl], warp_w1(pyr2[lvl], up), up], 1)
            flow = self.netBasic[lvl](feat) + up
        return flow


# ------------------------------------------------------------------------------------------
# hyperprior codecs (layers.py:43-191) -- MV (4 ch) and residual (3 ch) share the topology
# ------------------------------------------------------------------------------------------
class HyperpriorCodec(MeanScaleHyperprior):
    def __init__(self, io_channels, N=128):
        super().__init__(N=N, M=N)
        self.g_a = nn.Sequential(
            ResidualBlockWithStride(io_channels, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            conv3x3(N, N, stride=2))
        self.h_a = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2), nn.LeakyReLU(inplace=True), conv3x3(N, N),
            nn.LeakyReLU(inplace=True), conv3x3(N, N, stride=2))
        self.h_s = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), subpel_conv3x3(N, N, 2),
            nn.LeakyReLU(inplace=True), conv3x3(N, N * 3 // 2), nn.LeakyReLU(inplace=True),
            subpel_conv3x3(N * 3 // 2, N * 3 // 2, 2), nn.LeakyReLU(inplace=True),
            conv3x3(N * 3 // 2, N * 2))
        self.g_s = nn.Sequential(
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2), ResidualBlock(N, N),
            ResidualBlockUpsample(N, N, 2), ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), subpel_conv3x3(N, io_channels, 2))

    def compress(self, x):  # layers.py:93-104
        y = self.g_a(x)
        z = self.h_a(y)
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        scales, means = self.h_s(z_hat).chunk(2, 1)
        idx = self.gaussian_conditional.build_indexes(scales)
        y_strings = self.gaussian_conditional.compress(y, idx, means=means)
        return {"strings": [y_strings, z_strings], "shape": z.size()[-2:]}

    def decompress(self, strings, shape):  # layers.py:106-116
        assert isinstance(strings, list) and len(strings) == 2
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        scales, means = self.h_s(z_hat).chunk(2, 1)
        idx = self.gaussian_conditional.build_indexes(scales)
        y_hat = self.gaussian_conditional.decompress(strings[0], idx, means=means)
        return {"x_hat": self.g_s(y_hat)}


# ------------------------------------------------------------------------------------------
# Mask U-Net (layers.py:194-249)
# ------------------------------------------------------------------------------------------
class MaskNet(nn.Module):
    def __init__(self, ch=32):
        super().__init__()

        def c(i, o, k):
            return nn.Conv2d(i, o, kernel_size=k, stride=1, padding=k // 2)

        self.conv1, self.conv2, self.conv3 = c(6, ch, 5), c(ch, ch * 2, 5), c(ch * 2, ch * 4, 3)
        self.bottleneck = c(ch * 4, ch * 4, 3)
        self.deconv1 = c(ch * 8, ch * 4, 3)
        self.deconv2 = c(ch * 4 + ch * 2, ch * 2, 5)
        self.deconv3 = c(ch * 2 + ch, ch, 5)
        self.conv4 = c(ch, 1, 5)

    def forward(self, x):
        s1 = F.relu(self.conv1(x))
        s2 = F.relu(self.conv2(F.max_pool2d(s1, 2, 2)))
        s3 = F.relu(self.conv3(F.max_pool2d(s2, 2, 2)))
        x = F.relu(self.bottleneck(F.max_pool2d(s3, 2, 2)))
        for skip, conv in ((s3, self.deconv1), (s2, self.deconv2), (s1, self.deconv3)):
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
            x = F.relu(conv(torch.cat([x, skip], 1)))
        return torch.sigmoid(self.conv4(x))


# ------------------------------------------------------------------------------------------
# the B-frame model (m.py:20-126)
# ------------------------------------------------------------------------------------------
def _bits(likelihoods):
    return sum(torch.log(l).sum() / (-math.log(2)) for l in likelihoods.values())


def _rate(likelihoods, num_pixels):
    return sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in likelihoods.values())


class LhbdcModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.FlowNet = SpyNet()
        self.mv_compressor = HyperpriorCodec(4)
        self.residual_compressor = HyperpriorCodec(3)
        self.masknet = MaskNet()
        self.upsample_flow = nn.Upsample(scale_factor=4, mode="bilinear")

    pad = staticmethod(pad64)
    backwarp = staticmethod(warp_w1)

    def predictor_flows(self, x_before, x_after, cli_quirk=False):
        """m.py:38-44.  With ``cli_quirk`` reproduces encode_B.py:78-79 / decode_B.py:69-70 where both
        padded predictors end up equal to pad(flow_ab)."""
        flow_ba = F.avg_pool2d(self.FlowNet(x_before, x_after) / 2.0, 4)
        flow_ab = F.avg_pool2d(self.FlowNet(x_after, x_before) / 2.0, 4)
        hh, ww = flow_ab.shape[2], flow_ab.shape[3]
        if cli_quirk:
            flow_ba = pad64(flow_ab)
            flow_ab = pad64(flow_ba)
        else:
            flow_ba, flow_ab = pad64(flow_ba), pad64(flow_ab)
        return flow_ba, flow_ab, hh, ww

    def flow_difference(self, x_before, x_current, x_after, flow_ba, flow_ab):
        flow_cb = pad64(F.avg_pool2d(self.FlowNet(x_current, x_before), 4))
        flow_ca = pad64(F.avg_pool2d(self.FlowNet(x_current, x_after), 4))
        return torch.cat([flow_cb - flow_ab, flow_ca - flow_ba], 1)

    def predict_frame(self, x_before, x_after, mv_hat, flow_ba, flow_ab, hh, ww):
        """m.py:55-65: add predictors back, crop, x4 bilinear, warp both references, mask-blend."""
        cb_hat, ca_hat = torch.chunk(mv_hat, 2, dim=1)
        cb_hat = self.upsample_flow((cb_hat + flow_ab)[:, :, :hh, :ww])
        ca_hat = self.upsample_flow((ca_hat + flow_ba)[:, :, :hh, :ww])
        fw, bw = warp_w1(x_before, cb_hat), warp_w1(x_after, ca_hat)
        mask = self.masknet(torch.cat([fw, bw], 1)).repeat([1, 3, 1, 1])
        return mask * fw + (1.0 - mask) * bw

    def forward(self, x_before, x_current, x_after, train=False):
        n, _, h, w = x_current.size()
        num_pixels = n * h * w
        flow_ba, flow_ab, hh, ww = self.predictor_flows(x_before, x_after)
        diff = self.flow_difference(x_before, x_current, x_after, flow_ba, flow_ab)
        mv = self.mv_compressor(diff)
        pred = self.predict_frame(x_before, x_after, mv["x_hat"], flow_ba, flow_ab, hh, ww)
        res = self.residual_compressor(x_current - pred)
        x_hat = res["x_hat"] + pred
        size_flow, size_res = _bits(mv["likelihoods"]), _bits(res["likelihoods"])
        rate = (_rate(mv["likelihoods"], num_pixels) + _rate(res["likelihoods"], num_pixels)) / 2.0  # m.py:96,98 -- halved
        if train:
            return x_hat, rate
        return x_hat, rate, size_flow.item() + size_res.item()


# ------------------------------------------------------------------------------------------
# CLI functions (encode_B.py:71-105, decode_B.py:63-86)
# ------------------------------------------------------------------------------------------
def encode_B(model, x_after, x_current, x_before):
    """NB the argument order (encode_B.py:71) and the predictor quirk (:78-79)."""
    flow_ba, flow_ab, hh, ww = model.predictor_flows(x_before, x_after, cli_quirk=True)
    diff = model.flow_difference(x_before, x_current, x_after, flow_ba, flow_ab)
    mv = model.mv_compressor(diff)
    mv_bits = model.mv_compressor.compress(diff)
    pred = model.predict_frame(x_before, x_after, mv["x_hat"], flow_ba, flow_ab, hh, ww)
    res_bits = model.residual_compressor.compress(x_current - pred)
    return mv_bits, res_bits


def decode_B(x_before, x_after, model, string_flow, string_res, shape_flow, shape_res):
    flow_ba, flow_ab, hh, ww = model.predictor_flows(x_before, x_after, cli_quirk=True)
    mv_hat = model.mv_compressor.decompress(string_flow, shape_flow)["x_hat"]
    pred = model.predict_frame(x_before, x_after, mv_hat, flow_ba, flow_ab, hh, ww)
    return model.residual_compressor.decompress(string_res, shape_res)["x_hat"] + pred


def process_frame(img_u8):
    """encode_B.py:58-64: HWC uint8 -> [1,3,H,W] float in [0,1], reflection-padded to x64."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8.astype(np.float64).transpose(2, 0, 1)))[None]
    return pad64((x.float()) / 255.0)


def float_to_uint8(image_chw):
    """encode_B.py:44-47."""
    return np.round(np.clip(image_chw, 0, 1) * 255.0).astype(np.uint8).transpose(1, 2, 0)


def write_container(lmbda, mv_bits, res_bits):
    """encode_B.py:114-126: 24-byte little-endian header + 4 strings (last one runs to EOF)."""
    out = bytearray()
    out += np.array(lmbda, dtype=np.uint32).tobytes()
    out += np.array(tuple(mv_bits["shape"]), dtype=np.uint16).tobytes()
    out += np.array(len(mv_bits["strings"][0][0]), dtype=np.uint32).tobytes()
    out += np.array(len(mv_bits["strings"][1][0]), dtype=np.uint32).tobytes()
    out += np.array(tuple(res_bits["shape"]), dtype=np.uint16).tobytes()
    out += np.array(len(res_bits["strings"][0][0]), dtype=np.uint32).tobytes()
    for s in (mv_bits["strings"][0][0], mv_bits["strings"][1][0],
              res_bits["strings"][0][0], res_bits["strings"][1][0]):
        out += s
    return bytes(out)


def read_container(blob):
    """decode_B.py:88-104 (with the removed ``np.int`` read as plain int)."""
    u32 = lambda o: int(np.frombuffer(blob[o:o + 4], dtype=np.uint32)[0])  # noqa: E731
    lmbda = u32(0)
    shape_mv = torch.Size(np.frombuffer(blob[4:8], dtype=np.uint16).astype(int))
    len0_mv, len1_mv = u32(8), u32(12)
    shape_res = torch.Size(np.frombuffer(blob[16:20], dtype=np.uint16).astype(int))
    len0_res = u32(20)
    p = 24
    s0 = blob[p:p + len0_mv]; p += len0_mv
    s1 = blob[p:p + len1_mv]; p += len1_mv
    s2 = blob[p:p + len0_res]; p += len0_res
    s3 = blob[p:]
    return lmbda, [[s0], [s1]], [[s2], [s3]], shape_mv, shape_res


# ---------------------------------------------------------------------------------------------------
# the evaluation loop that defines the metric (LHBDC/test/testing.py:88-196, one video, i_interval == gop)
# ---------------------------------------------------------------------------------------------------
GOP8_ORDER = [0, 8, 4, 2, 1, 3, 6, 5, 7]
GOP8_REFS = {4: (0, 8), 2: (0, 4), 1: (0, 2), 3: (2, 4), 6: (4, 8), 5: (4, 6), 7: (6, 8)}


def harness_frames(seed, video, count, h=180, w=180):
    """Deterministic uint8 test clip (integer arithmetic only, so every host regenerates the same bytes): a box-blurred
    random texture translating by (2, 3) pixels per frame plus +-2 levels of noise."""
    import numpy as np
    g = np.random.Generator(np.random.PCG64([int(seed), int(video)]))
    big = g.integers(0, 256, size=(h + 64, w + 96, 3), dtype=np.int64)
    k = 5
    c = np.cumsum(np.cumsum(np.pad(big, ((k, 0), (k, 0), (0, 0))), 0), 1)
    blur = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) // (k * k)
    out = []
    for t in range(count):
        f = blur[2 * t:2 * t + h, 3 * t:3 * t + w] + g.integers(-2, 3, size=(h, w, 3))
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return out


def test_video(b_model, i_model, frames_u8, gop_size=8):
    """testing.py:125-188 for one video whose frame list is I B*7 I [B*7 I ...] (test_size GOPs, i_interval == gop_size):
    returns rows (frame_type, frame_num, psnr, size) in the order the reference's TestInfographic receives them.
    Frames are HWC uint8 arrays; padding (reflection to x64) and the uint8 PSNR on the unpadded crop as in test/utils.py."""
    import numpy as np
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
            decoded[8] = dec_last
            rows.append(("I", 0, psnr(dec_last, gop[-1]), size_last))
            for order in GOP8_ORDER[2:]:
                r0, r1 = GOP8_REFS[order]
                dec, _, size = b_model(decoded[r0], gop[order], decoded[r1], False)
                decoded[order] = dec
                rows.append(("B", order, psnr(dec, gop[order]), size))
            decoded = {0: dec_last}
    return rows
