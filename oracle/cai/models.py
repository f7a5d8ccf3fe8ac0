"""Restatement of compressai.models.MeanScaleHyperprior (v1.1.8) as the reference subclasses it
(LHBDC/model/layers.py:43,118; Flex-Rate.../b_model/layers.py:76,192).  SURVEY.md Appendix A.3.
PARITY UNPINNED (third-party library unavailable here)."""
import torch
import torch.nn as nn

from .entropy_models import EntropyBottleneck, GaussianConditional, get_scale_table

_CDF_BUFFERS = ("_quantized_cdf", "_offset", "_cdf_length")


def _resize_registered_buffers(module, prefix, names, state_dict):
    """CompressAI resizes the (variable-size) CDF buffers to the checkpoint's shapes before the
    ordinary strict load -- needed for checkpoints saved after update()."""
    for name in names:
        key = f"{prefix}.{name}"
        if key not in state_dict:
            raise RuntimeError(f'missing key "{key}" in state_dict')
        src = state_dict[key]
        buf = getattr(module, name)
        if buf.shape != src.shape:
            setattr(module, name, torch.empty(src.shape, dtype=src.dtype))


class CompressionModel(nn.Module):
    def __init__(self, entropy_bottleneck_channels, init_weights=True):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)

    def update(self, force=False):
        updated = False
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        _resize_registered_buffers(self.entropy_bottleneck, "entropy_bottleneck", _CDF_BUFFERS, state_dict)
        return super().load_state_dict(state_dict, strict=strict)


class MeanScaleHyperprior(CompressionModel):
    """g_a/h_a/h_s/g_s are always replaced by the reference subclasses, so none are built here."""

    def __init__(self, N, M, **kwargs):
        super().__init__(entropy_bottleneck_channels=N, **kwargs)
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = int(N), int(M)

    def forward(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_likelihoods = self.entropy_bottleneck(z)
        scales_hat, means_hat = self.h_s(z_hat).chunk(2, 1)
        y_hat, y_likelihoods = self.gaussian_conditional(y, scales_hat, means=means_hat)
        return {"x_hat": self.g_s(y_hat), "likelihoods": {"y": y_likelihoods, "z": z_likelihoods}}

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= super().update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        _resize_registered_buffers(self.gaussian_conditional, "gaussian_conditional",
                                   _CDF_BUFFERS + ("scale_table",), state_dict)
        return super().load_state_dict(state_dict, strict=strict)


# ---------------------------------------------------------------------------------------------------
# I-frame codec: compressai.zoo.mbt2018_mean = MeanScaleHyperprior(N, M) with its default networks
# (used at LHBDC/test/testing.py:209).  PARITY UNPINNED like the rest of this package.
# ---------------------------------------------------------------------------------------------------
def _conv(i, o, kernel_size=5, stride=2):
    return nn.Conv2d(i, o, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def _deconv(i, o, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(i, o, kernel_size=kernel_size, stride=stride, output_padding=stride - 1,
                              padding=kernel_size // 2)


class ImageMeanScaleHyperprior(MeanScaleHyperprior):
    def __init__(self, N, M, **kwargs):
        from .layers import GDN
        super().__init__(N, M, **kwargs)
        self.g_a = nn.Sequential(_conv(3, N), GDN(N), _conv(N, N), GDN(N), _conv(N, N), GDN(N), _conv(N, M))
        self.g_s = nn.Sequential(_deconv(M, N), GDN(N, inverse=True), _deconv(N, N), GDN(N, inverse=True),
                                 _deconv(N, N), GDN(N, inverse=True), _deconv(N, 3))
        self.h_a = nn.Sequential(_conv(M, N, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True), _conv(N, N),
                                 nn.LeakyReLU(inplace=True), _conv(N, N))
        self.h_s = nn.Sequential(_deconv(N, M), nn.LeakyReLU(inplace=True), _deconv(M, M * 3 // 2),
                                 nn.LeakyReLU(inplace=True), _conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))

    def compress(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        scales, means = self.h_s(z_hat).chunk(2, 1)
        idx = self.gaussian_conditional.build_indexes(scales)
        y_strings = self.gaussian_conditional.compress(y, idx, means=means)
        return {"strings": [y_strings, z_strings], "shape": z.size()[-2:]}

    def decompress(self, strings, shape):
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        scales, means = self.h_s(z_hat).chunk(2, 1)
        idx = self.gaussian_conditional.build_indexes(scales)
        y_hat = self.gaussian_conditional.decompress(strings[0], idx, means=means)
        return {"x_hat": self.g_s(y_hat).clamp_(0, 1)}


def mbt2018_mean(quality, metric="mse", pretrained=False):
    n, m = (128, 192) if quality <= 4 else (192, 320)
    return ImageMeanScaleHyperprior(n, m)


# ---------------------------------------------------------------------------------------------------
# compressai.models.JointAutoregressiveHierarchicalPriors (v1.1.8) -- the parent class of the ICIP2024
# compressors (ICIP2024/src/model/compression_bottlenecks.py:72,322).  Those subclasses replace h_a, h_s
# and entropy_parameters and never call the parent's forward; g_a, g_s and context_prediction survive
# only as (dead) entries of the state_dict, which a strict checkpoint load still needs.
# PARITY UNPINNED like the rest of this package.
# ---------------------------------------------------------------------------------------------------
class MaskedConv2d(nn.Conv2d):
    """Type-A/B masked convolution (PixelCNN); buffer ``mask`` is part of the state_dict."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        if mask_type not in ("A", "B"):
            raise ValueError(f'Invalid "mask_type" value "{mask_type}"')
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.size()
        self.mask[:, :, h // 2, w // 2 + (mask_type == "B"):] = 0
        self.mask[:, :, h // 2 + 1:] = 0

    def forward(self, x):
        self.weight.data *= self.mask
        return super().forward(x)


class JointAutoregressiveHierarchicalPriors(MeanScaleHyperprior):
    def __init__(self, N=192, M=192, **kwargs):
        from .layers import GDN
        super().__init__(N=N, M=M, **kwargs)
        self.g_a = nn.Sequential(_conv(3, N), GDN(N), _conv(N, N), GDN(N), _conv(N, N), GDN(N), _conv(N, M))
        self.g_s = nn.Sequential(_deconv(M, N), GDN(N, inverse=True), _deconv(N, N), GDN(N, inverse=True),
                                 _deconv(N, N), GDN(N, inverse=True), _deconv(N, 3))
        self.h_a = nn.Sequential(_conv(M, N, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True),
                                 _conv(N, N, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                 _conv(N, N, stride=2, kernel_size=5))
        self.h_s = nn.Sequential(_deconv(N, M, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                 _deconv(M, M * 3 // 2, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                 _conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))
        self.entropy_parameters = nn.Sequential(
            nn.Conv2d(M * 12 // 3, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = int(N), int(M)


class Cheng2020Anchor(nn.Module):
    """Imported by ICIP2024/src/model/compression_bottlenecks.py:7 and never instantiated on the path."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("Cheng2020Anchor is not on the hot path")
