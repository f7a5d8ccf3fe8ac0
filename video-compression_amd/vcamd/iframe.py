"""I-frame codec of the evaluation loop: CompressAI's ``mbt2018_mean`` architecture on MI355X.

The reference codes the GOP boundary frames with ``compressai.zoo.mbt2018_mean(quality, "mse",
pretrained=True)`` (LHBDC/test/testing.py:78-86,209; Flex-Rate.../test/testing.py:237) -- SURVEY.md section 8(f)
row 2.  This is the same mean-scale hyperprior entropy path as the B-frame compressors with 5x5 stride-2
(transposed) convolutions and GDN; module/attribute names follow compressai.models.MeanScaleHyperprior so
the zoo checkpoints' state_dict keys load unchanged.  The zoo weights are downloaded from S3 by CompressAI and
are not available offline: ``pretrained=True`` raises, benchmarks use seeded weights.
"""
import torch
import torch.nn as nn

from . import hip
from .layers import GDN, BitCounter, MeanScaleHyperprior
from .lhbdc import _require_cuda


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              output_padding=stride - 1, padding=kernel_size // 2)


class ImageMeanScaleHyperprior(MeanScaleHyperprior):
    def __init__(self, N, M, **kwargs):
        super().__init__(N=N, M=M, **kwargs)
        self.g_a = nn.Sequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = nn.Sequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True),
                                 deconv(N, N), GDN(N, inverse=True), deconv(N, 3))
        self.h_a = nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True), conv(N, N),
                                 nn.LeakyReLU(inplace=True), conv(N, N))
        self.h_s = nn.Sequential(deconv(N, M), nn.LeakyReLU(inplace=True), deconv(M, M * 3 // 2),
                                 nn.LeakyReLU(inplace=True), conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))

    def forward_device(self, x, likelihoods=None):
        """(x_hat NCHW, bits float64 device tensor [n, 2] = per image (y, z)); no host sync."""
        _require_cuda(x)
        n = x.shape[0]
        bits = BitCounter(x.device, max_rows=2 * n)
        x_hat = self.forward_t(hip.nchw_to_nhwc(x.contiguous().float()), bits, likelihoods=likelihoods)
        return hip.nhwc_to_nchw(x_hat), bits.totals().view(n, 2)

    def forward(self, x):
        """{"x_hat", "likelihoods": {"y","z"}} as compressai's MeanScaleHyperprior.forward returns (what image_compress
        reads: LHBDC/test/testing.py:58-63), plus "bits" = their -log2 sums reduced on the device."""
        lik = {}
        x_hat, tot = self.forward_device(x, likelihoods=lik)
        return {"x_hat": x_hat, "likelihoods": lik, "bits": {"y": tot[:, 0].sum(), "z": tot[:, 1].sum()}}

    def compress(self, x):
        _require_cuda(x)
        strings, (hz, wz) = self.compress_t(hip.nchw_to_nhwc(x.contiguous().float()))
        return {"strings": strings, "shape": torch.Size([hz, wz])}

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 2
        dev = self.entropy_bottleneck.quantiles.device
        # compressai's MeanScaleHyperprior.decompress clamps the reconstruction to [0, 1]
        return {"x_hat": hip.nhwc_to_nchw(self.decompress_t(strings, shape, dev, final_act=hip.ACT_CLAMP01))}


# compressai.zoo.image.cfgs["mbt2018-mean"]: quality -> (N, M)
_CFGS = {1: (128, 192), 2: (128, 192), 3: (128, 192), 4: (128, 192), 5: (192, 320), 6: (192, 320), 7: (192, 320), 8: (192, 320)}


def mbt2018_mean(quality, metric="mse", pretrained=False, progress=True, **kwargs):
    if metric not in ("mse", "ms-ssim"):
        raise ValueError(f'Invalid metric "{metric}"')
    if quality not in _CFGS:
        raise ValueError(f'Invalid quality "{quality}", should be between (1, 8)')
    if pretrained:
        raise hip.VcError("the CompressAI model-zoo weights are fetched from the network and are not available "
                          "offline; build with pretrained=False and load a state_dict")
    return ImageMeanScaleHyperprior(*_CFGS[quality], **kwargs)
