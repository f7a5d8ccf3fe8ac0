"""Restatement of compressai.layers (v1.1.8) -- only what the hot path touches.

Reference call sites: LHBDC/model/layers.py:6-17, Flex-Rate.../b_model/layers.py:6-16.
Described in SURVEY.md Appendix A.1/A.2.  PARITY UNPINNED (library not available here).
Attribute names are the checkpoint schema (state_dict keys) and must not change.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class LowerBound(nn.Module):
    """max(x, bound) with the bound held in a buffer named ``bound``."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return torch.max(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    """p_eff = max(p, sqrt(minimum + pedestal))**2 - pedestal, pedestal = 2**-36."""

    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        pedestal = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((float(minimum) + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return self.lower_bound(x) ** 2 - self.pedestal


class GDN(nn.Module):
    """y_i = x_i * rsqrt(beta_i + sum_j gamma_ij x_j^2)   (inverse: * sqrt)."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma_init * torch.eye(in_channels)))

    def forward(self, x):
        c = x.size(1)
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(c, c, 1, 1)
        norm = F.conv2d(x ** 2, gamma, beta)
        norm = torch.sqrt(norm) if self.inverse else torch.rsqrt(norm)
        return x * norm


def conv3x3(in_ch, out_ch, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def conv1x1(in_ch, out_ch, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=1, stride=stride)


def subpel_conv3x3(in_ch, out_ch, r=1):
    return nn.Sequential(nn.Conv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), nn.PixelShuffle(r))


class ResidualBlockWithStride(nn.Module):
    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def forward(self, x):
        out = self.gdn(self.conv2(self.leaky_relu(self.conv1(x))))
        identity = x if self.skip is None else self.skip(x)
        return out + identity


class ResidualBlockUpsample(nn.Module):
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def forward(self, x):
        out = self.igdn(self.conv(self.leaky_relu(self.subpel_conv(x))))
        return out + self.upsample(x)


class ResidualBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def forward(self, x):
        out = self.leaky_relu(self.conv2(self.leaky_relu(self.conv1(x))))
        identity = x if self.skip is None else self.skip(x)
        return out + identity


class AttentionBlock(nn.Module):
    """compressai.layers.AttentionBlock (v1.1.8), the simplified attention of Cheng et al. 2020:
    out = conv_a(x) * sigmoid(conv_b(x)) + x with conv_a = 3 residual units, conv_b = 3 residual units + conv1x1;
    residual unit: relu(conv1x1(N->N/2) -> ReLU -> conv3x3 -> ReLU -> conv1x1(N/2->N) + x).
    Imported by LHBDC/Flex (never instantiated there); used by the ELIC intra codec of ICIP2024 (elic.py:97-121)."""

    def __init__(self, N):
        super().__init__()

        class ResidualUnit(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv = nn.Sequential(conv1x1(N, N // 2), nn.ReLU(inplace=True), conv3x3(N // 2, N // 2),
                                          nn.ReLU(inplace=True), conv1x1(N // 2, N))
                self.relu = nn.ReLU(inplace=True)

            def forward(self, x):
                return self.relu(self.conv(x) + x)

        self.conv_a = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit())
        self.conv_b = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit(), conv1x1(N, N))

    def forward(self, x):
        return self.conv_a(x) * torch.sigmoid(self.conv_b(x)) + x
