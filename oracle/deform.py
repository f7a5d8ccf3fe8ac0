"""CPU restatement of torchvision.ops.deform_conv2d / DeformConv2d (modulated, grouped) -- TEST INFRASTRUCTURE ONLY.

ICIP2024 fuses its two warped references with ``DeformConv2d(2C, C, kernel_size=3, padding=1, groups=16)``
(ICIP2024/src/model/helpers.py:40,56).  torchvision is not installed in this image, so the operator is
restated from its published definition (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp, v0.12):

  out[b, o, y, x] = bias[o] + sum_{c in group(o)} sum_{k=(i,j)} W[o, c, i, j] * m[b, g(c), k, y, x]
                              * bilinear(in[b, c], y*s - p + i*d + off[b, g(c), k, 0, y, x],
                                                   x*s - p + j*d + off[b, g(c), k, 1, y, x])

with ``off`` = offset.view(B, G_off, K, 2, H, W) (dy first), ``m`` = mask.view(B, G_off, K, H, W), g(c) the
offset group of input channel c (C_in / G_off consecutive channels each), and ``bilinear`` returning 0 when
the sample lies at or beyond one pixel outside the image, corners outside the image contributing 0.
PARITY UNPINNED (third-party operator unavailable here); the reference's own wiring around it is pinned
by oracle/gen_golden.py, which gives the reference this class in place of torchvision's.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _bilinear_zero(inp, y, x):
    """inp [B, G, Cg, H, W]; y, x [B, G, Ho, Wo] -> [B, G, Cg, Ho, Wo] (torchvision bilinear_interpolate)."""
    b, g, cg, h, w = inp.shape
    ho, wo = y.shape[-2:]
    inside = (y > -1) & (y < h) & (x > -1) & (x < w)
    y0 = torch.floor(y)
    x0 = torch.floor(x)
    lh, lw = y - y0, x - x0
    hh, hw = 1 - lh, 1 - lw
    y0, x0 = y0.long(), x0.long()
    y1, x1 = y0 + 1, x0 + 1
    flat = inp.reshape(b, g, cg, h * w)

    def corner(yy, xx, ok):
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).reshape(b, g, 1, ho * wo).expand(-1, -1, cg, -1)
        v = torch.gather(flat, 3, idx).reshape(b, g, cg, ho, wo)
        return v * (ok & inside).unsqueeze(2).to(v.dtype)

    v1 = corner(y0, x0, (y0 >= 0) & (x0 >= 0))
    v2 = corner(y0, x1, (y0 >= 0) & (x1 <= w - 1))
    v3 = corner(y1, x0, (y1 <= h - 1) & (x0 >= 0))
    v4 = corner(y1, x1, (y1 <= h - 1) & (x1 <= w - 1))
    w1, w2, w3, w4 = (hh * hw).unsqueeze(2), (hh * lw).unsqueeze(2), (lh * hw).unsqueeze(2), (lh * lw).unsqueeze(2)
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    stride, padding, dilation = (tuple(v) if isinstance(v, (tuple, list)) else (v, v) for v in (stride, padding, dilation))
    b, cin, h, w = input.shape
    cout, cin_g, kh, kw = weight.shape
    groups = cin // cin_g
    k = kh * kw
    g_off = offset.shape[1] // (2 * k)
    ho = (h + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    wo = (w + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    off = offset.reshape(b, g_off, k, 2, ho, wo)
    msk = None if mask is None else mask.reshape(b, g_off, k, ho, wo)
    inp = input.reshape(b, g_off, cin // g_off, h, w)
    base_y = (torch.arange(ho, dtype=input.dtype) * stride[0] - padding[0]).view(1, 1, ho, 1)
    base_x = (torch.arange(wo, dtype=input.dtype) * stride[1] - padding[1]).view(1, 1, 1, wo)
    out = torch.zeros(b, cout, ho, wo, dtype=input.dtype)
    for i in range(kh):
        for j in range(kw):
            t = i * kw + j
            val = _bilinear_zero(inp, base_y + i * dilation[0] + off[:, :, t, 0], base_x + j * dilation[1] + off[:, :, t, 1])
            if msk is not None:
                val = val * msk[:, :, t].unsqueeze(2)
            out += F.conv2d(val.reshape(b, cin, ho, wo), weight[:, :, i:i + 1, j:j + 1], groups=groups)
    if bias is not None:
        out += bias.view(1, -1, 1, 1)
    return out


class DeformConv2d(nn.Module):
    """Parameter names and shapes of torchvision.ops.DeformConv2d (``weight`` [out, in/groups, kh, kw], ``bias``)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        pair = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)  # noqa: E731
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride, self.padding, self.dilation = pair(kernel_size), pair(stride), pair(padding), pair(dilation)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.weight.shape[1] * self.kernel_size[0] * self.kernel_size[1])
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input, offset, mask=None):
        return deform_conv2d(input, offset, self.weight, self.bias, self.stride, self.padding, self.dilation, mask)
