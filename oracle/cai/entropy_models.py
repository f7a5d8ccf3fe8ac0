"""Restatement of compressai.entropy_models (v1.1.8): EntropyBottleneck, GaussianConditional.

Reference call sites: LHBDC/model/layers.py:8,93-116,168-191; Flex-Rate.../b_model/layers.py:154-189;
LHBDC/encode_B.py:34-35 (update(force=True)).  Semantics follow SURVEY.md Appendix A.4.
PARITY UNPINNED: the real library is not installed and the reference has no vectors for it.
Buffer / parameter names are the checkpoint schema.
"""
import math

import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ans
from .layers import LowerBound


class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    # read-only views CompressAI exposes (used by ICIP2024/src/model/elic.py:310-312)
    @property
    def offset(self):
        return self._offset

    @property
    def quantized_cdf(self):
        return self._quantized_cdf

    @property
    def cdf_length(self):
        return self._cdf_length

    # -- quantisation ------------------------------------------------------------------
    @staticmethod
    def quantize(inputs, mode, means=None):
        if mode not in ("dequantize", "symbols"):
            raise ValueError("oracle supports inference modes only")
        out = inputs.clone()
        if means is not None:
            out -= means
        out = torch.round(out)
        if mode == "dequantize":
            if means is not None:
                out += means
            return out
        return out.int()

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            out = inputs.type_as(means)
            out += means
            return out
        return inputs.type(dtype)

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            q = ans.pmf_to_quantized_cdf(prob.tolist(), self.entropy_coder_precision)
            cdf[i, : len(q)] = torch.IntTensor(q)
        return cdf

    # -- real coding -------------------------------------------------------------------
    def compress(self, inputs, indexes, means=None):
        symbols = self.quantize(inputs, "symbols", means)
        if inputs.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        strings = []
        for i in range(symbols.size(0)):
            strings.append(ans.encode_with_indexes(
                symbols[i].reshape(-1).int().numpy(), indexes[i].reshape(-1).int().numpy(),
                self._quantized_cdf.numpy(), self._cdf_length.reshape(-1).int().numpy(),
                self._offset.reshape(-1).int().numpy()))
        return strings

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        cdf = self._quantized_cdf
        outputs = cdf.new_empty(indexes.size())
        for i, s in enumerate(strings):
            vals = ans.decode_with_indexes(s, indexes[i].reshape(-1).int().numpy(), cdf.numpy(),
                                           self._cdf_length.reshape(-1).int().numpy(),
                                           self._offset.reshape(-1).int().numpy())
            outputs[i] = torch.from_numpy(vals).to(outputs.dtype).reshape(outputs[i].size())
        return self.dequantize(outputs, means, dtype)


class EntropyBottleneck(EntropyModel):
    """Factorised prior: per-channel cumulative through a 1-3-3-3-3-1 monotone MLP."""

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        widths = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / widths[i + 1]))
            matrix = torch.Tensor(channels, widths[i + 1], widths[i])
            matrix.data.fill_(init)
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(matrix))
            bias = torch.Tensor(channels, widths[i + 1], 1)
            nn.init.uniform_(bias, -0.5, 0.5)
            self.register_parameter(f"_bias{i:d}", nn.Parameter(bias))
            if i < len(self.filters):
                factor = torch.Tensor(channels, widths[i + 1], 1)
                nn.init.zeros_(factor)
                self.register_parameter(f"_factor{i:d}", nn.Parameter(factor))
        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        init = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles.data = init.repeat(self.quantiles.size(0), 1, 1)
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def _logits_cumulative(self, inputs):
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(getattr(self, f"_matrix{i:d}").detach()), logits)
            logits = logits + getattr(self, f"_bias{i:d}").detach()
            if i < len(self.filters):
                factor = getattr(self, f"_factor{i:d}").detach()
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def _likelihood(self, values):
        lower = self._logits_cumulative(values - 0.5)
        upper = self._logits_cumulative(values + 0.5)
        sign = -torch.sign(lower + upper)
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))

    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        medians = self.quantiles[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - self.quantiles[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(self.quantiles[:, 0, 2] - medians).int(), min=0)
        self._offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative(samples - 0.5)
        upper = self._logits_cumulative(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = self._pmf_to_cdf(pmf.detach(), tail_mass.detach(), pmf_length, max_length)
        self._cdf_length = pmf_length + 2
        return True

    def forward(self, x, training=None):
        # channel-major view [C, 1, N*H*W] so that the per-channel matrices broadcast
        perm = list(range(x.dim()))
        perm[0], perm[1] = 1, 0
        xp = x.permute(*perm).contiguous()
        shape = xp.size()
        values = xp.reshape(xp.size(0), 1, -1)
        outputs = self.quantize(values, "dequantize", self._get_medians())
        likelihood = self._likelihood(outputs)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        outputs = outputs.reshape(shape).permute(*perm).contiguous()
        likelihood = likelihood.reshape(shape).permute(*perm).contiguous()
        return outputs, likelihood

    def _build_indexes(self, size):
        n, c = size[0], size[1]
        view = [1, c] + [1] * (len(size) - 2)
        return torch.arange(c).view(*view).int().repeat(n, 1, *size[2:])

    def _medians_like(self, n, spatial_dims):
        med = self._get_medians().detach()
        med = med.reshape(-1, *([1] * spatial_dims)) if spatial_dims > 0 else med.reshape(-1)
        return med.unsqueeze(0).expand(n, *([-1] * (spatial_dims + 1)))

    def compress(self, x):
        indexes = self._build_indexes(x.size())
        medians = self._medians_like(x.size(0), x.dim() - 2)
        return super().compress(x, indexes, medians)

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), *size)
        indexes = self._build_indexes(output_size)
        medians = self._medians_like(len(strings), len(size))
        return super().decompress(strings, indexes, medians.dtype, medians)


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("scale_table",
                             self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))
        self.tail_mass = float(tail_mass)
        self.lower_bound_scale = LowerBound(scale_bound)

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_cumulative(inputs):
        return 0.5 * torch.erfc(float(-(2 ** -0.5)) * inputs)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = self._prepare_scale_table(scale_table)
        self.update()
        return True

    def update(self):
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * float(multiplier)).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        scale = self.scale_table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / scale)
        lower = self._standardized_cumulative((-0.5 - samples) / scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def _likelihood(self, inputs, scales, means=None):
        values = inputs - means if means is not None else inputs
        scales = self.lower_bound_scale(scales)
        values = torch.abs(values)
        upper = self._standardized_cumulative((0.5 - values) / scales)
        lower = self._standardized_cumulative((-0.5 - values) / scales)
        return upper - lower

    def forward(self, inputs, scales, means=None, training=None):
        outputs = self.quantize(inputs, "dequantize", means)
        likelihood = self._likelihood(outputs, scales, means)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        return outputs, likelihood

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes


def get_scale_table(lo=0.11, hi=256, levels=64):
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))
