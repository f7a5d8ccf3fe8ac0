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
