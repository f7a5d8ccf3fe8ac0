"""Latent / symbol capture for a mean-scale hyperprior codec on the CPU -- TEST INFRASTRUCTURE ONLY.

Forward hooks on the sub-networks of a ``MeanScaleHyperprior``-shaped module (the reference's own
``MVCompressor`` / ``ResidualCompressor`` / ``FlowCompressor`` objects in gen_golden.py, or the oracle's
restatements in the -m gpu tests and bench.py's cpu_baseline leg) record what the entropy models see, so that
the integers the range coder consumes -- quantised symbols and scale-table indexes -- can be compared exactly
with the HIP path's, independently of any downstream amplification.

Follows (file:line relative to /root/reference):
  LHBDC/model/layers.py:72-104          forward / compress of the LHBDC compressors
  Flex-Rate.../b_model/layers.py:133-170  the gained variants (Gain_Module outputs hooked too; compress() codes the
                                          UN-gained y against the gained path's scales/means)
  CompressAI 1.1.8 EntropyModel.quantize("symbols", means) = round(x - means).int() and
  GaussianConditional.build_indexes (restated below; parity unpinned at that library, see oracle/cai).
"""
import torch

_HOOKED = ("g_a", "h_a", "h_s", "gain_unit", "hyper_gain_unit")


def build_indexes(scales, scale_table, scale_bound=0.11):
    """GaussianConditional.build_indexes: clamp to the scale bound, then count the table entries below."""
    table = torch.as_tensor(scale_table, dtype=torch.float32)
    s = torch.max(scales, torch.tensor([scale_bound], dtype=torch.float32))
    idx = torch.full(s.shape, len(table) - 1, dtype=torch.int32)
    for t in table[:-1]:
        idx -= (s <= t).int()
    return idx


class CodecTrace:
    """``with CodecTrace(codec) as tr: codec(x) / codec.compress(x)`` then ``tr.latents(...)``."""

    def __init__(self, codec):
        self.codec, self.last, self._handles = codec, {}, []
        for name in _HOOKED:
            mod = getattr(codec, name, None)
            if mod is not None:
                self._handles.append(mod.register_forward_hook(self._make(name)))
        self._handles.append(codec.g_a.register_forward_pre_hook(lambda m, args: self.last.__setitem__("x", args[0].detach())))

    def _make(self, name):
        def hook(_mod, _inp, out):
            self.last[name] = out.detach()
        return hook

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def latents(self, scale_table, code_ungained_y=False):
        """dict of the codec input ``x``, analysis outputs ``y`` / ``z`` (AFTER the gain units when the codec has them),
        the hyper-synthesis ``scales`` / ``means``, and the integers of the bitstream: ``y_sym``, ``y_idx``, ``z_sym``.
        ``code_ungained_y``: Flex compress() quantises g_a's raw output instead of the gained one."""
        c, L = self.codec, self.last
        y_raw = L["g_a"]
        y = L.get("gain_unit", y_raw)
        z = L.get("hyper_gain_unit", L["h_a"])
        scales, means = L["h_s"].chunk(2, 1)
        medians = c.entropy_bottleneck.quantiles[:, 0, 1].detach().view(1, -1, 1, 1)
        src = y_raw if code_ungained_y else y
        return {"x": L["x"], "y": y, "y_raw": y_raw, "z": z, "scales": scales, "means": means,
                "z_sym": torch.round(z - medians).int(), "y_sym": torch.round(src - means).int(),
                "y_idx": build_indexes(scales, scale_table)}


def symbol_mismatch(a, b):
    """(count, fraction) of integer entries that differ."""
    a, b = torch.as_tensor(a).reshape(-1).long(), torch.as_tensor(b).reshape(-1).long()
    if a.numel() != b.numel():
        raise ValueError(f"size mismatch: {a.numel()} vs {b.numel()}")
    n = int((a != b).sum().item())
    return n, n / max(1, a.numel())


class CallLog:
    """Outputs of a sub-module in call order (e.g. the four SPyNet calls of LHBDC's Model.forward, the mask network)."""

    def __init__(self, module):
        self.outputs = []
        self._h = module.register_forward_hook(lambda m, i, o: self.outputs.append(o.detach()))

    def close(self):
        self._h.remove()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
