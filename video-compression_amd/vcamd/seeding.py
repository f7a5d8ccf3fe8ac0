"""Deterministic synthetic checkpoints.

Pretrained weights (Google Drive, LHBDC/README.md:5) are not available offline, so benchmarks and
parity tests run on seeded weights.  Every tensor is drawn from a counter-based generator keyed by
(seed, crc32(key)), hence independent of key order and reproducible on any host with numpy.
"""
import zlib

import numpy as np
import torch

_SKIP_SUFFIXES = ("pedestal", "bound", "target", "scale_table", "scale_bound",
                  "_offset", "_quantized_cdf", "_cdf_length", "mask")
# ICIP2024 offset heads (last convolution of Offset_ELIC.g_o1..3): a trained model predicts small refinements
# around the optical flow; unit-variance outputs would saturate tanh(.)*magnitude into +-40 px of per-tap noise,
# which no real checkpoint produces and which turns the deformable gather into a random-access benchmark.
_SMALL_HEADS = ("offset_compressor.g_o1.4.", "offset_compressor.g_o2.4.", "offset_compressor.g_o3.4.")
_HEAD_GAIN = 0.05
_GAIN_LEAVES = ("gain_matrix", "Gain", "InverseGain", "HyperGain", "InverseHyperGain")


def _rng(seed, key):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(key.encode())]))


def seeded_state_dict(template, seed=1234, conv_gain=1.0):
    """Return a new state dict with the shapes/dtypes of ``template`` (a state_dict) and seeded values.

    Convolution weights ~ N(0, gain^2/fan_in); biases ~ N(0, 0.05^2); GDN beta/gamma perturbed around
    their defaults (kept inside the valid re-parametrised range); factorised-prior MLPs perturbed
    around the CompressAI initialisation; quantiles widened per channel; gain matrices ~ U[0.5, 2].
    Buffers that hold constants or derived tables are copied unchanged.
    """
    out = {}
    for key, ref in template.items():
        leaf = key.rsplit(".", 1)[-1]
        if leaf in _SKIP_SUFFIXES or not torch.is_floating_point(ref) or ref.numel() == 0:
            out[key] = ref.clone()
            continue
        g = _rng(seed, key)
        shape = tuple(ref.shape)
        gain = conv_gain * _HEAD_GAIN if any(h in key for h in _SMALL_HEADS) else conv_gain
        if leaf in _GAIN_LEAVES:
            val = g.uniform(0.5, 2.0, size=shape)
        elif leaf == "gamma":  # re-parametrised: stored value = sqrt(gamma_eff + pedestal)
            c = shape[0]
            eff = 0.1 * np.eye(c) + g.uniform(0.0, 0.004, size=shape)
            val = np.sqrt(eff + 2.0 ** -36)
        elif leaf == "beta":
            val = np.sqrt(g.uniform(0.5, 1.5, size=shape) + 2.0 ** -36)
        elif leaf.startswith("_matrix"):
            val = ref.detach().cpu().numpy().astype(np.float64) + g.normal(0.0, 0.2, size=shape)
        elif leaf.startswith("_bias"):
            val = g.uniform(-0.5, 0.5, size=shape)
        elif leaf.startswith("_factor"):
            val = g.normal(0.0, 0.3, size=shape)
        elif leaf == "quantiles":
            med = g.uniform(-1.0, 1.0, size=(shape[0], 1, 1))
            lo = med - g.uniform(4.0, 12.0, size=(shape[0], 1, 1))
            hi = med + g.uniform(4.0, 12.0, size=(shape[0], 1, 1))
            val = np.concatenate([lo, med, hi], axis=2)
        elif ref.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            val = g.normal(0.0, gain / np.sqrt(fan_in), size=shape)
        elif ref.dim() == 1:
            val = g.normal(0.0, 0.05 * gain / conv_gain, size=shape)
        else:
            val = g.normal(0.0, 0.1, size=shape)
        out[key] = torch.from_numpy(np.asarray(val, dtype=np.float64)).to(ref.dtype)
    return out
