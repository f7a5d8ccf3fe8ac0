"""Shared test helpers (no reference access: only fixtures + seeded weights)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name))


def frame_tensor(u8):
    """HWC uint8 -> [1,3,H,W] float32 in [0,1] (how gen_golden.py fed the reference)."""
    return torch.from_numpy(u8.astype(np.float32).transpose(2, 0, 1))[None] / 255.0


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10.0 * np.log10(1.0 / mse) if mse > 0 else float("inf")


def lhbdc_pair(seed, device=None, calibrated=False):
    """(oracle model, product model) carrying the same seeded checkpoint (``calibrated``: the trained-like variant of
    vcamd.seeding.calibrated_state_dict -- ~0.1-0.5 bpp, prediction-limited PSNR)."""
    from oracle import lhbdc as ol
    from vcamd import lhbdc
    from vcamd.seeding import calibrated_state_dict, seeded_state_dict
    prod = lhbdc.Model()
    sd = (calibrated_state_dict if calibrated else seeded_state_dict)(prod.state_dict(), seed=seed)
    prod.load_state_dict(sd)
    ora = ol.LhbdcModel().eval()
    ora.load_state_dict(sd)
    if device is not None:
        prod = prod.to(device)
    return ora, prod.eval()


def fixture_state_dict(fx, template, seed=None):
    """The B-frame checkpoint a test() loop fixture was generated on: ``fx["checkpoint"]`` = "calibrated" (trained-like
    statistics, vcamd.seeding.calibrated_state_dict) or absent / "seeded"."""
    from vcamd.seeding import calibrated_state_dict, seeded_state_dict
    fn = calibrated_state_dict if fx.get("checkpoint", "seeded") == "calibrated" else seeded_state_dict
    return fn(template, seed=fx["seed"] if seed is None else seed)


def fixture_intra_state_dict(fx, template, seed):
    """The I-frame checkpoint of a test() loop fixture (``fx["intra_checkpoint"]``: "calibrated" = the reconstructing
    linear transform codec of vcamd.seeding.calibrated_intra_state_dict; else seeded with the fixture's conv gain)."""
    from vcamd.seeding import calibrated_intra_state_dict, seeded_state_dict
    if fx.get("intra_checkpoint", "seeded") == "calibrated":
        return calibrated_intra_state_dict(template, seed=seed)
    return seeded_state_dict(template, seed=seed, conv_gain=fx["intra_conv_gain"])
