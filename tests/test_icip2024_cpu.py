"""CPU: the ICIP2024 FlowGuidedB restatement (oracle/icip2024.py, oracle/deform.py) against the fixtures recorded
from the reference, and the host side of the product path (state_dict schema, GOP-16 bookkeeping, gain vectors)."""
import json
import os

import pytest
import torch

from helpers import GOLDEN, frame_tensor, load_fixture
from oracle import deform as odeform, icip2024 as oi
from vcamd import icip2024
from vcamd.seeding import seeded_state_dict


@pytest.fixture(scope="module")
def oracle_model():
    fx = load_fixture("icip2024_forward_a.npz")
    m = oi.FlowGuidedB().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=int(fx["seed"])))
    return m


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_forward_matches_reference_fixture(oracle_model, tag):
    fx = load_fixture("icip2024_forward_a.npz")
    x1, xc, x2 = (frame_tensor(fx[k]) for k in ("ref_1", "current", "ref_2"))
    s1, s2, lvl, dr = (float(v) for v in fx[f"cfg_{tag}"])
    with torch.no_grad():
        out = oracle_model(x1, x2, s1, s2, xc, lvl if lvl != int(lvl) else int(lvl), int(dr))
        flow = oracle_model.estimate_flow(x1, x2, int(dr))
    # same container and thread count as the recording -> identical; other hosts may reorder fp32 sums slightly
    assert (flow - torch.from_numpy(fx[f"flow_{tag}"])).abs().max().item() < 1e-4
    ref = torch.from_numpy(fx[f"x_hat_{tag}"])
    assert ((out["x_hat"] - ref).abs() / (1 + ref.abs())).max().item() < 5e-3
    assert abs(out["size"].item() - float(fx[f"size_{tag}"])) / float(fx[f"size_{tag}"]) < 1e-3
    assert abs(out["rate"].item() - float(fx[f"rate_{tag}"])) / float(fx[f"rate_{tag}"]) < 1e-3


def test_oracle_down_ratio_search_matches_reference_fixture(oracle_model):
    fx = load_fixture("icip2024_forward_a.npz")
    x1, xc, x2 = (frame_tensor(fx[k]) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        for dr in (1, 2, 4, 8, 16):
            pred = oi.prediction_flowonly(oracle_model, xc, x1, x2, 0.5, 0.5, dr)
            assert (pred - torch.from_numpy(fx[f"pred_dr{dr}"])).abs().max().item() < 1e-4
        best, psnr = oi.get_best_down_ratio_prediction(oracle_model, x1, x2, 0.5, 0.5, xc)
    assert best == int(fx["best_down_ratio"]) and abs(psnr.item() - float(fx["best_pred_psnr"])) < 1e-3


def test_deform_conv_restatement_properties():
    """zero offsets + unit mask == grouped convolution; integer offsets == shifted input; mask scales linearly;
    samples at or beyond one pixel outside contribute nothing."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 9, 11, generator=g)
    w = torch.randn(16, 2, 3, 3, generator=g)
    b = torch.randn(16, generator=g)
    off = torch.zeros(2, 2 * 9 * 16, 9, 11)
    m = torch.ones(2, 9 * 16, 9, 11)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1, groups=16)
    assert (odeform.deform_conv2d(x, off, w, b, padding=(1, 1), mask=m) - ref).abs().max().item() < 1e-5
    assert (odeform.deform_conv2d(x, off, w, None, padding=(1, 1), mask=0.5 * m) - 0.5 * (ref - b.view(1, -1, 1, 1))).abs().max().item() < 1e-5
    off_dx = off.clone()
    off_dx[:, 1::2] = 2.0                                     # every tap two pixels to the right
    xs = torch.zeros_like(x)
    xs[..., :-2] = x[..., 2:]
    ref_s = torch.nn.functional.conv2d(xs, w, b, padding=1, groups=16)
    got = odeform.deform_conv2d(x, off_dx, w, b, padding=(1, 1), mask=m)
    assert (got - ref_s)[..., 1:-3].abs().max().item() < 1e-5
    far = off.clone()
    far[:, 0::2] = 100.0
    assert (odeform.deform_conv2d(x, far, w, b, padding=(1, 1), mask=m) - b.view(1, -1, 1, 1)).abs().max().item() == 0.0


def test_product_state_dict_schema_equals_reference_schema():
    ref = [l.strip() for l in open(os.path.join(GOLDEN, "icip2024_state_schema.txt"))]
    mine = sorted(f"{k} {list(v.shape)}" for k, v in icip2024.FlowGuidedB().state_dict().items())
    assert mine == ref and len(mine) == 1126
    sd = seeded_state_dict(icip2024.FlowGuidedB().state_dict(), seed=3)
    icip2024.FlowGuidedB().load_state_dict(sd, strict=True)
    oi.FlowGuidedB().load_state_dict(sd, strict=True)


def test_gop16_bookkeeping_matches_reference_lists():
    book = json.load(open(os.path.join(GOLDEN, "icip2024_gop16_bookkeeping.json")))
    for n_frames, rec in book["order_typ"].items():
        for mod in (icip2024, oi):
            order, typ = mod.get_order_typ_list(16, int(n_frames))
            assert [int(v) for v in order] == rec["order"] and "".join(typ) == rec["typ"]
        order, typ = icip2024.get_order_typ_list(16, int(n_frames))
        buf, buf_order, picks = [], [], []
        for o in order:
            if typ[o] == "B":
                _, _, o1, o2 = icip2024.select_references(None, o, buf, buf_order)
                s1, s2 = icip2024.get_scales(o, o1, o2)
                picks.append([int(o), int(o1), int(o2), float(s1), float(s2)])
            buf, buf_order = icip2024.update_buffer(buf, buf_order, o, o)
        assert picks == book["refs"][n_frames]


def test_gain_interpolation_matches_oracle():
    prod, ora = icip2024.Offset_ELIC(), oi.Offset_ELIC()
    sd = seeded_state_dict(prod.state_dict(), seed=11)
    prod.load_state_dict(sd)
    ora.load_state_dict(sd)
    for s in (0, 1, 2.5, 3.25, 4, 7, -1):
        for a, b in zip(prod.interpolate_gain(s), ora.interpolate_gain(s)):
            assert torch.allclose(a, b.detach(), rtol=1e-6, atol=0)
    assert icip2024.FlowGuidedB().convert_scales(0.33333, -0.125) == tuple(
        float(v) for v in oi.FlowGuidedB.convert_scales(0.33333, -0.125))


def test_no_cpu_path():
    m = icip2024.FlowGuidedB()
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(Exception):
        m(x, x, 0.5, 0.5, x, 1, 1)


def test_reference_import_paths_resolve():
    """`from src.model import m` (ICIP2024/main.py -> src/test.py:31-32) resolves to the HIP-backed classes."""
    from src.model import compression_bottlenecks, helpers, m
    from src import opt_helpers, utils
    assert m.FlowGuidedB is icip2024.FlowGuidedB and helpers.OffsetDiversity is icip2024.OffsetDiversity
    assert compression_bottlenecks.Offset_ELIC is icip2024.Offset_ELIC
    assert opt_helpers.get_best_down_ratio_prediction is icip2024.get_best_down_ratio_prediction
    assert utils.get_order_typ_list(16, 17)[0][:3] == [0, 16, 8]


def test_oracle_elic_matches_reference_fixture():
    fx = load_fixture("icip2024_elic_a.npz")
    m = oi.ELIC().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=int(fx["seed"]), conv_gain=float(fx["conv_gain"])))
    with torch.no_grad():
        dec, size = oi.image_compress(frame_tensor(fx["current"]), [m], 0)
    ref = torch.from_numpy(fx["x_hat"])
    assert ((dec - ref).abs() / (1 + ref.abs())).max().item() < 5e-3
    assert abs(size.item() - float(fx["size"])) / float(fx["size"]) < 1e-3


def test_product_elic_schema_equals_reference_schema():
    ref = [l.strip() for l in open(os.path.join(GOLDEN, "icip2024_elic_state_schema.txt"))]
    mine = sorted(f"{k} {list(v.shape)}" for k, v in icip2024.ELIC().state_dict().items())
    assert mine == ref and len(mine) == 387
    sd = seeded_state_dict(icip2024.ELIC().state_dict(), seed=5)
    icip2024.ELIC().load_state_dict(sd, strict=True)
    oi.ELIC().load_state_dict(sd, strict=True)


def test_oracle_sequence_loop_matches_reference_fixture():
    """val_sequence_level (src/test.py:37-101) on the 20-frame clip: one full GOP-16 plus an irregular 3-frame tail."""
    fx = load_fixture("icip2024_sequence_a.npz")
    clip = [torch.from_numpy(f.astype("float32"))[None] / 255.0 for f in fx["clip_u8"]]
    fa, fe = load_fixture("icip2024_forward_a.npz"), load_fixture("icip2024_elic_a.npz")
    model = oi.FlowGuidedB().eval()
    model.load_state_dict(seeded_state_dict(model.state_dict(), seed=int(fa["seed"])))
    intra = oi.ELIC().eval()
    intra.load_state_dict(seeded_state_dict(intra.state_dict(), seed=int(fe["seed"]), conv_gain=float(fe["conv_gain"])))
    order, typ = oi.get_order_typ_list(16, len(clip))
    assert order == fx["order"].tolist() and "".join(typ) == str(fx["typ"])
    first = order[:6]                                    # I, I, then the first B-frames (the full clip takes a minute)
    psnr, size = oi.val_sequence_level(clip, [intra] * 5, model, first, typ, int(fx["level"]))
    for o in first:
        assert abs(psnr[o] - fx["psnr"][o]) < 1e-2 and abs(size[o] - fx["size"][o]) / fx["size"][o] < 1e-3


def test_oracle_elic_bitstream_reproduces_the_reference_fixture():
    """oracle.icip2024.ELIC.compress / decompress / forward_stage2 against what the reference's own methods produced
    (icip2024_elic_codec_a.npz, written by oracle/gen_golden.py with max|d| = 0 and identical strings)."""
    import numpy as np
    import torch
    from helpers import frame_tensor, load_fixture
    from oracle import icip2024 as oi
    from vcamd.seeding import seeded_state_dict
    fx = load_fixture("icip2024_elic_codec_a.npz")
    ora = oi.ELIC().eval()
    ora.load_state_dict(seeded_state_dict(ora.state_dict(), seed=int(fx["seed"]), conv_gain=float(fx["conv_gain"])))
    ora.update(force=True)
    x = frame_tensor(fx["current"])
    with torch.no_grad():
        enc = ora.compress(x)
        for g in range(5):
            assert enc["strings"][0][g][0] == fx[f"y_string_{g}"].tobytes(), g
        assert enc["strings"][1][0] == fx["z_string"].tobytes()
        dec = ora.decompress(enc["strings"], enc["shape"])
        s2 = ora.forward_stage2(x)
    assert np.array_equal(torch.cat(dec["y_hat"], 1).numpy(), fx["y_hat"])
    assert np.array_equal(dec["x_hat"].numpy(), fx["decoded"])
    assert np.array_equal(s2["x_hat"].numpy(), fx["stage2_x_hat"])
