"""-m gpu: the fp16 MFMA conv path (BASELINE.json configs[4]) at model level.
Not bit-comparable with the fp32 CPU path by construction: judged on PSNR / bit-count tolerance."""
import pytest
import torch

from helpers import frame_tensor, load_fixture, psnr

pytestmark = pytest.mark.gpu

PSNR_TOL_DB = 0.05      # reconstruction PSNR vs the fp32 reference output
BITS_TOL = 0.02         # relative difference of the estimated size


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _fp16_model(dev):
    from vcamd import hip, lhbdc
    from vcamd.seeding import seeded_state_dict
    hip.set_conv_precision("fp16")
    try:
        m = lhbdc.Model()
        m.load_state_dict(seeded_state_dict(m.state_dict(), seed=1234))
        m = m.to(dev).eval()
        x = torch.zeros(1, 3, 192, 256, device=dev)
        with torch.no_grad():
            m(x, x, x, False)            # packs every layer while the fp16 mode is active
    finally:
        hip.set_conv_precision("fp32")
    return m


def test_lhbdc_forward_fp16_against_reference_fixture(dev):
    m = _fp16_model(dev)
    n16 = sum(1 for mod in m.modules() if getattr(mod, "_packed", None) is not None)
    assert n16 > 0
    fx = load_fixture("lhbdc_forward_a.npz")
    xb, xc, xa = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        x_hat, rate, bits = m(xb, xc, xa, False)
    ref = torch.from_numpy(fx["x_hat"])
    src = frame_tensor(fx["current"])
    d_psnr = abs(psnr(x_hat.cpu(), src) - psnr(ref, src))
    rel = abs(bits - float(fx["bits"])) / float(fx["bits"])
    print(f"fp16 path: dPSNR={d_psnr:.4f} dB, bits rel={rel:.4f}, PSNR(fp16 vs fp32 output)={psnr(x_hat.cpu(), ref):.2f} dB")
    assert d_psnr < PSNR_TOL_DB and rel < BITS_TOL


def test_fp16_layers_are_actually_half(dev):
    from vcamd import hip
    m = _fp16_model(dev)
    packs = []
    for mod in m.modules():
        p = getattr(mod, "_packed", None)
        if isinstance(p, hip.PackedConv):
            packs.append(p)
        elif isinstance(p, (tuple, list)):
            packs += [q for q in p if isinstance(q, hip.PackedConv)]
        elif isinstance(p, dict):
            for v in p.values():
                packs += [q for q in (v if isinstance(v, (list, tuple)) else [v]) if isinstance(q, hip.PackedConv)]
    half = [p for p in packs if p.wpk16 is not None]
    assert len(half) >= 40 and len(half) < len(packs)       # heavy layers in half, tiny-channel layers stay fp32


def test_half_precision_activations_do_not_change_a_bit(dev):
    """VC_CFG_IN_F16 / VC_CFG_OUT_F16: an activation consumed only by fp16-path convolutions is stored as half;
    the consumer rounds to half while staging anyway, so the result must equal the fp32-storage run exactly --
    LHBDC (SPyNet chains, residual blocks), Flex-Rate (U-Nets) and ICIP2024 (bottleneck blocks, conv chains)."""
    from vcamd import flex, hip, icip2024, lhbdc
    from vcamd.seeding import seeded_state_dict
    fx = load_fixture("lhbdc_forward_a.npz")
    xb, xc, xa = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    builders = {
        "lhbdc": (lhbdc.Model, lambda m: m(xb, xc, xa, False)[0]),
        "flex": (lambda: flex.BidirFlowRef(n=4), lambda m: m(xb, xc, xa, n=[1], l=1.0)["x_hat"]),
        "icip2024": (icip2024.FlowGuidedB, lambda m: m(xb, xa, 0.5, 0.5, xc, 2, 1)["x_hat"]),
    }
    hip.set_conv_precision("fp16")
    hip.HALF_RESIDUAL = False          # (the half-precision identity path is NOT bit-neutral: its own test below)
    hip.HALF_DEFORM = False            # (nor are half-precision features under the deformable gather: test_icip2024_gpu.py)
    hip.FUSE_TAIL = False              # (nor the fused bottleneck tail, which needs the half activation: test_ops_gpu.py)
    try:
        for name, (build, run) in builders.items():
            m = build()
            m.load_state_dict(seeded_state_dict(m.state_dict(), seed=1234))
            m = m.to(dev).eval()
            outs, halves = [], []
            for flag in (True, False):
                hip.HALF_ACTIVATIONS = flag
                made = []
                orig = hip.T.empty

                def counting(n, h, w, c, device, dtype="f32", _made=made, _orig=orig):
                    _made.append(dtype)
                    return _orig(n, h, w, c, device, dtype)
                hip.T.empty = staticmethod(counting)
                try:
                    with torch.no_grad():
                        outs.append(run(m))
                finally:
                    hip.T.empty = staticmethod(orig)
                halves.append(made.count("f16"))
            print(f"{name}: {halves[0]} half-precision activations per frame, max|d| = {(outs[0] - outs[1]).abs().max().item():.1e}")
            assert halves[0] > 0 and halves[1] == 0
            assert torch.equal(outs[0], outs[1])
    finally:
        hip.HALF_ACTIVATIONS = True
        hip.HALF_RESIDUAL = True
        hip.HALF_DEFORM = True
        hip.FUSE_TAIL = True
        hip.set_conv_precision("fp32")


def test_half_precision_identity_path_of_bottleneck_chains(dev):
    """VC_CFG_RES_F16 (hip.HALF_RESIDUAL): ICIP2024's chains of bottleneck blocks (elic.py:69-83, out = x + f(x)) keep x as half
    between the blocks on the fp16 path.  Not bit-neutral -- one more rounding of the identity per block -- so: it must be
    IN USE (half residuals reach the streaming kernel), and what it changes must stay inside the fp16 mode's own distance
    from the fp32 path (measured on this frame, seeded weights: max |d| of the reconstruction 0.47 against 0.45 for fp16 vs
    fp32 -- both are single flipped symbols of the quantiser, not drift; estimated size 827 046 vs 827 094 vs 827 137 bits)."""
    from vcamd import hip, icip2024
    from vcamd.seeding import seeded_state_dict
    fx = load_fixture("lhbdc_forward_a.npz")
    xb, xc, xa = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    outs = {}
    seen = []
    orig = hip.PackedConv.__call__

    def spy(self, x, *a, **k):
        if k.get("res") is not None:
            seen.append(k["res"].dtype)
        return orig(self, x, *a, **k)
    try:
        for mode, prec, hr in (("fp32", "fp32", False), ("fp16 fp32-identity", "fp16", False), ("fp16 half-identity", "fp16", True)):
            hip.set_conv_precision(prec)
            hip.HALF_RESIDUAL = hr
            m = icip2024.FlowGuidedB()
            m.load_state_dict(seeded_state_dict(m.state_dict(), seed=1234))
            m = m.to(dev).eval()
            del seen[:]
            hip.PackedConv.__call__ = spy
            try:
                with torch.no_grad():
                    r = m(xb, xa, 0.5, 0.5, xc, 2, 1)
            finally:
                hip.PackedConv.__call__ = orig
            outs[mode] = (r["x_hat"].float().cpu(), float(r["size"]), seen.count("f16"))
    finally:
        hip.HALF_RESIDUAL = True
        hip.set_conv_precision("fp32")
    x32, b32, _ = outs["fp32"]
    xf, bf, nf = outs["fp16 fp32-identity"]
    xh, bh, nh = outs["fp16 half-identity"]
    d_mode = (xf - x32).abs().max().item()
    d_half = (xh - xf).abs().max().item()
    print(f"half residuals per frame {nh} (fp32-identity run: {nf}); max|fp16 - fp32| = {d_mode:.2e}, max|half identity - fp32 identity| = {d_half:.2e}, "
          f"bits {b32:.1f} / {bf:.1f} / {bh:.1f}")
    assert nf == 0 and nh >= 20
    assert d_half <= max(3.4e-2, 2.0 * d_mode)
    if b32:
        assert abs(bh - bf) / b32 < 5e-3


def test_half_activation_never_reaches_other_kernels(dev):
    from vcamd import hip
    t = hip.T.empty(1, 8, 8, 16, dev, "f16")
    with pytest.raises(hip.VcError):
        hip.axpby(t, None)
    with pytest.raises(hip.VcError):
        hip.nhwc_to_nchw(t)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: "ICIP2024 config 2160p (4K) inference, fp16 MFMA conv path"
# ---------------------------------------------------------------------------------------------------------------------
def _icip_fp16(dev):
    from vcamd import hip, icip2024
    from vcamd.seeding import seeded_state_dict
    hip.set_conv_precision("fp16")
    m = icip2024.FlowGuidedB()
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=1234))
    return m.to(dev).eval()          # (layers pack lazily: the caller keeps the fp16 mode set during the first forward)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_icip2024_forward_fp16_against_reference_fixture(dev, tag):
    """FlowGuidedB.forward on the fp16 path against the fp32 REFERENCE outputs (icip2024_forward_a.npz): operands are
    rounded to half, so the bar is the fp16 tolerance of this file, not the 1e-3 dB of the exact path."""
    from vcamd import hip
    fx = load_fixture("icip2024_forward_a.npz")
    x1, xc, x2 = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    s1, s2, lvl, drr = (float(v) for v in fx[f"cfg_{tag}"])
    try:
        m = _icip_fp16(dev)
        with torch.no_grad():
            out = m(x1, x2, s1, s2, xc, lvl, int(drr))
    finally:
        hip.set_conv_precision("fp32")
    ref = torch.from_numpy(fx[f"x_hat_{tag}"])
    src = frame_tensor(fx["current"])
    d_psnr = abs(psnr(out["x_hat"].cpu(), src) - psnr(ref, src))
    rel = abs(out["size"].item() - float(fx[f"size_{tag}"])) / float(fx[f"size_{tag}"])
    print(f"icip2024 fp16 {tag}: dPSNR={d_psnr:.4f} dB, size rel={rel:.4f}, PSNR(fp16 vs fp32 output)={psnr(out['x_hat'].cpu(), ref):.2f} dB")
    assert d_psnr < PSNR_TOL_DB and rel < BITS_TOL


def _frames_4k(dev, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 2176 + 16, 3840 + 24, generator=g), 9, 1)
    return [base[..., 2 * t:2 * t + 2176, 3 * t:3 * t + 3840].contiguous().to(dev) for t in range(3)]


# (configs[4] at its own size, 2176x3840, is compared with the fp32 oracle in tests/test_fullsize_gpu.py::
#  test_icip2024_fp16_2160p_against_fp32_oracle -- that replaced the property-only run that lived here)


def test_lhbdc_2160p_properties(dev):
    """LHBDC at 2176x3840 on the exact fp32 path: deterministic, finite, rate = size / pixels / 2 (the m.py:96 halving),
    and the fp16 path of the same frame stays within this file's tolerance of it."""
    from vcamd import hip, lhbdc
    from vcamd.seeding import seeded_state_dict
    xb, xc, xa = _frames_4k(dev, 43)
    m = lhbdc.Model()
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=1234))
    m = m.to(dev).eval()
    with torch.no_grad():
        x1, rate1, bits1 = m(xb, xc, xa, False)
        x2, rate2, bits2 = m(xb, xc, xa, False)
    assert torch.equal(x1, x2) and bits1 == bits2
    assert torch.isfinite(x1).all() and bits1 > 0
    assert abs(float(rate1) - bits1 / (2176 * 3840) / 2.0) < 1e-5 * float(rate1)
    hip.set_conv_precision("fp16")
    try:
        h = lhbdc.Model()
        h.load_state_dict(seeded_state_dict(h.state_dict(), seed=1234))
        h = h.to(dev).eval()
        with torch.no_grad():
            x16, _, bits16 = h(xb, xc, xa, False)
    finally:
        hip.set_conv_precision("fp32")
    d_psnr = abs(psnr(x16.cpu(), xc.cpu()) - psnr(x1.cpu(), xc.cpu()))
    print(f"LHBDC 2160p fp16 vs fp32: dPSNR={d_psnr:.4f} dB, bits rel={abs(bits16 - bits1) / bits1:.4f}")
    assert d_psnr < PSNR_TOL_DB and abs(bits16 - bits1) / bits1 < BITS_TOL
