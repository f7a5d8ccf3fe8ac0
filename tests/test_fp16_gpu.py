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
        hip.set_conv_precision("fp32")


def test_half_activation_never_reaches_other_kernels(dev):
    from vcamd import hip
    t = hip.T.empty(1, 8, 8, 16, dev, "f16")
    with pytest.raises(hip.VcError):
        hip.axpby(t, None)
    with pytest.raises(hip.VcError):
        hip.nhwc_to_nchw(t)
