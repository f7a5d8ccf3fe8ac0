"""-m gpu: the ICIP2024 FlowGuidedB path (deformable compensation, checkerboard + channel-context entropy model)
over the HIP kernels, against the fixtures recorded from the reference and against the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import frame_tensor, load_fixture, psnr

pytestmark = pytest.mark.gpu
PSNR_TOL_DB = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def models(dev):
    from oracle import icip2024 as oi
    from vcamd import icip2024
    from vcamd.seeding import seeded_state_dict
    prod = icip2024.FlowGuidedB()
    sd = seeded_state_dict(prod.state_dict(), seed=1234)
    prod.load_state_dict(sd)
    ora = oi.FlowGuidedB().eval()
    ora.load_state_dict(sd)
    return ora, prod.to(dev).eval()


def _nhwc(x, dev):
    from vcamd import hip
    return hip.nchw_to_nhwc(x.to(dev))


@pytest.mark.parametrize("c,groups,h,w", [(64, 16, 24, 40), (96, 16, 17, 23), (128, 16, 16, 16), (32, 8, 9, 11)])
def test_deform_conv2d_matches_oracle(dev, c, groups, h, w):
    """Generic operator: random offsets up to +-6 px (many samples leave the image) and random masks."""
    from oracle import deform as od
    from vcamd import hip
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(2, 2 * c, h, w, generator=g)
    off = (torch.rand(2, groups * 18, h, w, generator=g) - 0.5) * 12
    msk = torch.rand(2, groups * 9, h, w, generator=g)
    wt = torch.randn(c, 2 * c // groups, 3, 3, generator=g) * 0.2
    b = torch.randn(c, generator=g)
    ref = od.deform_conv2d(x, off, wt, b, padding=(1, 1), mask=msk)
    pk = hip.PackedDeform(wt, b, groups, dev)
    out = hip.nhwc_to_nchw(pk.conv(_nhwc(x, dev), _nhwc(off, dev), _nhwc(msk, dev))).cpu()
    assert ((out - ref).abs() / (1 + ref.abs())).max().item() < 2e-5
    ref_nomask = od.deform_conv2d(x, off, wt, b, padding=(1, 1), mask=None)
    out = hip.nhwc_to_nchw(pk.conv(_nhwc(x, dev), _nhwc(off, dev))).cpu()
    assert ((out - ref_nomask).abs() / (1 + ref_nomask.abs())).max().item() < 2e-5


@pytest.mark.parametrize("c,mag", [(64, 40), (96, 20), (128, 10)])
def test_offset_diversity_matches_oracle(dev, c, mag):
    """Fused OffsetDiversity.forward: tanh/sigmoid preparation + flipped flow + grouped deformable conv."""
    from oracle import icip2024 as oi
    from vcamd import icip2024
    g = torch.Generator().manual_seed(c)
    h, w = 20, 28
    ora = oi.OffsetDiversity(c, mag)
    prod = icip2024.OffsetDiversity(c, mag)
    with torch.no_grad():
        ora.fusion.weight.copy_(torch.randn(ora.fusion.weight.shape, generator=g) * 0.2)
        ora.fusion.bias.copy_(torch.randn(c, generator=g))
    prod.load_state_dict(ora.state_dict())
    prod = prod.to(dev)
    x1, x2 = torch.randn(1, c, h, w, generator=g), torch.randn(1, c, h, w, generator=g)
    o1, o2 = torch.randn(1, 216, h, w, generator=g), torch.randn(1, 216, h, w, generator=g)
    f1, f2 = torch.randn(1, 2, h, w, generator=g) * 3, torch.randn(1, 2, h, w, generator=g) * 3
    f1[0, :, 0, 0] = float("inf")                       # a broken flow vector must not fault
    with torch.no_grad():
        ref = ora(x1, o1, f1, x2, o2, f2)
    from vcamd import hip
    out = hip.nhwc_to_nchw(prod.run(*[_nhwc(t, dev) for t in (x1, o1, f1, x2, o2, f2)])).cpu()
    ok = torch.isfinite(ref)
    assert ((out - ref).abs() / (1 + ref.abs()))[ok].max().item() < 5e-5


@pytest.mark.parametrize("planar", [True, False])
@pytest.mark.parametrize("c,mag", [(64, 40), (96, 20), (128, 10)])
def test_offset_diversity_half_precision_features(dev, c, mag, planar):
    """fp16 path (hip.HALF_DEFORM, vc_offset_diversity_hxp / _hx): the fusion gathers from half-precision copies of its feature maps
    -- the fp32 kernel's result on features that were rounded to half beforehand (offsets, modulation, bilinear weights and
    accumulation stay fp32), up to the compiler's choice of fused multiply-adds in the two instances: 1e-6 relative, 500x
    below one half-precision rounding; 8, 12 and 16 channels per group (12: a 16-byte and an 8-byte gather per corner).
    Both layouts of the half copies: group-planar (the default, hip.HALF_DEFORM_PLANAR) and pixel-interleaved -- the same
    values through the same arithmetic, so the two results are equal bit for bit."""
    from vcamd import hip, icip2024
    g = torch.Generator().manual_seed(c + 1)
    h, w, n = 27, 44, 2
    prod = icip2024.OffsetDiversity(c, mag)
    with torch.no_grad():
        prod.fusion.weight.copy_(torch.randn(prod.fusion.weight.shape, generator=g) * 0.2)
        prod.fusion.bias.copy_(torch.randn(c, generator=g))
    prod = prod.to(dev)
    x1, x2 = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
    o1, o2 = torch.randn(n, 216, h, w, generator=g), torch.randn(n, 216, h, w, generator=g)
    f1, f2 = torch.randn(n, 2, h, w, generator=g) * 3, torch.randn(n, 2, h, w, generator=g) * 3
    rest = [_nhwc(t, dev) for t in (o1, f1, o2, f2)]
    calls = []
    orig, orig_p, keep = hip.to_half, hip.to_half_planar, hip.HALF_DEFORM_PLANAR
    hip.to_half = lambda t: (calls.append(("interleaved", t.c)), orig(t))[1]
    hip.to_half_planar = lambda t, cg: (calls.append(("planar", t.c)), orig_p(t, cg))[1]
    try:
        ref = hip.nhwc_to_nchw(prod.run(_nhwc(x1.half().float(), dev), rest[0], rest[1], _nhwc(x2.half().float(), dev), rest[2], rest[3]))
        assert not calls                                        # fp32 precision: fp32 features
        hip.set_conv_precision("fp16")
        hip.HALF_DEFORM_PLANAR = planar
        out = hip.nhwc_to_nchw(prod.run(_nhwc(x1, dev), rest[0], rest[1], _nhwc(x2, dev), rest[2], rest[3]))
        assert calls == [("planar" if planar else "interleaved", c)] * 2
        hip.HALF_DEFORM_PLANAR = not planar
        other = hip.nhwc_to_nchw(prod.run(_nhwc(x1, dev), rest[0], rest[1], _nhwc(x2, dev), rest[2], rest[3]))
    finally:
        hip.to_half, hip.to_half_planar, hip.HALF_DEFORM_PLANAR = orig, orig_p, keep
        hip.set_conv_precision("fp32")
    d = ((out - ref).abs() / (1 + ref.abs())).max().item()
    print(f"c={c} planar={planar}: max relative difference {d:.2e}")
    assert d < 1e-6
    assert torch.equal(out, other)


@pytest.mark.parametrize("c,cg,h,w", [(64, 8, 9, 13), (96, 12, 5, 7), (128, 16, 4, 6), (8, 4, 3, 5)])
def test_to_half_planar_layout(dev, c, cg, h, w):
    """vc_to_half_planar: [n][c / cg][h][w][cg] halves, round to nearest even -- against torch's own cast and permutation"""
    from vcamd import hip
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, h, w, generator=g) * 3
    x[0, 0, 0, 0] = 1.0 + 2.0 ** -11                            # a tie: rounds to even (1.0)
    t = _nhwc(x, dev)
    out = hip.to_half_planar(t, cg)
    got = out.buf.flatten()                                     # (the window's flat half buffer)
    assert got.dtype == torch.float16 and got.numel() == 2 * c * h * w
    want = x.half().view(2, c // cg, cg, h, w).permute(0, 1, 3, 4, 2).contiguous().flatten()
    assert torch.equal(got.cpu(), want)


def test_quantize_mask(dev):
    from vcamd import hip
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 10, 7, 9, generator=g) * 3
    x[0, 0, 0, :4] = torch.tensor([0.5, 1.5, 2.5, -0.5])     # half-to-even
    gain = torch.rand(10, generator=g) + 0.5
    t = _nhwc(x, dev)
    yy, xx = torch.meshgrid(torch.arange(7), torch.arange(9), indexing="ij")
    odd = ((yy + xx) % 2 == 1).float()
    assert torch.equal(hip.nhwc_to_nchw(hip.quantize_mask(t)).cpu(), torch.round(x))
    assert torch.equal(hip.nhwc_to_nchw(hip.quantize_mask(t, keep_parity=1)).cpu(), torch.round(x) * odd)
    assert torch.equal(hip.nhwc_to_nchw(hip.quantize_mask(t, keep_parity=0, do_round=False)).cpu(), x * (1 - odd))
    out = hip.nhwc_to_nchw(hip.quantize_mask(t, gain=gain.to(dev))).cpu()
    assert torch.allclose(out, torch.round(x) * gain.view(1, -1, 1, 1), rtol=1e-6, atol=0)


@pytest.mark.parametrize("dr", [1, 2, 4, 8])
def test_estimate_flow_matches_oracle(dev, models, dr):
    ora, prod = models
    fx = load_fixture("icip2024_forward_a.npz")
    x1, x2 = (frame_tensor(fx[k]) for k in ("ref_1", "ref_2"))
    with torch.no_grad():
        ref = ora.estimate_flow(x1, x2, dr)
    out = prod.estimate_flow(x1.to(dev), x2.to(dev), dr).cpu()
    assert out.shape == ref.shape
    assert ((out - ref).abs() / (1 + ref.abs())).max().item() < 2e-4


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_forward_matches_reference_fixture(dev, models, tag):
    _, prod = models
    fx = load_fixture("icip2024_forward_a.npz")
    x1, xc, x2 = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    s1, s2, lvl, drr = (float(v) for v in fx[f"cfg_{tag}"])
    with torch.no_grad():
        out = prod(x1, x2, s1, s2, xc, lvl, int(drr))
    ref = torch.from_numpy(fx[f"x_hat_{tag}"])
    src = frame_tensor(fx["current"])
    d_psnr = abs(psnr(out["x_hat"].cpu(), src) - psnr(ref, src))
    rel = abs(out["size"].item() - float(fx[f"size_{tag}"])) / float(fx[f"size_{tag}"])
    print(f"icip2024 {tag}: max|d|={(out['x_hat'].cpu() - ref).abs().max():.3e} dPSNR={d_psnr:.2e} size rel={rel:.2e}")
    assert d_psnr < PSNR_TOL_DB and rel < 2e-3
    assert abs(out["rate"].item() - float(fx[f"rate_{tag}"])) / float(fx[f"rate_{tag}"]) < 2e-3


def test_forward_matches_oracle_on_other_frames(dev, models):
    """A different crop (256x192, asymmetric scales, interpolated quality level) against the oracle run live."""
    ora, prod = models
    fx = load_fixture("lhbdc_forward_b.npz")
    x1, xc, x2 = (frame_tensor(fx[k]) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        ref = ora(x1, x2, 0.75, 0.25, xc, 1.5, 2)
        out = prod(x1.to(dev), x2.to(dev), 0.75, 0.25, xc.to(dev), 1.5, 2)
    d_psnr = abs(psnr(out["x_hat"].cpu(), xc) - psnr(ref["x_hat"], xc))
    rel = abs(out["size"].item() - ref["size"].item()) / ref["size"].item()
    print(f"icip2024 live: dPSNR={d_psnr:.2e} size rel={rel:.2e}")
    assert d_psnr < PSNR_TOL_DB and rel < 2e-3


def test_down_ratio_search_matches_fixture(dev, models):
    from vcamd import icip2024
    _, prod = models
    fx = load_fixture("icip2024_forward_a.npz")
    x1, xc, x2 = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    for dr in (1, 2, 4, 8, 16):
        pred = icip2024.prediction_flowonly(prod, xc, x1, x2, 0.5, 0.5, dr).cpu()
        assert (pred - torch.from_numpy(fx[f"pred_dr{dr}"])).abs().max().item() < 2e-4
    best, best_psnr = icip2024.get_best_down_ratio_prediction(prod, x1, x2, 0.5, 0.5, xc)
    assert best == int(fx["best_down_ratio"]) and abs(best_psnr.item() - float(fx["best_pred_psnr"])) < 1e-3


def test_rejects_cpu_tensors_and_unpadded_frames(dev, models):
    from vcamd import hip
    _, prod = models
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(hip.VcError):
        prod(x, x, 0.5, 0.5, x, 1, 1)
    y = torch.zeros(1, 3, 72, 64, device=dev)
    with pytest.raises(hip.VcError):
        prod(y, y, 0.5, 0.5, y, 1, 1)


def test_device_search_matches_host_search(dev, models):
    """search_flow_t: same choice as the reference's host loop (fixture) and the chosen flow is estimate_flow(best)."""
    from vcamd import hip
    _, prod = models
    fx = load_fixture("icip2024_forward_a.npz")
    x1, xc, x2 = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    t1, tc, t2 = (hip.nchw_to_nhwc(x) for x in (x1, xc, x2))
    ratios = (1, 2, 4, 8, 16)
    flow, choice, sse = prod.search_flow_t(tc, t1, t2, 0.5, 0.5, ratios)
    best = ratios[int(choice.item())]
    assert best == int(fx["best_down_ratio"])
    psnr_dev = 10 * np.log10(1.0 / (sse.cpu().numpy() / xc.numel()))
    assert abs(psnr_dev.max() - float(fx["best_pred_psnr"])) < 1e-3
    ref = prod.estimate_flow(x1, x2, best)
    assert torch.equal(hip.nhwc_to_nchw(flow), ref)
    # a forward fed with the searched flow equals a forward that recomputes it
    with torch.no_grad():
        a = prod(x1, x2, 0.5, 0.5, xc, 1, best)
        from vcamd.layers import BitCounter
        bits = BitCounter(dev, max_rows=16)
        b = hip.nhwc_to_nchw(prod.forward_device(t1, t2, 0.5, 0.5, tc, 1, None, bits, flow=flow))
    assert torch.equal(a["x_hat"], b)


def _synthetic_gop16(dev, h=128, w=192):
    g = torch.Generator().manual_seed(7)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, h + 64, w + 64, generator=g), 5, 1, 2)
    frames = []
    for t in range(17):
        dx, dy = int(round(1.5 * t)), int(round(0.75 * t))
        f = base[:, :, dy:dy + h, dx:dx + w] + 0.01 * torch.randn(1, 3, h, w, generator=g)
        frames.append(f.clamp(0, 1).contiguous())
    return frames


def test_gop16_coder_modes_agree_and_follow_oracle(dev, models):
    """code_gop_icip2024: device search == host search, cached reference features == recomputed ones (bit-equal),
    and the first hierarchy levels follow the oracle's GOP loop (src/test.py:37-101 restated with oracle calls)."""
    from oracle import icip2024 as oi
    from vcamd import gop as vgop
    ora, prod = models
    frames = _synthetic_gop16(dev)
    dframes = [f.to(dev) for f in frames]
    h, w = frames[0].shape[2:]
    with torch.no_grad():
        recs_d, recs_h, recs_n = [], [], []
        dec_d, pick_d = vgop.code_gop_icip2024(prod, dframes, dframes[0], dframes[16], h, w, 2, recs_d, search="device")
        dec_h, pick_h = vgop.code_gop_icip2024(prod, dframes, dframes[0], dframes[16], h, w, 2, recs_h, search="host")
        dec_n, _ = vgop.code_gop_icip2024(prod, dframes, dframes[0], dframes[16], h, w, 2, recs_n, search="host",
                                          cache_features=False)
    ratios = (1, 2, 4, 8, 16)
    for o in vgop.ICIP_ORDER_16[1:]:
        assert ratios[int(pick_d[o].item())] == pick_h[o]
        assert torch.equal(dec_d[o], dec_h[o]) and torch.equal(dec_h[o], dec_n[o])
    assert [r[1] for r in recs_d] == vgop.ICIP_ORDER_16[1:] and len(recs_d) == 15
    # oracle loop for the first three coded B-frames (8, 4, 12): errors compound down the hierarchy, so compare early
    buf, buf_order = [frames[0], frames[16]], [0, 16]
    with torch.no_grad():
        for i, order in enumerate(vgop.ICIP_ORDER_16[1:4]):
            lo, hi = oi.select_references(order, buf_order)
            s1, s2 = oi.get_scales(order, buf_order[lo], buf_order[hi])
            best, _ = oi.get_best_down_ratio_prediction(ora, buf[lo], buf[hi], s1, s2, frames[order])
            out = ora(buf[lo], buf[hi], s1, s2, frames[order], 2, best)
            assert best == pick_h[order]
            d_psnr = abs(psnr(dec_h[order].cpu(), frames[order]) - psnr(out["x_hat"], frames[order]))
            rel = abs(recs_h[i][4].item() - out["size"].item()) / out["size"].item()
            print(f"gop16 frame {order}: down_ratio={best} dPSNR={d_psnr:.2e} size rel={rel:.2e}")
            assert d_psnr < 5e-3 and rel < 5e-3
            buf.append(torch.clamp(out["x_hat"], 0, 1))
            buf_order.append(order)


def test_gop16_graph_replay_equals_eager(dev, models):
    from vcamd import gop as vgop
    _, prod = models
    dframes = [f.to(dev) for f in _synthetic_gop16(dev)]
    h, w = dframes[0].shape[2:]
    with torch.no_grad():
        recs_e, recs_g = [], []
        dec_e, _ = vgop.code_gop_icip2024(prod, dframes, dframes[0], dframes[16], h, w, 3, recs_e, search="device")
        runner = vgop.GopGraph(prod, h, w, kind="icip2024", quality=3)
        runner.code(dframes, records=None)
        dec_g = runner.code(dframes, records=recs_g)
    for o in vgop.ICIP_ORDER_16[1:]:
        assert torch.equal(dec_e[o], dec_g[o])
    assert torch.equal(torch.stack([r[4] for r in recs_e]), torch.stack([r[4] for r in recs_g]))


def test_select_flow_follows_the_reference_comparison(dev):
    """vc_select_flow: first strictly greatest POSITIVE PSNR wins (opt_helpers.py:44-49 starts from best = 0 and
    compares with '>'); ties keep the earlier candidate; when no PSNR is positive (MSE >= 1) candidate 0 is used;
    NaN never wins."""
    from vcamd import hip
    L = hip.lib()
    cands = [hip.T.empty(1, 4, 6, 4, dev) for _ in range(5)]
    for i, c in enumerate(cands):
        c.buf.fill_(float(i + 1))
    views = (hip.View * 5)(*[c.view() for c in cands])
    n_elems = 100.0

    def pick(mses):
        sse = torch.tensor([m * n_elems for m in mses], dtype=torch.float64, device=dev)
        out = hip.T.empty(1, 4, 6, 4, dev)
        choice = torch.full((1,), -1, dtype=torch.int32, device=dev)
        hip.check(L.vc_select_flow(hip.stream(), sse.data_ptr(), 5, n_elems, views, out.view(), choice.data_ptr()), "vc_select_flow")
        c = int(choice.item())
        assert torch.equal(out.buf, cands[c].buf)
        return c
    assert pick([0.01, 0.002, 0.002, 0.5, 0.003]) == 1          # tie between 1 and 2 -> the earlier one
    assert pick([0.5, 0.4, 0.3, 0.2, 0.1]) == 4
    assert pick([2.0, 3.0, 1.5, 1.0, 7.0]) == 0                 # PSNR <= 0 everywhere: nothing beats best = 0
    assert pick([float("nan"), 0.2, float("nan"), 0.1, 0.3]) == 3
    assert pick([0.0, 0.2, 0.1, 0.1, 0.3]) == 0                 # MSE 0 -> PSNR +inf
    bad = (hip.View * 5)(*[c.view() for c in cands[:4]], hip.T.empty(1, 4, 6, 2, dev).view())
    assert L.vc_select_flow(hip.stream(), torch.zeros(5, dtype=torch.float64, device=dev).data_ptr(), 5, n_elems, bad,
                            cands[0].view(), None) != 0


def test_residual_before_activation_and_attention_gate(dev):
    """VC_CFG_RES_FIRST (relu(conv + res), compressai ResidualUnit) on the general and the streaming 1x1 kernel, and
    vc_attention_gate, against torch."""
    import torch.nn.functional as F
    from vcamd import hip
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 18, 37, generator=g)
    res = torch.randn(2, 96, 18, 37, generator=g)
    for k in (1, 3):
        wt = torch.randn(96, 64, k, k, generator=g) / (64 * k * k) ** 0.5
        b = torch.randn(96, generator=g) * 0.1
        ref = F.relu(F.conv2d(x, wt, b, padding=k // 2) + res)
        pc = hip.PackedConv(wt, b, device=dev)
        for cfg in ([pc.cfg, 6] if k == 1 else [pc.cfg]):
            pc.tuned = {(2, 18, 37, hip.CFG_RES_FIRST): cfg | hip.CFG_EXACT | hip.CFG_RES_FIRST}
            out = hip.nhwc_to_nchw(pc(hip.nchw_to_nhwc(x.to(dev)), act=hip.ACT_RELU, res=hip.nchw_to_nhwc(res.to(dev)), res_first=True))
            assert ((out.cpu() - ref).abs() / (1 + ref.abs())).max().item() < 2e-5
    a, bb, idt = (torch.randn(1, 24, 9, 11, generator=g) for _ in range(3))
    out = hip.nhwc_to_nchw(hip.attention_gate(*[hip.nchw_to_nhwc(t.to(dev)) for t in (a, bb, idt)])).cpu()
    assert (out - (a * torch.sigmoid(bb) + idt)).abs().max().item() < 1e-6


def test_elic_intra_codec_matches_reference_fixture_and_oracle(dev):
    from oracle import icip2024 as oi
    from vcamd import icip2024
    from vcamd.seeding import seeded_state_dict
    fx = load_fixture("icip2024_elic_a.npz")
    prod = icip2024.ELIC()
    sd = seeded_state_dict(prod.state_dict(), seed=int(fx["seed"]), conv_gain=float(fx["conv_gain"]))
    prod.load_state_dict(sd)
    prod = prod.to(dev).eval()
    x = frame_tensor(fx["current"])
    with torch.no_grad():
        dec, size = icip2024.image_compress(x.to(dev), [prod], 0)
    ref = torch.from_numpy(fx["x_hat"])
    d_psnr = abs(psnr(dec.cpu(), x) - psnr(ref, x))
    rel = abs(size.item() - float(fx["size"])) / float(fx["size"])
    print(f"ELIC: max|d|={(dec.cpu() - ref).abs().max():.3e} dPSNR={d_psnr:.2e} size rel={rel:.2e}")
    assert d_psnr < PSNR_TOL_DB and rel < 2e-3
    # a batch of two different frames against the oracle run live
    fy = load_fixture("lhbdc_forward_b.npz")
    xs = torch.cat([frame_tensor(fy["ref_1"]), frame_tensor(fy["current"])], 0)
    ora = oi.ELIC().eval()
    ora.load_state_dict(sd)
    with torch.no_grad():
        r = ora(xs)
        o = prod(xs.to(dev))
    assert abs(psnr(o["x_hat"].cpu(), xs) - psnr(r["x_hat"], xs)) < PSNR_TOL_DB
    assert abs(o["size"].item() - oi._bits(r["likelihoods"]).item()) / o["size"].item() < 2e-3


def _elic(dev, fx):
    from vcamd import icip2024
    from vcamd.seeding import seeded_state_dict
    prod = icip2024.ELIC()
    prod.load_state_dict(seeded_state_dict(prod.state_dict(), seed=int(fx["seed"]), conv_gain=float(fx["conv_gain"])))
    prod = prod.to(dev).eval()
    prod.update(force=True)
    return prod


def test_elic_forward_stage2_matches_reference_fixture(dev):
    """elic.py:247-305 against the reference's own forward_stage2 (icip2024_elic_codec_a.npz)."""
    fx = load_fixture("icip2024_elic_codec_a.npz")
    prod = _elic(dev, fx)
    x = frame_tensor(fx["current"])
    with torch.no_grad():
        out = prod.forward_stage2(x.to(dev))
    ref = torch.from_numpy(fx["stage2_x_hat"])
    d_psnr = abs(psnr(out["x_hat"].cpu(), x) - psnr(ref, x))
    rel = abs(out["size"].item() - float(fx["stage2_size"])) / float(fx["stage2_size"])
    print(f"ELIC forward_stage2: max|d|={(out['x_hat'].cpu() - ref).abs().max():.3e} dPSNR={d_psnr:.2e} size rel={rel:.2e}")
    assert d_psnr < PSNR_TOL_DB and rel < 2e-3


def test_elic_bitstream_against_reference_fixture(dev):
    """The two-pass checkerboard codec (elic.py:307-596): (1) decompress(compress(x)) rebuilds exactly the latents the
    encoder coded and the frame g_s gives for them; (2) against the strings the REFERENCE produced for the same image:
    every string of the same length class, byte-identical wherever no latent sits on a rounding boundary (counted); (3) the
    HIP decoder reads the reference's strings: the 10 decode_stream calls chain through five context networks, so it
    either lands on the reference's latents or -- after an index flip -- is reported as such."""
    fx = load_fixture("icip2024_elic_codec_a.npz")
    prod = _elic(dev, fx)
    x = frame_tensor(fx["current"])
    with torch.no_grad():
        enc = prod.compress(x.to(dev))
        dec = prod.decompress(enc["strings"], enc["shape"])
    assert tuple(enc["shape"]) == tuple(fx["shape"]) and len(enc["strings"][0]) == 5
    for a, b in zip(enc["y_hat"], dec["y_hat"]):
        assert torch.equal(a, b)                                   # the decoder recovers the encoder's latents exactly
    ref_y = torch.from_numpy(fx["y_hat"])
    mine_y = torch.cat(enc["y_hat"], 1).cpu()
    moved = int(((mine_y - ref_y).abs() > 0.5).sum())
    same = [enc["strings"][0][g][0] == fx[f"y_string_{g}"].tobytes() for g in range(5)] + [enc["strings"][1][0] == fx["z_string"].tobytes()]
    print(f"ELIC compress vs reference: latents that moved by a quantisation step: {moved} of {ref_y.numel()}; strings identical: {same}")
    assert moved <= max(2, ref_y.numel() // 1000)
    for g in range(5):
        ref_len = fx[f"y_string_{g}"].size
        assert abs(len(enc["strings"][0][g][0]) - ref_len) <= max(8, 0.02 * ref_len), g
    if moved == 0:
        assert all(same)
    ref_dec = torch.from_numpy(fx["decoded"])
    d_psnr = abs(psnr(dec["x_hat"].cpu(), x) - psnr(ref_dec, x))
    assert d_psnr < 5e-3
    # the reference's bitstream through the HIP decoder
    ref_strings = [[[fx[f"y_string_{g}"].tobytes()] for g in range(5)], [fx["z_string"].tobytes()]]
    with torch.no_grad():
        try:
            other = prod.decompress(ref_strings, tuple(fx["shape"]))
            y_err = (torch.cat(other["y_hat"], 1).cpu() - ref_y).abs().max().item()
            x_err = (other["x_hat"].cpu() - ref_dec).abs().max().item()
            print(f"HIP decode of the reference's ELIC strings: max|d y_hat|={y_err:.3e}, max|d x_hat|={x_err:.3e}")
            if y_err < 0.25:                                       # no index flip anywhere in the chain: same integers
                assert x_err < 1e-3
        except Exception as e:  # noqa: BLE001  (a flipped index desynchronises the range decoder: VC_EDATA)
            from vcamd import hip
            assert isinstance(e, hip.VcError)
            print("HIP decode of the reference's ELIC strings desynchronised:", e)


def test_full_size_properties_1080p(dev, models):
    """At BASELINE's size (1080x1920 padded to 1088x1920), size-independent properties instead of the CPU oracle:
    a batch of two frames equals the two single passes bit for bit, the device-side flow-resolution search picks
    what the host loop picks, feeding the searched flow equals recomputing it, and the rate rows add up."""
    from vcamd import hip, icip2024
    from vcamd.layers import BitCounter
    _, prod = models
    g = torch.Generator().manual_seed(11)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 1088 + 32, 1920 + 32, generator=g), 7, 1, 3)
    fr = [(base[:, :, 2 * t:2 * t + 1088, 3 * t:3 * t + 1920] + 0.01 * torch.randn(1, 3, 1088, 1920, generator=g)).clamp(0, 1).to(dev)
          for t in range(5)]
    with torch.no_grad():
        a = prod(fr[0], fr[2], 0.5, 0.5, fr[1], 2, 1)
        b = prod(fr[2], fr[4], 0.5, 0.5, fr[3], 2, 1)
        bits = BitCounter(dev, max_rows=24)
        both = prod.forward_device(hip.nchw_to_nhwc(torch.cat([fr[0], fr[2]])), hip.nchw_to_nhwc(torch.cat([fr[2], fr[4]])),
                                   0.5, 0.5, hip.nchw_to_nhwc(torch.cat([fr[1], fr[3]])), 2, 1, bits)
        both = hip.nhwc_to_nchw(both)
        rows = bits.totals().view(12, 2)
        assert torch.equal(both[0:1], a["x_hat"]) and torch.equal(both[1:2], b["x_hat"])
        assert abs(rows[:, 0].sum().item() - a["size"].item()) / a["size"].item() < 1e-6
        assert abs((a["size_offset"] + a["size_residual"]).item() - a["size"].item()) / a["size"].item() < 1e-6
        assert abs(a["rate"].item() - a["size"].item() / (1088 * 1920)) / a["rate"].item() < 1e-6
        best_host, _ = icip2024.get_best_down_ratio_prediction(prod, fr[0], fr[2], 0.5, 0.5, fr[1])
        t0, t1, t2 = (hip.nchw_to_nhwc(x) for x in (fr[0], fr[1], fr[2]))
        flow, choice, _ = prod.search_flow_t(t1, t0, t2, 0.5, 0.5)
        assert (1, 2, 4, 8, 16)[int(choice.item())] == best_host
        bits2 = BitCounter(dev, max_rows=12)
        again = hip.nhwc_to_nchw(prod.forward_device(t0, t2, 0.5, 0.5, t1, 2, None, bits2, flow=flow))
        ref = prod(fr[0], fr[2], 0.5, 0.5, fr[1], 2, best_host)
        assert torch.equal(again, ref["x_hat"])
    assert torch.isfinite(a["x_hat"]).all() and a["size"].item() > 0


def test_sequence_loop_matches_reference_fixture(dev, models):
    """code_sequence_icip2024 against the reference's own val_sequence_level output (fixture): 20 frames = I, I, a full
    hierarchical GOP-16 and an irregular tail (frames 17-19), ELIC intra frames, per-frame flow-resolution search."""
    from vcamd import gop as vgop, icip2024
    from vcamd.seeding import seeded_state_dict
    _, prod = models
    fx = load_fixture("icip2024_sequence_a.npz")
    fe = load_fixture("icip2024_elic_a.npz")
    intra = icip2024.ELIC()
    intra.load_state_dict(seeded_state_dict(intra.state_dict(), seed=int(fe["seed"]), conv_gain=float(fe["conv_gain"])))
    intra = intra.to(dev).eval()
    clip = [(torch.from_numpy(f.astype("float32"))[None] / 255.0).to(dev) for f in fx["clip_u8"]]
    with torch.no_grad():
        psnr, size = vgop.code_sequence_icip2024(prod, [intra] * 5, lambda i: clip[i], len(clip), int(fx["level"]))
        psnr_h, size_h = vgop.code_sequence_icip2024(prod, [intra] * 5, lambda i: clip[i], len(clip), int(fx["level"]),
                                                     search="host", cache_features=False)
    assert psnr == psnr_h and size == size_h                  # device search + cached features change nothing
    d_psnr = max(abs(a - b) for a, b in zip(psnr, fx["psnr"].tolist()))
    rel = max(abs(a - b) / b for a, b in zip(size, fx["size"].tolist()))
    print(f"sequence loop: max dPSNR={d_psnr:.2e} dB, max size rel={rel:.2e} over {len(clip)} frames")
    assert d_psnr < 2e-2 and rel < 5e-3                       # decoding errors compound down the hierarchy
    early = fx["order"].tolist()[:6]
    assert max(abs(psnr[o] - fx["psnr"][o]) for o in early) < 1e-3
