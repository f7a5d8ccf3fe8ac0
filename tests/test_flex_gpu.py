"""-m gpu: the Flex-Rate B-frame path (BidirFlowRef surface over the HIP kernels) against the golden
fixtures recorded from the reference and against the CPU oracle; gain units, interpolated rate point,
un-gained-y / clamp quirks included."""
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
    from oracle import flex as of
    from vcamd import flex
    from vcamd.seeding import seeded_state_dict
    prod = flex.BidirFlowRef(n=4)
    sd = seeded_state_dict(prod.state_dict(), seed=1234)
    prod.load_state_dict(sd)
    ora = of.FlexModel(n=4).eval()
    ora.load_state_dict(sd)
    return ora, prod.to(dev).eval()


@pytest.mark.parametrize("tag,n,l", [("n0", 0, 1.0), ("n2", 2, 1.0), ("n1l033", 1, 0.33)])
def test_forward_matches_reference_fixture(dev, models, tag, n, l):
    _, prod = models
    fx = load_fixture("flex_forward_a.npz")
    xb, xc, xa = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        out = prod(xb, xc, xa, n=[n], l=l, train=False)
    ref = torch.from_numpy(fx[f"x_hat_{tag}"])
    src = frame_tensor(fx["current"])
    d_psnr = abs(psnr(out["x_hat"].cpu(), src) - psnr(ref, src))
    rel = abs(out["size"].item() - float(fx[f"size_{tag}"][0])) / float(fx[f"size_{tag}"][0])
    print(f"flex {tag}: max|d|={(out['x_hat'].cpu() - ref).abs().max():.3e} dPSNR={d_psnr:.2e} size rel={rel:.2e}")
    assert d_psnr < PSNR_TOL_DB and rel < 2e-3
    assert abs(out["rate"].item() - float(fx[f"rate_{tag}"][0])) / float(fx[f"rate_{tag}"][0]) < 2e-3
    assert out["x_hat"].shape == ref.shape and out["size"].shape == (1,)


def test_unet_matches_oracle(dev, models):
    ora, prod = models
    x = torch.rand(1, 6, 64, 96, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = ora.flow_predictor(x)
    out = prod.flow_predictor(x.to(dev))
    assert ((out.cpu() - ref).abs() / (1 + ref.abs())).max().item() < 5e-5


def test_process_and_backwarp_surface(dev, models):
    ora, prod = models
    g = torch.Generator().manual_seed(3)
    x0, x1 = torch.rand(1, 3, 64, 64, generator=g), torch.rand(1, 3, 64, 64, generator=g)
    with torch.no_grad():
        r0, r1, rc = ora.process(x0, x1)
    p0, p1, pc = prod.process(x0.to(dev), x1.to(dev))
    assert pc.shape == rc.shape == (1, 16, 64, 64)
    for a, b in ((p0, r0), (p1, r1), (pc, rc)):
        assert (a.cpu() - b).abs().max().item() < 2e-4


def test_codec_roundtrip_through_container(dev, models):
    from vcamd import flex
    _, prod = models
    for c in (prod.flow_compressor, prod.residual_compressor):
        c.update(force=True)
    fx = load_fixture("flex_codec_a.npz")
    n, l = int(fx["n"]), float(fx["l"])
    xb, xc, xa = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        mv_bits, res_bits = flex.encode_B(prod, xb, xc, xa, n=n, l=l)
        blob = flex.write_container(None, l, mv_bits, res_bits)
        _, s_mv, s_res, sh_mv, sh_res = flex.read_container(blob)
        dec = flex.decode_B(prod, xb, xa, s_mv, s_res, sh_mv, sh_res, n, l)
    assert tuple(sh_mv) == tuple(fx["flow_shape"]) and tuple(sh_res) == tuple(fx["res_shape"])
    # (byte-identity of the strings and the integers behind them: test_bitstream_gpu.py)
    for k, s in (("flow_y", mv_bits["strings"][0][0]), ("flow_z", mv_bits["strings"][1][0]),
                 ("res_y", res_bits["strings"][0][0]), ("res_z", res_bits["strings"][1][0])):
        assert abs(len(s) - fx[k].size) <= max(8, 0.01 * fx[k].size), k
    ref = torch.from_numpy(fx["decoded"])
    src = frame_tensor(fx["current"])
    d_psnr = abs(psnr(dec.cpu(), src) - psnr(ref, src))
    assert d_psnr < 5e-3
    assert dec.min().item() >= -1.0  # residual path is clamped to [0,1] then added to the prediction


def test_gop16_graph_replay_equals_eager(dev, models):
    from vcamd import gop as vgop
    _, prod = models
    g = torch.Generator().manual_seed(11)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 136, 216, generator=g), 9, 1)   # 128x208
    frames = [base[..., :128, i:i + 192].contiguous().to(dev) for i in range(17)]
    quality = vgop.FLEX_QUALITIES[3]
    with torch.no_grad():
        rec_s = []
        dec_s = vgop.code_gop_flex(prod, frames, frames[0], frames[16], 120, 180, quality, rec_s, batch_levels=False)
        dec_s = {k: v.clone() for k, v in dec_s.items()}
        rec_e = []
        dec_e = vgop.code_gop_flex(prod, frames, frames[0], frames[16], 120, 180, quality, rec_e)
        dec_e = {k: v.clone() for k, v in dec_e.items()}
        for k in range(1, 16):             # level-batched passes == frame-by-frame passes, bit for bit
            assert torch.equal(dec_s[k], dec_e[k]), k
        for a, b in zip(rec_s, rec_e):
            assert a[:3] == b[:3] and float(a[3]) == float(b[3]) and float(a[4]) == float(b[4])
        runner = vgop.GopGraph(prod, 120, 180, kind="flex", quality=quality)
        rec_g = []
        runner.code(frames, records=rec_g)
        dec_g = runner.code(frames)
    assert len(rec_e) == len(rec_g) == 15
    for k in range(1, 16):
        assert torch.equal(dec_e[k], dec_g[k]), k
    for a, b in zip(rec_e, rec_g):
        assert a[:3] == b[:3] and float(a[3]) == float(b[3]) and float(a[4]) == float(b[4])


def test_gops_batched_together_equal_gops_coded_alone(dev, models):
    """code_gops_flex / GopGraph(kind="flex", gops=2): two independent GOP-16s with batched level passes == each alone."""
    from vcamd import gop as vgop
    _, prod = models
    g = torch.Generator().manual_seed(23)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 136, 240, generator=g), 9, 1)
    gops = [[base[..., :128, i + off:i + off + 192].contiguous().to(dev) for i in range(17)] for off in (0, 21)]
    quality = vgop.FLEX_QUALITIES[5]
    with torch.no_grad():
        alone, recs_alone = [], []
        for k, gp in enumerate(gops):
            d = vgop.code_gop_flex(prod, gp, gp[0], gp[16], 120, 180, quality, recs_alone, gop_index=3 + k)
            alone.append({o: v.clone() for o, v in d.items()})
        recs_b = []
        both = vgop.code_gops_flex(prod, gops, [(gp[0], gp[16]) for gp in gops], 120, 180, quality, recs_b, first_gop_index=3)
        for k in range(2):
            for o in range(1, 16):
                assert torch.equal(alone[k][o], both[k][o]), (k, o)
        runner = vgop.GopGraph(prod, 120, 180, kind="flex", quality=quality, gops=2)
        runner.code(gops[0] + gops[1])
        recs_g = []
        runner.code(gops[0] + gops[1], gop_index=3, records=recs_g)
    assert [r[:3] for r in recs_alone] == [r[:3] for r in recs_b] == [r[:3] for r in recs_g]
    assert all(float(a[4]) == float(b[4]) == float(c[4]) for a, b, c in zip(recs_alone, recs_b, recs_g))


def test_sequence_loop_against_the_reference_test_function(dev):
    """vcamd.gop.code_sequence_flex against the rows of the reference's own ``test()`` (fixture flex_test_loop.json: seven
    clips x all eight operating points of testing.py:86-89; three clips per point are run here).  Calibrated checkpoint
    (trained-like statistics), every hierarchy level held at 1e-3 dB / 1e-3 in size, like the LHBDC twin of this test."""
    import json
    import os
    from helpers import GOLDEN
    from oracle import lhbdc as ol
    from vcamd import flex, gop as vgop, iframe
    from vcamd.seeding import seeded_state_dict
    fx = json.load(open(os.path.join(GOLDEN, "flex_test_loop.json")))
    from helpers import fixture_intra_state_dict, fixture_state_dict
    b_model = flex.BidirFlowRef(n=4)
    b_model.load_state_dict(fixture_state_dict(fx, b_model.state_dict()))
    b_model = b_model.to(dev).eval()
    h, w = fx["frame_hw"]
    assert fx.get("checkpoint") == "calibrated"
    tol = {"I": (1e-3, 1e-4), 0: (1e-3, 1e-3), 1: (1e-3, 1e-3), 2: (1e-3, 1e-3), 3: (1e-3, 1e-3)}
    worst = {k: [0.0, 0.0] for k in tol}
    i_models = {}
    with torch.no_grad():
        for i_qual, table in fx["qualities"]:
            quality = (i_qual, {int(k): tuple(v) for k, v in table.items()})
            if i_qual not in i_models:
                m = iframe.mbt2018_mean(i_qual, "mse", pretrained=False)
                m.load_state_dict(fixture_intra_state_dict(fx, m.state_dict(), fx["seed"] + i_qual))
                i_models[i_qual] = m.to(dev).eval()
            lvl, itv = quality[1][3]
            for k, name in enumerate(fx["folders"][:3]):            # three clips per operating point keep the test short
                frames = [ol.pad64(torch.from_numpy(f.astype("float32").transpose(2, 0, 1))[None] / 255.0).to(dev)
                          for f in ol.harness_frames(fx["seed"], k, fx["frames_per_video"], h, w)]
                recs = vgop.code_sequence_flex(b_model, i_models, lambda i: frames[i], len(frames), h, w, quality, video=k, test_size=1)
                ref = [r for r in fx["rows"] if r[0] == name and r[1] == lvl and r[2] == itv]
                assert len(recs) == len(ref) == 17
                assert [int(r[6]) for r in recs] == [1 if r[3] == "I" else 0 for r in ref]
                assert [int(r[1]) % 16 for r in recs[2:]] == [int(r[4]) for r in ref[2:]]
                for mine, theirs in zip(recs, ref):
                    key = "I" if theirs[3] == "I" else vgop.HIER_LEVELS_16[int(theirs[4])]
                    worst[key][0] = max(worst[key][0], abs(float(mine[3]) - theirs[5]))
                    worst[key][1] = max(worst[key][1], abs(float(mine[4]) - theirs[6]) / theirs[6])
    print("flex test() loop, worst (dPSNR dB, size rel) per level:", {k: (f"{v[0]:.1e}", f"{v[1]:.1e}") for k, v in worst.items()})
    for key, (tp, ts) in tol.items():
        assert worst[key][0] < tp and worst[key][1] < ts, (key, worst[key])
