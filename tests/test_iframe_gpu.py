"""-m gpu: I-frame codec (mbt2018_mean architecture) and the sequence-level GOP loop
(8(f)-3) on the HIP path against the CPU oracle."""
import pytest
import torch

from helpers import psnr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def codecs(dev):
    from oracle.cai.models import mbt2018_mean as oracle_mean
    from vcamd import iframe
    from vcamd.seeding import seeded_state_dict
    ora = oracle_mean(7).eval()
    sd = seeded_state_dict(ora.state_dict(), seed=77, conv_gain=0.8)
    ora.load_state_dict(sd)
    prod = iframe.mbt2018_mean(7, "mse", pretrained=False)
    prod.load_state_dict(sd)
    return ora, prod.to(dev).eval()


def test_transposed_conv_as_subpel(dev):
    """ConvTranspose2d(5, s2, p2, op1) on the MFMA kernel (3x3 x 4 phases + fused pixel shuffle)"""
    from vcamd import hip
    from vcamd.layers import deconv_as_subpel_weights
    d = torch.nn.ConvTranspose2d(192, 320, 5, 2, 2, output_padding=1)
    x = torch.randn(1, 192, 17, 30, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = d(x)
    w3, b3 = deconv_as_subpel_weights(d)
    pc = hip.PackedConv(w3, b3, pixelshuffle=True, device=dev)
    out = hip.nhwc_to_nchw(pc(hip.nchw_to_nhwc(x.to(dev))))
    assert out.shape == ref.shape
    assert ((out.cpu() - ref).abs() / (1 + ref.abs())).max().item() < 2e-5


@pytest.mark.parametrize("cin,cout,h,w", [(3, 192, 128, 192), (192, 192, 64, 96), (192, 320, 34, 60), (320, 192, 17, 30)])
def test_conv5x5_stride2(dev, cin, cout, h, w):
    from vcamd import hip
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 5, 5, generator=g) / (cin * 25) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x, wt, b, stride=2, padding=2)
    pc = hip.PackedConv(wt, b, stride=2, device=dev)
    out = hip.nhwc_to_nchw(pc(hip.nchw_to_nhwc(x.to(dev))))
    assert ((out.cpu() - ref).abs() / (1 + ref.abs())).max().item() < 2e-5


def test_image_codec_forward(dev, codecs):
    ora, prod = codecs
    x = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 196, 260, generator=torch.Generator().manual_seed(3)), 5, 1)
    with torch.no_grad():
        ref = ora(x)
        out = prod(x.to(dev))
    assert abs(psnr(out["x_hat"].cpu(), x) - psnr(ref["x_hat"], x)) < 1e-3
    for k in "yz":
        rb = (-torch.log2(ref["likelihoods"][k])).sum().item()
        assert abs(out["bits"][k].item() - rb) / rb < 2e-3


def test_image_codec_bitstream_roundtrip(dev, codecs):
    ora, prod = codecs
    prod.update(force=True)
    ora.update(force=True)
    x = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 132, 196, generator=torch.Generator().manual_seed(4)), 5, 1)
    with torch.no_grad():
        enc = prod.compress(x.to(dev))
        dec = prod.decompress(enc["strings"], enc["shape"])["x_hat"]
        ref_enc = ora.compress(x)
        ref_dec = ora.decompress(ref_enc["strings"], ref_enc["shape"])["x_hat"]
    assert tuple(enc["shape"]) == tuple(ref_enc["shape"])
    for j in range(2):
        assert abs(len(enc["strings"][j][0]) - len(ref_enc["strings"][j][0])) <= max(8, 0.01 * len(ref_enc["strings"][j][0]))
    assert 0.0 <= dec.min().item() and dec.max().item() <= 1.0
    assert abs(psnr(dec.cpu(), x) - psnr(ref_dec, x)) < 5e-3
    # the oracle decodes the product's bitstream to the product's reconstruction (same format)
    with torch.no_grad():
        cross = ora.decompress(enc["strings"], enc["shape"])["x_hat"]
    assert abs(psnr(cross, x) - psnr(dec.cpu(), x)) < 5e-3


def test_sequence_loop_with_iframes(dev, codecs):
    """testing.py loop on a 17-frame synthetic clip: item list, I/B records, shard == whole"""
    from helpers import lhbdc_pair
    from vcamd import gop as vgop
    _, i_model = codecs
    _, b_model = lhbdc_pair(1234, dev)
    g = torch.Generator().manual_seed(5)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 200, 296, generator=g), 9, 1)     # 192x288
    clip = [base[..., :192, i:i + 256].contiguous().to(dev) for i in range(17)]
    with torch.no_grad():
        recs = vgop.code_sequence_lhbdc(b_model, i_model, lambda i: clip[i], 17, 180, 250, video=3)
        shard = vgop.code_sequence_lhbdc(b_model, i_model, lambda i: clip[i], 17, 180, 250, video=3, gop_range=(1, 2))
    intra = [r for r in recs if r[6] == 1]
    inter = [r for r in recs if r[6] == 0]
    assert [r[1] for r in intra] == [0, 8, 16] and len(inter) == 14
    assert sorted(r[1] for r in inter) == [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15]
    # the second GOP coded as a separate shard reproduces the whole-sequence records for its frames
    whole = {r[1]: (float(r[3]), float(r[4])) for r in recs if r[1] > 8}
    part = {r[1]: (float(r[3]), float(r[4])) for r in shard}
    assert part == whole
    table = vgop.RdTable()
    table.extend_from_records(recs, level=7)
    agg = table.per_level_frame_type()
    assert agg[(7, "I")]["frames"] == 3 and agg[(7, "B")]["frames"] == 14


def test_sequence_loop_against_the_reference_test_function(dev):
    """vcamd.gop.code_sequence_lhbdc + RdTable against the rows the reference's own ``test()``
    (LHBDC/test/testing.py:88-196) produced for seven synthetic clips -- fixture lhbdc_test_loop.json, frames
    regenerated bit-exactly by oracle.lhbdc.harness_frames.

    Every hierarchy level is held at the north-star tolerance (1e-3 dB, 1e-3 in size): the fixture is generated on the
    CALIBRATED checkpoint (vcamd.seeding.calibrated_state_dict: sub-pixel flow heads, a decoded residual that is a small
    correction), on which a 1e-6 difference in a decoded reference frame is not amplified from level to level.  (Rounds
    1-3 ran this loop on the plain seeded weights, where unclamped references grow a difference ~100x per level through
    the untrained flow network and level 2 had to be allowed 0.1 dB.)"""
    import json
    import os
    from helpers import GOLDEN, fixture_intra_state_dict, lhbdc_pair
    from oracle import lhbdc as ol
    from vcamd import gop as vgop, iframe
    fx = json.load(open(os.path.join(GOLDEN, "lhbdc_test_loop.json")))
    _, b_model = lhbdc_pair(fx["seed"], dev, calibrated=fx.get("checkpoint") == "calibrated")
    i_model = iframe.mbt2018_mean(7, "mse", pretrained=False)
    i_model.load_state_dict(fixture_intra_state_dict(fx, i_model.state_dict(), fx["intra_seed"]))
    i_model = i_model.to(dev).eval()
    h, w = fx["frame_hw"]
    table = vgop.RdTable()
    assert fx.get("checkpoint") == "calibrated"
    tol = {"I": (1e-3, 1e-4), 0: (1e-3, 1e-3), 1: (1e-3, 1e-3), 2: (1e-3, 1e-3)}
    worst = {k: [0.0, 0.0] for k in tol}
    with torch.no_grad():
        for k, name in enumerate(fx["folders"]):
            frames = [ol.pad64(torch.from_numpy(f.astype("float32").transpose(2, 0, 1))[None] / 255.0).to(dev)
                      for f in ol.harness_frames(fx["seed"], k, fx["frames_per_video"])]
            recs = vgop.code_sequence_lhbdc(b_model, i_model, lambda i: frames[i], len(frames), h, w, video=k, test_size=1)
            ref = [r for r in fx["rows"] if r[0] == name]
            assert len(recs) == len(ref) == 9
            assert [int(r[6]) for r in recs] == [1 if r[1] == "I" else 0 for r in ref]          # I, I, then 7 B-frames
            assert [int(r[1]) % 8 for r in recs[2:]] == [int(r[2]) for r in ref[2:]]            # coding order 4 2 1 3 6 5 7
            for mine, theirs in zip(recs, ref):
                key = "I" if theirs[1] == "I" else vgop.HIER_LEVELS[int(theirs[2])]
                worst[key][0] = max(worst[key][0], abs(float(mine[3]) - theirs[3]))
                worst[key][1] = max(worst[key][1], abs(float(mine[4]) - theirs[4]) / theirs[4])
            table.extend_from_records(recs, 7)
    print("test() loop, worst (dPSNR dB, size rel) per level:", {k: (f"{v[0]:.1e}", f"{v[1]:.1e}") for k, v in worst.items()})
    for key, (tp, ts) in tol.items():
        assert worst[key][0] < tp and worst[key][1] < ts, (key, worst[key])
    (bpp_ref, psnr_ref), = [(float(k), v) for k, v in fx["aggregate_bpp_to_psnr"]["per_level"].items()]
    agg = table.per_level()[7]
    assert abs(agg["bpp"] - bpp_ref) / bpp_ref < 1e-3 and abs(agg["psnr"] - psnr_ref) < 1e-3


def test_config4_gop_shards_union_equals_single_rank(dev, codecs):
    """BASELINE.json configs[3] (the multi-sequence set GOP-sharded over P GPUs, testing.py:99-188): the union of every
    rank's records equals the single-rank run RECORD FOR RECORD (bit-identical PSNR and bits), for P in {2, 3, 8} on two
    clips of 3 + 2 GOPs -- a GOP depends only on its two boundary I-frames, and intra frames on nothing.  The same code
    path drives bench.py --scaling strong; the N > 1 process-group leg runs under gloo in test_dist_cpu.py."""
    from helpers import lhbdc_pair
    from vcamd import gop as vgop
    _, i_model = codecs
    _, b_model = lhbdc_pair(1234, dev)
    g = torch.Generator().manual_seed(15)
    base = torch.nn.functional.avg_pool2d(torch.rand(2, 3, 200, 300, generator=g), 9, 1)      # 192 x 292
    clips = [[base[v:v + 1, :, :192, i:i + 256].contiguous().to(dev) for i in range(n)] for v, n in ((0, 25), (1, 17))]
    plan = vgop.workload_plan([25, 17])
    assert len(plan) == 5

    def run(world, rank, per_pass, graph=False):
        coder = vgop.LhbdcWorkloadCoder(b_model, i_model, lambda v, i: clips[v][i], 180, 250, graph=graph)
        recs = vgop.code_workload(plan, world, rank, coder.intra, coder.code_gops, gops_per_pass=per_pass)
        return [(int(r[0]), int(r[1]), int(r[2]), float(r[3]), float(r[4]), float(r[5]), int(r[6])) for r in recs]

    with torch.no_grad():
        whole = sorted(run(1, 0, 1))
        assert len(whole) == 25 + 17 and len({r[:2] for r in whole}) == 42
        for world in (2, 3, 8):
            union = sorted(sum((run(world, r, 1) for r in range(world)), []))
            assert union == whole, world
        # GOPs batched per pass and the HIP-graph runner give the same records too
        assert sorted(run(1, 0, 2)) == whole
        assert sorted(run(2, 0, 2, graph=True) + run(2, 1, 2, graph=True)) == whole
    rows = vgop.gather_records([tuple(r) for r in whole], dev)
    table = vgop.RdTable()
    table.extend_from_records(rows.tolist(), level=7)
    agg = table.per_level_frame_type()
    assert agg[(7, "I")]["frames"] == 7 and agg[(7, "B")]["frames"] == 35
