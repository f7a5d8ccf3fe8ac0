"""-m gpu: the product used the way the reference's own command-line scripts use their model -- module by module, with
ordinary torch operators on CUDA tensors in between (LHBDC/encode_B.py:71-126, decode_B.py:63-104): `model.FlowNet(a, b)`,
`F.avg_pool2d`, a reflection pad, `torch.cat`, `model.mv_compressor(x)["x_hat"]`, `.compress`, `.decompress`,
`model.upsample_flow`, `model.backwarp`, `model.masknet`.  This is the boundary a user who swaps the import
actually exercises; the fused entry points (`lhbdc.encode_B` / `decode_B`, `Model.forward`) are covered elsewhere.

The script-level path and the fused path are two different launch sequences of the same kernels (torch pools / pads / adds
instead of the fused resampling kernels), so they are compared the way two implementations are: same container header, string
lengths within 1 %, decoded frames within the PSNR bar -- and each path's own decoder reproduces its encoder's
reconstruction of the motion field exactly."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import frame_tensor, lhbdc_pair, load_fixture, psnr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    _, prod = lhbdc_pair(1234, dev)
    prod.mv_compressor.update(force=True)
    prod.residual_compressor.update(force=True)
    return prod


def _pad64(im):
    h, w = im.shape[-2:]
    return nn.ReflectionPad2d((0, (64 - w % 64) % 64, 0, (64 - h % 64) % 64))(im)


def _predictors(model, x_before, x_after):
    """The CLI's predictor wiring (both predictors end up equal to pad(flow_ab))."""
    flow_ab = F.avg_pool2d(model.FlowNet(x_after, x_before) / 2.0, 4)
    hh, ww = flow_ab.shape[-2:]
    flow_ba = _pad64(flow_ab)
    flow_ab = _pad64(flow_ba)
    return flow_ba, flow_ab, hh, ww


def _script_encode(model, x_after, x_current, x_before):
    flow_ba, flow_ab, hh, ww = _predictors(model, x_before, x_after)
    flow_cb = _pad64(F.avg_pool2d(model.FlowNet(x_current, x_before), 4))
    flow_ca = _pad64(F.avg_pool2d(model.FlowNet(x_current, x_after), 4))
    diff_flow = torch.cat([flow_cb - flow_ab, flow_ca - flow_ba], dim=1)
    coded = model.mv_compressor(diff_flow)
    assert set(coded) >= {"x_hat", "likelihoods"} and set(coded["likelihoods"]) == {"y", "z"}
    cb_hat, ca_hat = torch.chunk(coded["x_hat"], 2, dim=1)
    cb_hat = model.upsample_flow((cb_hat + flow_ab)[:, :, :hh, :ww])
    ca_hat = model.upsample_flow((ca_hat + flow_ba)[:, :, :hh, :ww])
    mv_bits = model.mv_compressor.compress(diff_flow)
    fw, bw = model.backwarp(x_before, cb_hat), model.backwarp(x_after, ca_hat)
    one_channel = model.masknet(torch.cat((fw, bw), 1))
    mask = one_channel.expand(-1, 3, -1, -1)
    pred = mask * fw + (1.0 - mask) * bw
    res_bits = model.residual_compressor.compress(x_current - pred)
    return mv_bits, res_bits, coded["x_hat"]


def _script_decode(model, x_before, x_after, string_flow, string_res, shape_flow, shape_res):
    flow_ba, flow_ab, hh, ww = _predictors(model, x_before, x_after)
    flow_hat = model.mv_compressor.decompress(string_flow, shape_flow)["x_hat"]
    cb_hat, ca_hat = torch.chunk(flow_hat, 2, dim=1)
    cb_hat = model.upsample_flow((cb_hat + flow_ab)[:, :, :hh, :ww])
    ca_hat = model.upsample_flow((ca_hat + flow_ba)[:, :, :hh, :ww])
    fw, bw = model.backwarp(x_before, cb_hat), model.backwarp(x_after, ca_hat)
    mask = model.masknet(torch.cat((fw, bw), 1)).expand(-1, 3, -1, -1)
    res_hat = model.residual_compressor.decompress(string_res, shape_res)["x_hat"]
    return res_hat + (mask * fw + (1.0 - mask) * bw), flow_hat


def test_cli_call_sequence_through_the_public_modules(dev, model):
    from vcamd import lhbdc
    fx = load_fixture("lhbdc_codec_a.npz")
    h, w = fx["current"].shape[:2]
    xb, xc, xa = (lhbdc.process_frame(fx[k].astype(np.float32), dev) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        mv_bits, res_bits, mv_hat_enc = _script_encode(model, xa, xc, xb)
        fused_mv, fused_res = lhbdc.encode_B(model, xa, xc, xb)
    # same container as the fused entry point and as the reference: shapes exactly, string lengths to 1 %
    assert tuple(mv_bits["shape"]) == tuple(fused_mv["shape"]) == tuple(fx["mv_shape"])
    assert tuple(res_bits["shape"]) == tuple(fused_res["shape"]) == tuple(fx["res_shape"])
    for mine, fused, ref in ((mv_bits["strings"][0][0], fused_mv["strings"][0][0], fx["mv_y"]),
                             (mv_bits["strings"][1][0], fused_mv["strings"][1][0], fx["mv_z"]),
                             (res_bits["strings"][0][0], fused_res["strings"][0][0], fx["res_y"]),
                             (res_bits["strings"][1][0], fused_res["strings"][1][0], fx["res_z"])):
        assert abs(len(mine) - len(ref)) <= max(2, 0.01 * len(ref))
        assert abs(len(mine) - len(fused)) <= max(2, 0.01 * len(fused))
    blob = lhbdc.write_container(None, 1626, mv_bits, res_bits)
    lm, s_mv, s_res, shape_mv, shape_res = lhbdc.read_container(blob)
    assert lm == 1626
    with torch.no_grad():
        dec, mv_hat_dec = _script_decode(model, xb, xa, s_mv, s_res, shape_mv, shape_res)
        dec_fused = lhbdc.decode_B(xb, xa, model, s_mv, s_res, shape_mv, shape_res)
    # the decoder rebuilds exactly the motion field the encoder reconstructed from the same integers
    assert torch.equal(mv_hat_dec, mv_hat_enc)
    src = frame_tensor(fx["current"])
    ref_dec = torch.from_numpy(fx["decoded"])
    p_script, p_fused, p_ref = (psnr(t.cpu()[..., :h, :w], src) for t in (dec, dec_fused, ref_dec))
    print(f"script-level decode {p_script:.4f} dB, fused decode of the same container {p_fused:.4f} dB, reference {p_ref:.4f} dB, "
          f"max|script - fused| = {(dec - dec_fused).abs().max().item():.2e}")
    assert abs(p_script - p_ref) < 5e-3 and abs(p_script - p_fused) < 1e-3
    u8 = lhbdc.float_to_uint8(dec[0].cpu().numpy())[:h, :w]
    assert (np.abs(u8.astype(int) - fx["decoded_u8"].astype(int)) > 1).mean() < 1e-3


def test_script_decoder_reads_the_reference_container(dev, model):
    """The reference's own bits_B.bin (fixture) through the module-level decode sequence."""
    from vcamd import lhbdc
    fx = load_fixture("lhbdc_codec_a.npz")
    h, w = fx["current"].shape[:2]
    xb, xa = (lhbdc.process_frame(fx[k].astype(np.float32), dev) for k in ("ref_1", "ref_2"))
    lm, s_mv, s_res, shape_mv, shape_res = lhbdc.read_container(fx["container"].tobytes())
    with torch.no_grad():
        dec, _ = _script_decode(model, xb, xa, s_mv, s_res, shape_mv, shape_res)
    ref_dec = torch.from_numpy(fx["decoded"])
    src = frame_tensor(fx["current"])
    assert abs(psnr(dec.cpu()[..., :h, :w], src) - psnr(ref_dec[..., :h, :w], src)) < 5e-3
    u8 = lhbdc.float_to_uint8(dec[0].cpu().numpy())[:h, :w]
    assert (np.abs(u8.astype(int) - fx["decoded_u8"].astype(int)) > 1).mean() < 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# Flex-Rate: the module-level sequence of Flex-Rate.../test/encode_B.py:73-109 and decode_B.py:74-96
# (`model.process`, the gained compressors called with ([n], l), `model.backwarp`, `model.Mask` + sigmoid, the weighted blend)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def flex_model(dev):
    from vcamd import flex
    from vcamd.seeding import seeded_state_dict
    m = flex.BidirFlowRef(n=4)
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed=1234))
    m = m.to(dev).eval()
    m.flow_compressor.update(force=True)
    m.residual_compressor.update(force=True)
    return m


def _flex_prediction(model, x_before, x_after, refinement):
    mv_b, mv_a, _ = model.process(x_before, x_after)
    mv_b, mv_a = mv_b + refinement[:, 0:2], mv_a + refinement[:, 2:4]
    warped_b, warped_a = model.backwarp(x_before, mv_b), model.backwarp(x_after, mv_a)
    gate = torch.sigmoid(model.Mask(torch.cat((mv_b, mv_a, x_before, x_after, warped_b, warped_a), 1)))
    wb, wa = 0.5 * gate[:, 0:1], 0.5 * gate[:, 1:2]
    return (wb * warped_b + wa * warped_a) / (wb + wa + 1e-8)


def test_flex_cli_call_sequence_through_the_public_modules(dev, flex_model):
    from vcamd import flex
    m = flex_model
    fx = load_fixture("flex_codec_a.npz")
    n, l = int(fx["n"]), float(fx["l"])
    xb, xc, xa = (frame_tensor(fx[k]).to(dev) for k in ("ref_1", "current", "ref_2"))
    with torch.no_grad():
        _, _, context = m.process(xb, xa)
        codec_in = torch.cat((context, xc), 1)
        mv_bits = m.flow_compressor.compress(codec_in, [n], l)
        coded = m.flow_compressor(codec_in, [n], l, False)
        assert set(coded) >= {"x_hat", "likelihoods"}
        pred = _flex_prediction(m, xb, xa, coded["x_hat"])
        res_bits = m.residual_compressor.compress(xc - pred, [n], l)
        fused_mv, fused_res = flex.encode_B(m, xb, xc, xa, n=n, l=l)
    assert tuple(mv_bits["shape"]) == tuple(fused_mv["shape"]) == tuple(fx["flow_shape"])
    assert tuple(res_bits["shape"]) == tuple(fused_res["shape"]) == tuple(fx["res_shape"])
    for mine, fused, ref in ((mv_bits["strings"][0][0], fused_mv["strings"][0][0], fx["flow_y"]),
                             (mv_bits["strings"][1][0], fused_mv["strings"][1][0], fx["flow_z"]),
                             (res_bits["strings"][0][0], fused_res["strings"][0][0], fx["res_y"]),
                             (res_bits["strings"][1][0], fused_res["strings"][1][0], fx["res_z"])):
        assert abs(len(mine) - ref.size) <= max(8, 0.01 * ref.size)
        assert abs(len(mine) - len(fused)) <= max(8, 0.01 * len(fused))
    # decoder side on the strings just written, and on the REFERENCE's own strings
    src = frame_tensor(fx["current"])
    ref_dec = torch.from_numpy(fx["decoded"])
    for s_mv, s_res in ((mv_bits["strings"], res_bits["strings"]),
                        ([[fx["flow_y"].tobytes()], [fx["flow_z"].tobytes()]], [[fx["res_y"].tobytes()], [fx["res_z"].tobytes()]])):
        with torch.no_grad():
            refinement = m.flow_compressor.decompress(s_mv, mv_bits["shape"], [n], l)["x_hat"]
            dec = m.residual_compressor.decompress(s_res, res_bits["shape"], [n], l)["x_hat"] + _flex_prediction(m, xb, xa, refinement)
            fused = flex.decode_B(m, xb, xa, s_mv, s_res, mv_bits["shape"], res_bits["shape"], n, l)
        assert abs(psnr(dec.cpu(), src) - psnr(ref_dec, src)) < 5e-3
        assert abs(psnr(dec.cpu(), src) - psnr(fused.cpu(), src)) < 1e-3
