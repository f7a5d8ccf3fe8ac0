"""-m gpu: the fp64 refinement kernels of the bitstream paths (vc_refine_y_symbols / vc_refine_z_symbols / vc_refine_scales) against
float64 CPU convolutions.

What they are for (LHBDC/model/layers.py:93-104,168-179): a coded integer is round(y - mu) or round(z - median); when the difference
sits within fp32 summation noise of a half-integer the integer depends on the platform's summation order.  The kernels recompute the
producing convolutions for exactly those elements in fp64 and round once.  The checks: (1) the set of rewritten elements is the set
within eps of a boundary, nothing else is touched; (2) a rewritten symbol equals round(fl32(conv64 y) - fl32(conv64 mu)), the value an
exact convolution gives; (3) y / mu in memory are untouched and y_hat follows the new symbol with the decoder's own mu.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

EPS = 2e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _nhwc(x_nchw, dev):
    from vcamd import hip
    return hip.nchw_to_nhwc(x_nchw.to(dev))


def _conv64(x, w, b, stride):
    return F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=w.shape[-1] // 2)


@pytest.mark.parametrize("k,stride,cin,c,h,w,n", [(3, 2, 128, 128, 18, 34, 1), (3, 1, 192, 64, 9, 12, 2), (5, 2, 48, 32, 21, 19, 1),
                                                   (1, 1, 32, 32, 8, 8, 1)])
def test_refined_y_symbols_are_those_of_an_exact_convolution(dev, k, stride, cin, c, h, w, n):
    from vcamd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(100 * k + stride)
    xa = torch.randn(n, cin, h, w, generator=g)
    wa = torch.randn(c, cin, k, k, generator=g) * (2.0 / (k * k * cin) ** 0.5)
    ba = torch.randn(c, generator=g)
    y64 = _conv64(xa, wa, ba, stride)
    ho, wo = y64.shape[-2:]
    # mu comes out of a 3x3 stride-1 convolution with 2c output channels (scales | means), like the hyper-synthesis transform's last layer
    cb = 40
    xb = torch.randn(n, cb, ho, wo, generator=g)
    wb = torch.randn(2 * c, cb, 3, 3, generator=g) * (1.0 / (9 * cb) ** 0.5)
    bb = torch.randn(2 * c, generator=g)
    mu64 = _conv64(xb, wb, bb, 1)[:, c:]
    # About a quarter of the elements are steered onto a rounding boundary (distance <= 4e-6).  A convolution output cannot be set
    # element by element, so the means layer gets c extra input channels that it copies through its centre tap (weight 1): the
    # correction map rides on them, and mu is still a genuine output of the layer the kernel recomputes.
    want = torch.rand(n, c, ho, wo, generator=g) < 0.25
    delta = (torch.rand(n, c, ho, wo, generator=g) - 0.5) * 8e-6
    j = torch.randint(-3, 4, (n, c, ho, wo), generator=g).double()
    corr = torch.where(want, (y64 - (j + 0.5) - delta.double()) - mu64, torch.zeros_like(mu64))
    xb2 = torch.cat([xb, corr.float()], 1)
    wb2 = torch.zeros(2 * c, cb + c, 3, 3)
    wb2[:, :cb] = wb
    for i in range(c):
        wb2[c + i, cb + i, 1, 1] = 1.0
    mu64 = _conv64(xb2, wb2, bb, 1)[:, c:]
    y32, mu32 = y64.float(), mu64.float()                   # correctly rounded values: what the kernel must arrive at
    noise = lambda t, s: (t.double() + (torch.rand(t.shape, generator=g).double() - 0.5) * s).float()
    y_mem, mu_mem = noise(y32, 3e-6), noise(mu32, 3e-6)     # what an fp32 engine leaves in memory
    d_mem = y_mem - mu_mem
    near = ((d_mem - torch.floor(d_mem)) - 0.5).abs() <= EPS
    assert int(near.sum()) >= int(0.2 * near.numel()) and int((~near).sum()) > 0
    sym_plain = torch.round(d_mem).to(torch.int32)
    expect = torch.where(near, torch.round(y32 - mu32).to(torch.int32), sym_plain)
    assert int((expect != sym_plain).sum()) > 0, "the case must contain symbols the noise flips"

    ta, tb = _nhwc(xa, dev), _nhwc(xb2, dev)
    ty, tmu = _nhwc(y_mem, dev), _nhwc(mu_mem, dev)
    wa_d, ba_d, wb_d, bb_d = wa.to(dev).contiguous(), ba.to(dev), wb2.to(dev).contiguous(), bb.to(dev)
    sym = sym_plain.to(dev).contiguous()
    y_hat = hip.T.empty(n, ho, wo, c, dev)
    y_hat.buf.fill_(-777.0)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    la = hip.RefineLayer(ta.view(), wa_d.data_ptr(), ba_d.data_ptr(), k, stride, 0)
    lb = hip.RefineLayer(tb.view(), wb_d.data_ptr(), bb_d.data_ptr(), 3, 1, c)
    y_keep, mu_keep = ty.buf.clone(), tmu.buf.clone()
    hip.check(L.vc_refine_y_symbols(hip.stream(), ty.view(), la, tmu.view(), lb, EPS, sym.data_ptr(), y_hat.view(), None,
                                    counter.data_ptr()), "vc_refine_y_symbols")
    torch.cuda.synchronize()
    assert int(counter.item()) == int(near.sum())
    got = sym.cpu()
    bad = int((got != expect).sum())
    # (fp64 accumulation in another order than the CPU's: a value whose fp64 sum sits within 1e-12 of an fp32 rounding boundary could
    # still round the other way -- none in these seeded cases)
    assert bad == 0, f"{bad} of {int(near.sum())} refined symbols differ from the exact convolution's"
    assert torch.equal(ty.buf, y_keep) and torch.equal(tmu.buf, mu_keep), "y / mu in memory must stay the decoder's"
    hat = hip.nhwc_to_nchw(y_hat).cpu()
    assert torch.equal(hat[near], (expect.float() + mu_mem)[near])
    assert bool((hat[~near] == -777.0).all())


def test_refined_z_symbols_and_z_hat(dev):
    from vcamd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(7)
    n, cin, c, h, w = 1, 64, 32, 12, 20
    x = torch.randn(n, cin, h, w, generator=g)
    wz = torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin) ** 0.5)
    bz = torch.randn(c, generator=g)
    z64 = _conv64(x, wz, bz, 2)
    ho, wo = z64.shape[-2:]
    params = torch.zeros(c, hip.EB_PARAMS_PER_CHANNEL)
    med = torch.randn(c, generator=g)
    params[:, 58] = med
    gain = torch.rand(c, generator=g) + 0.5
    inv_gain = torch.rand(c, generator=g) + 0.5
    # boundary cases through the bias: channel i's bias is chosen so that its first pixel lands 2e-6 from a half-integer
    for i in range(c):
        v = z64[0, i, 0, 0].item() * gain[i].item() - med[i].item()
        target = np.floor(v) + 0.5 + (2e-6 if i % 2 else -2e-6)
        bz[i] += float((target - v) / gain[i].item())
    z64 = _conv64(x, wz, bz, 2)
    z32 = z64.float()
    z_mem = (z32.double() + (torch.rand(z32.shape, generator=g).double() - 0.5) * 4e-6).float()
    d_mem = z_mem * gain.view(1, -1, 1, 1) - med.view(1, -1, 1, 1)
    near = ((d_mem - torch.floor(d_mem)) - 0.5).abs() <= EPS
    assert int(near.sum()) >= c // 2
    sym_plain = torch.round(d_mem).to(torch.int32)
    expect = torch.where(near, torch.round(z32 * gain.view(1, -1, 1, 1) - med.view(1, -1, 1, 1)).to(torch.int32), sym_plain)
    tx, tz = _nhwc(x, dev), _nhwc(z_mem, dev)
    w_d, b_d, p_d, g_d, ig_d = wz.to(dev).contiguous(), bz.to(dev), params.to(dev).contiguous(), gain.to(dev), inv_gain.to(dev)
    sym = sym_plain.to(dev).contiguous()
    z_hat = hip.T.empty(n, ho, wo, c, dev)
    z_hat.buf.fill_(-777.0)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    lz = hip.RefineLayer(tx.view(), w_d.data_ptr(), b_d.data_ptr(), 3, 2, 0)
    hip.check(L.vc_refine_z_symbols(hip.stream(), tz.view(), lz, p_d.data_ptr(), g_d.data_ptr(), EPS, sym.data_ptr(), z_hat.view(),
                                    ig_d.data_ptr(), counter.data_ptr()), "vc_refine_z_symbols")
    torch.cuda.synchronize()
    assert int(counter.item()) == int(near.sum())
    assert torch.equal(sym.cpu(), expect)
    hat = hip.nhwc_to_nchw(z_hat).cpu()
    want_hat = (expect.float() + med.view(1, -1, 1, 1)) * inv_gain.view(1, -1, 1, 1)
    assert torch.equal(hat[near], want_hat[near]) and bool((hat[~near] == -777.0).all())


def test_refine_rejects_shapes_that_do_not_belong_together(dev):
    from vcamd import hip
    L = hip.lib()
    t = hip.T.empty(1, 8, 8, 16, dev)
    y = hip.T.empty(1, 4, 4, 16, dev)
    w = torch.zeros(16, 16, 3, 3, device=dev)
    sym = torch.zeros(256, dtype=torch.int32, device=dev)
    ok = hip.RefineLayer(t.view(), w.data_ptr(), None, 3, 2, 0)
    s1 = hip.RefineLayer(t.view(), w.data_ptr(), None, 3, 1, 0)          # stride 1 would give 8x8, not y's 4x4
    assert L.vc_refine_y_symbols(hip.stream(), y.view(), s1, y.view(), ok, EPS, sym.data_ptr(), hip.NULL_VIEW, None, None) == -1
    assert L.vc_refine_y_symbols(hip.stream(), y.view(), ok, y.view(), ok, 0.0, sym.data_ptr(), hip.NULL_VIEW, None, None) == -1
    assert L.vc_refine_y_symbols(hip.stream(), y.view(), ok, y.view(), ok, EPS, None, hip.NULL_VIEW, None, None) == -1
    even = hip.RefineLayer(t.view(), w.data_ptr(), None, 4, 2, 0)
    assert L.vc_refine_z_symbols(hip.stream(), y.view(), even, w.data_ptr(), None, EPS, sym.data_ptr(), hip.NULL_VIEW, None, None) == -1


def test_bitstream_round_trip_with_symbol_refinement(dev):
    """The closed loop: what the encoder's one-pass coder (code_t) reconstructs is what the decoder rebuilds from the integers,
    with the refinement on -- the refined elements' y_hat uses the decoder's mu."""
    from helpers import lhbdc_pair
    from vcamd import hip
    _, prod = lhbdc_pair(1234, dev, calibrated=True)
    rc = prod.residual_compressor
    rc.update(force=True)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(1, 3, 128, 192, generator=g) - 0.5) * 0.2
    assert hip.SYMBOL_REFINE
    with torch.no_grad():
        x_hat, ints = rc.code_t(hip.nchw_to_nhwc(x.to(dev)))
        means, idx = rc.hyper_decode_t(ints["z_sym"], 1, ints["shape"])
        assert torch.equal(idx, ints["y_idx"])
        dec = rc.synth_decode_t(ints["y_sym"], means)
        torch.cuda.synchronize()
        assert torch.equal(hip.nhwc_to_nchw(dec), hip.nhwc_to_nchw(x_hat))


@pytest.mark.parametrize("h,w", [(384, 640), (1088, 1920)], ids=["384x640", "1088x1920"])
def test_streams_decode_across_fp32_modes(dev, h, w):
    """A container written under one fp32 mode (VC_FP32_MODE native / split: different summation orders in the analysis layers) decodes
    under the other: the hyper-synthesis transform of the bitstream paths -- the one place where the decoder's arithmetic decides how the
    stream is READ (scale-table indexes) -- is pinned to one pipeline (hip.BITSTREAM_HS_MODE).  The decoder's integers are the
    encoder's, both directions; the decoded frames of the two modes differ by synthesis-side summation noise only."""
    from helpers import lhbdc_pair
    from test_byte_equality_gpu import triple
    from vcamd import hip, lhbdc
    _, prod = lhbdc_pair(1234, dev, calibrated=True)
    prod.mv_compressor.update(force=True)
    prod.residual_compressor.update(force=True)
    keep = hip.fp32_mode()
    try:
        with torch.no_grad():
            for seed in (11, 37):
                xb, xc, xa = (t.to(dev) for t in triple(seed, h, w))
                for enc_mode, dec_mode in (("native", "split"), ("split", "native")):
                    hip.set_fp32_mode(enc_mode)
                    te = {}
                    mv_bits, res_bits = lhbdc.encode_B(prod, xa, xc, xb, trace=te)
                    blob = lhbdc.write_container(None, 1626, mv_bits, res_bits)
                    _, s_mv, s_res, sh_mv, sh_res = lhbdc.read_container(blob)
                    frames = {}
                    for mode in (dec_mode, enc_mode):
                        hip.set_fp32_mode(mode)
                        td = {}
                        frames[mode] = lhbdc.decode_B(xb, xa, prod, s_mv, s_res, sh_mv, sh_res, trace=td)
                        for c in ("mv", "res"):
                            for k in ("z_sym", "y_idx", "y_sym"):
                                assert np.array_equal(np.asarray(td[c][k]).reshape(-1), np.asarray(te[c][k]).reshape(-1)), (seed, enc_mode, mode, c, k)
                    d = (frames[dec_mode] - frames[enc_mode]).abs().max().item()
                    assert d < 1e-4, (seed, enc_mode, dec_mode, d)
    finally:
        hip.set_fp32_mode(keep)
