"""-m gpu: the split-operand fp32 convolution (VC_CFG_SPLIT, csrc/conv_split.h; the default fp32 mode since round 5) against an
fp64 CPU reference and against the native fp32 MFMA instances.

Bars: error against fp64 no worse than 1.5 x the native instance's; the reference-parity tests
(test_reference_1080p_gpu.py, test_lhbdc_gpu.py, test_fullsize_gpu.py) run in the default mode, i.e. on this path.
Reference call sites: LHBDC/model/flow.py:52-62 (7x7 Basic blocks), LHBDC/model/layers.py:202-209 (5x5 mask U-Net layers).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_mode():
    from vcamd import hip
    keep = hip.fp32_mode()
    yield
    hip.set_fp32_mode(keep)


def _layer(cin, cout, k, seed, dev, bias=True):
    from vcamd import hip
    g = torch.Generator().manual_seed(seed)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    return wt, b, hip.PackedConv(wt, b, stride=1, device=dev)


CASES = [  # cin, cout, k, n, h, w  (>= 48 tiles of 16 x 32 per image: the sizes the split pipeline takes)
    (128, 128, 3, 2, 100, 170),    # 3x3: two planes per chunk, 12-row tiles, ragged edges
    (64, 32, 3, 1, 150, 260),      # 3x3, one block of 32 output channels
    (256, 128, 3, 1, 61, 130),
    (32, 64, 7, 1, 112, 224),
    (64, 32, 7, 2, 100, 250),      # ragged right / bottom tiles, two images
    (8, 32, 7, 1, 128, 200),       # one chunk per tile
    (32, 16, 7, 2, 250, 330),      # one N-tile of 16 channels per workgroup, 24-row tiles
    (96, 32, 5, 1, 120, 230),
    (192, 64, 5, 1, 120, 200),
    (32, 128, 7, 1, 97, 130),      # two blocks of 64 output channels
    (16, 32, 7, 1, 128, 200),      # 7x7 period instance, ONE period of two 8-channel chunks per tile
    (48, 64, 5, 1, 120, 230),      # 5x5 period instance, three periods
    (24, 32, 5, 1, 120, 230),      # 5x5, input channels no multiple of 16: the per-chunk (tap-padded) instance
    (48, 64, 3, 1, 100, 170),      # 3x3, input channels no multiple of 32: the per-chunk instance
    (6, 32, 5, 1, 120, 230),       # input channels padded to the chunk with zero weights (vc_split3_pad): the mask U-Net's first layer
    (6, 32, 3, 2, 150, 260),       # 3x3: 6 -> 16 channels
    (24, 64, 3, 1, 100, 170),      # 3x3: 24 -> 32 channels (one period)
]


@pytest.mark.parametrize("cin,cout,k,n,h,w", CASES)
def test_split_conv_against_fp64_and_the_native_instance(dev, cin, cout, k, n, h, w):
    from vcamd import hip
    wt, b, pc = _layer(cin, cout, k, 3, dev)
    assert pc.split_ok and pc.split_pays(n, h, w)
    g = torch.Generator().manual_seed(4)
    xc = torch.randn(n, cin, h, w, generator=g)
    rc = torch.randn(n, cout, h, w, generator=g)
    gain = torch.rand(cout, generator=g) + 0.5
    x, res = hip.nchw_to_nhwc(xc.to(dev)), hip.nchw_to_nhwc(rc.to(dev))
    ref = F.leaky_relu(F.conv2d(xc.double(), wt.double(), b.double(), padding=k // 2), 0.1) * gain.double().view(1, -1, 1, 1) + rc.double()
    mag = F.conv2d(xc.double().abs(), wt.double().abs(), b.double().abs(), padding=k // 2) + rc.double().abs()
    err = {}
    for mode in ("native", "split"):
        hip.set_fp32_mode(mode)
        y = hip.nhwc_to_nchw(pc(x, act=hip.ACT_LRELU, slope=0.1, res=res, chscale=gain.to(dev))).cpu().double()
        e = (y - ref).abs() / mag
        err[mode] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    print(f"k{k} {cin}->{cout} @{n}x{h}x{w}: max / rms error over sum|a b| against fp64: native {err['native'][0]:.2e} / {err['native'][1]:.2e}, "
          f"split {err['split'][0]:.2e} / {err['split'][1]:.2e}")
    assert err["split"][0] <= 1.5 * err["native"][0] and err["split"][1] <= 1.5 * err["native"][1], err
    assert err["split"][0] < 6e-7


def test_split_tensor_is_exact_and_a_split_chain_never_converts(dev):
    """vc_split3: hi + mid + lo == x bit for bit (also for subnormal-range and huge values); the OUT_SP3 epilogue writes the same
    records vc_split3 would make of the fp32 result, so a chain through a split intermediate equals the chain through fp32."""
    from vcamd import hip
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 40, 72, generator=g)
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -3e38, 1.0, 2.0 ** -120, 1.0 + 2.0 ** -23, -(2.0 - 2.0 ** -23)])
    xt = hip.nchw_to_nhwc(x.to(dev))
    sp = hip.split3(xt)
    back = _unsplit(sp)
    assert torch.equal(back, x)
    hip.set_fp32_mode("split")
    wt1, b1, p1 = _layer(32, 64, 7, 6, dev)
    wt2, b2, p2 = _layer(64, 32, 7, 7, dev)
    xin = hip.T.empty(2, 112, 224, 32, dev)
    xin.buf.normal_()
    mid_sp = p1(xin, act=hip.ACT_RELU, out_sp3=True)
    assert mid_sp.dtype == "sp3"
    a = hip.nhwc_to_nchw(p2(mid_sp, act=hip.ACT_RELU))
    mid = p1(xin, act=hip.ACT_RELU)
    assert mid.dtype == "f32"
    bb = hip.nhwc_to_nchw(p2(mid, act=hip.ACT_RELU))
    assert torch.equal(a, bb)


def test_split_result_does_not_depend_on_the_batch(dev):
    """A frame gets the same bits coded alone or inside a level-batched pass (the pipeline choice is per image, the accumulation
    order per output fixed)."""
    from vcamd import hip
    hip.set_fp32_mode("split")
    _, _, pc = _layer(32, 64, 7, 8, dev)
    x = hip.T.empty(3, 112, 224, 32, dev)
    x.buf.normal_()
    full = hip.nhwc_to_nchw(pc(x, act=hip.ACT_RELU))
    for i in range(3):
        one = hip.nhwc_to_nchw(pc(x.images(i, i + 1), act=hip.ACT_RELU))
        assert torch.equal(one, full[i:i + 1])
    # three repetitions: no run-to-run variation (counted waits, no atomics)
    again = hip.nhwc_to_nchw(pc(x, act=hip.ACT_RELU))
    assert torch.equal(again, full)


@pytest.mark.parametrize("cin,cout,k,n,h,w", [(64, 32, 7, 3, 272, 480), (96, 32, 5, 2, 544, 960), (128, 128, 3, 3, 272, 480), (32, 16, 7, 2, 544, 960)])
def test_period_instances_same_bits_alone_in_a_batch_and_launch_after_launch(dev, cin, cout, k, n, h, w):
    """The period instances (two chunks per period; csrc/conv_split.h SplitPeriodCfg) and the 16-channel per-chunk instance at sizes where
    every CU walks several tiles and the next tile's first chunk is requested under the current tile's last period."""
    from vcamd import hip
    hip.set_fp32_mode("split")
    _, _, pc = _layer(cin, cout, k, 9, dev)
    x = hip.T.empty(n, h, w, cin, dev)
    x.buf.normal_()
    xs = hip.split3(x)
    full = pc(xs, act=hip.ACT_RELU).buf.clone()
    for _ in range(3):
        assert torch.equal(pc(xs, act=hip.ACT_RELU).buf, full)
    for i in range(n):
        assert torch.equal(pc(xs.images(i, i + 1), act=hip.ACT_RELU).buf.reshape(-1), full.reshape(n, -1)[i])


def test_split_path_refuses_what_it_does_not_serve(dev):
    from vcamd import hip
    hip.set_fp32_mode("split")
    _, _, pc = _layer(32, 64, 7, 9, dev)
    x = hip.T.empty(1, 112, 224, 32, dev)
    x.buf.normal_()
    sp = hip.split3(x)
    with pytest.raises(hip.VcError):
        pc(sp, act=hip.ACT_SIGMOID)            # sigmoid epilogue: not on the split pipeline
    small = hip.T.empty(1, 34, 60, 32, dev)   # a coarse pyramid level: stays on the native instances
    small.buf.normal_()
    assert not pc.split_pays(1, 34, 60)
    y = pc(small, act=hip.ACT_RELU)
    hip.set_fp32_mode("native")
    assert torch.equal(hip.nhwc_to_nchw(y), hip.nhwc_to_nchw(pc(small, act=hip.ACT_RELU)))


def test_lhbdc_forward_split_and_native_code_the_same_integers(dev):
    """Model level: the default (split) mode against the native mode on one fixture-sized frame triple -- same symbols, x_hat within
    fp32 summation noise."""
    from vcamd import hip, lhbdc
    from vcamd.seeding import calibrated_state_dict
    m = lhbdc.Model()
    m.load_state_dict(calibrated_state_dict(m.state_dict(), seed=1234))
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 264, 400, generator=g), 9, 1)
    xb, xc, xa = (base[..., :256, i:i + 384].contiguous().to(dev) for i in (0, 3, 6))
    out = {}
    with torch.no_grad():
        for mode in ("native", "split"):
            hip.set_fp32_mode(mode)
            tr = {}
            x_hat, tot = m.forward_device(xb, xc, xa, trace=tr)
            out[mode] = (x_hat.cpu(), tot.cpu(), {f"{k}_{w}": tr[k][w].cpu() for k in ("mv", "res") for w in ("y_sym", "z_sym")})
    d = float((out["native"][0] - out["split"][0]).abs().max())
    flips = {k: int((out["native"][2][k] != out["split"][2][k]).sum()) for k in out["native"][2]}
    print(f"LHBDC 256x384 native vs split: x_hat max|d| {d:.2e}, symbols differing {flips}")
    assert sum(flips.values()) <= 1 and (d < 1e-4 or sum(flips.values()) == 1)


def _unsplit(t):
    """split tensor window -> NCHW fp32 (exact: (lo + mid) + hi)"""
    full = t.buf.view(-1)[: t.buf.numel()]
    n, h, w, c = t.n, t.h, t.w, t.c
    per_img = t.sn * h * w * 24                     # int16 elements per image of the underlying buffer
    out = torch.empty(n, c, h, w)
    for i in range(n):
        start = i * per_img + t.off * h * w * 24
        rec = full[start:start + (c // 8) * h * w * 24].view(c // 8, h, w, 3, 8).to(torch.int32)
        pieces = (rec << 16).view(torch.float32)
        v = (pieces[..., 2, :] + pieces[..., 1, :]) + pieces[..., 0, :]
        out[i] = v.permute(0, 3, 1, 2).reshape(c, h, w).cpu()
    return out


def test_native_layers_write_split_tensors_and_split_layers_add_split_residuals(dev):
    """fp32 mode "split": the classic fp32 instances in front of a split consumer (stride-2 3x3, 1x1, GDN) write the three bf16
    pieces themselves (CFG_OUT_SP3) -- exactly the fp32 values they would have stored; a split layer adds an identity that is a
    split tensor (CFG_RES_SP3) exactly like the fp32 tensor of the same values; windows of planes inside a wider split tensor."""
    from vcamd import hip
    from vcamd.layers import GDN
    hip.set_fp32_mode("split")
    g = torch.Generator().manual_seed(21)
    # (a) stride-2 3x3 128 -> 128, LeakyReLU: split output == fp32 output
    wt, b, pc = _layer(128, 128, 3, 22, dev)
    pc2 = hip.PackedConv(wt, b, stride=2, device=dev)
    x = hip.T.empty(2, 96, 160, 128, dev)
    x.buf.normal_()
    y32 = hip.nhwc_to_nchw(pc2(x, act=hip.ACT_LRELU, slope=0.01)).cpu()
    ysp = pc2(x, act=hip.ACT_LRELU, slope=0.01, out_sp3=True)
    assert ysp.dtype == "sp3" and torch.equal(_unsplit(ysp), y32)
    # (b) GDN on the classic 1x1 instance, split output, with a residual
    gdn = GDN(128).to(dev)
    r = hip.T.empty(2, 96, 160, 128, dev)
    r.buf.normal_()
    z32 = hip.nhwc_to_nchw(gdn.run(x, res=r)).cpu()
    zsp = gdn.run(x, res=r, out_sp3=True)
    assert zsp.dtype == "sp3" and torch.equal(_unsplit(zsp), z32)
    # (c) a split layer with a split residual == the same with the fp32 residual; split input window of a wider split tensor
    xs_wide = hip.T.empty(2, 96, 160, 256, dev, "sp3")
    hip.split3(x, out=xs_wide.channels(128, 256))
    hip.split3(r, out=xs_wide.channels(0, 128))
    a = hip.nhwc_to_nchw(pc(xs_wide.channels(128, 256), act=hip.ACT_LRELU, slope=0.01, res=xs_wide.channels(0, 128)))
    bb = hip.nhwc_to_nchw(pc(x, act=hip.ACT_LRELU, slope=0.01, res=r))
    assert torch.equal(a, bb)
    # (d) split OUTPUT into a window of planes
    o_wide = hip.T.empty(2, 96, 160, 256, dev, "sp3")
    o_wide.buf.zero_()
    pc(x, act=hip.ACT_LRELU, slope=0.01, out=o_wide.channels(128, 256))
    assert torch.equal(_unsplit(o_wide.channels(128, 256)), hip.nhwc_to_nchw(pc(x, act=hip.ACT_LRELU, slope=0.01)).cpu())
    assert float(_unsplit(o_wide.channels(0, 128)).abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,k,n,h,w", [(128, 128, 3, 2, 100, 170), (32, 64, 7, 1, 101, 203), (64, 32, 7, 2, 100, 250), (192, 64, 5, 1, 131, 203),
                                                (32, 128, 7, 1, 97, 130), (64, 32, 3, 1, 150, 261)])
def test_record_epilogue_equals_the_plain_one(dev, cin, cout, k, n, h, w):
    """Round 6: a split output leaves the period kernels through LDS as whole records and a split residual is read as whole records
    (csrc/conv_split.h: split_epilogue_records).  Same arithmetic as the fp32-output / fp32-residual epilogue: the split result must hold
    exactly the fp32 result's values -- ragged right / bottom tiles (rows of records that must not run into the next image row), blocks of
    32 and 64 output channels, with and without residual, residual before / after the activation, channel gain."""
    from vcamd import hip
    hip.set_fp32_mode("split")
    wt, b, pc = _layer(cin, cout, k, 31, dev)
    assert pc.split_ok and pc.split_pays(n, h, w)
    g = torch.Generator().manual_seed(32)
    x = hip.nchw_to_nhwc(torch.randn(n, cin, h, w, generator=g).to(dev))
    r = hip.nchw_to_nhwc(torch.randn(n, cout, h, w, generator=g).to(dev))
    gain = (torch.rand(cout, generator=g) + 0.5).to(dev)
    xs, rs = hip.split3(x), hip.split3(r)
    guard = 7.5
    for res32, ressp, first, ch in ((None, None, False, None), (r, rs, False, None), (r, rs, True, gain), (None, None, False, gain)):
        want = hip.nhwc_to_nchw(pc(xs, act=hip.ACT_LRELU, slope=0.1, res=res32, res_first=first, chscale=ch)).cpu()
        # the split output sits in the middle planes of a wider tensor filled with a guard value: nothing outside its planes may change
        wide = hip.T.empty(n, h, w, cout + 16, dev, "sp3")
        hip.split3(hip.nchw_to_nhwc(torch.full((n, cout + 16, h, w), guard).to(dev)), out=wide)
        got = pc(xs, act=hip.ACT_LRELU, slope=0.1, res=ressp, res_first=first, chscale=ch, out=wide.channels(8, 8 + cout))
        assert got.dtype == "sp3"
        assert torch.equal(_unsplit(wide.channels(8, 8 + cout)), want), (res32 is not None, first, ch is not None)
        assert bool((_unsplit(wide.channels(0, 8)) == guard).all()) and bool((_unsplit(wide.channels(8 + cout, 16 + cout)) == guard).all())


def test_pooling_and_upsampling_on_split_tensors(dev):
    """vc_maxpool2_sp3 / vc_upsample_bilinear_sp3 (the mask U-Net between split-operand layers, LHBDC/model/layers.py:200-246): exactly the
    fp32 kernels followed by vc_split3, also into a window of planes of a wider split tensor."""
    from vcamd import hip
    g = torch.Generator().manual_seed(31)
    x = hip.nchw_to_nhwc(torch.randn(2, 64, 48, 80, generator=g).to(dev))
    xs = hip.split3(x)
    assert torch.equal(_unsplit(hip.maxpool2(xs)), hip.nhwc_to_nchw(hip.maxpool2(x)).cpu())
    wide = hip.T.empty(2, 96, 160, 96, dev, "sp3")
    wide.buf.zero_()
    hip.upsample_bilinear(x, 2, out=wide.channels(32, 96))
    assert torch.equal(_unsplit(wide.channels(32, 96)), hip.nhwc_to_nchw(hip.upsample_bilinear(x, 2)).cpu())
    assert float(_unsplit(wide.channels(0, 32)).abs().max()) == 0.0
    assert torch.equal(_unsplit(hip.avgpool2_split(xs, True)), hip.nhwc_to_nchw(hip.avgpool_reflectpad(x, 2)).cpu())
    assert torch.equal(hip.nhwc_to_nchw(hip.avgpool2_split(xs, False)), hip.nhwc_to_nchw(hip.avgpool_reflectpad(x, 2)))
    pooled = hip.maxpool2(wide.channels(32, 96))            # a window as the input
    assert torch.equal(_unsplit(pooled), hip.nhwc_to_nchw(hip.maxpool2(hip.upsample_bilinear(x, 2))).cpu())


def test_masknet_through_split_tensors_equals_the_conversion_path(dev):
    """Mask.run at a size where every layer takes the split pipeline: concat buffers as split tensors written by the producers == the
    same layers with fp32 concat buffers converted by vc_split3 (bit for bit: split tensors are exact)."""
    from vcamd import hip, lhbdc
    hip.set_fp32_mode("split")
    torch.manual_seed(5)
    m = lhbdc.Mask().to(dev).eval()
    x = hip.nchw_to_nhwc(torch.rand(1, 6, 544, 960, generator=torch.Generator().manual_seed(6)).to(dev))
    with torch.no_grad():
        a = hip.nhwc_to_nchw(m.run(x)).cpu()
        keep = hip.wants_split_at
        try:
            hip.wants_split_at = lambda pc, n, h, w: False if (pc is m._packed["deconv3"]) else keep(pc, n, h, w)   # forces the fp32-buffer path
            b = hip.nhwc_to_nchw(m.run(x)).cpu()
        finally:
            hip.wants_split_at = keep
        hip.set_fp32_mode("native")
        c = hip.nhwc_to_nchw(m.run(x)).cpu()
    print(f"mask net 544x960: split-streaming vs conversion path max|d| {float((a - b).abs().max()):.2e}; vs native {float((a - c).abs().max()):.2e}")
    assert float((a - c).abs().max()) < 1e-4


@pytest.mark.parametrize("cin,cout,h,w,n", [(128, 128, 96, 160, 2), (64, 64, 50, 75, 1), (128, 64, 33, 170, 1)])
def test_streaming_1x1_kernel_writes_split_tensors(dev, cin, cout, h, w, n):
    """VC_CFG_OUT_SP3 on the streaming 1x1 kernel (csrc/conv_pws.hip, its own instances with the store counts the counted waits need):
    plain / residual / residual-first epilogues -- the split tensor holds exactly the fp32 values the same configuration stores; three
    repetitions (a wrong wait count shows as run-to-run differences)."""
    from vcamd import hip
    hip.set_fp32_mode("split")
    g = torch.Generator().manual_seed(41)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    pc = hip.PackedConv(wt, torch.randn(cout, generator=g) * 0.1, device=dev)
    assert hip.CFG_PWS in pc.candidates
    x = hip.T.empty(n, h, w, cin, dev)
    x.buf.normal_()
    r = hip.T.empty(n, h, w, cout, dev)
    r.buf.normal_()
    for kw in (dict(act=hip.ACT_RELU), dict(act=hip.ACT_LRELU, slope=0.1, res=r), dict(act=hip.ACT_RELU, res=r, res_first=True)):
        pc.tuned = {}
        fl = hip.CFG_RES_FIRST if kw.get("res_first") else 0
        pc.tuned[(n, h, w, fl)] = hip.CFG_PWS | hip.CFG_EXACT | fl
        pc.tuned[(n, h, w, fl | hip.CFG_OUT_SP3)] = hip.CFG_PWS | hip.CFG_EXACT | fl | hip.CFG_OUT_SP3
        ref = hip.nhwc_to_nchw(pc(x, **kw)).cpu()
        for rep in range(3):
            o = pc(x, out_sp3=True, **kw)
            assert o.dtype == "sp3" and torch.equal(_unsplit(o), ref), (kw.keys(), rep)
