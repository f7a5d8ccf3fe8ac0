"""-m gpu: every HIP entry point against the CPU PyTorch operator the reference calls at that site.

The CPU side here is plain torch (fp32) on the same seeded inputs -- the operators `oracle/` is built
from.  Tolerances: convolutions 2e-5 * (1 + |ref|) (fp32 FMA chains in a different order), resampling /
warping 1e-5, entropy bit counts 1e-5 relative, integer outputs exact.
"""
import ctypes

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


@pytest.fixture(autouse=True, scope="module")
def _native_fp32_instances():
    """This module pins and compares the NATIVE fp32 / fp16 tile configurations; the split-operand pipeline (the default fp32 mode for
    the layers it serves) has its own module, tests/test_split_gpu.py."""
    from vcamd import hip
    keep = hip.fp32_mode()
    hip.set_fp32_mode("native")
    yield
    hip.set_fp32_mode(keep)


def _rand(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def _close(a, b, tol, what=""):
    a, b = a.double().cpu(), b.double().cpu()
    err = ((a - b).abs() / (1.0 + b.abs())).max().item()
    assert err <= tol, f"{what}: max rel-abs err {err:.3e} > {tol}"


CONV_CASES = [
    # cin, cout, k, stride, h, w, n, act, pixelshuffle
    (128, 128, 3, 1, 24, 40, 1, "lrelu", False),   # N128
    (128, 128, 3, 2, 34, 66, 1, "none", False),    # stride 2, odd tiles
    (3, 128, 3, 2, 64, 64, 2, "lrelu", False),     # tiny Cin, scalar staging path
    (4, 128, 3, 2, 64, 64, 1, "lrelu", False),
    (19, 128, 3, 2, 32, 64, 1, "lrelu", False),
    (128, 512, 3, 1, 17, 30, 1, "lrelu", True),    # subpel 128->512 + pixel shuffle
    (128, 192, 3, 1, 16, 32, 1, "lrelu", False),   # N64 x3
    (192, 768, 3, 1, 9, 33, 1, "lrelu", True),
    (192, 256, 3, 1, 20, 20, 1, "none", False),
    (128, 12, 3, 1, 16, 48, 1, "none", True),      # N16 + pixel shuffle to 3 channels
    (128, 16, 3, 1, 16, 48, 1, "none", True),
    (128, 128, 1, 2, 32, 64, 1, "none", False),    # skip projection
    (3, 128, 1, 2, 32, 64, 1, "none", False),
    (8, 32, 7, 1, 40, 72, 2, "relu", False),       # SPyNet
    (32, 64, 7, 1, 34, 60, 1, "relu", False),
    (64, 32, 7, 1, 34, 60, 1, "relu", False),
    (32, 16, 7, 1, 34, 60, 1, "relu", False),
    (16, 2, 7, 1, 34, 60, 4, "none", False),
    (6, 32, 5, 1, 32, 64, 1, "relu", False),       # mask net
    (32, 64, 5, 1, 32, 64, 1, "relu", False),
    (192, 64, 5, 1, 16, 32, 1, "relu", False),
    (96, 32, 5, 1, 16, 32, 1, "relu", False),
    (32, 1, 5, 1, 24, 40, 1, "sigmoid", False),
    (64, 128, 3, 1, 16, 32, 1, "relu", False),
    (256, 128, 3, 1, 16, 32, 1, "relu", False),
    (6, 32, 3, 1, 32, 32, 1, "lrelu", False),      # Flex U-Net
    (512, 512, 3, 1, 8, 16, 1, "lrelu", False),
    (32, 4, 3, 1, 32, 32, 1, "none", False),       # N4 (4x4x1 MFMA) heads
    (32, 2, 3, 1, 40, 100, 2, "sigmoid", False),
    (16, 2, 7, 1, 70, 130, 1, "none", False),
    (32, 1, 5, 1, 33, 65, 1, "relu", False),
]


@pytest.mark.parametrize("cin,cout,k,stride,h,w,n,act,ps", CONV_CASES)
def test_conv2d(dev, cin, cout, k, stride, h, w, n, act, ps):
    from vcamd import hip
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
    b = _rand((cout,), 3, 0.1)
    ref = F.conv2d(x, wt, b, stride=stride, padding=k // 2)
    if ps:
        ref = F.pixel_shuffle(ref, 2)
    slope = 0.01
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, slope),
           "sigmoid": torch.sigmoid}[act](ref)
    pc = hip.PackedConv(wt, b, stride=stride, pixelshuffle=ps, device=dev)
    code = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "lrelu": hip.ACT_LRELU, "sigmoid": hip.ACT_SIGMOID}[act]
    out = hip.nhwc_to_nchw(pc(hip.nchw_to_nhwc(x.to(dev)), act=code, slope=slope))
    assert out.shape == ref.shape
    _close(out, ref, 2e-5, f"conv {cin}->{cout} k{k} s{stride}")


def test_tile_configs_are_bit_identical(dev):
    """the autotuner may pick any 32-wide tile configuration: results must not depend on it"""
    from vcamd import hip
    x = hip.nchw_to_nhwc(_rand((1, 128, 40, 72), 31).to(dev))
    pc = hip.PackedConv(_rand((128, 128, 3, 3), 32, 0.03), _rand((128,), 33, 0.1), device=dev)
    outs = []
    for cfg in (0, 1, 2, 5):
        pc.tuned = {(x.n, x.h, x.w, 0): cfg | hip.CFG_EXACT}
        outs.append(hip.nhwc_to_nchw(pc(x, act=hip.ACT_LRELU)))
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_tile_configs_7x7_including_16_row_tiles(dev, precision):
    """VC_CFG_N32T16 (16-row tiles) on SPyNet-shaped 7x7 layers: heights that are not a multiple of 16, a batch,
    ReLU / residual epilogues -- bit-identical to the 8-row configurations on the same packed weights"""
    from vcamd import hip
    hip.set_conv_precision(precision)
    try:
        pcs = [hip.PackedConv(_rand((co, ci, 7, 7), 70 + co, 0.02), _rand((co,), 71, 0.1), device=dev) for ci, co in ((64, 32), (32, 64))]
    finally:
        hip.set_conv_precision("fp32")
    fl = hip.CFG_F16 if precision == "fp16" else 0
    for pc in pcs:
        assert 7 in pc.candidates
        x = hip.nchw_to_nhwc(_rand((2, pc.cin, 43, 70), 72).to(dev))
        res = hip.nchw_to_nhwc(_rand((2, pc.cout, 43, 70), 73).to(dev))
        outs = []
        for cfg in (pc.cfg, 2, 7):
            pc.tuned = {(x.n, x.h, x.w, fl): cfg | hip.CFG_EXACT | fl}
            outs.append((hip.nhwc_to_nchw(pc(x, act=hip.ACT_RELU)), hip.nhwc_to_nchw(pc(x, res=res))))
        for a, b in outs[1:]:
            assert torch.equal(outs[0][0], a) and torch.equal(outs[0][1], b)


F16_CASES = [(128, 128, 3, 1, 40, 72, "lrelu", False), (128, 512, 3, 1, 17, 30, "lrelu", True), (64, 32, 7, 1, 34, 60, "relu", False),
             (32, 64, 7, 1, 34, 60, "relu", False), (192, 64, 5, 1, 16, 32, "relu", False), (128, 128, 3, 2, 34, 66, "none", False),
             (256, 128, 3, 1, 16, 32, "relu", False), (128, 128, 1, 2, 32, 64, "none", False), (512, 512, 3, 1, 8, 16, "lrelu", False),
             (32, 16, 7, 1, 34, 60, "relu", False), (128, 12, 3, 1, 16, 48, "none", True), (96, 16, 5, 1, 20, 36, "relu", False)]


@pytest.mark.parametrize("cin,cout,k,stride,h,w,act,ps", F16_CASES)
def test_conv2d_fp16_path(dev, cin, cout, k, stride, h, w, act, ps):
    """fp16 MFMA conv path (BASELINE configs[4]): compared with the SAME convolution on half-rounded operands in
    fp32 (tight: only accumulation order differs) and with the exact fp32 result (loose: input rounding, 2^-11)."""
    from vcamd import hip
    x = _rand((1, cin, h, w), 41)
    wt = _rand((cout, cin, k, k), 42, 1.0 / np.sqrt(cin * k * k))
    b = _rand((cout,), 43, 0.1)
    f = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.01)}[act]

    def ref(xx, ww):
        r = F.conv2d(xx, ww, b, stride=stride, padding=k // 2)
        return f(F.pixel_shuffle(r, 2) if ps else r)
    exact, rounded = ref(x, wt), ref(x.half().float(), wt.half().float())
    hip.set_conv_precision("fp16")
    try:
        pc = hip.PackedConv(wt, b, stride=stride, pixelshuffle=ps, device=dev)
    finally:
        hip.set_conv_precision("fp32")
    assert pc.wpk16 is not None
    code = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "lrelu": hip.ACT_LRELU}[act]
    out = hip.nhwc_to_nchw(pc(hip.nchw_to_nhwc(x.to(dev)), act=code, slope=0.01))
    _close(out, rounded, 3e-5, "fp16 conv vs half-rounded operands")
    _close(out, exact, 5e-3, "fp16 conv vs exact fp32")


DMA_CASES = [  # cin, cout, k, n, h, w, classic tile configs to compare with
    (128, 128, 3, 1, 40, 72, (5, 1)), (128, 128, 3, 3, 17, 33, (5,)), (64, 32, 7, 2, 50, 70, (7, 2)), (32, 64, 7, 1, 34, 60, (1, 2)),
    (64, 128, 3, 1, 33, 31, (5,)), (256, 128, 3, 1, 16, 40, (1, 5)), (128, 512, 3, 1, 17, 30, (5,)), (192, 128, 3, 2, 20, 36, (5,)),
    (128, 128, 3, 2, 160, 288, (5,)),
    # more than 128 output channels that are not a multiple of 128: the last block of 128 is partly padding
    (128, 432, 3, 1, 20, 40, (5, 0)), (64, 208, 3, 2, 17, 33, (5, 0)),
    # 64 -> 64 (two N-tiles): 9 phases per tile, weight ring of 3
    (64, 64, 3, 2, 33, 50, (1, 2)), (64, 64, 3, 1, 100, 170, (1,)),
    # 96 input channels (27 phases, ring of 9); 96 output channels = one block of 128, a quarter of it padding
    (96, 384, 3, 1, 20, 40, (5, 0)), (96, 96, 3, 2, 33, 50, (5, 1, 2)), (96, 96, 3, 1, 100, 170, (5,)),
]


@pytest.mark.parametrize("cin,cout,k,n,h,w,cfgs", DMA_CASES)
def test_lds_dma_fp16_kernel_is_bit_identical(dev, cin, cout, k, n, h, w, cfgs):
    """VC_CFG_DMA (csrc/conv_dma.h): one persistent workgroup per CU, both operands streamed into LDS by global_load_lds
    behind counted vmcnt waits, the two waves of a SIMD half a phase apart.  Same accumulation order (32-channel chunk,
    tap, k-step) and the same epilogue arithmetic as the classic fp16 instances, so every result must be equal bit for
    bit: plain / residual / residual-first / sigmoid / channel-gain / pixel-shuffle epilogues, half and fp32 outputs,
    ragged sizes (partial tiles in both directions), several images, more tiles than workgroups, several channel blocks."""
    from vcamd import hip
    from vcamd.hip import T
    ps = cout == 512
    x = _rand((n, cin, h, w), 51)
    wt = _rand((cout, cin, k, k), 52, 1.0 / np.sqrt(cin * k * k))
    b = _rand((cout,), 53, 0.1)
    hip.set_conv_precision("fp16")
    try:
        pc = hip.PackedConv(wt, b, pixelshuffle=ps, device=dev)
    finally:
        hip.set_conv_precision("fp32")
    assert pc.wpk16 is not None and hip.CFG_DMA in pc.candidates
    xt = hip.nchw_to_nhwc(x.to(dev))
    ho, wo, co = pc.out_shape(h, w)
    res = hip.nchw_to_nhwc(_rand((n, co, ho, wo), 54).to(dev))
    gain = _rand((cout,), 55).abs().to(dev)
    fl = hip.CFG_F16
    # a half-precision copy of the input, produced the way the models produce one: by an fp16-path layer's epilogue
    ident = torch.zeros(cin, cin, 1, 1)
    ident[torch.arange(cin), torch.arange(cin), 0, 0] = 1.0
    hip.set_conv_precision("fp16")
    try:
        to_half = hip.PackedConv(ident, torch.zeros(cin), device=dev)
    finally:
        hip.set_conv_precision("fp32")
    xh = to_half(xt, out_f16=True)
    assert xh.dtype == "f16"
    seen = 0
    outs = {}
    for cfg in tuple(cfgs) + (hip.CFG_DMA,):
        if cfg not in pc.candidates and cfg != pc.cfg:
            continue
        fi = fl | hip.CFG_IN_F16
        pc.tuned = {(n, h, w, fi): cfg | hip.CFG_EXACT | fi, (n, h, w, fi | hip.CFG_OUT_F16): cfg | hip.CFG_EXACT | fi | hip.CFG_OUT_F16}
        for a in (hip.ACT_SIGMOID,):
            pc.tuned[(n, h, w, fi, a, hip.EPI_NONE)] = cfg | hip.CFG_EXACT | fi
        o = [pc(xh, act=hip.ACT_LRELU, slope=0.01, out_f16=True), pc(xh, act=hip.ACT_RELU, res=res),
             pc(xh, act=hip.ACT_RELU, res=res, res_first=True, out_f16=True), pc(xh, act=hip.ACT_SIGMOID)]
        assert o[0].dtype == "f16" and o[1].dtype == "f32"
        if not ps:
            o.append(pc(xh, act=hip.ACT_NONE, chscale=gain))
        outs[cfg] = [t.buf.clone() for t in o]
        seen += cfg != hip.CFG_DMA
    assert seen > 0 and hip.CFG_DMA in outs
    for cfg, ref in outs.items():
        for a, c in zip(ref, outs[hip.CFG_DMA]):
            assert torch.equal(a, c), (cfg, (a.float() - c.float()).abs().max().item())
    # an fp32 input is not offered by this kernel (the DMA cannot convert): the LIBRARY refuses the pinned configuration, and the
    # binding falls back to the layer's general configuration -- never silently wrong, never a failed layer
    pc.tuned = {(n, h, w, fl): hip.CFG_DMA | hip.CFG_EXACT | fl}
    y_fallback = pc(xt, act=hip.ACT_NONE).buf.clone()
    pc.tuned = {(n, h, w, fl): pc.cfg | hip.CFG_EXACT | fl}
    assert torch.equal(y_fallback, pc(xt, act=hip.ACT_NONE).buf)
    d = hip.ConvDesc()
    out_t = hip.T.empty(xt.n, *pc.out_shape(xt.h, xt.w), dev)
    d.inp, d.out = xt.view(), out_t.view()
    d.wpk, d.bias = pc.wpk16.data_ptr(), pc.bias.data_ptr()
    d.kh = d.kw = pc.k
    d.stride = 1
    d.out_mode = hip.OUT_PIXELSHUFFLE2 if pc.ps else hip.OUT_PLAIN
    d.cfg = hip.CFG_DMA | hip.CFG_EXACT | fl | (hip.CFG_PACK128 if pc.cfg == 0 else 0)
    assert hip.lib().vc_conv2d_nhwc(hip.stream(), ctypes.byref(d)) == -1
    if cout > 64 and cout % 128:
        # a partly padded last block reads weights / bias padded to 128s: a caller that does not state that packing
        # (VC_CFG_PACK128) is refused instead of reading past a narrower one
        fi = fl | hip.CFG_IN_F16
        d.inp = xh.view(True)
        d.cfg = hip.CFG_DMA | hip.CFG_EXACT | fi | hip.CFG_PACK128
        assert hip.lib().vc_conv2d_nhwc(hip.stream(), ctypes.byref(d)) == 0
        d.cfg = hip.CFG_DMA | hip.CFG_EXACT | fi
        assert hip.lib().vc_conv2d_nhwc(hip.stream(), ctypes.byref(d)) == -1


def test_conv_residual_and_channel_slices(dev):
    """residual add + reading/writing channel slices of wider buffers (concat-free U-Net plumbing)"""
    from vcamd import hip
    from vcamd.hip import T
    x = _rand((1, 96, 16, 32), 4)
    wt = _rand((32, 96, 5, 5), 5, 0.02)
    b = _rand((32,), 6, 0.1)
    res = _rand((1, 32, 16, 32), 7)
    ref = F.relu(F.conv2d(x, wt, b, padding=2)) + res
    wide_in = T.empty(1, 16, 32, 128, dev)
    wide_in.buf.fill_(123.0)
    hip.check(hip.lib().vc_nchw_to_nhwc(hip.stream(), x.to(dev).contiguous().data_ptr(), wide_in.channels(16, 112).view()), "x")
    wide_out = T.empty(1, 16, 32, 64, dev)
    wide_out.buf.fill_(-7.0)
    pc = hip.PackedConv(wt, b, device=dev)
    pc(wide_in.channels(16, 112), out=wide_out.channels(8, 40), act=hip.ACT_RELU, res=hip.nchw_to_nhwc(res.to(dev)))
    full = wide_out.to_nchw()
    _close(full[:, 8:40], ref, 2e-5, "sliced conv")
    assert (full[:, :8] == -7.0).all() and (full[:, 40:] == -7.0).all()


@pytest.mark.parametrize("inverse", [False, True])
def test_gdn(dev, inverse):
    from oracle.cai.layers import GDN as OGDN
    from vcamd import hip
    from vcamd.layers import GDN
    from vcamd.seeding import seeded_state_dict
    o = OGDN(128, inverse=inverse)
    sd = seeded_state_dict(o.state_dict(), 5)
    o.load_state_dict(sd)
    g = GDN(128, inverse=inverse)
    g.load_state_dict(sd)
    g = g.to(dev)
    x = _rand((2, 128, 20, 36), 8, 2.0)
    res = _rand((2, 128, 20, 36), 9)
    with torch.no_grad():
        ref = o(x) + res
    out = hip.nhwc_to_nchw(g.run(hip.nchw_to_nhwc(x.to(dev)), res=hip.nchw_to_nhwc(res.to(dev))))
    _close(out, ref, 2e-5, "gdn")


def test_layout_roundtrip(dev):
    from vcamd import hip
    x = _rand((2, 5, 7, 9), 10).to(dev)
    assert torch.equal(hip.nhwc_to_nchw(hip.nchw_to_nhwc(x)), x)


@pytest.mark.parametrize("k,scale,h,w,oh,ow", [(2, 1.0, 68, 120, 34, 60), (4, 0.5, 192, 256, 64, 64),
                                                (4, 1.0, 256, 192, 64, 64), (1, 1.0, 48, 80, 64, 128)])
def test_avgpool_reflectpad(dev, k, scale, h, w, oh, ow):
    from vcamd import hip
    x = _rand((2, 2, h, w), 11)
    ref = F.avg_pool2d(x * scale, k) if k > 1 else x * scale
    ref = F.pad(ref, (0, ow - ref.shape[3], 0, oh - ref.shape[2]), mode="reflect") if (oh, ow) != tuple(ref.shape[2:]) else ref
    out = hip.nhwc_to_nchw(hip.avgpool_reflectpad(hip.nchw_to_nhwc(x.to(dev)), k, scale, oh, ow))
    _close(out, ref, 1e-6, "avgpool+reflect")


def test_maxpool(dev):
    from vcamd import hip
    x = _rand((1, 32, 16, 24), 12)
    out = hip.nhwc_to_nchw(hip.maxpool2(hip.nchw_to_nhwc(x.to(dev))))
    assert torch.equal(out.cpu(), F.max_pool2d(x, 2, 2))


@pytest.mark.parametrize("factor,align,c,h,w", [(2, False, 128, 8, 12), (4, False, 2, 48, 64), (2, True, 2, 17, 30),
                                                (2, False, 512, 4, 4)])
def test_upsample(dev, factor, align, c, h, w):
    from vcamd import hip
    x = _rand((2, c, h, w), 13)
    ref = F.interpolate(x, scale_factor=factor, mode="bilinear", align_corners=align)
    out = hip.nhwc_to_nchw(hip.upsample_bilinear(hip.nchw_to_nhwc(x.to(dev)), factor, align))
    _close(out, ref, 1e-6, "upsample")


def _warp_w3(img, flow):
    """ICIP2024/src/model/m.py:262-282 restated (the third warp convention of SURVEY Appendix B.5)."""
    b, _, h, w = flow.shape
    xx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, -1, h, -1)
    yy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, -1, -1, w)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    return F.grid_sample(img, (torch.cat([xx, yy], 1) + fl).permute(0, 2, 3, 1), mode="bilinear",
                         padding_mode="border", align_corners=True)


@pytest.mark.parametrize("convention", [1, 2, 3])
def test_warp(dev, convention):
    from oracle.flex import warp_w2
    from oracle.lhbdc import warp_w1
    from vcamd import hip
    img = torch.rand(2, 3, 40, 56, generator=torch.Generator().manual_seed(14))
    flow = _rand((2, 2, 40, 56), 15, 4.0)
    flow[:, :, :4, :4] = 100.0     # far out of range: border clamp / zero padding
    flow[:, :, -4:, -4:] = -100.0
    ref = {1: warp_w1, 2: warp_w2, 3: _warp_w3}[convention](img, flow)
    out = hip.nhwc_to_nchw(hip.warp(convention, hip.nchw_to_nhwc(img.to(dev)), hip.nchw_to_nhwc(flow.to(dev))))
    _close(out, ref, 2e-5, f"warp W{convention}")


@pytest.mark.parametrize("convention", [1, 2, 3])
def test_warp_extreme_displacements(dev, convention):
    """random-weight flow predictors emit absurd displacements: 1e4 .. 1e30, inf -- must stay in bounds and
    agree with grid_sample (border clamp / zeros)"""
    from oracle.flex import warp_w2
    from oracle.lhbdc import warp_w1
    from vcamd import hip
    img = torch.rand(1, 3, 24, 40, generator=torch.Generator().manual_seed(24))
    flow = torch.zeros(1, 2, 24, 40)
    vals = torch.tensor([1e4, -1e4, 3e9, -3e9, 1e20, -1e20, 1e30, -1e30, float("inf"), -float("inf")])
    flow[0, 0, :10, 0] = vals
    flow[0, 1, :10, 1] = vals
    flow[0, :, :10, 2] = vals
    ref = {1: warp_w1, 2: warp_w2, 3: _warp_w3}[convention](img, flow)
    out = hip.nhwc_to_nchw(hip.warp(convention, hip.nchw_to_nhwc(img.to(dev)), hip.nchw_to_nhwc(flow.to(dev)))).cpu()
    finite = torch.isfinite(ref)
    assert torch.isfinite(out).all()
    assert ((out - ref).abs()[finite] < 2e-5).all()


@pytest.mark.parametrize("h,w", [(64, 96), (70, 100), (192, 256)])
def test_spynet(dev, h, w):
    """whole SPyNet (pre-process, pyramid, level inputs incl. odd-size replicate pad, 7x7 stacks)"""
    from oracle.lhbdc import SpyNet
    from vcamd.lhbdc import Network
    from vcamd.seeding import seeded_state_dict
    o = SpyNet()
    sd = seeded_state_dict(o.state_dict(), 21)
    o.load_state_dict(sd)
    net = Network()
    net.load_state_dict(sd)
    net = net.to(dev)
    g = torch.Generator().manual_seed(16)
    a = F.avg_pool2d(torch.rand(1, 3, h + 4, w + 4, generator=g), 5, 1)
    b = torch.roll(a, shifts=(1, 2), dims=(2, 3)) * 0.98 + 0.01
    with torch.no_grad():
        ref = o(a, b)
    out = net(a.to(dev), b.to(dev))
    _close(out, ref, 2e-4, "spynet")


def test_masknet(dev):
    from oracle.lhbdc import MaskNet
    from vcamd.lhbdc import Mask
    from vcamd.seeding import seeded_state_dict
    o = MaskNet()
    sd = seeded_state_dict(o.state_dict(), 22)
    o.load_state_dict(sd)
    m = Mask()
    m.load_state_dict(sd)
    m = m.to(dev)
    x = torch.rand(1, 6, 64, 96, generator=torch.Generator().manual_seed(17))
    with torch.no_grad():
        ref = o(x)
    _close(m(x.to(dev)), ref, 2e-5, "masknet")


def test_blend(dev):
    from vcamd import hip
    from vcamd.hip import T
    g = torch.Generator().manual_seed(18)
    fw, bw, cur = (torch.rand(1, 3, 16, 24, generator=g) for _ in range(3))
    m = torch.rand(1, 1, 16, 24, generator=g)
    pred_ref = m * fw + (1.0 - m) * bw
    fwbw = hip.nchw_to_nhwc(torch.cat([fw, bw], 1).to(dev))
    mt, ct = hip.nchw_to_nhwc(m.to(dev)), hip.nchw_to_nhwc(cur.to(dev))   # keep the buffers alive
    pred, resid = T.empty(1, 16, 24, 3, dev), T.empty(1, 16, 24, 3, dev)
    hip.check(hip.lib().vc_lhbdc_blend(hip.stream(), fwbw.view(), mt.view(), ct.view(), pred.view(), resid.view()), "blend")
    _close(hip.nhwc_to_nchw(pred), pred_ref, 1e-6, "blend pred")
    _close(hip.nhwc_to_nchw(resid), cur - pred_ref, 1e-6, "blend resid")


def _codec_pair(dev, io, seed):
    from oracle.lhbdc import HyperpriorCodec
    from vcamd.lhbdc import MVCompressor, ResidualCompressor
    from vcamd.seeding import seeded_state_dict
    o = HyperpriorCodec(io).eval()
    sd = seeded_state_dict(o.state_dict(), seed)
    o.load_state_dict(sd)
    c = (MVCompressor if io == 4 else ResidualCompressor)()
    c.load_state_dict(sd)
    return o, c.to(dev).eval()


@pytest.mark.parametrize("io", [3, 4])
def test_hyperprior_forward(dev, io):
    o, c = _codec_pair(dev, io, 30 + io)
    x = _rand((1, io, 128, 192), 19, 0.5)
    with torch.no_grad():
        ref = o(x)
    out = c(x.to(dev))
    _close(out["x_hat"], ref["x_hat"], 5e-4, "codec x_hat")
    for k in "yz":
        rb = (-torch.log2(ref["likelihoods"][k])).sum().item()
        assert abs(out["bits"][k].item() - rb) / rb < 2e-3, (k, out["bits"][k].item(), rb)


def test_entropy_kernels_exact_inputs(dev):
    """factorised + Gaussian likelihood kernels on IDENTICAL inputs (no conv noise): bits within 1e-5,
    quantised tensors / symbols / indexes exact."""
    from oracle.cai.entropy_models import EntropyBottleneck as OEB, GaussianConditional as OGC, get_scale_table
    from vcamd import hip
    from vcamd.hip import T
    from vcamd.layers import BitCounter, EntropyBottleneck
    from vcamd.seeding import seeded_state_dict
    o = OEB(128)
    sd = seeded_state_dict(o.state_dict(), 41)
    o.load_state_dict(sd)
    eb = EntropyBottleneck(128)
    eb.load_state_dict(sd)
    eb = eb.to(dev)
    z = _rand((2, 128, 9, 14), 20, 4.0)
    with torch.no_grad():
        z_ref, lik = o(z)
    bits = BitCounter(dev)
    zt = hip.nchw_to_nhwc(z.to(dev))
    z_hat = T.empty(zt.n, zt.h, zt.w, zt.c, dev)
    sym = torch.empty(z.numel(), dtype=torch.int32, device=dev)
    lik_d = torch.empty(z.shape, dtype=torch.float32, device=dev)
    hip.check(hip.lib().vc_eb_forward(hip.stream(), zt.view(), eb.device_params().data_ptr(), None, None, z_hat.view(),
                                      sym.data_ptr(), bits.next_row_ptr(), bits.slots, lik_d.data_ptr()), "eb")
    assert torch.equal(hip.nhwc_to_nchw(z_hat).cpu(), z_ref)
    med = o.quantiles[:, 0, 1].detach().view(1, -1, 1, 1)
    assert torch.equal(sym.cpu().view(z.shape), torch.round(z - med).int())
    rb = (-torch.log2(lik)).sum().item()
    assert abs(bits.totals()[0].item() - rb) / rb < 1e-5
    # the likelihood TENSOR the reference's callers read (m.py:73-91), element by element
    assert ((lik_d.cpu() - lik).abs() / lik).max().item() < 2e-5

    gc = OGC(None)
    gc.update_scale_table(get_scale_table(), force=True)
    y = _rand((1, 128, 12, 20), 21, 6.0)
    sc = torch.exp(_rand((1, 128, 12, 20), 22, 1.5)) * 0.3
    sc[0, 0, 0, :5] = torch.tensor([0.0, 0.11, -3.0, 256.0, 1e4])
    mu = _rand((1, 128, 12, 20), 23, 2.0)
    with torch.no_grad():
        y_ref, lik = gc(y, sc, means=mu)
        idx_ref = gc.build_indexes(sc)
    bits = BitCounter(dev)
    yt, st, mt = (hip.nchw_to_nhwc(t.to(dev)) for t in (y, sc, mu))
    y_hat = T.empty(yt.n, yt.h, yt.w, yt.c, dev)
    sym = torch.empty(y.numel(), dtype=torch.int32, device=dev)
    idx = torch.empty_like(sym)
    lik_d = torch.empty(y.shape, dtype=torch.float32, device=dev)
    table = gc.scale_table.to(dev)
    hip.check(hip.lib().vc_gc_forward(hip.stream(), yt.view(), st.view(), mt.view(), None, None, y_hat.view(),
                                      bits.next_row_ptr(), bits.slots, None, sym.data_ptr(), idx.data_ptr(),
                                      table.data_ptr(), table.numel(), lik_d.data_ptr()), "gc")
    assert torch.equal(hip.nhwc_to_nchw(y_hat).cpu(), y_ref)
    assert torch.equal(sym.cpu().view(y.shape), torch.round(y - mu).int())
    assert torch.equal(idx.cpu().view(y.shape), idx_ref)
    rb = (-torch.log2(lik)).sum().item()
    assert abs(bits.totals()[0].item() - rb) / rb < 1e-5
    # erfc differences of nearly equal values: absolute 1e-7 where the likelihood is tiny, relative 1e-4 elsewhere
    dl = (lik_d.cpu() - lik).abs()
    assert (dl <= 1e-7 + 1e-4 * lik).all(), float((dl / lik).max())


PW_CASES = [(128, 128, 37, 75, 2), (64, 64, 33, 64, 1), (96, 96, 20, 50, 1), (32, 32, 16, 96, 3), (128, 64, 18, 40, 1),
            (96, 64, 9, 31, 1), (128, 96, 12, 33, 1), (64, 32, 8, 129, 1),
            # VC_CFG_PWS: a row shorter than one tile; more tiles than the 2048 persistent waves (ring wrap-around, tile hand-over)
            (64, 64, 5, 17, 1), (32, 32, 200, 330, 2), (128, 128, 300, 250, 1)]


@pytest.mark.parametrize("cin,cout,h,w,n", PW_CASES)
@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_pointwise_streaming_kernel_is_bit_identical(dev, cin, cout, h, w, n, precision):
    """VC_CFG_PW (streaming 1x1 kernel, csrc/conv_pw.hip) and VC_CFG_PWS (its LDS-DMA successor, csrc/conv_pws.hip) against the
    general kernel on the same packed weights:
    plain and ReLU epilogues, channel gain + residual, inputs/outputs that are channel slices of wider buffers,
    widths that are not a multiple of the 32-pixel tile, batches; on the fp16 path also half-precision in/out."""
    from vcamd import hip
    hip.set_conv_precision(precision)
    try:
        pc = hip.PackedConv(_rand((cout, cin, 1, 1), 51, 1.0 / np.sqrt(cin)), _rand((cout,), 52, 0.1), device=dev)
    finally:
        hip.set_conv_precision("fp32")
    assert 6 in pc.candidates
    fl = hip.CFG_F16 if precision == "fp16" else 0
    wide_in = hip.nchw_to_nhwc(_rand((n, cin + 32, h, w), 53).to(dev))
    x = wide_in.channels(16, 16 + cin)                      # 64-byte aligned slice of a wider buffer
    res = hip.nchw_to_nhwc(_rand((n, cout, h, w), 54).to(dev))
    gain = (_rand((cout,), 55).abs() + 0.5).to(dev)

    def run(cfg, **kw):
        pc.tuned = {}
        outs = []
        for flags in (fl, fl | hip.CFG_IN_F16, fl | hip.CFG_OUT_F16, fl | hip.CFG_IN_F16 | hip.CFG_OUT_F16):
            pc.tuned[(n, h, w, flags)] = cfg | hip.CFG_EXACT | flags
        wide_out = hip.T.empty(n, h, w, cout + 8, dev)
        wide_out.buf.zero_()
        a = pc(x, act=hip.ACT_RELU)
        b = pc(x, out=wide_out.channels(4, 4 + cout), res=res, chscale=gain)
        outs += [hip.nhwc_to_nchw(a), hip.nhwc_to_nchw(wide_out)]
        if precision == "fp16":                               # half intermediate: producer and consumer both on cfg
            t = pc(x, act=hip.ACT_LRELU, slope=0.1, out_f16=True)
            assert t.dtype == "f16"
            if cin == cout:
                outs.append(hip.nhwc_to_nchw(pc(t, res=res)))
            outs.append(t.to_nchw())
        return outs

    base = run(pc.cfg if pc.cfg <= 2 else 2)
    pw = run(6)
    for a, b in zip(base, pw):
        assert torch.equal(a, b)
    if cin % 32 == 0 and cout % 32 == 0:                      # VC_CFG_PWS (csrc/conv_pws.hip): LDS-DMA rings, counted waits
        assert 9 in pc.candidates
        for rep in range(3):                                  # (a wrong wait count shows as run-to-run differences)
            for a, b in zip(base, run(9)):
                assert torch.equal(a, b)
    # and against torch on the exact path
    ref = F.relu(F.conv2d(hip.nhwc_to_nchw(x).cpu(), _rand((cout, cin, 1, 1), 51, 1.0 / np.sqrt(cin)), _rand((cout,), 52, 0.1)))
    _close(pw[0], ref, 2e-5 if precision == "fp32" else 5e-3, "pointwise kernel vs torch")


@pytest.mark.parametrize("cin,cout,h,w,n", [(128, 128, 37, 75, 2), (64, 64, 20, 33, 1), (96, 96, 9, 50, 1), (32, 32, 200, 330, 1), (128, 64, 18, 40, 1)])
def test_streaming_kernel_half_precision_residual(dev, cin, cout, h, w, n):
    """VC_CFG_RES_F16 (fp16 path, VC_CFG_PWS only): a half-precision residual gives exactly what the fp32 residual holding the
    same (half-representable) values gives -- fp32 and half output, residual before and after the activation, half and fp32
    input; any other configuration declines it."""
    from vcamd import hip
    hip.set_conv_precision("fp16")
    try:
        pc = hip.PackedConv(_rand((cout, cin, 1, 1), 61, 1.0 / np.sqrt(cin)), _rand((cout,), 62, 0.1), device=dev)
    finally:
        hip.set_conv_precision("fp32")
    assert pc.half_res_ok
    x32 = hip.nchw_to_nhwc(_rand((n, cin, h, w), 63).to(dev))
    x16 = hip.T.empty(n, h, w, cin, dev, "f16")
    x16.buf.copy_(x32.buf.half())
    r16 = hip.T.empty(n, h, w, cout, dev, "f16")
    r16.buf.copy_(hip.nchw_to_nhwc(_rand((n, cout, h, w), 64).to(dev)).buf.half())
    r32 = hip.T.empty(n, h, w, cout, dev)
    r32.buf.copy_(r16.buf.float())
    for x in (x32, x16):
        for out_f16 in (False, True):
            for res_first in (False, True):
                outs = []
                for res in (r32, r16):
                    for rep in range(2 if res is r16 else 1):
                        pc.tuned = {}
                        fl = hip.CFG_F16 | (hip.CFG_IN_F16 if x is x16 else 0) | (hip.CFG_OUT_F16 if out_f16 else 0)
                        pc.tuned[(n, h, w, fl)] = 9 | hip.CFG_EXACT | fl               # the fp32-residual twin on the same kernel
                        o = pc(x, act=hip.ACT_LRELU, slope=0.2, res=res, res_first=res_first, out_f16=out_f16)
                        assert o.dtype == ("f16" if out_f16 else "f32")
                        outs.append(o.buf.clone())
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    d = hip.ConvDesc()        # and the C ABI refuses the flag on any other configuration
    o = hip.T.empty(n, h, w, cout, dev)
    d.inp, d.out = x32.view(True), o.view(True)
    d.wpk, d.bias = pc.wpk16.data_ptr(), pc.bias.data_ptr()
    d.res, d.res_sn, d.res_sh, d.res_sw = r16.ptr, r16.sn, r16.sh, r16.sw
    d.kh = d.kw = 1
    d.stride = 1
    for cfg in (pc.cfg, 6):
        d.cfg = cfg | hip.CFG_EXACT | hip.CFG_F16 | hip.CFG_RES_F16
        assert hip.lib().vc_conv2d_nhwc(hip.stream(), ctypes.byref(d)) == -1     # VC_EINVAL


def test_pointwise_chain_padding_does_not_change_a_bit(dev):
    """run_sequential pads the odd intermediate channel counts of a chain of 1x1 convolutions (ICIP2024's entropy-parameter
    networks, 768 -> 426 -> 341 -> 2c) to multiples of 16 with zero weights: aligned rows for the consumer, and -- adding
    zeros -- the very same fp32 results as the unpadded chain."""
    import torch.nn as nn
    from vcamd import hip
    from vcamd.layers import run_sequential
    torch.manual_seed(5)
    seq = nn.Sequential(nn.Conv2d(96, 42, 1), nn.LeakyReLU(inplace=True), nn.Conv2d(42, 37, 1), nn.LeakyReLU(inplace=True),
                        nn.Conv2d(37, 12, 1)).to(dev)
    x = hip.nchw_to_nhwc(_rand((2, 96, 19, 45), 71).to(dev))
    padded_cache, plain_cache = {}, {"pads": {}}
    a = run_sequential(seq, x, padded_cache)
    b = run_sequential(seq, x, plain_cache)
    assert sorted(padded_cache["pads"].values()) == [(0, 6), (6, 11), (11, 0)]
    assert [p.cout for p in padded_cache["seq"].values()] == [48, 48, 12] and [p.cout for p in plain_cache["seq"].values()] == [42, 37, 12]
    assert a.c == 12 and torch.equal(hip.nhwc_to_nchw(a), hip.nhwc_to_nchw(b))
    ref = seq(hip.nhwc_to_nchw(x))
    _close(hip.nhwc_to_nchw(a).cpu(), ref.cpu(), 2e-5, "padded 1x1 chain vs torch")


@pytest.mark.parametrize("convention", [1, 2, 3])
def test_warp_feature_maps_vectorised(dev, convention):
    """multi-channel feature maps (ICIP2024 warps 64/96/128-channel pyramids) take the 16-byte path; it must give
    the scalar path's values (checked against grid_sample) also for views that are channel slices"""
    from oracle.flex import warp_w2
    from oracle.lhbdc import warp_w1
    from vcamd import hip
    img = torch.randn(2, 24, 30, 44, generator=torch.Generator().manual_seed(61))
    flow = _rand((2, 2, 30, 44), 62, 5.0)
    flow[:, :, :3, :3] = 80.0
    ref = {1: warp_w1, 2: warp_w2, 3: _warp_w3}[convention](img, flow)
    wide = hip.nchw_to_nhwc(torch.cat([torch.zeros(2, 8, 30, 44), img], 1).to(dev))
    dst = hip.T.empty(2, 30, 44, 40, dev)
    hip.warp(convention, wide.channels(8, 32), hip.nchw_to_nhwc(flow.to(dev)), out=dst.channels(12, 36))
    out = hip.nhwc_to_nchw(dst.channels(12, 36))
    _close(out, ref, 2e-5, f"vectorised warp W{convention}")


@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_on_streaming_kernel_is_bit_identical(dev, inverse):
    """GDN / IGDN (128 channels) through VC_CFG_PW: the x^2 contraction reads x once and multiplies the very registers
    it squared -- must equal the general kernel bit for bit (with and without the residual add) and the oracle GDN."""
    from oracle.cai.layers import GDN as OracleGDN
    from vcamd import hip
    from vcamd.layers import GDN
    from vcamd.seeding import seeded_state_dict
    ora = OracleGDN(128, inverse=inverse)
    sd = seeded_state_dict(ora.state_dict(), seed=77)
    ora.load_state_dict(sd)
    g = GDN(128, inverse=inverse)
    g.load_state_dict(sd)
    g = g.to(dev)
    x = _rand((2, 128, 37, 50), 78, 2.0)
    res = _rand((2, 128, 37, 50), 79)
    xt, rt = hip.nchw_to_nhwc(x.to(dev)), hip.nchw_to_nhwc(res.to(dev))
    g.run(xt)                                            # packs
    pc = g._packed
    assert 6 in pc.candidates
    outs = []
    for cfg in (pc.cfg, 6):
        pc.tuned = {(2, 37, 50, 0, hip.ACT_NONE, hip.EPI_IGDN if inverse else hip.EPI_GDN): cfg | hip.CFG_EXACT}
        outs.append((hip.nhwc_to_nchw(g.run(xt)), hip.nhwc_to_nchw(g.run(xt, res=rt))))
        assert len(pc.tuned) == 1                       # the forced entry was the one used
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    with torch.no_grad():
        ref = ora(x)
    _close(outs[1][0], ref, 3e-5, "GDN on the streaming kernel vs oracle")
    _close(outs[1][1], ref + res, 3e-5, "GDN + residual")


def test_survey_named_entry_points_forward_to_the_same_kernels(dev):
    """vc_gdn / vc_pad / vc_pool ... (the operator names of SURVEY.md 8(b)) give what the library's own entry points give"""
    from vcamd import hip
    from vcamd.layers import GDN
    from vcamd.seeding import seeded_state_dict
    L = hip.lib()
    g = GDN(128)
    g.load_state_dict(seeded_state_dict(g.state_dict(), seed=5))
    g = g.to(dev)
    x = hip.nchw_to_nhwc(_rand((1, 128, 20, 33), 91, 2.0).to(dev))
    res = hip.nchw_to_nhwc(_rand((1, 128, 20, 33), 92).to(dev))
    ref = g.run(x, res=res)
    pc = g._packed
    assert pc.cfg == L.vc_conv_select_cfg(128, 128, 1, 1)
    out = hip.T.empty(1, 20, 33, 128, dev)
    hip.check(L.vc_gdn(hip.stream(), x.view(), pc.wpk.data_ptr(), pc.bias.data_ptr(), 0, res.view(), out.view()), "vc_gdn")
    assert torch.equal(hip.nhwc_to_nchw(out), hip.nhwc_to_nchw(ref))
    img = _rand((1, 3, 30, 45), 93)
    t = hip.nchw_to_nhwc(img.to(dev))
    padded = hip.T.empty(1, 48, 64, 3, dev)
    hip.check(L.vc_pad(hip.stream(), t.view(), padded.view()), "vc_pad")
    ref_pad = torch.nn.ReflectionPad2d((0, 64 - 45, 0, 48 - 30))(img)
    assert torch.equal(hip.nhwc_to_nchw(padded).cpu(), ref_pad)


def test_psnr_uint8_kernel_matches_the_reference_formula(dev):
    """vc_psnr_uint8 == utils.py:32-51 / testing.py:176-182: clamp, x255, round half to even, MSE of the crop in double."""
    from vcamd import hip
    g = torch.Generator().manual_seed(61)
    a = (torch.rand(2, 3, 128, 192, generator=g) * 1.2 - 0.1)
    b = (a + 0.05 * torch.randn(2, 3, 128, 192, generator=g))
    a[0, 0, 0, :4] = torch.tensor([0.5 / 255.0, 1.5 / 255.0, 2.5 / 255.0, 254.5 / 255.0])      # exact ties -> half to even
    for h, w in ((128, 192), (120, 180), (1, 1)):
        qa = torch.round(torch.clamp(a[0, :, :h, :w], 0, 1) * 255.0).double()
        qb = torch.round(torch.clamp(b[0, :, :h, :w], 0, 1) * 255.0).double()
        mse = torch.mean((qa - qb) ** 2)
        ref = float(10.0 * torch.log10(255.0 ** 2 / mse)) if mse > 0 else float("inf")
        got = float(hip.psnr_uint8(a.to(dev), b.to(dev), h, w))
        assert got == pytest.approx(ref, rel=1e-12, abs=1e-12) or (ref == float("inf") and got == float("inf")), (h, w, got, ref)
    out = torch.zeros(3, dtype=torch.float64, device=dev)
    hip.psnr_uint8(a.to(dev), b.to(dev), 128, 192, out=out[1])                   # writes into a caller-provided slot
    assert float(out[0]) == 0.0 and float(out[2]) == 0.0 and float(out[1]) > 0.0
    with pytest.raises(hip.VcError):
        hip.psnr_uint8(a, b, 8, 8)                                               # CPU tensors: no fallback


def _sweep_cases(seed, count):
    """Seeded random convolution problems: ragged heights / widths (tile and band edges), batches, every kernel size and
    stride the dispatchers instantiate, channel counts on and off the packing granules, residual / activation epilogues."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        k = int(rng.choice([1, 3, 3, 5, 7]))
        stride = int(rng.choice([1, 2])) if k in (1, 3, 5) else 1
        cin = int(rng.choice([3, 6, 8, 16, 32, 64, 96, 128, 192]))
        cout = int(rng.choice([1, 2, 4, 12, 16, 32, 64, 96, 128, 192]))
        h, w = int(rng.integers(5, 75)), int(rng.integers(5, 140))
        n = int(rng.choice([1, 1, 2, 3]))
        act = str(rng.choice(["none", "relu", "lrelu"]))
        cases.append((cin, cout, k, stride, h, w, n, act, bool(rng.integers(0, 2))))
    return cases


def test_conv2d_random_sweep_every_candidate_configuration(dev):
    """Every tile configuration the tuner may pick, on random problems: equal to F.conv2d within the fp32 bar and
    BIT-identical to each other (the 2-D banded tile order, the 16-row tiles, the streaming 1x1 kernel, ragged last tiles)."""
    from vcamd import hip
    slope = 0.01
    for i, (cin, cout, k, stride, h, w, n, act, with_res) in enumerate(_sweep_cases(2024, 36)):
        x = _rand((n, cin, h, w), 100 + i)
        wt = _rand((cout, cin, k, k), 200 + i, 1.0 / np.sqrt(cin * k * k))
        b = _rand((cout,), 300 + i, 0.1)
        ref = F.conv2d(x, wt, b, stride=stride, padding=k // 2)
        ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, slope)}[act](ref)
        r = _rand(tuple(ref.shape), 400 + i) if with_res else None
        if r is not None:
            ref = ref + r
        pc = hip.PackedConv(wt, b, stride=stride, device=dev)
        code = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "lrelu": hip.ACT_LRELU}[act]
        xt = hip.nchw_to_nhwc(x.to(dev))
        rt = None if r is None else hip.nchw_to_nhwc(r.to(dev))
        outs = []
        for cfg in (pc.candidates or [pc.cfg]):
            pc.tuned = {(xt.n, xt.h, xt.w, 0): cfg | hip.CFG_EXACT}
            try:
                outs.append((cfg, hip.nhwc_to_nchw(pc(xt, act=code, slope=slope, res=rt))))
            except hip.VcError:
                # only the streaming 1x1 kernel and the LDS-DMA pipeline (half-precision inputs only) may decline a call
                assert cfg in (6, 8, 9)
        assert outs, f"no configuration ran for case {i}"
        what = f"sweep {i}: {cin}->{cout} k{k} s{stride} @{n}x{h}x{w} {act}{' +res' if with_res else ''}"
        _close(outs[0][1], ref, 2e-5, what)
        for cfg, o in outs[1:]:
            assert torch.equal(outs[0][1], o), f"{what}: configuration {cfg} differs from {outs[0][0]}"


def test_conv2d_half_precision_output_equals_rounded_fp32_output(dev):
    """fp16 path, VC_CFG_OUT_F16 on random problems (ragged sizes, residuals, every candidate configuration): the stored
    halves are exactly the fp32-output run rounded to half -- the paired-N-tile coalesced epilogue, the 32-channel one and
    the plain one must agree bit for bit."""
    from vcamd import hip
    hip.set_conv_precision("fp16")
    try:
        rng = np.random.default_rng(7)
        for i in range(16):
            k = int(rng.choice([1, 3, 3, 7]))
            cin = int(rng.choice([32, 64, 128]))
            cout = int(rng.choice([32, 64, 128, 192])) if k != 7 else int(rng.choice([32, 64]))
            h, w, n = int(rng.integers(6, 60)), int(rng.integers(6, 110)), int(rng.choice([1, 2]))
            pc = hip.PackedConv(_rand((cout, cin, k, k), 500 + i, 1.0 / np.sqrt(cin * k * k)), _rand((cout,), 600 + i, 0.1), device=dev)
            assert pc.wpk16 is not None
            x = hip.nchw_to_nhwc(_rand((n, cin, h, w), 700 + i).to(dev))
            res = hip.nchw_to_nhwc(_rand((n, cout, h, w), 800 + i).to(dev)) if i % 2 else None
            for cfg in (pc.candidates or [pc.cfg]):
                for fl_out in (0, hip.CFG_OUT_F16):
                    pc.tuned = {(x.n, x.h, x.w, hip.CFG_F16 | fl_out): cfg | hip.CFG_EXACT | hip.CFG_F16 | fl_out}
                try:
                    full = pc(x, act=hip.ACT_LRELU, res=res)
                    half = hip.T.empty(n, h, w, cout, dev, "f16")
                    pc(x, out=half, act=hip.ACT_LRELU, res=res)
                except hip.VcError:
                    assert cfg in (6, 8, 9)     # (the LDS-DMA pipeline takes half-precision inputs only: fp32 here)
                    continue
                torch.cuda.synchronize()
                assert torch.equal(full.buf.view(n, h, w, cout).to(torch.float16), half.buf.view(n, h, w, cout)), \
                    f"case {i}: {cin}->{cout} k{k} @{n}x{h}x{w} cfg {cfg}"
    finally:
        hip.set_conv_precision("fp32")


@pytest.mark.parametrize("cin,cout,k,n,h,w", [(16, 2, 7, 4, 70, 130), (32, 1, 5, 1, 33, 65), (32, 4, 3, 2, 40, 100), (8, 3, 7, 1, 20, 77)])
def test_four_channel_configuration_lds_weights_are_bit_identical(dev, cin, cout, k, n, h, w):
    """VC_CFG_N4 keeps the layer's weights in LDS when they fit beside the tile image; VC_N4_GLOBAL_WEIGHTS=1 forces the
    variant that streams them from global memory.  Same MFMA sequence, same operands: the outputs must be equal."""
    import os
    from vcamd import hip
    pc = hip.PackedConv(_rand((cout, cin, k, k), 900 + k, 1.0 / np.sqrt(cin * k * k)), _rand((cout,), 901, 0.1), device=dev)
    assert pc.cfg == 4
    x = hip.nchw_to_nhwc(_rand((n, cin, h, w), 902).to(dev))
    a = hip.nhwc_to_nchw(pc(x, act=hip.ACT_RELU))
    os.environ["VC_N4_GLOBAL_WEIGHTS"] = "1"
    try:
        b = hip.nhwc_to_nchw(pc(x, act=hip.ACT_RELU))
        torch.cuda.synchronize()
    finally:
        del os.environ["VC_N4_GLOBAL_WEIGHTS"]
    assert torch.equal(a, b)
    ref = F.relu(F.conv2d(_rand((n, cin, h, w), 902), _rand((cout, cin, k, k), 900 + k, 1.0 / np.sqrt(cin * k * k)), _rand((cout,), 901, 0.1),
                          padding=k // 2))
    _close(a, ref, 2e-5, f"N4 {cin}->{cout} k{k}")


def test_conv2d_on_tensors_beyond_2_31_elements(dev):
    """9 x 2176 x 3840 x 32 = 2.4e9 elements per tensor (a level-batched 2160p pass): offsets must be 64-bit everywhere.
    The last image (the one whose elements lie beyond 2^31) is checked against F.conv2d on crops."""
    from vcamd import hip
    n, h, w, c = 9, 2176, 3840, 32
    assert n * h * w * c > 2 ** 31
    wt = _rand((c, c, 3, 3), 950, 1.0 / np.sqrt(c * 9))
    b = _rand((c,), 951, 0.1)
    pc = hip.PackedConv(wt, b, device=dev)
    x = hip.T.empty(n, h, w, c, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    x.buf.copy_(torch.randn(x.buf.numel() // 8, generator=g, device=dev).repeat(8))      # (8 equal slabs: cheap to generate)
    out = pc(x, act=hip.ACT_RELU)
    torch.cuda.synchronize()
    xv, ov = x.buf.view(n, h, w, c), out.buf.view(n, h, w, c)
    for img, y0, x0 in ((n - 1, h - 40, w - 70), (n - 1, 0, 0), (5, 1000, 2000), (0, 17, 33)):
        crop = xv[img, max(y0 - 1, 0):y0 + 33, max(x0 - 1, 0):x0 + 65].permute(2, 0, 1)[None].cpu()
        ref = F.relu(F.conv2d(crop, wt, b, padding=1))
        oy, ox = (1 if y0 > 0 else 0), (1 if x0 > 0 else 0)          # rows / columns whose halo lies inside the crop
        got = ov[img, y0:y0 + 30, x0:x0 + 60].permute(2, 0, 1)[None].cpu()
        _close(got, ref[:, :, oy:oy + 30, ox:ox + 60], 2e-5, f"image {img} at ({y0},{x0})")
    del x, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("c,n,h,w", [(128, 1, 40, 72), (128, 3, 17, 33), (128, 2, 160, 288), (128, 1, 16, 32), (64, 2, 33, 50), (64, 1, 100, 170)])
def test_fused_bottleneck_tail_equals_the_two_layers(dev, c, n, h, w):
    """vc_conv_desc.tail_wpk (fp16 path, VC_CFG_DMA): conv3x3 C -> C + ReLU followed by conv1x1 C -> C + identity (C = 128, 64)
    (the tail of ICIP2024/src/model/elic.py:69-83) in ONE launch against the two launches it replaces.  The 3x3 result is
    rounded to half exactly as the unfused layer stores it and the 1x1 contraction sums the same C products per output
    (k order inside an MFMA permuted): bit-identical or within one half-precision ulp of the identity-carrying output.
    Half and fp32 residual, half and fp32 output, ragged sizes, several images, more tiles than workgroups."""
    from vcamd import hip
    hip.set_conv_precision("fp16")
    try:
        c2 = hip.PackedConv(_rand((c, c, 3, 3), 71, 1.0 / np.sqrt(c * 9)), _rand((c,), 72, 0.1), device=dev)
        c3 = hip.PackedConv(_rand((c, c, 1, 1), 73, 1.0 / np.sqrt(c)), _rand((c,), 74, 0.1), device=dev)
    finally:
        hip.set_conv_precision("fp32")
    assert c2.can_fuse_tail(c3)
    t16 = hip.T.empty(n, h, w, c, dev, "f16")
    t16.buf.copy_(hip.nchw_to_nhwc(_rand((n, c, h, w), 75).to(dev)).buf.half())
    r16 = hip.T.empty(n, h, w, c, dev, "f16")
    r16.buf.copy_(hip.nchw_to_nhwc(_rand((n, c, h, w), 76).to(dev)).buf.half())
    r32 = hip.T.empty(n, h, w, c, dev)
    r32.buf.copy_(r16.buf.float())
    for res in (r16, r32, None):
        for out_f16 in (True, False):
            mid = c2(t16, act=hip.ACT_RELU, out_f16=True)
            assert mid.dtype == "f16"
            two = c3(mid, res=res, out_f16=out_f16)
            one = c2(t16, act=hip.ACT_RELU, tail=c3, res=res, out_f16=out_f16)
            again = c2(t16, act=hip.ACT_RELU, tail=c3, res=res, out_f16=out_f16)
            assert one.dtype == two.dtype == ("f16" if out_f16 else "f32")
            assert torch.equal(one.buf, again.buf)                       # deterministic
            a, b = one.buf.float(), two.buf.float()
            err = ((a - b).abs() / (1.0 + b.abs())).max().item()
            same = torch.equal(one.buf, two.buf)
            print(f"fused tail C={c} {n}x{h}x{w} res={'none' if res is None else res.dtype} out={'f16' if out_f16 else 'f32'}: "
                  f"{'bit-identical' if same else f'max rel-abs err {err:.2e}'}")
            assert err < (2e-3 if out_f16 else 2e-5), err
    # refused where it has no instance: fp32 input, other channel counts
    x32 = hip.T.empty(n, h, w, c, dev)
    x32.buf.copy_(t16.buf.float())
    with pytest.raises(hip.VcError):
        c2(x32, act=hip.ACT_RELU, tail=c3)


@pytest.mark.parametrize("cin,cout,k,n,h,w", [(64, 32, 7, 2, 50, 70), (32, 64, 7, 1, 34, 60), (64, 32, 7, 4, 160, 288), (32, 64, 7, 3, 17, 33),
                                              (128, 128, 3, 1, 40, 72), (128, 128, 3, 3, 17, 33), (128, 512, 3, 1, 34, 60), (64, 64, 3, 2, 33, 50),
                                              (64, 128, 3, 1, 50, 70), (128, 64, 3, 1, 20, 36), (256, 128, 3, 1, 16, 40), (128, 128, 3, 2, 160, 288),
                                              (96, 32, 5, 2, 33, 50), (192, 64, 5, 1, 40, 72), (32, 64, 5, 1, 100, 170),
                                              (32, 64, 3, 1, 100, 170)])
def test_lds_dma_fp32_kernel_is_bit_identical(dev, cin, cout, k, n, h, w):
    """The EXACT fp32 instances of the LDS-DMA pipeline (csrc/conv_dma.h, DmaCfg::F32: fp32 tensors, v_mfma_f32_32x32x2_f32,
    16-channel chunks): SPyNet's two big 7x7 layers (LHBDC/model/flow.py:52-62) and the 3x3 stride-1 layers of the residual blocks
    and U-Nets (pixel shuffle included).  Same accumulation order as the classic fp32
    instances (16-channel chunk, tap, k-step, sub-step) and the same epilogue arithmetic: every result bit for bit equal --
    ReLU / plain / residual / sigmoid epilogues, ragged sizes, several images, more tiles than workgroups."""
    from vcamd import hip
    x = _rand((n, cin, h, w), 81)
    ps = cout == 512
    pc = hip.PackedConv(_rand((cout, cin, k, k), 82, 1.0 / np.sqrt(cin * k * k)), _rand((cout,), 83, 0.1), pixelshuffle=ps, device=dev)
    assert pc.dma_f32 and hip.CFG_DMA in pc.candidates and pc.wpk16 is None
    xt = hip.nchw_to_nhwc(x.to(dev))
    ho, wo, co = pc.out_shape(h, w)
    res = hip.nchw_to_nhwc(_rand((n, co, ho, wo), 84).to(dev))
    outs = {}
    for cfg in [c for c in pc.candidates]:
        pc.tuned = {(n, h, w, 0): cfg | hip.CFG_EXACT, (n, h, w, 0, hip.ACT_SIGMOID, hip.EPI_NONE): cfg | hip.CFG_EXACT}
        o = [pc(xt, act=hip.ACT_RELU), pc(xt, act=hip.ACT_NONE, res=res), pc(xt, act=hip.ACT_LRELU, slope=0.1, res=res),
             pc(xt, act=hip.ACT_SIGMOID)]
        outs[cfg] = [t.buf.clone() for t in o]
    assert hip.CFG_DMA in outs and len(outs) >= 2
    for cfg, ref in outs.items():
        for a, c in zip(ref, outs[hip.CFG_DMA]):
            assert torch.equal(a, c), (cfg, (a - c).abs().max().item())
    # and against the CPU's fp32 convolution (summation order differs: tolerance)
    exact = torch.relu(torch.nn.functional.conv2d(x, _rand((cout, cin, k, k), 82, 1.0 / np.sqrt(cin * k * k)), _rand((cout,), 83, 0.1), padding=k // 2))
    if ps:
        exact = torch.nn.functional.pixel_shuffle(exact, 2)
    _close(hip.nhwc_to_nchw(_as_t(outs[hip.CFG_DMA][0], n, ho, wo, co, dev)).cpu(), exact, 2e-5, "fp32 LDS-DMA conv vs F.conv2d")


def _as_t(buf, n, h, w, c, dev):
    from vcamd import hip
    t = hip.T.empty(n, h, w, c, dev)
    t.buf.copy_(buf)
    return t


@pytest.mark.parametrize("h,w,n", [(37, 75, 2), (136, 240, 1), (9, 50, 1), (300, 250, 1)])
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_on_the_streaming_kernel_is_bit_identical(dev, h, w, n, inverse):
    """GDN / IGDN (compressai.layers.GDN: a 1x1 contraction of x^2, then x * rsqrt / sqrt of it, + the block's skip path) on
    VC_CFG_PWS (csrc/conv_pws.hip, fp32 128 -> 128): the tile's own input stays in registers (B-operand layout == accumulator
    layout), the residual is requested one 16-pixel unit ahead.  Bit for bit the general kernel's result, with and without
    residual, three repetitions (a wrong wait count shows as run-to-run differences), ragged widths, more tiles than waves."""
    from vcamd import hip
    g = np.random.default_rng(91)
    gamma = torch.from_numpy((0.1 * np.eye(128) + g.uniform(0.0, 0.004, size=(128, 128))).astype(np.float32))
    beta = torch.from_numpy(g.uniform(0.5, 1.5, size=128).astype(np.float32))
    pc = hip.PackedConv(gamma.view(128, 128, 1, 1), beta, device=dev)
    assert hip.CFG_PWS in pc.candidates
    x = hip.nchw_to_nhwc(_rand((n, 128, h, w), 92).to(dev))
    res = hip.nchw_to_nhwc(_rand((n, 128, h, w), 93).to(dev))
    epi = hip.EPI_IGDN if inverse else hip.EPI_GDN

    def run(cfg):
        pc.tuned = {(n, h, w, 0, hip.ACT_NONE, epi): cfg | hip.CFG_EXACT}
        return [pc(x, epi=epi, mul=x, in_xform=hip.IN_SQUARE).buf.clone(), pc(x, epi=epi, mul=x, in_xform=hip.IN_SQUARE, res=res).buf.clone()]
    base = run(pc.cfg if pc.cfg <= 2 else 2)
    for rep in range(3):
        for a, b in zip(base, run(hip.CFG_PWS)):
            assert torch.equal(a, b), (a - b).abs().max().item()
    xc = hip.nhwc_to_nchw(x).cpu()
    norm = F.conv2d(xc * xc, gamma.view(128, 128, 1, 1), beta)
    ref = xc * torch.sqrt(norm) if inverse else xc * torch.rsqrt(norm)
    _close(hip.nhwc_to_nchw(_as_t(base[0], n, h, w, 128, dev)).cpu(), ref, 2e-5, "GDN vs torch")


@pytest.mark.parametrize("cin,cout,h,w,n", [(128, 128, 48, 96, 2), (128, 128, 37, 75, 1), (64, 64, 33, 70, 1), (128, 256, 32, 64, 1)])
def test_lds_dma_3x3_half_precision_residual(dev, cin, cout, h, w, n):
    """VC_CFG_RES_F16 on the LDS-DMA 3x3 kernel (round 5; the residual blocks of LHBDC/model/layers.py:48-56,82-91 on the fp16 path): a
    half-precision identity gives exactly what the fp32 identity holding the same values gives on the classic fp16 instance -- the sum is
    formed in fp32 and rounded once; whole tiles through the 16-byte exchange epilogue, ragged edges through the general one; half and
    fp32 output."""
    from vcamd import hip
    hip.set_conv_precision("fp16")
    try:
        pc = hip.PackedConv(_rand((cout, cin, 3, 3), 71, 1.0 / np.sqrt(9 * cin)), _rand((cout,), 72, 0.1), device=dev)
    finally:
        hip.set_conv_precision("fp32")
    assert pc.half_res_ok and hip.CFG_DMA in pc.candidates
    x16 = hip.T.empty(n, h, w, cin, dev, "f16")
    x16.buf.copy_(hip.nchw_to_nhwc(_rand((n, cin, h, w), 73).to(dev)).buf.half())
    r16 = hip.T.empty(n, h, w, cout, dev, "f16")
    r16.buf.copy_(hip.nchw_to_nhwc(_rand((n, cout, h, w), 74).to(dev)).buf.half())
    r32 = hip.T.empty(n, h, w, cout, dev)
    r32.buf.copy_(r16.buf.float())
    for out_f16 in (True, False):
        fl = hip.CFG_F16 | hip.CFG_IN_F16 | (hip.CFG_OUT_F16 if out_f16 else 0)
        pc.tuned = {(n, h, w, fl): pc.cfg | hip.CFG_EXACT | fl}                     # classic instance, fp32 identity
        ref = pc(x16, act=hip.ACT_LRELU, slope=0.01, res=r32, out_f16=out_f16).buf.clone()
        for rep in range(3):                                                        # the LDS-DMA kernel, half identity (picked by the flag)
            o = pc(x16, act=hip.ACT_LRELU, slope=0.01, res=r16, out_f16=out_f16)
            assert o.dtype == ("f16" if out_f16 else "f32")
            assert torch.equal(o.buf, ref), (out_f16, rep, (o.buf.float() - ref.float()).abs().max().item())
