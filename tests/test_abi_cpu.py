"""The C-ABI library loads without a GPU and exports every symbol include/vc_hip.h declares;
host-side entry points (weight packing, table construction) behave."""
import ctypes
import os
import re

import numpy as np
import torch

from vcamd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "vc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = hip.lib()
    names = _declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(hip.EXPORTED_SYMBOLS) == names
    assert lib.vc_target_arch() == b"gfx950"


def test_library_is_in_tree():
    assert os.path.dirname(hip.LIB_PATH) == os.path.join(ROOT, "video-compression_amd")


def _unpack(wpk, cfg, cout, cin, k, stride, ps):
    """inverse of the documented fragment layout [n-tile][tap][k-step][lane][4]"""
    L = hip.lib()
    mt = {3: 16, 4: 4}.get(cfg, 32)
    ks = {3: 16, 4: 4}.get(cfg, 8)
    ck = L.vc_conv_chunk(cfg, k, stride, cin)
    cin_pad = -(-cin // ck) * ck
    bn = {0: 128, 1: 64, 2: 32, 3: 16, 4: 4}[cfg]
    cout_pad = -(-cout // bn) * bn
    arr = wpk.reshape(cout_pad // mt, k * k, cin_pad // ks, 64, 4)
    w = np.zeros((cout, cin, k, k), dtype=np.float32)
    cps = cout // 4 if ps else 0
    for cop in range(cout):
        co = (cop % cps) * 4 + cop // cps if ps else cop
        for ci in range(cin):
            kst, r = divmod(ci, ks)
            kk, e = divmod(r, 4)
            lane = kk * mt + cop % mt       # cfg 4: ks == 4 so kk == 0; lanes 4..63 replicate lanes 0..3
            w[co, ci] = arr[cop // mt, :, kst, lane, e].reshape(k, k)
    return w


def test_weight_packing_is_a_permutation_with_zero_padding():
    L = hip.lib()
    g = np.random.default_rng(0)
    for cout, cin, k, stride, ps in ((128, 128, 3, 1, False), (512, 128, 3, 1, True), (32, 8, 7, 1, False),
                                     (2, 16, 7, 1, False), (128, 3, 3, 2, False), (12, 128, 3, 1, True),
                                     (192, 128, 3, 1, False), (128, 19, 1, 2, False)):
        w = g.standard_normal((cout, cin, k, k)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        pc = hip.PackedConv(torch.from_numpy(w), torch.from_numpy(b), stride=stride, pixelshuffle=ps, device="cpu")
        wpk = pc.wpk.numpy()
        assert wpk.size == L.vc_conv_packed_weight_floats(pc.cfg, cout, cin, k, k, stride)
        assert np.array_equal(_unpack(wpk, pc.cfg, cout, cin, k, stride, ps), w)
        rep = 16 if pc.cfg == 4 else 1                                           # N4 replicates over the 16 MFMA blocks
        assert np.isclose(np.abs(wpk).sum(), rep * np.abs(w).sum(), rtol=1e-5)   # padding is zeros
        bias = pc.bias.numpy()
        if ps:
            cps = cout // 4
            assert np.array_equal(bias[:cout], b.reshape(cps, 4).T.reshape(-1))
        else:
            assert np.array_equal(bias[:cout], b)
        assert (bias[cout:] == 0).all()


def test_conv_desc_rejects_bad_shapes_without_touching_a_gpu():
    d = hip.ConvDesc()
    assert hip.lib().vc_conv2d_nhwc(None, d) == -1


def test_planar_deformable_entry_rejects_inconsistent_planes_without_touching_a_gpu():
    """vc_offset_diversity_hxp: x1 / x2 describe ONE group's plane of a [n][G/2][h][w][cg] half tensor -- c = sw = cg, sh = w * cg,
    sn = (G/2) * h * w * cg; anything else is VC_EINVAL before any launch (include/vc_hip.h).  vc_to_half_planar: cg in {4, 8, 12, 16}
    dividing c."""
    L = hip.lib()
    n, h, w, cg, groups = 1, 4, 6, 8, 16
    hg = groups // 2
    buf = (ctypes.c_float * (n * h * w * 27 * hg))()
    p = ctypes.cast(buf, ctypes.c_void_p).value
    good = hip.View(p, n, h, w, cg, hg * h * w * cg, w * cg, cg)
    raw = hip.View(p, n, h, w, 27 * hg, h * w * 27 * hg, w * 27 * hg, 27 * hg)
    flow = hip.View(p, n, h, w, 2, h * w * 2, w * 2, 2)
    out = hip.View(p, n, h, w, groups * 4, h * w * groups * 4, w * groups * 4, groups * 4)
    for bad in (hip.View(p, n, h, w, cg, hg * h * w * cg, w * cg, 2 * cg),           # pixel stride is not cg
                hip.View(p, n, h, w, cg, hg * h * w * cg, w * cg + 8, cg),           # padded rows
                hip.View(p, n, h, w, cg, h * w * cg, w * cg, cg)):                   # image stride of ONE plane
        assert L.vc_offset_diversity_hxp(None, bad, raw, flow, good, raw, flow, 10.0, p, None, groups, out) == -1
        assert L.vc_offset_diversity_hxp(None, good, raw, flow, bad, raw, flow, 10.0, p, None, groups, out) == -1
    x = hip.View(p, n, h, w, 64, h * w * 64, w * 64, 64)
    for cg_bad in (0, 2, 6, 20, 24):
        assert L.vc_to_half_planar(None, x, cg_bad, p) == -1
