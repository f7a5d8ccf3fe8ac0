"""CPU: the host side of the split-operand fp32 path (csrc/conv_split.hip) -- the packed weights hold, per (output channel, input
channel, tap), three bf16 pieces whose sum is the fp32 weight EXACTLY, every weight exactly once, zeros in the padded k-groups; the
3x3 period order (cin % 32 == 0) has no padding at all, the 5x5 / 7x7 period order (cin % 16 == 0) two padded k-groups per period.  No GPU needed (vc_conv_pack_weights_split runs on the host)."""
import numpy as np
import pytest

from vcamd import hip


def _pack(w, k, ps=0):
    cout, cin = w.shape[:2]
    L = hip.lib()
    nbytes = L.vc_conv_packed_weight_bytes_split(cout, cin, k)
    assert nbytes > 0
    buf = np.zeros(nbytes // 2, dtype=np.uint16)
    bias = np.empty(cout, dtype=np.float32)
    b_in = np.arange(cout, dtype=np.float32)
    assert L.vc_conv_pack_weights_split(w.ctypes.data, b_in.ctypes.data, cout, cin, k, ps, buf.ctypes.data, bias.ctypes.data) == 0
    return buf, bias, b_in


def _pieces_to_float(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("cout,cin,k,units_per_block", [
    (64, 32, 7, 25 * 2),        # 7x7, cin % 16 == 0: periods of two 8-channel chunks = 25 units (two padded k-groups per period)
    (32, 64, 7, 25 * 4),
    (16, 32, 7, 13 * 4),        # one N-tile of 16 channels: the per-chunk order, 13 units per 8-channel chunk
    (32, 8, 7, 13 * 1),         # cin = 8: one chunk, the per-chunk order
    (64, 96, 5, 13 * 6),        # 5x5, cin % 16 == 0: periods of 13 units
    (32, 24, 5, 7 * 3),         # 5x5, cin = 24: the per-chunk order, 7 units per chunk
    (128, 64, 3, 9 * 2),        # 3x3, cin % 32 == 0: periods of 9 units, no padding
    (64, 48, 3, 5 * 3),         # 3x3, cin = 48: the padded order, 5 units per 16-channel chunk
])
def test_packed_split_weights_are_an_exact_permutation_of_the_weights(cout, cin, k, units_per_block):
    g = np.random.default_rng(3)
    w = g.standard_normal((cout, cin, k, k)).astype(np.float32)
    w[0, 0, 0, 0] = 1.0 + 2.0 ** -23          # needs all three pieces
    w[1, 0, 0, 0] = -3.0e-30
    buf, bias, b_in = _pack(w, k)
    bn = 64 if cout % 64 == 0 else (32 if cout % 32 == 0 else 16)
    ntw = bn // 16
    nfrag = (cout // bn) * units_per_block * 3 * ntw
    body = buf[: nfrag * 512].reshape(cout // bn, units_per_block, 3, ntw, 64, 8)
    assert not buf[nfrag * 512:].any()                                   # the slack behind the last unit is zero
    hi, mid, lo = (_pieces_to_float(body[:, :, p]) for p in range(3))
    total = (lo + mid) + hi                                              # exact: both partial sums are representable
    vals = total.reshape(-1)
    nz = vals[vals != 0]
    assert nz.size == w.size
    assert np.array_equal(np.sort(nz), np.sort(w.reshape(-1)))           # every weight exactly once, bit for bit
    # a piece only ever continues the one above it: where hi is zero, so are mid and lo
    assert not (mid[hi == 0] != 0).any() and not (lo[mid == 0] != 0).any()
    slots = total.size
    if k == 3 and cin % 32 == 0:
        assert slots == w.size                                           # the period order: no padded k-group at all
    elif k in (5, 7) and cin % 16 == 0 and bn > 16:
        assert slots == w.size + cout * (cin // 16) * 2 * 8              # two padded k-groups (of 8 channels) per period and output channel
    else:
        assert slots > w.size
    assert np.array_equal(bias, b_in)


def test_pixel_shuffle_permutes_output_channels_like_the_native_packing():
    g = np.random.default_rng(4)
    cout, cin = 128, 32
    w = g.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    buf, bias, b_in = _pack(w, 3, ps=1)
    cps = cout // 4
    want = np.array([b_in[(cop % cps) * 4 + cop // cps] for cop in range(cout)], dtype=np.float32)
    assert np.array_equal(bias, want)
    # lane (co, q) of n-tile 0, unit 0, piece hi, block 0 holds packed channel cop = lane & 15 -> module channel (cop % cps) * 4 + cop // cps;
    # k-group 0 of unit 0 = tap 0, plane 0 of the first chunk: channels 0..7
    body = buf[: 2 * 9 * 3 * 4 * 512].reshape(2, 9, 3, 4, 64, 8)
    got = (_pieces_to_float(body[0, 0, 2, 0, :16]) + _pieces_to_float(body[0, 0, 1, 0, :16])) + _pieces_to_float(body[0, 0, 0, 0, :16])
    co = np.array([(cop % cps) * 4 + cop // cps for cop in range(16)])
    assert np.array_equal(got, w[co, 0:8, 0, 0])


def test_shapes_the_split_path_does_not_serve_are_refused():
    L = hip.lib()
    assert L.vc_conv_packed_weight_bytes_split(64, 32, 9) == 0           # kernel size
    assert L.vc_conv_packed_weight_bytes_split(64, 12, 7) == 0           # cin not a multiple of 8
    assert L.vc_conv_packed_weight_bytes_split(24, 32, 7) == 0           # cout not a multiple of 16
    assert L.vc_conv_packed_weight_bytes_split(16, 32, 5) == 0           # 16 output channels: 7x7 only
    assert L.vc_conv_packed_weight_bytes_split(64, 24, 3) == 0           # 3x3: cin in chunks of 16
    assert L.vc_conv_chunk(hip.CFG_SPLIT, 7, 1, 32) == 8 and L.vc_conv_chunk(hip.CFG_SPLIT, 3, 1, 32) == 16
    assert L.vc_conv_chunk(hip.CFG_SPLIT, 3, 2, 32) == -1                # stride 2
