"""Range coder: hand-computed known answers from the published rans64 algorithm, round
trips, and byte-equality of the product coder (csrc/rans_host.cpp, through the C ABI) with the C oracle.
The reference holds no vectors for this third-party coder: parity with the real library is UNPINNED."""
import numpy as np
import pytest

from oracle.cai import ans
from vcamd import hip

# one table: two regular bins + escape bin; cdf = [0, 16384, 49152, 65536]
CDF = np.array([[0, 16384, 49152, 65536]], dtype=np.int32)
SIZES = np.array([4], dtype=np.int32)
OFFS = np.array([0], dtype=np.int32)

KATS = [
    ([], "0000008000000000"),                 # flush of the initial state 2^31
    ([1], "0040000001000000"),                # x = (2^31/2^15 << 16) + 16384
    ([5], "61c0000000020000"),                # escape: raw=6 -> nibbles 6,1 then the escape bin
    ([-1], "11c0000000020000"),               # negative: raw=1
]


@pytest.mark.parametrize("symbols,hexbytes", KATS)
def test_known_answers(symbols, hexbytes):
    idx = np.zeros(len(symbols), dtype=np.int32)
    sym = np.array(symbols, dtype=np.int32)
    expect = bytes.fromhex(hexbytes)
    assert ans.encode_with_indexes(sym, idx, CDF, SIZES, OFFS) == expect
    assert hip.rans_encode(sym, idx, CDF, SIZES, OFFS) == expect
    assert ans.decode_with_indexes(expect, idx, CDF, SIZES, OFFS).tolist() == symbols
    assert hip.rans_decode(expect, idx, CDF, SIZES, OFFS).tolist() == symbols


def test_pmf_to_quantized_cdf_known_answers():
    for pmf, expect in (([0.5, 0.25, 0.25], [0, 32768, 49152, 65536]),
                        ([1.0, 0.0, 0.0], [0, 65534, 65535, 65536]),       # two empty bins steal from bin 0
                        ([0.9999, 0.00001, 0.00009], [0, 65529, 65530, 65536])):
        assert ans.pmf_to_quantized_cdf(pmf) == expect
        assert hip.pmf_to_quantized_cdf(pmf).tolist() == expect


def _gaussian_tables():
    from oracle.cai.entropy_models import GaussianConditional, get_scale_table
    gc = GaussianConditional(None)
    gc.update_scale_table(get_scale_table(), force=True)
    return gc._quantized_cdf.numpy(), gc._cdf_length.numpy(), gc._offset.numpy(), gc.scale_table.numpy()


@pytest.mark.parametrize("count,seed", [(1, 0), (7, 1), (1000, 2), (200_000, 3)])
def test_roundtrip_and_product_equals_oracle(count, seed):
    cdfs, sizes, offs, table = _gaussian_tables()
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 64, count).astype(np.int32)
    sym = np.round(rng.standard_normal(count) * table[idx]).astype(np.int32)
    sym[::97] = rng.integers(-100_000, 100_000, sym[::97].size)          # out-of-table -> bypass path
    sym[::89] = offs[idx[::89]] + sizes[idx[::89]] - 2                    # exactly the escape boundary
    a = ans.encode_with_indexes(sym, idx, cdfs, sizes, offs)
    b = hip.rans_encode(sym, idx, cdfs, sizes, offs)
    assert a == b
    assert (hip.rans_decode(b, idx, cdfs, sizes, offs) == sym).all()
    assert (ans.decode_with_indexes(a, idx, cdfs, sizes, offs) == sym).all()


def test_random_pmfs_product_equals_oracle():
    rng = np.random.default_rng(11)
    for _ in range(200):
        n = int(rng.integers(2, 200))
        pmf = rng.random(n).astype(np.float32) ** 8
        pmf[rng.integers(0, n)] = 0.0
        if pmf.sum() <= 0:
            pmf[0] = 1.0
        pmf /= pmf.sum()
        a = np.array(ans.pmf_to_quantized_cdf(pmf))
        b = hip.pmf_to_quantized_cdf(pmf)
        assert (a == b).all() and a[0] == 0 and a[-1] == 65536 and (np.diff(a) > 0).all()
    with pytest.raises(ValueError):
        ans.pmf_to_quantized_cdf([0.0, 0.0])
    with pytest.raises(hip.VcError):
        hip.pmf_to_quantized_cdf([0.0, 0.0])


def test_decoder_rejects_truncated_stream():
    cdfs, sizes, offs, table = _gaussian_tables()
    idx = np.full(5000, 40, dtype=np.int32)
    sym = np.round(np.random.default_rng(0).standard_normal(5000) * table[40]).astype(np.int32)
    data = hip.rans_encode(sym, idx, cdfs, sizes, offs)
    with pytest.raises(hip.VcError):
        hip.rans_decode(data[:len(data) // 2 // 4 * 4], idx, cdfs, sizes, offs)
    with pytest.raises(hip.VcError):
        hip.rans_decode(b"\x00\x00", idx, cdfs, sizes, offs)


def test_extreme_bypass_values():
    """The largest magnitudes the format defines: the folded value must stay below 2^28 (seven nibbles) -- the format's
    own encoder counts nibbles with a 32-bit shift that is undefined beyond.  The product refuses larger values."""
    sym = np.array([2 ** 27 - 1, -(2 ** 27), 2 ** 27 - 3, 0, 3, -(2 ** 26)], dtype=np.int32)
    idx = np.zeros(sym.size, dtype=np.int32)
    a = ans.encode_with_indexes(sym, idx, CDF, SIZES, OFFS)
    b = hip.rans_encode(sym, idx, CDF, SIZES, OFFS)
    assert a == b
    assert hip.rans_decode(b, idx, CDF, SIZES, OFFS).tolist() == sym.tolist()
    assert ans.decode_with_indexes(a, idx, CDF, SIZES, OFFS).tolist() == sym.tolist()
    with pytest.raises(hip.VcError):
        hip.rans_encode(np.array([2 ** 30], dtype=np.int32), idx[:1], CDF, SIZES, OFFS)


def test_out_of_range_table_index_is_rejected_not_dereferenced():
    sym = np.array([0, 1, 0], dtype=np.int32)
    good = np.zeros(3, dtype=np.int32)
    data = hip.rans_encode(sym, good, CDF, SIZES, OFFS)
    for bad_value in (1, -1, 2 ** 20):
        bad = np.array([0, bad_value, 0], dtype=np.int32)
        with pytest.raises(hip.VcError):
            hip.rans_encode(sym, bad, CDF, SIZES, OFFS)
        with pytest.raises(hip.VcError):
            hip.rans_decode(data, bad, CDF, SIZES, OFFS)
    with pytest.raises(hip.VcError):                  # a table claiming more entries than its row holds
        hip.rans_encode(sym, good, CDF, np.array([9], dtype=np.int32), OFFS)
    with pytest.raises(hip.VcError):                  # one size / offset per table
        hip.rans_encode(sym, good, CDF, np.array([4, 4], dtype=np.int32), OFFS)


def test_stream_decoder_continues_across_calls():
    """RansDecoder.set_stream / decode_stream (ICIP2024/src/model/elic.py:428-429): one string read in several calls, product
    (vc_rans_decode_stream) and oracle; BufferedRansEncoder.flush over several queued calls == one encode of the lists."""
    cdfs, sizes, offs, table = _gaussian_tables()
    rng = np.random.default_rng(5)
    n = 6000
    idx = rng.integers(0, 64, n).astype(np.int32)
    sym = np.round(rng.standard_normal(n) * table[idx]).astype(np.int32)
    sym[::61] = rng.integers(-5000, 5000, sym[::61].size)
    enc = ans.BufferedRansEncoder()
    enc.encode_with_indexes(sym[:2500].tolist(), idx[:2500].tolist(), cdfs, sizes, offs)
    enc.encode_with_indexes(sym[2500:].tolist(), idx[2500:].tolist(), cdfs, sizes, offs)
    data = enc.flush()
    assert data == hip.rans_encode(sym, idx, cdfs, sizes, offs)
    for cuts in ((0, 2500, n), (0, 1, 17, 4000, n), (0, n)):
        o = ans.RansDecoder()
        o.set_stream(data)
        p = hip.RansStreamDecoder(data)
        got_o, got_p = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            got_o += o.decode_stream(idx[a:b], cdfs, sizes, offs)
            got_p += p.decode_stream(idx[a:b], cdfs, sizes, offs).tolist()
        assert got_o == sym.tolist() and got_p == sym.tolist()
    with pytest.raises(hip.VcError):                     # reading past the end of the string
        p = hip.RansStreamDecoder(data)
        p.decode_stream(np.concatenate([idx, idx, idx, idx]), cdfs, sizes, offs)
