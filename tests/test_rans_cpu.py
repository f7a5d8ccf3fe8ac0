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


def test_decoder_survives_garbage_and_bit_flips():
    """A decoder is fed by files: random bytes, truncated strings and valid strings with flipped bits must end in VcError
    or in SOME symbol array of the requested length -- never in a crash, an out-of-bounds read or an endless loop (the
    test's own timeout is the endless-loop check)."""
    cdfs, sizes, offs, table = _gaussian_tables()
    rng = np.random.default_rng(11)
    n = 3000
    idx = rng.integers(0, len(table), n).astype(np.int32)
    sym = np.round(rng.standard_normal(n) * np.asarray(table)[idx]).astype(np.int32)
    good = hip.rans_encode(sym, idx, cdfs, sizes, offs)
    assert hip.rans_decode(good, idx, cdfs, sizes, offs).tolist() == sym.tolist()
    outcomes = {"error": 0, "symbols": 0}

    def attempt(data):
        try:
            out = hip.rans_decode(bytes(data), idx, cdfs, sizes, offs)
        except hip.VcError:
            outcomes["error"] += 1
            return
        assert out.shape == (n,) and out.dtype == np.int32
        outcomes["symbols"] += 1

    for _ in range(40):                                   # garbage of every length class
        attempt(rng.integers(0, 256, int(rng.integers(0, 2 * len(good))), dtype=np.uint8).tobytes())
    for _ in range(60):                                   # a valid string with a few flipped bits
        buf = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            buf[int(rng.integers(0, len(buf)))] ^= 1 << int(rng.integers(0, 8))
        attempt(buf)
    for cut in (0, 1, 3, 4, 7, 8, len(good) - 4, len(good) - 1):
        attempt(good[:cut])
    attempt(good + b"\x00" * 64)                          # trailing bytes behind a valid string
    assert outcomes["error"] > 0 and outcomes["error"] + outcomes["symbols"] == 40 + 60 + 8 + 1
    # the resumable decoder takes the same abuse
    dec = hip.RansStreamDecoder(rng.integers(0, 256, 200, dtype=np.uint8).tobytes())
    try:
        dec.decode_stream(idx[:1000], cdfs, sizes, offs)
        dec.decode_stream(idx[1000:], cdfs, sizes, offs)
    except hip.VcError:
        pass


def test_container_parser_rejects_inconsistent_headers():
    from vcamd import lhbdc
    mv = {"strings": [[b"ab" * 10], [b"c" * 5]], "shape": (4, 6)}
    res = {"strings": [[b"d" * 30], [b"e" * 7]], "shape": (4, 6)}
    blob = lhbdc.write_container(None, 1626, mv, res)
    lm, s_mv, s_res, sh_mv, sh_res = lhbdc.read_container(blob)
    assert lm == 1626 and s_mv[0][0] == b"ab" * 10 and s_res[1][0] == b"e" * 7 and tuple(sh_mv) == (4, 6)
    with pytest.raises(hip.VcError):
        lhbdc.read_container(blob[:10])                   # shorter than the header
    with pytest.raises(hip.VcError):
        lhbdc.read_container(blob[:40])                   # header promises more than the file holds
    bad = bytearray(blob)
    bad[8:12] = np.array(10 ** 6, dtype=np.uint32).tobytes()
    with pytest.raises(hip.VcError):
        lhbdc.read_container(bytes(bad))


def test_host_coder_under_address_and_ub_sanitizers(tmp_path):
    """csrc/rans_host.cpp compiled for the CPU with -fsanitize=address,undefined and driven by tests/native/rans_fuzz.cpp:
    round trips with escapes, garbage / truncated / bit-flipped strings through both decoders, malformed tables and
    indexes.  (Sanitizers run on the CPU build only; the GPU pool refuses them.)"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "rans_fuzz")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all",
                            "-I", os.path.join(root, "include"), os.path.join(root, "tests", "native", "rans_fuzz.cpp"),
                            os.path.join(root, "video-compression_amd", "csrc", "rans_host.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "no sanitizer report" in run.stdout


def test_table_construction_refuses_what_is_not_a_probability_vector():
    for bad in ([float("nan"), 0.5, 0.5], [-0.25, 0.75, 0.5], [float("inf"), 0.0, 0.0], [2.0, 0.0, 0.0], [0.0, 0.0, 0.0]):
        with pytest.raises(hip.VcError):
            hip.pmf_to_quantized_cdf(np.array(bad, dtype=np.float32))
    cdf = hip.pmf_to_quantized_cdf(np.array([1.0, 1e-9, 0.0], dtype=np.float32))     # zero bins steal from the widest one
    assert cdf[0] == 0 and cdf[-1] == 65536 and np.all(np.diff(cdf.astype(np.int64)) > 0)
