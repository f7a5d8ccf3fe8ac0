"""ctypes front end of oracle/rans_oracle.c with the call shapes of compressai.ans
(RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes) and
compressai._CXX.pmf_to_quantized_cdf.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.dirname(_HERE)
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "librans_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _ORACLE_DIR, "librans_oracle.so"])
        lib = ctypes.CDLL(path)
        i32p = ctypes.POINTER(ctypes.c_int32)
        lib.vco_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_uint32)]
        lib.vco_pmf_to_quantized_cdf.restype = ctypes.c_int
        lib.vco_rans_encode_with_indexes.argtypes = [i32p, i32p, ctypes.c_size_t, i32p, ctypes.c_int,
                                                     i32p, i32p,
                                                     ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))]
        lib.vco_rans_encode_with_indexes.restype = ctypes.c_long
        lib.vco_rans_decode_with_indexes.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t,
                                                     i32p, ctypes.c_size_t, i32p, ctypes.c_int,
                                                     i32p, i32p, i32p]
        lib.vco_rans_decode_with_indexes.restype = ctypes.c_int
        lib.vco_rans_decode_stream.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64),
                                               i32p, ctypes.c_size_t, i32p, ctypes.c_int, i32p, i32p, i32p]
        lib.vco_rans_decode_stream.restype = ctypes.c_int
        lib.vco_free.argtypes = [ctypes.c_void_p]
        _LIB = lib
    return _LIB


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _ptr(a, ty=ctypes.c_int32):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _dense_cdfs(cdfs):
    """list-of-lists or 2-D array -> dense int32 [T, stride]."""
    if isinstance(cdfs, np.ndarray) and cdfs.ndim == 2:
        return _i32(cdfs)
    stride = max(len(c) for c in cdfs)
    out = np.zeros((len(cdfs), stride), dtype=np.int32)
    for t, c in enumerate(cdfs):
        out[t, :len(c)] = c
    return out


def pmf_to_quantized_cdf(pmf, precision=16):
    assert precision == 16
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    cdf = np.zeros(p.size + 1, dtype=np.uint32)
    rc = _lib().vco_pmf_to_quantized_cdf(_ptr(p, ctypes.c_float), p.size, _ptr(cdf, ctypes.c_uint32))
    if rc != 0:
        raise ValueError("degenerate pmf")
    return cdf.astype(np.int64).tolist()


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    assert sym.size == idx.size
    dense = _dense_cdfs(cdfs)
    sizes, offs = _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = _lib().vco_rans_encode_with_indexes(_ptr(sym), _ptr(idx), sym.size, _ptr(dense), dense.shape[1],
                                            _ptr(sizes), _ptr(offs), ctypes.byref(out))
    if n < 0:
        raise RuntimeError("oracle rANS encode failed")
    data = ctypes.string_at(out, n)
    _lib().vco_free(out)
    return data


def decode_with_indexes(data, indexes, cdfs, cdf_sizes, offsets):
    idx = _i32(indexes).reshape(-1)
    dense = _dense_cdfs(cdfs)
    sizes, offs = _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(idx.size, dtype=np.int32)
    rc = _lib().vco_rans_decode_with_indexes(_ptr(buf, ctypes.c_uint8), buf.size, _ptr(idx), idx.size,
                                             _ptr(dense), dense.shape[1], _ptr(sizes), _ptr(offs), _ptr(out))
    if rc != 0:
        raise RuntimeError("oracle rANS decode failed")
    return out


class RansEncoder:
    def encode_with_indexes(self, *args):
        return encode_with_indexes(*args)


class RansDecoder:
    """decode_with_indexes (whole string) and set_stream / decode_stream (ICIP2024/src/model/elic.py:428-429,566,584: one
    string read in several calls, each continuing from the coder state the previous call left)."""

    def decode_with_indexes(self, *args):
        return decode_with_indexes(*args).tolist()

    def set_stream(self, data):
        self._buf = np.frombuffer(bytes(data), dtype=np.uint8)
        self._state = np.zeros(2, dtype=np.uint64)

    def decode_stream(self, indexes, cdfs, cdf_sizes, offsets):
        idx = _i32(indexes).reshape(-1)
        dense = _dense_cdfs(cdfs)
        sizes, offs = _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
        out = np.zeros(idx.size, dtype=np.int32)
        rc = _lib().vco_rans_decode_stream(_ptr(self._buf, ctypes.c_uint8), self._buf.size, _ptr(self._state, ctypes.c_uint64),
                                           _ptr(idx), idx.size, _ptr(dense), dense.shape[1], _ptr(sizes), _ptr(offs), _ptr(out))
        if rc != 0:
            raise RuntimeError("oracle rANS stream decode failed")
        return out.tolist()


class BufferedRansEncoder:
    """compressai.ans.BufferedRansEncoder (ICIP2024/src/model/elic.py:329,404-407): encode_with_indexes only queues its
    symbols; flush() codes everything queued -- in reverse, as one rANS string -- and empties the queue.  One flush over
    several queued calls therefore equals RansEncoder.encode_with_indexes on the concatenated lists."""

    def __init__(self):
        self._sym, self._idx, self._tables = [], [], None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        self._sym.extend(int(v) for v in symbols)
        self._idx.extend(int(v) for v in indexes)
        self._tables = (cdfs, cdf_sizes, offsets)

    def flush(self):
        if self._tables is None:
            data = encode_with_indexes([], [], np.zeros((1, 3), dtype=np.int32), [3], [0])
        else:
            data = encode_with_indexes(self._sym, self._idx, *self._tables)
        self._sym, self._idx = [], []
        return data
