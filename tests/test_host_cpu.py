"""Host-side logic of the product without a GPU: checkpoint schema, table construction, container I/O,
and loud failure (no CPU fallback) when asked to compute without the device."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, lhbdc_pair
from vcamd import hip, lhbdc


def test_state_dict_schema_equals_reference_schema():
    ref = [l.strip() for l in open(os.path.join(GOLDEN, "lhbdc_state_schema.txt"))]
    mine = sorted(f"{k} {list(v.shape)}" for k, v in lhbdc.Model().state_dict().items())
    assert mine == ref and len(mine) == 378


def test_checkpoint_loads_strict_and_tables_match_oracle():
    ora, prod = lhbdc_pair(99)
    for name in ("mv_compressor", "residual_compressor"):
        a, b = getattr(prod, name), getattr(ora, name)
        a.update(force=True)
        b.update(force=True)
        for part in ("entropy_bottleneck", "gaussian_conditional"):
            for buf in ("_quantized_cdf", "_cdf_length", "_offset"):
                assert torch.equal(getattr(getattr(a, part), buf), getattr(getattr(b, part), buf))
    # a checkpoint saved AFTER update() carries populated, differently-sized CDF buffers
    sd = prod.residual_compressor.state_dict()
    fresh = lhbdc.ResidualCompressor()
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.gaussian_conditional._quantized_cdf, prod.residual_compressor.gaussian_conditional._quantized_cdf)


def test_container_roundtrip_matches_oracle_writer():
    from oracle import lhbdc as ol
    mv = {"strings": [[b"\x01\x02\x03\x04"], [b"zz"]], "shape": torch.Size([5, 8])}
    res = {"strings": [[os.urandom(40)], [os.urandom(12)]], "shape": torch.Size([17, 30])}
    blob = lhbdc.write_container(None, 845, mv, res)
    assert blob == ol.write_container(845, mv, res)
    lm, s_mv, s_res, sh_mv, sh_res = lhbdc.read_container(blob)
    assert lm == 845 and s_mv == mv["strings"] and s_res == res["strings"]
    assert tuple(sh_mv) == (5, 8) and tuple(sh_res) == (17, 30)
    with pytest.raises(hip.VcError):
        lhbdc.read_container(blob[:10])


def test_no_cpu_fallback():
    m = lhbdc.Model()
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(hip.VcError):
        m(x, x, x, False)
    with pytest.raises(hip.VcError):
        m.residual_compressor.compress(x)


def test_seeded_checkpoint_is_deterministic():
    from vcamd.seeding import seeded_state_dict
    t = lhbdc.Model().state_dict()
    a, b = seeded_state_dict(t, 5), seeded_state_dict(t, 5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = seeded_state_dict(t, 6)
    assert not torch.equal(a["FlowNet.netBasic.0.netBasic.0.weight"], c["FlowNet.netBasic.0.netBasic.0.weight"])


def test_flex_state_dict_schema_and_child_loading():
    from vcamd import flex
    ref = [l.strip() for l in open(os.path.join(GOLDEN, "flex_state_schema.txt"))]
    m = flex.BidirFlowRef(n=4)
    mine = sorted(f"{k} {list(v.shape)}" for k, v in m.state_dict().items())
    assert mine == ref and len(mine) == 396
    # Flex checkpoints are loaded child by child (test/utils.py:253-270), after update() was called
    m.flow_compressor.update(force=True)
    child = m.flow_compressor.state_dict()
    fresh = flex.BidirFlowRef(n=4)
    getattr(fresh, "flow_compressor").load_state_dict(child)
    assert fresh.flow_compressor.gaussian_conditional._quantized_cdf.shape[0] == 64


def test_flex_gain_vector_interpolation():
    from vcamd import flex
    g = flex.Gain_Module(n=4, N=128)
    with torch.no_grad():
        g.gain_matrix.copy_(torch.linspace(-2, 2, 4 * 128).reshape(4, 128))
    v = g.vector([1], 0.33)
    exp = torch.abs(g.gain_matrix[1]) ** 0.33 * torch.abs(g.gain_matrix[2]) ** 0.67
    assert torch.allclose(v, exp.detach(), rtol=1e-6)
    assert torch.equal(g.vector([3], 1), torch.abs(g.gain_matrix[3]).detach())
    with pytest.raises(IndexError):
        g.vector([3], 0.5)      # n+1 out of range, like the reference (Appendix B.7)
