"""-m gpu: how often is the END-TO-END bitstream byte-identical to the CPU path's?  (BASELINE.json north_star: "bit-exact on
the range-coder bitstream".)

Coder-level equality is exact by construction (same integers -> same bytes: test_bitstream_gpu.py).  End to end the
integers come out of fp32 convolution stacks whose summation order differs between the HIP kernels and oneDNN:
  * a latent within ~1e-6 of a rounding boundary may round the other way (0-2 symbols per 1.04 M at 1080p), and
  * a hyper-synthesis scale within ~3e-7 (relative) of one of the 64 log-spaced scale-table entries (0.123 apart in log)
    falls into the neighbouring bin: 2 x 3e-7 / 0.123 = 5e-6 per element whose scale is above the 0.11 floor.
The second effect is a property of the FORMAT (the stream carries no indexes; CompressAI streams are known not to be
portable across platforms for this reason), and it scales with the frame: ~0.1 expected differing indexes on a 192x256
crop, ~5 on a 1080p frame.  This test puts numbers on it, calibrated checkpoint (trained-like statistics), encode_B on
both sides, eight different frame triples per size:
  * 192x256: at least 6 of 8 bits_B containers are byte-identical to the CPU oracle's;
  * 1088x1920: the count is REPORTED (a container is identical only if all 1.2 M integers are), and every differing integer
    is shown to be a boundary case -- symbols: each codec alone on the oracle's input differs only where the oracle's own
    value sits within 2e-3 of a half-integer (test_fullsize_gpu.check_teacher_forced); indexes: the oracle's own scale
    within 2e-5 of a table entry, one bin apart, at most 2e-5 N of them.
bench.py reports the same count as ``byte_equality``.  The oracle passes run side by side (oracle.pool).

Round 6: three settings side by side per frame -- plain fp32, + scale refinement (vc_refine_scales, round 5), + symbol refinement
(vc_refine_y_symbols / vc_refine_z_symbols: latents and hyper-latents within 2e-5 of a rounding boundary decided from fp64
re-evaluations of the layers that produce them) -- with the differing indexes, differing symbols, identical containers and the number of
elements decided in fp64 printed for each.  Measured at 1088x1920: 92 / 66 / 66 indexes, 3 / 3 / 2 symbols, 0 / 5 / 6 of 8 containers.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEEDS = (11, 23, 37, 41, 59, 67, 73, 89)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def triple(seed, h, w):
    """Band-limited texture under a global translation + 1 % noise, 8-bit quantised (test_fullsize_gpu.frames_1080p at any size)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, h + 24, w + 32, generator=g), 9, 1)
    out = []
    for t in range(3):
        f = base[..., 2 * t:2 * t + h, 3 * t:3 * t + w] + 0.01 * torch.randn(1, 3, h, w, generator=g)
        out.append((torch.round(f.clamp(0, 1) * 255.0) / 255.0).contiguous())
    return out


def index_flips_are_boundary_cases(tag, mine, ref_lat):
    """Differing scale-table indexes: one bin apart, the oracle's own scale within 2e-5 (relative) of a table entry."""
    from oracle.cai.entropy_models import get_scale_table
    theirs = ref_lat["y_idx"].reshape(-1).numpy().astype(np.int64)
    mine = np.asarray(mine).reshape(-1).astype(np.int64)
    bad = np.nonzero(mine != theirs)[0]
    if bad.size == 0:
        return 0
    assert np.abs(mine[bad] - theirs[bad]).max() == 1, tag
    ls = np.log(np.maximum(ref_lat["scales"].reshape(-1).numpy()[bad].astype(np.float64), 0.11))
    lt = np.log(np.asarray(get_scale_table(), dtype=np.float64))
    dist = np.abs(ls[:, None] - lt[None, :]).min(1)
    assert dist.max() < 2e-5, (tag, float(dist.max()))
    assert bad.size <= max(2, int(2e-5 * theirs.size)), (tag, bad.size)
    return int(bad.size)


@pytest.mark.parametrize("h,w", [(192, 256), (1088, 1920)], ids=["192x256", "1088x1920"])
def test_end_to_end_containers_against_the_cpu_path(dev, h, w):
    from helpers import lhbdc_pair
    from oracle import pool
    from test_fullsize_gpu import check_teacher_forced, teacher_forced
    from vcamd import lhbdc
    ora, prod = lhbdc_pair(1234, dev, calibrated=True)
    prod.mv_compressor.update(force=True)
    prod.residual_compressor.update(force=True)
    triples = [triple(s, h, w) for s in SEEDS]
    refs = pool.run_jobs(triples, pool.lhbdc_encode_job(ora.state_dict()))
    identical, report = 0, []
    from vcamd import hip
    # settings: (vc_refine_scales, vc_refine_*_symbols).  Counted over the eight frames against the CPU path's integers:
    SETTINGS = ((False, False), (True, False), (True, True))
    flips = {k: 0 for k in SETTINGS}      # differing scale-table indexes
    sflips = {k: 0 for k in SETTINGS}     # differing symbols (y and z of both codecs)
    same_blob = {k: 0 for k in SETTINGS}  # byte-identical containers
    refined = [0, 0]                      # (y, z) elements the symbol refinement decided in fp64 (default setting)
    unaided = {False: 0, True: 0}         # CPU-path containers this decoder reads WITHOUT being handed the encoder's indexes
    with torch.no_grad():
        for seed, (xb, xc, xa), ref in zip(SEEDS, triples, refs):
            keep = hip.SCALE_REFINE, hip.SYMBOL_REFINE
            try:
                for setting in SETTINGS:
                    hip.SCALE_REFINE, hip.SYMBOL_REFINE = setting
                    tr = {}
                    mvb, resb = lhbdc.encode_B(prod, xa.to(dev), xc.to(dev), xb.to(dev), trace=tr)
                    flips[setting] += sum(int((torch.from_numpy(tr[c]["y_idx"]).reshape(-1) != ref[c]["y_idx"].reshape(-1)).sum()) for c in ("mv", "res"))
                    sflips[setting] += sum(int((torch.from_numpy(tr[c][k]).reshape(-1) != ref[c][k].reshape(-1)).sum())
                                           for c in ("mv", "res") for k in ("y_sym", "z_sym"))
                    same_blob[setting] += lhbdc.write_container(None, 1626, mvb, resb) == ref["container"]
                    if setting == (True, True):
                        for c in ("mv", "res"):
                            refined[0] += tr[c]["refined"][0]
                            refined[1] += tr[c]["refined"][1]
                    if setting[1]:
                        continue
                    _, s_mv, s_res, sh_mv, sh_res = lhbdc.read_container(ref["container"])
                    try:
                        td = {}
                        lhbdc.decode_B(xb.to(dev), xa.to(dev), prod, s_mv, s_res, sh_mv, sh_res, trace=td)
                        ok = all(int((torch.from_numpy(td[c]["y_sym"]).reshape(-1) != ref[c]["y_sym"].reshape(-1)).sum()) == 0 for c in ("mv", "res"))
                    except hip.VcError:
                        ok = False
                    unaided[setting[0]] += ok
            finally:
                hip.SCALE_REFINE, hip.SYMBOL_REFINE = keep
            trace = {}
            mv_bits, res_bits = lhbdc.encode_B(prod, xa.to(dev), xc.to(dev), xb.to(dev), trace=trace)
            blob = lhbdc.write_container(None, 1626, mv_bits, res_bits)
            same = blob == ref["container"]
            identical += same
            sym = {f"{c}_{k}": int((torch.from_numpy(trace[c][k]).reshape(-1) != ref[c][k].reshape(-1)).sum())
                   for c in ("mv", "res") for k in ("y_sym", "z_sym")}
            # (the scales are a function of the hyper-latent symbols alone: with those equal, index differences are first-order)
            idx = {f"{c}_idx": (index_flips_are_boundary_cases(f"triple {seed} {c}", trace[c]["y_idx"], ref[c]) if not sym[f"{c}_z_sym"] else
                                int((torch.from_numpy(trace[c]["y_idx"]).reshape(-1) != ref[c]["y_idx"].reshape(-1)).sum()))
                   for c in ("mv", "res")}
            report.append((seed, same, len(blob), len(ref["container"]), sym, idx))
            if same:
                assert not any(sym.values()) and not any(idx.values()), (seed, sym, idx)
                continue
            assert any(sym.values()) or any(idx.values()), (seed, "containers differ although every coded integer is equal")
            if any(sym.values()):       # every flipped symbol a boundary case: each codec alone on the ORACLE's input
                check_teacher_forced(f"triple {seed}: mv_compressor", teacher_forced(prod.mv_compressor, ref["mv"], dev))
                check_teacher_forced(f"triple {seed}: residual_compressor", teacher_forced(prod.residual_compressor, ref["res"], dev))
            assert abs(len(blob) - len(ref["container"])) <= max(64, 0.001 * len(blob)), (seed, len(blob), len(ref["container"]))
    print(f"end-to-end encode_B containers byte-identical to the CPU path: {identical} of {len(SEEDS)} ({h}x{w}, calibrated checkpoint)")
    print(f"over the {len(SEEDS)} frames, against the CPU path's integers -- plain fp32 / + vc_refine_scales / + vc_refine_*_symbols: "
          f"scale-table indexes differing {flips[SETTINGS[0]]} / {flips[SETTINGS[1]]} / {flips[SETTINGS[2]]}; symbols differing "
          f"{sflips[SETTINGS[0]]} / {sflips[SETTINGS[1]]} / {sflips[SETTINGS[2]]}; containers byte-identical "
          f"{same_blob[SETTINGS[0]]} / {same_blob[SETTINGS[1]]} / {same_blob[SETTINGS[2]]}; elements decided in fp64: {refined[0]} y, {refined[1]} z; "
          f"CPU-path containers the HIP decoder reads unaided (no index override): {unaided[False]} / {unaided[True]} of {len(SEEDS)}")
    for seed, same, n, m, sym, idx in report:
        print(f"  triple {seed}: {'identical' if same else 'DIFFERENT'} ({n} vs {m} bytes); symbols differing {sym}; indexes differing {idx}")
    if h * w <= 192 * 256:
        assert identical >= 6, report
