"""File-backed ingest and the CLI entry points on the GPU: the loader's frames equal the scripts' process_frame bit for
bit, the prefetch pipeline hands out every item of the dataset list, encode_B / decode_B through the command line give
the bytes / pixels of the library calls and decode the REFERENCE's container."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from helpers import load_fixture  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _clip(n, h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h + 2 * n, w + 2 * n, 3), dtype=np.uint8)
    return [np.ascontiguousarray(base[i:i + h, 2 * i:2 * i + w]) for i in range(n)]


def test_frame_kernels_equal_the_scripts_pixel_code(dev):
    """vc_u8hwc_to_f32nchw_pad == normalize + ReflectionPad2d (encode_B.py:39-64); vc_f32nchw_to_u8hwc == float_to_uint8 + crop."""
    from vcamd import hip
    for (h, w) in ((40, 72), (100, 130), (64, 128)):
        rgb = np.random.default_rng(h).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        x = hip.frame_from_uint8(torch.from_numpy(rgb).to(dev))
        ref = torch.from_numpy(rgb.transpose(2, 0, 1).astype(np.float64))[None].float() / 255.0
        ref = torch.nn.functional.pad(ref, (0, (64 - w % 64) % 64, 0, (64 - h % 64) % 64), mode="reflect")
        assert x.shape == ref.shape and torch.equal(x.cpu(), ref)
        noisy = (x + 0.3 * torch.randn_like(x))
        u8 = hip.frame_to_uint8(noisy, h, w).cpu().numpy()
        want = np.round(np.clip(noisy[0].cpu().numpy(), 0, 1) * 255.0).astype(np.uint8).transpose(1, 2, 0)[:h, :w]
        assert np.array_equal(u8, want)


def test_sequence_reader_serves_the_dataset_items(dev, tmp_path):
    from vcamd import data, gop, hip
    n0, n1, h, w = 19, 17, 40, 72
    clips = [_clip(n0, h, w, 1), _clip(n1, h, w, 2)]
    names = data.write_synthetic_sequences(str(tmp_path), clips, ["b_second", "a_first"])
    with data.SequenceReader(str(tmp_path), None, gop_size=8, test_size=2, device=dev, workers=3, depth=4, timeout=30.0) as rd:
        assert rd.video_names == sorted(names)                 # sorted folders, as glob order in the reference's lists
        order = {name: clips[names.index(name)] for name in names}
        assert rd.items == data.dataset_items([len(order[v]) for v in rd.video_names], 8, 1, 2)
        rd.prefetch(rd.items)                                   # more items than ring slots: the ring must recycle
        for vi, idx in rd.items:
            x = rd.load_frame(vi, idx)
            rgb = order[rd.video_names[vi]][idx]
            assert torch.equal(x, hip.frame_from_uint8(torch.from_numpy(rgb).to(dev))), (vi, idx)
        assert rd.stats["frames"] == len(rd.items) and rd.stats["sync_loads"] == 0
        x = rd.load_frame(1, 3)                                 # not announced: decoded on the spot
        assert rd.stats["sync_loads"] == 1 and x.shape == (1, 3, 64, 128)
        batches = gop.gop_batches([i for v, i in rd.items if v == 0], 8)
        assert [b[0] for b in batches] == [0, 8] and [b[-1] for b in batches] == [8, 16]


def test_cli_encode_decode_roundtrip_and_the_reference_container(dev, tmp_path):
    from vcamd import cli, data, lhbdc
    fx = load_fixture("lhbdc_codec_a.npz")
    seed = int(fx["seed"])
    paths = {}
    for k in ("ref_1", "ref_2", "current"):
        paths[k] = str(tmp_path / f"{k}.png")
        data.write_png(paths[k], fx[k])
    binp, outp = str(tmp_path / "bits_B.bin"), str(tmp_path / "decoded.png")
    blob = cli.main(["encode_B", "--ref_1", paths["ref_1"], "--ref_2", paths["ref_2"], "--current", paths["current"], "--bin", binp,
                     "--l", "1626", "--seeded", str(seed)])
    assert open(binp, "rb").read() == blob
    # the same bytes as the library calls on the same frames
    model = cli.load_b_model(1626, seeded=seed, device=dev)
    with torch.no_grad():
        xb, xa, xc = (lhbdc.process_frame(fx[k].astype(float), dev) for k in ("ref_1", "ref_2", "current"))
        mv_bits, res_bits = lhbdc.encode_B(model, xa, xc, xb)
    assert lhbdc.write_container(None, 1626, mv_bits, res_bits) == blob
    u8 = cli.main(["decode_B", "--ref_1", paths["ref_1"], "--ref_2", paths["ref_2"], "--bin", binp, "--out", outp, "--seeded", str(seed)])
    assert np.array_equal(data.read_png(outp), u8) and u8.shape == fx["current"].shape
    with torch.no_grad():
        dec = lhbdc.decode_B(xb, xa, model, *lhbdc.read_container(blob)[1:])
    assert np.array_equal(u8, lhbdc.float_to_uint8(dec[0].cpu().numpy())[:u8.shape[0], :u8.shape[1]])
    # the REFERENCE's bits_B.bin (written by LHBDC/encode_B.py on the same seeded checkpoint) through the command line
    refbin, refout = str(tmp_path / "ref_bits_B.bin"), str(tmp_path / "ref_decoded.png")
    open(refbin, "wb").write(fx["container"].tobytes())
    u8r = cli.main(["decode_B", "--ref_1", paths["ref_1"], "--ref_2", paths["ref_2"], "--bin", refbin, "--out", refout, "--seeded", str(seed)])
    diff = np.abs(u8r.astype(int) - fx["decoded_u8"].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
    # a header that does not belong to the frames is refused before anything is allocated from it
    bad = bytearray(fx["container"].tobytes())
    bad[4:8] = np.array([65535, 65535], dtype=np.uint16).tobytes()
    open(refbin, "wb").write(bytes(bad))
    with pytest.raises(Exception, match="latent shapes"):
        cli.main(["decode_B", "--ref_1", paths["ref_1"], "--ref_2", paths["ref_2"], "--bin", refbin, "--out", refout, "--seeded", str(seed)])


def test_cli_test_loop_over_png_folders(dev, tmp_path):
    from vcamd import cli, data
    clips = [_clip(17, 192, 256, 5), _clip(17, 192, 256, 6)]
    data.write_synthetic_sequences(str(tmp_path), clips)
    s = cli.main(["test", "--test_path", str(tmp_path), "--test_numbers", "2", "--i_qual", "3", "--seeded", "7", "--workers", "2"])
    assert s["overall"]["frames"] == 2 * 17 and np.isfinite(s["overall"]["psnr"]) and s["overall"]["bpp"] > 0
    # levels of the GOP-8 hierarchy: 4 | 2,6 | 1,3,5,7 per GOP (two GOPs per video) + intra frames at level -1
    assert {k: v["frames"] for k, v in s["per_level"].items()} == {-1: 6, 0: 4, 1: 8, 2: 16}
