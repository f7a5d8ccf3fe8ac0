"""File-backed ingest, host side (no GPU): item lists against the reference's own UVGTestDataset (fixture), natural frame
order, PNG round trip, the 4:2:0 conversion, the CLI's argument surface."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _frames(n, h=24, w=40, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]


def test_item_lists_of_png_folders_match_the_reference_dataset(tmp_path):
    """dataset_items over folders of N PNGs == the lists UVGTestDataset built in the reference run
    (tests/golden/uvg_dataset_indices.json, made by oracle/gen_golden.py from LHBDC/test/utils.py:162-203)."""
    from vcamd import data
    ref = json.load(open(os.path.join(GOLDEN, "uvg_dataset_indices.json")))
    made = {}
    for key, items in ref.items():
        _, n, g, t = key.split(":")
        n, g, t = int(n), int(g), int(t)
        if n not in made:
            made[n] = str(tmp_path / f"v{n}")
            data.write_synthetic_sequences(made[n], [_frames(n, 8, 8, n)], ["only"])
        folder = data.PngFolder(os.path.join(made[n], "only"))
        assert len(folder) == n
        got = data.dataset_items([len(folder)], g, 1, t)
        assert [i for _, i in got] == items, key
    # two videos: the second video's items follow the first's, each with its own boundary duplicates
    two = data.dataset_items([17, 25], 8, 1, 2)
    assert [i for v, i in two if v == 0] == ref["lhbdc:17:8:2"] and [i for v, i in two if v == 1] == ref["lhbdc:25:8:2"]
    # skip_frames strides the source frames: 33 frames, every 2nd -> 17 items' worth
    assert [i for _, i in data.dataset_items([33], 8, 2, 2)] == ref["lhbdc:17:8:2"]


def test_natural_order_and_png_roundtrip(tmp_path):
    from vcamd import data
    d = tmp_path / "seq"
    d.mkdir()
    fr = _frames(12)
    for i, f in enumerate(fr):
        data.write_png(str(d / f"im{i + 1}.png"), f)            # im1 ... im12: lexicographic order would put im10 before im2
    folder = data.PngFolder(str(d))
    assert [os.path.basename(p) for p in folder.paths] == [f"im{i + 1}.png" for i in range(12)]
    for i, f in enumerate(fr):
        assert np.array_equal(folder.read(i), f)
    out = np.empty((24, 40, 3), np.uint8)
    assert folder.read(3, out=out) is out and np.array_equal(out, fr[3])
    with pytest.raises(Exception):
        data.PngFolder(str(tmp_path / "missing"))


def test_yuv420_reader(tmp_path):
    from vcamd import data
    w, h, n = 16, 12, 3
    rng = np.random.default_rng(1)
    raw = rng.integers(16, 236, size=(n, w * h + 2 * (w // 2) * (h // 2)), dtype=np.uint8)
    path = tmp_path / "clip.yuv"
    raw.tofile(str(path))
    v = data.Yuv420File(str(path), w, h)
    assert len(v) == n and (v.h, v.w) == (h, w)
    rgb = v.read(1)
    assert rgb.shape == (h, w, 3) and rgb.dtype == np.uint8
    # grey: Y = 16 .. 235 with neutral chroma maps to 0 .. 255 on all three channels
    grey = np.concatenate([np.full(w * h, 235, np.uint8), np.full(2 * (w // 2) * (h // 2), 128, np.uint8)])
    assert (data.yuv420_frame_to_rgb(grey, w, h) == 255).all()
    grey[: w * h] = 16
    assert (data.yuv420_frame_to_rgb(grey, w, h) == 0).all()
    # a pixel by hand (BT.709 limited range)
    y, u, vv = float(raw[1, 0]), float(raw[1, w * h]), float(raw[1, w * h + (w // 2) * (h // 2)])
    r = (y - 16) * 255 / 219 + 255 / 224 * 1.5748 * (vv - 128)
    assert int(rgb[0, 0, 0]) == int(np.clip(np.rint(np.float32(r)), 0, 255)) or abs(int(rgb[0, 0, 0]) - np.clip(round(r), 0, 255)) <= 1


def test_cli_arguments_mirror_the_reference_scripts():
    from vcamd import cli
    p = cli.build_parser()
    e = p.parse_args(["encode_B"])
    # LHBDC/encode_B.py:21-28
    assert (e.ref_1, e.ref_2, e.current, e.bin, e.l) == ("frames/ref_1.png", "frames/ref_2.png", "frames/current.png", "bits_B.bin", 1626)
    with pytest.raises(SystemExit):
        p.parse_args(["encode_B", "--l", "1000"])                 # choices=[228, 436, 845, 1626, 3141]
    d = p.parse_args(["decode_B"])
    assert (d.ref_1, d.ref_2, d.bin) == ("frames/ref_1.png", "frames/ref_2.png", "bits_B.bin")      # decode_B.py:23-28
    t = p.parse_args(["test"])
    # LHBDC/test/testing.py:35-59
    assert (t.test_path, t.test_gop_size, t.i_interval, t.test_skip_frames, t.test_numbers, t.workers, t.i_qual, t.lmbda) == \
        ("/datasets/UVG/full_test/", 8, 8, 1, None, 4, 7, 1626)
    assert t.b_pretrained == "../new_compression_1626.pth"
