"""-m gpu: bench.py's multi-rank path, functionally, on a box with ONE GPU.

`python bench.py --gpus 2` started plainly must launch its own ranks as child processes and print rank 0's line.  RCCL
refuses two ranks on one device, so the check runs with VC_BENCH_SHARE_GPU=1 (ranks mapped onto the visible device) and
VC_BENCH_BACKEND=gloo (the R-D gather through host memory): same launcher, same sharding, same gather code -- never a
measurement.  BASELINE configs[3] property: the gathered RdTable of the sharded run equals the single-rank table exactly."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong", "--sequences", "2", "--frames-per-sequence", "17",
           "--steps", "1", "--warmup", "0", "--gops-per-step", "2", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_self_launched_two_rank_strong_run_equals_single_rank():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    one = _bench([])
    two = _bench(["--gpus", "2"], {"VC_BENCH_SHARE_GPU": "1", "VC_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong" and two.get("gloo_ranks") == 2
    assert one["quality"]["frames"] == two["quality"]["frames"] == 34
    assert one["quality"] == two["quality"]                    # bpp, PSNR and the per-frame-type table: identical
    assert two["config"]["parallelism"] == "gop-shard x2"
    # the multi-rank line explains itself: per-rank work, timings of the coding loop and of the R-D gather
    r = two["ranks"]
    assert r["world"] == 2 and r["backend"] == "gloo" and len(r["coded_s"]) == len(r["gather_s"]) == len(r["elapsed_s"]) == 2
    # 2 sequences x 2 GOPs: each rank codes one sequence (17 frames); a shard that started inside a sequence would add one I-frame
    assert r["frames_per_step_per_rank"] == [17, 17] and sum(r["records_last_step"]) == 34
    assert 0 < r["coded_s_min"] <= r["coded_s_max"] <= two["ms_per_step"] / 1000.0 * two["steps"] + 1e-6
    assert "ranks" not in one


def test_default_line_carries_the_strong_block_and_its_checksum_is_the_single_rank_one():
    """The DEFAULT (weak) line of `bench.py --gpus N` also holds BASELINE configs[3] as a `strong` block: one pass over a test
    set sized by --strong-seconds at N GPUs, GOP-sharded, R-D records gathered, per-rank coding times, and a checksum over the
    records of the frames every world size codes -- which must equal the single-rank run's, bit pattern for bit pattern."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    args = ["--sequences", "2", "--strong-seconds", "1", "--steps", "1", "--warmup", "0", "--gops-per-step", "1", "--no-cpu-baseline"]

    def run(extra, env_extra=None):
        env = dict(os.environ)
        env.update(env_extra or {})
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + extra, env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    one = run([])
    two = run(["--gpus", "2"], {"VC_BENCH_SHARE_GPU": "1", "VC_BENCH_BACKEND": "gloo"})
    assert one["scaling"] == two["scaling"] == "weak" and two["n_gpus"] == 2
    s1, s2 = one["strong"], two["strong"]
    # 17 frames/s x 1 s / 2 sequences -> 9 frames per sequence on one GPU, 17 on two
    assert s1["frames"] == 18 and s2["frames"] == 34 and s2["gloo_ranks"] == 2 and s2["frames_per_rank"] == [17, 17]
    assert len(s2["coded_s"]) == len(s2["gather_s"]) == 2 and all(t > 0 for t in s2["coded_s"])
    assert s1["rd_checksum_common"]["records"] == s2["rd_checksum_common"]["records"] == 18
    assert s1["rd_checksum_common"]["sha256"] == s2["rd_checksum_common"]["sha256"] == s1["rd_checksum"]["sha256"]
    assert s2["rd_checksum"]["records"] == 34 and s2["rd_checksum"]["sha256"] != s1["rd_checksum"]["sha256"]
    assert s1["value"] > 0 and s2["quality"]["frames"] == 34


def test_strong_run_sized_by_seconds_per_step():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong", "--sequences", "2", "--seconds-per-step", "1.5",
           "--steps", "1", "--warmup", "0", "--gops-per-step", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    # 17 frames/s x 1.5 s / 2 sequences = 12.75 frames -> the nearest whole number of GOP-8s: 9 frames per sequence
    assert line["quality"]["frames"] == 18 and line["config"]["frames_per_step"] == 18


def test_bench_streams_frames_from_png_files(tmp_path):
    """--data: the frames of every step come from PNG files through vcamd.data.SequenceReader; same R-D numbers as the
    device-resident run (the files hold exactly the synthetic clip), ingest statistics in the line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--gops-per-step", "1", "--no-cpu-baseline",
            "--no-strong-block"]

    def run(extra):
        out = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    resident = run([])
    streamed = run(["--data", str(tmp_path), "--data-workers", "4"])
    assert streamed["ingest"]["frames_per_step"] == 9 and streamed["ingest"]["frames_loaded"] >= 27
    assert streamed["quality"]["b_frames"] == resident["quality"]["b_frames"] == 7
    assert streamed["quality"]["bpp_estimated"] == resident["quality"]["bpp_estimated"]
    assert streamed["quality"]["psnr_db"] == resident["quality"]["psnr_db"]
