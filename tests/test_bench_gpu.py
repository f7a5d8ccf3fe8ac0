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
