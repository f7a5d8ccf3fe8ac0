"""-m gpu: two processes sharing ONE device must each get the same bits run after run.

Production is one process per GPU, but rocprofv3, a monitoring agent or a second tenant are "a second process" too.  Round 5 met an
intermittent difference that only a second process on the device brought out (DESIGN section 5f has the round-6 diagnosis: lanes
48-63 of single waves, second flow component only, in a packed-fp32 form of the SPyNet level-input kernel that the library no longer
ships; tools/r06.sh li-diag + the li_diag make target reproduce it).  This test keeps the shipped kernels under that condition:
  * tools/spynet_determinism.py: SPyNet alone, 40 runs per process, every stage of every level compared;
  * tools/forward_determinism.py: the whole B-frame (mask U-Net with its 3-D-grid split-tensor kernels: vc_split3, up-sampling and
    pooling on split tensors; both codecs; blend), 40 runs per process.
The children are started as ordinary child processes (never an exec from a process that has touched the GPU).
"""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(tool, reps):
    cmd = [sys.executable, os.path.join(ROOT, "tools", tool), str(reps)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT) for _ in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
        assert p.returncode == 0, out[-2000:]
    return outs


@pytest.mark.parametrize("tool", ["spynet_determinism.py", "forward_determinism.py"])
def test_two_processes_on_one_device_repeat_their_bits(tool):
    if torch.cuda.device_count() < 1:         # (counting devices does not initialise the GPU in this process)
        pytest.skip("no GPU")
    reps = 40
    for out in _pair(tool, reps):
        m = re.search(r"(\d+) of (\d+) runs differ from the first", out)
        assert m, out[-2000:]
        print(f"{tool}: {m.group(0)} (two processes on one device)")
        assert int(m.group(2)) == reps and int(m.group(1)) == 0, out[-3000:]
