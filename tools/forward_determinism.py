#!/usr/bin/env python
"""The whole LHBDC B-frame (Model.forward_device: SPyNet, pooling / reflection pad, both codecs, mask U-Net with its split-tensor
up-sampling / pooling / vc_split3 kernels, blend), launch after launch: reconstruction and bit counts must be the same bits every time.
Run two instances at once to share the GPU (tests/test_two_process_gpu.py does).
    python tools/forward_determinism.py [reps] [--native]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip, lhbdc  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
    if "--native" in sys.argv:
        hip.set_fp32_mode("native")
    dev = torch.device("cuda:0")
    from vcamd.seeding import calibrated_state_dict
    m = lhbdc.Model()
    m.load_state_dict(calibrated_state_dict(m.state_dict(), seed=1234))
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    H, W = 1088, 1920
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, H + 8, W + 16, generator=g), 9, 1, padding=4)
    xb, xc, xa = (base[..., :H, o:W + o].contiguous().to(dev) for o in (0, 3, 6))

    def run():
        with torch.no_grad():
            trace = {}
            x_hat, tot = m.forward_device(xb, xc, xa, trace=trace)
            out = {"x_hat": x_hat.clone(), "bits": tot.clone(), "mask": trace["mask"].buf.clone(), "flows": trace["flows"].buf.clone(),
                   "resid": trace["resid"].buf.clone()}
        torch.cuda.synchronize()
        return out

    ref = run()
    nbad = 0
    for r in range(reps):
        cur = run()
        bad = [(k, int((cur[k] != ref[k]).sum())) for k in ref if bool((cur[k] != ref[k]).any())]
        nbad += bool(bad)
        if bad:
            print(f"run {r}: differs from the first in {bad}", flush=True)
    print(f"{nbad} of {reps} runs differ from the first", flush=True)


if __name__ == "__main__":
    main()
