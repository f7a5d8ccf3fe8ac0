#!/usr/bin/env python
"""Run-to-run and batch independence of the split-operand instances at full size: every image of a batch must come out with the
same bits as the image alone, launch after launch (counted waits, no atomics: any difference is a race in the DMA schedule).

    python tools/split_determinism.py [--reps R] [cin,cout,k,n,h,w ...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

SHAPES = ["32,64,7,4,1088,1920", "64,32,7,4,1088,1920", "32,16,7,4,1088,1920", "96,32,5,2,1088,1920", "192,64,5,2,544,960", "128,128,3,4,544,960",
          "32,64,7,4,544,960", "64,32,7,4,272,480", "32,64,7,3,136,240"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("shapes", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.set_fp32_mode("split")
    bad = 0
    for spec in (args.shapes or SHAPES):
        cin, cout, k, n, h, w = [int(v) for v in spec.split(",")]
        g = torch.Generator().manual_seed(5)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        pc = hip.PackedConv(wt, b, stride=1, device=dev)
        x = hip.T.empty(n, h, w, cin, dev)
        x.buf.normal_()
        xs = hip.split3(x)
        first = pc(xs, act=hip.ACT_RELU).buf.clone()
        diffs = []
        for r in range(args.reps):
            again = pc(xs, act=hip.ACT_RELU).buf
            diffs.append(int((again != first).sum()))
        alone = []
        for i in range(n):
            one = pc(xs.images(i, i + 1), act=hip.ACT_RELU).buf
            alone.append(int((one.reshape(-1) != first.reshape(n, -1)[i]).sum()))
        ok = not any(diffs) and not any(alone)
        bad += not ok
        print(f"k{k} {cin}->{cout} @{n}x{h}x{w}: values that differ launch to launch {diffs}, image alone against in the batch {alone} -> {'same bits' if ok else 'DIFFERENT'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
