#!/usr/bin/env python
"""Micro-benchmark of the fused OffsetDiversity kernel (vc_offset_diversity) at the three ICIP2024 levels.

    python tools/deform_bench.py [--reps R] [--offset-std S] [--batch N]
Offsets are tanh(raw)*magnitude + flow with raw ~ N(0, S^2): S = 0.05 resembles a trained offset head (small
refinements around the flow), S = 1 saturates the tanh (random +-magnitude gathers, the worst case)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--offset-std", type=float, default=0.05)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for c, mag, h, w in ((64, 40, 544, 960), (96, 20, 272, 480), (128, 10, 136, 240)):
        n = args.batch
        pk = hip.PackedDeform(torch.randn(c, 2 * c // 16, 3, 3, generator=g) * 0.1, torch.zeros(c), 16, dev)
        x1, x2 = hip.T.empty(n, h, w, c, dev), hip.T.empty(n, h, w, c, dev)
        raw = hip.T.empty(n, h, w, 432, dev)
        f1, f2 = hip.T.empty(n, h, w, 2, dev), hip.T.empty(n, h, w, 2, dev)
        for t, s in ((x1, 1.0), (x2, 1.0), (raw, args.offset_std), (f1, 2.0), (f2, 2.0)):
            t.buf.normal_(0.0, s)
        out = hip.T.empty(n, h, w, c, dev)
        run = lambda: pk.offset_diversity(x1, raw.channels(0, 216), f1, x2, raw.channels(216, 432), f2, mag, out=out)  # noqa: E731
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        gb = n * h * w * (432 + 2 * c + 4 + c) * 4 / 1e9
        print(f"deform C={c} @{n}x{h}x{w} offset-std {args.offset_std}: {ms:7.3f} ms  compulsory traffic {gb:.2f} GB -> {gb / ms:.2f} TB/s")


if __name__ == "__main__":
    main()
