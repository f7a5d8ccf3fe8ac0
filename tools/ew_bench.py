#!/usr/bin/env python
"""Launch times of the HBM-bound kernels between the split-operand convolutions of the headline frame (1080p shapes): bilinear
up-sampling into a split tensor, the SPyNet level input, 2x2 max pooling, vc_split3 -- against their algorithmic bytes.

    python tools/ew_bench.py [--reps R]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = hip.lib()
    rows = []
    for n, h, w, c in [(1, 544, 960, 64), (1, 272, 480, 128), (1, 136, 240, 256), (1, 544, 960, 32)]:
        x = hip.T.empty(n, h, w, c, dev)
        x.buf.normal_()
        out = hip.T.empty(n, 2 * h, 2 * w, c, dev, "sp3")
        mb = n * c * (4.0 * h * w + 6.0 * 4 * h * w) / 1e6
        us = timeit(lambda: hip.check(L.vc_upsample_bilinear_sp3(hip.stream(), x.view(), out.ptr, out.image_bytes, 2, 0, 1.0), "up"), args.reps)
        rows.append((f"k_upsample_bilinear_sp3 x2 c{c} @{n}x{h}x{w}", mb, us))
    for n, h, w, c in [(1, 1088, 1920, 32), (1, 544, 960, 64), (1, 272, 480, 128)]:
        x = hip.T.empty(n, h, w, c, dev, "sp3")
        x.buf.zero_()
        out = hip.T.empty(n, h // 2, w // 2, c, dev, "sp3")
        mb = 6.0 * n * c * (h * w + h * w / 4) / 1e6
        us = timeit(lambda: hip.check(L.vc_maxpool2_sp3(hip.stream(), x.ptr, x.image_bytes, n, h, w, c, out.ptr, out.image_bytes), "mp"), args.reps)
        rows.append((f"k_maxpool2_sp3 c{c} @{n}x{h}x{w}", mb, us))
    for n, h, w in [(4, 1088, 1920), (4, 544, 960), (4, 272, 480)]:
        f1, f2 = hip.T.empty(n, h, w, 3, dev), hip.T.empty(n, h, w, 3, dev)
        f1.buf.uniform_()
        f2.buf.uniform_()
        fc = hip.T.empty(n, h // 2, w // 2, 2, dev)
        fc.buf.normal_()
        fc.buf.mul_(0.25)                   # (displacements of a fraction of a pixel after the x2: the gathers stay near the diagonal, as in the model)
        feat = hip.T.empty(n, h, w, 8, dev, "sp3")
        up = hip.T.empty(n, h, w, 2, dev)
        mb = n * h * w * (4.0 * (3 + 3 + 0.5 + 2) + 48.0) / 1e6
        us = timeit(lambda: hip.check(L.vc_spynet_level_input_sp3(hip.stream(), f1.view(), f2.view(), fc.view(), feat.ptr, up.view()), "li"), args.reps)
        rows.append((f"k_spynet_level_input (split) @{n}x{h}x{w}", mb, us))
    for n, h, w, c in [(1, 544, 960, 128), (1, 1088, 1920, 32)]:
        x = hip.T.empty(n, h, w, c, dev)
        x.buf.normal_()
        out = hip.T.empty(n, h, w, c, dev, "sp3")
        mb = 10.0 * n * c * h * w / 1e6
        us = timeit(lambda: hip.split3(x, out=out), args.reps)
        rows.append((f"vc_split3 c{c} @{n}x{h}x{w}", mb, us))
    for name, mb, us in rows:
        print(f"{name:48s} {us:8.1f} us  {mb:8.1f} MB  {mb / us * 1e3:7.1f} GB/s  ({mb / us * 1e3 / 8000 * 100:4.1f} % of the HBM peak)")


if __name__ == "__main__":
    main()
