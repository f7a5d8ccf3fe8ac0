"""Micro-benchmark of the fp16-path deformable fusion (csrc/deform.hip): pixel-interleaved half features (vc_offset_diversity_hx)
against group-planar ones (vc_offset_diversity_hxp), for raw offsets of a given spread -- 0: every tap follows the flow alone
(neighbouring pixels sample neighbouring positions), 1: unit-variance raw offsets (tanh * magnitude scatters the taps over +-magnitude px).
    python tools/deform_layout_bench.py [--h 1088 --w 1920 --c 64 --mag 40 --reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-compression_amd"))
from vcamd import hip, icip2024                                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1088)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--c", type=int, default=64)
    ap.add_argument("--mag", type=float, default=40)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    prod = icip2024.OffsetDiversity(args.c, args.mag)
    with torch.no_grad():
        prod.fusion.weight.copy_(torch.randn(prod.fusion.weight.shape, generator=g) * 0.2)
        prod.fusion.bias.copy_(torch.randn(args.c, generator=g))
    prod = prod.to(dev)
    n, h, w, c = 1, args.h, args.w, args.c
    x1, x2 = (hip.nchw_to_nhwc(torch.randn(n, c, h, w, generator=g).to(dev)) for _ in range(2))
    f1, f2 = (hip.nchw_to_nhwc((torch.randn(n, 2, 1, 1, generator=g) * 3).expand(n, 2, h, w).contiguous().to(dev)) for _ in range(2))
    hip.set_conv_precision("fp16")
    for spread in (0.0, 0.05, 0.3, 1.0):
        o1, o2 = (hip.nchw_to_nhwc((torch.randn(n, 216, h, w, generator=g) * spread).to(dev)) for _ in range(2))
        outs = {}
        for planar in (False, True):
            hip.HALF_DEFORM_PLANAR = planar
            try:
                for _ in range(3):
                    out = prod.run(x1, o1, f1, x2, o2, f2)
            except hip.VcError as e:                       # (an A/B library without the planar entry point)
                print(f"raw spread {spread:4.2f}  planar: {e}")
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                out = prod.run(x1, o1, f1, x2, o2, f2)
            e1.record()
            torch.cuda.synchronize()
            outs[planar] = hip.nhwc_to_nchw(out).clone()
            print(f"raw spread {spread:4.2f}  {'planar     ' if planar else 'interleaved'}  {e0.elapsed_time(e1) / args.reps * 1e3:8.1f} us per call (half copies included)")
        if True in outs:
            d = (outs[True] - outs[False]).abs().max().item()
            print(f"    planar vs interleaved: max|d| = {d:.3e}")
    hip.set_conv_precision("fp32")


if __name__ == "__main__":
    main()
