#!/usr/bin/env python
"""Split-operand fp32 convolution (VC_CFG_SPLIT, csrc/conv_split.h) against the native fp32 instances: errors of both against an
fp64 CPU reference on small shapes, then launch times on the layer shapes of the headline path.

    python tools/split_check.py [--reps R] [--no-accuracy] [cin,cout,k,n,h,w ...]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

SMALL = ["32,64,7,1,48,80", "64,32,7,2,37,53", "8,32,7,1,40,72", "96,32,5,1,48,80", "192,64,5,1,33,47",
         "128,128,3,1,60,96", "64,128,3,2,50,70", "128,512,3,1,48,64,1", "16,64,3,1,24,33"]
BIG = ["32,16,7,4,1088,1920", "32,64,7,4,1088,1920", "64,32,7,4,1088,1920", "128,128,3,1,544,960", "128,128,3,4,544,960", "128,512,3,1,272,480,1",
       "128,128,3,1,272,480", "256,128,3,1,272,480", "128,128,3,1,136,240", "96,32,5,1,1088,1920"]


def run(pc, x, mode, act, res=None, out_sp3=False):
    hip.set_fp32_mode(mode)
    return pc(x, act=act, slope=0.1, res=res, out_sp3=out_sp3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("shapes", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if not args.no_accuracy:
        for spec in (args.shapes or SMALL):
            cin, cout, k, n, h, w = [int(v) for v in spec.split(",")][:6]
            ps = spec.count(",") > 5
            g = torch.Generator().manual_seed(1)
            wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
            b = torch.randn(cout, generator=g) * 0.1
            xc = torch.randn(n, cin, h, w, generator=g)
            rc = torch.randn(n, cout // 4, 2 * h, 2 * w, generator=g) if ps else torch.randn(n, cout, h, w, generator=g)
            pc = hip.PackedConv(wt, b, stride=1, pixelshuffle=ps, device=dev)
            x = hip.nchw_to_nhwc(xc.to(dev))
            res = hip.nchw_to_nhwc(rc.to(dev))
            ref = F.leaky_relu(F.conv2d(xc.double(), wt.double(), b.double(), padding=k // 2), 0.1)
            mag = F.conv2d(xc.double().abs(), wt.double().abs(), b.double().abs(), padding=k // 2)
            if ps:
                ref, mag = F.pixel_shuffle(ref, 2), F.pixel_shuffle(mag, 2)
            ref, mag = ref + rc.double(), mag + rc.double().abs()
            outs = {}
            for mode in ("native", "split"):
                y = hip.nhwc_to_nchw(run(pc, x, mode, hip.ACT_LRELU, res=res)).cpu().double()
                outs[mode] = y
                err = ((y - ref).abs() / mag)
                print(f"k{k} {cin:3d}->{cout:3d} @{n}x{h}x{w} {mode:6s}: max |err| / sum|a b| = {err.max():.3e}  rms = {err.pow(2).mean().sqrt():.3e}")
            d = (outs["native"] - outs["split"]).abs().max().item()
            print(f"      native vs split max |d| = {d:.3e}")
            # a chain through a split intermediate: conv -> (split tensor) -> conv, against the same chain through fp32
            if cout % 8 == 0 and not ps:
                wt2 = torch.randn(64, cout, k, k, generator=g) / (cout * k * k) ** 0.5
                pc2 = hip.PackedConv(wt2, None, stride=1, device=dev)
                hip.set_fp32_mode("split")
                mid_sp = pc(x, act=hip.ACT_RELU, out_sp3=True)
                y_chain = hip.nhwc_to_nchw(pc2(mid_sp)).cpu()
                mid = pc(x, act=hip.ACT_RELU)
                y_two = hip.nhwc_to_nchw(pc2(mid)).cpu()
                print(f"      chain through a split intermediate == chain through fp32 + vc_split3: {torch.equal(y_chain, y_two)} (max |d| {float((y_chain - y_two).abs().max()):.2e})")
    if args.no_timing:
        return
    for spec in (args.shapes or BIG):
        cin, cout, k, n, h, w = [int(v) for v in spec.split(",")][:6]
        ps = spec.count(",") > 5
        g = torch.Generator().manual_seed(0)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        pc = hip.PackedConv(wt, b, stride=1, pixelshuffle=ps, device=dev)
        x = hip.T.empty(n, h, w, cin, dev)
        x.buf.normal_()
        flop = 2.0 * n * h * w * cout * cin * k * k
        line = f"conv k{k} s1 {cin:4d}->{cout:4d} @{n}x{h}x{w}:"
        hip.set_fp32_mode("native")
        out = hip.T.empty(n, 2 * h, 2 * w, cout // 4, dev) if ps else hip.T.empty(n, h, w, cout, dev)
        variants = [("native", lambda: pc(x, out=out, act=hip.ACT_RELU))]
        hip.set_fp32_mode("split")
        xs = hip.split3(x, c_out=pc.cin_split)
        variants.append(("split (input already split)", lambda: pc(xs, out=out, act=hip.ACT_RELU)))
        variants.append(("split + vc_split3 of the input", lambda: pc(hip.split3(x, out=xs, c_out=pc.cin_split), out=out, act=hip.ACT_RELU)))
        if cout % 32 == 0:
            osp = hip.T.empty(n, 2 * h, 2 * w, cout // 4, dev, "sp3") if ps else hip.T.empty(n, h, w, cout, dev, "sp3")
            variants.append(("split, split output", lambda: pc(xs, out=osp, act=hip.ACT_RELU)))
        for name, fn in variants:
            hip.set_fp32_mode("native" if name == "native" else "split")
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            line += f"\n    {name:34s} {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s fp32-equivalent ({flop / ms / 1e9 / 157.3 * 100:5.1f} % of the native fp32 peak)"
        print(line)


if __name__ == "__main__":
    main()
