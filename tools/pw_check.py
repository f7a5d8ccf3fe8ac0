#!/usr/bin/env python
"""A/B of the 1x1 kernels: general 32-wide configuration, VC_CFG_PW (registers, csrc/conv_pw.hip) and VC_CFG_PWS (LDS-DMA
rings, csrc/conv_pws.hip): bit-identity of the results and HIP-event timings, interleaved rounds in one process, per
tensor-type combination of the layer (fp32; fp16 path with fp32 / half input and output) with and without a residual.

    python tools/pw_check.py [--reps R] [--rounds N] [--modes ...] cin,cout,n,h,w [...]
Prints algorithmic TFLOP/s and GB/s (input + output + residual + weights, as the kernel table of bench.py counts them).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

DEFAULT = ["128,128,1,544,960", "128,128,4,136,240", "128,128,1,272,480", "64,64,1,544,960", "96,96,1,272,480", "32,32,1,544,960"]
# mode: (precision, half input, half output, residual)
MODES = {"f32": ("fp32", False, False, False), "f32+res": ("fp32", False, False, True),
         "h->h": ("fp16", True, True, False), "f->h": ("fp16", False, True, False), "h->f+res": ("fp16", True, False, True),
         "f->f+res": ("fp16", False, False, True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--modes", default=",".join(MODES))
    ap.add_argument("--cfgs", default="2,6,9")
    ap.add_argument("shapes", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfgs = [int(c) for c in args.cfgs.split(",")]
    bad = 0
    for spec in args.shapes:
        cin, cout, n, h, w = [int(v) for v in spec.split(",")]
        for mode in args.modes.split(","):
            prec, hin, hout, with_res = MODES[mode]
            hip.set_conv_precision(prec)
            g = torch.Generator().manual_seed(0)
            wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
            b = torch.randn(cout, generator=g) * 0.1
            pc = hip.PackedConv(wt, b, device=dev)
            hip.set_conv_precision("fp32")
            x = hip.T.empty(n, h, w, cin, dev, "f16" if hin else "f32")
            x.buf.normal_()
            res = None
            if with_res:
                res = hip.T.empty(n, h, w, cout, dev)
                res.buf.normal_()
            out = hip.T.empty(n, h, w, cout, dev, "f16" if hout else "f32")
            fl = (hip.CFG_F16 if prec == "fp16" else 0) | (hip.CFG_IN_F16 if hin else 0) | (hip.CFG_OUT_F16 if hout else 0)
            act = hip.ACT_NONE if with_res else hip.ACT_RELU
            outs, times = {}, {c: [] for c in cfgs}
            use = []
            for cfg in cfgs:
                base = cfg if cfg > 2 else min(max(pc.cfg, 0), 2)
                pc.tuned = {(n, h, w, fl): base | hip.CFG_EXACT | fl}
                try:
                    out.buf.zero_()
                    pc(x, out=out, act=act, res=res)
                except hip.VcError:
                    continue
                torch.cuda.synchronize()
                outs[cfg] = out.buf.clone()
                use.append(cfg)
            same = all(torch.equal(outs[use[0]], outs[c]) for c in use[1:])
            bad += not same
            for _ in range(args.rounds):
                for cfg in use:
                    base = cfg if cfg > 2 else min(max(pc.cfg, 0), 2)
                    pc.tuned = {(n, h, w, fl): base | hip.CFG_EXACT | fl}
                    pc(x, out=out, act=act, res=res)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.reps):
                        pc(x, out=out, act=act, res=res)
                    e1.record()
                    torch.cuda.synchronize()
                    times[cfg].append(e0.elapsed_time(e1) / args.reps)
            px = n * h * w
            flop = 2.0 * px * cin * cout
            nbytes = px * (cin * (2 if hin else 4) + cout * (2 if hout else 4) + (cout * 4 if with_res else 0))
            cells = "  ".join(f"cfg{c}: {min(times[c]) * 1e3:7.1f} us {flop / min(times[c]) / 1e9:6.1f} TF/s {nbytes / min(times[c]) / 1e6:5.0f} GB/s"
                              for c in use)
            print(f"k1 {cin:3d}->{cout:3d} @{n}x{h}x{w} {mode:9s} identical={same}  {cells}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
