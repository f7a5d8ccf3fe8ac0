#!/usr/bin/env python
"""Micro-benchmark of the convolution engine on chosen layer shapes (HIP events on the launch stream).

    python tools/conv_bench.py [--reps R] cin,cout,k,stride,n,h,w [...]
Prints achieved algorithmic TFLOP/s per shape; used for A/B work on csrc/conv_mfma.h and as the target of
`rocprofv3 --pmc` passes (see profiles/README.md)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

DEFAULT = ["128,128,3,1,1,544,960", "64,32,7,1,4,1088,1920", "32,64,7,1,4,1088,1920", "128,512,3,1,1,272,480",
           "192,64,5,1,1,544,960", "128,128,3,1,1,136,240", "128,128,3,1,1,40,64", "16,2,7,1,4,1088,1920",
           "128,128,1,1,1,544,960"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--precision", choices=["fp32", "fp16"], default="fp32")
    ap.add_argument("--half-io", action="store_true",
                    help="fp16 path: half-precision input AND output tensors (VC_CFG_IN_F16 / OUT_F16), as inside a chain of "
                         "fp16-path layers; default is fp32 tensors either side (the first / last layer of a chain)")
    ap.add_argument("--split-out", action="store_true", help="with --split: the output as a split tensor too (what the layer writes inside the "
                                                             "models when its consumer is a split layer: 6 instead of 4 bytes per element)")
    ap.add_argument("--split", action="store_true",
                    help="fp32 layers on the split-operand pipeline (VC_CFG_SPLIT, csrc/conv_split.h); the input is converted to a split "
                         "tensor ONCE before the timed launches (inside a chain the producing epilogue writes it)")
    ap.add_argument("--residual", action="store_true", help="add an fp32 residual tensor in the epilogue (bottleneck blocks)")
    ap.add_argument("--residual-split", action="store_true",
                    help="with --split: add a SPLIT-tensor residual in the epilogue (conv2 of a residual block inside a split chain)")
    ap.add_argument("--residual-half", action="store_true",
                    help="fp16 path with --half-io: add a HALF-precision residual (VC_CFG_RES_F16: the identity of a residual block)")
    ap.add_argument("shapes", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.set_conv_precision(args.precision)
    hip.set_fp32_mode("split" if args.split else "native")
    for spec in args.shapes:
        fields = [int(v) for v in spec.split(",")]
        cin, cout, k, s, n, h, w = fields[:7]
        g = torch.Generator().manual_seed(0)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        pc = hip.PackedConv(wt, b, stride=s, device=dev)
        if len(fields) > 7:            # optional 8th field: force a (layout-compatible) narrower tile config
            f16 = pc.wpk16 is not None
            fl = hip.CFG_F16 if f16 else 0
            if args.half_io:
                fl |= hip.CFG_IN_F16 | hip.CFG_OUT_F16
            pc.tuned = {(n, h, w, fl): fields[7] | hip.CFG_EXACT | fl}
        io = "f16" if (args.half_io and args.precision == "fp16") else "f32"
        x = hip.T.empty(n, h, w, cin, dev, io)
        x.buf.normal_()
        ho, wo, co = pc.out_shape(h, w)
        out = hip.T.empty(n, ho, wo, co, dev, io)
        res = None
        if args.residual:
            res = hip.T.empty(n, ho, wo, co, dev)
            res.buf.normal_()
        if args.residual_half:
            res = hip.T.empty(n, ho, wo, co, dev, "f16")
            res.buf.normal_()
        if args.residual_split and args.split and co % 8 == 0:
            res = hip.T.empty(n, ho, wo, co, dev, "sp3")
            res.buf.zero_()
        if args.split:
            if not pc.split_ok:
                print(f"conv k{k} s{s} {cin}->{cout}: no split-operand instance")
                continue
            x = hip.split3(x)
            if args.split_out and co % 8 == 0:
                out = hip.T.empty(n, ho, wo, co, dev, "sp3")
        for _ in range(2):
            pc(x, out=out, act=hip.ACT_LRELU, res=res)
        stamps = hasattr(hip.lib(), "vc_debug_dma_stamps") and "64" in (os.environ.get("VC_DMA_VARIANT"), os.environ.get("VC_SPLIT_VARIANT"))
        if stamps:          # diagnostic library (make dma_diag): shader-clock totals per segment of the LDS-DMA kernel
            import ctypes
            import numpy as np
            buf = np.zeros(8, dtype=np.uint64)
            hip.lib().vc_debug_dma_stamps.argtypes = [ctypes.c_void_p]
            hip.lib().vc_debug_dma_stamps(buf.ctypes.data)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            pc(x, out=out, act=hip.ACT_LRELU, res=res)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        if stamps:
            hip.lib().vc_debug_dma_stamps(buf.ctypes.data)
            tot = float(buf[7]) or 1.0
            names = ["DMA issue", "fragment reads", "vmcnt wait", "barrier after R", "MFMA issue", "barrier after M", "epilogue", "wave lifetime"]
            print("   stamps: " + "; ".join(f"{nm} {100.0 * float(v) / tot:.1f} %" for nm, v in zip(names, buf)))
        flop = 2.0 * n * ho * wo * cout * cin * k * k
        print(f"conv k{k} s{s} {cin:4d}->{cout:4d} @{n}x{h}x{w} cfg{[v for v in pc.tuned.values()][0] & 0xff if pc.tuned else pc.cfg} {args.precision}: {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s "
              f"({flop / ms / 1e9 / 157.3 * 100:5.1f}% of fp32 MFMA peak)")


if __name__ == "__main__":
    main()
