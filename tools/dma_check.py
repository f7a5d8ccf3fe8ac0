#!/usr/bin/env python
"""A/B of the fp16-path LDS-DMA convolution pipeline (VC_CFG_DMA, csrc/conv_dma.h) against the classic fp16 instances:
bit-identity of the results and HIP-event timings, interleaved rounds in one process.

    python tools/dma_check.py [--reps R] [--rounds N] cin,cout,k,n,h,w,classic_cfg [...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

DEFAULT = ["128,128,3,1,40,72,5", "128,128,3,2,37,50,5", "64,32,7,2,50,70,7", "32,64,7,1,34,60,1", "64,128,3,1,33,31,5",
           "128,128,3,4,544,960,5", "64,32,7,4,1088,1920,7", "32,64,7,4,1088,1920,1", "128,128,3,1,544,960,5"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--f32-out", action="store_true", help="fp32 output tensor instead of half")
    ap.add_argument("shapes", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.set_conv_precision("fp16")
    bad = 0
    for spec in args.shapes:
        cin, cout, k, n, h, w, ccfg = [int(v) for v in spec.split(",")]
        g = torch.Generator().manual_seed(0)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        pc = hip.PackedConv(wt, b, stride=1, device=dev)
        x = hip.T.empty(n, h, w, cin, dev, "f16")
        x.buf.normal_()
        ho, wo, co = pc.out_shape(h, w)
        res = hip.T.empty(n, ho, wo, co, dev)
        res.buf.normal_()
        fl = hip.CFG_F16 | hip.CFG_IN_F16
        outs, times = {}, {}
        for cfg in (ccfg, hip.CFG_DMA):
            pc.tuned = {(n, h, w, fl | hip.CFG_OUT_F16): cfg | hip.CFG_EXACT | fl | hip.CFG_OUT_F16, (n, h, w, fl): cfg | hip.CFG_EXACT | fl}
            o1 = pc(x, act=hip.ACT_LRELU, out_f16=True)
            o2 = pc(x, act=hip.ACT_RELU, res=res)
            o3 = pc(x, act=hip.ACT_NONE, res=res, res_first=True, out_f16=True)
            torch.cuda.synchronize()
            outs[cfg] = [o1.buf.clone(), o2.buf.clone(), o3.buf.clone()]
        same = all(torch.equal(a, c) for a, c in zip(outs[ccfg], outs[hip.CFG_DMA]))
        maxd = max((a.float() - c.float()).abs().max().item() for a, c in zip(outs[ccfg], outs[hip.CFG_DMA]))
        bad += not same
        io = "f32" if args.f32_out else "f16"
        out = hip.T.empty(n, ho, wo, co, dev, io)
        for cfg in (ccfg, hip.CFG_DMA):
            times[cfg] = []
        for _ in range(args.rounds):
            for cfg in (ccfg, hip.CFG_DMA):
                of = hip.CFG_OUT_F16 if io == "f16" else 0
                pc.tuned = {(n, h, w, fl | of): cfg | hip.CFG_EXACT | fl | of}
                pc(x, out=out, act=hip.ACT_LRELU)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    pc(x, out=out, act=hip.ACT_LRELU)
                e1.record()
                torch.cuda.synchronize()
                times[cfg].append(e0.elapsed_time(e1) / args.reps)
        if hasattr(hip.lib(), "vc_debug_dma_stamps") and os.environ.get("VC_DMA_VARIANT") in ("64", "192", "320", "832", "72", "68", "76", "96", "65"):
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            hip.lib().vc_debug_dma_stamps.argtypes = [ctypes.c_void_p]
            hip.lib().vc_debug_dma_stamps(buf)
            pc.tuned = {(n, h, w, fl | of): hip.CFG_DMA | hip.CFG_EXACT | fl | of}
            pc(x, out=out, act=hip.ACT_LRELU)
            hip.lib().vc_debug_dma_stamps(buf)
            names = ["DMA issue", "fragment reads (issue + return)", "vmcnt wait", "barrier after R", "MFMA issue", "barrier after M",
                     "epilogue", "wave lifetime"]
            waves = 256 * 8
            for i, nm in enumerate(names):
                print(f"   {nm:34s} {buf[i] / waves:12.0f} cycles/wave  {100.0 * buf[i] / max(1, buf[7]):5.1f}% of lifetime")
        flop = 2.0 * n * ho * wo * cout * cin * k * k
        tc, td = min(times[ccfg]), min(times[hip.CFG_DMA])
        print(f"conv k{k} {cin:4d}->{cout:4d} @{n}x{h}x{w}: bit-identical={same} (max|d|={maxd:.3g})  classic cfg{ccfg} {tc:7.3f} ms "
              f"{flop / tc / 1e9:7.1f} TF/s | dma {td:7.3f} ms {flop / td / 1e9:7.1f} TF/s  ({tc / td:4.2f}x)  rounds: "
              f"{['%.3f/%.3f' % (a, c) for a, c in zip(times[ccfg], times[hip.CFG_DMA])]}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
