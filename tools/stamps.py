#!/usr/bin/env python
"""Where does a convolution wave spend its life?  Diagnostic companion of `make -C video-compression_amd/csrc stamps`.

    VC_HIP_LIB=video-compression_amd/libvc_hip_stamps.so python tools/stamps.py [cin,cout,k,stride,n,h,w[,cfg] ...]
The stamps build accumulates shader-clock (s_memtime) intervals per wave around the phases of conv_mfma_kernel for the
fp32 3x3 and 7x7 instances; this script runs each shape a few times and prints the share of every phase in the waves'
lifetime.  Intervals are wall-clock per wave: a phase looks long both when it does much work and when the wave waits
for a co-resident wave that holds the matrix pipe -- read them together with the achieved TFLOP/s."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

PHASES = ["barrier-before-stage", "staging", "barrier-after-stage", "mfma loop", "epilogue", "lifetime"]


def main():
    argv = sys.argv[1:]
    f16 = "--fp16" in argv          # with libvc_hip_stamps16.so (`make stamps16`): the fp16-path instances
    knockout = "--knockout" in argv # time the kernel with phases switched off (garbage results, diagnostic only)
    half_io = "--half-io" in argv   # fp16 path: half-precision tensors either side
    argv = [a for a in argv if a not in ("--fp16", "--knockout", "--half-io")]
    if f16:
        hip.set_conv_precision("fp16")
    shapes = argv or ["64,32,7,1,4,1088,1920", "128,128,3,1,1,544,960,5", "128,128,3,1,1,544,960,1"]
    L = hip.lib()
    if not hasattr(L, "vc_debug_read_stamps"):
        raise SystemExit("this library has no stamps: build `make -C video-compression_amd/csrc stamps` and set VC_HIP_LIB")
    L.vc_debug_read_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    buf = (ctypes.c_ulonglong * 8)()
    for spec in shapes:
        f = [int(v) for v in spec.split(",")]
        cin, cout, k, s, n, h, w = f[:7]
        g = torch.Generator().manual_seed(0)
        pc = hip.PackedConv(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.zeros(cout), stride=s, device=dev)
        if len(f) > 7:
            fl = hip.CFG_F16 if f16 else 0
            pc.tuned = {(n, h, w, fl): f[7] | hip.CFG_EXACT | fl}
        io = "f16" if (f16 and half_io) else "f32"
        x = hip.T.empty(n, h, w, cin, dev, io)
        x.buf.normal_()
        out = hip.T.empty(n, *pc.out_shape(h, w), dev, io)
        pc(x, out=out)
        if knockout:
            L.vc_debug_set_skip.argtypes = [ctypes.c_int]
            names = {0: "everything", 1: "first chunk staged only", 2: "no contraction", 4: "no epilogue", 3: "epilogue only (+1 chunk)",
                     5: "contraction only (+1 chunk)", 6: "staging only"}
            print(f"conv k{k} {cin}->{cout} @{n}x{h}x{w} cfg={f[7] if len(f) > 7 else 'auto'} io={io}: phase knock-out")
            for mask in (0, 1, 2, 4, 3, 5, 6):
                L.vc_debug_set_skip(mask | 8)        # | 8: no stamps (their atomics would dominate the time)
                pc(x, out=out, act=hip.ACT_LRELU)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    pc(x, out=out, act=hip.ACT_LRELU)
                e1.record()
                torch.cuda.synchronize()
                print(f"   {names[mask]:30s} {e0.elapsed_time(e1) / 10:8.3f} ms")
            L.vc_debug_set_skip(0)
            L.vc_debug_read_stamps(buf)
            continue
        L.vc_debug_read_stamps(buf)                      # discard warm-up / autotune launches
        reps = 5
        for _ in range(reps):
            pc(x, out=out, act=hip.ACT_LRELU)
        L.vc_debug_read_stamps(buf)
        waves = buf[6] / reps
        print(f"conv k{k} {cin}->{cout} @{n}x{h}x{w} cfg={f[7] if len(f) > 7 else 'auto'}: waves/launch={waves:.0f}")
        for i, name in enumerate(PHASES):
            print(f"   {name:22s} {buf[i] / buf[6]:12.0f} cycles/wave  {100.0 * buf[i] / buf[5]:5.1f}% of lifetime")


if __name__ == "__main__":
    main()
