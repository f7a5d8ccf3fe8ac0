#!/usr/bin/env python
"""Which kernel writes into LDS it does not own?  A canary kernel (tools/micro/lds_canary.hip: 12 KiB of LDS per workgroup, pattern,
sleep, check) runs on one stream while a convolution of one kind loops on another: co-resident on the CUs.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/lds_canary.so tools/micro/lds_canary.hip     (once; *.so is not tracked)
    python tools/lds_canary.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    can = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "lds_canary.so"))
    can.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    g = torch.Generator().manual_seed(0)

    def conv(cin, cout, k, n, h, w, mode, half=False):
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        pc = hip.PackedConv(wt, None, stride=1, device=dev)
        x = hip.T.empty(n, h, w, cin, dev)
        x.buf.normal_()

        def run():
            hip.set_fp32_mode(mode)
            return pc(x, act=hip.ACT_RELU)
        return run

    cases = {
        "nothing beside it": None,
        "split 7x7 32->64": conv(32, 64, 7, 1, 1088, 1920, "split"),
        "split 7x7 64->32": conv(64, 32, 7, 1, 1088, 1920, "split"),
        "split 7x7 32->16": conv(32, 16, 7, 1, 1088, 1920, "split"),
        "split 7x7 8->32 (per-chunk instance)": conv(8, 32, 7, 1, 1088, 1920, "split"),
        "split 5x5 96->32": conv(96, 32, 5, 1, 1088, 1920, "split"),
        "split 3x3 128->128": conv(128, 128, 3, 1, 544, 960, "split"),
        "native 7x7 32->64 (LDS-DMA fp32)": conv(32, 64, 7, 1, 1088, 1920, "native"),
        "native 3x3 128->128": conv(128, 128, 3, 1, 544, 960, "native"),
        "1x1 128->128 (streaming kernel)": conv(128, 128, 1, 1, 544, 960, "native"),
        "native 7x7 16->2": conv(16, 2, 7, 1, 1088, 1920, "native"),
    }
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    sample = torch.zeros(128, dtype=torch.int32, device=dev)
    for name, run in cases.items():
        ref = None
        if run is not None:
            for _ in range(2):
                ref = run().buf.clone()            # tuning / packing outside the measurement; the result alone on the device
        torch.cuda.synchronize()
        wrong = 0
        bad.zero_()
        sample.zero_()
        torch.cuda.synchronize()
        for rep in range(6):
            if run is not None:
                with torch.cuda.stream(s2):         # canaries first: they sit on the CUs when the convolution's workgroups arrive
                    can.canary_launch(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sample.data_ptr()), 8192, 40)
                with torch.cuda.stream(s1):
                    for _ in range(3):
                        wrong += int((run().buf != ref).sum())
            else:
                with torch.cuda.stream(s2):
                    can.canary_launch(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sample.data_ptr()), 8192, 40)
        torch.cuda.synchronize()
        # few, long-lived canaries FIRST (one or two per CU, ~5 ms): the convolution's workgroups are placed beside them, at an LDS base
        # that is not 0 -- its own results must not change
        wrong_base = 0
        if run is not None:
            for rep in range(4):
                with torch.cuda.stream(s2):
                    can.canary_launch(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sample.data_ptr()), 384, 1500)
                with torch.cuda.stream(s1):
                    wrong_base += int((run().buf != ref).sum())
                torch.cuda.synchronize()
        nb = int(bad.item())
        sm = sample.cpu().tolist()
        print(f"{name:40s}: {wrong if run is not None else 0} convolution outputs differ from the run alone ({wrong_base if run is not None else 0} beside long-lived canaries); {nb} corrupted canary words" + ("" if not nb else "  first (index, value): " + ", ".join(f"({sm[2*i]}, {sm[2*i+1] & 0xffffffff:#010x})" for i in range(min(nb, 6)))), flush=True)


if __name__ == "__main__":
    main()
