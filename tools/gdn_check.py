#!/usr/bin/env python
"""A/B of the GDN layer's output formats (fp32 mode "split"): fp32 store, fp32 store + vc_split3, split store from the streaming 1x1
kernel, split store from the classic instance.   python tools/gdn_check.py [n,h,w ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402
from vcamd.layers import GDN  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    hip.set_fp32_mode("split")
    for spec in (sys.argv[1:] or ["1,544,960", "4,544,960", "1,272,480", "4,272,480", "1,136,240"]):
        n, h, w = [int(v) for v in spec.split(",")]
        gdn = GDN(128).to(dev)
        x = hip.T.empty(n, h, w, 128, dev)
        x.buf.normal_()
        r = hip.T.empty(n, h, w, 128, dev)
        r.buf.normal_()
        gdn.run(x, res=r)
        pc = gdn._packed
        out32 = hip.T.empty(n, h, w, 128, dev)
        outsp = hip.T.empty(n, h, w, 128, dev, "sp3")
        res = {}
        for name, cfg, o in (("pws fp32", hip.CFG_PWS, out32), ("classic fp32", 0, out32), ("pws split", hip.CFG_PWS, outsp), ("classic split", 0, outsp)):
            fl = hip.CFG_OUT_SP3 if o is outsp else 0
            pc.tuned = {(n, h, w, fl, hip.ACT_NONE, hip.EPI_GDN): cfg | hip.CFG_EXACT | fl}
            res[name] = timeit(lambda: gdn.run(x, res=r, out=o))
        res["vc_split3"] = timeit(lambda: hip.split3(out32, out=outsp))
        print(f"GDN 128 @{n}x{h}x{w}: " + "  ".join(f"{k} {v * 1000:.0f} us" for k, v in res.items()))


if __name__ == "__main__":
    main()
