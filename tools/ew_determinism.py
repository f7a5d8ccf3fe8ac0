#!/usr/bin/env python
"""Launch-to-launch equality of the split-tensor element-wise kernels (run two instances at once to share the GPU):
    python tools/ew_determinism.py & python tools/ew_determinism.py; wait
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    L = hip.lib()
    n, h, w = 4, 1088, 1920
    f1, f2 = hip.T.empty(n, h, w, 3, dev), hip.T.empty(n, h, w, 3, dev)
    f1.buf.uniform_()
    f2.buf.uniform_()
    fc = hip.T.empty(n, h // 2, w // 2, 2, dev)
    fc.buf.normal_()
    feat = hip.T.empty(n, h, w, 8, dev, "sp3")
    up = hip.T.empty(n, h, w, 2, dev)
    x = hip.T.empty(1, 544, 960, 64, dev)
    x.buf.normal_()
    o_up = hip.T.empty(1, 1088, 1920, 64, dev, "sp3")
    o_sp = hip.T.empty(1, 544, 960, 64, dev, "sp3")
    featf = hip.T.empty(n, h, w, 8, dev)
    kernels = {
        "level input (split)": (lambda: hip.check(L.vc_spynet_level_input_sp3(hip.stream(), f1.view(), f2.view(), fc.view(), feat.ptr, up.view()), "li"), [feat, up]),
        "level input (fp32)": (lambda: hip.check(L.vc_spynet_level_input(hip.stream(), f1.view(), f2.view(), fc.view(), featf.view(), up.view()), "li"), [featf, up]),
        "upsample into a split tensor": (lambda: hip.check(L.vc_upsample_bilinear_sp3(hip.stream(), x.view(), o_up.ptr, o_up.image_bytes, 2, 0, 1.0), "up"), [o_up]),
        "vc_split3": (lambda: hip.split3(x, out=o_sp), [o_sp]),
    }
    for name, (fn, outs) in kernels.items():
        for o in outs:
            o.buf.zero_()
        fn()
        ref = [o.buf.clone() for o in outs]
        bad = []
        for r in range(reps):
            for o in outs:
                o.buf.fill_(7)          # (stale content must not survive)
            fn()
            bad.append(sum(int((o.buf != q).sum()) for o, q in zip(outs, ref)))
        print(f"{name:32s}: values differing from the first launch, per launch: {'none in %d launches' % reps if not any(bad) else bad}", flush=True)


if __name__ == "__main__":
    main()
