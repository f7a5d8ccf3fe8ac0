#!/usr/bin/env python
"""Minimal form of the shared-GPU difference of DESIGN section 5e: K1 (a plain torch copy) writes one of two coarse flows in turn, K2 = the
SPyNet level-input kernel reads it at once: its two outputs must equal what the same launch gives on a quiet stream.  Run two instances at once; VC_HIP_LIB selects
the build (3-D-grid form of the kernel / grid-stride form).
    python tools/level_input_repro.py [iters]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    L = hip.lib()
    for n, h, w in [(2, 136, 240), (2, 68, 120), (2, 272, 480)]:
        f1, f2 = hip.T.empty(n, h, w, 3, dev), hip.T.empty(n, h, w, 3, dev)
        f1.buf.uniform_()
        f2.buf.uniform_()
        fc = hip.T.empty(n, h // 2, w // 2, 2, dev)
        fields = [torch.randn_like(fc.buf), torch.randn_like(fc.buf)]
        conv = "--conv" in sys.argv
        if conv:        # K1 = the level's last convolution (7x7 16 -> 2 with the up-sampled flow as residual), as inside SPyNet
            g = torch.Generator().manual_seed(3)
            pc = hip.PackedConv(torch.randn(2, 16, 7, 7, generator=g) / 28.0, torch.randn(2, generator=g) * 0.1, stride=1, device=dev)
            xs = [hip.T.empty(n, h // 2, w // 2, 16, dev) for _ in range(2)]
            for t in xs:
                t.buf.normal_()
            res = hip.T.empty(n, h // 2, w // 2, 2, dev)
            res.buf.normal_()
            hip.set_fp32_mode("native")
            for p_ in range(2):
                pc(xs[p_], res=res, out=fc)
                torch.cuda.synchronize()
                fields[p_] = fc.buf.clone()
        sp3 = "--fp32-records" not in sys.argv
        feat = hip.T.empty(n, h, w, 8, dev, "sp3" if sp3 else "f32")
        up = hip.T.empty(n, h, w, 2, dev)

        def k2():
            if sp3:
                hip.check(L.vc_spynet_level_input_sp3(hip.stream(), f1.view(), f2.view(), fc.view(), feat.ptr, up.view()), "li")
            else:
                hip.check(L.vc_spynet_level_input(hip.stream(), f1.view(), f2.view(), fc.view(), feat.view(), up.view()), "li")

        refs = []
        for p in range(2):                          # the two expected results, each confirmed by a second launch
            while True:
                fc.buf.copy_(fields[p])
                torch.cuda.synchronize()
                k2()
                a = (up.buf.clone(), feat.buf.clone())
                torch.cuda.synchronize()
                k2()
                if torch.equal(a[0], up.buf) and torch.equal(a[1], feat.buf):
                    refs.append(a)
                    break
        bad = torch.zeros(iters, dtype=torch.int64, device=dev)
        for r in range(iters):
            if conv:
                fc.buf.fill_(777.0)                                                                      # K0: marker
                pc(xs[r & 1], res=res, out=fc)                                                           # K1: the convolution
            else:
                fc.buf.copy_(fields[r & 1])                                                              # K1: a plain copy kernel
            k2()                                                                                         # K2 reads it at once
            bad[r] = (up.buf != refs[r & 1][0]).sum() + (feat.buf != refs[r & 1][1]).sum()
        torch.cuda.synchronize()
        b = bad.cpu()
        nz = b[b != 0]
        print(f"level input @{n}x{h}x{w}: {len(nz)} of {iters} iterations differ from the result alone" + (f" (values: {nz[:8].tolist()})" if len(nz) else ""), flush=True)


if __name__ == "__main__":
    main()
