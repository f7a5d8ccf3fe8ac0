#!/usr/bin/env python
"""K0 fills the output with a marker, K1 = a convolution writes it, K2 (same stream) copies it at once: the copy must hold no marker and
equal the convolution's result alone.  Per tile configuration of the layer.  Run two instances at once to share the GPU.
    python tools/conv_then_read.py [iters] [cin,cout,k,n,h,w]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    spec = sys.argv[2] if len(sys.argv) > 2 else "16,2,7,2,136,240"
    cin, cout, k, n, h, w = [int(v) for v in spec.split(",")]
    dev = torch.device("cuda:0")
    hip.set_fp32_mode("native")
    g = torch.Generator().manual_seed(0)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    pc = hip.PackedConv(wt, b, stride=1, device=dev)
    x = hip.T.empty(n, h, w, cin, dev)
    x.buf.normal_()
    res = hip.T.empty(n, h, w, cout, dev)
    res.buf.normal_()
    out = hip.T.empty(n, h, w, cout, dev)
    chk = torch.empty_like(out.buf)
    cands = list(pc.candidates) or [pc.cfg]
    for c in cands:
        pc.candidates = [c]
        pc.tuned.clear()
        pc(x, res=res, out=out)
        ref = out.buf.clone()
        flags = torch.zeros(iters, dtype=torch.int64, device=dev)
        marks = torch.zeros(iters, dtype=torch.int64, device=dev)
        for r in range(iters):
            out.buf.fill_(777.0)                    # K0
            pc(x, res=res, out=out)                 # K1
            chk.copy_(out.buf)                      # K2 reads at once
            flags[r] = (chk != ref).sum()
            marks[r] = (chk == 777.0).sum()
        torch.cuda.synchronize()
        f, mk = flags.cpu(), marks.cpu()
        print(f"conv k{k} {cin}->{cout} @{n}x{h}x{w} cfg {c:#x}: {int((f != 0).sum())} of {iters} iterations: the copy differs from the result alone "
              f"({int(f.sum())} values, {int(mk.sum())} of them the marker K0 wrote)", flush=True)


if __name__ == "__main__":
    main()
