#!/usr/bin/env python
"""Platform check, no code of this repo involved: kernel K1 writes a buffer, kernel K2 (same stream) reads it at once.  Run two
instances at the same time (two processes sharing one GPU):
    python tools/two_process_visibility.py & python tools/two_process_visibility.py; wait
Counts iterations in which K2 saw bytes K1 had not written yet.
"""
import sys
import torch

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad_iters, bad_vals = 0, 0
for shape in [(2, 68, 120, 2), (2, 136, 240, 2), (2, 272, 480, 2)]:
    buf = torch.zeros(shape, device=dev)
    src = torch.zeros(shape, device=dev)
    out = torch.empty(shape, device=dev)
    flags = torch.zeros(iters, dtype=torch.int64, device=dev)
    for r in range(1, iters + 1):
        torch.add(src, float(r), out=buf)          # K1: every element = r
        torch.mul(buf, 1.0, out=out)               # K2: reads it at once
        flags[r - 1] = (out != float(r)).sum()
    torch.cuda.synchronize()
    f = flags.cpu()
    bad_iters += int((f != 0).sum())
    bad_vals += int(f.sum())
    print(f"{shape}: {int((f != 0).sum())} of {iters} iterations saw stale values ({int(f.sum())} values)", flush=True)
sys.exit(1 if bad_iters else 0)
