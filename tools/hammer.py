#!/usr/bin/env python
"""Keeps the GPU busy with long persistent convolution kernels for N seconds (a second tenant for the shared-GPU checks).
    python tools/hammer.py 60 &
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
pc = hip.PackedConv(torch.randn(64, 32, 7, 7, generator=g) / 40, None, stride=1, device=dev)
x = hip.T.empty(4, 1088, 1920, 32, dev)
x.buf.normal_()
t_end = time.time() + float(sys.argv[1] if len(sys.argv) > 1 else 60)
print("hammer up", flush=True)
while time.time() < t_end:
    for _ in range(8):
        pc(x, act=hip.ACT_RELU)
    torch.cuda.synchronize()
