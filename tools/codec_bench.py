#!/usr/bin/env python
"""End-to-end bitstream path at 1080p: encode_B -> bits_B container -> decode_B (LHBDC), timing the GPU
stages and the host range coder separately, and checking that the decoder reproduces the encoder-side
reconstruction (property that holds at any size)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from bench import synthetic_gop  # noqa: E402
from vcamd import hip, lhbdc  # noqa: E402
from vcamd.seeding import seeded_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model = lhbdc.Model()
    model.load_state_dict(seeded_state_dict(model.state_dict(), seed=1234))
    model.mv_compressor.update(force=True)
    model.residual_compressor.update(force=True)
    model = model.to(dev).eval()
    frames = synthetic_gop(1234, 0, dev)
    xb, xc, xa = frames[0], frames[4], frames[8]
    with torch.no_grad():
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mv_bits, res_bits = lhbdc.encode_B(model, xa, xc, xb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            blob = lhbdc.write_container(None, 1626, mv_bits, res_bits)
            _, s_mv, s_res, sh_mv, sh_res = lhbdc.read_container(blob)
            t2 = time.perf_counter()
            dec = lhbdc.decode_B(xb, xa, model, s_mv, s_res, sh_mv, sh_res)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
        print(f"encode_B {1e3 * (t1 - t0):.1f} ms, container {1e3 * (t2 - t1):.2f} ms, decode_B {1e3 * (t3 - t2):.1f} ms, "
              f"bitstream {len(blob)} bytes = {8 * len(blob) / (1080 * 1920):.3f} bpp")
        # host coder alone on the residual latents
        sym = np.random.default_rng(0).integers(-3, 4, 128 * 68 * 120).astype(np.int32)
        cdf, ln, off = model.residual_compressor.gaussian_conditional.tables()
        idx = np.random.default_rng(1).integers(0, 64, sym.size).astype(np.int32)
        t0 = time.perf_counter(); s = hip.rans_encode(sym, idx, cdf, ln, off); t1 = time.perf_counter()
        hip.rans_decode(s, idx, cdf, ln, off); t2 = time.perf_counter()
        print(f"host rANS on {sym.size} symbols: encode {1e3 * (t1 - t0):.1f} ms, decode {1e3 * (t2 - t1):.1f} ms")


if __name__ == "__main__":
    main()
