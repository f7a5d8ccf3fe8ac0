#!/usr/bin/env python
"""Real-bitstream path at 1080p (LHBDC): (1) one frame through the CLI functions encode_B -> bits_B container ->
decode_B, (2) a whole GOP-8 through the pipelined stream codec (vcamd/bitstream.py: one analysis pass per codec, host
range coding on worker threads overlapped with the GPU), with the host coder timed alone for reference.  Prints one
JSON line with the rates; checks that the decoder reproduces the encoder-side reconstructions bit for bit."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from bench import synthetic_gop  # noqa: E402
from vcamd import bitstream, hip, lhbdc  # noqa: E402
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
    out = {}
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
            lhbdc.decode_B(xb, xa, model, s_mv, s_res, sh_mv, sh_res)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
        out["cli_one_frame"] = {"encode_B_ms": 1e3 * (t1 - t0), "decode_B_ms": 1e3 * (t3 - t2), "container_bytes": len(blob),
                                "bpp": 8 * len(blob) / (1080 * 1920)}
        # host coder alone on residual-sized input
        sym = np.random.default_rng(0).integers(-3, 4, 128 * 68 * 120).astype(np.int32)
        cdf, ln, off = model.residual_compressor.gaussian_conditional.tables()
        idx = np.random.default_rng(1).integers(0, 64, sym.size).astype(np.int32)
        t0 = time.perf_counter(); s = hip.rans_encode(sym, idx, cdf, ln, off); t1 = time.perf_counter()
        hip.rans_decode(s, idx, cdf, ln, off); t2 = time.perf_counter()
        out["host_rans_alone"] = {"symbols": int(sym.size), "encode_ms": 1e3 * (t1 - t0), "decode_ms": 1e3 * (t2 - t1)}

        # pipelined GOP codec: 7 B-frames per GOP, boundary frames taken as decoded
        codec = bitstream.LhbdcStreamCodec(model, workers=8)
        codec.encode_gop(frames, frames[0], frames[8])                 # warm-up: packs weights, tunes tile configurations
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            containers, recon = codec.encode_gop(frames, frames[0], frames[8])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        codec.decode_gop(containers, frames[0], frames[8])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(reps):
            decoded = codec.decode_gop(containers, frames[0], frames[8])
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        same = all(torch.equal(decoded[o], recon[o]) for o in range(1, 8))
        # the model-only rate of the same GOP (likelihood path, no bitstream), for the overhead of real coding
        from vcamd import gop as vgop
        vgop.code_gop_lhbdc(model, frames, frames[0], frames[8], 1080, 1920)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(reps):
            vgop.code_gop_lhbdc(model, frames, frames[0], frames[8], 1080, 1920)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        codec.close()
        total = sum(len(c) for c in containers.values())
        out["pipelined_gop"] = {"encode_fps": 7 * reps / (t1 - t0), "decode_fps": 7 * reps / (t3 - t2),
                                "estimate_only_fps_eager": 7 * reps / (t5 - t4),
                                "decoder_reproduces_encoder_bit_for_bit": bool(same),
                                "gop_bytes": total, "bpp": 8 * total / (7 * 1080 * 1920), "worker_threads": 8}
    print(json.dumps(out))
    if not same:
        raise SystemExit("decoder output differs from the encoder-side reconstruction")


if __name__ == "__main__":
    main()
