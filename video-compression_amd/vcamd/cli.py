"""Command-line entry points with the reference scripts' arguments.

    python -m vcamd.cli encode_B --ref_1 frames/ref_1.png --ref_2 frames/ref_2.png --current frames/current.png --bin bits_B.bin --l 1626
    python -m vcamd.cli decode_B --ref_1 frames/ref_1.png --ref_2 frames/ref_2.png --bin bits_B.bin [--out decoded.png]
    python -m vcamd.cli test --test_path /datasets/UVG/full_test/ --b_pretrained ../new_compression_1626.pth --lmbda 1626

``encode_B`` / ``decode_B`` mirror LHBDC/encode_B.py:21-28,108-126 and LHBDC/decode_B.py:23-28,88-124 (same argument
names and defaults, same ``bits_B.bin`` container, ``decoded.png`` written beside the bitstream); ``test`` mirrors the
argument parser and the loop of LHBDC/test/testing.py:35-59,89-196 on top of ``vcamd.data.SequenceReader``.

Checkpoints: ``--weights`` (default ``pretrained_weights/compression_<l>.pth``, as the reference) is loaded with
``torch.load(...)["state_dict"]``.  The reference's checkpoints live on Google Drive; where none is present ``--seeded
SEED`` runs the same code on the deterministic synthetic checkpoint of ``vcamd.seeding`` (encoder and decoder must use
the same seed).  Everything runs on the HIP path: a CUDA device is required.
"""
import argparse
import os
import sys

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from vcamd import hip, lhbdc  # noqa: E402
from vcamd.data import SequenceReader, read_png, write_png  # noqa: E402

LAMBDAS = [228, 436, 845, 1626, 3141]


def _device():
    if not torch.cuda.is_available():
        raise hip.VcError("vcamd.cli runs on the HIP path only: no CUDA/ROCm device is visible (there is no CPU fallback)")
    return torch.device("cuda")


def load_b_model(l, weights=None, seeded=None, device=None):
    """encode_B.py:32-37: Model(), load_state_dict, update(force=True) on both codecs, eval()."""
    device = device or _device()
    model = lhbdc.Model()
    if seeded is not None:
        from vcamd.seeding import seeded_state_dict
        model.load_state_dict(seeded_state_dict(model.state_dict(), seed=int(seeded)))
    else:
        path = weights or f"pretrained_weights/compression_{l}.pth"
        if not os.path.exists(path):
            raise hip.VcError(f"checkpoint {path} not found (the reference's weights are a separate download); pass --weights PATH "
                              "or --seeded SEED for the synthetic checkpoint")
        model.load_state_dict(torch.load(path, map_location=lambda storage, loc: storage)["state_dict"])
    model.mv_compressor.update(force=True)
    model.residual_compressor.update(force=True)
    return model.to(device).float().eval()


def frame_from_file(path, device):
    """``process_frame(imageio.imread(path).astype(float))`` (encode_B.py:58-64,109-111) with the pixel work on the device:
    returns the padded fp32 NCHW tensor and the original (h, w)."""
    rgb = read_png(path)
    x = hip.frame_from_uint8(torch.from_numpy(rgb).to(device))
    return x, rgb.shape[:2]


def cmd_encode_B(args):
    dev = _device()
    model = load_b_model(args.l, args.weights, args.seeded, dev)
    with torch.no_grad():
        x_before, _ = frame_from_file(args.ref_1, dev)
        x_after, _ = frame_from_file(args.ref_2, dev)
        x_current, _ = frame_from_file(args.current, dev)
        mv_bits, res_bits = lhbdc.encode_B(model, x_after, x_current, x_before)
    blob = lhbdc.write_container(args.bin, args.l, mv_bits, res_bits)
    print(f"{args.bin}: {len(blob)} bytes ({8.0 * len(blob) / (x_current.shape[-1] * x_current.shape[-2]):.4f} bpp on the padded frame)")
    return blob


def cmd_decode_B(args):
    dev = _device()
    l, mv_bits_dec, res_bits_dec, shape_mv, shape_res = lhbdc.read_container(args.bin)
    model = load_b_model(int(l), args.weights, args.seeded, dev)
    with torch.no_grad():
        x_before, (h, w) = frame_from_file(args.ref_1, dev)
        x_after, _ = frame_from_file(args.ref_2, dev)
        decoded = lhbdc.decode_B(x_before, x_after, model, mv_bits_dec, res_bits_dec, shape_mv, shape_res)
        u8 = hip.frame_to_uint8(decoded[:1], h, w).cpu().numpy()
    write_png(args.out, u8)
    print(f"{args.out}: {w}x{h}")
    return u8


def cmd_test(args):
    """testing.py:65-196: every sequence under --test_path, GOP by GOP, I-frames through mbt2018_mean(--i_qual),
    B-frames through the model; prints the per-level and overall (PSNR, bpp) table."""
    from vcamd import gop
    from vcamd.iframe import mbt2018_mean
    dev = _device()
    model = load_b_model(args.lmbda, args.b_pretrained, args.seeded, dev)
    i_model = mbt2018_mean(args.i_qual, "mse", pretrained=False)
    if args.seeded is not None or not args.i_pretrained:
        from vcamd.seeding import seeded_state_dict
        i_model.load_state_dict(seeded_state_dict(i_model.state_dict(), seed=4321, conv_gain=0.8))
    else:
        i_model.load_state_dict(torch.load(args.i_pretrained, map_location="cpu"))
    i_model.update(force=True)
    i_model = i_model.to(dev).float().eval()
    reader = SequenceReader(args.test_path, None, args.test_gop_size, args.test_skip_frames, args.test_numbers, dev, args.workers,
                            yuv_size=tuple(args.yuv_size) if args.yuv_size else None)
    table = gop.RdTable()
    with torch.no_grad():
        for vi, name in enumerate(reader.video_names):
            avail = (len(reader.videos[vi]) + reader.skip_frames - 1) // reader.skip_frames
            reader.prefetch([k for k in reader.items if k[0] == vi])
            rows = gop.code_sequence_lhbdc(model, i_model, lambda idx, vi=vi: reader.load_frame(vi, idx), avail, reader.h, reader.w,
                                           video=vi, gop_size=args.test_gop_size, test_size=args.test_numbers)
            rows = gop.gather_records(rows, dev)                 # one D2H of the per-frame scalars, (video, frame) order
            for r in rows.tolist():
                table.update("I" if int(r[6]) == 1 else "B", r[1], int(r[2]), int(r[0]), r[3], r[4], r[5])
            s = gop.summarize(rows)
            print(f"{name}: {s['frames']} frames  PSNR {s['psnr']:.3f} dB  {s['bpp']:.4f} bpp", flush=True)
    summary = {"per_level": table.per_level(), "per_level_frame_type": table.per_level_frame_type(),
               "overall": table._group(lambda r: 0).get(0)}
    for level, v in summary["per_level"].items():
        print(f"level {level:2d}: {v['frames']:4d} frames  PSNR {v['psnr']:.3f} dB  {v['bpp']:.4f} bpp")
    print(f"overall : {summary['overall']['frames']:4d} frames  PSNR {summary['overall']['psnr']:.3f} dB  {summary['overall']['bpp']:.4f} bpp")
    st = reader.stats
    print(f"ingest: {st['frames']} frames, decode {st['decode_s']:.2f} s (worker time), H2D {st['h2d_s']:.2f} s, consumer waited "
          f"{st['wait_s']:.2f} s")
    reader.close()
    return summary


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m vcamd.cli", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)

    def ckpt(p):
        p.add_argument("--weights", default=None, help="checkpoint with a 'state_dict' entry (default pretrained_weights/compression_<l>.pth)")
        p.add_argument("--seeded", type=int, default=None, help="use the synthetic checkpoint of vcamd.seeding with this seed instead")

    e = sub.add_parser("encode_B", formatter_class=argparse.ArgumentDefaultsHelpFormatter)      # encode_B.py:21-28
    e.add_argument("--ref_1", default="frames/ref_1.png")
    e.add_argument("--ref_2", default="frames/ref_2.png")
    e.add_argument("--current", default="frames/current.png")
    e.add_argument("--bin", default="bits_B.bin")
    e.add_argument("--l", type=int, default=1626, choices=LAMBDAS)
    ckpt(e)
    e.set_defaults(fn=cmd_encode_B)

    d = sub.add_parser("decode_B", formatter_class=argparse.ArgumentDefaultsHelpFormatter)      # decode_B.py:23-28
    d.add_argument("--ref_1", default="frames/ref_1.png")
    d.add_argument("--ref_2", default="frames/ref_2.png")
    d.add_argument("--bin", default="bits_B.bin")
    d.add_argument("--out", default="decoded.png")
    ckpt(d)
    d.set_defaults(fn=cmd_decode_B)

    t = sub.add_parser("test", formatter_class=argparse.ArgumentDefaultsHelpFormatter)          # test/testing.py:35-59
    t.add_argument("--project_name", type=str, default="LHBDC_test")
    t.add_argument("--model_name", type=str, default="Single_level_1626")
    t.add_argument("--test_path", type=str, default="/datasets/UVG/full_test/")
    t.add_argument("--test_gop_size", type=int, default=8)
    t.add_argument("--i_interval", type=int, default=8)
    t.add_argument("--test_skip_frames", type=int, default=1)
    t.add_argument("--test_numbers", type=int, default=None)
    t.add_argument("--device", type=str, default="cuda")
    t.add_argument("--workers", type=int, default=4)
    t.add_argument("--b_pretrained", type=str, default="../new_compression_1626.pth")
    t.add_argument("--i_pretrained", type=str, default=None, help="state dict of the mbt2018_mean I-frame codec (zoo weights are a download)")
    t.add_argument("--i_qual", type=int, default=7)
    t.add_argument("--lmbda", type=int, default=1626)
    t.add_argument("--yuv_size", type=int, nargs=2, default=None, metavar=("W", "H"), help="sequences are raw 8-bit 4:2:0 files of this size")
    t.add_argument("--seeded", type=int, default=None)
    t.set_defaults(fn=cmd_test)
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    return args.fn(args)


if __name__ == "__main__":
    main()
