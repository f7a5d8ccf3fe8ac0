#!/usr/bin/env python
"""Does a frame come out with the same bits while ANOTHER process keeps the GPU busy?  (Counted waits must not depend on how long a
DMA takes: a second process stretches every latency.)  Runs LHBDC's B-frame forward on one 1080p frame triple repeatedly, alone and
beside a child process that launches large convolutions, and names the first stage whose values differ from the first run.

    python tools/contention_check.py [--reps R] [--fp32-mode split|native] [--small]
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))

HAMMER = r"""
import sys, time, torch
sys.path.insert(0, sys.argv[1])
from vcamd import hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
wt = torch.randn(64, 32, 7, 7, generator=g) / 40
pc = hip.PackedConv(wt, None, stride=1, device=dev)
x = hip.T.empty(2, 1088, 1920, 32, dev); x.buf.normal_()
w1 = torch.randn(128, 128, 1, 1, generator=g) / 12
p1 = hip.PackedConv(w1, None, stride=1, device=dev)
y = hip.T.empty(1, 1088, 1920, 128, dev); y.buf.normal_()
t_end = time.time() + float(sys.argv[2])
print("hammer up", flush=True)
while time.time() < t_end:
    for _ in range(4):
        pc(x, act=hip.ACT_RELU)
        p1(y)
        torch.empty(64 << 20, device=dev).zero_()
    torch.cuda.synchronize()
"""


def flat(tr, prefix=""):
    import torch
    out = {}
    for k, v in tr.items():
        if isinstance(v, dict):
            out.update(flat(v, prefix + k + "."))
        elif isinstance(v, (list, tuple)):
            for i, e in enumerate(v):
                if torch.is_tensor(e):
                    out[f"{prefix}{k}[{i}]"] = e
                elif hasattr(e, "buf"):
                    out[f"{prefix}{k}[{i}]"] = e.buf
        elif torch.is_tensor(v):
            out[prefix + k] = v
        elif hasattr(v, "buf"):
            out[prefix + k] = v.buf
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--fp32-mode", default="split")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--hammer-seconds", type=float, default=60.0)
    ap.add_argument("--lds-poison", action="store_true",
                    help="instead of the hammer: fill every CU's LDS with a pattern after EVERY launch (tools/micro/lds_canary.hip) -- a kernel "
                         "that reads LDS it has not written gives different results for different patterns")
    args = ap.parse_args()
    # the child starts before this process touches the GPU
    child = subprocess.Popen([sys.executable, "-c", HAMMER, os.path.join(ROOT, "video-compression_amd"), str(args.hammer_seconds)],
                             stdout=subprocess.PIPE, text=True)
    import torch
    from vcamd import hip, lhbdc
    from vcamd.seeding import calibrated_state_dict
    dev = torch.device("cuda:0")
    hip.set_fp32_mode(args.fp32_mode)
    m = lhbdc.Model()
    m.load_state_dict(calibrated_state_dict(m.state_dict(), seed=1234))
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    H, W = (256, 384) if args.small else (1088, 1920)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, H + 8, W + 16, generator=g), 9, 1, padding=4)
    xb, xc, xa = (base[..., :H, i:i + W].contiguous().to(dev) for i in (0, 3, 6))
    line = child.stdout.readline()
    print("child:", line.strip(), flush=True)

    def run():
        tr = {}
        with torch.no_grad():
            x_hat, tot = m.forward_device(xb, xc, xa, trace=tr)
        d = {k: v.clone() for k, v in flat(tr).items()}
        d["x_hat"] = x_hat.clone()
        torch.cuda.synchronize()
        return d

    bad_total = 0
    if args.lds_poison:
        import ctypes
        can = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "lds_canary.so"))
        can.lds_fill.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        plain_check = hip.check
        pattern = [0]

        def check_and_poison(rc, what=""):
            plain_check(rc, what)
            if can.lds_fill(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), pattern[0], ctypes.c_void_p(sink.data_ptr())) != 0:
                raise RuntimeError("lds_fill")
        ref = run()
        hip.check = check_and_poison
        for mod in list(sys.modules.values()):           # (modules that did `from .hip import check` keep the plain one: they call hip.check)
            if getattr(mod, "hip", None) is hip:
                pass
        for name, pat in (("zeros", 0), ("quiet NaNs", 0x7fc00000), ("bf16 NaN pairs", 0x7fc07fc0), ("ones", 0x3f800000), ("large", 0x7f7f7f7f)):
            pattern[0] = pat
            cur = run()
            bad = {k: (int((cur[k] != ref[k]).sum()), float((cur[k].float() - ref[k].float()).abs().max())) for k in ref
                   if k in cur and cur[k].shape == ref[k].shape and bool((cur[k] != ref[k]).any())}
            bad_total += len(bad)
            print(f"LDS filled with {name} after every launch: " + ("same bits in all %d traced tensors" % len(ref) if not bad else f"DIFFERENT (count, max |d|): {bad}"), flush=True)
        child.wait()
        sys.exit(1 if bad_total else 0)
    ref = run()
    for phase in ("beside the hammer", "alone"):
        if phase == "alone":
            child.wait()
        for r in range(args.reps):
            cur = run()
            bad = {k: int((cur[k] != ref[k]).sum()) for k in ref if k in cur and cur[k].shape == ref[k].shape and bool((cur[k] != ref[k]).any())}
            bad_total += len(bad)
            print(f"{phase}, run {r}: " + ("same bits in all %d traced tensors" % len(ref) if not bad else f"DIFFERENT: {bad}"), flush=True)
    if child.poll() is None:
        child.wait()
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
