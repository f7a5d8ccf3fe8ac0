"""Race screen of the streaming 1x1 kernel (VC_CFG_PWS): short tiles with a residual, many repetitions, against cfg 2."""
import sys, os, torch
sys.path.insert(0, "video-compression_amd")
from vcamd import hip
dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "30"))
def run(prec, cin, cout, n, h, w, hin, hout, with_res, act=hip.ACT_NONE):
    hip.set_conv_precision(prec)
    g = torch.Generator().manual_seed(0)
    pc = hip.PackedConv(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, torch.randn(cout, generator=g) * 0.1, device=dev)
    hip.set_conv_precision("fp32")
    x = hip.T.empty(n, h, w, cin, dev, "f16" if hin else "f32"); x.buf.normal_()
    res = None
    if with_res:
        res = hip.T.empty(n, h, w, cout, dev); res.buf.normal_()
    fl = (hip.CFG_F16 if prec == "fp16" else 0) | (hip.CFG_IN_F16 if hin else 0) | (hip.CFG_OUT_F16 if hout else 0)
    bad = worst = 0
    for rep, cfg in enumerate([2] + [9] * REPS):
        out = hip.T.empty(n, h, w, cout, dev, "f16" if hout else "f32"); out.buf.zero_()
        pc.tuned = {(n, h, w, fl): cfg | hip.CFG_EXACT | fl}
        pc(x, out=out, act=act, res=res)
        torch.cuda.synchronize()
        o = out.buf.float()
        if cfg == 2:
            ref = o
            continue
        d = int((o != ref).sum())
        bad += d > 0
        worst = max(worst, d)
    print(f"{prec} {cin}->{cout} @{n}x{h}x{w} hin={hin} hout={hout} res={with_res}: {bad} of {REPS} runs differ (worst {worst} of {ref.numel()} values)", flush=True)
    return bad
tot = 0
for args in [("fp16", 32, 32, 1, 16, 96, True, False, True), ("fp16", 32, 32, 1, 16, 96, False, False, True), ("fp16", 32, 32, 2, 272, 480, True, False, True),
             ("fp16", 64, 64, 1, 64, 96, False, False, True), ("fp16", 64, 64, 1, 544, 960, True, False, True), ("fp16", 128, 128, 1, 544, 960, True, False, True),
             ("fp16", 96, 96, 1, 272, 480, True, True, True), ("fp32", 32, 32, 1, 544, 960, False, False, True), ("fp32", 64, 64, 1, 272, 480, False, False, True),
             ("fp32", 128, 128, 1, 272, 480, False, False, True), ("fp16", 128, 128, 1, 272, 480, True, True, False), ("fp16", 64, 64, 1, 544, 960, True, True, False)]:
    tot += run(*args)
print("TOTAL runs with differences:", tot)
sys.exit(1 if tot else 0)
