#!/usr/bin/env python
"""SPyNet alone, launch after launch (run two instances at once to share the GPU): per level, the level input, the up-sampled flow and
the level's flow must be the same bits every time.
    python tools/spynet_determinism.py [reps] [--native] [--prealloc] [--no-clones]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))
from vcamd import hip, lhbdc  # noqa: E402
from vcamd.hip import T  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
    if "--native" in sys.argv:
        hip.set_fp32_mode("native")
    dev = torch.device("cuda:0")
    from vcamd.seeding import calibrated_state_dict
    m = lhbdc.Model()
    m.load_state_dict(calibrated_state_dict(m.state_dict(), seed=1234))
    m = m.to(dev).eval()
    net = m.flow_predictor if hasattr(m, "flow_predictor") else [c for c in m.modules() if isinstance(c, lhbdc.Network)][0]
    g = torch.Generator().manual_seed(11)
    n, H, W = 2, 1088, 1920
    base = torch.nn.functional.avg_pool2d(torch.rand(n, 3, H + 8, W + 16, generator=g), 9, 1, padding=4)
    a, b = base[..., :H, 0:W].contiguous().to(dev), base[..., :H, 6:W + 6].contiguous().to(dev)
    L = hip.lib()

    prealloc = "--prealloc" in sys.argv      # every per-level tensor allocated once and reused by every run (no allocator reuse inside a run)
    pool = {}

    def buf(key, *shape_and_dtype):
        if not prealloc:
            return T.empty(*shape_and_dtype)
        if key not in pool:
            pool[key] = T.empty(*shape_and_dtype)
        return pool[key]

    def run():
        stages = {}
        with torch.no_grad():
            p1 = T.empty(n, H, W, 3, dev)
            p2 = T.empty(n, H, W, 3, dev)
            for i in range(n):
                net.preprocess_into(a[i], p1.images(i, i + 1))
                net.preprocess_into(b[i], p2.images(i, i + 1))
            pyr1, pyr2 = net.pyramid(p1), net.pyramid(p2)
            flow = None
            for lvl in range(len(pyr1)):
                f1, f2 = pyr1[lvl], pyr2[lvl]
                c = net._convs(lvl)
                sp = hip.fp32_mode() == "split" and c[0].split_ok and c[1].split_ok and c[1].split_pays(f1.n, f1.h, f1.w)
                feat = buf(("feat", lvl), f1.n, f1.h, f1.w, 8, dev, "sp3" if sp else "f32")
                up = buf(("up", lvl), f1.n, f1.h, f1.w, 2, dev)
                fv = flow.view() if flow is not None else lhbdc._zero_flow_view(f1)
                if sp:
                    hip.check(L.vc_spynet_level_input_sp3(hip.stream(), f1.view(), f2.view(), fv, feat.ptr, up.view()), "li")
                else:
                    hip.check(L.vc_spynet_level_input(hip.stream(), f1.view(), f2.view(), fv, feat.view(), up.view()), "li")
                stages[f"L{lvl} {f1.h}x{f1.w} pyramid"] = f1.buf.clone()
                stages[f"L{lvl} level input{' (split)' if sp else ''}"] = feat.buf.clone()
                stages[f"L{lvl} up"] = up.buf.clone()
                x = feat
                for j in range(4):
                    osp = sp and (j < 2 or (j == 2 and hip.wants_split(c[3], f1)))
                    o = buf(("x", lvl, j), f1.n, f1.h, f1.w, c[j].cout, dev, "sp3" if osp else "f32") if prealloc else None
                    x = c[j](x, out=o, act=hip.ACT_RELU, out_sp3=osp)
                    stages[f"L{lvl} conv{j}"] = x.buf.clone()
                flow = c[4](x, res=up, out=buf(("flow", lvl), f1.n, f1.h, f1.w, 2, dev) if prealloc else None)
                stages[f"L{lvl} flow"] = flow.buf.clone()
        torch.cuda.synchronize()
        return stages

    ref = run()
    nbad = 0
    for r in range(reps):
        cur = run()
        bad = [(k, int((cur[k] != ref[k]).sum())) for k in ref if bool((cur[k] != ref[k]).any())]
        nbad += bool(bad)
        if bad:
            print(f"run {r}: first stages that differ: {bad[:4]}", flush=True)
    print(f"{nbad} of {reps} runs differ from the first", flush=True)


if __name__ == "__main__":
    main()
