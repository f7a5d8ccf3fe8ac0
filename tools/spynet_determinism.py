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


def describe(key, cur, ref, n, h, w):
    """Error signature of a differing up-sampled flow ([n, h, w, 2] fp32): where the wrong pixels sit and what they hold."""
    cur, ref = cur.reshape(n, h, w, 2).cpu(), ref.reshape(n, h, w, 2).cpu()
    bad = (cur != ref).any(-1)
    rows = torch.nonzero(bad.any(-1))
    print(f"    {key}: {int(bad.sum())} wrong pixels in {rows.shape[0]} rows", flush=True)
    for img, y in rows[:6].tolist():
        xs = torch.nonzero(bad[img, y]).reshape(-1)
        runs, start, prev = [], int(xs[0]), int(xs[0])
        for x in xs[1:].tolist():
            if x != prev + 1:
                runs.append((start, prev))
                start = x
            prev = x
        runs.append((start, prev))
        x0, x1 = runs[0]
        got, want = cur[img, y, x0:x1 + 1], ref[img, y, x0:x1 + 1]
        like = [dy for dy in range(-6, 7) if dy and 0 <= y + dy < h and torch.equal(got, ref[img, y + dy, x0:x1 + 1])]
        other = [j for j in range(n) if j != img and torch.equal(got, ref[j, y, x0:x1 + 1])]
        print(f"      image {img} row {y}: x runs {runs[:6]} (first run: start % 16 = {x0 % 16}, length {x1 - x0 + 1}); max|d| {float((got - want).abs().max()):.3e}; "
              f"all zero: {bool((got == 0).all())}; NaN: {bool(torch.isnan(got).any())}; equals another row's values: {like}; another image's: {other}; "
              f"first wrong / right: {got[0].tolist()} / {want[0].tolist()}", flush=True)


def where(cur, ref, n, h, w, slot):
    """Diagnostic library only (vc_li_diag_read): the hardware unit every wave with a wrong result ran on."""
    import ctypes
    import numpy as np
    L = hip.lib()
    if not hasattr(L, "vc_li_diag_read"):
        return
    SLOT = 1 << 17
    buf = np.zeros(6 * SLOT * 4, dtype=np.uint32)
    L.vc_li_diag_read.argtypes = [ctypes.c_void_p]
    if L.vc_li_diag_read(buf.ctypes.data) != 0:
        return
    buf = buf.reshape(6, SLOT, 4)
    cur, ref = cur.reshape(n, h, w, 2).cpu(), ref.reshape(n, h, w, 2).cpu()
    bad = torch.nonzero((cur != ref).any(-1))
    gx = (w + 255) // 256
    units, bad_waves = {}, set()

    def unit(hw, xcc):
        return (int(xcc) & 15, (int(hw) >> 13) & 7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 15, (int(hw) >> 4) & 3)     # XCC, SE, SH, CU, SIMD
    for img, y, x in bad.tolist():
        idx = ((img * h + y) * gx + x // 256) * 4 + (x % 256) // 64
        if idx < SLOT:
            u = unit(buf[slot, idx, 0], buf[slot, idx, 1])
            units[u] = units.get(u, 0) + 1
            bad_waves.add(idx)
    nw = min(SLOT, n * h * gx * 4)
    life = buf[slot, :nw, 2].astype(np.int64) * 10           # ns (100 MHz ticks)
    moved = buf[slot, :nw, 0] != buf[slot, :nw, 3]            # HW_ID at the end differs from the start: the wave came back on another slot
    bw = np.array(sorted(bad_waves), dtype=np.int64)
    good = np.ones(nw, dtype=bool)
    good[bw] = False
    q = lambda a, p_: int(np.percentile(a, p_)) if a.size else -1          # noqa: E731
    print(f"    wrong pixels by (XCC, SE, SH, CU, SIMD) of the wave that produced them: {dict(list(units.items())[:12])}{' ...' if len(units) > 12 else ''}; "
          f"{len(set(unit(a, b) for a, b in buf[slot, :nw, :2].tolist()))} distinct SIMDs in the launch", flush=True)
    print(f"    wave lifetimes (s_memrealtime, ns): the {bw.size} waves with wrong lanes median {q(life[bw], 50)} min {int(life[bw].min()) if bw.size else -1} "
          f"max {int(life[bw].max()) if bw.size else -1}; the {int(good.sum())} others median {q(life[good], 50)} p99 {q(life[good], 99)} max {int(life[good].max())}; "
          f"waves longer than 100 us: {int((life > 100000).sum())} of {nw}, of them with wrong lanes: {int((life[bw] > 100000).sum())}; "
          f"HW_ID changed between start and end: {int(moved.sum())} waves, of them with wrong lanes: {int(moved[bw].sum())}", flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
    if "--native" in sys.argv:
        hip.set_fp32_mode("native")
    dev = torch.device("cuda:0")
    if os.environ.get("VC_VA_SHIFT_MB"):          # a different virtual-address layout than a twin process: every later allocation moves
        shift = torch.empty(int(os.environ["VC_VA_SHIFT_MB"]) << 20, dtype=torch.uint8, device=dev)
        print(f"virtual-address shift: {shift.numel() >> 20} MiB held at {shift.data_ptr():#x}", flush=True)
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

    global pyr_sizes
    pyr_sizes = []
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
            pyr_sizes[:] = [t.h for t in pyr1]
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
            for k, _ in bad[:3]:
                if k.endswith("level input (split)"):
                    size = [q for q in ref if q.startswith(k.split()[0] + " ") and q.endswith("pyramid")][0].split()[1]
                    hh, ww = (int(v) for v in size.split("x"))
                    c_, r_ = cur[k].view(torch.int16).reshape(n, hh, ww, 3, 8).cpu(), ref[k].view(torch.int16).reshape(n, hh, ww, 3, 8).cpu()
                    d_ = c_ != r_
                    px = torch.nonzero(d_.any(-1).any(-1))
                    lanes = sorted(set((px[:, 2] % 64).tolist()))
                    chans = sorted(set(torch.nonzero(d_.any(0).any(0).any(0).any(0)).reshape(-1).tolist()))
                    pieces = sorted(set(torch.nonzero(d_.any(0).any(0).any(0).any(-1)).reshape(-1).tolist()))
                    print(f"    {k}: {px.shape[0]} wrong pixels; x % 64 in {lanes[:4]}..{lanes[-4:]} ({len(lanes)} distinct); x % 256 // 64 = "
                          f"{sorted(set(((px[:, 2] % 256) // 64).tolist()))}; channels {chans}; pieces {pieces}; images {sorted(set(px[:, 0].tolist()))}; "
                          f"rows {sorted(set(px[:, 1].tolist()))[:8]}", flush=True)
                if k.endswith(" up"):
                    size = [q for q in ref if q.startswith(k.split()[0] + " ") and q.endswith("pyramid")][0].split()[1]
                    describe(k, cur[k], ref[k], n, *(int(v) for v in size.split("x")))
                    lvl = int(k.split()[0][1:])
                    where(cur[k], ref[k], n, *(int(v) for v in size.split("x")), len(pyr_sizes) - 1 - lvl)
                    dump = os.environ.get("VC_LI_DUMP")
                    if dump and lvl > 0 and not os.path.exists(dump):
                        hh, ww = (int(v) for v in size.split("x"))
                        c_, r_ = cur[k].reshape(n, hh, ww, 2).cpu(), ref[k].reshape(n, hh, ww, 2).cpu()
                        rows = torch.nonzero((c_ != r_).any(-1).any(-1))
                        torch.save({"level": lvl, "h": hh, "w": ww, "rows": rows, "cur_rows": [c_[i, y] for i, y in rows.tolist()],
                                    "ref_rows": [r_[i, y] for i, y in rows.tolist()], "coarse_cur": cur[f"L{lvl - 1} flow"].cpu(),
                                    "coarse_same": bool(torch.equal(cur[f"L{lvl - 1} flow"], ref[f"L{lvl - 1} flow"]))}, dump)
    print(f"{nbad} of {reps} runs differ from the first", flush=True)


if __name__ == "__main__":
    main()
