#!/usr/bin/env python
"""bench.py -- B-frame throughput of the LHBDC hot path on MI355X (BASELINE.json configs[1]).

One "step" = one GOP-8 of a synthetic 1080p video (SURVEY.md 8(d) Config 2): the 7 B-frames coded in
hierarchical order through Model.forward (flow -> warp -> mask/blend -> residual analysis -> hyperprior
-> likelihood/bit count -> synthesis), inputs resident in HBM, seeded random weights of the reference
architecture (pretrained weights and UVG are not available offline).  I-frames are the reference's
third-party mbt2018_mean codec -- outside the per-B-frame path -- so the boundary frames of each GOP
are taken as already decoded.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: launched by torch.distributed.run, one rank per GPU; GOPs shard across ranks (weak scaling, no
data-path collective), the per-frame R-D records are gathered over RCCL at the end.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-compression_amd"))

H, W = 1080, 1920
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 matrix peak (dense)
PEAK_F16_MFMA_TFLOPS = 2500.0    # BF16/FP16 matrix peak, dense
PEAK_HBM_GBPS = 8000.0           # HBM3E peak (same guide)
PEAK_HBM_GBS = 8000.0


def synthetic_gop(seed, gop_index, device, frames_per_gop=9, hw=None):
    H, W = hw if hw is not None else (1080, 1920)
    """Config 2: band-limited noise texture (Gaussian sigma=3 px) + global translation (1.5,0.75) px per
    frame + 2 % additive noise, quantised to uint8, then /255 and reflection-padded to 1088x1920."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    margin = 32
    tex = rng.random((3, H + 2 * margin, W + 2 * margin)).astype(np.float32)
    tex = np.stack([ndimage.gaussian_filter(c, 3.0) for c in tex])
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    frames = []
    for t in range(frames_per_gop):
        tt = gop_index * (frames_per_gop - 1) + t
        dx, dy = 1.5 * (tt % 16), 0.75 * (tt % 16)
        shifted = np.stack([ndimage.shift(c, (dy, dx), order=1, mode="nearest") for c in tex])
        f = shifted[:, margin:margin + H, margin:margin + W]
        noise = np.random.default_rng(seed + 1 + tt).standard_normal(f.shape).astype(np.float32) * 0.02
        u8 = np.clip(np.round((f + noise) * 255.0), 0, 255).astype(np.uint8)
        x = torch.from_numpy(u8.astype(np.float32) / 255.0)[None]
        x = torch.nn.functional.pad(x, (0, (64 - W % 64) % 64, 0, (64 - H % 64) % 64), mode="reflect")
        frames.append(x.to(device))
    return frames


def pick_cpu_threads():
    """Give the CPU baseline its best shot: time one representative convolution (SPyNet 7x7 32->64 at
    272x480) at a few thread counts and keep the fastest (oversubscribing a big host is much slower)."""
    cores = os.cpu_count() or 1
    x = torch.randn(1, 32, 272, 480)
    w = torch.randn(64, 32, 7, 7)
    best, best_t = 1, float("inf")
    for n in sorted({c for c in (8, 16, 32, 64, cores // 2, cores) if 1 <= c <= cores}):
        torch.set_num_threads(n)
        with torch.no_grad():
            torch.nn.functional.conv2d(x, w, padding=3)
            t0 = time.perf_counter()
            for _ in range(3):
                torch.nn.functional.conv2d(x, w, padding=3)
            dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = n, dt
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", choices=["lhbdc", "flex", "icip2024"], default="lhbdc",
                    help="lhbdc = BASELINE.json configs[1] (headline); flex = configs[2] (Flex-Rate, 4 rate points); "
                         "icip2024 = configs[4]'s model (FlowGuidedB, GOP-16, flow-resolution search, 5 quality levels)")
    ap.add_argument("--precision", choices=["fp32", "fp16"], default="fp32",
                    help="fp32 = exact path (headline); fp16 = half-precision MFMA conv path of BASELINE configs[4]")
    ap.add_argument("--resolution", choices=["1080p", "2160p"], default="1080p")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gops-per-step", type=int, default=None,
                    help="LHBDC / Flex-Rate: independent GOPs coded per step and GPU with their hierarchy levels batched together "
                         "(default at 1080p: 4 / 2; 1 at 2160p)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--kernel-table", default=None, help="write the per-kernel event timing table here (json)")
    args = ap.parse_args()

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from vcamd import flex, hip, lhbdc
    from vcamd import gop as vgop
    from vcamd.seeding import seeded_state_dict

    global H, W
    if args.resolution == "2160p":
        H, W = 2160, 3840
    hip.set_conv_precision(args.precision)
    f16 = args.precision == "fp16"
    is_flex = args.model == "flex"
    is_icip = args.model == "icip2024"
    if is_icip:
        from vcamd import icip2024
        model = icip2024.FlowGuidedB()
    else:
        model = flex.BidirFlowRef(n=4) if is_flex else lhbdc.Model()
    sd = seeded_state_dict(model.state_dict(), seed=1234)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    per_gop = 15 if (is_flex or is_icip) else 7
    # GOPs per step and GPU (LHBDC): 4 at 1080p (+2.5 % over one: the single-frame level and the coarse layers get 4x
    # the work per launch), 1 at 2160p where a level pass is already four 1080p frames' worth of pixels
    G = 1 if is_icip else (args.gops_per_step or ((2 if is_flex else 4) if args.resolution == "1080p" else 1))

    # every rank codes its own GOP (GOP index = rank): weak scaling, per-GPU work fixed
    frames = []
    for g in range(G):
        frames += synthetic_gop(1234, rank * G + g, dev, 17 if (is_flex or is_icip) else 9, (H, W))
    records = []
    # Flex: 4 rate points selected purely through the gain units (n = 0..3, l = 1), one per step in turn
    rate_points = [{lvl: (n, 1.0) for lvl in range(4)} for n in range(4)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if is_icip:                     # quality levels 0..4 in turn; the per-frame flow-resolution search runs on the device
        pool = None if args.no_graph else torch.cuda.graph_pool_handle()   # the five graphs replay in turn: one memory pool
        runners = [None if args.no_graph else vgop.GopGraph(model, H, W, kind="icip2024", quality=lvl, pool=pool)
                   for lvl in range(5)]
    elif is_flex:
        runners = [None if args.no_graph else vgop.GopGraph(model, H, W, kind="flex", quality=q, gops=G) for q in rate_points]
    else:
        runners = [None if args.no_graph else vgop.GopGraph(model, H, W, gops=G)]
    counter = [0]

    def step(keep):
        i = counter[0] % len(runners)
        counter[0] += 1
        recs = records if keep else None
        if runners[i] is not None:
            runners[i].code(frames, gop_index=rank * G, records=recs)
        elif is_icip:
            vgop.code_gop_icip2024(model, frames, frames[0], frames[16], H, W, i, recs, video=0, gop_index=rank)
        elif is_flex:
            gops = [frames[17 * g:17 * g + 17] for g in range(G)]
            vgop.code_gops_flex(model, gops, [(gp[0], gp[16]) for gp in gops], H, W, rate_points[i], recs, video=0,
                                first_gop_index=rank * G)
        else:
            gops = [frames[9 * g:9 * g + 9] for g in range(G)]
            vgop.code_gops_lhbdc(model, gops, [(gp[0], gp[8]) for gp in gops], H, W, recs, video=0, first_gop_index=rank * G)

    with torch.no_grad():
        for _ in range(max(args.warmup, len(runners))):   # every graph is captured before the clock starts
            step(False)
        counter[0] = 0
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i == args.steps - 1)
        rows = vgop.gather_records(records, dev)      # the only exchange: final R-D gather (RCCL)
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    b_frames = per_gop * G * args.steps * world
    try:        # the headline configuration reports BASELINE.json's metric verbatim (UVG itself is unavailable: see "data")
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            headline_metric = json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        headline_metric = "frames/sec + bpp/PSNR on UVG 1080p GOP-8, 1/2/4/8 MI355X"
    result = {
        # BASELINE.json: "frames/sec + bpp/PSNR on UVG 1080p GOP-8"; value = B-frames/s of the codec hot path,
        # bpp/PSNR of the same frames in "quality" (UVG is not available offline -> synthetic video)
        "metric": (f"frames/sec + bpp/PSNR on {args.resolution} GOP-16 (ICIP2024 FlowGuidedB B-frame path, 5 quality levels)" if is_icip else
                   f"frames/sec + bpp/PSNR on {args.resolution} GOP-16 (Flex-Rate B-frame path, 4 rate points)" if is_flex else
                   headline_metric if (args.resolution == "1080p") else
                   f"frames/sec + bpp/PSNR on {args.resolution} GOP-8 (LHBDC B-frame codec path)"),
        "value": b_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16 operands / f32 accumulate (eligible convolutions), f32 elsewhere" if f16 else "f32",
        "data": f"synthetic (band-limited texture + global translation + 2% noise, {H}x{W} reflection-padded to x64); seeded random weights",
        "config": {"workload": (f"ICIP2024 FlowGuidedB {args.resolution} GOP-16: 15 B-frames per GOP via FlowGuidedB.forward with the "
                                "per-frame flow-resolution search (5 flow+warp passes), quality level s=step%5, one GOP per GPU per step"
                                if is_icip else f"Flex-Rate b_model {args.resolution} GOP-16: 15 B-frames per GOP via BidirFlowRef.forward, rate point "
                                f"n=step%4 through the gain units, {G} independent GOP(s) per GPU per step, hierarchy levels batched" if is_flex else
                                f"LHBDC {args.resolution} GOP-8 inference, single lambda: 7 B-frames per GOP via Model.forward, "
                                f"{G} independent GOP(s) per GPU per step, hierarchy levels batched across them"),
                   "frames_per_step_per_gpu": per_gop * G, "gop": 16 if (is_flex or is_icip) else 8,
                   "resolution": f"{W}x{H}", "precision": args.precision, "parallelism": f"gop-shard x{world}",
                   "launch": "eager" if args.no_graph else "hip-graph per GOP"},
    }
    result["peak_hbm_gb"] = round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1)   # of 288 GB, this rank, graphs included
    q = vgop.summarize(rows)
    result["quality"] = {"b_frames": q["frames"], "bpp_estimated": q["bpp"], "psnr_db": q["psnr"],
                         "note": "seeded random weights: R-D values are parity references, not codec quality"}

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel: HIP events on the launch stream, one instrumented B-frame ----
        with torch.no_grad():
            hip.timer = hip.KernelTimer()
            if is_icip:
                model(frames[0], frames[16], 0.5, 0.5, frames[8], 2, 1)
            elif is_flex:
                model(frames[0], frames[8], frames[16], n=[1], l=1.0)
            else:
                model(frames[0], frames[4], frames[8], False)
            table = hip.timer.table()
            hip.timer = None
        total_ms = sum(v["ms"] for v in table.values())
        ranked = sorted(table.items(), key=lambda kv: -kv[1]["ms"])
        key, dom = ranked[0]
        per_launch_flop = dom["flops"] / dom["launches"]
        per_launch_bytes = dom["bytes"] / dom["launches"]
        avg_ms = dom["ms"] / dom["launches"]
        peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
        # which roofline bounds the dominant kernel: arithmetic intensity against the ridge peak_flops / peak_bandwidth
        intensity = per_launch_flop / max(per_launch_bytes, 1.0)
        if intensity >= peak * 1e12 / (PEAK_HBM_GBPS * 1e9):
            achieved = per_launch_flop / (avg_ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "kernel": key, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                  "frac": achieved / peak, "traffic": None}
        else:
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            result["roofline"] = {"bound": "hbm", "kernel": key, "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                  "frac": achieved / PEAK_HBM_GBPS, "traffic": None,
                                  "algorithmic_bytes_per_launch": per_launch_bytes}
        result["roofline"].update({"launches_per_frame": dom["launches"], "avg_launch_ms": avg_ms,
                                   "share_of_conv_time": dom["ms"] / total_ms,
                                   "algorithmic_flop_per_launch": per_launch_flop,
                                   "flop_per_byte": intensity})
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (profiles/r01/traffic.json:
        # FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per the gfx950 correction)
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as f:
                tr = json.load(f)["kernels"].get(key)
            if tr and not f16 and args.resolution == "1080p":
                result["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                result["roofline"]["traffic_source"] = "profiles/r01/traffic.json (rocprofv3 --pmc, separate passes)"
        except (OSError, KeyError, ValueError):
            pass
        ns = [kv for kv in ranked if kv[0].startswith("conv k3 s1 128->128 @1x544x960")]
        if ns:
            v = ns[0][1]
            a = v["flops"] / v["launches"] / (v["ms"] / v["launches"] * 1e-3) / 1e12
            result["roofline_3x3_analysis_conv"] = {"kernel": ns[0][0], "achieved": a, "peak": peak,
                                                    "unit": "TFLOP/s", "frac": a / peak,
                                                    "avg_launch_ms": v["ms"] / v["launches"], "launches": v["launches"]}
        all_flops = sum(v["flops"] for v in table.values())
        result["conv_engine"] = {"frame_conv_ms": total_ms, "frame_conv_tflop": all_flops / 1e12,
                                 "avg_tflops": all_flops / (total_ms * 1e-3) / 1e12}
        if args.kernel_table:
            with open(args.kernel_table, "w") as f:
                json.dump({k: v for k, v in ranked}, f, indent=1)

        if not is_flex and not is_icip and args.resolution == "1080p":
            # ---- whole GOP as testing.py codes it: 1 I-frame (mbt2018_mean q7 architecture) + 7 B-frames ----
            from vcamd import iframe
            i_model = iframe.mbt2018_mean(7, "mse", pretrained=False)
            i_model.load_state_dict(seeded_state_dict(i_model.state_dict(), seed=4321, conv_gain=0.8))
            i_model = i_model.to(dev).eval()
            with torch.no_grad():
                def full_gop():
                    dec_last, _ = i_model.forward_device(frames[8])
                    return vgop.code_gop_lhbdc(model, frames, frames[0], dec_last, H, W)
                full_gop()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    full_gop()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 2
            result["full_gop"] = {"frames_per_s": 8.0 / dt, "ms_per_gop": 1000.0 * dt,
                                  "what": "1 I-frame (mbt2018_mean q7 architecture, seeded) + 7 B-frames per GOP, eager launches"}

        if is_icip and args.resolution == "1080p":
            # ---- whole GOP-16 as src/test.py codes it: 1 intra frame (ELIC architecture, seeded) + 15 B-frames ----
            from vcamd.layers import BitCounter
            i_model = icip2024.ELIC()
            i_model.load_state_dict(seeded_state_dict(i_model.state_dict(), seed=4321, conv_gain=0.7))
            i_model = i_model.to(dev).eval()
            with torch.no_grad():
                def full_gop():
                    dec_last = hip.nhwc_to_nchw(i_model.forward_device(hip.nchw_to_nhwc(frames[16]), BitCounter(dev, 6)))
                    return vgop.code_gop_icip2024(model, frames, frames[0], torch.clamp(dec_last, 0, 1), H, W, 2)
                full_gop()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    full_gop()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 2
            result["full_gop"] = {"frames_per_s": 16.0 / dt, "ms_per_gop": 1000.0 * dt,
                                  "what": "1 I-frame (ELIC architecture of src/model/elic.py, seeded) + 15 B-frames per GOP-16, "
                                          "eager launches, quality level 2"}

        # ---- CPU baseline: the oracle (PyTorch-CPU restatement, tensor-equal to the reference) ----
        if not args.no_cpu_baseline and args.resolution == "1080p":
            from oracle import flex as oracle_flex
            from oracle import lhbdc as oracle_lhbdc
            if is_icip:
                from oracle import icip2024 as oracle_icip
                ora = oracle_icip.FlowGuidedB().eval()
            else:
                ora = (oracle_flex.FlexModel(n=4) if is_flex else oracle_lhbdc.LhbdcModel()).eval()
            ora.load_state_dict(sd)
            torch.set_num_threads(pick_cpu_threads())
            mid = 8 if (is_flex or is_icip) else 4
            xb, xc, xa = frames[0].cpu(), frames[mid].cpu(), frames[2 * mid].cpu()
            with torch.no_grad():
                t1 = time.perf_counter()
                if is_icip:
                    o = ora(xb, xa, 0.5, 0.5, xc, 2, 1)
                    ref_hat, ref_bits = o["x_hat"], float(o["size"].item())
                elif is_flex:
                    o = ora(xb, xc, xa, n=[1], l=1.0, train=False)
                    ref_hat, ref_bits = o["x_hat"], float(o["size"].item())
                else:
                    ref_hat, _, ref_bits = ora(xb, xc, xa, False)
                cpu_s = time.perf_counter() - t1
                if is_icip:
                    g = model(frames[0], frames[16], 0.5, 0.5, frames[8], 2, 1)
                    gpu_hat, gpu_bits = g["x_hat"], float(g["size"].item())
                elif is_flex:
                    g = model(frames[0], frames[mid], frames[2 * mid], n=[1], l=1.0)
                    gpu_hat, gpu_bits = g["x_hat"], float(g["size"].item())
                else:
                    gpu_hat, _, gpu_bits = model(frames[0], frames[4], frames[8], False)
            result["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": torch.get_num_threads(),
                                      "kind": "port", "sample": "1 B-frame 1088x1920 (middle frame of the same GOP), "
                                      "PyTorch-CPU fp32 oracle (tensor-equal to the reference), thread count "
                                      "chosen by a conv micro-calibration"}
            src = frames[mid]
            d_psnr = abs(float(vgop.psnr_uint8(gpu_hat, src, H, W)) - float(vgop.psnr_uint8(ref_hat.to(dev), src, H, W)))
            result["parity_vs_cpu"] = {"d_psnr_db": d_psnr, "bits_rel": abs(gpu_bits - ref_bits) / abs(ref_bits),
                                       "max_abs": float((gpu_hat.cpu() - ref_hat).abs().max())}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
