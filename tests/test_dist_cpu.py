"""Multi-process path without GPUs: GOP sharding and the final R-D gather over a 2-rank gloo group
(the same code runs over RCCL on the GPU box)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from vcamd import gop as vgop


def test_shard_gops_partition():
    for gops in (1, 7, 74, 518):
        for world in (1, 2, 3, 8):
            spans = [vgop.shard_gops(gops, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gops
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = vgop.shard_gops(5, world, rank)
    recs = []
    for g in range(lo, hi):                      # fake per-frame results: value encodes (gop, frame)
        for order in vgop.CODING_ORDER[2:]:
            recs.append((0, g * 8 + order, vgop.HIER_LEVELS[order], torch.tensor(30.0 + g + order / 10.0),
                         1000.0 * (g * 8 + order), 1080.0 * 1920.0))
    rows = vgop.gather_records(recs, torch.device("cpu"))
    q.put((rank, rows.tolist()))
    dist.destroy_process_group()


def test_rd_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]                      # every rank holds the same, frame-ordered table
    rows = torch.tensor(got[0], dtype=torch.float64)
    frames = rows[:, 1].tolist()
    assert frames == sorted(frames) and len(frames) == 35 and len(set(frames)) == 35
    s = vgop.summarize(rows)
    exp_bits = sum(1000.0 * (g * 8 + o) for g in range(5) for o in vgop.CODING_ORDER[2:])
    assert abs(s["bpp"] - exp_bits / (35 * 1080 * 1920)) < 1e-12
    exp_psnr = sum(30.0 + g + o / 10.0 for g in range(5) for o in vgop.CODING_ORDER[2:]) / 35
    assert abs(s["psnr"] - exp_psnr) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: the 7-sequence set sharded by GOP (vcamd.gop.workload_plan / code_workload)
# ---------------------------------------------------------------------------------------------------------------------
class FakeCoder:
    """Stands in for the HIP coder: every 'decoded frame' is a token naming what was coded from what, so the test sees
    exactly which boundary frames each B-frame was predicted from; R-D numbers are functions of (video, frame)."""

    def __init__(self):
        self.intra_calls = []

    def intra(self, video, idx):
        self.intra_calls.append((video, idx))
        return ("I", video, idx), (torch.tensor(40.0 + video + idx / 1000.0), torch.tensor(5000.0 + idx), 100.0)

    def code_gops(self, items, bounds):
        recs = []
        for (video, g, idxs), (first, last) in zip(items, bounds):
            assert first == ("I", video, idxs[0]) and last == ("I", video, idxs[-1])      # the right boundary frames
            for order in vgop.CODING_ORDER[2:]:
                frame = g * 8 + order
                recs.append((video, frame, vgop.HIER_LEVELS[order], torch.tensor(30.0 + video + frame / 1000.0),
                             torch.tensor(100.0 * frame + video), 100.0))
        return recs


def _rows(records):
    return sorted((int(r[0]), int(r[1]), int(r[2]), round(float(r[3]), 6), float(r[4]), float(r[5]), int(r[6])) for r in records)


def test_workload_shards_union_equals_single_rank():
    plan = vgop.workload_plan([33, 17, 41, 9, 25, 17, 33])        # 7 videos, 21 GOPs of 8
    assert len(plan) == 4 + 2 + 5 + 1 + 3 + 2 + 4
    single = FakeCoder()
    whole = vgop.code_workload(plan, 1, 0, single.intra, single.code_gops, gops_per_pass=4)
    n_frames = sum(n for n in (33, 17, 41, 9, 25, 17, 33))
    assert len(whole) == n_frames and len({(r[0], r[1]) for r in whole}) == n_frames        # every frame exactly once
    assert len(single.intra_calls) == len(set(single.intra_calls))                          # no I-frame coded twice
    for world in (2, 3, 8, 21, 30):
        for per_pass in (1, 4):
            union, extra = [], 0
            for rank in range(world):
                c = FakeCoder()
                union += vgop.code_workload(plan, world, rank, c.intra, c.code_gops, gops_per_pass=per_pass)
                extra += len(c.intra_calls)
            assert _rows(union) == _rows(whole), (world, per_pass)
            # the only duplicated work: one boundary I-frame per shard that starts inside a video
            assert extra - len(single.intra_calls) <= world - 1


def _workload_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = vgop.workload_plan([33, 17, 41, 9, 25, 17, 33])
    c = FakeCoder()
    recs = vgop.code_workload(plan, world, rank, c.intra, c.code_gops, gops_per_pass=4)
    rows = vgop.gather_records(recs, torch.device("cpu"), width=7)
    q.put((rank, rows.tolist()))
    dist.destroy_process_group()


def test_workload_two_ranks_gloo_gathers_the_single_rank_table():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_workload_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    plan = vgop.workload_plan([33, 17, 41, 9, 25, 17, 33])
    c = FakeCoder()
    whole = vgop.gather_records(vgop.code_workload(plan, 1, 0, c.intra, c.code_gops, 4), torch.device("cpu"))
    assert torch.tensor(got[0], dtype=torch.float64).tolist() == whole.tolist()
    table = vgop.RdTable()
    table.extend_from_records(got[0], level=7)
    agg = table.per_level_frame_type()
    assert agg[(7, "I")]["frames"] == 21 + 7 and agg[(7, "B")]["frames"] == 21 * 7
