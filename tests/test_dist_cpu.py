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
