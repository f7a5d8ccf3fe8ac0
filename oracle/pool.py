"""Several CPU-oracle runs side by side -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

One 1088x1920 pass of the oracle is 10-30 s at 8 threads, and more threads do not make ONE pass faster (oneDNN's
convolutions stop scaling around 8-16 cores); the GPU hosts have ~128 cores, so independent passes run concurrently:
Python threads (the GIL is released inside the operators), each with its own oracle instance (the capture hooks of
oracle.trace live on the instance) and its own intra-op thread count.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import torch


def host_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:  # noqa: BLE001
        pass
    return os.cpu_count() or 1


def plan(jobs, threads_per_job=8):
    """(workers, threads per worker) for ``jobs`` independent passes on this host."""
    cores = host_cores()
    workers = max(1, min(jobs, cores // threads_per_job))
    return workers, max(1, min(16, cores // workers))


def run_jobs(jobs, fn, threads_per_job=8):
    """``[fn(job) for job in jobs]`` with up to cores // threads_per_job jobs in flight; results in job order.  ``fn`` must
    build (or own) every module it hooks."""
    jobs = list(jobs)
    workers, threads = plan(len(jobs), threads_per_job)
    before = torch.get_num_threads()

    def call(job):
        torch.set_num_threads(threads)          # (per calling thread: OpenMP's nthreads-var is thread-local)
        with torch.no_grad():
            return fn(job)
    try:
        if workers == 1:
            return [call(j) for j in jobs]
        with ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(call, jobs))
    finally:
        torch.set_num_threads(before)


def lhbdc_encode_job(state_dict):
    """-> fn(frames) for :func:`run_jobs`: the oracle's encode_B (LHBDC/encode_B.py:71-105) on one (x_before, x_current,
    x_after) triple with the latents its entropy models saw captured: {"mv_bits", "res_bits", "container", "mv", "res"}."""
    from oracle import lhbdc as ol
    from oracle.cai.entropy_models import get_scale_table
    from oracle.trace import CodecTrace

    def fn(frames):
        xb, xc, xa = frames
        ora = ol.LhbdcModel().eval()
        ora.load_state_dict(state_dict)
        ora.mv_compressor.update(force=True)
        ora.residual_compressor.update(force=True)
        with CodecTrace(ora.mv_compressor) as t_mv, CodecTrace(ora.residual_compressor) as t_res:
            mv_bits, res_bits = ol.encode_B(ora, xa, xc, xb)
            table = get_scale_table()
            lat = {"mv": t_mv.latents(table), "res": t_res.latents(table)}
        return {"mv_bits": mv_bits, "res_bits": res_bits, "container": ol.write_container(1626, mv_bits, res_bits), **lat}
    return fn
