"""Real-bitstream GOP coding for LHBDC with the host range coder OFF the GPU's critical path.

The reference's CLI (LHBDC/encode_B.py:71-126, decode_B.py:63-104) codes one B-frame per process call and its test loop
(test/testing.py) only estimates rate, so nothing there overlaps entropy coding with the networks.  Here a GOP is coded
into / from the same ``bits_B.bin`` containers with

* ONE analysis pass per codec on the encoder (``MeanScaleHyperprior.code_t``): the reconstruction the next hierarchy level
  needs and the integers of the bitstream come out of the same launch sequence; the integers travel to pinned host memory on a
  side stream and are range-coded by worker threads (``vc_rans_*`` through ctypes releases the GIL) while the GPU is
  already coding the next level -- the encoder never waits for the coder;
* a decoder that enqueues the (tiny) hyper-synthesis of BOTH codecs first, so the scale-table indexes are on the host
  while the GPU estimates the predictor flows, decodes the y strings of all frames of the level in parallel threads, and
  only then enqueues the synthesis transforms;
* the frames of one hierarchy level batched through the same kernels, as in ``gop.code_gop_lhbdc``.

Wiring = the CLI's (both flow predictors equal pad(flow_ab): SURVEY.md Appendix B.1), so every container is decodable by
the reference's decode_B.py and vice versa.  The encoder-side reconstruction equals the decoder's output bit for bit
(same integers, same kernels) -- asserted in tests/test_stream_gpu.py.

Flex-Rate is not offered here on purpose: its compress() codes the UN-gained latent while forward() quantises the gained
one (quirk B.6), so the reference's own encoder and decoder disagree on the reconstruction whenever the gain differs from
one -- there is no closed loop to pipeline.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import hip
from .gop import DECODING_INFO, LEVEL_GROUPS
from .hip import T
from .lhbdc import Model, _cli_predictors, _count, check_container_shapes, frame_list, read_container


def _pack_container(lmbda, mv_shape, res_shape, mv_y, mv_z, res_y, res_z):
    """bits_B.bin (encode_B.py:114-126); see lhbdc.write_container."""
    head = (np.array(lmbda, dtype=np.uint32).tobytes() + np.array(tuple(mv_shape), dtype=np.uint16).tobytes() +
            np.array(len(mv_y), dtype=np.uint32).tobytes() + np.array(len(mv_z), dtype=np.uint32).tobytes() +
            np.array(tuple(res_shape), dtype=np.uint16).tobytes() + np.array(len(res_y), dtype=np.uint32).tobytes())
    return head + mv_y + mv_z + res_y + res_z


class _Tables:
    """Host copies of a codec's range-coder tables (read once; the model must have been update()d)."""

    def __init__(self, codec):
        self.eb = codec.entropy_bottleneck.tables()
        self.gc = codec.gaussian_conditional.tables()
        self.channels = codec.N

    def z_index(self, hz, wz):
        return np.repeat(np.arange(self.channels, dtype=np.int32), hz * wz)


class PendingContainers:
    """Containers of one encoded level pass; ``result()`` blocks until the worker threads have written them."""

    def __init__(self, futures, keep_alive):
        self._futures, self._keep = futures, keep_alive

    def result(self):
        out = [f.result() for f in self._futures]
        self._keep = None
        return out


class LhbdcStreamCodec:
    def __init__(self, model: Model, lmbda=1626, workers=4):
        self.model, self.lmbda = model, int(lmbda)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.copy_stream = None
        self.t_mv = _Tables(model.mv_compressor)
        self.t_res = _Tables(model.residual_compressor)

    # -- plumbing: device int32 tensors -> pinned host memory on a side stream --------------------------------
    def _to_host_async(self, tensors):
        """Start the D2H copies behind everything enqueued so far; returns (pinned arrays, event to wait on)."""
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record()
        host = []
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self.copy_stream)
                host.append(h)
            done = torch.cuda.Event()
            done.record()
        return host, done

    # -- encoder ------------------------------------------------------------------------------------------
    def encode_frames(self, x_before, x_current, x_after):
        """n independent B-frames (one hierarchy level).  Returns (x_hat NCHW = what decode_frames will output,
        PendingContainers).  No host synchronisation: the caller can enqueue the next level at once."""
        m = self.model
        xb_, xc_, xa_ = (frame_list(t) for t in (x_before, x_current, x_after))      # tensors or lists of frames
        n, dev = _count(xc_), xc_[0].device
        frames = {"b": xb_, "c": xc_, "a": xa_}
        flow_ba, flow_ab, hh, ww = _cli_predictors(m, frames, n)
        cur = m._flows(frames, [("c", "b"), ("c", "a")])
        cur_flows, _, _ = Model._pool_pad(cur, 1.0)
        diff = T.empty(n, flow_ab.h, flow_ab.w, 4, dev)
        hip.axpby(cur_flows.images(0, n), flow_ab, 1.0, -1.0, out=diff.channels(0, 2))
        hip.axpby(cur_flows.images(n, 2 * n), flow_ba, 1.0, -1.0, out=diff.channels(2, 4))
        mv_hat, mv_sym = m.mv_compressor.code_t(diff)
        host_mv, ev_mv = self._to_host_async([mv_sym["y_sym"], mv_sym["y_idx"], mv_sym["z_sym"]])
        xb, xc, xa = hip.nchw_frames_to_nhwc(xb_), hip.nchw_frames_to_nhwc(xc_), hip.nchw_frames_to_nhwc(xa_)
        pred, resid = m._predict(xb, xa, mv_hat, flow_ab, flow_ba, hh, ww, cur=xc)
        res_hat, res_sym = m.residual_compressor.code_t(resid)
        host_res, ev_res = self._to_host_async([res_sym["y_sym"], res_sym["y_idx"], res_sym["z_sym"]])
        x_hat = hip.nhwc_to_nchw(hip.axpby(res_hat, pred))
        mv_shape, res_shape = mv_sym["shape"], res_sym["shape"]
        zi_mv, zi_res = self.t_mv.z_index(*mv_shape), self.t_res.z_index(*res_shape)

        def code_one(i):
            ev_mv.synchronize()
            y, idx, z = (h.numpy() for h in host_mv)
            mv_z = hip.rans_encode(z[i], zi_mv, *self.t_mv.eb)
            mv_y = hip.rans_encode(y[i], idx[i], *self.t_mv.gc)
            ev_res.synchronize()
            y, idx, z = (h.numpy() for h in host_res)
            res_z = hip.rans_encode(z[i], zi_res, *self.t_res.eb)
            res_y = hip.rans_encode(y[i], idx[i], *self.t_res.gc)
            return _pack_container(self.lmbda, mv_shape, res_shape, mv_y, mv_z, res_y, res_z)

        futures = [self.pool.submit(code_one, i) for i in range(n)]
        return x_hat, PendingContainers(futures, (mv_sym, res_sym, host_mv, host_res))

    # -- decoder ------------------------------------------------------------------------------------------
    def decode_frames(self, x_before, x_after, containers):
        """n independent B-frames from their containers (bytes).  Returns x_hat NCHW [n,3,H,W]."""
        m = self.model
        xb_, xa_ = frame_list(x_before), frame_list(x_after)
        n, dev = _count(xb_), xb_[0].device
        if len(containers) != n:
            raise hip.VcError("one container per frame")
        parsed = [read_container(c) for c in containers]
        mv_shape, res_shape = tuple(parsed[0][3]), tuple(parsed[0][4])
        if any(tuple(p[3]) != mv_shape or tuple(p[4]) != res_shape for p in parsed):
            raise hip.VcError("frames of one pass must have the same latent shapes")
        check_container_shapes(mv_shape, res_shape, xb_[0].shape[-2], xb_[0].shape[-1])   # before anything is allocated from them
        zi_mv, zi_res = self.t_mv.z_index(*mv_shape), self.t_res.z_index(*res_shape)
        # 1. hyper-latents: fixed tables, a few thousand symbols per frame -- decoded at once, in parallel
        fz = [self.pool.submit(lambda p=p: (hip.rans_decode(p[1][1][0], zi_mv, *self.t_mv.eb),
                                            hip.rans_decode(p[2][1][0], zi_res, *self.t_res.eb))) for p in parsed]
        zs = [f.result() for f in fz]
        z_mv = torch.from_numpy(np.stack([z[0] for z in zs])).to(dev)
        z_res = torch.from_numpy(np.stack([z[1] for z in zs])).to(dev)
        # 2. device: both hyper-syntheses FIRST (tiny), their indexes start travelling to the host ...
        means_mv, idx_mv = m.mv_compressor.hyper_decode_t(z_mv, n, mv_shape)
        means_res, idx_res = m.residual_compressor.hyper_decode_t(z_res, n, res_shape)
        (h_idx_mv, h_idx_res), ev_idx = self._to_host_async([idx_mv, idx_res])
        # 3. ... while the GPU estimates the predictor flows (the heavy part that needs no bitstream at all)
        flow_ba, flow_ab, hh, ww = _cli_predictors(m, {"b": xb_, "a": xa_}, n)
        xb, xa = hip.nchw_frames_to_nhwc(xb_), hip.nchw_frames_to_nhwc(xa_)
        # 4. host: every y string of the level in its own worker, in parallel with (3)
        ev_idx.synchronize()
        i_mv, i_res = h_idx_mv.numpy(), h_idx_res.numpy()
        f_mv = [self.pool.submit(hip.rans_decode, parsed[i][1][0][0], i_mv[i], *self.t_mv.gc) for i in range(n)]
        f_res = [self.pool.submit(hip.rans_decode, parsed[i][2][0][0], i_res[i], *self.t_res.gc) for i in range(n)]
        y_mv = torch.from_numpy(np.stack([f.result() for f in f_mv])).to(dev)
        mv_hat = m.mv_compressor.synth_decode_t(y_mv, means_mv)
        pred, _ = m._predict(xb, xa, mv_hat, flow_ab, flow_ba, hh, ww)
        y_res = torch.from_numpy(np.stack([f.result() for f in f_res])).to(dev)
        res_hat = m.residual_compressor.synth_decode_t(y_res, means_res)
        return hip.nhwc_to_nchw(hip.axpby(res_hat, pred))

    # -- GOP-8 in hierarchical order ------------------------------------------------------------------------
    def encode_gop(self, gop, dec_first, dec_last):
        """``gop``: 9 padded NCHW frames.  Returns ({order: container bytes}, {order: reconstruction}).  The host coding of
        level l runs while the GPU codes level l+1; the only synchronisation is the final collection."""
        decoded, pending = {0: dec_first, 8: dec_last}, []
        for group in LEVEL_GROUPS:
            xb = [decoded[DECODING_INFO[o][0]] for o in group]
            xc = [gop[o] for o in group]
            xa = [decoded[DECODING_INFO[o][1]] for o in group]
            x_hat, pend = self.encode_frames(xb, xc, xa)
            for i, o in enumerate(group):
                decoded[o] = x_hat[i:i + 1]
            pending.append((group, pend))
        containers = {}
        for group, pend in pending:
            for o, c in zip(group, pend.result()):
                containers[o] = c
        return containers, decoded

    def decode_gop(self, containers, dec_first, dec_last):
        decoded = {0: dec_first, 8: dec_last}
        for group in LEVEL_GROUPS:
            xb = [decoded[DECODING_INFO[o][0]] for o in group]
            xa = [decoded[DECODING_INFO[o][1]] for o in group]
            x_hat = self.decode_frames(xb, xa, [containers[o] for o in group])
            for i, o in enumerate(group):
                decoded[o] = x_hat[i:i + 1]
        return decoded

    def close(self):
        self.pool.shutdown(wait=True)
