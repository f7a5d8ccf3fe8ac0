"""Hierarchical bi-directional GOP coding and the GOP sharding used for multi-GPU runs.

Mirrors the evaluation loop that defines the reference's metric (LHBDC/test/testing.py:65-196):
coding order [0, 8, 4, 2, 1, 3, 6, 5, 7], each B-frame predicted from the two already decoded frames in
``DECODING_INFO``, PSNR on uint8-rounded crops, bpp = sum(bits) / sum(h*w) (test/utils.py:415-432).
GOPs depend only on their two boundary I-frames, so they shard across GPUs with no data-path collective;
the only exchange is the final gather of per-frame rate-distortion records.
"""
import math

import numpy as np

import torch

CODING_ORDER = [0, 8, 4, 2, 1, 3, 6, 5, 7]                     # testing.py:70
DECODING_INFO = {4: (0, 8), 2: (0, 4), 1: (0, 2), 3: (2, 4), 6: (4, 8), 5: (4, 6), 7: (6, 8)}  # :72
HIER_LEVELS = {4: 0, 2: 1, 1: 2, 3: 2, 6: 1, 5: 2, 7: 2}       # :74


# Flex-Rate GOP-16 (Flex-Rate.../test/testing.py:71-89)
CODING_ORDER_16 = [0, 16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10, 9, 11, 14, 13, 15]
DECODING_INFO_16 = {8: (0, 16), 4: (0, 8), 2: (0, 4), 1: (0, 2), 3: (2, 4), 6: (4, 8), 5: (4, 6), 7: (6, 8),
                    12: (8, 16), 10: (8, 12), 9: (8, 10), 11: (10, 12), 14: (12, 16), 13: (12, 14), 15: (14, 16)}
HIER_LEVELS_16 = {8: 0, 4: 1, 2: 2, 1: 3, 3: 3, 6: 2, 5: 3, 7: 3, 12: 1, 10: 2, 9: 3, 11: 3, 14: 2, 13: 3, 15: 3}
# (I-frame quality, {hierarchy level: (n, l)}) -- the 8 operating points of the published curve (:86-89)
FLEX_QUALITIES = [
    (5, {0: (1, 1.0), 1: (0, 0.33), 2: (0, 0.66), 3: (0, 1.0)}), (6, {0: (1, 0.66), 1: (1, 1.0), 2: (0, 0.33), 3: (0, 0.66)}),
    (6, {0: (1, 0.33), 1: (1, 0.66), 2: (1, 1.0), 3: (0, 0.33)}), (6, {0: (2, 1.0), 1: (1, 0.33), 2: (1, 0.66), 3: (1, 1.0)}),
    (7, {0: (2, 0.66), 1: (2, 1.0), 2: (1, 0.33), 3: (1, 0.66)}), (7, {0: (2, 0.33), 1: (2, 0.66), 2: (2, 1.0), 3: (1, 0.33)}),
    (7, {0: (3, 1.0), 1: (2, 0.33), 2: (2, 0.66), 3: (2, 1.0)}), (8, {0: (3, 1.0), 1: (3, 1.0), 2: (2, 0.33), 3: (2, 0.66)}),
]


def psnr_uint8(x_hat, x, h, w):
    """testing.py:176-182 / utils.py:32-51: PSNR of round(clip(.,0,1)*255) on the un-padded crop of the first image --
    one HIP kernel pair (vc_psnr_uint8), result left on the device."""
    from . import hip
    return hip.psnr_uint8(x_hat, x, h, w)


LEVEL_GROUPS = [[4], [2, 6], [1, 3, 5, 7]]                    # frames of one hierarchy level are independent


def code_gop_lhbdc(model, gop, dec_first, dec_last, h, w, records=None, video=0, gop_index=0, batch_levels=True):
    """Code the 7 B-frames of one GOP-8.  ``gop``: list of 9 NCHW frames (padded), ``dec_first`` /
    ``dec_last``: decoded boundary frames.  Appends (video, frame, level, psnr, bits, pixels) to
    ``records`` in the reference's coding order (psnr and bits stay device scalars: no host sync inside
    the GOP) and returns the decoded dict.

    ``batch_levels``: the frames of one hierarchy level ({4}, {2,6}, {1,3,5,7}) depend only on already
    decoded levels, so each level runs as ONE batched pass (1, 2, 4 frames) through the same kernels --
    identical per-frame arithmetic, but the small feature maps of the hyper-networks / MV codec / coarse
    SPyNet levels get 2-4x more workgroups and the big layers lose their partial last wave."""
    decoded = {0: dec_first, 8: dec_last}
    stats = {}
    groups = LEVEL_GROUPS if batch_levels else [[o] for o in CODING_ORDER[2:]]
    for group in groups:
        # (lists of frames: the layout kernels assemble the batch -- no torch.cat copies)
        xb = [decoded[DECODING_INFO[o][0]] for o in group]
        xc = [gop[o] for o in group]
        xa = [decoded[DECODING_INFO[o][1]] for o in group]
        x_hat, tot = model.forward_device(xb, xc, xa)
        for i, o in enumerate(group):
            decoded[o] = x_hat[i:i + 1]
            stats[o] = tot[i].sum()
    if records is not None:
        for order in CODING_ORDER[2:]:
            records.append((video, gop_index * 8 + order, HIER_LEVELS[order],
                            psnr_uint8(decoded[order], gop[order], h, w), stats[order], float(h * w)))
    return decoded


def code_gops_lhbdc(model, gops, bounds, h, w, records=None, video=0, first_gop_index=0):
    """Several independent GOP-8s in one go: GOPs share nothing but the model, so hierarchy level l of ALL of them runs
    as one batched pass (G, 2G, 4G frames) -- the same per-frame arithmetic as :func:`code_gop_lhbdc`, with the
    single-frame level {4} and the coarse layers getting G times more work per launch.  ``gops``: list of G lists of 9
    padded NCHW frames; ``bounds``: per GOP its (decoded first, decoded last) boundary frames.  Records are appended
    GOP by GOP in the reference's coding order; returns the list of decoded dicts."""
    decoded = [{0: b[0], 8: b[1]} for b in bounds]
    stats = [{} for _ in gops]
    for group in LEVEL_GROUPS:
        xb = [decoded[g][DECODING_INFO[o][0]] for g in range(len(gops)) for o in group]
        xc = [gops[g][o] for g in range(len(gops)) for o in group]
        xa = [decoded[g][DECODING_INFO[o][1]] for g in range(len(gops)) for o in group]
        x_hat, tot = model.forward_device(xb, xc, xa)
        i = 0
        for g in range(len(gops)):
            for o in group:
                decoded[g][o] = x_hat[i:i + 1]
                stats[g][o] = tot[i].sum()
                i += 1
    if records is not None:
        for g in range(len(gops)):
            for order in CODING_ORDER[2:]:
                records.append((video, (first_gop_index + g) * 8 + order, HIER_LEVELS[order],
                                psnr_uint8(decoded[g][order], gops[g][order], h, w), stats[g][order], float(h * w)))
    return decoded


LEVEL_GROUPS_16 = [[8], [4, 12], [2, 6, 10, 14], [1, 3, 5, 7, 9, 11, 13, 15]]


def code_gops_flex(model, gops, bounds, h, w, quality, records=None, video=0, first_gop_index=0, batch_levels=True):
    """The 15 B-frames of each of several independent GOP-16s with the per-hierarchy-level (n, l) of ``quality`` (an entry
    of FLEX_QUALITIES or a plain {level: (n, l)} dict), like Flex-Rate.../test/testing.py:192-201.
    ``batch_levels``: the frames of one hierarchy level share their rate point and depend only on lower levels, so each
    level -- of ALL the GOPs given -- runs as one batched pass with identical per-frame arithmetic.
    ``gops``: list of lists of 17 padded NCHW frames; ``bounds``: per GOP (decoded first, decoded last)."""
    table = quality[1] if isinstance(quality, tuple) else quality
    decoded = [{0: b[0], 16: b[1]} for b in bounds]
    stats = [{} for _ in gops]
    groups = LEVEL_GROUPS_16 if batch_levels else [[o] for o in CODING_ORDER_16[2:]]
    ng = range(len(gops))
    for group in groups:
        n, l = table[HIER_LEVELS_16[group[0]]]
        xb = [decoded[g][DECODING_INFO_16[o][0]] for g in ng for o in group]
        xc = [gops[g][o] for g in ng for o in group]
        xa = [decoded[g][DECODING_INFO_16[o][1]] for g in ng for o in group]
        x_hat, tot = model.forward_device(xb, xc, xa, n=[n], l=l)
        i = 0
        for g in ng:
            for o in group:
                decoded[g][o] = x_hat[i:i + 1]
                stats[g][o] = tot[i].sum()
                i += 1
    if records is not None:
        for g in ng:
            for order in CODING_ORDER_16[2:]:
                records.append((video, (first_gop_index + g) * 16 + order, HIER_LEVELS_16[order],
                                psnr_uint8(decoded[g][order], gops[g][order], h, w), stats[g][order], float(h * w)))
    return decoded


def code_gop_flex(model, gop, dec_first, dec_last, h, w, quality, records=None, video=0, gop_index=0, batch_levels=True):
    """One GOP-16 (see :func:`code_gops_flex`); returns its decoded dict."""
    return code_gops_flex(model, [gop], [(dec_first, dec_last)], h, w, quality, records, video, gop_index, batch_levels)[0]


# ICIP2024 GOP-16 (ICIP2024/src/utils.py:188-221, src/test.py:37-101): frame 16 is intra-coded first, then the
# B-frames in this order, each predicted from the two buffered frames closest in display order.
ICIP_ORDER_16 = [16, 8, 4, 12, 2, 14, 6, 10, 1, 15, 3, 13, 5, 11, 7, 9]
ICIP_LEVELS_16 = {8: 0, 4: 1, 12: 1, 2: 2, 6: 2, 10: 2, 14: 2, 1: 3, 3: 3, 5: 3, 7: 3, 9: 3, 11: 3, 13: 3, 15: 3}


def icip2024_gop_plan(batch_levels=True, max_batch=8):
    """The reference loop (src/test.py:56-96) replayed on frame numbers only: for every B-frame of a GOP-16 its two
    references and temporal scales, grouped into passes.  Frames of one hierarchy level never reference each other
    (their bracketing references are strictly closer than any same-level frame), so a level whose frames share the
    same scales runs as ONE batched pass ({8}, {4,12}, {2,14,6,10}, {1,15,...,9})."""
    from . import icip2024
    buffer_order, plan = [0, 16], []
    for order in ICIP_ORDER_16[1:]:
        _, _, o1, o2 = icip2024.select_references(None, order, buffer_order, buffer_order)
        plan.append((order, o1, o2) + tuple(icip2024.get_scales(order, o1, o2)))
        buffer_order, _ = icip2024.update_buffer(buffer_order, buffer_order, order, order)
    groups = []
    for item in plan:
        key = (ICIP_LEVELS_16[item[0]], item[3], item[4])
        if batch_levels and groups and groups[-1][0] == key and len(groups[-1][1]) < max_batch:
            groups[-1][1].append(item)
        else:
            groups.append((key, [item]))
    return [g for _, g in groups]


def code_gop_icip2024(model, gop, dec_first, dec_last, h, w, level, records=None, video=0, gop_index=0,
                      search="device", down_ratio=1, cache_features=True, batch_levels=True, max_batch=None):
    """Code the 15 B-frames of one ICIP2024 GOP-16 at quality ``level`` (0..4, fractional values interpolate the
    gain vectors), following src/test.py:37-101.

    ``search``: "device" scores the five flow resolutions of get_best_down_ratio_prediction (opt_helpers.py:41-51)
    and picks the winner without leaving the GPU (no host sync in the whole GOP: capturable as one HIP graph);
    "host" compares the five PSNRs on the host like the reference loop (one sync per frame, no batching); None uses
    ``down_ratio``.  ``cache_features``: a decoded frame's feature pyramid is computed once and reused for every
    B-frame it serves as reference.  ``batch_levels`` / ``max_batch``: see :func:`icip2024_gop_plan` (default cap: 8 frames and sixteen 1080p frames' worth of pixels per pass -- 4 frames at 2160p, 167 GB peak).  Decoded frames are clamped to
    [0,1] before they serve as references (src/test.py:94).  Records are appended in the reference's coding order.
    Returns ({order: decoded}, {order: chosen down_ratio -- a device int32 index into (1,2,4,8,16) for "device"})."""
    from . import hip, icip2024
    from .layers import BitCounter
    ratios = (1, 2, 4, 8, 16)
    decoded, picked, feats, stats = {0: dec_first, 16: dec_last}, {}, {}, {}
    # references live as channels-last windows (clamped to [0,1] once, by vc_clamp01): no torch operator touches a pixel here
    nhwc = {0: hip.nchw_to_nhwc(dec_first), 16: hip.nchw_to_nhwc(dec_last)}

    def features(o):
        if o not in feats:
            feats[o] = model.feature_extractor.run(nhwc[o])
        return feats[o]

    if max_batch is None:        # at most 8 frames and sixteen 1080p frames' worth of activations per batched pass
        max_batch = max(1, min(8, (16 * 1088 * 1920) // (gop[0].shape[2] * gop[0].shape[3])))
    for group in icip2024_gop_plan(batch_levels and search != "host", max_batch):
        orders = [g[0] for g in group]
        s1, s2 = group[0][3], group[0][4]
        n = len(group)
        tc = hip.nchw_frames_to_nhwc([gop[o] for o in orders])
        t1 = hip.stack_images([nhwc[g[1]] for g in group])
        t2 = hip.stack_images([nhwc[g[2]] for g in group])
        flow, dr = None, down_ratio
        if search == "device":
            flow, choice, _ = model.search_flow_t(tc, t1, t2, s1, s2, ratios)
            for i, o in enumerate(orders):
                picked[o] = choice[i]
        elif search == "host":
            # the reference loop: five candidate predictions, five MSEs compared on the host (one sync per frame)
            L, slots = hip.lib(), hip.lib().vc_bits_slots()
            partial = torch.empty(len(ratios) * slots, dtype=torch.float64, device=tc.buf.device)
            sse = torch.empty(len(ratios), dtype=torch.float64, device=tc.buf.device)
            for i, cand in enumerate(ratios):
                pred = model.prediction_flowonly_t(tc, t1, t2, s1, s2, cand)
                hip.check(L.vc_sse_clamp01(hip.stream(), pred.view(), tc.view(), partial.data_ptr() + 8 * i * slots, slots), "vc_sse_clamp01")
            hip.check(L.vc_bits_reduce(hip.stream(), partial.data_ptr(), slots, len(ratios), sse.data_ptr()), "vc_bits_reduce")
            count = float(tc.n * tc.h * tc.w * tc.c)
            best, dr = 0.0, None
            for cand, e in zip(ratios, sse.cpu().tolist()):     # strict '>' keeps the first maximum, like the reference
                # (float32 like the reference's tensor arithmetic: 10 * log10(1 / mean))
                p = float(np.float32(10.0) * np.log10(np.float32(1.0) / np.float32(e / count))) if e > 0 else math.inf
                if p > best:
                    best, dr = p, cand
            if dr is None:
                raise hip.VcError("flow-resolution search found no finite PSNR")
            picked[orders[0]] = dr
        else:
            for o in orders:
                picked[o] = dr
        f1 = [features(g[1]) for g in group] if cache_features else None
        f2 = [features(g[2]) for g in group] if cache_features else None
        bits = BitCounter(tc.buf.device, max_rows=12 * n)
        y = model.forward_device(t1, t2, s1, s2, tc, level, dr, bits, flow=flow, feats1=f1, feats2=f2)
        x_hat = hip.nhwc_to_nchw(y)
        sizes = bits.totals().view(12, n).sum(0)
        is_ref = ICIP_LEVELS_16[orders[0]] < 3     # the deepest level is never referenced
        if is_ref:
            cl = hip.clamp01(y)
            batch_feats = model.feature_extractor.run(cl) if cache_features else None
        for i, o in enumerate(orders):
            decoded[o], stats[o] = x_hat[i:i + 1], sizes[i]
            if is_ref:
                nhwc[o] = cl.images(i, i + 1)
                if cache_features:
                    feats[o] = [f.images(i, i + 1) for f in batch_feats]
    if records is not None:
        for o in ICIP_ORDER_16[1:]:
            records.append((video, gop_index * 16 + o, ICIP_LEVELS_16[o], psnr_uint8(decoded[o], gop[o], h, w), stats[o],
                            float(h * w)))
    return decoded, picked


def code_sequence_icip2024(model, i_models, load_frame, n_frames, level, h=1080, w=1920, intra_size=16, search="device",
                           cache_features=True):
    """The reference's whole-sequence loop (ICIP2024/src/test.py:37-101, ``val_sequence_level``) for ANY frame count:
    coding order and frame types from get_order_typ_list (full GOP-16s plus the irregular tail the reference codes with
    whatever references are nearest), I-frames through ``i_models[level]`` (ELIC), B-frames through the flow-resolution
    search + FlowGuidedB, references = the two decoded frames (clamped to [0,1], at most 32 kept) closest in display
    order.  ``load_frame(i)`` returns the padded NCHW CUDA frame i.  One frame per pass (the regular GOPs of a long
    sequence go through :func:`code_gop_icip2024`, which batches levels); the only host sync is the final readback.
    Returns (psnr list, size list) indexed by display order, PSNR on uint8-rounded [:h,:w] crops, sizes in bits/(h*w)
    with the reference's 1080x1920 defaults."""
    from . import hip, icip2024
    from .layers import BitCounter
    order_list, typ_list = icip2024.get_order_typ_list(intra_size, n_frames)
    psnr, size = [None] * n_frames, [None] * n_frames
    buffer, buffer_order, feats = [], [], {}
    for order in order_list:
        cur = load_frame(order)
        tc = hip.nchw_to_nhwc(cur)
        if typ_list[order] == "I":
            bits = BitCounter(cur.device, max_rows=6)
            dec = hip.nhwc_to_nchw(i_models[int(level)].forward_device(tc, bits))
        else:
            ref1, ref2, o1, o2 = icip2024.select_references(None, order, buffer, buffer_order)
            s1, s2 = icip2024.get_scales(order, o1, o2)
            t1, t2 = hip.nchw_to_nhwc(ref1), hip.nchw_to_nhwc(ref2)
            flow, dr = None, None
            if search == "device":
                flow, _, _ = model.search_flow_t(tc, t1, t2, s1, s2)
            else:
                dr, _ = icip2024.get_best_down_ratio_prediction(model, ref1, ref2, s1, s2, cur)
            f1 = f2 = None
            if cache_features:
                for o, t in ((o1, t1), (o2, t2)):
                    if o not in feats:
                        feats[o] = model.feature_extractor.run(t)
                f1, f2 = [feats[o1]], [feats[o2]]
            bits = BitCounter(cur.device, max_rows=12)
            dec = hip.nhwc_to_nchw(model.forward_device(t1, t2, s1, s2, tc, level, dr, bits, flow=flow, feats1=f1, feats2=f2))
        psnr[order] = psnr_uint8(dec, cur, h, w)
        size[order] = bits.totals().sum() / float(h * w)
        buffer, buffer_order = icip2024.update_buffer(buffer, buffer_order, hip.clamp01_nchw(dec), order)
        feats = {o: f for o, f in feats.items() if o in buffer_order}     # frames that left the buffer can go
    return torch.stack(psnr).tolist(), torch.stack(size).tolist()


class GopGraph:
    """One GOP of B-frame coding captured ONCE as a HIP graph and replayed per GOP.

    The per-frame path is ~200 kernel launches issued from Python.  Every kernel of libvc_hip.so launches
    on the caller's stream without synchronising or allocating, so the whole dependency chain of a GOP
    (7 B-frames for LHBDC GOP-8, 15 for Flex GOP-16) captures into one graph: static input slots for the
    frames, intermediates in the graph's private pool, per-frame PSNR/bits left in static device tensors."""

    def __init__(self, model, h, w, video=0, kind="lhbdc", quality=None, pool=None, gops=1):
        """``pool``: a torch.cuda.graph_pool_handle() shared by several GopGraphs that are replayed one after the
        other (e.g. one per quality level): their intermediates then reuse the same memory, and the tensors a
        replay returns are only valid until the next replay of ANY graph of the pool.
        ``gops`` (LHBDC, Flex-Rate): number of consecutive GOPs coded per replay with their level passes batched together
        (:func:`code_gops_lhbdc`, :func:`code_gops_flex`); ``code`` then takes the frames of those GOPs in order."""
        self.model, self.h, self.w, self.video, self.kind, self.quality = model, h, w, video, kind, quality
        self.pool, self.gops = pool, int(gops)
        if self.gops != 1 and kind == "icip2024":
            raise ValueError("multi-GOP graphs exist for the LHBDC and Flex-Rate coders")
        self.orders = {"lhbdc": CODING_ORDER[2:], "flex": CODING_ORDER_16[2:], "icip2024": ICIP_ORDER_16[1:]}[kind]
        self.levels = {"lhbdc": HIER_LEVELS, "flex": HIER_LEVELS_16, "icip2024": ICIP_LEVELS_16}[kind]
        self.span = 8 if kind == "lhbdc" else 16
        self.graph = None
        self.static_in = None
        self.out_psnr = self.out_bits = None
        self.decoded = None

    def _run(self, frames):
        recs = []
        if self.kind == "lhbdc" and self.gops > 1:
            gops = [frames[9 * g:9 * g + 9] for g in range(self.gops)]
            dec = code_gops_lhbdc(self.model, gops, [(gp[0], gp[8]) for gp in gops], self.h, self.w, recs, self.video, 0)
        elif self.kind == "lhbdc":
            dec = code_gop_lhbdc(self.model, frames, frames[0], frames[8], self.h, self.w, recs, self.video, 0)
        elif self.kind == "icip2024":       # quality = level; flow-resolution search on the device
            dec, _ = code_gop_icip2024(self.model, frames, frames[0], frames[16], self.h, self.w, self.quality, recs,
                                       self.video, 0, search="device")
        elif self.gops > 1:
            gops = [frames[17 * g:17 * g + 17] for g in range(self.gops)]
            dec = code_gops_flex(self.model, gops, [(gp[0], gp[16]) for gp in gops], self.h, self.w, self.quality, recs, self.video, 0)
        else:
            dec = code_gop_flex(self.model, frames, frames[0], frames[16], self.h, self.w, self.quality, recs, self.video, 0)
        psnr = torch.stack([r[3] for r in recs])
        bits = torch.stack([r[4] for r in recs])
        return dec, psnr, bits

    def code(self, frames, gop_index=0, records=None):
        """frames: the GOP's padded NCHW device tensors (boundary frames taken as decoded I-frames)."""
        if self.graph is None:
            self.static_in = [f.clone() for f in frames]
            with torch.no_grad():
                self._run(self.static_in)                      # eager warm-up: packs weights, fills caches
                torch.cuda.synchronize()
                torch.cuda.empty_cache()                     # hand the eager warm-up's blocks back before capturing
                g = torch.cuda.CUDAGraph()
                # thread_local: only THIS thread's calls are checked against the capture -- a process-group watchdog
                # thread polling events (multi-GPU runs) must not invalidate it
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                    self.decoded, self.out_psnr, self.out_bits = self._run(self.static_in)
            self.graph = g
        for dst, src in zip(self.static_in, frames):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        if records is not None:     # (with gops > 1, gop_index is the index of the FIRST GOP of the replay)
            for i in range(len(self.orders) * self.gops):
                g, order = divmod(i, len(self.orders))
                order = self.orders[order]
                records.append((self.video, (gop_index + g) * self.span + order, self.levels[order], self.out_psnr[i].clone(),
                                self.out_bits[i].clone(), float(self.h * self.w)))
        return self.decoded


def shard_gops(num_gops, world_size, rank):
    """Contiguous GOP range [lo, hi) of ``rank`` (rank r gets [r*G/P, (r+1)*G/P))."""
    lo = (rank * num_gops) // world_size
    hi = ((rank + 1) * num_gops) // world_size
    return lo, hi


def gather_records(records, device, width=None):
    """All-gather per-frame R-D records over the default process group (RCCL on GPUs, gloo on CPU) and
    return them sorted in (video, frame) order on every rank.  Payload is a few KB: latency-bound.
    Records are (video, frame, level, psnr, bits, pixels) or, from the sequence loops, the same + is_intra; a rank
    without records (more ranks than GOPs) passes ``width`` or takes part with the 7-column layout."""
    import torch.distributed as dist
    if width is None:
        width = len(records[0]) if records else 7
    if any(len(r) != width for r in records):
        raise ValueError("records of one gather must all have the same number of fields")
    # psnr / bits are device scalars (no host sync inside a GOP): stack them on the device, one D2H at the very end
    cols = []
    for j in range(width):
        vals = [r[j] for r in records]
        if any(isinstance(v, torch.Tensor) for v in vals):
            cols.append(torch.stack([v.to(device=device, dtype=torch.float64).reshape(()) if isinstance(v, torch.Tensor)
                                     else torch.tensor(float(v), dtype=torch.float64, device=device) for v in vals]))
        else:
            cols.append(torch.tensor([float(v) for v in vals], dtype=torch.float64, device=device))
    local = torch.stack(cols, 1) if records else torch.zeros((0, width), dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        allrows = local
    else:
        world = dist.get_world_size()
        if dist.get_backend() != "nccl":        # gloo gathers host tensors (CPU tests, single-GPU functional checks)
            local, device = local.cpu(), torch.device("cpu")
        count = torch.tensor([local.shape[0], width], dtype=torch.int64, device=device)
        counts = [torch.zeros_like(count) for _ in range(world)]
        dist.all_gather(counts, count)
        counts = [c.cpu() for c in counts]
        if any(int(c[1]) != width for c in counts if int(c[0]) > 0):
            raise ValueError("ranks disagree on the record layout")
        m = max(1, int(max(int(c[0]) for c in counts)))
        padded = torch.zeros((m, width), dtype=torch.float64, device=device)
        padded[: local.shape[0]] = local
        parts = [torch.zeros_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
        allrows = torch.cat([p[: int(c[0])] for p, c in zip(parts, counts)], 0)
    allrows = allrows.cpu()
    key = allrows[:, 0] * 1e9 + allrows[:, 1]
    return allrows[torch.argsort(key, stable=True)]


def summarize(rows):
    """bpp = sum(bits)/sum(pixels); PSNR = mean of per-frame PSNR (utils.py:425-426), summed in frame order."""
    if rows.numel() == 0:
        return {"frames": 0, "bpp": math.nan, "psnr": math.nan}
    bits, pix, ps = 0.0, 0.0, 0.0
    for r in rows.tolist():
        ps += r[3]
        bits += r[4]
        pix += r[5]
    return {"frames": int(rows.shape[0]), "bpp": bits / pix, "psnr": ps / rows.shape[0]}


# ------------------------------------------------------------------------------------------------------
# Sequence-level evaluation (LHBDC/test/testing.py:89-196, test/utils.py:162-203,393-489)
# ------------------------------------------------------------------------------------------------------
def uvg_frame_indices(num_available, gop_size=8, test_size=2):
    """Source-frame index of every dataset item, as UVGTestDataset builds its list (utils.py:181-188):
    the first ``test_size*gop_size+1`` frames (all frames when ``test_size`` is falsy), with every interior
    GOP-boundary frame listed twice so that consecutive batches of ``gop_size+1`` items are I-B...B-I."""
    n = min(num_available, test_size * gop_size + 1) if test_size else num_available
    out = []
    for idx in range(n):
        out.append(idx)
        if idx % gop_size == 0 and idx != 0 and idx // gop_size != test_size:
            out.append(idx)
    return out


def gop_batches(indices, gop_size=8):
    """DataLoader(batch_size=gop_size+1, drop_last=True) over the item list (testing.py:117-120)."""
    step = gop_size + 1
    return [indices[i:i + step] for i in range(0, len(indices) - step + 1, step)]


def code_sequence_lhbdc(b_model, i_model, load_frame, num_available, h, w, video=0, gop_size=8, test_size=2,
                        gop_range=None, runner=None):
    """testing.py:125-188 for one video: I-frame 0 once, I-frame at the end of every GOP, 7 B-frames between.
    ``load_frame(idx)`` returns the padded NCHW device tensor of source frame ``idx``.  ``gop_range=(lo,hi)``
    restricts to a shard of GOPs (multi-GPU): a shard that does not start at GOP 0 re-codes its first
    boundary I-frame itself (intra frames do not depend on neighbours) but does not record it again.
    Returns records (video, frame_num, level|-1 for I, psnr, bits, pixels, is_intra)."""
    batches = gop_batches(uvg_frame_indices(num_available, gop_size, test_size), gop_size)
    lo, hi = gop_range if gop_range is not None else (0, len(batches))
    records = []

    def intra(idx, record):
        x = load_frame(idx)
        x_hat, tot = i_model.forward_device(x)
        if record:
            records.append((video, idx, -1, psnr_uint8(x_hat, x, h, w), tot.sum(), float(h * w), 1))
        return x_hat

    dec_last = None
    for g in range(lo, hi):
        idxs = batches[g]
        gop = [load_frame(i) for i in idxs]
        dec_first = dec_last if dec_last is not None else intra(idxs[0], record=(g == 0))
        dec_last = intra(idxs[-1], record=True)
        recs = []
        if runner is not None:
            frames = [dec_first] + gop[1:-1] + [dec_last]
            runner.code(frames, gop_index=g, records=recs)
        else:
            code_gop_lhbdc(b_model, gop, dec_first, dec_last, h, w, recs, video, g)
        records.extend(r + (0,) for r in recs)
    return records


# ------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: the whole test set (7 UVG sequences in testing.py:99-188), GOP-sharded across GPUs
# ------------------------------------------------------------------------------------------------------
def workload_plan(frames_per_video, gop_size=8, test_size=0):
    """The GOPs of a multi-video test set in the order testing.py walks them (video by video, GOP by GOP):
    [(video, gop_in_video, [source frame index of the gop_size+1 dataset items])].  ``frames_per_video``: frames
    available per video; ``test_size`` as UVGTestDataset (0 = all frames)."""
    plan = []
    for video, n in enumerate(frames_per_video):
        for g, idxs in enumerate(gop_batches(uvg_frame_indices(n, gop_size, test_size), gop_size)):
            plan.append((video, g, list(idxs)))
    return plan


def code_workload(plan, world_size, rank, intra, code_gops, gops_per_pass=1):
    """Rank ``rank``'s contiguous share (:func:`shard_gops`) of a :func:`workload_plan`, with NO data-path exchange:
    a GOP needs only its two boundary I-frames, and intra frames depend on nothing, so a shard that starts inside a
    video codes that boundary frame itself (it is RECORDED by the rank that owns the GOP it closes).

    ``intra(video, frame_idx) -> (decoded, record_tail)`` codes one I-frame; ``record_tail`` = (psnr, bits, pixels).
    ``code_gops(items, bounds) -> records`` codes the B-frames of up to ``gops_per_pass`` GOPs in one batched pass:
    ``items`` = plan entries, ``bounds`` = per GOP (decoded first, decoded last); returns 6-field records
    (video, frame, level, psnr, bits, pixels) with frame = gop_in_video * gop_size + order.
    Returns 7-field records (…, is_intra) in coding order; the union over all ranks is exactly the single-rank result."""
    lo, hi = shard_gops(len(plan), world_size, rank)
    records, batch, prev = [], [], None

    def flush():
        if batch:
            recs = code_gops([b[0] for b in batch], [(b[1], b[2]) for b in batch])
            records.extend(tuple(r) + (0,) for r in recs)
            batch.clear()

    for k in range(lo, hi):
        video, g, idxs = plan[k]
        if prev is not None and prev[0] == video and prev[1] == idxs[0]:
            dec_first = prev[2]
        else:
            dec_first, tail = intra(video, idxs[0])
            if g == 0:                            # frame 0 of a video belongs to its first GOP
                records.append((video, idxs[0], -1) + tuple(tail) + (1,))
        dec_last, tail = intra(video, idxs[-1])
        records.append((video, idxs[-1], -1) + tuple(tail) + (1,))
        batch.append((plan[k], dec_first, dec_last))
        if len(batch) == gops_per_pass:
            flush()
        prev = (video, idxs[-1], dec_last)
    flush()
    return records


class LhbdcWorkloadCoder:
    """The two callbacks of :func:`code_workload` on the HIP path: mbt2018_mean I-frames + LHBDC B-frames, ``G`` GOPs per
    pass with their hierarchy levels batched (one HIP graph per pass size when ``graph``)."""

    def __init__(self, b_model, i_model, load_frame, h, w, graph=True):
        self.b_model, self.i_model, self.load_frame, self.h, self.w, self.graph = b_model, i_model, load_frame, h, w, graph
        self.runners = {}

    def intra(self, video, idx):
        x = self.load_frame(video, idx)
        x_hat, tot = self.i_model.forward_device(x)
        return x_hat, (psnr_uint8(x_hat, x, self.h, self.w), tot.sum(), float(self.h * self.w))

    def code_gops(self, items, bounds):
        recs = []
        gops = [[self.load_frame(v, i) for i in idxs] for v, _, idxs in items]
        if self.graph:
            g = len(items)
            if g not in self.runners:
                self.runners[g] = GopGraph(self.b_model, self.h, self.w, gops=g)
            frames = []
            for gp, (first, last) in zip(gops, bounds):
                frames += [first] + gp[1:-1] + [last]
            self.runners[g].code(frames, gop_index=0, records=recs)
            out = []
            for j, r in enumerate(recs):          # the runner numbers GOPs 0..g-1: re-label with the plan's (video, gop)
                video, gop_in_video, _ = items[j // 7]
                out.append((video, gop_in_video * 8 + int(r[1]) % 8) + tuple(r[2:]))
            return out
        for (video, gop_in_video, _), gp, (first, last) in zip(items, gops, bounds):
            code_gop_lhbdc(self.b_model, gp, first, last, self.h, self.w, recs, video, gop_in_video)
        return recs


def code_sequence_flex(b_model, i_models, load_frame, num_available, h, w, quality, video=0, gop_size=16, test_size=2):
    """Flex-Rate.../test/testing.py:124-224 for one video and ONE operating point ``quality`` = (i_qual, {hierarchy level:
    (n, l)}) (an entry of FLEX_QUALITIES): intra frame 0 once and one at the end of every GOP through
    ``i_models[i_qual]``, the 15 B-frames between them with the (n, l) of their level (level-batched passes).
    Returns records (video, frame_num, level|-1 for I, psnr, bits, pixels, is_intra) like code_sequence_lhbdc."""
    i_model = i_models[quality[0]]
    batches = gop_batches(uvg_frame_indices(num_available, gop_size, test_size), gop_size)
    records = []

    def intra(idx):
        x = load_frame(idx)
        x_hat, tot = i_model.forward_device(x)
        records.append((video, idx, -1, psnr_uint8(x_hat, x, h, w), tot.sum(), float(h * w), 1))
        return x_hat

    dec_last = None
    for g, idxs in enumerate(batches):
        gop = [load_frame(i) for i in idxs]
        dec_first = dec_last if dec_last is not None else intra(idxs[0])
        dec_last = intra(idxs[-1])
        recs = []
        code_gop_flex(b_model, gop, dec_first, dec_last, h, w, quality, recs, video, g)
        records.extend(r + (0,) for r in recs)
    return records


class RdTable:
    """Aggregation of TestInfographic (utils.py:393-489) without pandas: PSNR = mean of per-frame PSNR,
    bpp = sum(size)/sum(pixels), grouped like print_per_level / per_video_level / per_frame_type."""

    def __init__(self):
        self.rows = []   # (video, level, frame_num, frame_type, psnr, size, pixels)

    def update(self, frame_type, frame_num, level, video, psnr, size, pixels):
        self.rows.append((video, level, int(frame_num), frame_type, float(psnr), float(size), float(pixels)))

    def extend_from_records(self, rows, level):
        for r in rows:
            intra = len(r) > 6 and int(r[6]) == 1
            self.update("I" if intra else "B", r[1], level, int(r[0]), r[3], r[4], r[5])

    def _group(self, key):
        out = {}
        for row in self.rows:
            acc = out.setdefault(key(row), [0.0, 0.0, 0.0, 0])
            acc[0] += row[4]; acc[1] += row[5]; acc[2] += row[6]; acc[3] += 1
        return {k: {"psnr": v[0] / v[3], "bpp": v[1] / v[2], "frames": v[3]} for k, v in sorted(out.items())}

    def per_level(self):
        return self._group(lambda r: r[1])

    def per_video_level(self):
        return self._group(lambda r: (r[0], r[1]))

    def per_level_frame_type(self):
        return self._group(lambda r: (r[1], r[3]))
