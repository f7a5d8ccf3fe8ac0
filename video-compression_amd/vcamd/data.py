"""File-backed sequence ingest for the evaluation loops and the CLI scripts.

Mirrors the reference's data path -- ``UVGTestDataset`` + a 4-worker ``DataLoader`` (LHBDC/test/utils.py:162-203,
test/testing.py:117-131) and ``process_frame`` of the CLI scripts (LHBDC/encode_B.py:58-64) -- as a pipeline that keeps
the GPU fed:

    PNG folder / raw 8-bit YUV 4:2:0 file
        -> decode workers (threads; the zlib / numpy work releases the GIL)
        -> ring of PINNED uint8 [h, w, 3] host buffers
        -> asynchronous H2D copies on a side stream into a ring of device uint8 buffers
        -> ``vc_u8hwc_to_f32nchw_pad`` on the caller's stream (x / 255, reflection pad to multiples of 64)
        -> the padded fp32 NCHW tensor ``load_frame(video, idx)`` hands to the codec loops (vcamd.gop).

The item list (which frames, boundary frames listed twice) is ``gop.uvg_frame_indices`` -- pinned against the
reference's own ``UVGTestDataset`` in tests/golden/uvg_dataset_indices.json.  Nothing here touches a pixel with a torch
operator; decoding a frame is host work that runs beside the GPU, and the reader reports how long the consumer had to
wait for it.
"""
import glob
import os
import queue
import re
import threading
import time

import numpy as np
import torch

from . import hip
from .gop import uvg_frame_indices


def natural_key(path):
    """natsort.natsorted order (utils.py:176): digit runs compare as numbers."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]


def read_png(path):
    """8-bit RGB [h, w, 3] (what ``imageio.imread`` returns for the reference's PNG frames)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8)


def write_png(path, rgb_u8):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(rgb_u8), "RGB").save(path, format="PNG", compress_level=1)


def yuv420_frame_to_rgb(buf, w, h):
    """One 8-bit planar 4:2:0 frame (Y, U, V planes: how UVG ships its sequences) -> RGB uint8 [h, w, 3].
    BT.709 limited range, chroma replicated to full resolution (nearest) -- the conversion ``ffmpeg -pix_fmt rgb24``
    performs up to its chroma interpolation; the reference only ever sees the PNGs produced that way."""
    y = buf[: w * h].reshape(h, w).astype(np.float32)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    u = buf[w * h: w * h + cw * ch].reshape(ch, cw).astype(np.float32)
    v = buf[w * h + cw * ch: w * h + 2 * cw * ch].reshape(ch, cw).astype(np.float32)
    u = np.repeat(np.repeat(u, 2, 0), 2, 1)[:h, :w] - 128.0
    v = np.repeat(np.repeat(v, 2, 0), 2, 1)[:h, :w] - 128.0
    yy = (y - 16.0) * (255.0 / 219.0)
    r = yy + (255.0 / 224.0) * 1.5748 * v
    g = yy - (255.0 / 224.0) * (0.1873 * u + 0.4681 * v)
    b = yy + (255.0 / 224.0) * 1.8556 * u
    return np.clip(np.rint(np.stack([r, g, b], -1)), 0, 255).astype(np.uint8)


class PngFolder:
    """One video = a folder of PNG frames in natural order."""

    def __init__(self, folder):
        self.paths = sorted(glob.glob(os.path.join(folder, "*.png")), key=natural_key)
        if not self.paths:
            raise hip.VcError(f"no PNG frames under {folder}")
        self.h, self.w = read_png(self.paths[0]).shape[:2]

    def __len__(self):
        return len(self.paths)

    def read(self, idx, out=None):
        rgb = read_png(self.paths[idx])
        if rgb.shape[:2] != (self.h, self.w):
            raise hip.VcError(f"{self.paths[idx]}: frame size changes inside the sequence")
        if out is None:
            return rgb
        np.copyto(out, rgb)
        return out


class Yuv420File:
    """One video = a raw 8-bit planar YUV 4:2:0 file of known size."""

    def __init__(self, path, w, h):
        self.path, self.w, self.h = path, w, h
        self.frame_bytes = w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)
        self.n = os.path.getsize(path) // self.frame_bytes
        if self.n == 0:
            raise hip.VcError(f"{path} holds no complete {w}x{h} 4:2:0 frame")
        self.map = np.memmap(path, dtype=np.uint8, mode="r")

    def __len__(self):
        return self.n

    def read(self, idx, out=None):
        rgb = yuv420_frame_to_rgb(self.map[idx * self.frame_bytes:(idx + 1) * self.frame_bytes], self.w, self.h)
        if out is None:
            return rgb
        np.copyto(out, rgb)
        return out


def dataset_items(frame_counts, gop_size=8, skip_frames=1, test_size=2):
    """``UVGTestDataset.frames`` as (video index, item index) pairs for videos of ``frame_counts`` source frames
    (utils.py:173-188): per video the first test_size*gop_size+1 items (all when ``test_size`` is falsy), interior GOP
    boundaries listed twice; item i is source frame i*skip_frames."""
    items = []
    for vi, n in enumerate(frame_counts):
        avail = (n + skip_frames - 1) // skip_frames
        items += [(vi, i) for i in uvg_frame_indices(avail, gop_size, test_size)]
    return items


def open_video(path, yuv_size=None):
    if os.path.isdir(path):
        return PngFolder(path)
    if yuv_size is None:
        raise hip.VcError(f"{path} is not a PNG folder; raw YUV files need their frame size (width, height)")
    return Yuv420File(path, *yuv_size)


class _Slot:
    __slots__ = ("pinned", "dev", "copied", "consumed_ev")

    def __init__(self, h, w, device):
        self.pinned = torch.empty((h, w, 3), dtype=torch.uint8).pin_memory()
        self.dev = torch.empty((h, w, 3), dtype=torch.uint8, device=device)
        self.copied = torch.cuda.Event()
        self.consumed_ev = None


class _Pending:
    """Result cell of one queued frame (a minimal future: the workers are plain daemon threads)."""
    __slots__ = ("done", "slot", "error")

    def __init__(self):
        self.done, self.slot, self.error = threading.Event(), None, None


class SequenceReader:
    """``UVGTestDataset`` semantics over a directory of videos, with decode / H2D running ahead of the consumer.

    ``data_path``: a directory whose sub-directories are PNG sequences (``video_names`` selects / orders them; default:
    all, sorted) -- or whose ``*.yuv`` files are raw 4:2:0 sequences when ``yuv_size=(w, h)`` is given.
    ``items``: the dataset's item list [(video index, source frame index)], boundary frames duplicated exactly as the
    reference does (``gop.uvg_frame_indices``).  ``prefetch(keys)`` queues frames in the order the consumer is going to
    ask for them; ``load_frame(video, idx)`` returns the padded fp32 NCHW CUDA tensor.  ``depth`` frames are in flight
    at most (pinned + device uint8 buffers of one frame each); a frame that was not announced is decoded on the spot.
    Worker threads are daemons and every wait is bounded by ``timeout`` seconds: a failing decode surfaces as an
    exception in ``load_frame``, never as a hang.
    """

    def __init__(self, data_path, video_names=None, gop_size=8, skip_frames=1, test_size=2, device="cuda:0", workers=4,
                 depth=12, yuv_size=None, timeout=120.0):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise hip.VcError("SequenceReader feeds the HIP path: it needs a CUDA device (there is no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if video_names is None:
            if yuv_size is None:
                video_names = sorted(d for d in os.listdir(data_path) if os.path.isdir(os.path.join(data_path, d)))
            else:
                video_names = sorted(f for f in os.listdir(data_path) if f.endswith(".yuv"))
        if not video_names:
            raise hip.VcError(f"no sequences under {data_path}")
        self.video_names = list(video_names)
        self.videos = [open_video(os.path.join(data_path, v), yuv_size) for v in self.video_names]
        self.h, self.w = self.videos[0].h, self.videos[0].w
        if any((v.h, v.w) != (self.h, self.w) for v in self.videos):
            raise hip.VcError("all sequences of one reader must have the same frame size")
        self.hp, self.wp = self.h + (64 - self.h % 64) % 64, self.w + (64 - self.w % 64) % 64
        self.gop_size, self.skip_frames, self.test_size = gop_size, skip_frames, test_size
        self.items = dataset_items([len(v) for v in self.videos], gop_size, skip_frames, test_size)
        self.timeout = float(timeout)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.free_slots = queue.Queue()
        for _ in range(max(2, depth)):
            self.free_slots.put(_Slot(self.h, self.w, self.device))
        self.jobs = queue.Queue()
        self.inflight = {}              # key -> list of _Pending (a key may be queued more than once)
        self.lock = threading.Lock()
        self.dispatch = threading.Lock()
        self.closed = False
        self.stats = {"frames": 0, "decode_s": 0.0, "h2d_s": 0.0, "wait_s": 0.0, "sync_loads": 0}
        self.threads = [threading.Thread(target=self._worker, name=f"vc-decode-{i}", daemon=True) for i in range(max(1, workers))]
        for t in self.threads:
            t.start()

    # -- producer side --------------------------------------------------------------------------------------------
    def _take(self):
        """Next (key, cell, slot) -- job AND ring slot are taken under one lock, so slots are granted in the order the
        frames were announced: a later frame can never hold the last slot while the consumer waits for an earlier one."""
        with self.dispatch:
            while not self.closed:
                try:
                    key, cell = self.jobs.get(timeout=0.2)
                    break
                except queue.Empty:
                    continue
            else:
                return None
            while not self.closed:
                try:
                    return key, cell, self.free_slots.get(timeout=0.2)
                except queue.Empty:
                    continue
            cell.error = hip.VcError("reader closed")
            cell.done.set()
            return None

    def _worker(self):
        torch.cuda.set_device(self.device)
        while not self.closed:
            taken = self._take()
            if taken is None:
                continue
            key, cell, slot = taken
            try:
                if slot.consumed_ev is not None:              # the conversion kernel that read slot.dev last time has finished
                    slot.consumed_ev.synchronize()
                    slot.consumed_ev = None
                video, idx = key
                t0 = time.perf_counter()
                self.videos[video].read(idx * self.skip_frames, out=slot.pinned.numpy())
                t1 = time.perf_counter()
                with torch.cuda.stream(self.copy_stream):
                    slot.dev.copy_(slot.pinned, non_blocking=True)
                    slot.copied.record(self.copy_stream)
                slot.copied.synchronize()                     # the pinned buffer may be overwritten after this
                t2 = time.perf_counter()
                with self.lock:
                    self.stats["decode_s"] += t1 - t0
                    self.stats["h2d_s"] += t2 - t1
                cell.slot = slot
            except BaseException as e:  # noqa: BLE001 -- handed to the consumer
                cell.error = e
                self.free_slots.put(slot)
            finally:
                cell.done.set()

    def prefetch(self, keys):
        """Queue frames in consumption order.  Not more than ``depth`` of them are held decoded at a time."""
        for key in keys:
            cell = _Pending()
            with self.lock:
                self.inflight.setdefault(tuple(key), []).append(cell)
            self.jobs.put((tuple(key), cell))

    # -- consumer side --------------------------------------------------------------------------------------------
    def load_frame(self, video, idx, out=None):
        key = (video, idx)
        with self.lock:
            cells = self.inflight.get(key)
            cell = cells.pop(0) if cells else None
            if cells is not None and not cells:
                del self.inflight[key]
        t0 = time.perf_counter()
        if cell is None:                                   # not announced: decode now, on this thread
            rgb = torch.from_numpy(self.videos[video].read(idx * self.skip_frames)).to(self.device)
            x = hip.frame_from_uint8(rgb, self.hp, self.wp, out=out)
            with self.lock:
                self.stats["sync_loads"] += 1
                self.stats["frames"] += 1
                self.stats["wait_s"] += time.perf_counter() - t0
            return x
        if not cell.done.wait(self.timeout):
            raise hip.VcError(f"frame {key} was not decoded within {self.timeout} s (frames must be requested in the order they were "
                              "announced to prefetch())")
        if cell.error is not None:
            raise hip.VcError(f"decoding frame {key} failed: {cell.error!r}") from cell.error
        slot = cell.slot
        torch.cuda.current_stream(self.device).wait_event(slot.copied)
        x = hip.frame_from_uint8(slot.dev, self.hp, self.wp, out=out)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot.consumed_ev = ev
        self.free_slots.put(slot)
        with self.lock:
            self.stats["frames"] += 1
            self.stats["wait_s"] += time.perf_counter() - t0
        return x

    def load(self, key):
        return self.load_frame(*key)

    def close(self):
        """Stops the workers (queued frames are dropped) and waits for them: no thread of the reader outlives it."""
        self.closed = True
        for t in self.threads:
            if t is not threading.current_thread():
                t.join(timeout=5.0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def write_synthetic_sequences(root, frames_by_video, names=None):
    """Test / bench helper: ``frames_by_video`` = list of lists of uint8 [h, w, 3] arrays -> <root>/<name>/im00001.png ..."""
    names = names or [f"seq{i:02d}" for i in range(len(frames_by_video))]
    for name, frames in zip(names, frames_by_video):
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        for i, f in enumerate(frames):
            write_png(os.path.join(d, f"im{i + 1:05d}.png"), f)
    return names
