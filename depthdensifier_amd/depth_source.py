"""Where per-view depth / normal / mask come from.

The reference runs MoGe-2 per image (``scripts/test.py:101-106, 161-168``).  ``moge`` and its weights
are not available offline, so the source is pluggable:

* ``MoGeSource``   -- ``MoGeModel.from_pretrained(checkpoint).infer(image)`` when ``moge`` imports;
* ``CachedSource`` -- precomputed maps ``<cache_dir>/<image stem>.npz`` with ``depth`` (H,W) f32/f16,
  ``mask`` (H,W) bool, optional ``normal`` (H,W,3) -- or the same arrays as ``<stem>_depth.npy`` /
  ``<stem>_mask.npy`` / ``<stem>_normal.npy`` -- e.g. dumped once on a machine that has MoGe
  (``dump_cache`` below, ``tools/dump_moge_cache.py``).

Every source returns DEVICE tensors: the maps go from the producer to the densify kernels
without the reference's ``.cpu().numpy()`` round trip (``scripts/test.py:166-168``).
"""

from __future__ import annotations

import os
import threading
from pathlib import Path
from typing import Optional

import numpy as np
import torch


class StagingSlot:
    """Page-locked host buffers for ONE view's maps, reused round-robin by the pipeline's I/O threads: a map read into
    pinned memory is uploaded by DMA, asynchronously, instead of through the driver's pageable staging copy on the main
    thread (3.6 ms per 1080p view).  Buffers are allocated on first use, inside the I/O thread (page-locking 41 MB costs
    ~30 ms -- hidden there, ruinous per view on the main thread, which round 1 tried), and kept for the views that follow.
    ``release`` is called by the consumer once the uploads are enqueued; ``wait`` by the next producer of the slot."""

    # page-locked bytes all slots of the process may hold (12 MP views need 244 MB per slot); beyond it a slot falls back to
    # ordinary memory -- still correct, the upload is then the driver's synchronous pageable copy again
    budget = int(os.environ.get("DD_PINNED_BUDGET_MB", "4096")) << 20
    _lock = threading.Lock()

    def __init__(self):
        self._bufs: dict = {}
        self._free: Optional[torch.cuda.Event] = None
        self._ev: Optional[torch.cuda.Event] = None      # the slot's own event, recorded natively by dd_upload_async (event_handle)

    def array(self, key: str, shape, dtype) -> np.ndarray:
        tdt = torch.from_numpy(np.empty(0, dtype=dtype)).dtype
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != tdt:
            need = int(np.prod(shape)) * np.dtype(dtype).itemsize
            with StagingSlot._lock:
                if t is not None and t.is_pinned():
                    StagingSlot.budget += t.numel() * t.element_size()
                pin = StagingSlot.budget >= need
                if pin:
                    StagingSlot.budget -= need
            t = torch.empty(tuple(shape), dtype=tdt, pin_memory=pin)
            self._bufs[key] = t
        return t.numpy()

    def pointer(self, key: str) -> int:
        """Address of the buffer ``array(key, ...)`` returned last (for the native readers, ``dd_npy_read``)."""
        return self._bufs[key].data_ptr()

    def put(self, key: str, arr: np.ndarray) -> np.ndarray:
        dst = self.array(key, arr.shape, arr.dtype)
        np.copyto(dst, arr)
        return dst

    def wait(self) -> None:
        if self._free is not None:
            self._free.synchronize()
            self._free = None

    def release(self, stream) -> None:
        ev = torch.cuda.Event()
        ev.record(stream)
        self._free = ev

    def event_handle(self, stream) -> int:
        """The raw handle of the slot's own event for a native call that records it behind the uploads (``dd_upload_async``); the
        caller then calls ``released_natively()``."""
        if self._ev is None:
            self._ev = torch.cuda.Event()
            self._ev.record(stream)                  # (creates the underlying event)
        return self._ev.cuda_event

    def released_natively(self) -> None:
        self._free = self._ev


_TORCH_OF = {"float32": torch.float32, "float16": torch.float16, "uint8": torch.uint8, "bool": torch.bool, "float64": torch.float64}


class DepthSource:
    def cached_rgb(self, image_name: str) -> Optional[np.ndarray]:
        """The image at processing resolution as (H,W,3) uint8 if the source holds it (``CachedSource`` with ``<stem>_rgb.npy``),
        else None: the pipeline then decodes and resizes the image file like the reference (``scripts/test.py:145-152``)."""
        return None

    def prepare(self, image_name: str, rgb_u8: np.ndarray, staging: Optional[StagingSlot] = None):
        """Optional host-side stage (file reads, decoding) that the pipeline may run ahead on an I/O thread;
        whatever it returns is handed to ``infer`` as ``prepared``.  Launches nothing on the GPU.  ``staging``: pinned
        buffers to leave the maps in."""
        return None

    def infer(self, image_name: str, rgb_u8: np.ndarray, device: torch.device, prepared=None) -> dict:
        """``{'depth': (H,W) float, 'normal': (H,W,3) float32 | None, 'mask': (H,W) bool}`` on ``device``."""
        raise NotImplementedError


class MoGeSource(DepthSource):
    def __init__(self, checkpoint: Path, device: torch.device):
        from moge.model.v2 import MoGeModel          # scripts/test.py:4
        self.model = MoGeModel.from_pretrained(checkpoint).to(device).eval()     # :104-105

    def infer(self, image_name, rgb_u8, device, prepared=None):
        x = torch.from_numpy(rgb_u8).to(device).permute(2, 0, 1).unsqueeze(0).float() / 255.0   # :154-155
        with torch.no_grad():
            out = self.model.infer(x)                # :161-162
        sq = lambda t: None if t is None else t.squeeze(0)
        return {"depth": sq(out["depth"]), "normal": sq(out.get("normal")), "mask": sq(out["mask"]).bool()}


class CachedSource(DepthSource):
    accepts_staging = True            # prepare() can leave the maps in the pipeline's pinned staging buffers

    def __init__(self, cache_dir: Path):
        self.dir = Path(cache_dir)
        if not self.dir.is_dir():
            raise FileNotFoundError(f"depth cache directory not found: {self.dir}")
        self._stems: dict = {}            # image name -> stem of its files (looked up once)
        self._npy: dict = {}              # stem -> None (no .npy layout) or {key: (path bytes, dtype code, numpy dtype, shape)} from the headers
        self._lock = threading.Lock()

    def _stem(self, image_name: str) -> str:
        got = self._stems.get(image_name)
        if got is None:
            got = self._stems[image_name] = self._find_stem(image_name)
        return got

    def _find_stem(self, image_name: str) -> str:
        # COLMAP image names may carry sub-folders ("cam1/0001.jpg"): a cache laid out the same way wins, so that two
        # cameras' "0001" do not collide; otherwise the flat <stem> files
        nested, flat = str(Path(image_name).with_suffix("")), Path(image_name).stem
        return nested if nested != flat and any((self.dir / (nested + ext)).exists() for ext in (".npz", "_depth.npy")) else flat

    def cached_rgb(self, image_name):
        """``<stem>_rgb.npy`` written by ``dump_cache(..., with_rgb=True)``: the image already decoded and resized (decoding a
        1080p PNG costs 30-40 ms of CPU, reading 6 MB does not) -- memory-mapped, the pipeline copies it into its staging slot."""
        f = self.dir / (self._stem(image_name) + "_rgb.npy")
        return np.load(f, mmap_mode="r") if f.exists() else None

    # ---- the .npy layout read natively (csrc/ddingest.hip): an I/O thread holds the interpreter lock for a few microseconds per view,
    # not for the header parsing and array handling of np.load -- sixteen such threads made every call of the main thread queue for
    # the lock (profiles/r06_bench_pipeline.txt) ----
    _NPY_DTYPES = {0: np.float32, 1: np.float16, 2: np.uint8, 3: np.bool_}

    def _npy_files(self, stem: str):
        """{key: (path, dtype code, numpy dtype, shape)} of the view's .npy files from their headers, or None (no .npy layout)."""
        got = self._npy.get(stem, False)
        if got is not False:
            return got
        import ctypes as C
        from ._lib import lib
        files = {}
        for key in ("depth", "mask", "normal", "rgb"):
            f = self.dir / f"{stem}_{key}.npy"
            if not f.exists():
                continue
            dt, nd, shape = C.c_int32(), C.c_int32(), (C.c_int64 * 4)()
            if lib.dd_npy_header(str(f).encode(), C.byref(dt), C.byref(nd), shape, None) < 0:
                files = None                     # (an element type the native reader does not take, say float64: np.load handles it)
                break
            files[key] = (str(f).encode(), int(dt.value), self._NPY_DTYPES[int(dt.value)], tuple(int(shape[k]) for k in range(nd.value)))
        if files is not None and "depth" not in files:
            files = None
        with self._lock:
            self._npy[stem] = files
        return files

    def _read_native(self, spec, key: str, staging: "StagingSlot") -> np.ndarray:
        import ctypes as C
        from ._lib import lib
        path, code, dtype, shape = spec
        dst = staging.array(key, shape, dtype)
        rc = lib.dd_npy_read(path, code, len(shape), (C.c_int64 * 4)(*shape), staging.pointer(key), dst.nbytes)
        if rc < 0:
            raise OSError(f"libddcore: {lib.dd_ingest_last_error().decode('utf-8', 'replace')}")
        return dst

    def read_rgb(self, image_name: str, hw: tuple, staging: "StagingSlot") -> Optional[np.ndarray]:
        """The cached image at processing resolution read straight into the staging slot, or None (no such file, or another size)."""
        files = self._npy_files(self._stem(image_name))
        if not files or "rgb" not in files or files["rgb"][3] != (hw[0], hw[1], 3) or files["rgb"][1] != 2:
            return None
        return self._read_native(files["rgb"], "rgb", staging)

    def prepare(self, image_name, rgb_u8, staging=None):
        stem = self._stem(image_name)
        if staging is not None:
            files = self._npy_files(stem)
            if files is not None:
                h, w = rgb_u8.shape[:2]
                if files["depth"][3] != (h, w):
                    raise ValueError(f"{files['depth'][0].decode()}: depth is {files['depth'][3]}, image at processing resolution is {(h, w)}")
                return {k: self._read_native(files[k], k, staging) for k in ("depth", "mask", "normal") if k in files}
        f = self.dir / (stem + ".npz")
        keep = (lambda k, a: staging.put(k, a)) if staging is not None else (lambda k, a: np.asarray(a))
        if f.exists():
            with np.load(f) as z:
                maps = {k: keep(k, z[k]) for k in ("depth", "mask", "normal") if k in z.files}
        else:
            f = self.dir / (stem + "_depth.npy")
            if not f.exists():
                raise FileNotFoundError(f"no cached depth for {image_name}: {self.dir / (stem + '.npz')} or {f}")
            mode = "r" if staging is not None else None          # straight from the page cache into the pinned buffer
            maps = {"depth": keep("depth", np.load(f, mmap_mode=mode))}
            for k in ("mask", "normal"):
                g = self.dir / f"{stem}_{k}.npy"
                if g.exists():
                    maps[k] = keep(k, np.load(g, mmap_mode=mode))
        h, w = rgb_u8.shape[:2]
        if maps["depth"].shape != (h, w):
            raise ValueError(f"{f}: depth is {maps['depth'].shape}, image at processing resolution is {(h, w)}")
        return maps

    def upload_staged(self, maps: dict, rgb_u8: np.ndarray, slot: StagingSlot, device: torch.device, copy_stream=None, fork_event=None, timing=None,
                      dest: Optional[dict] = None):
        """The maps ``prepare(..., staging=slot)`` left in the slot and the image (``slot.put("rgb", ...)``) to the device with ONE native
        call (``dd_upload_async``: the copies and the event that frees the slot) -> (``infer``'s dictionary, the image on the device).
        ``copy_stream`` (with ``fork_event``, an event of the caller's): the copies run on that stream -- beside the kernels of the
        views before, not in line with them (41 MB per 1080p view: 0.76 ms of PCIe against 0.03 ms of kernels) -- ordered behind what the
        current stream has enqueued so far (the destination blocks may have had readers there) and in front of what it enqueues next.
        ``dest``: device tensors of the caller's to copy INTO (keys as ``maps`` + ``"rgb"``; the pipeline's resident group stacks) --
        nothing is allocated, and with ``fork_event=None`` the copy stream waits for nothing: ordering against earlier readers of
        those tensors is the caller's."""
        import ctypes as C
        from ._lib import lib
        keys = [k for k in ("depth", "mask", "normal") if k in maps]
        if dest is not None:
            out, rgb_dev = {k: dest[k] for k in keys}, dest["rgb"]
        else:
            out = {k: torch.empty(maps[k].shape, dtype=_TORCH_OF[maps[k].dtype.name], device=device) for k in keys}
            rgb_dev = torch.empty(rgb_u8.shape, dtype=torch.uint8, device=device)
        n = len(keys) + 1
        src = (C.c_void_p * n)(*[slot.pointer(k) for k in keys], slot.pointer("rgb"))
        dst = (C.c_void_p * n)(*[out[k].data_ptr() for k in keys], rgb_dev.data_ptr())
        size = (C.c_int64 * n)(*[maps[k].nbytes for k in keys], rgb_u8.nbytes)
        stream = torch.cuda.current_stream(device)
        ev = slot.event_handle(stream)
        if copy_stream is not None and fork_event is not None:
            lib.dd_stream_fork(fork_event.cuda_event, stream.cuda_stream, copy_stream.cuda_stream)
        if timing is not None:                        # (a measurement run: how long the copies themselves take on their stream)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(copy_stream or stream)
        if lib.dd_upload_async(n, src, dst, size, ev, (copy_stream or stream).cuda_stream) < 0:
            raise RuntimeError(f"libddcore: {lib.dd_ingest_last_error().decode('utf-8', 'replace')}")
        if timing is not None:
            t1.record(copy_stream or stream)
            timing.append((t0, t1))
        if copy_stream is not None:
            lib.dd_stream_wait(stream.cuda_stream, ev)
        slot.released_natively()
        h, w = rgb_u8.shape[:2]
        mask = out.get("mask")
        if mask is not None and mask.dtype != torch.bool:
            mask = mask.bool()
        return {"depth": out["depth"], "normal": out.get("normal"),
                "mask": mask if mask is not None else torch.ones((h, w), dtype=torch.bool, device=device)}, rgb_dev

    def infer(self, image_name, rgb_u8, device, prepared=None):
        maps = prepared if prepared is not None else self.prepare(image_name, rgb_u8)
        h, w = rgb_u8.shape[:2]
        g = lambda k: torch.from_numpy(maps[k]).to(device, non_blocking=True) if k in maps else None      # DMA when the map is pinned
        mask = g("mask")
        return {"depth": g("depth"), "normal": g("normal"),
                "mask": mask.bool() if mask is not None else torch.ones((h, w), dtype=torch.bool, device=device)}


_PREFETCHERS: dict = {}              # (threads, slots) -> [handle, slot bytes, the slots' events, weak reference to the feeder using it]
_PREFETCHERS_LOCK = threading.Lock()


class _Staged:
    """What ``upload_staged`` needs to know of an array that lies in a staging slot: shape, element type, bytes."""
    __slots__ = ("shape", "dtype", "nbytes")

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize


class _NativeSlot:
    """One job of the native prefetcher as the pipeline's staging slot: addresses of the view's arrays, the event that frees it."""

    def __init__(self, feeder: "NativeFeeder", ticket: int, base: int, offsets: dict):
        self.feeder, self.ticket, self.base, self.offsets = feeder, ticket, base, offsets
        self._bufs = offsets                       # (the keys the slot holds, as StagingSlot has them)

    def pointer(self, key: str) -> int:
        return self.base + self.offsets[key]

    def event_handle(self, stream) -> int:
        return self.feeder.event_handle(self.ticket, stream)

    def released_natively(self) -> None:
        self.feeder.release(self.ticket)


class NativeFeeder:
    """The views' cached ``.npy`` files read ahead by native threads (``dd_prefetch_*``, csrc/ddingest.hip) instead of a Python thread
    pool: with no second Python thread nobody competes with the main thread for the interpreter lock (16 prefetch threads cost the
    loop 0.6-1.3 ms per view in waiting for it: profiles/r06_bench_pipeline.txt).  ``get(k)`` -> what ``fetch`` returns."""

    KEYS = ("depth", "mask", "normal", "rgb")
    CODES = {"depth": -1, "mask": 3, "normal": 0, "rgb": 2}      # DD_NPY_*: the depth may be float32 or float16
    DTYPES = {0: np.float32, 1: np.float16, 2: np.uint8, 3: np.bool_}

    def __init__(self, source: "CachedSource", names, sizes: dict, threads: int, ahead: int):
        import ctypes as C
        from ._lib import lib
        self._C, self._lib = C, lib
        self.source, self.names, self.sizes = source, list(names), sizes
        first = source._stem(self.names[0])
        self.keys = [k for k in self.KEYS if (source.dir / f"{first}_{k}.npy").exists()]
        if "depth" not in self.keys or "rgb" not in self.keys:
            raise FileNotFoundError("the cache holds no <stem>_depth.npy / <stem>_rgb.npy")
        w0, h0 = sizes[self.names[0]]
        for key in self.keys:                        # the first view's files must be what its size says (an image cached at another size: not this path)
            dt, nd, shape = C.c_int32(), C.c_int32(), (C.c_int64 * 4)()
            if lib.dd_npy_header(str(source.dir / f"{first}_{key}.npy").encode(), C.byref(dt), C.byref(nd), shape, None) < 0 \
                    or (int(shape[0]), int(shape[1])) != (h0, w0) or (self.CODES[key] >= 0 and int(dt.value) not in ((2, 3) if key == "mask" else (self.CODES[key],))):
                raise FileNotFoundError(f"{first}_{key}.npy is not the {key} of a {w0} x {h0} view")
        self.ahead = max(1, int(ahead))
        self.nslots = self.ahead + 2
        self.slot_bytes = max(sum(self._bytes(k, sizes[n]) for k in self.keys) + 64 * len(self.keys) for n in self.names)
        # one prefetcher per process and shape of use, kept across scans (scripts/run_batch.py runs scan after scan): its slots are
        # page-locked memory, and locking 1.4 GB of it again for every scan costs more than reading the scan
        import weakref
        key = (int(threads), self.nslots)
        self._h = None
        with _PREFETCHERS_LOCK:
            have = _PREFETCHERS.get(key)
            if have is not None:
                # a scan that ended in an exception may have left its feeder open, with jobs read ahead in the slots: they are given
                # back before this scan queues its own (scans of a process run one after the other; two at a time are not supported)
                before = have[3]() if have[3] is not None else None
                if before is not None:
                    before.close()
                if have[1] < self.slot_bytes:
                    lib.dd_prefetch_destroy(have[0])
                    del _PREFETCHERS[key]
                    have = None
            if have is None:
                h = C.c_void_p()
                if lib.dd_prefetch_create(int(threads), self.nslots, self.slot_bytes, C.byref(h)) < 0:
                    raise RuntimeError(lib.dd_ingest_last_error().decode("utf-8", "replace"))
                have = _PREFETCHERS[key] = [h, self.slot_bytes, [None] * self.nslots, None]
            have[3] = weakref.ref(self)
        self._h, _, self._events, _ = have
        self._submitted = 0
        self._tickets: dict = {}          # view index -> ticket (tickets count on over the scans of a process)
        self._offsets: dict = {}
        for _ in range(min(self.ahead, len(self.names))):
            self._submit()

    @staticmethod
    def _bytes(key: str, size) -> int:
        w, h = size
        return h * w * {"depth": 4, "mask": 1, "normal": 12, "rgb": 3}[key]

    def _submit(self) -> None:
        C, k = self._C, self._submitted
        if k >= len(self.names):
            return
        name = self.names[k]
        w, h = self.sizes[name]
        stem = self.source._stem(name)
        n = len(self.keys)
        offs, at = {}, 0
        for key in self.keys:
            offs[key], at = at, at + ((self._bytes(key, (w, h)) + 63) & ~63)
        paths = (C.c_char_p * n)(*[str(self.source.dir / f"{stem}_{key}.npy").encode() for key in self.keys])
        codes = (C.c_int32 * n)(*[self.CODES[key] for key in self.keys])
        nds = (C.c_int32 * n)(*[3 if key in ("normal", "rgb") else 2 for key in self.keys])
        shapes = (C.c_int64 * (4 * n))(*[x for key in self.keys for x in ((h, w, 3, 1) if key in ("normal", "rgb") else (h, w, 1, 1))])
        t = self._lib.dd_prefetch_submit(self._h, n, paths, codes, nds, shapes, (C.c_int64 * n)(*[offs[key] for key in self.keys]))
        if t < 0:
            raise RuntimeError(f"prefetcher: {self._lib.dd_ingest_last_error().decode('utf-8', 'replace')}")
        self._tickets[k], self._offsets[k] = int(t), offs
        self._submitted += 1

    def get(self, k: int):
        C = self._C
        base, found = C.c_void_p(), (C.c_int32 * 8)()
        t = self._tickets.pop(k)
        if self._lib.dd_prefetch_wait(self._h, t, C.byref(base), found) < 0:
            why = self._lib.dd_ingest_last_error().decode("utf-8", "replace")
            self._offsets.pop(k, None)
            self._lib.dd_prefetch_release(self._h, t, None)      # (the slot goes on; the caller reads this view its other way)
            self._submit()
            raise OSError(f"libddcore: {why}")
        w, h = self.sizes[self.names[k]]
        maps = {}
        for i, key in enumerate(self.keys):
            if key != "rgb":
                maps[key] = _Staged((h, w, 3) if key == "normal" else (h, w), self.DTYPES[int(found[i])])
        return _Staged((h, w, 3), np.uint8), maps, _NativeSlot(self, t, int(base.value), self._offsets.pop(k))

    def event_handle(self, ticket: int, stream) -> int:
        s = ticket % self.nslots
        if self._events[s] is None:
            self._events[s] = torch.cuda.Event()
            self._events[s].record(stream)           # (creates the underlying event)
        return self._events[s].cuda_event

    def release(self, ticket: int) -> None:
        ev = self._events[ticket % self.nslots]
        if self._lib.dd_prefetch_release(self._h, ticket, None if ev is None else ev.cuda_event) < 0:
            raise RuntimeError(self._lib.dd_ingest_last_error().decode("utf-8", "replace"))
        self._submit()                               # the slot is spoken for again: the next view to come

    def close(self) -> None:
        """Jobs that were read ahead and never taken are waited for and given back: the prefetcher goes on to the next scan."""
        if self._h is not None:
            for k, t in sorted(self._tickets.items()):
                self._lib.dd_prefetch_wait(self._h, t, None, None)
                self._lib.dd_prefetch_release(self._h, t, None)
            self._tickets.clear()
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass


def dump_cache(source: DepthSource, image_dir: Path, cache_dir: Path, device: torch.device, factor: int = 1,
               fp16_depth: bool = False, layout: str = "npy", with_rgb: bool = False) -> int:
    """Run ``source`` over every image of ``image_dir`` (resized like the pipeline, ``scripts/test.py:145-152``) and
    write the maps ``CachedSource`` reads back; returns the number of images written.  ``layout="npy"`` (default)
    writes ``<stem>_depth.npy`` / ``_mask.npy`` / ``_normal.npy`` -- a plain read, 10x faster to load than the
    single-file ``layout="npz"`` whose zip container is CRC-checked on every read (4.7 vs 50 ms per 1080p view).
    ``with_rgb``: also ``<stem>_rgb.npy``, the image at processing resolution (``CachedSource.cached_rgb``)."""
    if layout not in ("npy", "npz"):
        raise ValueError("layout must be 'npy' or 'npz'")
    from PIL import Image as PILImage
    cache_dir.mkdir(parents=True, exist_ok=True)
    n = 0
    for f in sorted(p for p in Path(image_dir).iterdir() if p.suffix.lower() in (".png", ".jpg", ".jpeg")):
        img = PILImage.open(f).convert("RGB")
        w, h = img.size
        rgb = np.array(img.resize((w // factor, h // factor), PILImage.Resampling.LANCZOS))
        maps = source.infer(f.name, rgb, device)
        out = {"depth": maps["depth"].float().cpu().numpy().astype(np.float16 if fp16_depth else np.float32),
               "mask": maps["mask"].cpu().numpy().astype(bool)}
        if maps.get("normal") is not None:
            out["normal"] = maps["normal"].float().cpu().numpy()
        if layout == "npz":
            np.savez(cache_dir / (f.stem + ".npz"), **out)
        else:
            for k, v in out.items():
                np.save(cache_dir / f"{f.stem}_{k}.npy", v)
        if with_rgb:            # the resized image itself: a scan run again from the cache then decodes nothing
            np.save(cache_dir / f"{f.stem}_rgb.npy", rgb)
        n += 1
    return n


def make_depth_source(checkpoint: Path, cache_dir: Optional[Path], device: torch.device) -> DepthSource:
    if cache_dir is not None:
        return CachedSource(cache_dir)
    try:
        return MoGeSource(checkpoint, device)
    except ImportError as e:
        raise ImportError("MoGe is not installed; pass --config.moge.cache-dir with precomputed depth maps "
                          "(<stem>.npz: depth, mask, normal)") from e
