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

    def infer(self, image_name, rgb_u8, device, prepared=None):
        maps = prepared if prepared is not None else self.prepare(image_name, rgb_u8)
        h, w = rgb_u8.shape[:2]
        g = lambda k: torch.from_numpy(maps[k]).to(device, non_blocking=True) if k in maps else None      # DMA when the map is pinned
        mask = g("mask")
        return {"depth": g("depth"), "normal": g("normal"),
                "mask": mask.bool() if mask is not None else torch.ones((h, w), dtype=torch.bool, device=device)}


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
