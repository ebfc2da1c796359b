"""Host side of the depth->points hot path (PyTorch-ROCm tensors in, fused cloud out).

Mirrors, for this path only, what the reference does inline in
``scripts/test.py:203-244, 262-266`` ("script" semantics) and in
``COLMAPVisualizer.add_rgbd_pointcloud`` (``src/depthdensifier/visualizer.py:246-376``,
"viz" semantics).  All arithmetic runs in ``libddcore.so`` (hand-written HIP for
gfx950) through the C ABI of ``include/ddcore.h``; PyTorch only owns device
memory and the stream.  There is no CPU implementation here: without a GPU and
the built library these functions raise.
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from ._lib import DDCloudOut, DDViewBatch, check, lib

ArrayLike = Union[np.ndarray, torch.Tensor]

SEMANTICS = ("script", "viz")


# --------------------------------------------------------------------------------------
# camera blocks (host, float64 -> float32 once)
# --------------------------------------------------------------------------------------

def intrinsics_matrix(params_or_K: ArrayLike) -> np.ndarray:
    """(V,3,3) float64 intrinsics from either (V,4) ``fx,fy,cx,cy`` (``camera.params`` of a
    PINHOLE camera, ``scripts/test.py:81``) or (V,3,3) calibration matrices
    (``visualizer.py:320``; skew honoured)."""
    a = np.asarray(params_or_K.cpu() if isinstance(params_or_K, torch.Tensor) else params_or_K, dtype=np.float64)
    if a.ndim == 1 and a.shape[0] == 4:
        a = a[None]
    if a.ndim == 2 and a.shape == (3, 3):
        a = a[None]
    if a.ndim == 2 and a.shape[1] == 4:
        K = np.zeros((a.shape[0], 3, 3))
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = a[:, 0], a[:, 1], a[:, 2], a[:, 3], 1.0
        return K
    if a.ndim == 3 and a.shape[1:] == (3, 3):
        return a
    raise ValueError(f"intrinsics must be (V,4) fx,fy,cx,cy or (V,3,3); got {a.shape}")


def camera_blocks(intrinsics: ArrayLike, cam_from_world: ArrayLike) -> np.ndarray:
    """(V,32) float32 ``DDViewParams`` rows (``include/ddcore.h``).

    ``ray_to_world = R^T K^-1``, ``centre = -R^T t``, ``rot = R^T`` computed in
    float64 -- the fusion of ``scripts/test.py:79-90`` with the ``Rigid3d``
    inverse of ``:233`` (equivalently ``visualizer.py:320-334``).  Accepts
    (V,3,4) or (V,4,4) extrinsics like ``visualizer.py:325-327``.
    """
    K = intrinsics_matrix(intrinsics)
    E = np.asarray(cam_from_world.cpu() if isinstance(cam_from_world, torch.Tensor) else cam_from_world, dtype=np.float64)
    if E.ndim == 2:
        E = E[None]
    if E.ndim != 3 or E.shape[1] not in (3, 4) or E.shape[2] != 4:
        raise ValueError(f"cam_from_world must be (V,3,4) or (V,4,4); got {E.shape}")
    if K.shape[0] == 1 and E.shape[0] > 1:
        K = np.repeat(K, E.shape[0], axis=0)
    if K.shape[0] != E.shape[0]:
        raise ValueError(f"{K.shape[0]} intrinsics for {E.shape[0]} poses")
    R = E[:, :3, :3]
    t = E[:, :3, 3]
    Rt = np.transpose(R, (0, 2, 1))
    blocks = np.zeros((E.shape[0], 32), dtype=np.float64)
    blocks[:, 0:9] = (Rt @ np.linalg.inv(K)).reshape(-1, 9)
    blocks[:, 9:12] = -np.einsum("vij,vj->vi", Rt, t)
    blocks[:, 12:21] = Rt.reshape(-1, 9)
    return blocks.astype(np.float32)


# --------------------------------------------------------------------------------------
# result container
# --------------------------------------------------------------------------------------

@dataclass
class FusedCloud:
    """The fused dense cloud: ``final_point_cloud / final_colors / final_normals`` of
    ``scripts/test.py:264-266`` and the ``PointCloud`` of ``visualizer.py:71-80``, as device
    tensors, plus what the reference does not keep: ``view_offsets`` (V+1, int64),
    ``pixel_index`` (``y*W+x``) and ``view_index`` so order can be checked bit-exactly."""

    points: torch.Tensor                   # (N,3) float32
    colors: Optional[torch.Tensor]         # (N,3) uint8
    normals: Optional[torch.Tensor]        # (N,3) float32
    pixel_index: Optional[torch.Tensor]    # (N,) int32
    view_index: Optional[torch.Tensor]     # (N,) int32
    view_offsets: torch.Tensor             # (V+1,) int64
    name: str = "Dense Cloud"
    packed: Optional[torch.Tensor] = None  # (N,4) float32 rows x, y, z, bits(r | g<<8 | b<<16 | 255<<24): the 16-byte gather record
    rgb_passthrough: Optional[list] = None # viz semantics: per view, the image whose colours the reference hands on in a non-uint8 dtype

    def __len__(self) -> int:
        return int(self.points.shape[0])

    @classmethod
    def from_packed(cls, packed: torch.Tensor, view_offsets: torch.Tensor, name: str = "Dense Cloud", **fields) -> "FusedCloud":
        """A cloud held as 16-byte records (``DDCloudOut.xyz_rgba``): ``points`` / ``colors`` are strided views of it."""
        n = packed.shape[0]
        rgba = packed.view(torch.uint8).view(n, 16)
        return cls(points=packed[:, :3], colors=rgba[:, 12:15], normals=fields.get("normals"), pixel_index=fields.get("pixel_index"),
                   view_index=fields.get("view_index"), view_offsets=view_offsets, name=name, packed=packed)

    @property
    def counts(self) -> torch.Tensor:
        return self.view_offsets[1:] - self.view_offsets[:-1]

    def numpy(self) -> dict:
        """Host copies with the reference's dtypes (points float64 like ``scripts/test.py:264``)."""
        f = lambda x: None if x is None else x.cpu().numpy()
        return {
            "points": self.points.cpu().numpy().astype(np.float64),
            "colors": f(self.colors),
            "normals": f(self.normals),
            "pixel_index": f(self.pixel_index),
            "view_index": f(self.view_index),
            "view_offsets": self.view_offsets.cpu().numpy(),
        }


# --------------------------------------------------------------------------------------
# batch description
# --------------------------------------------------------------------------------------

def _gpu(x: Optional[ArrayLike], device: torch.device, dtype: Optional[torch.dtype] = None) -> Optional[torch.Tensor]:
    if x is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device, non_blocking=True).contiguous()


class _PinnedRing:
    """Small host arrays (camera blocks, a view's sparse points) on their way to the GPU.  A copy from ordinary memory
    makes the host wait until the stream has reached it -- behind the 41 MB of maps of a 1080p view that is ~1 ms for 128
    bytes.  Here the bytes are put into one of a few page-locked slots first and the copy is enqueued without waiting;
    a slot is reused only after the event recorded behind its copy has passed."""

    def __init__(self, slots: int = 64, nbytes: int = 1 << 18):
        self.nbytes, self.slots = nbytes, slots
        self._bufs: list = []
        self._events: list = []
        self._next = 0
        import threading
        self._lock = threading.Lock()        # (clouds are built from several threads: tests/test_streaming_calls.py, the batch driver)

    def upload(self, arr: np.ndarray, device: torch.device) -> torch.Tensor:
        arr = np.ascontiguousarray(arr)
        if arr.nbytes == 0 or arr.nbytes > self.nbytes:
            return torch.from_numpy(arr).to(device)
        with self._lock:
            return self._upload(arr, device)

    def stage(self, arr: np.ndarray):
        """``arr`` copied into a page-locked slot WITHOUT an upload of its own: ``(address, slot)`` for a native call that reads the
        slot on a stream (``dd_refine_fit_async`` copies the sparse points up itself); the caller then hands ``slot`` and an event
        recorded behind that stream operation to ``staged_until``.  None if the array does not fit a slot."""
        arr = np.ascontiguousarray(arr)
        if arr.nbytes == 0 or arr.nbytes > self.nbytes:
            return None
        with self._lock:
            if not self._bufs:
                self._allocate()
            k = self._next % self.slots
            self._next += 1
            if self._events[k] is not None:
                self._events[k].synchronize()
                self._events[k] = None
            self._views[k][:arr.nbytes] = arr.view(np.uint8).reshape(-1)
            return self._ptrs[k], k

    def staged_until(self, slot: int, stream) -> None:
        """The stream operation that reads the slot has been enqueued on ``stream``: the slot is free once the stream has passed here.
        (An event of the slot's own -- never one the caller re-records for something later: a reuse of the slot would wait for THAT.)"""
        ev = self._slot_events[slot]
        if ev is None:
            ev = self._slot_events[slot] = torch.cuda.Event()
        ev.record(stream)
        self._events[slot] = ev

    def _allocate(self) -> None:
        # one page-locked block for all slots (a pinned allocation costs ~0.3 ms whatever its size)
        block = torch.empty(self.slots * self.nbytes, dtype=torch.uint8, pin_memory=True)
        self._bufs = [block[i * self.nbytes:(i + 1) * self.nbytes] for i in range(self.slots)]
        self._views = [b.numpy() for b in self._bufs]
        self._ptrs = [b.data_ptr() for b in self._bufs]
        self._events = [None] * self.slots
        self._slot_events = [None] * self.slots

    def _upload(self, arr: np.ndarray, device: torch.device) -> torch.Tensor:
        if not self._bufs:
            self._allocate()
        k = self._next % self.slots
        self._next += 1
        if self._events[k] is not None:
            self._events[k].synchronize()
        self._views[k][:arr.nbytes] = arr.view(np.uint8).reshape(-1)
        dev = self._bufs[k][:arr.nbytes].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._events[k] = ev
        return dev.view(_TORCH_DTYPE[arr.dtype.name]).view(arr.shape)


_TORCH_DTYPE = {n: getattr(torch, n) for n in ("float32", "float64", "float16", "int32", "int64", "int16", "int8", "uint8", "bool")}

_small = _PinnedRing()


class _HostSlots:
    """Page-locked slots of 8 int64 for results on their way BACK from the GPU (a cloud's row count and the scan status words):
    the copies are enqueued without making the host wait, an event behind them says when the slot can be read.  A slot belongs to
    its ``PendingCheck`` until that has been read (or dropped); a caller that asks for many checks before reading any -- scene after
    scene, results at the end -- makes the ring grow instead of coming round onto a slot that is still waited for."""

    def __init__(self, slots: int = 128, words: int = 8):
        self.block, self.words = slots, words
        self._slots: list = []             # page-locked (words,) int64 views
        self._owned: list = []
        self._next = 0
        import threading
        self._lock = threading.Lock()

    def _grow(self) -> None:
        block = torch.zeros(self.block * self.words, dtype=torch.int64, pin_memory=True)
        self._slots += [block[i * self.words:(i + 1) * self.words] for i in range(self.block)]
        self._owned += [False] * self.block

    def take(self):
        with self._lock:
            n = len(self._slots)
            for i in range(n):
                k = (self._next + i) % n
                if not self._owned[k]:
                    break
            else:
                k = n
                self._grow()
            self._owned[k] = True
            self._next = k + 1
            return k, self._slots[k]

    def release(self, k: int) -> None:
        with self._lock:
            self._owned[k] = False


_results = _HostSlots()


class GuessPolicy:
    """The score of the "no holes" guesses (``CloudBuilder.fuse_tuning``) of the clouds that share this object: a cloud guesses only
    while the misses do not outnumber the guesses that held, so a caller that builds cloud after cloud from sensor depth with holes
    pays one redo, not one per cloud.  Every ``CloudBuilder`` has a policy of its own unless the caller hands one in
    (``CloudBuilder(..., guess_policy=p)``: the batch driver does, for the clouds of its scans); nothing is shared behind the
    caller's back.  Thread-safe."""

    def __init__(self):
        import threading
        self.hits = 0
        self.misses = 0
        self._lock = threading.Lock()

    def allows(self) -> bool:
        return self.misses <= self.hits

    def held(self, n: int = 1) -> None:
        with self._lock:
            self.hits += n

    def missed(self) -> None:
        with self._lock:
            self.misses += 1


class PendingCheck:
    """A ``CloudBuilder.check_async()`` in flight: the row count and the scan status of everything appended UP TO THAT CALL are on
    their way to the host; ``result()`` waits for them (only for them: kernels enqueued later keep running) and does what
    ``check()`` does -- for those batches.  Batches appended after the call are neither verified nor released by it."""

    def __init__(self, builder: "CloudBuilder", key: int, slot: torch.Tensor, nws: int, workspaces: list, event, late=()):
        self._b, self._key, self._slot, self._nws, self._ws, self._event, self._late = builder, key, slot, nws, workspaces, event, list(late)
        self._epoch = builder._epoch                 # reset() since then: the cloud asked about is no longer the builder's
        self._retained_then = builder._released + len(builder._retained)      # the batches this check covers, counted since the last reset() (released ones included) ...
        self._appends_then = builder._appends                # ... (all of them, if nothing was appended since)
        self._guesses_then = builder._guesses_pending
        self._done = False

    def __del__(self):
        try:
            if not self._done:
                _results.release(self._key)
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    def result(self, heal: bool = True) -> int:
        self._event.synchronize()
        b = self._b
        total = int(self._slot[0])
        words = [(w, v >> 32) for w, v in zip(self._ws, self._slot[1:1 + self._nws].tolist())]
        if not self._done:
            self._done = True
            _results.release(self._key)
        words += [(w, int(w[:8].view(torch.int32)[1].item())) for w in self._late]
        bad = [w for w, code in words if code != 0]
        stale = b._epoch != self._epoch
        if bad:
            dense_miss = any(code == 2 for _, code in words)          # (2: a batch run as 'assume dense' was not)
            b._join_side()                                # chained calls may still run on the side streams: nothing of a workspace is wiped under them
            for w in bad:
                w.zero_()                                 # sticky word: cleared only here, once seen -- and with it the whole workspace:
                                                          # after a give-up the single-pass state (epoch, granules) is not to be trusted
            what = ("a batch run without a counting pass ('assume dense', tuning bit 17) was not dense" if dense_miss
                    else "an in-kernel scan timed out in one of the appended batches")
            if stale:
                raise RuntimeError(f"libddcore: {what} (workspace error word set), and the builder was reset() since this check was asked "
                                   "for: the rows it is about are gone -- nothing to redo")
            if not heal:
                raise RuntimeError(f"libddcore: {what} (workspace error word set); rows are invalid -- append the batches again with tuning=4")
            total = b._heal(dense_miss=dense_miss)
            self._retained_then, self._appends_then, self._guesses_then = b._released + len(b._retained), b._appends, 0      # the redo covered everything held
        if total > b.capacity:
            raise OverflowError(f"cloud capacity {b.capacity} < {total} valid points; "
                                "allocate with capacity=batch.max_points or count_valid() first")
        if not stale:
            if not bad and self._guesses_then:
                b.guess_policy.held(self._guesses_then)      # every guess appended before the check held
                b._guesses_pending = max(0, b._guesses_pending - self._guesses_then)
                self._guesses_then = 0
            b._release_retained(total, self._retained_then, self._appends_then)
        return total


def upload_small(arr: np.ndarray, device: torch.device) -> torch.Tensor:
    """Host array -> device tensor without making the host wait for the stream (up to 256 KiB; larger arrays take the
    ordinary path)."""
    return _small.upload(arr, device)


def _require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("depthdensifier_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback for the densify path")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


class ViewBatch:
    """A stack of V equally sized views resident in HBM (``DDViewBatch`` of ``include/ddcore.h``).

    ``depth`` (V,H,W) float32/float16; ``mask`` (V,H,W) bool/uint8; ``conf`` (V,H,W)
    float32/float16; ``normal`` (V,H,W,3) float32; ``rgb`` (V,H,W,3) uint8;
    ``intrinsics`` (V,4) or (V,3,3); ``cam_from_world`` (V,3,4)/(V,4,4).  A single view may be
    passed without the leading axis.
    """

    def __init__(self, depth: ArrayLike, intrinsics: ArrayLike, cam_from_world: ArrayLike, *,
                 mask: Optional[ArrayLike] = None, conf: Optional[ArrayLike] = None,
                 conf_threshold: Optional[float] = None, normal: Optional[ArrayLike] = None,
                 rgb: Optional[ArrayLike] = None, stride: int = 1, semantics: str = "script",
                 rotate_normals: Optional[bool] = None, view_index_base: int = 0, device=None,
                 tuning: int = 0, depth_positive_on_mask: bool = False, refine=None, refined_out=None, lab: int = 0):
        """``tuning``: ``DDViewBatch.tuning`` (``_lib.DD_TUNE_*``: what a caller may choose); ``lab``: the library's experiment
        switches (``_lib.DD_LAB_*``, ``include/ddcore_lab.h``) for the calls made with this batch -- tests and A/B tools only.
        ``refine``: per view ``(knots_x, knots_y, skip_smoothing)`` (or one tuple for a single view), sorted float32 device
        knots of the refiner's transfer curve: ``depth`` is then the RAW monocular map and the densify kernel refines it on
        the fly (``DD_REFINE``: ``depth_refiner.py:180-205`` fused with ``scripts/test.py:194-233``; stride 1, width <= 3071,
        2..512 knots).  ``refined_out``: ``True`` or a (V,H,W) float32 tensor to receive the refined, mask-zeroed map (the
        filter's cache, ``scripts/test.py:197-201``); available as ``self.refined`` afterwards."""
        if semantics not in SEMANTICS:
            raise ValueError(f"semantics must be one of {SEMANTICS}")
        dev = _require_gpu(device)
        d = _gpu(depth, dev)
        if d.dim() == 2:
            d = d[None]
            mask, conf, normal, rgb = (None if x is None else x[None] for x in (mask, conf, normal, rgb))
        if d.dim() != 3:
            raise ValueError(f"depth must be (V,H,W) or (H,W); got {tuple(d.shape)}")
        if d.dtype not in (torch.float32, torch.float16):
            d = d.float()
        self.depth = d.contiguous()
        V, H, W = self.depth.shape
        self.mask = _gpu(mask, dev)
        if self.mask is not None:
            if self.mask.dtype == torch.bool:
                self.mask = self.mask.view(torch.uint8)
            elif self.mask.dtype != torch.uint8:
                self.mask = (self.mask > 0).view(torch.uint8)          # visualizer.py:312 "mask > 0"
        self.conf = _gpu(conf, dev)
        if self.conf is not None and self.conf.dtype not in (torch.float32, torch.float16):
            self.conf = self.conf.float()
        if (self.conf is None) != (conf_threshold is None):
            raise ValueError("conf and conf_threshold must be given together")
        # `conf > thr` as NumPy evaluates it (NEP 50): a Python-float threshold is rounded to the map's
        # dtype before the comparison, so an f16 map is compared against f16(thr)
        if conf_threshold is None:
            self.conf_threshold = 0.0
        elif self.conf.dtype == torch.float16:
            self.conf_threshold = float(np.float16(conf_threshold))
        else:
            self.conf_threshold = float(np.float32(conf_threshold))
        self.normal = _gpu(normal, dev, torch.float32)
        self.rgb = _gpu(rgb, dev)
        if self.rgb is not None:
            if semantics == "viz":
                # visualizer.py:337-342, per view (the reference converts one view per call): the colours OF THE VALID PIXELS
                # are multiplied by 255 and cast to uint8 when their maximum is <= 1 -- whatever the image's dtype -- and are
                # otherwise left as they are.  The kernel gathers uint8 colours; a view whose colours stay non-uint8 is
                # kept in `rgb_passthrough` for COLMAPVisualizer.add_rgbd_pointcloud, which gathers them by pixel index.
                self.rgb, self.rgb_passthrough = _viz_colors(self.rgb, self.depth, self.mask)
            elif self.rgb.dtype != torch.uint8:
                # scripts/test.py:215 only ever sees uint8 (np.array of a PIL image); other dtypes are accepted as a convenience
                self.rgb = (self.rgb * 255).to(torch.uint8) if float(self.rgb.max()) <= 1.0 else self.rgb.to(torch.uint8)
        for name, t, shape in (("mask", self.mask, (V, H, W)), ("conf", self.conf, (V, H, W)),
                               ("normal", self.normal, (V, H, W, 3)), ("rgb", self.rgb, (V, H, W, 3))):
            if t is not None and tuple(t.shape) != shape:
                raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shape}")
        if stride < 1:
            raise ValueError("stride (downsample_density) must be >= 1")
        self.stride = int(stride)
        self.semantics = semantics
        blocks = camera_blocks(intrinsics, cam_from_world)
        if blocks.shape[0] != V:
            raise ValueError(f"{blocks.shape[0]} cameras for {V} views")
        self._knots = None
        self.refined = None
        if refine is not None:
            curves = [refine] if (isinstance(refine, tuple) and len(refine) == 3 and not isinstance(refine[0], tuple)) else list(refine)
            if len(curves) != V:
                raise ValueError(f"{len(curves)} transfer curves for {V} views")
            if semantics != "script" or conf is not None or stride != 1 or W > 3071:
                raise ValueError("refine= needs script semantics, stride 1, no confidence map and width <= 3071")
            i32, u64 = blocks.view(np.int32), blocks.view(np.uint64)       # DDViewParams: n_knots / skip at floats 21, 22; pointers at bytes 96, 104
            keep = []
            for v, (kx, ky, skip) in enumerate(curves):
                kx = _gpu(kx, dev, torch.float32)
                ky = _gpu(ky, dev, torch.float32)
                if kx.dim() != 1 or kx.shape != ky.shape or not 2 <= kx.numel() <= 512:
                    raise ValueError("a transfer curve needs 2..512 knots, knots_x and knots_y of equal length")
                keep.append((kx, ky))
                i32[v, 21], i32[v, 22] = kx.numel(), 1 if skip else 0
                u64[v, 12], u64[v, 13] = kx.data_ptr(), ky.data_ptr()
            self._knots = keep
            if refined_out is True:
                refined_out = torch.empty((V, H, W), dtype=torch.float32, device=dev)
            if refined_out is not None:
                if refined_out.dim() == 2:
                    refined_out = refined_out[None]
                if tuple(refined_out.shape) != (V, H, W) or refined_out.dtype != torch.float32 or not refined_out.is_contiguous() or refined_out.device != dev:
                    raise ValueError("refined_out must be a contiguous float32 (V,H,W) tensor on the batch's device")
                self.refined = refined_out
        elif refined_out is not None:
            raise ValueError("refined_out needs refine=")
        self.params = upload_small(blocks, dev)
        if not hasattr(self, "rgb_passthrough"):
            self.rgb_passthrough = None
        self.view_index_base = int(view_index_base)
        self.tuning = int(tuning)
        if self.tuning & ~_lib.DD_TUNE_ALL:
            raise ValueError(f"tuning {self.tuning:#x}: reserved bits set (the experiment switches are `lab=` since ABI 14: _lib.DD_LAB_*)")
        self.lab = int(lab)

        flags = 0
        if semantics == "script":
            # scripts/test.py:194 + :210 -- mask folded into depth, then depth > 0.
            # depth_positive_on_mask: the caller guarantees depth > 0 wherever the mask is set (true for
            # DepthRefiner output with skip_smoothing, floored at 1e-3 on the mask, depth_refiner.py:176;
            # NOT with the 3x3 median, which can zero an isolated masked pixel); the rule then reduces
            # to the mask and pass 1 reads 1 B/px instead of 5.
            if self.mask is None or not depth_positive_on_mask:
                flags |= _lib.DD_VALID_DEPTH_POSITIVE
            if self.mask is not None:
                flags |= _lib.DD_VALID_MASK
            rot_default = False                      # scripts/test.py:220 camera-frame normals
        else:
            # visualizer.py:311-314 -- mask only when given, else depth > 0
            flags |= _lib.DD_VALID_MASK if self.mask is not None else _lib.DD_VALID_DEPTH_POSITIVE
            rot_default = True                       # visualizer.py:363-374
        if self.conf is not None:
            flags |= _lib.DD_VALID_CONF
        self.rotate_normals = rot_default if rotate_normals is None else bool(rotate_normals)
        if self.rotate_normals:
            flags |= _lib.DD_ROTATE_NORMALS
        if self._knots is not None:
            flags |= _lib.DD_REFINE | _lib.DD_VALID_DEPTH_POSITIVE
        self.flags = flags
        self.device = dev

    def slice(self, lo: int, hi: int) -> "ViewBatch":
        """Views ``[lo, hi)`` of this batch as a batch of their own -- no copy, the maps are shared (used to
        overlap the exchange of one chunk of views with the kernel of the next, ``distributed.fuse_replicated``)."""
        if not 0 <= lo <= hi <= self.num_views:
            raise ValueError(f"slice [{lo}, {hi}) outside a batch of {self.num_views} views")
        sub = object.__new__(ViewBatch)
        sub.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_cstruct", "_ws_bytes")})
        cut = lambda t: None if t is None else t[lo:hi]
        sub.depth, sub.mask, sub.conf, sub.normal, sub.rgb, sub.params, sub.refined = (
            cut(t) for t in (self.depth, self.mask, self.conf, self.normal, self.rgb, self.params, self.refined))
        sub.rgb_passthrough = None if self.rgb_passthrough is None else self.rgb_passthrough[lo:hi]
        sub.view_index_base = self.view_index_base + lo
        return sub

    # -- sizes -------------------------------------------------------------------------
    @property
    def num_views(self) -> int:
        return int(self.depth.shape[0])

    @property
    def visited_per_view(self) -> int:
        _, H, W = self.depth.shape
        s = self.stride
        return ((H + s - 1) // s) * ((W + s - 1) // s)

    @property
    def max_points(self) -> int:
        return self.num_views * self.visited_per_view

    def c_struct(self) -> DDViewBatch:
        """The C view of this batch (cached: the tensors it points to are owned by ``self``)."""
        key = (self.view_index_base, self.tuning, self.flags, self.conf_threshold)
        cached = getattr(self, "_cstruct", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        cs = self._make_c_struct()
        self._cstruct = (key, cs)
        self._ws_bytes = None
        return cs

    def workspace_bytes(self) -> int:
        cs = self.c_struct()
        if getattr(self, "_ws_bytes", None) is None:
            self._ws_bytes = check(lib.dd_workspace_bytes(C.byref(cs)))
        return self._ws_bytes

    def _make_c_struct(self) -> DDViewBatch:
        V, H, W = self.depth.shape
        ptr = lambda t: None if t is None else t.data_ptr()
        return DDViewBatch(
            num_views=V, height=H, width=W, stride=self.stride,
            depth=ptr(self.depth), mask=ptr(self.mask), conf=ptr(self.conf), normal=ptr(self.normal),
            rgb=ptr(self.rgb), params=ptr(self.params),
            depth_dtype=_lib.DD_F16 if self.depth.dtype == torch.float16 else _lib.DD_F32,
            conf_dtype=_lib.DD_F16 if (self.conf is not None and self.conf.dtype == torch.float16) else _lib.DD_F32,
            conf_threshold=self.conf_threshold, flags=self.flags,
            view_index_base=self.view_index_base, tuning=self.tuning, refined_out=ptr(self.refined),
        )


def _viz_colors(rgb: torch.Tensor, depth: torch.Tensor, mask: Optional[torch.Tensor]):
    """``visualizer.py:337-342`` for a stack of views: -> (uint8 image stack for the kernel, list of per-view images that the
    reference would hand on unchanged in a non-uint8 dtype, or None).  The test ``max <= 1`` is made on the colours of the
    valid pixels (``mask > 0`` when a mask is given, else ``depth > 0``, ``visualizer.py:311-314``) and NaN compares false,
    as in NumPy; the product ``colors * 255`` is formed in the image's own dtype, like NumPy's."""
    valid = (mask > 0) if mask is not None else (depth > 0)                      # (V,H,W)
    out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
    keep = [None] * rgb.shape[0]
    for v in range(rgb.shape[0]):
        img = rgb[v]
        cols = img[valid[v]]                                                    # (N,3) in the image's dtype
        scale = False
        if cols.numel() > 0:
            m = cols.max()
            scale = bool((m <= 1.0).item()) if cols.dtype.is_floating_point else int(m.item()) <= 1
        if scale:
            out[v] = (img * 255).to(torch.uint8)                                 # (values of invalid pixels may wrap: never gathered)
        elif img.dtype == torch.uint8:
            out[v] = img
        else:
            out[v] = 0
            keep[v] = img
    return out, (keep if any(k is not None for k in keep) else None)


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def count_valid(batch: ViewBatch) -> torch.Tensor:
    """(V,) int64 device tensor: valid visited pixels per view (the N of ``scripts/test.py:210-212``)."""
    counts = torch.empty(batch.num_views, dtype=torch.int64, device=batch.device)
    if batch.num_views == 0:
        return counts
    cb = batch.c_struct()
    with _lib.lab_switches(batch.lab):
        check(lib.dd_count_valid(C.byref(cb), counts.data_ptr(), _stream(batch.device)))
    return counts


@dataclass
class BatchPlan:
    """Result of pass 1 (``dd_plan``): absolute rows of every view and the per-tile rows kept in
    ``workspace`` for ``dd_scatter``."""

    view_offsets: torch.Tensor      # (V+1,) int64 device
    workspace: torch.Tensor         # uint8 device scratch, consumed by dd_scatter

    @property
    def num_points(self) -> torch.Tensor:
        return self.view_offsets[-1] - self.view_offsets[0]


def plan_batch(batch: ViewBatch, cursor: Optional[torch.Tensor] = None,
               reuse: Optional[BatchPlan] = None) -> BatchPlan:
    """Pass 1: one streaming read of depth/mask/conf -> exact row of every view and tile.  The
    replacement of ``valid_pixels`` bookkeeping (``scripts/test.py:210-212``) that lets the cloud
    be allocated exactly before a single point is written."""
    if cursor is None:
        cursor = torch.zeros(1, dtype=torch.int64, device=batch.device)
    cb = batch.c_struct()
    nbytes = batch.workspace_bytes()
    if reuse is not None and reuse.workspace.numel() >= nbytes and reuse.view_offsets.numel() == batch.num_views + 1:
        ws, offsets = reuse.workspace, reuse.view_offsets        # steady-state: no allocation per step
    else:
        ws = torch.zeros(max(nbytes, 1024), dtype=torch.uint8, device=batch.device)
        offsets = torch.empty(batch.num_views + 1, dtype=torch.int64, device=batch.device)
    with _lib.lab_switches(batch.lab):
        check(lib.dd_plan(C.byref(cb), cursor.data_ptr(), offsets.data_ptr(), ws.data_ptr(), ws.numel(), _stream(batch.device)))
    return BatchPlan(offsets, ws)


# --------------------------------------------------------------------------------------
# the cloud under construction
# --------------------------------------------------------------------------------------

class CloudBuilder:
    """Pre-allocated fused cloud that batches append to on the GPU.

    Replaces the three Python lists + ``np.concatenate`` of ``scripts/test.py:124-127,
    238-240, 264-266``: every appended batch writes its points directly at their final
    slot; a device-side cursor carries the running count, so appends never synchronise
    the host.  ``finish()`` reads the count once and returns exact-size views.
    """

    # A cloud that is ONE large row array (points, no normals) and was placed with its thirds in three classes of HBM is
    # filled two-pass with the scatter taking tiles of the three thirds in turn (round 4): consecutive
    # workgroups then write three classes at once -- 0.72 instead of 0.66 of the roofline on BASELINE configs[4] with the count
    # pass included, 0.815 for the scatter kernel alone (0.81 for the whole step where the count pass is guessed away: fuse_tuning).
    # Below this many rows (3 GiB of points) the count pass costs more than the interleaving wins back.
    CHAIN_MAX_TILES = 700            # (DD_CHAIN_MAX_TILES overrides) small appends up to this many 12288-pixel tiles (4 views of 1080p) are chained
                                     # across the two side streams.  us per call on 185 x 1080p, one stream / chained (profiles/r06_early_gate.txt):
                                     # 1 view 21.2 / 16.1, 2 views 35.1 / 30.8, 4 views 63.1 / 62.5, 8 views 118.6 / 120.0, 16 views 226.1 / 225.8 --
                                     # a call's scan is over only a few us before its last rows are written, so from 4 views on there is no tail left
                                     # to hide a launch behind.  (A gate that opens earlier -- when the previous call's workgroups are all running --
                                     # was built and measured in round 6: the counting costs more than the earlier start wins.)
    INTERLEAVE_MIN_ROWS = 256 << 20
    GUESS_MIN_PIXELS = 4 << 20       # unmasked batches from this size on run count-free (1.41x the single pass at 24 M pixels, 1.22x at 61 M, 1.11x at
                                     # 244 M, 1.23x at 6.1 G: profiles/r04_ab_count_free_small_batches.txt); below, a launch is a few microseconds either way
    INTERLEAVE_REGIONS = 8           # = the number of XCDs: workgroup b runs on XCD b mod 8 (round-robin dispatch) and takes a tile of stretch b mod 8, so
                                     # every XCD writes an eighth of the cloud of its own.  Interleaved A/B in one process (profiles/r04_ab_interleave_count.txt):
                                     # 8 -> 0.822, 16 -> 0.818, 24 / 32 -> 0.811, 3 ... 15 (no multiple of 8) -> 0.797-0.804, 18 -> 0.785

    FIELDS = {"points": ((3,), torch.float32), "normals": ((3,), torch.float32), "colors": ((3,), torch.uint8),
              "pixel_index": ((), torch.int32), "view_index": ((), torch.int32), "packed": ((4,), torch.float32)}

    def __init__(self, capacity: int, *, normals: bool = False, colors: bool = False,
                 pixel_index: bool = True, view_index: bool = False, packed: bool = False, points: bool = True,
                 buffers: Optional[dict] = None, start=None, device=None, placement: Optional[str] = None,
                 guess_policy: Optional[GuessPolicy] = None, exclusive_gpu: Optional[bool] = None):
        """``packed``: also (or, with ``points=False``, only) write the 16-byte ``x, y, z, rgba`` record per point
        (``DDCloudOut.xyz_rgba``).  ``buffers``: caller-owned tensors to write into instead of allocating, keyed like
        ``FIELDS`` -- the multi-GPU fuse hands in the GLOBAL cloud so that every point is written once, at its final
        row.  ``start``: first row (int or (1,) int64 device tensor), e.g. ``rank_offsets[rank]``.  ``placement``:
        ``None`` (default) = ``DD_PLACEMENT`` or ``"probed"`` for clouds of at least 128 Mi rows that carry normals, plain
        allocation below; ``"probed"`` builds the row arrays from physical chunks spread over the three classes of HBM
        address ranges whatever the size, so that points and normals -- written in lock step -- never share a class
        (``placement.place_outputs``; ``self.placement`` reports what was done); ``"first"`` takes the arrays as the
        allocator returns them.  ``guess_policy``: the score of the "no holes" guesses (``fuse_tuning``) this cloud shares with the
        caller's other clouds; default: one of its own.  ``exclusive_gpu``: this process's densify stream has the GPU to itself (one
        process per GPU, the deployment of ``scripts/run_batch.py`` under ``torchrun``; default: the environment's ``DD_EXCLUSIVE_GPU``
        = 1, else False).  The single-pass kernel then takes its tiles by workgroup index instead of drawing tickets from one
        contended counter (``DD_TUNE_BY_INDEX``): 2-8 % faster -- but on a GPU that another launch of the kind shares
        (a second process, a second stream) two launches can hold each other's slots until a spin limit ends it and the batches
        are redone, so it is never assumed.

        The builder keeps the batches it is given (and with them their maps) until the next ``check()`` / ``finish()`` /
        ``reset()`` so that it can redo them if an in-kernel scan gives up (``self.healed``); it stops keeping them once they
        exceed a quarter of the device's free memory.  **The maps of an appended batch must stay unmodified until then**: a
        redo reads them again (an in-place write to one of them is noticed -- tensor version counter -- and turns the redo
        into the error it was before round 3)."""
        dev = _require_gpu(device)
        self.device = dev
        self.capacity = int(capacity)
        n = max(self.capacity, 1)
        want = {"points": points, "normals": normals, "colors": colors, "pixel_index": pixel_index, "view_index": view_index, "packed": packed}
        if not (points or packed):
            raise ValueError("a cloud needs points or the packed record")
        self.placement = None
        if buffers is None and points and (normals or placement == "probed" or (placement is None and n >= self.INTERLEAVE_MIN_ROWS)):
            # the kernel is bound by its row stores: the row arrays are built from chunks spread over the three classes of HBM
            from . import placement as _placement
            p_xyz, p_nrm, p_rgb, self.placement = _placement.place_outputs(n, colors=colors, normals=normals, device=dev, mode=placement)
            placed = {"points": p_xyz, "normals": p_nrm, "colors": p_rgb}
        else:
            placed = {}
        got = {}
        for name, on in want.items():
            tail, dtype = self.FIELDS[name]
            if placed.get(name) is not None:
                got[name] = placed[name]
                continue
            t = None if buffers is None else buffers.get(name)
            if t is not None:
                if t.dtype != dtype or tuple(t.shape) != (t.shape[0],) + tail or t.shape[0] < self.capacity or not t.is_contiguous() or t.device != dev:
                    raise ValueError(f"buffer '{name}' must be a contiguous {dtype} tensor of shape (>= {self.capacity},{','.join(map(str, tail))}) on {dev}")
                got[name] = t
            elif on and buffers is None:
                got[name] = torch.empty((n,) + tail, dtype=dtype, device=dev)
            elif on:
                raise ValueError(f"buffers given without '{name}'")
            else:
                got[name] = None
        self.xyz, self.normal, self.rgb, self.pix, self.view, self.packed = (got[k] for k in ("points", "normals", "colors", "pixel_index", "view_index", "packed"))
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self._start = start
        self._set_start()
        self._offsets: list[torch.Tensor] = []
        self._workspaces: list[torch.Tensor] = []
        self._ws_cache: Optional[torch.Tensor] = None
        # batches appended since the last reset, kept so that they can be redone if an in-kernel scan gives up (see
        # _check_status); a batch holds its maps alive, so retention stops at a quarter of the device's free memory
        self._retained: list = []
        self._retained_bytes = 0
        self._retain_limit = torch.cuda.mem_get_info(dev)[0] // 4
        self._retain_complete = True
        self._retain_base: Optional[int] = None      # row the retained batches start from (None: the cloud's start)
        self._released = 0                           # retained batches released since the last reset(): checks count in absolute numbers
        self._epoch = 0                              # counts reset(): a PendingCheck knows which cloud it was asked about
        self.healed = 0
        self._guesses_pending = 0
        self._appends = 0                            # append() calls since the last reset(): what a PendingCheck covers
        self.guess_policy = guess_policy if guess_policy is not None else GuessPolicy()
        import os
        self.exclusive_gpu = (os.environ.get("DD_EXCLUSIVE_GPU", "0") == "1") if exclusive_gpu is None else bool(exclusive_gpu)
        # consecutive SMALL appends (a streamed view per call, scripts/test.py:131) chained across two side streams so that call n + 1
        # runs beside the tail of call n (DDViewBatch.chain, include/ddcore.h): only where this stream has the GPU to itself
        self.overlap_small = os.environ.get("DD_OVERLAP_SMALL", "1") == "1"
        if os.environ.get("DD_CHAIN_MAX_TILES"):
            self.CHAIN_MAX_TILES = int(os.environ["DD_CHAIN_MAX_TILES"])
        self._side: list = []                        # two side streams + their workspaces, made at the first chained append
        self.side_stream_probes = 0                  # candidates tried until two streams ran side by side (_ensure_side)
        self._side_ws: list = []
        self._chain = None                           # (1,) int64 device: the chain word
        self._chain_seq = 0
        self._side_busy = False                      # chained calls are in flight on the side streams: join before anything else
        self._fork_ev = None
        self.speculate_dense = True                  # fuse_tuning may run unmasked batches of a blocked cloud without the counting pass
        self.dense_misses = 0                        # ... until one of them was not dense (then never again on this cloud)

    def _set_start(self) -> None:
        if self._start is None:
            self.cursor.zero_()
        elif isinstance(self._start, torch.Tensor):
            self.cursor.copy_(self._start.reshape(1), non_blocking=True)
        else:
            self.cursor.fill_(int(self._start))

    def join(self) -> None:
        """Order the caller's stream behind everything appended so far (small appends may run on the builder's side streams:
        ``exclusive_gpu``).  ``check()`` / ``finish()`` / ``reset()`` do it by themselves; a caller that reads the cloud's arrays or
        records a timing event without them calls this first."""
        self._join_side()

    def _join_side(self) -> None:
        """The side streams' chained calls into the caller's stream: everything enqueued from here on runs behind them."""
        if self._side_busy:
            cur = torch.cuda.current_stream(self.device)
            for s in self._side:
                cur.wait_stream(s)
            self._side_busy = False

    def _chained_ok(self, batch: "ViewBatch", tuning: int) -> bool:
        """May this append run chained on a side stream?  A small single-pass call on a GPU this stream has to itself (the later of
        two calls in flight occupies workgroup slots while it waits for the earlier one's scan: never more than 384 of the 512)."""
        if not (self.exclusive_gpu and self.overlap_small) or batch.stride != 1 or batch._knots is not None:
            return False
        if tuning & (1 | 4 | 8 | 0x3F00 | _lib.DD_TUNE_ASSUME_DENSE | (3 << 18)) or batch.lab & _lib.DD_LAB_LOOKBACK:
            return False
        _, H, W = batch.depth.shape
        # small calls only (a large batch fills the chip by itself and its launch latency is nothing): up to CHAIN_MAX_TILES tiles of
        # 12288 pixels.  (Up to 383 tiles of 6144 the call waits inside its scan workgroup; above, behind a one-wave gate kernel.)
        if not (batch.num_views * (-(-(H * W) // 12288)) <= self.CHAIN_MAX_TILES and H * W >= 8):
            return False
        return self._ensure_side()

    def _ensure_side(self) -> bool:
        """The two side streams of the chained appends, made once: two streams whose kernels really run side by side.  The HIP
        runtime deals its streams to a few hardware queues and two streams on one queue run strictly in order -- chained across such
        a pair a one-view call takes 26 us instead of 17 (and 23 on the caller's stream alone: profiles/r05_streaming_queue_collision.txt;
        which streams collide depends on what else the process has created).  So the pair is probed (``dd_streams_overlap``, ~50 us per
        candidate); when no candidate runs beside the first stream this cloud does not chain its appends."""
        if self._side:
            return True
        if not self.overlap_small or torch.cuda.is_current_stream_capturing():      # (the probe synchronises: never inside a capture)
            return False
        scratch = torch.zeros(2, dtype=torch.int32, device=self.device)
        seen = C.c_int32(0)
        with torch.cuda.device(self.device):
            first = torch.cuda.Stream(self.device)
            for _ in range(8):
                second = torch.cuda.Stream(self.device)
                self.side_stream_probes += 1
                check(lib.dd_streams_overlap(first.cuda_stream, second.cuda_stream, scratch.data_ptr(), C.byref(seen)))
                if seen.value:
                    break
            else:
                self.overlap_small = False
                return False
        self._side = [first, second]
        self._side_raw = [s.cuda_stream for s in self._side]
        self._chain = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._chain_ptr = self._chain.data_ptr()
        self._fork_ev = torch.cuda.Event()
        self._fork_ev.record(torch.cuda.current_stream(self.device))      # (creates the underlying event)
        self._fork_raw = self._fork_ev.cuda_event
        return True

    def reset(self) -> None:
        self._join_side()
        self._epoch += 1
        self._appends = 0
        self._guesses_pending = 0                    # (guesses nobody looked at are neither hits nor misses)
        self._set_start()
        self._offsets.clear()
        self._workspaces.clear()
        self._retained.clear()
        self._retained_bytes = 0
        self._retain_complete = True
        self._retain_base = None
        self._released = 0

    def _retain_cost(self, batch: "ViewBatch") -> int:
        """Bytes of maps that holding ``batch`` for a redo keeps alive beyond what is held already."""
        if self._retained and self._retained[-1][0] is batch:          # the same batch again (a timing loop): nothing new is held
            return 0
        n = getattr(batch, "_map_bytes", None)
        if n is None:
            n = batch._map_bytes = sum(t.numel() * t.element_size() for t in (batch.depth, batch.mask, batch.conf, batch.normal, batch.rgb) if t is not None)
        return n

    def _will_retain(self, batch: "ViewBatch") -> bool:
        return self._retain_complete and self._retained_bytes + self._retain_cost(batch) <= self._retain_limit

    def _retain(self, batch: "ViewBatch", offsets: torch.Tensor) -> None:
        if not self._retain_complete:
            return
        nbytes = self._retain_cost(batch)
        if self._retained_bytes + nbytes > self._retain_limit:
            self._retain_complete = False
            self._retained.clear()
            return
        self._retained.append((batch, offsets, self._versions(batch), nbytes))
        self._retained_bytes += nbytes

    @staticmethod
    def _versions(batch: "ViewBatch") -> tuple:
        """(data_ptr, version counter) of every map of a batch: what a redo must find unchanged."""
        maps = getattr(batch, "_maps", None)
        if maps is None:          # (the maps of a batch and where they live never change; what is written INTO them may: the version counters)
            maps = batch._maps = tuple(t for t in (batch.depth, batch.mask, batch.conf, batch.normal, batch.rgb) if t is not None)
            batch._map_ptrs = tuple(t.data_ptr() for t in maps)
        return batch._map_ptrs + tuple(t._version for t in maps)

    def _offsets_slice(self, n: int) -> torch.Tensor:
        """(n,) int64 device slice from a pooled tensor (one allocation per ~4096 offsets, not per append)."""
        pool, used = getattr(self, "_off_pool", (None, 0))
        if pool is None or used + n > pool.numel():
            pool, used = torch.empty(max(4096, n), dtype=torch.int64, device=self.device), 0
        self._off_pool = (pool, used + n)
        return pool[used:used + n]

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws_cache is None or self._ws_cache.numel() < nbytes:
            # zero-filled once: the header's error word is sticky (the library only ever sets it), so a
            # look-back timeout in ANY append since the last finish() is still there when finish() looks
            old = self._ws_cache
            self._ws_cache = torch.zeros(max(nbytes, 1024), dtype=torch.uint8, device=self.device)
            if old is not None:
                self._ws_cache[:16].copy_(old[:16])          # carry a pending error over to the larger buffer
        return self._ws_cache

    def append(self, batch: ViewBatch, _offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Enqueue one batch; returns its (V+1,) absolute view offsets (device, valid once the
        stream has run).  With ``exclusive_gpu`` a small batch (up to 4 views of 1080p) is enqueued on one of the builder's two side
        streams (``_append_chained``): its offsets and rows are ordered with the caller's stream by ``join()`` / ``check()`` /
        ``finish()`` / ``reset()``, not by the append itself -- and so are its READS: a caller that writes into the batch's maps again
        (a staging buffer it refills in place) calls ``join()`` first.  (Maps that are merely dropped are safe: the builder holds them
        until the next check, or tells the allocator which stream still reads them.)"""
        if self.normal is not None and batch.normal is None:
            raise ValueError("this cloud carries normals but the batch has no normal map")
        if self.rgb is not None and batch.rgb is None:
            raise ValueError("this cloud carries colours but the batch has no rgb image")
        redo = _offsets is not None
        if not redo:
            self._appends += 1
        if batch.num_views == 0:                       # an empty chunk of views: nothing to enqueue, the cursor stays
            offsets = _offsets if redo else self._offsets_slice(1)
            self._join_side()                          # (chained calls on the side streams advance the cursor: read it behind them)
            offsets.copy_(self.cursor, non_blocking=True)
            if not redo:
                self._offsets.append(offsets)
                self._retain(batch, offsets)
            return offsets
        saved = batch.tuning
        batch.tuning = self.fuse_tuning(batch)
        if (batch.tuning & ~saved) & _lib.DD_TUNE_ASSUME_DENSE:
            self._guesses_pending += 1                # (the policy's guess, not the caller's bit: scored when the status is read)
        try:
            cb = batch.c_struct()
            out = self._out_struct()
            offsets = _offsets if redo else self._offsets_slice(batch.num_views + 1)
            with _lib.lab_switches(batch.lab):
                if not redo and self._chained_ok(batch, batch.tuning):
                    ws = self._append_chained(batch, cb, out, offsets)
                else:
                    self._join_side()
                    ws = self._workspace(batch.workspace_bytes())
                    check(lib.dd_unproject_compact(C.byref(cb), C.byref(out), offsets.data_ptr(), self.cursor.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), _stream(self.device)))
        finally:
            batch.tuning = saved
        if not redo:
            self._offsets.append(offsets)
            self._workspaces.append(ws)
            self._retain(batch, offsets)
        return offsets

    def _append_chained(self, batch: "ViewBatch", cb, out, offsets: torch.Tensor) -> torch.Tensor:
        """One small call on the next of two side streams, chained to the previous one through the chain word (``DDViewBatch.chain``):
        the call's scan starts from the row the previous call ends at as soon as THAT call's scan is over -- its tiles load and count
        meanwhile, and the previous call's last rows are still being written.  A one-view call is bound by the launch-to-launch latency
        of a stream (4-5 us of the 25 it takes), not by its kernel: two streams take 1.5x as many calls per second
        (``tools/experiments/two_stream_chains.py``)."""
        self._side_workspaces(batch.workspace_bytes())
        if not self._side_busy:
            self._chain.copy_(self.cursor, non_blocking=True)      # sequence 0, the row this chain starts from
            self._chain_seq = 0
        k = self._chain_seq & 1
        side, ws = self._side_raw[k], self._side_ws[k]
        # the maps of the batch (and, the first time, the chain word) are ready on the caller's stream: the side stream waits for them
        check(lib.dd_stream_fork(self._fork_raw, _stream(self.device), side))
        if not self._will_retain(batch):                     # nobody keeps the maps alive for the side stream: tell the allocator
            for t in (batch.depth, batch.mask, batch.conf, batch.normal, batch.rgb, batch.params):
                if t is not None:
                    t.record_stream(self._side[k])
        cb.chain, cb.chain_seq = self._chain_ptr, self._chain_seq
        try:
            check(lib.dd_unproject_compact(C.byref(cb), C.byref(out), offsets.data_ptr(), self.cursor.data_ptr(),
                                           ws.data_ptr(), ws.numel(), side))
        finally:
            cb.chain, cb.chain_seq = None, 0
        self._chain_seq += 1
        self._side_busy = True
        return ws

    def _side_workspaces(self, need: int) -> None:
        """One workspace per side stream, of at least ``need`` bytes."""
        if len(self._side_ws) < 2 or self._side_ws[0].numel() < need:
            self._join_side()
            old = self._side_ws
            self._side_ws = [torch.zeros(max(need, 1024), dtype=torch.uint8, device=self.device) for _ in range(2)]
            for o, n in zip(old, self._side_ws):
                n[:16].copy_(o[:16])                         # (a pending error word travels with the workspace)

    def __del__(self):
        try:
            import sys
            if sys.is_finalizing():                  # (no stream work while the interpreter -- and with it the HIP runtime -- goes down)
                return
            self._join_side()
        except BaseException:      # noqa: BLE001
            pass

    def fuse_tuning(self, batch: "ViewBatch") -> int:
        """``DDViewBatch.tuning`` with which ``append`` runs ``batch``: the batch's own, plus -- for a large stride-1 batch without
        an explicit choice of a path in its tuning --

        * a large batch (half of ``INTERLEAVE_MIN_ROWS`` pixels) into a placed cloud of points only whose thirds lie in three classes (``placement.layout == "blocked"``): two-pass with the
          scatter interleaving ``INTERLEAVE_REGIONS`` stretches of tiles (several store fronts in several classes at once);
        * a cloud without normals and a batch WITHOUT a mask or a confidence map (a depth map its producer considers complete):
          the same scatter against a count-free plan (bit 17: every pixel guessed valid, every tile verified by the scatter; a miss
          is redone by ``check()`` / ``finish()`` like a scan that gave up, once -- then this cloud stops guessing)."""
        t = batch.tuning
        if self.exclusive_gpu and not (t & (1 | 4)):
            t |= _lib.DD_TUNE_BY_INDEX       # (the fused refine stage and an explicit single pass included)
        if (t & (1 | 4 | 8 | 0x3F00 | _lib.DD_TUNE_ASSUME_DENSE)) or batch.stride != 1 or batch._knots is not None:
            return t
        dense_tiles = 128 if (self.normal is None and self.packed is None and not batch.rotate_normals) else 0
        npx = batch.max_points
        if npx < self.GUESS_MIN_PIXELS:               # a streamed view or two: neither the interleaved scatter nor the guess is for it
            return t | dense_tiles
        blocked = (self.placement is not None and self.placement.layout == "blocked"
                   and self.placement.mode.startswith(("probed", "degraded")) and npx >= self.INTERLEAVE_MIN_ROWS // 2)
        guess = (self.normal is None and batch.mask is None and batch.conf is None and self.speculate_dense and not self.dense_misses
                 and self.guess_policy.allows()
                 and self.capacity >= batch.max_points        # (a cloud sized below the pixel count says the maps have holes)
                 and self._will_retain(batch))                # (a guess that misses is redone from the batch: only if it will be held)
        t |= dense_tiles    # a cloud of points (and colours): tiles whose pixels all survive take the list-free path -- 1.5 % on 100 x 12 MP now
                            # that the single pass no longer waits for a look-back (profiles/r05_ab_scan_service_3.txt); nothing with normals
        if blocked or guess:
            # (+ bit 128: dense tiles take the list-free path -- in the scatter pass, which waits for no look-back, its smaller
            # instruction count is worth 0.3-1.3 %; in the single-pass kernel it is not)
            t |= 128 | ((self.INTERLEAVE_REGIONS - 1) << 8)
            t |= _lib.DD_TUNE_ASSUME_DENSE if guess else 4
        return t

    def _out_struct(self) -> DDCloudOut:
        if getattr(self, "_out_cached", None) is None:
            ptr = lambda t: None if t is None else t.data_ptr()
            self._out_cached = DDCloudOut(xyz=ptr(self.xyz), normal=ptr(self.normal), rgb=ptr(self.rgb),
                                          pixel_index=ptr(self.pix), view_index=ptr(self.view), capacity=self.capacity,
                                          xyz_rgba=ptr(self.packed))
        return self._out_cached

    def scatter(self, batch: ViewBatch, plan: "BatchPlan") -> torch.Tensor:
        """Pass 2 only (``dd_scatter``) for a batch planned with :func:`plan_batch` against this
        cloud's cursor; advances the cursor.  Used by ``unproject_views`` (exact allocation) and
        by ``bench.py`` to time the dominant kernel on its own."""
        self._join_side()
        cb = batch.c_struct()
        out = self._out_struct()
        with _lib.lab_switches(batch.lab):
            check(lib.dd_scatter(C.byref(cb), C.byref(out), plan.view_offsets.data_ptr(), plan.workspace.data_ptr(),
                                 plan.workspace.numel(), _stream(self.device)))
        self._retain_complete = False               # (a redo replays append() calls only: not with scatter() calls in between)
        self._retained.clear()
        self.cursor.copy_(plan.view_offsets[-1:], non_blocking=True)
        self._offsets.append(plan.view_offsets)
        self._workspaces.append(plan.workspace)
        return plan.view_offsets

    def check(self) -> int:
        """Synchronise once: the row after the last appended point.  If an in-kernel scan gave up in one of the appended
        batches, the batches are redone once with the dependency-free two-pass kernels (``self.healed`` counts that);
        raises if that is not possible, or if the cloud overflowed."""
        return self.check_async().result()

    def check_async(self) -> PendingCheck:
        """``check()`` without the wait: the cursor and the scan status words are copied to page-locked memory behind the
        kernels enqueued so far; ``.result()`` waits for that copy alone.  A caller that runs scene after scene
        (``scripts/run_batch.py:57-91``) asks here, enqueues the next scene and reads the answer later -- the GPU never
        idles while the host looks at a number."""
        self._join_side()
        ws = list({id(w): w for w in self._workspaces}.values())
        ws, late = ws[:7], ws[7:]                     # (more than 7 distinct workspaces: the rest are read when the result is)
        k, slot = _results.take()
        slot[0:1].copy_(self.cursor, non_blocking=True)
        for i, w in enumerate(ws):
            slot[1 + i:2 + i].copy_(w[:8].view(torch.int64), non_blocking=True)      # bytes 4..7 = the error word
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return PendingCheck(self, k, slot, len(ws), ws, ev, late)

    def _release_retained(self, total: int, covered: Optional[int] = None, appends_then: Optional[int] = None) -> None:
        """Everything a check covered is final: those batches (and their maps) need not be held any longer, a later redo starts
        behind them.  ``covered``: how many batches had been retained since the last reset() when the check was asked for -- an
        ABSOLUTE count (released ones included: several checks may be in flight, and the one read first shifts the list under the
        others); ``appends_then``: the append() calls the check saw.  Batches appended after the check stay held (or stay not held,
        if retention had given up by then) -- the check says nothing about them."""
        if appends_then is None or appends_then == self._appends:
            self._released += len(self._retained)
            self._retained.clear()                    # nothing was appended since: the whole cloud so far is final
            self._retained_bytes = 0
            self._retain_complete = True
            self._retain_base = total
            return
        if not self._retain_complete:
            return                                    # batches behind the check are not held: a redo of them stays impossible
        k = covered - self._released                  # of the batches this check covers, those still held
        if k <= 0:
            return                                    # a later check was read first: it released them all and moved the base beyond
        del self._retained[:k]
        self._released += k
        self._retained_bytes = sum(e[3] for e in self._retained)
        self._retain_base = total                     # the row the first batch behind the check starts from

    def finish(self, name: str = "Dense Cloud") -> FusedCloud:
        """Synchronise once, check the scan status words and the capacity, return exact-size views."""
        total = self.check()
        if self._offsets:
            offs = torch.cat([self._offsets[0]] + [o[1:] for o in self._offsets[1:]])
        else:
            offs = torch.zeros(1, dtype=torch.int64, device=self.device)
        first = 0 if self._start is None else int(offs[0].item())
        cut = lambda t: None if t is None else t[first:total]
        if self.xyz is None:
            return FusedCloud.from_packed(cut(self.packed), offs, name, normals=cut(self.normal), pixel_index=cut(self.pix), view_index=cut(self.view))
        return FusedCloud(points=cut(self.xyz), colors=cut(self.rgb), normals=cut(self.normal),
                          pixel_index=cut(self.pix), view_index=cut(self.view), view_offsets=offs, name=name, packed=cut(self.packed))

    def _scan_gave_up(self) -> bool:
        bad = False
        self._join_side()                                    # (never wipe a workspace under a chained call that still runs)
        for ws in {id(w): w for w in self._workspaces}.values():
            if int(ws[:8].view(torch.int32)[1].item()) != 0:
                ws.zero_()                                   # sticky word: cleared only here, once seen (the whole workspace, as above)
                bad = True
        return bad

    def _heal(self, dense_miss: bool = False) -> int:
        """A look-back of the single-pass kernel timed out (a workgroup was parked for ~2 s: another tenant, ranks sharing
        the GPU): every batch appended since the last reset is run again through dd_plan + dd_scatter (``tuning`` bit 4),
        whose workgroups do not depend on each other, writing the same rows and the same offset tensors."""
        self._join_side()
        what = ("a batch run without a counting pass ('assume dense', tuning bit 17) was not dense" if dense_miss
                else "an in-kernel scan timed out in one of the appended batches")
        if not self._retain_complete or not self._retained:
            raise RuntimeError(f"libddcore: {what} (workspace error word set) "
                               "and the batches are no longer held (more than a quarter of the device's memory): rows are "
                               "invalid -- append the batches again with tuning=4")
        for batch, _, versions, _ in self._retained:
            if self._versions(batch) != versions:
                raise RuntimeError(f"libddcore: {what} (workspace error word set) and "
                                   "one of its maps was modified in place after append(): the rows cannot be redone -- append the "
                                   "batches again with tuning=4 (inputs must stay unmodified until check() / finish())")
        if self._retain_base is None:
            self._set_start()
        else:
            self.cursor.fill_(self._retain_base)
        for batch, offsets, _, _ in self._retained:
            saved, saved_lab = batch.tuning, batch.lab
            batch.tuning = (saved | 4) & ~(8 | _lib.DD_TUNE_ASSUME_DENSE)
            batch.lab = 0                                    # (a redo never runs with an injected fault or another experiment switch)
            try:
                self.append(batch, _offsets=offsets)
            finally:
                batch.tuning, batch.lab = saved, saved_lab
        self.healed += 1
        if not dense_miss and self.exclusive_gpu:
            # a scan that timed out while tiles were taken by workgroup index: the GPU was not this stream's alone after all (two
            # launches holding each other's workgroup slots is the one way that order can stall) -- tickets from here on
            self.exclusive_gpu = False
        if dense_miss:
            self.dense_misses += 1                           # this cloud stops guessing (fuse_tuning)
            if self._guesses_pending:                        # (the policy's guess, not a bit the caller set)
                self.guess_policy.missed()
            self._guesses_pending = 0
        total = int(self.cursor.item())
        if self._scan_gave_up():                             # cannot happen: the two-pass kernels have no look-back
            raise RuntimeError("libddcore: the two-pass redo reported a scan time-out")
        return total


# --------------------------------------------------------------------------------------
# one-call entry points
# --------------------------------------------------------------------------------------

def unproject_views(depth: ArrayLike, intrinsics: ArrayLike, cam_from_world: ArrayLike, *,
                    mask: Optional[ArrayLike] = None, conf: Optional[ArrayLike] = None,
                    conf_threshold: Optional[float] = None, normal: Optional[ArrayLike] = None,
                    rgb: Optional[ArrayLike] = None, downsample_density: int = 1,
                    semantics: str = "script", rotate_normals: Optional[bool] = None,
                    capacity: Union[None, int, str] = None, pixel_index: bool = True,
                    view_index: bool = False, device=None, tuning: int = 0, record: str = "rows",
                    _allow_passthrough: bool = False, lab: int = 0) -> FusedCloud:
    """Densify + fuse a stack of views: ``scripts/test.py:203-244`` per view and ``:262-266``.

    ``downsample_density`` is ``ProcessingConfig.downsample_density`` (``scripts/test.py:37``; the
    reference default is 32, the benchmarks use 1).  ``capacity``: ``None`` counts the valid pixels
    (``dd_count_valid``, one streaming read of depth/mask), allocates exactly and runs the fused
    call ``dd_unproject_compact``; ``"max"`` allocates for every visited pixel and skips the count
    (no host round trip); an int is taken as given.  ``tuning=4`` uses ``dd_plan`` + ``dd_scatter``.
    ``record``: ``"rows"`` = the reference's (N,3) arrays; ``"xyz_rgba"`` = one 16-byte record per point
    (``FusedCloud.packed``; ``points`` / ``colors`` are views of it) -- the compact form of the multi-GPU gather;
    ``"both"`` writes the two.
    """
    if record not in ("rows", "xyz_rgba", "both"):
        raise ValueError("record must be 'rows', 'xyz_rgba' or 'both'")
    batch = ViewBatch(depth, intrinsics, cam_from_world, mask=mask, conf=conf, conf_threshold=conf_threshold,
                      normal=normal, rgb=rgb, stride=downsample_density, semantics=semantics,
                      rotate_normals=rotate_normals, device=device, tuning=tuning, lab=lab)
    if batch.rgb_passthrough is not None and not _allow_passthrough:
        raise ValueError("the colours of some views stay non-uint8 in the reference (visualizer.py:341: valid colours above 1 are "
                         "handed on unchanged); a fused cloud carries uint8 colours -- convert the image, or go through "
                         "COLMAPVisualizer.add_rgbd_pointcloud, which returns them in their own dtype")
    with_normals = batch.normal is not None and (semantics == "script" or batch.mask is not None)
    rows = record != "xyz_rgba"
    fields = dict(normals=with_normals, colors=batch.rgb is not None and rows, pixel_index=pixel_index,
                  view_index=view_index, device=batch.device, packed=record != "rows", points=rows)
    if capacity is None and (tuning & 4):        # forced two-pass: the plan's offsets feed the scatter directly
        plan = plan_batch(batch)
        builder = CloudBuilder(int(plan.num_points.item()), **fields)
        builder.scatter(batch, plan)
        return builder.finish()
    if capacity is None:          # exact allocation: one counting pass, then the fused single-pass call
        cap = int(count_valid(batch).sum().item())
    elif capacity == "max":
        cap = batch.max_points
    else:
        cap = int(capacity)
    builder = CloudBuilder(cap, **fields)
    if cap != batch.max_points:
        builder.speculate_dense = False          # fewer rows than pixels (counted, or said so by the caller): the maps are known to have holes
    builder.append(batch)
    cloud = builder.finish()
    cloud.rgb_passthrough = batch.rgb_passthrough
    return cloud


class CapturedChain:
    """``builder.reset()`` + ``builder.append(b)`` for every batch of ``batches``, captured ONCE in a HIP graph: ``replay()`` enqueues
    the whole chain with one graph launch (on the current stream) instead of one kernel launch per call from Python.  Possible
    since ABI 11: a single-pass call is one kernel whose per-call state (look-back epoch, cursor) lives on the device, so a replay
    finds it as any later call would.  For a caller that feeds the same resident buffers again and again (a fixed ring of staging
    stacks); the maps of the batches must stay where they are, and every replay rebuilds the cloud from the builder's first row.
    Measured (``bench.py`` ``streaming``, ``profiles/r05_streaming_*.txt``): the chain takes the same GPU time either way -- it is
    bound by the dependent launches on the GPU (4-5 us each), not by the host's enqueueing."""

    def __init__(self, builder: "CloudBuilder", batches: Sequence["ViewBatch"]):
        dev = builder.device
        need = max([b.workspace_bytes() for b in batches if b.num_views] + [1024])
        builder._workspace(need)                                                                     # (allocated outside the capture)
        if builder.exclusive_gpu and builder._ensure_side():      # ... and so are the side streams of chained appends (probed: synchronises)
            builder._side_workspaces(need)
        self.builder, self.batches = builder, list(batches)
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(side):
            self.graph.capture_begin()
            builder.reset()
            for b in self.batches:
                builder.append(b)
            builder._join_side()                      # (chained small calls fork onto the builder's side streams: joined inside the capture)
            self.graph.capture_end()
        torch.cuda.synchronize(dev)

    def replay(self) -> None:
        self.graph.replay()


def capture_chain(builder: "CloudBuilder", batches: Sequence["ViewBatch"]) -> CapturedChain:
    return CapturedChain(builder, batches)


def fuse_batches(batches: Sequence[ViewBatch], capacity: Optional[int] = None, **cloud_fields) -> FusedCloud:
    """Fuse several batches (e.g. views of different resolutions) into one cloud, in order."""
    if capacity is None:
        capacity = sum(int(count_valid(b).sum().item()) for b in batches)
    base = 0
    builder = CloudBuilder(capacity, device=batches[0].device, **cloud_fields)
    if capacity != sum(b.max_points for b in batches):
        builder.speculate_dense = False          # (as in unproject_views)
    for b in batches:
        b.view_index_base = base
        builder.append(b)
        base += b.num_views
    return builder.finish()
