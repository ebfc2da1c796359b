"""Monocular-depth -> metric-depth alignment against the COLMAP sparse points.

Drop-in for the reference's exported API (``src/depthdensifier/__init__.py:3-6``):
``RefinerConfig`` (``depth_refiner.py:16-31``) and ``DepthRefiner`` with the same constructor
(``:47-82``), ``refine_depth`` signature and result dictionary (``:207-216, 323-328``), the same
early exits that hand back the caller's own ``depth_map`` object (``:259, 278, 285, 299``), and the
same arithmetic: project the sparse points (``:92-115``), sample the depth map bilinearly at the
projections (``:266-272``), drop ratio outliers by 2.5 IQR (``:117-139``), build a sorted
look-up table depth -> metric depth and interpolate it linearly for every masked pixel
(``:141-178``), 3x3 median (``:194-200``), re-zero outside the mask (``:203``).

What is different (MI355X-first): everything runs as device tensor ops on the ROCm GPU and the
refined map can STAY there (``return_tensor=True``) so that the densify kernels read it straight
from HBM -- the reference's D2H at ``:312`` and the H2D of the next stage disappear; the median is a
19-exchange sorting network over 9 shifted views instead of materialising a 9x ``unfold`` copy.
Inputs may be NumPy arrays or tensors.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional, Union

import time

import numpy as np
import torch
import torch.nn.functional as F

ArrayLike = Union[np.ndarray, torch.Tensor]


@dataclass
class RefinerConfig:
    """Same fields and defaults as the reference's ``RefinerConfig`` (``depth_refiner.py:16-31``)."""

    min_correspondences: int = 50
    edge_margin: int = 10
    robust: bool = True
    outlier_threshold: float = 2.5
    use_fp16: bool = True
    skip_smoothing: bool = False
    adaptive_correspondences: bool = True
    verbose: int = 0


_FIELDS = ("min_correspondences", "edge_margin", "robust", "outlier_threshold", "use_fp16", "skip_smoothing",
           "adaptive_correspondences", "verbose")

# median of 9 by compare-exchange (Paeth / Smith network, 19 exchanges); exact for finite inputs
_MED9 = ((1, 2), (4, 5), (7, 8), (0, 1), (3, 4), (6, 7), (1, 2), (4, 5), (7, 8), (0, 3), (5, 8), (4, 7),
         (3, 6), (1, 4), (2, 5), (4, 7), (4, 2), (6, 4), (4, 2))


def median3x3(img: torch.Tensor) -> torch.Tensor:
    """3x3 median with replicated borders (what ``depth_refiner.py:194-200`` computes)."""
    p = F.pad(img[None, None], (1, 1, 1, 1), mode="replicate")[0, 0]
    h, w = img.shape
    v = [p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)]
    has_nan = torch.isnan(torch.stack(v)).any(dim=0)        # torch.median of a window holding a NaN is NaN
    v = [torch.nan_to_num(x, nan=0.0, posinf=float("inf"), neginf=float("-inf")) for x in v]
    for a, b in _MED9:
        lo, hi = torch.minimum(v[a], v[b]), torch.maximum(v[a], v[b])
        v[a], v[b] = lo, hi
    return torch.where(has_nan, torch.full_like(v[4], float("nan")), v[4]).contiguous()


class DepthRefiner:
    """GPU depth refiner; see the module docstring for the mapping to the reference."""

    def __init__(self, config: Optional[RefinerConfig] = None, **overrides):
        unknown = set(overrides) - set(_FIELDS)
        if unknown:
            raise TypeError(f"DepthRefiner got unexpected argument(s): {sorted(unknown)}")
        config = config or RefinerConfig()
        for name in _FIELDS:                              # per-field overrides win (depth_refiner.py:75-82)
            val = overrides.get(name)
            setattr(self, name, getattr(config, name) if val is None else val)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.dtype = torch.float16 if self.use_fp16 and torch.cuda.is_available() else torch.float32
        if self.verbose > 0:
            print(f"[DepthRefiner] Using {self.device.type.upper()} backend with "
                  f"{'FP16' if self.dtype == torch.float16 else 'FP32'}")

    trace: Optional[dict] = None           # a caller's dict: host seconds per step of begin_refine (the pipeline's loop report)

    def _lap(self, key: str, t0: float) -> float:
        now = time.perf_counter()
        if self.trace is not None:
            self.trace[key] = self.trace.get(key, 0.0) + now - t0
        return now

    # ---- pieces -----------------------------------------------------------------------
    def _to(self, x: ArrayLike, dtype=None) -> torch.Tensor:
        t = torch.from_numpy(x) if isinstance(x, np.ndarray) else x
        return t.to(self.device, dtype=dtype or self.dtype)

    def _project_sparse(self, pts: torch.Tensor, cam_from_world: torch.Tensor, K: torch.Tensor):
        """Pixel coordinates and camera depths of the sparse points (``depth_refiner.py:92-115``):
        points behind the camera keep coordinate (0,0) and are rejected later by ``depth > 0``."""
        ones = torch.ones(pts.shape[0], 1, device=self.device, dtype=self.dtype)
        bottom = torch.tensor([[0, 0, 0, 1]], device=self.device, dtype=self.dtype)
        H = torch.cat([cam_from_world[:3], bottom], dim=0)
        cam = (H @ torch.cat([pts, ones], dim=1).T).T[:, :3]
        z = cam[:, 2]
        front = z > 0
        uv = torch.zeros((len(z), 2), device=self.device, dtype=self.dtype)
        if front.any():
            ray = cam[front] / z[front, None]
            uv[front] = (K[:2, :2] @ ray[:, :2].T).T + K[:2, 2]
        return uv, z

    def _iqr_inliers(self, z_metric: torch.Tensor, z_mono: torch.Tensor):
        """Ratio test of ``depth_refiner.py:117-139``: keep ``|r - median(r)| < outlier_threshold * IQR(r)``."""
        if len(z_metric) < 10:
            return z_metric, z_mono, 0
        r = z_metric / (z_mono + 1e-6)
        med = torch.median(r)
        q = torch.quantile(r.float(), torch.tensor([0.75, 0.25], device=self.device, dtype=torch.float32))
        thr = self.outlier_threshold * (q[0] - q[1])
        if r.dtype == torch.float16:
            thr, med = thr.half(), med.half()
        keep = torch.abs(r - med) < thr
        return z_metric[keep], z_mono[keep], int((~keep).sum().item())

    def _lut_interpolate(self, d: torch.Tensor, x: torch.Tensor, y: torch.Tensor, reciprocal: bool = False) -> torch.Tensor:
        """Piecewise-linear transfer curve through the sorted knots ``(x, y)`` (``depth_refiner.py:141-178``);
        clamped at the end knots, floored at 1e-3.  ``reciprocal``: ``t = (d - x0) * (1 / dx)`` instead of the reference's
        ``(d - x0) / dx`` -- what the HIP kernels compute since round 6 (the reciprocal is a property of the interval,
        ``csrc/ddrefine_math.h``; at most two ulps of ``t`` apart): what the tests compare the kernels with, bit for bit."""
        if len(d) < 4:
            return d * torch.median(y / (d + 1e-6))
        order = torch.argsort(x)
        xs, ys = x[order], y[order]
        if len(xs) < 2:
            return d * torch.median(y / (x + 1e-6))
        hi = torch.clamp(torch.searchsorted(xs, d, right=False), 1, len(xs) - 1)
        x0, x1, y0, y1 = xs[hi - 1], xs[hi], ys[hi - 1], ys[hi]
        dx = x1 - x0
        dx = torch.where(dx == 0, torch.tensor(1e-6, device=self.device, dtype=self.dtype), dx)
        t = torch.clamp((d - x0) * (1.0 / dx) if reciprocal else (d - x0) / dx, 0, 1)
        out = y0 + t * (y1 - y0)
        return torch.maximum(out, torch.tensor(1e-3, device=self.device, dtype=self.dtype))

    def _sorted_knots(self, x: torch.Tensor, y: torch.Tensor):
        """The knots ordered by x (``torch.argsort``, ``depth_refiner.py:149-151``), float32, on the device."""
        from ._lib import DDCoreError, lib
        xf, yf = x.float().contiguous(), y.float().contiguous()
        if xf.numel() <= 4096:                                # one launch instead of argsort + two gathers
            kx, ky = torch.empty_like(xf), torch.empty_like(yf)
            rc = lib.dd_sort_knots(xf.data_ptr(), yf.data_ptr(), xf.numel(), kx.data_ptr(), ky.data_ptr(),
                                   torch.cuda.current_stream(xf.device).cuda_stream)
            if rc < 0:
                raise DDCoreError(rc, lib.dd_refine_last_error().decode())
            return kx, ky
        order = torch.argsort(xf)
        return xf[order].contiguous(), yf[order].contiguous()

    def _apply_curve_hip(self, depth: torch.Tensor, mask: Optional[torch.Tensor], x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """The same per-pixel work as ``_apply_curve`` in one hand-written kernel (``dd_refine_apply``,
        ``csrc/ddrefine.hip``): the view is read once and written once.  Always float32 arithmetic."""
        from ._lib import DD_F16, DD_F32, DDCoreError, lib
        kx, ky = self._sorted_knots(x, y)
        d = depth if depth.dtype in (torch.float16, torch.float32) else depth.float()
        d = d.contiguous()
        m = None if mask is None else mask.contiguous().view(torch.uint8)
        out = torch.empty(d.shape, dtype=torch.float32, device=d.device)
        rc = lib.dd_refine_apply(d.data_ptr(), DD_F16 if d.dtype == torch.float16 else DD_F32,
                                 None if m is None else m.data_ptr(), d.shape[0], d.shape[1], kx.data_ptr(), ky.data_ptr(),
                                 kx.numel(), 1 if self.skip_smoothing else 0, out.data_ptr(),
                                 torch.cuda.current_stream(d.device).cuda_stream)
        if rc < 0:
            raise DDCoreError(rc, lib.dd_refine_last_error().decode())
        return out

    def _apply_curve(self, depth: torch.Tensor, mask: torch.Tensor, x: torch.Tensor, y: torch.Tensor, reciprocal: bool = False,
                     n_masked: Optional[int] = None) -> torch.Tensor:
        """``depth_refiner.py:180-205``: curve on masked pixels, 3x3 median, zeros outside the mask.  ``n_masked``: the number of set
        mask pixels if the caller knows it (the fit counts them: no host synchronisation here then)."""
        if depth.is_cuda and len(x) >= 2 and (n_masked if n_masked is not None else int(mask.sum().item())) >= 4:       # (:143-145, 153-154 keep the tensor path)
            return self._apply_curve_hip(depth, mask, x, y)
        out = torch.zeros_like(depth)
        if mask.any():
            out[mask] = self._lut_interpolate(depth[mask], x, y, reciprocal)
        if not self.skip_smoothing:
            out = median3x3(out)
        out[~mask] = 0
        return out

    def _fit_launch(self, depth: torch.Tensor, points3D: ArrayLike, cam_from_world: ArrayLike, K: ArrayLike, mask_or_count: Optional[torch.Tensor] = None):
        """The correspondence half (``depth_refiner.py:244-299``) as ONE kernel launch (``dd_refine_fit``,
        ``csrc/ddrefine.hip``) instead of ~25 tensor launches and several synchronisations; the eight result words are
        copied to page-locked host memory asynchronously and an event marks them ready, so the caller may enqueue other
        work (the next view's uploads and fit) before ``_fit_finish`` reads them.  In the reference's FP16 mode (``:85-86``)
        the inputs and the correspondences are quantised to half like there; the arithmetic in between is float32 (no
        golden exists for that mode, and this is at least as close to the FP32 result).  ``mask_or_count``: the view's mask (bool / uint8,
        (H,W): its number of set pixels rides along in the same read; None counts ``depth > 0``) or a 0-dim device count."""
        import ctypes as C
        from ._lib import DD_F16, DD_F32, DDCoreError, lib
        half = self.dtype == torch.float16
        q = (lambda a: np.asarray(a, dtype=np.float64).astype(np.float16).astype(np.float32)) if half else (lambda a: np.asarray(a, dtype=np.float32))
        E = q(cam_from_world.cpu().numpy() if isinstance(cam_from_world, torch.Tensor) else cam_from_world)[:3, :4].reshape(-1)
        Kq = q(K.cpu().numpy() if isinstance(K, torch.Tensor) else K)[:2, :3].reshape(-1)
        d = depth.contiguous()
        stream = torch.cuda.current_stream(self.device)
        # page-locked result slots with their events: a handle OWNS its slot from here until _fit_finish hands it back, so any number
        # of begun handles may be open at once (the free list grows on demand; the pipeline's one-group lag uses two groups' worth)
        if not hasattr(self, "_meta_free"):
            self._meta_free = []
        if self._meta_free:
            host, ready = self._meta_free.pop()
        else:
            host, ready = torch.empty(8, dtype=torch.int32, pin_memory=True), torch.cuda.Event()
            ready.record(stream)                      # (creates the underlying event: its handle goes to the native call)
        staged = None
        t_ = time.perf_counter()
        if isinstance(points3D, np.ndarray):
            from .densify import _small
            staged = _small.stage(np.ascontiguousarray(points3D, dtype=np.float32).reshape(-1, 3))
        t_ = self._lap("fit_launch.stage_points", t_)
        if staged is not None and isinstance(mask_or_count, (torch.Tensor, type(None))) and (mask_or_count is None or mask_or_count.dtype in (torch.bool, torch.uint8)):
            # ONE native call: points up, fit, masked-pixel count, result words down, event (dd_refine_fit_async) -- the calls the host
            # makes per view are what bounds the loop around the kernels (profiles/r06_bench_pipeline.txt)
            n = int(np.asarray(points3D).reshape(-1, 3).shape[0])
            work = torch.empty(6 * max(n, 1) + 8, dtype=torch.float32, device=self.device)
            t_ = self._lap("fit_launch.empty_work", t_)
            m = mask_or_count
            rc = lib.dd_refine_fit_async(staged[0], n, (C.c_float * 12)(*E.tolist()), (C.c_float * 6)(*Kq.tolist()), d.data_ptr(),
                                         DD_F16 if d.dtype == torch.float16 else DD_F32, d.shape[0], d.shape[1], int(self.edge_margin),
                                         1 if self.robust else 0, float(self.outlier_threshold), 1 if half else 0,
                                         None if m is None else m.data_ptr(), work.data_ptr(), work.data_ptr() + 24 * max(n, 1), host.data_ptr(),
                                         ready.cuda_event, stream.cuda_stream)
            if rc < 0:
                raise DDCoreError(rc, lib.dd_refine_last_error().decode())
            self._lap("fit_launch.native_call", t_)
            _small.staged_until(staged[1], stream)
            nn = max(n, 1)
            return {"work": work, "n": nn, "host": host, "ready": ready, "keep": (d, m)}
        if isinstance(points3D, np.ndarray):
            from .densify import upload_small
            pts = upload_small(np.ascontiguousarray(points3D, dtype=np.float32).reshape(-1, 3), self.device)   # no host wait
        else:
            pts = points3D.float().to(self.device).contiguous()
        n = int(pts.shape[0])
        buf = torch.empty((3, max(n, 1)), dtype=torch.float32, device=self.device)
        meta = torch.zeros(8, dtype=torch.int32, device=self.device)
        rc = lib.dd_refine_fit(pts.data_ptr(), n, (C.c_float * 12)(*E.tolist()), (C.c_float * 6)(*Kq.tolist()), d.data_ptr(),
                               DD_F16 if d.dtype == torch.float16 else DD_F32, d.shape[0], d.shape[1], int(self.edge_margin),
                               1 if self.robust else 0, float(self.outlier_threshold), 1 if half else 0,
                               buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), meta.data_ptr(), stream.cuda_stream)
        if rc < 0:
            raise DDCoreError(rc, lib.dd_refine_last_error().decode())
        src = mask_or_count if mask_or_count is not None else d > 0          # depth_refiner.py:241
        also = src if src.dim() == 0 else src.sum().clamp(max=2 ** 31 - 1)
        meta[5] = also.to(torch.int32)
        host.copy_(meta, non_blocking=True)
        ready.record(stream)
        return {"buf": buf, "meta": meta, "host": host, "ready": ready, "keep": (pts, d)}

    def _fit_finish(self, fit: dict):
        """``(z_mono, z_metric, in_bounds, positive, kept, removed, scale, extra)``: the one synchronisation of the fit."""
        if fit.get("host") is None:
            raise RuntimeError("finish_refine: this handle has been finished already")
        t0 = time.perf_counter()
        fit["ready"].synchronize()
        self.wait_seconds = getattr(self, "wait_seconds", 0.0) + time.perf_counter() - t0      # (host time spent waiting for the GPU: the pipeline's report)
        inb, pos, kept, removed, scale_bits, extra = fit["host"][:6].tolist()
        self._meta_free.append((fit["host"], fit["ready"]))      # the slot (and its event) is free for the next begun handle
        fit["host"] = None
        scale = float(np.array([scale_bits], dtype=np.int32).view(np.float32)[0])
        if "work" in fit:                                        # (dd_refine_fit_async: points, z_mono, z_metric, scratch in one array)
            w, n = fit["work"], fit["n"]
            return w[3 * n:3 * n + kept], w[4 * n:4 * n + kept], inb, pos, kept, removed, scale, extra
        buf = fit["buf"]
        return buf[0][:kept], buf[1][:kept], inb, pos, kept, removed, scale, extra

    def _fit_hip(self, depth: torch.Tensor, points3D: ArrayLike, cam_from_world: ArrayLike, K: ArrayLike):
        """Launch + finish in one go: ``(z_mono, z_metric, in_bounds, positive, kept, removed, scale)``."""
        return self._fit_finish(self._fit_launch(depth, points3D, cam_from_world, K))[:7]

    # ---- API --------------------------------------------------------------------------
    def refine_depth(self, depth_map: ArrayLike, normal_map: Optional[ArrayLike], points3D: ArrayLike,
                     cam_from_world: ArrayLike, K: ArrayLike, mask: Optional[ArrayLike] = None,
                     return_tensor: bool = False, generator: Optional[torch.Generator] = None, fit_only: bool = False,
                     **kwargs) -> dict[str, Any]:
        """Same contract as the reference (``depth_refiner.py:207-328``).  Extras: ``return_tensor=True``
        leaves ``refined_depth`` on the device (float32 tensor) for the densify kernels;
        ``generator`` seeds the 500-correspondence subsample (the reference's is unseeded, ``:304``);
        ``fit_only=True`` (GPU) stops after the correspondence fit and returns ``curve = (knots_x, knots_y,
        skip_smoothing)`` with ``refined_depth = None`` when the curve can be applied inside the densify kernel
        (``ViewBatch(refine=...)``; ``raw_depth`` is the map in the refiner's working precision, which is what the curve
        must be applied to) -- early exits and unusual curves still return a map.

        ``begin_refine`` + ``finish_refine`` are the two halves of this call: on a GPU the first only ENQUEUES the
        correspondence fit, so a caller that streams views can start the next view before it asks for this one's result."""
        return self.finish_refine(self.begin_refine(depth_map, normal_map, points3D, cam_from_world, K, mask=mask,
                                                    return_tensor=return_tensor, generator=generator, fit_only=fit_only, **kwargs))

    def begin_refine(self, depth_map: ArrayLike, normal_map: Optional[ArrayLike], points3D: ArrayLike,
                     cam_from_world: ArrayLike, K: ArrayLike, mask: Optional[ArrayLike] = None,
                     return_tensor: bool = False, generator: Optional[torch.Generator] = None, fit_only: bool = False,
                     working_out: Optional[torch.Tensor] = None, **kwargs) -> dict:
        """First half of ``refine_depth``: inputs to the device and, on a GPU, the correspondence fit enqueued (no host
        synchronisation).  Returns the handle ``finish_refine`` takes.  ``working_out``: a device tensor of the depth map's shape in the
        refiner's working precision (``self.dtype``) to hold the map in that precision (the pipeline's resident stacks: the
        conversion then writes where the densify kernel will read, nothing is allocated or stacked later)."""
        if self.verbose > 1:
            print(f"[DepthRefiner] Input depth shape: {tuple(depth_map.shape)}")
            print(f"[DepthRefiner] COLMAP points: {len(points3D)}")
        t_ = time.perf_counter()
        if working_out is not None and isinstance(depth_map, torch.Tensor) and depth_map.is_cuda and working_out.dtype == self.dtype \
                and tuple(working_out.shape) == tuple(depth_map.shape):
            depth = working_out if working_out.data_ptr() == depth_map.data_ptr() else working_out.copy_(depth_map)
        else:
            depth = self._to(depth_map)
        t_ = self._lap("begin_refine.depth_to_working_precision", t_)
        gpu_fit = depth.is_cuda and depth.dim() == 2 and min(depth.shape) >= 2
        # (on the GPU path a missing mask stays None -- the kernels test depth > 0 themselves -- and is only made when a tensor branch asks)
        m = self._to(mask, torch.bool) if mask is not None else (None if gpu_fit else depth > 0)
        h = dict(depth_map=depth_map, depth=depth, m=m, points3D=points3D, cam_from_world=cam_from_world, K=K,
                 return_tensor=return_tensor, generator=generator, fit_only=fit_only)
        if gpu_fit:
            # GPU: the whole correspondence half is one hand-written kernel (the number of masked pixels, needed later to
            # choose the apply path, is counted beside it and read in the fit's own synchronisation)
            t_ = self._lap("begin_refine.mask", t_)
            h["fit"] = self._fit_launch(depth, points3D, cam_from_world, K, mask_or_count=m.contiguous() if m is not None else None)
            self._lap("begin_refine.fit_launch", t_)
        return h

    def finish_refine(self, h: dict) -> dict[str, Any]:
        """Second half of ``refine_depth``: waits for the fit (GPU) and produces the result dictionary."""
        depth_map, depth, m = h["depth_map"], h["depth"], h["m"]
        if m is None and not (h["fit_only"] and "fit" in h):
            m = depth > 0                                        # depth_refiner.py:241 (only the apply paths need the tensor)
        points3D, cam_from_world, K = h["points3D"], h["cam_from_world"], h["K"]
        return_tensor, generator, fit_only = h["return_tensor"], h["generator"], h["fit_only"]

        def unchanged(n, why):
            if self.verbose > 0:
                print(f"[DepthRefiner] {why}")
            return {"refined_depth": depth_map, "num_correspondences": n, "scale_factor": 1.0}

        if "fit" in h:
            z_mono, z_metric, inb, pos, kept, removed, scale, n_masked = self._fit_finish(h["fit"])
            if inb == 0:
                return unchanged(0, "No valid correspondences found")
            if pos == 0:
                return unchanged(0, "No valid depth correspondences")
            if kept < self.min_correspondences:
                return unchanged(kept, f"Too few correspondences ({kept} < {self.min_correspondences})")
            if self.adaptive_correspondences and kept > 500:
                pick = torch.randperm(kept, device=self.device, generator=generator)[:500]
                z_mono, z_metric = z_mono[pick], z_metric[pick]
                scale = torch.median(z_metric / (z_mono + 1e-6))               # stays on the device unless somebody asks (below)
            n_corr = int(z_mono.numel())
            if fit_only:
                # hand the curve out instead of applying it: the densify kernel refines on the fly (ViewBatch(refine=...)).
                # Only where dd_refine_apply would have been taken (>= 4 masked pixels, see _apply_curve) and the curve fits
                # the kernel's LDS table; otherwise the caller gets the refined map as usual.
                if 2 <= n_corr <= 512 and n_masked >= 4:
                    kx, ky = self._sorted_knots(z_mono, z_metric)
                    if self.verbose > 0:                       # the reference's messages (:317-321); float(scale) synchronises, so only when asked
                        print(f"[DepthRefiner] Refined using {n_corr} correspondences")
                        if removed > 0:
                            print(f"[DepthRefiner] Removed {removed} outliers")
                        print(f"[DepthRefiner] Effective scale: {float(scale):.3f}")
                    # scale_factor: a float, or (adaptive subsample) a 0-dim device tensor -- no second synchronisation here;
                    # float(result["scale_factor"]) gives the number in either case
                    return {"refined_depth": None, "curve": (kx, ky, bool(self.skip_smoothing)), "raw_depth": depth,
                            "num_correspondences": n_corr, "outliers_removed": removed, "scale_factor": scale}
            if not (return_tensor and self.verbose <= 0):
                scale = float(scale)                             # (a device scalar after the adaptive subsample: reading it synchronises -- a
                                                                 # caller that keeps the map on the device gets the scalar as it is, see fit_only)
            if m is None:
                m = depth > 0                                    # depth_refiner.py:241
            refined = self._apply_curve(depth, m, z_mono, z_metric, n_masked=n_masked)
            if self.verbose > 0:
                print(f"[DepthRefiner] Refined using {n_corr} correspondences")
                if removed > 0:
                    print(f"[DepthRefiner] Removed {removed} outliers")
                print(f"[DepthRefiner] Effective scale: {scale:.3f}")
            out = refined.float() if return_tensor else refined.cpu().numpy().astype(np.float32)
            return {"refined_depth": out, "num_correspondences": n_corr, "outliers_removed": removed, "scale_factor": scale}

        pts = self._to(points3D)
        E = self._to(cam_from_world)
        Kt = self._to(K)

        uv, z = self._project_sparse(pts, E, Kt)
        h, w = depth.shape
        e = self.edge_margin
        ok = (uv[:, 0] >= e) & (uv[:, 0] < w - e) & (uv[:, 1] >= e) & (uv[:, 1] < h - e) & (z > 0)
        if not ok.any():
            return unchanged(0, "No valid correspondences found")
        uv, z = uv[ok], z[ok]
        # bilinear sample at the projections (align_corners=True <=> pixel centres at integers)
        grid = torch.stack([uv[:, 0] / (w - 1) * 2 - 1, uv[:, 1] / (h - 1) * 2 - 1], dim=-1)[None, None]
        sampled = F.grid_sample(depth[None, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True).squeeze()
        if sampled.numel() == 0:
            return unchanged(0, "No depth values sampled")
        has = sampled > 0
        if not has.any():
            return unchanged(0, "No valid depth correspondences")
        z_mono, z_metric = sampled[has], z[has]
        removed = 0
        if self.robust and len(z_mono) > 10:
            z_metric, z_mono, removed = self._iqr_inliers(z_metric, z_mono)
        if len(z_mono) < self.min_correspondences:
            return unchanged(len(z_mono), f"Too few correspondences ({len(z_mono)} < {self.min_correspondences})")
        if self.adaptive_correspondences and len(z_mono) > 500:
            pick = torch.randperm(len(z_mono), device=self.device, generator=generator)[:500]
            z_mono, z_metric = z_mono[pick], z_metric[pick]

        refined = self._apply_curve(depth, m, z_mono, z_metric)
        scale = float(torch.median(z_metric / (z_mono + 1e-6)).cpu())
        if self.verbose > 0:
            print(f"[DepthRefiner] Refined using {len(z_mono)} correspondences")
            if removed > 0:
                print(f"[DepthRefiner] Removed {removed} outliers")
            print(f"[DepthRefiner] Effective scale: {scale:.3f}")
        out = refined.float() if return_tensor else refined.cpu().numpy().astype(np.float32)
        return {"refined_depth": out, "num_correspondences": len(z_mono), "outliers_removed": removed, "scale_factor": scale}
