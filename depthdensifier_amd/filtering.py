"""Multi-view floater filter on the GPU.

Mirrors ``scripts/test.py:269-335`` of the reference: every fused point is projected into every
cached view (``project_points``, ``:58-76``) and collects a vote when it lies clearly in front of
that view's refined depth; points with ``votes >= vote_threshold`` are dropped.  The O(N*V) vote
loop runs in ``libddcore.so`` (``dd_floater_votes``, float64 decisions like NumPy's); PyTorch owns
the memory.  No CPU fallback.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from ._lib import DDCloudOut, DDFilterViews, DDCoreError, lib
from .densify import ArrayLike, FusedCloud, _gpu, _require_gpu, _stream, intrinsics_matrix


@dataclass
class FilteringConfig:
    """``FilteringConfig`` of ``scripts/test.py:40-46``."""

    vote_threshold: int = 5
    """Number of votes required to remove a 'floater' point."""
    depth_threshold: float = 0.7
    """Threshold to identify a floater (projected_depth < T * refined_depth)."""


GRAZING_COS = 0.087   # scripts/test.py:295


def filter_cameras(intrinsics: ArrayLike, cam_from_world: ArrayLike) -> np.ndarray:
    """(V,24) float64 camera blocks of ``DDFilterViews.cams``: ``[R|t]`` row-major, the first two rows
    of the calibration matrix, the projection centre ``-R^T t``."""
    K = intrinsics_matrix(intrinsics)
    E = np.asarray(cam_from_world.cpu() if isinstance(cam_from_world, torch.Tensor) else cam_from_world, dtype=np.float64)
    if E.ndim == 2:
        E = E[None]
    E = E[:, :3, :]
    if K.shape[0] == 1 and E.shape[0] > 1:
        K = np.repeat(K, E.shape[0], axis=0)
    if K.shape[0] != E.shape[0]:
        raise ValueError(f"{K.shape[0]} intrinsics for {E.shape[0]} poses")
    cams = np.zeros((E.shape[0], 24))
    cams[:, 0:12] = E.reshape(-1, 12)
    cams[:, 12:18] = K[:, :2, :].reshape(-1, 6)
    cams[:, 18:21] = -np.einsum("vji,vj->vi", E[:, :, :3], E[:, :, 3])
    return cams


VOTE_MODES = {"auto": 4, "float64": 1, "float64_cull": 3, "float64_cull1": 3}


def floater_votes(points: torch.Tensor, normals: torch.Tensor, depth: ArrayLike, intrinsics: ArrayLike,
                  cam_from_world: ArrayLike, mask: Optional[ArrayLike] = None, depth_threshold: float = 0.7,
                  votes: Optional[torch.Tensor] = None, mode: str = "auto", stats: Optional[dict] = None) -> torch.Tensor:
    """(N,) int32 votes of ``scripts/test.py:273-328``.  ``depth`` (V,H,W) float32 is the refined depth
    of the cached views (``:197-201``); with ``mask`` given, masked-out pixels read as 0 (``:194``).
    Pass ``votes`` to accumulate over several calls (views in chunks).

    ``mode``: ``"auto"`` (default) = ``"float64"`` or ``"float64_cull"``, chosen on the device from a sample of the
    workgroups (no host round trip): ``"float64_cull"`` skips, for each workgroup of 256 consecutive points, every view
    whose frustum the workgroup's bounding sphere cannot touch (conservative, same votes) -- several times faster when the
    views look at different parts of the scene, ~20 % slower when every view sees everything (an inward-facing ring);
    ``"float64_cull1"`` = the same without the super-tile masks (level 1 of the cull), for A/B;
    ``"float64"`` = every decision in float64, the fastest un-culled form on MI355X (division-free image-bounds
    test first, grazing second, reciprocal only for pairs that reach the lookup; a 256-byte table per view is built in a
    scratch buffer).  (The round-1 kernel -- a reciprocal for every pair in front of the camera, no scratch -- was removed
    in round 4, the float32 first pass of round 2 --
    same votes, 0.7-0.8x the rate -- was removed in round 3.)  ``stats`` (a dict) receives ``pairs`` and, in auto mode, the
    counters of the on-device choice (one host synchronisation)."""
    if mode not in VOTE_MODES:
        raise ValueError(f"mode must be one of {sorted(VOTE_MODES)}")
    dev = _require_gpu(points.device if isinstance(points, torch.Tensor) and points.is_cuda else None)
    pts = _gpu(points, dev, torch.float32)
    nrm = _gpu(normals, dev, torch.float32)
    if pts.dim() != 2 or pts.shape[1] != 3 or nrm.shape != pts.shape:
        raise ValueError("points and normals must both be (N,3)")
    d = _gpu(depth, dev, torch.float32)
    if d.dim() == 2:
        d = d[None]
    m = _gpu(mask, dev)
    if m is not None:
        if m.dim() == 2:
            m = m[None]
        m = m.view(torch.uint8) if m.dtype == torch.bool else (m > 0).view(torch.uint8)
        if m.shape != d.shape:
            raise ValueError("mask and depth shapes differ")
    if m is not None:
        # scripts/test.py:194 once, instead of a second gather per (point, view) pair: the cached map IS the
        # mask-zeroed depth in the reference (:197-201)
        d = torch.where(m.bool(), d, torch.zeros((), dtype=d.dtype, device=dev))
        m = None
    cams = torch.from_numpy(filter_cameras(intrinsics, cam_from_world)).to(dev)
    if cams.shape[0] != d.shape[0]:
        raise ValueError(f"{cams.shape[0]} cameras for {d.shape[0]} depth maps")
    accumulate = votes is not None
    if votes is None:
        votes = torch.empty(pts.shape[0], dtype=torch.int32, device=dev)
    V, H, W = d.shape
    # culling modes: two 256-byte tables per view, the decision counters, one mask of V bits per 65 536 points (level 1)
    cull_bytes = 512 * V + 64 + -(-pts.shape[0] // 65536) * (-(-V // 64)) * 8
    ws_bytes = {"float64": 256 * V, "float64_cull": cull_bytes, "float64_cull1": 512 * V + 64, "auto": cull_bytes}[mode]
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    fv = DDFilterViews(num_views=V, height=H, width=W, depth=d.data_ptr(), mask=None if m is None else m.data_ptr(),
                       cams=cams.data_ptr(), grazing_cos=GRAZING_COS, depth_threshold=float(depth_threshold),
                       workspace=ws.data_ptr(), workspace_bytes=ws.numel(), mode=VOTE_MODES[mode])
    rc = lib.dd_floater_votes(C.byref(fv), pts.data_ptr(), nrm.data_ptr(), pts.shape[0], votes.data_ptr(),
                              1 if accumulate else 0, _stream(dev))
    if rc < 0:
        raise DDCoreError(rc, lib.dd_filter_last_error().decode())
    if stats is not None:
        if mode == "auto":          # the two counters of the on-device choice (workgroup x view cells that survive / were sampled)
            d_ = ws[512 * V:512 * V + 16].view(torch.int64).tolist()
            stats.update(cull_sample_survived=int(d_[0]), cull_sample_cells=int(d_[1]), culled=bool(d_[0] * 10 < d_[1] * 9))
        stats.update(pairs=int(pts.shape[0]) * V, mode=mode)
    # asynchronous: temporaries freed here are only reused by later work on the same stream
    return votes


def compact_cloud(cloud: FusedCloud, votes: torch.Tensor, vote_threshold: int) -> FusedCloud:
    """``keep = votes < vote_threshold`` applied to every per-point field, stable, on the GPU
    (``dd_compact_cloud``; ``scripts/test.py:330-332``).  One host read for the kept count."""
    dev = cloud.points.device
    n = len(cloud)
    if n == 0:                      # nothing to compact (empty tensors have no device pointer)
        return cloud
    ptr = lambda t: None if t is None else t.data_ptr()
    mk = lambda t: None if t is None else torch.empty_like(t)
    o_xyz, o_rgb, o_nrm, o_pix, o_view = (mk(t) for t in (cloud.points, cloud.colors, cloud.normals, cloud.pixel_index,
                                                           cloud.view_index))
    src = DDCloudOut(xyz=ptr(cloud.points.contiguous()), normal=ptr(cloud.normals), rgb=ptr(cloud.colors),
                     pixel_index=ptr(cloud.pixel_index), view_index=ptr(cloud.view_index), capacity=n)
    dst = DDCloudOut(xyz=ptr(o_xyz), normal=ptr(o_nrm), rgb=ptr(o_rgb), pixel_index=ptr(o_pix), view_index=ptr(o_view),
                     capacity=n)
    kept = torch.zeros(1, dtype=torch.int64, device=dev)
    old = cloud.view_offsets.contiguous()
    new = torch.empty_like(old)
    nb = int(lib.dd_compact_workspace_bytes(n))
    ws = torch.empty(max(nb, 64), dtype=torch.uint8, device=dev)
    rc = lib.dd_compact_cloud(C.byref(src), n, votes.data_ptr(), int(vote_threshold), C.byref(dst), kept.data_ptr(),
                              old.data_ptr(), new.data_ptr(), old.numel() - 1, ws.data_ptr(), ws.numel(), _stream(dev))
    if rc < 0:
        raise DDCoreError(rc, lib.dd_filter_last_error().decode())
    k = int(kept.item())
    cut = lambda t: None if t is None else t[:k]
    return FusedCloud(points=o_xyz[:k], colors=cut(o_rgb), normals=cut(o_nrm), pixel_index=cut(o_pix), view_index=cut(o_view),
                      view_offsets=new, name=cloud.name)


def kept_per_view(cloud: FusedCloud, votes: torch.Tensor, vote_threshold: int) -> torch.Tensor:
    """(V,) int64 device: rows of each view that pass ``votes < vote_threshold`` -- what a rank announces before a
    filtered cloud is fused across GPUs (the rows themselves are compacted later, straight into the global cloud)."""
    keep = (votes < int(vote_threshold)).to(torch.int64)
    csum = torch.zeros(len(cloud) + 1, dtype=torch.int64, device=votes.device)
    torch.cumsum(keep, 0, out=csum[1:])
    at = csum[cloud.view_offsets - cloud.view_offsets[0]]
    return at[1:] - at[:-1]


def compact_into(cloud: FusedCloud, votes: torch.Tensor, vote_threshold: int, buffers: dict, row0: int, rows: int) -> None:
    """``dd_compact_cloud`` writing the kept rows of ``cloud`` at rows ``[row0, row0 + rows)`` of caller-owned tensors
    (keys ``points / normals / colors / pixel_index / view_index / packed``; ``packed`` = the 16-byte ``x, y, z, rgba``
    record built from points + colours).  Asynchronous; kept rows beyond ``rows`` would be dropped (``rows`` is the
    count announced by ``kept_per_view``)."""
    n = len(cloud)
    if n == 0 or rows == 0:
        return
    dev = cloud.points.device
    ptr = lambda t: None if t is None else t.data_ptr()
    at = lambda name: None if buffers.get(name) is None else buffers[name][row0:row0 + rows].data_ptr()
    need = {"normals": cloud.normals, "colors": cloud.colors, "pixel_index": cloud.pixel_index, "view_index": cloud.view_index}
    for name, srct in need.items():
        if buffers.get(name) is not None and srct is None:
            raise ValueError(f"buffer '{name}' given but the cloud has no such field")
    src = DDCloudOut(xyz=ptr(cloud.points.contiguous()), normal=ptr(cloud.normals), rgb=ptr(cloud.colors),
                     pixel_index=ptr(cloud.pixel_index), view_index=ptr(cloud.view_index), capacity=n)
    dst = DDCloudOut(xyz=at("points"), normal=at("normals"), rgb=at("colors"), pixel_index=at("pixel_index"),
                     view_index=at("view_index"), capacity=rows, xyz_rgba=at("packed"))
    kept = torch.zeros(1, dtype=torch.int64, device=dev)
    nb = int(lib.dd_compact_workspace_bytes(n))
    ws = torch.empty(max(nb, 64), dtype=torch.uint8, device=dev)
    rc = lib.dd_compact_cloud(C.byref(src), n, votes.data_ptr(), int(vote_threshold), C.byref(dst), kept.data_ptr(),
                              None, None, 0, ws.data_ptr(), ws.numel(), _stream(dev))
    if rc < 0:
        raise DDCoreError(rc, lib.dd_filter_last_error().decode())


def filter_floaters(cloud: FusedCloud, depth: ArrayLike, intrinsics: ArrayLike, cam_from_world: ArrayLike,
                    mask: Optional[ArrayLike] = None, config: Optional[FilteringConfig] = None):
    """``scripts/test.py:269-335``: returns ``(filtered_cloud, votes)``.  The reference filters points and
    colours (``:331-332``) and leaves ``final_normals`` untouched; here every per-point field is
    filtered so the cloud stays consistent."""
    cfg = config or FilteringConfig()
    if cloud.normals is None:
        raise ValueError("the floater filter needs per-point normals (scripts/test.py:291)")
    votes = floater_votes(cloud.points, cloud.normals, depth, intrinsics, cam_from_world, mask, cfg.depth_threshold)
    return compact_cloud(cloud, votes, cfg.vote_threshold), votes          # :330-332
