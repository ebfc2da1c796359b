"""CPU oracle (NumPy, float64) for DepthDensifier's per-view depth->points path.

TEST INFRASTRUCTURE ONLY.  This module is the checker for the HIP path and the
``cpu_baseline`` ("port") that ``bench.py`` times; it is never imported by the
product package ``depthdensifier_amd``.

It restates, in our own code, the arithmetic of the reference
(OpsiClear/DepthDensifier @ 2025-09-12).  Every function cites the reference
lines it follows (paths relative to the reference root):

* script formulation   ``scripts/test.py:194, 203-244, 262-266`` and the helper
  ``unproject_points`` ``scripts/test.py:79-90``;
* package formulation  ``src/depthdensifier/visualizer.py:291-344`` (points),
  ``:346-376`` (normals).

Third-party arithmetic that is NOT in the reference tree: ``pycolmap`` 3.12.5
(``pyproject.toml:19``) supplies ``image.cam_from_world().inverse() * points``
at ``scripts/test.py:233``.  COLMAP's ``Rigid3d`` maps ``x -> R x + t``; its
inverse is ``(R^T, -R^T t)``; ``Rigid3d * points`` applies the map row-wise.
That published behaviour is restated in :func:`rigid_inverse_apply`.

Parity pin: ``tests/golden/*.npz`` hold outputs captured from the reference's
own importable functions (``tests/golden/make_goldens.py``);
``tests/test_oracle_golden.py`` checks this file against them.  The inline
densify block itself (``scripts/test.py:194, 203-232``) is pinned by
``script_block_small.npz`` -- the arrays its own statements leave behind when
``make_goldens.py`` executes them, as they stand, on seeded inputs -- which
``tests/test_reference_blocks.py`` requires this file to reproduce bit for
bit (pixel order, colours, normals, float64 camera-frame points).  The pycolmap
step itself has no reference-side test, so it is pinned through the package
formulation (``visualizer.py:325-334`` computes the same map with
``np.linalg.inv`` and needs no pycolmap).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np


# --------------------------------------------------------------------------
# script formulation (scripts/test.py)
# --------------------------------------------------------------------------

def fold_cull_into_depth(depth: np.ndarray,
                         mask: Optional[np.ndarray],
                         conf: Optional[np.ndarray] = None,
                         conf_threshold: Optional[float] = None) -> np.ndarray:
    """Depth with every culled pixel forced to zero.

    Follows ``scripts/test.py:194`` (``refined_depth[~moge_mask] = 0``) but on a
    copy, so the caller's array is never mutated (the reference mutates in
    place, which is an aliasing quirk -- SURVEY.md section 8 a3 -- not arithmetic).

    The reference has no confidence map.  The build's confidence cull is
    defined as the reference fed ``mask & (conf > thr)``;
    that is what happens here when ``conf`` is given.
    """
    out = np.array(depth, copy=True)
    keep = np.ones(depth.shape, dtype=bool) if mask is None else (np.asarray(mask) > 0)
    if conf is not None:
        keep = keep & (np.asarray(conf) > conf_threshold)
    out[~keep] = 0
    return out


def strided_valid_pixels(culled_depth: np.ndarray, stride: int):
    """Row-major (y outer, x inner) coordinates of the visited, valid pixels.

    Follows ``scripts/test.py:205-212``: ``np.mgrid[0:h:s, 0:w:s]`` visits rows
    0, s, 2s, ... and columns 0, s, 2s, ...; a pixel is valid iff its (already
    mask-zeroed) depth is ``> 0`` (NaN -> False, +inf -> True); boolean-mask
    indexing keeps row-major order.
    """
    h, w = culled_depth.shape
    ys = np.arange(0, h, stride, dtype=np.int64)
    xs = np.arange(0, w, stride, dtype=np.int64)
    sub = culled_depth[::stride, ::stride]
    with np.errstate(invalid="ignore"):
        keep = sub > 0
    gy, gx = np.nonzero(keep)          # C-order == boolean-mask order
    return xs[gx], ys[gy]


def unproject_pinhole(px: np.ndarray, py: np.ndarray, depth_values: np.ndarray, params) -> np.ndarray:
    """Camera-frame points of integer pixel coordinates (no +0.5 offset).

    Follows ``scripts/test.py:79-90``: ``x = (u - cx) / fx * d``,
    ``y = (v - cy) / fy * d``, ``z = d`` with ``fx, fy, cx, cy = camera.params``
    (a 4-parameter PINHOLE camera).  int64 pixels and float64 params promote
    the float32/float16 depth to float64, exactly as NumPy does there.
    """
    fx, fy, cx, cy = (float(p) for p in params)
    xn = (px - cx) / fx
    yn = (py - cy) / fy
    with np.errstate(invalid="ignore", over="ignore"):      # inf / NaN depths pass through like in NumPy
        return np.stack([xn * depth_values, yn * depth_values, depth_values * np.float64(1.0)], axis=-1)


def rigid_inverse_apply(cam_from_world: np.ndarray, points_cam: np.ndarray) -> np.ndarray:
    """``cam_from_world.inverse() * points_cam`` of ``scripts/test.py:233``.

    pycolmap ``Rigid3d`` semantics restated: with ``cam_from_world = [R | t]``
    the inverse is ``[R^T | -R^T t]`` and applying it gives ``R^T p - R^T t``.
    """
    E = np.asarray(cam_from_world, dtype=np.float64)
    R, t = E[:3, :3], E[:3, 3]
    t_inv = -(R.T @ t)
    with np.errstate(invalid="ignore", over="ignore"):
        return points_cam @ R + t_inv      # rows: (R^T p)^T = p^T R


def densify_view_script(depth: np.ndarray,
                        params,
                        cam_from_world: np.ndarray,
                        mask: Optional[np.ndarray] = None,
                        normal: Optional[np.ndarray] = None,
                        rgb: Optional[np.ndarray] = None,
                        stride: int = 1,
                        conf: Optional[np.ndarray] = None,
                        conf_threshold: Optional[float] = None) -> dict:
    """One iteration of the densify block, ``scripts/test.py:203-233``.

    Returns ``points`` (N,3) float64 world, ``colors`` (N,3) uint8 (``:215-216``),
    ``normals`` (N,3) in the dtype of ``normal`` -- camera frame, NOT rotated
    (``:220``) -- plus ``pixel_index`` int64 = ``y*W + x`` of the full-resolution
    map (the reference keeps no indices; this is what lets order be checked
    bit-exactly).
    """
    culled = fold_cull_into_depth(depth, mask, conf, conf_threshold)
    px, py = strided_valid_pixels(culled, stride)
    d = culled[py, px]
    pts_cam = unproject_pinhole(px, py, d, params)
    pts_world = rigid_inverse_apply(cam_from_world, pts_cam)
    out = {
        "points": pts_world,
        "pixel_index": py * depth.shape[1] + px,
        "colors": None if rgb is None else rgb[py, px],
        "normals": None if normal is None else normal[py, px],
    }
    return out


def densify_view_script_literal(depth: np.ndarray, params, cam_from_world: np.ndarray, mask: np.ndarray,
                                normal: np.ndarray, rgb: np.ndarray, stride: int = 1) -> dict:
    """The densify block with the reference's OWN op sequence, for timing the reference's cost
    (``bench.py`` ``cpu_baseline.reference_formulation``): in-place mask fold (``scripts/test.py:194``),
    ``np.mgrid`` index grids (``:206``), fancy-index validity test (``:210``), boolean-mask gathers of the two
    int64 grids (``:212``), fancy-index gathers of colours / normals / depths with int64 index pairs
    (``:216, 220, 229``), ``np.stack`` of the pixel pairs (``:231``), ``unproject_points`` (``:79-90``:
    three float64 temporaries + ``np.stack``) and the rigid inverse (``:233``).  Same results as
    :func:`densify_view_script`, which reaches them with ``np.nonzero`` on a strided view -- about the same speed on the GPU
    box's host (27.7 against 26.2 Mpixels/s in ``BENCH_r03.json``; the "~20x" once written here came from the slower survey
    container and an earlier form of the port).
    """
    refined = np.array(depth, copy=True)
    refined[~np.asarray(mask, bool)] = 0
    h, w = refined.shape
    pixels_y, pixels_x = np.mgrid[0:h:stride, 0:w:stride]
    with np.errstate(invalid="ignore"):
        valid = refined[pixels_y, pixels_x] > 0
    px, py = pixels_x[valid], pixels_y[valid]
    colors = rgb[py, px]
    normals = normal[py, px]
    d = refined[py, px]
    points2d = np.stack([px, py], axis=-1)
    fx, fy, cx, cy = (float(p) for p in params)
    u, v = points2d[:, 0], points2d[:, 1]
    with np.errstate(invalid="ignore", over="ignore"):
        x = (u - cx) / fx * d
        y = (v - cy) / fy * d
        z = d
        cam = np.stack([x, y, z], axis=-1)
        world = rigid_inverse_apply(cam_from_world, cam)
    return {"points": world, "pixel_index": py * w + px, "colors": colors, "normals": normals}


# --------------------------------------------------------------------------
# package formulation (src/depthdensifier/visualizer.py)
# --------------------------------------------------------------------------

def _as_4x4(cam_from_world: np.ndarray) -> np.ndarray:
    E = np.asarray(cam_from_world, dtype=np.float64)
    if E.shape[0] == 3:                       # visualizer.py:325-327
        E = np.vstack([E, [0.0, 0.0, 0.0, 1.0]])
    return E


def depth_to_pointcloud_viz(depth: np.ndarray, K: np.ndarray, cam_from_world: np.ndarray,
                            color: Optional[np.ndarray] = None,
                            mask: Optional[np.ndarray] = None):
    """``COLMAPVisualizer._depth_to_pointcloud``, ``visualizer.py:291-344``.

    Full resolution; validity is ``mask > 0`` when a mask is given (NO depth
    test then) else ``depth > 0`` (``:311-314``); rays through the full
    ``inv(K)`` (skew honoured, ``:320-322``); world points through
    ``inv([E; 0 0 0 1])`` (``:325-334``); colours ``*255 -> uint8`` when their
    maximum is <= 1 (``:337-342``).
    """
    h, w = depth.shape
    flat_depth = depth.reshape(-1)
    with np.errstate(invalid="ignore"):
        keep = (np.asarray(mask).reshape(-1) > 0) if mask is not None else (flat_depth > 0)
    lin = np.nonzero(keep)[0]
    u = lin % w
    v = lin // w
    homog = np.stack([u, v, np.ones_like(u)], axis=-1)          # int64 like the reference grid
    rays = (np.linalg.inv(K) @ homog.T).T
    pts_cam = rays * flat_depth[keep][:, None]
    world_from_cam = np.linalg.inv(_as_4x4(cam_from_world))
    pts_h = np.hstack([pts_cam, np.ones((len(pts_cam), 1))])
    pts_world = (world_from_cam @ pts_h.T).T[:, :3]
    cols = None
    if color is not None:
        cols = color.reshape(-1, 3)[keep]
        if cols.size > 0 and np.max(cols) <= 1.0:
            cols = (cols * 255).astype(np.uint8)
    return pts_world, cols, lin


def transform_normals_viz(normal_map: np.ndarray, cam_from_world: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """``COLMAPVisualizer._transform_normals``, ``visualizer.py:346-376``.

    ``n_w = R^T n_c`` then ``n_w / (||n_w|| + 1e-8)`` for pixels with
    ``mask > 0``.
    """
    R = np.asarray(cam_from_world)[:3, :3]
    keep = np.asarray(mask).reshape(-1) > 0
    n_cam = normal_map.reshape(-1, 3)[keep]
    n_world = (R.T @ n_cam.T).T
    length = np.linalg.norm(n_world, axis=1, keepdims=True)
    return n_world / (length + 1e-8)


# --------------------------------------------------------------------------
# fuse (scripts/test.py:124-127, 238-240, 262-266)
# --------------------------------------------------------------------------

@dataclass
class OracleCloud:
    points: np.ndarray            # (N,3) float64
    colors: Optional[np.ndarray]  # (N,3) uint8
    normals: Optional[np.ndarray] # (N,3)
    pixel_index: np.ndarray       # (N,) int64, y*W+x inside the view
    view_offsets: np.ndarray      # (V+1,) int64, exclusive scan of per-view counts

    @property
    def view_index(self) -> np.ndarray:
        counts = np.diff(self.view_offsets)
        return np.repeat(np.arange(len(counts), dtype=np.int64), counts)


def fuse_views(per_view: Sequence[dict]) -> OracleCloud:
    """List-append + ``np.concatenate`` of ``scripts/test.py:238-240, 264-266``."""
    counts = np.array([len(v["points"]) for v in per_view], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def cat(key, empty_shape, dtype):
        parts = [v[key] for v in per_view]
        if any(p is None for p in parts):
            return None
        if not parts:
            return np.zeros(empty_shape, dtype)
        return np.concatenate(parts, axis=0)

    return OracleCloud(
        points=cat("points", (0, 3), np.float64),
        colors=cat("colors", (0, 3), np.uint8),
        normals=cat("normals", (0, 3), np.float32),
        pixel_index=cat("pixel_index", (0,), np.int64),
        view_offsets=offsets,
    )


def densify_scene_script(depth, params, cam_from_world, mask=None, normal=None, rgb=None,
                         stride: int = 1, conf=None, conf_threshold=None) -> OracleCloud:
    """The outer loop ``scripts/test.py:131`` over a stack of views + the fuse.

    ``depth`` (V,H,W); ``params`` (V,4) ``fx,fy,cx,cy``; ``cam_from_world``
    (V,3,4); optional ``mask`` (V,H,W), ``normal`` (V,H,W,3), ``rgb`` (V,H,W,3),
    ``conf`` (V,H,W).  Views are taken in index order.
    """
    V = depth.shape[0]
    pick = lambda a, i: None if a is None else a[i]
    views = [
        densify_view_script(depth[i], params[i], cam_from_world[i], pick(mask, i), pick(normal, i),
                            pick(rgb, i), stride, pick(conf, i), conf_threshold)
        for i in range(V)
    ]
    return fuse_views(views)


def densify_scene_viz(depth, K, cam_from_world, mask=None, normal=None, rgb=None) -> OracleCloud:
    """``add_rgbd_pointcloud`` (``visualizer.py:246-289``) over a stack of views, fused.

    Normals are produced only when both ``normal`` and ``mask`` are given
    (``visualizer.py:276-278``), rotated to the world frame and re-normalised.
    """
    V = depth.shape[0]
    views = []
    for i in range(V):
        m = None if mask is None else mask[i]
        pts, cols, lin = depth_to_pointcloud_viz(depth[i], K[i], cam_from_world[i],
                                                 None if rgb is None else rgb[i], m)
        nrm = None
        if normal is not None and m is not None:
            nrm = transform_normals_viz(normal[i], cam_from_world[i], m)
        views.append({"points": pts, "colors": cols, "normals": nrm, "pixel_index": lin.astype(np.int64)})
    return fuse_views(views)
