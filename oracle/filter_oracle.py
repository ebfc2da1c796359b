"""CPU oracle (NumPy, float64) for the multi-view floater filter, SURVEY.md 8(f) row f1.

TEST INFRASTRUCTURE ONLY (same rules as oracle/densify_oracle.py).

Restates ``scripts/test.py:58-76`` (``project_points``) and the vote loop
``scripts/test.py:273-335`` of the reference.  pycolmap accessors used there are restated from
COLMAP's published semantics: ``image.cam_from_world().matrix()`` = 3x4 ``[R|t]``,
``camera.calibration_matrix()`` = ``[[fx,0,cx],[0,fy,cy],[0,0,1]]``,
``image.projection_center()`` = ``-R^T t``.

NumPy promotion matters for parity and is reproduced as NumPy >= 2 (NEP 50) does it:
``depth_threshold * refined_depth`` is a Python float times a float32 array and therefore a
FLOAT32 product, which is then compared with the float64 projected depth.

Parity pin: ``project_points`` is importable from the reference (goldens ``filter_small.npz``).  The vote
loop is inline in ``main``; ``tests/golden/make_goldens.py`` picks its statements (``:273-332``) out of the
reference's syntax tree, executes them as they stand on a seeded scene and commits what they leave behind
(``votes_small.npz``: votes, keep mask, filtered points / colours); ``tests/test_reference_blocks.py`` requires
this file to reproduce those arrays bit for bit.
"""

from __future__ import annotations

import numpy as np

GRAZING_COS = 0.087          # scripts/test.py:295  (cos 85 deg)


def project_points(points3d: np.ndarray, cam_from_world: np.ndarray, K: np.ndarray):
    """``scripts/test.py:58-76``: pixel coordinates (N,2) and camera-frame depths (N,)."""
    E = np.asarray(cam_from_world)[:3, :]
    pts_h = np.hstack([points3d, np.ones((len(points3d), 1))])
    pts_cam = (E @ pts_h.T).T[:, :3]
    depths = pts_cam[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        normalized = pts_cam / (depths[:, np.newaxis] + 1e-8)
    pix_h = (np.asarray(K) @ normalized.T).T
    return pix_h[:, :2], depths


def projection_center(cam_from_world: np.ndarray) -> np.ndarray:
    E = np.asarray(cam_from_world, dtype=np.float64)
    return -(E[:3, :3].T @ E[:3, 3])


def floater_votes(points: np.ndarray, normals: np.ndarray, culled_depth: np.ndarray, K: np.ndarray,
                  cam_from_world: np.ndarray, depth_threshold: float = 0.7) -> np.ndarray:
    """Votes per point over all cached views, ``scripts/test.py:273-328``.

    ``culled_depth`` (V,H,W) float32 is ``refined_depth`` with the mask already folded in
    (``scripts/test.py:194, 197-201``); ``K`` (V,3,3); ``cam_from_world`` (V,3,4).
    """
    votes = np.zeros(len(points), dtype=int)
    for v in range(culled_depth.shape[0]):
        depth_map = culled_depth[v]
        h, w = depth_map.shape
        pix, depths = project_points(points, cam_from_world[v], K[v])
        dirs = points - projection_center(cam_from_world[v])
        with np.errstate(divide="ignore", invalid="ignore"):
            dirs = dirs / np.linalg.norm(dirs, axis=1)[:, np.newaxis]
            facing = np.sum(normals * -dirs, axis=1)
            not_grazing = facing > GRAZING_COS
            u, vv = pix[:, 0], pix[:, 1]
            inside = (u >= 0) & (u < w) & (vv >= 0) & (vv < h) & (depths > 0) & not_grazing
        if not np.any(inside):
            continue
        ui = u[inside].astype(int)
        vi = vv[inside].astype(int)
        seen = depth_map[vi, ui]
        with np.errstate(invalid="ignore"):
            has_depth = seen > 0
            floater = depths[inside][has_depth] < depth_threshold * seen[has_depth]     # float32 product (NEP 50)
        idx = np.where(inside)[0][has_depth][floater]
        votes[idx] += 1
    return votes


def filter_floaters(points, colors, normals, culled_depth, K, cam_from_world, vote_threshold: int = 5,
                    depth_threshold: float = 0.7):
    """``scripts/test.py:330-332``: keep ``votes < vote_threshold``; points and colours are filtered
    (the reference leaves ``final_normals`` unfiltered -- returned here filtered as a convenience)."""
    votes = floater_votes(points, normals, culled_depth, K, cam_from_world, depth_threshold)
    keep = votes < vote_threshold
    return points[keep], None if colors is None else colors[keep], normals[keep], votes
