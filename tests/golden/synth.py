"""Seeded synthetic inputs shared by ``make_goldens.py`` and the tests.

Own code (no reference content): random COLMAP-style poses, depth / mask /
normal / rgb / confidence stacks.  ``numpy.random.default_rng`` streams are
stable across NumPy releases for the methods used here; the VGA fixture also
stores SHA-256 digests of the generated inputs so a drift would be detected.
"""

from __future__ import annotations

import hashlib

import numpy as np


def random_pose(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])
    t = rng.standard_normal(3)
    return np.hstack([R, t[:, None]])


def make_views(seed, V, H, W, rho=0.8, specials=False, depth_dtype=np.float32):
    rng = np.random.default_rng(seed)
    depth = rng.uniform(0.5, 5.0, size=(V, H, W)).astype(depth_dtype)
    if specials:
        flat = depth.reshape(V, -1)
        for v in range(V):
            idx = rng.choice(H * W, size=min(24, H * W), replace=False)
            vals = [0.0, -1.5, np.nan, np.inf, -np.inf, -0.0]
            for k, i in enumerate(idx):
                flat[v, i] = vals[k % len(vals)]
    mask = rng.uniform(size=(V, H, W)) < rho
    normal = rng.standard_normal((V, H, W, 3)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=-1, keepdims=True)
    rgb = rng.integers(0, 256, size=(V, H, W, 3), dtype=np.uint8)
    conf = rng.uniform(size=(V, H, W)).astype(np.float32)
    E = np.stack([random_pose(rng) for _ in range(V)])
    return dict(depth=depth, mask=mask, normal=normal, rgb=rgb, conf=conf, cam_from_world=E)


def pinhole_K(params):
    fx, fy, cx, cy = params
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])


def refiner_case(seed, H=96, W=128, n_pts=900):
    """Mono depth = distorted true depth; sparse points = true-depth unprojections + noise + outliers."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    true_depth = 3.0 + 1.2 * np.sin(xs / 17.0) + 0.8 * np.cos(ys / 11.0)
    mono = (0.35 * true_depth ** 1.15 + 0.1).astype(np.float32)        # unknown monotone distortion
    mask = rng.uniform(size=(H, W)) < 0.9
    fx, fy, cx, cy = 110.0, 112.0, W / 2.0, H / 2.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    E = random_pose(rng)
    u = rng.uniform(2, W - 3, n_pts); v = rng.uniform(2, H - 3, n_pts)
    d = true_depth[v.astype(int), u.astype(int)] * (1 + 0.01 * rng.standard_normal(n_pts))
    d[: n_pts // 20] *= rng.uniform(1.5, 3.0, n_pts // 20)            # outliers
    cam = np.stack([(u - cx) / fx * d, (v - cy) / fy * d, d], axis=-1)
    world = (cam - E[:, 3]) @ E[:, :3]                                 # R^T (p - t)
    behind = rng.standard_normal((40, 3)) * 0.2 - E[:, :3].T @ E[:, 3] - 3.0 * E[2, :3]
    return dict(depth=mono, mask=mask, points3D=np.vstack([world, behind]), cam_from_world=E, K=K)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
