#!/usr/bin/env python3
"""CPU baseline of the floater filter: the NumPy restatement (oracle, one core) on 2 M points x a few views.
Lives under tests/ because only tests, smoke() and bench.py's cpu_baseline leg may use the oracle."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import filter_oracle as forc  # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
n, H, W = 2_000_000, 1080, 1920
pts = (rng.uniform(-1, 1, (n, 3)) * [3, 2, 3]).astype(np.float32)
nrm = rng.standard_normal((n, 3)).astype(np.float32)
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
depth = rng.uniform(1, 8, (views, H, W)).astype(np.float32)
K = np.tile(np.array([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1.0]]), (views, 1, 1))
E = np.zeros((views, 3, 4))
for v in range(views):
    a = 2 * np.pi * v / views
    c = np.array([4 * np.cos(a), 0.0, 4 * np.sin(a)])
    z = -c / np.linalg.norm(c); x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x); y = np.cross(z, x)
    R = np.stack([x, y, z]); E[v, :, :3] = R; E[v, :, 3] = -R @ c
t0 = time.perf_counter()
votes = forc.floater_votes(pts, nrm, depth, K, E)
dt = time.perf_counter() - t0
print(f"oracle (NumPy, 1 core): {n * views / dt / 1e6:.1f} Mpairs/s  (votes>0: {(votes > 0).mean() * 100:.1f} %)")
