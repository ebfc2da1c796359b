#!/usr/bin/env python3
"""One-off soak: floater votes of the HIP kernel vs the NumPy oracle on many random ring scenes (bit-exact)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
import depthdensifier_amd as dd
from oracle import filter_oracle as forc
from test_filter import _scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tot = 0
for seed in range(100, 100 + n):
    rng = np.random.default_rng(seed)
    V, H, W = int(rng.integers(2, 9)), int(rng.integers(20, 90)), int(rng.integers(20, 120))
    d = _scene(seed, V, H, W)
    depth = np.where(np.isfinite(d["depth"]) & (d["depth"] > 0), d["depth"], 1.0).astype(np.float32)
    cloud = dd.unproject_views(depth, d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"])
    K = dd.intrinsics_matrix(d["params"])
    thr = float(rng.choice([0.7, 0.9, 0.5]))
    votes = dd.floater_votes(cloud.points, cloud.normals, depth, K, d["cam_from_world"], mask=d["mask"], depth_threshold=thr).cpu().numpy()
    culled = np.where(d["mask"], depth, 0).astype(np.float32)
    ref = forc.floater_votes(cloud.points.cpu().numpy(), cloud.normals.cpu().numpy(), culled, K, d["cam_from_world"], depth_threshold=thr)
    assert np.array_equal(votes, ref), f"seed {seed}: {np.count_nonzero(votes != ref)} of {len(ref)} votes differ"
    tot += len(ref) * V
print(f"{n} scenes, {tot/1e6:.1f} M pairs: all votes equal")
