#!/usr/bin/env python3
"""Per-view streaming appends (V=1 batches, as scripts/test.py consumes views) vs one batch."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd

V = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = V
ids = np.arange(V)
scene = bench.make_scene(cfg, ids, dev)
H, W = cfg["H"], cfg["W"]
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)
big = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"])
singles = [dd.ViewBatch(scene["depth"][i:i + 1], params[i:i + 1], E[i:i + 1], mask=scene["mask"][i:i + 1],
                        normal=scene["normal"][i:i + 1], rgb=scene["rgb"][i:i + 1], view_index_base=i) for i in range(V)]
n = int(dd.count_valid(big).sum())
b = dd.CloudBuilder(n, normals=True, colors=True, pixel_index=False)
def run(batches):
    b.reset()
    for x in batches:
        b.append(x)
for name, batches in (("one batch", [big]), ("per-view appends", singles)):
    for _ in range(3): run(batches)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(batches)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:18s}: {dt*1e3:8.3f} ms for {V} views = {dt/V*1e6:7.1f} us/view, {V*H*W/dt/1e9:6.1f} Gpix/s")
