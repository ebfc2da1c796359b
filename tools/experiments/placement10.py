#!/usr/bin/env python3
"""Round 3: the product's placement (CloudBuilder(placement="probed"), the HBM zone arena) against "first", re-allocated R times
in one process; then the input stacks moved into the arena's third class as well.  GPU box only.
usage: placement10.py [rounds] [workload]"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
from depthdensifier_amd import placement as pl

dev = torch.device("cuda", 0)
wl = sys.argv[2] if len(sys.argv) > 2 else "garden185"
cfg = dict(bench.WORKLOADS[wl]); cfg["mask_kind"] = "blob"
V, H, W = cfg["V"], cfg["H"], cfg["W"]
ids = np.arange(V)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def time_it(batch, builder, n=10, warm=3):
    for _ in range(warm):
        builder.reset(); builder.append(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        builder.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); builder.append(batch); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def mk_batch(sc):
    return dd.ViewBatch(sc["depth"], params, E, mask=sc["mask"], normal=sc["normal"], rgb=sc["rgb"], conf=sc.get("conf"),
                        conf_threshold=cfg.get("conf"), device=dev)


scene = bench.make_scene(cfg, ids, dev)
batch = mk_batch(scene)
P = batch.max_points
print(f"{wl}: {V} views, cloud capacity {P} rows", flush=True)
for r in range(R):
    for mode, layout in (("first", ""), ("probed", "rotated")):
        import os
        os.environ["DD_PLACEMENT_LAYOUT"] = layout
        t0 = time.perf_counter()
        b = dd.CloudBuilder(P, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=False, device=dev, placement=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = time_it(batch, b)
        rep = b.placement.as_dict() if b.placement is not None else None
        alg = bench.algorithmic_bytes(cfg, V, int(b.cursor.item()), False)
        print(f"round {r} {mode:7s} {layout:11s}: kernel {t:.3f} ms  frac {alg / t / 1e6 / 8000:.3f}   builder {dt:.2f} s   {rep}", flush=True)
        if rep is not None:
            pass
        del b
        torch.cuda.empty_cache()
        pl.trim(dev)
    ballast = torch.empty((r + 1) * 3 * 2**30, dtype=torch.uint8, device=dev)      # the next round starts somewhere else
print(pl.get_arena(dev).stats(), flush=True)
