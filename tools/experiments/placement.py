#!/usr/bin/env python3
"""Does scatter-kernel speed depend on where the output rows live?  One process, many placements."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd

V = 96
dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = V
ids = np.arange(V)
scene = bench.make_scene(cfg, ids, dev)
H, W = cfg["H"], cfg["W"]
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)
batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"])
n = int(dd.count_valid(batch).sum())
arena = torch.empty(n * 27 + (256 << 20), dtype=torch.uint8, device=dev)
print("arena base %x  depth %x mask %x normal %x rgb %x" % (arena.data_ptr(), batch.depth.data_ptr(), batch.mask.data_ptr(),
                                                         batch.normal.data_ptr(), batch.rgb.data_ptr()))

def timed(offs):
    ox, on, oc = offs
    b = dd.CloudBuilder(n, normals=False, colors=False, pixel_index=False)
    b.xyz = arena[ox:ox + n * 12].view(torch.float32).view(n, 3)
    b.normal = arena[on:on + n * 12].view(torch.float32).view(n, 3)
    b.rgb = arena[oc:oc + n * 3].view(n, 3)
    plan = dd.plan_batch(batch, b.cursor)
    ts = []
    for _ in range(6):
        b.reset(); plan = dd.plan_batch(batch, b.cursor, reuse=plan)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.scatter(batch, plan); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))

rng = np.random.default_rng(0)
base_n = n * 12
results = []
for trial in range(14):
    if trial == 0:
        offs = (0, (base_n + 255) // 256 * 256, (2 * base_n + 511) // 256 * 256)            # packed, 256-B aligned
    else:
        gaps = rng.integers(0, 1 << 14, 3) * 4096 + rng.integers(0, 16, 3) * 256            # random 256-B aligned gaps
        ox = int(gaps[0]); on = ox + base_n + int(gaps[1]); on = on // 256 * 256
        oc = on + base_n + int(gaps[2]); oc = oc // 256 * 256
        offs = (ox, on, oc)
    t = timed(offs)
    results.append(t)
    print(f"trial {trial:2d} offsets {offs[0]:>12d} {offs[1]:>12d} {offs[2]:>12d}: {t:.3f} ms  ({t/V*1000:.2f} us/view)")
print("min %.3f max %.3f spread %.1f %%" % (min(results), max(results), 100 * (max(results) / min(results) - 1)))
