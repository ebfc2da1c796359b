#!/usr/bin/env python3
"""Round 3: does the class of the INPUT stacks matter once points and normals are in different classes?  Outputs class-pure
(points group 0, normals group 1, colours group 2); the four input stacks of garden185 allocated by the arena in group 0, 1,
2 in turn (and by torch, as the bench does).  GPU box only."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
from depthdensifier_amd import placement as pl

dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["mask_kind"] = "blob"
V, H, W = cfg["V"], cfg["H"], cfg["W"]
ids = np.arange(V)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)


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
    return dd.ViewBatch(sc["depth"], params, E, mask=sc["mask"], normal=sc["normal"], rgb=sc["rgb"], device=dev)


scene = bench.make_scene(cfg, ids, dev)
P = V * H * W
arena = pl.get_arena(dev)
outs, _ = arena.alloc({"points": ((P, 3), torch.float32, 0), "normals": ((P, 3), torch.float32, 1), "colors": ((P, 3), torch.uint8, 2)})
print({k: sorted(set(arena.classes_of(t))) for k, t in outs.items()}, flush=True)
b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=outs, device=dev)
for rep in range(2):
    print(f"inputs from torch: {time_it(mk_batch(scene), b):.3f} ms", flush=True)
    for g in (0, 1, 2):
        specs = {k: (tuple(v.shape), torch.uint8 if v.dtype == torch.bool else v.dtype, g) for k, v in scene.items() if v is not None}
        placed, deg = arena.alloc(specs)
        for k, v in placed.items():
            v.copy_(scene[k].view(torch.uint8) if scene[k].dtype == torch.bool else scene[k])
        sc = {k: (placed[k].view(torch.bool) if (k in placed and scene[k].dtype == torch.bool) else placed.get(k)) for k in scene}
        cls = sorted(set(arena.classes_of(placed["normal"])))
        print(f"inputs in group {g} (class {cls}, the class of {['points', 'normals', 'colours'][g]}): {time_it(mk_batch(sc), b):.3f} ms", flush=True)
        del placed, sc
