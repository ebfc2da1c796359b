#!/usr/bin/env python3
"""Do two chains of one-view calls on two HIP streams overlap on the GPU?  Two independent clouds (own cursor, own workspace), appends
interleaved from one host thread: if the streams' kernels run side by side, 2 x 185 calls take about as long as 185 (feasibility of
chaining consecutive calls of ONE cloud across two streams: DESIGN.md section 4, streaming).  GPU box only."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
os.environ.setdefault("DD_EXCLUSIVE_GPU", "1")
import bench, depthdensifier_amd as dd

dev = torch.device("cuda", 0)
V = 185
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = V; cfg["mask_kind"] = "blob"
H, W = cfg["H"], cfg["W"]
ids = np.arange(V)
scene = bench.make_scene(cfg, ids, dev)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
batch = dd.ViewBatch(scene["depth"], params, bench.ring_poses(ids, V), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
subs = [batch.slice(i, i + 1) for i in range(V)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
builders = [dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=False, device=dev, placement="first") for _ in range(2)]

def run(two: bool):
    for b in builders: b.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    if two:
        for s in subs:
            for k in (0, 1):
                with torch.cuda.stream(streams[k]):
                    builders[k].append(s)
        for k in (0, 1):
            torch.cuda.current_stream().wait_stream(streams[k])
    else:
        for k in (0, 1):
            for s in subs:
                builders[k].append(s)
    e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), host * 1e3

for two in (False, True, False, True):
    ts = [run(two) for _ in range(4)][1:]
    print(("two streams, interleaved" if two else "one stream, back to back "), "2 x 185 one-view calls:", " ".join(f"{t:.2f} ms (host {h:.2f})" for t, h in ts))
print("rows:", [b.check() for b in builders])
