#!/usr/bin/env python3
"""Is the run-to-run spread of the hot kernel (2.6 .. 3.0 ms on the same binary) a matter of where the driver places the
buffers physically?  Re-allocate everything several times inside ONE process (really freeing to the driver in between)
and time the kernel on each allocation; then the same with all tensors carved out of one big arena.  GPU box only."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd

dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["mask_kind"] = "blob"
V, H, W = cfg["V"], cfg["H"], cfg["W"]
ids = np.arange(V)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)

def time_it(batch, builder, n=12):
    for _ in range(4):
        builder.reset(); builder.append(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        builder.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); builder.append(batch); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

print("separate allocations (torch caching allocator, emptied between rounds):")
for r in range(6):
    scene = bench.make_scene(cfg, ids, dev)
    batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
    builder = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=False, device=dev)
    print(f"  round {r}: {time_it(batch, builder):.3f} ms   depth@{scene['depth'].data_ptr():#x} xyz@{builder.xyz.data_ptr():#x}", flush=True)
    keep = scene
    del batch, builder, scene
    if r % 2 == 1:                      # every other round also holds some ballast so that the next allocation lands elsewhere
        ballast = torch.empty((r + 1) * 300_000_000, dtype=torch.uint8, device=dev)
    del keep
    torch.cuda.empty_cache()
print("one arena (a single 24 GB allocation, tensors carved at 2 MiB boundaries):")
for r in range(4):
    torch.cuda.empty_cache()
    arena = torch.empty(24 * 1024 ** 3, dtype=torch.uint8, device=dev)
    off = [0]
    def carve(shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        a = off[0]; off[0] = (a + n + (1 << 21) - 1) & ~((1 << 21) - 1)
        return arena[a:a + n].view(dtype).view(shape)
    scene = bench.make_scene(cfg, ids, dev)
    tens = {k: (carve(tuple(v.shape), v.dtype) if v is not None else None) for k, v in scene.items()}
    for k, v in scene.items():
        if v is not None:
            tens[k].copy_(v)
    del scene; torch.cuda.empty_cache()
    batch = dd.ViewBatch(tens["depth"], params, E, mask=tens["mask"], normal=tens["normal"], rgb=tens["rgb"], device=dev)
    P = batch.max_points
    bufs = {"points": carve((P, 3), torch.float32), "normals": carve((P, 3), torch.float32), "colors": carve((P, 3), torch.uint8)}
    builder = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, device=dev)
    print(f"  round {r}: {time_it(batch, builder):.3f} ms   arena@{arena.data_ptr():#x}", flush=True)
    del batch, builder, tens, bufs, arena
