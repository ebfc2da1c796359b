#!/usr/bin/env python3
"""Does the kernel time depend on where the WORKSPACE (look-back granules: 8 bytes per tile, polled by every workgroup) or the
per-view camera blocks live?  placement2 / placement3 moved the seven data streams; here they stay where they are and
only the two small tables move (offsets inside one pool, then fresh allocations)."""
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
scene = bench.make_scene(cfg, ids, dev)
batch = dd.ViewBatch(scene["depth"], params, bench.ring_poses(ids, V), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
builder = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=False, device=dev)


def time_it(n=12):
    for _ in range(3):
        builder.reset(); builder.append(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        builder.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); builder.append(batch); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


print(f"baseline (torch's own placement): {time_it():.3f} ms", flush=True)
nb = batch.workspace_bytes()
pool = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
rng = np.random.default_rng(0)
res = []
for j in range(20):
    off = (int(rng.integers(0, (1 << 30) - nb - 4096)) // 256) * 256 if j >= 8 else j * (1 << 27) // 8 * 8
    ws = pool[off:off + nb]
    ws.zero_()
    builder._ws_cache = ws
    t = time_it()
    res.append(t)
    print(f"workspace at pool + {off:>10d}: {t:.3f} ms", flush=True)
print(f"workspace placements: min {min(res):.3f} max {max(res):.3f} ms")
keep = []
res = []
for j in range(10):
    ws = torch.zeros(nb + 4096, dtype=torch.uint8, device=dev)       # fresh allocations, the old ones kept alive
    keep.append(ws)
    builder._ws_cache = ws
    res.append(time_it())
    print(f"fresh workspace {j}: {res[-1]:.3f} ms", flush=True)
print(f"fresh workspaces: min {min(res):.3f} max {max(res):.3f} ms")
p0 = batch.params
res = []
for j in range(10):
    off = int(rng.integers(0, (1 << 30) - p0.numel() * 4 - 4096)) // 256 * 256
    batch.params = pool[off:off + p0.numel() * p0.element_size()].view(p0.dtype).view(p0.shape)
    batch.params.copy_(p0)
    batch._cstruct = None
    res.append(time_it())
    print(f"camera blocks at pool + {off:>10d}: {res[-1]:.3f} ms", flush=True)
print(f"camera-block placements: min {min(res):.3f} max {max(res):.3f} ms")
