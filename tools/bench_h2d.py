#!/usr/bin/env python3
"""PCIe-inclusive rate: host-resident maps (pageable NumPy / pinned tensors) -> ViewBatch upload -> fused kernel.
Never the bench `value` (inputs resident in HBM); recorded in DESIGN.md section 5."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
dev = torch.device("cuda", 0)
H, W, V = 1080, 1920, 24
rng = np.random.default_rng(0)
depth = rng.uniform(0.5, 8.0, (V, H, W)).astype(np.float32)
mask = rng.uniform(size=(V, H, W)) < 0.85
normal = rng.standard_normal((V, H, W, 3)).astype(np.float32)
rgb = rng.integers(0, 256, (V, H, W, 3), dtype=np.uint8)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1)); E = bench.ring_poses(np.arange(V), V)
nbytes = depth.nbytes + mask.nbytes + normal.nbytes + rgb.nbytes
b = dd.CloudBuilder(V * H * W, normals=True, colors=True, pixel_index=False)
def run(d, m, n, c):
    b.reset(); b.append(dd.ViewBatch(d, params, E, mask=m, normal=n, rgb=c)); torch.cuda.synchronize()
pin = lambda a: torch.from_numpy(a).pin_memory()
cases = {"pageable numpy": (depth, mask, normal, rgb), "pinned tensors": (pin(depth), pin(mask), pin(normal), pin(rgb)),
         "device tensors": tuple(torch.from_numpy(a).to(dev) for a in (depth, mask, normal, rgb))}
for name, args in cases.items():
    run(*args)
    t0 = time.perf_counter()
    for _ in range(3): run(*args)
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:16s}: {dt*1e3:8.2f} ms for {V} views, {V*H*W/dt/1e9:7.2f} Gpix/s, host->device {nbytes/dt/1e9:6.1f} GB/s equivalent", flush=True)
