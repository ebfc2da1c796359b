#!/usr/bin/env python3
"""Throughput of the strided path (the reference's `downsample_density`, default 32: scripts/test.py:37, :206)."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
dev = torch.device("cuda", 0)
H, W, V = 1080, 1920, 185
g = torch.Generator(device=dev).manual_seed(1)
depth = torch.empty((V, H, W), device=dev).uniform_(0.5, 8.0, generator=g)
mask = torch.rand((V, H, W), device=dev, generator=g) < 0.85
normal = torch.nn.functional.normalize(torch.randn((V, H, W, 3), device=dev, generator=g), dim=-1)
rgb = torch.randint(0, 256, (V, H, W, 3), device=dev, generator=g, dtype=torch.uint8)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1)); E = bench.ring_poses(np.arange(V), V)
for s in (1, 2, 3, 4, 8, 16, 32):
    batch = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, stride=s)
    n = int(dd.count_valid(batch).sum())
    b = dd.CloudBuilder(n, normals=True, colors=True, pixel_index=False)
    for _ in range(3): b.reset(); b.append(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): b.reset(); b.append(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    samples = V * ((H + s - 1) // s) * ((W + s - 1) // s)
    print(f"stride {s:2d}: {dt*1e3:7.3f} ms/step, {samples/1e6:8.2f} M samples, {samples/dt/1e9:7.2f} Gsamples/s, {n/dt/1e9:6.2f} Gpoints/s out", flush=True)
