#!/usr/bin/env python3
"""Per-view time of the refiner's per-pixel kernel (dd_refine_apply) vs the tensor-op formulation."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from depthdensifier_amd.depth_refiner import DepthRefiner
H, W = 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
depth = torch.rand((H, W), device="cuda", generator=g) * 6 + 0.5
mask = torch.rand((H, W), device="cuda", generator=g) < 0.85
x = torch.rand(500, device="cuda", generator=g) * 6 + 0.5
y = 2 * x + 0.1 * torch.rand(500, device="cuda", generator=g)
r = DepthRefiner(use_fp16=False)
def tensor_path():
    out = torch.zeros_like(depth); out[mask] = r._lut_interpolate(depth[mask], x, y)
    from depthdensifier_amd.depth_refiner import median3x3
    out = median3x3(out); out[~mask] = 0; return out
for name, fn in (("dd_refine_apply kernel", lambda: r._apply_curve_hip(depth, mask, x, y)), ("tensor ops", tensor_path)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:24s}: {dt*1e6:8.1f} us per 1080p view ({H*W*9/dt/1e9:7.1f} GB/s of the 9 B/px minimum)")
print("equal:", bool(torch.equal(r._apply_curve_hip(depth, mask, x, y), tensor_path())))
