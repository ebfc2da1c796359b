#!/usr/bin/env python3
"""Whole `DepthRefiner.refine_depth` per 1080p view (fit + per-pixel kernel), inputs resident on the GPU."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from depthdensifier_amd.depth_refiner import DepthRefiner
H, W = 1080, 1920
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
K = torch.tensor([[1500.0, 0, W / 2], [0, 1500.0, H / 2], [0, 0, 1]], device=dev)
E = torch.eye(4, device=dev)[:3]
for npts in (2000, 20000, 200000):
    xy = torch.rand((npts, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)
    z = torch.rand(npts, device=dev, generator=g) * 6 + 1
    pts = torch.stack([(xy[:, 0] - W / 2) / 1500 * z, (xy[:, 1] - H / 2) / 1500 * z, z], -1)
    depth = (torch.rand((H, W), device=dev, generator=g) * 0.2 + 1.0) * 2.0
    mask = torch.rand((H, W), device=dev, generator=g) < 0.85
    for fp16 in (False,):
        r = DepthRefiner(use_fp16=fp16, verbose=0)
        fn = lambda: r.refine_depth(depth, None, pts, E, K, mask=mask, return_tensor=True)
        for _ in range(3): out = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): out = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{npts:7d} sparse points: {dt*1e3:7.3f} ms per view, n_corr {out['num_correspondences']}", flush=True)
