#!/usr/bin/env python3
"""Companion of placement2.py: inside ONE arena (one physical placement), does the kernel time depend on the RELATIVE
offsets of the seven streams (depth, mask, normal, rgb in; xyz, normal, rgb out)?  Carve them with different paddings."""
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
scene = bench.make_scene(cfg, ids, dev)
arena = torch.empty(26 * 1024 ** 3, dtype=torch.uint8, device=dev)
rng = np.random.default_rng(0)

def time_it(batch, builder, n=10):
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

for trial in range(14):
    gran = [0, 256, 4096, 65536, 1 << 21][trial % 5] if trial < 5 else None
    off = [0]
    pads = []
    def carve(shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        pad = (gran * (len(pads) + 1) if gran is not None else int(rng.integers(0, 1 << 24)) // 256 * 256)
        pads.append(pad)
        a = (off[0] + pad + 255) & ~255
        off[0] = a + n
        return arena[a:a + n].view(dtype).view(shape)
    tens = {k: (carve(tuple(v.shape), v.dtype) if v is not None else None) for k, v in scene.items()}
    for k, v in scene.items():
        if v is not None:
            tens[k].copy_(v)
    batch = dd.ViewBatch(tens["depth"], params, E, mask=tens["mask"], normal=tens["normal"], rgb=tens["rgb"], device=dev)
    P = batch.max_points
    bufs = {"points": carve((P, 3), torch.float32), "normals": carve((P, 3), torch.float32), "colors": carve((P, 3), torch.uint8)}
    builder = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, device=dev)
    print(f"trial {trial:2d} pads {('k*%d' % gran) if gran is not None else 'random'}: {time_it(batch, builder):.3f} ms", flush=True)
    del batch, builder, tens, bufs
