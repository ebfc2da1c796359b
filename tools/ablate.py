#!/usr/bin/env python3
"""Interleaved A/B timing of the hot kernel under different field sets / tuning words (GPU box).

    python tools/ablate.py [--views 64] [--rounds 7] [--tunings 0,2,4]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import depthdensifier_amd as dd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--tunings", default="0,8")
    ap.add_argument("--workload", default="garden185")
    ap.add_argument("--fields", default="full,nocolor,nonormal,xyz,xyz_nomask")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(bench.WORKLOADS[a.workload]); cfg["V"] = a.views
    ids = np.arange(a.views)
    scene = bench.make_scene(cfg, ids, dev)
    H, W = cfg["H"], cfg["W"]
    params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (a.views, 1))
    E = bench.ring_poses(ids, a.views)
    field_sets = {
        "full": dict(mask=True, normal=True, rgb=True),
        "nocolor": dict(mask=True, normal=True, rgb=False),
        "nonormal": dict(mask=True, normal=False, rgb=True),
        "xyz": dict(mask=True, normal=False, rgb=False),
        "xyz_nomask": dict(mask=False, normal=False, rgb=False),
    }
    variants = []
    for fname in a.fields.split(","):
        fs = field_sets[fname]
        if any(fs[k] and scene[k] is None for k in fs):
            continue
        for t in (int(x, 0) for x in a.tunings.split(",")):
            c = dict(cfg, mask=fs["mask"], normal=fs["normal"], rgb=fs["rgb"])
            batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"] if fs["mask"] else None,
                                 normal=scene["normal"] if fs["normal"] else None, rgb=scene["rgb"] if fs["rgb"] else None,
                                 device=dev, tuning=t)
            n = int(dd.count_valid(batch).sum().item())
            b = dd.CloudBuilder(n, normals=fs["normal"], colors=fs["rgb"], pixel_index=False, device=dev)
            variants.append(dict(name=f"{fname}/t{t:#x}", batch=batch, builder=b, n=n,
                                 bytes=bench.algorithmic_bytes(c, a.views, n, False), times=[]))
    for r in range(a.rounds + 1):
        for v in variants:
            v["builder"].reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); v["builder"].append(v["batch"]); e1.record()
            torch.cuda.synchronize()
            if r:
                v["times"].append(e0.elapsed_time(e1))
    pix = a.views * H * W
    print(f"{'variant':24s} {'med ms':>8s} {'min ms':>8s} {'Gpix/s':>8s} {'alg GB/s':>9s} {'frac8T':>7s}")
    for v in variants:
        med, mn = float(np.median(v["times"])), float(np.min(v["times"]))
        print(f"{v['name']:24s} {med:8.3f} {mn:8.3f} {pix/med/1e6:8.1f} {v['bytes']/med/1e6:9.1f} {v['bytes']/med/1e6/8000:7.3f}")
        v["builder"].finish()


if __name__ == "__main__":
    main()
