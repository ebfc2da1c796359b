#!/usr/bin/env python3
"""Chains of k-view appends over a 185-view 1080p scan: one stream (calls behind one another) against chained across the two side streams.  On a
tree with tools/experiments/r06_early_gate.patch applied (a gate that opens when every workgroup of the previous call is running: built and
measured in round 6, not kept -- profiles/r06_early_gate.txt) the chained chain runs with both gates.  GPU box only.
    python tools/experiments/early_gate.py [--per-call 2,4,8,16]"""
import argparse, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
from depthdensifier_amd import _lib

ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=185); ap.add_argument("--per-call", default="1,2,4,8,16"); ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = a.views
ids = np.arange(a.views)
scene = bench.make_scene(cfg, ids, dev)
H, W, V = cfg["H"], cfg["W"], a.views
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)
alg = None
for k in [int(x) for x in a.per_call.split(",")]:
    res = {}
    patched = hasattr(_lib, "DD_LAB_LATE_GATE")
    for mode in (("one stream", "late gate", "early gate") if patched else ("one stream", "chained")):
        lab = _lib.DD_LAB_LATE_GATE if mode == "late gate" else 0
        subs = [dd.ViewBatch(scene["depth"][i:i + k], params[i:i + k], E[i:i + k], mask=scene["mask"][i:i + k], normal=scene["normal"][i:i + k],
                             rgb=scene["rgb"][i:i + k], view_index_base=i, device=dev, lab=lab) for i in range(0, V, k)]
        b = dd.CloudBuilder(V * H * W, normals=True, colors=True, pixel_index=False, device=dev, exclusive_gpu=True)
        b.CHAIN_MAX_TILES = 0 if mode == "one stream" else 1 << 20
        ts = []
        for r in range(a.rounds + 2):
            b.reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s in subs: b.append(s)
            b.join(); e1.record(); torch.cuda.synchronize()
            n = b.check()
            if r >= 2: ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[mode] = (ts[len(ts) // 2], ts[0], n)
        del b
    bytes_ = bench.algorithmic_bytes(cfg, V, res["one stream"][2], False)
    print(f"{k:3d} views per call, {len(subs):3d} calls: " + "   ".join(f"{m}: {t[0] * 1e3 / len(subs):7.2f} us/call (min {t[1] * 1e3 / len(subs):7.2f}) frac {bytes_ / (t[0] * 1e-3) / 8e12:.3f}" for m, t in res.items())
          + ("   points equal" if len({t[2] for t in res.values()}) == 1 else "   POINT COUNTS DIFFER"))
