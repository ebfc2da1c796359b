#!/usr/bin/env python3
"""Round 3: WHERE does the placement state of the hot kernel live, and can a short probe see it?  One process, garden185.

  A. inputs fixed, the three output arrays re-allocated R times (old ones held so the driver hands out other pages)
  B. outputs fixed, the four input stacks re-allocated R times
  C. per allocation of A: a probe over the first 16 views against the full 185-view time (is the probe predictive?)
  D. one allocation, the 185 views in 5 groups of 37, each written where the full run writes it: do REGIONS of one
     allocation differ?

GPU box only.  Prints a table; the summary goes to profiles/r03_placement_probe.txt by hand."""
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
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def time_it(batch, builder, n=10, warm=3):
    for _ in range(warm):
        builder.reset(); builder.append(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        builder.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); builder.append(batch); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def mk_batch(sc, lo=0, hi=V):
    return dd.ViewBatch(sc["depth"][lo:hi], params[lo:hi], E[lo:hi], mask=sc["mask"][lo:hi], normal=sc["normal"][lo:hi], rgb=sc["rgb"][lo:hi], device=dev)


def mk_out(P):
    return {"points": torch.empty((P, 3), dtype=torch.float32, device=dev), "normals": torch.empty((P, 3), dtype=torch.float32, device=dev),
            "colors": torch.empty((P, 3), dtype=torch.uint8, device=dev)}


scene = bench.make_scene(cfg, ids, dev)
batch = mk_batch(scene)
P = batch.max_points
probe = mk_batch(scene, 0, 16)

print(f"A. inputs fixed, outputs re-allocated {R} times (full 185 views | 16-view probe x 185/16)", flush=True)
held, res_a = [], []
for r in range(R):
    bufs = mk_out(P)
    b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, device=dev)
    full = time_it(batch, b)
    pr = time_it(probe, b, n=20)
    res_a.append((full, pr))
    print(f"   out#{r}: full {full:.3f} ms   probe16 {pr:.4f} ms (x11.56 = {pr * 185 / 16:.3f})   xyz@{bufs['points'].data_ptr():#x}", flush=True)
    held.append(bufs)
fa = np.array(res_a)
print(f"   full: min {fa[:, 0].min():.3f} median {np.median(fa[:, 0]):.3f} max {fa[:, 0].max():.3f};  corr(full, probe) = {np.corrcoef(fa[:, 0], fa[:, 1])[0, 1]:.3f}", flush=True)
best = int(np.argmin(fa[:, 0])); worst = int(np.argmax(fa[:, 0]))
best_bufs, worst_bufs = held[best], held[worst]
held = None
torch.cuda.empty_cache()

print(f"B. outputs fixed (best of A = out#{best} and worst = out#{worst}), inputs re-allocated {R} times", flush=True)
bb = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=best_bufs, device=dev)
bw = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=worst_bufs, device=dev)
held_in, res_b = [scene], []
for r in range(R):
    sc = {k: (None if v is None else v.clone()) for k, v in scene.items()}
    bt = mk_batch(sc)
    t1, t2 = time_it(bt, bb), time_it(bt, bw)
    res_b.append((t1, t2))
    print(f"   in#{r}: on best out {t1:.3f} ms   on worst out {t2:.3f} ms   depth@{sc['depth'].data_ptr():#x}", flush=True)
    held_in.append(sc)
    if len(held_in) > 5:
        held_in.pop(1)
fb = np.array(res_b)
print(f"   on best out: min {fb[:, 0].min():.3f} max {fb[:, 0].max():.3f};  on worst out: min {fb[:, 1].min():.3f} max {fb[:, 1].max():.3f}", flush=True)
held_in = None
torch.cuda.empty_cache()

print("D. regions of one allocation: 5 groups of 37 views, each group written at the rows the full run gives it", flush=True)
full_offs = None
for name, bufs in (("best", best_bufs), ("worst", worst_bufs)):
    b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, device=dev)
    b.reset(); offs = b.append(batch); torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    row = []
    for g in range(5):
        lo, hi = g * 37, (g + 1) * 37
        sub = mk_batch(scene, lo, hi)
        bg = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, start=int(offs[lo]), device=dev)
        row.append(time_it(sub, bg, n=12))
    print(f"   {name}: " + "  ".join(f"{t:.4f}" for t in row) + f"   sum {sum(row):.3f} ms", flush=True)
