#!/usr/bin/env python3
"""Round 3, third step: placement6 showed the state is a property of EACH output array on its own (array n3 was 0.14 ms
faster in either role, xyz or normals; a write-only kernel, fill_ and copy_ do not see it).  Survey: how many arrays
are fast, how many levels are there, does the class survive free + re-allocate, does the allocation size matter?

The candidate is tested in the `normals` role of the full garden185 kernel with every other buffer fixed.  GPU box only."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def time_it(batch, builder, n=6, warm=2):
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


scene = bench.make_scene(cfg, ids, dev)
batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
P = batch.max_points
x0 = torch.empty((P, 3), dtype=torch.float32, device=dev)
c0 = torch.empty((P, 3), dtype=torch.uint8, device=dev)


def test(arr):
    b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers={"points": x0, "normals": arr[:P], "colors": c0}, device=dev)
    return time_it(batch, b)


def survey(title, make, k):
    print(title, flush=True)
    arrs, ts = [], []
    for i in range(k):
        a = make()
        arrs.append(a)
        ts.append(test(a))
    print("   " + " ".join(f"{t:.3f}" for t in ts), flush=True)
    return arrs, np.array(ts)


free0 = torch.cuda.mem_get_info(dev)[0]
print(f"free HBM {free0 / 2**30:.1f} GiB; candidate = (P,3) float32 = {P * 12 / 2**30:.3f} GiB", flush=True)
arrs, ts = survey(f"1. {K} candidates allocated one after the other (all held):", lambda: torch.empty((P, 3), dtype=torch.float32, device=dev), K)
thr = (ts.min() + ts.max()) / 2
fast = ts < thr
print(f"   fast {int(fast.sum())} / {K}  (threshold {thr:.3f}); addresses of the fast ones: " + " ".join(f"{arrs[i].data_ptr():#x}" for i in np.nonzero(fast)[0]), flush=True)
print("   re-test of the same arrays (is the class stable?):", flush=True)
print("   " + " ".join(f"{test(a):.3f}" for a in arrs), flush=True)

# free the slow ones and allocate again: does the class come back with the pages?
keep = [a for a, f in zip(arrs, fast) if f]
n_slow = K - len(keep)
arrs = None
torch.cuda.empty_cache()
arrs2, ts2 = survey(f"2. the {n_slow} slow ones freed (fast ones held), {n_slow} allocated again:", lambda: torch.empty((P, 3), dtype=torch.float32, device=dev), n_slow)
arrs2 = None
keep = None
torch.cuda.empty_cache()

for mult, label in ((2.0, "2x the size"), (0.5 * 2**33 / (P * 12), "exactly 4 GiB (smaller than needed: first rows only)")):
    rows = int(P * mult)
    if rows < P:
        continue
    a3, t3 = survey(f"3. candidates of {label} = {rows * 12 / 2**30:.3f} GiB, first P rows used:", lambda: torch.empty((rows, 3), dtype=torch.float32, device=dev), 8)
    a3 = None
    torch.cuda.empty_cache()
rows8 = (8 * 2**30) // 12 + 1
a4, t4 = survey("4. candidates of 8 GiB + 12 B:", lambda: torch.empty((rows8, 3), dtype=torch.float32, device=dev), 8)
a4 = None
torch.cuda.empty_cache()
# one 64 GiB arena cut into 12 candidates
arena = torch.empty(64 * 2**30, dtype=torch.uint8, device=dev)
step = (P * 12 + (1 << 21) - 1) & ~((1 << 21) - 1)
ts5 = []
for i in range(12):
    a = arena[i * step:i * step + P * 12].view(torch.float32).view(P, 3)
    ts5.append(test(a))
print("5. 12 candidates cut from ONE 64 GiB allocation:\n   " + " ".join(f"{t:.3f}" for t in ts5), flush=True)
