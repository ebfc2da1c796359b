#!/usr/bin/env python3
"""Round 3, fourth step: placement7 showed fast and slow REGIONS inside one 64 GiB allocation (and runs of consecutive
allocations of one class): the state belongs to physical address ranges of several GiB.  Map it: one big arena, a
16-view probe of the real kernel whose `normals` output is a 0.5 GiB window sliding through the arena, every other
buffer fixed.  Then move the fixed buffers (xyz / colours) somewhere else and map again: is the class of a region
absolute, or relative to where the other streams are?

usage: placement8.py [arena GiB] [step MiB]      GPU box only."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd

dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["mask_kind"] = "blob"
VP = 16
H, W = cfg["H"], cfg["W"]
ids = np.arange(VP)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (VP, 1))
E = bench.ring_poses(ids, 185)
ARENA_GIB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEP = (int(sys.argv[2]) if len(sys.argv) > 2 else 512) << 20

scene = bench.make_scene(cfg, ids, dev)
batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
P = batch.max_points
WIN = (P * 12 + (1 << 21) - 1) & ~((1 << 21) - 1)


def time_it(builder, n=8, warm=2):
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


def carve(arena, off, rows, dtype=torch.float32):
    nb = rows * 3 * (4 if dtype == torch.float32 else 1)
    return arena[off:off + nb].view(dtype).view(rows, 3)


def sweep(arena, x, c, role="normals"):
    out = []
    for off in range(0, arena.numel() - WIN + 1, STEP):
        cand = carve(arena, off, P)
        bufs = {"points": x, "normals": cand, "colors": c} if role == "normals" else {"points": cand, "normals": x, "colors": c}
        out.append(time_it(dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, device=dev)))
    return np.array(out)


def show(ts):
    lo, hi = ts.min(), ts.max()
    thr = (lo + hi) / 2
    print(f"   min {lo:.4f} max {hi:.4f} ms; one character per {STEP >> 20} MiB step, '#' = slow half, '.' = fast half, digits = position inside the range (0 fast .. 9 slow)", flush=True)
    line = "".join(str(min(9, int(10 * (t - lo) / max(hi - lo, 1e-9)))) for t in ts)
    for i in range(0, len(line), 64):
        print(f"   {i * (STEP >> 20) / 1024:6.1f} GiB  {line[i:i + 64]}", flush=True)


x_sep = torch.empty((P, 3), dtype=torch.float32, device=dev)
c_sep = torch.empty((P, 3), dtype=torch.uint8, device=dev)
arena = torch.empty(ARENA_GIB << 30, dtype=torch.uint8, device=dev)
print(f"arena {ARENA_GIB} GiB @ {arena.data_ptr():#x}; window {WIN / 2**20:.0f} MiB ({VP} views); xyz / colours in their own allocations", flush=True)
print("1. candidate = normals output, xyz + colours in separate allocations:", flush=True)
t1 = sweep(arena, x_sep, c_sep)
show(t1)
print("2. the same sweep again (stability):", flush=True)
t2 = sweep(arena, x_sep, c_sep)
show(t2)
print(f"   corr(1, 2) = {np.corrcoef(t1, t2)[0, 1]:.3f}", flush=True)
# move the fixed buffers: xyz at the fastest spot of the arena, then at the slowest
order = np.argsort(t1)
for label, k in (("fastest", order[0]), ("slowest", order[-1])):
    x_in = carve(arena, int(k) * STEP, P)
    print(f"3. xyz moved to the {label} window of the arena (offset {k * (STEP >> 20) / 1024:.1f} GiB), normals swept:", flush=True)
    t3 = sweep(arena, x_in, c_sep)
    show(t3)
    print(f"   corr(1, 3) = {np.corrcoef(t1, t3)[0, 1]:.3f}", flush=True)
print("4. candidate in the xyz role (normals in the separate allocation):", flush=True)
t4 = sweep(arena, x_sep, c_sep, role="points")
show(t4)
print(f"   corr(1, 4) = {np.corrcoef(t1, t4)[0, 1]:.3f}", flush=True)
np.save(str(ROOT / "gpurun_out" / "r3a" / "placement8_sweeps.npy"), np.stack([t1, t2, t4]))
