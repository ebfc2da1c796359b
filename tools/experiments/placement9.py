#!/usr/bin/env python3
"""Round 3, fifth step: placement8 showed the state is RELATIVE -- the kernel is fast when the xyz and the normals
outputs lie in different physical "zones" of HBM and slow when they share one.  Map the relation: one arena over most
of the free HBM cut into cells; T[i, j] = 16-view probe with xyz in cell i and normals in cell j.  Then, with xyz and
normals in different zones, sweep the colours, and sweep the INPUT stacks.

usage: placement9.py [arena GiB] [cell GiB]      GPU box only."""
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
ARENA_GIB = int(sys.argv[1]) if len(sys.argv) > 1 else 224
CELL = int(float(sys.argv[2]) * 2**30) if len(sys.argv) > 2 else 4 << 30

scene = bench.make_scene(cfg, ids, dev)
P = VP * H * W
c_sep = torch.empty((P, 3), dtype=torch.uint8, device=dev)
free = torch.cuda.mem_get_info(dev)[0]
ARENA_GIB = min(ARENA_GIB, int(free / 2**30) - 6)
arena = torch.empty(ARENA_GIB << 30, dtype=torch.uint8, device=dev)
NC = arena.numel() // CELL
print(f"arena {ARENA_GIB} GiB @ {arena.data_ptr():#x}, {NC} cells of {CELL / 2**30:.1f} GiB; probe = {VP} views, an output window is {P * 12 / 2**20:.0f} MiB", flush=True)


def carve(off, rows, tail, dtype):
    nb = rows * int(np.prod(tail)) * torch.empty((), dtype=dtype).element_size()
    return arena[off:off + nb].view(dtype).view((rows,) + tail)


def mk_batch(sc):
    return dd.ViewBatch(sc["depth"], params, E, mask=sc["mask"], normal=sc["normal"], rgb=sc["rgb"], device=dev)


def time_it(batch, bufs, n=5, warm=1):
    builder = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers=bufs, device=dev)
    for _ in range(warm):
        builder.reset(); builder.append(batch)
    ts = []
    for _ in range(n):
        builder.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); builder.append(batch); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


batch = mk_batch(scene)
T = np.zeros((NC, NC))
for i in range(NC):
    x = carve(i * CELL, P, (3,), torch.float32)
    for j in range(NC):
        nrm = carve(j * CELL + (CELL // 2 if i == j else 0), P, (3,), torch.float32)     # same cell: the other half of it
        T[i, j] = time_it(batch, {"points": x, "normals": nrm, "colors": c_sep})
lo, hi = np.percentile(T, 2), np.percentile(T, 98)
print(f"T[xyz cell, normals cell]: 2nd / 98th percentile {lo:.4f} / {hi:.4f} ms; digit = position in that range (0 fast .. 9 slow)", flush=True)
for i in range(NC):
    print(f"   x{i:02d} " + "".join(str(int(np.clip(10 * (T[i, j] - lo) / (hi - lo), 0, 9))) for j in range(NC)), flush=True)
np.save(str(ROOT / "gpurun_out" / "r3a" / "placement9_T.npy"), T)

# classes: cells i, k are in one zone when their rows agree
slow = T > (lo + hi) / 2
zone = -np.ones(NC, dtype=int)
nz = 0
for i in range(NC):
    if zone[i] >= 0:
        continue
    same = [k for k in range(NC) if zone[k] < 0 and np.mean(slow[i] == slow[k]) > 0.9]
    zone[same] = nz
    nz += 1
print("zone of every cell (cells with equal rows): " + "".join(chr(ord('A') + min(z, 25)) for z in zone), flush=True)

# a fast pair, then sweep the colours and the inputs
i0, j0 = np.unravel_index(np.argmin(T + 10 * np.eye(NC)), T.shape)
x = carve(i0 * CELL, P, (3,), torch.float32)
nrm = carve(j0 * CELL, P, (3,), torch.float32)
print(f"fast pair: xyz in cell {i0}, normals in cell {j0}: {T[i0, j0]:.4f} ms.  colours swept over the cells (second half of each cell):", flush=True)
tc = [time_it(batch, {"points": x, "normals": nrm, "colors": carve(k * CELL + CELL // 2, P, (3,), torch.uint8)}) for k in range(NC)]
print("   " + " ".join(f"{t:.4f}" for t in tc), flush=True)
kc = int(np.argmin(tc))
col = carve(kc * CELL + CELL // 2, P, (3,), torch.uint8)
print(f"colours in cell {kc}.  INPUT stacks (depth, mask, normal, rgb of the {VP} views) moved over the cells (at +1 GiB inside each cell):", flush=True)
ti = []
for k in range(NC):
    off = k * CELL + (1 << 30)
    sc = {}
    for name, tail, dt in (("depth", (H, W), torch.float32), ("mask", (H, W), torch.bool), ("normal", (H, W, 3), torch.float32), ("rgb", (H, W, 3), torch.uint8)):
        t = carve(off, VP, tail, torch.uint8 if dt == torch.bool else dt)
        if dt == torch.bool:
            t = t.view(torch.bool)
        t.copy_(scene[name])
        sc[name] = t
        off += (t.numel() * t.element_size() + (1 << 21) - 1) & ~((1 << 21) - 1)
    sc["conf"] = None
    ti.append(time_it(mk_batch(sc), {"points": x, "normals": nrm, "colors": col}))
print("   " + " ".join(f"{t:.4f}" for t in ti), flush=True)
