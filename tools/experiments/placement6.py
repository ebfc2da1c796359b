#!/usr/bin/env python3
"""Round 3, second step: placement5 showed that the state lives in the OUTPUT arrays (three discrete levels 2.60 / 2.74 /
2.94 ms over 8 allocations of the set, inputs +-1 %, all regions of one allocation alike).  Is it a property of each
array on its own, or of the combination?

  E. K xyz candidates x K normal candidates (colours fixed): is time = f(xyz) + g(normal)?
  F. each xyz candidate alone (xyz-only instantiation of the kernel), and a plain fill_ / copy_ of it
  G. K colour candidates on the best pair

GPU box only."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def time_it(batch, builder, n=8, warm=2):
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


def time_op(fn, n=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


scene = bench.make_scene(cfg, ids, dev)
batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
P = batch.max_points

# interleaved allocation order (x0 n0 c0 x1 n1 c1 ...), like K fresh sets
xs, ns, cs = [], [], []
for k in range(K):
    xs.append(torch.empty((P, 3), dtype=torch.float32, device=dev))
    ns.append(torch.empty((P, 3), dtype=torch.float32, device=dev))
    cs.append(torch.empty((P, 3), dtype=torch.uint8, device=dev))

print("F. each xyz / normal candidate alone: xyz-only kernel | fill_ | copy_ from another candidate  (ms)", flush=True)
src = torch.empty((P, 3), dtype=torch.float32, device=dev)
for name, arr in (("x", xs), ("n", ns)):
    for k in range(K):
        b = dd.CloudBuilder(P, normals=False, colors=False, pixel_index=False, buffers={"points": arr[k]}, device=dev)
        t = time_it(batch, b)
        tf = time_op(lambda: arr[k].fill_(1.0))
        tc = time_op(lambda: arr[k].copy_(src))
        print(f"   {name}{k}: xyz-only {t:.3f}   fill {tf:.3f} ({arr[k].numel() * 4 / tf / 1e6:.0f} GB/s)   copy {tc:.3f}   @{arr[k].data_ptr():#x}", flush=True)
del src

print("E. xyz candidate (row) x normal candidate (column), colours c0:", flush=True)
grid = np.zeros((K, K))
for i in range(K):
    for j in range(K):
        b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers={"points": xs[i], "normals": ns[j], "colors": cs[0]}, device=dev)
        grid[i, j] = time_it(batch, b)
    print("   x%d: " % i + "  ".join(f"{t:.3f}" for t in grid[i]), flush=True)
print("E'. the same arrays with the roles swapped (normal candidates as xyz, xyz candidates as normals):", flush=True)
for i in range(K):
    row = []
    for j in range(K):
        b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers={"points": ns[i], "normals": xs[j], "colors": cs[0]}, device=dev)
        row.append(time_it(batch, b))
    print("   n%d: " % i + "  ".join(f"{t:.3f}" for t in row), flush=True)

bi, bj = np.unravel_index(np.argmin(grid), grid.shape)
print(f"G. colour candidates on the best pair (x{bi}, n{bj}):", flush=True)
for k in range(K):
    b = dd.CloudBuilder(P, normals=True, colors=True, pixel_index=False, buffers={"points": xs[bi], "normals": ns[bj], "colors": cs[k]}, device=dev)
    print(f"   c{k}: {time_it(batch, b):.3f}", flush=True)
