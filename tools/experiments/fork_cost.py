#!/usr/bin/env python3
"""What does the fork in front of every chained call (dd_stream_fork: event record on the caller's stream + wait on the side
stream) cost a chain of one-view appends, and does dd_stream_fork's "the caller's stream is idle" shortcut trigger?
   variants: as shipped | no fork after a chain's first call | the caller's stream is a non-default stream | HIP graph replay
GPU box only."""
import ctypes as C
import os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
os.environ.setdefault("DD_EXCLUSIVE_GPU", "1")
import bench, depthdensifier_amd as dd
from depthdensifier_amd import densify

dev = torch.device("cuda", 0)
V = 185
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = V; cfg["mask_kind"] = "blob"
H, W = cfg["H"], cfg["W"]
ids = np.arange(V)
scene = bench.make_scene(cfg, ids, dev)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
batch = dd.ViewBatch(scene["depth"], params, bench.ring_poses(ids, V), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
subs = [batch.slice(i, i + 1) for i in range(V)]
builder = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=False, device=dev, placement="first")
real = densify.lib


class Lib:
    """The library with dd_stream_fork timed, or skipped after a chain's first call."""
    def __init__(self, skip):
        self.skip, self.t, self.n = skip, [], 0
    def __getattr__(self, name):
        return getattr(real, name)
    def dd_stream_fork(self, ev, a, b):
        self.n += 1
        if self.skip and builder._side_busy:
            return 0
        t0 = time.perf_counter()
        rc = real.dd_stream_fork(ev, a, b)
        self.t.append(time.perf_counter() - t0)
        return rc


def chain():
    builder.reset()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for s in subs:
        builder.append(s)
    builder.join()
    e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), host * 1e3


def report(tag, proxy):
    densify.lib = proxy
    ts = [chain() for _ in range(6)][1:]
    densify.lib = real
    rows = builder.check()
    ft = np.array(proxy.t[-(V if not proxy.skip else 1):]) * 1e6
    best = min(t for t, _ in ts)
    print(f"{tag:46s} chain {np.median([t for t, _ in ts]):6.3f} ms (min {best:6.3f} = {1e3 * best / V:5.2f} us/call)  host {np.median([h for _, h in ts]):6.3f} ms"
          f"  dd_stream_fork: median {np.median(ft):5.2f} us, under 1.5 us: {int((ft < 1.5).sum())}/{len(ft)}  rows {rows}")


report("as shipped (default stream)", Lib(False))
report("no fork after the chain's first call", Lib(True))
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    report("as shipped (caller on a non-default stream)", Lib(False))
    report("no fork after the first call (non-default)", Lib(True))
report("as shipped (default stream) again", Lib(False))
# (more than two side streams: measured once with 3 and 4 -- 16.4 us per call, and a 10.7 s chain when three successors held every
#  workgroup slot while they waited for a call that had not been dispatched: the 384-workgroup bound of a waiting call is for ONE successor)
import gc
g = dd.capture_chain(builder, subs)
report("eager again, a captured graph exists", Lib(False))
ts = []
for _ in range(6):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"{'HIP graph replay of the chained chain':46s} chain {np.median(ts[1:]):6.3f} ms (min {min(ts[1:]):6.3f} = {1e3 * min(ts[1:]) / V:5.2f} us/call)")
report("eager again, the graph has been replayed", Lib(False))
del g; gc.collect(); torch.cuda.synchronize()
report("eager again, the graph is gone", Lib(False))
# a graph of something else, captured on a stream of its own (what tools/bench_streaming.py --graph does before it times the appends)
side = torch.cuda.Stream(dev)
tiny = torch.zeros(8, device=dev)
g2 = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.stream(side):
    g2.capture_begin()
    for _ in range(50):
        tiny.add_(1)
    g2.capture_end()
torch.cuda.synchronize()
report("eager, a graph of 50 tiny kernels exists", Lib(False))
g2.replay(); torch.cuda.synchronize()
report("eager, that graph has been replayed", Lib(False))
builder._join_side(); builder._side = []; builder._side_ws = []; builder._side_busy = False
report("eager, fresh side streams", Lib(False))
del g2; gc.collect(); torch.cuda.synchronize()
report("eager, that graph is gone", Lib(False))
