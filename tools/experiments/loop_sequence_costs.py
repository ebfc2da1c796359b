#!/usr/bin/env python3
"""The pipeline's per-view sequence of stream operations, step by step on the host: where does a call block?  (round 6: 0.6-1.2 ms
per view sat in whichever runtime call came first behind the 41 MB of uploads.)  Variants: everything on one stream / uploads on a
copy stream of their own with an event for the compute stream."""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import depthdensifier_amd as dd
from depthdensifier_amd._lib import lib
dev = torch.device("cuda", 0)
H, W = 1080, 1920
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
host = {"depth": torch.empty((H, W), dtype=torch.float32, pin_memory=True), "mask": torch.empty((H, W), dtype=torch.uint8, pin_memory=True),
        "normal": torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True), "rgb": torch.empty((H, W, 3), dtype=torch.uint8, pin_memory=True)}
for t in host.values(): t.zero_()
host["depth"].fill_(2.0); host["mask"].fill_(1)
pts_h = torch.rand((300, 3), pin_memory=True)
meta_h = torch.empty(8, dtype=torch.int32, pin_memory=True)
E = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0); K = (C.c_float * 6)(1500, 0, 960, 0, 1500, 540)

def run(label, copy_stream):
    s = torch.cuda.current_stream(dev)
    cs = torch.cuda.Stream(dev) if copy_stream else s
    ev_up = torch.cuda.Event(); ev_up.record(s); ev_ready = torch.cuda.Event(); ev_ready.record(s)
    laps = {}
    keep = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(N):
        t = time.perf_counter()
        def lap(k):
            nonlocal t
            now = time.perf_counter(); laps[k] = laps.get(k, 0.0) + now - t; t = now
        dst = {k: torch.empty(v.shape, dtype=v.dtype, device=dev) for k, v in host.items()}
        lap("4 x torch.empty")
        n = 4
        src = (C.c_void_p * n)(*[host[k].data_ptr() for k in host]); d = (C.c_void_p * n)(*[dst[k].data_ptr() for k in host]); sz = (C.c_int64 * n)(*[host[k].numel() * host[k].element_size() for k in host])
        lib.dd_upload_async(n, src, d, sz, ev_up.cuda_event, cs.cuda_stream)
        lap("dd_upload_async (4 copies + event)")
        if copy_stream:
            s.wait_event(ev_up)
            lap("stream.wait_event")
        d16 = dst["depth"].to(torch.float16)
        lap("depth.to(float16)")
        work = torch.empty(6 * 300 + 8, dtype=torch.float32, device=dev)
        lap("torch.empty(work)")
        lib.dd_refine_fit_async(pts_h.data_ptr(), 300, E, K, d16.data_ptr(), 1, H, W, 10, 1, 3.0, 1, dst["mask"].data_ptr(), work.data_ptr(), work.data_ptr() + 24 * 300, meta_h.data_ptr(), ev_ready.cuda_event, s.cuda_stream)
        lap("dd_refine_fit_async")
        keep.append((dst, d16, work))
        if len(keep) > 20:
            if copy_stream:
                for tt in keep[0][0].values(): tt.record_stream(cs)
            keep.pop(0)
        lap("drop a view's tensors")
    host_s = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    print(f"== {label}: host {1e3 * host_s / N:.3f} ms per view, until drained {1e3 * wall / N:.3f} ms per view")
    for k, v in laps.items():
        print(f"   {k:<40s} {1e6 * v / N:8.1f} us")

run("one stream", False)
run("uploads on a copy stream", True)
run("one stream", False)
run("uploads on a copy stream", True)
