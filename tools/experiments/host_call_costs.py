#!/usr/bin/env python3
"""What the pipeline's per-view host calls cost on the host (the loop around the kernel costs 100x the kernel):
H2D copies from page-locked memory through torch and through hipMemcpyAsync directly, allocations, small tensor ops, events, a
ctypes kernel launch -- on an idle stream and behind a stream that is busy (the maps of the previous views still being copied)."""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import depthdensifier_amd as dd
from depthdensifier_amd._lib import lib
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream(dev)

def bench(name, fn, n=200, busy=None):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    if busy: busy()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:<72s} {1e6 * (t1 - t0) / n:8.1f} us per call on the host   (+ {1e3 * (t2 - t1):6.2f} ms until the stream drained)")

big_h = torch.empty(25 << 20, dtype=torch.uint8, pin_memory=True)
big_d = torch.empty(25 << 20, dtype=torch.uint8, device=dev)
def busy():                     # ~8 ms of copies queued on the stream
    for _ in range(16): big_d.copy_(big_h, non_blocking=True)

for nbytes in (128, 24 << 10, 2 << 20, 8 << 20, 25 << 20):
    src = big_h[:nbytes]; dst = big_d[:nbytes]
    for label, b in (("idle", None), ("busy", busy)):
        bench(f"[{label}] {nbytes:>9d} B  pinned.to(device, non_blocking=True)", lambda: src.to(dev, non_blocking=True), busy=b)
        bench(f"[{label}] {nbytes:>9d} B  dst.copy_(pinned, non_blocking=True)", lambda: dst.copy_(src, non_blocking=True), busy=b)
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        dp, sp, st = dst.data_ptr(), src.data_ptr(), s.cuda_stream
        bench(f"[{label}] {nbytes:>9d} B  hipMemcpyAsync(dst, pinned, H2D) via ctypes", lambda: hip.hipMemcpyAsync(dp, sp, nbytes, 1, st), busy=b)
m = torch.ones((1080, 1920), dtype=torch.bool, device=dev)
d32 = torch.rand((1080, 1920), device=dev)
meta = torch.zeros(8, dtype=torch.int32, device=dev)
host8 = torch.empty(8, dtype=torch.int32, pin_memory=True)
for label, b in (("idle", None), ("busy", busy)):
    bench(f"[{label}] m.sum()", lambda: m.sum(), busy=b)
    bench(f"[{label}] m.sum().clamp(max=2**31-1).to(int32)", lambda: m.sum().clamp(max=2 ** 31 - 1).to(torch.int32), busy=b)
    bench(f"[{label}] depth.to(float16)", lambda: d32.to(torch.float16), busy=b)
    bench(f"[{label}] torch.empty((3, 2000), device)", lambda: torch.empty((3, 2000), dtype=torch.float32, device=dev), busy=b)
    bench(f"[{label}] torch.zeros(8, int32, device)", lambda: torch.zeros(8, dtype=torch.int32, device=dev), busy=b)
    bench(f"[{label}] host8.copy_(meta, non_blocking=True)  (D2H 32 B into pinned)", lambda: host8.copy_(meta, non_blocking=True), busy=b)
    bench(f"[{label}] torch.cuda.Event().record(stream)", lambda: torch.cuda.Event().record(s), busy=b)
    ev = torch.cuda.Event()
    bench(f"[{label}] ev.record(stream) (one event object)", lambda: ev.record(s), busy=b)
    x = torch.rand(500, device=dev); y = torch.rand(500, device=dev); kx = torch.empty_like(x); ky = torch.empty_like(y)
    bench(f"[{label}] dd_sort_knots via ctypes (one small kernel launch)", lambda: lib.dd_sort_knots(x.data_ptr(), y.data_ptr(), 500, kx.data_ptr(), ky.data_ptr(), s.cuda_stream), busy=b)
    bench(f"[{label}] torch.stack of 16 (1080,1920) float32 maps", lambda: torch.stack([d32] * 16), n=20, busy=b)
    bench(f"[{label}] upload_small(128 B)", lambda: dd.densify.upload_small(np.zeros(32, np.float32), dev), busy=b)
    bench(f"[{label}] upload_small(24 KB)", lambda: dd.densify.upload_small(np.zeros((2000, 3), np.float32), dev), busy=b)
