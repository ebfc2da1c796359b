#!/usr/bin/env python3
"""The streaming regime of the hot path: the views of a scan appended to the cloud k at a time, call after call, as the
reference's loop does (scripts/test.py:131, 203-240: one view per iteration) and as pipeline.py does (8 views per launch).

    python tools/bench_streaming.py [--views 185] [--per-call 1,8] [--variants auto,t1w3,...] [--graph]

For every k and every variant: the whole chain of ceil(V / k) dd_unproject_compact calls between two events (inputs resident in
HBM), three ways -- CloudBuilder.append (the product's host path), the bare C-ABI call with structs built beforehand (what
the host path costs on top), and the same chain replayed from a captured HIP graph.  A variant is a DDViewBatch.tuning word:
tN = bits 18-19 (1 small tile, 3 large tile), wN = bits 20-21 (1 / 2 / 3 = 16 / 32 / 64 polling lanes), s1 = bit 22 (tiles by
workgroup index, no tickets), c1 = bit 26 (the decoupled look-back instead of the scan service), pN = plain tuning N.  --libs tag:-Dflag,-Dflag ...: experiment builds of csrc/ddcore.hip
(tools/ab_builds.py builds them), timed through the bare C-ABI call beside the product library.
"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
import lab_bits  # noqa: E402  (variant words: a product tuning + the experiment switches of include/ddcore_lab.h)
import depthdensifier_amd as dd  # noqa: E402
from depthdensifier_amd._lib import lib, check  # noqa: E402


def tuning_of(tag: str) -> int:
    if tag == "auto":
        return 0
    t, i = 0, 0
    while i < len(tag):
        c = tag[i]
        j = i + 1
        while j < len(tag) and tag[j].isdigit():
            j += 1
        n = int(tag[i + 1:j])
        t |= (n << 18) if c == "t" else (n << 20) if c == "w" else (n << 22) if c == "s" else (n << 26) if c == "c" else n
        i = j
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden185")
    ap.add_argument("--views", type=int, default=185)
    ap.add_argument("--per-call", default="1,8")
    ap.add_argument("--variants", default="auto")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--libs", nargs="*", default=[], help="experiment builds: tag:-Dflag,-Dflag")
    ap.add_argument("--only", default="", help="run only this mode (builder.append | C ABI | HIP graph | a --libs tag): for profiler runs")
    ap.add_argument("--build-only", action="store_true")
    a = ap.parse_args()
    xlibs = []
    if a.libs:
        import ab_builds
        ab_builds.BUILD_ONLY = a.build_only
        for spec in a.libs:
            tag, _, fl = spec.partition(":")
            xlibs.append((tag, ab_builds.build(tag, [f for f in fl.split(",") if f])))
    if a.build_only:
        return
    dev = torch.device("cuda", 0)
    cfg = dict(bench.WORKLOADS[a.workload]); cfg["V"] = a.views; cfg["mask_kind"] = "blob"
    V, H, W = a.views, cfg["H"], cfg["W"]
    ids = np.arange(V)
    scene = bench.make_scene(cfg, ids, dev)
    params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
    batch = dd.ViewBatch(scene["depth"], params, bench.ring_poses(ids, V), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"],
                         conf=scene["conf"], conf_threshold=cfg.get("conf"), device=dev)
    builder = dd.CloudBuilder(batch.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=False, device=dev)
    builder.append(batch)
    ref = builder.finish()
    n = len(ref)
    ref_sum = float(ref.points.double().sum())
    ref_offs = ref.view_offsets.clone()
    alg = bench.algorithmic_bytes(cfg, V, n, False)
    print(f"# {a.workload} x {V} views, {n} points, algorithmic {alg / 1e9:.3f} GB; one batch of {V}: see bench.py")
    stream = torch.cuda.current_stream(dev)
    # the floor: the same number of dependent launches of a kernel that does nothing
    tiny = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(V):
            tiny.add_(1)
        e1.record(); torch.cuda.synchronize()
    print(f"# floor: {V} dependent launches of a one-element kernel: {1e3 * e0.elapsed_time(e1) / V:.2f} us per launch")
    for k in [int(x) for x in a.per_call.split(",")]:
        subs = [batch.slice(lo, min(lo + k, V)) for lo in range(0, V, k)]
        for tag in a.variants.split(","):
            tun = tuning_of(tag)
            for s in subs:
                lab_bits.set_on(s, tun)
            res = {}
            # (i) the product's host path
            def chain_builder():
                builder.reset()
                for s in subs:
                    builder.append(s)
                builder.join()
            # (ii) the bare C-ABI call
            structs = [(s.c_struct(), torch.empty(s.num_views + 1, dtype=torch.int64, device=dev)) for s in subs]
            out = builder._out_struct()
            ws = builder._workspace(max(s.workspace_bytes() for s in subs))
            sp = stream.cuda_stream

            from depthdensifier_amd._lib import lab_switches
            t_lab = lab_bits.split(tun)[1]

            def chain_abi():
                builder.cursor.zero_()
                with lab_switches(t_lab):
                  for cs, offs in structs:
                    rc = lib.dd_unproject_compact(C.byref(cs), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), ws.data_ptr(), ws.numel(), sp)
                    assert rc == 0
            modes = [("builder.append", chain_builder), ("C ABI", chain_abi)]
            # (iii) the bare C-ABI calls chained across the builder's two side streams (DDViewBatch.chain; what builder.append does for
            #       small calls with exclusive_gpu, without its Python): structs with the chain word and the sequence numbers set beforehand
            if builder.exclusive_gpu and tun == 0 and builder._chained_ok(subs[0], 1 << 22):
                structs2 = []
                for i, (cs, offs) in enumerate(structs):
                    c2 = type(cs)()
                    C.memmove(C.byref(c2), C.byref(cs), C.sizeof(cs))
                    c2.chain, c2.chain_seq, c2.tuning = builder._chain_ptr, i, cs.tuning | (1 << 22)
                    structs2.append((c2, offs))
                builder._side_workspaces(max(s.workspace_bytes() for s in subs))

                def chain_abi2():
                    builder.cursor.zero_()
                    builder._chain.copy_(builder.cursor, non_blocking=True)
                    for sd in builder._side_raw:
                        assert lib.dd_stream_fork(builder._fork_raw, sp, sd) == 0
                    for i, (cs, offs) in enumerate(structs2):
                        w = builder._side_ws[i & 1]
                        rc = lib.dd_unproject_compact(C.byref(cs), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), w.data_ptr(), w.numel(), builder._side_raw[i & 1])
                        assert rc == 0
                    for sd in builder._side:
                        stream.wait_stream(sd)
                modes.append(("C ABI chained", chain_abi2))
            for xtag, xlib in xlibs:
                def chain_x(xlib=xlib):
                    builder.cursor.zero_()
                    for cs, offs in structs:
                        rc = xlib.dd_unproject_compact(C.byref(cs), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), ws.data_ptr(), ws.numel(), sp)
                        assert rc == 0
                modes.append((xtag, chain_x))
            graph = None
            if a.graph:
                side = torch.cuda.Stream(dev)
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    sps = side.cuda_stream
                    graph.capture_begin()
                    for cs, offs in structs:
                        rc = lib.dd_unproject_compact(C.byref(cs), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), ws.data_ptr(), ws.numel(), sps)
                        assert rc == 0
                    graph.capture_end()
                torch.cuda.synchronize()

                def chain_graph():
                    builder.cursor.zero_()
                    graph.replay()
                modes.append(("HIP graph", chain_graph))
            for name, fn in modes:
                if a.only and name != a.only:
                    continue
                fn(); torch.cuda.synchronize()
                ts, hs = [], []
                for r in range(a.rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    h0 = time.perf_counter()
                    e0.record(); fn(); e1.record()
                    h1 = time.perf_counter()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1)); hs.append((h1 - h0) * 1e3)
                tot = int(builder.cursor.item())
                ok = tot == n and abs(float(builder.xyz[:n].double().sum()) - ref_sum) == 0.0
                if name != "builder.append":
                    last = structs[-1][1]
                    ok = ok and int(last[-1].item()) == n
                err = int(ws[:8].view(torch.int32)[1].item()) | (sum(int(w[:8].view(torch.int32)[1].item()) for w in builder._side_ws) if name == "C ABI chained" else 0)
                med = float(np.median(ts))
                res[name] = med
                print(f"k={k:3d} {tag:10s} {name:15s} chain {med:8.3f} ms (min {min(ts):7.3f})  per call {1e3 * med / len(subs):7.2f} us  frac {alg / med / 1e6 / 8000:5.3f}  "
                      f"host enqueue {float(np.median(hs)):7.3f} ms  calls {len(subs)}  rows {'ok' if ok else 'WRONG'} err {err}", flush=True)
            for s in subs:
                lab_bits.set_on(s, 0)


if __name__ == "__main__":
    main()
