#!/usr/bin/env python3
"""One-view calls (the reference's loop, scripts/test.py:131) chained across S = 2 or 3 streams through the bare C ABI, with the small chained tile
(6144 pixels: 339 workgroups per 1080p view) and with the large one (12288: 170 -- three calls are 510 of the device's 512 slots): does a third call
in flight fill the HBM's idle time?  GPU box only.   python tools/experiments/three_in_flight.py [--views 185]"""
import argparse, ctypes as C, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
from depthdensifier_amd import _lib
lib = _lib.lib

ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=185); ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--variants", nargs="*", default=[], help="experiment builds of csrc/ddcore.hip (tools/ab_builds.build) timed beside the product library: tag:-Dflag,-Dflag")
a = ap.parse_args()
libs = [("product", lib)]
if a.variants:
    sys.path.insert(0, str(ROOT / "tools"))
    import ab_builds
    for spec in a.variants:
        tag, _, fl = spec.partition(":")
        libs.append((tag, ab_builds.build(tag, [f for f in fl.split(",") if f])))
dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = a.views
ids = np.arange(a.views)
scene = bench.make_scene(cfg, ids, dev)
H, W, V = cfg["H"], cfg["W"], a.views
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)
whole = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
b = dd.CloudBuilder(V * H * W, normals=True, colors=True, pixel_index=False, device=dev, exclusive_gpu=True)
b.append(whole); n_ref = b.check(); ref_sum = float(b.xyz[:n_ref].double().sum())
out = b._out_struct()
subs = [whole.slice(v, v + 1) for v in range(V)]
chain = torch.zeros(2, dtype=torch.int64, device=dev)
main = torch.cuda.current_stream(dev)
alg = bench.algorithmic_bytes(cfg, V, n_ref, False)
# streams that really run side by side (the runtime deals streams to a few hardware queues): keep making streams until S of them overlap pairwise
def overlapping_streams(S):
    got = [torch.cuda.Stream(dev)]
    w = torch.zeros(4, dtype=torch.int32, device=dev)
    tries = 0
    while len(got) < S and tries < 24:
        tries += 1
        s = torch.cuda.Stream(dev)
        ok = True
        for g in got:
            seen = C.c_int32()
            torch.cuda.synchronize()
            if lib.dd_streams_overlap(g.cuda_stream, s.cuda_stream, w.data_ptr(), C.byref(seen)) != 0 or not seen.value: ok = False; break
        if ok: got.append(s)
    return got if len(got) == S else None

for S in (2, 3):
    streams = overlapping_streams(S)
    if streams is None:
        print(f"{S} streams: no set of streams that run side by side found"); continue
    for tile, tname in ((0, "small tile (6144 px)"), (_lib.DD_TUNE_TILE_LARGE, "large tile (12288 px)")):
        structs = []
        for i, s in enumerate(subs):
            cs = s.c_struct(); c2 = type(cs)(); C.memmove(C.byref(c2), C.byref(cs), C.sizeof(cs))
            c2.chain, c2.chain_seq, c2.tuning = chain.data_ptr(), i, _lib.DD_TUNE_BY_INDEX | tile
            structs.append((c2, torch.empty(2, dtype=torch.int64, device=dev)))
        wss = [torch.zeros(max(s.workspace_bytes() for s in subs) + 4096, dtype=torch.uint8, device=dev) for _ in range(S)]
        for ltag, L in libs:
            ts = []
            for r in range(a.rounds + 2):
                b.cursor.zero_(); chain[0:1].copy_(b.cursor); chain[1:2].fill_(-1)      # (the second word: only read by a tree with r06_early_gate.patch)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                for st in streams: st.wait_event(e0)
                for i, (cs, offs) in enumerate(structs):
                    rc = L.dd_unproject_compact(C.byref(cs), C.byref(out), offs.data_ptr(), b.cursor.data_ptr(), wss[i % S].data_ptr(), wss[i % S].numel(), streams[i % S].cuda_stream)
                    assert rc == 0, rc
                for st in streams: main.wait_stream(st)
                e1.record(main); torch.cuda.synchronize()
                n = int(b.cursor.item())
                err = int(max(w[4:8].view(torch.int32)[0].item() for w in wss))
                if err != 0:      # a scan gave up: with three calls of 339 workgroups in flight the later ones can hold the slots the earlier one still needs
                    print(f"{S} streams, {tname:<22s} {ltag:<10s}: a call waited for ~2 s and gave up (error word {err}) -- more workgroups in flight than slots")
                    for w in wss: w.zero_()
                    ts = None
                    break
                assert n == n_ref, (n, n_ref)
                if r >= 2: ts.append(e0.elapsed_time(e1))
            if ts is None: continue
            assert abs(float(b.xyz[:n].double().sum()) - ref_sum) <= 1e-9 * abs(ref_sum)
            ts.sort()
            print(f"{S} streams, {tname:<22s} {ltag:<10s}: {ts[len(ts) // 2] * 1e3 / V:6.2f} us per one-view call (min {ts[0] * 1e3 / V:6.2f})  frac {alg / (ts[len(ts) // 2] * 1e-3) / 8e12:.3f}")
