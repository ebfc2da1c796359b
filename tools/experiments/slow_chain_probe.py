#!/usr/bin/env python3
"""bench.py's default run reports 25 us per chained one-view call where the stand-alone tools measure 16-17 us for the same chain.
This runs bench.main() with the streaming record wrapped: behind the record's own measurement the same chain is timed again
with one thing changed at a time (same builder again; fresh side streams; a fresh plainly allocated cloud; GPU busy right before;
GPU idle for half a second before).  Everything goes to stderr; GPU box only.
   usage: tools/experiments/slow_chain_probe.py [bench.py arguments]"""
import os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench

orig = bench.streaming_record


def say(*a):
    print(*a, file=sys.stderr, flush=True)


def timed(builder, subs, device, before=None, n=5):
    ts, hs = [], []
    for _ in range(n):
        builder.reset()
        torch.cuda.synchronize(device)
        if before:
            before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record()
        for sb in subs:
            builder.append(sb)
        builder.join()
        e1.record()
        h1 = time.perf_counter()
        torch.cuda.synchronize(device)
        ts.append(e0.elapsed_time(e1)); hs.append(1e3 * (h1 - h0))
    rows = builder.check()
    return f"chain {np.median(ts):6.3f} ms (min {min(ts):6.3f}) = {1e3 * np.median(ts) / len(subs):5.2f} us/call  host {np.median(hs):6.3f} ms  rows {rows}"


def wrapped(args, dd, cfg, scene, params, E, batch, builder, device, view_base, alg_bytes):
    rec = orig(args, dd, cfg, scene, params, E, batch, builder, device, view_base, alg_bytes)
    V = batch.num_views
    subs = [batch.slice(i, i + 1) for i in range(V)]
    say(f"[probe] the record itself: {rec['per_call'].get('1', {}).get('us_per_call')} us/call; streams of the builder: {[hex(s) for s in builder._side_raw]}")
    say("[probe] same builder again           ", timed(builder, subs, device))
    builder._join_side(); builder._side = []; builder._side_ws = []; builder._side_busy = False
    say("[probe] fresh side streams           ", timed(builder, subs, device), [hex(s) for s in builder._side_raw])
    small = dd.CloudBuilder(batch.max_points, normals=builder.normal is not None, colors=builder.rgb is not None, pixel_index=False,
                            device=device, placement="first")
    say("[probe] fresh plainly allocated cloud", timed(small, subs, device), [hex(s) for s in small._side_raw])

    def busy():
        builder2 = small
        for _ in range(3):
            builder2.reset(); builder2.append(batch)
        builder2.reset()
    say("[probe] ... GPU busy right before    ", timed(small, subs, device, before=busy))
    say("[probe] ... GPU idle 0.5 s before    ", timed(small, subs, device, before=lambda: time.sleep(0.5)))
    say("[probe] first builder once more      ", timed(builder, subs, device))
    s = torch.cuda.Stream(device)
    with torch.cuda.stream(s):
        say("[probe] caller on a non-default stream", timed(small, subs, device))
    import threading
    say(f"[probe] threads alive: {[t.name for t in threading.enumerate()]}")
    return rec


if __name__ == "__main__":          # (bench.py's all-cores baseline spawns workers that import this module again: they must not run it)
    bench.streaming_record = wrapped
    sys.argv = ["bench.py"] + sys.argv[1:]
    bench.main()
