#!/usr/bin/env python3
"""Raw MoGe depth -> points: dd_refine_apply + densify (two kernels, refined map written and re-read) against the fused
DD_REFINE kernel (LUT + 3x3 median + validity + unprojection in one pass; the refined map is still written once for the
filter cache).  GPU box only.   python tools/bench_fused_refine.py [--views 64]"""
import argparse, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd
from depthdensifier_amd.depth_refiner import DepthRefiner

ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=64)
ap.add_argument("--tuning", type=lambda x: int(x, 0), default=0, help="variant word of the fused batch (tests/lab_bits.py: a product tuning + experiment switches of include/ddcore_lab.h)")
ap.add_argument("--variants", nargs="*", default=[], help="experiment builds of csrc/ddcore.hip timed on the fused batch, interleaved with the product library: tag:-Dflag,-Dflag (tools/ab_builds.build); tags that start with x_ may compute something else")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--refined-placements", action="store_true", help="time the fused launch with the refined map (the filter's cache) allocated plainly and through the zone arena, rotated from class 0 / 1 / 2")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = a.views
ids = np.arange(a.views)
scene = bench.make_scene(cfg, ids, dev)
H, W = cfg["H"], cfg["W"]
V = a.views
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
E = bench.ring_poses(ids, V)
r = DepthRefiner(use_fp16=False)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(500, device=dev, generator=g) * 7 + 1
kx, ky = r._sorted_knots(x, 1.1 * x + 0.05 * torch.rand(500, device=dev, generator=g))
refined = torch.empty((V, H, W), dtype=torch.float32, device=dev)

def unfused():
    for v in range(V):
        refined[v] = r._apply_curve_hip(scene["depth"][v], scene["mask"][v], kx, ky)      # (sorts the sorted knots again: one tiny launch)
    batch = dd.ViewBatch(refined, params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev)
    b.reset(); b.append(batch)

fused_batch = dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev,
                           refine=[(kx, ky, False)] * V, refined_out=True)
sys.path.insert(0, str(ROOT / "tests"))
import lab_bits
lab_bits.set_on(fused_batch, a.tuning)
def fused():
    b.reset(); b.append(fused_batch)

b = dd.CloudBuilder(fused_batch.max_points, normals=True, colors=True, pixel_index=False, device=dev)
res = {}
for name, fn in (("unfused", unfused), ("fused", fused), ("unfused", unfused), ("fused", fused)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    n = b.check()
    res[name] = e0.elapsed_time(e1) / 5
    print(f"{name:8s} {res[name]:8.3f} ms per {V} views = {res[name] / V * 1e3:7.1f} us per 1080p view, {n} points")
pix = V * H * W
rho = n / pix
print(f"bytes per pixel by the model: unfused {5 + 4 + 5 + rho * 42:.1f} (refine 5 r + 4 w, densify 5 r + rho 42), fused {5 + 4 + rho * 42:.1f} "
      f"(5 r + 4 w for the filter cache + rho 42); rho = {rho:.3f}")

if a.variants:
    import ctypes as C
    sys.path.insert(0, str(ROOT / "tools"))
    import ab_builds
    from depthdensifier_amd import _lib
    libs = [("product", _lib.lib)]
    for spec in a.variants:
        tag, _, fl = spec.partition(":")
        libs.append((tag, ab_builds.build(tag, [f for f in fl.split(",") if f])))
    cb, out = fused_batch.c_struct(), b._out_struct()
    cb.tuning |= _lib.DD_TUNE_BY_INDEX
    ws = torch.zeros(4 * fused_batch.workspace_bytes() + 4096, dtype=torch.uint8, device=dev)
    offs = torch.empty(V + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    times = {tag: [] for tag, _ in libs}
    ref = None
    for r in range(a.rounds + 1):
        for tag, lib in libs:
            b.cursor.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.dd_unproject_compact(C.byref(cb), C.byref(out), offs.data_ptr(), b.cursor.data_ptr(), ws.data_ptr(), ws.numel(), stream)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0, (tag, rc)
            if r == 0:
                chk = (int(b.cursor.item()), float(b.xyz[: int(b.cursor.item())].double().sum()))
                ref = ref or chk
                assert chk == ref or tag.startswith("x_"), f"variant {tag} computes something else: {chk} vs {ref}"
            else:
                times[tag].append(e0.elapsed_time(e1))
    for tag, _ in libs:
        t = sorted(times[tag])
        print(f"{tag:<28s} median {t[len(t) // 2] / V * 1e3:6.2f} us per view   min {t[0] / V * 1e3:6.2f}   ({V} views, kernel alone through the C ABI)")

if a.refined_placements:
    from depthdensifier_amd import placement as PL
    print("placement of the cloud:", None if b.placement is None else (b.placement.mode, b.placement.classes))
    outs = {"plain": torch.empty((V, H, W), dtype=torch.float32, device=dev)}
    for ph in (0, 1, 2):
        t, rep = PL.place_arrays({"refined": ((V, H, W), torch.float32, PL.rotated(ph))}, dev, "probed")
        outs[f"rotated({ph})"] = t["refined"]
        print(f"rotated({ph}):", rep.mode, rep.classes)
    batches = {k: dd.ViewBatch(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=dev,
                               refine=[(kx, ky, False)] * V, refined_out=o) for k, o in outs.items()}
    times = {k: [] for k in batches}
    for r in range(a.rounds + 1):
        for k, fb in batches.items():
            b.reset(); b.append(fb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b.reset(); e0.record(); b.append(fb); e1.record(); torch.cuda.synchronize()
            if r: times[k].append(e0.elapsed_time(e1))
    for k, t in times.items():
        t = sorted(t)
        print(f"refined map {k:<12s} median {t[len(t) // 2] / V * 1e3:6.2f} us per view   min {t[0] / V * 1e3:6.2f}")
