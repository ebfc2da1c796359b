#!/usr/bin/env python3
"""Throughput of the floater-vote kernel on a garden-like scene (GPU box)."""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, depthdensifier_amd as dd

ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=32)
ap.add_argument("--modes", default="auto,float64,float64_cull", help="comma-separated vote modes, timed in this order")
ap.add_argument("--normals", default="random", choices=("random", "smooth"),
                help="random: independent unit normals per pixel (bench.py's scene; the grazing test then differs lane by lane); "
                     "smooth: a slowly varying field facing the camera, like a monocular normal map (coherent within a wave)")
ap.add_argument("--pmc-json", default="", help="summary.json of tools/pmc_votes.sh (default: profiles/r04_pmc_votes.json, else r02): "
                "the roofline line is derived from its SQ_INSTS_VALU / SQ_WAVES / SQ_WAIT_INST_ANY")
ap.add_argument("--layout", default="ring", choices=("ring", "corridor"),
                help="ring: cameras around the scene looking inward, every view sees almost every point (bench.py's poses); "
                     "corridor: cameras 1 m apart along a line looking sideways at a surface 1-8 m away -- a point is inside "
                     "5-6 frusta whatever the number of views, like any scan larger than a table top")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = dict(bench.WORKLOADS["garden185"]); cfg["V"] = a.views
ids = np.arange(a.views)
scene = bench.make_scene(cfg, ids, dev)
H, W = cfg["H"], cfg["W"]
if a.normals == "smooth":
    ys = torch.linspace(0, 1, H, device=dev)[:, None].expand(H, W)
    xs = torch.linspace(0, 1, W, device=dev)[None, :].expand(H, W)
    for i in range(a.views):
        n = torch.stack([0.6 * torch.sin(5.0 * xs + i), 0.6 * torch.cos(4.0 * ys + 0.5 * i), -torch.ones_like(xs)], dim=-1)
        scene["normal"][i] = torch.nn.functional.normalize(n, dim=-1)
params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (a.views, 1))
E = bench.ring_poses(ids, a.views)
if a.layout == "corridor":
    E = np.zeros((a.views, 3, 4))
    for v in range(a.views):
        E[v, :, :3] = np.eye(3)                     # looking along +z, x to the right
        E[v, :, 3] = -np.array([1.0 * v, 0.0, 0.0])
cloud = dd.unproject_views(scene["depth"], params, E, mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"])
K = dd.intrinsics_matrix(params)
torch.cuda.synchronize()
pairs = len(cloud) * a.views
sums, rate = {}, {}
for mode in tuple(a.modes.split(",")):
    st = {}
    dd.floater_votes(cloud.points, cloud.normals, scene["depth"], K, E, mask=scene["mask"], mode=mode)   # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    votes = dd.floater_votes(cloud.points, cloud.normals, scene["depth"], K, E, mask=scene["mask"], mode=mode, stats=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sums[mode] = (int(votes.long().sum()), int((votes.long() * (torch.arange(len(votes), device=votes.device) % 1000003)).sum()))
    rate[mode] = pairs / dt / 1e9
    print(f"[{mode:13s}] points {len(cloud)/1e6:.1f} M x views {a.views} = {pairs/1e9:.2f} G pairs in {dt*1e3:.1f} ms -> {pairs/dt/1e9:.1f} Gpairs/s; "
          f"votes>=5: {(votes >= 5).float().mean().item()*100:.2f} %  max {int(votes.max())}; checksum {sums[mode]}"
          + (f"; sampled cells surviving the cull {st['cull_sample_survived'] / max(st['cull_sample_cells'], 1) * 100:.1f} % -> {'culling' if st['culled'] else 'plain'} kernel" if mode == "auto" else ""))
assert len(set(sums.values())) == 1, f"votes differ between modes: {sums}"
print("votes identical in every mode")
# the NumPy baseline of this stage is timed by tests/time_filter_oracle.py (the oracle is test infrastructure)


# ---- roofline of the vote kernel: bound by VALU ISSUE (float64 arithmetic on one lane per point; no MFMA, almost no memory) ----
# bound = 64 pairs per wave and view / (VALU wave-instructions per wave and view) x (wave-instructions the chip issues per second);
# the chip issues one wave-instruction per 4 cycles and SIMD nominally (256 CUs x 4 SIMDs x 2.4 GHz / 4 = 614 G/s) and one
# float64 FMA per 5.35 cycles measured (profiles/r02_ubench_fma_rates.txt: 459 G/s).  The counters come from a separate
# profiling run of this script (tools/pmc_votes.sh): they are per-kernel averages, not measured here.
import json
cands = [Path(a.pmc_json)] if a.pmc_json else [ROOT / "profiles" / "r04_pmc_votes.json", ROOT / "profiles" / "r02_pmc_votes.json"]
pmc = next((p for p in cands if p.is_file()), None)
if pmc is not None:
    doc = json.loads(pmc.read_text())
    ctr = doc.get("counters", doc)
    views_pmc = int(doc.get("views", 48))
    for kern, mode in (("floater_votes_kernel2", "float64"), ("floater_votes_kernel_cull", "float64_cull")):
        c = ctr.get(kern)
        if not c or mode not in rate or not c.get("SQ_WAVES"):
            continue
        valu = c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / views_pmc
        nominal, measured = 64.0 / valu * 614.4, 64.0 / valu * 614.4 * 4.0 / 5.35
        print(json.dumps({"roofline": {"kernel": kern, "bound": "valu_issue", "valu_per_wave_and_view": round(valu, 1),
                                       "valu_issue_bound_gpairs": round(nominal, 1), "valu_issue_bound_gpairs_at_measured_f64_rate": round(measured, 1),
                                       "achieved_gpairs": round(rate[mode], 1), "frac": round(rate[mode] / nominal, 3),
                                       "frac_of_measured_f64_rate": round(rate[mode] / measured, 3),
                                       "waiting_fraction": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None,
                                       "counters_from": str(pmc.relative_to(ROOT)), "counters_views": views_pmc, "this_run_views": a.views}}))
