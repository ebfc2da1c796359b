#!/usr/bin/env python3
"""bench.py -- Mpixels/s unprojected+fused on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload garden185|scene2000|roofline12mp]

One "step" = one pass of the hot path (cull + unproject + cam->world + stable compaction +
fuse) over one batch of synthetic views already resident in HBM.  At N=1 the workload is
BASELINE.json configs[1] ("garden": 185 views, 1920x1080, depth f32 + normal f32x3 + mask u8
+ rgb u8x3; synthetic stand-in).  For N>1 the driver launches one rank per GPU
(torch.distributed.run) and the default workload is BASELINE configs[2], the north-star curve: the
2000-view 1080p scene split over the N ranks in contiguous shards (strong scaling; views are
independent), `value` = the sharded fuse (fused kernel + RCCL all-gather of the per-view
counts: every rank holds the global view offsets of the distributed cloud); `--workload garden185` at
N>1 is the weak-scaling variant (185 views per rank).
Every line, at every N, also carries a "strong2000" sub-record: BASELINE configs[2], the 2000-view
1080p scene split over the N ranks (strong scaling), timed three ways -- sharded fuse (count
exchange only), replicated fuse in place (all-gatherv of xyz + normals + colours, 27 B/point) and
replicated fuse of the compact 16-byte xyz+rgba record -- with the bytes each rank received and
the per-link rate against xGMI's 153 GB/s.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

WORKLOADS = {
    # name: (views per GPU, H, W, depth dtype, mask, normal, rgb, valid fraction target)
    "garden185": dict(V=185, H=1080, W=1920, depth="float32", mask=True, normal=True, rgb=True, rho=0.85,
                      note="BASELINE configs[1] stand-in: 185 views ~1080p, blob mask"),
    "scene2000": dict(V=2000, H=1080, W=1920, depth="float32", mask=True, normal=True, rgb=True, rho=0.8,
                      note="BASELINE configs[2]: 2000-view synthetic scene (strong scaling: V is the total)"),
    "mip360conf": dict(V=232, H=1080, W=1920, depth="float32", mask=True, normal=True, rgb=True, rho=0.85, conf=0.5,
                       note="BASELINE configs[3], one GPU's share as ONE batch: 232 = 1626 / 7 views of the Mip-NeRF 360 set "
                            "(synthetic stand-ins), mask AND conf > 0.5; use --views to change"),
    "mip360conf_smooth": dict(V=232, H=1080, W=1920, depth="float32", mask=True, normal=True, rgb=True, rho=0.85, conf=0.5, conf_kind="smooth",
                              note="the same share with a spatially COHERENT confidence map (blobs: what a network's confidence looks like) -- the "
                                   "per-pixel noise of mip360conf keeps a survivor on every 128-byte line of the normal / colour maps, so its gathers "
                                   "fetch the whole maps (traffic 1.26x the algorithmic bytes: line granularity, not kernel quality)"),
    "mip360x7": dict(V=1626, H=1080, W=1920, depth="float32", mask=True, normal=True, rgb=True, rho=0.85, conf=0.5,
                     scenes=[("bicycle", 194), ("bonsai", 292), ("counter", 240), ("garden", 185), ("kitchen", 279),
                             ("room", 311), ("stump", 125)],
                     note="BASELINE configs[3]: the 7 Mip-NeRF 360 scenes back to back (1626 views, synthetic stand-ins at 1080p), "
                          "mask AND conf > 0.5; whole scenes dealt to the ranks most expensive first (no data-path collective), "
                          "every scene gets a fresh cloud and a host read of its point count (V is the total: strong scaling)"),
    "roofline12mp": dict(V=500, H=3024, W=4032, depth="float16", mask=False, normal=False, rgb=False, rho=1.0,
                         note="BASELINE configs[4]: 500 views 12 MP, f16 depth in / f32 xyz out, dense"),
}


# ------------------------------------------------------------------------------ synthetic scene

def ring_poses(view_ids: np.ndarray, total: int, radius: float = 4.0) -> np.ndarray:
    """cam_from_world (n,3,4): cameras on a ring of radius 4 looking at the origin."""
    E = np.zeros((len(view_ids), 3, 4))
    for i, v in enumerate(view_ids):
        a = 2 * np.pi * float(v) / total
        c = np.array([radius * np.cos(a), 0.3 * np.sin(3 * a), radius * np.sin(a)])
        z = -c / np.linalg.norm(c)
        x = np.cross([0.0, 1.0, 0.0], z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        E[i, :, :3] = R
        E[i, :, 3] = -R @ c
    return E


def make_scene(cfg: dict, view_ids: np.ndarray, device) -> dict:
    """Seeded per-view synthetic maps generated on the device (seed = 1000 + global view id, so
    sharding does not change the data): smooth depth field 1-8 m, blob mask with the target
    valid fraction, unit normals, random colours."""
    V, H, W = len(view_ids), cfg["H"], cfg["W"]
    ddt = getattr(torch, cfg["depth"])
    depth = torch.empty((V, H, W), dtype=ddt, device=device)
    mask = torch.empty((V, H, W), dtype=torch.bool, device=device) if cfg["mask"] else None
    normal = torch.empty((V, H, W, 3), dtype=torch.float32, device=device) if cfg["normal"] else None
    rgb = torch.empty((V, H, W, 3), dtype=torch.uint8, device=device) if cfg["rgb"] else None
    conf = torch.empty((V, H, W), dtype=torch.float32, device=device) if cfg.get("conf") else None
    ys = torch.linspace(0, 1, H, device=device)[:, None]
    xs = torch.linspace(0, 1, W, device=device)[None, :]
    for i, vid in enumerate(view_ids):
        g = torch.Generator(device=device).manual_seed(1000 + int(vid))
        ph = torch.rand(6, generator=g, device=device) * 6.283
        f = 1.0 + 3.0 * torch.rand(6, generator=g, device=device)
        field = (torch.sin(f[0] * 6.283 * xs + ph[0]) * torch.sin(f[1] * 6.283 * ys + ph[1])
                 + 0.5 * torch.sin(f[2] * 6.283 * (xs + ys) + ph[2]) + 0.25 * torch.sin(f[3] * 12.566 * xs + ph[3]))
        depth[i] = (4.5 + 2.0 * field).clamp(1.0, 8.0).to(ddt)
        if mask is not None:
            if cfg.get("mask_kind") == "bernoulli":
                mask[i] = torch.rand((H, W), generator=g, device=device) < cfg["rho"]
            else:
                coarse = torch.rand((1, 1, 18, 32), generator=g, device=device)
                blob = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=False)[0, 0]
                thr = torch.quantile(blob.flatten()[:: max(1, (H * W) // 200000)], 1.0 - cfg["rho"])
                mask[i] = blob > thr
        if normal is not None:
            n = torch.randn((H, W, 3), generator=g, device=device)
            normal[i] = torch.nn.functional.normalize(n, dim=-1)
        if rgb is not None:
            rgb[i] = torch.randint(0, 256, (H, W, 3), generator=g, device=device, dtype=torch.uint8)
        if conf is not None:
            if cfg.get("conf_kind") == "smooth":
                # a spatially coherent confidence (what a network's confidence map looks like): half of every view above 0.5, in blobs
                coarse = torch.rand((1, 1, 18, 32), generator=g, device=device)
                blob = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=False)[0, 0]
                med = torch.quantile(blob.flatten()[:: max(1, (H * W) // 200000)], 0.5)
                conf[i] = ((blob - med) * 2.0 + 0.5).clamp(0.0, 1.0)
            else:
                conf[i].uniform_(0.0, 1.0, generator=g)     # independent per pixel: every 128-byte line of the attribute maps keeps a survivor
    return dict(depth=depth, mask=mask, normal=normal, rgb=rgb, conf=conf)


# ------------------------------------------------------------------------------ byte model

def algorithmic_bytes(cfg: dict, V: int, n_valid: int, pixel_index: bool, reads_only: bool = False) -> int:
    """read P*(b_depth+b_mask) + N*(12[normal]+3[rgb]) + 64 B params per view;
    write N*(12 + 12[normal] + 3[rgb] + 4[pixel_index]) + 8 B offset per view.  Attributes are
    charged only for surviving pixels; nothing is credited for re-reads."""
    P = cfg["H"] * cfg["W"]
    b_depth = 2 if cfg["depth"] == "float16" else 4
    per_px = b_depth + (1 if cfg["mask"] else 0) + (4 if cfg.get("conf") else 0)
    per_pt_r = (12 if cfg["normal"] else 0) + (3 if cfg["rgb"] else 0)
    per_pt_w = 12 + (12 if cfg["normal"] else 0) + (3 if cfg["rgb"] else 0) + (4 if pixel_index else 0)
    if reads_only:
        return V * P * per_px + n_valid * per_pt_r + V * 64
    return V * P * per_px + n_valid * (per_pt_r + per_pt_w) + V * (64 + 8)


# ------------------------------------------------------------------------------ CPU baseline

def cpu_baseline(cfg: dict, scene: dict, params: np.ndarray, E: np.ndarray, budget_s: float) -> dict:
    """The oracle (NumPy restatement of scripts/test.py:194-266, kind "port") timed single-process
    -- exactly how the reference runs -- on the first views of the same workload, for about
    `budget_s` seconds of CPU work."""
    from oracle import densify_oracle as orc          # reported baseline only, never the product path
    from threadpoolctl import threadpool_limits

    P = cfg["H"] * cfg["W"]
    n_views = 0
    t_total = 0.0
    pts = 0
    max_views = scene["depth"].shape[0]
    one_thread = threadpool_limits(limits=1)            # "cores": 1 means one thread: BLAS would otherwise fan the small
    while t_total < budget_s and n_views < max_views:   # matmuls out over every host core it can see
        i = n_views
        d = scene["depth"][i].cpu().numpy()
        m = None if scene["mask"] is None else scene["mask"][i].cpu().numpy()
        n = None if scene["normal"] is None else scene["normal"][i].cpu().numpy()
        c = None if scene["rgb"] is None else scene["rgb"][i].cpu().numpy()
        t0 = time.perf_counter()
        cf = None if scene.get("conf") is None else scene["conf"][i].cpu().numpy()
        out = orc.fuse_views([orc.densify_view_script(d, params[i], E[i], mask=m, normal=n, rgb=c, conf=cf,
                                                      conf_threshold=cfg.get("conf"))])
        t_total += time.perf_counter() - t0
        pts += len(out.points)
        n_views += 1
    # the reference's own op sequence (np.mgrid + int64 fancy-index gathers, scripts/test.py:205-233), same views, same thread
    lit = None
    if cfg["mask"] and cfg["normal"] and cfg["rgb"] and not cfg.get("conf"):
        lit_t, lit_n = 0.0, 0
        while lit_t < budget_s / 3.0 and lit_n < min(n_views, max_views):
            i = lit_n
            d, m = scene["depth"][i].cpu().numpy(), scene["mask"][i].cpu().numpy()
            n, c = scene["normal"][i].cpu().numpy(), scene["rgb"][i].cpu().numpy()
            t0 = time.perf_counter()
            orc.fuse_views([orc.densify_view_script_literal(d, params[i], E[i], m, n, c)])
            lit_t += time.perf_counter() - t0
            lit_n += 1
        lit = {"value": round(lit_n * P / lit_t / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
               "sample": f"first {lit_n} views, {lit_t:.1f} s",
               "what": "the densify block restated with the reference's literal op sequence: np.mgrid index grids, boolean-mask "
                       "gathers of the int64 grids, fancy-index gathers, np.stack + three float64 temporaries (scripts/test.py:205-233); "
                       "same outputs as the tuned port above, which uses np.nonzero on a strided view"}
    one_thread.restore_original_limits()
    return {
        "value": round(n_views * P / t_total / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"first {n_views} of {max_views} views of the same workload, {t_total:.1f} s single-process, single-thread NumPy "
                  f"{np.__version__} (the reference is one Python process; oracle/densify_oracle.py)",
        "mpoints_per_s": round(pts / t_total / 1e6, 3),
        "reference_formulation": lit,
        "note": "SURVEY.md / BASELINE.md quote 1.4 Mpixels/s for the same block: that probe ran in the survey container (one thread of a "
                "2.1 GHz Xeon, a slower host than the GPU box's) -- the figure here is measured on this box, now",
    }


def verify_views(dd, cfg: dict, scene: dict, params: np.ndarray, E: np.ndarray, views, offsets: torch.Tensor, cloud: dict, device,
                 view_index_base: int = 0) -> dict:
    """The timed cloud against the oracle (scripts/test.py:194-266 restated in NumPy) on the given local views: per-view counts
    bit-exact; rows of the timed cloud equal to an untimed pass that also emits pixel_index (same kernel, one more output);
    pixel_index order, colours and pass-through normals bit-exact against the oracle; xyz within 1e-4 of the scene scale
    (SURVEY.md 8d).  Raises AssertionError on any mismatch."""
    from oracle import densify_oracle as orc          # checker only, never the product path
    offs = offsets.cpu().numpy()
    worst, n_pts = 0.0, 0
    for i in views:
        d = scene["depth"][i].cpu().numpy()
        m = None if scene["mask"] is None else scene["mask"][i].cpu().numpy()
        nm = None if scene["normal"] is None else scene["normal"][i].cpu().numpy()
        c = None if scene["rgb"] is None else scene["rgb"][i].cpu().numpy()
        cf = None if scene.get("conf") is None else scene["conf"][i].cpu().numpy()
        ref = orc.fuse_views([orc.densify_view_script(d, params[i], E[i], mask=m, normal=nm, rgb=c, conf=cf, conf_threshold=cfg.get("conf"))])
        lo, hi = int(offs[i]), int(offs[i + 1])
        assert hi - lo == len(ref.points), f"view {i}: {hi - lo} points, oracle {len(ref.points)}"
        # an untimed pass of this one view with pixel_index on: the rows must be the timed cloud's rows
        vb = dd.ViewBatch(scene["depth"][i:i + 1], params[i:i + 1], E[i:i + 1], mask=None if scene["mask"] is None else scene["mask"][i:i + 1],
                          normal=None if scene["normal"] is None else scene["normal"][i:i + 1], rgb=None if scene["rgb"] is None else scene["rgb"][i:i + 1],
                          conf=None if scene.get("conf") is None else scene["conf"][i:i + 1], conf_threshold=cfg.get("conf"), device=device)
        b = dd.CloudBuilder(vb.max_points, normals=nm is not None, colors=c is not None, pixel_index=True, device=device, placement="first")
        b.append(vb)
        one = b.finish()
        assert np.array_equal(one.pixel_index.cpu().numpy(), ref.pixel_index), f"view {i}: pixel_index order differs from the oracle"
        for name, timed, again in (("points", cloud["points"], one.points), ("normals", cloud.get("normals"), one.normals), ("colors", cloud.get("colors"), one.colors)):
            if timed is not None:
                assert torch.equal(timed[lo:hi], again), f"view {i}: {name} of the timed cloud differ from the untimed pass"
        if c is not None:
            assert np.array_equal(one.colors.cpu().numpy(), ref.colors), f"view {i}: colours differ from the oracle"
        if nm is not None:
            assert np.array_equal(one.normals.cpu().numpy(), ref.normals), f"view {i}: normals differ from the oracle"
        if len(ref.points):
            # per point, |delta|_inf / max(|p_ref|_inf, scene radius); radius = camera centre + deepest finite depth
            centre = -np.asarray(E[i], dtype=np.float64)[:3, :3].T @ np.asarray(E[i], dtype=np.float64)[:3, 3]
            dfin = d[np.isfinite(d)].astype(np.float64)
            radius = float(np.linalg.norm(centre) + (dfin.max() if dfin.size else 0.0))
            delta = np.abs(one.points.cpu().numpy().astype(np.float64) - ref.points).max(axis=1)
            denom = np.maximum(np.abs(ref.points).max(axis=1), radius)
            fin = np.isfinite(ref.points).all(axis=1)
            err = float((delta[fin] / denom[fin]).max()) if fin.any() else 0.0
            assert err <= 1e-4, f"view {i}: xyz relative error {err:.3e} > 1e-4"
            worst = max(worst, err)
        n_pts += hi - lo
    return {"views": len(list(views)), "points": n_pts, "xyz_max_rel": float(f"{worst:.3e}"), "tolerance": 1e-4,
            "bit_exact": "per-view counts, pixel_index order, colours, pass-through normals; timed rows == untimed re-run"}


def remask_bernoulli(scene: dict, cfg: dict, view_ids: np.ndarray, device) -> None:
    """The masks of `scene` overwritten IN PLACE with the independent per-pixel cull of SURVEY.md 8d config 3 (rho = cfg["rho"]),
    exactly the masks make_scene draws with mask_kind = "bernoulli" (same generator, same order of draws)."""
    H, W = cfg["H"], cfg["W"]
    for i, vid in enumerate(view_ids):
        g = torch.Generator(device=device).manual_seed(1000 + int(vid))
        torch.rand(6, generator=g, device=device); torch.rand(6, generator=g, device=device)      # (phases and frequencies of the depth field)
        scene["mask"][i] = torch.rand((H, W), generator=g, device=device) < cfg["rho"]


def strong_scaling_record(args, dd, D, dist, use_dist, rank, world, device, fence, reuse=None) -> dict:
    """BASELINE configs[2] on the N ranks of this job: `--strong-views` (2000) synthetic 1080p views, rank r holds the
    contiguous shard shard_views(V, N, r), inputs resident in HBM.  Timed three ways (max over ranks, mean of the timed
    passes): "sharded" = one fused kernel over the shard + the all-gather of per-view counts (cloud stays distributed,
    globally indexed); "gathered" = distributed.fuse_replicated, every point written once at its final global row and
    the chunks exchanged in place while the next chunk's kernel runs (xyz + normals + colours, 27 B/point);
    "gathered_compact" = the same with one 16-byte xyz+rgba record per point; and "bernoulli" = the sharded fuse once more with
    the blob masks replaced by SURVEY.md 8d's per-pixel Bernoulli cull (the worst case for the compaction).
    `reuse`: the scene the main workload already holds in HBM (scene2000 at its default size), instead of a second copy."""
    cfg = dict(WORKLOADS["scene2000"])
    cfg["mask_kind"] = args.mask_kind          # blob (default) or the per-pixel Bernoulli cull of SURVEY.md 8d config 3
    V_total = args.strong_views
    lo, hi = D.shard_views(V_total, world, rank)
    H, W = cfg["H"], cfg["W"]
    ids = np.arange(lo, hi)
    t_gen = time.perf_counter()
    if reuse is not None:
        scene, batch, params = reuse["scene"], reuse["batch"], reuse["params"]
    else:
        scene = make_scene(cfg, ids, device)
        params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (len(ids), 1))
        batch = dd.ViewBatch(scene["depth"], params, ring_poses(ids, V_total), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"],
                             view_index_base=int(lo), device=device)
    torch.cuda.synchronize(device)
    t_gen = time.perf_counter() - t_gen
    rec = {"views_total": V_total, "views_per_gpu": len(ids), "height": H, "width": W, "scaling": "strong", "chunks": args.chunks,
           "gather_dst": args.gather_dst, "allgatherv": os.environ.get("DD_ALLGATHERV", "p2p"),
           "mask_kind": args.mask_kind,
           "scene_generation_s": round(t_gen, 2) if reuse is None else "the main workload's scene, already resident"}

    def timed(fn, passes):
        fn()                                            # warm-up (allocations, RCCL channels)
        fence()
        t0 = time.perf_counter()
        for _ in range(passes):
            out = fn()
        fence()
        dt = (time.perf_counter() - t0) / passes
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    # -- sharded fuse: the cloud stays distributed; max capacity, no sizing pass
    builder = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=False, device=device, placement=args.placement)
    rec["placement"] = None if builder.placement is None else builder.placement.mode
    pixels = V_total * H * W

    def sharded_leg(kind_cfg):
        def sharded():
            builder.reset()
            offs = builder.append(batch)
            builder.join()                  # (a tiny batch may have run on the builder's side streams: its offsets are read right here)
            counts = offs[1:] - offs[:-1]
            return D.offsets_from_counts(D.exchange_counts(counts, V_total) if use_dist else counts)

        dt, goffs = timed(sharded, args.strong_steps)
        builder.check()
        out = {}
        if len(ids) and not args.no_verify:      # the last view of this rank's shard of the TIMED cloud against the oracle
            try:
                local_offs = builder._offsets[-1]
                v = verify_views(dd, kind_cfg, scene, params, ring_poses(ids, V_total), [len(ids) - 1], local_offs,
                                 {"points": builder.xyz, "normals": builder.normal, "colors": builder.rgb}, device)
                ok = 1
            except AssertionError as e:
                v, ok = {"error": str(e)[:300]}, 0
            if use_dist:
                flag = torch.tensor([ok], dtype=torch.int64, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            out["verified"] = dict(v, what="last view of every rank's shard of the timed sharded cloud vs the oracle", all_ranks_ok=bool(ok))
        n_total = int(goffs[-1].item())
        n_own = int((goffs[hi] - goffs[lo]).item())
        alg = algorithmic_bytes(kind_cfg, len(ids), n_own, False)
        out["points_total"] = n_total
        out["sharded"] = {"ms": round(dt * 1e3, 3), "mpixels_per_s": round(pixels / dt / 1e6, 1), "mpoints_per_s": round(n_total / dt / 1e6, 1),
                          "whole_step_frac": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4), "valid_fraction": round(n_own / max(1, len(ids) * H * W), 4),
                          "what": "fused kernel over the shard + all-gather of per-view counts; cloud left distributed, globally indexed "
                                  "(whole_step_frac: this rank's algorithmic bytes over the wall time of the step, host included)"}
        return out, n_total, n_own

    first, n_total, n_own = sharded_leg(cfg)
    rec.update(first)
    if args.mask_kind == "blob" and not args.no_bernoulli and len(ids):
        # SURVEY.md 8d config 3 names a per-pixel Bernoulli variant beside the blob masks: the same scene, masks redrawn in place
        bcfg = dict(cfg, mask_kind="bernoulli")
        keep_masks = scene["mask"].clone()
        remask_bernoulli(scene, bcfg, ids, device)
        try:
            bern, _, _ = sharded_leg(bcfg)
            rec["bernoulli"] = dict(bern, mask_kind="bernoulli", what="the sharded leg with an independent per-pixel cull (rho = 0.8) instead of blob masks")
        except Exception as e:      # noqa: BLE001
            rec["bernoulli"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        scene["mask"].copy_(keep_masks)
        del keep_masks
    del builder
    torch.cuda.empty_cache()

    for key, record, rec_bytes in (("gathered", "rows", 27), ("gathered_compact", "xyz_rgba", 16)):
        need = n_total * rec_bytes
        free = torch.cuda.mem_get_info(device)[0]
        if use_dist:
            fmin = torch.tensor([free], dtype=torch.int64, device=device)
            dist.all_reduce(fmin, op=dist.ReduceOp.MIN)
            free = int(fmin.item())
        if need > 0.9 * free:
            rec[key] = {"skipped": f"replicated cloud of {need / 1e9:.1f} GB does not fit the free HBM ({free / 1e9:.1f} GB)"}
            continue
        placed_report = None
        if record == "rows" and os.environ.get("DD_FUSE_PLACEMENT") == "probed":
            # the global arrays RCCL receives into, placed by the zone arena (classes of HBM) instead of as the allocator returns them:
            # what distributed.fuse_replicated's comment says should work and no one-GPU box could show (tools/run_multi_gpu.sh runs it)
            from depthdensifier_amd import placement as _plf
            p_xyz, p_nrm, p_rgb, placed_report = _plf.place_outputs(n_total, colors=True, normals=True, device=device, mode="probed")
            bufs = {"points": p_xyz, "normals": p_nrm, "colors": p_rgb}
        elif record == "rows":
            bufs = {"points": torch.empty((n_total, 3), dtype=torch.float32, device=device),
                    "normals": torch.empty((n_total, 3), dtype=torch.float32, device=device),
                    "colors": torch.empty((n_total, 3), dtype=torch.uint8, device=device)}
        else:
            bufs = {"packed": torch.empty((n_total, 4), dtype=torch.float32, device=device)}

        def replicated():
            if use_dist:
                return D.fuse_replicated(batch, V_total, record=record, chunks=args.chunks, buffers=bufs,
                                         dst=None if args.gather_dst == "all" else int(args.gather_dst))
            counts = dd.count_valid(batch)              # N = 1: the same steps minus the wire
            b = dd.CloudBuilder(n_total, points=record == "rows", normals=record == "rows", colors=record == "rows", pixel_index=False,
                                packed=record != "rows", buffers=bufs, device=device)
            for clo, chi in D._chunk_bounds(0, len(ids), args.chunks):
                b.append(batch.slice(clo, chi))
            return b.check(), counts

        try:
            dt, _ = timed(replicated, args.strong_steps)
        except Exception as e:      # noqa: BLE001  (one leg failing must not cost the others)
            rec[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            del bufs
            torch.cuda.empty_cache()
            continue
        recv = (n_total - n_own) * rec_bytes
        rec[key] = {"ms": round(dt * 1e3, 3), "mpixels_per_s": round(pixels / dt / 1e6, 1), "mpoints_per_s": round(n_total / dt / 1e6, 1),
                    "record_bytes": rec_bytes, "cloud_bytes": need, "bytes_received_per_rank": recv,
                    "ingress_GBps_per_rank": round(recv / dt / 1e9, 1),
                    "GBps_per_link": round(recv / max(world - 1, 1) / dt / 1e9, 1) if world > 1 else None,
                    "xgmi_link_peak_GBps": 153.0,
                    "what": "count pass + count all-gather + fused kernel writing at final global rows, per-chunk grouped send/recv "
                            "in place overlapped with the next chunk's kernel" if world > 1 else
                            "N = 1: count pass + fused kernel per chunk (no wire)"}
        if placed_report is not None:
            rec[key]["buffers_placement"] = placed_report.mode
        del bufs
        torch.cuda.empty_cache()
    return rec


def streaming_record(args, dd, cfg, scene, params, E, batch, builder, device, view_base: int, alg_bytes: int) -> dict:
    """The path as the reference calls it: ONE view per loop iteration (scripts/test.py:131, 203-240) -- and eight, as pipeline.py
    stacks them -- appended call after call to the same cloud, inputs resident in HBM.  Events bracket the whole chain of
    ceil(V / k) CloudBuilder.append calls (each ONE kernel launch since ABI 11); frac = the workload's algorithmic bytes over that
    time.  The same chain replayed from a captured HIP graph is timed beside it (the chain is bound by the GPU, not by the host's
    enqueueing: `host_enqueue_ms`), and the chained cloud is compared with the oracle like the one-batch cloud."""
    V = batch.num_views
    rec = {"views": V, "what": "chains of CloudBuilder.append calls of k views each over the whole workload, events around the chain; "
                              "frac = algorithmic bytes of the workload / chain time / peak", "per_call": {}}
    # the floor of any chain: dependent launches of a kernel that does nothing
    tiny = torch.zeros(1, dtype=torch.int64, device=device)
    fl = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device); e0.record()
        for _ in range(100):
            tiny.add_(1)
        e1.record(); torch.cuda.synchronize(device)
        fl.append(1e3 * e0.elapsed_time(e1) / 100)
    rec["dependent_launch_floor_us"] = round(min(fl), 2)
    ok_all = True
    for k in args.streaming:
        subs = [batch.slice(lo, min(lo + k, V)) for lo in range(0, V, k)]

        def chain():
            builder.reset()
            for sb in subs:
                builder.append(sb)
            builder.join()                  # (small appends run on the builder's side streams: the event behind the chain must see them)

        chain(); builder.check()
        ts, hs = [], []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            h0 = time.perf_counter()
            e0.record(); chain(); e1.record()
            h1 = time.perf_counter()
            torch.cuda.synchronize(device)
            ts.append(e0.elapsed_time(e1)); hs.append((h1 - h0) * 1e3)
        total = builder.check()
        med = float(np.median(ts))
        shared = None
        if builder.exclusive_gpu:                    # the same chain with tickets (the library's default on a GPU that may be shared)
            builder.exclusive_gpu = False
            chain(); builder.check()
            t2 = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(device)
                e0.record(); chain(); e1.record()
                torch.cuda.synchronize(device)
                t2.append(e0.elapsed_time(e1))
            builder.check()
            shared = float(np.median(t2))
            builder.exclusive_gpu = True
            chain(); total = builder.check()
        item = {"calls": len(subs), "chain_ms": round(med, 4), "chain_ms_min": round(min(ts), 4), "us_per_call": round(1e3 * med / len(subs), 2),
                "frac_shared_gpu_mode": None if shared is None else round(alg_bytes / (shared * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "frac": round(alg_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "mpixels_per_s": round(V * cfg["H"] * cfg["W"] / (med * 1e-3) / 1e6, 1),
                "host_enqueue_ms": round(float(np.median(hs)), 3), "redone": {"healed": int(builder.healed), "dense_misses": int(builder.dense_misses)}}
        if not args.no_verify:
            try:
                offs = torch.cat([builder._offsets[0]] + [o[1:] for o in builder._offsets[1:]])
                # (1 and 8 views per call against the oracle on four views; the chains in between on two: each costs seconds of host time)
                views = sorted(set(list(range(min(V, 2))) + [V // 2, V - 1])) if k in (1, 8) else sorted({0, V - 1})
                v = verify_views(dd, cfg, scene, params, E, views, offs, {"points": builder.xyz, "normals": builder.normal, "colors": builder.rgb},
                                 device, view_base)
                v["rows_total"] = total
                item["verified"] = v
            except AssertionError as e:
                item["verified"] = {"error": str(e)[:300]}
                ok_all = False
        # the same chain from a captured graph (one graph launch instead of len(subs) kernel launches from Python)
        try:
            graph = dd.capture_chain(builder, subs)
            graph.replay(); torch.cuda.synchronize(device)
            tg = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(device)
                e0.record(); graph.replay(); e1.record()
                torch.cuda.synchronize(device)
                tg.append(e0.elapsed_time(e1))
            item["hip_graph_chain_ms"] = round(float(np.median(tg)), 4)
            item["hip_graph_rows_equal"] = bool(int(builder.cursor.item()) == total)
            del graph
        except Exception as e:      # noqa: BLE001
            item["hip_graph_chain_ms"] = None
            item["hip_graph_error"] = f"{type(e).__name__}: {e}"[:200]
        rec["per_call"][str(k)] = item
    rec["side_stream_probes"] = int(getattr(builder, "side_stream_probes", 0))      # pairs of streams tried until two ran side by side
    rec["all_ok"] = ok_all
    return rec


def fused_refine_record(dd, cfg, scene, params, E, builder, device) -> dict:
    """The call scripts/test.py's loop amounts to (:179-240: refine the raw depth with the fitted curve, then densify) as pipeline.py
    issues it: ONE kernel (DD_REFINE: LUT + 3x3 median + validity + unprojection; the refined map written once for the filter cache).
    The whole workload in one batch with a synthetic 500-knot curve per view; the first views are compared, bit for bit, with
    dd_refine_apply followed by the plain call.  This stage is bound by the vector ALU (profiles/r05_fused_refine.txt), so `frac` is
    reported as what it is -- the share of the HBM peak its own bytes amount to -- next to the plain kernel's."""
    from depthdensifier_amd.depth_refiner import DepthRefiner
    V, H, W = scene["depth"].shape
    r = DepthRefiner(use_fp16=False)
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.rand(500, device=device, generator=g) * 7 + 1
    kx, ky = r._sorted_knots(x, 1.1 * x + 0.05 * torch.rand(500, device=device, generator=g))
    ids = np.arange(V)
    kw = dict(mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"], device=device)
    fused = dd.ViewBatch(scene["depth"], params, E, refine=[(kx, ky, False)] * V, refined_out=True, **kw)
    builder.reset(); builder.append(fused); n = builder.check()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        # the launch's duration, not the host's way to it: an untimed launch keeps the GPU busy while the timed one is enqueued behind
        # it (round 5 recorded e0 on an idle stream: the ~0.2 ms the host needs to reach the launch were counted as kernel time)
        builder.reset(); builder.append(fused)
        builder.reset(); e0.record(); builder.append(fused); e1.record()
        torch.cuda.synchronize(device)
        ts.append(e0.elapsed_time(e1))
    n = builder.check()
    med = float(np.median(ts))
    # bytes by the model of section 8d with the refined map's write added: P (depth + mask + 4) + N (rows + gathers)
    per_point = 12 + (12 if cfg["normal"] else 0) * 2 + (3 if cfg["rgb"] else 0) * 2
    alg = V * H * W * (scene["depth"].element_size() + 1 + 4) + n * per_point
    k = min(V, 3)
    small = dd.ViewBatch(scene["depth"][:k], params[:k], E[:k], refine=[(kx, ky, False)] * k, refined_out=True,
                         **{a: (b[:k] if torch.is_tensor(b) else b) for a, b in kw.items()})
    b2 = dd.CloudBuilder(small.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=False, device=device, placement="first")
    b2.append(small); got = b2.finish()
    refined = torch.stack([r._apply_curve_hip(scene["depth"][v], scene["mask"][v], kx, ky) for v in range(k)])
    plain = dd.ViewBatch(refined, params[:k], E[:k], **{a: (b[:k] if torch.is_tensor(b) else b) for a, b in kw.items()})
    b3 = dd.CloudBuilder(plain.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=False, device=device, placement="first")
    b3.append(plain); want = b3.finish()
    same = (len(got) == len(want) and torch.equal(got.points, want.points) and torch.equal(small.refined.view(torch.int32), refined.view(torch.int32))
            and (got.normals is None or torch.equal(got.normals, want.normals)) and (got.colors is None or torch.equal(got.colors, want.colors)))
    traffic = None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        trec = json.loads(tfile.read_text()).get("garden185:fused_refine")
        if trec:
            traffic = int(trec["hbm_bytes_per_launch"] * V / trec["views"])
    return {"traffic": traffic, "traffic_frac": None if traffic is None else round(traffic / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "traffic_source": "profiles/traffic.json[garden185:fused_refine]: PMC FETCH_SIZE / WRITE_SIZE of this kernel, separate profiling runs -- NOT measured in this run",
            "what": "raw depth -> points in ONE kernel (DD_REFINE: transfer curve + 3x3 median fused into the densify kernel; refined map written for the filter "
                    "cache), the whole workload in one batch, 500 knots", "ms": round(med, 4), "us_per_view": round(1e3 * med / V, 2), "points": int(n),
            "algorithmic_bytes": int(alg), "frac": round(alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "bound": "hbm + valu (profiles/r06_fused_refine.txt: the look-ups cost 1.7 us of 17 per view, the windows 0.9, the refined map's write 1.4)",
            "equals_refine_apply_then_plain": bool(same), "views_compared": k}


def pipeline_record(device, views: int = 48) -> dict:
    """The path as the drop-in drives it (scripts/run_batch.py:57-91 -> scripts/test.py:131-251 -> depthdensifier_amd/pipeline.py): the
    image loop of one synthetic 1080p scan with every cache warm (maps and decoded image as .npy files), at full density -- host
    milliseconds per view, how much of that is waiting (for the GPU: 41 MB of maps per view come over PCIe; for the prefetcher), the
    loop's wall clock and the densify kernel's share of it.  The scan is written to a temporary directory first (not timed); the
    loop runs three times, the last one is reported.  profiles/r06_bench_pipeline.txt has the 185-view runs at densities 1 and 32."""
    import contextlib, io, tempfile
    from PIL import Image as PILImage
    from depthdensifier_amd import pipeline as P
    from depthdensifier_amd.colmap_io import Camera, Image, Reconstruction
    H, W, V = 1080, 1920, int(views)
    g = torch.Generator(device=device).manual_seed(11)
    with tempfile.TemporaryDirectory() as tmp:
        scan = Path(tmp) / "scan"
        (scan / "images").mkdir(parents=True); (scan / "sparse" / "0").mkdir(parents=True); (scan / "cache").mkdir()
        rec = Reconstruction()
        fx = 0.9 * W
        rec.cameras[1] = Camera(1, 1, W, H, np.array([fx, fx, W / 2.0, H / 2.0]))
        us = torch.arange(W, device=device, dtype=torch.float32)[None, :]; vs = torch.arange(H, device=device, dtype=torch.float32)[:, None]
        flat = PILImage.fromarray(np.full((H, W, 3), 128, np.uint8))
        ids, xyz, next_id = [], [], 1
        rng = np.random.default_rng(11)
        for v in range(V):
            a = -0.5 + v / max(V - 1, 1)
            c = np.array([3.0 * np.sin(a), -2.0, -3.0 * np.cos(a)])                       # a camera above the plane y = 0, looking at the origin
            zax = -c / np.linalg.norm(c); xax = np.cross([0, 1.0, 0], zax); xax /= np.linalg.norm(xax); yax = np.cross(zax, xax)
            R = np.stack([xax, yax, zax]); t = -R @ c
            Rt = torch.tensor(R, device=device, dtype=torch.float32)
            ry = ((us - W / 2.0) / fx) * Rt[0, 1] + ((vs - H / 2.0) / fx) * Rt[1, 1] + Rt[2, 1]     # y of R^T r
            tt = -float(c[1]) / ry
            depth_true = torch.where((tt > 0) & torch.isfinite(tt), tt, torch.zeros_like(tt))
            mask = (depth_true > 0) & (depth_true < 12) & (torch.rand((H, W), device=device, generator=g) < 0.97)
            mono = 0.5 * depth_true.clamp(min=1e-3) ** 1.1
            stem = f"img_{v:03d}"
            np.save(scan / "cache" / f"{stem}_depth.npy", mono.cpu().numpy())
            np.save(scan / "cache" / f"{stem}_mask.npy", mask.cpu().numpy())
            np.save(scan / "cache" / f"{stem}_normal.npy", np.tile((np.array([0.0, -1.0, 0.0]) @ R.T).astype(np.float32), (H, W, 1)))
            np.save(scan / "cache" / f"{stem}_rgb.npy", torch.randint(0, 256, (H, W, 3), device=device, generator=g, dtype=torch.uint8).cpu().numpy())
            flat.save(scan / "images" / f"{stem}.png")
            pu = rng.uniform(12, W - 13, 300); pv = rng.uniform(12, H - 13, 300)
            d = depth_true[torch.as_tensor(pv.astype(int), device=device), torch.as_tensor(pu.astype(int), device=device)].cpu().numpy().astype(np.float64)
            ok = (d > 0) & (d < 12)
            pu, pv, d = pu[ok], pv[ok], d[ok]
            cam = np.stack([(np.floor(pu) - W / 2.0) / fx * d, (np.floor(pv) - H / 2.0) / fx * d, d], -1)
            world = (cam - t) @ R
            pid = np.arange(next_id, next_id + len(world)); next_id += len(world)
            ids.append(pid); xyz.append(world)
            w4 = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            q = np.array([w4, (R[2, 1] - R[1, 2]) / (4 * w4), (R[0, 2] - R[2, 0]) / (4 * w4), (R[1, 0] - R[0, 1]) / (4 * w4)])
            rec.images[v + 1] = Image(v + 1, q, t, 1, f"{stem}.png", np.stack([np.floor(pu), np.floor(pv)], -1), pid.astype(np.int64))
        rec.point_ids = np.concatenate(ids).astype(np.uint64); rec.point_xyz = np.concatenate(xyz)
        rec.point_rgb = np.full((len(rec.point_ids), 3), 200, np.uint8); rec.point_error = np.zeros(len(rec.point_ids))
        rec._tracks = [np.zeros((0, 2), np.int32)] * len(rec.point_ids)
        rec.write_binary(scan / "sparse" / "0")
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=Path(tmp) / "out")
        cfg.moge.cache_dir = scan / "cache"
        cfg.processing.downsample_density = 1
        cfg.refiner.verbose = 0
        rep = None
        for _ in range(3):
            with contextlib.redirect_stdout(io.StringIO()):
                rep = P.main(cfg, _loop_only=True)
            torch.cuda.synchronize(device)
    tm, det, n = rep["timings"], rep["loop_detail"], rep["views"]
    host = sum(tm[k] for k in ("image_decode", "depth_source", "refine", "densify"))
    waits = det.get("finish_refine_of_which_waiting_for_the_gpu", 0.0) + det.get("wait_for_io_thread", 0.0)
    wall = rep["loop_seconds"]
    kernel_us = 17.0                       # the fused refine launch per 1080p view (this line's garden185.fused_refine.us_per_view)
    return {"what": "pipeline.main's image loop on a synthetic scan, caches warm (maps + decoded image as .npy), downsample_density 1, "
                    f"{cfg.processing.io_threads} native prefetch threads, {cfg.processing.views_per_launch} views per launch",
            "views": n, "dense_points": int(rep["dense_points"]),
            "host_ms_per_view": round(1e3 * host / n, 3), "host_ms_per_view_without_waits": round(1e3 * (host - waits) / n, 3),
            "waiting_ms_per_view": round(1e3 * waits / n, 3), "loop_wall_ms_per_view": round(1e3 * wall / n, 3),
            "upload_bytes_per_view": int(H * W * (4 + 1 + 12 + 3)), "pcie_floor_ms_per_view": round(H * W * 20 / 54e9 * 1e3, 3),
            "densify_share_of_loop": round(kernel_us * 1e-6 * n / wall, 4),
            "loop_detail_us_per_view": {k: round(1e6 * v / n, 1) for k, v in det.items() if 1e6 * v / n >= 5},
            "r03": {"host_ms_per_view": 2.04}}


def _cpu_worker(job):
    """One process of the all-cores courtesy baseline: the oracle over this worker's views (arrays staged as .npy in
    shared memory by the parent; only the oracle calls are timed)."""
    import numpy as _np
    from oracle import densify_oracle as orc          # reported baseline only, never the product path
    stage, views, params, E, fields, conf_thr = job
    load = lambda k, i: _np.load(f"{stage}/{k}_{i}.npy", mmap_mode="r") if k in fields else None
    t, pts = 0.0, 0
    for i in views:
        d, m, n, c, cf = (None if a is None else _np.array(a) for a in (load(k, i) for k in ("depth", "mask", "normal", "rgb", "conf")))
        t0 = time.perf_counter()
        out = orc.fuse_views([orc.densify_view_script(d, params[i], E[i], mask=m, normal=n, rgb=c, conf=cf, conf_threshold=conf_thr)])
        t += time.perf_counter() - t0
        pts += len(out.points)
    return t, pts


def cpu_baseline_all_cores(cfg: dict, scene: dict, params: np.ndarray, E: np.ndarray, budget_s: float, single_mpix_s: float,
                           procs: int) -> dict:
    """Courtesy upper bound (SURVEY.md 8d ii): the same oracle, one process per host core over views -- NOT something
    the reference does (it is a single Python process).  Workers are spawned (no fork of a GPU process), the sample
    views are staged in /dev/shm, throughput = sample pixels / slowest worker's oracle time."""
    import multiprocessing as mp
    import shutil
    import tempfile
    P = cfg["H"] * cfg["W"]
    max_views = scene["depth"].shape[0]
    per_proc = max(1, int(single_mpix_s * 1e6 * budget_s / P))            # views one core gets through in the budget
    n_views = min(max_views, per_proc * procs)
    procs = min(procs, n_views)
    stage = tempfile.mkdtemp(prefix="dd_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        fields = [k for k in ("depth", "mask", "normal", "rgb", "conf") if scene.get(k) is not None]
        for i in range(n_views):
            for k in fields:
                np.save(f"{stage}/{k}_{i}.npy", scene[k][i].cpu().numpy())
        jobs = [(stage, list(range(r, n_views, procs)), params, E, fields, cfg.get("conf")) for r in range(procs)]
        saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        os.environ.update({k: "1" for k in saved})          # one math thread per worker (inherited by the spawned children)
        try:
            with mp.get_context("spawn").Pool(procs) as pool:
                res = pool.map(_cpu_worker, jobs)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    finally:
        shutil.rmtree(stage, ignore_errors=True)
    wall = max(t for t, _ in res)
    return {"value": round(n_views * P / wall / 1e6, 3), "unit": "Mpixels/s", "cores": procs, "kind": "port",
            "sample": f"first {n_views} views over {procs} spawned processes (one per host core available to this job), "
                      f"slowest worker {wall:.1f} s; a courtesy upper bound, the reference itself is one process"}


def deal_scenes(sizes, world: int):
    """Owner rank of every scene: largest first to the least loaded rank (what batch.assign_scans does with scan
    folders); a pure function of the sizes, identical on every rank."""
    load, owner = [0] * world, [0] * len(sizes)
    for k in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k], load[r] = r, load[r] + sizes[k]
    return owner


# ------------------------------------------------------------------------------ main

_GUARD_SRC = r"""
import signal, sys
for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(s, signal.SIG_IGN)
data = sys.stdin.buffer.read()                      # returns when rank 0 closes the pipe -- or dies
if data and not data.endswith(b"\0DONE"):
    sys.stdout.buffer.write(data.split(b"\0")[0]); sys.stdout.buffer.flush()
"""


def _claim_stdout():
    """stdout is reserved for the ONE result line: whatever a library prints to fd 1 (RCCL's version banner
    on the GPU boxes, for one) is sent to stderr; the returned file is the real stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _spawn_line_guard(real_out):
    """A GPU-less child that holds rank 0's finished main result while the (separately timed) all-gatherv
    leg runs and prints it only if rank 0 dies without printing -- so stdout carries exactly one JSON
    line whatever RCCL does.  Own session + ignored SIGTERM: it outlives torchrun's clean-up of a failed job."""
    import subprocess
    return subprocess.Popen([sys.executable, "-c", _GUARD_SRC], stdin=subprocess.PIPE, stdout=real_out, start_new_session=True)


def _release_line_guard(guard) -> None:
    if guard is None:
        return
    try:
        guard.stdin.write(b"\0DONE")
        guard.stdin.close()
        guard.wait(timeout=10)
    except Exception:      # noqa: BLE001
        pass


def launch_ranks(n: int, argv, grace_s: float = 15.0) -> int:
    """Start ``n`` ranks of ``python argv...`` on this node, one fresh process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, exactly what ``python -m torch.distributed.run --nnodes=1 --nproc-per-node n`` hands its workers), wait for
    them and return 0 only if every one of them did.  The caller must not have initialised the GPU: the children are started with
    ``subprocess`` (fork + exec of a process that holds no GPU state), never by replacing a running program.  When a rank fails the
    others get ``grace_s`` seconds to notice (a collective that errors out), then SIGTERM, then SIGKILL -- by PID, nothing else.
    stdout is shared with the children: rank 0 prints the ONE result line, every other rank sends its stdout to stderr itself."""
    import signal
    import socket
    import subprocess
    env = dict(os.environ)
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")                 # (as torchrun does: N ranks must not each take every core)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    sys.stdout.flush()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")))

    def forward(signum, _frame):                            # the driver's time-out reaches the ranks too
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    old = {s: signal.signal(s, forward) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        failed_at = None
        while any(p.poll() is None for p in procs):
            time.sleep(0.05)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad and failed_at is None:
                failed_at = time.monotonic()
                print(f"[bench] rank {procs.index(bad[0])} exited with {bad[0].returncode}; the other ranks get {grace_s:.0f} s", file=sys.stderr, flush=True)
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                t_kill = time.monotonic() + 10.0
                while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                    time.sleep(0.05)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                for p in procs:
                    p.wait()
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    codes = [p.returncode for p in procs]
    if any(codes):
        print(f"[bench] rank exit codes {codes}", file=sys.stderr, flush=True)
        return next((c for c in codes if 0 < c < 256), 1)      # (a rank's own exit code; ranks ended by a signal have negative ones)
    return 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="the workload `value` is measured on.  Default: scene2000 (BASELINE configs[2], the scene the metric is quoted on; strong "
                         "scaling: the 2000 views are split over the N ranks) with, at N = 1, the other single-GPU configurations as sub-records of "
                         "the same line: garden185 (configs[1]) with its streaming chains, roofline12mp (configs[4]), mip360conf (configs[3])")
    ap.add_argument("--sub", default="auto", help="sub-records of the default line: auto (all, at N = 1 with the default workload), none, or a comma list")
    ap.add_argument("--sub-steps", type=int, default=10, help="timed steps of a sub-record (its warm-up: 3)")
    ap.add_argument("--streaming", default="1,2,4,8,16", help="views per call of the streaming chains timed on garden185 (16 = the pipeline's launch size; '' = skip)")
    ap.add_argument("--no-bernoulli", action="store_true", help="skip the per-pixel Bernoulli leg of the strong2000 record")
    ap.add_argument("--placement", default="probed", choices=("probed", "first"),
                    help="probed: the cloud's points / normals / colours built from different classes of HBM address ranges (the arena of "
                         "depthdensifier_amd/placement.py); first: as the allocator returns them (rounds 1-2)")
    ap.add_argument("--alloc-rounds", type=int, default=5, help="fresh allocations of the cloud the kernel is re-timed on (roofline.frac_min / median / max); 0 = skip")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed cloud with the oracle")
    ap.add_argument("--verify-views", type=int, default=4, help="leading views of the timed cloud compared with the oracle (the last view is always added)")
    ap.add_argument("--n1-strong-mpix", type=float, default=0.0, help="N = 1 strong2000 sharded Mpixels/s of an earlier run: speedup_vs_n1 is computed against it")
    ap.add_argument("--views", type=int, default=0, help="override views per GPU (garden185) / total views (scene2000)")
    ap.add_argument("--pixel-index", action="store_true", help="also emit the int32 pixel index per point")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="also time the oracle over this many processes (courtesy all-cores figure); -1 = the cores this job may use, 0/1 = skip")
    ap.add_argument("--strong-views", type=int, default=2000, help="views of the strong-scaling sub-record (BASELINE configs[2]); 0 = skip")
    ap.add_argument("--strong-steps", type=int, default=3, help="timed passes per leg of the strong-scaling sub-record")
    ap.add_argument("--chunks", type=int, default=5, help="view chunks per rank whose exchange overlaps the next chunk's kernel")
    ap.add_argument("--gather-dst", default="all", choices=("all", "0"), help="gathered legs: every rank receives the whole cloud (all) or only rank 0 does")
    ap.add_argument("--gather-timeout", type=float, default=240.0, help="watchdog for the strong-scaling leg, seconds")
    ap.add_argument("--no-dense-guess", action="store_true",
                    help="unmasked batches: count before scattering (CloudBuilder.speculate_dense = False) -- the counted figure of configs[4]")
    ap.add_argument("--conf-kind", default="noise", choices=("noise", "smooth"),
                    help="confidence maps of the conf workloads: independent per pixel (default; the worst case for the attribute gathers) or spatially coherent blobs")
    ap.add_argument("--mask-kind", default="blob", choices=("blob", "bernoulli"),
                    help="blob: smooth regions (default); bernoulli: independent per-pixel cull, worst case for compaction")
    ap.add_argument("--colmap-path", type=Path, default=ROOT / "data" / "360_v2" / "garden" / "sparse" / "0",
                    help="COLMAP model whose poses/intrinsics replace the synthetic ring (used only if it exists)")
    ap.add_argument("--tuning", type=int, default=0)
    ap.add_argument("--two-pass", action="store_true", help="dd_plan + dd_scatter per step instead of the fused single-pass call")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched plainly (`python bench.py --gpus N`): this process -- which has not touched the GPU and never will -- starts the N
        # ranks as fresh children, one per GPU, and leaves with their verdict; rank 0's single line goes straight to the shared stdout
        sys.exit(launch_ranks(args.gpus, [str(Path(__file__).resolve())] + sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world
    explicit = args.workload is not None
    if args.workload is None:
        args.workload = "scene2000"        # the scene BASELINE.json's metric is quoted on, at every N
    args.streaming = [int(x) for x in args.streaming.split(",") if x.strip()]
    if os.environ.get("DD_BENCH_SHARE_GPU") == "1":
        args.placement = "first"         # rehearsal ranks share one GPU: no scouting of its memory by several processes at once
    # one process per GPU, one densify stream: the deployment the north star names.  The single-pass kernel then takes its tiles by
    # workgroup index instead of drawing tickets (CloudBuilder.exclusive_gpu); the line also carries the figure WITH tickets, the
    # library's default for a GPU that may be shared (`roofline.frac_shared_gpu_mode`).  Rehearsal ranks that share a GPU: tickets.
    os.environ.setdefault("DD_EXCLUSIVE_GPU", "0" if os.environ.get("DD_BENCH_SHARE_GPU") == "1" else "1")
    real_out = _claim_stdout()
    guard = None
    if rank == 0 and args.strong_views > 0 and (world > 1 or os.environ.get("DD_BENCH_FORCE_DIST") == "1"):
        guard = _spawn_line_guard(real_out)      # before anything touches the GPU
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if os.environ.get("DD_BENCH_SHARE_GPU") == "1":      # rehearsal: every rank on cuda:0 (1-GPU box), gloo collectives
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    # DD_BENCH_FORCE_DIST=1 runs the N>1 code path (RCCL group, count exchange, all-gatherv) with a
    # single rank, so it can be rehearsed on a 1-GPU box.
    use_dist = world > 1 or os.environ.get("DD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("DD_BENCH_SHARE_GPU") == "1":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)     # "nccl" is RCCL on ROCm

    import depthdensifier_amd as dd
    from depthdensifier_amd import distributed as D

    # under rocprofv3 a physical allocation of the virtual-memory API that is released does not come back to the device: the arena
    # keeps every chunk it is given back in its pool instead (the next workload's cloud is built from them), and the fresh-allocation
    # rounds -- which release and scout anew by design -- are left out (roofline.frac_min / median / max absent in a profiled run)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if profiled and args.placement == "probed":
        from depthdensifier_amd import placement as _plp
        _plp.keep_everything(device)
        args.alloc_rounds = 0

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def run_workload(workload: str, steps: int, warmup: int, alloc_rounds: int, views_override: int, cpu_seconds: float):
        """One workload of WORKLOADS on this job's ranks: scene in HBM, warm-up, the timed steps between two fences, the timed cloud against
        the oracle, the kernel re-timed on fresh allocations.  Returns (the line rank 0 would print for it, what is still resident)."""
        cfg = dict(WORKLOADS[workload])
        cfg["mask_kind"] = args.mask_kind
        cfg["conf_kind"] = cfg.get("conf_kind", args.conf_kind)      # (a workload may fix its kind of confidence map: mip360conf_smooth)
        strong = workload == "scene2000"
        multi = "scenes" in cfg                          # whole scenes back to back, dealt to the ranks
        if views_override:
            if multi:                                    # scale every scene (quick runs)
                cfg["scenes"] = [(n, max(1, round(v * views_override / cfg["V"]))) for n, v in cfg["scenes"]]
            cfg["V"] = views_override
        scene_sets = None
        if multi:
            names, sizes = zip(*cfg["scenes"])
            owner = deal_scenes(sizes, world)
            starts = np.concatenate([[0], np.cumsum(sizes)])
            total_views = int(starts[-1])
            scene_sets = [(names[k], np.arange(starts[k], starts[k + 1])) for k in range(len(sizes)) if owner[k] == rank]
            lo, hi = (int(scene_sets[0][1][0]), int(scene_sets[0][1][-1]) + 1) if scene_sets else (0, 0)
            scaling = "strong"
        elif strong:
            total_views = cfg["V"]
            lo, hi = D.shard_views(total_views, world, rank)
            scaling = "strong"
        else:
            total_views = cfg["V"] * world
            lo, hi = rank * cfg["V"], (rank + 1) * cfg["V"]
            scaling = "weak"
        view_ids = np.arange(lo, hi)
        V = len(view_ids)
        H, W = cfg["H"], cfg["W"]

        # The cloud's arrays are allocated BEFORE the maps (capacity = every pixel of the shard, known from the shapes alone): with the
        # device still empty the zone arena has its pick of all three classes of HBM; behind 83 GB of maps it may not find enough
        # chunks of each (round 6: 'degraded' placement and 0.75 instead of 0.77 on one box in three).  A caller of the library
        # can do the same -- pipeline.main creates its CloudBuilder before the image loop.  DD_BENCH_CLOUD_FIRST=0: the old order.
        early_builder = None
        if not multi and V > 0 and os.environ.get("DD_BENCH_CLOUD_FIRST", "1") == "1":
            early_builder = dd.CloudBuilder(V * H * W, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=args.pixel_index,
                                            device=device, placement=args.placement)
        scene = make_scene(cfg, view_ids, device)
        params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
        E = ring_poses(view_ids, total_views)
        poses_from = "synthetic ring"
        if workload == "garden185" and (args.colmap_path / "images.bin").exists():
            # real registered poses / intrinsics when the dataset is on disk; maps stay synthetic
            from depthdensifier_amd.colmap_io import Reconstruction
            rec = Reconstruction(args.colmap_path)
            imgs = [rec.images[i] for i in sorted(rec.images)]
            if len(imgs) >= hi:
                for j, vid in enumerate(view_ids):
                    im = imgs[int(vid)]
                    cam = rec.cameras[im.camera_id]
                    E[j] = im.cam_from_world().matrix()
                    params[j] = cam.pinhole_params() * [W / cam.width, H / cam.height, W / cam.width, H / cam.height]
                poses_from = str(args.colmap_path)
        def view_batch(sc, pr, Ek, base):
            return dd.ViewBatch(sc["depth"], pr, Ek, mask=sc["mask"], normal=sc["normal"], rgb=sc["rgb"], conf=sc["conf"],
                                conf_threshold=cfg.get("conf"), view_index_base=int(base), device=device, tuning=args.tuning)

        if multi:       # this rank's scenes, each its own ring of cameras and its own batch; `scene` stays the first (CPU baseline)
            batches = []
            for k, (_, ids) in enumerate(scene_sets):
                sc = scene if k == 0 else make_scene(cfg, ids, device)
                batches.append(view_batch(sc, np.tile(params[:1], (len(ids), 1)), ring_poses(np.arange(len(ids)), len(ids)), 0))
            if scene_sets:
                E = ring_poses(np.arange(V), V)
            V = sum(len(ids) for _, ids in scene_sets)
            batch = builder = None
        else:
            batch = view_batch(scene, params, E, lo)
            # capacity = every visited pixel: no sizing pass exists anywhere, timed or not (SURVEY.md 8d defines the metric over
            # count + scan + unproject + compact, which the fused kernel does in its one pass)
            builder = early_builder if early_builder is not None and early_builder.capacity == batch.max_points else \
                dd.CloudBuilder(batch.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=args.pixel_index,
                                device=device, placement=args.placement)
            early_builder = None
            builder.speculate_dense = not args.no_dense_guess

        ev = []
        state = {"plan": None}
        single_pass = not args.two_pass
        auto_two_pass = speculative = None
        if builder is not None and single_pass:
            # the path CloudBuilder.append would take for this cloud and batch: for ONE large row array placed with its thirds in three
            # classes of HBM that is plan + scatter with the scatter walking the thirds in turn.  The same calls are made here
            # separately so that the events bracket the dominant kernel (the scatter) and the count pass on their own.
            tun = builder.fuse_tuning(batch)
            if (tun & 4) and not (batch.tuning & 4):
                batch.tuning, single_pass = tun, False
                auto_two_pass = f"CloudBuilder.fuse_tuning: two-pass, scatter interleaving {1 + ((tun >> 8) & 63)} stretches of tiles (cloud placed '{builder.placement.layout}')"
            elif tun & (1 << 17):
                # unmasked depth maps on a blocked cloud: the fused call runs the scatter against a count-free plan and the scatter verifies
                # it (DDViewBatch.tuning bit 17); builder.append makes that call itself, builder.check() below redoes the batch on a miss
                speculative = (f"CloudBuilder.fuse_tuning: no counting pass -- plan_dense + the scatter kernel, which verifies that every pixel is valid; "
                               f"scatter interleaving {1 + ((tun >> 8) & 63)} stretches of tiles (cloud placed '{builder.placement.layout}')")

        scene_pool, state_placement, big = {}, None, None
        if multi and batches:
            # one long-lived set of arrays sized for the largest scene, placed once (outside the timed region) and handed to every
            # scene's cloud -- what torch's caching allocator did for the per-scene clouds of round 2 anyway (the same memory every
            # time), now in HBM classes of our choosing
            big = dd.CloudBuilder(max(b.max_points for b in batches), normals=cfg["normal"], colors=cfg["rgb"], pixel_index=args.pixel_index,
                                  device=device, placement=args.placement)
            scene_pool = {"points": big.xyz, "normals": big.normal, "colors": big.rgb, "pixel_index": big.pix}
            state_placement = big.placement

        def step_scenes(record: bool):
            """mip360x7: this rank's scenes back to back -- per scene a cloud sized for every visited pixel (on the pooled arrays
            above), the fused call, and the read of the point count and the error word that writing the scene's model needs --
            asked for behind each scene's kernel (``CloudBuilder.check_async``) and looked at once all scenes are enqueued, so the
            host never stands between two kernels (round 3 read them scene by scene: 0.58 of the roofline against 0.60 for one scene).
            The events bracket the whole sequence."""
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
            if record:
                e[0].record()
            pending = []
            for b in batches:
                big.reset()                                 # the next scene's cloud: the same pooled arrays, rows from 0
                big.append(b)
                pending.append(big.check_async())           # count + status on their way to the host; the next scene is enqueued meanwhile
            n = sum(p.result(heal=False) for p in pending)   # (the scenes share the pooled arrays: a redo after the fact has nothing to redo into)
            if record:
                e[1].record(); e[2].record()
                ev.append(e)
            state["n_local"] = n
            return None

        def step(record: bool):
            """One pass of the hot path.  Default: the fused call dd_unproject_compact (one kernel reads the
            inputs once: cull + unproject + transform + look-back scan + compaction + write).  --two-pass:
            dd_plan (count + scans) then dd_scatter.  Events bracket the kernels on the launch stream."""
            if multi:
                return step_scenes(record)
            builder.reset()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
            if record:
                e[0].record()
            if single_pass:
                offs = builder.append(batch)
                builder.join()              # (no-op unless the batch was small enough to run on the builder's side streams)
                if record:
                    e[1].record(); e[2].record()
            else:
                state["plan"] = dd.plan_batch(batch, builder.cursor, reuse=state["plan"])
                if record:
                    e[1].record()
                offs = builder.scatter(batch, state["plan"])
                if record:
                    e[2].record()
            if record:
                ev.append(e)
            if use_dist:                               # the fuse exchange: global view offsets on every rank
                counts = offs[1:] - offs[:-1]
                return D.offsets_from_counts(D.exchange_counts(counts, total_views))
            return offs

        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            goffs = step(True)
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())

        if multi:
            n_local = state["n_local"]
            n_total = n_local
            if use_dist:
                nt = torch.tensor([n_local], dtype=torch.int64, device=device)
                dist.all_reduce(nt)
                n_total = int(nt.item())
        else:
            n_local = builder.check()
            n_total = int(goffs[-1].item())
        if os.environ.get("DD_BENCH_TRACE_STEPS") and rank == 0:      # per-step kernel times (diagnostic)
            print("steps_ms", [round(e[0].elapsed_time(e[2]), 3) for e in ev], file=sys.stderr)
        plan_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        k_all = [e[1].elapsed_time(e[2]) for e in ev] if not single_pass else [e[0].elapsed_time(e[1]) for e in ev]
        kernel_ms = float(np.mean(k_all))
        if single_pass:
            plan_ms = 0.0
        # the same step with tiles drawn by ticket: what a caller gets who cannot promise to have the GPU to itself (the library's default)
        shared_ms = None
        if builder is not None and single_pass and builder.exclusive_gpu and V > 0:
            builder.exclusive_gpu = False
            ts = []
            for _ in range(7):
                builder.reset()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); builder.append(batch); builder.join(); e1.record()
                torch.cuda.synchronize(device)
                ts.append(e0.elapsed_time(e1))
            builder.check()
            shared_ms = float(np.median(ts[2:]))
            builder.exclusive_gpu = True
            builder.reset(); builder.append(batch); builder.check()      # (the cloud that is verified below is the default mode's again)

        # ---- the timed cloud against the oracle (untimed; every rank checks views of its own shard)
        verified = None
        if not multi and not args.no_verify and V > 0:
            views = sorted(set(list(range(min(V, args.verify_views))) + [V - 1]))
            try:
                verified = verify_views(dd, cfg, scene, params, E, views, builder._offsets[-1],
                                        {"points": builder.xyz, "normals": builder.normal, "colors": builder.rgb}, device, lo)
                ok = 1
            except AssertionError as e:
                verified, ok = {"error": str(e)[:300]}, 0
            if use_dist:
                flag = torch.tensor([ok], dtype=torch.int64, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            verified["all_ranks_ok"] = bool(ok)
            verified["what"] = "rows of the TIMED cloud (last timed step) vs oracle/densify_oracle.py on this rank's first views and its last one"

        # ---- the same kernel on fresh allocations of the cloud (the placement of the output arrays is the one thing that moves it)
        alloc_ms, alloc_how = [], []
        if not multi and single_pass and alloc_rounds > 0 and V > 0:
            keep = []
            from depthdensifier_amd import placement as _pl
            cloud_bytes = batch.max_points * (12 + (12 if cfg["normal"] else 0) + (3 if cfg["rgb"] else 0) + (4 if args.pixel_index else 0))
            for r in range(alloc_rounds):
                _pl.trim(device)              # no spare chunks from the last round: every round scouts the device's memory anew
                torch.cuda.empty_cache()      # (blocks torch keeps cached -- the verification's temporaries -- are not free memory to the driver)
                if torch.cuda.mem_get_info(device)[0] < 1.15 * cloud_bytes + (4 << 30):
                    break                     # no room for a second cloud beside the timed one (2000 views on one GPU)
                b2 = None
                try:      # (a courtesy figure: ranks that share a card -- rehearsals -- may both have seen the room and not both get it; no collective in here)
                    b2 = dd.CloudBuilder(batch.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=args.pixel_index, device=device,
                                         placement=args.placement)
                    b2.speculate_dense = not args.no_dense_guess
                    for _ in range(2):
                        b2.reset(); b2.append(batch)
                    ts = []
                    for _ in range(5):
                        b2.reset()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); b2.append(batch); b2.join(); e1.record()
                        torch.cuda.synchronize(device)
                        ts.append(e0.elapsed_time(e1))
                    alloc_ms.append(float(np.median(ts)))
                    alloc_how.append("first" if b2.placement is None else f"{b2.placement.mode[:120]} / {b2.placement.layout}")
                    if args.placement == "first":
                        keep.append(torch.empty((r + 1) << 30, dtype=torch.uint8, device=device))    # the next allocation starts elsewhere
                except (torch.OutOfMemoryError, MemoryError) as e:
                    print(f"[bench] rank {rank}: fresh-allocation round {r} left out ({type(e).__name__})", file=sys.stderr)
                    del b2
                    torch.cuda.empty_cache()
                    break
                del b2
                torch.cuda.empty_cache()
            del keep

        devices = None
        if use_dist:
            props_r = torch.cuda.get_device_properties(device)
            mine = f"rank {rank}: {torch.cuda.get_device_name(device)} pci {getattr(props_r, 'pci_bus_id', 0):02x}:{getattr(props_r, 'pci_device_id', 0):02x} uuid {getattr(props_r, 'uuid', '?')}"
            devices = [None] * world
            dist.all_gather_object(devices, mine)

        if rank == 0:
            props = torch.cuda.get_device_properties(device)
            ms_per_step = elapsed / steps * 1e3
            pixels = total_views * H * W
            alg = algorithmic_bytes(cfg, V, n_local, args.pixel_index)
            alg_r = algorithmic_bytes(cfg, V, n_local, args.pixel_index, reads_only=True)
            achieved = alg / (kernel_ms * 1e-3) / 1e9
            traffic, traffic_source = None, None
            tfile = ROOT / "profiles" / "traffic.json"
            if tfile.exists():
                tkey = ("mip360conf" if multi or workload == "mip360conf_smooth" else workload) + (":bernoulli" if args.mask_kind == "bernoulli" else "") + (":smooth" if cfg.get("conf") and cfg["conf_kind"] == "smooth" else "") + ("" if single_pass and not speculative else ":two-pass")      # (the speculative call runs the two-pass scatter kernel)
                # (mip360x7 runs the mip360conf kernel scene after scene on the same kind of maps: its bytes per view)
                rec = json.loads(tfile.read_text()).get(tkey)
                if rec:     # PMC bytes were collected on the full workload; a launch over fewer views moves proportionally fewer
                    traffic = int(rec["hbm_bytes_per_launch"] * V / rec["views"])
                    traffic_source = (f"profiles/traffic.json[{tkey}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel on this workload, "
                                      "collected in separate profiling runs (tools/pmc_traffic.sh) -- NOT measured in this run")
            if multi:
                cfg_scenes = {"scenes": [f"{n}:{len(i)}" for n, i in scene_sets], "scenes_total": len(cfg["scenes"])}
            line = {
                "metric": "Mpixels/s unprojected+fused",
                "value": round(pixels / (elapsed / steps) / 1e6, 1),
                "unit": "Mpixels/s",
                "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "dtype": "f32" if cfg["depth"] == "float32" else "f16-in/f32-out",
                "data": "synthetic",
                "device": f"{torch.cuda.get_device_name(device)} pci {getattr(props, 'pci_bus_id', '?'):02x}:{getattr(props, 'pci_device_id', 0):02x}",
                "mpoints_per_s": round(n_total / (elapsed / steps) / 1e6, 1),
                "config": {"workload": workload, "note": cfg["note"], "views_total": total_views, "views_per_gpu": V,
                           "height": H, "width": W, "downsample_density": 1, "valid_fraction": round(n_local / (V * H * W), 4), "mask_kind": args.mask_kind,
                           "conf_kind": cfg["conf_kind"] if cfg.get("conf") else None,
                           "poses": poses_from,
                           "tiles": ("by workgroup index: exclusive_gpu, one process per GPU -- the default of the drop-in pipeline (ProcessingConfig) since round 6; "
                                     "roofline.frac_shared_gpu_mode has the figure with tickets, the library's default for a GPU that may be shared")
                                    if os.environ.get("DD_EXCLUSIVE_GPU") == "1" else "by ticket (a GPU that may be shared)",
                           "inputs": "+".join(k for k in ("depth", "mask", "conf", "normal", "rgb") if scene[k] is not None),
                           "outputs": "xyz f32" + (" + normal f32" if cfg["normal"] else "") + (" + rgb u8" if cfg["rgb"] else "")
                                      + (" + pixel_index i32" if args.pixel_index else ""),
                           "fuse": "whole scenes per rank, one cloud per scene (on one pooled set of arrays), no data-path collective" if multi else
                                   "single GPU: one global scan, points written at final slots" if world == 1 else
                                   "sharded: contiguous view shards + RCCL all-gather of per-view counts (global offsets)"},
                "roofline": {"bound": "hbm",
                             "kernel": "compact_lean<single-pass> (dd_unproject_compact: cull+unproject+transform+scan+compact+write)"
                                       if single_pass and not speculative else "compact_lean (dd_scatter: cull+unproject+transform+compact+write)"
                                       + (f"; {auto_two_pass}" if auto_two_pass else "") + (f"; {speculative}" if speculative else ""),
                             "achieved": round(achieved, 1),
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                             "read_frac": round(alg_r / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "read_frac_note": "algorithmic READ bytes / kernel time / peak (the other "
                                               f"{100 * (1 - alg_r / alg):.0f} % of the bytes are writes sharing the same interface)",
                             "traffic": traffic, "traffic_source": traffic_source,
                             # the counters' bytes over the same time: what the kernel really moves through the fabric (per-pixel culls
                             # fetch whole 128-byte lines of the normal / colour maps for one survivor: frac undercounts them)
                             "traffic_frac": None if not traffic else round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "traffic_over_algorithmic": None if traffic is None else round(traffic / alg, 4),
                             "algorithmic_bytes_per_launch": alg,
                             "kernel_ms": round(kernel_ms, 4), "kernel_ms_min": round(float(np.min(k_all)), 4),
                             "kernel_ms_median": round(float(np.median(k_all)), 4),
                             "timer": "HIP events on the launch stream; achieved uses the mean over the timed steps",
                             "sizing_pass": "none: the cloud is allocated for every visited pixel (capacity = V*H*W rows)",
                             "pass1_ms": round(plan_ms, 4),
                             "pass1_note": "two-pass mode only: count_lean + scan kernels re-read depth+mask (not credited); whole_step_frac and `value` include them",
                             "whole_step_frac": round(alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "exclusive_gpu": None if builder is None else bool(builder.exclusive_gpu),
                             "frac_shared_gpu_mode": None if shared_ms is None else round(alg / (shared_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "shared_gpu_mode_note": "the same launch with its tiles drawn by ticket instead of taken by workgroup index: the library's default, "
                                                     "safe when other launches of the kind share the GPU (DD_EXCLUSIVE_GPU=0; median of 5 launches)"},
            }
            rf = line["roofline"]
            if builder is not None:                     # redone batches: look-back give-ups and dense speculations that missed (0 / 0 expected)
                rf["redone"] = {"healed": int(builder.healed), "dense_misses": int(builder.dense_misses)}
            prep = (builder.placement if builder is not None else state_placement) if (builder is not None or multi) else None
            rf["placement"] = "first" if prep is None else (args.placement if prep.mode == "probed" else prep.mode)
            rf["placement_report"] = None if prep is None else prep.as_dict()
            if alloc_ms:
                fr = [alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS for t in alloc_ms]
                rf.update({"frac_min": round(min(fr), 4), "frac_median": round(float(np.median(fr)), 4), "frac_max": round(max(fr), 4),
                           "alloc_rounds": len(fr), "kernel_ms_per_allocation": [round(t, 4) for t in alloc_ms], "placement_per_allocation": alloc_how,
                           "frac_note": "frac is the driver's timed run; frac_min / median / max re-time the same kernel on this many FRESH allocations "
                                        "of the cloud in the same process (placement as above), median of 5 launches each"})
            if verified is not None:
                line["verified"] = verified
            if devices is not None:
                line["devices"] = devices
                line["rccl_world_size"] = dist.get_world_size()
                line["collective_backend"] = dist.get_backend()
            if strong and args.n1_strong_mpix > 0:
                line["speedup_vs_n1"] = round(line["value"] / args.n1_strong_mpix, 3)
            if multi:
                line["config"]["rank0_scenes"] = cfg_scenes["scenes"]
                line["roofline"]["kernel"] += f"; {len(batches)} launches, the events also bracket the per-scene allocation and host read"
            if cpu_seconds > 0 and world == 1:      # reported at N=1 only, on rank 0
                line["cpu_baseline"] = cpu_baseline(cfg, scene, params, E, cpu_seconds)
                procs = args.cpu_procs if args.cpu_procs >= 0 else min(len(os.sched_getaffinity(0)), 16)      # a GPU box's CPU share is 16 cores
                profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
                if profiled:        # spawned workers would inherit the profiler's preload (and its GPU initialisation): keep to one process
                    line["cpu_baseline"]["all_cores"] = {"skipped": "running under a profiler"}
                elif procs > 1:
                    try:
                        line["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(cfg, scene, params, E, cpu_seconds / 2,
                                                                                   line["cpu_baseline"]["value"], procs)
                    except Exception as e:      # noqa: BLE001  (a courtesy figure must not cost the result)
                        line["cpu_baseline"]["all_cores"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        else:
            line = None
        state = dict(cfg=cfg, scene=scene, params=params, E=E, batch=batch, builder=builder, lo=lo, V=V, n_local=n_local, multi=multi,
                     verified=verified, alg=None if multi else algorithmic_bytes(cfg, V, n_local, args.pixel_index), total_views=total_views)
        return line, state


    t_main = time.perf_counter()
    phases = {}                          # wall seconds of this process's phases (what the driver's clock around the run is made of)
    line, st = run_workload(args.workload, args.steps, args.warmup, args.alloc_rounds, args.views, args.cpu_seconds)
    phases[args.workload] = round(time.perf_counter() - t_main, 1)
    verified, strong = st["verified"], None
    failed = verified is not None and not verified.get("all_ranks_ok", True)

    # the garden185 streaming chains belong to whichever record holds garden185 (here: the main one, when asked for explicitly)
    if args.workload == "garden185" and args.streaming and world == 1 and st["builder"] is not None and st["V"] > 0:
        sr = streaming_record(args, dd, st["cfg"], st["scene"], st["params"], st["E"], st["batch"], st["builder"], device, st["lo"], st["alg"])
        failed = failed or not sr["all_ok"]
        if rank == 0:
            line["streaming"] = sr
        fr = fused_refine_record(dd, st["cfg"], st["scene"], st["params"], st["E"], st["builder"], device)
        failed = failed or not fr["equals_refine_apply_then_plain"]
        if rank == 0:
            line["fused_refine"] = fr

    # BASELINE configs[2] (2000-view strong scaling: sharded / gathered / gathered-compact / Bernoulli), timed AFTER the main
    # result exists.  Doubly guarded: an exception is reported in the line; a collective that does not finish within
    # the watchdog's limit makes every rank print the main result and exit NON-ZERO (a hung GPU process is a failed leg).
    if args.strong_views > 0:
        import threading

        reuse = None
        if args.workload == "scene2000" and st["total_views"] == args.strong_views and args.tuning == 0 and not st["multi"]:
            reuse = dict(scene=st["scene"], batch=st["batch"], params=st["params"])      # the same views are resident already
        st["builder"] = None
        if reuse is None:
            st.clear()
        import gc
        gc.collect()                     # (the main cloud -- 112 GB of arena memory on the default workload -- goes back to the driver NOW)
        torch.cuda.empty_cache()

        def bail():
            if rank == 0:
                line["strong2000"] = {"error": f"strong-scaling leg did not finish within {args.gather_timeout:.0f} s"}
                print(json.dumps(line), file=real_out, flush=True)
                _release_line_guard(guard)
            os._exit(3)

        if guard is not None:       # from here on a hard crash of rank 0 still leaves the main result on stdout
            held = dict(line, strong2000={"error": "rank 0 died inside the strong-scaling leg (main result printed by the guard process)"})
            guard.stdin.write((json.dumps(held) + "\n").encode())
            guard.stdin.flush()
        if os.environ.get("DD_BENCH_TEST_ABORT") == "1":      # test hook for the guard
            os.abort()

        dog = threading.Timer(args.gather_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            strong = strong_scaling_record(args, dd, D, dist, use_dist, rank, world, device, fence, reuse=reuse)
        except Exception as e:      # noqa: BLE001  (reported, not fatal)
            strong = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
        if rank == 0:
            line["strong2000"] = strong
        reuse = None
    st.clear()
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if torch.cuda.is_available() and world == 1 and not profiled:
        from depthdensifier_amd import placement as _pl0
        _pl0.trim(device)                # spare chunks of the arena go back too: the sub-records start from an empty device

    phases["strong2000"] = round(time.perf_counter() - t_main - sum(phases.values()), 1)
    # the other single-GPU configurations of BASELINE.json as sub-records of the default line (N = 1 only: every one of them is a
    # one-GPU workload; the driver's N > 1 runs measure the scaling curve of the main workload)
    subs = []
    if args.sub == "auto":
        subs = ["garden185", "roofline12mp", "mip360conf", "mip360conf_smooth"] if (world == 1 and not explicit and not use_dist) else []
    elif args.sub != "none":
        subs = [x for x in args.sub.split(",") if x.strip()]
    for name in subs:
        if name == args.workload or name not in WORKLOADS or "scenes" in WORKLOADS[name]:
            continue
        try:
            from depthdensifier_amd import placement as _pl
            if not profiled:
                _pl.trim(device)
            torch.cuda.empty_cache()
            v_sub = min(args.views, WORKLOADS[name]["V"]) if args.views else 0
            sub_line, sst = run_workload(name, args.sub_steps, 3, min(args.alloc_rounds, 3), v_sub, 0.0)
            rec = None
            if rank == 0:
                r = sub_line["roofline"]
                rec = {"workload": name, "note": WORKLOADS[name]["note"], "value": sub_line["value"], "unit": sub_line["unit"], "ms_per_step": sub_line["ms_per_step"],
                       "mpoints_per_s": sub_line["mpoints_per_s"], "steps": args.sub_steps, "warmup": 3, "dtype": sub_line["dtype"],
                       "config": {k: sub_line["config"][k] for k in ("views_total", "height", "width", "valid_fraction", "mask_kind", "conf_kind", "inputs", "outputs")},
                       "roofline": {k: r.get(k) for k in ("kernel", "achieved", "frac", "whole_step_frac", "traffic_frac", "traffic_over_algorithmic", "read_frac",
                                                          "algorithmic_bytes_per_launch", "kernel_ms", "kernel_ms_min", "pass1_ms", "frac_min", "frac_median",
                                                          "frac_max", "alloc_rounds", "redone", "placement", "exclusive_gpu", "frac_shared_gpu_mode")},
                       "verified": sub_line.get("verified")}
            sv = sst["verified"]
            failed = failed or (sv is not None and not sv.get("all_ranks_ok", True))
            if name == "garden185" and args.streaming and sst["builder"] is not None and sst["V"] > 0:
                sr = streaming_record(args, dd, sst["cfg"], sst["scene"], sst["params"], sst["E"], sst["batch"], sst["builder"], device, sst["lo"], sst["alg"])
                failed = failed or not sr["all_ok"]
                if rank == 0:
                    rec["streaming"] = sr
                fr = fused_refine_record(dd, sst["cfg"], sst["scene"], sst["params"], sst["E"], sst["builder"], device)
                failed = failed or not fr["equals_refine_apply_then_plain"]
                if rank == 0:
                    rec["fused_refine"] = fr
                try:
                    pr = pipeline_record(device)
                    pr["densify_share_of_loop"] = round(fr["us_per_view"] * 1e-6 / (pr["loop_wall_ms_per_view"] * 1e-3), 4)
                except Exception as e:      # noqa: BLE001  (no writable temporary directory ...: the line goes on without it)
                    pr = {"error": f"{type(e).__name__}: {e}"[:300]}
                if rank == 0:
                    line["pipeline"] = pr
            sst.clear()
            del sst
            gc.collect()
        except Exception as e:      # noqa: BLE001  (one sub-record failing must not cost the line)
            rec = {"workload": name, "error": f"{type(e).__name__}: {e}"[:300]}
            failed = True
        if rank == 0:
            line[name] = rec
        torch.cuda.empty_cache()
        phases[name] = round(time.perf_counter() - t_main - sum(phases.values()), 1)

    if rank == 0:
        phases["total_since_main"] = round(time.perf_counter() - t_main, 1)
        line["wall_s"] = phases
        print(json.dumps(line), file=real_out, flush=True)
        _release_line_guard(guard)

    if isinstance(strong, dict):
        for leg in (strong, strong.get("bernoulli") or {}):
            if isinstance(leg.get("verified"), dict) and not leg["verified"].get("all_ranks_ok", True):
                failed = True
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: a timed cloud does NOT match the oracle, or a sub-record failed (see the \"verified\" / \"error\" records of the line)")


if __name__ == "__main__":
    main()
