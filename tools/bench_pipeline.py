#!/usr/bin/env python3
"""End-to-end `pipeline.main` on a synthetic 1080p COLMAP scan (cached depth maps): stage breakdown.
The scan is generated into a temp dir by tests/scan_factory.py (not timed)."""
import contextlib, io, json, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from scan_factory import make_scan
from depthdensifier_amd import pipeline as P
V = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 185
JPG = "--jpg" in sys.argv      # photograph-like JPEG images (what real scans hold) instead of PNG files of noise
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.time()
    scan, _cache, _truth = make_scan(Path(tmp), "scan", V=V, H=1080, W=1920, seed=1, **({"image_ext": ".jpg", "photo_like": True} if JPG else {}))
    print(f"scan of {V} views generated in {time.time() - t0:.1f}s", flush=True)
    import numpy as np
    npy_cache = scan / "moge_cache_npy"
    npy_cache.mkdir()
    from PIL import Image as PILImage
    rgb_cache = scan / "moge_cache_npy_rgb"          # the same maps plus the decoded image (dump_cache(with_rgb=True))
    rgb_cache.mkdir()
    for f in sorted((scan / "moge_cache").glob("*.npz")):
        with np.load(f) as z:
            for k in z.files:
                np.save(npy_cache / f"{f.stem}_{k}.npy", z[k])
                (rgb_cache / f"{f.stem}_{k}.npy").symlink_to(npy_cache / f"{f.stem}_{k}.npy")
        img = next(p for p in (scan / "images").iterdir() if p.stem == f.stem)
        np.save(rgb_cache / f"{f.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))
    import os
    ncores = len(os.sched_getaffinity(0))
    MATRIX = "--matrix" in sys.argv       # round 3's sweep over strides / io threads / cache layouts; default: the two runs VERDICT r5 asks for
    nio_default = max(2, min(16, ncores - 2))
    if "--io" in sys.argv:
        nio_default = int(sys.argv[sys.argv.index("--io") + 1])
    STRIDES = [int(x) for x in sys.argv[sys.argv.index("--strides") + 1].split(",")] if "--strides" in sys.argv else [1, 32]
    runs = ([(32, 0), (32, 4), (32, 8), (4, 8), (1, 8), (1, nio_default)] if MATRIX else [(st, nio_default) for st in STRIDES])
    for stride, nio in runs:
      for cache in (("moge_cache", "moge_cache_npy") + (("moge_cache_npy_rgb",) if stride == 1 or nio == 8 else ()) if MATRIX else ("moge_cache_npy_rgb",)):
       for rep in range(1 if MATRIX else 3):        # (warm caches: the first run of a configuration pages the files in and pins the staging slots)
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=Path(tmp) / f"out_s{stride}")
        cfg.moge.cache_dir = scan / cache
        cfg.processing.downsample_density = stride
        cfg.processing.pipeline_downsample_factor = 1
        cfg.processing.io_threads = nio
        cfg.refiner.verbose = 0
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rep_ = P.main(cfg)
        torch.cuda.synchronize()
        t = {k: round(v, 3) for k, v in rep_["timings"].items()}
        loop = sum(rep_["timings"][k] for k in ("image_decode", "depth_source", "refine", "densify"))
        wait = rep_["loop_detail"].get("finish_refine_of_which_waiting_for_the_gpu", 0.0) + rep_["loop_detail"].get("wait_for_io_thread", 0.0)
        line = {"stride": stride, "io_threads": nio, "run": rep, "views": rep_["views"],
                "host_ms_per_view": round(loop / rep_["views"] * 1e3, 3),
                "host_ms_per_view_without_waits": round((loop - wait) / rep_["views"] * 1e3, 3),
                "loop_wall_ms_per_view": round(rep_["loop_seconds"] / rep_["views"] * 1e3, 3),
                "images": "jpg" if JPG else "png", "cache": "npy+rgb" if cache.endswith("rgb") else "npy" if cache.endswith("npy") else "npz",
                "dense_points": rep_["dense_points"], "removed": rep_["removed"], "seconds": t,
                "upload_ms_per_view_on_the_copy_stream": None if "upload_seconds_on_the_copy_stream" not in rep_ else round(rep_["upload_seconds_on_the_copy_stream"] / rep_["views"] * 1e3, 3),
                "copy_engine_busy_fraction_of_the_loop": None if "upload_seconds_on_the_copy_stream" not in rep_ else round(rep_["upload_seconds_on_the_copy_stream"] / rep_["loop_seconds"], 3),
                "loop_detail_us_per_view": {k: round(v / rep_["views"] * 1e6, 1) for k, v in rep_["loop_detail"].items()}}
        print(json.dumps(line), flush=True)
