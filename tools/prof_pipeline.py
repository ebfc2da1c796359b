import sys, tempfile, time, contextlib, io, cProfile, pstats
from pathlib import Path
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from scan_factory import make_scan
from depthdensifier_amd import pipeline as P
with tempfile.TemporaryDirectory() as tmp:
    import os
    scan, _c, _t = make_scan(Path(tmp), "scan", V=int(os.environ.get("DD_PROF_VIEWS", "48")), H=1080, W=1920, seed=1)
    npy = scan / "moge_cache_npy"; npy.mkdir()
    from PIL import Image as PILImage
    for f in sorted((scan / "moge_cache").glob("*.npz")):
        with np.load(f) as z:
            for k in z.files: np.save(npy / f"{f.stem}_{k}.npy", z[k])
        if os.environ.get("DD_PROF_RGB", "1") == "1":          # the decoded image in the cache too (dump_cache(with_rgb=True)): nothing is decoded
            img = next(p for p in (scan / "images").iterdir() if p.stem == f.stem)
            np.save(npy / f"{f.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))
    cfg = P.ScriptConfig()
    cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=Path(tmp) / "out")
    cfg.moge.cache_dir = npy
    cfg.processing.downsample_density = int(os.environ.get('DD_PROF_STRIDE', '32'))
    cfg.refiner.verbose = 0
    with contextlib.redirect_stdout(io.StringIO()):
        P.main(cfg)
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable(); r = P.main(cfg); pr.disable()
    print({k: round(v, 3) for k, v in r["timings"].items()})
    print(f"host ms per view (sum of the loop's stages): {1e3 * sum(r['timings'][k] for k in ('image_decode', 'depth_source', 'refine', 'densify')) / r['views']:.3f}")
    st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(int(os.environ.get("DD_PROF_LINES", "70")))
    st.sort_stats("tottime"); st.print_stats(35)
    st.print_callers("method 'to' of")
    st.print_callers("synchronize")
