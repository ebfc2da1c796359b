"""Drop-in surfaces (SURVEY.md 8b iii-iv, 8f f3/f4): COLMAP binary I/O, the tyro-style CLI, the
batch driver's error conventions, and -- on the GPU -- the whole scan pipeline."""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "scripts"), str(ROOT / "tests")]


def test_colmap_roundtrip_and_bulk_append(tmp_path):
    from depthdensifier_amd.colmap_io import Reconstruction, write_ply
    from scan_factory import make_scan
    scan, _, _ = make_scan(tmp_path, "s0", V=3)
    rec = Reconstruction(scan / "sparse" / "0")
    assert rec.num_reg_images() == 3 and rec.num_points3D() > 100
    im = rec.images[2]
    E = im.cam_from_world().matrix()
    assert E.shape == (3, 4) and np.allclose(E[:, :3] @ E[:, :3].T, np.eye(3), atol=1e-12)
    assert np.allclose(im.projection_center(), -E[:, :3].T @ E[:, 3])
    inv = im.cam_from_world().inverse()
    p = np.random.default_rng(0).standard_normal((5, 3))
    assert np.allclose(inv * (im.cam_from_world() * p), p)
    ids = im.observed_point3D_ids()
    assert np.array_equal(rec.xyz_of(ids), np.array([rec.points3D[int(i)].xyz for i in ids]))
    assert sum(1 for q in im.points2D if q.has_point3D()) == len(ids)
    n0 = rec.num_points3D()
    new = rec.add_points3D(np.arange(30, dtype=float).reshape(10, 3), np.full((10, 3), 7, np.uint8))
    assert new[0] == rec.point_ids[:n0].max() + 1
    rec.write_binary(tmp_path / "out")
    back = Reconstruction(tmp_path / "out")
    assert back.num_points3D() == n0 + 10
    assert np.array_equal(back.point_xyz, rec.point_xyz) and np.array_equal(back.point_rgb, rec.point_rgb)
    assert np.array_equal(back.point_error[-10:], np.full(10, -1.0))
    assert back.images[2].name == im.name and np.array_equal(back.images[2].point3D_ids, im.point3D_ids)
    assert np.array_equal(back.cameras[1].params, rec.cameras[1].params)
    write_ply(tmp_path / "c.ply", rec.point_xyz, rec.point_rgb, rec.point_xyz)
    head = (tmp_path / "c.ply").read_bytes()[:200].decode("ascii", "replace")
    assert f"element vertex {n0 + 10}" in head and "property uchar red" in head


def test_camera_rescale_semantics():
    from depthdensifier_amd.colmap_io import Camera
    c = Camera(1, 1, 1920, 1080, np.array([1500.0, 1400.0, 960.0, 540.0]))
    c.rescale(960, 540)
    assert (c.width, c.height) == (960, 540) and np.allclose(c.params, [750, 700, 480, 270])
    c.rescale(960, 540)                                   # idempotent at the same size (scripts/test.py:173 per image)
    assert np.allclose(c.params, [750, 700, 480, 270])
    s = Camera(2, 0, 100, 50, np.array([80.0, 50.0, 25.0]))  # SIMPLE_PINHOLE: one focal, mean scale
    s.rescale(50, 50)
    assert np.allclose(s.params, [80 * 0.75, 25, 25]) and np.allclose(s.calibration_matrix()[0, 0], 60)


def test_cli_spellings_match_tyro():
    import run_batch
    from depthdensifier_amd.cli import parse
    c = parse(run_batch.BatchConfig, ["--root-dir", "/d/scans", "--output-dir", "/d/out", "--config.filtering.vote-threshold", "3",
                                      "--config.processing.downsample-density", "1", "--config.refiner.no-use-fp16",
                                      "--config.refiner.verbose", "1", "--config.paths.recon-path", "/x"])
    assert c.root_dir == Path("/d/scans") and c.config.filtering.vote_threshold == 3
    assert c.config.processing.downsample_density == 1 and c.config.processing.pipeline_downsample_factor == 1
    assert c.config.refiner.use_fp16 is False and c.config.refiner.verbose == 1 and c.config.refiner.robust is True
    assert c.config.filtering.depth_threshold == 0.7 and c.config.paths.recon_path == Path("/x")
    with pytest.raises(SystemExit):
        parse(run_batch.BatchConfig, ["--output-dir", "/d/out"])          # root_dir is required


def test_run_batch_conventions_without_gpu(tmp_path, capsys):
    """scripts/run_batch.py:48-50, 69-71, 82-91: bad root -> message; scans without sparse/0 or images are
    skipped; a scan that raises is recorded as FAILED and the batch goes on."""
    import torch
    import run_batch
    from scan_factory import make_scan
    assert run_batch.main(run_batch.BatchConfig(tmp_path / "nope", tmp_path / "out")) == []
    assert "Root directory not found" in capsys.readouterr().out
    (tmp_path / "scans" / "a_incomplete" / "images").mkdir(parents=True)
    make_scan(tmp_path / "scans", "b_scan", V=2)
    cfg = run_batch.BatchConfig(tmp_path / "scans", tmp_path / "out")
    cfg.config.moge.cache_dir = tmp_path / "scans" / "b_scan" / "missing_cache"
    rep = run_batch.main(cfg)
    out = capsys.readouterr().out
    assert "Skipping 'a_incomplete'" in out
    assert rep == [("b_scan", "FAILED")] and "An error occurred while processing 'b_scan'" in out
    assert "Batch Processing Time Report" in out
    if not torch.cuda.is_available():
        assert "no CPU fallback" in out or "AMD GPU" in out


@pytest.mark.gpu
def test_pipeline_end_to_end_on_gpu(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import run_batch
    import depthdensifier_amd as dd
    from depthdensifier_amd.colmap_io import Reconstruction
    from oracle import filter_oracle as forc
    from scan_factory import make_scan

    scan, cache, truth = make_scan(tmp_path / "scans", "plane", V=5, floaters=0.03)
    n_sparse = Reconstruction(scan / "sparse" / "0").num_points3D()
    cfg = run_batch.BatchConfig(tmp_path / "scans", tmp_path / "out")
    cfg.config.moge.cache_dir = cache
    cfg.config.processing.downsample_density = 2
    cfg.config.refiner.use_fp16 = False
    cfg.config.refiner.adaptive_correspondences = False
    cfg.config.filtering.vote_threshold = 2
    rep = run_batch.main(cfg)
    assert len(rep) == 1 and isinstance(rep[0][1], float)
    out = Reconstruction(tmp_path / "out" / "plane" / "sparse" / "0")
    dense = out.point_xyz[n_sparse:]
    assert len(dense) > 1000 and np.array_equal(out.point_xyz[:n_sparse], Reconstruction(scan / "sparse" / "0").point_xyz)
    assert np.all(out.point_error[n_sparse:] == -1.0)

    # the refiner recovered metric scale: surviving dense points lie on the plane y = 0
    assert np.median(np.abs(dense[:, 1])) < 0.05       # (the transfer curve clamps beyond the sparse depth range)
    # expected result rebuilt stage by stage: our refiner + densify, then the ORACLE filter on that cloud
    src = Reconstruction(scan / "sparse" / "0")
    refiner = dd.DepthRefiner(use_fp16=False, adaptive_correspondences=False)
    depths, Ks, Es, builder = [], [], [], None
    for i in sorted(src.images):
        im = src.images[i]; tr = truth[i - 1]
        E = im.cam_from_world().matrix(); K = src.cameras[1].calibration_matrix()
        r = refiner.refine_depth(tr["mono"], tr["normal"], src.xyz_of(im.observed_point3D_ids()), E, K, tr["mask"])
        depths.append(r["refined_depth"]); Ks.append(K); Es.append(E)
    depths = np.stack(depths); masks = np.stack([t["mask"] for t in truth])
    cloud = dd.unproject_views(depths, np.stack(Ks), np.stack(Es), mask=masks, normal=np.stack([t["normal"] for t in truth]),
                               rgb=np.stack([t["rgb"] for t in truth]), downsample_density=2)
    c = cloud.numpy()
    culled = np.where(masks, depths, 0).astype(np.float32)
    ep, ec, _, votes = forc.filter_floaters(c["points"].astype(np.float32), c["colors"], c["normals"], culled, np.stack(Ks),
                                            np.stack(Es), vote_threshold=2, depth_threshold=0.7)
    assert (votes >= 2).sum() > 20                                       # the injected floaters are caught
    assert len(dense) == len(ep)
    assert np.array_equal(dense.astype(np.float32), ep) and np.array_equal(out.point_rgb[n_sparse:], ec)

    # reading ahead on I/O threads (the default) changes nothing: the inline loop writes the same model
    cfg2 = run_batch.BatchConfig(tmp_path / "scans", tmp_path / "out_inline")
    cfg2.config = cfg.config
    cfg2.config.processing.io_threads = 0
    run_batch.main(cfg2)
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert (tmp_path / "out_inline" / "plane" / "sparse" / "0" / name).read_bytes() == \
               (tmp_path / "out" / "plane" / "sparse" / "0" / name).read_bytes(), name


@pytest.mark.gpu
def test_pipeline_downsample_factor_two_on_gpu(tmp_path):
    """``pipeline_downsample_factor = 2`` (``scripts/test.py:145-152, 172-173``): the image FILES and the sparse model's
    cameras are twice the size of the maps; the pipeline must LANCZOS-resize the picture, rescale the camera in place,
    size the cloud from the files -- and write the rescaled cameras.  Checked stage by stage like the f = 1 test."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image as PILImage
    import run_batch
    import depthdensifier_amd as dd
    from depthdensifier_amd.colmap_io import Reconstruction
    from oracle import filter_oracle as forc
    from scan_factory import make_scan

    H, W = 96, 128
    scan, cache, truth = make_scan(tmp_path / "scans", "plane2x", V=5, H=H, W=W, floaters=0.03, image_scale=2)
    src = Reconstruction(scan / "sparse" / "0")
    assert (src.cameras[1].width, src.cameras[1].height) == (2 * W, 2 * H)
    n_sparse = src.num_points3D()
    cfg = run_batch.BatchConfig(tmp_path / "scans", tmp_path / "out")
    cfg.config.moge.cache_dir = cache
    cfg.config.processing.pipeline_downsample_factor = 2
    cfg.config.processing.downsample_density = 1
    cfg.config.refiner.use_fp16 = False
    cfg.config.refiner.adaptive_correspondences = False
    cfg.config.filtering.vote_threshold = 2
    rep = run_batch.main(cfg)
    assert len(rep) == 1 and isinstance(rep[0][1], float), rep
    out = Reconstruction(tmp_path / "out" / "plane2x" / "sparse" / "0")
    cam = out.cameras[1]                                                 # :172-173 rescaled in place, then written (:363)
    assert (cam.width, cam.height) == (W, H)
    assert cam.params.tolist() == [0.9 * W, 0.9 * W, W / 2.0, H / 2.0]  # exact: the scale is a power of two
    dense = out.point_xyz[n_sparse:]
    assert len(dense) > 5000 and np.median(np.abs(dense[:, 1])) < 0.05
    # stage by stage: PIL's LANCZOS picture, the rescaled camera, our refiner + densify, the ORACLE filter
    K = np.array([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1.0]])
    refiner = dd.DepthRefiner(use_fp16=False, adaptive_correspondences=False)
    depths, Es, rgbs = [], [], []
    for i in sorted(src.images):
        im = src.images[i]; tr = truth[i - 1]
        E = im.cam_from_world().matrix()
        pic = PILImage.open(scan / "images" / im.name).convert("RGB")
        assert pic.size == (2 * W, 2 * H)
        rgbs.append(np.array(pic.resize((W, H), PILImage.Resampling.LANCZOS)))
        r = refiner.refine_depth(tr["mono"], tr["normal"], src.xyz_of(im.observed_point3D_ids()), E, K, tr["mask"])
        depths.append(r["refined_depth"]); Es.append(E)
    depths = np.stack(depths); masks = np.stack([t["mask"] for t in truth]); Es = np.stack(Es)
    Ks = np.stack([K] * len(Es))
    c = dd.unproject_views(depths, Ks, Es, mask=masks, normal=np.stack([t["normal"] for t in truth]), rgb=np.stack(rgbs)).numpy()
    culled = np.where(masks, depths, 0).astype(np.float32)
    ep, ec, _, votes = forc.filter_floaters(c["points"].astype(np.float32), c["colors"], c["normals"], culled, Ks, Es,
                                            vote_threshold=2, depth_threshold=0.7)
    assert (votes >= 2).sum() > 20
    assert len(dense) == len(ep)
    assert np.array_equal(dense.astype(np.float32), ep) and np.array_equal(out.point_rgb[n_sparse:], ec)


@pytest.mark.gpu
def test_view_sharded_scan_with_downsample_factor_and_a_late_camera(tmp_path):
    """Two ranks, ``pipeline_downsample_factor = 2``, and a second camera that only the LAST views use (all of them in
    rank 1's shard): rank 0 writes the model, so it has to rescale that camera too (``scripts/test.py:172-173`` does it
    for every processed view).  ``cameras.bin`` / ``images.bin`` / ``points3D.bin`` equal the one-process model byte
    for byte."""
    import os
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scan_factory import make_scan
    scan, cache, _ = make_scan(tmp_path / "scans", "plane2x", V=7, seed=4, floaters=0.03, second_size=(72, 104), image_scale=2, second_tail=2)
    root = Path(__file__).resolve().parent.parent
    args = ["--paths.recon-path", str(scan / "sparse" / "0"), "--paths.image-dir", str(scan / "images"),
            "--moge.cache-dir", str(cache), "--processing.downsample-density", "1", "--processing.pipeline-downsample-factor", "2",
            "--refiner.no-use-fp16", "--refiner.no-adaptive-correspondences", "--filtering.vote-threshold", "2", "--refiner.verbose", "0"]
    one = subprocess.run([sys.executable, str(root / "scripts" / "test.py"), *args, "--paths.output-model-dir", str(tmp_path / "one")],
                         capture_output=True, text=True, timeout=240)
    assert one.returncode == 0, one.stdout + one.stderr
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DD_DIST_BACKEND="gloo", DD_ALLGATHERV="broadcast")      # gloo has no CUDA send/recv
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(root / "scripts" / "test.py"), *args, "--paths.output-model-dir", str(tmp_path / "two")],
                         capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode == 0, two.stdout + two.stderr
    from depthdensifier_amd.colmap_io import Reconstruction
    m = Reconstruction(tmp_path / "two")
    assert (m.cameras[2].width, m.cameras[2].height) == (104, 72) and (m.cameras[1].width, m.cameras[1].height) == (128, 96)
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert (tmp_path / "one" / name).read_bytes() == (tmp_path / "two" / name).read_bytes(), name


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ("npz", "npy+rgb"))
def test_view_sharded_scan_matches_single_gpu(tmp_path, layout):
    """``scripts/test.py`` under torchrun: 2 ranks (sharing this box's one GPU, gloo collectives) shard the views, filter
    sharded by points, all-gatherv the surviving clouds; the model written by rank 0 equals the one-process model
    byte for byte -- with the cache read by Python threads (``npz``) and by every rank's own native prefetcher (``npy+rgb``)."""
    import os
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scan_factory import make_scan
    scan, cache, _ = make_scan(tmp_path / "scans", "plane", V=7, seed=3, floaters=0.03, second_size=(71, 103))   # two view sizes, one with an odd pixel count
    if layout == "npy+rgb":
        from PIL import Image as PILImage
        npy = scan / "cache_npy"
        npy.mkdir()
        for f in sorted(cache.glob("*.npz")):
            with np.load(f) as z:
                for k in z.files:
                    np.save(npy / f"{f.stem}_{k}.npy", z[k])
        for img in sorted((scan / "images").iterdir()):
            np.save(npy / f"{img.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))
        cache = npy
    root = Path(__file__).resolve().parent.parent
    args = ["--paths.recon-path", str(scan / "sparse" / "0"), "--paths.image-dir", str(scan / "images"),
            "--moge.cache-dir", str(cache), "--processing.downsample-density", "1", "--refiner.no-use-fp16",
            "--refiner.no-adaptive-correspondences", "--filtering.vote-threshold", "2", "--refiner.verbose", "0"]
    one = subprocess.run([sys.executable, str(root / "scripts" / "test.py"), *args, "--paths.output-model-dir", str(tmp_path / "one")],
                         capture_output=True, text=True, timeout=240)
    assert one.returncode == 0, one.stdout + one.stderr
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DD_DIST_BACKEND="gloo", DD_ALLGATHERV="broadcast")      # gloo has no CUDA send/recv
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(root / "scripts" / "test.py"), *args, "--paths.output-model-dir", str(tmp_path / "two")],
                         capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode == 0, two.stdout + two.stderr
    assert "Sharding 7 views over 2 GPUs" in two.stdout
    from depthdensifier_amd.colmap_io import Reconstruction
    n_sparse = Reconstruction(scan / "sparse" / "0").num_points3D()
    assert Reconstruction(tmp_path / "one").num_points3D() > n_sparse + 1000        # dense points were produced
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert (tmp_path / "one" / name).read_bytes() == (tmp_path / "two" / name).read_bytes(), name


def test_moge_adapter_with_a_stub_model(monkeypatch, tmp_path):
    """``MoGeSource`` feeds ``MoGeModel.infer`` a (1,3,H,W) float image in [0,1] (scripts/test.py:154-155) and returns
    squeezed depth / normal / bool mask; ``make_depth_source`` prefers the cache and explains a missing MoGe."""
    import sys
    import types
    import torch
    from depthdensifier_amd import depth_source as ds

    seen = {}

    class StubModel:
        @classmethod
        def from_pretrained(cls, ckpt):
            seen["ckpt"] = ckpt
            return cls()

        def to(self, device):
            seen["device"] = device
            return self

        def eval(self):
            return self

        def infer(self, x):
            seen["x"] = x
            _, _, h, w = x.shape
            return {"depth": torch.full((1, h, w), 2.0), "normal": torch.zeros((1, h, w, 3)), "mask": torch.ones((1, h, w))}

    pkg, model, v2 = types.ModuleType("moge"), types.ModuleType("moge.model"), types.ModuleType("moge.model.v2")
    v2.MoGeModel = StubModel
    for name, mod in (("moge", pkg), ("moge.model", model), ("moge.model.v2", v2)):
        monkeypatch.setitem(sys.modules, name, mod)
    cpu = torch.device("cpu")
    src = ds.make_depth_source(Path("models/x.pt"), None, cpu)
    assert isinstance(src, ds.MoGeSource) and seen["ckpt"] == Path("models/x.pt") and seen["device"] == cpu
    rgb = np.full((6, 8, 3), 255, np.uint8); rgb[0, 0] = (0, 51, 102)
    out = src.infer("a.png", rgb, cpu)
    x = seen["x"]
    assert x.shape == (1, 3, 6, 8) and x.dtype == torch.float32 and float(x.max()) == 1.0
    assert torch.allclose(x[0, :, 0, 0], torch.tensor([0.0, 0.2, 0.4]))
    assert out["depth"].shape == (6, 8) and out["normal"].shape == (6, 8, 3) and out["mask"].dtype == torch.bool
    assert src.prepare("a.png", rgb) is None                              # nothing to read ahead for a live model

    # dump_cache writes what CachedSource reads back (.npz), and the per-map .npy flavour is accepted too
    from PIL import Image as PILImage
    (tmp_path / "imgs").mkdir()
    PILImage.fromarray(rgb).save(tmp_path / "imgs" / "a.png")
    for layout in ("npz", "npy"):
        cdir = tmp_path / f"cache_{layout}"
        assert ds.dump_cache(src, tmp_path / "imgs", cdir, cpu, layout=layout) == 1
        assert (cdir / ("a.npz" if layout == "npz" else "a_depth.npy")).exists()
        cached = ds.make_depth_source(Path("models/x.pt"), cdir, cpu)
        assert isinstance(cached, ds.CachedSource)
        back = cached.infer("a.png", rgb, cpu)
        assert torch.equal(back["depth"], out["depth"]) and torch.equal(back["mask"], out["mask"]) and back["normal"].shape == (6, 8, 3)
    np.save(cached.dir / "b_depth.npy", np.full((6, 8), 3.0, np.float32))
    only_depth = cached.infer("b.jpg", rgb, cpu)
    assert float(only_depth["depth"][0, 0]) == 3.0 and only_depth["normal"] is None and bool(only_depth["mask"].all())
    with pytest.raises(FileNotFoundError, match="c_depth.npy"):
        cached.infer("c.png", rgb, cpu)
    for name in ("moge", "moge.model", "moge.model.v2"):
        monkeypatch.setitem(sys.modules, name, None)                      # import now fails
    with pytest.raises(ImportError, match="cache-dir"):
        ds.make_depth_source(Path("models/x.pt"), None, cpu)
    with pytest.raises(FileNotFoundError):
        ds.CachedSource(tmp_path / "nope")


def test_batch_cache_dir_placeholder(tmp_path):
    """``moge.cache_dir`` with ``{scan}`` is resolved per scan folder; a plain path is shared as given."""
    import run_batch
    for name in ("a", "b"):
        (tmp_path / "scans" / name / "images").mkdir(parents=True)
        (tmp_path / "scans" / name / "sparse" / "0").mkdir(parents=True)
    seen = []
    cfg = run_batch.BatchConfig(tmp_path / "scans", tmp_path / "out")
    cfg.config.moge.cache_dir = tmp_path / "scans" / "{scan}" / "moge_cache"
    run_batch.main(cfg, run_scan=lambda c: seen.append((c.paths.recon_path.parent.parent.name, c.moge.cache_dir)))
    assert seen == [("a", tmp_path / "scans" / "a" / "moge_cache"), ("b", tmp_path / "scans" / "b" / "moge_cache")]
    seen.clear()
    cfg.config.moge.cache_dir = tmp_path / "shared"
    run_batch.main(cfg, run_scan=lambda c: seen.append(c.moge.cache_dir))
    assert seen == [tmp_path / "shared"] * 2


def test_colmap_text_model(tmp_path):
    """The text flavour of a COLMAP model reads to the same reconstruction as its binary twin."""
    from depthdensifier_amd.colmap_io import CAMERA_MODELS, Reconstruction, load_colmap_model
    from scan_factory import make_scan
    scan, _, _ = make_scan(tmp_path, "s", V=2)
    b = Reconstruction(scan / "sparse" / "0")
    t = tmp_path / "txt"; t.mkdir()
    with open(t / "cameras.txt", "w") as f:
        f.write("# Camera list\n")
        for c in b.cameras.values():
            f.write(f"{c.camera_id} {CAMERA_MODELS[c.model_id][0]} {c.width} {c.height} " + " ".join(repr(float(p)) for p in c.params) + "\n")
    with open(t / "images.txt", "w") as f:
        f.write("# Image list with two lines of data per image\n")
        for im in b.images.values():
            f.write(f"{im.image_id} " + " ".join(repr(float(v)) for v in [*im.qvec, *im.tvec]) + f" {im.camera_id} {im.name}\n")
            f.write(" ".join(f"{repr(float(x))} {repr(float(y))} {int(p)}" for (x, y), p in zip(im.xys, im.point3D_ids)) + "\n")
    with open(t / "points3D.txt", "w") as f:
        for i, pid in enumerate(b.point_ids):
            f.write(f"{int(pid)} " + " ".join(repr(float(v)) for v in b.point_xyz[i]) + " " + " ".join(str(int(v)) for v in b.point_rgb[i])
                    + f" {repr(float(b.point_error[i]))}\n")
    r = load_colmap_model(t)
    assert np.array_equal(r.point_ids, b.point_ids) and np.array_equal(r.point_xyz, b.point_xyz) and np.array_equal(r.point_rgb, b.point_rgb)
    assert set(r.images) == set(b.images)
    for k in b.images:
        assert r.images[k].name == b.images[k].name and np.array_equal(r.images[k].qvec, b.images[k].qvec)
        assert np.array_equal(r.images[k].point3D_ids, b.images[k].point3D_ids) and np.array_equal(r.images[k].xys, b.images[k].xys)
    assert np.array_equal(r.cameras[1].params, b.cameras[1].params) and r.cameras[1].model_name == "PINHOLE"
    with pytest.raises(FileNotFoundError):
        Reconstruction(tmp_path)


def test_staging_slot_falls_back_to_ordinary_memory_without_budget(monkeypatch):
    """The pinned staging slots of the pipeline's ingest hold at most DD_PINNED_BUDGET_MB of page-locked memory; beyond it
    a slot keeps its maps in ordinary memory (same contents, the upload is simply synchronous again)."""
    import torch
    from depthdensifier_amd.depth_source import StagingSlot
    monkeypatch.setattr(StagingSlot, "budget", 0)
    s = StagingSlot()
    a = s.put("depth", np.arange(12, dtype=np.float16).reshape(3, 4))
    assert a.dtype == np.float16 and a[2, 3] == 11 and not torch.from_numpy(a).is_pinned()
    m = s.put("mask", np.array([[True, False]]))
    assert m.dtype == np.bool_ and m.tolist() == [[True, False]]
    again = s.put("depth", np.zeros((3, 4), np.float16))
    assert again.ctypes.data == a.ctypes.data                   # same buffer reused for the next view of that shape
    s.wait()                                                     # nothing pending: returns at once


def test_cached_source_prefers_a_nested_layout(tmp_path):
    """Image names with sub-folders: ``<cache>/cam1/0001_depth.npy`` is used for ``cam1/0001.jpg`` when it exists (two
    cameras' 0001 must not collide); without it the flat ``<cache>/0001_depth.npy`` is read as before."""
    from depthdensifier_amd.depth_source import CachedSource
    rgb = np.zeros((2, 3, 3), np.uint8)
    (tmp_path / "cam1").mkdir()
    np.save(tmp_path / "0001_depth.npy", np.full((2, 3), 1.0, np.float32))
    np.save(tmp_path / "cam1" / "0001_depth.npy", np.full((2, 3), 7.0, np.float32))
    src = CachedSource(tmp_path)
    assert src.prepare("cam1/0001.jpg", rgb)["depth"][0, 0] == 7.0
    assert src.prepare("cam2/0001.jpg", rgb)["depth"][0, 0] == 1.0
    assert src.prepare("0001.jpg", rgb)["depth"][0, 0] == 1.0
    with pytest.raises(FileNotFoundError):
        src.prepare("cam1/0002.jpg", rgb)


@pytest.mark.gpu
def test_cached_image_gives_the_same_model(tmp_path):
    """dump_cache(with_rgb=True) / <stem>_rgb.npy: the pipeline takes the decoded image from the cache instead of decoding the
    file -- same model, byte for byte; an image cached at another size is ignored."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image as PILImage
    from scan_factory import make_scan
    from depthdensifier_amd import pipeline as P
    scan, cache, _ = make_scan(tmp_path, "s", V=5, H=72, W=96, seed=4)
    npy = scan / "cache_npy"
    npy.mkdir()
    for f in sorted(cache.glob("*.npz")):
        with np.load(f) as z:
            for k in z.files:
                np.save(npy / f"{f.stem}_{k}.npy", z[k])
    outs = []
    for with_rgb in (False, True, "wrong size"):
        if with_rgb:
            for img in sorted((scan / "images").iterdir()):
                a = np.array(PILImage.open(img).convert("RGB"))
                np.save(npy / f"{img.stem}_rgb.npy", a if with_rgb is True else a[:10])
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=tmp_path / f"out_{with_rgb}")
        cfg.moge.cache_dir = npy
        cfg.processing.downsample_density = 1
        cfg.refiner.verbose = 0
        cfg.refiner.adaptive_correspondences = False
        P.main(cfg)
        outs.append((tmp_path / f"out_{with_rgb}" / "points3D.bin").read_bytes())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("early_exit", (False, True))
@pytest.mark.parametrize("density", (1, 4))
def test_resident_stacks_give_the_same_model(tmp_path, monkeypatch, early_exit, density):
    """Round 6: the views' maps are uploaded straight into resident group stacks that are reused every third launch group
    (``pipeline._GroupRing``).  Ten views in groups of two -- the stacks wrap around -- must give the model of the run without them, byte
    for byte: with the fused launch (density 1), the per-view launches (density 4), and with every view an early exit of the refiner
    (the refiner hands its INPUT map back, a view of a stack: the filter's cache must keep a copy, not the view)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image as PILImage
    from scan_factory import make_scan
    from depthdensifier_amd import pipeline as P
    scan, cache, _ = make_scan(tmp_path, "s", V=10, H=72, W=96, seed=9)
    npy = scan / "cache_npy"
    npy.mkdir()
    for f in sorted(cache.glob("*.npz")):
        with np.load(f) as z:
            for k in z.files:
                np.save(npy / f"{f.stem}_{k}.npy", z[k])
    for img in sorted((scan / "images").iterdir()):
        np.save(npy / f"{img.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))
    outs = []
    for ring in ("1", "0"):
        monkeypatch.setenv("DD_GROUP_RING", ring)
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=tmp_path / f"out_{ring}")
        cfg.moge.cache_dir = npy
        cfg.processing.downsample_density = density
        cfg.processing.views_per_launch = 2
        cfg.refiner.verbose = 0
        cfg.refiner.adaptive_correspondences = False
        if early_exit:
            cfg.refiner.min_correspondences = 10 ** 6          # "Too few correspondences": every view comes back unrefined
        rep = P.main(cfg)
        assert rep["views"] == 10
        outs.append((tmp_path / f"out_{ring}" / "points3D.bin").read_bytes())
    assert outs[0] == outs[1] and len(outs[0]) > 1000


@pytest.mark.gpu
def test_a_scan_that_raises_leaves_the_process_fit_for_the_next(tmp_path, monkeypatch):
    """scripts/run_batch.py runs scan after scan in one process and goes on when one fails.  A scan that raises in the middle of its image
    loop -- views read ahead in the native prefetcher's slots, uploads in flight on the copy stream -- must leave nothing behind that the next
    scan trips over: its feeder is closed in a ``finally`` (the slots are free), and the next scan's model is the model of a fresh process."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image as PILImage
    from scan_factory import make_scan
    from depthdensifier_amd import pipeline as P
    from depthdensifier_amd import depth_source as DS
    from depthdensifier_amd.depth_refiner import DepthRefiner
    scan, cache, _ = make_scan(tmp_path, "s", V=12, H=72, W=96, seed=11)
    npy = scan / "cache_npy"
    npy.mkdir()
    for f in sorted(cache.glob("*.npz")):
        with np.load(f) as z:
            for k in z.files:
                np.save(npy / f"{f.stem}_{k}.npy", z[k])
    for img in sorted((scan / "images").iterdir()):
        np.save(npy / f"{img.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))

    def config(tag):
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=tmp_path / f"out_{tag}")
        cfg.moge.cache_dir = npy
        cfg.processing.downsample_density = 1
        cfg.processing.views_per_launch = 2
        cfg.refiner.verbose = 0
        cfg.refiner.adaptive_correspondences = False
        return cfg

    P.main(config("clean"))
    want = (tmp_path / "out_clean" / "points3D.bin").read_bytes()
    real, calls = DepthRefiner.begin_refine, {"n": 0}

    def failing(self, *a, **kw):
        calls["n"] += 1
        if calls["n"] == 6:
            raise RuntimeError("the sixth view of this scan cannot be refined")
        return real(self, *a, **kw)

    monkeypatch.setattr(DepthRefiner, "begin_refine", failing)
    with pytest.raises(RuntimeError, match="sixth view"):
        P.main(config("failed"))
    monkeypatch.setattr(DepthRefiner, "begin_refine", real)
    for entry in DS._PREFETCHERS.values():                    # the failed scan's feeder is closed: no job of it is left in a slot
        user = entry[3]() if entry[3] is not None else None
        assert user is None or user._h is None
    P.main(config("after"))
    assert (tmp_path / "out_after" / "points3D.bin").read_bytes() == want and len(want) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("density", (1, 3))
def test_two_view_sizes_through_the_native_read_path(tmp_path, density, monkeypatch):
    """A scan whose views come in two sizes (two cameras, one with an odd pixel count): no resident stacks (they hold one size), but the
    native prefetcher and the one-call uploads still carry every view -- slots sized for the larger, a job per view in its own shape.
    The model must be the model of the Python read path (the .npz cache), byte for byte."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image as PILImage
    from scan_factory import make_scan
    from depthdensifier_amd import pipeline as P
    scan, cache, _ = make_scan(tmp_path, "s", V=9, H=72, W=96, seed=21, second_size=(71, 103))
    npy = scan / "cache_npy"
    npy.mkdir()
    for f in sorted(cache.glob("*.npz")):
        with np.load(f) as z:
            for k in z.files:
                np.save(npy / f"{f.stem}_{k}.npy", z[k])
    for img in sorted((scan / "images").iterdir()):
        np.save(npy / f"{img.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))
    from depthdensifier_amd import depth_source as DS
    taken, real_get = [], DS.NativeFeeder.get
    monkeypatch.setattr(DS.NativeFeeder, "get", lambda self, k: (taken.append(k), real_get(self, k))[1])
    outs = []
    for tag, cache_dir in (("native", npy), ("python", cache)):
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=tmp_path / f"out_{tag}")
        cfg.moge.cache_dir = cache_dir
        cfg.processing.downsample_density = density
        cfg.processing.views_per_launch = 4
        cfg.refiner.verbose = 0
        cfg.refiner.adaptive_correspondences = False
        rep = P.main(cfg)
        assert rep["views"] == 9
        outs.append((tmp_path / f"out_{tag}" / "points3D.bin").read_bytes())
    assert taken == list(range(9))                         # every view of the first run came through the native prefetcher, none of the second
    assert outs[0] == outs[1] and len(outs[0]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("refiner_fp16", (False, True))
def test_resident_stacks_with_a_half_precision_cache(tmp_path, monkeypatch, refiner_fp16):
    """A cache that holds the depth maps as float16 (``dump_cache(fp16_depth=True)``): the resident stack keeps the maps in the cache's type
    and -- when the refiner works in another precision -- a second stack in the refiner's; with both in float16 they are ONE array.  Either way
    the model of the run without stacks, byte for byte."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image as PILImage
    from scan_factory import make_scan
    from depthdensifier_amd import pipeline as P
    scan, cache, _ = make_scan(tmp_path, "s", V=8, H=72, W=96, seed=13)
    npy = scan / "cache_npy"
    npy.mkdir()
    for f in sorted(cache.glob("*.npz")):
        with np.load(f) as z:
            for k in z.files:
                np.save(npy / f"{f.stem}_{k}.npy", z[k].astype(np.float16) if k == "depth" else z[k])
    for img in sorted((scan / "images").iterdir()):
        np.save(npy / f"{img.stem}_rgb.npy", np.array(PILImage.open(img).convert("RGB")))
    outs = []
    for ring in ("1", "0"):
        monkeypatch.setenv("DD_GROUP_RING", ring)
        cfg = P.ScriptConfig()
        cfg.paths = P.PathsConfig(recon_path=scan / "sparse" / "0", image_dir=scan / "images", output_model_dir=tmp_path / f"out_{ring}")
        cfg.moge.cache_dir = npy
        cfg.processing.downsample_density = 1
        cfg.processing.views_per_launch = 3
        cfg.refiner.verbose = 0
        cfg.refiner.adaptive_correspondences = False
        cfg.refiner.use_fp16 = refiner_fp16
        rep = P.main(cfg)
        assert rep["views"] == 8
        outs.append((tmp_path / f"out_{ring}" / "points3D.bin").read_bytes())
    assert outs[0] == outs[1] and len(outs[0]) > 1000
