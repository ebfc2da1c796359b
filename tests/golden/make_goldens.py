#!/usr/bin/env python3
"""Capture golden vectors for the depth->points path FROM THE REFERENCE ITSELF.

Runs only in the build container (needs ``/root/reference``); its outputs
(``tests/golden/*.npz``) are committed and are what travels to the GPU box.
No reference source is copied: every ``exp_*`` array below is the return value
of a function imported from the reference tree:

* ``COLMAPVisualizer._depth_to_pointcloud``  (src/depthdensifier/visualizer.py:291-344)
* ``COLMAPVisualizer._transform_normals``    (src/depthdensifier/visualizer.py:346-376)
* ``COLMAPVisualizer.add_rgbd_pointcloud``   (src/depthdensifier/visualizer.py:246-289)
* ``unproject_points``                       (scripts/test.py:79-90)
* ``project_points``                         (scripts/test.py:58-76)   -> filter_small.npz
* ``DepthRefiner.refine_depth``              (src/depthdensifier/depth_refiner.py:207-328, CPU/FP32) -> refiner_small.npz

The densify block of ``scripts/test.py:203-233`` is inline in ``main`` and not
callable.  Its validity/order/stride semantics are pinned through the package
formulation, which is the same map: the reference's
``_depth_to_pointcloud`` is fed the mask-zeroed depth sub-sampled with
``[::s, ::s]``, the mask ``depth > 0`` and the intrinsics ``diag(1/s,1/s,1) @ K``
(so that sub-sampled pixel ``u'`` back-projects along the ray of pixel
``u = s*u'``).  The key names say which reference function produced each array.

``scripts/test.py`` imports ``pycolmap``, ``moge`` and ``tyro`` at module top;
none is installed here and none is used by ``unproject_points``, so empty
placeholder modules satisfy those three import statements.  Nothing of those
packages is emulated.

Usage:  python tests/golden/make_goldens.py
"""

from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF / "src"))
sys.path.insert(0, str(OUT))

from depthdensifier.visualizer import COLMAPVisualizer  # noqa: E402  (the reference)


def _load_reference_script_helpers():
    for name in ("pycolmap", "tyro", "moge", "moge.model", "moge.model.v2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["pycolmap"].Image = object
    sys.modules["pycolmap"].Camera = object
    sys.modules["moge.model.v2"].MoGeModel = object
    spec = importlib.util.spec_from_file_location("_ref_script", REF / "scripts" / "test.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REF_SCRIPT = _load_reference_script_helpers()
VIZ = COLMAPVisualizer()


class _Cam:
    """Duck-typed camera for ``unproject_points`` (it only reads ``.params``)."""

    def __init__(self, params):
        self.params = np.asarray(params, dtype=np.float64)


from synth import make_views, pinhole_K, random_pose, sha  # noqa: E402  (tests/golden/synth.py)


# ------------------------------------------------------- reference callers

def ref_script_view(depth, mask, params, E, stride, rgb=None, conf=None, thr=None):
    """Script-semantics expected values, produced by reference functions only."""
    keep = np.asarray(mask, bool) if mask is not None else np.ones(depth.shape, bool)
    if conf is not None:
        keep = keep & (conf > thr)
    culled = depth.copy()
    culled[~keep] = 0                      # the input fed to the reference; test.py:194 semantics
    sub = np.ascontiguousarray(culled[::stride, ::stride])
    with np.errstate(invalid="ignore"):
        vmask = sub > 0
    Ks = np.diag([1.0 / stride, 1.0 / stride, 1.0]) @ pinhole_K(params)
    sub_rgb = None if rgb is None else np.ascontiguousarray(rgb[::stride, ::stride])
    with np.errstate(invalid="ignore", over="ignore"):
        pts, cols = VIZ._depth_to_pointcloud(sub, Ks, E, sub_rgb, vmask)
    return pts, cols, int(vmask.sum())


def ref_unproject(px, py, d, params):
    return REF_SCRIPT.unproject_points(np.stack([px, py], axis=-1), d, _Cam(params))


# -------------------------------------------------------------- fixtures

def build_small():
    """Small cases with full inputs stored (ragged sizes, special depths, both forks)."""
    g = {}
    cases = [
        # name, seed, V, H, W, params, strides, specials, dtype
        ("a", 11, 2, 23, 37, (41.5, 43.25, 18.0, 11.5), (1, 3, 32), True, np.float32),
        ("b", 12, 2, 16, 64, (100.0, 100.0, 32.0, 8.0), (1, 2), False, np.float16),
        ("c", 13, 2, 5, 7, (1.0, 1.0, 0.0, 0.0), (1, 4), True, np.float32),
    ]
    for name, seed, V, H, W, params, strides, specials, dt in cases:
        d = make_views(seed, V, H, W, specials=specials, depth_dtype=dt)
        if name == "c":                       # view 0: nothing valid; view 1: one valid pixel
            d["mask"][0] = False
            d["mask"][1] = False
            d["mask"][1, 3, 4] = True
            d["depth"][1, 3, 4] = 2.5
        for k, v in d.items():
            g[f"{name}_in_{k}"] = v
        g[f"{name}_in_params"] = np.asarray(params, np.float64)
        g[f"{name}_in_strides"] = np.asarray(strides, np.int64)
        for s in strides:
            pts, cols, cnt = [], [], []
            for v in range(V):
                p, c, n = ref_script_view(d["depth"][v], d["mask"][v], params, d["cam_from_world"][v], s, d["rgb"][v])
                pts.append(p); cols.append(c); cnt.append(n)
            g[f"{name}_exp_script_s{s}_points__depth_to_pointcloud"] = np.concatenate(pts)
            g[f"{name}_exp_script_s{s}_colors__depth_to_pointcloud"] = np.concatenate(cols)
            g[f"{name}_exp_script_s{s}_counts"] = np.asarray(cnt, np.int64)
        # confidence cull at thr=0.5, stride 1 (reference fed mask & conf>thr)
        pts, cnt = [], []
        for v in range(V):
            p, _, n = ref_script_view(d["depth"][v], d["mask"][v], params, d["cam_from_world"][v], 1,
                                      None, d["conf"][v], 0.5)
            pts.append(p); cnt.append(n)
        g[f"{name}_exp_conf_s1_points__depth_to_pointcloud"] = np.concatenate(pts)
        g[f"{name}_exp_conf_s1_counts"] = np.asarray(cnt, np.int64)
        # package formulation: mask-only validity + rotated normals; and no-mask (depth>0) fork
        Kskew = pinhole_K(params).copy()
        Kskew[0, 1] = 0.37                    # skew is honoured by inv(K)
        g[f"{name}_in_Kskew"] = Kskew
        vp, vc, vn, vcnt, dp, dcnt, api = [], [], [], [], [], [], []
        for v in range(V):
            dep32 = d["depth"][v].astype(np.float32)
            E = d["cam_from_world"][v]
            with np.errstate(invalid="ignore", over="ignore"):
                p, c = VIZ._depth_to_pointcloud(dep32, Kskew, E, d["rgb"][v], d["mask"][v])
                n = VIZ._transform_normals(d["normal"][v], E, d["mask"][v])
                p2, _ = VIZ._depth_to_pointcloud(dep32, Kskew, np.vstack([E, [0, 0, 0, 1]]), None, None)
                viz = COLMAPVisualizer()
                p3 = viz.add_rgbd_pointcloud(dep32, d["rgb"][v], Kskew, E, d["mask"][v], d["normal"][v])
            assert np.array_equal(p3, p, equal_nan=True)
            assert np.array_equal(viz.point_clouds[0].normals, n, equal_nan=True)
            vp.append(p); vc.append(c); vn.append(n); vcnt.append(len(p)); dp.append(p2); dcnt.append(len(p2))
        g[f"{name}_exp_viz_points__depth_to_pointcloud"] = np.concatenate(vp)
        g[f"{name}_exp_viz_colors__depth_to_pointcloud"] = np.concatenate(vc)
        g[f"{name}_exp_viz_normals__transform_normals"] = np.concatenate(vn)
        g[f"{name}_exp_viz_counts"] = np.asarray(vcnt, np.int64)
        g[f"{name}_exp_viznomask_points__depth_to_pointcloud"] = np.concatenate(dp)
        g[f"{name}_exp_viznomask_counts"] = np.asarray(dcnt, np.int64)
        # scripts/test.py:79-90 on a fixed pixel list (camera-frame points)
        rng = np.random.default_rng(seed + 100)
        px = rng.integers(0, W, size=200).astype(np.int64)
        py = rng.integers(0, H, size=200).astype(np.int64)
        dd = d["depth"][0][py, px]
        g[f"{name}_in_unproj_px"], g[f"{name}_in_unproj_py"] = px, py
        with np.errstate(invalid="ignore", over="ignore"):
            g[f"{name}_exp_unproj__unproject_points"] = ref_unproject(px, py, dd, params)
    np.savez_compressed(OUT / "densify_small.npz", **g)
    return g


VGA_SUB = 997        # keep every 997th expected point


def build_vga():
    """BASELINE config 1: 4 synthetic 640x480 views, identity K (and a realistic K),
    random poses, dense depth.  Inputs are regenerated from the seed by
    ``synth.make_views`` (tests/golden/synth.py, shared with the tests) and
    checked by SHA-256; expected outputs are stored sub-sampled."""
    g = {}
    V, H, W, seed = 4, 480, 640, 0
    d = make_views(seed, V, H, W, rho=0.8)
    for k, v in d.items():
        g[f"in_sha_{k}"] = np.frombuffer(bytes.fromhex(sha(v)), np.uint8)
    g["in_seed"] = np.int64(seed)
    g["in_shape"] = np.asarray([V, H, W], np.int64)
    g["cam_from_world"] = d["cam_from_world"]
    for kname, params in (("ident", (1.0, 1.0, 0.0, 0.0)), ("real", (500.0, 510.0, 320.0, 240.0))):
        g[f"{kname}_params"] = np.asarray(params, np.float64)
        for dense in (True, False):
            mask = None if dense else d["mask"]
            tag = "dense" if dense else "masked"
            for s in (1, 32):
                pts, cols, cnt = [], [], []
                for v in range(V):
                    p, c, n = ref_script_view(d["depth"][v], None if mask is None else mask[v], params,
                                              d["cam_from_world"][v], s, d["rgb"][v])
                    pts.append(p); cols.append(c); cnt.append(n)
                pts = np.concatenate(pts); cols = np.concatenate(cols)
                key = f"{kname}_{tag}_s{s}"
                g[f"{key}_counts"] = np.asarray(cnt, np.int64)
                g[f"{key}_points_sub__depth_to_pointcloud"] = pts[::VGA_SUB]
                g[f"{key}_colors_sha__depth_to_pointcloud"] = np.frombuffer(bytes.fromhex(sha(cols)), np.uint8)
        # package formulation at full res, masked, rotated normals
        vp, vn = [], []
        for v in range(V):
            p, _ = VIZ._depth_to_pointcloud(d["depth"][v], pinhole_K(params), d["cam_from_world"][v], None, d["mask"][v])
            n = VIZ._transform_normals(d["normal"][v], d["cam_from_world"][v], d["mask"][v])
            vp.append(p); vn.append(n)
        g[f"{kname}_viz_points_sub__depth_to_pointcloud"] = np.concatenate(vp)[::VGA_SUB]
        g[f"{kname}_viz_normals_sub__transform_normals"] = np.concatenate(vn)[::VGA_SUB]
    g["sub"] = np.int64(VGA_SUB)
    np.savez_compressed(OUT / "densify_vga.npz", **g)
    return g


class _Rigid:
    def __init__(self, E):
        self._E = np.asarray(E, np.float64)

    def matrix(self):
        return self._E


class _Image:
    """Duck-typed image for ``project_points`` (it only calls ``cam_from_world().matrix()``)."""

    def __init__(self, E):
        self._E = E

    def cam_from_world(self):
        return _Rigid(self._E)


class _CamK:
    def __init__(self, K):
        self._K = np.asarray(K, np.float64)

    def calibration_matrix(self):
        return self._K


def build_filter():
    """``project_points`` (scripts/test.py:58-76) of the reference on seeded points/poses."""
    g = {}
    rng = np.random.default_rng(77)
    V, N = 3, 600
    E = np.stack([random_pose(rng) for _ in range(V)])
    K = np.stack([pinhole_K((60.0 + v, 61.0, 32.0, 24.0)) for v in range(V)])
    pts64 = rng.standard_normal((N, 3)) * 3.0
    pts32 = pts64.astype(np.float32)
    g["in_cam_from_world"], g["in_K"], g["in_points64"], g["in_points32"] = E, K, pts64, pts32
    for tag, pts in (("f64", pts64), ("f32", pts32)):
        for v in range(V):
            with np.errstate(divide="ignore", invalid="ignore"):
                p2, d = REF_SCRIPT.project_points(pts, _Image(E[v]), _CamK(K[v]))
            g[f"exp_{tag}_v{v}_points2d__project_points"] = p2
            g[f"exp_{tag}_v{v}_depths__project_points"] = d
    np.savez_compressed(OUT / "filter_small.npz", **g)
    return g


def refiner_case(seed, H=96, W=128, n_pts=900):
    """Mono depth = distorted true depth; sparse points = true-depth unprojections + noise + outliers."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    true_depth = 3.0 + 1.2 * np.sin(xs / 17.0) + 0.8 * np.cos(ys / 11.0)
    mono = (0.35 * true_depth ** 1.15 + 0.1).astype(np.float32)        # unknown monotone distortion
    mask = rng.uniform(size=(H, W)) < 0.9
    fx, fy, cx, cy = 110.0, 112.0, W / 2.0, H / 2.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    E = random_pose(rng)
    u = rng.uniform(2, W - 3, n_pts); v = rng.uniform(2, H - 3, n_pts)
    d = true_depth[v.astype(int), u.astype(int)] * (1 + 0.01 * rng.standard_normal(n_pts))
    d[: n_pts // 20] *= rng.uniform(1.5, 3.0, n_pts // 20)            # outliers
    cam = np.stack([(u - cx) / fx * d, (v - cy) / fy * d, d], axis=-1)
    world = (cam - E[:, 3]) @ E[:, :3]                                 # R^T (p - t)
    behind = rng.standard_normal((40, 3)) * 0.2 - E[:, :3].T @ E[:, 3] - 3.0 * E[2, :3]
    return dict(depth=mono, mask=mask, points3D=np.vstack([world, behind]), cam_from_world=E, K=K)


def build_refiner():
    """``DepthRefiner.refine_depth`` (src/depthdensifier/depth_refiner.py:207-328) of the reference on
    CPU / FP32, deterministic (adaptive_correspondences=False avoids the unseeded randperm)."""
    import torch
    from depthdensifier import DepthRefiner as RefRefiner            # the reference's class
    g = {}
    variants = {
        "default": dict(),
        "nosmooth": dict(skip_smoothing=True),
        "notrobust": dict(robust=False),
        "nomask": dict(),
        "toofew": dict(min_correspondences=100000),
    }
    for i, (name, kw) in enumerate(variants.items()):
        c = refiner_case(500 + i)
        if name == "nomask":
            c["mask"] = None
        for k, v in c.items():
            if v is not None:
                g[f"{name}_in_{k}"] = v
        ref = RefRefiner(adaptive_correspondences=False, use_fp16=False, verbose=0, **kw)
        assert ref.device.type == "cpu" and ref.dtype == torch.float32
        depth_in = c["depth"].copy()
        out = ref.refine_depth(depth_in, None, c["points3D"], c["cam_from_world"][:3], c["K"], c["mask"])
        g[f"{name}_exp_refined_depth__refine_depth"] = out["refined_depth"]
        g[f"{name}_exp_num_correspondences__refine_depth"] = np.int64(out["num_correspondences"])
        g[f"{name}_exp_scale_factor__refine_depth"] = np.float64(out["scale_factor"])
        g[f"{name}_exp_outliers_removed__refine_depth"] = np.int64(out.get("outliers_removed", -1))
        g[f"{name}_exp_returns_input_object"] = np.bool_(out["refined_depth"] is depth_in)
    np.savez_compressed(OUT / "refiner_small.npz", **g)
    return g


if __name__ == "__main__":
    a = build_small()
    b = build_vga()
    build_filter()
    build_refiner()
    for f in ("densify_small.npz", "densify_vga.npz", "filter_small.npz", "refiner_small.npz"):
        print(f, (OUT / f).stat().st_size, "bytes")
    print("keys:", len(a), len(b))
