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

The densify block of ``scripts/test.py:203-233`` and the vote loop ``:273-332`` are inline
in ``main`` and not callable.  They are pinned twice.  (1) ``build_script_block`` /
``build_votes`` take the reference's OWN statements: the module source is parsed with ``ast`` at
run time, the statements of those two ranges are picked out of ``main``'s syntax tree (the mask
fold-in ``:194`` + ``:203-232`` up to, not including, the pycolmap call at ``:233``; ``:273-332``),
compiled as they stand and executed in a namespace holding seeded inputs and duck-typed
``camera`` / ``image`` objects -> ``script_block_small.npz``, ``votes_small.npz``.  Only arrays
land in the repository, never the statements' text.  (2) The validity/order/stride semantics
are also pinned through the package
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


from synth import make_views, pinhole_K, random_pose, refiner_case, sha  # noqa: E402  (tests/golden/synth.py)


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


# ----------------------------------------------- the reference's own inline statements (ast)

def _main_tree():
    import ast
    tree = ast.parse((REF / "scripts" / "test.py").read_text())
    return next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")


def _names(node):
    import ast
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name):
            out.add(n.id)
        elif isinstance(n, ast.Attribute):
            out.add(n.attr)
    return out


def _assigns(node, name):
    import ast
    if not isinstance(node, ast.Assign):
        return False
    return any(name in {getattr(e, "id", None) for e in ([t] if not isinstance(t, ast.Tuple) else t.elts)} for t in node.targets)


def _compile(stmts, label):
    import ast
    mod = ast.Module(body=list(stmts), type_ignores=[])
    ast.fix_missing_locations(mod)
    return compile(mod, f"<reference scripts/test.py {label}>", "exec")


def _densify_block_code():
    """``refined_depth[~moge_mask] = 0`` (:194) and the densify statements :203-232 of the per-image loop, minus the
    [DEBUG] twin on the unrefined depth (:223-227, it calls pycolmap's Rigid3d) -- stops before :233."""
    import ast
    main = _main_tree()
    loop = next(n for n in ast.walk(main) if isinstance(n, ast.For) and getattr(n.target, "id", None) == "image")
    body = loop.body
    fold = next(st for st in body if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Subscript)
                and getattr(st.targets[0].value, "id", None) == "refined_depth" and "moge_mask" in _names(st))
    first = next(i for i, st in enumerate(body) if _assigns(st, "h") and "refined_depth" in _names(st))
    last = next(i for i, st in enumerate(body) if _assigns(st, "points3D_camera"))
    block = [st for st in body[first:last + 1] if not any("unrefined" in n for n in _names(st))]
    assert fold.lineno == 194 and block[0].lineno == 205 and block[-1].lineno == 232, (fold.lineno, block[0].lineno, block[-1].lineno)
    assert not any("inverse" in _names(st) for st in block)
    return _compile([fold] + block, "194+203-232")


def _vote_block_code():
    """``floater_votes = np.zeros(...)`` (:273) through ``final_colors = final_colors[points_to_keep_mask]`` (:332)."""
    import ast
    main = _main_tree()
    branch = next(n for n in ast.walk(main) if isinstance(n, ast.If) and getattr(n.test, "id", None) == "all_dense_points")
    body = branch.body
    first = next(i for i, st in enumerate(body) if _assigns(st, "floater_votes"))
    last = next(i for i, st in enumerate(body) if _assigns(st, "final_colors") and "points_to_keep_mask" in _names(st))
    block = body[first:last + 1]
    assert block[0].lineno == 273 and block[-1].lineno == 332, (block[0].lineno, block[-1].lineno)
    return _compile(block, "273-332")


def run_reference_densify_block(depth, mask, normal, rgb, params, stride):
    """One pass of the reference's own statements over one view; returns what they leave behind."""
    import time
    import types as _t
    ns = dict(np=np, time=time, refined_depth=np.array(depth, copy=True), moge_mask=np.asarray(mask, bool), moge_normal=normal,
              pil_image_rescaled=rgb, camera=_Cam(params), unproject_points=REF_SCRIPT.unproject_points,
              config=_t.SimpleNamespace(processing=_t.SimpleNamespace(downsample_density=int(stride))),
              refiner_config={"verbose": 0})
    with np.errstate(invalid="ignore", over="ignore"):
        exec(_densify_block_code(), ns)
    return {k: ns[k] for k in ("pixels_x_valid", "pixels_y_valid", "colors", "normals", "depth_values", "points3D_camera")}


def build_script_block():
    """script_block_small.npz: the densify block's own outputs at strides 1, 3, 32 (two ragged views with special
    depths, one f16 view stack, one identity-pose stack so that camera frame == world frame)."""
    g = {}
    cases = [("p", 21, 2, 23, 37, (41.5, 43.25, 18.0, 11.5), True, np.float32),
             ("q", 22, 2, 40, 72, (80.0, 82.0, 36.0, 20.0), False, np.float16),
             ("r", 23, 1, 67, 131, (150.0, 149.0, 65.5, 33.5), True, np.float32)]
    for name, seed, V, H, W, params, specials, dt in cases:
        d = make_views(seed, V, H, W, specials=specials, depth_dtype=dt)
        for k, v in d.items():
            g[f"{name}_in_{k}"] = v
        g[f"{name}_in_params"] = np.asarray(params, np.float64)
        g[f"{name}_in_strides"] = np.asarray((1, 3, 32), np.int64)
        for s in (1, 3, 32):
            outs = [run_reference_densify_block(d["depth"][v], d["mask"][v], d["normal"][v], d["rgb"][v], params, s) for v in range(V)]
            g[f"{name}_exp_s{s}_counts"] = np.asarray([len(o["pixels_x_valid"]) for o in outs], np.int64)
            for key in ("pixels_x_valid", "pixels_y_valid", "colors", "normals", "depth_values", "points3D_camera"):
                g[f"{name}_exp_s{s}_{key}__test_py_203_232"] = np.concatenate([o[key] for o in outs])
    np.savez_compressed(OUT / "script_block_small.npz", **g)
    return g


class _ImageP(_Image):
    """``project_points`` calls ``cam_from_world().matrix()``; the vote loop also ``projection_center()``
    (pycolmap: the camera centre ``-R^T t``; an input to the statements, not part of them)."""

    def projection_center(self):
        E = np.asarray(self._E, np.float64)
        return -(E[:3, :3].T @ E[:3, 3])


def votes_scene(seed=5, V=8, H=48, W=64, n_extra=300):
    """Cameras on a ring INSIDE a sphere of radius 4, looking across its centre: every ray ends on the far wall, so the
    depth maps are dense (own synthetic code: ray/sphere intersections).  The cloud = wall points seen by the views
    (consistent: no votes) + floaters around the centre, in front of the wall for every camera (votes from each view
    whose direction their normal faces: upward normals face all cameras, random ones about half) + random points
    anywhere, some outside the wall / behind cameras.  Points are float32-representable (the HIP kernel takes
    float32 rows)."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 40.0, W / 2.0, H / 2.0
    K = np.array([[f, 0, cx], [0, f + 1.5, cy], [0, 0, 1.0]])
    Es, depths, pts, nrm = [], [], [], []
    ys, xs = np.mgrid[0:H, 0:W]
    for v in range(V):
        a = 2 * np.pi * v / V
        c = np.array([2.0 * np.cos(a), 0.8 + 0.2 * np.sin(2 * a), 2.0 * np.sin(a)])
        z = -c / np.linalg.norm(c)
        x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        E = np.hstack([R, (-R @ c)[:, None]])
        rays = np.stack([(xs - cx) / K[0, 0], (ys - cy) / K[1, 1], np.ones_like(xs, float)], -1)
        dirs = rays @ R                                            # world directions of the unit-z rays
        b = dirs @ c
        aa = np.sum(dirs * dirs, -1)
        t = (-b + np.sqrt(b * b - aa * (c @ c - 16.0))) / aa       # far root: the camera is inside the sphere
        depth = t.astype(np.float32)                               # z-depth: the ray's z component is 1
        depth[rng.uniform(size=depth.shape) < 0.05] = 0.0          # holes, like a folded mask
        Es.append(E); depths.append(depth)
        sel = (depth > 0) & (rng.uniform(size=depth.shape) < 0.06)
        p = c + dirs[sel] * depth[sel][:, None].astype(np.float64)
        pts.append(p); nrm.append(-p / np.linalg.norm(p, axis=1, keepdims=True))
    n_fl = 500
    fl = rng.standard_normal((n_fl, 3)); fl *= (0.7 * rng.uniform(size=n_fl) ** (1 / 3) / np.linalg.norm(fl, axis=1))[:, None]
    up = np.tile([0.0, 1.0, 0.0], (n_fl, 1)) + 0.2 * rng.standard_normal((n_fl, 3))
    rnd = rng.standard_normal((n_fl, 3))
    nf = np.where((np.arange(n_fl) % 2 == 0)[:, None], up, rnd)
    pts.append(fl); nrm.append(nf / np.linalg.norm(nf, axis=1, keepdims=True))
    extra = rng.standard_normal((n_extra, 3)) * 3.0
    pts.append(extra); n = rng.standard_normal((n_extra, 3)); nrm.append(n / np.linalg.norm(n, axis=1, keepdims=True))
    points = np.concatenate(pts).astype(np.float32)
    normals = np.concatenate(nrm).astype(np.float32)
    colors = rng.integers(0, 256, size=(len(points), 3), dtype=np.uint8)
    return dict(points=points, normals=normals, colors=colors, depth=np.stack(depths), K=np.stack([K] * V), cam_from_world=np.stack(Es))


def run_reference_vote_block(points64, normals, colors, depth, K, E, depth_threshold, vote_threshold):
    import types as _t
    cached = {i + 1: {"refined_depth": depth[i], "image": _ImageP(E[i]), "camera": _CamK(K[i])} for i in range(len(depth))}
    ns = dict(np=np, tqdm=lambda it, **kw: it, project_points=REF_SCRIPT.project_points, cached_refinement_data=cached,
              final_point_cloud=points64, final_normals=normals, final_colors=colors,
              config=_t.SimpleNamespace(filtering=_t.SimpleNamespace(depth_threshold=depth_threshold, vote_threshold=vote_threshold)))
    with np.errstate(divide="ignore", invalid="ignore"):
        exec(_vote_block_code(), ns)
    return ns["floater_votes"], ns["points_to_keep_mask"], ns["final_point_cloud"], ns["final_colors"]


def build_votes():
    """votes_small.npz: floater_votes / keep mask / filtered points+colours left behind by the reference's own vote
    loop (scripts/test.py:273-332) on the seeded sphere scene, at the default thresholds and at (0.9, 2)."""
    g = {}
    sc = votes_scene()
    for k, v in sc.items():
        g[f"in_{k}"] = v
    for tag, dthr, vthr in (("default", 0.7, 5), ("tight", 0.9, 2)):
        votes, keep, pts, cols = run_reference_vote_block(sc["points"].astype(np.float64), sc["normals"], sc["colors"], sc["depth"],
                                                          sc["K"], sc["cam_from_world"], dthr, vthr)
        g[f"{tag}_in_thresholds"] = np.asarray([dthr, vthr], np.float64)
        g[f"{tag}_exp_floater_votes__test_py_273_332"] = np.asarray(votes, np.int64)
        g[f"{tag}_exp_keep_mask__test_py_273_332"] = np.asarray(keep, bool)
        g[f"{tag}_exp_points__test_py_273_332"] = pts
        g[f"{tag}_exp_colors__test_py_273_332"] = cols
    np.savez_compressed(OUT / "votes_small.npz", **g)
    return g


def build_colors():
    """colors_small.npz: the colour branch of ``_depth_to_pointcloud`` (src/depthdensifier/visualizer.py:337-342) and what
    ``add_rgbd_pointcloud`` stores, for the image types the uint8 fixtures do not reach:
      f01      float32 image in [0, 1]                      -> x255, uint8
      f255     float32 image in [0, 255]                    -> kept as float32
      fmasked  float32 image in [0, 1] whose only values above 1 lie under the mask's holes -> x255, uint8 (the test is on the VALID colours)
      u01      uint8 image with maximum 1                   -> x255, uint8
      f64      float64 image in [0, 1]                      -> x255 in float64, uint8
      nomask   float32 image in [0, 1], no mask: validity = depth > 0, colours of the pixels with depth <= 0 above 1
    Every case: one 24 x 32 view, skewed K, random pose."""
    rng = np.random.default_rng(77)
    H, W = 24, 32
    K = np.array([[30.0, 0.4, 15.5], [0.0, 31.0, 11.5], [0.0, 0.0, 1.0]])
    E = random_pose(rng)
    depth = rng.uniform(0.5, 4.0, (H, W)).astype(np.float32)
    depth[rng.uniform(size=(H, W)) < 0.15] = 0.0
    mask = rng.uniform(size=(H, W)) < 0.7
    base = rng.uniform(0.0, 1.0, (H, W, 3))
    f01 = base.astype(np.float32)
    f255 = (base * 255.0).astype(np.float32)
    fmasked = f01.copy(); fmasked[~mask] = 7.5
    u01 = (base > 0.5).astype(np.uint8)
    f64 = base.copy()
    nomask = f01.copy(); nomask[depth <= 0] = 3.0
    g = dict(in_K=K, in_cam_from_world=E, in_depth=depth, in_mask=mask)
    for name, img, m in (("f01", f01, mask), ("f255", f255, mask), ("fmasked", fmasked, mask), ("u01", u01, mask), ("f64", f64, mask),
                         ("nomask", nomask, None)):
        pts, cols = VIZ._depth_to_pointcloud(depth, K, E, img, m)
        viz = COLMAPVisualizer()
        ret = viz.add_rgbd_pointcloud(depth, img, K, E, m, None)
        assert np.array_equal(ret, pts) and np.array_equal(viz.point_clouds[0].colors, cols)
        g[f"{name}_in_rgb"] = img
        g[f"{name}_exp_points__depth_to_pointcloud"] = pts
        g[f"{name}_exp_colors__depth_to_pointcloud"] = cols
    np.savez_compressed(OUT / "colors_small.npz", **g)
    return g


ALL_FIXTURES = ("densify_small.npz", "densify_vga.npz", "filter_small.npz", "refiner_small.npz", "script_block_small.npz",
                "votes_small.npz", "colors_small.npz")


def build_all():
    a = build_small()
    b = build_vga()
    build_filter()
    build_refiner()
    build_script_block()
    build_votes()
    build_colors()
    return a, b


if __name__ == "__main__":
    a = build_small()
    b = build_vga()
    build_filter()
    build_refiner()
    build_script_block()
    v = build_votes()
    c = build_colors()
    print("colour cases:", {k.split("_exp_")[0]: str(v_.dtype) for k, v_ in c.items() if "_exp_colors" in k})
    for f in ALL_FIXTURES:
        print(f, (OUT / f).stat().st_size, "bytes")
    for tag in ("default", "tight"):
        vv = v[f"{tag}_exp_floater_votes__test_py_273_332"]
        print(f"votes[{tag}]: n={len(vv)} histogram={np.bincount(vv).tolist()} kept={int(v[f'{tag}_exp_keep_mask__test_py_273_332'].sum())}")
    print("keys:", len(a), len(b))
