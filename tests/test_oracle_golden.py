"""The oracle (oracle/densify_oracle.py) against vectors captured from the reference.

Every expected array in tests/golden/*.npz was returned by a function imported
from the reference (see tests/golden/make_goldens.py; key suffixes name it).
Float64 tolerance: 1e-12 relative to the scene scale (BLAS summation order may
differ between the reference's matmul formulation and ours).
"""

import numpy as np
import pytest

from oracle import densify_oracle as orc

CASES = ("a", "b", "c")
RTOL = 1e-12


def _close(got, exp, scale=None):
    assert got.shape == exp.shape
    fin = np.isfinite(exp)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.array_equal(got[~fin & ~np.isnan(exp)], exp[~fin & ~np.isnan(exp)])
    if fin.any():
        s = scale if scale is not None else max(1.0, float(np.abs(exp[fin]).max()))
        assert np.abs(got[fin] - exp[fin]).max() <= RTOL * s


def _inputs(g, c):
    return {k: g[f"{c}_in_{k}"] for k in ("depth", "mask", "normal", "rgb", "conf", "cam_from_world", "params", "strides", "Kskew")}


@pytest.mark.parametrize("c", CASES)
def test_script_semantics_all_strides(golden_small, c):
    i = _inputs(golden_small, c)
    V = i["depth"].shape[0]
    for s in i["strides"]:
        cloud = orc.densify_scene_script(i["depth"], np.tile(i["params"], (V, 1)), i["cam_from_world"],
                                         mask=i["mask"], normal=i["normal"], rgb=i["rgb"], stride=int(s))
        assert np.array_equal(np.diff(cloud.view_offsets), golden_small[f"{c}_exp_script_s{s}_counts"])
        with np.errstate(invalid="ignore"):
            _close(cloud.points, golden_small[f"{c}_exp_script_s{s}_points__depth_to_pointcloud"])
        assert np.array_equal(cloud.colors, golden_small[f"{c}_exp_script_s{s}_colors__depth_to_pointcloud"])
        # normals are a pass-through gather of the camera-frame map (test.py:220)
        W = i["depth"].shape[2]
        vi = cloud.view_index
        exp_n = i["normal"][vi, cloud.pixel_index // W, cloud.pixel_index % W]
        assert np.array_equal(cloud.normals, exp_n)


@pytest.mark.parametrize("c", CASES)
def test_confidence_cull(golden_small, c):
    i = _inputs(golden_small, c)
    V = i["depth"].shape[0]
    cloud = orc.densify_scene_script(i["depth"], np.tile(i["params"], (V, 1)), i["cam_from_world"],
                                     mask=i["mask"], stride=1, conf=i["conf"], conf_threshold=0.5)
    assert np.array_equal(np.diff(cloud.view_offsets), golden_small[f"{c}_exp_conf_s1_counts"])
    with np.errstate(invalid="ignore"):
        _close(cloud.points, golden_small[f"{c}_exp_conf_s1_points__depth_to_pointcloud"])


@pytest.mark.parametrize("c", CASES)
def test_package_formulation(golden_small, c):
    i = _inputs(golden_small, c)
    V = i["depth"].shape[0]
    d32 = i["depth"].astype(np.float32)
    K = np.tile(i["Kskew"], (V, 1, 1))
    with np.errstate(invalid="ignore", over="ignore"):
        cloud = orc.densify_scene_viz(d32, K, i["cam_from_world"], mask=i["mask"], normal=i["normal"], rgb=i["rgb"])
        nomask = orc.densify_scene_viz(d32, K, i["cam_from_world"])
    assert np.array_equal(np.diff(cloud.view_offsets), golden_small[f"{c}_exp_viz_counts"])
    _close(cloud.points, golden_small[f"{c}_exp_viz_points__depth_to_pointcloud"])
    assert np.array_equal(cloud.colors, golden_small[f"{c}_exp_viz_colors__depth_to_pointcloud"])
    _close(cloud.normals, golden_small[f"{c}_exp_viz_normals__transform_normals"], scale=1.0)
    assert np.array_equal(np.diff(nomask.view_offsets), golden_small[f"{c}_exp_viznomask_counts"])
    _close(nomask.points, golden_small[f"{c}_exp_viznomask_points__depth_to_pointcloud"])


@pytest.mark.parametrize("c", CASES)
def test_unproject_points_bit_exact(golden_small, c):
    """scripts/test.py:79-90 is pure NumPy element-wise math: the restatement must match bit for bit."""
    g = golden_small
    px, py = g[f"{c}_in_unproj_px"], g[f"{c}_in_unproj_py"]
    d = g[f"{c}_in_depth"][0][py, px]
    with np.errstate(invalid="ignore", over="ignore"):
        got = orc.unproject_pinhole(px, py, d, g[f"{c}_in_params"])
    exp = g[f"{c}_exp_unproj__unproject_points"]
    assert got.dtype == exp.dtype == np.float64
    assert np.array_equal(got, exp, equal_nan=True)


def test_edge_views(golden_small):
    """Case c: view 0 has no valid pixel, view 1 exactly one."""
    assert list(golden_small["c_exp_script_s1_counts"]) == [0, 1]
    i = _inputs(golden_small, "c")
    cloud = orc.densify_scene_script(i["depth"], np.tile(i["params"], (2, 1)), i["cam_from_world"], mask=i["mask"])
    assert list(cloud.view_offsets) == [0, 0, 1]
    assert list(cloud.pixel_index) == [3 * 7 + 4]


@pytest.mark.parametrize("kname", ("ident", "real"))
@pytest.mark.parametrize("tag", ("dense", "masked"))
@pytest.mark.parametrize("s", (1, 32))
def test_vga_config1(golden_vga, vga_inputs, kname, tag, s):
    """BASELINE config 1: 4 synthetic 640x480 views, identity K, random poses, dense depth."""
    g, d = golden_vga, vga_inputs
    V = d["depth"].shape[0]
    sub = int(g["sub"])
    key = f"{kname}_{tag}_s{s}"
    cloud = orc.densify_scene_script(d["depth"], np.tile(g[f"{kname}_params"], (V, 1)), d["cam_from_world"],
                                     mask=None if tag == "dense" else d["mask"], rgb=d["rgb"], stride=s)
    assert np.array_equal(np.diff(cloud.view_offsets), g[f"{key}_counts"])
    _close(cloud.points[::sub], g[f"{key}_points_sub__depth_to_pointcloud"])
    from synth import sha
    assert sha(cloud.colors) == bytes(g[f"{key}_colors_sha__depth_to_pointcloud"]).hex()


@pytest.mark.parametrize("kname", ("ident", "real"))
def test_vga_package_formulation(golden_vga, vga_inputs, kname):
    g, d = golden_vga, vga_inputs
    V = d["depth"].shape[0]
    sub = int(g["sub"])
    fx, fy, cx, cy = g[f"{kname}_params"]
    K = np.tile(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]), (V, 1, 1))
    cloud = orc.densify_scene_viz(d["depth"], K, d["cam_from_world"], mask=d["mask"], normal=d["normal"])
    _close(cloud.points[::sub], g[f"{kname}_viz_points_sub__depth_to_pointcloud"])
    _close(cloud.normals[::sub], g[f"{kname}_viz_normals_sub__transform_normals"], scale=1.0)


def test_two_formulations_agree(vga_inputs):
    """script block and package formulation are the same map on equal validity."""
    d = vga_inputs
    params = np.array([500.0, 510.0, 320.0, 240.0])
    a = orc.densify_view_script(d["depth"][0], params, d["cam_from_world"][0], mask=d["mask"][0])
    K = np.array([[500.0, 0, 320], [0, 510.0, 240], [0, 0, 1]])
    b, _, lin = orc.depth_to_pointcloud_viz(d["depth"][0], K, d["cam_from_world"][0], None, d["mask"][0])
    assert np.array_equal(a["pixel_index"], lin)
    assert np.abs(a["points"] - b).max() < 1e-12 * 10


@pytest.mark.parametrize("case", ("f01", "f255", "fmasked", "u01", "f64", "nomask"))
def test_package_colour_conversion(case):
    """The oracle's restatement of ``visualizer.py:337-342`` against the reference's own return values for float images, a
    uint8 image with maximum 1 and colours above 1 that lie under the mask's holes (``tests/golden/colors_small.npz``)."""
    from pathlib import Path
    g = np.load(Path(__file__).parent / "golden" / "colors_small.npz")
    mask = None if case == "nomask" else g["in_mask"]
    pts, cols, _ = orc.depth_to_pointcloud_viz(g["in_depth"], g["in_K"], g["in_cam_from_world"], g[f"{case}_in_rgb"], mask)
    want = g[f"{case}_exp_colors__depth_to_pointcloud"]
    assert cols.dtype == want.dtype and np.array_equal(cols, want)
    assert np.array_equal(pts, g[f"{case}_exp_points__depth_to_pointcloud"])
