"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference goldens.

Bar (BASELINE.json north_star / SURVEY.md 8d): counts, view offsets, pixel / view indices and
colours bit-exact; pass-through normals bit-exact; rotated normals <= 1e-6 abs; xyz within
1e-4 relative, measured per point as  max|d| / max(|p_ref|_inf, scene_radius)  with
scene_radius = max |camera centre| + max finite depth (guards cancellation when |t| >> |p|).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAULT_INJECT = 2       # _lib.DD_LAB_FAULT_INJECT (include/ddcore_lab.h): an in-kernel scan gives up; `ViewBatch(lab=...)`

XYZ_RTOL = 1e-4
NORMAL_ATOL = 1e-6


@pytest.fixture(scope="module")
def dd():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd
    return depthdensifier_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import densify_oracle
    return densify_oracle


def scene_radius(cam_from_world, depth):
    E = np.asarray(cam_from_world, np.float64)
    centres = -np.einsum("vji,vj->vi", E[:, :3, :3], E[:, :3, 3])
    d = np.asarray(depth, np.float64)
    fin = np.isfinite(d)
    return float(np.abs(centres).max() + (np.abs(d[fin]).max() if fin.any() else 0.0))


def assert_xyz(got, ref, radius, rtol=XYZ_RTOL):
    assert got.shape == ref.shape
    fin = np.isfinite(ref).all(axis=1)
    # non-finite reference rows (inf / NaN depth kept by the mask-only or depth>0 rule): the fused
    # f32 formula need not reproduce which component is inf vs NaN, only that the row is not finite
    assert not np.isfinite(got[~fin]).all(axis=1).any()
    if fin.any():
        denom = np.maximum(np.abs(ref[fin]).max(axis=1), radius)
        err = np.abs(got[fin] - ref[fin]).max(axis=1) / denom
        assert err.max() <= rtol, f"xyz rel err {err.max():.3e}"


def assert_cloud(cloud, ref, radius, normals="exact"):
    c = cloud.numpy()
    assert np.array_equal(c["view_offsets"], ref.view_offsets)
    assert np.array_equal(c["pixel_index"].astype(np.int64), ref.pixel_index)
    if c["view_index"] is not None:
        assert np.array_equal(c["view_index"].astype(np.int64), ref.view_index)
    assert_xyz(c["points"], ref.points, radius)
    if ref.colors is not None:
        assert np.array_equal(c["colors"], ref.colors)
    if ref.normals is not None:
        if normals == "exact":
            assert np.array_equal(c["normals"], ref.normals, equal_nan=True)
        else:
            assert np.abs(c["normals"] - ref.normals).max() <= NORMAL_ATOL
    else:
        assert c["normals"] is None


def _inputs(g, c):
    keys = ("depth", "mask", "normal", "rgb", "conf", "cam_from_world", "params", "strides", "Kskew")
    return {k: g[f"{c}_in_{k}"] for k in keys}


# ------------------------------------------------------------------ reference goldens

# tuning words: 0 = default, 1 = generic scalar kernels, 4 = force plan + scatter in the fused call,
# 8 = single-pass look-back kernel (the fused call's default on aligned maps), 9 = single-pass generic kernel.
# With capacity=None the host layer counts first and then runs the fused call; tuning=4 -> dd_plan + dd_scatter.
TUNINGS = (0, 1, 4, 8, 9)


@pytest.mark.parametrize("c", ("a", "b", "c"))
@pytest.mark.parametrize("tuning", TUNINGS)
def test_golden_script(dd, golden_small, c, tuning):
    """HIP vs arrays returned by the reference's own _depth_to_pointcloud (see make_goldens.py)."""
    i = _inputs(golden_small, c)
    V = i["depth"].shape[0]
    rad = scene_radius(i["cam_from_world"], i["depth"])
    for s in i["strides"]:
        cloud = dd.unproject_views(i["depth"], np.tile(i["params"], (V, 1)), i["cam_from_world"], mask=i["mask"],
                                   normal=i["normal"], rgb=i["rgb"], downsample_density=int(s), tuning=tuning).numpy()
        assert np.array_equal(np.diff(cloud["view_offsets"]), golden_small[f"{c}_exp_script_s{s}_counts"])
        assert_xyz(cloud["points"], golden_small[f"{c}_exp_script_s{s}_points__depth_to_pointcloud"], rad)
        assert np.array_equal(cloud["colors"], golden_small[f"{c}_exp_script_s{s}_colors__depth_to_pointcloud"])


@pytest.mark.parametrize("c", ("a", "b", "c"))
def test_golden_conf_cull(dd, golden_small, c):
    i = _inputs(golden_small, c)
    V = i["depth"].shape[0]
    cloud = dd.unproject_views(i["depth"], np.tile(i["params"], (V, 1)), i["cam_from_world"], mask=i["mask"],
                               conf=i["conf"], conf_threshold=0.5).numpy()
    assert np.array_equal(np.diff(cloud["view_offsets"]), golden_small[f"{c}_exp_conf_s1_counts"])
    assert_xyz(cloud["points"], golden_small[f"{c}_exp_conf_s1_points__depth_to_pointcloud"],
               scene_radius(i["cam_from_world"], i["depth"]))


@pytest.mark.parametrize("c", ("a", "b", "c"))
def test_golden_package_formulation(dd, golden_small, c):
    """visualizer.py semantics: mask-only validity, skewed K, rotated + renormalised normals."""
    i = _inputs(golden_small, c)
    V = i["depth"].shape[0]
    d32 = i["depth"].astype(np.float32)
    K = np.tile(i["Kskew"], (V, 1, 1))
    rad = scene_radius(i["cam_from_world"], d32)
    cloud = dd.unproject_views(d32, K, i["cam_from_world"], mask=i["mask"], normal=i["normal"], rgb=i["rgb"],
                               semantics="viz").numpy()
    assert np.array_equal(np.diff(cloud["view_offsets"]), golden_small[f"{c}_exp_viz_counts"])
    assert_xyz(cloud["points"], golden_small[f"{c}_exp_viz_points__depth_to_pointcloud"], rad)
    assert np.array_equal(cloud["colors"], golden_small[f"{c}_exp_viz_colors__depth_to_pointcloud"])
    assert np.abs(cloud["normals"] - golden_small[f"{c}_exp_viz_normals__transform_normals"]).max() <= NORMAL_ATOL
    nomask = dd.unproject_views(d32, K, i["cam_from_world"], normal=i["normal"], semantics="viz").numpy()
    assert np.array_equal(np.diff(nomask["view_offsets"]), golden_small[f"{c}_exp_viznomask_counts"])
    assert_xyz(nomask["points"], golden_small[f"{c}_exp_viznomask_points__depth_to_pointcloud"], rad)
    assert nomask["normals"] is None          # visualizer.py:276: normals only with a mask


@pytest.mark.parametrize("case", ("f01", "f255", "fmasked", "u01", "f64", "nomask"))
def test_golden_package_colour_conversion(dd, case):
    """The colour branch of ``_depth_to_pointcloud`` (``visualizer.py:337-342``) on the image types the uint8 fixtures do not
    reach (``tests/golden/colors_small.npz``, captured from the reference): the test ``max <= 1`` is made on the VALID colours,
    x255 applies to any dtype (also uint8 with maximum 1), and colours that are not scaled keep their dtype -- through
    ``COLMAPVisualizer.add_rgbd_pointcloud``, values and dtype equal to the reference's; through ``unproject_views`` the
    uint8 outcomes equal, the float outcome refused (a fused cloud carries uint8 colours)."""
    from pathlib import Path
    from depthdensifier_amd.visualizer import COLMAPVisualizer
    g = np.load(Path(__file__).parent / "golden" / "colors_small.npz")
    depth, K, E = g["in_depth"], g["in_K"], g["in_cam_from_world"]
    mask = None if case == "nomask" else g["in_mask"]
    img = g[f"{case}_in_rgb"]
    want_p, want_c = g[f"{case}_exp_points__depth_to_pointcloud"], g[f"{case}_exp_colors__depth_to_pointcloud"]
    viz = COLMAPVisualizer()
    pts = viz.add_rgbd_pointcloud(depth, img, K, E, mask, None)
    got_c = viz.point_clouds[0].colors
    assert_xyz(pts, want_p, scene_radius(E[None], depth[None]))
    assert got_c.dtype == want_c.dtype and np.array_equal(got_c, want_c), case
    kw = dict(mask=None if mask is None else mask[None], rgb=img[None], semantics="viz")
    if want_c.dtype == np.uint8:
        cloud = dd.unproject_views(depth[None], K[None], E[None], **kw).numpy()
        assert np.array_equal(cloud["colors"], want_c)
    else:
        with pytest.raises(ValueError, match="non-uint8"):
            dd.unproject_views(depth[None], K[None], E[None], **kw)


@pytest.mark.parametrize("kname", ("ident", "real"))
@pytest.mark.parametrize("tag", ("dense", "masked"))
@pytest.mark.parametrize("s", (1, 32))
def test_vga_config1(dd, orc, golden_vga, vga_inputs, kname, tag, s):
    """BASELINE config 1 (4 synthetic 640x480 views): HIP vs reference golden and vs the oracle."""
    g, d = golden_vga, vga_inputs
    V = d["depth"].shape[0]
    params = np.tile(g[f"{kname}_params"], (V, 1))
    mask = None if tag == "dense" else d["mask"]
    cloud = dd.unproject_views(d["depth"], params, d["cam_from_world"], mask=mask, normal=d["normal"], rgb=d["rgb"],
                               downsample_density=s, view_index=True)
    rad = scene_radius(d["cam_from_world"], d["depth"])
    key = f"{kname}_{tag}_s{s}"
    c = cloud.numpy()
    assert np.array_equal(np.diff(c["view_offsets"]), g[f"{key}_counts"])
    assert_xyz(c["points"][::int(g["sub"])], g[f"{key}_points_sub__depth_to_pointcloud"], rad)
    ref = orc.densify_scene_script(d["depth"], params, d["cam_from_world"], mask=mask, normal=d["normal"],
                                   rgb=d["rgb"], stride=s)
    assert_cloud(cloud, ref, rad)


# ------------------------------------------------------------------ oracle sweeps

def _rand_case(seed, V, H, W, dtype=np.float32, rho=0.8, specials=True):
    from synth import make_views
    d = make_views(seed, V, H, W, rho=rho, specials=specials, depth_dtype=dtype)
    rng = np.random.default_rng(seed + 1)
    d["params"] = np.stack([[W * rng.uniform(0.6, 1.2), W * rng.uniform(0.6, 1.2),
                             W / 2 + rng.uniform(-3, 3), H / 2 + rng.uniform(-3, 3)] for _ in range(V)])
    return d


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 3, 5), (3, 67, 129), (2, 128, 256), (1, 255, 257), (5, 64, 64), (2, 96, 172)])
@pytest.mark.parametrize("dtype", (np.float32, np.float16))
@pytest.mark.parametrize("stride", (1, 2, 7))
@pytest.mark.parametrize("tuning", (0, 4))
def test_oracle_sweep_script(dd, orc, shape, dtype, stride, tuning):
    """Ragged sizes (lean kernels with a ragged view end at stride 1, scalar kernels otherwise), H*W % 8 == 0 sizes, both depth dtypes, strides,
    two-pass and single-pass variants."""
    V, H, W = shape
    d = _rand_case(1000 + H * W + stride, V, H, W, dtype)
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"],
                               rgb=d["rgb"], downsample_density=stride, view_index=True, tuning=tuning)
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"],
                                   rgb=d["rgb"], stride=stride)
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"], d["depth"]))


@pytest.mark.parametrize("dtype", (np.float32, np.float16))
@pytest.mark.parametrize("shape", [(67, 129), (33, 35), (1, 9)])
@pytest.mark.parametrize("tuning", (0, 4))
def test_element_aligned_inputs(dd, orc, shape, dtype, tuning):
    """Device tensors whose base pointers are only ELEMENT aligned: views 1.. of a stack with an odd pixel count
    (depth on a 4- / 2-byte boundary, mask and colours on a 1-byte boundary, confidence likewise).  The lean kernels
    take them with element-aligned wide loads; every output equals the oracle."""
    import torch
    H, W = shape
    d = _rand_case(4242 + H * W, 4, H, W, dtype)
    dev = torch.device("cuda")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    depth, mask, normal, rgb, conf = (up(d[k])[1:] for k in ("depth", "mask", "normal", "rgb", "conf"))
    assert depth.data_ptr() % 16 != 0 or (H * W * depth.element_size()) % 16 == 0
    cloud = dd.unproject_views(depth, d["params"][1:], d["cam_from_world"][1:], mask=mask, normal=normal, rgb=rgb, conf=conf,
                               conf_threshold=0.4, view_index=True, tuning=tuning)
    ref = orc.densify_scene_script(d["depth"][1:], d["params"][1:], d["cam_from_world"][1:], mask=d["mask"][1:], normal=d["normal"][1:],
                                   rgb=d["rgb"][1:], conf=d["conf"][1:], conf_threshold=0.4)
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"][1:], d["depth"][1:]))


@pytest.mark.parametrize("conf_dtype", (np.float32, np.float16))
@pytest.mark.parametrize("dtype", (np.float32, np.float16))
def test_oracle_conf_dtypes(dd, orc, dtype, conf_dtype):
    d = _rand_case(77, 2, 96, 160, dtype)
    conf = d["conf"].astype(conf_dtype)
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], conf=conf,
                               conf_threshold=0.5, rgb=d["rgb"])
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], rgb=d["rgb"],
                                   conf=conf, conf_threshold=0.5)
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"], d["depth"]))


def test_no_mask_dense_and_all_invalid(dd, orc):
    d = _rand_case(5, 3, 48, 64, specials=False)
    d["depth"][1] = 0.0                                   # whole view culled by depth > 0
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], normal=d["normal"])
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], normal=d["normal"])
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"], d["depth"]))
    assert int(cloud.counts[1]) == 0 and int(cloud.counts[0]) == 48 * 64
    empty = dd.unproject_views(np.zeros((2, 8, 8), np.float32), d["params"][:2], d["cam_from_world"][:2])
    assert len(empty) == 0 and list(empty.view_offsets.cpu().numpy()) == [0, 0, 0]


def test_viz_oracle_rotated_normals(dd, orc):
    d = _rand_case(9, 2, 120, 200, specials=False)
    K = dd.intrinsics_matrix(d["params"])
    K[:, 0, 1] = 0.25
    cloud = dd.unproject_views(d["depth"], K, d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"],
                               semantics="viz")
    ref = orc.densify_scene_viz(d["depth"], K, d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"])
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"], d["depth"]), normals="close")


def test_script_with_rotated_normals_option(dd, orc):
    """rotate_normals=True on the script path = a9 normals on a4 validity."""
    d = _rand_case(10, 1, 64, 64, specials=False)
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"],
                               rotate_normals=True).numpy()
    exp = orc.transform_normals_viz(d["normal"][0], d["cam_from_world"][0], d["mask"][0])
    assert np.abs(cloud["normals"] - exp).max() <= NORMAL_ATOL


# ------------------------------------------------------------------ fuse / capacity

def test_chained_batches_equal_one_batch(dd, orc):
    """scripts/test.py:238-240, 264-266: appending batch after batch == one concatenated cloud."""
    d = _rand_case(21, 6, 72, 96)
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"],
                                   rgb=d["rgb"])
    parts = [(0, 1), (1, 4), (4, 6)]
    batches = [dd.ViewBatch(d["depth"][a:b], d["params"][a:b], d["cam_from_world"][a:b], mask=d["mask"][a:b],
                            normal=d["normal"][a:b], rgb=d["rgb"][a:b]) for a, b in parts]
    cloud = dd.fuse_batches(batches, normals=True, colors=True, view_index=True)
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"], d["depth"]))


def test_capacity_overflow_is_detected_and_safe(dd):
    import torch
    d = _rand_case(22, 2, 64, 64, specials=False)
    batch = dd.ViewBatch(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"])
    n = int(dd.count_valid(batch).sum())
    builder = dd.CloudBuilder(n - 100)
    guard = builder.xyz.clone()
    builder.append(batch)
    with pytest.raises(OverflowError):
        builder.finish()
    torch.cuda.synchronize()
    assert int(builder.cursor.item()) == n      # the count is still exact
    full = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], capacity="max")
    assert torch.equal(builder.xyz[: n - 100], full.points[: n - 100])
    assert guard.shape == builder.xyz.shape


def test_count_valid_matches_oracle(dd, orc):
    d = _rand_case(23, 4, 100, 164)
    for s in (1, 3):
        batch = dd.ViewBatch(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], stride=s)
        ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], stride=s)
        assert np.array_equal(dd.count_valid(batch).cpu().numpy(), np.diff(ref.view_offsets))


def test_error_codes(dd):
    with pytest.raises(ValueError):
        dd.ViewBatch(np.ones((1, 4, 4), np.float32), np.ones((2, 4)), np.eye(4)[None, :3])
    with pytest.raises(ValueError):
        dd.ViewBatch(np.ones((1, 4, 4), np.float32), np.ones((1, 4)), np.eye(4)[None, :3], stride=0)
    with pytest.raises(ValueError):   # visualizer.py:266-267 raises ValueError without K / E
        dd.ViewBatch(np.ones((1, 4, 4), np.float32), np.ones((1, 5)), np.eye(4)[None, :3])


@pytest.mark.parametrize("fields", [(), ("normal",), ("rgb",), ("normal", "rgb")])
@pytest.mark.parametrize("use_mask", (True, False))
@pytest.mark.parametrize("tuning", (0, 4))
def test_field_subsets(dd, orc, fields, use_mask, tuning):
    """Every template instantiation of the lean kernel: {mask} x {normal} x {rgb} x {two-pass, single-pass}.
    320x240 views span several 4096-pixel tiles with ragged last tiles; the valid-pixel run ends
    mid-quad so the byte-store tail of the colour path is exercised."""
    d = _rand_case(31 + len(fields), 3, 240, 324)
    kw = {k: d[k] for k in fields}
    mask = d["mask"] if use_mask else None
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=mask, view_index=True, tuning=tuning, **kw)
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=mask, **kw)
    assert_cloud(cloud, ref, scene_radius(d["cam_from_world"], d["depth"]))


def test_plan_then_scatter_api(dd, orc):
    """dd_plan gives the exact rows before anything is written; dd_scatter fills them."""
    import torch
    d = _rand_case(41, 4, 120, 160)
    batch = dd.ViewBatch(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], rgb=d["rgb"])
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], rgb=d["rgb"])
    cursor = torch.full((1,), 1000, dtype=torch.int64, device="cuda")      # cloud already holds 1000 points
    plan = dd.plan_batch(batch, cursor)
    assert np.array_equal(plan.view_offsets.cpu().numpy(), ref.view_offsets + 1000)
    assert int(plan.num_points) == len(ref.points)
    builder = dd.CloudBuilder(1000 + len(ref.points), colors=True)
    builder.cursor.fill_(1000)
    builder.scatter(batch, plan)
    cloud = builder.finish()
    assert len(cloud) == 1000 + len(ref.points)
    assert np.array_equal(cloud.colors[1000:].cpu().numpy(), ref.colors)
    assert np.array_equal(cloud.pixel_index[1000:].cpu().numpy().astype(np.int64), ref.pixel_index)


# ------------------------------------------------------------------ full BASELINE sizes

def _device_stack(V, H, W, seed, depth_dtype="float32"):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    depth = torch.empty((V, H, W), device="cuda").uniform_(0.5, 8.0, generator=g).to(getattr(torch, depth_dtype))
    mask = torch.rand((V, H, W), device="cuda", generator=g) < 0.8
    normal = torch.nn.functional.normalize(torch.randn((V, H, W, 3), device="cuda", generator=g), dim=-1)
    rgb = torch.randint(0, 256, (V, H, W, 3), device="cuda", generator=g, dtype=torch.uint8)
    return depth, mask, normal, rgb


def _ring_poses(V, radius=4.0):
    E = np.zeros((V, 3, 4))
    for v in range(V):
        a = 2 * np.pi * v / V
        c = np.array([radius * np.cos(a), 0.3 * np.sin(3 * a), radius * np.sin(a)])
        z = -c / np.linalg.norm(c)
        x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])                 # cam_from_world rotation
        E[v, :, :3] = R
        E[v, :, 3] = -R @ c
    return E


def test_full_size_1080p_properties_and_oracle_sample(dd, orc):
    """BASELINE configs 2/3 shape (1920x1080, depth+normal+mask+rgb): size-independent properties
    on all views, full oracle comparison on two of them."""
    import torch
    V, H, W = 6, 1080, 1920
    depth, mask, normal, rgb = _device_stack(V, H, W, 1234)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    cloud = dd.unproject_views(depth, params, E, mask=mask, normal=normal, rgb=rgb, view_index=True, capacity="max")
    counts = cloud.counts.cpu().numpy()
    assert np.array_equal(counts, mask.sum(dim=(1, 2)).cpu().numpy())           # depth > 0 everywhere
    offs = cloud.view_offsets.cpu().numpy()
    pix = cloud.pixel_index.cpu().numpy()
    for v in range(V):                                                           # sorted inside each view
        assert np.all(np.diff(pix[offs[v]:offs[v + 1]]) > 0)
    # pixel_index is exactly the set of kept pixels
    v = 3
    assert np.array_equal(pix[offs[v]:offs[v + 1]], np.nonzero(mask[v].reshape(-1).cpu().numpy())[0])
    # round trip: project the points back through [R|t] and K -> original pixel and depth
    P = cloud.points[offs[v]:offs[v + 1]].double().cpu().numpy()
    cam = P @ E[v, :, :3].T + E[v, :, 3]
    u = cam[:, 0] / cam[:, 2] * params[v, 0] + params[v, 2]
    w_ = cam[:, 1] / cam[:, 2] * params[v, 1] + params[v, 3]
    assert np.abs(u - pix[offs[v]:offs[v + 1]] % W).max() < 2e-2
    assert np.abs(w_ - pix[offs[v]:offs[v + 1]] // W).max() < 2e-2
    dsel = depth[v].reshape(-1).cpu().numpy()[pix[offs[v]:offs[v + 1]]]
    assert np.abs(cam[:, 2] - dsel).max() < 1e-4 * 12
    # idempotence: a second run is bit-identical; so are the single-pass and the scalar kernels
    for tuning in (0, 4, 8, 1):     # fused call: single-pass (default), forced two-pass, explicit single-pass, scalar
        again = dd.unproject_views(depth, params, E, mask=mask, normal=normal, rgb=rgb, view_index=True,
                                   capacity="max", tuning=tuning)
        assert torch.equal(again.points, cloud.points) and torch.equal(again.colors, cloud.colors)
        assert torch.equal(again.normals, cloud.normals) and torch.equal(again.view_offsets, cloud.view_offsets)
        assert torch.equal(again.pixel_index, cloud.pixel_index) and torch.equal(again.view_index, cloud.view_index)
    # oracle on views 0 and 5 (about 1.5 s of NumPy each)
    dn, mn, nn, cn = (t.cpu().numpy() for t in (depth, mask, normal, rgb))
    rad = scene_radius(E, dn)
    for v in (0, 5):
        ref = orc.densify_view_script(dn[v], params[v], E[v], mask=mn[v], normal=nn[v], rgb=cn[v])
        sl = slice(offs[v], offs[v + 1])
        assert np.array_equal(pix[sl], ref["pixel_index"])
        assert_xyz(cloud.points[sl].double().cpu().numpy(), ref["points"], rad)
        assert np.array_equal(cloud.colors[sl].cpu().numpy(), ref["colors"])
        assert np.array_equal(cloud.normals[sl].cpu().numpy(), ref["normals"])


def test_full_size_12mp_f16(dd, orc):
    """BASELINE config 5 shape: 4032x3024 float16 depth in, float32 xyz out, dense (depth > 0)."""
    import torch
    V, H, W = 2, 3024, 4032
    g = torch.Generator(device="cuda").manual_seed(99)
    depth = torch.empty((V, H, W), device="cuda").uniform_(0.5, 8.0, generator=g).half()
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    cloud = dd.unproject_views(depth, params, E, capacity="max", pixel_index=True)
    assert len(cloud) == V * H * W
    assert torch.equal(cloud.pixel_index[: H * W], torch.arange(H * W, device="cuda", dtype=torch.int32))
    dn = depth[1].cpu().numpy()
    ref = orc.densify_view_script(dn, params[1], E[1])
    assert_xyz(cloud.points[H * W:].double().cpu().numpy(), ref["points"], scene_radius(E, dn))


def test_visualizer_add_rgbd_pointcloud_dropin(dd, golden_small):
    """The reference's package API (visualizer.py:246-289) through the alias package."""
    from depthdensifier.visualizer import COLMAPVisualizer
    import depthdensifier
    assert depthdensifier.__version__ == "0.1.0" and depthdensifier.DepthRefiner is dd.DepthRefiner
    i = _inputs(golden_small, "a")
    viz = COLMAPVisualizer()
    with pytest.raises(ValueError):
        viz.add_rgbd_pointcloud(i["depth"][0])
    d32 = i["depth"][0].astype(np.float32)
    pts = viz.add_rgbd_pointcloud(d32, i["rgb"][0], i["Kskew"], i["cam_from_world"][0], i["mask"][0], i["normal"][0], name="x")
    n0 = int(golden_small["a_exp_viz_counts"][0])
    assert pts.dtype == np.float64 and pts.shape == (n0, 3)
    assert_xyz(pts, golden_small["a_exp_viz_points__depth_to_pointcloud"][:n0], scene_radius(i["cam_from_world"][:1], d32))
    pc = viz.point_clouds[0]
    assert pc.name == "x" and np.array_equal(pc.colors, golden_small["a_exp_viz_colors__depth_to_pointcloud"][:n0])
    assert np.abs(pc.normals - golden_small["a_exp_viz_normals__transform_normals"][:n0]).max() <= NORMAL_ATOL
    # 4x4 extrinsics and no mask -> depth > 0 validity, no normals (visualizer.py:276, 325-327)
    E4 = np.vstack([i["cam_from_world"][0], [0, 0, 0, 1.0]])
    viz.add_rgbd_pointcloud(d32, None, i["Kskew"], E4, None, i["normal"][0])
    assert viz.point_clouds[1].normals is None and viz.point_clouds[1].colors is None
    assert len(viz.point_clouds[1].points) == int(golden_small["a_exp_viznomask_counts"][0])


def test_mask_only_pass1_and_positive_depth_hint(dd, orc):
    """Pass 1 without depth loads (mask-only validity): viz semantics and the refiner-output hint give the
    same cloud as the full rule when depth > 0 on the mask."""
    d = _rand_case(51, 3, 128, 192, specials=False)          # depth in [0.5, 5): positive everywhere
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], rgb=d["rgb"])
    hinted = dd.fuse_batches([dd.ViewBatch(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], rgb=d["rgb"],
                                           depth_positive_on_mask=True)], colors=True, view_index=True)
    assert hinted.numpy()["view_offsets"].tolist() == ref.view_offsets.tolist()
    assert_cloud(hinted, ref, scene_radius(d["cam_from_world"], d["depth"]))


def test_offsets_beyond_32_bits(dd, orc):
    """480 views x 1080p: 995 M pixels, ~800 M points -> xyz element offsets > 2^31 and byte offsets > 4 GiB
    (BASELINE config 3 territory).  Counts, per-view sortedness and the LAST views against the oracle."""
    import torch
    V, H, W = 480, 1080, 1920
    g = torch.Generator(device="cuda").manual_seed(2024)
    depth = torch.empty((V, H, W), device="cuda")
    mask = torch.empty((V, H, W), dtype=torch.bool, device="cuda")
    for v in range(V):                      # per-view generation keeps the temporaries small
        depth[v].uniform_(0.5, 8.0, generator=g)
        mask[v] = torch.rand((H, W), device="cuda", generator=g) < 0.8
    rgb = torch.empty((V, H, W, 3), dtype=torch.uint8, device="cuda")
    for v in range(0, V, 20):
        rgb[v:v + 20] = torch.randint(0, 256, (min(20, V - v), H, W, 3), device="cuda", generator=g, dtype=torch.uint8)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    cloud = dd.unproject_views(depth, params, E, mask=mask, rgb=rgb, view_index=True)
    offs = cloud.view_offsets.cpu().numpy()
    assert offs[-1] == len(cloud) and len(cloud) * 3 > 2 ** 31 and len(cloud) * 12 > 2 ** 32
    assert np.array_equal(np.diff(offs), mask.sum(dim=(1, 2)).cpu().numpy())
    for v in (V - 1, V - 2, 200):
        sl = slice(int(offs[v]), int(offs[v + 1]))
        ref = orc.densify_view_script(depth[v].cpu().numpy(), params[v], E[v], mask=mask[v].cpu().numpy(), rgb=rgb[v].cpu().numpy())
        assert np.array_equal(cloud.pixel_index[sl].cpu().numpy().astype(np.int64), ref["pixel_index"])
        assert torch.all(cloud.view_index[sl] == v)
        assert np.array_equal(cloud.colors[sl].cpu().numpy(), ref["colors"])
        assert_xyz(cloud.points[sl].double().cpu().numpy(), ref["points"], scene_radius(E, np.array([8.0])))
    # the fused single-pass call (ticket + look-back over ~81 000 tiles) gives the same rows, bit for bit
    fused = dd.unproject_views(depth, params, E, mask=mask, rgb=rgb, view_index=True, capacity=len(cloud) + 12345)
    assert torch.equal(fused.view_offsets, cloud.view_offsets) and torch.equal(fused.points, cloud.points)
    assert torch.equal(fused.colors, cloud.colors) and torch.equal(fused.pixel_index, cloud.pixel_index)
    assert torch.equal(fused.view_index, cloud.view_index)


@pytest.mark.parametrize("conf_dtype", ("float32", "float16"))
def test_full_size_1080p_confidence_cull(dd, orc, conf_dtype):
    """BASELINE configs[3] shape (the per-GPU share of the 7 Mip-NeRF 360 scenes): 1920x1080, mask AND conf > 0.5, all four
    fields, float32 and float16 confidence maps.  Properties on all 6 views, the oracle on two; every kernel variant
    gives the same bits."""
    import torch
    V, H, W = 6, 1080, 1920
    depth, mask, normal, rgb = _device_stack(V, H, W, 4321)
    g = torch.Generator(device="cuda").manual_seed(77)
    conf = torch.rand((V, H, W), device="cuda", generator=g).to(getattr(torch, conf_dtype))
    depth.view(-1)[::100003] = float("nan")                       # a few culled by depth > 0 as well
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    kw = dict(mask=mask, conf=conf, conf_threshold=0.5, normal=normal, rgb=rgb, view_index=True)
    cloud = dd.unproject_views(depth, params, E, capacity="max", **kw)
    thr = torch.tensor(0.5, dtype=conf.dtype, device="cuda")    # NEP 50: the threshold is rounded to the map's dtype
    keep = mask & (conf > thr) & (depth > 0)
    assert np.array_equal(cloud.counts.cpu().numpy(), keep.sum(dim=(1, 2)).cpu().numpy())
    assert 0.35 < len(cloud) / (V * H * W) < 0.45
    offs = cloud.view_offsets.cpu().numpy()
    pix = cloud.pixel_index.cpu().numpy()
    for v in range(V):
        assert np.array_equal(pix[offs[v]:offs[v + 1]], np.nonzero(keep[v].reshape(-1).cpu().numpy())[0])   # sorted, exactly the kept set
    for tuning in (4, 1):                                       # two-pass lean, scalar kernels: bit-identical
        again = dd.unproject_views(depth, params, E, capacity="max", tuning=tuning, **kw)
        for name in ("points", "colors", "normals", "view_offsets", "pixel_index", "view_index"):
            assert torch.equal(getattr(again, name), getattr(cloud, name)), (tuning, name)
    exact = dd.unproject_views(depth, params, E, **kw)          # exact allocation (count pass with the conf rule)
    assert len(exact) == len(cloud) and torch.equal(exact.points, cloud.points)
    dn, mn, nn, cn, fn = (t.cpu().numpy() for t in (depth, mask, normal, rgb, conf))
    rad = scene_radius(E, np.array([8.0]))
    for v in (1, 4):
        ref = orc.densify_view_script(dn[v], params[v], E[v], mask=mn[v], normal=nn[v], rgb=cn[v], conf=fn[v], conf_threshold=0.5)
        sl = slice(offs[v], offs[v + 1])
        assert np.array_equal(pix[sl], ref["pixel_index"])
        assert_xyz(cloud.points[sl].double().cpu().numpy(), ref["points"], rad)
        assert np.array_equal(cloud.colors[sl].cpu().numpy(), ref["colors"])
        assert np.array_equal(cloud.normals[sl].cpu().numpy(), ref["normals"])


def test_per_gpu_share_of_the_2000_view_scene_with_normals(dd, orc):
    """BASELINE configs[2], what ONE of 8 GPUs holds: 250 views x 1080p with depth + mask + normal + rgb (10.4 GB in,
    ~11 GB out), one fused call.  Counts, order and ranges on all views; first, middle and LAST view against the oracle
    (rows beyond the 4 GiB byte offset carry normals here, unlike test_offsets_beyond_32_bits)."""
    import torch
    V, H, W = 250, 1080, 1920
    g = torch.Generator(device="cuda").manual_seed(250)
    depth = torch.empty((V, H, W), device="cuda")
    mask = torch.empty((V, H, W), dtype=torch.bool, device="cuda")
    normal = torch.empty((V, H, W, 3), device="cuda")
    rgb = torch.empty((V, H, W, 3), dtype=torch.uint8, device="cuda")
    for v in range(V):                      # per-view generation keeps the temporaries small
        depth[v].uniform_(0.5, 8.0, generator=g)
        mask[v] = torch.rand((H, W), device="cuda", generator=g) < 0.8
        normal[v] = torch.nn.functional.normalize(torch.randn((H, W, 3), device="cuda", generator=g), dim=-1)
        rgb[v] = torch.randint(0, 256, (H, W, 3), device="cuda", generator=g, dtype=torch.uint8)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    cloud = dd.unproject_views(depth, params, E, mask=mask, normal=normal, rgb=rgb, view_index=True, capacity="max")
    offs = cloud.view_offsets.cpu().numpy()
    assert len(cloud) * 12 > 2 ** 32                                          # byte offsets of the xyz / normal rows pass 4 GiB
    assert np.array_equal(np.diff(offs), mask.sum(dim=(1, 2)).cpu().numpy())
    vi = cloud.view_index
    assert int(vi[0]) == 0 and int(vi[-1]) == V - 1 and bool(torch.all(vi[1:] >= vi[:-1]))
    inc = cloud.pixel_index[1:] > cloud.pixel_index[:-1]                      # sorted inside every view
    assert bool(torch.all(inc | (vi[1:] != vi[:-1])))
    nn_ = torch.linalg.vector_norm(cloud.normals, dim=1)
    assert float((nn_ - 1).abs().max()) < 1e-5                                # pass-through unit normals, no stray rows
    rad = scene_radius(E, np.array([8.0]))
    for v in (0, 125, V - 1):
        sl = slice(int(offs[v]), int(offs[v + 1]))
        ref = orc.densify_view_script(depth[v].cpu().numpy(), params[v], E[v], mask=mask[v].cpu().numpy(), normal=normal[v].cpu().numpy(),
                                      rgb=rgb[v].cpu().numpy())
        assert np.array_equal(cloud.pixel_index[sl].cpu().numpy().astype(np.int64), ref["pixel_index"])
        assert np.array_equal(cloud.colors[sl].cpu().numpy(), ref["colors"])
        assert np.array_equal(cloud.normals[sl].cpu().numpy(), ref["normals"])
        assert_xyz(cloud.points[sl].double().cpu().numpy(), ref["points"], rad)
    # the 16-byte record of the compact gather holds the same xyz bits and colours
    del cloud
    torch.cuda.empty_cache()
    packed = dd.unproject_views(depth, params, E, mask=mask, rgb=rgb, record="xyz_rgba", pixel_index=False, capacity="max")
    rows = dd.unproject_views(depth, params, E, mask=mask, rgb=rgb, pixel_index=False, capacity="max")
    assert torch.equal(packed.view_offsets, rows.view_offsets)
    assert torch.equal(packed.packed[:, :3], rows.points)
    assert torch.equal(packed.colors, rows.colors)


@pytest.mark.parametrize("tuning", (0, 8))
def test_12mp_f16_dense_beyond_2_31_rows(dd, orc, tuning):
    """BASELINE configs[4] at scale in its OWN instantiations (f16 depth, no mask, no normals, no colours, xyz out): 180 views
    of 4032 x 3024 = 2.19 G rows (> 2^31), 26 GB of points.  Every pixel is valid, so the cloud is the pixel grid: length,
    per-view offsets, pixel_index of a late view == arange, and the LAST view against the oracle.  ``tuning`` 0: the path the
    builder chooses (an unmasked batch: no counting pass, the scatter verifies -- ``CloudBuilder.fuse_tuning``); 8: the single pass."""
    import torch
    V, H, W = 180, 3024, 4032
    P = H * W
    assert V * P > 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(12)
    depth = torch.empty((V, H, W), dtype=torch.float16, device="cuda")
    for v in range(V):
        depth[v] = torch.rand((H, W), device="cuda", generator=g) * 7.0 + 1.0          # 1 .. 8 m, all > 0
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    batch = dd.ViewBatch(depth, params, E, tuning=tuning)
    b = dd.CloudBuilder(batch.max_points, pixel_index=True)
    assert bool(b.fuse_tuning(batch) & (1 << 17)) == (tuning == 0)
    b.append(batch)
    cloud = b.finish()
    assert len(cloud) == V * P and b.healed == 0 and b.dense_misses == 0
    offs = cloud.view_offsets
    assert torch.equal(offs, torch.arange(V + 1, device="cuda", dtype=torch.int64) * P)
    for v in (0, V // 2, V - 3, V - 1):                                                # rows beyond 2^31 included
        assert torch.equal(cloud.pixel_index[v * P:(v + 1) * P], torch.arange(P, device="cuda", dtype=torch.int32)), v
    v = V - 1
    ref = orc.densify_view_script(depth[v].cpu().numpy(), params[v], E[v])
    assert len(ref["points"]) == P
    assert_xyz(cloud.points[v * P:].double().cpu().numpy(), ref["points"], scene_radius(E, np.array([8.0])))
    # a sample of rows across the whole cloud: finite and inside the scene
    idx = torch.randint(0, V * P, (1 << 20,), device="cuda", generator=g)
    assert bool(torch.isfinite(cloud.points[idx]).all()) and float(cloud.points[idx].abs().max()) < 20.0


@pytest.mark.parametrize("fields", ("all", "xyz"))
def test_scan_timeout_is_healed_by_a_two_pass_redo(dd, fields):
    """The fault-injection switch (``DD_LAB_FAULT_INJECT``, include/ddcore_lab.h) makes every look-back of the single-pass kernel give up at its first wait (what a workgroup parked for
    ~2 s would cause): the error word is set, rows are garbage.  finish() must notice, redo the retained batches with the
    dependency-free two-pass kernels into the SAME rows and offset tensors, and report it in `healed`."""
    import torch
    V, H, W = 5, 540, 960
    depth, mask, normal, rgb = _device_stack(V, H, W, 99)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    kw = dict(mask=mask, normal=normal, rgb=rgb) if fields == "all" else dict(mask=mask)
    good = dd.ViewBatch(depth, params, E, **kw)
    want_b = dd.CloudBuilder(good.max_points, normals=fields == "all", colors=fields == "all", pixel_index=True)
    want_b.append(good)
    want = want_b.finish()
    assert want_b.healed == 0
    # two chained batches, the second one sabotaged
    first = dd.ViewBatch(depth[:2], params[:2], E[:2], **{k: t[:2] for k, t in kw.items()})
    second = dd.ViewBatch(depth[2:], params[2:], E[2:], view_index_base=2, lab=FAULT_INJECT, **{k: t[2:] for k, t in kw.items()})
    b = dd.CloudBuilder(good.max_points, normals=fields == "all", colors=fields == "all", pixel_index=True)
    o1 = b.append(first)
    o2 = b.append(second)
    got = b.finish()
    assert b.healed == 1 and second.lab == FAULT_INJECT and second.tuning == 0
    assert len(got) == len(want) and torch.equal(got.view_offsets, want.view_offsets)
    assert int(o1[-1]) == int(o2[0]) == int(want.view_offsets[2])                     # the offset tensors handed out were rewritten
    for name in ("points", "normals", "colors", "pixel_index"):
        a, w = getattr(got, name), getattr(want, name)
        assert (a is None and w is None) or torch.equal(a, w), name
    # and the builder is usable afterwards
    b.reset()
    b.append(good)
    assert torch.equal(b.finish().points, want.points) and b.healed == 1
    # check() between appends: what it has verified is final and no longer held; a later give-up redoes only what came after it
    b.reset()
    b.append(first)
    n1 = b.check()
    assert n1 == int(want.view_offsets[2]) and not b._retained
    b.append(second)
    got2 = b.finish()
    assert b.healed == 2 and torch.equal(got2.points, want.points) and torch.equal(got2.pixel_index, want.pixel_index)
    assert not b._retained                                                            # finish() lets the maps go
    # an input overwritten in place between append() and finish(): the redo would read other pixels -- refused, like before round 3
    b.reset()
    scratch = depth[2:].clone()
    third = dd.ViewBatch(scratch, params[2:], E[2:], view_index_base=2, lab=FAULT_INJECT, **{k: t[2:] for k, t in kw.items()})
    b.append(first)
    b.append(third)
    scratch.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        b.finish()


def test_check_async_reads_later_what_check_reads_now(dd):
    """``CloudBuilder.check_async``: count and scan status travel to page-locked memory behind the kernels enqueued so far; the
    answer is that of ``check()``, also when the builder has been reset and refilled in the meantime (scene after scene on one
    pooled set of arrays, ``scripts/run_batch.py:57-91``); a give-up seen late is an error when a redo is no longer possible."""
    import torch
    V, H, W = 4, 300, 500
    depth, mask, normal, rgb = _device_stack(V, H, W, 5)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    a = dd.ViewBatch(depth[:3], params[:3], E[:3], mask=mask[:3])
    c = dd.ViewBatch(depth[1:], params[1:], E[1:], mask=mask[1:])
    b = dd.CloudBuilder(a.max_points + c.max_points, pixel_index=True)
    b.append(a)
    p1 = b.check_async()
    b.reset()                                              # the next "scene" on the same arrays
    b.append(c)
    p2 = b.check_async()
    assert p1.result() == int(mask[:3].sum()) and p2.result() == int(mask[1:].sum())
    assert b.check() == int(mask[1:].sum())
    # a fault seen by a late reader: the cloud it was asked about is gone -> an error, not a silent redo of another cloud
    bad = dd.ViewBatch(depth, params, E, mask=mask, lab=FAULT_INJECT)
    b.reset(); b.append(bad)
    p3 = b.check_async()
    b.reset(); b.append(a)
    with pytest.raises(RuntimeError, match="timed out"):
        p3.result()
    assert b.check() == int(mask[:3].sum())                # the sticky word was cleared by the reader that saw it
    b.reset(); b.append(bad)
    with pytest.raises(RuntimeError, match="timed out"):
        b.check_async().result(heal=False)
    b.reset(); b.append(bad)
    assert b.check() == int(mask.sum()) and b.healed >= 1   # the synchronous form heals


@pytest.mark.parametrize("rho", (0.05, 0.5, 0.97))
def test_every_row_alignment_of_the_first_row(dd, rho):
    """The lean kernel shifts its sweeps so that wave runs start on 128-byte lines of the outputs (32-row period); the
    shift depends on the tile's first row.  Start the cloud at every row 0..40 (all 32 phases), with sparse / medium /
    nearly dense tiles (lists shorter than one sweep, a few sweeps, almost full), all fields, and with the capacity cutting
    the cloud at an odd row: the rows written are always the same rows, shifted."""
    import torch
    V, H, W = 3, 211, 307                          # ~65 k pixels per view: 5-6 tiles, ragged end
    g = torch.Generator(device="cuda").manual_seed(int(rho * 100))
    depth = torch.empty((V, H, W), device="cuda").uniform_(0.5, 8.0, generator=g)
    mask = torch.rand((V, H, W), device="cuda", generator=g) < rho
    normal = torch.nn.functional.normalize(torch.randn((V, H, W, 3), device="cuda", generator=g), dim=-1)
    rgb = torch.randint(0, 256, (V, H, W, 3), device="cuda", generator=g, dtype=torch.uint8)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring_poses(V)
    batch = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    ref = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=True, view_index=True, packed=True)
    ref.append(batch)
    want = ref.finish()
    n = len(want)
    assert n == int(mask.sum())
    fields = ("points", "normals", "colors", "pixel_index", "view_index", "packed")
    for start in list(range(0, 41)) + [95, 127, 1000003]:
        for cut in (0, 777):                       # capacity = everything, or 777 rows short
            cap = start + n - cut
            b = dd.CloudBuilder(cap, normals=True, colors=True, pixel_index=True, view_index=True, packed=True, start=start)
            for t in (b.xyz, b.normal, b.packed):
                t.fill_(-7.0)
            b.rgb.fill_(201); b.pix.fill_(-7); b.view.fill_(-7)
            b.append(batch)
            torch.cuda.synchronize()
            assert int(b.cursor.item()) == start + n                      # counted even where not written
            kept = n - cut
            got = dict(points=b.xyz, normals=b.normal, colors=b.rgb, pixel_index=b.pix, view_index=b.view, packed=b.packed)
            for name in fields:
                w = getattr(want, name)[:kept]
                t = got[name]
                assert torch.equal(t[start:start + kept].view(torch.uint8), w.contiguous().view(torch.uint8)), (start, cut, name)
                if start:                                                  # nothing before the first row was touched
                    head = t[:start]
                    assert bool((head == (201 if name == "colors" else -7)).all()), (start, cut, name, "rows before the start")


def test_c_abi_client_without_python(tmp_path):
    """A C++/HIP program linking libddcore.so (no torch, no Python in the loop) drives dd_plan + dd_scatter and
    the fused call and checks them against its own float64 loop (tests/c_client/abi_client.cpp)."""
    import shutil, subprocess
    from pathlib import Path
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "abi_client"
    lib_dir = root / "depthdensifier_amd"
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", f"-I{root / 'include'}",
                            str(root / "tests" / "c_client" / "abi_client.cpp"), f"-L{lib_dir}", "-lddcore",
                            f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "C ABI OK" in run.stdout, run.stdout + run.stderr
    # round 5: the per-view loop on one stream (mode 3) and chained across two probed streams (mode 4: DDViewBatch.chain, dd_stream_fork,
    # dd_streams_overlap) -- or, said so, not chained where no two streams of the process run side by side
    assert "mode 3:" in run.stdout and ("chained across two streams" in run.stdout or "chaining skipped" in run.stdout), run.stdout
