"""Floater filter: oracle vs the reference's project_points golden (CPU) and the
HIP vote kernel vs the oracle (GPU; votes are integers -> bit-exact)."""

import os

import numpy as np
import pytest

from oracle import filter_oracle as forc


@pytest.fixture(scope="module")
def gfilter():
    from pathlib import Path
    return dict(np.load(Path(__file__).parent / "golden" / "filter_small.npz"))


@pytest.mark.parametrize("tag", ("f64", "f32"))
def test_project_points_against_reference(gfilter, tag):
    """scripts/test.py:58-76 as returned by the reference itself (make_goldens.build_filter)."""
    pts = gfilter["in_points64"] if tag == "f64" else gfilter["in_points32"]
    for v in range(gfilter["in_K"].shape[0]):
        p2, d = forc.project_points(pts, gfilter["in_cam_from_world"][v], gfilter["in_K"][v])
        e2, ed = gfilter[f"exp_{tag}_v{v}_points2d__project_points"], gfilter[f"exp_{tag}_v{v}_depths__project_points"]
        assert p2.dtype == e2.dtype == np.float64
        assert np.abs(d - ed).max() <= 1e-12 * np.abs(ed).max()
        rel = np.abs(p2 - e2) / np.maximum(1.0, np.abs(e2))
        assert rel.max() <= 1e-9            # points near depth ~ 0 amplify the last-bit matmul difference


def test_votes_small_scene_cpu():
    """Hand-checkable case: a wall at depth 4 seen by two identical cameras; a point at depth 2 is a
    floater in both (2 < 0.7*4), a point on the wall in neither, a point behind the camera in neither."""
    E = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))]), (2, 1, 1))
    K = np.tile(np.array([[50.0, 0, 16], [0, 50.0, 12], [0, 0, 1]]), (2, 1, 1))
    depth = np.full((2, 24, 32), 4.0, np.float32)
    pts = np.array([[0, 0, 2.0], [0, 0, 4.0], [0, 0, -1.0], [100.0, 0, 2.0]])
    nrm = np.tile([0, 0, -1.0], (4, 1)).astype(np.float32)     # facing the cameras
    assert list(forc.floater_votes(pts, nrm, depth, K, E)) == [2, 0, 0, 0]
    grazing = np.tile([1.0, 0, 0], (4, 1)).astype(np.float32)
    assert list(forc.floater_votes(pts, grazing, depth, K, E)) == [0, 0, 0, 0]
    depth[1] = 0                                                # second view has no valid depth there
    assert list(forc.floater_votes(pts, nrm, depth, K, E)) == [1, 0, 0, 0]


def _scene(seed, V, H, W):
    from synth import make_views
    d = make_views(seed, V, H, W, rho=0.85, specials=True)
    rng = np.random.default_rng(seed)
    params = np.tile([0.9 * W, 0.95 * W, W / 2.0, H / 2.0], (V, 1))
    E = np.zeros((V, 3, 4))
    for v in range(V):                        # cameras on a ring looking at the origin: views overlap
        a = 2 * np.pi * v / V + rng.uniform(-0.05, 0.05)
        c = np.array([3.5 * np.cos(a), 0.2 * rng.standard_normal(), 3.5 * np.sin(a)])
        z = -c / np.linalg.norm(c); x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x); y = np.cross(z, x)
        R = np.stack([x, y, z]); E[v, :, :3] = R; E[v, :, 3] = -R @ c
    d["cam_from_world"] = E
    d["params"] = params
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("with_mask", (True, False))
def test_gpu_votes_match_oracle(with_mask):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    from oracle import densify_oracle as orc

    d = _scene(5, 7, 72, 96)
    mask = d["mask"] if with_mask else None
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=mask, normal=d["normal"], rgb=d["rgb"])
    K = dd.intrinsics_matrix(d["params"])
    culled = np.stack([orc.fold_cull_into_depth(d["depth"][v], None if mask is None else mask[v]) for v in range(7)])
    with np.errstate(invalid="ignore", over="ignore"):
        culled = np.where(np.isfinite(culled), culled, 0).astype(np.float32)   # keep the scene finite
    depth_in = np.where(np.isfinite(d["depth"]), d["depth"], 0).astype(np.float32)
    pts = cloud.points.cpu().numpy()
    fin = np.isfinite(pts).all(axis=1)
    votes = dd.floater_votes(cloud.points, cloud.normals, depth_in, K, d["cam_from_world"], mask=mask).cpu().numpy()
    ref = forc.floater_votes(pts[fin], cloud.normals.cpu().numpy()[fin], culled, K, d["cam_from_world"])
    assert ref.max() >= 3, "scene too tame to exercise the vote path"
    assert np.array_equal(votes[fin], ref)
    plain = dd.floater_votes(cloud.points, cloud.normals, depth_in, K, d["cam_from_world"], mask=mask, mode="float64")
    assert np.array_equal(plain.cpu().numpy(), votes)               # the un-culled kernel: same votes as the default (auto)
    # accumulate over view chunks == one call
    v2 = dd.floater_votes(cloud.points, cloud.normals, depth_in[:3], K[:3], d["cam_from_world"][:3], mask=None if mask is None else mask[:3])
    v2 = dd.floater_votes(cloud.points, cloud.normals, depth_in[3:], K[3:], d["cam_from_world"][3:], mask=None if mask is None else mask[3:], votes=v2)
    assert np.array_equal(v2.cpu().numpy(), votes)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 100 + int(os.environ.get("DD_VOTE_SEEDS", "8"))))      # soak: DD_VOTE_SEEDS=60
def test_gpu_votes_random_scenes(seed):
    """Random ring scenes (views, size, threshold drawn per seed): every vote equals the oracle's."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    rng = np.random.default_rng(seed)
    V, H, W = int(rng.integers(2, 9)), int(rng.integers(20, 90)), int(rng.integers(20, 120))
    d = _scene(seed, V, H, W)
    depth = np.where(np.isfinite(d["depth"]) & (d["depth"] > 0), d["depth"], 1.0).astype(np.float32)
    cloud = dd.unproject_views(depth, d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"])
    K = dd.intrinsics_matrix(d["params"])
    thr = float(rng.choice([0.7, 0.9, 0.5]))
    votes = dd.floater_votes(cloud.points, cloud.normals, depth, K, d["cam_from_world"], mask=d["mask"], depth_threshold=thr).cpu().numpy()
    culled = np.where(d["mask"], depth, 0).astype(np.float32)
    ref = forc.floater_votes(cloud.points.cpu().numpy(), cloud.normals.cpu().numpy(), culled, K, d["cam_from_world"], depth_threshold=thr)
    assert np.array_equal(votes, ref)
    plain = dd.floater_votes(cloud.points, cloud.normals, depth, K, d["cam_from_world"], mask=d["mask"], depth_threshold=thr, mode="float64")
    assert np.array_equal(plain.cpu().numpy(), ref)


@pytest.mark.gpu
def test_gpu_votes_on_decision_boundaries():
    """Pairs built to sit ON the decision thresholds: projections at (or 1e-14 from) integer pixel coordinates and
    the image edges -- the 1e-8 in the denominator puts them just below the integer -- and normals whose facing
    cosine equals the grazing threshold up to float32 rounding.  Both the cheap evaluation and the exact fallback
    of the filtered predicates are hit; every vote must equal the float64 NumPy restatement."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd

    H, W, fx = 48, 64, 64.0
    K = np.tile(np.array([[fx, 0, 32.0], [0, fx, 24.0], [0, 0, 1.0]]), (3, 1, 1))
    E = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))]), (3, 1, 1))
    E[1, 0, 3] = 0.25; E[2, 1, 3] = -0.5                      # exactly representable shifts
    depth = np.full((3, H, W), 1e9, np.float32)
    rng = np.random.default_rng(11)
    pts, nrm = [], []
    # (a) pixel-grid points: u = j / (1 + 1e-8 / z) for integer j -- far from / inside the guard band as z grows
    for z in (1.0, 1024.0, 1048576.0):
        j, i = np.meshgrid(np.arange(-1, W + 2), np.arange(-1, H + 2))
        p = np.stack([(j - 32.0) / fx * z, (i - 24.0) / fx * z, np.full(j.shape, z)], -1).reshape(-1, 3)
        pts.append(p); nrm.append(np.tile([0.0, 0.0, -1.0], (len(p), 1)))
    # (b) facing cosine = 0.087 in exact arithmetic, perturbed by the float32 rounding of the normal
    g = 0.087
    p = np.stack([rng.uniform(-0.3, 0.3, 20000), rng.uniform(-0.2, 0.2, 20000), np.ones(20000)], -1) * rng.uniform(1, 5, (20000, 1))
    p = p.astype(np.float32).astype(np.float64)
    d = p / np.linalg.norm(p, axis=1, keepdims=True)
    perp = np.cross(d, [0.0, 1.0, 0.0]); perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    pts.append(p); nrm.append(-(g * d + np.sqrt(1 - g * g) * perp))
    pts = np.concatenate(pts).astype(np.float32); nrm = np.concatenate(nrm).astype(np.float32)
    ref = forc.floater_votes(pts, nrm, depth, K, E)
    got = dd.floater_votes(torch.from_numpy(pts).cuda(), torch.from_numpy(nrm).cuda(), depth, K, E).cpu().numpy()
    assert np.array_equal(got, ref)
    got = dd.floater_votes(torch.from_numpy(pts).cuda(), torch.from_numpy(nrm).cuda(), depth, K, E, mode="float64").cpu().numpy()
    assert np.array_equal(got, ref)
    assert 0 < (ref[-20000:] > 0).mean() < 1 and ref[:-20000].max() == 3 and ref[:-20000].min() == 0     # both outcomes occur


@pytest.mark.gpu
def test_gpu_filter_floaters_matches_oracle():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd

    d = _scene(6, 6, 64, 80)
    d["depth"] = np.where(np.isfinite(d["depth"]) & (d["depth"] > 0), d["depth"], 1.0).astype(np.float32)
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"],
                               view_index=True)
    K = dd.intrinsics_matrix(d["params"])
    cfg = dd.FilteringConfig(vote_threshold=2, depth_threshold=0.7)
    out, votes = dd.filter_floaters(cloud, d["depth"], K, d["cam_from_world"], mask=d["mask"], config=cfg)
    culled = np.where(d["mask"], d["depth"], 0).astype(np.float32)
    c = cloud.numpy()
    ep, ec, en, ev = forc.filter_floaters(c["points"].astype(np.float32), c["colors"], c["normals"], culled, K,
                                          d["cam_from_world"], vote_threshold=2, depth_threshold=0.7)
    assert np.array_equal(votes.cpu().numpy(), ev)
    assert 0 < len(ep) < len(c["points"])
    o = out.numpy()
    assert np.array_equal(o["points"].astype(np.float32), ep) and np.array_equal(o["colors"], ec)
    assert np.array_equal(o["normals"], en)
    # per-view offsets of the filtered cloud follow the kept points
    counts = np.bincount(o["view_index"], minlength=6)
    assert np.array_equal(np.diff(o["view_offsets"]), counts)


@pytest.mark.gpu
@pytest.mark.parametrize("n", (0, 1, 63, 4095, 4096, 4097, 300_001))
@pytest.mark.parametrize("fields", ("all", "xyz_rgb", "xyz"))
def test_gpu_compact_cloud(n, fields):
    """dd_compact_cloud vs boolean indexing: stable order, every field, view offsets."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    g = torch.Generator(device="cuda").manual_seed(n + 7)
    pts = torch.randn((n, 3), device="cuda", generator=g)
    rgb = torch.randint(0, 256, (n, 3), device="cuda", generator=g, dtype=torch.uint8) if fields != "xyz" else None
    nrm = torch.randn((n, 3), device="cuda", generator=g) if fields == "all" else None
    pix = torch.arange(n, device="cuda", dtype=torch.int32) if fields == "all" else None
    counts = torch.tensor([n // 3, 0, n - n // 3 - n // 5, n // 5], dtype=torch.int64)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).cuda()
    view = torch.repeat_interleave(torch.arange(4, dtype=torch.int32), counts).cuda() if fields == "all" else None
    cloud = dd.FusedCloud(points=pts, colors=rgb, normals=nrm, pixel_index=pix, view_index=view, view_offsets=offs)
    votes = torch.randint(0, 8, (n,), device="cuda", generator=g, dtype=torch.int32)
    out = dd.compact_cloud(cloud, votes, 5)
    keep = votes < 5
    assert len(out) == int(keep.sum())
    assert torch.equal(out.points, pts[keep])
    if rgb is not None:
        assert torch.equal(out.colors, rgb[keep])
    if fields == "all":
        assert torch.equal(out.normals, nrm[keep]) and torch.equal(out.pixel_index, pix[keep]) and torch.equal(out.view_index, view[keep])
        assert torch.equal(out.view_offsets.cpu(), torch.cat([torch.zeros(1, dtype=torch.int64),
                                                               torch.bincount(view[keep].long(), minlength=4).cumsum(0).cpu()]))
    # keep everything / drop everything
    assert len(dd.compact_cloud(cloud, votes, 100)) == n and len(dd.compact_cloud(cloud, votes, 0)) == 0


@pytest.mark.gpu
def test_every_vote_kernel_gives_the_oracle_votes_on_special_points():
    """The float64 kernels (round-1 form, division-free bounds test, culling, the on-device choice, and what a zero-initialised
    C struct selects): the same votes on a ring scene, on points that reproject exactly onto pixel centres / image borders /
    the camera centre, on huge and non-finite coordinates -- and the oracle's votes on every finite point."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    from depthdensifier_amd import _lib
    from depthdensifier_amd.filtering import filter_cameras

    d = _scene(11, 9, 96, 128)
    K = dd.intrinsics_matrix(d["params"])
    E = d["cam_from_world"]
    depth_in = np.where(np.isfinite(d["depth"]), d["depth"], 0).astype(np.float32)
    cloud = dd.unproject_views(depth_in, d["params"], E, mask=d["mask"], normal=d["normal"])      # points ON pixel centres of their own view
    pts, nrm = cloud.points.clone(), cloud.normals.clone()
    n = len(pts)
    centres = torch.as_tensor(np.stack([-E[v, :, :3].T @ E[v, :, 3] for v in range(9)]), dtype=torch.float32, device="cuda")
    pts[:9] = centres                                                          # exactly at a camera centre
    pts[100:110] = float("nan"); pts[110:120] = float("inf"); pts[120:130] = 3e30; nrm[130:140] = float("nan")
    pts[140:150] *= 1e-30                                                      # underflow territory
    v64 = dd.floater_votes(pts, nrm, depth_in, K, E, mask=d["mask"], mode="float64")
    for mode in ("float64", "float64_cull", "float64_cull1", "auto"):
        assert torch.equal(dd.floater_votes(pts, nrm, depth_in, K, E, mask=d["mask"], mode=mode), v64), mode
    fin = torch.isfinite(pts).all(dim=1).cpu().numpy()                         # and the oracle on every finite point
    culled = np.where(d["mask"], depth_in, 0).astype(np.float32)
    ref = forc.floater_votes(pts.cpu().numpy()[fin], nrm.cpu().numpy()[fin], culled, K, E)
    assert np.array_equal(v64.cpu().numpy()[fin], ref)
    assert int(v64.max()) >= 3
    # mode 0 -- a zero-initialised C struct -- is the best the workspace allows: all three sizes, same votes
    V, H, W = depth_in.shape
    cams = torch.from_numpy(filter_cameras(K, E)).cuda()
    dz = torch.where(torch.as_tensor(d["mask"]).cuda(), torch.as_tensor(depth_in).cuda(), torch.zeros((), device="cuda")).contiguous()
    for nbytes in (0, 256 * V, int(_lib.lib.dd_votes_workspace_bytes(V, n))):
        ws = torch.zeros(max(nbytes, 32), dtype=torch.uint8, device="cuda")
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        fv = _lib.DDFilterViews(num_views=V, height=H, width=W, depth=dz.data_ptr(), mask=None, cams=cams.data_ptr(), grazing_cos=0.087,
                                depth_threshold=0.7, workspace=ws.data_ptr() if nbytes else None, workspace_bytes=nbytes, mode=0)
        rc = _lib.lib.dd_floater_votes(C.byref(fv), pts.data_ptr(), nrm.data_ptr(), n, out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        if nbytes == 0:                   # ABI 10: the table-free kernel a NULL workspace used to select is gone -- a clean error, nothing launched
            assert rc == -1 and b"256 * num_views" in _lib.lib.dd_filter_last_error()
            continue
        assert rc == 0, _lib.lib.dd_filter_last_error()
        assert torch.equal(out, v64), nbytes


@pytest.mark.gpu
@pytest.mark.parametrize("normals", ("random", "smooth"))
@pytest.mark.parametrize("layout", ("ring", "outward", "corridor"))
def test_view_culling_keeps_the_votes(layout, normals):
    """Per-workgroup view culling ("float64_cull": a view is skipped for 256 consecutive points when their bounding sphere
    cannot touch its frustum) and the on-device choice ("auto") give the votes of the un-culled kernels and of the oracle
    -- on a ring where nothing can be culled, with cameras looking outward from one spot (every view sees its own points
    only) and along a corridor (each view overlaps a few neighbours); special depths (inf -> infinite spheres) included."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    V, H, W = (12, 72, 96) if normals == "random" else (12, 20, 640)     # wide rows: a workgroup's 256 pixels stay in a small sphere
    d = _scene(21, V, H, W)
    rng = np.random.default_rng(3)
    E = d["cam_from_world"]
    if layout != "ring":
        for v in range(V):
            if layout == "outward":       # at the origin, looking away from it in 12 directions
                a = 2 * np.pi * v / V
                c = np.array([0.05 * np.cos(a), 0.0, 0.05 * np.sin(a)]); z = np.array([np.cos(a), 0.0, np.sin(a)])
            else:                          # walking along x, looking sideways (+z), slightly panning
                c = np.array([1.2 * v, 0.0, 0.0]); z = np.array([0.15 * np.sin(v), 0.0, 1.0]); z /= np.linalg.norm(z)
            x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x); y = np.cross(z, x)
            R = np.stack([x, y, z]); E[v, :, :3] = R; E[v, :, 3] = -R @ c
    if layout != "ring" or normals == "smooth":
        # a smooth surface ~3 m away (a workgroup's 256 pixels then lie in a small sphere); the special depths stay
        smooth = (3.0 + 0.2 * rng.standard_normal(d["depth"].shape)).astype(np.float32)
        ordinary = np.isfinite(d["depth"]) & (d["depth"] > 0)
        d["depth"] = np.where(ordinary, smooth, d["depth"])
    if normals == "smooth":                # a slowly varying field like a monocular normal map (waves take one path through
        ys, xs = np.mgrid[0:H, 0:W]        # the grazing test), with a few NaN / inf / zero / huge normals thrown in
        for v in range(V):
            n = np.stack([0.6 * np.sin(5.0 * xs / W + v), 0.6 * np.cos(4.0 * ys / H + 0.5 * v), -np.ones((H, W))], -1)
            d["normal"][v] = (n / np.linalg.norm(n, axis=-1, keepdims=True)).astype(np.float32)
        flat = d["normal"].reshape(-1, 3)
        flat[11::9377] = np.nan; flat[13::9833, 1] = np.inf; flat[17::9901] = 0.0; flat[19::9973] *= 1e20
    K = dd.intrinsics_matrix(d["params"])
    depth_in = np.where(np.isfinite(d["depth"]), d["depth"], 0).astype(np.float32)
    cloud = dd.unproject_views(d["depth"], d["params"], E, mask=d["mask"], normal=d["normal"])     # inf depths -> inf points
    res, stats = {}, {}
    for mode in ("float64", "float64_cull", "float64_cull1", "auto"):
        st = {}
        res[mode] = dd.floater_votes(cloud.points, cloud.normals, depth_in, K, E, mask=d["mask"], mode=mode, stats=st).cpu().numpy()
        stats[mode] = st
    for mode in ("float64", "float64_cull", "float64_cull1", "auto"):
        assert np.array_equal(res[mode], res["float64"]), mode
    pts = cloud.points.cpu().numpy()
    fin = np.isfinite(pts).all(axis=1)
    culled = np.where(d["mask"], depth_in, 0).astype(np.float32)
    ref = forc.floater_votes(pts[fin], cloud.normals.cpu().numpy()[fin], culled, K, E)
    assert np.array_equal(res["auto"][fin], ref)
    s = stats["auto"]
    assert s["cull_sample_cells"] > 0 and 0 <= s["cull_sample_survived"] <= s["cull_sample_cells"]
    frac = s["cull_sample_survived"] / s["cull_sample_cells"]
    if layout == "outward":              # each point is inside two or three of the twelve 58-degree frusta
        assert s["culled"] and frac < 0.7, frac
    if layout == "ring" and normals == "random":
        assert not s["culled"], frac         # every view sees every point and the normals of a workgroup point everywhere
    # accumulating over chunks of views goes through the same choice per call
    v2 = dd.floater_votes(cloud.points, cloud.normals, depth_in[:5], K[:5], E[:5], mask=d["mask"][:5], mode="float64_cull")
    v2 = dd.floater_votes(cloud.points, cloud.normals, depth_in[5:], K[5:], E[5:], mask=d["mask"][5:], votes=v2, mode="auto")
    assert np.array_equal(v2.cpu().numpy(), res["float64"])


@pytest.mark.gpu
def test_floater_votes_argument_errors():
    """dd_floater_votes refuses what it cannot run, with a message: an unknown mode, a culling mode without its workspace,
    NULL pointers -- error code and dd_filter_last_error(), nothing thrown, nothing launched."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd import _lib
    lib = _lib.lib
    n, V, H, W = 300, 2, 8, 8
    pts = torch.zeros((n, 3), device="cuda"); nrm = torch.zeros((n, 3), device="cuda")
    depth = torch.ones((V, H, W), device="cuda"); cams = torch.zeros((V, 24), dtype=torch.float64, device="cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    ws = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def call(**kw):
        f = dict(num_views=V, height=H, width=W, depth=depth.data_ptr(), mask=None, cams=cams.data_ptr(), grazing_cos=0.087, depth_threshold=0.7,
                 workspace=None, workspace_bytes=0, mode=1)
        f.update(kw)
        rc = lib.dd_floater_votes(C.byref(_lib.DDFilterViews(**f)), pts.data_ptr(), nrm.data_ptr(), n, out.data_ptr(), 0, stream)
        return rc, lib.dd_filter_last_error().decode()

    rc, msg = call()                                                                  # ABI 10: no workspace, no kernel
    assert rc == -1 and "256 * num_views" in msg
    assert call(workspace=ws.data_ptr(), workspace_bytes=256 * V)[0] == 0
    rc, msg = call(mode=7)
    assert rc == -1 and "mode" in msg
    rc, msg = call(mode=4)
    assert rc == -1 and "workspace" in msg
    rc, msg = call(mode=3, workspace=ws.data_ptr(), workspace_bytes=100)
    assert rc == -1 and "workspace" in msg
    rc, msg = call(mode=3, workspace=ws.data_ptr(), workspace_bytes=ws.numel())      # 512 * 2 + 64 fit: runs (single-level cull)
    assert rc == 0
    rc, msg = call(cams=None)
    assert rc == -1 and "cams" in msg
    rc, msg = call(mode=2)
    assert rc == -1 and "removed" in msg
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks, views, layout", [(2, 12, "corridor"), (3, 13, "corridor"), (2, 9, "ring"), (3, 2, "corridor")])
def test_sharded_filter_gathers_only_the_views_in_reach(ranks, views, layout):
    """distributed.floater_votes_sharded with ranks sharing the GPU over gloo: votes equal the one-GPU votes with the selective
    gather and with the all-gather; on a corridor a rank receives fewer views than there are (tests/filter_worker.py)."""
    import socket
    import subprocess
    import sys
    from pathlib import Path
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, DD_FILTER_VIEWS=str(views), DD_FILTER_LAYOUT=layout, DD_PLACEMENT="first")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "tests" / "filter_worker.py")], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": ok,") == ranks, r.stdout


@pytest.mark.gpu
def test_views_in_reach_is_conservative():
    """Every view that casts a vote on some point must be in reach of the points' bounding spheres (the converse need not hold)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    from depthdensifier_amd import distributed as D
    d = _scene(31, 10, 48, 64)
    E = d["cam_from_world"]
    for v in range(10):                            # half the cameras turned away from the scene
        if v % 2:
            E[v, :, :3] = E[v, :, :3] * np.array([[-1.0], [1.0], [-1.0]])
            E[v, :, 3] = E[v, :, 3] * np.array([-1.0, 1.0, -1.0])
    ok = np.isfinite(d["depth"]) & (d["depth"] > 0)
    depth = np.where(ok, 3.0 + 0.1 * np.sin(np.where(ok, d["depth"], 0.0)), 0).astype(np.float32)      # a surface ~3 m away
    K = dd.intrinsics_matrix(d["params"])
    cloud = dd.unproject_views(depth[:3], d["params"][:3], E[:3], mask=d["mask"][:3], normal=d["normal"][:3])
    fin = torch.isfinite(cloud.points).all(dim=1)          # (the special depths of the scene give a few non-finite points: those put every view in reach)
    pts, nrm = cloud.points[fin].contiguous(), cloud.normals[fin].contiguous()
    reach = D.views_in_reach(pts, K, E, [depth.shape[1:]] * 10, chunk=16).cpu().numpy()
    for v in range(10):
        votes = dd.floater_votes(pts, nrm, depth[v:v + 1], K[v:v + 1], E[v:v + 1], mask=d["mask"][v:v + 1])
        if int(votes.sum()) > 0:
            assert reach[v], v
    assert not reach.all()                         # something was excluded, or the test says nothing
    assert D.views_in_reach(torch.full((5, 3), float("nan"), device="cuda"), K, E, [depth.shape[1:]] * 10).all()
