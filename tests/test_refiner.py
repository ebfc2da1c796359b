"""DepthRefiner against outputs of the reference's own class (CPU/FP32 goldens)."""

import os
from pathlib import Path

import numpy as np
import pytest
import torch

VARIANTS = {"default": {}, "nosmooth": {"skip_smoothing": True}, "notrobust": {"robust": False}, "nomask": {},
            "toofew": {"min_correspondences": 100000}}


@pytest.fixture(scope="module")
def g():
    return dict(np.load(Path(__file__).parent / "golden" / "refiner_small.npz"))


def _refiner(**kw):
    from depthdensifier_amd.depth_refiner import DepthRefiner
    r = DepthRefiner(adaptive_correspondences=False, use_fp16=False, **kw)
    return r


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_against_reference_golden_cpu(g, name, monkeypatch):
    """Same torch ops on the CPU in FP32 -> same result as the reference returned."""
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    r = _refiner(**VARIANTS[name])
    assert r.device.type == "cpu" and r.dtype == torch.float32
    depth = g[f"{name}_in_depth"].copy()
    mask = g.get(f"{name}_in_mask")
    out = r.refine_depth(depth, None, g[f"{name}_in_points3D"], g[f"{name}_in_cam_from_world"][:3], g[f"{name}_in_K"], mask)
    exp = g[f"{name}_exp_refined_depth__refine_depth"]
    assert out["num_correspondences"] == int(g[f"{name}_exp_num_correspondences__refine_depth"])
    assert bool(g[f"{name}_exp_returns_input_object"]) == (out["refined_depth"] is depth)   # early exits alias the input
    if not g[f"{name}_exp_returns_input_object"]:
        assert out["outliers_removed"] == int(g[f"{name}_exp_outliers_removed__refine_depth"])
        assert abs(out["scale_factor"] - float(g[f"{name}_exp_scale_factor__refine_depth"])) < 1e-6
    assert out["refined_depth"].dtype == np.float32
    assert np.abs(out["refined_depth"] - exp).max() <= 1e-6 * np.abs(exp).max()


def test_median_network_equals_torch_median():
    from depthdensifier_amd.depth_refiner import median3x3
    torch.manual_seed(0)
    img = torch.rand(37, 53)
    img[img < 0.2] = 0
    pad = torch.nn.functional.pad(img[None, None], (1, 1, 1, 1), mode="replicate")
    ref = torch.nn.functional.unfold(pad, 3).view(9, -1).median(dim=0).values.view(37, 53)
    assert torch.equal(median3x3(img), ref)


def test_constructor_overrides_and_config():
    from depthdensifier_amd import DepthRefiner, RefinerConfig
    import dataclasses
    cfg = RefinerConfig(min_correspondences=7, verbose=0)
    r = DepthRefiner(**dataclasses.asdict(cfg))                    # scripts/test.py:118-119 construction
    assert r.min_correspondences == 7 and r.edge_margin == 10 and r.robust
    r2 = DepthRefiner(config=cfg, edge_margin=3)
    assert r2.min_correspondences == 7 and r2.edge_margin == 3
    with pytest.raises(TypeError):
        DepthRefiner(bogus=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("default", "nosmooth", "notrobust", "nomask"))
def test_gpu_fp32_matches_golden_and_stays_on_device(g, name):
    """GPU / FP32 (dd_refine_fit + dd_refine_apply) against the outputs of the reference's own class: the same
    correspondences, the same outliers, the same scale, the refined map to 2e-4 of its range (measured 3e-5 .. 9e-5: the
    bilinear samples differ from the CPU's in the last bit and the steep segments of the transfer curve amplify that)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _refiner(**VARIANTS[name])
    assert r.device.type == "cuda"
    mask = g.get(f"{name}_in_mask")
    out = r.refine_depth(g[f"{name}_in_depth"], None, g[f"{name}_in_points3D"], g[f"{name}_in_cam_from_world"][:3],
                         g[f"{name}_in_K"], mask, return_tensor=True)
    t = out["refined_depth"]
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32
    exp = g[f"{name}_exp_refined_depth__refine_depth"]
    assert out["num_correspondences"] == int(g[f"{name}_exp_num_correspondences__refine_depth"])
    assert out["outliers_removed"] == int(g[f"{name}_exp_outliers_removed__refine_depth"])
    assert abs(out["scale_factor"] - float(g[f"{name}_exp_scale_factor__refine_depth"])) <= 1e-6 * abs(out["scale_factor"])
    diff = np.abs(t.cpu().numpy() - exp)
    assert np.array_equal(t.cpu().numpy() > 0, exp > 0)
    assert diff.max() <= 2e-4 * exp.max() and np.median(diff) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_gpu_fp16_default_mode(g, name):
    """The reference's DEFAULT on a GPU is half precision (``use_fp16=True``, depth_refiner.py:85-86).  No golden
    exists for it (the goldens are the reference on CPU/FP32); this pins the contract -- float32 output, early exits
    alias the input, same correspondence counts up to the few that sit on an edge -- and closeness to the FP32
    result at half-precision accuracy."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd.depth_refiner import DepthRefiner
    r = DepthRefiner(adaptive_correspondences=False, **VARIANTS[name])          # use_fp16 defaults to True
    assert r.device.type == "cuda" and r.dtype == torch.float16
    depth = g[f"{name}_in_depth"].copy()
    out = r.refine_depth(depth, None, g[f"{name}_in_points3D"], g[f"{name}_in_cam_from_world"][:3], g[f"{name}_in_K"],
                         g.get(f"{name}_in_mask"))
    exp = g[f"{name}_exp_refined_depth__refine_depth"]
    aliased = bool(g[f"{name}_exp_returns_input_object"])
    assert aliased == (out["refined_depth"] is depth)
    if aliased:
        return
    assert out["refined_depth"].dtype == np.float32 and out["refined_depth"].shape == exp.shape
    assert abs(out["num_correspondences"] - int(g[f"{name}_exp_num_correspondences__refine_depth"])) <= 0.05 * out["num_correspondences"] + 3
    assert abs(out["scale_factor"] - float(g[f"{name}_exp_scale_factor__refine_depth"])) <= 2e-2 * abs(out["scale_factor"])
    valid = exp > 0
    assert np.array_equal(out["refined_depth"] > 0, valid) or (np.count_nonzero((out["refined_depth"] > 0) != valid) <= 1e-3 * valid.size)
    rel = np.abs(out["refined_depth"][valid] - exp[valid]) / exp[valid]
    assert np.median(rel) <= 5e-3 and np.quantile(rel, 0.99) <= 5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("skip", (False, True))
@pytest.mark.parametrize("shape", [(96, 128), (37, 53), (1, 1), (33, 65)])
@pytest.mark.parametrize("with_mask", (True, False))
def test_hip_apply_kernel_equals_tensor_path(shape, skip, with_mask):
    """dd_refine_apply (csrc/ddrefine.hip) vs the tensor formulation evaluated on the CPU: bit-exact with the kernels' blend, two ulps
    of t from the reference's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd.depth_refiner import DepthRefiner
    g = torch.Generator().manual_seed(shape[0] * 1000 + shape[1])
    H, W = shape
    depth = torch.rand((H, W), generator=g) * 4 + 0.2
    depth[torch.rand((H, W), generator=g) < 0.1] = 0.0
    mask = (torch.rand((H, W), generator=g) < 0.8) if with_mask else None
    x = torch.rand(300, generator=g) * 3 + 0.5            # knots cover only part of the depth range (clamping)
    x[5] = x[6]                                           # duplicate knot: dx == 0 branch
    y = 2.0 * x + 0.3 * torch.rand(300, generator=g)
    y[5] = y[6]                                           # (equal y too: argsort may order equal keys either way)
    r = DepthRefiner(use_fp16=False, skip_smoothing=skip)
    m_gpu = None if mask is None else mask.cuda()
    got = r._apply_curve_hip(depth.cuda(), m_gpu, x.cuda(), y.cuda()).cpu()
    cpu = DepthRefiner.__new__(DepthRefiner)
    cpu.__dict__.update(r.__dict__); cpu.device = torch.device("cpu"); cpu.dtype = torch.float32
    m_cpu = mask if mask is not None else depth > 0
    if int(m_cpu.sum()) < 4:
        pytest.skip("fewer than 4 masked pixels: the tensor path takes its own branch")
    # bit-exact against the tensor formulation with the kernels' blend (t = (d - x0) * (1 / dx): round 6) ...
    want = cpu._apply_curve(depth.clone(), m_cpu, x, y, reciprocal=True)
    assert torch.equal(got, want)
    # ... and within two ulps of t -- 1.2e-7 of the interval's |dy| -- of the reference's quotient form (:160-176)
    ref = cpu._apply_curve(depth.clone(), m_cpu, x, y)
    assert float((got - ref).abs().max()) <= 3e-7 * float(y.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("DD_APPLY_SEEDS", "24"))))      # soak: DD_APPLY_SEEDS=1000
def test_hip_apply_kernel_random_against_the_tensor_path(seed):
    """dd_refine_apply against the tensor formulation on the CPU -- the independent side of the round-5 rewrite (both GPU kernels
    share the grid look-up and the shared-column medians): random sizes (several 64x16 tiles, ragged edges), 2..2048 knots in
    uniform / tight / clustered / narrow sets, with and without NaN (a NaN anywhere in a tile selects the careful medians),
    +-inf, negative and zero depths, float16 / float32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd.depth_refiner import DepthRefiner
    rng = np.random.default_rng(80_000 + seed)
    g = torch.Generator().manual_seed(80_000 + seed)
    H, W = (int(rng.integers(1, 70)), int(rng.integers(1, 300))) if rng.uniform() < 0.5 else (int(rng.integers(20, 200)), int(rng.integers(60, 200)))
    depth = torch.rand((H, W), generator=g) * 4 + 0.2
    depth[torch.rand((H, W), generator=g) < 0.1] = 0.0
    flat = depth.view(-1)
    if rng.uniform() < 0.5:
        flat[::89] = float("nan")
    flat[3::97] = float("inf"); flat[5::101] = -1.0; flat[7::211] = float("-inf")
    half = bool(rng.uniform() < 0.3)
    if half:
        depth = depth.half().float()                      # (the kernel reads float16, the tensor path the same values as float32)
    mask = (torch.rand((H, W), generator=g) < float(rng.uniform(0.3, 1.0))) if rng.uniform() < 0.7 else None
    n = int(rng.choice([2, 3, 40, 500, 512, 513, 2048]))
    x = torch.rand(n, generator=g) * 3 + 0.5
    kind = str(rng.choice(["uniform", "uniform", "tight", "two_clusters", "narrow"]))
    if kind == "tight":
        x = 2.0 + 1e-4 * (x - 0.5)
    elif kind == "two_clusters":
        x = torch.where(torch.arange(n) % 2 == 0, 0.7 + 1e-3 * x, 3.9 + 1e-3 * x)
    elif kind == "narrow":
        x = 1.9 + 0.1 * x
    y = 2.0 * x + 0.3 * torch.frac(x * 7919.0)            # a function of x: equal knots (frequent in the tight sets) carry equal values,
    skip = bool(rng.uniform() < 0.3)                      # whichever way a sort orders them
    r = DepthRefiner(use_fp16=False, skip_smoothing=skip)
    d_gpu = depth.half().cuda() if half else depth.cuda()
    got = r._apply_curve_hip(d_gpu, None if mask is None else mask.cuda(), x.cuda(), y.cuda()).cpu()
    cpu = DepthRefiner.__new__(DepthRefiner)
    cpu.__dict__.update(r.__dict__); cpu.device = torch.device("cpu"); cpu.dtype = torch.float32
    m_cpu = mask if mask is not None else depth > 0
    if int(m_cpu.sum()) < 4:
        pytest.skip("fewer than 4 masked pixels: the tensor path takes its own branch")
    want = cpu._apply_curve(depth.clone(), m_cpu, x, y, reciprocal=True)      # the kernels' blend (round 6), bit for bit
    nan = torch.isnan(want)
    assert torch.equal(torch.isnan(got), nan)             # (a NaN is a NaN: the CPU's carries another payload)
    assert torch.equal(got[~nan].view(torch.int32), want[~nan].view(torch.int32))
    ref = cpu._apply_curve(depth.clone(), m_cpu, x, y)    # the reference's quotient form: two ulps of t away at most
    assert torch.equal(torch.isnan(ref), nan) and float((got[~nan] - ref[~nan]).abs().max()) <= 3e-7 * float(y.abs().max())


def _tensor_fit_cpu(r, depth, pts, E, K):
    """The correspondence half as the tensor formulation evaluates it on the CPU (= the reference's op sequence, which
    the CPU golden test pins): returns z_mono, z_metric (kept, in order), counts and the scale."""
    import torch.nn.functional as F
    cpu = type(r).__new__(type(r))
    cpu.__dict__.update(r.__dict__); cpu.device = torch.device("cpu"); cpu.dtype = torch.float32
    d = torch.as_tensor(depth, dtype=torch.float32)
    uv, z = cpu._project_sparse(torch.as_tensor(pts, dtype=torch.float32), torch.as_tensor(E, dtype=torch.float32), torch.as_tensor(K, dtype=torch.float32))
    h, w = d.shape
    e = cpu.edge_margin
    ok = (uv[:, 0] >= e) & (uv[:, 0] < w - e) & (uv[:, 1] >= e) & (uv[:, 1] < h - e) & (z > 0)
    uv, z = uv[ok], z[ok]
    grid = torch.stack([uv[:, 0] / (w - 1) * 2 - 1, uv[:, 1] / (h - 1) * 2 - 1], dim=-1)[None, None]
    sampled = F.grid_sample(d[None, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True).reshape(-1)
    has = sampled > 0
    z_mono, z_metric = sampled[has], z[has]
    pos, removed = int(has.sum()), 0
    if cpu.robust and len(z_mono) > 10:
        z_metric, z_mono, removed = cpu._iqr_inliers(z_metric, z_mono)
    scale = float(torch.median(z_metric / (z_mono + 1e-6))) if len(z_mono) else 1.0
    return z_mono, z_metric, int(ok.sum()), pos, removed, scale


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("default", "notrobust", "nomask"))
def test_hip_fit_kernel_equals_tensor_path_on_the_golden_inputs(g, name):
    """dd_refine_fit (one launch) against the tensor formulation on the CPU, on the inputs of the reference goldens: same
    correspondences in the same order (counts identical, values to float32 rounding), same outlier count, same scale."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _refiner(**VARIANTS[name])
    depth, pts = g[f"{name}_in_depth"], g[f"{name}_in_points3D"]
    E, K = g[f"{name}_in_cam_from_world"][:3], g[f"{name}_in_K"]
    zm, zt, inb, pos, kept, removed, scale = r._fit_hip(torch.as_tensor(depth).cuda(), pts, E, K)
    em, et, e_inb, e_pos, e_removed, e_scale = _tensor_fit_cpu(r, depth, pts, E, K)
    assert (inb, pos, kept, removed) == (e_inb, e_pos, len(em), e_removed)
    assert np.allclose(zm.cpu().numpy(), em.numpy(), rtol=2e-6, atol=0) and np.allclose(zt.cpu().numpy(), et.numpy(), rtol=2e-6, atol=0)
    assert abs(scale - e_scale) <= 2e-6 * abs(e_scale)
    assert kept == int(g[f"{name}_exp_num_correspondences__refine_depth"])
    assert removed == int(g[f"{name}_exp_outliers_removed__refine_depth"])
    assert abs(scale - float(g[f"{name}_exp_scale_factor__refine_depth"])) <= 2e-6 * abs(scale)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("nosmooth", "default", "notrobust"))
def test_gpu_fp32_error_budget_step_by_step(g, name):
    """Where the 3e-5 .. 9e-5 of range between the GPU FP32 refiner and the reference's golden comes from, as a
    budget that every pixel must meet:

      step 1  the knots of the transfer curve: the sampled depths / projected depths of the sparse points, GPU against the tensor
              formulation on the CPU (the reference's op sequence): relative 2e-6 -- last-bit differences of the projection's
              dot products and of the four-term bilinear sum (``depth_refiner.py:99-112, 266-272``);
      step 2  the curve evaluated through these knots: a pixel between knots i, i+1 moves by at most
              |dy_i| + |dy_i+1| + |slope_i| (|dx_i| + |dx_i+1|) when the knots move by dx, dy -- evaluated here exactly, in
              float64, as the difference of the two curves (``:141-178``).  Sparse points that sample almost the same depth make
              segments with slopes in the hundreds: THAT is the amplification, and it is a property of the reference's curve;
      step 3  the float32 evaluation of the curve and the 3x3 median (``:180-205``): 4e-6 of range on top (the median of values
              that each meet their budget meets the largest budget of its window).

    Every pixel: |GPU - golden| <= (step 2 of its window) + 4e-6 range.  Pixels on segments of slope <= 8: <= 2e-5 of range."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _refiner(**VARIANTS[name])
    depth, pts = g[f"{name}_in_depth"], g[f"{name}_in_points3D"]
    E, K = g[f"{name}_in_cam_from_world"][:3], g[f"{name}_in_K"]
    mask = g.get(f"{name}_in_mask")
    exp = g[f"{name}_exp_refined_depth__refine_depth"]
    out = r.refine_depth(depth, None, pts, E, K, mask, return_tensor=True)["refined_depth"].cpu().numpy()
    rng = float(exp.max())
    # step 1
    zm, zt, *_ = r._fit_hip(torch.as_tensor(depth).cuda(), pts, E, K)
    em, et, *_ = _tensor_fit_cpu(r, depth, pts, E, K)
    gx, gy = zm.double().cpu(), zt.double().cpu()
    cx, cy = em.double(), et.double()
    assert len(gx) == len(cx) and float(((gx - cx).abs() / cx.abs()).max()) <= 2e-6 and float(((gy - cy).abs() / cy.abs()).max()) <= 2e-6
    # step 2: both curves in float64 at every masked pixel
    cpu64 = type(r).__new__(type(r))
    cpu64.__dict__.update(r.__dict__); cpu64.device = torch.device("cpu"); cpu64.dtype = torch.float64
    m = torch.as_tensor(mask if mask is not None else depth > 0)
    d = torch.as_tensor(depth, dtype=torch.float64)
    lut_gpu_knots = cpu64._lut_interpolate(d[m], gx, gy)
    lut_cpu_knots = cpu64._lut_interpolate(d[m], cx, cy)
    moved = torch.zeros_like(d)
    moved[m] = (lut_gpu_knots - lut_cpu_knots).abs()
    order = torch.argsort(cx)
    xs, ys = cx[order], cy[order]
    hi = torch.clamp(torch.searchsorted(xs, d[m]), 1, len(xs) - 1)
    slope = torch.zeros_like(d)
    slope[m] = ((ys[hi] - ys[hi - 1]) / torch.clamp(xs[hi] - xs[hi - 1], min=1e-6)).abs()
    if not r.skip_smoothing:      # the median of a window meets the largest budget (and sees the steepest segment) of the window
        pool = lambda t: torch.nn.functional.max_pool2d(torch.nn.functional.pad(t[None, None], (1, 1, 1, 1), mode="replicate"), 3, 1)[0, 0]
        moved, slope = pool(moved), pool(slope)
    # step 3 + the verdict
    diff = np.abs(out.astype(np.float64) - exp.astype(np.float64))
    budget = moved.numpy() + 4e-6 * rng
    assert (diff <= budget).all(), f"worst excess {float((diff - budget).max()):.3e} of range {rng:.3f}"
    gentle = (slope.numpy() <= 8.0) & m.numpy()
    assert gentle.any() and diff[gentle].max() <= 2e-5 * rng, (float(gentle.mean()), float(diff[gentle].max() / rng))
    steep = float(slope.max())
    assert diff.max() <= 2e-4 * rng and steep > 20.0        # (the steepest segment of these curves: what the 9e-5 comes from)


@pytest.mark.gpu
@pytest.mark.parametrize("holes", (False, True))
@pytest.mark.parametrize("seed, n, shape", [(1, 7000, (270, 480)), (2, 40000, (1080, 1920)), (3, 9, (64, 64)), (4, 1, (16, 16)), (5, 300, (2, 2000))])
def test_hip_fit_kernel_random_scenes(seed, n, shape, holes):
    """Larger / odd cases: tens of thousands of points (chunked stable compaction, radix select over > 4096 values), very
    few points (no IQR below 11), points behind the camera and outside the image, zeros in the depth map."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from synth import refiner_case                 # tests/golden/synth.py: own synthetic scene builder
    H, W = shape
    try:
        c = refiner_case(seed, H=H, W=W, n_pts=n) if min(H, W) > 8 else None
    except Exception:
        c = None
    rng = np.random.default_rng(seed)
    if c is None:
        c = dict(depth=rng.uniform(0.5, 3.0, (H, W)).astype(np.float32), points3D=rng.standard_normal((n, 3)) * 2 + [0, 0, 3],
                 cam_from_world=np.hstack([np.eye(3), np.zeros((3, 1))]), K=np.array([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1.0]]))
    c["depth"] = np.array(c["depth"], copy=True)
    if holes:
        c["depth"][rng.uniform(size=c["depth"].shape) < 0.05] = 0.0
    r = _refiner(edge_margin=3 if min(H, W) < 30 else 10)
    if min(H, W) <= 6:
        r.edge_margin = 0
    zm, zt, inb, pos, kept, removed, scale = r._fit_hip(torch.as_tensor(c["depth"]).cuda(), c["points3D"], c["cam_from_world"][:3], c["K"])
    em, et, e_inb, e_pos, e_removed, e_scale = _tensor_fit_cpu(r, c["depth"], c["points3D"], c["cam_from_world"][:3], c["K"])
    # a projection within float32 rounding of a bound, or a ratio within rounding of the IQR threshold, may fall either way
    assert abs(inb - e_inb) <= 2 and abs(pos - e_pos) <= 2 and abs(kept - len(em)) <= 3 + 2e-3 * len(em)
    if kept == len(em) and kept:
        # A projected coordinate differs by an ulp between two float32 evaluations (1.2e-4 px at u ~ 1500).  On a smooth map
        # that moves the sample by ~1e-7; next to a hole the sample is (weight x one neighbour) and the weight u - floor(u)
        # inherits the whole ulp: relative differences up to ~1e-3 there, in either implementation, against float64.
        rel_m = np.abs(zm.cpu().numpy() - em.numpy()) / np.abs(em.numpy())
        rel_t = np.abs(zt.cpu().numpy() - et.numpy()) / np.abs(et.numpy())
        assert rel_t.max() <= 1e-5
        assert rel_m.max() <= (5e-3 if holes else 2e-5) and np.median(rel_m) <= 1e-6
        assert abs(scale - e_scale) <= (1e-4 if holes else 1e-5) * abs(e_scale)


@pytest.mark.gpu
def test_sort_knots_kernel():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd._lib import lib
    for n in (1, 2, 3, 500, 777, 4096):
        gen = torch.Generator().manual_seed(n)
        x = torch.rand(n, generator=gen).cuda() * 5
        y = torch.rand(n, generator=gen).cuda()
        xs, ys = torch.empty_like(x), torch.empty_like(y)
        assert lib.dd_sort_knots(x.data_ptr(), y.data_ptr(), n, xs.data_ptr(), ys.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        order = torch.argsort(x)
        assert torch.equal(xs, x[order]) and torch.equal(ys, y[order])
    assert lib.dd_sort_knots(x.data_ptr(), y.data_ptr(), 5000, xs.data_ptr(), ys.data_ptr(), None) == -1
    # ties and +inf keys: equal x keep their input order, and a real +inf knot is not displaced by the padding
    x = torch.tensor([3.0, float("inf"), 1.0, 3.0, float("inf"), 0.5, 3.0], device="cuda")      # n = 7 -> one padding slot
    y = torch.arange(7, dtype=torch.float32, device="cuda") + 10
    xs, ys = torch.empty_like(x), torch.empty_like(y)
    assert lib.dd_sort_knots(x.data_ptr(), y.data_ptr(), 7, xs.data_ptr(), ys.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    order = torch.argsort(x, stable=True)
    assert torch.equal(xs, x[order]) and torch.equal(ys, y[order]), (xs, ys)


@pytest.mark.gpu
def test_many_begun_handles_keep_their_own_results(g):
    """begin_refine / finish_refine are public; more than 8 open handles used to share page-locked result slots."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd.depth_refiner import DepthRefiner
    r = _refiner()
    assert r.device.type == "cuda"
    depth, pts, E, K, mask = (g[f"default_in_{k}"] for k in ("depth", "points3D", "cam_from_world", "K", "mask"))
    handles, want = [], []
    for k in range(12):                                    # each view sees a different subset of the sparse points
        sub = pts[: max(40, len(pts) - 60 * k)]
        want.append(r.refine_depth(depth, None, sub, E[:3], K, mask))
        handles.append((sub, r.begin_refine(depth, None, sub, E[:3], K, mask)))
    for (sub, h), w in zip(reversed(handles), reversed(want)):          # finished in another order than begun
        got = r.finish_refine(h)
        assert got["num_correspondences"] == w["num_correspondences"] and got.get("outliers_removed") == w.get("outliers_removed")
        assert np.array_equal(got["refined_depth"], w["refined_depth"])
    with pytest.raises(RuntimeError):
        r.finish_refine(handles[0][1])


@pytest.mark.gpu
@pytest.mark.parametrize("skip", (False, True))
@pytest.mark.parametrize("dtype", ("float32", "float16"))
@pytest.mark.parametrize("shape, with_mask", [((96, 128), True), ((37, 53), True), ((200, 333), False), ((540, 960), True), ((5, 3000), True),
                                              ((541, 961), True), ((270, 1920), False)])
def test_fused_refine_densify_equals_apply_then_densify(shape, with_mask, dtype, skip):
    """DD_REFINE: the densify kernel applying the transfer curve to the RAW depth itself (LUT + 3x3 median over a halo held
    in LDS + mask) must give, bit for bit, what dd_refine_apply followed by the plain densify call gives -- cloud AND the
    refined map it writes for the filter cache.  Several tiles per view, ragged sizes, NaN / inf / zero raw depths, holes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    from depthdensifier_amd.depth_refiner import DepthRefiner
    H, W = shape
    V = 3
    g = torch.Generator().manual_seed(H * 7 + W)
    raw = (torch.rand((V, H, W), generator=g) * 4 + 0.2)
    raw[torch.rand((V, H, W), generator=g) < 0.05] = 0.0
    flat = raw.view(-1)
    if shape not in ((541, 961), (270, 1920)):                    # (two shapes without a NaN: tiles whose windows share their sorted columns)
        flat[::997] = float("nan")
    flat[5::1013] = float("inf"); flat[7::1019] = -1.0
    raw = raw.to(getattr(torch, dtype)).cuda()
    mask = (torch.rand((V, H, W), generator=g) < 0.85).cuda() if with_mask else None
    normal = torch.nn.functional.normalize(torch.randn((V, H, W, 3), generator=g), dim=-1).cuda()
    rgb = torch.randint(0, 256, (V, H, W, 3), generator=g, dtype=torch.uint8).cuda()
    params = np.tile([0.9 * W, 0.9 * W, W / 2.0, H / 2.0], (V, 1))
    from synth import random_pose
    rng = np.random.default_rng(H + W)
    E = np.stack([random_pose(rng) for _ in range(V)])
    r = DepthRefiner(use_fp16=False, skip_smoothing=skip)
    curves, refined = [], []
    for v in range(V):
        n = 120 + 130 * v
        x = torch.rand(n, generator=g) * 3 + 0.5
        y = 2.0 * x + 0.3 * torch.rand(n, generator=g)
        kx, ky = r._sorted_knots(x.cuda(), y.cuda())
        curves.append((kx, ky, skip))
        m = mask[v] if mask is not None else raw[v] > 0
        refined.append(r._apply_curve_hip(raw[v], m if mask is not None else None, x.cuda(), y.cuda()))
    refined = torch.stack(refined)
    want = dd.unproject_views(refined, params, E, mask=mask, normal=normal, rgb=rgb, view_index=True, capacity="max")
    batch = dd.ViewBatch(raw, params, E, mask=mask, normal=normal, rgb=rgb, refine=curves, refined_out=True)
    b = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=True, view_index=True)
    b.append(batch)
    got = b.finish()
    assert torch.equal(got.view_offsets, want.view_offsets)
    for name in ("points", "colors", "normals", "pixel_index", "view_index"):
        assert torch.equal(getattr(got, name), getattr(want, name)), name
    assert torch.equal(batch.refined.view(torch.int32), refined.view(torch.int32))          # NaNs included, bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("DD_REFINE_SEEDS", "16"))))      # soak: DD_REFINE_SEEDS=200
def test_fused_refine_random(seed):
    """Seeded sweep of the fused refine stage against dd_refine_apply + plain densify: random view sizes (rows shorter than
    a wave, widths around the vector / tile sizes), 2..512 knots with repeated x values (the dx == 0 branch), field subsets,
    both record forms, float16 / float32 raw depth, special values.  Everything must be equal bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    from depthdensifier_amd.depth_refiner import DepthRefiner
    from synth import random_pose
    rng = np.random.default_rng(50_000 + seed)
    g = torch.Generator().manual_seed(50_000 + seed)
    V = int(rng.integers(1, 4))
    H, W = (int(rng.integers(2, 60)), int(rng.integers(2, 400))) if rng.uniform() < 0.6 else (int(rng.integers(60, 260)), int(rng.integers(2, 130)))
    if H * W < 4:
        W = 4
    skip = bool(rng.uniform() < 0.3)
    dtype = torch.float16 if rng.uniform() < 0.4 else torch.float32
    raw = torch.rand((V, H, W), generator=g) * 4 + 0.2
    raw[torch.rand((V, H, W), generator=g) < 0.1] = 0.0
    flat = raw.view(-1)
    r5 = np.random.default_rng(60_000 + seed)                     # round 5's draws: a stream of their own, the older ones keep theirs
    if r5.uniform() < 0.5:
        flat[::97] = float("nan")                                 # (a NaN anywhere in a tile sends its windows down the careful path:
    flat[5::101] = float("inf"); flat[7::103] = -1.0              #  half of the cases have none -- the shared-column medians)
    knots_kind = str(r5.choice(["uniform", "uniform", "tight", "two_clusters", "narrow"]))
    raw = raw.to(dtype).cuda()
    mask = (torch.rand((V, H, W), generator=g) < float(rng.uniform(0.2, 1.0))).cuda() if rng.uniform() < 0.7 else None
    normal = torch.nn.functional.normalize(torch.randn((V, H, W, 3), generator=g), dim=-1).cuda() if rng.uniform() < 0.6 else None
    rgb = torch.randint(0, 256, (V, H, W, 3), generator=g, dtype=torch.uint8).cuda() if rng.uniform() < 0.6 else None
    params = np.tile([0.9 * W, 0.9 * W, W / 2.0, H / 2.0], (V, 1))
    E = np.stack([random_pose(rng) for _ in range(V)])
    r = DepthRefiner(use_fp16=False, skip_smoothing=skip)
    curves, refined = [], []
    for v in range(V):
        n = int(rng.choice([2, 3, 17, 200, 511, 512]))
        x = torch.rand(n, generator=g) * 3 + 0.5
        if n > 4:
            x[1::3] = x[0::3][: len(x[1::3])]                     # repeated knots
        if knots_kind == "tight":                                 # a range far below 1/128 of the values: the grid of buckets is switched off
            x = 2.0 + 1e-4 * (x - 0.5)
        elif knots_kind == "two_clusters":                        # nearly all knots inside one or two buckets: long bisections behind the grid
            x = torch.where(torch.arange(n) % 2 == 0, 0.7 + 1e-3 * x, 3.9 + 1e-3 * x)
        elif knots_kind == "narrow":                              # most depths lie outside the knots' range
            x = 1.9 + 0.1 * x
        y = 2.0 * x + 0.3 * torch.rand(n, generator=g)
        kx, ky = r._sorted_knots(x.cuda(), y.cuda())
        curves.append((kx, ky, skip))
        refined.append(r._apply_curve_hip(raw[v], None if mask is None else mask[v], x.cuda(), y.cuda()))
    refined = torch.stack(refined)
    record = str(rng.choice(["rows", "xyz_rgba"]))
    want = dd.unproject_views(refined, params, E, mask=mask, normal=normal, rgb=rgb, view_index=True, capacity="max", record=record)
    batch = dd.ViewBatch(raw, params, E, mask=mask, normal=normal, rgb=rgb, refine=curves, refined_out=True)
    rows = record == "rows"
    b = dd.CloudBuilder(batch.max_points, points=rows, normals=normal is not None, colors=rgb is not None and rows, pixel_index=True,
                        view_index=True, packed=not rows)
    b.append(batch)
    got = b.finish()
    assert torch.equal(got.view_offsets, want.view_offsets)
    for name in ("points", "colors", "normals", "pixel_index", "view_index", "packed"):
        a_, b_ = getattr(got, name), getattr(want, name)
        assert (a_ is None) == (b_ is None), name
        if a_ is not None:
            assert torch.equal(a_.contiguous().view(torch.uint8), b_.contiguous().view(torch.uint8)), name
    assert torch.equal(batch.refined.view(torch.int32), refined.view(torch.int32))
