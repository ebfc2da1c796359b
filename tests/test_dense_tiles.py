"""The list-free path for dense tiles (``dense_wave`` in ``csrc/ddcore.hip``): tiles whose pixels all survive are written
as line-aligned 16-byte pieces from an LDS staging area (xyz) and as shifted copies (normals, colours), per wave.  Its
rows must be the rows of the list path (``tuning`` bit 128 switches the dense path on; it is off by default -- it writes the same rows with a third of the
instructions but is not faster) bit for bit, for every phase
of the first row against the 128-byte lines of every output array, with the capacity cutting a dense tile anywhere, in
the single-pass and the two-pass kernels, for float32 and float16 depth -- and equal to the oracle
(``scripts/test.py:203-233`` restated)."""

import numpy as np
import pytest

from test_gpu_parity import assert_cloud, scene_radius

pytestmark = pytest.mark.gpu

DENSE = 128          # DDViewBatch.tuning bit 128: dense tiles take the list-free path (off by default)
ASSUME = 1 << 17     # DDViewBatch.tuning bit 17: count-free plan, verified by the scatter pass


@pytest.fixture(scope="module")
def dd():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd
    return depthdensifier_amd


def _ring(V):
    from test_gpu_parity import _ring_poses
    return _ring_poses(V)


def _scene(V, H, W, dtype, seed, mask_kind):
    """Depth without holes; ``mask_kind``: None (every tile dense), "blob" (a rectangle cut out of every view: dense,
    partial and empty tiles side by side), "speck" (one pixel missing per view: one partial tile among dense ones)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    depth = torch.empty((V, H, W), device="cuda").uniform_(0.5, 8.0, generator=g).to(dtype)
    normal = torch.nn.functional.normalize(torch.randn((V, H, W, 3), device="cuda", generator=g), dim=-1)
    rgb = torch.randint(0, 256, (V, H, W, 3), device="cuda", generator=g, dtype=torch.uint8)
    mask = None
    if mask_kind == "blob":
        mask = torch.ones((V, H, W), dtype=torch.bool, device="cuda")
        for v in range(V):
            y0, x0 = (17 * v + 5) % (H // 2), (29 * v + 3) % (W // 2)
            mask[v, y0:y0 + H // 3, x0:x0 + W // 2] = False
            mask[v, H - 1 - v, :] = False
    elif mask_kind == "speck":
        mask = torch.ones((V, H, W), dtype=torch.bool, device="cuda")
        for v in range(V):
            mask[v, (H // 2 + v) % H, (W // 3 + 7 * v) % W] = False
    return depth, mask, normal, rgb


FIELDS = ("points", "normals", "colors", "pixel_index", "view_index")


def _build(dd, batch, cap, start, fields, fill=True):
    """A cloud of capacity ``cap`` in buffers 64 rows longer (rows behind the capacity must stay untouched)."""
    import torch
    bufs = {}
    for name in ("points", "normals", "colors", "pixel_index", "view_index", "packed"):
        bufs[name] = None
        if name in fields:
            tail, dt = dd.CloudBuilder.FIELDS[name]
            bufs[name] = torch.empty((cap + 64,) + tail, dtype=dt, device="cuda")
            if fill:
                bufs[name].fill_(201 if name == "colors" else -7)
    b = dd.CloudBuilder(cap, normals="normals" in fields, colors="colors" in fields, pixel_index="pixel_index" in fields,
                        view_index="view_index" in fields, start=start, buffers=bufs)
    b.append(batch)
    torch.cuda.synchronize()
    return b


def _arrays(b):
    return dict(points=b.xyz, normals=b.normal, colors=b.rgb, pixel_index=b.pix, view_index=b.view)


@pytest.mark.parametrize("dtype_name", ("float32", "float16"))
@pytest.mark.parametrize("mask_kind", (None, "blob", "speck"))
@pytest.mark.parametrize("tuning", (0, 4))
def test_dense_path_equals_list_path(dd, dtype_name, mask_kind, tuning):
    import torch
    dtype = getattr(torch, dtype_name)
    V, H, W = 3, 150, 331                       # 49 650 pixels per view: 4 tiles of 12 288 + a ragged one (two-pass: 12 + 1)
    depth, mask, normal, rgb = _scene(V, H, W, dtype, 11, mask_kind)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring(V)
    mk = lambda tun: dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, tuning=tun)
    dense, plain = mk(tuning | DENSE), mk(tuning)
    n = V * H * W if mask is None else int(mask.sum())
    ref = _build(dd, plain, n, None, FIELDS, fill=False)
    assert int(ref.cursor.item()) == n
    want = _arrays(ref)
    for start in list(range(0, 34)) + [127, 1000003]:
        for cut in (0, 1, 5000):
            cap = start + n - cut
            b = _build(dd, dense, cap, start, FIELDS)
            assert int(b.cursor.item()) == start + n
            kept = n - cut
            for name, t in _arrays(b).items():
                w = want[name][:kept]
                assert torch.equal(t[start:start + kept].view(torch.uint8), w.view(torch.uint8)), (start, cut, name)
                assert bool((t[:start] == (201 if name == "colors" else -7)).all()), (start, cut, name, "rows in front of the cloud")
                assert bool((t[start + kept:] == (201 if name == "colors" else -7)).all()), (start, cut, name, "rows behind the capacity")


@pytest.mark.parametrize("fields", (("points",), ("points", "normals"), ("points", "colors"), ("points", "pixel_index"),
                                    ("points", "colors", "view_index")))
def test_dense_path_field_subsets_and_odd_base_pointers(dd, fields):
    """Every subset runs its own instantiation; sub-tensor views of larger buffers shift each array's base pointer off
    the 128-byte grid by a different amount."""
    import torch
    V, H, W = 2, 128, 400
    depth, mask, normal, rgb = _scene(V, H, W, torch.float32, 5, "speck")
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring(V)
    n = int(mask.sum())
    outs = {}
    for tun in (DENSE, 0):
        batch = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, tuning=tun)
        pad = {"points": 1, "normals": 3, "colors": 5, "pixel_index": 7, "view_index": 9}      # rows in front: 12 / 36 / 15 / 28 / 36 bytes
        bufs = {}
        for name in fields:
            tail, dt = dd.CloudBuilder.FIELDS[name]
            big = torch.full((n + 16,) + tail, 3, dtype=dt, device="cuda")
            bufs[name] = big[pad[name]:pad[name] + n]
        for name in ("points", "normals", "colors", "pixel_index", "view_index", "packed"):
            bufs.setdefault(name, None)
        b = dd.CloudBuilder(n, normals="normals" in fields, colors="colors" in fields, pixel_index="pixel_index" in fields,
                            view_index="view_index" in fields, buffers=bufs)
        b.append(batch)
        torch.cuda.synchronize()
        outs[tun] = {k: v.clone() for k, v in bufs.items() if v is not None}
    for name in fields:
        assert torch.equal(outs[0][name].view(torch.uint8), outs[DENSE][name].view(torch.uint8)), name


@pytest.mark.parametrize("dtype_name", ("float32", "float16"))
def test_dense_tiles_against_the_oracle(dd, dtype_name):
    from oracle import densify_oracle as orc
    rng = np.random.default_rng(3)
    V, H, W = 2, 96, 640                        # 61 440 pixels per view = 5 dense tiles exactly
    depth = rng.uniform(0.5, 6.0, (V, H, W)).astype(dtype_name)
    normal = rng.standard_normal((V, H, W, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (V, H, W, 3), dtype=np.uint8)
    params = np.stack([[500.0, 510.0, 320.0, 48.0], [480.0, 470.0, 300.5, 50.25]])
    E = _ring(V)
    for tuning in (DENSE, DENSE | 4):
        cloud = dd.unproject_views(depth, params, E, normal=normal, rgb=rgb, view_index=True, tuning=tuning, capacity="max")
        ref = orc.densify_scene_script(depth, params, E, normal=normal, rgb=rgb)
        assert_cloud(cloud, ref, scene_radius(E, depth))


def test_narrow_views_take_the_list_path(dd):
    """width < 64: the dense path's x += 64 stepping does not apply; the result is the oracle's all the same."""
    from oracle import densify_oracle as orc
    rng = np.random.default_rng(4)
    V, H, W = 1, 1024, 48
    depth = rng.uniform(0.5, 6.0, (V, H, W)).astype(np.float32)
    params = np.array([[40.0, 41.0, 24.0, 512.0]])
    E = _ring(V)
    cloud = dd.unproject_views(depth, params, E, view_index=True, tuning=DENSE)
    assert_cloud(cloud, orc.densify_scene_script(depth, params, E), scene_radius(E, depth))


@pytest.mark.parametrize("shape", ((3, 211, 307), (1, 70, 70), (2, 128, 400)))
def test_interleaved_scatter_order_writes_the_same_cloud(dd, shape):
    """``DDViewBatch.tuning`` bits 8-13: the scatter pass of the two-pass path takes tiles of K = 2 .. 64 stretches of the batch in
    turn (consecutive workgroups then write K distant regions of the output -- different classes of HBM for a cloud placed in
    thirds).  Every tile knows its rows before the pass starts, so the order cannot change a byte; K may exceed the number of
    tiles and need not divide it."""
    import torch
    V, H, W = shape
    depth, mask, normal, rgb = _scene(V, H, W, torch.float32, 21, "blob")
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring(V)
    n = int(mask.sum())
    ref = _build(dd, dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, tuning=4), n, None, FIELDS, fill=False)
    want = _arrays(ref)
    for k in list(range(1, 16)) + [20, 44, 63]:
        b = _build(dd, dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, tuning=4 | (k << 8)), n, None, FIELDS)
        assert int(b.cursor.item()) == n
        for name, t in _arrays(b).items():
            assert torch.equal(t[:n].view(torch.uint8), want[name][:n].view(torch.uint8)), (k + 1, name)
            assert bool((t[n:] == (201 if name == "colors" else -7)).all()), (k + 1, name)


@pytest.mark.parametrize("dtype_name", ("float32", "float16"))
@pytest.mark.parametrize("shape", ((3, 150, 331), (1, 70, 70), (5, 64, 64), (2, 128, 400)))
def test_assume_dense_scatter_writes_the_counted_cloud(dd, dtype_name, shape):
    """``DDViewBatch.tuning`` bit 17: no counting pass -- the plan is arithmetic (every visited pixel taken as valid) and the scatter
    pass verifies it.  On maps without holes the cloud is the counted cloud bit for bit: ragged last tiles, views of exactly one
    tile, a cloud that starts at an odd row, two batches chained, all fields, with the dense path and the interleaved order."""
    import torch
    dtype = getattr(torch, dtype_name)
    V, H, W = shape
    depth, _, normal, rgb = _scene(V, H, W, dtype, 31, None)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring(V)
    n = V * H * W
    ref = _build(dd, dd.ViewBatch(depth, params, E, normal=normal, rgb=rgb), n, None, FIELDS, fill=False)
    want, want_off = _arrays(ref), ref._offsets[0].clone()
    for extra in (0, DENSE, DENSE | (14 << 8), 2 << 8):
        b = _build(dd, dd.ViewBatch(depth, params, E, normal=normal, rgb=rgb, tuning=ASSUME | extra), n, None, FIELDS)
        assert int(b.cursor.item()) == n and b.check() == n and b.healed == 0 and b.dense_misses == 0
        assert torch.equal(b._offsets[0], want_off)
        for name, t in _arrays(b).items():
            assert torch.equal(t[:n].view(torch.uint8), want[name][:n].view(torch.uint8)), (extra, name)
            assert bool((t[n:] == (201 if name == "colors" else -7)).all()), (extra, name)
    half = max(V // 2, 1)
    b = dd.CloudBuilder(37 + n, normals=True, colors=True, pixel_index=True, view_index=True, start=37)
    b.append(dd.ViewBatch(depth[:half], params[:half], E[:half], normal=normal[:half], rgb=rgb[:half], tuning=ASSUME))
    if half < V:
        b.append(dd.ViewBatch(depth[half:], params[half:], E[half:], normal=normal[half:], rgb=rgb[half:], tuning=ASSUME | DENSE, view_index_base=half))
    got = b.finish()
    assert len(got) == n and b.healed == 0
    for name in FIELDS:
        assert torch.equal(getattr(got, name).view(torch.uint8), want[name][:n].view(torch.uint8)), name


@pytest.mark.parametrize("hole", ("depth", "mask", "last_pixel"))
def test_a_dense_guess_that_misses_is_redone(dd, hole):
    """One invalid pixel anywhere voids a batch run as 'assume dense': the tile that holds it sets the workspace's error word to 2;
    ``check()`` / ``finish()`` redo the batch through the counting path (``healed``, ``dense_misses``) and the cloud is the counted
    cloud; without the redo the error surfaces."""
    import torch
    V, H, W = 4, 150, 331
    depth, _, normal, rgb = _scene(V, H, W, torch.float32, 37, None)
    mask = None
    if hole == "depth":
        depth[2, 77, 200] = 0.0
    elif hole == "last_pixel":
        depth[V - 1, H - 1, W - 1] = float("nan")
    else:
        mask = torch.ones((V, H, W), dtype=torch.bool, device="cuda")
        mask[1, 3, 5] = False
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring(V)
    n = V * H * W - 1
    want = _arrays(_build(dd, dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb), n, None, FIELDS, fill=False))
    b = _build(dd, dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, tuning=ASSUME), n, None, FIELDS)
    with pytest.raises(RuntimeError, match="not dense"):
        b.check_async().result(heal=False)
    b2 = _build(dd, dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, tuning=ASSUME | DENSE), n, None, FIELDS)
    assert b2.check() == n and b2.healed == 1 and b2.dense_misses == 1
    for name, t in _arrays(b2).items():
        assert torch.equal(t[:n].view(torch.uint8), want[name][:n].view(torch.uint8)), name


def test_a_blocked_cloud_of_points_is_filled_in_thirds(dd):
    """A cloud of points only, placed with its thirds in three classes, chooses the interleaved two-pass path by itself for a large
    batch (``CloudBuilder.fuse_tuning``) and leaves small batches, explicit choices and clouds with normals alone; the cloud it
    builds is the single-pass cloud."""
    import torch
    V, H, W = 40, 1080, 1920                                     # 83 M pixels: above half of a lowered threshold
    g = torch.Generator(device="cuda").manual_seed(5)
    depth = torch.empty((V, H, W), device="cuda", dtype=torch.float16).uniform_(0.5, 8.0, generator=g)
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = _ring(V)
    holes = depth.clone()
    holes[:, 100:300, 200:900] = 0
    mask = torch.ones((V, H, W), dtype=torch.bool, device="cuda")
    mask[:, 100:300, 200:900] = False
    old = dd.CloudBuilder.INTERLEAVE_MIN_ROWS
    dd.CloudBuilder.INTERLEAVE_MIN_ROWS = 64 << 20
    K1 = dd.CloudBuilder.INTERLEAVE_REGIONS - 1
    try:
        # a masked batch: counted, the scatter interleaved
        masked = dd.ViewBatch(depth, params, E, mask=mask)
        b = dd.CloudBuilder(masked.max_points, pixel_index=False, placement="probed")
        assert b.placement.layout == "blocked", b.placement.as_dict()
        tun = b.fuse_tuning(masked)
        assert tun & 4 and not tun & ASSUME and tun & DENSE and (tun >> 8) & 63 == K1
        # (bit 128 -- dense tiles without a list -- is set for every batch into a cloud of points only since round 5)
        assert b.fuse_tuning(dd.ViewBatch(depth[:1], params[:1], E[:1])) == DENSE             # a small batch (2 M pixels): the fused single pass
        assert b.fuse_tuning(dd.ViewBatch(depth[:4], params[:4], E[:4], mask=mask[:4])) == DENSE  # a masked batch below half the threshold: too
        assert b.fuse_tuning(dd.ViewBatch(depth, params, E, tuning=8)) == 8                   # an explicit choice stands
        b.append(masked)
        got_masked = b.finish()
        assert b.healed == 0
        # unmasked maps without holes: no counting pass, the guess holds
        dense = dd.ViewBatch(depth, params, E)
        tun = b.fuse_tuning(dense)
        assert tun & ASSUME and not tun & 4 and tun & DENSE and (tun >> 8) & 63 == K1
        b.reset(); b.append(dense)
        got_dense = b.finish()
        assert b.healed == 0 and b.dense_misses == 0 and len(got_dense) == V * H * W
        got_dense_points = got_dense.points.clone()
        # unmasked maps WITH holes: the guess misses once, the batch is redone, and this cloud stops guessing
        holed = dd.ViewBatch(holes, params, E)
        assert b.fuse_tuning(holed) & ASSUME
        b.reset(); b.append(holed)
        got_holed = b.finish()
        assert b.healed == 1 and b.dense_misses == 1
        tun = b.fuse_tuning(holed)
        assert tun & 4 and not tun & ASSUME
        assert (b.guess_policy.hits, b.guess_policy.misses) == (1, 1)
        # the score belongs to whoever owns the policy object: a cloud that SHARES it stops guessing once the misses outnumber the
        # guesses that held; a cloud with a policy of its own (the default) knows nothing of it
        b.guess_policy.missed()
        other = dd.CloudBuilder(dense.max_points, pixel_index=False, placement="first", guess_policy=b.guess_policy)
        assert not other.fuse_tuning(dense) & ASSUME
        assert dd.CloudBuilder(dense.max_points, pixel_index=False, placement="first").fuse_tuning(dense) & ASSUME
        # plainly allocated arrays guess too (an unmasked batch, no normals), and run masked batches in the single pass
        plain = dd.CloudBuilder(holed.max_points, pixel_index=False, placement="first")
        assert plain.fuse_tuning(holed) & ASSUME and not plain.fuse_tuning(holed) & 4 and plain.fuse_tuning(masked) == DENSE
    finally:
        dd.CloudBuilder.INTERLEAVE_MIN_ROWS = old
    # the references: the single pass, asked for explicitly
    plain.append(dd.ViewBatch(holes, params, E, tuning=8))
    want = plain.finish()
    assert plain.healed == 0
    assert len(got_holed) == len(want) and torch.equal(got_holed.view_offsets, want.view_offsets)
    assert torch.equal(got_holed.points.view(torch.int32), want.points.view(torch.int32))
    assert len(got_masked) == len(want)
    plain.reset(); plain.append(dd.ViewBatch(depth, params, E, tuning=8))
    want = plain.finish()
    assert torch.equal(got_dense_points.view(torch.int32), want.points.view(torch.int32))
