"""The single-pass call as a streaming caller uses it (scripts/test.py:131, 203-240: one view per loop iteration; pipeline.py:
eight per launch): ONE stream operation per call -- the look-back granules carry the workspace's call epoch instead of being
zeroed, the batch's last tile writes the cursor -- and a tile geometry chosen by the size of the batch.  Everything here is
checked against the oracle or against the same cloud built another way, bit for bit."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from lab_bits import CLASSIC, FAULT, W32, W64, kw, set_on      # variant words: a product tuning + experiment switches (tests/lab_bits.py)

EPOCH_MAX = (1 << 18) - 1
T_SMALL, T_LARGE, STATIC = 1 << 18, 3 << 18, 1 << 22
# CLASSIC: the decoupled look-back of rounds 1-4 (default since round 5: the scan service -- one workgroup of the launch scans the
# tiles' counts, the tiles poll their own row)


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


def _case(seed, V, H, W, dtype=np.float32, rho=0.8):
    rng = np.random.default_rng(seed)
    depth = rng.uniform(0.5, 5.0, (V, H, W)).astype(dtype)
    depth[rng.uniform(size=depth.shape) < 0.03] = 0.0
    mask = rng.uniform(size=(V, H, W)) < rho
    normal = rng.normal(size=(V, H, W, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (V, H, W, 3), dtype=np.uint8)
    params = np.tile([0.8 * W, 0.9 * W, W / 2.0, H / 2.0], (V, 1))
    E = np.zeros((V, 3, 4))
    for v in range(V):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        E[v, :, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
        E[v, :, 3] = rng.normal(size=3)
    return depth, mask, normal, rgb, params, E


def _oracle(orc, depth, mask, normal, rgb, params, E):
    return orc.fuse_views([orc.densify_view_script(depth[v], params[v], E[v], mask=mask[v], normal=normal[v], rgb=rgb[v])
                           for v in range(depth.shape[0])])


def _header(builder):
    h = builder._ws_cache[:64].view(__import__("torch").int32).cpu().numpy()
    return {"error": int(h[1]), "ticket": int(h[4]), "done": int(h[5]), "epoch": int(h[6])}


def _equal(a, b):
    import torch
    assert len(a) == len(b) and torch.equal(a.view_offsets, b.view_offsets)
    for f in ("points", "normals", "colors", "pixel_index"):
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None and y is None) or torch.equal(x, y), f


@pytest.mark.parametrize("dtype", (np.float32, np.float16))
@pytest.mark.parametrize("shape", [(3, 67, 129), (2, 128, 256), (1, 255, 257), (2, 200, 331), (9, 96, 172), (70, 216, 384)])
def test_every_single_pass_geometry_writes_the_oracles_cloud(dd, orc, shape, dtype):
    """Small tile (8 pixels per lane, 16 waves: 8192) / large tile (16, 12288), tiles by ticket / by workgroup index, 16 / 32 / 64
    polling lanes: indices, colours and normals bit-exact against the oracle, and the rows of all variants identical."""
    V, H, W = shape
    depth, mask, normal, rgb, params, E = _case(11 + H, V, H, W, dtype)
    ref = _oracle(orc, depth, mask, normal, rgb, params, E)
    first = None
    for tuning in (0, T_SMALL, T_LARGE, 8 | T_LARGE, CLASSIC, CLASSIC | T_SMALL, CLASSIC | T_SMALL | STATIC | W64, CLASSIC | T_LARGE,
                   CLASSIC | T_LARGE | STATIC, CLASSIC | T_LARGE | W32, CLASSIC | 8 | T_LARGE | W64):
        cloud = dd.unproject_views(depth, params, E, mask=mask, normal=normal, rgb=rgb, capacity="max", **kw(tuning))
        c = cloud.numpy()
        assert np.array_equal(c["view_offsets"], ref.view_offsets), tuning
        assert np.array_equal(c["pixel_index"].astype(np.int64), ref.pixel_index), tuning
        assert np.array_equal(c["colors"], ref.colors) and np.array_equal(c["normals"], ref.normals), tuning
        if first is None:
            first = cloud
            err = np.abs(c["points"] - ref.points).max() / max(np.abs(ref.points).max(), 1.0)
            assert err <= 1e-4
        else:
            _equal(cloud, first)


def test_a_chain_of_calls_on_one_workspace_equals_one_batch(dd, orc):
    """60 views appended one at a time, then eight at a time, through one workspace that is never zeroed again: the cloud of
    one 60-view batch, bit for bit; every call advances the workspace's epoch by one and leaves ticket and done at zero."""
    V, H, W = 60, 72, 200
    depth, mask, normal, rgb, params, E = _case(5, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    b.append(whole)
    want = b.finish()
    ref = _oracle(orc, depth, mask, normal, rgb, params, E)
    assert np.array_equal(want.view_offsets.cpu().numpy(), ref.view_offsets)
    assert np.array_equal(want.pixel_index.cpu().numpy().astype(np.int64), ref.pixel_index)
    e0 = _header(b)["epoch"]
    calls = 0
    for k in (1, 8):
        b.reset()
        for lo in range(0, V, k):
            b.append(whole.slice(lo, min(lo + k, V)))
            calls += 1
        _equal(b.finish(), want)
    h = _header(b)
    assert h == {"error": 0, "ticket": 0, "done": 0, "epoch": e0 + calls}


@pytest.mark.parametrize("tuning", (0, T_LARGE, T_SMALL, 9, CLASSIC, CLASSIC | T_LARGE, CLASSIC | T_SMALL))
def test_the_epoch_wraps_without_a_trace(dd, tuning):
    """The call in which the 18-bit epoch wraps zeroes every record of the workspace (so a granule of 2^18 calls ago can never
    read as this call's): forced here by setting the epoch by hand, with stale granules planted that carry the tags of the
    calls to come."""
    import torch
    V, H, W = 8, 64, 200
    depth, mask, normal, rgb, params, E = _case(9, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb, **kw(tuning))
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    b.append(whole)
    want = b.finish()
    ws = b._ws_cache
    # a workspace far larger than these batches need, as after a large batch: records behind the ones in use hold old granules
    big = torch.zeros(ws.numel() + (1 << 16), dtype=torch.uint8, device=ws.device)
    b._ws_cache = big
    words = big[64:64 + ((big.numel() - 64) // 8) * 8].view(torch.int64)
    b.reset()
    big[:64].view(torch.int32)[6] = EPOCH_MAX - 1
    # "inclusive, row 12345" with the tags of the two calls after the wrap (epochs 0 and 1), in every word behind the header
    for epoch in (0, 1):
        words[epoch::2] = ((2 << 62) | (epoch << 44) | 12345) - (1 << 64)
    for lo in range(0, V, 2):              # epochs MAX-1, MAX (wraps: every record zeroed), 0, 1
        b.append(whole.slice(lo, lo + 2))
    got = b.finish()
    _equal(got, want)
    assert _header(b) == {"error": 0, "ticket": 0, "done": 0, "epoch": 2}
    assert int((words[words.shape[0] // 2:] != 0).sum()) == 0          # the planted granules behind the part in use are gone


def test_two_pass_calls_between_single_pass_calls_share_the_workspace(dd):
    """plan + scatter (tuning 4) and the count-free plan write their counts and rows beside the granules, never over them."""
    V, H, W = 8, 80, 160
    depth, mask, normal, rgb, params, E = _case(21, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    b.append(whole)
    want = b.finish()
    for order in ((0, 4, 0, 4), (4, 0, T_LARGE, 4), (T_LARGE, 4, 4, 0), (CLASSIC, 4, 0, CLASSIC | T_LARGE)):
        b.reset()
        for i, tuning in enumerate(order):
            sub = whole.slice(2 * i, 2 * i + 2)
            b.append(set_on(sub, tuning))
        _equal(b.finish(), want)


@pytest.mark.parametrize("how", (0, CLASSIC))
def test_appends_after_a_healed_give_up_start_from_a_clean_workspace(dd, how):
    """A look-back that gives up (fault injection, tuning 64) writes nothing it does not know the place of -- the rows of the
    batches before it stay intact -- and the redo zeroes the whole workspace, so the single-pass calls after it run as on a new one."""
    V, H, W = 8, 96, 160
    depth, mask, normal, rgb, params, E = _case(33, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    b.append(whole)
    want = b.finish()
    b.reset()
    b.append(whole.slice(0, 2))
    n2 = b.check()
    bad = whole.slice(2, 4)
    set_on(bad, FAULT | how)
    b.append(bad)
    # the sabotaged call has not touched a row of the first batch
    import torch
    torch.cuda.synchronize()
    assert torch.equal(b.xyz[:n2], want.points[:n2]) and torch.equal(b.pix[:n2], want.pixel_index[:n2])
    assert b.check() == int(want.view_offsets[4]) and b.healed == 1
    assert _header(b) == {"error": 0, "ticket": 0, "done": 0, "epoch": 0}
    b.append(whole.slice(4, 8))
    _equal(b.finish(), want)


def test_a_check_covers_what_was_appended_when_it_was_asked_for(dd):
    """append(A); p = check_async(); append(B); p.result(): A is verified and released, B is neither -- it stays held, and a
    give-up inside B that a LATER check sees is redone from the row behind A (result() used to treat B as verified,
    dropped it, and a later redo wrote the batch after it over B's rows)."""
    import torch
    V, H, W = 9, 96, 160
    depth, mask, normal, rgb, params, E = _case(41, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    b.append(whole)
    want = b.finish()
    b.reset()
    A, B, Cc = whole.slice(0, 3), whole.slice(3, 6), whole.slice(6, 9)
    set_on(B, FAULT)                                 # every look-back of B that has to wait gives up
    b.append(A)
    p = b.check_async()
    b.append(B)
    assert p.result() == int(want.view_offsets[3])   # the count behind A; B's error is not this check's business ...
    assert len(b._retained) == 1 and b._retained[0][0] is B and b._retain_base == int(want.view_offsets[3]) and b.healed == 0
    b.append(Cc)
    got = b.finish()                                 # ... it is this one's: B and C are redone from the row behind A
    assert b.healed == 1
    _equal(got, want)
    # the same with the fault in A: the redo covers everything held, the answer is the count behind all of it
    b.reset()
    set_on(A, FAULT); set_on(B, 0)
    b.append(A)
    p = b.check_async()
    b.append(B)
    assert p.result() == int(want.view_offsets[6]) and b.healed == 2 and not b._retained
    b.append(Cc)
    _equal(b.finish(), want)
    # a reader that comes after a reset() finds nothing to redo: an error that says so
    set_on(A, FAULT)
    b.reset(); b.append(A)
    p = b.check_async()
    b.reset(); b.append(B)
    with pytest.raises(RuntimeError, match="reset"):
        p.result()
    b.append(Cc)
    assert b.check() == int(want.view_offsets[9] - want.view_offsets[3])


def test_more_checks_in_flight_than_the_ring_had_slots(dd):
    """300 check_async() before the first result() is read (scene after scene, results at the end): every answer is its own."""
    V, H, W = 2, 40, 96
    depth, mask, normal, rgb, params, E = _case(43, V, H, W)
    one = dd.ViewBatch(depth[:1], params[:1], E[:1], mask=mask[:1])
    n1 = int((mask[0] & (depth[0] > 0)).sum())
    b = dd.CloudBuilder(400 * one.max_points, pixel_index=False)
    pending = []
    for i in range(300):
        b.append(one)
        pending.append(b.check_async())
    assert [p.result() for p in pending] == [n1 * (i + 1) for i in range(300)]


@pytest.mark.parametrize("order", ("A then B", "B then A"))
def test_two_checks_in_flight_release_only_what_each_covers(dd, orc, order):
    """check A covers 2 batches, check B covers 4, two more are appended; A and B are read (in either order); then a later
    batch faults (injection, tuning bit 64) and is healed: the redo must start at the row behind batch 4, from batches 5.. only --
    the cloud equals the one-batch cloud.  (Round 5 counted a check's batches relative to a list the other check had shifted.)"""
    import torch
    V, H, W = 9, 48, 112
    depth, mask, normal, rgb, params, E = _case(47, V, H, W)
    want = dd.unproject_views(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    one = lambda v, tuning=0, n=1: dd.ViewBatch(depth[v:v + n], params[v:v + n], E[v:v + n], mask=mask[v:v + n], normal=normal[v:v + n],
                                                rgb=rgb[v:v + n], view_index_base=v, **kw(tuning))
    b = dd.CloudBuilder(V * H * W, normals=True, colors=True, placement="first")
    offs = want.view_offsets.tolist()
    for v in (0, 1):
        b.append(one(v))
    A = b.check_async()
    for v in (2, 3):
        b.append(one(v))
    B = b.check_async()
    for v in (4, 5):
        b.append(one(v))
    if order == "A then B":
        assert A.result() == offs[2] and len(b._retained) == 4 and b._retain_base == offs[2]
        assert B.result() == offs[4]
    else:
        assert B.result() == offs[4] and len(b._retained) == 2 and b._retain_base == offs[4]
        assert A.result() == offs[2]                     # the older answer, read late: releases nothing more, moves nothing back
    assert len(b._retained) == 2 and b._retain_base == offs[4] and b._released == 4
    b.append(one(6, tuning=FAULT, n=2))                     # an in-kernel scan gives up here (two tiles: the second one's row is "unknown") ...
    b.append(one(8))
    cloud = b.finish()                                    # ... and the redo replays the batches of views 4..8 from the row behind view 3
    assert b.healed == 1
    _equal(cloud, want)


def test_no_guess_without_a_way_back_and_policies_are_the_callers(dd):
    """A batch is run count-free ("no holes") only if the builder will hold it for the redo a miss needs; the score of the guesses
    lives in an object the caller owns -- two builders in two threads, each with its own, do not see each other's misses."""
    import threading
    import torch
    V, H, W = 3, 1080, 1920                          # 6.2 M pixels: above GUESS_MIN_PIXELS
    g = torch.Generator(device="cuda").manual_seed(3)
    depth = torch.empty((V, H, W), device="cuda", dtype=torch.float16).uniform_(0.5, 8.0, generator=g)
    holes = depth.clone()
    holes[:, 500:520, 100:900] = 0
    params = np.tile([0.8 * W, 0.8 * W, W / 2, H / 2], (V, 1))
    E = np.tile(np.eye(4)[:3], (V, 1, 1))
    dense, holed = dd.ViewBatch(depth, params, E), dd.ViewBatch(holes, params, E)
    b = dd.CloudBuilder(dense.max_points, pixel_index=False, placement="first")
    assert b.fuse_tuning(dense) & (1 << 17)
    b._retain_limit = 1 << 20                        # (a device with little memory left: the maps of a batch will not be held)
    assert not b.fuse_tuning(dense) & (1 << 17)
    b.append(holed)                                  # ... so the batch with holes runs counted, and finishes without a redo
    assert b.check() == int((holes > 0).sum()) and b.healed == 0
    ref = dd.CloudBuilder(dense.max_points, pixel_index=False, placement="first")
    ref.append(dd.ViewBatch(holes, params, E, tuning=8))
    want = ref.finish()

    results = {}

    def worker(name, batch, rounds):
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            mine = dd.CloudBuilder(batch.max_points, pixel_index=False, placement="first")
            for _ in range(rounds):
                mine.reset()
                mine.append(batch)
                n = mine.check()
            results[name] = (n, mine.guess_policy.hits, mine.guess_policy.misses, mine.healed, mine.xyz[:n].clone())

    t1 = threading.Thread(target=worker, args=("dense", dd.ViewBatch(depth, params, E), 4))
    t2 = threading.Thread(target=worker, args=("holed", dd.ViewBatch(holes, params, E), 4))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert results["dense"][:4] == (V * H * W, 4, 0, 0)                     # four guesses, all held
    assert results["holed"][:4] == (len(want), 0, 1, 1)                     # one miss, one redo, then counted
    assert torch.equal(results["holed"][4], want.points)
    assert not hasattr(dd.CloudBuilder, "guess_hits") and not hasattr(dd.CloudBuilder, "guess_misses")


@pytest.mark.parametrize("shape", [(96, 160), (1080, 1920)])
def test_small_appends_chained_across_two_streams_write_the_same_cloud(dd, shape):
    """``CloudBuilder(exclusive_gpu=True)``: consecutive one-view appends run on two side streams, each call's scan taking its first
    row from the chain word the previous call leaves (``DDViewBatch.chain``): the cloud of one batch, bit for bit -- also when a
    large append, a check and a reset come in between, and when the chain is replayed from a captured graph."""
    import torch
    H, W = shape
    V = 21 if H > 500 else 40
    depth, mask, normal, rgb, params, E = _case(51, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    ref = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    ref.append(whole)
    want = ref.finish()
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True, exclusive_gpu=True)
    ones = [whole.slice(v, v + 1) for v in range(V)]
    for rep in range(3):
        b.reset()
        for i, s in enumerate(ones):
            b.append(s)
            if rep == 1 and i == V // 3:
                assert b.check() == int(want.view_offsets[i + 1])          # a check in the middle joins the streams
        assert (b._chain_seq >= V // 2 and len(b._side) == 2) or not b.overlap_small      # (the calls did go through the side streams -- unless
                                                                                           #  the probe found no two streams that run side by side)
        _equal(b.finish(), want)
    # a large batch between small ones: joins, runs on the caller's stream, and the chain starts again behind it
    b.reset()
    b.append(ones[0]); b.append(ones[1])
    b.append(whole.slice(2, V - 2))
    b.append(ones[V - 2]); b.append(ones[V - 1])
    _equal(b.finish(), want)
    # the same chain from a captured graph, twice
    g = dd.capture_chain(b, ones)
    for _ in range(2):
        b.xyz.zero_()
        g.replay()
        _equal(b.finish(), want)
    # a give-up in the middle of a chain: the calls behind it learn from the chain word that their rows are unknown and write nothing
    b.reset()
    for i, s in enumerate(ones[:6]):
        set_on(s, FAULT if i == 2 else 0)
        b.append(s)
    s_bad = ones[2]
    got = b.finish()
    set_on(s_bad, 0)
    assert b.healed == 1 and not b.exclusive_gpu                           # healed two-pass; tickets from here on
    assert torch.equal(got.points, want.points[:len(got)]) and len(got) == int(want.view_offsets[6])


def test_gated_chained_appends_write_the_same_cloud(dd):
    """Appends of two to four 1080p views are chained across the two side streams behind a GATE (they are too large to wait inside their
    own workgroups: dd_chain_workgroup_limit) that opens when the previous call's scan is over: the cloud of one batch, bit for bit, over
    and over."""
    import ctypes as C
    from depthdensifier_amd import _lib
    H, W, V = 1080, 1920, 14
    depth, mask, normal, rgb, params, E = _case(77, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    ref = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    ref.append(whole)
    want = ref.finish()
    cuts = [0, 2, 4, 7, 11, 14]                                              # calls of 2, 2, 3, 4 and 3 views
    subs = [whole.slice(a, b) for a, b in zip(cuts, cuts[1:])]
    out = (C.c_int32 * 8)()
    cb = subs[0].c_struct()
    cb.chain, cb.chain_seq = 0x1000, 0                                       # (what the plan of such a call says once it is chained)
    with _lib.lab_switches(subs[0].lab):
        assert _lib.lib.dd_debug_plan(C.byref(cb), out) == 0 and out[6] == 1 # gated
    cb.chain, cb.chain_seq = None, 0
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True, exclusive_gpu=True)
    chained, inner = [], b._append_chained
    b._append_chained = lambda batch, *a, **kw: (chained.append(batch.num_views), inner(batch, *a, **kw))[1]
    for rep in range(6):
        b.reset()
        for s in subs:
            b.append(s)
        _equal(b.finish(), want)
    assert (chained == [2, 2, 3, 4, 3] * 6 and len(b._side) == 2) or not b.overlap_small      # (every one of them went through the side streams)
    assert b.healed == 0


def test_side_streams_are_probed_before_calls_are_chained_across_them(dd, monkeypatch):
    """ABI 13: two HIP streams may share a hardware queue and then run strictly in order (``dd_streams_overlap``): a stream beside
    itself never overlaps; the pair a builder settles on does; and a builder that finds no pair does not chain -- same cloud."""
    import ctypes as C
    import torch
    from depthdensifier_amd import _lib, densify
    lib = _lib.lib
    scratch = torch.zeros(2, dtype=torch.int32, device="cuda")
    seen = C.c_int32(7)
    s = torch.cuda.Stream()
    assert lib.dd_streams_overlap(s.cuda_stream, s.cuda_stream, scratch.data_ptr(), C.byref(seen)) == 0 and seen.value == 0
    assert lib.dd_streams_overlap(s.cuda_stream, s.cuda_stream, None, C.byref(seen)) != 0
    V, H, W = 6, 120, 200
    depth, mask, normal, rgb, params, E = _case(77, V, H, W)
    whole = dd.ViewBatch(depth, params, E, mask=mask, normal=normal, rgb=rgb)
    ref = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True)
    ref.append(whole)
    want = ref.finish()
    ones = [whole.slice(v, v + 1) for v in range(V)]
    b = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True, exclusive_gpu=True)
    for o in ones:
        b.append(o)
    _equal(b.finish(), want)
    if not b._side:
        assert b.side_stream_probes == 8 and b.overlap_small is False
        pytest.skip("no two streams of this process run side by side: nothing was chained")
    assert len(b._side) == 2 and 1 <= b.side_stream_probes <= 8
    ok = False
    for _ in range(4):                 # (the probe gives its second kernel 1 ms to start: a host thread that lost the CPU in between says 0)
        seen.value = 0
        assert lib.dd_streams_overlap(b._side_raw[0], b._side_raw[1], scratch.data_ptr(), C.byref(seen)) == 0
        ok = ok or seen.value == 1
    assert ok
    # a process in which no second stream runs beside the first: every probe says no -> the appends stay on the caller's stream

    class NoOverlap:
        def __getattr__(self, name):
            return getattr(lib, name)

        def dd_streams_overlap(self, a, b_, w, out):
            out._obj.value = 0
            return 0
    monkeypatch.setattr(densify, "lib", NoOverlap())
    c = dd.CloudBuilder(whole.max_points, normals=True, colors=True, pixel_index=True, exclusive_gpu=True)
    for o in ones:
        c.append(o)
    _equal(c.finish(), want)
    assert c._side == [] and c.side_stream_probes == 8 and c.overlap_small is False and c._chain_seq == 0


N_STREAM_SEEDS = int(__import__("os").environ.get("DD_STREAM_SEEDS", "16"))          # soak: DD_STREAM_SEEDS=400


@pytest.mark.parametrize("seed", range(N_STREAM_SEEDS))
def test_random_chains_of_appends_write_the_one_batch_cloud(dd, seed):
    """Random view sizes / dtypes / fields, the views cut at random into calls of 1-5 views, appended with checks, joins and resets
    thrown in, on a shared GPU (tickets, one stream) or an exclusive one (by index; small calls chained across two streams, waiting
    inside their scan or behind a gate): always the cloud of ONE batch, bit for bit."""
    import torch
    rng = np.random.default_rng(70_000 + seed)
    if rng.uniform() < 0.12:
        V, H, W = int(rng.integers(2, 6)), 1080, 1920                        # one call of 1-5 such views: with and without a gate
    elif rng.uniform() < 0.5:
        V, H, W = int(rng.integers(3, 25)), int(rng.integers(1, 60)) * 4, int(rng.integers(1, 60)) * 8
    else:
        V, H, W = int(rng.integers(3, 25)), int(rng.integers(3, 300)), int(rng.integers(3, 400))
    dtype = np.float16 if rng.uniform() < 0.4 else np.float32
    depth, mask, normal, rgb, params, E = _case(int(rng.integers(1 << 30)), V, H, W, dtype=dtype, rho=float(rng.uniform(0.1, 1.0)))
    fields = dict(normals=bool(rng.uniform() < 0.6), colors=bool(rng.uniform() < 0.6))
    whole = dd.ViewBatch(depth, params, E, mask=mask if rng.uniform() < 0.7 else None, normal=normal if fields["normals"] else None,
                         rgb=rgb if fields["colors"] else None)
    ref = dd.CloudBuilder(whole.max_points, pixel_index=True, **fields)
    ref.speculate_dense = False
    ref.append(whole)
    want = ref.finish()
    b = dd.CloudBuilder(whole.max_points, pixel_index=True, exclusive_gpu=bool(rng.uniform() < 0.7), **fields)
    b.speculate_dense = False
    for _ in range(2):
        b.reset()
        lo = 0
        while lo < V:
            hi = min(V, lo + int(rng.integers(1, 6)))
            b.append(whole.slice(lo, hi))
            lo = hi
            u = rng.uniform()
            if u < 0.12:
                assert b.check() == int(want.view_offsets[lo])
            elif u < 0.2:
                b.join()
        _equal(b.finish(), want)
    assert b.healed == 0 and b.dense_misses == 0


def test_the_ungated_chain_limit_follows_the_device(dd):
    """how many workgroups a chained call may have before a gate kernel precedes it is three quarters of the slots the
    device offers that kernel (occupancy x compute units, from the runtime) -- not a constant: 384 on a whole MI355X, and a call one
    tile larger is gated (dd_debug_plan), one at the limit is not."""
    import ctypes as C
    import torch
    from depthdensifier_amd import _lib
    limit = _lib.lib.dd_chain_workgroup_limit()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert limit > 0 and limit % 3 == 0 and limit * 4 // 3 % cus == 0, (limit, cus)          # 3/4 of (a whole number of workgroups per CU) x CUs
    if cus == 256:
        assert limit == 384
    out = (C.c_int32 * 8)()
    word = torch.zeros(1, dtype=torch.int64, device="cuda")

    def gate(tiles):                                  # a one-view batch of `tiles` 6144-pixel tiles, chained
        px = tiles * 6144
        b = _lib.DDViewBatch(num_views=1, height=1, width=px, stride=1, depth=0x1000, params=0x2000, depth_dtype=_lib.DD_F32,
                             flags=_lib.DD_VALID_DEPTH_POSITIVE, chain=word.data_ptr(), chain_seq=0)
        assert _lib.lib.dd_debug_plan(C.byref(b), out) == 0
        return out[6]
    assert gate(limit - 1) == 0 and gate(limit) == 1           # (+ 1 workgroup: the scan)
