"""The round-6 entry points of the loop around the kernels, called directly through the C ABI (include/ddcore.h): dd_upload_async,
dd_stream_wait, dd_refine_fit_async -- the pipeline tests cover them only as parts of a whole scan."""

import ctypes as C

import numpy as np
import pytest
import torch


def test_upload_async_refuses_bad_arguments_before_touching_the_device():
    from depthdensifier_amd import _lib
    L = _lib.lib
    one = (C.c_void_p * 1)(0x1000)
    assert L.dd_upload_async(-1, None, None, None, None, None) == -1
    assert L.dd_upload_async(1, None, one, (C.c_int64 * 1)(8), None, None) == -1 and b"NULL array" in L.dd_ingest_last_error()
    assert L.dd_upload_async(1, (C.c_void_p * 1)(None), one, (C.c_int64 * 1)(8), None, None) == -1
    assert L.dd_upload_async(1, one, one, (C.c_int64 * 1)(-8), None, None) == -1 and b"negative size" in L.dd_ingest_last_error()
    assert L.dd_upload_async(0, None, None, None, None, None) == 0                  # nothing to copy, no event: nothing to do


@pytest.mark.gpu
def test_upload_async_and_stream_wait_order_two_streams():
    """n host -> device copies and the event behind them in one call on a copy stream; a second stream made to wait for the event sees
    every byte (the pipeline's copy stream / compute stream hand-over)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd import _lib
    L = _lib.lib
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    sizes = [1 << 22, 0, 12345, 1 << 24]
    host = [torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8)).pin_memory() if n else torch.empty(0, dtype=torch.uint8) for n in sizes]
    dst = [torch.zeros(max(n, 1), dtype=torch.uint8, device=dev) for n in sizes]
    copy, compute = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev = torch.cuda.Event()
    ev.record(copy)                                        # (creates the underlying event)
    torch.cuda.synchronize()
    for rep in range(3):
        for d in dst:
            d.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(copy):
            torch.cuda._sleep(20_000_000)                  # the copies are NOT done when the other stream is told to wait
        n = len(sizes)
        rc = L.dd_upload_async(n, (C.c_void_p * n)(*[h.data_ptr() if h.numel() else None for h in host]), (C.c_void_p * n)(*[d.data_ptr() for d in dst]),
                               (C.c_int64 * n)(*sizes), ev.cuda_event, copy.cuda_stream)
        assert rc == 0, L.dd_ingest_last_error()
        assert L.dd_stream_wait(compute.cuda_stream, ev.cuda_event) == 0
        with torch.cuda.stream(compute):
            sums = [d[:max(n_, 1)].to(torch.int64).sum() for d, n_ in zip(dst, sizes)]
        compute.synchronize()
        for s, h, n_ in zip(sums, host, sizes):
            assert int(s) == (int(h.to(torch.int64).sum()) if n_ else 0)
    copy.synchronize()
    assert L.dd_stream_wait(compute.cuda_stream, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n_points, with_mask, half_depth", [(0, True, False), (7, True, False), (3000, True, False), (3000, False, False), (1500, True, True)])
def test_the_one_call_fit_equals_the_stepwise_fit(n_points, with_mask, half_depth):
    """dd_refine_fit_async (sparse points up from page-locked memory, fit, masked-pixel count, result words down, event: ONE call) against
    dd_refine_fit on device-resident points with the count taken by a tensor reduction: the same correspondences, counters, scale and count,
    bit for bit -- with no sparse point at all, a handful, thousands; with a mask and with ``depth > 0`` in its place; float16 maps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd.depth_refiner import DepthRefiner
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(100 + n_points)
    H, W = 120, 168
    depth = rng.uniform(0.5, 6.0, (H, W)).astype(np.float32)
    depth[rng.uniform(size=(H, W)) < 0.1] = 0.0
    mask = rng.uniform(size=(H, W)) < 0.7
    K = np.array([[140.0, 0, W / 2], [0, 140.0, H / 2], [0, 0, 1]])
    E = np.eye(4)[:3]
    uv = np.stack([rng.uniform(-20, W + 20, n_points), rng.uniform(-20, H + 20, n_points)], 1)
    z = rng.uniform(0.5, 8.0, n_points)
    pts = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0] * z, (uv[:, 1] - K[1, 2]) / K[1, 1] * z, z], 1).astype(np.float32)
    r = DepthRefiner(adaptive_correspondences=False, use_fp16=False)
    d = torch.from_numpy(depth).to(dev)
    if half_depth:
        d = d.half()
    m = torch.from_numpy(mask).to(dev) if with_mask else None
    one = r._fit_finish(r._fit_launch(d, pts, E, K, m))                                       # numpy points: the one-call path
    two = r._fit_finish(r._fit_launch(d, torch.from_numpy(pts).to(dev), E, K, m))             # device points: fit, then the count by a reduction
    assert one[2:] == two[2:], (one[2:], two[2:])
    assert torch.equal(one[0], two[0]) and torch.equal(one[1], two[1])
    want = int(mask.sum()) if with_mask else int((d > 0).sum())
    assert one[7] == want
    if n_points == 0:
        assert one[2:6] == (0, 0, 0, 0)
