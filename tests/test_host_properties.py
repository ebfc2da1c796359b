"""Property tests (hypothesis) of the host-side logic: sharding, CLI, COLMAP I/O, camera blocks."""

import dataclasses
from pathlib import Path

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


@given(V=st.integers(0, 5000), R=st.integers(1, 16))
def test_shards_tile_the_views(V, R):
    from depthdensifier_amd.distributed import shard_sizes, shard_views
    spans = [shard_views(V, R, r) for r in range(R)]
    assert spans[0][0] == 0 and spans[-1][1] == V
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = shard_sizes(V, R)
    assert sum(sizes) == V and max(sizes) - min(sizes) <= 1 and all(s >= 0 for s in sizes)


@given(thr=st.integers(0, 50), density=st.integers(1, 64), fp16=st.booleans(), dt=st.floats(0.01, 2.0))
def test_cli_round_trip(thr, density, fp16, dt):
    from depthdensifier_amd.batch import BatchConfig
    from depthdensifier_amd.cli import parse
    argv = ["--root-dir", "/r", "--output-dir", "/o", "--config.filtering.vote-threshold", str(thr),
            "--config.processing.downsample-density", str(density), "--config.filtering.depth-threshold", repr(dt),
            "--config.refiner.use-fp16" if fp16 else "--config.refiner.no-use-fp16"]
    c = parse(BatchConfig, argv)
    assert (c.config.filtering.vote_threshold, c.config.processing.downsample_density) == (thr, density)
    assert c.config.refiner.use_fp16 is fp16 and c.config.filtering.depth_threshold == dt
    assert c.config.refiner.min_correspondences == 50                      # untouched defaults survive


@settings(max_examples=25, deadline=None)
@given(n_pts=st.integers(0, 40), n_new=st.integers(0, 60), seed=st.integers(0, 2 ** 16))
def test_colmap_points_round_trip(tmp_path_factory, n_pts, n_new, seed):
    from depthdensifier_amd.colmap_io import Camera, Image, Reconstruction
    rng = np.random.default_rng(seed)
    rec = Reconstruction()
    rec.cameras[3] = Camera(3, 1, 640, 480, np.array([500.0, 505.0, 320.0, 240.0]))
    ids = np.arange(10, 10 + n_pts, dtype=np.uint64)
    rec.point_ids, rec.point_xyz = ids, rng.standard_normal((n_pts, 3))
    rec.point_rgb, rec.point_error = rng.integers(0, 256, (n_pts, 3)).astype(np.uint8), rng.uniform(0, 2, n_pts)
    rec._tracks = [rng.integers(0, 9, (int(rng.integers(0, 4)), 2)).astype(np.int32) for _ in range(n_pts)]
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    obs = int(rng.integers(0, 6))
    rec.images[7] = Image(7, q, rng.standard_normal(3), 3, "a b/ü.png", rng.uniform(0, 600, (obs, 2)),
                          rng.choice(np.concatenate([ids.astype(np.int64), [-1]]), obs) if obs else np.zeros(0, np.int64))
    rec.add_points3D(rng.standard_normal((n_new, 3)), rng.integers(0, 256, (n_new, 3)))
    d = tmp_path_factory.mktemp("m")
    rec.write_binary(d)
    back = Reconstruction(d)
    assert back.num_points3D() == n_pts + n_new
    assert np.array_equal(back.point_ids, rec.point_ids) and np.array_equal(back.point_xyz, rec.point_xyz)
    assert np.array_equal(back.point_rgb, rec.point_rgb) and np.array_equal(back.point_error, rec.point_error)
    assert all(np.array_equal(a, b) for a, b in zip(back._tracks[:n_pts], rec._tracks))
    assert all(len(t) == 0 for t in back._tracks[n_pts:])
    im = back.images[7]
    assert im.name == "a b/ü.png" and np.array_equal(im.qvec, q) and np.array_equal(im.point3D_ids, rec.images[7].point3D_ids)


@settings(max_examples=30, deadline=None)
@given(seed=st.integers(0, 2 ** 16), fx=st.floats(50, 5000), skew=st.floats(-2, 2))
def test_camera_blocks_are_the_fused_reference_map(seed, fx, skew):
    """d * (R^T K^-1 [u,v,1]) - R^T t  ==  inv([R|t]) applied to d * K^-1 [u,v,1]   (visualizer.py:320-334)."""
    import depthdensifier_amd as dd
    from synth import random_pose
    rng = np.random.default_rng(seed)
    E = random_pose(rng)
    K = np.array([[fx, skew, 300.0], [0, fx * 1.01, 200.0], [0, 0, 1.0]])
    b = dd.camera_blocks(K, E[None]).astype(np.float64)[0]
    u, v, d = rng.uniform(0, 600), rng.uniform(0, 400), rng.uniform(0.1, 20)
    got = d * (b[:9].reshape(3, 3) @ [u, v, 1.0]) + b[9:12]
    cam = d * (np.linalg.inv(K) @ [u, v, 1.0])
    want = (np.linalg.inv(np.vstack([E, [0, 0, 0, 1]])) @ [*cam, 1.0])[:3]
    assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
