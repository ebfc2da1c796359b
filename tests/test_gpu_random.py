"""Randomised parity sweep: HIP path vs oracle over random shapes, strides, dtypes, field subsets,
semantics and kernel variants (seeded, so failures reproduce)."""

import os

import numpy as np
import pytest

from test_gpu_parity import assert_cloud, scene_radius  # noqa: E402  (shared helpers)

pytestmark = pytest.mark.gpu

# soak runs: DD_RANDOM_SEEDS=1000 DD_RANDOM_SCALE=6 python -m pytest tests/test_gpu_random.py  (defaults: 60 seeds, scale 1)
N_SEEDS = int(os.environ.get("DD_RANDOM_SEEDS", "60"))
SCALE = int(os.environ.get("DD_RANDOM_SCALE", "1"))      # multiplies H and W: several look-back tiles per view


def _case(seed):
    from synth import make_views
    rng = np.random.default_rng(10_000 + seed)
    V = int(rng.integers(1, 5))
    # mix of vector-aligned shapes (H*W % 8 == 0) and ragged ones
    if rng.uniform() < 0.5:
        H, W = int(rng.integers(1, 40)) * 4, int(rng.integers(1, 40)) * 8
    else:
        H, W = int(rng.integers(1, 150)), int(rng.integers(1, 150))
    dtype = np.float16 if rng.uniform() < 0.4 else np.float32
    H, W = H * SCALE, W * SCALE
    d = make_views(seed, V, H, W, rho=float(rng.uniform(0.05, 1.0)), specials=bool(rng.uniform() < 0.7), depth_dtype=dtype)
    d["params"] = np.stack([[W * rng.uniform(0.5, 1.5), W * rng.uniform(0.5, 1.5), W / 2 + rng.uniform(-5, 5),
                             H / 2 + rng.uniform(-5, 5)] for _ in range(V)])
    opts = dict(
        stride=int(rng.choice([1, 1, 1, 2, 3, 5, 32])),
        use_mask=bool(rng.uniform() < 0.7), use_conf=bool(rng.uniform() < 0.3),
        conf_dtype=np.float16 if rng.uniform() < 0.5 else np.float32,
        use_normal=bool(rng.uniform() < 0.6), use_rgb=bool(rng.uniform() < 0.6),
        viz=bool(rng.uniform() < 0.25), tuning=int(rng.choice([0, 0, 1, 4, 5, 9])),
    )
    # options added in round 2, drawn from their own stream so that a seed keeps the configuration it always had
    rng2 = np.random.default_rng(20_000 + seed)
    opts["record"] = str(rng2.choice(["rows", "rows", "xyz_rgba", "both"]))
    opts["capacity"] = [None, "max"][int(rng2.integers(0, 2))]
    if rng2.uniform() < 0.3:
        opts["tuning"] |= 32                      # wave runs NOT aligned to 128-byte lines: same bits out
    if np.random.default_rng(30_000 + seed).uniform() < 0.4:
        opts["tuning"] |= 128                     # round 4: dense tiles take the list-free path (needs >= 12 288 dense pixels: DD_RANDOM_SCALE)
    r4 = np.random.default_rng(40_000 + seed)
    if r4.uniform() < 0.2:
        # round 4: the count-free plan (every pixel guessed valid, the scatter verifies).  The random maps have holes and masks, so the
        # guess misses nearly always: what is tested is that the miss is noticed and the batch redone into the right cloud
        opts["tuning"] |= 1 << 17
        if r4.uniform() < 0.5:
            opts["tuning"] |= int(r4.integers(1, 64)) << 8     # with the tiles of the scatter pass in an interleaved order
    r5 = np.random.default_rng(50_000 + seed)
    if r5.uniform() < 0.6 and not (opts["tuning"] & (1 | 4 | (1 << 17))):
        # round 5: the single-pass kernel's geometries -- tiles by workgroup index (bit 22), the decoupled look-back instead of the scan
        # service (bit 26), the small / the large tile forced (bits 18-19 = 1 / 3), 32 / 64 polling lanes (bits 20-21 = 2 / 3)
        opts["tuning"] |= (int(r5.integers(0, 2)) << 22) | (int(r5.integers(0, 2)) << 26) | (int(r5.choice([0, 1, 3])) << 18) | (int(r5.choice([0, 2, 3])) << 20)
    return d, opts      # (opts["tuning"] is a variant word of tests/lab_bits.py: a product tuning + experiment switches)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_configuration(seed):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    import lab_bits
    from oracle import densify_oracle as orc

    d, o = _case(seed)
    mask = d["mask"] if o["use_mask"] else None
    normal = d["normal"] if o["use_normal"] else None
    rgb = d["rgb"] if o["use_rgb"] else None
    rad = scene_radius(d["cam_from_world"], d["depth"])
    if o["viz"]:
        depth32 = d["depth"].astype(np.float32)
        K = dd.intrinsics_matrix(d["params"])
        cloud = dd.unproject_views(depth32, K, d["cam_from_world"], mask=mask, normal=normal, rgb=rgb, semantics="viz",
                                   view_index=True, **lab_bits.kw(o["tuning"]))
        with np.errstate(invalid="ignore", over="ignore"):
            ref = orc.densify_scene_viz(depth32, K, d["cam_from_world"], mask=mask, normal=normal, rgb=rgb)
        assert_cloud(cloud, ref, rad, normals="close")
        return
    conf = d["conf"].astype(o["conf_dtype"]) if o["use_conf"] else None
    thr = 0.37 if o["use_conf"] else None
    cloud = dd.unproject_views(d["depth"], d["params"], d["cam_from_world"], mask=mask, normal=normal, rgb=rgb, conf=conf,
                               conf_threshold=thr, downsample_density=o["stride"], view_index=True, **lab_bits.kw(o["tuning"]),
                               record=o["record"], capacity=o["capacity"])
    ref = orc.densify_scene_script(d["depth"], d["params"], d["cam_from_world"], mask=mask, normal=normal, rgb=rgb,
                                   stride=o["stride"], conf=conf, conf_threshold=thr)
    assert_cloud(cloud, ref, rad)
    if o["record"] != "rows":                     # the 16-byte record: x, y, z bits + r | g<<8 | b<<16 | 255<<24
        rec = cloud.packed.cpu().numpy().view(np.uint32)
        assert rec.shape == (len(ref.points), 4)
        want = np.full(len(rec), 255 << 24, dtype=np.uint32)
        if rgb is not None:
            c = ref.colors.astype(np.uint32)
            want |= c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16)
        assert np.array_equal(rec[:, 3], want)
        if o["record"] == "both":                 # rows and record written by the same lanes: identical xyz bits
            assert np.array_equal(rec[:, :3], cloud.points.cpu().numpy().view(np.uint32))
