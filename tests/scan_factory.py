"""Synthetic COLMAP scan folders for the pipeline tests: cameras on an arc above a ground plane,
true depth by ray/plane intersection, monocular depth = distorted true depth, sparse points on the
plane, cached <stem>.npz maps, PNG images.  Written with the package's own COLMAP writer."""

from pathlib import Path

import numpy as np

from depthdensifier_amd.colmap_io import Camera, Image, Reconstruction


def rot_to_qvec(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = (R[2, 1] - R[1, 2]) / (4 * w); y = (R[0, 2] - R[2, 0]) / (4 * w); z = (R[1, 0] - R[0, 1]) / (4 * w)
    return np.array([w, x, y, z])


def make_scan(root: Path, name: str, V=5, H=96, W=128, seed=0, floaters=0.02, second_size=None, image_scale=1, second_tail=0,
              image_ext=".png", photo_like=False):
    """``second_size=(H2, W2)``: odd-numbered views use a second camera of that size (mixed resolutions in one scan);
    with ``second_tail=k`` only the LAST k views use it (a camera no early view touches).  ``image_scale=f``: the image
    files and the sparse model's cameras / 2-D observations are f times larger than the cached maps, i.e. a scan meant
    to be run with ``pipeline_downsample_factor=f`` (``scripts/test.py:145-152, 172-173``).  ``image_ext=".jpg"`` with
    ``photo_like=True`` writes smooth pictures as JPEG files (what real scans hold; lossy, so for timing only)."""
    from PIL import Image as PILImage
    rng = np.random.default_rng(seed)
    scan = Path(root) / name
    (scan / "images").mkdir(parents=True)
    (scan / "sparse" / "0").mkdir(parents=True)
    cache = scan / "moge_cache"
    cache.mkdir()
    rec = Reconstruction()
    sizes = [(H, W)] + ([tuple(second_size)] if second_size else [])
    for k, (h, w) in enumerate(sizes):
        f = image_scale
        rec.cameras[k + 1] = Camera(k + 1, 1, w * f, h * f, np.array([0.9 * w * f, 0.9 * w * f, w * f / 2.0, h * f / 2.0]))
    ids, xyz, rgbs, next_id = [], [], [], 1
    truth = []
    for v in range(V):
        cam_id = (2 if v >= V - second_tail else 1) if (second_tail and len(sizes) > 1) else 1 + (v % len(sizes))
        H, W = sizes[cam_id - 1]
        fx = fy = 0.9 * W
        cx, cy = W / 2.0, H / 2.0
        a = -0.5 + v * (1.0 / max(V - 1, 1))
        c = np.array([3.0 * np.sin(a), -2.0, -3.0 * np.cos(a)])          # camera centre above the plane y = 0
        zax = -c / np.linalg.norm(c)                                      # looks at the origin
        xax = np.cross([0, 1.0, 0], zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        R = np.stack([xax, yax, zax]); t = -R @ c
        us, vs = np.meshgrid(np.arange(W), np.arange(H))
        rays_cam = np.stack([(us - cx) / fx, (vs - cy) / fy, np.ones_like(us, float)], -1)
        rays_w = rays_cam @ R                                             # R^T r
        tt = -c[1] / rays_w[..., 1]                                       # plane y = 0
        depth_true = np.where((tt > 0) & np.isfinite(tt), tt, 0.0)
        mask = (depth_true > 0) & (depth_true < 12) & (rng.uniform(size=(H, W)) < 0.97)
        mono = (0.5 * np.maximum(depth_true, 1e-3) ** 1.1).astype(np.float32)
        fl = np.zeros((H, W), bool)                                       # floaters: 5x5 blocks far too close
        for _ in range(int(floaters * H * W / 25)):                       # (blocks survive the refiner's 3x3 median)
            y0, x0 = rng.integers(10, H - 15), rng.integers(10, W - 15)
            fl[y0:y0 + 5, x0:x0 + 5] = True
        mono_f = np.where(fl, mono * 0.3, mono).astype(np.float32)
        normal = np.tile(np.array([0.0, -1.0, 0.0]) @ R.T, (H, W, 1)).astype(np.float32)   # plane normal in the camera frame
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        stem = f"img_{v:03d}"
        if photo_like:               # low-frequency content: a photograph's share of zero coefficients, not noise
            small = rng.integers(0, 256, (H // 16 + 1, W // 16 + 1, 3), dtype=np.uint8)
            img = np.asarray(PILImage.fromarray(small).resize((W, H), PILImage.Resampling.BICUBIC))
        if image_scale > 1:          # a smooth full-resolution picture (so that LANCZOS down-sampling is not pure noise)
            big = np.asarray(PILImage.fromarray(img).resize((W * image_scale, H * image_scale), PILImage.Resampling.BICUBIC))
            PILImage.fromarray(big).save(scan / "images" / f"{stem}{image_ext}")
        else:
            PILImage.fromarray(img).save(scan / "images" / f"{stem}{image_ext}", **({"quality": 92} if image_ext != ".png" else {}))
        np.savez(cache / f"{stem}.npz", depth=mono_f, mask=mask, normal=normal)
        # sparse observations on the plane
        n_obs = 300
        pu = rng.uniform(12, W - 13, n_obs); pv = rng.uniform(12, H - 13, n_obs)
        d = depth_true[pv.astype(int), pu.astype(int)]
        ok = (d > 0) & (d < 12)
        pu, pv, d = pu[ok], pv[ok], d[ok]
        cam = np.stack([(np.floor(pu) - cx) / fx * d, (np.floor(pv) - cy) / fy * d, d], -1)
        world = (cam - t) @ R
        pid = np.arange(next_id, next_id + len(world)); next_id += len(world)
        ids.append(pid); xyz.append(world); rgbs.append(np.full((len(world), 3), 200, np.uint8))
        xys = np.stack([np.floor(pu), np.floor(pv)], -1) * image_scale
        rec.images[v + 1] = Image(v + 1, rot_to_qvec(R), t, cam_id, f"{stem}{image_ext}", xys, pid.astype(np.int64))
        truth.append(dict(R=R, t=t, depth_true=depth_true, mono=mono_f, mask=mask, normal=normal, rgb=img))
    rec.point_ids = np.concatenate(ids).astype(np.uint64)
    rec.point_xyz = np.concatenate(xyz)
    rec.point_rgb = np.concatenate(rgbs)
    rec.point_error = np.zeros(len(rec.point_ids))
    rec._tracks = [np.zeros((0, 2), np.int32)] * len(rec.point_ids)
    rec.write_binary(scan / "sparse" / "0")
    return scan, cache, truth
