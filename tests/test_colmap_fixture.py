"""A COLMAP binary model written HERE, byte by byte with ``struct``, from the published format description -- not by
``colmap_io``'s own writer -- and read by ``Reconstruction`` (``scripts/test.py:111``).

Layout (little endian, COLMAP ``src/colmap/scene/reconstruction_io.cc`` as documented in the COLMAP manual,
"Output format / binary"):

    cameras.bin   uint64 num_cameras; per camera: int32 camera_id, int32 model_id, uint64 width, uint64 height,
                  float64 params[num_params(model)]
    images.bin    uint64 num_reg_images; per image: int32 image_id, float64 qvec[4] (w x y z), float64 tvec[3],
                  int32 camera_id, char name[] + NUL, uint64 num_points2D, per point2D: float64 x, float64 y,
                  uint64 point3D_id (kInvalidPoint3DId = 2^64 - 1 when the feature has no 3-D point)
    points3D.bin  uint64 num_points3D; per point: uint64 point3D_id, float64 xyz[3], uint8 rgb[3], float64 error,
                  uint64 track_length, per track element: int32 image_id, int32 point2D_idx

The model has a SIMPLE_RADIAL and a PINHOLE camera, images with observations that do and do not have 3-D points,
non-contiguous ids, non-empty tracks of different lengths.  Also checked: ``Camera.rescale`` against COLMAP's rule, the
bytes our writer produces for the same model (must be identical to the hand-packed ones) and the bulk ``add_points3D``.
"""

import struct

import numpy as np
import pytest

INVALID = 2 ** 64 - 1

CAMERAS = [  # id, model id, width, height, params
    (1, 2, 4946, 3286, [4627.3, 2473.0, 1643.0, 0.0172]),          # SIMPLE_RADIAL: f, cx, cy, k
    (7, 1, 1237, 822, [1159.5, 1164.7, 618.5, 411.0]),             # PINHOLE: fx, fy, cx, cy
]
IMAGES = [  # id, qvec (w x y z), tvec, camera id, name, [(x, y, point3D id)]
    (3, [0.8775825618903728, 0.0, 0.479425538604203, 0.0], [0.5, -0.25, 2.0], 7, "_DSC8679.JPG",
     [(10.5, 20.25, 11), (600.0, 400.5, INVALID), (1200.75, 800.0, 42), (3.0, 4.0, 11)]),
    (12, [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 1, "sub dir/frame 0012.png", [(100.0, 200.0, INVALID)]),
    (5, [0.7071067811865476, 0.7071067811865476, 0.0, 0.0], [-1.0, 2.0, 3.5], 7, "a.png", []),
]
POINTS = [  # id, xyz, rgb, error, track [(image id, point2D idx)]
    (11, [1.5, -2.25, 7.125], [255, 0, 17], 0.73, [(3, 0), (3, 3)]),
    (42, [-0.5, 0.0, 3.0], [1, 2, 3], 1.25, [(3, 2)]),
    (1000000007, [9.0, 8.0, 7.0], [200, 100, 50], -1.0, []),
]


def pack_model(folder):
    b = struct.pack("<Q", len(CAMERAS))
    for cid, model, w, h, params in CAMERAS:
        b += struct.pack("<iiQQ", cid, model, w, h) + struct.pack(f"<{len(params)}d", *params)
    (folder / "cameras.bin").write_bytes(b)
    b = struct.pack("<Q", len(IMAGES))
    for iid, q, t, cid, name, obs in IMAGES:
        b += struct.pack("<i4d3di", iid, *q, *t, cid) + name.encode() + b"\0" + struct.pack("<Q", len(obs))
        for x, y, pid in obs:
            b += struct.pack("<ddQ", x, y, pid)
    (folder / "images.bin").write_bytes(b)
    b = struct.pack("<Q", len(POINTS))
    for pid, xyz, rgb, err, track in POINTS:
        b += struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, err, len(track))
        for iid, idx in track:
            b += struct.pack("<ii", iid, idx)
    (folder / "points3D.bin").write_bytes(b)


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_reads_a_hand_packed_model(tmp_path):
    from depthdensifier_amd.colmap_io import INVALID_POINT3D, Reconstruction
    pack_model(tmp_path)
    rec = Reconstruction(tmp_path)
    assert sorted(rec.cameras) == [1, 7] and rec.num_reg_images() == 3 and rec.num_points3D() == 3
    for cid, model, w, h, params in CAMERAS:
        c = rec.cameras[cid]
        assert (c.camera_id, c.model_id, c.width, c.height) == (cid, model, w, h)
        assert c.params.dtype == np.float64 and c.params.tolist() == params
    assert rec.cameras[1].model_name == "SIMPLE_RADIAL" and rec.cameras[7].model_name == "PINHOLE"
    assert rec.cameras[1].calibration_matrix().tolist() == [[4627.3, 0.0, 2473.0], [0.0, 4627.3, 1643.0], [0.0, 0.0, 1.0]]
    assert rec.cameras[7].calibration_matrix().tolist() == [[1159.5, 0.0, 618.5], [0.0, 1164.7, 411.0], [0.0, 0.0, 1.0]]
    assert rec.cameras[7].pinhole_params().tolist() == [1159.5, 1164.7, 618.5, 411.0]
    for iid, q, t, cid, name, obs in IMAGES:
        im = rec.images[iid]
        assert (im.image_id, im.camera_id, im.name, im.has_pose) == (iid, cid, name, True)
        assert im.qvec.tolist() == q and im.tvec.tolist() == t
        assert len(im.points2D) == len(obs)
        for p2, (x, y, pid) in zip(im.points2D, obs):
            assert p2.xy.tolist() == [x, y]
            assert p2.has_point3D() == (pid != INVALID)
            assert p2.point3D_id == (pid if pid != INVALID else INVALID_POINT3D)
        assert im.observed_point3D_ids().tolist() == [pid for _, _, pid in obs if pid != INVALID]       # scripts/test.py:135
        E = im.cam_from_world().matrix()                                                               # :63, :177
        assert np.allclose(E[:, :3], quat_to_R(q), atol=1e-15) and E[:, 3].tolist() == t
        assert np.allclose(im.projection_center(), -quat_to_R(q).T @ np.array(t), atol=1e-15)          # :284
        p = np.array([[0.3, -0.2, 1.0]])
        assert np.allclose(im.cam_from_world().inverse() * (im.cam_from_world() * p), p, atol=1e-14)   # :233
    for pid, xyz, rgb, err, track in POINTS:
        assert pid in rec.points3D
        pt = rec.points3D[pid]
        assert pt.xyz.tolist() == xyz and pt.color.tolist() == rgb and pt.error == err
    assert rec.xyz_of(np.array([42, 11, 11])).tolist() == [POINTS[1][1], POINTS[0][1], POINTS[0][1]]    # :139
    assert [t.tolist() for t in rec._tracks] == [[list(e) for e in tr] for *_, tr in POINTS]
    with pytest.raises(KeyError):
        rec.xyz_of(np.array([12345]))


def test_writer_reproduces_the_hand_packed_bytes_and_appends_in_bulk(tmp_path):
    from depthdensifier_amd.colmap_io import Reconstruction
    src, dst, grown = tmp_path / "src", tmp_path / "dst", tmp_path / "grown"
    src.mkdir()
    pack_model(src)
    rec = Reconstruction(src)
    rec.write_binary(dst)
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert (dst / name).read_bytes() == (src / name).read_bytes(), name
    # scripts/test.py:355-358 in bulk: consecutive new ids after the largest, empty track, error -1
    xyz = np.array([[0.1, 0.2, 0.3], [4.0, 5.0, 6.0]])
    rgb = np.array([[9, 8, 7], [255, 254, 253]], np.uint8)
    rec.add_points3D(xyz, rgb)
    rec.write_binary(grown)
    raw = (grown / "points3D.bin").read_bytes()
    head = (src / "points3D.bin").read_bytes()
    assert struct.unpack_from("<Q", raw, 0)[0] == 5 and raw[8:len(head)] == head[8:]
    tail = raw[len(head):]
    assert len(tail) == 2 * (8 + 24 + 3 + 8 + 8)
    for k in range(2):
        pid, x, y, z, r, g, b, err, tl = struct.unpack_from("<Q3d3BdQ", tail, k * 51)
        assert (pid, [x, y, z], [r, g, b], err, tl) == (1000000008 + k, xyz[k].tolist(), rgb[k].tolist(), -1.0, 0)
    again = Reconstruction(grown)
    assert again.num_points3D() == 5 and again.points3D[1000000009].color.tolist() == [255, 254, 253]


def test_rescale_follows_colmap(tmp_path):
    """``Camera::Rescale(new_width, new_height)`` (scripts/test.py:172-173): per-axis scale for two-focal models and the
    principal point; the mean of the two scales for the single focal length of SIMPLE_* / RADIAL models."""
    from depthdensifier_amd.colmap_io import Reconstruction
    pack_model(tmp_path)
    rec = Reconstruction(tmp_path)
    pin, rad = rec.cameras[7], rec.cameras[1]
    pin.rescale(new_width=618, new_height=411)
    sx, sy = 618 / 1237, 411 / 822
    assert (pin.width, pin.height) == (618, 411)
    assert np.allclose(pin.params, [1159.5 * sx, 1164.7 * sy, 618.5 * sx, 411.0 * sy], rtol=0, atol=1e-12)
    rad.rescale(new_width=2473, new_height=1643)
    sx, sy = 2473 / 4946, 1643 / 3286
    assert np.allclose(rad.params, [4627.3 * (sx + sy) / 2, 2473.0 * sx, 1643.0 * sy, 0.0172], rtol=0, atol=1e-12)
    pin.rescale(new_width=618, new_height=411)                     # same size again: identity (several images share a camera)
    assert np.allclose(pin.params[0], 1159.5 * 618 / 1237, atol=1e-12)


# ------------------------------------------------------------------ dense points streamed from the GPU

@pytest.mark.gpu
@pytest.mark.parametrize("form", ("rows", "packed"))
@pytest.mark.parametrize("n,chunk", ((0, 100), (1, 100), (255, 100), (256, 256), (1000, 257), (70_001, 4096)))
def test_streamed_dense_records_equal_the_host_writer(tmp_path, form, n, chunk):
    """``write_binary(dense=device cloud)`` (records formatted by ``dd_format_points3d`` and streamed in chunks) writes
    the bytes that ``add_points3D`` + ``write_binary`` write: ids continuing after the sparse points, float64 xyz, rgb,
    error -1, empty tracks -- ``scripts/test.py:355-358, 363``."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import depthdensifier_amd as dd
    from depthdensifier_amd.colmap_io import Reconstruction

    src = tmp_path / "src"
    src.mkdir()
    pack_model(src)
    rng = np.random.default_rng(n)
    xyz = (rng.standard_normal((n, 3)) * 10).astype(np.float32)
    rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    host = Reconstruction(src)
    host.add_points3D(xyz.astype(np.float64), rgb)
    host.write_binary(tmp_path / "host")

    pts, col = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
    offs = torch.tensor([0, n], dtype=torch.int64, device="cuda")
    if form == "rows":
        cloud = dd.FusedCloud(points=pts, colors=col, normals=None, pixel_index=None, view_index=None, view_offsets=offs)
    else:
        rec = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        rec[:, :3] = pts
        rgba = col.to(torch.int32)
        word = rgba[:, 0] | (rgba[:, 1] << 8) | (rgba[:, 2] << 16) | (0xFF << 24)
        rec.view(torch.int32)[:, 3] = word
        cloud = dd.FusedCloud.from_packed(rec, offs)
    dev = Reconstruction(src)
    where = dev.write_binary(tmp_path / "dev", dense=cloud, chunk_points=chunk)
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert (tmp_path / "dev" / name).read_bytes() == (tmp_path / "host" / name).read_bytes(), name
    back = Reconstruction(tmp_path / "dev")
    assert back.num_points3D() == host.num_points3D()
    assert where["first_dense_id"] == int(Reconstruction(src).point_ids.max()) + 1

    # the sharded form: one call lays the file out, slices are written in place in any order
    if n >= 256:
        from depthdensifier_amd.model_writer import RECORD_BYTES, write_dense_at
        lay = Reconstruction(src).write_binary(tmp_path / "shard", dense=None, dense_total=n)
        cut = n // 3

        def piece(a, b):
            return dd.FusedCloud(points=pts[a:b], colors=col[a:b], normals=None, pixel_index=None, view_index=None,
                                 view_offsets=torch.tensor([0, b - a], dtype=torch.int64, device="cuda"))
        f = tmp_path / "shard" / "points3D.bin"
        write_dense_at(f, lay["dense_offset"] + cut * RECORD_BYTES, piece(cut, n), lay["first_dense_id"] + cut, chunk_points=chunk)
        write_dense_at(f, lay["dense_offset"], piece(0, cut), lay["first_dense_id"], chunk_points=chunk)
        assert f.read_bytes() == (tmp_path / "host" / "points3D.bin").read_bytes()


@pytest.mark.gpu
def test_format_points3d_argument_errors():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from depthdensifier_amd import _lib
    lib = _lib.lib
    xyz = torch.zeros((4, 3), device="cuda"); out = torch.zeros(4 * 51 + 64, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    base = out.data_ptr() + (-out.data_ptr()) % 16
    assert lib.dd_format_points3d(xyz.data_ptr(), None, None, 4, 1, base, s) == 0
    assert lib.dd_format_points3d(xyz.data_ptr(), None, None, 0, 1, None, s) == 0            # nothing to do
    for args, word in (((None, None, None, 4, 1, base, s), "xyz"), ((xyz.data_ptr(), None, None, -1, 1, base, s), "negative"),
                       ((xyz.data_ptr(), None, None, 4, 1, base + 4, s), "aligned"), ((xyz.data_ptr(), None, None, 4, 1, None, s), "out"),
                       ((None, None, xyz.data_ptr() + 4, 4, 1, base, s), "aligned")):
        assert lib.dd_format_points3d(*args) == -1 and word in lib.dd_model_last_error().decode(), word
    torch.cuda.synchronize()
    rec = out[base - out.data_ptr(): base - out.data_ptr() + 51].cpu().numpy()
    assert struct.unpack("<Q3d3BdQ", rec.tobytes()) == (1, 0.0, 0.0, 0.0, 0, 0, 0, -1.0, 0)         # rgb NULL -> black
