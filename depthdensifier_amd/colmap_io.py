"""COLMAP sparse-model binary I/O and PLY export.

Stands in for what the reference gets from ``pycolmap`` (not installable here, and its per-point
Python loop ``rec.add_point3D(...)`` at ``scripts/test.py:355-358`` would take hours at full
density):

* ``Reconstruction(path)`` / ``read_binary`` / ``write_binary``  (``scripts/test.py:111, 363``;
  ``src/depthdensifier/utils.py:46-51``) for ``cameras.bin`` / ``images.bin`` / ``points3D.bin``;
* ``Camera.params / calibration_matrix() / rescale()``            (``scripts/test.py:81, 73, 173``);
* ``Image.cam_from_world() / projection_center() / has_pose / points2D`` (``:63, 130, 135, 284``);
* ``add_points3D(xyz, rgb)`` -- the BULK replacement of the ``add_point3D`` loop (``:355-358``):
  new points get consecutive ids, an empty track and error -1, and are written as one
  structured array;
* ``write_ply`` for the fused cloud.

The on-disk layout is COLMAP's public binary format (little endian):
``cameras.bin``  u64 n; per camera: i32 id, i32 model, u64 width, u64 height, f64 params[k(model)];
``images.bin``   u64 n; per image: i32 id, f64 qvec[4] (w,x,y,z), f64 tvec[3], i32 camera_id,
                 name\\0, u64 m, m x (f64 x, f64 y, i64 point3D_id);
``points3D.bin`` u64 n; per point: u64 id, f64 xyz[3], u8 rgb[3], f64 error, u64 t, t x (i32 image_id, i32 point2D_idx).
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Optional, Union

import numpy as np

# model id -> (name, number of params)
CAMERA_MODELS = {
    0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
    5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
    9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12),
}
_MODEL_IDS = {name: mid for mid, (name, _) in CAMERA_MODELS.items()}
_SINGLE_FOCAL = {0, 2, 3, 8, 9}          # models whose first parameter is the one focal length
INVALID_POINT3D = -1                     # COLMAP stores kInvalidPoint3DId (2^64-1) which reads as -1 in int64


def qvec_to_rotmat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


class Rigid3d:
    """``x -> R x + t`` (COLMAP ``Rigid3d``): the transform ``image.cam_from_world()`` returns."""

    def __init__(self, rotation: np.ndarray, translation: np.ndarray):
        self.R = np.asarray(rotation, dtype=np.float64)
        self.t = np.asarray(translation, dtype=np.float64)

    def matrix(self) -> np.ndarray:
        return np.hstack([self.R, self.t[:, None]])

    def inverse(self) -> "Rigid3d":
        return Rigid3d(self.R.T, -self.R.T @ self.t)

    def __mul__(self, points: np.ndarray) -> np.ndarray:
        p = np.asarray(points, dtype=np.float64)
        return p @ self.R.T + self.t


@dataclass
class Camera:
    camera_id: int
    model_id: int
    width: int
    height: int
    params: np.ndarray

    @property
    def model_name(self) -> str:
        return CAMERA_MODELS[self.model_id][0]

    def _fxfycxcy(self):
        p = self.params
        if self.model_id in _SINGLE_FOCAL:
            return p[0], p[0], p[1], p[2]
        return p[0], p[1], p[2], p[3]

    def calibration_matrix(self) -> np.ndarray:
        fx, fy, cx, cy = self._fxfycxcy()
        return np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])

    def pinhole_params(self) -> np.ndarray:
        """``fx, fy, cx, cy`` whatever the model (distortion ignored, as the reference does by reading
        only ``calibration_matrix()`` / the first four PINHOLE params)."""
        return np.array(self._fxfycxcy(), dtype=np.float64)

    def rescale(self, new_width: int, new_height: int) -> None:
        """COLMAP ``Camera::Rescale(new_width, new_height)``: principal point scales per axis, focal
        lengths per axis (two-focal models) or by the mean scale (single-focal models); in place."""
        sx, sy = new_width / self.width, new_height / self.height
        p = self.params
        if self.model_id in _SINGLE_FOCAL:
            p[0] *= (sx + sy) / 2.0
            p[1] *= sx
            p[2] *= sy
        else:
            p[0] *= sx
            p[1] *= sy
            p[2] *= sx
            p[3] *= sy
        self.width, self.height = int(new_width), int(new_height)


@dataclass
class Point2D:
    xy: np.ndarray
    point3D_id: int

    def has_point3D(self) -> bool:
        return self.point3D_id != INVALID_POINT3D


class _Points2DView:
    """Lazy sequence over an image's observations (arrays underneath; objects only on demand)."""

    def __init__(self, xys: np.ndarray, ids: np.ndarray):
        self.xys, self.ids = xys, ids

    def __len__(self):
        return len(self.ids)

    def __iter__(self):
        for xy, pid in zip(self.xys, self.ids):
            yield Point2D(xy, int(pid))

    def __getitem__(self, i):
        return Point2D(self.xys[i], int(self.ids[i]))


@dataclass
class Image:
    image_id: int
    qvec: np.ndarray
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    point3D_ids: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int64))
    has_pose: bool = True

    @property
    def points2D(self) -> _Points2DView:
        return _Points2DView(self.xys, self.point3D_ids)

    def cam_from_world(self) -> Rigid3d:
        return Rigid3d(qvec_to_rotmat(self.qvec), self.tvec)

    def projection_center(self) -> np.ndarray:
        return -qvec_to_rotmat(self.qvec).T @ self.tvec

    def observed_point3D_ids(self) -> np.ndarray:
        """ids of the 3-D points this image observes (``scripts/test.py:135`` without the Python loop)."""
        return self.point3D_ids[self.point3D_ids != INVALID_POINT3D]


@dataclass
class Point3D:
    xyz: np.ndarray
    color: np.ndarray
    error: float = -1.0


_POINT_NO_TRACK = np.dtype([("id", "<u8"), ("xyz", "<f8", 3), ("rgb", "u1", 3), ("error", "<f8"), ("track", "<u8")])


class _Points3DView:
    """Dictionary-like access ``rec.points3D[pid].xyz`` over array storage."""

    def __init__(self, rec: "Reconstruction"):
        self._rec = rec

    def __len__(self):
        return self._rec.num_points3D()

    def __contains__(self, pid):
        return int(pid) in self._rec._index()

    def __getitem__(self, pid) -> Point3D:
        i = self._rec._index()[int(pid)]
        r = self._rec
        return Point3D(r.point_xyz[i], r.point_rgb[i], float(r.point_error[i]))

    def keys(self):
        return self._rec._index().keys()

    def values(self):
        r = self._rec
        return (Point3D(r.point_xyz[i], r.point_rgb[i], float(r.point_error[i])) for i in range(len(r.point_ids)))


class Reconstruction:
    """Minimal COLMAP sparse model: cameras, registered images, 3-D points (array storage)."""

    def __init__(self, path: Union[str, Path, None] = None):
        self.cameras: Dict[int, Camera] = {}
        self.images: Dict[int, Image] = {}
        self.point_ids = np.zeros((0,), np.uint64)
        self.point_xyz = np.zeros((0, 3), np.float64)
        self.point_rgb = np.zeros((0, 3), np.uint8)
        self.point_error = np.zeros((0,), np.float64)
        # tracks of the first len(_track_len) points, CSR: (sum T, 2) int32 (image_id, point2D_idx) rows + lengths;
        # points appended later (add_points3D) have empty tracks
        self._track_flat = np.zeros((0, 2), np.int32)
        self._track_len = np.zeros((0,), np.int64)
        self._idx: Optional[dict] = None
        if path is not None:
            self.read(path)

    @property
    def _tracks(self) -> list:
        """Per-point (T,2) int32 arrays (a list view of the CSR storage)."""
        cuts = np.cumsum(self._track_len)[:-1] if len(self._track_len) else []
        return np.split(self._track_flat, cuts) if len(self._track_len) else []

    @_tracks.setter
    def _tracks(self, tracks) -> None:
        tracks = [np.asarray(t, dtype=np.int32).reshape(-1, 2) for t in tracks]
        self._track_len = np.array([len(t) for t in tracks], dtype=np.int64)
        self._track_flat = np.concatenate(tracks).astype(np.int32) if tracks else np.zeros((0, 2), np.int32)

    # -- queries ------------------------------------------------------------------------
    @property
    def points3D(self) -> _Points3DView:
        return _Points3DView(self)

    def _index(self) -> dict:
        if self._idx is None:
            self._idx = {int(p): i for i, p in enumerate(self.point_ids)}
        return self._idx

    def num_reg_images(self) -> int:
        return sum(1 for im in self.images.values() if im.has_pose)

    def num_points3D(self) -> int:
        return len(self.point_ids)

    def xyz_of(self, ids: np.ndarray) -> np.ndarray:
        """(n,3) coordinates of the given point ids (vectorised ``[rec.points3D[p].xyz for p in ids]``,
        ``scripts/test.py:139``)."""
        cache = getattr(self, "_lookup", None)
        if cache is None or cache[0] is not self.point_ids:          # id array replaced (read / add_points3D) -> rebuild
            n = len(self.point_ids)
            top = int(self.point_ids.max()) if n else 0
            if top <= 8 * n + (1 << 20):                             # COLMAP ids are dense: direct table id -> row
                table = np.full(top + 1, -1, np.int64)
                table[self.point_ids.astype(np.int64)] = np.arange(n, dtype=np.int64)
                cache = (self.point_ids, table, None, None)
            else:
                order = np.argsort(self.point_ids, kind="stable")
                cache = (self.point_ids, None, order, self.point_ids[order])
            self._lookup = cache
        _, table, order, sorted_ids = cache
        ids = np.asarray(ids)
        if table is not None:
            idx = ids.astype(np.int64)
            if len(idx) and (idx.min() < 0 or idx.max() >= len(table)):
                raise KeyError("xyz_of: unknown point3D id")
            rows = table[idx]
            if len(rows) and rows.min() < 0:
                raise KeyError("xyz_of: unknown point3D id")
            return self.point_xyz[rows]
        want = ids.astype(np.uint64)
        pos = np.minimum(np.searchsorted(sorted_ids, want), max(len(sorted_ids) - 1, 0))
        if len(want) and (len(sorted_ids) == 0 or np.any(sorted_ids[pos] != want)):
            raise KeyError("xyz_of: unknown point3D id")
        return self.point_xyz[order[pos]]

    # -- reading ------------------------------------------------------------------------
    def read(self, path: Union[str, Path]) -> None:
        """Binary model if ``cameras.bin`` exists, else the text model (like ``pycolmap.Reconstruction(path)``)."""
        path = Path(path)
        if (path / "cameras.bin").exists():
            self.read_binary(path)
        elif (path / "cameras.txt").exists():
            self.read_text(path)
        else:
            raise FileNotFoundError(f"{path}: neither cameras.bin nor cameras.txt found")

    def read_text(self, path: Union[str, Path]) -> None:
        """COLMAP text model: ``cameras.txt`` (ID MODEL W H PARAMS...), ``images.txt`` (two lines per image:
        ID QW QX QY QZ TX TY TZ CAMERA_ID NAME / X Y POINT3D_ID ...), ``points3D.txt`` (ID X Y Z R G B ERROR TRACK...)."""
        path = Path(path)
        rows = lambda f: [ln for ln in (path / f).read_text().splitlines() if ln.strip() and not ln.startswith("#")]
        self.cameras = {}
        for ln in rows("cameras.txt"):
            t = ln.split()
            if t[1] not in _MODEL_IDS:
                raise ValueError(f"cameras.txt: unknown camera model {t[1]}")
            self.cameras[int(t[0])] = Camera(int(t[0]), _MODEL_IDS[t[1]], int(t[2]), int(t[3]), np.array(t[4:], dtype=np.float64))
        self.images = {}
        lines = [ln for ln in (path / "images.txt").read_text().splitlines() if not ln.startswith("#")]
        lines = lines[next((i for i, ln in enumerate(lines) if ln.strip()), len(lines)):]
        for i in range(0, len(lines) - 1, 2):
            head = lines[i].split()
            if len(head) < 10:
                continue
            obs = np.array(lines[i + 1].split(), dtype=np.float64).reshape(-1, 3) if lines[i + 1].strip() else np.zeros((0, 3))
            self.images[int(head[0])] = Image(int(head[0]), np.array(head[1:5], dtype=np.float64), np.array(head[5:8], dtype=np.float64),
                                              int(head[8]), " ".join(head[9:]), obs[:, :2].copy(), obs[:, 2].astype(np.int64))
        ids, xyz, rgb, err, tracks = [], [], [], [], []
        for ln in rows("points3D.txt"):
            t = ln.split()
            ids.append(int(t[0])); xyz.append([float(v) for v in t[1:4]]); rgb.append([int(v) for v in t[4:7]]); err.append(float(t[7]))
            tracks.append(np.array(t[8:], dtype=np.int32).reshape(-1, 2))
        self.point_ids = np.array(ids, dtype=np.uint64)
        self.point_xyz = np.array(xyz, dtype=np.float64).reshape(-1, 3)
        self.point_rgb = np.array(rgb, dtype=np.uint8).reshape(-1, 3)
        self.point_error = np.array(err, dtype=np.float64)
        self._tracks, self._idx = tracks, None

    def read_binary(self, path: Union[str, Path]) -> None:
        path = Path(path)
        self._read_cameras(path / "cameras.bin")
        self._read_images(path / "images.bin")
        self._read_points(path / "points3D.bin")

    def _read_cameras(self, f: Path) -> None:
        buf = f.read_bytes()
        (n,), off = struct.unpack_from("<Q", buf, 0), 8
        self.cameras = {}
        for _ in range(n):
            cid, model, w, h = struct.unpack_from("<iiQQ", buf, off)
            off += 24
            if model not in CAMERA_MODELS:
                raise ValueError(f"{f}: unknown camera model id {model}")
            k = CAMERA_MODELS[model][1]
            params = np.frombuffer(buf, "<f8", k, off).copy()
            off += 8 * k
            self.cameras[cid] = Camera(cid, model, int(w), int(h), params)

    def _read_images(self, f: Path) -> None:
        buf = f.read_bytes()
        (n,), off = struct.unpack_from("<Q", buf, 0), 8
        self.images = {}
        obs = np.dtype([("xy", "<f8", 2), ("pid", "<i8")])
        for _ in range(n):
            iid = struct.unpack_from("<i", buf, off)[0]
            qt = np.frombuffer(buf, "<f8", 7, off + 4)
            cam = struct.unpack_from("<i", buf, off + 60)[0]
            off += 64
            end = buf.index(b"\x00", off)
            name = buf[off:end].decode("utf-8")
            off = end + 1
            (m,) = struct.unpack_from("<Q", buf, off)
            off += 8
            o = np.frombuffer(buf, obs, m, off)
            off += obs.itemsize * m
            self.images[iid] = Image(iid, qt[:4].copy(), qt[4:].copy(), cam, name, o["xy"].copy(), o["pid"].copy())

    def _read_points(self, f: Path) -> None:
        """``points3D.bin``: per point a 43-byte head (id u64, xyz 3 f64, rgb 3 u8, error f64, track length u64)
        followed by the track (8 bytes per element).  Only the walk over the record boundaries is sequential
        (one integer per point, and it stops as soon as the rest of the file can hold no track element);
        heads and tracks are then gathered with array operations."""
        raw = f.read_bytes()
        buf = np.frombuffer(raw, np.uint8)
        n = struct.unpack_from("<Q", raw, 0)[0]
        H = _POINT_NO_TRACK.itemsize                      # 43
        off, size, unpack = 8, len(raw), struct.Struct("<Q").unpack_from
        walked, left = [], n * H                          # left = bytes the remaining heads need
        while left and size - off != left:                # records with (possibly) non-empty tracks
            walked.append(off)
            off += H + 8 * unpack(raw, off + H - 8)[0]
            left -= H
        i = len(walked)
        starts = np.array(walked, dtype=np.int64)
        if off + left != size:
            raise ValueError(f"{f}: file size does not match the {n} points announced")
        # heads of the walked records are gathered; the rest of the file (no track elements) IS a record array
        parts = []
        if i:
            parts.append(buf[starts[:, None] + np.arange(H)].reshape(-1).view(_POINT_NO_TRACK))
        if n - i:
            parts.append(np.frombuffer(raw, _POINT_NO_TRACK, n - i, off))
        head = np.concatenate(parts) if parts else np.zeros(0, _POINT_NO_TRACK)
        self.point_ids = head["id"].astype(np.uint64)
        self.point_xyz = head["xyz"].astype(np.float64).reshape(-1, 3)
        self.point_rgb = head["rgb"].astype(np.uint8).reshape(-1, 3)
        self.point_error = head["error"].astype(np.float64)
        self._track_len = head["track"].astype(np.int64)
        if int(self._track_len.sum()):
            is_head = np.zeros(off, bool)                 # tracks live inside [8, off), between the walked heads
            is_head[:8] = True
            is_head[(starts[:, None] + np.arange(H)).reshape(-1)] = True
            self._track_flat = buf[:off][~is_head].view("<i4").reshape(-1, 2).astype(np.int32)
        else:
            self._track_flat = np.zeros((0, 2), np.int32)
        self._idx = None

    # -- growing ------------------------------------------------------------------------
    def add_points3D(self, xyz: np.ndarray, rgb: np.ndarray) -> np.ndarray:
        """Append points with empty tracks (``rec.add_point3D(xyz, Track(), color)`` for every row,
        ``scripts/test.py:355-358``); returns their new ids."""
        xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
        rgb = np.asarray(rgb, dtype=np.uint8).reshape(-1, 3)
        if len(xyz) != len(rgb):
            raise ValueError("xyz and rgb row counts differ")
        first = int(self.point_ids.max()) + 1 if len(self.point_ids) else 1
        new_ids = np.arange(first, first + len(xyz), dtype=np.uint64)
        self.point_ids = np.concatenate([self.point_ids, new_ids])
        self.point_xyz = np.concatenate([self.point_xyz, xyz])
        self.point_rgb = np.concatenate([self.point_rgb, rgb])
        self.point_error = np.concatenate([self.point_error, np.full(len(xyz), -1.0)])
        self._idx = None
        return new_ids

    # -- writing ------------------------------------------------------------------------
    def write_binary(self, path: Union[str, Path], dense=None, dense_total: Optional[int] = None, chunk_points: Optional[int] = None) -> dict:
        """``rec.write_binary(path)`` (``scripts/test.py:363``).  ``dense``: a DEVICE cloud (``FusedCloud``) whose points
        are written behind this reconstruction's own, exactly as if ``add_points3D(dense.points, dense.colors)`` had
        been called first -- but formatted on the GPU and streamed to the file chunk by chunk
        (``model_writer.write_dense_records``), never materialised on the host.  ``dense_total``: the number of dense
        records the file is announced to hold when this call writes only some (or none) of them -- the other ranks of a
        multi-GPU run then write their slices with ``model_writer.write_dense_at`` at the returned offsets.
        Returns ``{"dense_offset": byte offset of the first dense record, "first_dense_id": its id}``."""
        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        n_dense = 0 if dense is None else len(dense)
        n_announced = n_dense if dense_total is None else int(dense_total)
        with open(path / "cameras.bin", "wb") as f:
            f.write(struct.pack("<Q", len(self.cameras)))
            for c in self.cameras.values():
                f.write(struct.pack("<iiQQ", c.camera_id, c.model_id, c.width, c.height))
                f.write(np.asarray(c.params, "<f8").tobytes())
        with open(path / "images.bin", "wb") as f:
            reg = [im for im in self.images.values() if im.has_pose]
            f.write(struct.pack("<Q", len(reg)))
            obs = np.dtype([("xy", "<f8", 2), ("pid", "<i8")])
            for im in reg:
                f.write(struct.pack("<i", im.image_id))
                f.write(np.concatenate([im.qvec, im.tvec]).astype("<f8").tobytes())
                f.write(struct.pack("<i", im.camera_id))
                f.write(im.name.encode("utf-8") + b"\x00")
                f.write(struct.pack("<Q", len(im.point3D_ids)))
                o = np.empty(len(im.point3D_ids), obs)
                o["xy"], o["pid"] = im.xys, im.point3D_ids
                f.write(o.tobytes())
        with open(path / "points3D.bin", "wb") as f:
            n_old = len(self._track_len)                             # points that may carry tracks come first
            f.write(struct.pack("<Q", len(self.point_ids) + n_announced))
            if n_old:                                                # heads scattered, tracks poured into the gaps
                H = _POINT_NO_TRACK.itemsize
                head = np.empty(n_old, _POINT_NO_TRACK)
                head["id"], head["xyz"], head["rgb"] = self.point_ids[:n_old], self.point_xyz[:n_old], self.point_rgb[:n_old]
                head["error"], head["track"] = self.point_error[:n_old], self._track_len
                sizes = H + 8 * self._track_len
                starts = np.cumsum(sizes) - sizes
                out = np.empty(int(sizes.sum()), np.uint8)
                at = (starts[:, None] + np.arange(H)).reshape(-1)
                out[at] = head.view(np.uint8)
                gaps = np.ones(len(out), bool)
                gaps[at] = False
                out[gaps] = np.ascontiguousarray(self._track_flat, dtype="<i4").view(np.uint8).reshape(-1)
                out.tofile(f)
            n_new = len(self.point_ids) - n_old                      # dense points: one structured array
            if n_new:
                rec = np.empty(n_new, _POINT_NO_TRACK)
                rec["id"], rec["xyz"], rec["rgb"] = self.point_ids[n_old:], self.point_xyz[n_old:], self.point_rgb[n_old:]
                rec["error"], rec["track"] = self.point_error[n_old:], 0
                rec.tofile(f)
            first_id = int(self.point_ids.max()) + 1 if len(self.point_ids) else 1
            f.flush()
            where = {"dense_offset": f.tell(), "first_dense_id": first_id}
            if n_dense:
                from .model_writer import DEFAULT_CHUNK_POINTS, write_dense_records
                write_dense_records(f, dense, first_id, chunk_points or DEFAULT_CHUNK_POINTS)
            if n_announced > n_dense:                                # room for the slices the other ranks write in place
                f.truncate(where["dense_offset"] + n_announced * _POINT_NO_TRACK.itemsize)
        return where


def load_colmap_model(model_path: Union[str, Path]) -> Reconstruction:
    """``src/depthdensifier/utils.py:10-51``: a model directory (binary or text files)."""
    p = Path(model_path)
    if not p.is_dir():
        raise ValueError(f"{p}: expected a COLMAP model directory (cameras/images/points3D .bin or .txt)")
    return Reconstruction(p)


def write_ply(path: Union[str, Path], points: np.ndarray, colors: Optional[np.ndarray] = None,
              normals: Optional[np.ndarray] = None) -> None:
    """Binary little-endian PLY of a fused cloud."""
    n = len(points)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.empty(n, np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = np.asarray(points, np.float32).T
    if normals is not None:
        rec["nx"], rec["ny"], rec["nz"] = np.asarray(normals, np.float32).T
    if colors is not None:
        rec["red"], rec["green"], rec["blue"] = np.asarray(colors, np.uint8).T
    names = {"<f4": "float", "u1": "uchar"}
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {n}\n" + \
        "".join(f"property {names[t]} {k}\n" for k, t in fields) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        rec.tofile(f)
