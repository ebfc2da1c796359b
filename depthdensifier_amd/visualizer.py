"""``COLMAPVisualizer.add_rgbd_pointcloud`` on the MI355X core (SURVEY.md 8b ii, 8 a9).

Only the data path of the reference's visualizer is rebuilt -- the RGBD -> world point cloud
conversion ``add_rgbd_pointcloud`` / ``_depth_to_pointcloud`` / ``_transform_normals``
(``src/depthdensifier/visualizer.py:246-376``) and the ``PointCloud`` container (``:71-80``);
the Plotly figure code (``:378-951``) is presentation and out of scope.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .densify import unproject_views


@dataclass
class PointCloud:
    """Container for point cloud data (``visualizer.py:71-80``)."""
    points: np.ndarray
    colors: Optional[np.ndarray] = None
    normals: Optional[np.ndarray] = None
    name: str = "Point Cloud"
    visible: bool = True
    point_size: int = 1
    opacity: float = 0.8


@dataclass
class COLMAPVisualizer:
    """Collects point clouds; RGBD inputs are unprojected on the GPU."""

    point_clouds: list = field(default_factory=list)
    max_points_display: int = 100000

    def add_pointcloud(self, points, colors=None, normals=None, name="Point Cloud", visible=True, point_size=1,
                       opacity=0.8) -> None:
        self.point_clouds.append(PointCloud(points, colors, normals, name, visible, point_size, opacity))

    def add_rgbd_pointcloud(self, depth_map, rgb_image=None, K=None, cam_from_world=None, mask=None, normal_map=None,
                            name: str = "RGBD Point Cloud", **kwargs) -> np.ndarray:
        """Same signature, return value ((N,3) float64 world points) and errors as ``visualizer.py:246-289``:
        validity is ``mask > 0`` when a mask is given else ``depth > 0``; normals are produced (rotated
        to the world frame and re-normalised) only when both ``normal_map`` and ``mask`` are given."""
        if K is None or cam_from_world is None:
            raise ValueError("Camera intrinsics (K) and extrinsics (cam_from_world) are required")
        image = None if rgb_image is None else np.asarray(rgb_image)
        # colours (visualizer.py:337-342): x255 -> uint8 when the VALID colours' maximum is <= 1, otherwise handed on in the
        # image's own dtype.  uint8 results are gathered by the kernel; a float image that stays float is gathered here by
        # the pixel index the kernel emits.
        keeps_dtype = image is not None and image.dtype != np.uint8
        fused = unproject_views(np.asarray(depth_map)[None], np.asarray(K, dtype=np.float64)[None],
                                np.asarray(cam_from_world, dtype=np.float64)[None],
                                mask=None if mask is None else np.asarray(mask)[None],
                                normal=None if normal_map is None else np.asarray(normal_map, dtype=np.float32)[None],
                                rgb=None if image is None else image[None],
                                semantics="viz", pixel_index=keeps_dtype, _allow_passthrough=True)
        cloud = fused.numpy()
        colors = cloud["colors"]
        if keeps_dtype and fused.rgb_passthrough is not None:
            colors = image.reshape(-1, image.shape[-1])[cloud["pixel_index"]]
        self.add_pointcloud(points=cloud["points"], colors=colors, normals=cloud["normals"], name=name, **kwargs)
        return cloud["points"]
