"""depthdensifier_amd -- MI355X-native densification core with DepthDensifier's surface.

Only the per-view depth->points hot path of OpsiClear/DepthDensifier is rebuilt here
(``scripts/test.py:203-244, 262-266``; ``src/depthdensifier/visualizer.py:246-376``):
unproject, camera-to-world, cull, compact and fuse, as hand-written HIP kernels for
gfx950 behind the C ABI of ``include/ddcore.h``.  Importing this package loads
``libddcore.so`` and fails loudly if it has not been built.
"""

__version__ = "0.1.0"

from ._lib import DDCoreError  # noqa: F401
from .densify import (  # noqa: F401
    BatchPlan,
    CapturedChain,
    CloudBuilder,
    FusedCloud,
    GuessPolicy,
    ViewBatch,
    camera_blocks,
    capture_chain,
    count_valid,
    fuse_batches,
    intrinsics_matrix,
    plan_batch,
    unproject_views,
)

from .depth_refiner import DepthRefiner, RefinerConfig  # noqa: F401,E402
from .filtering import FilteringConfig, compact_cloud, filter_cameras, filter_floaters, floater_votes  # noqa: F401,E402

__all__ = [
    "DepthRefiner", "RefinerConfig", "FilteringConfig", "compact_cloud", "filter_cameras", "filter_floaters", "floater_votes",
    "CloudBuilder", "FusedCloud", "ViewBatch", "camera_blocks", "count_valid", "fuse_batches",
    "intrinsics_matrix", "plan_batch", "BatchPlan", "unproject_views", "DDCoreError", "__version__", "CapturedChain", "capture_chain", "GuessPolicy",
]
