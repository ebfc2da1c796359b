"""``depthdensifier.visualizer`` of the reference, data path only (see depthdensifier_amd.visualizer)."""
from depthdensifier_amd.visualizer import COLMAPVisualizer, PointCloud  # noqa: F401
