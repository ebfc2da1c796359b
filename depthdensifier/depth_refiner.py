"""``depthdensifier.depth_refiner`` of the reference (see depthdensifier_amd.depth_refiner)."""
from depthdensifier_amd.depth_refiner import DepthRefiner, RefinerConfig  # noqa: F401
