"""``depthdensifier.utils.load_colmap_model`` of the reference, on the package's own COLMAP reader."""
from depthdensifier_amd.colmap_io import load_colmap_model  # noqa: F401
