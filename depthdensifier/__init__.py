"""``depthdensifier`` -- the reference's import name, served by the MI355X-native core.

``from depthdensifier import DepthRefiner, RefinerConfig`` and
``from depthdensifier.visualizer import COLMAPVisualizer`` keep working
(``src/depthdensifier/__init__.py:3-6`` of the reference); everything resolves to
``depthdensifier_amd``.
"""

from depthdensifier_amd import DepthRefiner, RefinerConfig  # noqa: F401

__version__ = "0.1.0"
__all__ = ["DepthRefiner", "RefinerConfig"]
