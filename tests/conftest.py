"""pytest configuration: markers, import paths, shared fixtures."""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_small():
    return dict(np.load(GOLDEN / "densify_small.npz"))


@pytest.fixture(scope="session")
def golden_vga():
    return dict(np.load(GOLDEN / "densify_vga.npz"))


@pytest.fixture(scope="session")
def vga_inputs(golden_vga):
    """Regenerate the BASELINE config-1 inputs from the seed and verify their digests."""
    from synth import make_views, sha

    V, H, W = (int(x) for x in golden_vga["in_shape"])
    d = make_views(int(golden_vga["in_seed"]), V, H, W, rho=0.8)
    for k in ("depth", "mask", "normal", "rgb", "conf", "cam_from_world"):
        want = bytes(golden_vga[f"in_sha_{k}"]).hex()
        if sha(d[k]) != want:
            pytest.skip(f"NumPy RNG stream drifted for '{k}'; VGA golden inputs cannot be regenerated")
    return d
