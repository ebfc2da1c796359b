"""The committed fixtures ARE what the reference returns: re-run tests/golden/make_goldens.py (which imports the
reference's functions) into a scratch directory and compare every array with the committed .npz files.
Runs where /root/reference exists (the build container); skipped elsewhere (the GPU box has no reference)."""

import importlib.util
import sys
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).parent / "golden"


@pytest.mark.skipif(not Path("/root/reference/src/depthdensifier").is_dir(), reason="the reference tree is not mounted here")
def test_committed_goldens_equal_a_fresh_capture(tmp_path, monkeypatch):
    monkeypatch.setattr(sys, "dont_write_bytecode", True)              # the reference tree is read-only
    spec = importlib.util.spec_from_file_location("make_goldens_fresh", GOLDEN / "make_goldens.py")
    mg = importlib.util.module_from_spec(spec)
    sys.path.insert(0, str(GOLDEN))
    try:
        spec.loader.exec_module(mg)
        monkeypatch.setattr(mg, "OUT", tmp_path)
        mg.build_all()
    finally:
        sys.path.remove(str(GOLDEN))
    for name in mg.ALL_FIXTURES:
        fresh, kept = np.load(tmp_path / name), np.load(GOLDEN / name)
        assert sorted(fresh.files) == sorted(kept.files), name
        for k in kept.files:
            a, b = fresh[k], kept[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), (name, k)
