#!/usr/bin/env python3
"""Build three small synthetic scans under gpurun_out/batch_scans (not timed) for the sharded-batch rehearsal."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from scan_factory import make_scan
root = Path(sys.argv[1])
for name, V in (("s_big", 6), ("s_mid", 4), ("s_small", 3)):
    make_scan(root, name, V=V, H=96, W=128, seed=V)
print("scans ready")
