"""Densify one COLMAP scan (drop-in for the reference's ``scripts/test.py``).

Same configuration tree and flags (``--paths.recon-path``, ``--processing.downsample-density``,
``--filtering.vote-threshold`` ...); the work is done by ``depthdensifier_amd.pipeline`` on the
MI355X kernels.  Imported as module ``test`` by ``scripts/run_batch.py`` like in the reference.
"""

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from depthdensifier_amd.cli import cli  # noqa: E402
from depthdensifier_amd.pipeline import (  # noqa: E402,F401
    FilteringConfig, MoGeConfig, PathsConfig, ProcessingConfig, RefinerConfig, ScriptConfig, main,
)

if __name__ == "__main__":
    cli(main)
