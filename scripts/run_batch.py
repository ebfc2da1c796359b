"""Entry point with the reference's name and flags; the work is in ``depthdensifier_amd.batch``.

    python scripts/run_batch.py --root-dir data/scans --output-dir results \\
        --config.processing.downsample-density 1 --config.filtering.vote-threshold 3
"""

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from depthdensifier_amd.batch import BatchConfig, main  # noqa: E402,F401
from depthdensifier_amd.cli import cli  # noqa: E402
from depthdensifier_amd.pipeline import PathsConfig, ScriptConfig  # noqa: E402,F401

if __name__ == "__main__":
    cli(main)
