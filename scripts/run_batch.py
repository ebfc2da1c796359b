"""Run the densification on every scan folder of a root directory (drop-in for the reference's
``scripts/run_batch.py:26-110``): ``<scan>/images`` + ``<scan>/sparse/0`` ->
``<output_dir>/<scan>/sparse/0``; a failing scan is reported and the batch continues.

    python scripts/run_batch.py --root-dir data/scans --output-dir results \\
        --config.processing.downsample-density 1 --config.filtering.vote-threshold 3
"""

import os
import sys
import time
from dataclasses import dataclass, field
from pathlib import Path

sys.path.append(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from test import main as densify_main, ScriptConfig, PathsConfig  # noqa: E402  (sibling scripts/test.py)
from depthdensifier_amd.cli import cli  # noqa: E402


@dataclass
class BatchConfig:
    """Configuration for the batch processing script."""
    root_dir: Path
    """The root directory containing the individual scan folders."""
    output_dir: Path
    """The directory to save the output point clouds and models."""
    config: ScriptConfig = field(default_factory=ScriptConfig)


def main(batch_config: BatchConfig):
    """Runs the densification on all valid scan folders found in the root directory."""
    t_batch = time.time()
    root = batch_config.root_dir.resolve()
    if not root.is_dir():
        print(f"Error: Root directory not found at {root}")
        return []
    scans = sorted(d for d in root.iterdir() if d.is_dir())
    print(f"Found {len(scans)} potential scan folders in {root}.")
    reports = []
    for scan in scans:
        print(f"\n{'=' * 80}\nProcessing scan: {scan.name}\n{'=' * 80}")
        recon, images = scan / "sparse" / "0", scan / "images"
        out_model = batch_config.output_dir / scan.name / "sparse" / "0"
        if not recon.is_dir() or not images.is_dir():
            print(f"Skipping '{scan.name}': Missing 'sparse/0' or 'images' directory.")
            continue
        run_config = batch_config.config                       # shared and mutated, like the reference
        run_config.paths = PathsConfig(recon_path=recon, image_dir=images, output_model_dir=out_model)
        t0 = time.time()
        try:
            densify_main(run_config)
            reports.append((scan.name, time.time() - t0))
            print(f"\nSuccessfully finished processing scan: {scan.name}")
        except Exception as e:                                 # one failed scan must not stop the batch
            reports.append((scan.name, "FAILED"))
            print(f"\n!!!!!!!!!!\nAn error occurred while processing '{scan.name}': {e}\n!!!!!!!!!!")
    print(f"\n\n{'=' * 63}\n{'Batch Processing Time Report':^63}\n{'=' * 63}")
    print(f"{'Scan Name':<40} | {'Duration (s)':>18}\n{'-' * 40}-+-{'-' * 18}")
    for name, dur in reports:
        print(f"{name:<40} | {dur:>18.2f}" if isinstance(dur, float) else f"{name:<40} | {dur:>18}")
    print(f"{'-' * 40}-+-{'-' * 18}\n{'Total Time':<40} | {time.time() - t_batch:>18.2f}\n{'=' * 63}\n")
    return reports


if __name__ == "__main__":
    cli(main)
