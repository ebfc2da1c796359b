"""Batch driver: densify every scan folder under a root (the job of the reference's
``scripts/run_batch.py:41-110``).

Conventions kept from the reference: a scan is a sub-directory holding ``images/`` and
``sparse/0/``; incomplete folders are skipped with a message; the output model goes to
``<output_dir>/<scan>/sparse/0``; an exception inside one scan is reported and recorded as
``FAILED`` without stopping the batch; a duration table closes the run; the embedded
``ScriptConfig`` is shared by all scans and only its ``paths`` are replaced per scan.
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, Iterator, List, Optional, Tuple, Union

from .pipeline import PathsConfig, ScriptConfig, main as densify_scan

Outcome = Union[float, str]          # seconds, or "FAILED"


@dataclass
class BatchConfig:
    """Configuration for the batch processing script."""
    root_dir: Path
    """The root directory containing the individual scan folders."""
    output_dir: Path
    """The directory to save the output point clouds and models."""
    config: ScriptConfig = field(default_factory=ScriptConfig)


@dataclass
class ScanJob:
    name: str
    recon: Path
    images: Path
    output_model: Path

    @property
    def complete(self) -> bool:
        return self.recon.is_dir() and self.images.is_dir()


def discover_scans(root: Path, output_dir: Path) -> Iterator[ScanJob]:
    """Sub-directories of ``root`` in name order, each described as a job."""
    for d in sorted(p for p in root.iterdir() if p.is_dir()):
        yield ScanJob(d.name, d / "sparse" / "0", d / "images", output_dir / d.name / "sparse" / "0")


def format_report(rows: List[Tuple[str, Outcome]], total_s: float, width: int = 63) -> str:
    name_w, val_w = 40, width - 40 - 3
    bar, rule = "=" * width, "-" * name_w + "-+-" + "-" * val_w
    cell = lambda v: f"{v:>{val_w}.2f}" if isinstance(v, float) else f"{v:>{val_w}}"
    lines = ["", "", bar, f"{'Batch Processing Time Report':^{width}}", bar,
             f"{'Scan Name':<{name_w}} | {'Duration (s)':>{val_w}}", rule]
    lines += [f"{n:<{name_w}} | {cell(v)}" for n, v in rows]
    lines += [rule, f"{'Total Time':<{name_w}} | {cell(float(total_s))}", bar, ""]
    return "\n".join(lines)


def main(batch_config: BatchConfig, run_scan: Optional[Callable[[ScriptConfig], object]] = None) -> List[Tuple[str, Outcome]]:
    """Run the densification on all valid scan folders found in the root directory."""
    run_scan = run_scan or densify_scan
    started = time.time()
    root = batch_config.root_dir.resolve()
    if not root.is_dir():
        print(f"Error: Root directory not found at {root}")
        return []
    jobs = list(discover_scans(root, batch_config.output_dir))
    print(f"Found {len(jobs)} potential scan folders in {root}.")
    outcomes: List[Tuple[str, Outcome]] = []
    shared = batch_config.config                     # one config object for the whole batch
    for job in jobs:
        print("\n" + "=" * 80 + f"\nProcessing scan: {job.name}\n" + "=" * 80)
        if not job.complete:
            print(f"Skipping '{job.name}': Missing 'sparse/0' or 'images' directory.")
            continue
        shared.paths = PathsConfig(recon_path=job.recon, image_dir=job.images, output_model_dir=job.output_model)
        tick = time.time()
        try:
            run_scan(shared)
        except Exception as err:                     # noqa: BLE001 -- a broken scan must not end the batch
            outcomes.append((job.name, "FAILED"))
            print(f"\n!!!!!!!!!!\nAn error occurred while processing '{job.name}': {err}\n!!!!!!!!!!")
        else:
            outcomes.append((job.name, time.time() - tick))
            print(f"\nSuccessfully finished processing scan: {job.name}")
    print(format_report(outcomes, time.time() - started))
    return outcomes
