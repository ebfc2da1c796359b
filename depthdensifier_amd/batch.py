"""Batch driver: densify every scan folder under a root (the job of the reference's
``scripts/run_batch.py:41-110``).

Conventions kept from the reference: a scan is a sub-directory holding ``images/`` and
``sparse/0/``; incomplete folders are skipped with a message; the output model goes to
``<output_dir>/<scan>/sparse/0``; an exception inside one scan is reported and recorded as
``FAILED`` without stopping the batch; a duration table closes the run; the embedded
``ScriptConfig`` is shared by all scans and only its ``paths`` are replaced per scan (plus
``moge.cache_dir`` when it contains the placeholder ``{scan}``, e.g. ``/data/{scan}/moge_cache``).

Multi-GPU (BASELINE config 4, "all scenes back-to-back on 8 GPUs"): launched with
``python -m torch.distributed.run --nproc-per-node N scripts/run_batch.py ...`` every rank owns one
GPU (``LOCAL_RANK``) and a subset of the scans -- scans are independent, so there is no data-path
collective; the scans are dealt longest-first (by image count) to the least loaded rank, the same
on every rank, and the per-rank outcomes are merged over a gloo group so that rank 0 prints one table.
"""

from __future__ import annotations

import os
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


def assign_scans(jobs: List[ScanJob], world_size: int) -> List[int]:
    """Owner rank of every job: most expensive scan first (images x pixels per image; incomplete folders cost
    nothing) to the least loaded rank, ties to the lower rank / earlier name -- a pure function of the folders."""
    def cost(job: ScanJob) -> int:
        if not job.complete:
            return 0
        n = sum(1 for f in job.images.iterdir() if f.is_file())
        try:                                             # mean camera size from the (tiny) camera file; scenes differ 2.5x
            from .colmap_io import Reconstruction
            rec = Reconstruction()
            if (job.recon / "cameras.bin").exists():
                rec._read_cameras(job.recon / "cameras.bin")
            cams = list(rec.cameras.values())
            px = sum(c.width * c.height for c in cams) // len(cams) if cams else 1
        except Exception:                                # noqa: BLE001 -- unreadable model: fall back to the image count
            px = 1
        return n * max(px, 1)

    costs = [cost(j) for j in jobs]
    load = [0] * world_size
    owner = [0] * len(jobs)
    for k in sorted(range(len(jobs)), key=lambda k: (-costs[k], jobs[k].name)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        owner[k] = r
        load[r] += max(costs[k], 1)
    return owner


def _launch_env() -> Tuple[int, int, int]:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    return rank, int(os.environ.get("LOCAL_RANK", str(rank))), world


def main(batch_config: BatchConfig, run_scan: Optional[Callable[[ScriptConfig], object]] = None) -> List[Tuple[str, Outcome]]:
    """Run the densification on all valid scan folders found in the root directory."""
    run_scan = run_scan or densify_scan
    started = time.time()
    rank, local_rank, world = _launch_env()
    root = batch_config.root_dir.resolve()
    if not root.is_dir():
        print(f"Error: Root directory not found at {root}")
        return []
    jobs = list(discover_scans(root, batch_config.output_dir))
    print(f"Found {len(jobs)} potential scan folders in {root}.")
    owner = assign_scans(jobs, world)
    if world > 1:
        import torch
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        created_group = not dist.is_initialized()
        if created_group:                            # control plane only (the merged report): gloo
            dist.init_process_group("gloo")
        batch_config.config.processing.shard_views = False      # whole scans per rank, not views
        print(f"[rank {rank}/{world}] owns {sum(1 for o in owner if o == rank)} of {len(jobs)} scan folders")
    outcomes: List[Tuple[str, Outcome]] = []
    shared = batch_config.config                     # one config object for the whole batch
    cache_template = shared.moge.cache_dir           # may hold "{scan}": cached maps live per scan folder
    for job in (j for j, o in zip(jobs, owner) if o == rank):
        print("\n" + "=" * 80 + f"\nProcessing scan: {job.name}\n" + "=" * 80)
        if not job.complete:
            print(f"Skipping '{job.name}': Missing 'sparse/0' or 'images' directory.")
            continue
        shared.paths = PathsConfig(recon_path=job.recon, image_dir=job.images, output_model_dir=job.output_model)
        if cache_template is not None and "{scan}" in str(cache_template):
            shared.moge.cache_dir = Path(str(cache_template).format(scan=job.name))
        tick = time.time()
        try:
            run_scan(shared)
        except Exception as err:                     # noqa: BLE001 -- a broken scan must not end the batch
            outcomes.append((job.name, "FAILED"))
            print(f"\n!!!!!!!!!!\nAn error occurred while processing '{job.name}': {err}\n!!!!!!!!!!")
        else:
            outcomes.append((job.name, time.time() - tick))
            print(f"\nSuccessfully finished processing scan: {job.name}")
    if world > 1:
        gathered: List[Optional[List[Tuple[str, Outcome]]]] = [None] * world
        dist.all_gather_object(gathered, outcomes)
        by_name = {name: val for part in gathered for name, val in (part or [])}
        outcomes = [(j.name, by_name[j.name]) for j in jobs if j.name in by_name]      # folder order, as on one GPU
        if created_group:
            dist.destroy_process_group()
        if rank != 0:
            return outcomes
    print(format_report(outcomes, time.time() - started))
    return outcomes
