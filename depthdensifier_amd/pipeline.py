"""The densification pipeline of ``scripts/test.py:main`` (``:95-370``) on the MI355X core.

Same configuration tree (``PathsConfig / MoGeConfig / ProcessingConfig / RefinerConfig /
FilteringConfig / ScriptConfig``, ``scripts/test.py:20-55``), same steps and messages, same
outputs (a COLMAP binary model with the dense points appended).  What changed is where the work
happens: depth / normal / mask stay on the GPU from the depth source through the refiner to the
densify kernels; the per-view NumPy block (``:203-244``) is ``CloudBuilder.append``; the
multi-view filter (``:269-335``) is ``filter_floaters``; the per-point ``add_point3D`` loop
(``:355-358``) is one bulk append.
"""

from __future__ import annotations

import dataclasses
import os
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .colmap_io import Reconstruction
from .densify import CloudBuilder, ViewBatch
from .depth_refiner import DepthRefiner, RefinerConfig
from .depth_source import make_depth_source
from .filtering import FilteringConfig, compact_cloud, floater_votes


@dataclass
class PathsConfig:
    """Configuration for input and output paths."""
    recon_path: Path = Path("data/360_v2/bicycle/sparse/0")
    image_dir: Path = Path("data/360_v2/bicycle/images")
    output_model_dir: Path = Path("results/0")


@dataclass
class MoGeConfig:
    """Configuration for the MoGe model."""
    checkpoint: Path = Path("models/moge/moge-2-vitl-normal/model.pt")
    cache_dir: Optional[Path] = None
    """Directory of precomputed <image stem>.npz maps (depth, mask, normal); used instead of MoGe."""


def _default_io_threads() -> int:
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:                      # not on Linux
        cores = os.cpu_count() or 4
    return max(2, min(16, cores - 2))


@dataclass
class ProcessingConfig:
    """Parameters for processing and densification."""
    pipeline_downsample_factor: int = 1
    """Factor to downsample images before processing. Larger is faster."""
    downsample_density: int = 32
    """Controls final point cloud density (1=densest)."""
    io_threads: int = field(default_factory=lambda: _default_io_threads())
    """Host threads that decode images / read cached maps ahead of the GPU (0 = inline, like the reference).  Default: the
    cores this process may use minus two, between 2 and 16 -- image decoding is what bounds a scan once the maps are cached."""
    views_per_launch: int = 16
    """Consecutive equally sized views densified by ONE kernel launch at full density (downsample_density = 1): their fits are
    enqueued back to back and read in one go, their maps stacked on the device, one ViewBatch / dd_unproject_compact with a
    transfer curve per view.  1 = a launch per view (rounds 1-2).  Same model either way.  (16: a launch of 2700 tiles fills the
    256 CUs five times over -- 0.72 of the roofline in a chain of such calls where 8 views reach 0.68, profiles/r05_streaming_*.txt.)"""
    exclusive_gpu: Optional[bool] = None
    """This process's densify stream has its GPU to itself -- the reference is ONE process that owns its GPU, and scripts/run_batch.py
    under torchrun is one process per GPU: the single-pass kernel then takes its tiles by workgroup index and small appends are
    chained across two side streams (CloudBuilder.exclusive_gpu; a launch that finds the GPU shared after all is redone and the
    cloud falls back to tickets by itself).  None (default) = yes, unless DD_EXCLUSIVE_GPU=0 or the job's local ranks outnumber
    the visible GPUs (a rehearsal: ranks share a card)."""
    shard_views: bool = True
    """Under torchrun (one process per GPU): shard this scan's views over the ranks.  The batch driver turns it
    off because it shards by scan."""
    sharded_model_write: bool = True
    """With sharded views: every rank writes its own slice of the dense points into points3D.bin (the clouds are never
    gathered: N PCIe links and N writers).  Off: the kept clouds travel to rank 0 over xGMI, which writes alone."""


@dataclass
class ScriptConfig:
    """Main configuration for the densification script."""
    paths: PathsConfig = field(default_factory=PathsConfig)
    moge: MoGeConfig = field(default_factory=MoGeConfig)
    processing: ProcessingConfig = field(default_factory=ProcessingConfig)
    refiner: RefinerConfig = field(default_factory=RefinerConfig)
    filtering: FilteringConfig = field(default_factory=FilteringConfig)


def _processing_size(path: Path, factor: int) -> tuple[int, int]:
    """(width, height) a view is processed at: the image FILE's size over ``pipeline_downsample_factor``
    (``scripts/test.py:149-152``) -- header-only read, the pixels are not decoded."""
    from PIL import Image as PILImage
    with PILImage.open(path) as img:
        w, h = img.size
    return w // factor, h // factor


def _load_rgb(path: Path, factor: int) -> np.ndarray:
    from PIL import Image as PILImage
    img = PILImage.open(path).convert("RGB")                                   # scripts/test.py:146
    w, h = img.size
    img = img.resize((w // factor, h // factor), PILImage.Resampling.LANCZOS)  # :149-152
    return np.array(img)


class _Ranks:
    """The ranks one scan's views are sharded over: a torchrun launch (``WORLD_SIZE`` > 1) with
    ``processing.shard_views`` on -- one process per GPU, RCCL (``DD_DIST_BACKEND=gloo`` for rehearsals)."""

    def __init__(self, enabled: bool):
        self.rank, self.world, self.created = 0, 1, False
        if enabled and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch.distributed as dist
            self.rank, self.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", self.rank)) % torch.cuda.device_count())
            if not dist.is_initialized():
                backend = os.environ.get("DD_DIST_BACKEND", "nccl")
                extra = {"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}
                dist.init_process_group(backend, **extra)
                self.created = True

    def total(self, value: int, device) -> int:
        """Sum of ``value`` over the ranks."""
        if self.world == 1:
            return int(value)
        import torch.distributed as dist
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        return int(t.item())

    def close(self) -> None:
        if self.created:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


class _GroupRing:
    """Device-resident stacks for the views of ``R`` launch groups of ``K`` views each (round 6): a view's maps are uploaded -- on the copy
    stream, waiting for nothing but the densify launch that last read the stack -- straight to where the group's ONE ViewBatch will read
    them: no per-view allocations, no torch.stack of 41 MB per view, and no fork of the copy stream behind every view's fit (which put
    0.4 ms of gaps between the copies of consecutive views: profiles/r06_bench_pipeline.txt).  The masks and the refined maps (the
    filter's cache, scripts/test.py:197-201) live in two arrays of their own for the whole scan."""

    def __init__(self, n_views: int, K: int, H: int, W: int, device, work_dtype, R: int = 3):
        self.K, self.R, self.H, self.W, self.device, self.work_dtype = K, R, H, W, device, work_dtype
        self.depth: list = [None] * R                # (K,H,W) in the cache's dtype, allocated when the first view says which
        self.work: list = [None] * R                 # (K,H,W) in the refiner's working precision (the same tensor when the dtypes agree)
        self.normal = [torch.empty((K, H, W, 3), dtype=torch.float32, device=device) for _ in range(R)]
        self.rgb = [torch.empty((K, H, W, 3), dtype=torch.uint8, device=device) for _ in range(R)]
        self.mask_all = torch.empty((n_views, H, W), dtype=torch.bool, device=device)
        self.n_views = n_views
        self._refined_all = None                     # (n_views,H,W) float32, made when the first fused launch asks for it
        self.events: list = [None] * R               # behind the densify launch that last read stack r (compute stream)
        self.checks: list = [None] * R               # the builder's check behind it: what it covers is final, its maps may be overwritten

    @property
    def refined_all(self) -> torch.Tensor:
        if self._refined_all is None:
            self._refined_all = torch.empty((self.n_views, self.H, self.W), dtype=torch.float32, device=self.device)
        return self._refined_all

    def place(self, k: int, prepared: dict, rgb) -> Optional[dict]:
        """Where view ``k``'s arrays go, or None if they are not what the ring holds (another size or element type: the caller's other way)."""
        g, j = divmod(k, self.K)
        r = g % self.R
        d = prepared.get("depth")
        if d is None or tuple(d.shape) != (self.H, self.W) or tuple(rgb.shape) != (self.H, self.W, 3) or "normal" not in prepared or "mask" not in prepared:
            return None
        dt = torch.float16 if d.dtype.name == "float16" else torch.float32 if d.dtype.name == "float32" else None
        if dt is None or prepared["mask"].dtype.name not in ("bool", "uint8") or prepared["normal"].dtype.name != "float32":
            return None
        if self.depth[r] is None:
            self.depth[r] = torch.empty((self.K, self.H, self.W), dtype=dt, device=self.device)
            self.work[r] = self.depth[r] if dt == self.work_dtype else torch.empty((self.K, self.H, self.W), dtype=self.work_dtype, device=self.device)
        if self.depth[r].dtype != dt:
            return None
        return dict(r=r, j=j, depth=self.depth[r][j], work=self.work[r][j], mask=self.mask_all[k], normal=self.normal[r][j], rgb=self.rgb[r][j])


def _exclusive_gpu(processing: "ProcessingConfig") -> bool:
    """``ProcessingConfig.exclusive_gpu`` resolved: an explicit choice stands; otherwise on, unless the environment says
    ``DD_EXCLUSIVE_GPU=0`` or more local ranks than GPUs were started (they share a card)."""
    if processing.exclusive_gpu is not None:
        return bool(processing.exclusive_gpu)
    if os.environ.get("DD_EXCLUSIVE_GPU") is not None:
        return os.environ["DD_EXCLUSIVE_GPU"] == "1"
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return local_world <= max(1, torch.cuda.device_count())


def _votes_single(cloud, cached, depth_threshold):
    """Votes of every point against every view held by this process; views of one size share a launch."""
    votes = None
    groups: dict = {}
    for c in cached:
        groups.setdefault(tuple(c["depth"].shape), []).append(c)
    for views in groups.values():
        votes = floater_votes(cloud.points, cloud.normals, torch.stack([c["depth"] for c in views]),
                              np.stack([c["K"] for c in views]), np.stack([c["E"] for c in views]),
                              mask=torch.stack([c["mask"] for c in views]), depth_threshold=depth_threshold, votes=votes)
    return votes


def main(config: ScriptConfig, _loop_only: bool = False) -> dict:
    """Densify one COLMAP scan; returns a small report (counts, timings).  (``_loop_only``: stop behind the image loop and the
    fused cloud -- no filter, no model written: what ``bench.py`` times as the path's own share of a scan.)

    Under ``torchrun`` (one process per GPU) with ``processing.shard_views`` the scan's views are sharded
    contiguously over the ranks (``distributed.shard_views``): every rank refines and densifies its views into
    its own cloud, the multi-view filter runs sharded by points, the surviving per-GPU clouds are fused with the
    all-gatherv of ``distributed.gather_cloud`` (rank order = view order, so the result is the one-GPU cloud),
    and rank 0 writes the model."""
    t_total = time.time()
    if not torch.cuda.is_available():
        raise RuntimeError("the densification core needs an AMD GPU (no CPU fallback)")
    ranks = _Ranks(config.processing.shard_views)
    device = torch.device("cuda", torch.cuda.current_device())
    say = print if ranks.rank == 0 else (lambda *a, **k: None)
    try:
        return _run(config, ranks, device, say, t_total, _loop_only)
    finally:
        ranks.close()


def _run(config: ScriptConfig, ranks: _Ranks, device, say, t_total: float, loop_only: bool = False) -> dict:
    t0 = time.time()
    say(f"Loading depth source ({config.moge.cache_dir or config.moge.checkpoint})...")
    source = make_depth_source(config.moge.checkpoint, config.moge.cache_dir, device)
    say(f"-> Depth source ready in {time.time() - t0:.2f}s.")

    t0 = time.time()
    say(f"Loading COLMAP reconstruction from {config.paths.recon_path}...")
    rec = Reconstruction(config.paths.recon_path)
    say(f"Loaded model with {rec.num_reg_images()} images and {rec.num_points3D()} sparse points.")
    say(f"-> COLMAP reconstruction loaded in {time.time() - t0:.2f}s.")

    refiner_cfg = dataclasses.asdict(config.refiner)
    if ranks.rank != 0:
        refiner_cfg["verbose"] = 0
    refiner = DepthRefiner(**refiner_cfg)                                       # :118-119
    verbose = refiner_cfg["verbose"] > 0
    f = config.processing.pipeline_downsample_factor
    s = config.processing.downsample_density

    # view order: registered images by id (pycolmap iterates an unordered map; sorted is deterministic)
    image_list = [rec.images[i] for i in sorted(rec.images) if rec.images[i].has_pose]     # :130
    todo = [im for im in image_list if len(im.observed_point3D_ids()) > 0]      # :135-137
    num_views = len(todo)
    if ranks.world > 1:
        from .distributed import shard_views
        lo, hi = shard_views(num_views, ranks.world, ranks.rank)
        say(f"Sharding {num_views} views over {ranks.world} GPUs (rank 0: views [{lo}, {hi})).")
    else:
        lo, hi = 0, num_views
    mine = todo[lo:hi]
    # capacity from the size the views are really processed at -- the image files', which the reference reconciles
    # with the model through camera.rescale (:172-173); the sparse model's camera may have another size
    capacity = 0
    sizes = {}                                                                  # image name -> (width, height) it is processed at
    for im in mine:
        pw, ph = sizes[im.name] = _processing_size(config.paths.image_dir / im.name, f)
        capacity += (-(-ph // s)) * (-(-pw // s))
    # (views are appended a few at a time here: launches of 26 us whose rows sit in one chunk anyway -- the default rule applies: a
    # placed cloud when the scan is large or the arena already holds classified spares from an earlier scan, no scouting otherwise)
    builder = CloudBuilder(capacity, normals=True, colors=True, pixel_index=False, device=device, exclusive_gpu=_exclusive_gpu(config.processing))
    cached = []                                                                 # :128 cached_refinement_data
    stage = {"image_decode": 0.0, "depth_source": 0.0, "refine": 0.0, "densify": 0.0}     # host seconds per stage
    clock = time.perf_counter
    detail: dict = {}                                                          # finer: seconds per step of the loop (report["loop_detail"])

    refiner.trace = detail

    def lap(key: str, t_from: float) -> float:
        now = clock()
        detail[key] = detail.get(key, 0.0) + (now - t_from)
        return now

    read_rgb = getattr(source, "read_rgb", None)

    def fetch(im, slot=None):                                                   # no GPU work: safe on an I/O thread
        pw, ph = sizes[im.name]
        if slot is not None and read_rgb is not None:                           # cached image + staging slot: read natively, straight into the slot
            slot.wait()                                                         # the uploads of the slot's previous view are done
            rgb = read_rgb(im.name, (ph, pw), slot)
            if rgb is not None:
                return rgb, source.prepare(im.name, rgb, staging=slot), slot
        rgb = source.cached_rgb(im.name)                                        # the cache may hold the resized image itself
        if rgb is None or tuple(rgb.shape[:2]) != (ph, pw):
            rgb = _load_rgb(config.paths.image_dir / im.name, f)                # :145-152
        if slot is None:
            rgb = np.array(rgb) if isinstance(rgb, np.memmap) else rgb        # (a private, writable copy of a memory-mapped image)
            return rgb, source.prepare(im.name, rgb), None
        slot.wait()                                                             # the uploads of the slot's previous view are done
        rgb = slot.put("rgb", rgb)
        if getattr(source, "accepts_staging", False):
            return rgb, source.prepare(im.name, rgb, staging=slot), slot
        return rgb, source.prepare(im.name, rgb), slot

    # decode / cache reads run `ahead` views in front of the GPU on a small pool; order of consumption is unchanged.
    # Each view in flight owns a slot of pinned staging buffers (ahead + 1 of them, round-robin): the maps are uploaded by
    # asynchronous DMA instead of the driver's pageable copy on this thread.
    ahead = 2 * config.processing.io_threads
    # a cache that holds every map AND the decoded image as .npy files is read ahead by NATIVE threads (depth_source.NativeFeeder): no
    # second Python thread, nobody to share the interpreter lock with.  Anything else (images to decode, .npz archives): Python threads.
    feeder = None
    if config.processing.io_threads > 0 and mine and hasattr(source, "upload_staged") and os.environ.get("DD_NATIVE_PREFETCH", "1") == "1":
        try:
            from .depth_source import NativeFeeder
            feeder = NativeFeeder(source, [im.name for im in mine], sizes, config.processing.io_threads, ahead)
        except (FileNotFoundError, OSError, RuntimeError):
            feeder = None
    pool = ThreadPoolExecutor(max_workers=config.processing.io_threads) if config.processing.io_threads > 0 and feeder is None else None
    # the uploads of staged views run on a stream of their own, beside the kernels of the views before them
    copy_stream, fork_event = None, None
    upload_events = [] if os.environ.get("DD_PIPELINE_TRACE") == "1" else None      # measurement runs: the copies timed on their stream
    if config.processing.io_threads > 0 and hasattr(source, "upload_staged") and os.environ.get("DD_COPY_STREAM", "1") == "1":
        copy_stream, fork_event = torch.cuda.Stream(device), torch.cuda.Event()
        fork_event.record(torch.cuda.current_stream(device))                   # (creates the underlying event)
    # Every view of one size, staged uploads: the views' maps go straight into resident stacks (one per launch group)
    K_LAUNCH = 1 if verbose else max(1, int(config.processing.views_per_launch))
    ring = None
    if copy_stream is not None and not verbose and mine and len(set(sizes.values())) == 1 and os.environ.get("DD_GROUP_RING", "1") == "1":
        pw0, ph0 = next(iter(sizes.values()))
        ring = _GroupRing(len(mine), K_LAUNCH, ph0, pw0, device, refiner.dtype)
    from .depth_source import StagingSlot
    slots = [StagingSlot() for _ in range(ahead + 2)] if pool else []
    pending: deque = deque(pool.submit(fetch, im, slots[j % len(slots)]) for j, im in enumerate(mine[:ahead])) if pool else deque()
    t_loop = time.time()

    def begin(k, image) -> dict:
        """Everything of a view up to the ENQUEUED correspondence fit: nothing here waits for the GPU."""
        t0_ = clock()
        pts_world = rec.xyz_of(image.observed_point3D_ids())                    # :139
        t1 = lap("sparse_points", t0_)
        if feeder is not None:
            try:
                rgb, prepared, slot = feeder.get(k)
            except OSError:                                                     # this view's files are not what its size says: read it the other way
                rgb, prepared, slot = fetch(image)
        elif pool:
            rgb, prepared, slot = pending.popleft().result()
            if k + ahead < len(mine):                                           # its slot was released two views ago at the latest
                pending.append(pool.submit(fetch, mine[k + ahead], slots[(k + ahead) % len(slots)]))
        else:
            rgb, prepared, slot = fetch(image)
        new_h, new_w = rgb.shape[:2]
        t2 = lap("wait_for_io_thread", t1)
        staged = slot is not None and prepared is not None and hasattr(source, "upload_staged") and "rgb" in slot._bufs and rgb is not None \
            and all(a.dtype.name in ("float32", "float16", "uint8", "bool") for a in prepared.values())
        spot = ring.place(k, prepared, rgb) if (ring is not None and staged) else None
        if spot is not None:
            r_ = spot["r"]
            if spot["j"] == 0:                                                  # the first view of a group: the stack's last readers
                if ring.checks[r_] is not None:
                    ring.checks[r_].result()                                    # (long done: the batches that read the stack are final and released)
                    ring.checks[r_] = None
                if ring.events[r_] is not None:
                    from ._lib import lib as _l
                    _l.dd_stream_wait(copy_stream.cuda_stream, ring.events[r_].cuda_event)
            maps, rgb_dev = source.upload_staged(prepared, rgb, slot, device, copy_stream, None, upload_events, dest=spot)    # :161-168 + :215
            tx = t3 = lap("upload_maps", t2)
        elif staged:
            maps, rgb_dev = source.upload_staged(prepared, rgb, slot, device, copy_stream, fork_event, upload_events)    # :161-168 + :215: maps and colours up, slot freed
            tx = t3 = lap("upload_maps", t2)
        else:
            maps = source.infer(image.name, rgb, device, prepared=prepared)         # :161-168, stays on device
            tx = lap("upload_maps", t2)
            rgb_dev = torch.from_numpy(rgb).to(device, non_blocking=True)           # :215 the colours, uploaded with the maps
            if slot is not None:
                slot.release(torch.cuda.current_stream(device))                     # every upload from the slot is enqueued by now
            t3 = lap("upload_rgb_release_slot", tx)
        camera = rec.cameras[image.camera_id]
        camera.rescale(new_width=new_w, new_height=new_h)                       # :172-173 (in place, like the reference)
        E = image.cam_from_world().matrix()[:3, :]                              # :177
        K = camera.calibration_matrix()                                         # :178
        if verbose:
            print(f"\n--- Refining depth for {image.name} (ID: {image.image_id}) ---")
        normal = maps["normal"]
        if normal is None:
            normal = torch.zeros((new_h, new_w, 3), dtype=torch.float32, device=device)
        fuse = s == 1 and new_w <= 3071          # full density: the densify kernel applies the transfer curve itself
        tc = lap("camera", t3)
        handle = refiner.begin_refine(depth_map=maps["depth"], normal_map=normal, points3D=pts_world, cam_from_world=E, K=K,
                                      mask=maps["mask"], return_tensor=True, fit_only=fuse,
                                      working_out=None if spot is None else spot["work"])   # :179-186, first half
        t4 = lap("begin_refine", tc)
        stage["image_decode"] += t2 - t1; stage["depth_source"] += t3 - t2; stage["refine"] += t4 - t3
        # (the camera may be rescaled again by the next view before this one is finished: its intrinsics are taken now)
        return dict(rgb=rgb_dev, maps=maps, normal=normal, E=E, K=K, pinhole=camera.pinhole_params().copy(), handle=handle,
                    spot=None if spot is None else (spot["r"], spot["j"], k))

    def densify_run(run: list) -> None:
        """Consecutive views whose transfer curves the kernel applies itself, all of one size: ONE ViewBatch, one launch.
        :194 "refined_depth[~moge_mask] = 0" is the kernels' validity rule (mask AND depth > 0); :203-240 densify + append.
        Raw depth -> LUT + 3x3 median -> validity -> unprojection in one kernel; the refined maps it writes on the way are the
        filter's cache (:197-201).  Same bits as dd_refine_apply followed by the plain densify call, view by view."""
        spots = [v.get("spot") for v in run]
        in_ring = ring is not None and all(sp is not None for sp in spots) and all(
            spots[i][0] == spots[0][0] and spots[i][1] == spots[0][1] + i and spots[i][2] == spots[0][2] + i for i in range(len(run))) \
            and all(v["raw"].data_ptr() == ring.work[spots[0][0]][sp[1]].data_ptr() for v, sp in zip(run, spots))
        if in_ring:                                   # consecutive views of one resident stack: slices, nothing is copied
            r_, j0, k0 = spots[0]
            n_ = len(run)
            masks = ring.mask_all[k0:k0 + n_]
            batch = ViewBatch(ring.work[r_][j0:j0 + n_], np.stack([v["pinhole"] for v in run]), np.stack([v["E"] for v in run]), mask=masks,
                              normal=ring.normal[r_][j0:j0 + n_], rgb=ring.rgb[r_][j0:j0 + n_], stride=s, view_index_base=lo + len(cached), device=device,
                              refine=[v["curve"] for v in run], refined_out=ring.refined_all[k0:k0 + n_])
            builder.append(batch)
        else:
            st = (lambda key: torch.stack([v[key] for v in run])) if len(run) > 1 else (lambda key: run[0][key][None])
            masks = st("mask")
            batch = ViewBatch(st("raw"), np.stack([v["pinhole"] for v in run]), np.stack([v["E"] for v in run]), mask=masks, normal=st("normal"),
                              rgb=st("rgb"), stride=s, view_index_base=lo + len(cached), device=device,
                              refine=[v["curve"] for v in run], refined_out=True)
            builder.append(batch)
        for i, v in enumerate(run):
            cached.append(dict(depth=batch.refined[i], mask=masks[i], K=v["K"], E=v["E"]))         # :197-201

    def finish(group: list) -> None:
        """The fits' results (the first read waits for the group's GPU work, the others are there), then the densify launches:
        runs of consecutive views with a curve and one size go together, anything else (early exits of the refiner, a coarser
        density, very wide images) view by view.  View order is preserved."""
        t3 = clock()
        w0 = getattr(refiner, "wait_seconds", 0.0)
        results = [refiner.finish_refine(v["handle"]) for v in group]
        t4 = lap("finish_refine", t3)
        detail["finish_refine_of_which_waiting_for_the_gpu"] = detail.get("finish_refine_of_which_waiting_for_the_gpu", 0.0) + getattr(refiner, "wait_seconds", 0.0) - w0
        run: list = []
        for v, res in zip(group, results):
            maps = v["maps"]
            if res["refined_depth"] is None:
                item = dict(raw=res["raw_depth"], curve=res["curve"], mask=maps["mask"], normal=v["normal"], rgb=v["rgb"], pinhole=v["pinhole"],
                            E=v["E"], K=v["K"], spot=v.get("spot"))
                if run and (tuple(run[0]["raw"].shape) != tuple(item["raw"].shape) or run[0]["raw"].dtype != item["raw"].dtype or len(run) >= max(1, K_LAUNCH)):
                    densify_run(run); run = []
                run.append(item)
                continue
            if run:
                densify_run(run); run = []
            refined = res["refined_depth"]
            refined = refined if isinstance(refined, torch.Tensor) else torch.as_tensor(refined, device=device)
            refined = refined.float()
            if v.get("spot") is not None:                 # an early exit of the refiner hands the INPUT map back (depth_refiner.py:259 ...): here a
                r_ = v["spot"][0]                         # view of a resident stack, which the views of three groups on overwrite -- the filter's
                stacks = {t.untyped_storage().data_ptr() for t in (ring.depth[r_], ring.work[r_]) if t is not None}      # cache keeps a copy
                if refined.untyped_storage().data_ptr() in stacks:
                    refined = refined.clone()
            batch = ViewBatch(refined, v["pinhole"][None], v["E"][None], mask=maps["mask"], normal=v["normal"], rgb=v["rgb"],
                              stride=s, view_index_base=lo + len(cached), device=device)
            builder.append(batch)
            cached.append(dict(depth=refined, mask=maps["mask"], K=v["K"], E=v["E"]))         # :197-201
        if run:
            densify_run(run)
        used = {v["spot"][0] for v in group if v.get("spot") is not None}
        for r_ in used:                                   # (a group's views share one resident stack)
            if ring.events[r_] is None:
                ring.events[r_] = torch.cuda.Event()
            builder.join()
            ring.events[r_].record(torch.cuda.current_stream(device))       # the stack may be overwritten once this has passed ...
            ring.checks[r_] = builder.check_async()                         # ... and what read it is final (its batches released) once this is read
        t5 = lap("densify_launch", t4)
        stage["refine"] += t4 - t3; stage["densify"] += t5 - t4

    # Groups of `views_per_launch` views, one group of lag: while the GPU runs group g's uploads and fits, the host starts
    # group g + 1; by the time it asks for group g's fits the answers are there.  (With refiner messages on: view by view, no
    # lag, so that the log keeps its order.)
    lag = 0 if verbose else 1
    inflight: deque = deque()
    group: list = []
    try:
        for k, image in enumerate(mine):
            group.append(begin(k, image))
            if len(group) >= K_LAUNCH:
                inflight.append(group); group = []
                if len(inflight) > lag:
                    finish(inflight.popleft())
        if group:
            inflight.append(group)
        while inflight:
            finish(inflight.popleft())
    finally:                                                                    # (also when a view raised: the next scan of this process finds the slots free)
        if pool:
            pool.shutdown(wait=False, cancel_futures=True)
        if feeder is not None:
            torch.cuda.synchronize(device)                                      # (no upload still reads a slot)
            feeder.close()
    say(f"-> Image processing loop finished in {time.time() - t_loop:.2f}s.")

    report = {"views": num_views, "dense_points": 0, "removed": 0, "timings": stage, "loop_detail": detail, "loop_seconds": time.time() - t_loop}
    if num_views == 0:
        say("No dense points were generated. Skipping save.")                   # :366-367
        return report
    cloud = builder.finish()
    n_before = ranks.total(len(cloud), device)
    say(f"number of dense points: {n_before}")                                  # :244-245
    if upload_events:
        torch.cuda.synchronize(device)
        report["upload_seconds_on_the_copy_stream"] = sum(a.elapsed_time(b) for a, b in upload_events) * 1e-3
    if loop_only:
        report["dense_points"] = n_before
        return report
    if n_before == 0:
        say("No dense points were generated. Skipping save.")
        return report

    say("\n--- Filtering point cloud for geometric consistency ---")            # :270
    t0 = time.time()
    if ranks.world == 1:
        votes = _votes_single(cloud, cached, config.filtering.depth_threshold)
    else:
        from .distributed import floater_votes_sharded
        fstats: dict = {}
        votes = floater_votes_sharded(cloud, cached, num_views, config.filtering.depth_threshold, stats=fstats)
        say(f"-> Sharded filter: rank 0 received {fstats['views_received']} of the other ranks' {num_views - (hi - lo)} views "
            f"({fstats['bytes_received'] / 1e6:.1f} MB of depth maps and masks; only views its points can reach travel)")
        report["filter_views_received"] = fstats["views_received"]
    plan = None
    sharded_write = ranks.world > 1 and config.processing.sharded_model_write
    if ranks.world == 1:
        kept = compact_cloud(cloud, votes, config.filtering.vote_threshold)    # :330-332
        n_kept = len(kept)
    elif sharded_write:
        # no gather at all: the ranks agree on the row range of every rank's kept points (one all-gather of per-view
        # counts), each compacts its own cloud, and each writes its own slice of the model file (below)
        from . import distributed as D
        from .filtering import kept_per_view
        plan = D.plan_fuse(kept_per_view(cloud, votes, config.filtering.vote_threshold), num_views, 1)
        kept = compact_cloud(cloud, votes, config.filtering.vote_threshold)
        n_kept = plan.total_points
    else:
        # fuse: every rank compacts its kept rows straight into its slice of the global cloud (16-byte xyz + rgba
        # records, all the model writer needs) and the slices travel to rank 0 only; rank order = view order
        from . import distributed as D
        kept, plan = D.fuse_filtered(cloud, votes, config.filtering.vote_threshold, num_views, record="xyz_rgba", dst=0)
        n_kept = plan.total_points
    removed = n_before - n_kept
    say(f"-> Filtering removed {removed} points ({removed / n_before * 100:.2f}%)")
    say(f"-> Filtering finished in {time.time() - t0:.2f}s.")
    stage["filter"] = time.time() - t0

    t0 = time.time()
    report.update(dense_points=n_kept, removed=removed, total_points=rec.num_points3D() + n_kept)
    where, layout_error = None, None
    if ranks.rank == 0:
        try:
            if ranks.world > 1:
                # The reference rescales the camera of EVERY processed view in place (:172-173) before the model is
                # written; this rank did so for its own shard only.  Same calls, same order, for the other ranks' views.
                for im in todo[hi:]:
                    pw, ph = _processing_size(config.paths.image_dir / im.name, f)
                    rec.cameras[im.camera_id].rescale(new_width=pw, new_height=ph)
            say(f"Adding {n_kept} new dense points...")
            config.paths.output_model_dir.mkdir(parents=True, exist_ok=True)
            # :355-358 + :363 -- the dense records are formatted on the GPU and streamed behind the sparse points
            # (model_writer.py); nothing of the dense cloud is materialised on the host
            # (sharded write: rank 0 lays the file out -- cameras, images, sparse points, room for every dense record --
            # and then writes its slice like everybody else)
            where = rec.write_binary(config.paths.output_model_dir, dense=None if sharded_write else kept, dense_total=n_kept)
        except Exception as e:      # noqa: BLE001  (disk full, permissions ...): the other ranks must not wait for a file that never comes
            if not sharded_write:
                raise
            layout_error = f"{type(e).__name__}: {e}"
    if sharded_write:
        import torch.distributed as dist
        from .model_writer import RECORD_BYTES, write_dense_at
        box = [(where, layout_error) if ranks.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)          # also orders the writes: the file exists at its full size from here on
        where, layout_error = box[0]
        if layout_error is not None:                    # every rank raises the same error instead of hanging in a collective
            raise RuntimeError(f"rank 0 could not lay out the model file: {layout_error}")
        own_lo, own_hi = plan.rank_rows[ranks.rank]
        # a rank writing more or fewer records than its slot would corrupt its neighbours' -- checked by all ranks TOGETHER:
        # one rank raising on its own would leave the others in the barrier below until the collective times out
        mismatch = [None] * ranks.world
        dist.all_gather_object(mismatch, None if len(kept) == own_hi - own_lo else
                               f"rank {ranks.rank} holds {len(kept)} kept points but its slice of the model file has room for {own_hi - own_lo}")
        if any(m is not None for m in mismatch):
            raise RuntimeError("; ".join(m for m in mismatch if m is not None))
        if len(kept):
            write_dense_at(config.paths.output_model_dir / "points3D.bin", where["dense_offset"] + own_lo * RECORD_BYTES,
                           kept, where["first_dense_id"] + own_lo)
        dist.barrier()
    if ranks.rank == 0:
        say(f"COLMAP binary model saved to: {config.paths.output_model_dir}")
        say(f"-> COLMAP model written in {time.time() - t0:.2f}s.")
    stage["write_model"] = time.time() - t0
    stage["total"] = time.time() - t_total
    say(f"\nTotal script execution time: {time.time() - t_total:.2f}s")
    return report
