"""Multi-GPU fuse: one process per GPU, views sharded contiguously, RCCL over xGMI.

The reference is a single process looping over views (``scripts/test.py:131``) and fusing by
list append + ``np.concatenate`` (``:238-240, 264-266``).  Views are independent, so the
only exchange the path needs is the fuse itself:

* ``shard_views``   -- rank r owns views ``[floor(rV/R), floor((r+1)V/R))`` so that rank-order
  concatenation IS the reference's view order (global indices stay bit-exact);
* ``exchange_counts`` -- one small all-gather of per-view point counts; afterwards every rank
  knows the global ``view_offsets`` and the slot range of every rank's slice.  With only this
  step the fused cloud exists *distributed*: rank r holds global slots
  ``[rank_offsets[r], rank_offsets[r+1])`` ("sharded" fuse, no data-path collective);
* ``allgatherv_rows`` -- the all-gatherv of the per-GPU compacted clouds (replicated fuse).
  RCCL has no gatherv primitive and xGMI is point-to-point (one link per peer): every rank posts
  ONE grouped batch of sends (its slice to each peer) and receives (each peer's slice straight into
  its final rows of the pre-allocated global buffer) -- no staging copy, no padding, every link
  driven at once.

Works on any ``torch.distributed`` backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` in the
CPU tests.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous block of views owned by ``rank``."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return (rank * num_views) // world_size, ((rank + 1) * num_views) // world_size


def shard_sizes(num_views: int, world_size: int) -> list[int]:
    return [b - a for a, b in (shard_views(num_views, world_size, r) for r in range(world_size))]


def shard_views_balanced(costs: Sequence[float], world_size: int) -> list[tuple[int, int]]:
    """Contiguous split of views with unequal cost (e.g. pixel counts of mixed-resolution scenes, SURVEY.md
    8e / BASELINE config 4): cut points at the cumulative-cost quantiles, so rank order is still view
    order and no rank gets more than its fair share plus one view."""
    total = float(sum(costs))
    bounds, acc, v = [0], 0.0, 0
    n = len(costs)
    for r in range(1, world_size):
        target = total * r / world_size
        while v < n and acc + costs[v] / 2.0 <= target:
            acc += costs[v]
            v += 1
        bounds.append(v)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def exchange_counts(local_counts: torch.Tensor, num_views: int, group=None) -> torch.Tensor:
    """All-gather the per-view counts of every rank; returns (num_views,) int64 on the input's device.

    Shard sizes differ by at most one view, so the shards are padded to the largest and sent
    with one ``all_gather_into_tensor`` (a few KB).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(num_views, world)
    if local_counts.numel() != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local_counts.numel()} views, shard has {sizes[rank]}")
    if min(sizes) == max(sizes) and sizes[0] > 0:          # equal shards (weak scaling): no padding, no concat
        recv = torch.empty(world * sizes[0], dtype=torch.int64, device=local_counts.device)
        dist.all_gather_into_tensor(recv, local_counts.contiguous(), group=group)
        return recv
    width = max(max(sizes), 1)
    send = torch.zeros(width, dtype=torch.int64, device=local_counts.device)
    send[: sizes[rank]] = local_counts
    recv = torch.empty(world * width, dtype=torch.int64, device=local_counts.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, width)
    return torch.cat([recv[r, : sizes[r]] for r in range(world)])


def offsets_from_counts(counts: torch.Tensor) -> torch.Tensor:
    """(n+1,) exclusive scan with the total in the last entry."""
    out = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=counts.device)
    torch.cumsum(counts, 0, out=out[1:])
    return out


def allgatherv_rows(local: torch.Tensor, rows_per_rank: Sequence[int], out: Optional[torch.Tensor] = None,
                    group=None) -> torch.Tensor:
    """All-gatherv along dim 0: rank r contributes ``rows_per_rank[r]`` rows; every rank ends with
    the concatenation in rank order.  ``out`` (sum(rows), *local.shape[1:]) may be pre-allocated."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = [int(x) for x in rows_per_rank]
    if len(rows) != world:
        raise ValueError("rows_per_rank must have one entry per rank")
    if local.shape[0] != rows[rank]:
        raise ValueError(f"rank {rank} passes {local.shape[0]} rows, announced {rows[rank]}")
    total = sum(rows)
    if out is None:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    elif out.shape[0] != total or out.shape[1:] != local.shape[1:] or out.dtype != local.dtype:
        raise ValueError("out has the wrong shape or dtype")
    starts = [0]
    for r in range(world):
        starts.append(starts[-1] + rows[r])
    pieces = [out[starts[r]:starts[r + 1]] for r in range(world)]
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if rows[rank]:
        pieces[rank].copy_(local)
    if world == 1:
        return out
    if os.environ.get("DD_ALLGATHERV", "p2p") == "broadcast":
        work = [dist.broadcast(pieces[r], src=glob(r), group=group, async_op=True) for r in range(world) if rows[r]]
    else:
        ops = []
        for k in range(1, world):                      # peer order staggered per rank: no hot receiver
            dst, src = (rank + k) % world, (rank - k) % world
            if rows[rank]:
                ops.append(dist.P2POp(dist.isend, local, glob(dst), group))
            if rows[src]:
                ops.append(dist.P2POp(dist.irecv, pieces[src], glob(src), group))
        work = dist.batch_isend_irecv(ops) if ops else []
    for w in work:
        w.wait()
    return out


@dataclass
class ShardedCloud:
    """A fused cloud left distributed: this rank's slice plus the global index space."""

    local: "object"                    # FusedCloud of this rank's views (local slots)
    view_offsets: torch.Tensor         # (V_total+1,) int64 global slot of each view's first point
    rank_offsets: torch.Tensor         # (R+1,)       int64 global slot range of each rank
    rank: int
    world_size: int

    @property
    def total_points(self) -> int:
        return int(self.view_offsets[-1])

    @property
    def global_slots(self) -> tuple[int, int]:
        return int(self.rank_offsets[self.rank]), int(self.rank_offsets[self.rank + 1])


def fuse_sharded(local_cloud, num_views_total: int, group=None) -> ShardedCloud:
    """Exchange counts only: global ``view_offsets`` / ``rank_offsets`` for a cloud that stays sharded."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = exchange_counts(local_cloud.counts, num_views_total, group)
    view_offsets = offsets_from_counts(counts)
    bounds = [shard_views(num_views_total, world, r)[0] for r in range(world)] + [num_views_total]
    rank_offsets = view_offsets[torch.tensor(bounds, device=view_offsets.device)]
    return ShardedCloud(local_cloud, view_offsets, rank_offsets, rank, world)


def gather_cloud(sharded: ShardedCloud, group=None, out: Optional[dict] = None):
    """All-gatherv every field of a sharded cloud; returns a ``FusedCloud`` replicated on all ranks."""
    from .densify import FusedCloud

    ro = sharded.rank_offsets.cpu()
    rows = [int(ro[r + 1] - ro[r]) for r in range(sharded.world_size)]
    loc = sharded.local
    out = out or {}

    def g(name, t):
        return None if t is None else allgatherv_rows(t, rows, out.get(name), group)

    view_index = loc.view_index
    return FusedCloud(points=g("points", loc.points), colors=g("colors", loc.colors), normals=g("normals", loc.normals),
                      pixel_index=g("pixel_index", loc.pixel_index), view_index=g("view_index", view_index),
                      view_offsets=sharded.view_offsets, name=loc.name)


def allgather_views(local: torch.Tensor, num_views_total: int, group=None) -> torch.Tensor:
    """Every rank's per-view stack (depth / mask / camera blocks ..., views along dim 0, sharded with
    ``shard_views``) gathered into the full ``(num_views_total, ...)`` stack on every rank."""
    world = dist.get_world_size(group)
    return allgatherv_rows(local.contiguous(), shard_sizes(num_views_total, world), group=group)


def floater_votes_sharded(local_cloud, local_views: Sequence[dict], num_views_total: int, depth_threshold: float = 0.7,
                          group=None) -> torch.Tensor:
    """Votes of this rank's points against ALL views of a view-sharded scan.

    ``local_views``: this rank's views in order, each ``{"depth": (H,W) f32 device tensor, "mask": (H,W) bool or None,
    "K": (3,3), "E": (3,4)}``; sizes may differ between views.  Depth maps, masks and cameras are all-gathered size
    group by size group (2000 x 1080p = 16.6 GB, small next to 288 GB of HBM), then every rank votes on its own
    points: the O(N*V) work splits by points and no point moves.  Votes are counts over the same set of views, so
    they equal the one-GPU votes."""
    import numpy as np
    from .filtering import floater_votes

    world = dist.get_world_size(group)
    dev = local_cloud.points.device
    parts: list = [None] * world
    dist.all_gather_object(parts, [tuple(v["depth"].shape) for v in local_views], group=group)    # control plane
    all_shapes = [shp for part in parts for shp in part]
    if len(all_shapes) != num_views_total:
        raise ValueError(f"ranks hold {len(all_shapes)} views in total, expected {num_views_total}")
    bounds, start = [], 0
    for part in parts:
        bounds.append((start, start + len(part)))
        start += len(part)
    votes = torch.zeros(len(local_cloud), dtype=torch.int32, device=dev)
    for shp in dict.fromkeys(all_shapes):                         # distinct sizes, first-seen order (same on all ranks)
        rows = [sum(1 for k in range(lo, hi) if all_shapes[k] == shp) for lo, hi in bounds]
        mine = [v for v in local_views if tuple(v["depth"].shape) == shp]

        def stacked(make, tail, dtype):
            if mine:
                return torch.stack([make(v) for v in mine]).contiguous()
            return torch.empty((0,) + tuple(tail), dtype=dtype, device=dev)

        f64 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
        depth = allgatherv_rows(stacked(lambda v: v["depth"].to(torch.float32), shp, torch.float32), rows, group=group)
        ones = lambda v: torch.ones(shp, dtype=torch.uint8, device=dev)
        mask = allgatherv_rows(stacked(lambda v: ones(v) if v.get("mask") is None else v["mask"].view(torch.uint8) if v["mask"].dtype == torch.bool
                                       else (v["mask"] > 0).view(torch.uint8), shp, torch.uint8), rows, group=group)
        K = allgatherv_rows(stacked(lambda v: f64(v["K"])[:3, :3], (3, 3), torch.float64), rows, group=group)
        E = allgatherv_rows(stacked(lambda v: f64(v["E"])[:3, :4], (3, 4), torch.float64), rows, group=group)
        floater_votes(local_cloud.points, local_cloud.normals, depth, K.cpu().numpy(), E.cpu().numpy(), mask=mask,
                      depth_threshold=depth_threshold, votes=votes)
    return votes


def filter_floaters_sharded(local_cloud, local_depth: torch.Tensor, local_intrinsics, local_cam_from_world,
                            num_views_total: int, local_mask: Optional[torch.Tensor] = None, config=None, group=None):
    """The multi-view filter on a sharded cloud: ``floater_votes_sharded`` on this rank's stack of views, then the
    stable compaction of its own slice.  Returns ``(kept_local_cloud, votes)``."""
    import numpy as np
    from .densify import intrinsics_matrix
    from .filtering import FilteringConfig, compact_cloud

    cfg = config or FilteringConfig()
    K = intrinsics_matrix(local_intrinsics)
    E = np.asarray(local_cam_from_world.cpu() if isinstance(local_cam_from_world, torch.Tensor) else local_cam_from_world, dtype=np.float64)
    n = local_depth.shape[0]
    if K.shape[0] == 1 and n > 1:
        K = np.repeat(K, n, axis=0)
    views = [dict(depth=local_depth[i], mask=None if local_mask is None else local_mask[i], K=K[i], E=E[i]) for i in range(n)]
    votes = floater_votes_sharded(local_cloud, views, num_views_total, cfg.depth_threshold, group)
    return compact_cloud(local_cloud, votes, cfg.vote_threshold), votes
