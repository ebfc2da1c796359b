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

* ``fuse_replicated`` -- the replicated fuse done IN PLACE (what ``scripts/test.py:238-240, 262-266`` becomes on
  N GPUs): count (``dd_count_valid``) -> ``plan_fuse`` (count all-gather, ONE device->host copy of the offsets) ->
  allocate the global cloud once -> ``dd_unproject_compact`` with ``cursor = rank_offsets[rank]`` writes every point
  at its final global row -> each chunk of views is exchanged (grouped send/recv straight from / into the final
  rows, no staging, no local copy) on RCCL's stream while the kernel of the next chunk runs on the compute stream.
  ``record="xyz_rgba"`` moves one 16-byte record per point instead of 27 bytes in three arrays.

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
    if rows[rank] and local.data_ptr() != pieces[rank].data_ptr():     # already in place when `local` IS its slice of `out`
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
    rank_rows: Optional[list] = None   # host copy of rank_offsets (R+1 ints), read once when the counts were exchanged

    @property
    def total_points(self) -> int:
        return int(self.rank_rows[-1]) if self.rank_rows is not None else int(self.view_offsets[-1])

    @property
    def global_slots(self) -> tuple[int, int]:
        ro = self.rank_rows if self.rank_rows is not None else self.rank_offsets.tolist()
        return int(ro[self.rank]), int(ro[self.rank + 1])


def fuse_sharded(local_cloud, num_views_total: int, group=None) -> ShardedCloud:
    """Exchange counts only: global ``view_offsets`` / ``rank_offsets`` for a cloud that stays sharded."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = exchange_counts(local_cloud.counts, num_views_total, group)
    view_offsets = offsets_from_counts(counts)
    bounds = [shard_views(num_views_total, world, r)[0] for r in range(world)] + [num_views_total]
    rank_offsets = view_offsets[torch.tensor(bounds, device=view_offsets.device)]
    # the ONE host read of the exchange: send / receive sizes are host arguments of the collective
    return ShardedCloud(local_cloud, view_offsets, rank_offsets, rank, world, rank_rows=rank_offsets.tolist())


def gather_cloud(sharded: ShardedCloud, group=None, out: Optional[dict] = None):
    """All-gatherv every field of a sharded cloud; returns a ``FusedCloud`` replicated on all ranks."""
    from .densify import FusedCloud

    ro = sharded.rank_rows if sharded.rank_rows is not None else sharded.rank_offsets.tolist()
    rows = [int(ro[r + 1] - ro[r]) for r in range(sharded.world_size)]
    loc = sharded.local
    out = out or {}

    def g(name, t):
        return None if t is None else allgatherv_rows(t, rows, out.get(name), group)

    view_index = loc.view_index
    return FusedCloud(points=g("points", loc.points), colors=g("colors", loc.colors), normals=g("normals", loc.normals),
                      pixel_index=g("pixel_index", loc.pixel_index), view_index=g("view_index", view_index),
                      view_offsets=sharded.view_offsets, name=loc.name)


# --------------------------------------------------------------------------------------
# replicated fuse, in place
# --------------------------------------------------------------------------------------

@dataclass
class FusePlan:
    """What every rank knows once the per-view counts have been exchanged (host-side numbers)."""

    view_offsets: torch.Tensor            # (V_total+1,) int64 device: global row of every view's first point
    offsets_host: list                    # the same on the host
    rank: int
    world_size: int
    num_views_total: int
    chunk_views: list                     # [chunk] -> (lo, hi) LOCAL view range of this rank's chunk
    chunk_rows: list                      # [chunk][rank] -> (row_lo, row_hi) global rows of that rank's chunk

    @property
    def total_points(self) -> int:
        return int(self.offsets_host[-1])

    @property
    def rank_rows(self) -> list:
        """[rank] -> (row_lo, row_hi) of each rank's whole slice."""
        return [(c[0][0], c[-1][1]) for c in zip(*self.chunk_rows)] if self.chunk_rows else []


def _chunk_bounds(lo: int, hi: int, chunks: int) -> list:
    n = hi - lo
    return [(lo + (k * n) // chunks, lo + ((k + 1) * n) // chunks) for k in range(chunks)]


def plan_fuse(local_counts: torch.Tensor, num_views_total: int, chunks: int = 1, group=None) -> FusePlan:
    """All-gather the per-view counts and lay the global cloud out: views in order, rank r's shard split into
    ``chunks`` contiguous pieces whose k-th pieces are exchanged together.  One device->host copy (V_total+1
    int64) -- the sizes of the sends and receives are host arguments."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if chunks < 1:
        raise ValueError("chunks must be >= 1")
    counts = exchange_counts(local_counts, num_views_total, group)
    view_offsets = offsets_from_counts(counts)
    host = view_offsets.tolist()
    chunk_views, chunk_rows = layout_chunks(host, world, rank, chunks)
    return FusePlan(view_offsets, host, rank, world, num_views_total, chunk_views, chunk_rows)


def layout_chunks(offsets_host: Sequence[int], world: int, rank: int, chunks: int):
    """Pure layout of the fused cloud from the global view offsets (``len = V_total + 1``): rank r's contiguous shard of
    views is cut into ``chunks`` contiguous pieces; returns ``(chunk_views, chunk_rows)`` -- this rank's LOCAL view range
    per chunk and, per chunk, every rank's global row range.  Rows of (rank, chunk) cells tile ``[0, total)`` in
    (rank, chunk) order, i.e. in view order."""
    num_views_total = len(offsets_host) - 1
    shards = [shard_views(num_views_total, world, r) for r in range(world)]
    per_rank = [_chunk_bounds(lo, hi, chunks) for lo, hi in shards]
    chunk_rows = [[(offsets_host[per_rank[r][k][0]], offsets_host[per_rank[r][k][1]]) for r in range(world)] for k in range(chunks)]
    lo0 = shards[rank][0]
    chunk_views = [(a - lo0, b - lo0) for a, b in per_rank[rank]]
    return chunk_views, chunk_rows


def exchange_rows(tensors: Sequence[torch.Tensor], rows: Sequence[tuple], group=None, dst: Optional[int] = None, base: int = 0) -> list:
    """In-place all-gatherv of row ranges of pre-allocated GLOBAL tensors: rank r owns rows ``rows[r] = (lo, hi)`` of
    every tensor (already written), and receives every other rank's rows straight into place.  ONE grouped batch of
    sends and receives (``ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd`` on RCCL: one xGMI link per peer, all
    driven at once); nothing is staged or copied locally.  ``dst``: only that rank receives (gather-to-owner); a pure
    sender may then hold just its own rows, ``base`` = the global row of its tensors' row 0.
    Returns the outstanding work handles (``wait_all`` them before reading the rows on another stream)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if len(rows) != world:
        raise ValueError("rows must have one (lo, hi) per rank")
    if world == 1:
        return []
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    mine = rows[rank]
    if base and (dst is None or rank == dst):
        raise ValueError("base is for pure senders of a gather-to-owner exchange")
    if os.environ.get("DD_ALLGATHERV", "p2p") == "broadcast" and dst is None:
        return [dist.broadcast(t[rows[r][0]:rows[r][1]], src=glob(r), group=group, async_op=True)
                for r in range(world) if rows[r][1] > rows[r][0] for t in tensors]
    # backends without device send / recv (gloo: CPU rehearsals, ranks sharing one GPU) run the SAME schedule -- same peers, same
    # row ranges, same order -- staged through host memory; the received rows are copied into place before this returns
    staged = dist.get_backend(group) != "nccl" and any(t.is_cuda for t in tensors)
    ops, landed, keep = [], [], []
    for k in range(1, world):                          # peer order staggered per rank: no hot receiver
        to, frm = (rank + k) % world, (rank - k) % world
        for t in tensors:
            if mine[1] > mine[0] and (dst is None or to == dst):
                out = t[mine[0] - base:mine[1] - base]
                if staged:
                    out = out.cpu()
                    keep.append(out)
                ops.append(dist.P2POp(dist.isend, out, glob(to), group))
            if rows[frm][1] > rows[frm][0] and (dst is None or rank == dst):
                into = t[rows[frm][0]:rows[frm][1]]
                if staged:
                    buf = torch.empty(into.shape, dtype=into.dtype)
                    landed.append((into, buf))
                    into = buf
                ops.append(dist.P2POp(dist.irecv, into, glob(frm), group))
    work = list(dist.batch_isend_irecv(ops)) if ops else []
    if staged:
        wait_all(work)
        for into, buf in landed:
            into.copy_(buf)
        return []
    return work


def wait_all(work: Sequence) -> None:
    for w in work:
        w.wait()


def fuse_replicated(batch, num_views_total: int, *, normals: bool = True, colors: bool = True, pixel_index: bool = False,
                    view_index: bool = False, record: str = "rows", chunks: int = 4, group=None, dst: Optional[int] = None,
                    buffers: Optional[dict] = None, counts: Optional[torch.Tensor] = None, plan: Optional[FusePlan] = None):
    """Densify this rank's shard of views and fuse all shards into ONE cloud held by every rank (or by ``dst``):
    ``scripts/test.py:203-244`` per view and the fuse of ``:238-240, 262-266`` across GPUs.

    ``batch``: the ``ViewBatch`` of this rank's views (``shard_views`` order; ``view_index_base`` = its first global
    view).  Every point is written exactly once, by the kernel, at its final global row; the exchange of chunk k
    overlaps the kernel of chunk k+1.  ``record``: ``"rows"`` = xyz / normals / colours arrays (27 B per point with all
    three), ``"xyz_rgba"`` = one 16-byte record (what the model writer needs).  ``buffers``: global tensors to reuse
    (keys as ``CloudBuilder.FIELDS``).  Returns ``(FusedCloud, FusePlan)``; rows of other ranks are valid on the
    current stream when this returns (on ranks that receive)."""
    from .densify import CloudBuilder, FusedCloud, count_valid

    if record not in ("rows", "xyz_rgba"):
        raise ValueError("record must be 'rows' or 'xyz_rgba'")
    if plan is None:
        plan = plan_fuse(count_valid(batch) if counts is None else counts, num_views_total, chunks, group)
    total = plan.total_points
    receives = dst is None or plan.rank == dst
    own_lo, own_hi = plan.rank_rows[plan.rank]
    base = 0 if receives else own_lo                                   # a pure sender holds only its own rows
    cap = total if receives else own_hi - own_lo
    rows = record == "rows"
    # Where the global arrays live.  RCCL's send / recv copy between the user's buffer and RCCL's own (IPC-shared) FIFO buffers with
    # kernels of the local GPU, so a user buffer needs no export of its own -- arrays placed by the zone arena (virtual-memory API,
    # exportable as shareable handles but not through hipIpcGetMemHandle: profiles/r05_ubench_vmm_export.txt) should do.  That could
    # not be run on this build's one-GPU boxes (RCCL refuses two ranks on one device), so the default stays the plain allocation that
    # is known to work; DD_FUSE_PLACEMENT=probed places them (tools/run_multi_gpu.sh times both on the first multi-GPU box).
    builder = CloudBuilder(cap, points=rows, normals=normals and rows, colors=colors and rows, pixel_index=pixel_index,
                           view_index=view_index, packed=not rows, buffers=buffers, start=own_lo - base, device=batch.device,
                           placement=os.environ.get("DD_FUSE_PLACEMENT", "first"))
    # the counts are KNOWN here (plan_fuse): nothing to guess -- and a guess that missed would scatter rows beyond this rank's
    # [own_lo, own_hi) into regions of the shared buffers that RCCL is receiving the peers' rows into
    builder.speculate_dense = False
    builder.overlap_small = False           # every chunk's rows are handed to RCCL right behind its kernel: the caller's stream, no side streams
    moved = [t for t in (builder.xyz, builder.normal, builder.rgb, builder.pix, builder.view, builder.packed) if t is not None]
    work = []
    for (lo, hi), ranges in zip(plan.chunk_views, plan.chunk_rows):
        builder.append(batch.slice(lo, hi))                             # kernel of this chunk on the current stream ...
        work += exchange_rows(moved, ranges, group, dst, base)          # ... its exchange waits for it, then runs on RCCL's stream
    wait_all(work)
    healed_before = builder.healed
    end = builder.check()
    if end != own_hi - base:
        raise RuntimeError(f"rank {plan.rank}: kernel wrote up to row {end + base}, plan says {own_hi}")
    # check() may have healed THIS rank's rows (an in-kernel scan gave up; the batches were redone two-pass, CloudBuilder._heal)
    # -- after they went out.  The peers hold the rows of the failed attempt and cannot know: agree on it, and if any rank
    # healed, every rank sends its (now final) rows of every chunk once more.
    if dist.get_world_size(group) > 1:
        flag = torch.tensor([1 if builder.healed != healed_before else 0], dtype=torch.int32, device=batch.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()):
            work = []
            for ranges in plan.chunk_rows:
                work += exchange_rows(moved, ranges, group, dst, base)
            wait_all(work)
    cut = lambda t: None if t is None else t[:cap]
    if rows:
        cloud = FusedCloud(points=cut(builder.xyz), colors=cut(builder.rgb), normals=cut(builder.normal), pixel_index=cut(builder.pix),
                           view_index=cut(builder.view), view_offsets=plan.view_offsets)
    else:
        cloud = FusedCloud.from_packed(cut(builder.packed), plan.view_offsets, pixel_index=cut(builder.pix), view_index=cut(builder.view))
    return cloud, plan


def fuse_filtered(local_cloud, votes: torch.Tensor, vote_threshold: int, num_views_total: int, *, record: str = "xyz_rgba",
                  dst: Optional[int] = None, group=None):
    """Fuse the per-GPU clouds AFTER the floater filter (``scripts/test.py:330-332`` then ``:355-358`` needs points and
    colours in one place): every rank announces its kept rows per view, the global cloud is laid out, each rank
    compacts its kept rows straight into its slice (``dd_compact_cloud`` -> final rows, no intermediate local cloud) and
    the slices are exchanged in place.  ``record="xyz_rgba"``: 16 bytes per point on the wire; ``"rows"``: points, colours
    and normals arrays.  ``dst``: only that rank receives.  Returns ``(FusedCloud or None on a pure sender, FusePlan)``."""
    from .densify import CloudBuilder, FusedCloud
    from .filtering import compact_into, kept_per_view

    if os.environ.get("DD_ALLGATHERV", "p2p") == "broadcast":
        dst = None                                           # rehearsal backends without send/recv replicate
    plan = plan_fuse(kept_per_view(local_cloud, votes, vote_threshold), num_views_total, 1, group)
    dev = local_cloud.points.device
    receives = dst is None or plan.rank == dst
    own_lo, own_hi = plan.rank_rows[plan.rank]
    base = 0 if receives else own_lo
    cap = plan.total_points if receives else own_hi - own_lo
    names = ["packed"] if record == "xyz_rgba" else [k for k, t in (("points", local_cloud.points), ("colors", local_cloud.colors),
                                                                    ("normals", local_cloud.normals)) if t is not None]
    bufs = {k: torch.empty((max(cap, 1),) + CloudBuilder.FIELDS[k][0], dtype=CloudBuilder.FIELDS[k][1], device=dev) for k in names}
    compact_into(local_cloud, votes, vote_threshold, bufs, own_lo - base, own_hi - own_lo)
    wait_all(exchange_rows([bufs[k] for k in names], plan.chunk_rows[0], group, dst, base))
    if not receives:
        return None, plan
    if record == "xyz_rgba":
        return FusedCloud.from_packed(bufs["packed"][:cap], plan.view_offsets, local_cloud.name), plan
    return FusedCloud(points=bufs["points"][:cap], colors=None if "colors" not in bufs else bufs["colors"][:cap],
                      normals=None if "normals" not in bufs else bufs["normals"][:cap], pixel_index=None, view_index=None,
                      view_offsets=plan.view_offsets, name=local_cloud.name), plan


def allgather_views(local: torch.Tensor, num_views_total: int, group=None) -> torch.Tensor:
    """Every rank's per-view stack (depth / mask / camera blocks ..., views along dim 0, sharded with
    ``shard_views``) gathered into the full ``(num_views_total, ...)`` stack on every rank."""
    world = dist.get_world_size(group)
    return allgatherv_rows(local.contiguous(), shard_sizes(num_views_total, world), group=group)


def views_in_reach(points: torch.Tensor, K: "np.ndarray", E: "np.ndarray", sizes: Sequence[tuple], chunk: int = 65536) -> torch.Tensor:
    """(V,) bool on the points' device: views whose image at least one of ``points`` COULD project into, in front of the
    camera -- conservative (bounding spheres of runs of ``chunk`` consecutive points against the five half-spaces "in front of
    the camera, inside the image", the same linear forms as the vote kernel's culling, ``csrc/ddfilter.hip``): a view outside
    the result cannot vote on any of the points (``scripts/test.py:297-312`` never reaches the depth lookup), so leaving it
    out of ``floater_votes`` changes no vote.  Points that are not finite put every view in reach."""
    import numpy as np
    V = len(sizes)
    dev = points.device
    n = points.shape[0]
    if n == 0 or V == 0:
        return torch.zeros(V, dtype=torch.bool, device=dev)
    # per-run bounding boxes on the points as they are (min / max are exact in any dtype): no float64 copy of the cloud, no padded
    # second copy -- at 2 G points per rank those were ~100 GB of transients.  The ragged tail is a run of its own.
    whole = (n // chunk) * chunk
    parts_lo, parts_hi = [], []
    if whole:
        body = points[:whole].view(-1, chunk, 3)
        parts_lo.append(body.amin(dim=1)); parts_hi.append(body.amax(dim=1))
    if n > whole:
        tail = points[whole:]
        parts_lo.append(tail.amin(dim=0, keepdim=True)); parts_hi.append(tail.amax(dim=0, keepdim=True))
    lo, hi = torch.cat(parts_lo).to(torch.float64), torch.cat(parts_hi).to(torch.float64)    # (S,3)
    if not bool((torch.isfinite(lo) & torch.isfinite(hi)).all()):          # amin / amax propagate NaN; an infinity shows in one of them
        return torch.ones(V, dtype=torch.bool, device=dev)
    c = 0.5 * (lo + hi)                                                     # (S,3) sphere centres
    r = 0.5 * torch.linalg.vector_norm(hi - lo, dim=1) * (1.0 + 1e-6) + 1e-9     # (S,) half diagonals: every point is inside
    Kt = torch.as_tensor(np.asarray(K, dtype=np.float64), device=dev)       # (V,3,3)
    Et = torch.as_tensor(np.asarray(E, dtype=np.float64)[:, :3, :4], device=dev)      # (V,3,4)
    M = Kt[:, :2, :] @ Et                                                   # (V,2,4): rows nu, nw of K [R|t]
    Z = Et[:, 2, :]                                                         # (V,4): depth in the camera frame
    W = torch.as_tensor([float(sz[1]) for sz in sizes], dtype=torch.float64, device=dev)[:, None]
    H = torch.as_tensor([float(sz[0]) for sz in sizes], dtype=torch.float64, device=dev)[:, None]
    planes = torch.stack([Z, M[:, 0], W * Z - M[:, 0], M[:, 1], H * Z - M[:, 1]], dim=1)      # (V,5,4): f(p) = a.p + b must be > 0 (>= 0)
    csum = (1.0 + c.abs().sum(dim=1))[None, None, :]
    out = torch.zeros(V, dtype=torch.bool, device=dev)
    for v0 in range(0, V, 128):                                             # blocks of views: (128, 5, S) float64 temporaries
        a, b = planes[v0:v0 + 128, :, :3], planes[v0:v0 + 128, :, 3]        # (v,5,3), (v,5)
        f = torch.einsum("vki,si->vks", a, c) + b[..., None]                # value at the sphere centres
        reach = f + torch.linalg.vector_norm(a, dim=2)[..., None] * r[None, None, :]      # the largest value inside the sphere
        # slack: 1e-6 relative on every term (the float64 rounding of either formulation is 1e-16), plus the W * 1e-8 of den = zc + 1e-8
        slack = 1e-6 * (a.abs().sum(dim=2) + b.abs())[..., None] * csum + 1e-7 * torch.maximum(W, H)[v0:v0 + 128, :, None]
        out[v0:v0 + 128] = (reach + slack > 0).all(dim=1).any(dim=1)        # some sphere touches all five half-spaces
    return out


def _exchange_selected(local_stack: torch.Tensor, own_ids: Sequence[int], need_by_rank: Sequence[Sequence[int]], owner_of, group=None) -> torch.Tensor:
    """Rows of a view-sharded stack sent only where they are needed: this rank holds the rows of ``own_ids`` (ascending global
    view ids) in ``local_stack``; ``need_by_rank[r]`` (ascending) is what rank r wants.  Returns this rank's wanted rows in
    that order.  One grouped batch of sends / receives (xGMI is point to point); backends without device send / recv
    (gloo rehearsals) stage through host memory."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    want = list(need_by_rank[rank])
    out = torch.empty((len(want),) + tuple(local_stack.shape[1:]), dtype=local_stack.dtype, device=local_stack.device)
    pos_own = {v: i for i, v in enumerate(own_ids)}
    pos_out = {v: i for i, v in enumerate(want)}
    mine = [v for v in want if owner_of(v) == rank]
    if mine:
        out[[pos_out[v] for v in mine]] = local_stack[[pos_own[v] for v in mine]]
    if world == 1:
        return out
    host = dist.get_backend(group) != "nccl"
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    ops, landed, keep = [], [], []
    for k in range(1, world):                              # peer order staggered per rank
        to, frm = (rank + k) % world, (rank - k) % world
        send_ids = [v for v in need_by_rank[to] if owner_of(v) == rank]
        if send_ids:
            rows = local_stack[[pos_own[v] for v in send_ids]].contiguous()
            rows = rows.cpu() if host else rows
            keep.append(rows)
            ops.append(dist.P2POp(dist.isend, rows, glob(to), group))
        recv_ids = [v for v in want if owner_of(v) == frm]
        if recv_ids:                                       # ownership is contiguous in view order: one slice of `out`
            a, b_ = pos_out[recv_ids[0]], pos_out[recv_ids[-1]] + 1
            if host:
                buf = torch.empty((b_ - a,) + tuple(out.shape[1:]), dtype=out.dtype)
                landed.append((buf, a, b_))
                ops.append(dist.P2POp(dist.irecv, buf, glob(frm), group))
            else:
                ops.append(dist.P2POp(dist.irecv, out[a:b_], glob(frm), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    for buf, a, b_ in landed:
        out[a:b_].copy_(buf)
    return out


def floater_votes_sharded(local_cloud, local_views: Sequence[dict], num_views_total: int, depth_threshold: float = 0.7,
                          group=None, stats: Optional[dict] = None, selective: Optional[bool] = None) -> torch.Tensor:
    """Votes of this rank's points against ALL views of a view-sharded scan.

    ``local_views``: this rank's views in order, each ``{"depth": (H,W) f32 device tensor, "mask": (H,W) bool or None,
    "K": (3,3), "E": (3,4)}``; sizes may differ between views.  The cameras of all views are all-gathered (168 bytes per
    view); every rank then works out which views its own points can reach at all (``views_in_reach``), the ranks exchange
    these lists, and each depth map / mask travels only to the ranks that asked for it (round 3; an inward-facing ring
    degrades to the all-gather of rounds 1-2, a corridor scan moves a few views per rank instead of 2000 x 1080p = 20 GB).
    Every rank votes on its own points: the O(N*V) work splits by points and no point moves.  A view outside a rank's list
    cannot vote on its points, so the votes equal the one-GPU votes.  ``selective=False`` (or ``DD_FILTER_GATHER=all``)
    forces the all-gather.  ``stats`` receives ``views_total``, ``views_received``, ``bytes_received``."""
    import numpy as np
    from .filtering import floater_votes

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local_cloud.points.device
    if selective is None:
        selective = os.environ.get("DD_FILTER_GATHER", "needed") != "all"
    f64 = lambda a: np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    parts: list = [None] * world                                   # control plane: sizes and cameras of every view
    dist.all_gather_object(parts, [(tuple(v["depth"].shape), f64(v["K"])[:3, :3], f64(v["E"])[:3, :4]) for v in local_views], group=group)
    flat = [x for part in parts for x in part]
    if len(flat) != num_views_total:
        raise ValueError(f"ranks hold {len(flat)} views in total, expected {num_views_total}")
    all_shapes = [x[0] for x in flat]
    K_all = np.stack([x[1] for x in flat]) if flat else np.zeros((0, 3, 3))
    E_all = np.stack([x[2] for x in flat]) if flat else np.zeros((0, 3, 4))
    bounds, start = [], 0
    for part in parts:
        bounds.append((start, start + len(part)))
        start += len(part)
    owner_of = lambda v: next(r for r, (lo, hi) in enumerate(bounds) if lo <= v < hi)
    my_lo = bounds[rank][0]
    if selective:
        reach = views_in_reach(local_cloud.points, K_all, E_all, all_shapes).cpu().numpy()
        needs: list = [None] * world
        dist.all_gather_object(needs, np.nonzero(reach)[0].tolist(), group=group)
    else:
        needs = [list(range(num_views_total))] * world
    votes = torch.zeros(len(local_cloud), dtype=torch.int32, device=dev)
    received = received_bytes = 0
    for shp in dict.fromkeys(all_shapes):                         # distinct sizes, first-seen order (same on all ranks)
        group_ids = [k for k in range(num_views_total) if all_shapes[k] == shp]
        in_group = set(group_ids)
        need_by_rank = [[v for v in needs[r] if v in in_group] for r in range(world)]
        own_ids = [k for k in group_ids if bounds[rank][0] <= k < bounds[rank][1]]
        mine = [local_views[k - my_lo] for k in own_ids]

        def stacked(make, tail, dtype):
            if mine:
                return torch.stack([make(v) for v in mine]).contiguous()
            return torch.empty((0,) + tuple(tail), dtype=dtype, device=dev)

        ones = lambda v: torch.ones(shp, dtype=torch.uint8, device=dev)
        d_loc = stacked(lambda v: v["depth"].to(torch.float32), shp, torch.float32)
        m_loc = stacked(lambda v: ones(v) if v.get("mask") is None else v["mask"].view(torch.uint8) if v["mask"].dtype == torch.bool
                        else (v["mask"] > 0).view(torch.uint8), shp, torch.uint8)
        if selective:
            depth = _exchange_selected(d_loc, own_ids, need_by_rank, owner_of, group)
            mask = _exchange_selected(m_loc, own_ids, need_by_rank, owner_of, group)
            ids = need_by_rank[rank]
        else:
            rows = [sum(1 for k in group_ids if lo <= k < hi) for lo, hi in bounds]
            depth = allgatherv_rows(d_loc, rows, group=group)
            mask = allgatherv_rows(m_loc, rows, group=group)
            ids = group_ids
        got = [v for v in ids if owner_of(v) != rank]
        received += len(got)
        received_bytes += len(got) * int(np.prod(shp)) * 5
        if ids and len(local_cloud):
            floater_votes(local_cloud.points, local_cloud.normals, depth, K_all[ids], E_all[ids], mask=mask,
                          depth_threshold=depth_threshold, votes=votes)
    if stats is not None:
        stats.update(views_total=num_views_total, views_received=received, bytes_received=received_bytes, selective=bool(selective))
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
