"""One rank of the GPU rehearsal of the in-place replicated fuse (``distributed.fuse_replicated``).

Launched by tests/test_fuse_gpu.py through ``python -m torch.distributed.run``: either one rank per GPU over RCCL
(backend ``nccl``; needs >= 2 GPUs) or, on a one-GPU box, two ranks sharing ``cuda:0`` with gloo collectives
(``DD_DIST_BACKEND=gloo DD_ALLGATHERV=broadcast DD_SHARE_GPU=1`` -- gloo has no CUDA send/recv).  Every rank also
computes the WHOLE scene on its own GPU; the fused cloud must equal it bit for bit in every field.
"""

import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("DD_DIST_BACKEND", "nccl")
    local = 0 if os.environ.get("DD_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import depthdensifier_amd as dd
    from depthdensifier_amd import distributed as D
    from synth import make_views

    repeat = int(os.environ.get("DD_FUSE_REPEAT", "1"))          # (soak: the whole exchange several times in one set of processes)
    for _ in range(repeat):
        n_points = body(rank, world, backend, dev)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} [{backend}]: ok, {n_points} points" + (f", {repeat} executions" if repeat > 1 else ""))


def body(rank, world, backend, dev):
    import depthdensifier_amd as dd
    from depthdensifier_amd import distributed as D
    from synth import make_views

    V, H, W = int(os.environ.get("DD_FUSE_VIEWS", "9")), 120, 200
    d = make_views(77, V, H, W, rho=0.75, specials=True)
    d["mask"][min(3, V - 1)] = False                                          # an empty view
    params = np.tile([150.0, 152.0, 100.0, 60.0], (V, 1))
    kw = dict(mask=d["mask"], normal=d["normal"], rgb=d["rgb"])
    def stage(what):                       # (how far this rank got, should it die: tests/test_fuse_gpu.py)
        torch.cuda.synchronize()
        print(f"rank {rank}: stage {what} done", file=sys.stderr, flush=True)

    stage("start")
    full = dd.unproject_views(d["depth"], params, d["cam_from_world"], view_index=True, **kw)
    stage("whole scene, rows")
    full_packed = dd.unproject_views(d["depth"], params, d["cam_from_world"], record="xyz_rgba", pixel_index=False, **kw)
    stage("whole scene, records")

    lo, hi = D.shard_views(V, world, rank)
    cut = lambda a: a[lo:hi]
    batch = dd.ViewBatch(cut(d["depth"]), cut(params), cut(d["cam_from_world"]), mask=cut(d["mask"]), normal=cut(d["normal"]),
                         rgb=cut(d["rgb"]), view_index_base=lo, device=dev)

    def same(a, b, what):
        assert a.shape == b.shape, (what, a.shape, b.shape)
        assert torch.equal(a.view(torch.uint8) if a.dtype.is_floating_point else a, b.view(torch.uint8) if b.dtype.is_floating_point else b), what

    stage("own batch on the device")
    for chunks in (1, 3, 20):
        cloud, plan = D.fuse_replicated(batch, V, pixel_index=True, view_index=True, chunks=chunks)
        torch.cuda.synchronize()
        stage(f"rows, chunks={chunks}")
        assert plan.total_points == len(full) and torch.equal(cloud.view_offsets, full.view_offsets)
        for name in ("points", "colors", "normals", "pixel_index", "view_index"):
            same(getattr(cloud, name), getattr(full, name), f"rows, chunks={chunks}, {name}")
    cloud, plan = D.fuse_replicated(batch, V, record="xyz_rgba", chunks=2)
    stage("xyz_rgba")
    same(cloud.packed, full_packed.packed, "xyz_rgba record")
    same(cloud.points.contiguous(), full.points, "xyz_rgba points view")
    same(cloud.colors.contiguous(), full.colors, "xyz_rgba colours view")
    fault = os.environ.get("DD_FUSE_FAULT_RANK")
    if fault is not None:
        # an in-kernel scan gives up on ONE rank (fault injection, DD_LAB_FAULT_INJECT of include/ddcore_lab.h): that rank heals its rows after they went out;
        # every rank must still end up with the whole cloud (the re-exchange is agreed on collectively)
        bad = dd.ViewBatch(cut(d["depth"]), cut(params), cut(d["cam_from_world"]), mask=cut(d["mask"]), normal=cut(d["normal"]),
                           rgb=cut(d["rgb"]), view_index_base=lo, device=dev, lab=2 if rank == int(fault) else 0)
        cloud, plan = D.fuse_replicated(bad, V, pixel_index=True, view_index=True, chunks=3)
        torch.cuda.synchronize()
        for name in ("points", "colors", "normals", "pixel_index", "view_index"):
            same(getattr(cloud, name), getattr(full, name), f"healed on rank {fault}, {name}")
    p2p = os.environ.get("DD_ALLGATHERV", "p2p") == "p2p"
    if p2p and world > 1:                                                     # gather-to-owner needs send/recv (gloo: host-staged)
        owner = world - 1
        cloud, plan = D.fuse_replicated(batch, V, record="xyz_rgba", chunks=2, dst=owner)
        stage("gather-to-owner")
        if rank == owner:
            same(cloud.packed, full_packed.packed, "gather-to-owner")
        else:
            s0, s1 = plan.rank_rows[rank]
            same(cloud.packed, full_packed.packed[s0:s1], "a sender keeps just its own rows")
    # the count-only (sharded) fuse and the gather of an already compacted local cloud (what the pipeline does after the filter)
    if hi > lo:
        part = dd.unproject_views(cut(d["depth"]), cut(params), cut(d["cam_from_world"]), view_index=True, mask=cut(d["mask"]),
                                  normal=cut(d["normal"]), rgb=cut(d["rgb"]))
    else:                                   # a rank without views still takes part in every collective, with an empty cloud
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        part = dd.FusedCloud(points=z((0, 3), torch.float32), colors=z((0, 3), torch.uint8), normals=z((0, 3), torch.float32),
                             pixel_index=z((0,), torch.int32), view_index=z((0,), torch.int32), view_offsets=z((1,), torch.int64))
    if True:
        sharded = D.fuse_sharded(part, V)
        assert torch.equal(sharded.view_offsets, full.view_offsets)
        fused = D.gather_cloud(sharded)
        torch.cuda.synchronize()
        same(fused.points, full.points, "gather_cloud points")
        same(fused.colors, full.colors, "gather_cloud colours")
    return len(full)


if __name__ == "__main__":
    main()
