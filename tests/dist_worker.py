"""One rank of the world_size-N CPU (gloo) rehearsal of the multi-GPU fuse.

Launched by tests/test_distributed_cpu.py.  The per-rank cloud is produced by the ORACLE
(test infrastructure) so that the exchange logic of depthdensifier_amd.distributed -- view
sharding, count exchange, global offsets, all-gatherv -- is exercised without a GPU.
"""

import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--views", type=int, default=7)
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port))
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)

    from depthdensifier_amd import distributed as D
    from depthdensifier_amd.densify import FusedCloud
    from oracle import densify_oracle as orc
    from synth import make_views

    V, H, W = a.views, 24, 40
    d = make_views(42, V, H, W, rho=0.7, specials=True)
    if V > 2:
        d["mask"][2] = False                                # an empty view in the middle
    params = np.tile([30.0, 31.0, 20.0, 12.0], (V, 1))
    full = orc.densify_scene_script(d["depth"], params, d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"])

    lo, hi = D.shard_views(V, a.world, a.rank)
    if hi > lo:
        part = orc.densify_scene_script(d["depth"][lo:hi], params[lo:hi], d["cam_from_world"][lo:hi], mask=d["mask"][lo:hi],
                                        normal=d["normal"][lo:hi], rgb=d["rgb"][lo:hi])
        vi = part.view_index + lo
    else:
        part = orc.fuse_views([])
        part.normals = np.zeros((0, 3), np.float32); part.colors = np.zeros((0, 3), np.uint8)
        vi = np.zeros((0,), np.int64)
    local = FusedCloud(points=torch.from_numpy(part.points.astype(np.float32)), colors=torch.from_numpy(part.colors),
                       normals=torch.from_numpy(part.normals), pixel_index=torch.from_numpy(part.pixel_index.astype(np.int32)),
                       view_index=torch.from_numpy(vi.astype(np.int32)), view_offsets=torch.from_numpy(part.view_offsets))

    sharded = D.fuse_sharded(local, V)
    assert np.array_equal(sharded.view_offsets.numpy(), full.view_offsets), "global view offsets"
    s0, s1 = sharded.global_slots
    assert s1 - s0 == len(local)
    assert np.array_equal(full.pixel_index[s0:s1], part.pixel_index), "rank slice sits at its global slots"
    assert sharded.total_points == len(full.points)

    fused = D.gather_cloud(sharded)
    assert np.array_equal(fused.points.numpy(), full.points.astype(np.float32), equal_nan=True)
    assert np.array_equal(fused.colors.numpy(), full.colors)
    assert np.array_equal(fused.normals.numpy(), full.normals)
    assert np.array_equal(fused.pixel_index.numpy(), full.pixel_index.astype(np.int32))
    assert np.array_equal(fused.view_index.numpy(), full.view_index.astype(np.int32))
    # ---- the replicated fuse IN PLACE: plan from the counts, own rows already at their final global rows, chunked
    # grouped exchange straight into place (the kernel is played by the oracle part; GPU twin: tests/test_fuse_gpu.py)
    want = {"points": full.points.astype(np.float32), "colors": full.colors, "normals": full.normals,
            "pixel_index": full.pixel_index.astype(np.int32)}
    mine = {"points": local.points, "colors": local.colors, "normals": local.normals, "pixel_index": local.pixel_index}
    for chunks in (1, 3):
        plan = D.plan_fuse(local.counts, V, chunks=chunks)
        assert plan.offsets_host == full.view_offsets.tolist() and plan.total_points == len(full.points)
        assert plan.rank_rows[a.rank] == (s0, s1)
        assert [r for c in plan.chunk_views for r in c][0] == 0 and plan.chunk_views[-1][1] == hi - lo
        N = plan.total_points
        glob = {k: torch.zeros((N,) + tuple(v.shape[1:]), dtype=v.dtype) for k, v in mine.items()}
        for k, v in mine.items():
            glob[k][s0:s1] = v                                   # "the kernel wrote its rows at rank_offsets[rank]"
        ptrs = {k: t.data_ptr() for k, t in glob.items()}
        work = []
        for ranges in plan.chunk_rows:
            assert ranges[a.rank][0] >= s0 and ranges[a.rank][1] <= s1
            work += D.exchange_rows(list(glob.values()), ranges)
        D.wait_all(work)
        for k, t in glob.items():
            assert t.data_ptr() == ptrs[k]
            assert np.array_equal(t.numpy(), want[k], equal_nan=t.dtype.is_floating_point), f"in-place fuse, chunks={chunks}, field {k}"
    # gather-to-owner: only the last rank receives; the others hold just their own rows (base = their first global row)
    owner = a.world - 1
    plan = D.plan_fuse(local.counts, V, chunks=2)
    if a.rank == owner:
        glob = {k: torch.zeros((plan.total_points,) + tuple(v.shape[1:]), dtype=v.dtype) for k, v in mine.items()}
        for k, v in mine.items():
            glob[k][s0:s1] = v
        tens, base = list(glob.values()), 0
    else:
        tens, base = [v.contiguous() for v in mine.values()], s0
    work = []
    for ranges in plan.chunk_rows:
        work += D.exchange_rows(tens, ranges, dst=owner, base=base)
    D.wait_all(work)
    if a.rank == owner:
        for k, t in glob.items():
            assert np.array_equal(t.numpy(), want[k], equal_nan=t.dtype.is_floating_point), f"gather-to-owner, field {k}"
    # per-view stacks (what the sharded filter all-gathers): every rank ends with all views, in view order
    depth_all = D.allgather_views(torch.from_numpy(d["depth"][lo:hi].astype(np.float32)), V)
    assert np.array_equal(depth_all.numpy(), d["depth"].astype(np.float32), equal_nan=True)
    E_all = D.allgather_views(torch.from_numpy(d["cam_from_world"][lo:hi]), V)
    assert np.array_equal(E_all.numpy(), d["cam_from_world"])
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {a.rank}/{a.world}: ok, {len(local)} local of {len(full.points)} points")


if __name__ == "__main__":
    main()
