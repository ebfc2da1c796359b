"""One rank of the rehearsal of the view-sharded floater filter (``distributed.floater_votes_sharded``): ranks share cuda:0,
gloo collectives (device send / recv staged through the host).  Every rank also computes the votes of the WHOLE scene on its
own; its shard's votes must equal that slice -- with the selective gather (each depth map only to the ranks whose points can
reach the view) and with the all-gather of rounds 1-2.  Launched by tests/test_filter.py."""

import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    import depthdensifier_amd as dd
    from depthdensifier_amd import distributed as D
    from test_filter import _scene

    layout = os.environ.get("DD_FILTER_LAYOUT", "corridor")
    V, H, W = int(os.environ.get("DD_FILTER_VIEWS", "12")), 40, 320
    d = _scene(5, V, H, W)
    rng = np.random.default_rng(3)
    E = d["cam_from_world"]
    if layout == "corridor":                       # walking along x, looking sideways: a view overlaps a few neighbours only
        for v in range(V):
            c = np.array([1.6 * v, 0.0, 0.0]); z = np.array([0.15 * np.sin(v), 0.0, 1.0]); z /= np.linalg.norm(z)
            x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x); y = np.cross(z, x)
            R = np.stack([x, y, z]); E[v, :, :3] = R; E[v, :, 3] = -R @ c
        smooth = (3.0 + 0.2 * rng.standard_normal(d["depth"].shape)).astype(np.float32)
        ordinary = np.isfinite(d["depth"]) & (d["depth"] > 0)
        d["depth"] = np.where(ordinary, smooth, d["depth"])
    depth = np.where(np.isfinite(d["depth"]), d["depth"], 0).astype(np.float32)
    K = dd.intrinsics_matrix(d["params"])
    full = dd.unproject_views(depth, d["params"], E, mask=d["mask"], normal=d["normal"], view_index=True)
    want = dd.floater_votes(full.points, full.normals, depth, K, E, mask=d["mask"])
    offs = full.view_offsets.cpu().numpy()

    lo, hi = D.shard_views(V, world, rank)
    if hi > lo:
        part = dd.unproject_views(depth[lo:hi], d["params"][lo:hi], E[lo:hi], mask=d["mask"][lo:hi], normal=d["normal"][lo:hi], view_index=True)
    else:
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        part = dd.FusedCloud(points=z((0, 3), torch.float32), colors=None, normals=z((0, 3), torch.float32), pixel_index=z((0,), torch.int32),
                             view_index=z((0,), torch.int32), view_offsets=z((1,), torch.int64))
    views = [dict(depth=torch.as_tensor(depth[v], device=dev), mask=torch.as_tensor(d["mask"][v], device=dev), K=K[v], E=E[v]) for v in range(lo, hi)]
    st_sel, st_all = {}, {}
    got_sel = D.floater_votes_sharded(part, views, V, stats=st_sel)
    got_all = D.floater_votes_sharded(part, views, V, stats=st_all, selective=False)
    torch.cuda.synchronize()
    mine = want[int(offs[lo]):int(offs[hi])]
    assert torch.equal(got_all, mine), "all-gather votes differ from the one-GPU votes"
    assert torch.equal(got_sel, mine), "selective-gather votes differ from the one-GPU votes"
    assert st_all["views_received"] == V - (hi - lo) and st_sel["views_received"] <= st_all["views_received"]
    if layout == "corridor" and V >= 4 * world and hi > lo:
        assert st_sel["views_received"] < st_all["views_received"], (st_sel, st_all)      # a corridor rank does not need the far end
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} [{layout}]: ok, {len(mine)} points, max votes {int(mine.max()) if len(mine) else 0}, "
          f"views received {st_sel['views_received']} of {st_all['views_received']} ({st_sel['bytes_received']} of {st_all['bytes_received']} bytes)")


if __name__ == "__main__":
    main()
