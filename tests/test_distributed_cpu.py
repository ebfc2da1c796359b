"""world_size-2 (and 3) gloo rehearsal of the multi-GPU fuse on CPU, plus sharding arithmetic."""

import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world, views", [(2, 7), (3, 8), (2, 1)])
def test_gloo_fuse(world, views):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "dist_worker.py"), "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--views", str(views)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert "ok" in out


def test_shard_views_partition():
    from depthdensifier_amd.distributed import shard_sizes, shard_views
    for V in (1, 7, 185, 2000):
        for R in (1, 2, 4, 8):
            spans = [shard_views(V, R, r) for r in range(R)]
            assert spans[0][0] == 0 and spans[-1][1] == V
            assert all(spans[i][1] == spans[i + 1][0] for i in range(R - 1))      # contiguous, in rank order
            sizes = shard_sizes(V, R)
            assert sum(sizes) == V and max(sizes) - min(sizes) <= 1
    assert shard_views(2000, 8, 3) == (750, 1000)
    with pytest.raises(ValueError):
        shard_views(10, 2, 2)


def test_balanced_contiguous_split():
    from depthdensifier_amd.distributed import shard_views_balanced
    costs = [1920 * 1080] * 100 + [4032 * 3024] * 20 + [640 * 480] * 300        # mixed resolutions, in view order
    for R in (1, 2, 3, 8):
        spans = shard_views_balanced(costs, R)
        assert spans[0][0] == 0 and spans[-1][1] == len(costs)
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        loads = [sum(costs[a:b]) for a, b in spans]
        assert max(loads) <= sum(costs) / R + max(costs)
    assert shard_views_balanced([1.0] * 8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert shard_views_balanced([], 2) == [(0, 0), (0, 0)]
