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


def test_fuse_layout_tiles_the_cloud_in_view_order():
    """layout_chunks (what plan_fuse derives from the exchanged counts): for every world size / chunk count the
    (rank, chunk) cells are contiguous, in view order, cover [0, total) exactly, agree between ranks, and a rank's local
    view ranges cover its shard."""
    import numpy as np
    from depthdensifier_amd.distributed import layout_chunks, shard_views
    rng = np.random.default_rng(0)
    for V in (0, 1, 7, 185, 2000):
        counts = rng.integers(0, 5000, V)
        if V > 3:
            counts[2] = 0
        offs = [0] + np.cumsum(counts).tolist()
        for R in (1, 2, 3, 8):
            for C in (1, 3, 5, 20):
                views0, rows0 = layout_chunks(offs, R, 0, C)
                for r in range(R):
                    views, rows = layout_chunks(offs, R, r, C)
                    assert rows == rows0 and len(rows) == C and all(len(c) == R for c in rows)
                    lo, hi = shard_views(V, R, r)
                    assert views[0][0] == 0 and views[-1][1] == hi - lo and all(a[1] == b[0] for a, b in zip(views, views[1:]))
                    for k, (a, b) in enumerate(views):               # the chunk's rows are the rows of its views
                        assert rows[k][r] == (offs[lo + a], offs[lo + b])
                cells = [rows0[k][r] for r in range(R) for k in range(C)]          # rank-major, chunk-minor = view order
                assert cells[0][0] == 0 and cells[-1][1] == offs[-1]
                assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(cells, cells[1:]))


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


def test_batch_assignment_is_balanced_and_deterministic(tmp_path):
    from depthdensifier_amd.batch import assign_scans, discover_scans
    sizes = {"a": 9, "b": 1, "c": 5, "d": 4, "e": 0, "f": 3}
    for name, n in sizes.items():
        (tmp_path / name / "images").mkdir(parents=True)
        if name != "e":
            (tmp_path / name / "sparse" / "0").mkdir(parents=True)       # "e" is incomplete
        for k in range(n):
            (tmp_path / name / "images" / f"{k}.png").write_bytes(b"")
    jobs = list(discover_scans(tmp_path, tmp_path / "out"))
    owner = assign_scans(jobs, 2)
    assert owner == assign_scans(jobs, 2) and set(owner) == {0, 1}
    load = [sum(sizes[j.name] for j, o in zip(jobs, owner) if o == r and j.complete) for r in range(2)]
    assert abs(load[0] - load[1]) <= 2, load                             # 9+1+.. vs 5+4+3: longest-first deals evenly
    assert assign_scans(jobs, 1) == [0] * len(jobs)
    # resolution counts: a 2-image scan of 4000x3000 pictures outweighs a 9-image scan of 640x480 ones
    from depthdensifier_amd.colmap_io import Camera, Reconstruction
    import numpy as np
    for name, (w, h) in {"a": (640, 480), "b": (4000, 3000)}.items():
        rec = Reconstruction()
        rec.cameras[1] = Camera(1, 1, w, h, np.array([1.0, 1.0, w / 2, h / 2]))
        rec.write_binary(tmp_path / name / "sparse" / "0")
    (tmp_path / "b" / "images" / "1.png").write_bytes(b"")                       # b now has 2 images
    jobs = [j for j in discover_scans(tmp_path, tmp_path / "out") if j.name in ("a", "b")]
    owner = assign_scans(jobs, 2)
    assert owner[[j.name for j in jobs].index("b")] == 0                          # dealt first -> rank 0


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_batch(tmp_path, world):
    """Every complete scan runs exactly once, on the rank the assignment names; every rank ends with the same
    merged report in folder order; a failing scan is FAILED without stopping its rank."""
    import json
    root, out = tmp_path / "scans", tmp_path / "out"
    out.mkdir()
    names = ["alpha", "beta", "broken", "delta", "empty", "gamma"]
    for i, name in enumerate(names):
        (root / name / "images").mkdir(parents=True)
        if name != "empty":
            (root / name / "sparse" / "0").mkdir(parents=True)
        for k in range(i + 1):
            (root / name / "images" / f"{k}.png").write_bytes(b"")
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "batch_worker.py"), str(r), str(world), str(port), str(root), str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
    from depthdensifier_amd.batch import assign_scans, discover_scans
    jobs = list(discover_scans(root, out))
    owner = dict(zip((j.name for j in jobs), assign_scans(jobs, world)))
    for name in ("alpha", "beta", "delta", "gamma"):
        assert int((out / f"{name}.ran_on").read_text()) == owner[name]
    assert not (out / "empty.ran_on").exists() and not (out / "broken.ran_on").exists()
    reports = [json.loads((out / f"report.rank{r}.json").read_text()) for r in range(world)]
    assert all(rep == reports[0] for rep in reports)
    assert [n for n, _ in reports[0]] == ["alpha", "beta", "broken", "delta", "gamma"]
    assert dict(map(tuple, reports[0]))["broken"] == "FAILED"
    assert "Batch Processing Time Report" in outs[0] and all("Batch Processing Time Report" not in o for o in outs[1:])


def test_c_abi_allgatherv_schedule_on_a_fake_rccl(tmp_path):
    """``dd_allgatherv`` at world 2, 3 and 8 without a GPU: ``tests/c_client/fake_rccl.cpp`` exports the five RCCL entry points
    itself (libddcore.so resolves them from the running process), plays every rank on host buffers, matches the logged
    sends with the logged receives the way NCCL does and moves the bytes: every receiver ends with the whole cloud, a pure
    sender's base offset is honoured, nothing is sent twice or left unmatched -- replicate and gather-to-owner, all six
    fields, ragged and empty shards."""
    import shutil
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    lib_dir = ROOT / "depthdensifier_amd"
    if not (lib_dir / "libddcore.so").exists():
        pytest.skip("libddcore.so not built")
    exe = tmp_path / "fake_rccl"
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-rdynamic", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_client" / "fake_rccl.cpp"),
                            f"-L{lib_dir}", "-lddcore", "-ldl", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "exchanges OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


def test_c_abi_allgatherv_schedule_under_sanitizers(tmp_path):
    """The same schedule with ``csrc/ddcomm.hip`` compiled for the host under AddressSanitizer + UBSan (world 2 / 3 / 8, ragged and
    empty shards, gather-to-owner): no out-of-bounds offset, no overflow in the byte arithmetic of > 4 GiB clouds.  (``function`` is left
    out of UBSan: the FAKE's entry points return int where RCCL's return an enum.)  Sanitizers run on the CPU build only."""
    clang = Path("/opt/rocm/lib/llvm/bin/clang++")
    if not clang.exists():
        pytest.skip("no clang++ under /opt/rocm")
    exe = tmp_path / "fake_rccl_asan"
    build = subprocess.run([str(clang), "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=function", "-fno-omit-frame-pointer",
                            "-fno-sanitize-recover=all", "-rdynamic", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}", "-x", "c++",
                            str(ROOT / "depthdensifier_amd" / "csrc" / "ddcomm.hip"), str(ROOT / "tests" / "c_client" / "fake_rccl.cpp"),
                            "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "exchanges OK" in run.stdout and "Sanitizer" not in run.stderr, run.stdout[-1000:] + run.stderr[-3000:]

