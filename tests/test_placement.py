"""HBM zone arena (csrc/ddarena.hip, depthdensifier_amd/placement.py): the arrays of a cloud are built from physical chunks of
different HBM classes.  Placement changes addresses only: the cloud must equal the unplaced one bit for bit."""

import ctypes as C
import gc

import numpy as np
import pytest


def test_arena_entry_points_validate_without_a_gpu():
    """Argument errors come back as codes with a message; nothing is launched (runs on a machine without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from depthdensifier_amd import _lib
    L = _lib.lib
    h = C.c_void_p()
    assert L.dd_arena_create(0, 12345, C.byref(h)) == -1 and b"2 MiB" in L.dd_arena_last_error()
    assert L.dd_arena_create(0, 0, None) == -1
    assert L.dd_arena_alloc(None, 1, None, None, 0, None) == -1
    assert L.dd_arena_free(None, None) == -1 and L.dd_arena_stats(None, None) == -1
    assert L.dd_arena_classes(None, None, None, 0) == -1 and L.dd_arena_probe(None, None, None, None) == -1
    assert L.dd_arena_destroy(None) == 0
    assert L.dd_arena_trim(None, 0) == -1
    assert C.sizeof(_lib.DDArenaStats) == 8 + 8 + 4 + 4 + 3 * 8 + 3 * 8 + 3 * 8 + 4 + 4 + 8


def test_placement_mode_from_environment(monkeypatch):
    from depthdensifier_amd import placement as pl
    monkeypatch.delenv("DD_PLACEMENT", raising=False)
    assert pl.default_mode() == "probed"
    monkeypatch.setenv("DD_PLACEMENT", "first")
    assert pl.default_mode() == "first"
    monkeypatch.setenv("DD_PLACEMENT", "nonsense")
    assert pl.default_mode() == "probed"


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda", 0)


@pytest.mark.gpu
def test_arena_gives_every_group_its_own_class(gpu):
    import torch
    from depthdensifier_amd import placement as pl
    arena = pl.get_arena(gpu)
    rows = 40 << 20                                   # 480 MiB of float32 rows: more than one probe window
    specs = {"a": ((rows, 3), torch.float32, 0), "b": ((rows, 3), torch.float32, 1), "c": ((3 * rows,), torch.uint8, 2)}
    t, degraded = arena.alloc(specs)
    assert not degraded
    cls = {k: set(arena.classes_of(v)) for k, v in t.items()}
    assert all(len(c) == 1 for c in cls.values()), cls                      # every array from ONE class
    assert len(set.union(*cls.values())) == 3, cls                          # three arrays, three classes
    st = arena.stats()
    assert st["num_classes"] == 3 and st["cross_class_ms"] < 0.9 * st["same_class_ms"], st
    # the arrays are ordinary device memory
    t["a"].fill_(1.5); t["c"].fill_(7)
    assert float(t["a"].sum(dtype=torch.float64)) == 1.5 * 3 * rows and int(t["c"][-1]) == 7
    # the probe sees what the classes say: two arrays of different groups are a fast pair, two halves of one array a slow one
    cross = arena.probe_ms(t["a"], t["b"])
    same = arena.probe_ms(t["a"], t["a"][rows // 2:]) if (rows // 2) * 12 >= st["probe_bytes"] else None
    assert cross < 0.92 * st["same_class_ms"], (cross, st)
    if same is not None:
        assert same > cross * 1.08, (same, cross)
    # a later call keeps the groups where they are
    t2, _ = arena.alloc({"d": ((rows, 3), torch.float32, 1)})
    assert set(arena.classes_of(t2["d"])) == cls["b"]
    # rotated layout: chunk k from class (phase + k) mod 3 -- two arrays of different phase never share a class at equal rows
    big = 3 * (st["chunk_bytes"] // 12) + 1000                                # a little more than three chunks of float32 rows
    t3, deg3 = arena.alloc({"p": ((big, 3), torch.float32, pl.rotated(0)), "q": ((big, 3), torch.float32, pl.rotated(1))})
    cp, cq = arena.classes_of(t3["p"]), arena.classes_of(t3["q"])
    assert not deg3 and len(cp) == len(cq) == 4 and all(a != b for a, b in zip(cp, cq)), (cp, cq)      # (the exact rotation when the supply is even)
    t3["p"][-1].fill_(2.0)                                                     # the last row lies in the fourth chunk
    assert float(t3["p"][-1].sum()) == 6.0
    held = sum(arena.stats()["chunks_held"])
    del t, t2, t3
    gc.collect()
    st2 = arena.stats()
    assert sum(st2["chunks_pooled"]) > 0 and sum(st2["chunks_held"]) <= held  # spares are kept (small arrays stay mapped in the cache), nothing was added
    arena.trim()
    st3 = arena.stats()
    assert sum(st3["chunks_pooled"]) == 0 and sum(st3["chunks_held"]) == 3    # only the three anchors stay


@pytest.mark.gpu
def test_placed_cloud_equals_unplaced_cloud(gpu):
    """18 views of 1080p (37 M rows: above the placement threshold): same rows, same bits, whichever pages hold them; and the
    cloud stays alive after its builder is gone."""
    import torch
    import depthdensifier_amd as dd
    from synth import make_views
    V, H, W = 18, 1080, 1920
    d = make_views(11, V, H, W, rho=0.8)
    params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (V, 1))
    batch = dd.ViewBatch(d["depth"], params, d["cam_from_world"], mask=d["mask"], normal=d["normal"], rgb=d["rgb"], device=gpu)
    clouds = {}
    for mode in ("first", "probed"):
        b = dd.CloudBuilder(batch.max_points, normals=True, colors=True, pixel_index=True, device=gpu, placement=mode)
        assert b.placement.mode == mode, b.placement.as_dict()
        b.append(batch)
        clouds[mode] = b.finish()
        if mode == "probed":
            cl = b.placement.classes
            assert b.placement.layout == "rotated" and cl["points"][0] != cl["normals"][0], b.placement.as_dict()
        del b
    gc.collect()
    a, p = clouds["first"], clouds["probed"]
    assert len(a) == len(p) > 0
    for f in ("points", "normals", "colors", "pixel_index", "view_offsets"):
        assert torch.equal(getattr(a, f), getattr(p, f)), f


@pytest.mark.gpu
def test_small_clouds_are_left_alone_unless_spares_are_at_hand(gpu):
    """Below the threshold a cloud is allocated plainly -- scouting the memory would cost more than it wins -- UNLESS the arena
    already holds classified spare chunks (left by an earlier, larger cloud): then placing costs nothing and is done."""
    import depthdensifier_amd as dd
    from depthdensifier_amd import placement
    arena = placement.get_arena(gpu)
    arena.trim(0)                                                                  # no spares
    b = dd.CloudBuilder(40 << 20, normals=True, colors=True, device=gpu)          # 40 Mi rows: below the default threshold
    assert b.placement is not None and b.placement.mode.startswith("skipped"), b.placement.as_dict()
    b2 = dd.CloudBuilder(1000, normals=False, colors=False, device=gpu)
    assert b2.placement is None
    del b, b2
    arena.trim(4)                                                                  # the default pool again: 4 spares per class
    big = dd.CloudBuilder(100 << 20, normals=True, colors=True, device=gpu, placement="probed")     # scouts, then leaves its chunks as spares
    assert big.placement.mode == "probed"
    del big
    gc.collect()
    assert sum(arena.stats()["chunks_pooled"]) >= 3
    probes = arena.stats()["probes"]
    small = dd.CloudBuilder(40 << 20, normals=True, colors=True, device=gpu)
    assert small.placement.mode == "probed" and small.placement.classes["points"][0] != small.placement.classes["normals"][0], small.placement.as_dict()
    assert arena.stats()["probes"] == probes                                       # served from the pool: nothing was scouted


@pytest.mark.gpu
def test_arena_churn_without_python(gpu, tmp_path):
    """Arrays allocated, filled, read back and freed 1500 times through the C ABI alone, with ordinary hipMalloc / hipFree traffic in
    between and a change of shape every fifth round (``tests/c_client/arena_churn.cpp``): every array reads what was written to it.
    Round 4 found that a virtual range that is mapped a second time keeps translating to its FIRST physical memory on this stack
    (``tools/experiments/ubench_vmm_remap.hip``); the arena of rounds 3-4 reused ranges and failed this test in its second round."""
    import shutil, subprocess
    from pathlib import Path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "arena_churn"
    lib_dir = root / "depthdensifier_amd"
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", f"-I{root / 'include'}",
                            str(root / "tests" / "c_client" / "arena_churn.cpp"), f"-L{lib_dir}", "-lddcore",
                            f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe), "1500", "1"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "arena churn OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.gpu
def test_set_pool_releases_nothing(gpu):
    """``dd_arena_set_pool`` (ABI 14; ``placement.keep_everything``: a process under rocprofv3 must not give chunks back, they would
    be lost): it only moves the number of spare chunks kept per class -- unlike ``dd_arena_trim`` nothing is released at the call, and
    arrays freed afterwards stay with the arena, every chunk of them."""
    import torch
    from depthdensifier_amd import placement as pl
    from depthdensifier_amd._lib import lib
    arena = pl.ZoneArena(torch.device(gpu))                    # an arena of its own: the process-wide one keeps its settings
    chunk = arena.stats()["chunk_bytes"]
    rows = 2 * (chunk // 12) + 1000                            # three chunks per array
    t, _ = arena.alloc({"a": ((rows, 3), torch.float32, 0), "b": ((rows, 3), torch.float32, 1), "c": ((rows, 3), torch.float32, 2)})
    before = arena.stats()
    assert lib.dd_arena_set_pool(arena._handle, 1 << 20) == 0
    mid = arena.stats()
    assert mid["chunks_released"] == before["chunks_released"] and mid["chunks_held"] == before["chunks_held"]      # nothing moved at the call
    del t
    gc.collect()
    after = arena.stats()
    assert after["chunks_released"] == before["chunks_released"], (before, after)                                  # ... nor when the arrays go
    assert sum(after["chunks_held"]) >= 9
    t2, degraded = arena.alloc({"d": ((rows, 3), torch.float32, 0), "e": ((rows, 3), torch.float32, 1)})           # served from what was kept
    assert not degraded and arena.stats()["chunks_created"] == after["chunks_created"]
    assert lib.dd_arena_set_pool(arena._handle, -1) < 0                                                           # a negative pool is refused
    del t2
    gc.collect()
    arena.trim(0)
    assert sum(arena.stats()["chunks_pooled"]) == 0
