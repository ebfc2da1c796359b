"""CPU-side checks: the C-ABI library loads and exports every symbol of include/ddcore.h,
argument validation returns error codes without touching a GPU, and the host camera-block
math agrees with the oracle's float64 formulation."""

import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def libmod():
    import __graft_entry__ as g
    g.build()
    from depthdensifier_amd import _lib
    return _lib


def test_header_symbols_are_exported(libmod):
    header = (ROOT / "include" / "ddcore.h").read_text()
    declared = set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", header))
    lab = set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", (ROOT / "include" / "ddcore_lab.h").read_text()))
    assert lab == set(libmod.LAB_EXPORTS) and not (declared & lab)          # the experiment switches live in a header of their own
    assert declared | lab == set(libmod.EXPORTS)
    declared |= lab
    handle = C.CDLL(str(libmod.LIB_PATH))
    for sym in declared:
        assert getattr(handle, sym) is not None
    assert libmod.lib.dd_abi_version() == libmod.DD_ABI_VERSION
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if nm:                      # ... and nothing else called dd_* leaves the library: helpers shared between its translation units stay hidden
        out = subprocess.run([nm, "-D", "--defined-only", str(libmod.LIB_PATH)], capture_output=True, text=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("dd_")}
        assert exported == declared, exported ^ declared


def test_tuning_holds_only_what_a_caller_chooses(libmod):
    """DDViewBatch.tuning carries the caller's choices (DD_TUNE_*), reserved bits are refused; the A/B and
    fault-injection switches are a thread-local debug word behind include/ddcore_lab.h; no getenv in the compute sources; the
    header's description of `tuning` is short."""
    L = libmod.lib
    header = (ROOT / "include" / "ddcore.h").read_text()
    for name in ("GENERIC", "TWO_PASS", "SINGLE_PASS", "DENSE_TILES", "ASSUME_DENSE", "TILE_SMALL", "TILE_LARGE", "BY_INDEX"):
        m = re.search(r"#define DD_TUNE_%s\s+\(?([0-9a-fx]+)u(?: << (\d+))?\)?" % name, header)
        assert m, name
        value = int(m.group(1), 0) << int(m.group(2) or 0)
        assert value == getattr(libmod, "DD_TUNE_" + name), name
    field = re.search(r"uint32_t tuning;(.*?)\n    float \*refined_out", header, re.S).group(1)
    assert field.count("\n") <= 12
    for bits in (32, 64, 3 << 20, 1 << 26, 1 << 27, 1 << 28, 1 << 30):       # where the experiment switches sat up to ABI 13
        assert L.dd_workspace_bytes(C.byref(_batch(libmod, tuning=bits))) == -1 and b"reserved" in L.dd_last_error()
    assert L.dd_workspace_bytes(C.byref(_batch(libmod, tuning=libmod.DD_TUNE_ALL))) > 0
    assert L.dd_debug_tuning(libmod.DD_LAB_FAULT_INJECT) == 0 and L.dd_debug_tuning(0) == libmod.DD_LAB_FAULT_INJECT
    import threading
    seen = []
    L.dd_debug_tuning(5)
    t = threading.Thread(target=lambda: seen.append(L.dd_debug_tuning(0)))          # another thread: its own word
    t.start(); t.join()
    assert seen == [0] and L.dd_debug_tuning(0) == 5
    for src in (ROOT / "depthdensifier_amd" / "csrc").glob("*.hip"):
        hits = [ln for ln in src.read_text().splitlines() if "getenv(" in ln]
        assert src.name == "ddarena.hip" and len(hits) == 2 or not hits, (src.name, hits)      # the arena's DEBUG / TRACE flags only


def test_the_plan_of_a_batch_on_the_host(libmod):
    """dd_debug_plan: which kernels a batch takes, decided on the host (no GPU): lean on stride-1 maps, generic otherwise; the small
    tile for a streamed view, the large one for a scene; a gate only in front of a chained call."""
    L = libmod.lib
    out = (C.c_int32 * 8)()
    plan = lambda **kw: (L.dd_debug_plan(C.byref(_batch(libmod, **kw)), out), list(out))[1]
    small = plan(height=120, width=200)
    assert small[:3] == [1, 1, 0] and small[5] == 8 and small[6] == 0 and small[7] == 2 * 6          # 6 tiles of 4096 per view
    assert plan(height=1080, width=1920, num_views=200)[5] == 16
    assert plan(height=120, width=200, stride=2)[:2] == [0, 0]
    assert plan(height=120, width=200, tuning=libmod.DD_TUNE_TWO_PASS)[:2] == [1, 0]
    assert plan(height=120, width=200, tuning=libmod.DD_TUNE_GENERIC)[0] == 0
    assert plan(height=120, width=200, tuning=libmod.DD_TUNE_TILE_LARGE)[5] == 16
    assert L.dd_chain_workgroup_limit() == 0             # no GPU here: unknown -> every chained call would be gated ...
    assert plan(height=120, width=200, chain=0x3000)[6] == 1       # ... like this one


def test_the_plan_does_not_depend_on_the_callers_stack(tmp_path):
    """Round 6, the root cause of the rare GPU fault of the three-rank rehearsal: a member of the host-side plan that only chained calls
    wrote was read by every call -- with whatever the stack held there.  tests/c_client/plan_stack_test.cpp paints the stack."""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("hipcc")
    exe = tmp_path / "plan_stack_test"
    lib_dir = ROOT / "depthdensifier_amd"
    build = subprocess.run([cxx, "-std=c++17", "-O1", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_client" / "plan_stack_test.cpp"),
                            f"-L{lib_dir}", "-lddcore", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-1500:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and "OK" in run.stdout, run.stdout + run.stderr


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_the_prefetcher_under_sanitizers(tmp_path, sanitizer):
    """csrc/ddingest.hip's prefetcher (native threads, a mutex, a condition variable, slots handed back and forth) compiled for the host
    with ThreadSanitizer and with AddressSanitizer + UBSan and driven by tests/c_client/prefetch_stress.cpp: 1-8 workers, 2-9 slots,
    changing file sizes, missing files, over-submission, destroy with jobs in every state.  (Sanitizers run on the CPU build only.)"""
    import subprocess
    clang = Path("/opt/rocm/lib/llvm/bin/clang++")
    if not clang.exists():
        pytest.skip("no clang++ under /opt/rocm")
    exe = tmp_path / "prefetch_stress"
    build = subprocess.run([str(clang), "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-fno-sanitize-recover=all",
                            "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}", "-x", "c++",
                            str(ROOT / "depthdensifier_amd" / "csrc" / "ddingest.hip"), str(ROOT / "tests" / "c_client" / "prefetch_stress.cpp"),
                            "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "sanitizer" in build.stderr.lower() and "unsupported" in build.stderr.lower():
        pytest.skip(build.stderr[-300:])
    assert build.returncode == 0, build.stderr[-1500:]
    data = tmp_path / "files"
    data.mkdir()
    run = subprocess.run([str(exe), str(data), "120"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "prefetch_stress ok" in run.stdout and "Sanitizer" not in run.stderr, run.stdout[-500:] + run.stderr[-3000:]


def test_host_planning_at_its_limits_under_sanitizers(tmp_path):
    """The HOST code of csrc/ddcore.hip and csrc/ddfilter.hip built with AddressSanitizer + UBSan (``-fno-gpu-sanitize``: the kernels are
    compiled as usual and never run here) and driven by tests/c_client/host_limits_test.cpp: plans and workspace sizes from one pixel
    to views of 2^31 - 1 pixels, 2^31 - 1 views, every stride and caller's tuning, chained and not, point counts up to INT64_MAX -- a
    sane answer or a clean refusal, no overflow on the way.  (Round 6: ``height + stride - 1`` was added in 32 bits.)"""
    import shutil
    import subprocess
    clang = Path("/opt/rocm/lib/llvm/bin/clang++")
    if shutil.which("hipcc") is None or not clang.exists():
        pytest.skip("no hipcc / clang++")
    lib = tmp_path / "libddcore_asan.so"
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    build = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O0", "-std=c++17", "-fPIC", "-shared", *san, "-fno-gpu-sanitize", f"-I{ROOT / 'include'}",
                            str(ROOT / "depthdensifier_amd" / "csrc" / "ddcore.hip"), str(ROOT / "depthdensifier_amd" / "csrc" / "ddfilter.hip"), "-o", str(lib)],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    exe = tmp_path / "host_limits"
    build = subprocess.run([str(clang), "-std=c++17", "-O1", "-g", *san, f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_client" / "host_limits_test.cpp"),
                            f"-L{tmp_path}", "-lddcore_asan", f"-Wl,-rpath,{tmp_path}", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "host limits OK" in run.stdout and "Sanitizer" not in run.stderr, run.stdout[-800:] + run.stderr[-3000:]


def test_struct_layout_matches_header(libmod):
    # DDViewParams is 32 floats; DDViewBatch / DDCloudOut sizes for the LP64 layout in the header
    assert C.sizeof(libmod.DDViewBatch) == 4 * 4 + 6 * 8 + 6 * 4 + 8 + 2 * 8      # (+ chain, chain_seq: ABI 12)
    assert C.sizeof(libmod.DDCloudOut) == 5 * 8 + 8 + 8


def _batch(libmod, **kw):
    b = libmod.DDViewBatch(num_views=2, height=8, width=8, stride=1, depth=0x1000, params=0x2000,
                           depth_dtype=libmod.DD_F32, flags=libmod.DD_VALID_DEPTH_POSITIVE)
    for k, v in kw.items():
        setattr(b, k, v)
    return b


@pytest.mark.parametrize("kw, text", [
    (dict(num_views=0), "positive"),
    (dict(stride=0), "stride"),
    (dict(depth=None), "depth is NULL"),
    (dict(params=None), "params is NULL"),
    (dict(depth_dtype=7), "depth_dtype"),
    (dict(flags=0), "no validity rule"),
    (dict(flags=0x2), "mask is NULL"),
    (dict(flags=0x4), "conf is NULL"),
])
def test_invalid_arguments_return_codes(libmod, kw, text):
    b = _batch(libmod, **kw)
    rc = libmod.lib.dd_workspace_bytes(C.byref(b))
    assert rc == -1
    assert text in libmod.lib.dd_last_error().decode()
    with pytest.raises(libmod.DDCoreError):
        libmod.check(rc)
    assert libmod.lib.dd_count_valid(C.byref(b), None, None) == -1


def test_neighbour_entry_points_validate_before_launching(libmod):
    """dd_floater_votes / dd_compact_cloud / dd_refine_apply reject bad arguments with DD_ERR_* and a message -- no GPU
    work is attempted (this runs on a machine without one)."""
    L = libmod.lib
    fv = libmod.DDFilterViews(num_views=0, height=8, width=8, depth=0x1000, cams=0x2000, grazing_cos=0.087, depth_threshold=0.7)
    assert L.dd_floater_votes(C.byref(fv), 0x10, 0x20, 5, 0x30, 0, None) == -1 and b"positive" in L.dd_filter_last_error()
    fv.num_views, fv.depth = 2, None
    assert L.dd_floater_votes(C.byref(fv), 0x10, 0x20, 5, 0x30, 0, None) == -1 and b"NULL" in L.dd_filter_last_error()
    fv.depth = 0x1000
    assert L.dd_floater_votes(C.byref(fv), None, 0x20, 5, 0x30, 0, None) == -1
    assert L.dd_floater_votes(C.byref(fv), 0x10, 0x20, -1, 0x30, 0, None) == -1 and b"negative" in L.dd_filter_last_error()
    assert L.dd_floater_votes(C.byref(fv), None, None, 0, None, 0, None) == 0            # nothing to do is not an error
    assert L.dd_floater_votes(None, 0x10, 0x20, 5, 0x30, 0, None) == -1

    src = libmod.DDCloudOut(xyz=0x100, capacity=10)
    dst = libmod.DDCloudOut(xyz=0x200, normal=0x300, capacity=10)
    assert L.dd_compact_workspace_bytes(-1) == -1 and L.dd_compact_workspace_bytes(0) >= 16
    assert L.dd_compact_cloud(C.byref(src), 10, 0x40, 5, C.byref(dst), 0x50, None, None, 0, 0x1000, 1 << 20, None) == -1
    assert b"no input field" in L.dd_filter_last_error()
    dst.normal = None
    dst.capacity = -1
    assert L.dd_compact_cloud(C.byref(src), 10, 0x40, 5, C.byref(dst), 0x50, None, None, 0, 0x1000, 1 << 20, None) == -1
    assert b"capacity" in L.dd_filter_last_error()
    dst.capacity = 10
    dst.xyz_rgba = 0x208                                                                    # the 16-byte record must be 16-byte aligned
    assert L.dd_compact_cloud(C.byref(src), 10, 0x40, 5, C.byref(dst), 0x50, None, None, 0, 0x1000, 1 << 20, None) == -1
    assert b"xyz_rgba" in L.dd_filter_last_error()
    dst.xyz_rgba = None
    assert L.dd_compact_cloud(C.byref(src), 10, 0x40, 5, C.byref(dst), 0x50, 0x60, None, 3, 0x1000, 1 << 20, None) == -1
    assert L.dd_compact_cloud(C.byref(src), 10, 0x40, 5, C.byref(dst), 0x50, None, None, 0, None, 0, None) == -3        # DD_ERR_WORKSPACE
    assert L.dd_compact_cloud(C.byref(src), 10, 0x40, 5, C.byref(dst), 0x50, None, None, 0, 0x1008, 1 << 20, None) == -3   # mis-aligned

    assert L.dd_refine_apply(None, 0, None, 8, 8, 0x10, 0x20, 4, 0, 0x30, None) == -1 and b"NULL" in L.dd_refine_last_error()
    assert L.dd_refine_apply(0x100, 0, None, 0, 8, 0x10, 0x20, 4, 0, 0x30, None) == -1
    assert L.dd_refine_apply(0x100, 9, None, 8, 8, 0x10, 0x20, 4, 0, 0x30, None) == -1 and b"depth_dtype" in L.dd_refine_last_error()
    assert L.dd_refine_apply(0x100, 0, None, 8, 8, 0x10, 0x20, 1, 0, 0x30, None) == -1 and b"two" in L.dd_refine_last_error()


def test_workspace_size_and_null_outputs(libmod):
    b = _batch(libmod, height=1080, width=1920, num_views=3)
    n = libmod.lib.dd_workspace_bytes(C.byref(b))
    tiles = -(-1080 * 1920 // 4096) * 3
    # 16 B sticky + 48 B single-pass state, look-back granules (8 B per tile, padded to 16), the tiles' first rows from the scan service
    # (the same again), count + first row for the two-pass kernels (8 B per tile), 8 B per view
    assert n == 64 + 2 * ((8 * tiles + 15) // 16 * 16) + 8 * tiles + 8 * 3
    out = libmod.DDCloudOut(capacity=10)
    assert libmod.lib.dd_unproject_compact(C.byref(b), C.byref(out), None, None, None, 0, None) == -1
    assert "xyz" in libmod.lib.dd_last_error().decode()
    assert libmod.lib.dd_count_valid(C.byref(b), None, None) == -1


def test_camera_blocks_against_oracle():
    import depthdensifier_amd as dd
    from oracle import densify_oracle as orc
    from synth import random_pose

    rng = np.random.default_rng(3)
    V = 5
    E = np.stack([random_pose(rng) for _ in range(V)])
    params = np.stack([[500 + v, 510 - v, 320.5, 240.25] for v in range(V)]).astype(np.float64)
    blocks = dd.camera_blocks(params, E).astype(np.float64)
    px = rng.integers(0, 640, 50)
    py = rng.integers(0, 480, 50)
    d = rng.uniform(0.5, 5, 50)
    for v in range(V):
        ref = orc.rigid_inverse_apply(E[v], orc.unproject_pinhole(px, py, d, params[v]))
        M, c = blocks[v, :9].reshape(3, 3), blocks[v, 9:12]
        got = d[:, None] * (np.stack([px, py, np.ones_like(px)], -1) @ M.T) + c
        assert np.abs(got - ref).max() < 5e-6 * np.abs(ref).max()
        assert np.allclose(blocks[v, 12:21].reshape(3, 3), E[v, :, :3].T, atol=1e-7)
    # (4,4) extrinsics and a single (3,3) K broadcast over views
    E4 = np.concatenate([E, np.tile([[[0, 0, 0, 1.0]]], (V, 1, 1))], axis=1)
    K = np.array([[500.0, 0.3, 320], [0, 510, 240], [0, 0, 1]])
    assert dd.camera_blocks(K, E4).shape == (V, 32)
    with pytest.raises(ValueError):
        dd.camera_blocks(np.ones((V, 5)), E)


def test_no_gpu_means_loud_failure():
    import torch
    import depthdensifier_amd as dd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dd.unproject_views(np.ones((1, 4, 4), np.float32), np.array([[1.0, 1, 0, 0]]), np.eye(4)[None, :3])


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (the timed baseline, and since round 3 the
    check of the timed cloud against it) may touch oracle/: the package, the alias package, the entry
    scripts and the measurement tools must not even mention it in code."""
    import re
    for folder in ("depthdensifier_amd", "depthdensifier", "scripts"):
        for f in (ROOT / folder).rglob("*.py"):
            assert "oracle" not in f.read_text(), f
    for f in (ROOT / "tools").rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle", f.read_text(), re.M), f
    bench_src = (ROOT / "bench.py").read_text()
    uses = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle", bench_src, re.M)]
    assert uses, "bench.py times the oracle as its CPU baseline"
    for u in uses:          # every use sits inside a function of the cpu_baseline leg
        owner = re.findall(r"^def (\w+)\(", bench_src[:u], re.M)[-1]
        assert owner in ("cpu_baseline", "_cpu_worker", "verify_views"), f"bench.py uses the oracle in {owner}()"
    # verify_views is a checker: it runs after the timed region and its result never feeds the product path
    timed = bench_src[bench_src.index("t0 = time.perf_counter()\n        for _ in range(steps):"):bench_src.index("elapsed = time.perf_counter() - t0")]
    assert "verify_views" not in timed and "oracle" not in timed


def test_bench_byte_model_matches_survey_examples():
    """config 3 (f32 depth + u8 mask + f32x3 normal + u8x3 rgb, rho 0.8) = 17 B read + 21.6 B
    written per pixel; config 5 (f16 depth in, f32 xyz out, dense) = 2 + 12 B per pixel."""
    import bench
    P = 1920 * 1080
    b3 = bench.algorithmic_bytes(bench.WORKLOADS["scene2000"], 1, int(0.8 * P), False) - 72
    assert abs(b3 / P - 38.6) < 1e-6
    P5 = 4032 * 3024
    b5 = bench.algorithmic_bytes(bench.WORKLOADS["roofline12mp"], 1, P5, False) - 72
    assert b5 / P5 == 14.0
    b4 = bench.algorithmic_bytes(bench.WORKLOADS["mip360conf"], 1, int(0.425 * P), True) - 72
    assert abs(b4 / P - (9 + 0.425 * (15 + 31))) < 1e-6


def test_votes_workspace_covers_every_mode(libmod):
    """dd_votes_workspace_bytes() is documented to suffice for the float32 first pass AND the culling modes (two
    256-byte tables per view, the decision counters, one mask of V bits per 65 536 points)."""
    for V, n in ((1, 0), (1, 1), (185, 326_000_000), (2000, 3_300_000_000), (1_000_000, 10), (70_000, 5_000_000_000)):
        need = 512 * V + 64 + -(-n // 65536) * (-(-V // 64)) * 8
        got = libmod.lib.dd_votes_workspace_bytes(V, n)
        assert got >= need and got % 8 == 0, (V, n, got, need)
    assert libmod.lib.dd_votes_workspace_bytes(0, 10) < 0 and libmod.lib.dd_votes_workspace_bytes(4, -1) < 0


def test_integration_md_stub_matches_the_binding(libmod):
    """The ctypes stub shown to a reference maintainer in INTEGRATION.md declares the same fields, in the same order, as
    the binding the package itself uses (a missing trailing field would make the library read past the struct)."""
    text = (ROOT / "INTEGRATION.md").read_text()
    for name, cls in (("DDViewBatch", libmod.DDViewBatch), ("DDCloudOut", libmod.DDCloudOut)):
        body = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\s*(#.*)?\n\n" % name, text, re.S).group(1)
        shown = re.findall(r'\("(\w+)",\s*C\.(\w+)\)', body)
        assert [n for n, _ in shown] == [f[0] for f in cls._fields_], name
        assert [getattr(C, t) for _, t in shown] == [f[1] for f in cls._fields_], name
    assert "dd_abi_version() == %d" % libmod.DD_ABI_VERSION in text


def test_bench_byte_model_matches_the_survey():
    """bench.py's algorithmic bytes are SURVEY.md 8(d)'s: config 3 (f32 depth + u8 mask + normals + rgb, rho = 0.8) is
    17 B read + 21.6 B written = 38.6 B/px, config 5 (f16 depth in, xyz out, dense) 2 + 12 = 14 B/px (+ 72 B per view);
    whole scenes are dealt to the ranks largest first."""
    import bench
    P = 1080 * 1920
    V, rho = 10, 0.8
    n = int(rho * V * P)
    cfg = dict(bench.WORKLOADS["scene2000"])
    total = bench.algorithmic_bytes(cfg, V, n, pixel_index=False)
    reads = bench.algorithmic_bytes(cfg, V, n, pixel_index=False, reads_only=True)
    assert abs(total / (V * P) - 38.6) < 1e-3 and abs(reads / (V * P) - 17.0) < 1e-3
    assert bench.algorithmic_bytes(cfg, V, n, pixel_index=True) - total == 4 * n
    cfg5 = dict(bench.WORKLOADS["roofline12mp"])
    P5 = cfg5["H"] * cfg5["W"]
    assert bench.algorithmic_bytes(cfg5, 3, 3 * P5, pixel_index=False) == 3 * P5 * 14 + 3 * 72
    conf = dict(bench.WORKLOADS["mip360conf"])
    assert bench.algorithmic_bytes(conf, 1, 0, pixel_index=False) == P * 9 + 72          # depth + mask + f32 confidence
    sizes = [v for _, v in bench.WORKLOADS["mip360x7"]["scenes"]]
    assert sum(sizes) == bench.WORKLOADS["mip360x7"]["V"] == 1626
    for world in (1, 2, 3, 7, 8):
        owner = bench.deal_scenes(sizes, world)
        load = [sum(s for s, o in zip(sizes, owner) if o == r) for r in range(world)]
        assert sorted(set(owner)) == list(range(min(world, len(sizes)))) and max(load) - min(l for l in load if l) <= max(sizes)


def test_csrc_makefile_builds_a_loadable_library(libmod, tmp_path):
    """`make -C depthdensifier_amd/csrc` is a documented build path (_lib.py): it must compile the same
    sources with the same flags as __graft_entry__.build() and its result must export every symbol the binding resolves."""
    import shutil
    import subprocess
    import __graft_entry__ as g
    mk = (ROOT / "depthdensifier_amd" / "csrc" / "Makefile").read_text()
    srcs = re.search(r"^SRCS\s*:=\s*(.+)$", mk, re.M).group(1).split()
    assert sorted(srcs) == sorted(p.name for p in g.HIP_SOURCES)
    for name, flags in g.EXTRA_FLAGS.items():
        assert re.search(rf"{Path(name).stem}\.o: EXTRA := {' '.join(flags)}", mk), name
    assert "-ldl" in mk and "ddrefine_math.h" in mk
    if shutil.which("hipcc") is None or shutil.which("make") is None:
        pytest.skip("hipcc / make not on PATH")
    out = tmp_path / "libddcore.so"
    subprocess.run(["make", "-j5", "-C", str(ROOT / "depthdensifier_amd" / "csrc"), f"OUT={out}", f"OBJDIR={tmp_path / 'obj'}"],
                   check=True, capture_output=True, timeout=600)
    handle = C.CDLL(str(out))
    for sym in libmod.EXPORTS:
        assert getattr(handle, sym) is not None
    handle.dd_abi_version.restype = C.c_int
    assert handle.dd_abi_version() == libmod.DD_ABI_VERSION


def test_arena_layout_planning(tmp_path):
    """The HBM zone arena's planning (which class every chunk of every array comes from) is plain C++
    (csrc/ddarena_plan.h): compiled with g++ and run here, no GPU involved."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not installed")
    exe = tmp_path / "arena_plan_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    f"-I{ROOT / 'include'}", f"-I{ROOT / 'depthdensifier_amd' / 'csrc'}",
                    "-o", str(exe), str(ROOT / "tests" / "c_client" / "arena_plan_test.cpp")], check=True, timeout=120)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)             # (under AddressSanitizer + UBSan)
    assert out.returncode == 0 and "plan OK" in out.stdout and "Sanitizer" not in out.stderr, out.stdout + out.stderr


def test_native_npy_reader_and_prefetcher(libmod, tmp_path):
    """csrc/ddingest.hip on the host: .npy headers and arrays read without the interpreter (element type and shape checked against what
    the view needs), and the prefetcher -- native threads filling staging slots ahead of the consumer, a slot refilled only after its
    release -- against np.load.  (No GPU here: the slots are ordinary memory and events are NULL; the GPU twin is the pipeline itself.)"""
    import numpy as np
    L = libmod.lib
    rng = np.random.default_rng(3)
    H, W, V = 12, 20, 9
    want = []
    for v in range(V):
        maps = {"depth": rng.uniform(0.5, 5, (H, W)).astype(np.float16 if v % 2 else np.float32), "mask": rng.uniform(size=(H, W)) < 0.7,
                "normal": rng.normal(size=(H, W, 3)).astype(np.float32), "rgb": rng.integers(0, 256, (H, W, 3), dtype=np.uint8)}
        for k, a in maps.items():
            np.save(tmp_path / f"v{v}_{k}.npy", a)
        want.append(maps)
    np.save(tmp_path / "f64.npy", np.zeros((2, 2)))
    np.save(tmp_path / "fortran.npy", np.asfortranarray(np.zeros((3, 4), np.float32)))
    dt, nd, shape, off = C.c_int32(), C.c_int32(), (C.c_int64 * 4)(), C.c_int64()
    assert L.dd_npy_header(str(tmp_path / "v1_depth.npy").encode(), C.byref(dt), C.byref(nd), shape, C.byref(off)) == 0
    assert (dt.value, nd.value, list(shape)[:2]) == (libmod.DD_NPY_F16, 2, [H, W]) and off.value % 64 == 0
    assert L.dd_npy_header(str(tmp_path / "f64.npy").encode(), C.byref(dt), C.byref(nd), shape, None) == -4 and b"element type" in L.dd_ingest_last_error()
    assert L.dd_npy_header(str(tmp_path / "fortran.npy").encode(), C.byref(dt), C.byref(nd), shape, None) == -4
    assert L.dd_npy_header(str(tmp_path / "nothing.npy").encode(), C.byref(dt), C.byref(nd), shape, None) == -1 and b"No such file" in L.dd_ingest_last_error()
    out = np.empty((H, W, 3), np.float32)
    ok = lambda name, code, shp: L.dd_npy_read(str(tmp_path / name).encode(), code, len(shp), (C.c_int64 * 4)(*shp), out.ctypes.data, out.nbytes)
    assert ok("v0_normal.npy", libmod.DD_NPY_F32, (H, W, 3)) == 0 and np.array_equal(out, want[0]["normal"])
    assert ok("v0_normal.npy", libmod.DD_NPY_F32, (H, W + 1, 3)) == -1 and ok("v0_normal.npy", libmod.DD_NPY_F16, (H, W, 3)) == -1
    assert ok("v0_mask.npy", libmod.DD_NPY_U8, (H, W)) == 0                       # bool files read as bytes
    # the prefetcher: 4 slots, 3 threads, 3 jobs ahead
    sizes = {"depth": H * W * 4, "mask": H * W, "normal": H * W * 12, "rgb": H * W * 3}
    offs, at = {}, 0
    for k, n in sizes.items():
        offs[k], at = at, at + ((n + 63) & ~63)
    handle = C.c_void_p()
    assert L.dd_prefetch_create(3, 4, at, C.byref(handle)) == 0
    keys = list(sizes)

    def submit(v):
        paths = (C.c_char_p * 4)(*[str(tmp_path / f"v{v}_{k}.npy").encode() for k in keys])
        codes = (C.c_int32 * 4)(-1, libmod.DD_NPY_BOOL, libmod.DD_NPY_F32, libmod.DD_NPY_U8)
        nds = (C.c_int32 * 4)(2, 2, 3, 3)
        shp = (C.c_int64 * 16)(H, W, 1, 1, H, W, 1, 1, H, W, 3, 1, H, W, 3, 1)
        return L.dd_prefetch_submit(handle, 4, paths, codes, nds, shp, (C.c_int64 * 4)(*[offs[k] for k in keys]))

    tickets = [submit(v) for v in range(3)]
    assert tickets == [0, 1, 2]
    for v in range(V):
        base, found = C.c_void_p(), (C.c_int32 * 8)()
        assert L.dd_prefetch_wait(handle, tickets[v], C.byref(base), found) == 0, L.dd_ingest_last_error()
        assert found[0] == (libmod.DD_NPY_F16 if v % 2 else libmod.DD_NPY_F32)
        for i, k in enumerate(keys):
            a = want[v][k]
            got = np.frombuffer((C.c_char * a.nbytes).from_address(base.value + offs[k]), dtype=a.dtype).reshape(a.shape)
            assert np.array_equal(got, a), (v, k)
        assert L.dd_prefetch_release(handle, tickets[v], None) == 0
        assert L.dd_prefetch_release(handle, tickets[v], None) == -1                   # once
        if v + 3 < V:
            tickets.append(submit(v + 3))
    # a job whose file is missing reports it at the wait; more jobs than slots without a release is refused
    paths = (C.c_char_p * 1)(str(tmp_path / "gone.npy").encode())
    t = L.dd_prefetch_submit(handle, 1, paths, (C.c_int32 * 1)(-1), (C.c_int32 * 1)(2), (C.c_int64 * 4)(H, W, 1, 1), (C.c_int64 * 1)(0))
    assert t == V and L.dd_prefetch_wait(handle, t, None, None) == -1 and b"gone.npy" in L.dd_ingest_last_error()
    for _ in range(3):
        assert submit(0) >= 0
    assert submit(0) == -3 and b"released" in L.dd_ingest_last_error()
    assert L.dd_prefetch_destroy(handle) == 0


def test_a_feeder_left_open_gives_its_slots_to_the_next_scan(libmod, tmp_path):
    """depth_source.NativeFeeder keeps one native prefetcher per process across scans.  A scan that ended in an exception may leave its
    feeder open with views read ahead: the next scan's feeder gives those jobs back first (same slot size), or replaces the prefetcher
    (larger views) -- and the old feeder's late close is harmless either way.  No GPU: slots are ordinary memory."""
    import gc
    from depthdensifier_amd.depth_source import CachedSource, NativeFeeder, _PREFETCHERS
    rng = np.random.default_rng(5)

    def scan(folder, n, h, w):
        folder.mkdir()
        want = {}
        for v in range(n):
            want[f"im{v}.png"] = maps = {"depth": rng.uniform(0.5, 5, (h, w)).astype(np.float32), "mask": rng.uniform(size=(h, w)) < 0.6,
                                         "normal": rng.normal(size=(h, w, 3)).astype(np.float32), "rgb": rng.integers(0, 256, (h, w, 3), dtype=np.uint8)}
            for k, a in maps.items():
                np.save(folder / f"im{v}_{k}.npy", a)
        return CachedSource(folder), want

    def read_all(feeder, want, upto=None):
        for k, name in enumerate(list(want)[:upto]):
            rgb, maps, slot = feeder.get(k)
            for key, a in want[name].items():
                got = np.frombuffer((C.c_char * a.nbytes).from_address(slot.pointer(key)), dtype=a.dtype).reshape(a.shape)
                assert np.array_equal(got, a), (name, key)
            assert rgb.shape == want[name]["rgb"].shape and maps["depth"].dtype == np.float32
            slot.released_natively()

    src_a, want_a = scan(tmp_path / "a", 7, 6, 10)
    a = NativeFeeder(src_a, list(want_a), {n: (10, 6) for n in want_a}, threads=2, ahead=3)
    read_all(a, want_a, upto=2)                      # ... and the scan "fails" here: views 2-4 are read ahead, the feeder stays open
    handle = _PREFETCHERS[(2, 5)][0]
    src_b, want_b = scan(tmp_path / "b", 6, 6, 10)
    b = NativeFeeder(src_b, list(want_b), {n: (10, 6) for n in want_b}, threads=2, ahead=3)
    assert a._h is None and _PREFETCHERS[(2, 5)][0] is handle          # the same prefetcher, the old feeder closed
    read_all(b, want_b)
    with pytest.raises((RuntimeError, KeyError)):
        a.get(2)                                     # the old feeder is of no use any more, and says so
    # larger views with the second feeder still open: a new prefetcher; the old feeders' finalisers touch nothing that is gone
    src_c, want_c = scan(tmp_path / "c", 4, 9, 16)
    c = NativeFeeder(src_c, list(want_c), {n: (16, 9) for n in want_c}, threads=2, ahead=3)
    assert b._h is None and _PREFETCHERS[(2, 5)][0] is not handle
    del a, b
    gc.collect()
    read_all(c, want_c)
    c.close()
