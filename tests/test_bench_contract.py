"""bench.py keeps the driver's contract: ONE JSON line on stdout with the agreed keys, at N = 1 and (rehearsed with the
ranks sharing the box's GPU over gloo) at N = 2, including the strong-scaling sub-record and the multi-scene workload."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _run(cmd, env_extra=None, timeout=240):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, **(env_extra or {}))
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}: {out.stdout[:500]}"
    return json.loads(lines[0])


def _check(line, n_gpus, steps, warmup):
    assert LINE_KEYS <= set(line), LINE_KEYS - set(line)
    assert line["n_gpus"] == n_gpus and line["steps"] == steps and line["warmup"] == warmup
    assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert line["unit"] == "Mpixels/s" and line["value"] > 0 and line["ms_per_step"] > 0
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert ROOFLINE_KEYS <= set(r) and r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1 and 0 < r["read_frac"] < r["frac"]


def test_single_gpu_line():
    """The driver's N = 1 command with small sizes: `value` on scene2000 (the scene BASELINE.json's metric is quoted on) and the other
    single-GPU configurations -- garden185 with its streaming chains, roofline12mp, mip360conf -- as sub-records of the same line."""
    line = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--views", "6", "--strong-views", "6", "--strong-steps", "1",
                 "--cpu-seconds", "1", "--cpu-procs", "2", "--alloc-rounds", "2", "--sub-steps", "2"], timeout=420)
    _check(line, 1, 3, 1)
    assert line["scaling"] == "strong" and line["config"]["workload"] == "scene2000" and line["config"]["views_total"] == 6
    cpu = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["cores"] == 1 and cpu["kind"] == "port" and cpu["value"] > 0
    assert cpu["reference_formulation"]["value"] > 0
    s = line["strong2000"]
    assert s["views_total"] == 6 and all(s[k]["ms"] > 0 for k in ("sharded", "gathered", "gathered_compact")), s
    assert s["bernoulli"]["sharded"]["ms"] > 0 and s["bernoulli"]["verified"]["all_ranks_ok"] and 0.7 < s["bernoulli"]["sharded"]["valid_fraction"] < 0.9
    # round 3: the timed cloud is checked against the oracle inside the bench, and the kernel is re-timed on fresh allocations
    v = line["verified"]
    assert v["all_ranks_ok"] and v["views"] >= 2 and v["points"] > 0 and v["xyz_max_rel"] <= 1e-4, v
    assert s["verified"]["all_ranks_ok"], s["verified"]
    r = line["roofline"]
    assert r["placement"] in ("probed", "first") or r["placement"].startswith(("skipped", "degraded")), r["placement"]
    assert r["alloc_rounds"] == 2 and r["frac_min"] <= r["frac_median"] <= r["frac_max"] and len(r["kernel_ms_per_allocation"]) == 2
    # round 5: every single-GPU configuration in the driver-run line
    for name in ("garden185", "roofline12mp", "mip360conf"):
        sub = line[name]
        assert "error" not in sub, sub
        assert sub["value"] > 0 and 0 < sub["roofline"]["frac"] < 1 and 0 < sub["roofline"]["whole_step_frac"] < 1, sub
        assert sub["verified"]["all_ranks_ok"] and sub["roofline"]["redone"] == {"healed": 0, "dense_misses": 0}, sub
    st = line["garden185"]["streaming"]
    assert st["all_ok"] and set(st["per_call"]) == {"1", "2", "4", "8", "16"} and st["dependent_launch_floor_us"] > 0
    for k, item in st["per_call"].items():
        assert item["calls"] == -(-6 // int(k)) and 0 < item["frac"] < 1 and item["verified"]["points"] > 0 and item["hip_graph_rows_equal"], item
    fr = line["garden185"]["fused_refine"]          # the pipeline's own call (raw depth -> points in one kernel), bit-equal to refine + plain call
    assert fr["equals_refine_apply_then_plain"] and fr["ms"] > 0 and fr["points"] > 0 and 0 < fr["frac"] < 1, fr


def test_an_explicit_workload_prints_that_workload_only():
    line = _run([sys.executable, "bench.py", "--workload", "garden185", "--steps", "2", "--warmup", "1", "--views", "5", "--strong-views", "0",
                 "--cpu-seconds", "0", "--alloc-rounds", "0"])
    _check(line, 1, 2, 1)
    assert line["scaling"] == "weak" and line["config"]["workload"] == "garden185" and "roofline12mp" not in line and "strong2000" not in line
    assert set(line["streaming"]["per_call"]) == {"1", "2", "4", "8", "16"} and line["fused_refine"]["equals_refine_apply_then_plain"]


def test_two_ranks_sharing_the_gpu():
    """N > 1: the default workload is the north-star curve (2000-view scene, strong scaling; here 8 views over 2 ranks)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--views", "8", "--strong-views", "9",
           "--strong-steps", "1", "--chunks", "2", "--alloc-rounds", "0", "--n1-strong-mpix", "1000"]
    line = _run(cmd, {"DD_BENCH_SHARE_GPU": "1", "DD_ALLGATHERV": "broadcast"})
    _check(line, 2, 2, 1)
    assert line["config"]["workload"] == "scene2000" and line["scaling"] == "strong"
    assert line["config"]["views_total"] == 8 and line["config"]["views_per_gpu"] == 4 and "cpu_baseline" not in line
    assert len(line["devices"]) == 2 and all(d.startswith(f"rank {i}:") for i, d in enumerate(line["devices"])) and line["rccl_world_size"] == 2
    assert line["speedup_vs_n1"] == round(line["value"] / 1000, 3)
    assert line["verified"]["all_ranks_ok"], line["verified"]
    s = line["strong2000"]
    assert s["views_per_gpu"] in (4, 5) and all(s[k]["ms"] > 0 for k in ("sharded", "gathered", "gathered_compact")), s
    assert s["gathered"]["bytes_received_per_rank"] > 0 and s["verified"]["all_ranks_ok"]


def test_plain_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher in front of it (the shape of the driver's 1-GPU command) starts
    its two ranks itself and prints ONE line with n_gpus 2.  CPU twin of the launcher: tests/test_bench_launcher.py."""
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--views", "8", "--strong-views", "9",
           "--strong-steps", "1", "--chunks", "2", "--alloc-rounds", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run(cmd, cwd=ROOT, env=dict(env, DD_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[:500]
    line = json.loads(lines[0])
    _check(line, 2, 2, 1)
    assert line["config"]["workload"] == "scene2000" and line["scaling"] == "strong" and line["rccl_world_size"] == 2
    assert line["verified"]["all_ranks_ok"] and line["strong2000"]["verified"]["all_ranks_ok"]


def test_weak_scaling_variant_still_runs_on_two_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29657", "bench.py", "--gpus", "2", "--workload", "garden185", "--steps", "2", "--warmup", "1", "--views", "4",
           "--strong-views", "0", "--alloc-rounds", "0"]
    line = _run(cmd, {"DD_BENCH_SHARE_GPU": "1"})
    _check(line, 2, 2, 1)
    assert line["scaling"] == "weak" and line["config"]["views_total"] == 8 and line["verified"]["all_ranks_ok"]


def test_scene_set_workload_on_three_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", "29656", "bench.py", "--gpus", "3", "--workload", "mip360x7", "--views", "40", "--steps", "2", "--warmup", "1",
           "--strong-views", "0"]
    line = _run(cmd, {"DD_BENCH_SHARE_GPU": "1"})
    _check(line, 3, 2, 1)
    assert line["scaling"] == "strong" and line["config"]["workload"] == "mip360x7" and len(line["config"]["rank0_scenes"]) >= 2
