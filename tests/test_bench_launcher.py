"""``python bench.py --gpus N`` launched plainly starts its own ranks (``bench.launch_ranks``): environment of every rank, one
shared stdout, the verdict of a failing rank -- on the CPU, with a stand-in program instead of the GPU job.  The GPU twin:
tests/test_bench_contract.py::test_plain_launch_starts_its_own_ranks."""

import json
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

RANK_PROGRAM = r"""
import json, os, sys, time
out, mode = sys.argv[1], sys.argv[2]
rank = int(os.environ["RANK"])
keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
json.dump({k: os.environ.get(k) for k in keys}, open(f"{out}/rank{rank}.json", "w"))
if mode == "fail" and rank == 1:
    sys.exit(3)
if mode == "fail":
    time.sleep(120)              # a rank stuck in a collective whose peer died: the launcher ends it
if rank == 0:
    print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"])}))
"""


def _launch(tmp_path, n, mode, grace="1.0"):
    prog = tmp_path / "rank_program.py"
    prog.write_text(RANK_PROGRAM)
    code = (f"import sys; sys.path.insert(0, {str(ROOT)!r}); import bench; "
            f"sys.exit(bench.launch_ranks({n}, [{str(prog)!r}, {str(tmp_path)!r}, {mode!r}], grace_s={grace}))")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return r, time.monotonic() - t0


def test_every_rank_gets_its_environment_and_rank_0_owns_stdout(tmp_path):
    r, _ = _launch(tmp_path, 3, "ok")
    assert r.returncode == 0, r.stderr[-1500:]
    assert [json.loads(l) for l in r.stdout.splitlines() if l.strip()] == [{"n_gpus": 3}]
    envs = [json.loads((tmp_path / f"rank{k}.json").read_text()) for k in range(3)]
    for k, e in enumerate(envs):
        assert e["RANK"] == e["LOCAL_RANK"] == str(k) and e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "3"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0


def test_a_failing_rank_fails_the_job_and_the_others_are_ended(tmp_path):
    r, took = _launch(tmp_path, 3, "fail")
    assert r.returncode == 3 and r.stdout.strip() == "", (r.returncode, r.stdout, r.stderr[-800:])
    assert took < 60 and "rank 1 exited with 3" in r.stderr


def test_plain_launch_without_a_gpu_reports_the_ranks_failure():
    """The real entry, here without a GPU: the parent starts the ranks, every rank says what it needs, the parent's code is theirs."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the CPU half of the launcher test")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs an MI355X" in r.stderr and "must be launched with" not in r.stderr


def test_a_signal_to_the_launcher_reaches_every_rank(tmp_path):
    """The driver ends a run that overstays with SIGTERM to the process it started: the ranks must go with it."""
    import os
    import signal
    prog = tmp_path / "rank_program.py"
    prog.write_text("import os, sys, time\n"
                    "open(f'{sys.argv[1]}/pid{os.environ[\"RANK\"]}', 'w').write(str(os.getpid()))\n"
                    "time.sleep(300)\n")
    code = (f"import sys; sys.path.insert(0, {str(ROOT)!r}); import bench; "
            f"sys.exit(bench.launch_ranks(2, [{str(prog)!r}, {str(tmp_path)!r}], grace_s=1.0))")
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.monotonic() + 60
    while time.monotonic() < deadline and not all((tmp_path / f"pid{k}").exists() and (tmp_path / f"pid{k}").read_text() for k in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / f"pid{k}").read_text()) for k in range(2)]
    p.send_signal(signal.SIGTERM)
    try:
        p.communicate(timeout=60)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode != 0
    time.sleep(0.5)
    for pid in pids:                       # gone (a zombie of another parent would still answer kill -0: the launcher reaped them)
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, pid
