"""The small tools that condense rocprofv3 traces into the files under profiles/ (tools/kernel_timeline.py, tools/kernel_gaps.py), on
synthetic traces: what they report is what the numbers in profiles/r05_streaming_queue_collision.txt and r05_timeline_8views_gated.txt
are read from."""

import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEAD = "Kind,Agent_Id,Queue_Id,Kernel_Name,Grid_Size_X,Start_Timestamp,End_Timestamp\n"
NAME = '"void (anonymous namespace)::compact_lean<float, true, true, true, true, 12, false, 8>((anonymous namespace)::KArgs)"'


def _trace(tmp_path, rows):
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    (d / "123_kernel_trace.csv").write_text(HEAD + "".join(f"KERNEL_DISPATCH,1,{q},{name},{grid},{t0},{t1}\n" for q, name, grid, t0, t1 in rows))
    return tmp_path / "prof"


def test_timeline_of_a_chain_on_two_queues(tmp_path):
    # ten calls of 30 us alternating between queues 3 and 4, each starting 14 us before the previous one ends; one stray launch on another grid
    rows = [(3 + (i & 1), NAME, 260352, 1000 + 16000 * i, 1000 + 16000 * i + 30000) for i in range(10)]
    rows.append((1, NAME, 24012288, 10_000_000, 12_000_000))
    out = tmp_path / "t.txt"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_timeline.py"), str(_trace(tmp_path, rows)), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    assert "grid 260352 (10 dispatches)" in text
    line = {ln.split("  ")[0].strip(): ln for ln in text.splitlines()}
    assert "median    30.00" in line["kernel duration"] and "median    16.00" in line["end-to-end period (end n -> end n+1)"]
    assert "median   -14.00" in line["start of n+1 relative to the end of n"] and "median    14.00" in line["overlap of n and n+1"]
    assert "('3', '4'): 5" in text and "('4', '3'): 4" in text


def test_gaps_of_a_chain_on_one_queue(tmp_path):
    rows = [(1, NAME, 261120, 1000 + 25000 * i, 1000 + 25000 * i + 22000) for i in range(8)]          # 22 us kernels, 3 us apart
    out = tmp_path / "g.txt"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_gaps.py"), str(_trace(tmp_path, rows)), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    row = [ln for ln in out.read_text().splitlines() if "compact_lean" in ln][0].split(",")
    assert row[-10:] == ["261120", "7", "22.00", "22.00", "22.00", "7", "3.00", "3.00", "3.00", "25.00"]
