#!/usr/bin/env python3
"""Timeline of a chain of calls that runs on more than one stream, from a rocprofv3 --kernel-trace CSV: for every dispatch of
the densify kernel on the grid that occurs most often, when it started and ended relative to the END of the previous one, how
long the two overlapped, and what the gate kernel in front of it did.  (What tools/kernel_gaps.py cannot say: it assumes one
queue.)
   usage: tools/kernel_timeline.py <rocprof output dir> [out.txt] [grid threads, default: the most frequent one]"""
import csv
import sys
from collections import Counter
from pathlib import Path

import numpy as np


def short(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(src: str, dst: str = "", want_grid: str = "") -> None:
    traces = sorted(Path(src).rglob("*_kernel_trace.csv"))
    if not traces:
        sys.exit(f"no *_kernel_trace.csv under {src}")
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]), r.get("Queue_Id", "?"))
            for r in csv.DictReader(traces[0].open())]
    rows.sort()
    mains = [r for r in rows if "compact_lean" in r[2]]
    if not mains:
        sys.exit("no compact_lean dispatch in the trace")
    top = Counter((r[3], r[2]) for r in mains).most_common(6)
    grid = int(want_grid) if want_grid else top[0][0][0]
    mains = [r for r in mains if r[3] == grid]
    gates = [r for r in rows if "chain_gate" in r[2]]
    start_after_prev_end, overlap, period, dur, queues = [], [], [], [], Counter()
    for prev, cur in zip(mains, mains[1:]):
        if cur[0] - prev[1] > 200_000:                    # another chain (a new round of the bench): not a neighbour
            continue
        start_after_prev_end.append((cur[0] - prev[1]) / 1e3)
        overlap.append(max(0, min(prev[1], cur[1]) - cur[0]) / 1e3)
        period.append((cur[1] - prev[1]) / 1e3)
        dur.append((cur[1] - cur[0]) / 1e3)
        queues[(prev[4], cur[4])] += 1
    gate_dur = [(g[1] - g[0]) / 1e3 for g in gates]
    # the gate that belongs to a main kernel: the last gate on the same queue that ended before the main kernel started
    gate_to_main = []
    for m in mains:
        mine = [g for g in gates if g[4] == m[4] and g[1] <= m[0]]
        if mine:
            gate_to_main.append((m[0] - mine[-1][1]) / 1e3)

    def q(v):
        return f"median {np.median(v):8.2f}  p10 {np.percentile(v, 10):8.2f}  p90 {np.percentile(v, 90):8.2f}" if len(v) else "-"
    lines = [f"# {traces[0].name}: compact_lean on grid {grid} ({len(mains)} dispatches), microseconds",
             f"kernel duration                         {q(dur)}",
             f"end-to-end period (end n -> end n+1)    {q(period)}",
             f"start of n+1 relative to the end of n   {q(start_after_prev_end)}   (negative: they overlap)",
             f"overlap of n and n+1                    {q(overlap)}",
             f"gate kernels: {len(gates)}, duration             {q(gate_dur)}",
             f"gate's end -> its main kernel's start   {q(gate_to_main)}",
             f"queues of (n, n+1): {dict(queues)}",
             "# most frequent grids: " + "; ".join(f"{g} x{n} {k[:60]}" for (g, k), n in top)]
    text = "\n".join(lines) + "\n"
    if dst:
        Path(dst).write_text(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
