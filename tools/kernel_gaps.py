#!/usr/bin/env python3
"""Chains of one kernel in a rocprofv3 --kernel-trace CSV: per (kernel, grid) the kernel's duration AND the gap between the
end of one dispatch and the start of the next in the chain -- what a streaming caller pays per call besides the kernel.
   usage: tools/kernel_gaps.py <rocprof output dir> [out.txt]"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

import numpy as np


def main(src: str, dst: str = "") -> None:
    traces = sorted(Path(src).rglob("*_kernel_trace.csv"))
    if not traces:
        sys.exit(f"no *_kernel_trace.csv under {src}")
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                   r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], int(r["Grid_Size_X"]))
                  for r in csv.DictReader(traces[0].open()))
    dur, gap = defaultdict(list), defaultdict(list)
    for prev, cur in zip(rows, rows[1:]):
        key = (cur[2], cur[3])
        dur[key].append((cur[1] - cur[0]) / 1e3)
        if (prev[2], prev[3]) == key:                      # the same kernel on the same grid back to back: a chain
            gap[key].append((cur[0] - prev[1]) / 1e3)
    lines = [f"# {traces[0].name}: kernel duration and the gap to the previous dispatch of the same chain, microseconds",
             "kernel,grid_threads,calls,dur_median,dur_min,dur_p90,chained,gap_median,gap_min,gap_p90,period_median"]
    for key, d in sorted(dur.items(), key=lambda kv: -len(kv[1])):
        if not any(m in key[0] for m in ("compact_lean", "count_lean", "compact_generic", "plan_dense", "scan_")):
            continue
        g = gap.get(key, [])
        gm = f"{np.median(g):.2f},{min(g):.2f},{np.percentile(g, 90):.2f},{np.median(d) + np.median(g):.2f}" if g else ",,,"
        lines.append(f"\"{key[0]}\",{key[1]},{len(d)},{np.median(d):.2f},{min(d):.2f},{np.percentile(d, 90):.2f},{len(g)},{gm}")
    text = "\n".join(lines) + "\n"
    if dst:
        Path(dst).write_text(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
