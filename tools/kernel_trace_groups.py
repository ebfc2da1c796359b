#!/usr/bin/env python3
"""Per (kernel, grid size) statistics from a rocprofv3 --kernel-trace CSV: the bench's default run launches the hot kernel on
several workloads (the 185-view headline, one-view verification passes, the 2000-view leg), which `--stats` averages into one
row.   usage: tools/kernel_trace_groups.py <rocprof output dir> <out.csv> [note]"""
import csv
import sys
from collections import defaultdict
from pathlib import Path


def main(src: str, dst: str, note: str = "") -> None:
    traces = sorted(Path(src).rglob("*_kernel_trace.csv"))
    if not traces:
        sys.exit(f"no *_kernel_trace.csv under {src}")
    groups = defaultdict(list)
    for row in csv.DictReader(traces[0].open()):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not any(m in name for m in ("compact_lean", "count_lean", "zone_pair_store", "compact_generic")):
            continue
        groups[(name, int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace, dispatches grouped by kernel and grid ({traces[0].name})\n")
        if note:
            f.write(f"# {note}\n")
        f.write("name,grid_threads,workgroup,calls,avg_us,min_us,median_us,max_us\n")
        for (name, grid, wg), ts in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            ts = sorted(ts)
            f.write(f"\"{name}\",{grid},{wg},{len(ts)},{sum(ts) / len(ts):.2f},{ts[0]:.2f},{ts[len(ts) // 2]:.2f},{ts[-1]:.2f}\n")
    print(open(dst).read())


if __name__ == "__main__":
    main(*sys.argv[1:])
