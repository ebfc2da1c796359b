#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the short
summary we commit under profiles/ (our kernels in full, everything else aggregated)."""
import csv
import sys
from pathlib import Path


def main(src: str, dst: str, note: str = "") -> None:
    stats = sorted(Path(src).rglob("*_kernel_stats.csv"))
    if not stats:
        sys.exit(f"no *_kernel_stats.csv under {src}")
    rows = list(csv.DictReader(stats[0].open()))
    mine = ("compact_generic", "compact_lean", "count_lean", "count_generic", "scan_view_tiles", "scan_views",
            "floater_", "dd_", "compact_count", "compact_scan", "compact_scatter", "compact_view", "refine_apply")
    ours = [r for r in rows if any(m in r["Name"] for m in mine)]
    rest = [r for r in rows if r not in ours]
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({stats[0].name})\n")
        if note:
            f.write(f"# {note}\n")
        f.write("name,calls,avg_us,min_us,max_us,stddev_us,total_ms,pct\n")
        for r in ours:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0]
            f.write(f"\"{name}\",{r['Calls']},{float(r['AverageNs'])/1e3:.2f},{float(r['MinNs'])/1e3:.2f},"
                    f"{float(r['MaxNs'])/1e3:.2f},{float(r['StdDev'])/1e3:.2f},{float(r['TotalDurationNs'])/1e6:.3f},{r['Percentage']}\n")
        tot = sum(float(r["TotalDurationNs"]) for r in rest)
        calls = sum(int(r["Calls"]) for r in rest)
        f.write(f"\"(all other kernels: torch scene generation, memset/copy)\",{calls},,,,,{tot/1e6:.3f},"
                f"{sum(float(r['Percentage']) for r in rest):.2f}\n")
    print(open(dst).read())


if __name__ == "__main__":
    main(*sys.argv[1:])
