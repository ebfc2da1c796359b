#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE per dispatch of our kernels from tools/pmc_traffic.sh output."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

MINE = ("compact_lean", "count_lean", "unproject_compact_kernel", "count_valid_kernel", "scan_view", "scan_views",
        "vectorized_elementwise_kernel")


def main(src, workload="garden185", dst=None, alg_bytes=None):
    src = Path(src)
    res = defaultdict(dict)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted((src / counter).rglob("*counter_collection.csv"))
        if not files:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(files[0].open()):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            key = next((m for m in MINE if m in name), None)
            if key:
                acc[key].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k][counter] = sum(v) / len(v)
            res[k]["dispatches"] = len(v)
    print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    main(*sys.argv[1:])
