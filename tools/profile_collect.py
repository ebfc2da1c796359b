#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into profiles/: one kernel-stats CSV and one bench line per
workload (named per round) and the HBM-traffic entries of profiles/traffic.json.

    python tools/profile_collect.py <raw dir> r02 [<dest dir, default profiles/>]

(on the GPU box the raw rocprofv3 output stays in /tmp; only this condensed form travels back under gpurun_out/)

FETCH_SIZE is doubled (gfx950 tallies the 128-byte read requests of a wide streaming read at 64 B --
MI355X_MICROARCH.md, HBM; verified in round 1 on count_lean, whose bytes are known exactly); WRITE_SIZE is taken as is.
"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import summarize_prof  # noqa: E402


def mean_counter(folder: Path, counter: str, kernel: str = "compact_lean"):
    files = sorted(folder.rglob("*counter_collection.csv"))
    if not files:
        return None, 0
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(files[0].open()) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(src: str, rnd: str, dest: str = "") -> None:
    src = Path(src)
    prof = Path(dest) if dest else ROOT / "profiles"
    prof.mkdir(parents=True, exist_ok=True)
    tfile = prof / "traffic.json"
    base = ROOT / "profiles" / "traffic.json"
    traffic = json.loads(tfile.read_text()) if tfile.exists() else (json.loads(base.read_text()) if base.exists() else {})
    for d in sorted(p for p in src.iterdir() if p.is_dir() and (p / "bench.json").exists()):
        tag = d.name
        line = json.loads((d / "bench.json").read_text())
        (prof / f"{rnd}_bench_{tag}.json").write_text(json.dumps(line, indent=1) + "\n")
        under = d / "bench_under_rocprof.json"
        if under.exists() and under.stat().st_size:
            (prof / f"{rnd}_bench_{tag}_under_rocprof.json").write_text(json.dumps(json.loads(under.read_text()), indent=1) + "\n")
        if (d / "stats").is_dir():
            summarize_prof.main(str(d / "stats"), str(prof / f"{rnd}_{tag}_kernel_stats.csv"),
                                f"{rnd}: rocprofv3 --kernel-trace --stats -- python3 bench.py { {'bernoulli': '--workload garden185 --mask-kind bernoulli', 'mip360conf_smooth': '--workload mip360conf --conf-kind smooth'}.get(tag, '--workload ' + tag) } --cpu-seconds 0 --strong-views 0")
        fetch, nf = mean_counter(d / "pmc" / "FETCH_SIZE", "FETCH_SIZE")
        write, nw = mean_counter(d / "pmc" / "WRITE_SIZE", "WRITE_SIZE")
        if fetch is None or write is None:
            print(f"{tag}: no PMC data")
            continue
        cfg = line["config"]
        rb, wb = fetch * 1024 * 2, write * 1024
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        key = {"bernoulli": "garden185:bernoulli", "mip360conf_smooth": "mip360conf:smooth"}.get(tag, tag)
        if "dd_scatter" in line["roofline"]["kernel"]:      # the builder chose plan + scatter for this cloud: bench.py looks the scatter up under this key
            key += ":two-pass"
        traffic[key] = {
            "views": cfg["views_per_gpu"], "hbm_bytes_per_launch": int(rb + wb), "kernel": line["roofline"]["kernel"],
            "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "read_bytes_corrected": int(rb), "write_bytes": int(wb),
            "dispatches_averaged": [nf, nw], "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rb + wb) / alg, 4),
            "valid_fraction": cfg["valid_fraction"],
            "correction": "FETCH_SIZE*1024*2 (gfx950 tallies 128-B read requests at 64 B; verified on count_lean in round 1), WRITE_SIZE*1024 as is",
            "source": f"{rnd}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs with --kernel-trace only (tools/profile_round.sh)",
        }
        print(f"{key}: read {rb / 1e9:.3f} GB + write {wb / 1e9:.3f} GB = {(rb + wb) / 1e9:.3f} GB vs algorithmic {alg / 1e9:.3f} GB -> x{(rb + wb) / alg:.3f}")
    tfile.write_text(json.dumps(traffic, indent=1) + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
