#!/bin/bash
# UTCL1 (per-CU address translation cache) hit / miss counters of the densify kernel over several re-allocations inside one
# process (tools/experiments/placement2.py): does the slow placement state show up as translation misses?   usage: tools/pmc_tlb.sh <outdir>
set -uo pipefail
OUT=$(realpath -m "$1"); R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_tlb; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE \
    --kernel-include-regex "compact_lean" --kernel-trace --output-format csv -d /tmp/prof_tlb -- python3 "$R/tools/experiments/placement2.py" > "$OUT/placement2_under_pmc.txt" 2>&1 || { echo failed; tail -5 "$OUT/placement2_under_pmc.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, sys, collections
from pathlib import Path
files = sorted(Path("/tmp/prof_tlb").rglob("*counter_collection.csv"))
rows = list(csv.DictReader(files[0].open()))
by = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    by.setdefault(k, {"start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])})[r["Counter_Name"]] = float(r["Counter_Value"])
out = Path(sys.argv[1]) / "tlb_per_dispatch.txt"
with out.open("w") as f:
    f.write("dispatch  kernel_us  utcl1_miss  utcl1_hit  utcl1_req  miss_rate  utcl2_busy/gui_active\n")
    for k, v in by.items():
        us = (v["end"] - v["start"]) / 1e3
        miss, hit, req = v.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0), v.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0), v.get("TCP_UTCL1_REQUEST_sum", 0)
        busy = v.get("GRBM_UTCL2_BUSY", 0) / max(v.get("GRBM_GUI_ACTIVE", 1), 1)
        f.write(f"{k:8d} {us:10.1f} {miss:12.0f} {hit:12.0f} {req:12.0f} {miss / max(miss + hit, 1):9.5f} {busy:8.3f}\n")
print(out.read_text()[:6000])
PY
