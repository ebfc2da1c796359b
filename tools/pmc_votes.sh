#!/bin/bash
# Wave-level PMC counters of the floater-vote kernels (GPU box), summarised per kernel.
#   usage: tools/pmc_votes.sh <outdir> [views] [extra bench_filter.py arguments, e.g. --normals smooth --modes float64,float64_cull]
set -euo pipefail
OUT=$(realpath -m "$1"); V=${2:-48}; shift; shift || true; EXTRA="$*"
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "floater_votes" --kernel-trace --output-format csv -d "$OUT/set$i" -- \
      python3 "$R/tools/bench_filter.py" --views "$V" $EXTRA > "$OUT/set$i.log" 2>&1 || echo "set$i failed"
done
PMC_VOTES_VIEWS=$V python3 - "$OUT" <<'PY'
import csv, sys, json, re
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1]); res = defaultdict(lambda: defaultdict(list))
for f in out.rglob("*counter_collection.csv"):
    for r in csv.DictReader(f.open()):
        if "floater_votes" in r["Kernel_Name"]:
            res[re.search(r"floater_votes\w*(<[^>]*>)?", r["Kernel_Name"]).group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {kern: {k: sum(v) / len(v) for k, v in cs.items()} for kern, cs in res.items()}
for kern, a in avg.items():
    if a.get("SQ_WAVES") and a.get("SQ_INSTS_VALU"):
        a["valu_insts_per_wave"] = a["SQ_INSTS_VALU"] / a["SQ_WAVES"]
    if a.get("SQ_ACTIVE_INST_VALU") and a.get("SQ_BUSY_CYCLES"):
        a["valu_active_per_sq_busy_cycle"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_BUSY_CYCLES"]
    if a.get("SQ_THREAD_CYCLES_VALU") and a.get("SQ_ACTIVE_INST_VALU"):
        a["lanes_active_fraction"] = a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"] / 64
import os
json.dump({"views": int(os.environ.get("PMC_VOTES_VIEWS", "48")), "counters": avg}, open(out / "summary.json", "w"), indent=1)
print(json.dumps(avg, indent=1))
PY
rm -rf "$OUT"/set*/
