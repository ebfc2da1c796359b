#!/bin/bash
# Wave-level PMC counters of the floater-vote kernel (GPU box).  usage: tools/pmc_votes.sh <outdir> [views]
set -euo pipefail
OUT=$(realpath -m "$1"); V=${2:-48}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "floater_votes" --kernel-trace --output-format csv -d "$OUT/set$i" -- \
      python3 "$R/tools/bench_filter.py" --views "$V" > "$OUT/set$i.log" 2>&1 || echo "set$i failed"
done
python3 - "$OUT" <<'PY'
import csv, sys, json
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1]); res = defaultdict(list)
for f in out.rglob("*counter_collection.csv"):
    for r in csv.DictReader(f.open()):
        if "floater_votes" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open(out / "summary.json", "w"), indent=1)
print(json.dumps(avg, indent=1))
PY
rm -rf "$OUT"/set*/
