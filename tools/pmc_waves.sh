#!/bin/bash
# Wave-level PMC counters of the hot kernel (GPU box).  usage: tools/pmc_waves.sh <outdir> [bench args]
set -euo pipefail
OUT=$(realpath -m "$1"); shift
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "compact_lean|count_lean" --kernel-trace --output-format csv -d "$OUT/set$i" -- \
      python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 --strong-views 0 "$@" > "$OUT/set$i.log" 2>&1 || echo "set$i failed"
done
python3 - "$OUT" <<'PY'
import csv, sys, json
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1]); res = defaultdict(list)
for f in out.rglob("*counter_collection.csv"):
    for r in csv.DictReader(f.open()):
        if "compact_lean" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open(out / "summary.json", "w"), indent=1)
print(json.dumps(avg, indent=1))
PY
rm -rf "$OUT"/set*/
