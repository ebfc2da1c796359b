#!/bin/bash
# Counters the round-3 verdict asked for, list path (tuning 0) against dense path (tuning 128) on BASELINE configs[4] (GPU box):
# VALU instructions, LDS bank conflicts, waits, and the L2's write requests to the fabric (all sizes / 64-byte ones).
#   usage: tools/pmc_dense.sh <outdir> [bench args, e.g. --workload roofline12mp --views 100]
set -uo pipefail
OUT=$(realpath -m "$1"); shift
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
rocprofv3 -L 2>/dev/null | grep -io "TCC_EA0_WRREQ[A-Za-z0-9_]*" | sort -u > "$OUT/wrreq_counters_available.txt"
for tun in 0 128; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "WRITE_SIZE" "FETCH_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "compact_lean" --kernel-trace --output-format csv -d "$OUT/t${tun}_set$i" -- \
        python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 --strong-views 0 --alloc-rounds 0 --no-verify --tuning $tun "$@" > "$OUT/t${tun}_set$i.log" 2>&1 || echo "tuning $tun set$i failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, sys, json
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1]); res = {}
for tun in (0, 128):
    acc = defaultdict(list)
    for f in out.glob(f"t{tun}_set*/**/*counter_collection.csv"):
        for r in csv.DictReader(f.open()):
            if "compact_lean" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[f"tuning_{tun}"] = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open(out / "summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/t*_set*/
