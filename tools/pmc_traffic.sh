#!/bin/bash
# HBM traffic of the hot kernels from rocprofv3 PMC counters (GPU box only).
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots): one rocprofv3 run per counter,
# with --kernel-trace only (no other trace domains), as MI355X_MICROARCH.md prescribes.
#   usage: tools/pmc_traffic.sh <outdir> [bench args...]
set -euo pipefail
OUT=$(realpath -m "$1"); shift
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-include-regex "compact_lean|count_lean|scan_view|compact_generic" --kernel-trace --output-format csv -d "$OUT/$c" -- \
      python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --strong-views 0 "$@" > "$OUT/$c.log" 2>&1
  echo "$c rc=$?"
done
