#!/bin/bash
# rocprofv3 kernel statistics of the neighbouring stages (GPU box): the fused refine + densify kernel against the
# two-kernel path (tools/bench_fused_refine.py) and the floater-vote kernels (tools/bench_filter.py).
#   usage: tools/profile_side.sh <round tag, e.g. r02>      -> gpurun_out/profiles_side/<tag>_{refine,filter}_kernel_stats.csv
set -uo pipefail
TAG=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT="$R/gpurun_out/profiles_side"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for what in refine filter; do
  case $what in
    refine) PROG="$R/tools/bench_fused_refine.py" ;;
    filter) PROG="$R/tools/bench_filter.py" ;;
  esac
  RAW=/tmp/dd_side_$what; rm -rf "$RAW"
  python3 "$PROG" > "$OUT/${TAG}_${what}_bench.txt" 2> "$OUT/${what}.err" || { echo "$what: bench failed"; tail -5 "$OUT/${what}.err"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW" -- python3 "$PROG" > "$OUT/${what}_under_rocprof.txt" 2> "$OUT/${what}_stats.err" \
      || { echo "$what: stats failed"; tail -5 "$OUT/${what}_stats.err"; exit 1; }
  python3 -c "import sys; sys.path.insert(0, '$R/tools'); import summarize_prof; summarize_prof.main('$RAW', '$OUT/${TAG}_${what}_kernel_stats.csv', '$TAG: rocprofv3 --kernel-trace --stats -- python3 tools/$(basename $PROG)')" || exit 1
  echo "$what done"
done
