#!/bin/bash
# The driver's exact bench command under rocprofv3 --kernel-trace --stats in N fresh processes (GPU box), condensed per
# (kernel, grid) into profiles/<prefix>_run<k>_kernel_groups.csv + the bench line of each run.
#   usage: tools/profile_driver_cmd.sh <outdir> <prefix> [runs] [extra bench args...]
set -uo pipefail
OUT=$(realpath -m "$1"); PFX=$2; N=${3:-3}; shift 3 || shift $#
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for k in $(seq 1 "$N"); do
  D="$OUT/${PFX}_run$k"; mkdir -p "$D"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/prof" -- \
      python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 "$@" > "$D/bench.json" 2> "$D/prof.err" || { echo "run $k failed"; grep -v "^[EWI]20[0-9][0-9]" "$D/prof.err" | tail -25; rm -rf "$D/prof"; exit 1; }
  python3 "$R/tools/kernel_trace_groups.py" "$D/prof" "$D/kernel_groups.csv" "fresh process $k of: python3 bench.py --gpus 1 --steps 20 --warmup 5 $*" > /dev/null
  python3 "$R/tools/summarize_prof.py" "$D/prof" "$D/kernel_stats.csv" "fresh process $k (all grids of a kernel in one row)" > /dev/null
  head -8 "$D/kernel_groups.csv"
  python3 "$R/tools/show_bench.py" "$D/bench.json" | head -3
  rm -rf "$D/prof"
done
