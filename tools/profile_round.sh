#!/bin/bash
# One profiling sweep of the hot kernel over the bench workloads (GPU box): for each workload
#   1. bench.py line (un-profiled)                                  -> <out>/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command          -> <out>/<tag>/stats/   (+ bench line under the profiler)
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs -> <out>/<tag>/pmc/
# then tools/profile_collect.py condenses them into profiles/ (kernel-stats CSVs, traffic.json).
#   usage: tools/profile_round.sh <outdir> [workload tags...]      tags: garden185 bernoulli mip360conf mip360conf_smooth roofline12mp scene2000
set -uo pipefail
OUT=$(realpath -m "$1"); shift          # (raw traces are large: give a directory under /tmp, not under gpurun_out/ -- only 64 MiB travel back)
R=$(cd "$(dirname "$0")/.." && pwd)
TAGS=${@:-garden185 bernoulli mip360conf roofline12mp}
export TMPDIR=/tmp
cd /tmp
for tag in $TAGS; do
  case $tag in
    bernoulli) ARGS="--workload garden185 --mask-kind bernoulli" ;;
    mip360conf_smooth) ARGS="--workload mip360conf --conf-kind smooth" ;;
    *) ARGS="--workload $tag" ;;
  esac
  case $tag in      # (the streaming chains of garden185 are hundreds of small launches of the same kernel: they would be averaged into its counters)
    garden185|bernoulli) ARGS="$ARGS --streaming=" ;;
  esac
  D="$OUT/$tag"; mkdir -p "$D"
  python3 "$R/bench.py" $ARGS --cpu-seconds 0 --strong-views 0 > "$D/bench.json" 2> "$D/bench.err" || { echo "$tag: bench failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -- \
      python3 "$R/bench.py" $ARGS --cpu-seconds 0 --strong-views 0 --no-verify --alloc-rounds 0 > "$D/bench_under_rocprof.json" 2> "$D/stats.err" || { echo "$tag: stats failed"; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "compact_lean|count_lean" --kernel-trace --output-format csv -d "$D/pmc/$c" -- \
        python3 "$R/bench.py" $ARGS --steps 3 --warmup 1 --cpu-seconds 0 --strong-views 0 --no-verify --alloc-rounds 0 > "$D/pmc_$c.json" 2> "$D/pmc_$c.err" || { echo "$tag: pmc $c failed"; exit 1; }
  done
  echo "$tag done"
done
