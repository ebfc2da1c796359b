#!/bin/bash
# Everything that needs MORE THAN ONE GPU, in one go (a box with >= 2 MI355X; the one-GPU boxes of this build never ran it):
#   1. the two RCCL tests the one-GPU suite skips (tests/test_fuse_gpu.py: the Python exchange and the C-ABI dd_allgatherv);
#   2. bench.py at N = 1 (the strong2000 reference figure), then N in {2,4,8} (as many as the box has): the default line (2000-view
#      scene, strong scaling, chunks 5, grouped send/recv, every rank receives) and the sweep chunks {1,5,10} x DD_ALLGATHERV
#      {p2p,broadcast} x gather-dst {all,0};
#   3. one rocprofv3 --kernel-trace --memory-copy-trace of rank 0 at the largest N (the program directly after `--`).
# Results: profiles/$TAG_scale_*.json (one bench line each), profiles/$TAG_scale_summary.txt, profiles/$TAG_scale_rank0_trace_*.csv (TAG defaults to r06).
#   usage: tools/run_multi_gpu.sh [outdir]          (run from the repo root)
set -uo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "${1:-$R/profiles}")
mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU"
# REHEARSE=1: the script's own plumbing on a ONE-GPU box -- two ranks share the GPU over gloo, tiny workloads, no RCCL tests,
# no profiler; the numbers mean nothing, the point is that every step runs and every line parses
REHEARSE=${REHEARSE:-0}
SMALL=""
if [ "$REHEARSE" = 1 ]; then
  NGPU=2; export DD_BENCH_SHARE_GPU=1        # (gloo has no device send / recv: the p2p flavour runs its schedule staged through host memory)
  SMALL="--views 8 --strong-views 8 --strong-steps 1 --steps 2 --warmup 1"
fi
if [ "$NGPU" -lt 2 ]; then echo "needs at least 2 GPUs"; exit 2; fi
TAG=${TAG:-r06}
SUM="$OUT/${TAG}_scale_summary.txt"
: > "$SUM"

echo "== 1. RCCL tests" | tee -a "$SUM"
if [ "$REHEARSE" = 1 ]; then echo "(rehearsal: skipped)" | tee -a "$SUM"; else
timeout -k 10 600 python3 -m pytest tests/test_fuse_gpu.py -q -m gpu -k "two_ranks_over_rccl or c_abi_allgatherv_over_rccl" 2>&1 | tail -5 | tee -a "$SUM"
fi

line() {   # line <tag> <N> <env assignments...> -- <bench args...>
  local tag=$1 n=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local port=$((29800 + RANDOM % 150))
  # one entry at every N: bench.py starts its own ranks when it is launched plainly (bench.launch_ranks) -- the driver's command shape
  env "${envs[@]}" MASTER_PORT="$port" timeout -k 10 900 python3 bench.py --gpus "$n" "$@" > "$OUT/${TAG}_scale_$tag.json" 2> "$OUT/${TAG}_scale_$tag.err"
  local rc=$?
  python3 - "$OUT/${TAG}_scale_$tag.json" "$tag" "$rc" <<'PY' | tee -a "$SUM"
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(f"{sys.argv[2]}: rc {sys.argv[3]}, no line ({e})"); sys.exit(0)
s = d.get("strong2000") or {}
g = lambda k: (s.get(k) or {}).get("ms")
print(f"{sys.argv[2]}: rc {sys.argv[3]}  N={d['n_gpus']} {d['config']['workload']} {d['value']} Mpix/s ({d['ms_per_step']} ms)  speedup_vs_n1 {d.get('speedup_vs_n1')}  "
      f"strong2000 sharded/gathered/compact ms {g('sharded')}/{g('gathered')}/{g('gathered_compact')}  "
      f"GB/s per link {(s.get('gathered') or {}).get('GBps_per_link')}/{(s.get('gathered_compact') or {}).get('GBps_per_link')}  verified {(d.get('verified') or {}).get('all_ranks_ok')}  devices {len(d.get('devices') or [1])}")
PY
}

echo "== 2. bench lines" | tee -a "$SUM"
line n1 1 -- --workload scene2000 --cpu-seconds 0 --alloc-rounds 0 $SMALL
N1=$(python3 -c "import json; print(json.load(open('$OUT/${TAG}_scale_n1.json'))['value'])" 2>/dev/null || echo 0)
NS=""; for n in 2 4 8; do [ "$n" -le "$NGPU" ] && NS="$NS $n"; done
for n in $NS; do
  line "n${n}_default" "$n" -- --n1-strong-mpix "$N1" $SMALL
  line "n${n}_placed" "$n" "DD_FUSE_PLACEMENT=probed" -- --n1-strong-mpix "$N1" --steps 5 --warmup 2 --alloc-rounds 0 $SMALL     # the gathered legs with arena-placed global arrays
  for chunks in 1 5 10; do for ag in p2p broadcast; do for dst in all 0; do
    [ "$ag" = broadcast ] && [ "$dst" = 0 ] && continue        # the broadcast flavour replicates by construction
    line "n${n}_c${chunks}_${ag}_dst${dst}" "$n" "DD_ALLGATHERV=$ag" -- --steps 5 --warmup 2 --chunks "$chunks" --gather-dst "$dst" --n1-strong-mpix "$N1" --alloc-rounds 0 $SMALL
  done; done; done
done

if [ "$REHEARSE" = 1 ]; then echo "== 3. (rehearsal: no profiler run)" | tee -a "$SUM"; echo "done; summary in $SUM"; exit 0; fi
echo "== 3. rank-0 trace at N = $(echo $NS | awk '{print $NF}')" | tee -a "$SUM"
NMAX=$(echo $NS | awk '{print $NF}')
export TMPDIR=/tmp
# rank 0 runs under the profiler (the program directly after `--`), the other ranks plainly: one launcher per rank, no re-exec
PORT=$((29950 + RANDOM % 40))
for r in $(seq 1 $((NMAX - 1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$NMAX MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT timeout -k 10 600 python3 bench.py --gpus "$NMAX" --steps 5 --warmup 2 --alloc-rounds 0 > /dev/null 2> "$OUT/${TAG}_scale_trace_rank$r.err" &
done
( cd /tmp && RANK=0 LOCAL_RANK=0 WORLD_SIZE=$NMAX MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv \
    -d /tmp/${TAG}_scale_trace -- python3 "$R/bench.py" --gpus "$NMAX" --steps 5 --warmup 2 --alloc-rounds 0 > "$OUT/${TAG}_scale_trace_rank0.json" 2> "$OUT/${TAG}_scale_trace_rank0.err" )
wait
for f in $(find /tmp/${TAG}_scale_trace -name "*_kernel_stats.csv" -o -name "*_memory_copy_stats.csv" 2>/dev/null); do cp "$f" "$OUT/${TAG}_scale_rank0_trace_$(basename "$f" | sed 's/^[0-9]*_//')"; done
python3 tools/kernel_trace_groups.py /tmp/${TAG}_scale_trace "$OUT/${TAG}_scale_rank0_kernel_groups.csv" "rank 0 of $NMAX, bench.py --gpus $NMAX --steps 5 --warmup 2" > /dev/null 2>&1 || true
echo "done; summary in $SUM"
