# One parameterised job script for the GPU box (round 6; the 40 one-off r05_*.sh scripts are gone):   r06_job.sh <what> [args...]
#   suite [pytest args]     the GPU suite, log in gpurun_out/r06_suite.log
#   soak <launches> <reps>  the three-rank rehearsal (torchrun, gloo, p2p, views 2 / 4 in turn), <reps> executions per launch
#   bench [bench args]      python3 bench.py ... -> gpurun_out/r06_bench.json
#   py <script> [args]      python3 <script> ... -> gpurun_out/r06_<script name>.log
#   sweeps                  the seeded parity sweeps at soak sizes -> gpurun_out/r06_soak.txt
#   driver_cmd [runs]       the driver's bench command under rocprofv3 --kernel-trace --stats in fresh processes -> gpurun_out/r06_driver_cmd/
#   traffic_refine          FETCH_SIZE / WRITE_SIZE (separate passes) of the fused refine kernel -> gpurun_out/r06_traffic_refine.json
#   pmc_refine              SQ counters (two passes) of the fused refine kernel beside the plain kernel -> gpurun_out/r06_refine_pmc.json
#   prof <tag> <cmd...>     rocprofv3 --kernel-trace --stats of a python3 command -> gpurun_out/r06_prof_<tag>_*.csv
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; what=$1; shift
case "$what" in
suite)
  timeout -k 10 ${SUITE_SECONDS:-1000} python3 -m pytest tests -x -q -m gpu "$@" > gpurun_out/r06_suite.log 2>&1; rc=$?
  tail -n 6 gpurun_out/r06_suite.log; exit $rc ;;
soak)
  launches=$1; reps=$2; out=gpurun_out/r06_three_rank_soak.txt; t0=$(date +%s); done_=0
  echo "three-rank rehearsal soak: torchrun, 3 ranks sharing the GPU, gloo, p2p; $launches launches x $reps executions of the whole exchange (tests/fuse_worker.py, DD_FUSE_REPEAT)" > $out
  for i in $(seq 1 $launches); do
    v=$(( (i % 2) * 2 + 2 ))
    DD_DIST_BACKEND=gloo DD_ALLGATHERV=p2p DD_SHARE_GPU=1 DD_FUSE_VIEWS=$v DD_FUSE_REPEAT=$reps timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 \
      --nproc-per-node 3 --master-addr 127.0.0.1 --master-port $((20000 + (RANDOM % 20000))) tests/fuse_worker.py > /tmp/soak.out 2> /tmp/soak.err; rc=$?
    if [ $rc -ne 0 ]; then
      if grep -q -E "EADDRINUSE|Connection refused|Connection reset|Rendezvous|connectFullMesh" /tmp/soak.err && ! grep -q "Memory access fault" /tmp/soak.err; then echo "launch $i: rendezvous trouble, not counted" >> $out; continue; fi
      echo "launch $i (views $v): rc $rc after $done_ clean executions" | tee -a $out; grep -E "fault|Error|error|stage" /tmp/soak.err | tail -20 | tee -a $out; exit 1
    fi
    [ $(grep -o ": ok," /tmp/soak.out | wc -l) -eq 3 ] || { echo "launch $i: a rank did not report ok" | tee -a $out; cat /tmp/soak.out | tail -5 | tee -a $out; grep -v amdgpu.ids /tmp/soak.err | tail -12 | tee -a $out; exit 1; }
    done_=$((done_ + reps))
    [ $((i % 10)) -eq 0 ] && echo "$i launches, $done_ executions clean, $(( $(date +%s) - t0 )) s" | tee -a $out
  done
  echo "$done_ consecutive three-rank executions clean ($launches launches, $(( $(date +%s) - t0 )) s); no retry anywhere" | tee -a $out ;;
bench)
  timeout -k 10 ${BENCH_SECONDS:-600} python3 bench.py "$@" > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; rc=$?
  tail -c 600 gpurun_out/r06_bench.err; python3 tools/show_bench.py gpurun_out/r06_bench.json 2>/dev/null | head -60; exit $rc ;;
py)
  s=$1; shift; timeout -k 10 ${PY_SECONDS:-600} python3 $s "$@" > gpurun_out/r06_$(basename $s .py).log 2>&1; rc=$?
  tail -n ${TAIL:-40} gpurun_out/r06_$(basename $s .py).log; exit $rc ;;
sweeps)
  out=gpurun_out/r06_soak.txt; : > $out
  run() { local envs="$1"; shift; local t0=$(date +%s); env $envs timeout -k 10 900 python3 -m pytest "$@" -x -q -m gpu > /tmp/sweep.log 2>&1; local rc=$?
          printf "%-44s %s: %s\n" "$envs" "$*" "$(tail -n 1 /tmp/sweep.log)" | tee -a $out; [ $rc -eq 0 ] || { tail -n 30 /tmp/sweep.log; exit 1; }; }
  run "DD_APPLY_SEEDS=1500 DD_REFINE_SEEDS=3000" tests/test_refiner.py
  run "DD_RANDOM_SEEDS=12000" tests/test_gpu_random.py
  run "DD_RANDOM_SEEDS=1500 DD_RANDOM_SCALE=6" tests/test_gpu_random.py
  run "DD_STREAM_SEEDS=2000" tests/test_streaming_calls.py -k random_chains
  run "DD_VOTE_SEEDS=400" tests/test_filter.py
  if [ "${LONG:-0}" = 1 ]; then      # the protocols that depend on timing (scan service, chained calls), at length
    run "DD_STREAM_SEEDS=6000" tests/test_streaming_calls.py -k random_chains
    run "DD_RANDOM_SEEDS=60000" tests/test_gpu_random.py
  fi
  ;;
driver_cmd)
  bash tools/profile_driver_cmd.sh gpurun_out/r06_driver_cmd r06_driver_cmd ${1:-2} ;;
traffic_refine)
  export TMPDIR=/tmp DD_EXCLUSIVE_GPU=1
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/tr_$c
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "compact_lean" --kernel-trace --output-format csv -d /tmp/tr_$c -- python3 "$GRAFT_REPO_ROOT/tools/bench_fused_refine.py" --views 185 > "$GRAFT_REPO_ROOT/gpurun_out/r06_traffic_refine_$c.log" 2>&1) || { echo "$c failed"; tail -5 gpurun_out/r06_traffic_refine_$c.log; exit 1; }
  done
  python3 - <<'P'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/tr_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "compact_lean" not in k: continue
            tag = "fused_refine" if ("true, 16>" in k or "ELb1ELi16" in k) else "plain"
            acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
px = 185 * 1080 * 1920
for t, d in acc.items():
    fetch = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    rd, wr = fetch * 1024 * 2, write * 1024          # gfx950: FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md; verified on count_lean in round 1)
    res[t] = {"FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
              "bytes_per_pixel": round((rd + wr) / px, 2), "read_per_pixel": round(rd / px, 2), "write_per_pixel": round(wr / px, 2), "dispatches": [len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])]}
json.dump(res, open("gpurun_out/r06_traffic_refine.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
  ;;
pmc_refine)
  export TMPDIR=/tmp DD_EXCLUSIVE_GPU=1
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1)); rm -rf /tmp/rp_$i
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "compact_lean" --kernel-trace --output-format csv -d /tmp/rp_$i -- python3 "$GRAFT_REPO_ROOT/tools/bench_fused_refine.py" --views 185 > "$GRAFT_REPO_ROOT/gpurun_out/r06_refine_pmc_$i.log" 2>&1) || { echo "set $i failed"; tail -5 gpurun_out/r06_refine_pmc_$i.log; exit 1; }
  done
  python3 - <<'P'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"/tmp/rp_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "compact_lean" not in k: continue
            tag = "fused_refine" if ("true, 16>" in k or "ELb1ELi16" in k) else "plain"
            acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {t: {c: sum(v) / len(v) for c, v in d.items()} for t, d in acc.items()}
for t, d in res.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        d["share_wait_any"] = round(d["SQ_WAIT_ANY"] / wc, 3); d["share_wait_inst"] = round(d["SQ_WAIT_INST_ANY"] / wc, 3); d["share_active"] = round(d["SQ_ACTIVE_INST_ANY"] / wc, 3)
    if d.get("SQ_BUSY_CYCLES"):
        d["valu_busy_per_simd"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (d["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
    d["valu_per_pixel"] = round(d.get("SQ_INSTS_VALU", 0) * 64 / (185 * 1080 * 1920), 1)
json.dump(res, open("gpurun_out/r06_refine_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
  ;;
prof)
  tag=$1; shift; export TMPDIR=/tmp; rm -rf /tmp/prof_$tag
  (cd /tmp && timeout -k 10 ${PROF_SECONDS:-600} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 "$@" > "$GRAFT_REPO_ROOT/gpurun_out/r06_prof_$tag.out" 2> "$GRAFT_REPO_ROOT/gpurun_out/r06_prof_$tag.err"); rc=$?
  for f in $(find /tmp/prof_$tag -name "*_kernel_stats.csv" 2>/dev/null); do cp "$f" "gpurun_out/r06_prof_${tag}_kernel_stats.csv"; done
  python3 tools/kernel_trace_groups.py /tmp/prof_$tag "gpurun_out/r06_prof_${tag}_kernel_groups.csv" "$tag: python3 $*" 2>/dev/null | tail -n 12
  exit $rc ;;
*) echo "unknown job $what"; exit 2 ;;
esac
