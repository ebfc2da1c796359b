# One parameterised job script for the GPU box (round 6; the 40 one-off r05_*.sh scripts are gone):   r06_job.sh <what> [args...]
#   suite [pytest args]     the GPU suite, log in gpurun_out/r06_suite.log
#   soak <launches> <reps>  the three-rank rehearsal (torchrun, gloo, p2p, views 2 / 4 in turn), <reps> executions per launch
#   bench [bench args]      python3 bench.py ... -> gpurun_out/r06_bench.json
#   py <script> [args]      python3 <script> ... -> gpurun_out/r06_<script name>.log
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
*) echo "unknown job $what"; exit 2 ;;
esac
