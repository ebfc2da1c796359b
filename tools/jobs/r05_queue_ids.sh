# chained one-view calls: with a graph captured before the builder's side streams are first used (slow: 26 us per call) and without (17 us):
# which hardware queues do the two side streams' kernels run on?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp DD_EXCLUSIVE_GPU=1
for v in graph plain; do
  fl=""; [ $v = graph ] && fl="--graph"
  rm -rf /tmp/q_$v
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/q_$v -- python3 "$GRAFT_REPO_ROOT/tools/bench_streaming.py" --per-call 1 $fl --only builder.append --rounds 5 > "$GRAFT_REPO_ROOT/gpurun_out/r05_q_$v.log" 2>&1) || { echo "profile $v failed"; tail -5 gpurun_out/r05_q_$v.log; exit 1; }
  echo "== $v"; grep -E "^k=" gpurun_out/r05_q_$v.log
  python3 tools/kernel_timeline.py /tmp/q_$v gpurun_out/r05_queue_ids_$v.txt 260352 | head -9
done
