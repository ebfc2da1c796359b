# the chained one-view calls with the fork skipped when the caller's stream is idle; then the whole GPU suite with its full log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export DD_EXCLUSIVE_GPU=1
timeout -k 10 300 python3 tools/bench_streaming.py --per-call 1,2,4,8 --graph --rounds 7 > gpurun_out/r05_fork_query_streaming.log 2>&1; rc=$?
grep -E "^k=|^#" gpurun_out/r05_fork_query_streaming.log
if [ $rc -ge 124 ]; then echo "killed ($rc)"; exit 1; fi
unset DD_EXCLUSIVE_GPU
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite.log 2>&1; rc=$?
tail -5 gpurun_out/r05_gpu_suite.log
exit $rc
