# the whole GPU suite with its full log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -W always::UserWarning > gpurun_out/r05_gpu_suite.log 2>&1; rc=$?
tail -3 gpurun_out/r05_gpu_suite.log; grep -n "retried after" gpurun_out/r05_gpu_suite.log | head -3
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_gpu_suite.log | head -20; fi
exit $rc
