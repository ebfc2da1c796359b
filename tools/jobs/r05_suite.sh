# the whole GPU suite with its full log, then the refiner / pipeline soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite.log 2>&1; rc=$?
tail -3 gpurun_out/r05_gpu_suite.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_gpu_suite.log | head -20; exit 1; fi
DD_REFINE_SEEDS=2000 timeout -k 10 600 python -m pytest tests/test_refiner.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_soak_refine.log 2>&1; rc=$?
echo "DD_REFINE_SEEDS=2000: $(tail -1 gpurun_out/r05_soak_refine.log)"
exit $rc
