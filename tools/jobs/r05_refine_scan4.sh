# the look-up's range scanned four knots per step instead of bisected: same bits? how fast?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_APPLY_SEEDS=600 DD_REFINE_SEEDS=1000 timeout -k 10 900 python -m pytest tests/test_refiner.py tests/test_pipeline.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_scan4_tests.log 2>&1; rc=$?
tail -1 gpurun_out/r05_scan4_tests.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_scan4_tests.log | head -20; exit 1; fi
for t in 0 0x8000000 0; do
  echo "== tuning $t, 185 views"
  DD_EXCLUSIVE_GPU=1 timeout -k 10 200 python3 tools/bench_fused_refine.py --views 185 --tuning $t 2>&1 | grep "^fused"
done 2>&1 | tee gpurun_out/r05_scan4.log
