# the fused refine stage with the curve found through a grid and four windows sharing their sorted columns: same bits? how fast?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_REFINE_SEEDS=400 timeout -k 10 600 python -m pytest tests/test_refiner.py tests/test_pipeline.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_refine_fast_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r05_refine_fast_tests.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_refine_fast_tests.log | head -20; exit 1; fi
for t in 0 0x8000000 0x10000000 0x18000000 0; do
  echo "== tuning $t (bit 27: bisect all knots, bit 28: one median per window), 185 views"
  DD_EXCLUSIVE_GPU=1 timeout -k 10 200 python3 tools/bench_fused_refine.py --views 185 --tuning $t 2>&1 | grep "^fused"
done > gpurun_out/r05_refine_fast.log 2>&1
cat gpurun_out/r05_refine_fast.log
