# dd_refine_apply: the persistent-tile kernel (grid of buckets, lock-step look-ups, shared sorted columns) against the 32x32-tile one
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_REFINE_SEEDS=300 timeout -k 10 600 python -m pytest tests/test_refiner.py tests/test_pipeline.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_refine_apply_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r05_refine_apply_tests.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_refine_apply_tests.log | head -20; exit 1; fi
for p in 0 1 0 1; do
  echo "== DD_REFINE_APPLY_PLAIN=$p"
  DD_REFINE_APPLY_PLAIN=$p timeout -k 10 200 python3 tools/bench_refine.py 2>&1 | grep -v "^[EWI]20\|amdgpu.ids" | grep -E "kernel|equal"
done > gpurun_out/r05_refine_apply.log 2>&1
cat gpurun_out/r05_refine_apply.log
