# dd_refine_apply against the tensor formulation on the CPU, seeded; then the fused stage against dd_refine_apply
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_APPLY_SEEDS=1500 DD_REFINE_SEEDS=3000 timeout -k 10 1000 python -m pytest tests/test_refiner.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_soak_apply.log 2>&1; rc=$?
echo "DD_APPLY_SEEDS=1500 DD_REFINE_SEEDS=3000: $(tail -1 gpurun_out/r05_soak_apply.log)"
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_soak_apply.log | head -20; fi
exit $rc
