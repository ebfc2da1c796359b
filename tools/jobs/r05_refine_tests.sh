# the fused refine stage against dd_refine_apply + plain densify with NaN-free cases and odd knot distributions in the draw
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_REFINE_SEEDS=3000 timeout -k 10 900 python -m pytest tests/test_refiner.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_soak_refine.log 2>&1; rc=$?
echo "DD_REFINE_SEEDS=3000: $(tail -1 gpurun_out/r05_soak_refine.log)"
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_soak_refine.log | head -20; fi
exit $rc
