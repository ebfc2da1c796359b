# soak of the random chains of appends (tests/test_streaming_calls.py) on the final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_STREAM_SEEDS=16 timeout -k 10 600 python -m pytest tests/test_streaming_calls.py -x -q -m gpu -p no:cacheprovider -k random_chains > gpurun_out/r05_soak_e0.log 2>&1; rc=$?
echo "16 seeds: $(tail -1 gpurun_out/r05_soak_e0.log)"
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED|Error" gpurun_out/r05_soak_e0.log | head -20; exit 1; fi
DD_STREAM_SEEDS=600 timeout -k 10 1000 python -m pytest tests/test_streaming_calls.py -x -q -m gpu -p no:cacheprovider -k random_chains > gpurun_out/r05_soak_e.log 2>&1; rc=$?
echo "600 seeds: $(tail -1 gpurun_out/r05_soak_e.log)"
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED|Error" gpurun_out/r05_soak_e.log | head -20; exit 1; fi
