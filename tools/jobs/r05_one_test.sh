cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_streaming_calls.py -x -q -k "probed" > gpurun_out/r05_one_test.log 2>&1
grep -E "^E|Error|assert" gpurun_out/r05_one_test.log | head -30
