# tests/test_streaming_calls.py and the C client once more
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_streaming_calls.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_stream_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r05_stream_tests.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_stream_tests.log | head -20; fi
exit $rc
