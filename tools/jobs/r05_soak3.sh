# a longer soak of the protocols that depend on timing (scan service, chained calls): random chains of appends and the random sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 1100 python -m pytest $FILES -x -q -m gpu -p no:cacheprovider $KSEL > gpurun_out/r05_soak_$tag.log 2>&1; rc=$?; echo "$* $FILES $KSEL: $(tail -1 gpurun_out/r05_soak_$tag.log)"; if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_soak_$tag.log | head; exit 1; fi; }
FILES=tests/test_streaming_calls.py; KSEL="-k random_chains"
run f DD_STREAM_SEEDS=6000
FILES=tests/test_gpu_random.py; KSEL=""
run g DD_RANDOM_SEEDS=60000
run h DD_RANDOM_SEEDS=3000 DD_RANDOM_SCALE=6
