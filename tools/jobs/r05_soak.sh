# soak of the randomised parity sweeps on the final tree (the round-5 single-pass geometries in the draw)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 1000 python -m pytest $FILES -x -q -m gpu -p no:cacheprovider > gpurun_out/r05_soak_$tag.log 2>&1; rc=$?; echo "$* $FILES: $(tail -1 gpurun_out/r05_soak_$tag.log)"; if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_soak_$tag.log | head; exit 1; fi; }
FILES=tests/test_gpu_random.py
run a DD_RANDOM_SEEDS=12000
run b DD_RANDOM_SEEDS=1500 DD_RANDOM_SCALE=6
run c DD_RANDOM_SEEDS=200 DD_RANDOM_SCALE=14
FILES="tests/test_filter.py tests/test_refiner.py"
run d DD_VOTE_SEEDS=400 DD_REFINE_SEEDS=2000
