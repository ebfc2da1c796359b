# chained calls through the bare C ABI beside builder.append: is the host the bound of the one-view chain?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export DD_EXCLUSIVE_GPU=1
timeout -k 10 300 python3 tools/bench_streaming.py --per-call 1,2,4 --rounds 7 > gpurun_out/r05_abi_chained.log 2>&1 || { tail -8 gpurun_out/r05_abi_chained.log; exit 1; }
grep -E "^k=|^#" gpurun_out/r05_abi_chained.log
