# with the forks gone: does a gate in front of 8-view (and larger) calls win now?  (DD_CHAIN_MAX_TILES 700 against 4000, twice)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export DD_EXCLUSIVE_GPU=1
for m in 700 4000 700 4000; do
  DD_CHAIN_MAX_TILES=$m timeout -k 10 300 python3 tools/bench_streaming.py --per-call 4,8,16,32 --only builder.append --rounds 7 > gpurun_out/r05_chain_max_$m.log 2>&1 || { tail -5 gpurun_out/r05_chain_max_$m.log; exit 1; }
  echo "== DD_CHAIN_MAX_TILES=$m"; grep -E "^k=" gpurun_out/r05_chain_max_$m.log
done
