# bench.py's default run with its streaming record wrapped: the chained chain again with one thing changed at a time
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocm-smi --showbus 2>/dev/null | grep -i "pci bus" | head -1
timeout -k 10 400 python3 tools/experiments/slow_chain_probe.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_slow_chain_default.json 2> gpurun_out/r05_slow_chain_default.err; rc=$?
grep "\[probe\]" gpurun_out/r05_slow_chain_default.err
if [ $rc -ne 0 ]; then tail -5 gpurun_out/r05_slow_chain_default.err; exit 1; fi
timeout -k 10 300 python3 tools/experiments/slow_chain_probe.py --workload garden185 --cpu-seconds 0 --strong-views 0 > gpurun_out/r05_slow_chain_garden.json 2> gpurun_out/r05_slow_chain_garden.err; rc=$?
grep "\[probe\]" gpurun_out/r05_slow_chain_garden.err
exit $rc
