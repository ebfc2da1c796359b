# what the fork in front of every chained call costs, and whether dd_stream_fork's idle shortcut triggers (tools/experiments/fork_cost.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export DD_EXCLUSIVE_GPU=1
rocm-smi --showbus 2>/dev/null | grep -i "pci bus" | head -1
timeout -k 10 300 python3 tools/experiments/fork_cost.py > gpurun_out/r05_fork_cost.log 2>&1 || { tail -5 gpurun_out/r05_fork_cost.log; exit 1; }
grep -v "^[EWI]20" gpurun_out/r05_fork_cost.log | tail -10
