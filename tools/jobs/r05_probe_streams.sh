# dd_streams_overlap: the tests, then the scenario in which the two side streams used to share a hardware queue (bench_streaming --graph)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_streaming_calls.py tests/test_host_cpu.py -x -q 2>&1 | tail -4
export TMPDIR=/tmp DD_EXCLUSIVE_GPU=1
timeout -k 10 300 python3 tools/bench_streaming.py --per-call 1,2,4 --graph --rounds 7 > gpurun_out/r05_probe_bs_graph.log 2>&1 || { tail -5 gpurun_out/r05_probe_bs_graph.log; exit 1; }
grep -E "^k=" gpurun_out/r05_probe_bs_graph.log
rm -rf /tmp/q_graph
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/q_graph -- python3 "$GRAFT_REPO_ROOT/tools/bench_streaming.py" --per-call 1 --graph --only builder.append --rounds 5 > "$GRAFT_REPO_ROOT/gpurun_out/r05_q_graph_probed.log" 2>&1) || { echo "profile failed"; tail -5 gpurun_out/r05_q_graph_probed.log; exit 1; }
grep -E "^k=" gpurun_out/r05_q_graph_probed.log
python3 tools/kernel_timeline.py /tmp/q_graph gpurun_out/r05_queue_ids_graph_probed.txt 260352 | head -9
