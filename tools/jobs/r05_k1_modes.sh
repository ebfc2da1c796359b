# which flows see the chained one-view calls at 17 us and which at 25 us?  (same box, one after the other)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocm-smi --showbus 2>/dev/null | grep -i "pci bus" | head -1
show() { python3 - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['garden185']['streaming']['per_call']
print(sys.argv[1], 'pci', d['device'].get('pci'), 'k1', s['1']['frac'], s['1']['us_per_call'], 'shared', s['1']['frac_shared_gpu_mode'], 'graph', s['1']['hip_graph_chain_ms'], 'host', s['1']['host_enqueue_ms'], '| k8', s['8']['frac'], '| scene2000', d['roofline']['frac'])
P
}
for i in 1 2; do
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_k1_default_$i.json 2> gpurun_out/r05_k1_default_$i.err || { tail -5 gpurun_out/r05_k1_default_$i.err; exit 1; }
  show gpurun_out/r05_k1_default_$i.json
done
export DD_EXCLUSIVE_GPU=1
timeout -k 10 300 python3 tools/bench_streaming.py --per-call 1 --graph --rounds 7 > gpurun_out/r05_k1_bs_graph.log 2>&1 || exit 1
grep -E "^k=" gpurun_out/r05_k1_bs_graph.log
timeout -k 10 300 python3 tools/bench_streaming.py --per-call 1 --rounds 7 > gpurun_out/r05_k1_bs_plain.log 2>&1 || exit 1
grep -E "^k=" gpurun_out/r05_k1_bs_plain.log
