# the default bench under rocprofv3 --kernel-trace with the trace kept: do the chained one-view calls of its streaming record overlap?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showbus 2>/dev/null | grep -i "pci" | head -1
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlb -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$GRAFT_REPO_ROOT/gpurun_out/r05_tlb_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r05_tlb.err") || { echo failed; tail -5 gpurun_out/r05_tlb.err; exit 1; }
python3 tools/kernel_timeline.py /tmp/tlb gpurun_out/r05_tlb_top.txt | tail -12
for g in 260352; do python3 tools/kernel_timeline.py /tmp/tlb gpurun_out/r05_tlb_$g.txt $g 2>&1 | head -9; done
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r05_tlb_bench.json').read().strip().splitlines()[-1])
s=d['garden185']['streaming']['per_call']['1']; print('k1', s['frac'], s['us_per_call'], s['frac_shared_gpu_mode'], s['hip_graph_chain_ms'], s['host_enqueue_ms'])
P
