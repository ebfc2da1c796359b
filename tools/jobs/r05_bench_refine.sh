# the bench contract tests and the default line with the fused-refine record
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bench_contract.py -x -q -m gpu > gpurun_out/r05_bench_contract.log 2>&1; rc=$?
tail -3 gpurun_out/r05_bench_contract.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_bench_contract.log | head -20; exit 1; fi
( time timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default_7.json 2> gpurun_out/r05_bench_default_7.err ) 2>&1 | tail -4
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r05_bench_default_7.json').read().strip().splitlines()[-1])
print('scene2000', d['roofline']['frac'], 'garden185', d['garden185']['roofline']['frac'])
print(json.dumps(d['garden185']['fused_refine']))
s=d['garden185']['streaming']['per_call']; print('k1', s['1']['frac'], 'k8', s['8']['frac'])
P
