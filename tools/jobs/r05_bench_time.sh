# the default bench with its own account of where the wall time goes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default_9.json 2> gpurun_out/r05_bench_default_9.err ) 2>&1 | tail -4
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r05_bench_default_9.json').read().strip().splitlines()[-1])
print(d['wall_s']); print('scene2000', d['roofline']['frac'])
P
