# smoke(), the whole GPU suite and the default bench line on the tree as it is
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite.log 2>&1; rc=$?
tail -3 gpurun_out/r05_gpu_suite.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_gpu_suite.log | head -20; exit 1; fi
( time timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default_8.json 2> gpurun_out/r05_bench_default_8.err ) 2>&1 | tail -4
python3 tools/show_bench.py gpurun_out/r05_bench_default_8.json | head -1
