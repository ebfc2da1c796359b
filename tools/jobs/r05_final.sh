# the whole GPU suite, the default bench line and three profiled runs of the driver's command on the tree as it is
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite.log 2>&1; rc=$?
tail -4 gpurun_out/r05_gpu_suite.log
if [ $rc -ne 0 ]; then grep -E "^E|^FAILED" gpurun_out/r05_gpu_suite.log | head -20; exit 1; fi
( time timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default_6.json 2> gpurun_out/r05_bench_default_6.err ) 2>&1 | tail -4
tail -2 gpurun_out/r05_bench_default_6.err
timeout -k 10 800 bash tools/profile_driver_cmd.sh gpurun_out/driver_cmd r05_driver_cmd 3 > gpurun_out/r05_driver_cmd.log 2>&1
grep -v "^[EWI]20[0-9][0-9]" gpurun_out/r05_driver_cmd.log | grep -E "scene2000 N=1|failed|259584768"
