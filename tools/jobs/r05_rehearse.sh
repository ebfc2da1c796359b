# tools/run_multi_gpu.sh REHEARSE=1: two ranks share the one GPU over gloo, every step of the script runs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rehearse
REHEARSE=1 timeout -k 10 1000 bash tools/run_multi_gpu.sh gpurun_out/rehearse > gpurun_out/rehearse/log.txt 2>&1; rc=$?
tail -30 gpurun_out/rehearse/r05_scale_summary.txt | cut -c1-300
exit $rc
