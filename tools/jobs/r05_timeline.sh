# the sharing tests once more, then 8-view calls with and without a gate under rocprofv3 --kernel-trace (tools/kernel_timeline.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_fuse_gpu.py -x -q -m gpu -k "sharing_the_gpu" > gpurun_out/r05_fuse_share.log 2>&1; rc=$?
tail -5 gpurun_out/r05_fuse_share.log
if [ $rc -ge 124 ]; then echo "test step killed ($rc): stopping"; exit 1; fi
export TMPDIR=/tmp DD_EXCLUSIVE_GPU=1
for m in 4000 0; do
  export DD_CHAIN_MAX_TILES=$m
  rm -rf /tmp/tl_$m
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$m -- python3 /root/repo/tools/bench_streaming.py --per-call 8 --only builder.append --rounds 5 > /root/repo/gpurun_out/r05_tl_$m.log 2>&1) || { echo "profile $m failed"; tail -5 gpurun_out/r05_tl_$m.log; exit 1; }
  python3 tools/kernel_timeline.py /tmp/tl_$m gpurun_out/r05_timeline_8views_chain_max_$m.txt
  grep -v "^[EWI]20" gpurun_out/r05_tl_$m.log | tail -4
done
