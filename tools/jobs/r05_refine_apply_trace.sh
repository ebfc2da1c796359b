# kernel durations of dd_refine_apply (rocprofv3 --kernel-trace --stats of tools/bench_refine.py): workgroups of the persistent form, LDS by the number of knots
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DD_APPLY_SEEDS=300 timeout -k 10 600 python -m pytest tests/test_refiner.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
export TMPDIR=/tmp
for w in 1024 1536 2048 4096; do
  rm -rf /tmp/ra_$w
  (cd /tmp && DD_REFINE_APPLY_WGS=$w timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ra_$w -- python3 "$GRAFT_REPO_ROOT/tools/bench_refine.py" > "$GRAFT_REPO_ROOT/gpurun_out/r05_ra_$w.log" 2>&1) || { echo "failed $w"; tail -5 gpurun_out/r05_ra_$w.log; exit 1; }
  echo "== DD_REFINE_APPLY_WGS=$w: $(python3 tools/summarize_prof.py /tmp/ra_$w gpurun_out/r05_refine_apply_kernels_$w.csv "dd_refine_apply, DD_REFINE_APPLY_WGS=$w" | grep refine_apply)"
done
