# the pipeline's own call: the densify kernel with the refiner's transfer curve fused in (DD_REFINE), 8 views per launch against 64 and 185
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for ex in 0 1; do for v in 8 64 185; do
  echo "== DD_EXCLUSIVE_GPU=$ex views $v"
  DD_EXCLUSIVE_GPU=$ex timeout -k 10 200 python3 tools/bench_fused_refine.py --views $v 2>&1 | grep -v "^[EWI]20\|amdgpu.ids" | tail -6
done; done > gpurun_out/r05_fused_refine.log 2>&1
cat gpurun_out/r05_fused_refine.log
