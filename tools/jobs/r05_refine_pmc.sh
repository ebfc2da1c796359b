# where the fused refine kernel's wave cycles go (SQ counters, two passes), beside the plain kernel of the same process
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp DD_EXCLUSIVE_GPU=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/rp_$i
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "compact_lean" --kernel-trace --output-format csv -d /tmp/rp_$i -- python3 "$GRAFT_REPO_ROOT/tools/bench_fused_refine.py" --views 185 > "$GRAFT_REPO_ROOT/gpurun_out/r05_refine_pmc_$i.log" 2>&1) || { echo "set $i failed"; tail -5 gpurun_out/r05_refine_pmc_$i.log; exit 1; }
done
python3 - <<'P'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"/tmp/rp_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "compact_lean" not in k: continue
            tag = "fused_refine" if ("true, 16>" in k or "ELb1ELi16" in k) else "plain"
            acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {t: {c: sum(v) / len(v) for c, v in d.items()} for t, d in acc.items()}
for t, d in res.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        d["share_wait_any"] = round(d["SQ_WAIT_ANY"] / wc, 3); d["share_wait_inst"] = round(d["SQ_WAIT_INST_ANY"] / wc, 3); d["share_active"] = round(d["SQ_ACTIVE_INST_ANY"] / wc, 3)
    if d.get("SQ_BUSY_CYCLES"):
        d["valu_busy_per_simd"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (d["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
json.dump(res, open("gpurun_out/r05_refine_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
