#!/bin/bash
# One rocprofv3 --kernel-trace --stats run of the default workload on whatever box this lands on; prints the box's PCI id,
# the bench line's kernel time and rocprof's average for compact_lean, and leaves the condensed CSV + bench line under <out>.
#   usage: tools/profile_garden_once.sh <outdir>
set -uo pipefail
OUT=$(realpath -m "$1"); R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_once; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_once -- \
    python3 "$R/bench.py" --cpu-seconds 0 --strong-views 0 > /tmp/prof_once_bench.json 2> /tmp/prof_once.err || { echo "profile failed"; tail -3 /tmp/prof_once.err; exit 1; }
python3 - "$OUT" "$R" <<'PY'
import json, sys
from pathlib import Path
out, root = Path(sys.argv[1]), Path(sys.argv[2])
sys.path.insert(0, str(root / "tools"))
import summarize_prof
line = json.loads(Path("/tmp/prof_once_bench.json").read_text())
pci = line["device"].split("pci ")[-1].replace(":", "")
(out / f"bench_garden185_under_rocprof_pci{pci}.json").write_text(json.dumps(line, indent=1) + "\n")
summarize_prof.main("/tmp/prof_once", str(out / f"garden185_kernel_stats_pci{pci}.csv"),
                    "rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --strong-views 0   (box pci " + pci + ")")
print("BOX", pci, "bench kernel_ms", line["roofline"]["kernel_ms"], "frac", line["roofline"]["frac"])
PY
