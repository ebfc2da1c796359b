#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_resources.py depthdensifier_amd/csrc/ddcore.hip [-DDD_DENSE=0 ...]
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    src, flags = sys.argv[1], sys.argv[2:]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + flags
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r": remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:") or txt.startswith("Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'spill':>5} {'SGPR':>5} {'LDS':>7} {'occ':>4}  kernel")
    for r, n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "").replace("((anonymous namespace)::KArgs)", "")
        print(f"{r.get('VGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>5} {r.get('TotalSGPRs', '?'):>5} {r.get('LDS Size [bytes/block]', '?'):>7} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4}  {n}")


if __name__ == "__main__":
    main()
