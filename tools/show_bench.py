#!/usr/bin/env python3
"""Condensed view of one bench.py line: tools/show_bench.py <file.json>"""
import json, signal, sys
signal.signal(signal.SIGPIPE, signal.SIG_DFL)      # piped into head
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"{d['config']['workload']} N={d['n_gpus']}: {d['value']} {d['unit']}  {d['ms_per_step']} ms/step  frac {r['frac']} (kernel {r['kernel_ms']} ms)  device {d.get('device')}")
print("  fresh allocations:", r.get("frac_min"), r.get("frac_median"), r.get("frac_max"), r.get("kernel_ms_per_allocation"), "placement:", r.get("placement"))
rep = r.get("placement_report") or {}
print("  placement report:", {k: v for k, v in rep.items() if k != "arena"}, (rep.get("arena") or {}))
print("  verified:", d.get("verified"))
s = d.get("strong2000") or {}
for k in ("sharded", "gathered", "gathered_compact"):
    if k in s:
        print(f"  strong2000.{k}:", {a: b for a, b in s[k].items() if a not in ("what",)})
print("  strong2000.verified:", s.get("verified"), "placement:", s.get("placement"), "error:", s.get("error"))
c = d.get("cpu_baseline") or {}
print("  cpu_baseline:", c.get("value"), c.get("unit"), "cores", c.get("cores"), "| all_cores:", (c.get("all_cores") or {}).get("value"))
for k in ("devices", "rccl_world_size", "speedup_vs_n1"):
    if k in d:
        print(f"  {k}:", d[k])
