#!/usr/bin/env python3
"""Joins profiles/pmc_traffic.json (L2-side bytes per launch) with a rocprofv3 kernel-stats CSV (average duration per launch):
bytes per step, time per step and achieved TB/s per kernel, convolutions and element-wise passes summed separately.

    python tools/traffic_table.py profiles/pmc_traffic.json profiles/r02_step_kernel_stats_single_stream.csv [steps=8]
"""
import csv
import json
import re
import sys

pmc = json.load(open(sys.argv[1]))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else pmc.get("steps_profiled", 8)


def canon(name):
    m = re.match(r"_Z\d+(conv_igemm_kernel|conv_wgrad_kernel)I(DF16b|DF16_|f)((?:L[ib]\d+E)+)", name)
    if m:
        t = {"DF16b": "bf16", "DF16_": "f16", "f": "f32"}[m.group(2)]
        return f"{m.group(1)}<{t}," + ",".join(re.findall(r"L[ib](\d+)E", m.group(3))) + ">"
    return name.split("(")[0]


dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = canon(r["Name"])
    c, t = dur.get(k, (0, 0))
    dur[k] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
rows = []
for name, v in pmc["kernels"].items():
    k = canon(name) if not name.startswith("conv_") else name
    calls, tot = dur.get(k, (0, 0))
    b = v["launches"] / steps * v["hbm_bytes_per_launch"]
    us = tot / steps / 1e3 if calls else 0.0
    rows.append((b, us, k, v["launches"] / steps))
rows.sort(reverse=True)
print(f"# csrc {pmc['csrc_sha16']}; L2-side bytes (FETCH_SIZE x2 + WRITE_SIZE) per step, kernel time per step (one stream, every kernel alone)")
print(f"{'kernel':58s} {'launches':>8s} {'GB/step':>8s} {'us/step':>8s} {'TB/s':>6s}")
tot = {"conv": [0, 0], "elementwise": [0, 0]}
for b, us, k, n in rows:
    kind = "conv" if k.startswith("conv_") or "wgrad_reduce" in k else "elementwise"
    k = re.sub(r"^_Z\d+|^void ", "", k)
    tot[kind][0] += b
    tot[kind][1] += us
    if b > 2e7:
        print(f"{k[:58]:58s} {n:8.1f} {b / 1e9:8.2f} {us:8.1f} {b / us / 1e6 if us else 0:6.2f}")
for kind, (b, us) in tot.items():
    print(f"# {kind}: {b / 1e9:.2f} GB/step in {us / 1e3:.2f} ms = {b / us / 1e6:.2f} TB/s")
