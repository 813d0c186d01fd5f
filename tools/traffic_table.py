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
    """Base kernel name: rocprofv3 reports some instantiations mangled, some with a garbled template list, and the PMC and
    kernel-trace outputs disagree on which - the template variants of a kernel are therefore ONE row here."""
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return re.split(r"[<(]", name)[0].strip()


dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = canon(r["Name"])
    c, t = dur.get(k, (0, 0))
    dur[k] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
agg = {}
for name, v in pmc["kernels"].items():
    k = canon(name)
    b, n = agg.get(k, (0.0, 0.0))
    agg[k] = (b + v["launches"] / steps * v["hbm_bytes_per_launch"], n + v["launches"] / steps)
rows = []
for k, (b, n) in agg.items():
    calls, tot = dur.get(k, (0, 0))
    rows.append((b, tot / steps / 1e3 if calls else 0.0, k, n))
rows.sort(reverse=True)
print(f"# csrc {pmc['csrc_sha16']}; HBM-side bytes (FETCH_SIZE + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes) per step, kernel time per step")
print("# (one stream, every kernel alone on the whole chip); the template variants of a kernel are summed")
print(f"{'kernel':44s} {'launches':>8s} {'GB/step':>8s} {'us/step':>8s} {'TB/s':>6s}")
tot = {"conv": [0, 0], "elementwise": [0, 0]}
for b, us, k, n in rows:
    kind = "conv" if k.startswith("conv") or "wgrad" in k or k.startswith("up1x1") else "elementwise"
    tot[kind][0] += b
    tot[kind][1] += us
    if b > 2e7:
        print(f"{k[:44]:44s} {n:8.1f} {b / 1e9:8.2f} {us:8.1f} {b / us / 1e6 if us else 0:6.2f}")
for kind, (b, us) in tot.items():
    print(f"# {kind}: {b / 1e9:.2f} GB/step in {us / 1e3:.2f} ms = {b / us / 1e6:.2f} TB/s")
