#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace rocpd database: every dispatch between the last two
adam kernels, ordered by start, with its queue / stream, duration and the idle gap since the previous dispatch ended
on the same queue; then a summary (busy time per queue, idle gaps, dispatch count).

    python tools/step_timeline.py <results.db> [--full]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
rows = list(db.execute(f"select {sel} from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "adam" in r[0] and "advance" not in r[0]]
if len(adam) < 2:
    sys.exit("fewer than two adam kernels in the trace")
a, b = adam[-2], adam[-1]
step = rows[a + 1:b + 1]
t0 = step[0][1]
last_end = {}
busy = {}
gaps = {}
n = {}
for name, s, e, q, st in step:
    key = (q, st)
    gap = s - last_end[key] if key in last_end else 0
    last_end[key] = max(e, last_end.get(key, 0))
    busy[key] = busy.get(key, 0) + (e - s)
    gaps[key] = gaps.get(key, 0) + max(gap, 0)
    n[key] = n.get(key, 0) + 1
    if "--full" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  q{q} s{st}  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {name[:90]}")
print(f"# step: {(step[-1][2] - t0) / 1e6:.3f} ms, {len(step)} dispatches")
for key in busy:
    print(f"# queue {key}: {n[key]} dispatches, busy {busy[key] / 1e6:.3f} ms, idle gaps {gaps[key] / 1e6:.3f} ms")
# union busy time over all queues
iv = sorted((s, e) for _, s, e, _, _ in step)
u, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print(f"# GPU busy (union over queues): {u / 1e6:.3f} ms; idle inside the step: {(step[-1][2] - t0 - u) / 1e6:.3f} ms")
