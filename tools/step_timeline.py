#!/usr/bin/env python3
"""Timeline of ONE env step of a relay rollout from a rocprofv3 --kernel-trace CSV: every kernel between two consecutive
pool_ingest launches in the steady state, with its duration and the gap to its predecessor.
    python tools/step_timeline.py <kernel_trace.csv> [n_steps_to_average]"""
import csv
import re
import sys
from collections import OrderedDict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:48]
idx = [i for i, r in enumerate(rows) if "pool_ingest" in r[2]]
nav = int(sys.argv[2]) if len(sys.argv) > 2 else 200
idx = idx[len(idx) // 2:len(idx) // 2 + nav + 1]          # steady state: the middle of the trace
agg = OrderedDict()
tot = 0
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    if b - a > 40:
        continue
    tot += 1
    prev_end = seg[0][0]
    for k, (s, e, n) in enumerate(seg):
        key = (k, short(n))
        d = agg.setdefault(key, [0, 0, 0])
        d[0] += e - s
        d[1] += max(0, s - prev_end)
        d[2] += 1
        prev_end = max(prev_end, e)
    agg.setdefault((999, "step total"), [0, 0, 0])
    agg[(999, "step total")][0] += rows[b][0] - seg[0][0]
    agg[(999, "step total")][2] += 1
print(f"{tot} steps averaged")
for (k, n), (d, g, c) in agg.items():
    if c:
        print(f"{k:3d} {n:50s} dur {d / c / 1e3:8.1f} us   gap before {g / c / 1e3:6.1f} us   ({c})")
