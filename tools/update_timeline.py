#!/usr/bin/env python3
"""Kernels of ONE steady-state epoch (from one marker kernel launch to the next) of a rocprofv3 --kernel-trace CSV, in launch
order with duration and the idle gap before each:   python tools/update_timeline.py <kernel_trace.csv> <marker substring> [epochs]"""
import csv
import re
import sys
from collections import OrderedDict

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:56]
mark = sys.argv[2]
idx = [i for i, r in enumerate(rows) if mark in r[2]]
nav = int(sys.argv[3]) if len(sys.argv) > 3 else 10
idx = idx[-nav - 1:]
agg = OrderedDict()
n_ep = 0
lens = [b - a for a, b in zip(idx[:-1], idx[1:])]
mode = max(set(lens), key=lens.count)
for a, b in zip(idx[:-1], idx[1:]):
    if b - a != mode:
        continue
    n_ep += 1
    prev_end = rows[a][0]
    for k, (s, e, n) in enumerate(rows[a:b]):
        d = agg.setdefault((k, short(n)), [0, 0])
        d[0] += e - s
        d[1] += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    agg.setdefault((9999, "epoch total"), [0, 0])[0] += rows[b][0] - rows[a][0]
print(f"{n_ep} epochs of {mode} launches averaged")
for (k, n), (d, g) in agg.items():
    if d / n_ep > 3000 or g / n_ep > 3000 or k == 9999:
        print(f"{k:4d} {n:58s} dur {d / n_ep / 1e3:9.1f} us   gap before {g / n_ep / 1e3:7.1f} us")
