#!/usr/bin/env python3
"""print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}")
