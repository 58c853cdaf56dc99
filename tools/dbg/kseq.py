"""print the kernel launch sequence (name, duration us, gap us) of a window of a rocprofv3 kernel trace: kseq.py <dir> <start> <count>"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
s, n = int(sys.argv[2]), int(sys.argv[3])
if s < 0:
    s = len(rows) + s
prev = None
for r in rows[s:s + n]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (a - prev) / 1e3 if prev else 0.0
    prev = b
    print(f"{r['Kernel_Name'][:90]:90s} {(b - a) / 1e3:8.1f} us  gap {gap:7.1f}")
