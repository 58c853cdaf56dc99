"""run tools/ingest_bench.py with the process pinned to a CPU list (first-touch places the pinned region there)"""
import os, sys, runpy
cpus = sys.argv[1]
lst = []
for part in cpus.split(","):
    a, _, b = part.partition("-")
    lst += list(range(int(a), int(b or a) + 1))
os.sched_setaffinity(0, lst)
sys.argv = ["tools/ingest_bench.py"] + sys.argv[2:]
runpy.run_path("tools/ingest_bench.py", run_name="__main__")
