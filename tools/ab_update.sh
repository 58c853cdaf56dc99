#!/bin/bash
# headline, N reps: value rollout_ms update_ms + the five largest update launch sites (from the side file):  tools/ab_update.sh [reps]
for i in $(seq ${1:-2}); do
python bench.py --no-configs --no-secondary --no-cpu-baseline --steps 60 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open(d["full_report"]))
print(d["value"], d["rollout_ms"], d["update_ms"], {k:round(v["avg_ms"],3) for k,v in list((f.get("update_launch_sites_ms") or {}).items())[:5]})'
done
