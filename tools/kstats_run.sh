#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py run, top kernels printed:  tools/kstats_run.sh <bench args...>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers > /tmp/tr.json 2>/tmp/tr.err
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:36]:
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"]).split("(")[0][:58]
    print(f"{n:58s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
tail -1 /tmp/tr.json | cut -c1-160
