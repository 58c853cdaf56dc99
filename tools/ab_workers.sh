#!/bin/bash
# headline rollout time for several env-thread counts, alternating (same box):  tools/ab_workers.sh "12 14" reps
for i in $(seq ${2:-3}); do
  for n in $1; do
    python bench.py --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers --steps 150 --n-workers $n 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers', $n, d['value'], d['rollout_ms'], d['update_ms'])"
  done
done
