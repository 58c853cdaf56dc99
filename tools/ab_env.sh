#!/bin/bash
# headline value for two values of an env switch, alternating, no per-step events:  tools/ab_env.sh VAR v0 v1 reps
VAR=$1; V0=$2; V1=$3; REPS=${4:-5}
for i in $(seq $REPS); do
  for v in $V0 $V1; do
    env $VAR=$v python bench.py --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'])"
  done
done
