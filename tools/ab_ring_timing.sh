#!/bin/bash
# phase stamps of the ring kernel for two values of an env switch, alternating, lazy-states mode.  usage: VAR v0 v1 [reps]
VAR=$1; V0=$2; V1=$3; REPS=${4:-3}
for i in $(seq $REPS); do
  for v in $V0 $V1; do
    env $VAR=$v python tools/ring_timing.py 14 bits frame_store 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=list(d)
print('$VAR=$v', 'sum', d['sum_us_per_step'], 'rollout_ms', d['rollout_ms_timed'], [round(d[x],2) for x in k[:9]])"
  done
done
