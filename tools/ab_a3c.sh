#!/bin/bash
# A/B of the headline on ONE box, alternating runs: tools/ab_a3c.sh VAR val0 val1 [repeats] [extra bench args]
V=$1; A=$2; B=$3; N=${4:-3}; shift 4
for i in $(seq $N); do
  for x in $A $B; do
    env $V=$x timeout 500 python bench.py --steps 100 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V=$x', d['value'], d['rollout_ms'], d['update_ms'])"
  done
done
