#!/bin/bash
# the headline with ONE env thread (the host floor of an 8-rank run on a 16-CPU box) under environment switches:
#   tools/ab_one_thread.sh "A2C_X=0" "A2C_TAPE_WARM=0" ...      (each argument: space-separated VAR=val list)
for e in "$@"; do
  for i in 1 2; do
    env $e timeout 300 python bench.py --n-workers 1 --steps 30 --warmup 5 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['value']/1e6,3), 'M  rollout', d['rollout_ms'], 'update', d['update_ms'], ' us/env-step', round(d['rollout_ms']*1e3/(128*256),3))"
  done
done
