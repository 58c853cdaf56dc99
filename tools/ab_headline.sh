#!/bin/bash
# headline on ONE box, three repeats per setting: ring kernel at 8 / 12 / 14 env threads, old persistent kernel at 12
for rep in 1 2 3; do
for cfg in "ring 8" "ring 12" "ring 14" "noring 12"; do
  set -- $cfg
  if [ $1 = noring ]; then export A2C_NO_RING=1; else unset A2C_NO_RING; fi
  python bench.py --steps 100 --warmup 5 --sustain-steps 0 --no-configs --no-cpu-baseline --no-secondary --no-kernel-timers --n-workers $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'threads', $2, 'value', d['value'], 'ms', d['ms_per_step'])"
done; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "numa\|model name\|socket" | head
