#!/bin/bash
# headline A/B on ONE box: old persistent kernel vs ring kernel, several env thread counts
for nw in 8 12 16; do
  for mode in ring noring; do
    if [ $mode = noring ]; then export A2C_NO_RING=1; else unset A2C_NO_RING; fi
    python bench.py --steps 60 --warmup 5 --sustain-steps 0 --no-configs --no-cpu-baseline --no-secondary --no-kernel-timers --n-workers $nw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', 'threads', $nw, 'value', d['value'], 'ms', d['ms_per_step'])"
  done
done
