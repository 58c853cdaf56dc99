#!/bin/bash
for i in 1 2 3; do
python bench.py --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers --steps 150 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no events', d['value'])"
python bench.py --no-configs --no-secondary --no-cpu-baseline --steps 150 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('events   ', d['value'], d['rollout_ms'], d['update_ms'])"
done
