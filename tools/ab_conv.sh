#!/bin/bash
# alternating runs of the ConvModel 32 x 64 config under environment switches (edit the list)
for i in 1 2 3; do for v in "A2C_X=0" "A2C_SPLITK_TARGET=128" "A2C_SPLITK_TARGET=64"; do
env $v timeout 500 python bench.py --workload conv --steps 10 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['rollout_ms'], d['update_ms'])"
done; done
