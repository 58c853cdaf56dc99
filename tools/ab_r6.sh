#!/bin/bash
# round 6 A/B of the headline update on ONE box, alternating runs.  tools/ab_r6.sh [repeats]
# legs: v1 = first form of bwd_stream_kernel + float mask (round 5), v2f = second form + float mask, v2 = second form + lane masks
N=${1:-3}
run() {
  env "$@" timeout 500 python bench.py --steps 100 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>gpurun_out/ab_r6.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['rollout_ms'], d['update_ms'])"
  grep -o '"conv2.bwd_data": {[^}]*}' gpurun_out/ab_r6.err | head -1
}
for i in $(seq $N); do
  run A2C_BWD_STREAM_V1=1 A2C_NO_LANEMASK=1
  run A2C_BWD_STREAM_V1=0 A2C_NO_LANEMASK=1
  run A2C_BWD_STREAM_V1=0 A2C_NO_LANEMASK=0
done
