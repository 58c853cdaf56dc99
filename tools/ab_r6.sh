#!/bin/bash
# round 6 A/B of the headline update on ONE box, alternating runs.  tools/ab_r6.sh [repeats]
# legs: form 1 of the conv2 backward-data kernel + float mask (round 5), form 2 + lane masks (default), form 3 + lane masks (opt-in)
N=${1:-3}
run() {
  env "$@" timeout 500 python bench.py --steps 100 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>gpurun_out/ab_r6.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); f=json.load(open(d['full_report'])); print('$*', d['value'], d['rollout_ms'], d['update_ms'], round(f['update_launch_sites_ms']['conv2.bwd_data']['avg_ms'],4))"
}
for i in $(seq $N); do
  run A2C_BWD_STREAM_V1=1 A2C_NO_LANEMASK=1
  run A2C_BWD_STREAM_FORM=2
  run A2C_BWD_STREAM_FORM=3
done
