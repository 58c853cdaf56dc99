#!/bin/bash
run() {
  env "$@" timeout 500 python bench.py --steps 60 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); f=json.load(open(d['full_report'])); print('$*', d['value'], d['rollout_ms'], d['update_ms'], round(f['update_launch_sites_ms']['conv2.bwd_data']['avg_ms'],4))"
}
for i in 1 2; do
  run A2C_BWD_STREAM_FORM=2
  run A2C_BWD_STREAM_FORM=3 A2C_BS3_ORDER=0
  run A2C_BWD_STREAM_FORM=3 A2C_BS3_ORDER=1
  run A2C_BWD_STREAM_FORM=3 A2C_BS3_ORDER=2
done
