# headline, alternating: fp32 MFMA backward-data (A2C_BWD_X6=0) against the bf16 x 6 kernel (default)
for i in 1 2 3; do for v in "A2C_BWD_X6=0" "A2C_X=0"; do
env $v timeout 500 python bench.py --steps ${STEPS:-200} --warmup 5 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('a3c $v', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done; done
