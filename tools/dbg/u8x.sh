C="--warmup 5 --steps 150 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
for i in 1 2 3 4 5 6; do for x in 0 1; do
A2C_PUSH=$x python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('push=$x', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done; done
