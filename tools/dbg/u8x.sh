C="--warmup 5 --steps 150 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
for i in 1 2 3 4; do for x in "0 0" "1 0" "1 1"; do
set -- $x
A2C_RING_STORES_EARLY=$1 A2C_RING_EXPAND_ALL=$2 python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('se=$1 ea=$2', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done; done
