C="--warmup 5 --steps 150 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
for i in 1 2 3 4; do for x in "0 0" "1 0" "0 1" "1 1"; do
set -- $x
A2C_TAPE_PREFETCH=$1 A2C_POLL_RR=$2 python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pf=$1 rr=$2', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done; done
