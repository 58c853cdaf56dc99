C="--warmup 5 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
run() { python bench.py $* $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d.get('rollout_ms'), d.get('update_ms'))"; }
for i in 1 2 3; do
echo "push=0"; A2C_PUSH=0 run --steps 200
echo "push=1"; A2C_PUSH=1 run --steps 200
done
echo "timing push=0"; A2C_PUSH=0 python tools/ring_timing.py 14 bits frame_store 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if isinstance(v,float)})"
echo "timing push=1"; A2C_PUSH=1 python tools/ring_timing.py 14 bits frame_store 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if isinstance(v,float)})"
