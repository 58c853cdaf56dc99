for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20/5:', d['value'], d.get('rollout_ms'), d.get('update_ms'), d['ms_per_step'])"; done
for i in 1 2; do python bench.py --gpus 1 --steps 200 --warmup 5 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('200/5:', d['value'], d.get('rollout_ms'), d.get('update_ms'), d['ms_per_step'])"; done
