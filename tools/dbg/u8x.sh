python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frames.py tests/test_gpu_timed_path.py -q -x 2>&1 | tail -3
C="--steps 10 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
for fs in "" "--frame-store"; do
echo "== gru $fs"; python bench.py --workload gru_bptt $C $fs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done
for fs in "" "--frame-store"; do
echo "== conv32 $fs"; python bench.py --workload conv $C $fs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('rollout_ms'), d.get('update_ms'))"; done
echo "== shard"; python bench.py --workload conv --n-envs 256 --steps 4 --warmup 2 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('rollout_ms'), d.get('update_ms'))"
