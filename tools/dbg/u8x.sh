python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_timed_path.py tests/test_gpu_frames.py -q -x 2>&1 | tail -3
C="--warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
for i in 1 2; do python bench.py --workload gru_bptt --steps 10 $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gru', d['value'], d.get('rollout_ms'), d.get('update_ms'))"; done
