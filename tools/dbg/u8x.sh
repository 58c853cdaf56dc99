python -m pytest tests/test_gpu_kernels.py -q -x -k "rank_bwd_fused" 2>&1 | tail -5
python -m pytest tests/test_gpu_models.py tests/test_gpu_timed_path.py tests/test_gpu_fullsize.py -q -x -k "a3c or A3C" 2>&1 | tail -3
C="--warmup 5 --steps 100 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
for i in 1 2 3; do for x in 1 0; do
A2C_NO_RANK_FUSED=$x python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nofuse=$x', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done; done
