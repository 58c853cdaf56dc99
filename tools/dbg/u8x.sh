python -m pytest tests/test_gpu_models.py tests/test_gpu_timed_path.py tests/test_gpu_updater.py -q -x -k "GRU or gru or bptt or Gru" 2>&1 | tail -3
C="--warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
run() { python bench.py $* $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d.get('rollout_ms'), d.get('update_ms'))"; }
for i in 1 2; do
echo "unfused"; A2C_NO_GRU_CARRY=1 run --workload gru_bptt --steps 10
echo "fused"; run --workload gru_bptt --steps 10
done
