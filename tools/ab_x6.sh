for v in "A2C_GEMM_X9=0" "A2C_X=0"; do
env $v timeout 500 python bench.py --workload conv --steps 10 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('conv32 $v', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
env $v timeout 800 python bench.py --workload conv --n-envs 256 --transport u8 --grey --steps 4 --warmup 2 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shard $v', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done
