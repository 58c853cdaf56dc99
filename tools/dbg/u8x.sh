C="--warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
run() { python bench.py $* $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d.get('rollout_ms'), d.get('update_ms'))"; }
for i in 1 2 3; do
echo "gru rows"; run --workload gru_bptt --steps 10
echo "gru frames"; run --workload gru_bptt --steps 10 --frame-store
done
for i in 1 2 3; do
echo "conv32 rows"; run --workload conv --steps 20
echo "conv32 frames"; run --workload conv --steps 20 --frame-store
done
for i in 1 2; do
echo "shard rows"; run --workload conv --n-envs 256 --steps 5
echo "shard frames"; run --workload conv --n-envs 256 --steps 5 --frame-store
done
