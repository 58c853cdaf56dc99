C="--warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
run() { python bench.py $* $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d.get('rollout_ms'), d.get('update_ms'))"; }
for i in 1 2 3; do
echo "gru old"; A2C_TAPE_PREFETCH=0 A2C_POLL_RR=0 run --workload gru_bptt --steps 10
echo "gru new"; run --workload gru_bptt --steps 10
done
for i in 1 2 3; do
echo "conv32 old"; A2C_TAPE_PREFETCH=0 A2C_POLL_RR=0 run --workload conv --steps 20
echo "conv32 new"; run --workload conv --steps 20
done
