C="--warmup 2 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
run() { python bench.py $* $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d.get('rollout_ms'), d.get('update_ms'), d.get('roofline',{}).get('kernel','')[:40], d.get('roofline',{}).get('frac'))"; }
for i in 1 2; do
echo "2048 body (RING_BLOCKS=0)"; A2C_RING_BLOCKS=0 run --n-envs 2048 --steps 10
echo "2048 ring blocks, rows"; run --n-envs 2048 --steps 10 --no-frame-store
echo "2048 ring blocks, frames lazy"; run --n-envs 2048 --steps 10
done
echo "512 body"; A2C_RING_BLOCKS=0 run --n-envs 512 --steps 30
echo "512 ring blocks"; run --n-envs 512 --steps 30
python -m pytest tests/test_gpu_rounds.py tests/test_gpu_ingest.py -q -x 2>&1 | tail -3
