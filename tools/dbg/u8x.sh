python -m pytest tests/test_gpu_frames.py tests/test_gpu_rounds.py -q -x 2>&1 | tail -2
for i in 1 2 3; do for x in 0 1; do
A2C_RING_STORES_EARLY=$x python tools/ring_timing.py 14 bits frame_store 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=list(d); print('early=$x sum', d['sum_us_per_step'], 'rollout', d['rollout_ms_timed'], [round(d[x],2) for x in k[:9]])"
done; done
