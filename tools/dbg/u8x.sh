python -m pytest tests/test_gpu_ingest.py tests/test_gpu_rounds.py tests/test_gpu_frames.py -q -x 2>&1 | tail -3
for i in 1 2; do for x in 0 1; do
echo "xsplit=$x"; A2C_RING_XSPLIT=$x python tools/ring_timing.py 14 bits frame_store 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=list(d); print('  sum', d['sum_us_per_step'], 'rollout', d['rollout_ms_timed'], [round(d[x],2) for x in k[:9]])"
done; done
