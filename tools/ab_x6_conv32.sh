# alternating runs of the ConvModel 32 x 64 config: fp32 MFMA kernels (A2C_GEMM_X9=0) against the bf16 x 6 default
for i in 1 2 3 4; do for v in "A2C_GEMM_X9=0" "A2C_X=0"; do
env $v timeout 500 python bench.py --workload conv --steps ${STEPS:-20} --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('conv32 $v', d['value'], d.get('rollout_ms'), d.get('update_ms'))"
done; done
