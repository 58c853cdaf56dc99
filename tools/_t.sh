timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "a3c_step" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers > $GRAFT_REPO_ROOT/gpurun_out/prof_step.log 2>&1
