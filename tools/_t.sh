timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "streaming" 2>&1 | tail -8
python tools/run_kernel.py conv2_fwd 32768 5 2>&1 | grep TFLOP
env A2C_NO_STREAM=1 python tools/run_kernel.py conv2_fwd 32768 5 2>&1 | grep TFLOP
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
