python tools/wgrad_bench.py 32768 2>&1 | grep "^L"
python tools/conv3_check.py 2>&1 | grep "^L" > gpurun_out/c3_32bit.txt
python -m pytest tests/test_gpu_frames.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -3
