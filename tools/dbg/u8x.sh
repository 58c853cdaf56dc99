python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frames.py tests/test_gpu_models.py tests/test_gpu_timed_path.py -q -x 2>&1 | tail -3
