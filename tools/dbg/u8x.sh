python tools/dbg/band_tune.py 32768 2>&1 | grep -v amdgpu
A2C_NO_BAND_GROUPS=1 python tools/dbg/band_tune.py 32768 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -3
