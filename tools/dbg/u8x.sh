python tools/conv3_check.py 6 7 2>&1 | grep "^L" > gpurun_out/c3_band4.txt
A2C_BAND_NTU2=1 python tools/conv3_check.py 6 7 2>&1 | grep "^L" > gpurun_out/c3_band2.txt
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -3
