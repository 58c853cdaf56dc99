#!/usr/bin/env python3
"""Phase stamps of bwd_stream2_kernel (A3CModel conv2 backward-data at update batch), workgroup 0, per wave:
    python tools/bwd_stream_timing.py [N]      -> shader clocks per sample and phase, waves 0 (half 0) and 4 (half 1)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda")
d = ops.conv_desc(16, 20, 20, 32, 4, 2, 0)
w = (torch.rand(32, 16, 4, 4, device=dev) - 0.5) * 0.2
wb = torch.empty(ops.conv_prep_floats(d, 1), device=dev)
ops.conv_prep(d, 1, w, wb)
dout = torch.randn(N, 32, 9, 9, device=dev)
act = torch.relu(torch.randn(N, 16, 20, 20, device=dev))
lm = torch.zeros(N, 100, dtype=torch.int64, device=dev)
ops.lanemask_from_act(act, lm)
din = torch.empty(N, 16, 20, 20, device=dev)
fn = lambda: ops.conv_bwd_data_lanemask(d, dout, wb, lm, din, N)
fn(); fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"N={N}: {e0.elapsed_time(e1) / 10:.3f} ms per launch")
lib = _lib.load()
lib.a2c_debug_bwd_stream_timing.argtypes = [ctypes.c_void_p]
buf = torch.zeros(64, dtype=torch.int64, device=dev)
lib.a2c_debug_bwd_stream_timing(buf.data_ptr())
fn()
torch.cuda.synchronize()
lib.a2c_debug_bwd_stream_timing(None)
t = buf.cpu().reshape(8, 8)
names = ["wait dOut + staging", "barrier 1", "flush prev", "tile pair 0", "issue loads", "other tile pairs", "barrier 2"]
for wv in (0, 4, 3, 7):
    n = max(int(t[wv, 7]), 1)
    per = [float(t[wv, i]) / n for i in range(7)]
    print(f"wave {wv} ({n} samples): total {sum(per):.0f} clocks/sample: " + ", ".join(f"{a} {b:.0f}" for a, b in zip(names, per)))
