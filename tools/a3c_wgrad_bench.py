#!/usr/bin/env python3
"""A3CModel conv1 (8x8 / stride 4) weight gradient from the single-frame uint8 store at update batch, through the C ABI:
HIP-event time per launch.   python tools/a3c_wgrad_bench.py [N]      (A2C_WGRAD_F32=1: the fp32 MFMA kernel;
A2C_WSB_DBG=1 / 2 / 3: timing-only variants of the bf16-pipe kernel without its matrix phase / commit phase / both)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda")
d = ops.conv_desc(4, 84, 84, 16, 8, 4, 0)
T = 128
R = N // T
Fs = (torch.rand(R, T + 4, 84 * 84, device=dev) < 0.25).to(torch.uint8)
nv = torch.full((N,), 4, dtype=torch.int32, device=dev)
dout = torch.randn(N, 16, 20, 20, device=dev)
dW, db = torch.empty(16, 4, 8, 8, device=dev), torch.empty(16, device=dev)
ws = torch.empty((ops.conv_bwd_weight_ws_bytes(d, N) + 3) // 4, device=dev)
fn = lambda: ops.conv_bwd_weight_frames(d, Fs, Fs.stride(0), T, nv, dout, dW, db, N, ws)
fn(); fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 2.0 * N * 16 * 400 * 256
print(f"N={N} F32={os.environ.get('A2C_WGRAD_F32', '0')} DBG={os.environ.get('A2C_WSB_DBG', '0')}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF (fp32-equivalent)")
