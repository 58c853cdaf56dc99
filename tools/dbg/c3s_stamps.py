#!/usr/bin/env python3
"""Per-role timeline of workgroup 0 of conv1's staged forward (c3s_kernel, uint8 frames in, sign words out) at 256 samples:
needs a build with the stamps compiled in (make -C pytorch-a2c_amd/csrc EXTRA=-DA2C_C3_STAMPS after a `make clean`).  Times in us
relative to the first stamp; A2C_C3S_DS=0 selects the single-image kernel.   python tools/dbg/c3s_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

dev = "cuda"
B, H, W = 256, 84, 84
d = ops.conv_desc(4, H, W, 16, 3, 1, 1)
F = torch.randint(0, 2, (B, 4 * H * W), dtype=torch.uint8, device=dev)
nv = torch.full((B,), 4, dtype=torch.int32, device=dev)
w = torch.randn(16, 4, 3, 3, device=dev) / 6
bias = torch.randn(16, device=dev) * 0.1
wf = torch.empty(ops.conv_prep_floats(d, 0), device=dev)
ops.conv_prep(d, 0, w, wf)
out = torch.empty(B, 16, H, W, device=dev)
nsw = ops.conv_sign_words(d)
sg = torch.zeros(B, nsw, dtype=torch.int32, device=dev)
buf = torch.zeros(3 * 16 * 4, dtype=torch.int64, device=dev)
lib = ops.lib()
lib.a2c_debug_c3_timing.argtypes = [ctypes.c_void_p]


def run():
    ops.conv_fwd_frames(d, F.data_ptr(), 4 * H * W, 1, nv.data_ptr(), 1, wf, bias, True, out, B, signs=(sg.data_ptr(), nsw))


for _ in range(3):
    run()
torch.cuda.synchronize()
lib.a2c_debug_c3_timing(buf.data_ptr())
run()
torch.cuda.synchronize()
lib.a2c_debug_c3_timing(None)
t = buf.cpu().view(3, 16, 4).double()
t0 = t[t > 0].min()
names = {0: ("compute", ["start", "mfma done", "after X", "image done"]), 1: ("storer ", ["start", "drain done", "after X", "after A"]),
         2: ("loader ", ["start", "issued", "landed", "after A"])}
for k in range(8):
    for r in range(3):
        if t[r, k].max() > 0:
            print(f"band {k} {names[r][0]}: " + "  ".join(f"{n} {(float(v) - float(t0)) / 100:7.2f}" for n, v in zip(names[r][1], t[r, k])))
