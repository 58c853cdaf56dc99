#!/usr/bin/env python3
"""Phase timestamps of a2c_a3c_step (library built with -DA2C_STEP_TIMING):
    make -C pytorch-a2c_amd/csrc clean all EXTRA=-DA2C_STEP_TIMING && python tools/step_timing.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
import a2c_amd
from a2c_amd import ops, _lib

B, A = 256, 6
net = a2c_amd.models.A3CModel([4, 84, 84], A, h_size=256)
net._ensure_device()
st = ops.stream()
net._refresh(st)
S = 4 * 84 * 84
prev = (torch.rand(B, S, device="cuda") < 0.25).float()
frame = (torch.rand(B, 7056, device="cuda") < 0.25).float()
out = torch.empty(B, S, device="cuda")
u = torch.rand(B, device="cuda")
acts = torch.zeros(B, dtype=torch.int64, device="cuda")
reset = torch.zeros(B, device="cuda")
lib = _lib.load()
dbg = hasattr(lib, "a2c_debug_step_ts")


def run(label, n=20):
    """hipGraph of n launches (no host launch cost), replayed 5 times"""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            net._step(B, ops.stream(), prev=prev.data_ptr(), prev_stride=S, frame_new=frame.data_ptr(),
                      reset_mask=reset.data_ptr(), out=out.data_ptr(), out_stride=S, u=u.data_ptr(),
                      actions=acts.data_ptr(), act_stride=1)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1) / (5 * n) * 1e3:7.1f} us per launch (hipGraph of {n})")


run("full")
if dbg:
    ts = (ctypes.c_ulonglong * 16)()
    assert lib.a2c_debug_step_ts(ts) == 0
    print("timestamps (10 ns ticks):", [int(ts[i + 1] - ts[i]) for i in range(8)])
    for mask, label in ((1, "no row stores"), (2, "no conv1"), (4, "no conv2"), (8, "no heads"), (16, "no state loads"),
                        (17, "no state loads/stores"), (6, "no conv1/conv2"), (14, "no conv1/conv2/heads"),
                        (31, "nothing (launch + weights + barriers)"), (63, "nothing, no weight loads"),
                        (64, "empty kernel (same grid and LDS)")):
        assert lib.a2c_debug_step_skip(mask) == 0
        run(label)
        assert lib.a2c_debug_step_ts(ts) == 0
        print("      timestamps (10 ns ticks):", [int(ts[i + 1] - ts[i]) for i in range(8)])
