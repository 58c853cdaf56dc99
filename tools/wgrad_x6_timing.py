#!/usr/bin/env python3
"""A3CModel conv2's weight gradient alone at N = 32768: wgrad_x6_kernel against the fp32 MFMA kernel and the opt-in pipelined
form (A2C_WGRAD_X6=2), each with parts switched off (A2C_WGRAD_X6_DBG bits: 1 = no a1 conversion, 2 = no dOut conversion,
4 = no MFMAs, 8 = no DMA / loads, 16 = pipelined form with every wave multiplying first; wrong sums)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = ops.conv_desc(16, 20, 20, 32, 4, 2, 0)
a1 = torch.relu(torch.rand(B, 6400, device="cuda") - 0.3)
dout = (torch.rand(B, 32, 9, 9, device="cuda") - 0.5) * (torch.rand(B, 32, 9, 9, device="cuda") < 0.6)
ws = torch.empty(ops.conv_bwd_weight_ws_bytes(d, B) // 4 + 1, device="cuda")
dW, db = torch.empty(32, 16, 4, 4, device="cuda"), torch.empty(32, device="cuda")


def t(reps=10):
    ops.conv_bwd_weight(d, a1.data_ptr(), 6400, dout, dW, db, B, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_bwd_weight(d, a1.data_ptr(), 6400, dout, dW, db, B, ws)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(2):
    os.environ["A2C_WGRAD_X6"] = "0"
    print(f"fp32 MFMA kernel            {t():.3f} ms")
    os.environ["A2C_WGRAD_X6"] = "2"
    for dbg, name in ((0, "bf16 x 6, pipelined"), (1, "  p: no conversion"), (4, "  p: no MFMAs"), (8, "  p: no loads"), (16, "  p: all waves multiply first"),
                      (9, "  p: matrix only"), (12, "  p: conversion only"), (5, "  p: loads + barrier only")):
        os.environ["A2C_WGRAD_X6_DBG"] = str(dbg)
        print(f"{name:28s}{t():.3f} ms")
    os.environ.pop("A2C_WGRAD_X6_DBG")
    os.environ["A2C_WGRAD_X6"] = "1"
    for dbg, name in ((0, "bf16 x 6"), (1, "  no a1 conversion"), (2, "  no dOut conversion"), (3, "  no conversion"), (4, "  no MFMAs"),
                      (8, "  no DMA"), (7, "  DMA + barriers only"), (11, "  MFMAs only"), (12, "  conversion only")):
        os.environ["A2C_WGRAD_X6_DBG"] = str(dbg)
        print(f"{name:28s}{t():.3f} ms")
    os.environ.pop("A2C_WGRAD_X6_DBG")
