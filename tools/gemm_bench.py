#!/usr/bin/env python3
"""Time a2c_gemm_f32 on the three linear-layer shapes of the A3C update (and variants)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, F, H = 32768, 2592, 256
a2 = (torch.rand(M, F, device=dev, generator=g) < 0.4).float() * torch.rand(M, F, device=dev, generator=g)
W = (torch.rand(H, F, device=dev, generator=g) - 0.5) * 0.05
demb = torch.randn(M, H, device=dev, generator=g) * 0.01
emb = torch.empty(M, H, device=dev)
da2 = torch.empty(M, F, device=dev)
dW = torch.empty(H, F, device=dev)
bias = torch.zeros(H, device=dev)


def timeit(name, fn, flops, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:44s} {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TFLOP/s")


fl = 2.0 * M * F * H
sk = ops.pick_splitk(M, H, F)
ws = torch.empty(max(1, ops.gemm_ws_bytes(M, H, sk) // 4), device=dev)
timeit(f"fwd   emb = a2 W^T (splitk {sk})", lambda: ops.gemm(0, 1, M, H, F, a2.data_ptr(), F, W.data_ptr(), F, emb.data_ptr(), H, bias=bias, splitk=sk, ws=ws), fl)
timeit("bwd_d da2 = demb W  * mask", lambda: ops.gemm(0, 0, M, F, H, demb.data_ptr(), H, W.data_ptr(), F, da2.data_ptr(), F, mask_ptr=a2.data_ptr(), ldmask=F), fl)
timeit("bwd_d da2 = demb W  (no mask)", lambda: ops.gemm(0, 0, M, F, H, demb.data_ptr(), H, W.data_ptr(), F, da2.data_ptr(), F), fl)
sk2 = ops.pick_splitk(H, F, M)
ws2 = torch.empty(max(1, ops.gemm_ws_bytes(H, F, sk2) // 4), device=dev)
timeit(f"bwd_w dW = demb^T a2 (splitk {sk2})", lambda: ops.gemm(1, 0, H, F, M, demb.data_ptr(), H, a2.data_ptr(), F, dW.data_ptr(), F, splitk=sk2, ws=ws2), fl)
