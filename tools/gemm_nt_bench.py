#!/usr/bin/env python3
"""The weight stream of ConvModel's hidden layer at rollout batch (M = 32 envs, 2000 x 28224 fp32 = 226 MB per env step):
a2c_gemm_f32_partial timed alone, between launches that sweep the caches.   python tools/gemm_nt_bench.py [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, K = 2000, 28224
x = torch.rand(M, K, device=dev)
Ws = [(torch.rand(N, K, device=dev) - 0.5) * 0.05 for _ in range(3)]          # 3 x 226 MB: round robin past the 256 MB MALL
for tgt in os.environ.get("TARGETS", "512").split():
    os.environ["A2C_SPLITK_TARGET"] = tgt
    sk = ops.pick_splitk(M, N, K)
    ws = torch.empty(max(1, (ops.gemm_ws_bytes(M, N, sk) + 3) // 4), device=dev)
    for rr in (1, 3):
        reps = 30
        for i in range(3):
            ops.gemm_partial(0, 1, M, N, K, x.data_ptr(), K, Ws[i % rr].data_ptr(), K, sk, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            ops.gemm_partial(0, 1, M, N, K, x.data_ptr(), K, Ws[i % rr].data_ptr(), K, sk, ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"target {tgt} splitk {sk:3d}  {'same W' if rr == 1 else '3 Ws  '}  {us:7.1f} us  {N * K * 4 / us / 1e6:6.2f} TB/s")
