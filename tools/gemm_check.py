#!/usr/bin/env python3
"""fp32 GEMM sites of the models through the C ABI: error against torch (fp64 on CPU for a sample of rows) and HIP-event
timings.  Usage: python tools/gemm_check.py [M ...]   (default M = 32 256: the rollout batch of configs 2 and 5)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

dev = torch.device("cuda")
Ms = [int(a) for a in sys.argv[1:]] or [32, 256]
N, K = 2000, 28224
g = torch.Generator().manual_seed(1)
W = ((torch.rand(N, K, generator=g) - 0.5) * 0.02).to(dev)
b = ((torch.rand(N, generator=g) - 0.5) * 0.1).to(dev)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for M in Ms:
    x = (torch.rand(M, K, generator=g) - 0.3).to(dev)
    out = torch.empty(M, N, device=dev)
    sk = ops.pick_splitk(M, N, K)
    ws = torch.empty(max(ops.gemm_ws_bytes(M, N, sk), 4) // 4, device=dev) if sk > 1 else None

    def run():
        ops.gemm(0, 1, M, N, K, x.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, bias=b, relu=True, splitk=sk, ws=ws)
    run()
    rows = [0, M // 2, M - 1]
    ref = torch.relu(x[rows].cpu().double() @ W.cpu().double().t() + b.cpu().double())
    err = float((out[rows].cpu().double() - ref).abs().max())
    ms = timeit(run, 30)
    fl = 2.0 * M * N * K
    print(f"linear.fwd M={M} N={N} K={K} splitk={sk}: max|err| {err:.2e} | {ms * 1e3:.1f} us {fl / ms / 1e9:.1f} TF "
          f"weights {4.0 * N * K / ms / 1e6:.0f} GB/s", flush=True)
