#!/usr/bin/env python3
"""gemm_x6_kernel alone (prebuilt images) and its split passes on ConvModel's 28224 x 2000 layer shapes.
   python tools/gemm_x6_bench.py [rows=2048]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

dev = "cuda"
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
F, H = 28224, 2000


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def img(rows, K):
    return torch.empty(ops.gemm_x6_image_bytes(rows, K) // 2, dtype=torch.int16, device=dev)


def case(name, M, N, K, sk):
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) * 0.05
    ia, ib = img(M, K), img(N, K)
    ta = timeit(lambda: ops.gemm_x6_split(a.data_ptr(), K, M, K, True, ia))
    tb = timeit(lambda: ops.gemm_x6_split(b.data_ptr(), K, N, K, True, ib))
    at = a.t().contiguous()
    tat = timeit(lambda: ops.gemm_x6_split(at.data_ptr(), M, M, K, False, ia))
    ops.gemm_x6_split(a.data_ptr(), K, M, K, True, ia)
    c = torch.empty(M, N, device=dev)
    ws = torch.empty(max(1, sk * M * N), device=dev) if sk > 1 else None
    tg = timeit(lambda: ops.gemm_x6_images(M, N, K, ia, ib, c.data_ptr(), N, splitk=sk, ws=ws))
    fl = 2.0 * M * N * K
    print(f"{name:26s} M {M:6d} N {N:6d} K {K:6d} sk {sk:2d}: gemm {tg * 1e3:8.1f} us = {fl / tg / 1e9:6.1f} TF fp32-equiv "
          f"({6 * fl / tg / 1e9 / 2516.6:.3f} of the bf16 peak) | split A {ta * 1e3:7.1f} us (k-major source {tat * 1e3:7.1f}), B {tb * 1e3:7.1f} us")


case("bwd_data dx = de W", Nb, F, H, 1)
case("bwd_weight dW = de^T x", H, F, Nb, 1)
case("fwd (update batch)", Nb, H, F, 4)
case("fwd (rollout batch 256)", 256, H, F, 32)
