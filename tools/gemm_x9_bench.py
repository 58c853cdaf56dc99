#!/usr/bin/env python3
"""a2c_gemm_f32 on ConvModel's 28224 x 2000 layer at update batch (and the A3C update shapes): the bf16 x 9 form against
the fp32 MFMA form (A2C_GEMM_X9=0), time and error against fp64.   python tools/gemm_x9_bench.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

dev = "cuda"
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
F, H = 28224, 2000
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(Nb, F, device=dev, generator=g) < 0.4).float() * torch.rand(Nb, F, device=dev, generator=g)
W = (torch.rand(H, F, device=dev, generator=g) - 0.5) * 0.05
de = torch.randn(Nb, H, device=dev, generator=g) * 0.01


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, tA, tB, M, N, K, a, lda, b, ldb, ref_fn):
    out = {}
    sk = int(os.environ.get("X9_SPLITK", 0)) or ops.pick_splitk(M, N, K)
    os.environ["A2C_GEMM_X9"] = "1"
    ws = torch.empty(max(1, (ops.gemm_ws_bytes(M, N, sk, K) + 3) // 4), device=dev)
    for x9 in ("1", "0") + tuple(os.environ.get("X9_EXTRA", "").split()):
        os.environ["A2C_GEMM_X9"] = x9
        c = torch.empty(M, N, device=dev)
        ms = timeit(lambda: ops.gemm(tA, tB, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), N, splitk=sk, ws=ws))
        out[x9] = (ms, c)
    rows = torch.arange(0, M, max(1, M // 64), device=dev)[:64]
    ref = ref_fn(rows)                                      # fp64 reference of 64 rows
    rms = float(ref.pow(2).mean().sqrt())
    e9 = float((out["1"][1][rows].double() - ref).pow(2).mean().sqrt()) / rms
    e32 = float((out["0"][1][rows].double() - ref).pow(2).mean().sqrt()) / rms
    fl = 2.0 * M * N * K
    for k in out:
        if k not in ("0", "1"):
            ek = float((out[k][1][rows].double() - ref).pow(2).mean().sqrt()) / rms
            print(f"   variant {k}: {out[k][0]:7.3f} ms {fl / out[k][0] / 1e9:6.1f} TF (err {ek:.2e})")
    print(f"{name:34s} splitk {sk:2d}  x9 {out['1'][0]:7.3f} ms {fl / out['1'][0] / 1e9:6.1f} TF (err {e9:.2e}) | "
          f"fp32 {out['0'][0]:7.3f} ms {fl / out['0'][0] / 1e9:6.1f} TF (err {e32:.2e})")


run("fwd   e = x W^T  (NT)", 0, 1, Nb, H, F, x, F, W, F, lambda r: x[r].double() @ W.double().t())
run("bwd_d dx = de W  (NN)", 0, 0, Nb, F, H, de, H, W, F, lambda r: de[r].double() @ W.double())
run("bwd_w dW = de^T x (TN)", 1, 0, H, F, Nb, de, H, x, F, lambda r: de[:, r].double().t() @ x.double())
