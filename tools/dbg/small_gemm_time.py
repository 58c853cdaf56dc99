import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops
dev = "cuda"
for (tB, M, N, K) in [(1, 256, 256, 2304), (1, 256, 256, 256), (0, 256, 768, 256), (0, 256, 512, 256), (1, 32, 512, 2000), (1, 2048, 512, 2000)]:
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if tB else (K, N), device=dev); C = torch.empty(M, N, device=dev)
    row = []
    for env in ("", "1"):
        if env: os.environ["A2C_NO_SMALL_GEMM"] = "1"
        sk = ops.pick_splitk(M, N, K)
        ws = torch.empty(max(1, ops.gemm_ws_bytes(M, N, sk) // 4), device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        ops.gemm(0, tB, M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], C.data_ptr(), N, splitk=sk, ws=ws)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(50):
                ops.gemm(0, tB, M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], C.data_ptr(), N, splitk=sk, ws=ws)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 50 * 1e3)
        if env: del os.environ["A2C_NO_SMALL_GEMM"]
    print(f"tB={tB} M={M} N={N} K={K}: small {row[0]:.1f} us  split-K {row[1]:.1f} us")
