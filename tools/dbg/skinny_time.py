import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
dev = "cuda"
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); bias = torch.zeros(N, device=dev)
for sk in (8, 16, 32, 64, 128):
    ws = torch.empty(max(1, ops.gemm_ws_bytes(M, N, sk) // 4), device=dev)
    row = []
    for env in ("", "1"):
        if env: os.environ["A2C_NO_SKINNY_STREAM"] = "1"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(21):
            if it == 1: e0.record()
            ops.gemm(0, 1, M, N, K, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, bias=bias, relu=True, splitk=sk, ws=ws)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 20 * 1e3)
        if env: del os.environ["A2C_NO_SKINNY_STREAM"]
    print(f"M={M} N={N} K={K} splitk={sk}: stream {row[0]:.1f} us  tiled {row[1]:.1f} us   ({N*K*4/row[0]/1e6:.2f} TB/s)")
