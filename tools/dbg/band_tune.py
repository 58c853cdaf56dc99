import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for (Cin, H, W, Cout, S) in [(32, 21, 21, 48, 2), (48, 11, 11, 64, 2)]:
    d = ops.conv_desc(Cin, H, W, Cout, 3, S, 1)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=dev)
    ops.conv_prep(d, 1, w, wb)
    dout = torch.randn(N, Cout, d.OH, d.OW, device=dev)
    mask = torch.randn(N, Cin, H, W, device=dev)
    din = torch.empty(N, Cin, H, W, device=dev)
    for _ in range(3):
        ops.conv_bwd_data(d, dout, wb, mask, din, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv_bwd_data(d, dout, wb, mask, din, N)
    e1.record(); torch.cuda.synchronize()
    print(f"{Cin}<-{Cout} @{H}: {e0.elapsed_time(e1)/5:.3f} ms", flush=True)
