import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops
s2, B = int(sys.argv[1]), int(sys.argv[2])
H = W = 84
dev = "cuda"
d1 = ops.conv_desc(4, H, W, 16, 3, 1, 1); d2 = ops.conv_desc(16, H, W, 24, 3, s2, 1)
x = torch.rand(B, 4, H, W, device=dev); a1 = torch.rand(B, 16, H, W, device=dev) - 0.3
w2 = torch.randn(24, 16, 3, 3, device=dev) * 0.1
dout = torch.randn(B, 24, d2.OH, d2.OW, device=dev)
wb = torch.empty(ops.conv_prep_floats(d2, 1), device=dev); ops.conv_prep(d2, 1, w2, wb)
dW1 = torch.empty(16, 4, 3, 3, device=dev); db1 = torch.empty(16, device=dev)
da1 = torch.empty(B, 16, H, W, device=dev)
ws2 = torch.empty(ops.conv_bwd_weight_ws_bytes(d1, B) // 4, device=dev)
def t(fn, reps=4):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
sep = t(lambda: (ops.conv_bwd_data(d2, dout, wb, a1, da1, B), ops.conv_bwd_weight(d1, x.data_ptr(), 4 * H * W, da1, dW1, db1, B, ws2)))
print(f"s2={s2} B={B}: separate {sep:.3f} ms")
for kb in (48, 64, 80, 100, 128, 156):
    os.environ["A2C_W1_LDS_KB"] = str(kb)
    nb = ops.conv_bwd_data_w1_ws_bytes(d2, d1, B)
    if not nb: print(kb, "n/a"); continue
    ws = torch.empty(nb // 4, device=dev)
    print(f"   fused lds<={kb} KB: {t(lambda: ops.conv_bwd_data_w1(d2, dout, wb, a1, d1, x.data_ptr(), 4 * H * W, dW1, db1, B, ws)):.3f} ms")
