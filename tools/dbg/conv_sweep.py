"""A/B sweep of env toggles on one box: python tools/dbg/conv_sweep.py <pass> <B> spec1 spec2 ... -- K1=v1,K2=v2 K1=v3 ...
Each setting is a comma list of env assignments ('-' = none); prints ms per (spec, setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops

what, B = sys.argv[1], int(sys.argv[2])
rest = sys.argv[3:]
specs = rest[:rest.index("--")]
settings = rest[rest.index("--") + 1:]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for sp in specs:
    spec = tuple(int(v) for v in sp.split(","))
    d = ops.conv_desc(*spec)
    Cin, H, W, Cout = spec[:4]
    x = (torch.rand(B, Cin, H, W, device=dev, generator=g) < 0.25).float()
    w = (torch.rand(Cout, Cin, spec[4], spec[4], device=dev, generator=g) - 0.5) * 0.1
    bias = torch.zeros(Cout, device=dev)
    out = torch.empty(B, Cout, d.OH, d.OW, device=dev)
    dout = torch.randn(B, Cout, d.OH, d.OW, device=dev, generator=g)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=dev)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=dev)
    ops.conv_prep(d, 0, w, wf); ops.conv_prep(d, 1, w, wb)
    dW, db = torch.empty_like(w), torch.empty(Cout, device=dev)
    din = torch.empty_like(x)
    row = []
    for st in settings:
        kv = [a.split("=") for a in st.split(",") if a != "-"]
        for k, v in kv:
            os.environ[k] = v
        ws = torch.empty(max(1, ops.conv_bwd_weight_ws_bytes(d, B) // 4), device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            if what == "fwd":
                ops.conv_fwd(d, x.data_ptr(), Cin * H * W, wf, bias, True, out, B)
            elif what == "wgrad":
                ops.conv_bwd_weight(d, x.data_ptr(), Cin * H * W, dout, dW, db, B, ws)
            else:
                ops.conv_bwd_data(d, dout, wb, x, din, B)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / reps)
        for k, _ in kv:
            del os.environ[k]
    print(f"{what} {sp:22s} B={B}: " + "  ".join(f"{st}: {ms:.3f}" for st, ms in zip(settings, row)), flush=True)
