#!/usr/bin/env python3
"""Run one kernel family in isolation (for rocprofv3 --pmc / --kernel-trace passes).
    python tools/run_kernel.py conv1_fwd|conv1_wgrad|conv2_fwd|conv2_bwd_data|conv2_wgrad|scan|step [B] [reps]
"step" = a2c_a3c_step walking the rows of a (B*16, 4, 84, 84) rollout buffer like Runner does (B envs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops

what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
if what == "step":
    import a2c_amd
    T, A = 16, 6
    net = a2c_amd.models.A3CModel([4, 84, 84], A, h_size=256)
    net._ensure_device()
    st = ops.stream()
    net._refresh(st)
    S = 4 * 84 * 84
    states = (torch.rand(B * T, S, device=dev, generator=g) < 0.25).float()
    frames = (torch.rand(T, B, 7056, device=dev, generator=g) < 0.25).float()
    u = torch.rand(B, device=dev, generator=g)
    acts = torch.zeros(B, dtype=torch.int64, device=dev)
    reset = torch.zeros(B, device=dev)
    sp = lambda t: states.data_ptr() + 4 * t * S
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    for it in range(reps + 1):
        if it == 1:
            e0.record()
        for t in range(1, T):
            net._step(B, st, prev=sp(t - 1), prev_stride=T * S, frame_new=frames[t].data_ptr(), reset_mask=reset.data_ptr(),
                      out=sp(t), out_stride=T * S, u=u.data_ptr(), actions=acts.data_ptr(), act_stride=1)
            n += it > 0
    e1.record()
    torch.cuda.synchronize()
    print(f"step B={B}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch (eager, host-launch bound if > kernel time)")
elif what == "scan":
    n_seg, T = B, 128
    N = n_seg * T
    x, r = torch.randn(N, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
    d = (torch.rand(N, device=dev, generator=g) < 0.01).float()
    d[T - 1::T] = 1
    a, b = torch.empty_like(x), torch.empty_like(x)
    for _ in range(reps):
        ops.gae_returns(x, r, d, .9702, .99, n_seg, T, a, b)
else:
    spec = (4, 84, 84, 16, 8, 4, 0) if what.startswith("conv1") else (16, 20, 20, 32, 4, 2, 0)
    if what.startswith("spec:"):      # spec:Cin,H,W,Cout,ks,stride,pad:fwd|wgrad|bwd_data
        spec = tuple(int(v) for v in what.split(":")[1].split(","))
        what = "x_" + what.split(":")[2]
    d = ops.conv_desc(*spec)
    Cin, H, W, Cout = spec[:4]
    x = (torch.rand(B, Cin, H, W, device=dev, generator=g) < 0.25).float()
    w = (torch.rand(Cout, Cin, spec[4], spec[4], device=dev, generator=g) - 0.5) * 0.1
    bias = torch.zeros(Cout, device=dev)
    out = torch.empty(B, Cout, d.OH, d.OW, device=dev)
    dout = torch.randn(B, Cout, d.OH, d.OW, device=dev, generator=g)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=dev)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=dev)
    ops.conv_prep(d, 0, w, wf)
    ops.conv_prep(d, 1, w, wb)
    dW, db = torch.empty_like(w), torch.empty(Cout, device=dev)
    ws = torch.empty(max(1, ops.conv_bwd_weight_ws_bytes(d, B) // 4), device=dev)
    din = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(reps + 1):
        if it == 1:
            e0.record()
        if what.endswith("_fwd"):
            ops.conv_fwd(d, x.data_ptr(), Cin * H * W, wf, bias, True, out, B)
        elif what.endswith("_wgrad"):
            ops.conv_bwd_weight(d, x.data_ptr(), Cin * H * W, dout, dW, db, B, ws)
        else:
            ops.conv_bwd_data(d, dout, wb, x, din, B)
    e1.record()
    torch.cuda.synchronize()
    fl = 2.0 * B * Cout * d.OH * d.OW * Cin * spec[4] * spec[4]
    ms = e0.elapsed_time(e1) / reps
    print(f"{what} {spec} B={B}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
torch.cuda.synchronize()
print("done", what, B, reps)
