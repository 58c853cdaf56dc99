#!/usr/bin/env python3
"""weight-gradient kernels of the 3x3 stacks through the C ABI at update batch: HIP-event time, TF, TB/s; first layer also
from the single-frame uint8 store.   python tools/wgrad_bench.py [N] [layer indices]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
from a2c_amd import ops  # noqa: E402

LAYERS = [(4, 84, 84, 16, 1), (16, 84, 84, 24, 1), (24, 84, 84, 32, 2), (32, 42, 42, 64, 2), (16, 84, 84, 24, 2), (24, 42, 42, 32, 2),
          (32, 21, 21, 48, 2), (48, 11, 11, 64, 2)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
only = [int(a) for a in sys.argv[2:]]
dev = torch.device("cuda")


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for li, (Cin, H, W, Cout, S) in enumerate(LAYERS):
    if only and li not in only:
        continue
    d = ops.conv_desc(Cin, H, W, Cout, 3, S, 1)
    x = (torch.rand(N, Cin, H, W, device=dev) < 0.25).float()
    dout = torch.randn(N, Cout, d.OH, d.OW, device=dev)
    dW, db = torch.empty(Cout, Cin, 3, 3, device=dev), torch.empty(Cout, device=dev)
    ws = torch.empty((ops.conv_bwd_weight_ws_bytes(d, N) + 3) // 4, device=dev)
    ms = timeit(lambda: ops.conv_bwd_weight(d, x.data_ptr(), Cin * H * W, dout, dW, db, N, ws))
    fl = 2.0 * N * Cout * d.OH * d.OW * Cin * 9
    by = 4.0 * N * (Cin * H * W + Cout * d.OH * d.OW)
    line = f"L{li} {Cin:2d}->{Cout:2d} @{H} s{S}  N={N}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF  {by / ms / 1e9:5.2f} TB/s"
    if li == 0:
        T = 128
        R = N // T
        Fs = (torch.rand(R, T + 4, H * W, device=dev) < 0.25).to(torch.uint8)
        nv = torch.full((N,), 4, dtype=torch.int32, device=dev)
        ms8 = timeit(lambda: ops.conv_bwd_weight_frames(d, Fs, Fs.stride(0), T, nv, dout, dW, db, N, ws))
        by8 = N * (Cin * H * W + 4.0 * Cout * d.OH * d.OW)
        line += f" | from the uint8 store: {ms8:7.3f} ms  {by8 / ms8 / 1e9:5.2f} TB/s"
    print(line, flush=True)
    del x, dout
