#!/usr/bin/env python3
"""conv3.hip dispatch boundaries: forward (+ sign words) and sign-word backward-data of every streaming layer at batch sizes
around the small/large switch (64) and around one band per CU, against torch on the CPU.  Prints one line per (layer, B)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from a2c_amd import ops  # noqa: E402

LAYERS = [(4, 84, 84, 16, 1), (16, 84, 84, 24, 1), (24, 84, 84, 32, 2), (32, 42, 42, 64, 2), (16, 84, 84, 24, 2),
          (24, 42, 42, 32, 2), (32, 21, 21, 48, 2), (48, 11, 11, 64, 2)]
dev = torch.device("cuda")
bad = 0
for li, (Cin, H, W, Cout, S) in enumerate(LAYERS):
    d = ops.conv_desc(Cin, H, W, Cout, 3, S, 1)
    g = torch.Generator().manual_seed(100 + li)
    wt = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.4
    bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=dev)
    ops.conv_prep(d, 0, wt.to(dev), wf)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=dev)
    ops.conv_prep(d, 1, wt.to(dev), wb)
    nsw = ops.conv_sign_words(d)
    for B in (1, 2, 37, 64, 65, 129, 257):
        x = torch.rand(B, Cin, H, W, generator=g) - 0.3
        ref = F.relu(F.conv2d(x, wt, bias, stride=S, padding=1))
        out = torch.empty(B, Cout, d.OH, d.OW, device=dev)
        xd, bd = x.to(dev), bias.to(dev)      # (kept alive: a temporary would be freed, and possibly reused, before the launch)
        ops.conv_fwd(d, xd.data_ptr(), Cin * H * W, wf, bd, True, out, B)
        e0 = float((out.cpu() - ref).abs().max())
        msg = f"L{li} B={B}: fwd {e0:.1e}"
        ok = e0 < 1e-5
        if nsw:
            out2 = torch.empty_like(out)
            sg = torch.zeros(B, nsw, dtype=torch.int32, device=dev)
            ops.conv_fwd_signs(d, xd.data_ptr(), Cin * H * W, wf, bd, True, out2, sg.data_ptr(), nsw, B)
            same = torch.equal(out2, out) or float((out2 - out).abs().max()) < 1e-6
            RW = (d.OW + 31) // 32
            pad = torch.zeros(B, Cout, d.OH, RW * 32, dtype=torch.bool)
            pad[..., :d.OW] = out2.cpu() > 0
            words = (pad.reshape(B, Cout, d.OH, RW, 32).long() * (2 ** torch.arange(32, dtype=torch.int64))).sum(-1)
            words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).int().reshape(B, -1)
            wok = bool((sg.cpu() == words).all())
            msg += f" | signs fwd same {same} words {wok}"
            ok = ok and same and wok
        if li > 0 and ops.conv_bwd_data_signs_supported(d):
            dout = torch.rand(B, Cout, d.OH, d.OW, generator=g) - 0.5
            mask = torch.rand(B, Cin, H, W, generator=g) - 0.4
            RWi = (W + 31) // 32
            pad = torch.zeros(B, Cin, H, RWi * 32, dtype=torch.bool)
            pad[..., :W] = mask > 0
            words = (pad.reshape(B, Cin, H, RWi, 32).long() * (2 ** torch.arange(32, dtype=torch.int64))).sum(-1)
            words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).int().reshape(B, -1).contiguous().to(dev)
            din = torch.empty(B, Cin, H, W, device=dev)
            dd = dout.to(dev)
            ops.conv_bwd_data_signs(d, dd, wb, words, din, B)
            refb = F.conv_transpose2d(dout, wt, stride=S, padding=1, output_padding=(H + 2 - 3) % S) * (mask > 0)
            e1 = float((din.cpu() - refb).abs().max())
            msg += f" | bwd signs {e1:.1e}"
            ok = ok and e1 < 1e-5
        bad += 0 if ok else 1
        print(msg + ("" if ok else "   <<<<<< MISMATCH"), flush=True)
print("mismatches:", bad)
