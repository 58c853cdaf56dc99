#!/usr/bin/env python3
"""3x3 conv layers of ConvModel / GRUModel through the C ABI: max error against torch (CPU, fp32) at a small batch and
HIP-event timings at rollout / update batch sizes.  Run twice to compare the kernel families:
    python tools/conv3_check.py            # shape-specialised streaming kernels (conv3.hip) where they apply
    A2C_NO_C3=1 python tools/conv3_check.py    # conv.hip's generic kernels"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from a2c_amd import ops  # noqa: E402

LAYERS = [  # Cin, H, W, Cout, stride, has bwd-data
    (4, 84, 84, 16, 1, False), (16, 84, 84, 24, 1, True), (24, 84, 84, 32, 2, True), (32, 42, 42, 64, 2, True),
    (16, 84, 84, 24, 2, True), (24, 42, 42, 32, 2, True), (32, 21, 21, 48, 2, True), (48, 11, 11, 64, 2, True)]
dev = torch.device("cuda")
only = [int(a) for a in sys.argv[1:]]
tag = "generic (A2C_NO_C3=1)" if os.environ.get("A2C_NO_C3") == "1" else "conv3 where supported"
print("kernels:", tag)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for li, (Cin, H, W, Cout, S, has_bwd) in enumerate(LAYERS):
    if only and li not in only:
        continue
    d = ops.conv_desc(Cin, H, W, Cout, 3, S, 1)
    g = torch.Generator().manual_seed(li)
    wt = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.4
    bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=dev)
    ops.conv_prep(d, 0, wt.to(dev), wf)
    wb = None
    if has_bwd:
        wb = torch.empty(ops.conv_prep_floats(d, 1), device=dev)
        ops.conv_prep(d, 1, wt.to(dev), wb)
    # ---- correctness at B = 3
    B = 3
    x = torch.rand(B, Cin, H, W, generator=g) - 0.3
    out = torch.empty(B, Cout, d.OH, d.OW, device=dev)
    xd = x.to(dev)
    ops.conv_fwd(d, xd.data_ptr(), Cin * H * W, wf, bias.to(dev), True, out, B)
    ref = F.relu(F.conv2d(x, wt, bias, stride=S, padding=1))
    err_f = float((out.cpu() - ref).abs().max())
    err_b = float("nan")
    if has_bwd:
        dout = torch.rand(B, Cout, d.OH, d.OW, generator=g) - 0.5
        mask = torch.rand(B, Cin, H, W, generator=g) - 0.4
        din = torch.empty(B, Cin, H, W, device=dev)
        ops.conv_bwd_data(d, dout.to(dev), wb, mask.to(dev), din, B)
        refb = F.conv_transpose2d(dout, wt, stride=S, padding=1, output_padding=(H + 2 - 3) % S) * (mask > 0)
        err_b = float((din.cpu() - refb).abs().max())
    # ---- sign words: forward leaves (out > 0) as bits, the layer above masks with them; B = 150: several bands per workgroup
    err_s = ""
    nsw = ops.conv_sign_words(d)
    if nsw:
        Bs = 150
        xs = torch.rand(Bs, Cin, H, W, generator=g) - 0.3
        outs = torch.empty(Bs, Cout, d.OH, d.OW, device=dev)
        sg = torch.full((Bs, nsw + 3), -1, dtype=torch.int32, device=dev)
        xsd, bsd = xs.to(dev), bias.to(dev)
        ops.conv_fwd_signs(d, xsd.data_ptr(), Cin * H * W, wf, bsd, True, outs, sg.data_ptr(), nsw + 3, Bs)
        refs = F.relu(F.conv2d(xs, wt, bias, stride=S, padding=1))
        e1 = float((outs.cpu() - refs).abs().max())
        RW = (d.OW + 31) // 32
        bits = (outs.cpu() > 0).reshape(Bs, Cout, d.OH, d.OW)
        pad = torch.zeros(Bs, Cout, d.OH, RW * 32, dtype=torch.bool)
        pad[..., :d.OW] = bits
        wgt = (2 ** torch.arange(32, dtype=torch.int64))
        words = (pad.reshape(Bs, Cout, d.OH, RW, 32).long() * wgt).sum(-1)
        words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).int().reshape(Bs, -1)
        ok_w = bool((sg.cpu()[:, :nsw] == words).all()) and bool((sg.cpu()[:, nsw:] == -1).all())
        err_s = f" | signs B={Bs}: fwd err {e1:.2e} words {'ok' if ok_w else 'MISMATCH'}"
    if has_bwd and ops.conv_bwd_data_signs_supported(d):
        Bs = 150
        dout = torch.rand(Bs, Cout, d.OH, d.OW, generator=g) - 0.5
        mask = torch.rand(Bs, Cin, H, W, generator=g) - 0.4
        RWi = (W + 31) // 32
        pad = torch.zeros(Bs, Cin, H, RWi * 32, dtype=torch.bool)
        pad[..., :W] = mask > 0
        wgt = (2 ** torch.arange(32, dtype=torch.int64))
        words = (pad.reshape(Bs, Cin, H, RWi, 32).long() * wgt).sum(-1)
        words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).int().reshape(Bs, -1).contiguous().to(dev)
        din_s = torch.empty(Bs, Cin, H, W, device=dev)
        din_f = torch.empty(Bs, Cin, H, W, device=dev)
        dd = dout.to(dev)
        ops.conv_bwd_data_signs(d, dd, wb, words, din_s, Bs)
        ops.conv_bwd_data(d, dd, wb, mask.to(dev), din_f, Bs)
        refb = F.conv_transpose2d(dout, wt, stride=S, padding=1, output_padding=(H + 2 - 3) % S) * (mask > 0)
        e2 = float((din_s.cpu() - refb).abs().max())
        same = bool(torch.equal(din_s, din_f))
        ops.conv_bwd_data_signs(d, dd, wb, None, din_s, Bs)
        refn = F.conv_transpose2d(dout, wt, stride=S, padding=1, output_padding=(H + 2 - 3) % S)
        e3 = float((din_s.cpu() - refn).abs().max())
        err_s += f" | bwd_data signs: err {e2:.2e} (no mask {e3:.2e}) == float-mask kernel: {same}"
    # weight gradient against autograd
    xg = x.clone().requires_grad_(False)
    wt_g = wt.clone().requires_grad_(True)
    bias_g = bias.clone().requires_grad_(True)
    dout_w = torch.rand(B, Cout, d.OH, d.OW, generator=g) - 0.5
    F.conv2d(xg, wt_g, bias_g, stride=S, padding=1).backward(dout_w)
    dW = torch.empty(Cout, Cin, 3, 3, device=dev)
    dbv = torch.empty(Cout, device=dev)
    ws = torch.empty(max(ops.conv_bwd_weight_ws_bytes(d, B), 4) // 4, device=dev)
    ops.conv_bwd_weight(d, xd.data_ptr(), Cin * H * W, dout_w.to(dev), dW, dbv, B, ws)
    sc = float(wt_g.grad.abs().max())
    err_w = float((dW.cpu() - wt_g.grad).abs().max()) / sc
    err_db = float((dbv.cpu() - bias_g.grad).abs().max()) / float(bias_g.grad.abs().max())
    line = f"L{li} {Cin}->{Cout} {H}x{W} s{S}: max|err| fwd {err_f:.2e} bwd_data {err_b:.2e} wgrad(rel) {err_w:.2e} db(rel) {err_db:.2e}"
    line += err_s
    # ---- timings
    for Bt in (32, 256, 4096):
        xb = torch.rand(Bt, Cin, H, W, device=dev) - 0.3
        ob = torch.empty(Bt, Cout, d.OH, d.OW, device=dev)
        bd = bias.to(dev)
        ms = timeit(lambda: ops.conv_fwd(d, xb.data_ptr(), Cin * H * W, wf, bd, True, ob, Bt), 20 if Bt <= 256 else 3)
        fl = 2.0 * Bt * Cout * d.OH * d.OW * Cin * 9
        by = 4.0 * Bt * (Cin * H * W + Cout * d.OH * d.OW)
        line += f" | fwd B={Bt}: {ms * 1e3:.1f} us {fl / ms / 1e9:.1f} TF {by / ms / 1e6:.0f} GB/s"
        if ops.conv_sign_words(d):
            nsw_ = ops.conv_sign_words(d)
            sgo = torch.empty(Bt, nsw_, dtype=torch.int32, device=dev)
            ms = timeit(lambda: ops.conv_fwd_signs(d, xb.data_ptr(), Cin * H * W, wf, bd, True, ob, sgo.data_ptr(), nsw_, Bt),
                        20 if Bt <= 256 else 3)
            line += f" (+signs {ms * 1e3:.1f} us)"
        if has_bwd and Bt == 4096:
            do = torch.rand(Bt, Cout, d.OH, d.OW, device=dev) - 0.5
            di = torch.empty(Bt, Cin, H, W, device=dev)
            ms = timeit(lambda: ops.conv_bwd_data(d, do, wb, xb, di, Bt), 3)
            by = 4.0 * Bt * (2 * Cin * H * W + Cout * d.OH * d.OW)
            line += f" | bwd_data B={Bt}: {ms * 1e3:.1f} us {fl / ms / 1e9:.1f} TF {by / ms / 1e6:.0f} GB/s"
            if ops.conv_bwd_data_signs_supported(d):
                sgi = torch.randint(-2 ** 31, 2 ** 31 - 1, (Bt, Cin * H * ((W + 31) // 32)), dtype=torch.int32, device=dev)
                ms = timeit(lambda: ops.conv_bwd_data_signs(d, do, wb, sgi, di, Bt), 3)
                by = 4.0 * Bt * (Cin * H * W + Cout * d.OH * d.OW)
                line += f" | bwd_data(signs): {ms * 1e3:.1f} us {fl / ms / 1e9:.1f} TF {by / ms / 1e6:.0f} GB/s"
                if S == 2 and hasattr(ops.lib(), "a2c_debug_c3_timing"):      # phase stamps of workgroup 0 (one wave per role)
                    import ctypes
                    lib = ops.lib()
                    lib.a2c_debug_c3_timing.argtypes = [ctypes.c_void_p]
                    buf = torch.zeros(8, dtype=torch.int64, device=dev)
                    lib.a2c_debug_c3_timing(buf.data_ptr())
                    ops.conv_bwd_data_signs(d, do, wb, sgi, di, Bt)
                    torch.cuda.synchronize()
                    lib.a2c_debug_c3_timing(None)
                    t = [v / 100 for v in buf.cpu().tolist()]
                    line += (f" [staged wg0 over {int(t[6] * 100)} chunks: mfma {t[0]:.0f} us, wait@X {t[1]:.0f}, image write {t[2]:.0f}, "
                             f"wait@chunk {t[3]:.0f}; storer drain {t[4]:.0f}; loader issue+wait {t[5]:.0f}]")
            if S == 2 and hasattr(ops.lib(), "a2c_debug_c3_timing"):      # phase stamps of workgroup 0 / wave 0
                import ctypes
                lib = ops.lib()
                lib.a2c_debug_c3_timing.argtypes = [ctypes.c_void_p]
                buf = torch.zeros(4, dtype=torch.int64, device=dev)
                lib.a2c_debug_c3_timing(buf.data_ptr())
                ops.conv_bwd_data(d, do, wb, xb, di, Bt)
                torch.cuda.synchronize()
                lib.a2c_debug_c3_timing(None)
                t = buf.cpu().tolist()
                if t[3]:
                    line += f" [wg0: compute {t[0] / 100:.0f} us, epilogue {t[1] / 100:.0f} us, barrier wait {t[2] / 100:.0f} us over {t[3]} chunks]"
        if Bt == 4096:
            do = torch.rand(Bt, Cout, d.OH, d.OW, device=dev) - 0.5
            dWt = torch.empty(Cout, Cin, 3, 3, device=dev)
            dbt = torch.empty(Cout, device=dev)
            wst = torch.empty(max(ops.conv_bwd_weight_ws_bytes(d, Bt), 4) // 4, device=dev)
            ms = timeit(lambda: ops.conv_bwd_weight(d, xb.data_ptr(), Cin * H * W, do, dWt, dbt, Bt, wst), 3)
            by = 4.0 * Bt * (Cin * H * W + Cout * d.OH * d.OW)
            line += f" | wgrad B={Bt}: {ms * 1e3:.1f} us {fl / ms / 1e9:.1f} TF {by / ms / 1e6:.0f} GB/s"
        del xb, ob
    print(line, flush=True)
