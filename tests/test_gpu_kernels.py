"""-m gpu: every kernel family of liba2c_mi355x.so, called through the C ABI (a2c_amd.ops), against
the CPU oracle / the same torch CPU operator the reference calls, on seeded inputs.
Tolerances: bit-exact for the scans, sampler and integer outputs; 1e-5 (fp32, the north-star
tolerance) for dense arithmetic, scaled by the magnitude of the accumulated terms."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import hashf  # noqa: E402

DEV = "cuda"


def _ops():
    from a2c_amd import ops
    return ops


def close(name, got, want, atol=1e-5, rtol=1e-5):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    want = want.detach().cpu().double().numpy() if torch.is_tensor(want) else np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{name}: {bad.sum()}/{bad.size} off, max abs err {err.max():.3e} at {i}: "
                             f"got {got[i]:.8g} want {want[i]:.8g} (scale {np.abs(want).max():.3g})")


def rnd(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy(hashf(n, seed, lo, hi).reshape(shape))


# ------------------------------------------------------------------ scans
def test_discount_golden_bitexact(golden):
    from a2c_amd.utils import discount
    g = golden["g1_discount"]
    for i in range(int(g["n_cases"])):
        y = discount(g[f"x{i}"], g[f"d{i}"], float(g[f"g{i}"]))
        assert np.array_equal(y.cpu().numpy(), g[f"y{i}"]), i


@pytest.mark.parametrize("n_seg,T", [(1, 1), (3, 5), (64, 128), (257, 33), (2, 2047), (2, 2048), (3, 5000), (2048, 128)])
def test_discount_rows_vs_oracle(n_seg, T):
    ops = _ops()
    x = rnd((n_seg * T,), 1, -1, 1)
    d = (rnd((n_seg * T,), 2, 0, 1) < 0.07).float()
    d[T - 1::T] = 1.0
    r = rnd((n_seg * T,), 3, -1, 1)
    want = np.concatenate([O.discount_np(x[i * T:(i + 1) * T], d[i * T:(i + 1) * T], 0.9702) for i in range(n_seg)])
    xd, dd, rd = x.to(DEV), d.to(DEV), r.to(DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    y = ops.discount_rows(xd, dd, 0.9702, n_seg, T, err=err)
    assert np.array_equal(y.cpu().numpy(), want)
    assert int(err.item()) == 0
    advs, rets = torch.empty_like(xd), torch.empty_like(xd)
    ops.gae_returns(xd, rd, dd, 0.9702, 0.99, n_seg, T, advs, rets)
    want_r = np.concatenate([O.discount_np(r[i * T:(i + 1) * T], d[i * T:(i + 1) * T], 0.99) for i in range(n_seg)])
    assert np.array_equal(advs.cpu().numpy(), want)
    assert np.array_equal(rets.cpu().numpy(), want_r)


def test_discount_row_invariant_flag_and_flat_fallback():
    from a2c_amd.utils import discount
    ops = _ops()
    x = rnd((40,), 4).to(DEV)
    d = torch.zeros(40, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    flat = O.discount_np(x.cpu().numpy(), d.cpu().numpy(), 0.9)
    y = ops.discount_rows(x, d, 0.9, 4, 10, err=err)
    assert int(err.item()) == 1                       # rows do not end with done == 1 ...
    assert np.array_equal(y.cpu().numpy(), flat)       # ... so the call fell back to the reference's flat scan
    assert np.array_equal(discount(x, d, 0.9, n_tsteps=10).cpu().numpy(), flat)
    d2 = d.clone()
    d2[19] = 1                                         # only one of the four rows ends properly: still flat
    assert np.array_equal(ops.discount_rows(x, d2, 0.9, 4, 10, err=err).cpu().numpy(),
                          O.discount_np(x.cpu().numpy(), d2.cpu().numpy(), 0.9))
    a_, r_ = torch.empty_like(x), torch.empty_like(x)
    ops.gae_returns(x, -x, d2, 0.9, 0.8, 4, 10, a_, r_, err=err)
    assert int(err.item()) == 1
    assert np.array_equal(a_.cpu().numpy(), O.discount_np(x.cpu().numpy(), d2.cpu().numpy(), 0.9))
    assert np.array_equal(r_.cpu().numpy(), O.discount_np(-x.cpu().numpy(), d2.cpu().numpy(), 0.8))
    y = discount(x, d, 0.9)            # flat: carries across everything, exactly like the reference
    assert np.array_equal(y.cpu().numpy(), flat)
    assert discount(torch.zeros(0), torch.zeros(0), 0.9).numel() == 0


def test_discount_full_size_properties():
    """BASELINE size (n_envs=2048 x T=128) and the saturating size: linearity in x and a sampled
    comparison with the oracle (the oracle's python loop cannot walk 2^24 elements in seconds)."""
    ops = _ops()
    for n_seg, T in [(2048, 128), (1 << 17, 128)]:
        N = n_seg * T
        gen = torch.Generator(device=DEV).manual_seed(5)
        x = torch.rand(N, device=DEV, generator=gen) * 2 - 1
        d = (torch.rand(N, device=DEV, generator=gen) < 0.02).float()
        d[T - 1::T] = 1
        y = ops.discount_rows(x, d, 0.99, n_seg, T)
        y2 = ops.discount_rows(2 * x, d, 0.99, n_seg, T)
        assert torch.equal(y2, 2 * y)                       # scaling by 2 is exact in fp32
        for row in (0, 1, n_seg // 2, n_seg - 1):
            sl = slice(row * T, (row + 1) * T)
            assert np.array_equal(y[sl].cpu().numpy(), O.discount_np(x[sl].cpu().numpy(), d[sl].cpu().numpy(), 0.99))
        # a done step keeps its own value
        assert torch.equal(y[d == 1], x[d == 1])


def test_moments_normalize_add():
    ops = _ops()
    x = rnd((5000,), 7, -3, 5)
    xd = x.to(DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    ops.moments(xd, sums)
    y = torch.empty_like(xd)
    ops.normalize(xd, y, sums, x.numel(), 1e-6)
    close("normalize", y, (x - x.mean()) / (x.std() + 1e-6), 2e-6, 1e-5)
    z = torch.empty_like(xd)
    ops.add(xd, y, z)
    assert torch.equal(z, xd + y)
    assert torch.allclose(torch.tensor([1., 2, 3, 4]).std(), torch.tensor(1.2909944))


# ------------------------------------------------------------------ sampler / rollout step kernels
def test_sample_action_golden(golden):
    from a2c_amd.utils import sample_action
    g = golden["g2_sample_action"]
    for i in range(int(g["n_cases"])):
        a = sample_action(torch.from_numpy(g[f"p{i}"]), torch.from_numpy(g[f"u{i}"]))
        assert np.array_equal(a.cpu().numpy(), g[f"a{i}"]), i


@pytest.mark.parametrize("A", [2, 3, 4, 6, 18])
def test_softmax_sample_vs_oracle(A):
    ops = _ops()
    B = 777
    logits = rnd((B, A + 1), 10 + A, -4, 4)          # strided rows, like the heads buffer
    u = rnd((B,), 30 + A, 0, 1)
    want = O.sample_action(F.softmax(logits[:, :A], dim=-1), u).long()
    want[want < 0] = A - 1          # the rollout sampler returns the last action where utils.sample_action falls through (-1)
    ld, ud = logits.to(DEV), u.to(DEV)
    acts = torch.full((B, 3), -7, dtype=torch.int64, device=DEV)
    probs = torch.empty(B, A, device=DEV)
    ops.softmax_sample(ld[:, :A], ud, acts.data_ptr() + 8, 3, B, A, probs=probs)
    close("probs", probs, F.softmax(logits[:, :A], dim=-1), 1e-6, 1e-5)
    got = acts[:, 1].cpu()
    mism = (got != want).nonzero().flatten()
    # a flip is only legitimate when the cumsum is within rounding of u
    cs = torch.cumsum(F.softmax(logits[:, :A], -1), -1)
    for i in mism.tolist():
        assert (cs[i] - u[i]).abs().min() < 1e-6, (i, got[i], want[i])
    assert len(mism) <= 1
    assert (acts[:, 0] == -7).all() and (acts[:, 2] == -7).all()


def test_frame_stack_push():
    ops = _ops()
    B, C, HW, T = 5, 4, 84 * 84, 3
    buf = rnd((B, T, C * HW), 40).to(DEV)           # rollout-major rows
    frames = rnd((B, HW), 41).to(DEV)
    reset = torch.tensor([0., 1, 0, 0, 1], device=DEV)
    want = buf.clone()
    prev = buf[:, 0].view(B, C, HW)
    new = torch.cat([prev[:, 1:], frames[:, None]], 1)
    new[reset.bool(), :C - 1] = 0
    want[:, 1] = new.view(B, -1)
    ops.frame_stack_push(frames, reset, buf.data_ptr(), T * C * HW, buf.data_ptr() + 4 * C * HW, T * C * HW, B, C, HW)
    assert torch.equal(buf, want)
    # unaligned / odd size path
    B, C, HW = 3, 3, 7
    buf = rnd((B, 2, C * HW), 42).to(DEV)
    frames = rnd((B, HW), 43).to(DEV)
    want = buf.clone()
    want[:, 1] = torch.cat([buf[:, 0].view(B, C, HW)[:, 1:], frames[:, None]], 1).view(B, -1)
    ops.frame_stack_push(frames, None, buf.data_ptr(), 2 * C * HW, buf.data_ptr() + 4 * C * HW, 2 * C * HW, B, C, HW)
    assert torch.equal(buf, want)


def test_rollout_record_and_bootstrap():
    ops = _ops()
    B, T, slot0, n_slots, hd = 6, 5, 2, 9, 8
    gamma = 0.99
    rewards = rnd((n_slots * T,), 50)
    dones = (rnd((n_slots * T,), 51, 0, 1) < 0.3).float()
    deltas = rnd((n_slots * T,), 52)
    rw, dn, dl = rewards.to(DEV), dones.to(DEV), deltas.to(DEV)
    val_prev = rnd((B,), 53)
    vp = val_prev.to(DEV)
    h = rnd((B, hd), 54).to(DEV)
    h0 = h.clone()
    t = 3
    rew = torch.tensor([0., 1, -1, 0, 0, 2])
    done = torch.tensor([0., 0, 0, 1, 0, 0])
    val = rnd((B, 4), 55)                      # strided value column
    de = torch.zeros(B, device=DEV)
    val_d, rew_d, done_d = val.to(DEV), rew.to(DEV), done.to(DEV)
    ops.rollout_record(rew_d, done_d, val_d.data_ptr() + 12, 4, vp, rw, dn, dl, de, h, B, T, t, slot0, gamma, True)
    for b in range(B):
        e = (slot0 + b) * T + t
        d_eff = 1.0 if (done[b] != 0 or rew[b] != 0) else 0.0
        assert rw[e].item() == rew[b].item() and dn[e].item() == d_eff
        v = val[b, 3]
        want = (rewards[e - 1] + (torch.tensor(gamma) * v) * (1 - dones[e - 1]) - val_prev[b]).float()
        assert dl[e - 1].item() == want.item(), b
        assert vp[b].item() == v.item()
        assert torch.equal(h[b], torch.zeros(hd, device=DEV) if d_eff else h0[b])
    # bootstrap
    vb = rnd((B,), 56)
    r2, d2, l2 = rw.clone(), dn.clone(), dl.clone()
    vb_d = vb.to(DEV)
    ops.rollout_bootstrap(vb_d.data_ptr(), 1, vp, r2, d2, l2, B, T, slot0, gamma)
    for b in range(B):
        e = (slot0 + b) * T + T - 1
        r = rw[e].cpu()
        if dn[e].item() == 0:
            r = (r + torch.tensor(gamma) * vb[b]).float()
        assert r2[e].item() == r.item() and d2[e].item() == 1.0
        assert l2[e].item() == (r - vp[b].cpu()).float().item()


@pytest.mark.parametrize("u8,with_src", [(True, False), (True, True), (False, True)])
def test_rollout_post_rec_equals_the_separate_launches(u8, with_src):
    """a2c_rollout_post_rec (bookkeeping + frame stack + hidden-state reset + h_states row, runner.py:199-232) against
    a2c_rollout_record + a2c_frame_stack_push[_u8] + a2c_copy_rows, every output bit for bit; h_src = where the
    previous step's cell left its new hidden rows"""
    ops = _ops()
    B, T, slot0, n_slots, hd, C, HW = 7, 5, 1, 9, 24, 4, 84 * 84
    gamma, t, S = 0.99, 2, 4 * 84 * 84
    mk = lambda: dict(rw=rnd((n_slots * T,), 50).to(DEV), dn=(rnd((n_slots * T,), 51, 0, 1) < 0.3).float().to(DEV),
                      dl=rnd((n_slots * T,), 52).to(DEV), vp=rnd((B,), 53).to(DEV), de=torch.zeros(B, device=DEV),
                      h=rnd((B, hd), 54).to(DEV), hs=torch.zeros(n_slots * T, hd, device=DEV),
                      states=rnd((n_slots * T, S), 57, 0, 1).to(DEV))
    rew = torch.tensor([0., 1, -1, 0, 0, 2, 0]).to(DEV)
    done = torch.tensor([0., 0, 0, 1, 0, 0, 1]).to(DEV)
    val = rnd((B, 4), 55).to(DEV)
    h_new = rnd((B, hd), 58).to(DEV)                         # what the cell of step t left (when with_src)
    f8 = (rnd((B, HW), 59, 0, 1) < 0.3).to(torch.uint8).to(DEV)
    f32 = f8.float()
    sp = lambda D, k: D["states"].data_ptr() + 4 * (slot0 * T + k) * S
    a, b = mk(), mk()
    # separate launches
    if with_src:
        a["h"].copy_(h_new)
    ops.rollout_record(rew, done, val.data_ptr() + 12, 4, a["vp"], a["rw"], a["dn"], a["dl"], a["de"], a["h"], B, T, t, slot0,
                       gamma, True)
    if u8:
        ops.frame_stack_push_u8(f8.data_ptr(), HW, done, sp(a, t), T * S, sp(a, t + 1), T * S, B, C, HW)
    else:
        ops.frame_stack_push(f32, done, sp(a, t), T * S, sp(a, t + 1), T * S, B, C, HW)
    ops.copy_rows(a["h"].data_ptr(), hd, a["hs"].data_ptr() + 4 * (slot0 * T + t + 1) * hd, T * hd, B, hd)
    # one launch
    ops.rollout_post_rec(rew, done, val.data_ptr() + 12, 4, b["vp"], b["rw"], b["dn"], b["dl"], T, t, slot0, gamma, True,
                         0 if u8 else f32.data_ptr(), f8.data_ptr() if u8 else 0, HW if u8 else 0, done, sp(b, t), T * S,
                         sp(b, t + 1), T * S, B, C, HW, b["de"], b["h"], b["hs"].data_ptr() + 4 * (slot0 * T + t + 1) * hd, T * hd,
                         h_src_ptr=h_new.data_ptr() if with_src else 0)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_copy_mask_permute_rows():
    ops = _ops()
    R, T, n = 5, 7, 12
    x = rnd((R, T, n), 60).to(DEV)
    y = torch.empty(T, R, n, device=DEV)
    ops.permute_rows(x, y, R, T, n)
    assert torch.equal(y, x.permute(1, 0, 2).contiguous())
    z = torch.empty(R, n, device=DEV)
    ops.copy_rows(x.data_ptr() + 4 * 2 * n, T * n, z.data_ptr(), n, R, n)
    assert torch.equal(z, x[:, 2])
    d = (rnd((R * T,), 61, 0, 1) < 0.5).float().to(DEV)
    w = z.clone()
    ops.mask_rows(w, d.data_ptr() + 4 * 3, T)
    assert torch.equal(w, z * (1 - d.view(R, T)[:, 3:4]))


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("A,norm", [(2, True), (3, True), (3, False), (4, True), (6, False), (18, True)])
def test_loss_fwd_bwd_vs_autograd(A, norm):
    ops = _ops()
    N, NG = 301, 301
    heads = rnd((N, A + 1), 70 + A, -3, 3)
    acts = (rnd((N,), 71, 0, 1) * A).long().clamp(0, A - 1)
    advs = rnd((N,), 72, -2, 2)
    rets = rnd((N,), 73, -2, 2)
    pc, vc, ec = 1.0, 0.5, 0.005
    lg = heads[:, :A].clone().requires_grad_(True)
    vl = heads[:, A].clone().requires_grad_(True)
    a_n = (advs - advs.mean()) / (advs.std() + 1e-6) if norm else advs
    lsm = F.log_softmax(lg, -1)
    entr = -ec * (lsm * F.softmax(lg, -1)).sum(-1).mean()
    pi_loss = pc * -(lsm[torch.arange(N), acts] * a_n).mean()
    val_loss = vc * F.mse_loss(vl, rets)
    (pi_loss + val_loss - entr).backward()
    hd = heads.to(DEV)
    dh = torch.zeros(N, A + 1, device=DEV)
    sums = torch.zeros(3, dtype=torch.float64, device=DEV)
    adv_sums = None
    if norm:
        adv_sums = torch.zeros(2, dtype=torch.float64, device=DEV)
        ops.moments(advs.to(DEV), adv_sums)
    ops.loss_fwd_bwd(hd[:, :A], hd[:, A], acts.to(DEV), advs.to(DEV), rets.to(DEV), adv_sums, NG, pc, vc, ec,
                     dh[:, :A], dh[:, A], sums)
    s = sums.cpu()
    assert pc * -(s[0] / NG) == pytest.approx(pi_loss.item(), rel=1e-5, abs=1e-7)
    assert vc * (s[1] / NG) == pytest.approx(val_loss.item(), rel=1e-5, abs=1e-7)
    assert -ec * (s[2] / NG) == pytest.approx(entr.item(), rel=1e-5, abs=1e-8)
    close("dlogits", dh[:, :A], lg.grad, 1e-8, 1e-4)
    close("dvals", dh[:, A], vl.grad, 1e-9, 1e-5)


# ------------------------------------------------------------------ GEMM
def _gemm_ref(A, B, tA, tB):
    a = A.double().t() if tA else A.double()
    b = B.double().t() if tB else B.double()
    return a @ b


@pytest.mark.parametrize("tA,tB", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 3, 7), (128, 128, 16), (130, 257, 100), (256, 256, 2592), (300, 4, 256),
                                   (3, 256, 1000), (200, 300, 256), (33, 95, 64), (256, 768, 288), (32, 512, 2000), (70, 130, 40)])
def test_gemm_f32(tA, tB, M, N, K):
    ops = _ops()
    A = rnd((K, M) if tA else (M, K), 80)
    B = rnd((N, K) if tB else (K, N), 81)
    want = _gemm_ref(A, B, tA, tB)
    scale = 1e-6 * K ** 0.5 + 1e-7
    Ad, Bd = A.to(DEV), B.to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(tA, tB, M, N, K, Ad.data_ptr(), A.shape[1], Bd.data_ptr(), B.shape[1], C.data_ptr(), N)
    close("plain", C, want, 5 * scale, 1e-5)
    # epilogue: bias + relu + mask + accumulate, split-K
    bias = rnd((N,), 82).to(DEV)
    mask = rnd((M, N), 83).to(DEV)
    C0 = rnd((M, N), 84).to(DEV)
    for sk in (1, 3):
        Cx = C0.clone()
        ws = torch.empty(max(1, ops.gemm_ws_bytes(M, N, sk) // 4), device=DEV)
        ops.gemm(tA, tB, M, N, K, Ad.data_ptr(), A.shape[1], Bd.data_ptr(), B.shape[1], Cx.data_ptr(), N, bias=bias,
                 relu=True, mask_ptr=mask.data_ptr(), ldmask=N, accumulate=True, splitk=sk, ws=ws)
        ref = torch.relu(C0.cpu().double() + want + bias.cpu().double()) * (mask.cpu() > 0)
        close(f"epilogue sk={sk}", Cx, ref, 5 * scale, 1e-5)


@pytest.mark.parametrize("M,N,K,sk", [(32, 2000, 28224, 32), (8, 1030, 4104, 5), (33, 1500, 4096, 7), (64, 2048, 2048, 2)])
def test_gemm_skinny_stream(M, N, K, sk):
    """rollout-batch rows against a large k-contiguous weight matrix (ConvModel proj_matrx at n_envs = 32,
    models.py:246-264): the LDS-free streaming split-K kernel, ragged M / N and a short last split"""
    ops = _ops()
    A, B = rnd((M, K), 90), rnd((N, K), 91)
    want = _gemm_ref(A, B, 0, 1)
    scale = 1e-6 * K ** 0.5 + 1e-7
    Ad, Bd = A.to(DEV), B.to(DEV)
    bias = rnd((N,), 92).to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(max(1, ops.gemm_ws_bytes(M, N, sk) // 4), device=DEV)
    ops.gemm(0, 1, M, N, K, Ad.data_ptr(), K, Bd.data_ptr(), K, C.data_ptr(), N, bias=bias, relu=True, splitk=sk, ws=ws)
    close("stream", C, torch.relu(want + bias.cpu().double()), 5 * scale, 1e-5)
    # the block-tiled kernel on the same call (A2C_NO_SKINNY_STREAM): same result up to summation order
    os.environ["A2C_NO_SKINNY_STREAM"] = "1"
    try:
        C2 = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(0, 1, M, N, K, Ad.data_ptr(), K, Bd.data_ptr(), K, C2.data_ptr(), N, bias=bias, relu=True, splitk=sk, ws=ws)
    finally:
        del os.environ["A2C_NO_SKINNY_STREAM"]
    close("tiled", C2, torch.relu(want + bias.cpu().double()), 5 * scale, 1e-5)


def test_gemm_skinny_head_paths():
    """N <= 8 layers (policy/value heads) take dedicated wave-per-row kernels"""
    ops = _ops()
    M, A, h = 1000, 3, 256
    emb = rnd((M, h), 87).to(DEV)
    Wh = rnd((A + 1, h), 88).to(DEV)
    bh = rnd((A + 1,), 89).to(DEV)
    heads = torch.zeros(M, A + 1, device=DEV)
    ops.gemm(0, 1, M, A + 1, h, emb.data_ptr(), h, Wh.data_ptr(), h, heads.data_ptr(), A + 1, bias=bh)
    close("heads fwd", heads, emb.cpu().double() @ Wh.cpu().double().t() + bh.cpu().double(), 1e-5, 1e-5)
    dh = rnd((M, A + 1), 90).to(DEV)
    demb = torch.full((M, h), float("nan"), device=DEV)
    mask = rnd((M, h), 91).to(DEV)
    ops.gemm(0, 0, M, h, A, dh.data_ptr(), A + 1, Wh.data_ptr(), h, demb.data_ptr(), h, mask_ptr=mask.data_ptr(), ldmask=h)
    ref = (dh[:, :A].cpu().double() @ Wh[:A].cpu().double()) * (mask.cpu() > 0)
    close("heads bwd_data", demb, ref, 1e-5, 1e-5)
    ops.gemm(0, 0, M, h, A, dh.data_ptr(), A + 1, Wh.data_ptr(), h, demb.data_ptr(), h, accumulate=True)
    close("heads bwd_data acc", demb, ref + dh[:, :A].cpu().double() @ Wh[:A].cpu().double(), 1e-5, 1e-5)
    dW = torch.full((A + 1, h), float("nan"), device=DEV)
    ws = torch.empty(ops.gemm_ws_bytes(A + 1, h, 2) // 4, device=DEV)
    ops.gemm(1, 0, A + 1, h, M, dh.data_ptr(), A + 1, emb.data_ptr(), h, dW.data_ptr(), h, splitk=2, ws=ws)
    want = dh.cpu().double().t() @ emb.cpu().double()
    close("heads bwd_weight", dW, want, 1e-5 * float(want.abs().max()), 1e-5)


@pytest.mark.parametrize("M,K,nslab,A", [(32, 2000, 5, 6), (256, 256, 1, 3), (7, 516, 3, 7)])
def test_heads_fused_tail_samples_and_publishes(M, K, nslab, A):
    """a2c_heads_fused_publish (the tail of a rollout forward: split-K slab sum + bias + ReLU -> [emb] -> pi / value heads
    (models.py:84-85) -> softmax sample (runner.py:94-97) -> the device relay's cmd granule, runner.py:129-131
    "pipe.send(action)"): against fp64, the sampler kernel on the same logits, and a2c_pool_publish_actions."""
    ops = _ops()
    xs = rnd((nslab, M, K), 71).to(DEV)
    bias, W, b = rnd((K,), 72).to(DEV), (rnd((A + 1, K), 73) / K ** 0.5).to(DEV), rnd((A + 1,), 74).to(DEV)
    u = rnd((M,), 75, 0, 1).to(DEV)
    emb = torch.full((M, K), float("nan"), device=DEV)
    heads = torch.full((M, 8), float("nan"), device=DEV)
    acts = torch.full((M, 2), -1, dtype=torch.int64, device=DEV)           # stride 2
    cmd = torch.zeros(M, dtype=torch.int64, device=DEV)
    seq = torch.tensor([41], dtype=torch.int32, device=DEV)
    ops.heads_fused(xs.data_ptr(), nslab, M * K, K, bias, True, emb, W, b, heads, M, u, A, acts.data_ptr(), 2,
                    publish=(cmd.data_ptr(), seq, 3))
    torch.cuda.synchronize()
    x64 = torch.relu(xs.cpu().double().sum(0) + bias.cpu().double())
    close("emb", emb, x64, 1e-5, 1e-5)
    close("heads", heads[:, :A + 1], x64 @ W.cpu().double().t() + b.cpu().double(), 2e-5, 1e-5)
    want = torch.full((M,), -1, dtype=torch.int64, device=DEV)
    ops.softmax_sample(heads[:, :A], u, want.data_ptr(), 1, M, A)
    assert torch.equal(acts[:, 0], want) and (acts[:, 1] == -1).all()
    assert torch.equal(cmd, (44 << 32) | want)
    cmd2 = torch.zeros(M, dtype=torch.int64, device=DEV)
    ops.pool_publish_actions(cmd2.data_ptr(), acts.data_ptr(), 2, M, seq, 3)
    assert torch.equal(cmd2, cmd)
    # without publish nothing else is written; publishing needs the sampler
    cmd.zero_()
    ops.heads_fused(xs.data_ptr(), nslab, M * K, K, bias, True, None, W, b, heads, M, u, A, acts.data_ptr(), 2)
    assert (cmd == 0).all()
    from a2c_amd import _lib
    with pytest.raises(_lib.A2CKernelError):
        ops.heads_fused(xs.data_ptr(), nslab, M * K, K, bias, True, None, W, b, heads, M, None, 0, 0, 0,
                        publish=(cmd.data_ptr(), seq, 3))


@pytest.mark.parametrize("M", [7, 32, 65, 100, 256])
def test_gemm_rollout_batch_forward_against_long_k_contiguous_weights(M, monkeypatch):
    """64 < M <= 256 rows against a big k-contiguous weight matrix (ConvModel's resize_emb at the rollout batch of
    configs 4 / 5): the LDS-DMA kernel (gemm_nt_kernel: all rows in one tile, split K, ragged last column tile, K tail
    shorter than a 32-deep tile) against fp64 and against the block-tiled kernel it replaces."""
    ops = _ops()
    N, K = 300, 14008                    # 3 column tiles (the last 44 wide); N * K >= 2^22; K % 32 == 24
    x = rnd((M, K + 4), 300).to(DEV)     # lda > K
    W = (rnd((N, K), 301) * 0.05).to(DEV)
    b = rnd((N,), 302).to(DEV)
    sk = 6
    ws = torch.empty(ops.gemm_ws_bytes(M, N, sk) // 4, device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(0, 1, M, N, K, x.data_ptr(), K + 4, W.data_ptr(), K, out.data_ptr(), N, bias=b, relu=True, splitk=sk, ws=ws)
    want = torch.relu(x[:, :K].cpu().double() @ W.cpu().double().t() + b.cpu().double())
    close("gemm_nt", out, want, 1e-5 * float(want.abs().max()), 1e-5)
    monkeypatch.setenv("A2C_NO_GEMM_NT", "1")
    ref = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(0, 1, M, N, K, x.data_ptr(), K + 4, W.data_ptr(), K, ref.data_ptr(), N, bias=b, relu=True, splitk=sk, ws=ws)
    close("gemm_nt vs tiled", out, ref, 1e-5 * float(want.abs().max()), 1e-5)


def test_gemm_strided_views_and_colsum():
    ops = _ops()
    M, N, K = 70, 5, 33
    X = rnd((M, K + 3), 85).to(DEV)          # lda > K
    W = rnd((N, K), 86).to(DEV)
    out = torch.zeros(M, N + 2, device=DEV)  # ldc > N
    ops.gemm(0, 1, M, N, K, X.data_ptr(), K + 3, W.data_ptr(), K, out.data_ptr() + 4, N + 2)
    close("strided", out[:, 1:N + 1], X[:, :K].cpu().double() @ W.cpu().double().t(), 1e-5, 1e-5)
    assert (out[:, 0] == 0).all() and (out[:, N + 1] == 0).all()
    cs = torch.empty(N + 2, device=DEV)
    ws = torch.empty(ops.colsum_ws_bytes(N + 2) // 4, device=DEV)
    ops.colsum(out.data_ptr(), N + 2, M, N + 2, cs, ws)
    close("colsum", cs, out.cpu().double().sum(0), 1e-5, 1e-5)


# ------------------------------------------------------------------ conv
CONV_SPECS = [  # Cin, H, W, Cout, ks, stride, pad
    (4, 84, 84, 16, 8, 4, 0), (16, 20, 20, 32, 4, 2, 0),                                              # A3CModel
    (4, 84, 84, 16, 3, 1, 1), (16, 84, 84, 24, 3, 1, 1), (24, 84, 84, 32, 3, 2, 1), (32, 42, 42, 64, 3, 2, 1),  # ConvModel
    (16, 84, 84, 24, 3, 2, 1), (24, 42, 42, 32, 3, 2, 1), (32, 21, 21, 48, 3, 2, 1), (48, 11, 11, 64, 3, 2, 1),  # GRUModel
    (4, 20, 20, 16, 3, 1, 1), (8, 13, 17, 12, 3, 2, 1), (4, 9, 9, 8, 4, 2, 0), (4, 4, 4, 16, 3, 1, 1),          # small / odd
]


@pytest.mark.parametrize("spec", CONV_SPECS, ids=[str(s) for s in CONV_SPECS])
def test_conv2d_fwd_bwd(spec):
    ops = _ops()
    Cin, H, W, Cout, ks, s, p = spec
    B = 3
    d = ops.conv_desc(*spec)
    x = rnd((B, Cin, H, W), 90, 0, 1)
    w = rnd((Cout, Cin, ks, ks), 91) / (Cin * ks * ks) ** 0.5
    bias = rnd((Cout,), 92) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, stride=s, padding=p)
    y = F.relu(pre)
    gy = rnd(tuple(y.shape), 93)
    gpre = gy * (pre > 0)
    pre.backward(gpre)
    # strided batch input (rows of a wider buffer, like states[idx*T+t])
    xbuf = torch.zeros(B, Cin * H * W + 8, device=DEV)
    xbuf[:, :Cin * H * W] = x.view(B, -1).to(DEV)
    wd, bd = w.to(DEV), bias.to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 0, wd, wf)
    ops.conv_prep(d, 1, wd, wb)
    out = torch.full(tuple(y.shape), float("nan"), device=DEV)
    ops.conv_fwd(d, xbuf.data_ptr(), xbuf.stride(0), wf, bd, True, out, B)
    close("fwd", out, y, 2e-6, 1e-5)
    # backward data (+ fused ReLU mask of the layer below: use x > 0.5 as that mask)
    dout = gpre.contiguous().to(DEV)
    mask = (x - 0.5).to(DEV)
    din = torch.full((B, Cin, H, W), float("nan"), device=DEV)
    ops.conv_bwd_data(d, dout, wb, mask, din, B)
    close("bwd_data", din, xr.grad * (x > 0.5), 2e-6, 1e-5)
    # backward weight + bias
    dW = torch.full_like(wd, float("nan"))
    db = torch.full_like(bd, float("nan"))
    ws = torch.empty(ops.conv_bwd_weight_ws_bytes(d, B) // 4, device=DEV)
    ops.conv_bwd_weight(d, xbuf.data_ptr(), xbuf.stride(0), dout, dW, db, B, ws)
    # sums over B*OH*OW (up to 21k) fp32 terms: tolerance 1e-5 of the tensor's scale
    close("bwd_weight", dW, wr.grad, 1e-5 * float(wr.grad.abs().max()), 1e-5)
    close("bwd_bias", db, br.grad, 1e-5 * float(br.grad.abs().max()), 1e-5)


def _sign_words(t):
    """(B, C, H, W) bool -> (B, C*H*ceil(W/32)) int32: bit x & 31 of word [c][y][x >> 5] (a2c_conv2d_fwd_signs' layout)"""
    B, C, H, W = t.shape
    RW = (W + 31) // 32
    pad = torch.zeros(B, C, H, RW * 32, dtype=torch.bool)
    pad[..., :W] = t
    words = (pad.reshape(B, C, H, RW, 32).long() * (2 ** torch.arange(32, dtype=torch.int64))).sum(-1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).int().reshape(B, -1).contiguous()


SIGN_SPECS = [s for s in CONV_SPECS if s[4] == 3 and s[1] in (84, 42, 21, 11)]


@pytest.mark.parametrize("spec", SIGN_SPECS, ids=[str(s) for s in SIGN_SPECS])
@pytest.mark.parametrize("B", [3, 300])
def test_conv2d_sign_words(spec, B):
    """The ReLU mask as one bit per activation: a2c_conv2d_fwd_signs leaves (out > 0) next to the output (which is the
    plain forward's, bit for bit), a2c_conv2d_bwd_data_signs masks with those bits exactly as a2c_conv2d_bwd_data does
    with the float activation.  B = 300: several bands per workgroup (the staged kernels' pipeline), rows of a wider
    sign buffer (whatever lies between the rows is not touched)."""
    ops = _ops()
    Cin, H, W, Cout, ks, s, p = spec
    d = ops.conv_desc(*spec)
    x = rnd((B, Cin, H, W), 190, 0, 1)
    w = rnd((Cout, Cin, ks, ks), 191) / (Cin * ks * ks) ** 0.5
    bias = rnd((Cout,), 192) * 0.1
    xd, wd, bd = x.to(DEV), w.to(DEV), bias.to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 0, wd, wf)
    ops.conv_prep(d, 1, wd, wb)
    nsw = ops.conv_sign_words(d)
    assert nsw == (Cout * d.OH * ((d.OW + 31) // 32) if H != 11 else 0)      # (the last layer's output feeds no backward-data)
    ref = torch.empty(B, Cout, d.OH, d.OW, device=DEV)
    ops.conv_fwd(d, xd.data_ptr(), Cin * H * W, wf, bd, True, ref, B)
    close("fwd", ref, F.relu(F.conv2d(x, w, bias, stride=s, padding=p)), 2e-6, 1e-5)
    if nsw:
        out = torch.full_like(ref, float("nan"))
        sg = torch.full((B, nsw + 5), -7, dtype=torch.int32, device=DEV)
        ops.conv_fwd_signs(d, xd.data_ptr(), Cin * H * W, wf, bd, True, out, sg.data_ptr(), nsw + 5, B)
        assert torch.equal(out, ref)
        assert torch.equal(sg[:, :nsw].cpu(), _sign_words(ref.cpu() > 0)) and bool((sg[:, nsw:] == -7).all())
    if not ops.conv_bwd_data_signs_supported(d):
        return
    dout = rnd((B, Cout, d.OH, d.OW), 193).to(DEV)
    mask = rnd((B, Cin, H, W), 194)
    words = _sign_words(mask > 0).to(DEV)
    din_f = torch.full((B, Cin, H, W), float("nan"), device=DEV)
    din_s = torch.full((B, Cin, H, W), float("nan"), device=DEV)
    ops.conv_bwd_data(d, dout, wb, mask.to(DEV), din_f, B)
    ops.conv_bwd_data_signs(d, dout, wb, words, din_s, B)
    # even images: the float-mask and the sign-word kernels are the same streaming family (same summation order): bit for bit.
    # The odd images (21 <- 11, 11 <- 6: GRUModel conv4 / conv5) have a sign-word kernel only; its float-mask twin is conv.hip's generic band
    # kernel, another order of the same sums.
    odd = H % 2 == 1
    if odd:
        close("bwd_data signs vs float mask (generic kernel)", din_s, din_f, 2e-6 * float(din_f.abs().max()), 1e-5)
        assert bool(((din_s != 0) == (din_f != 0)).all()) or float((din_s - din_f).abs().max()) < 1e-6
    else:
        assert torch.equal(din_s, din_f)
    want = F.conv_transpose2d(dout[:3].cpu(), w, stride=s, padding=p, output_padding=(H + 2 * p - ks) % s)
    close("bwd_data", din_s[:3], want * (mask[:3] > 0), 2e-6, 1e-5)
    if B == 3:
        ops.conv_bwd_data(d, dout, wb, None, din_f, B)
        ops.conv_bwd_data_signs(d, dout, wb, None, din_s, B)
        if odd:
            close("bwd_data no mask, both kernels", din_s, din_f, 2e-6 * float(din_f.abs().max()), 1e-5)
        else:
            assert torch.equal(din_s, din_f)
        close("bwd_data no mask", din_s, want, 2e-6, 1e-5)


@pytest.mark.parametrize("s2,hw,B", [(2, (84, 84), 5), (1, (84, 84), 3), (2, (20, 24), 37), (1, (12, 16), 300)])
def test_conv2d_bwd_data_fused_with_first_layer_weight_gradient(s2, hw, B, monkeypatch):
    """conv1 (4 -> 16, 3x3/1/1) + ReLU + conv2 (16 -> 24, 3x3/s2 (GRUModel) or s1 (ConvModel)): da1 = conv2's masked
    input gradient only feeds conv1's weight gradient (models.py:196-215), so a2c_conv2d_bwd_data_w1 keeps it on
    chip.  Against torch autograd of the two-layer stack in fp64, and against the two separate C-ABI calls.
    Opt-in (A2C_FUSE_W1=1): measured slower than the two passes on MI355X."""
    ops = _ops()
    monkeypatch.setenv("A2C_FUSE_W1", "1")
    H, W = hw
    d1 = ops.conv_desc(4, H, W, 16, 3, 1, 1)
    d2 = ops.conv_desc(16, H, W, 24, 3, s2, 1)
    nb = ops.conv_bwd_data_w1_ws_bytes(d2, d1, B)
    assert nb > 0
    x = rnd((B, 4, H, W), 60, 0, 1)
    w1 = rnd((16, 4, 3, 3), 61) / 6.0
    b1 = rnd((16,), 62) * 0.1
    w2 = rnd((24, 16, 3, 3), 63) / 12.0
    w1r, b1r = w1.double().requires_grad_(True), b1.double().requires_grad_(True)
    a1 = F.relu(F.conv2d(x.double(), w1r, b1r, stride=1, padding=1))
    y = F.conv2d(a1, w2.double(), None, stride=s2, padding=1)
    gy = rnd(tuple(y.shape), 64)
    y.backward(gy.double())
    xd, w2d, dout = x.to(DEV), w2.to(DEV), gy.contiguous().to(DEV)
    a1d = a1.detach().float().to(DEV)                     # the mask: conv1's ReLU output
    wb = torch.empty(ops.conv_prep_floats(d2, 1), device=DEV)
    ops.conv_prep(d2, 1, w2d, wb)
    dW1 = torch.full((16, 4, 3, 3), float("nan"), device=DEV)
    db1 = torch.full((16,), float("nan"), device=DEV)
    ws = torch.empty(nb // 4, device=DEV)
    ops.conv_bwd_data_w1(d2, dout, wb, a1d, d1, xd.data_ptr(), 4 * H * W, dW1, db1, B, ws)
    sw, sb = float(w1r.grad.abs().max()), float(b1r.grad.abs().max())
    close("fused dW1", dW1, w1r.grad, 1e-5 * sw, 1e-5)
    close("fused db1", db1, b1r.grad, 1e-5 * sb, 1e-5)
    # the two separate passes
    da1 = torch.empty(B, 16, H, W, device=DEV)
    ops.conv_bwd_data(d2, dout, wb, a1d, da1, B)
    dW1s, db1s = torch.empty_like(dW1), torch.empty_like(db1)
    ws2 = torch.empty(ops.conv_bwd_weight_ws_bytes(d1, B) // 4, device=DEV)
    ops.conv_bwd_weight(d1, xd.data_ptr(), 4 * H * W, da1, dW1s, db1s, B, ws2)
    close("fused vs separate dW1", dW1, dW1s, 2e-6 * sw, 1e-5)
    close("fused vs separate db1", db1, db1s, 2e-6 * sb, 1e-5)
    # layers it does not apply to report 0 (callers fall back to the two passes)
    assert ops.conv_bwd_data_w1_ws_bytes(ops.conv_desc(24, 42, 42, 32, 3, 2, 1), ops.conv_desc(16, 84, 84, 24, 3, 2, 1), B) == 0


@pytest.mark.parametrize("spec", [(16, 84, 84, 24, 3, 2, 1), (24, 42, 42, 32, 3, 2, 1), (32, 21, 21, 48, 3, 2, 1),
                                  (16, 84, 84, 24, 3, 1, 1)], ids=str)
def test_conv2d_tile_height_does_not_change_a_bit(spec, monkeypatch):
    """The tile tuners (bwd_band_tuned / conv_fwd_tuned) pick band / tile heights by timing.  That is only sound
    because the height changes the tiling and nothing else: every output element keeps its tap and channel order.
    Forced heights / LDS budgets must therefore agree BIT FOR BIT (and the tuned call with them)."""
    ops = _ops()
    Cin, H, W, Cout, ks, s, p = spec
    B = 40
    d = ops.conv_desc(*spec)
    x = rnd((B, Cin, H, W), 70, 0, 1).to(DEV)
    w = (rnd((Cout, Cin, ks, ks), 71) / (Cin * ks * ks) ** 0.5).to(DEV)
    bias = (rnd((Cout,), 72) * 0.1).to(DEV)
    dout = rnd((B, Cout, d.OH, d.OW), 73).to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 0, w, wf)
    ops.conv_prep(d, 1, w, wb)

    def bwd():
        din = torch.full((B, Cin, H, W), float("nan"), device=DEV)
        ops.conv_bwd_data(d, dout, wb, x - 0.5, din, B)
        return din

    def fwd():
        out = torch.full((B, Cout, d.OH, d.OW), float("nan"), device=DEV)
        ops.conv_fwd(d, x.data_ptr(), Cin * H * W, wf, bias, True, out, B)
        return out

    monkeypatch.setenv("A2C_NO_TUNE", "1")
    ref_b, ref_f = bwd(), fwd()
    assert torch.isfinite(ref_b).all() and torch.isfinite(ref_f).all()
    for ty in (2, 4, 6, 10, 12):
        monkeypatch.setenv("A2C_BAND_TY", str(ty))
        assert torch.equal(bwd(), ref_b), ty
    # the whole sample as one band (what the tuner picks for the 21 x 21 layer at update batch): its 32 input channels then
    # go to two workgroups of 16 (two resident per CU instead of one); A2C_NO_BAND_GROUPS=1 = one workgroup, all channels
    monkeypatch.setenv("A2C_BAND_TY", str((H + s - 1) // s * s))
    assert torch.equal(bwd(), ref_b)
    monkeypatch.setenv("A2C_NO_BAND_GROUPS", "1")
    assert torch.equal(bwd(), ref_b)
    monkeypatch.delenv("A2C_NO_BAND_GROUPS")
    monkeypatch.delenv("A2C_BAND_TY")
    for kb in (24, 32, 48, 96, 128):
        monkeypatch.setenv("A2C_IGEMM_LDS_KB", str(kb))
        monkeypatch.setenv("A2C_RUN3_LDS_KB", str(kb))
        assert torch.equal(fwd(), ref_f), kb
        assert torch.equal(bwd(), ref_b), kb
    monkeypatch.delenv("A2C_IGEMM_LDS_KB")
    monkeypatch.delenv("A2C_RUN3_LDS_KB")
    monkeypatch.delenv("A2C_NO_TUNE")            # the tuned calls (large enough to be tuned: B * planes >= 2^24 / 2^16)
    assert torch.equal(fwd(), ref_f) and torch.equal(fwd(), ref_f)
    assert torch.equal(bwd(), ref_b) and torch.equal(bwd(), ref_b)


def test_conv2d_many_samples_persistent_grid():
    """more tiles than workgroups: exercises the grid-stride / persistent accumulation paths"""
    ops = _ops()
    spec = (4, 20, 20, 16, 3, 1, 1)
    B = 2100
    d = ops.conv_desc(*spec)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(B, 4, 20, 20, generator=gen)
    w = (torch.rand(16, 4, 3, 3, generator=gen) - 0.5) * 0.3
    dout = torch.rand(B, 16, 20, 20, generator=gen) - 0.5
    wr = w.double().requires_grad_(True)          # fp64 reference: 840k-term sums
    y = F.conv2d(x.double(), wr, None, stride=1, padding=1)
    y.backward(dout.double())
    xd, wd, dd = x.to(DEV), w.to(DEV), dout.to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    ops.conv_prep(d, 0, wd, wf)
    out = torch.empty(B, 16, 20, 20, device=DEV)
    ops.conv_fwd(d, xd.data_ptr(), 1600, wf, None, False, out, B)
    close("fwd", out, y, 2e-6, 1e-5)
    dW, db = torch.empty_like(wd), torch.empty(16, device=DEV)
    ws = torch.empty(ops.conv_bwd_weight_ws_bytes(d, B) // 4, device=DEV)
    ops.conv_bwd_weight(d, xd.data_ptr(), 1600, dd, dW, db, B, ws)
    close("bwd_weight", dW, wr.grad, 1e-5 * float(wr.grad.abs().max()), 1e-5)
    close("bwd_bias", db, dout.double().sum((0, 2, 3)), 1e-5 * float(dout.double().sum((0, 2, 3)).abs().max()), 1e-5)


def test_conv2d_streaming_forward_matches_tiled_kernel(monkeypatch):
    """A3C conv1 at update-sized batch goes through the persistent streaming kernel; the tiled
    kernel walks K in the same (c4, ky, kx) order, so the two must agree bit for bit -- except the
    25th (leftover) 16-pixel tile, whose 8 kernel rows are summed by 8 waves and combined in a fixed
    order (fp32 re-association: 1e-6).  Batch not a multiple of the grid, strided input rows, with
    and without bias / ReLU."""
    ops = _ops()
    spec = (4, 84, 84, 16, 8, 4, 0)
    B = 2048 + 77
    d = ops.conv_desc(*spec)
    gen = torch.Generator().manual_seed(11)
    xbuf = torch.zeros(B, 4 * 84 * 84 + 12, device=DEV)
    xbuf[:, :4 * 84 * 84] = (torch.rand(B, 4 * 84 * 84, generator=gen) < 0.3).float().to(DEV) * \
        torch.rand(B, 4 * 84 * 84, generator=gen).to(DEV)
    w = ((torch.rand(16, 4, 8, 8, generator=gen) - 0.5) * 0.2).to(DEV)
    bias = ((torch.rand(16, generator=gen) - 0.5) * 0.2).to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    ops.conv_prep(d, 0, w, wf)
    for b_, relu in ((bias, True), (None, False)):
        got = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
        ops.conv_fwd(d, xbuf.data_ptr(), xbuf.stride(0), wf, b_, relu, got, B)
        monkeypatch.setenv("A2C_NO_STREAM", "1")
        want = torch.empty(B, 16, 20, 20, device=DEV)
        ops.conv_fwd(d, xbuf.data_ptr(), xbuf.stride(0), wf, b_, relu, want, B)
        monkeypatch.delenv("A2C_NO_STREAM")
        torch.cuda.synchronize()
        gf, wf_ = got.view(B, 16, 400), want.view(B, 16, 400)
        assert torch.equal(gf[:, :, :384], wf_[:, :, :384])
        close("leftover tile", gf[:, :, 384:], wf_[:, :, 384:], 1e-6, 1e-6)
    ref = F.conv2d(xbuf[:64, :4 * 84 * 84].view(64, 4, 84, 84).cpu().double(), w.cpu().double(), None, stride=4)
    close("fwd vs fp64", got[:64], ref, 1e-5, 1e-5)


def test_conv2d_streaming_backward_weight(monkeypatch):
    """A3C conv1 weight gradient at update-sized batch (persistent streaming kernel) against an fp64
    reference and against the tiled kernel (same sums, different association over samples)."""
    ops = _ops()
    spec = (4, 84, 84, 16, 8, 4, 0)
    B = 2048 + 77
    d = ops.conv_desc(*spec)
    gen = torch.Generator().manual_seed(12)
    x = (torch.rand(B, 4, 84, 84, generator=gen) < 0.3).float() * torch.rand(B, 4, 84, 84, generator=gen)
    dout = (torch.rand(B, 16, 20, 20, generator=gen) - 0.5) * (torch.rand(B, 16, 20, 20, generator=gen) < 0.5).float()
    xbuf = torch.zeros(B, 4 * 84 * 84 + 12, device=DEV)
    xbuf[:, :4 * 84 * 84] = x.view(B, -1).to(DEV)
    dd = dout.to(DEV)
    # fp64 reference of dW = sum_b corr(x_b, dout_b) on a few hundred samples is too slow on CPU for all B:
    # use the unfold identity on the device in fp64 instead
    cols = F.unfold(x.to(DEV).double(), kernel_size=8, stride=4)                  # (B, 256, 400)
    want = torch.einsum("bcp,bkp->ck", dd.double().view(B, 16, 400), cols).view(16, 4, 8, 8)
    want_b = dd.double().sum((0, 2, 3))
    ws = torch.empty(max(1, ops.conv_bwd_weight_ws_bytes(d, B) // 4), device=DEV)
    dW, db = torch.full((16, 4, 8, 8), float("nan"), device=DEV), torch.full((16,), float("nan"), device=DEV)
    ops.conv_bwd_weight(d, xbuf.data_ptr(), xbuf.stride(0), dd, dW, db, B, ws)
    monkeypatch.setenv("A2C_NO_STREAM", "1")
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    ops.conv_bwd_weight(d, xbuf.data_ptr(), xbuf.stride(0), dd, dW2, db2, B, ws)
    monkeypatch.delenv("A2C_NO_STREAM")
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    close("dW vs fp64", dW, want, 1e-5 * scale, 1e-5)
    close("db vs fp64", db, want_b, 1e-5 * float(want_b.abs().max()), 1e-5)
    close("dW vs tiled", dW, dW2, 1e-5 * scale, 1e-5)
    close("db vs tiled", db, db2, 1e-5 * float(want_b.abs().max()), 1e-5)


@pytest.mark.parametrize("B", [4096 + 54, 4096 + 55])
def test_conv2d_streaming_forward_conv2_matches_tiled_kernel(monkeypatch, B):
    """A3C conv2 forward at update-sized batch (two samples per iteration of the persistent kernel;
    odd batch: the last pair is one sample twice) == the tiled kernel bit for bit."""
    ops = _ops()
    spec = (16, 20, 20, 32, 4, 2, 0)
    d = ops.conv_desc(*spec)
    gen = torch.Generator().manual_seed(14)
    x = (torch.rand(B, 16, 20, 20, generator=gen) * (torch.rand(B, 16, 20, 20, generator=gen) < 0.5).float()).to(DEV)
    w = ((torch.rand(32, 16, 4, 4, generator=gen) - 0.5) * 0.2).to(DEV)
    bias = ((torch.rand(32, generator=gen) - 0.5) * 0.2).to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    ops.conv_prep(d, 0, w, wf)
    got = torch.full((B, 32, 9, 9), float("nan"), device=DEV)
    ops.conv_fwd(d, x.data_ptr(), 6400, wf, bias, True, got, B)
    monkeypatch.setenv("A2C_NO_STREAM", "1")
    want = torch.empty(B, 32, 9, 9, device=DEV)
    ops.conv_fwd(d, x.data_ptr(), 6400, wf, bias, True, want, B)
    monkeypatch.delenv("A2C_NO_STREAM")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = F.relu(F.conv2d(x[:64].cpu().double(), w.cpu().double(), bias.cpu().double(), stride=2))
    close("fwd vs fp64", got[:64], ref, 1e-5, 1e-5)


@pytest.mark.parametrize("with_mask", [True, False])
def test_conv2d_streaming_backward_data_matches_tiled_kernel(monkeypatch, with_mask):
    """A3C conv2 input gradient at update-sized batch (persistent streaming kernel) == the tiled
    kernel bit for bit (same (tap, channel-quad) order of the fp32 MFMA chain), and == fp64 to 1e-5."""
    ops = _ops()
    monkeypatch.setenv("A2C_BWD_X6", "0")        # (the fp32 MFMA streaming kernel; the bf16 x 6 form has its own test)
    spec = (16, 20, 20, 32, 4, 2, 0)
    B = 2048 + 55
    d = ops.conv_desc(*spec)
    gen = torch.Generator().manual_seed(13)
    w = ((torch.rand(32, 16, 4, 4, generator=gen) - 0.5) * 0.2)
    dout = ((torch.rand(B, 32, 9, 9, generator=gen) - 0.5) * (torch.rand(B, 32, 9, 9, generator=gen) < 0.6).float())
    mask = (torch.rand(B, 16, 20, 20, generator=gen) - 0.4).to(DEV) if with_mask else None
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 1, w.to(DEV), wb)
    dd = dout.to(DEV)
    got = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
    ops.conv_bwd_data(d, dd, wb, mask, got, B)
    monkeypatch.setenv("A2C_NO_STREAM", "1")
    want = torch.empty(B, 16, 20, 20, device=DEV)
    ops.conv_bwd_data(d, dd, wb, mask, want, B)
    monkeypatch.delenv("A2C_NO_STREAM")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = F.conv_transpose2d(dout[:32].double(), w.double(), stride=2)
    if with_mask:
        ref = ref * (mask[:32].cpu() > 0)
    close("vs fp64", got[:32], ref, 1e-5 * float(ref.abs().max()), 1e-5)


def _lanemask_ref(act):
    """(B, n) float -> (B, n/64) int64: bit (i & 7) of byte (i >> 3) of a row = (act[i] > 0)
    (include/a2c_mi355x.h: a2c_conv2d_bwd_data_lanemask)"""
    bits = np.packbits((act > 0).numpy().astype(np.uint8), axis=1, bitorder="little")
    return torch.from_numpy(bits.copy().view(np.int64))


@pytest.mark.parametrize("B", [2048 + 55, 4096])
def test_conv2d_streaming_backward_data_with_lane_masks(monkeypatch, B):
    """round 6: the streaming backward-data kernel of A3CModel's conv2 takes its ReLU mask as LANE MASKS (800 B per sample
    instead of the 25.6 KB activation row): a2c_lanemask_from_act == the layout's definition, and the masked dX equals the
    float-mask call bit for bit -- on the second form of the kernel (two dX images, default) and against the first
    (A2C_BWD_STREAM_V1=1)."""
    ops = _ops()
    monkeypatch.setenv("A2C_BWD_X6", "0")                    # (the fp32 MFMA kernels; the bf16 x 6 form has its own test below)
    d = ops.conv_desc(16, 20, 20, 32, 4, 2, 0)
    assert ops.conv_bwd_data_lanemask_supported(d, B) and not ops.conv_bwd_data_lanemask_supported(d, 64)
    gen = torch.Generator().manual_seed(14)
    w = ((torch.rand(32, 16, 4, 4, generator=gen) - 0.5) * 0.2)
    dout = ((torch.rand(B, 32, 9, 9, generator=gen) - 0.5) * (torch.rand(B, 32, 9, 9, generator=gen) < 0.6).float()).to(DEV)
    act = torch.relu(torch.rand(B, 16, 20, 20, generator=gen) - 0.4)
    act[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-45, float("nan")])       # zero, negative zero, a denormal, NaN: (x > 0)
    actd = act.to(DEV)
    lm = torch.zeros(B, 6400 // 64, dtype=torch.int64, device=DEV)
    ops.lanemask_from_act(actd, lm)
    torch.cuda.synchronize()
    assert torch.equal(lm.cpu()[:64], _lanemask_ref(act.reshape(B, -1)[:64]))
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 1, w.to(DEV), wb)
    got = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
    ops.conv_bwd_data_lanemask(d, dout, wb, lm, got, B)
    want = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
    ops.conv_bwd_data(d, dout, wb, actd, want, B)
    for order in ("0", "1", "2"):                            # the opt-in third form (one barrier per sample), its three schedules
        monkeypatch.setenv("A2C_BWD_STREAM_FORM", "3")
        monkeypatch.setenv("A2C_BS3_ORDER", order)
        f3 = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
        ops.conv_bwd_data_lanemask(d, dout, wb, lm, f3, B)
        torch.cuda.synchronize()
        assert torch.equal(got, f3), order
    monkeypatch.delenv("A2C_BWD_STREAM_FORM")
    monkeypatch.delenv("A2C_BS3_ORDER")
    monkeypatch.setenv("A2C_BWD_STREAM_V1", "1")
    old = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
    ops.conv_bwd_data(d, dout, wb, actd, old, B)
    with pytest.raises(RuntimeError):
        ops.conv_bwd_data_lanemask(d, dout, wb, lm, got.clone(), B)            # the first form does not read lane masks
    monkeypatch.delenv("A2C_BWD_STREAM_V1")
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(want, old)
    assert not bool(torch.isnan(got).any())


@pytest.mark.parametrize("B", [2048 + 55, 4096])
def test_conv2d_backward_data_on_the_bf16_pipe_is_the_fp32_sum(monkeypatch, B):
    """bwd_x6_kernel (default for A3CModel's conv2 at streaming batch, mask as bits or none): dOut and the weights split into
    three bf16 pieces each (exact), the six piece products with qa + qb <= 2 on the bf16 pipe, fp32 sums -- against the fp64
    transposed convolution no worse than 1.5 x the fp32 MFMA kernel's error (A2C_BWD_X6=0), masks applied exactly, every
    sample written (ragged batch), bit-identical run to run."""
    ops = _ops()
    d = ops.conv_desc(16, 20, 20, 32, 4, 2, 0)
    gen = torch.Generator().manual_seed(21)
    w = ((torch.rand(32, 16, 4, 4, generator=gen) - 0.5) * 0.2)
    dout = ((torch.rand(B, 32, 9, 9, generator=gen) - 0.5) * (torch.rand(B, 32, 9, 9, generator=gen) < 0.6).float())
    dout[1] = torch.randn(32, 9, 9, generator=gen) * 1e3                      # a dense sample at another scale
    act = torch.relu(torch.rand(B, 16, 20, 20, generator=gen) - 0.4)
    doutd, actd = dout.to(DEV), act.to(DEV)
    lm = torch.zeros(B, 6400 // 64, dtype=torch.int64, device=DEV)
    ops.lanemask_from_act(actd, lm)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 1, w.to(DEV), wb)
    res = {}
    for x6 in ("1", "0"):
        monkeypatch.setenv("A2C_BWD_X6", x6)
        a = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
        ops.conv_bwd_data_lanemask(d, doutd, wb, lm, a, B)
        b = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
        ops.conv_bwd_data(d, doutd, wb, None, b, B)
        a2 = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
        ops.conv_bwd_data_lanemask(d, doutd, wb, lm, a2, B)
        torch.cuda.synchronize()
        assert torch.equal(a, a2) and not bool(torch.isnan(a).any()) and not bool(torch.isnan(b).any())
        assert torch.equal(a, b * (actd > 0))                                 # the mask bits, exactly
        res[x6] = b.cpu()
    assert not torch.equal(res["1"], res["0"])                                # (the bf16 kernel did run)
    idx = torch.cat([torch.arange(0, 48), torch.arange(B - 16, B)])
    ref = torch.nn.functional.conv_transpose2d(dout[idx].double(), w.double(), stride=2)
    for i in range(len(idx)):                                                 # per sample: the scales differ
        rms = float(ref[i].pow(2).mean().sqrt())
        e6 = float((res["1"][idx[i]].double() - ref[i]).pow(2).mean().sqrt()) / rms
        e32 = float((res["0"][idx[i]].double() - ref[i]).pow(2).mean().sqrt()) / rms
        assert e6 <= 1.5 * e32 + 1e-7, (i, e6, e32)


@pytest.mark.parametrize("B,pad", [(2048 + 55, 0), (4096, 64)])
def test_conv2d_weight_gradient_on_the_bf16_pipe_is_the_fp32_sum(monkeypatch, B, pad):
    """wgrad_x6_kernel (default for A3CModel's conv2 at streaming batch): dOut and the activations split into three bf16 pieces
    each (exact), the six piece products with qa + qb <= 2 on the bf16 pipe with k = pixels (row groups, the ox = 8 column
    group, the corner), fp32 sums -- against the fp64 weight gradient no worse than 1.5 x the fp32 MFMA kernel's error
    (A2C_WGRAD_X6=0); db too; strided activation rows; bit-identical run to run."""
    ops = _ops()
    d = ops.conv_desc(16, 20, 20, 32, 4, 2, 0)
    gen = torch.Generator().manual_seed(22)
    a1 = torch.relu(torch.rand(B, 6400 + pad, generator=gen) - 0.3).to(DEV)
    dout = ((torch.rand(B, 32, 9, 9, generator=gen) - 0.5) * (torch.rand(B, 32, 9, 9, generator=gen) < 0.6).float()).to(DEV)
    ws = torch.empty(ops.conv_bwd_weight_ws_bytes(d, B) // 4 + 1, device=DEV)
    res = {}
    for x6 in ("1", "0"):
        monkeypatch.setenv("A2C_WGRAD_X6", x6)
        outs = []
        for _ in range(2):
            dW = torch.full((32, 16, 4, 4), float("nan"), device=DEV)
            db = torch.full((32,), float("nan"), device=DEV)
            ops.conv_bwd_weight(d, a1.data_ptr(), 6400 + pad, dout, dW, db, B, ws)
            torch.cuda.synchronize()
            outs.append((dW.clone(), db.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        res[x6] = (outs[0][0].double().reshape(32, 256), outs[0][1].double())
    assert not torch.equal(res["1"][0], res["0"][0])                          # (the bf16 kernel did run)
    monkeypatch.setenv("A2C_WGRAD_X6", "2")               # the opt-in form with the conversion under the matrix phase: the same dW
    dW2, db2 = torch.full((32, 16, 4, 4), float("nan"), device=DEV), torch.full((32,), float("nan"), device=DEV)
    ops.conv_bwd_weight(d, a1.data_ptr(), 6400 + pad, dout, dW2, db2, B, ws)
    torch.cuda.synchronize()
    assert torch.equal(dW2.double().reshape(32, 256), res["1"][0])
    close("db, pipelined form", db2, res["1"][1], 1e-5 * float(res["1"][1].abs().max()) + 1e-4, 1e-5)
    x = a1[:, :6400].reshape(B, 16, 20, 20).double()
    ref = torch.zeros(32, 256, dtype=torch.float64, device=DEV)
    for i in range(0, B, 512):
        patches = F.unfold(x[i:i + 512], 4, stride=2)                         # (b, 256, 81), k = (ci, ky, kx)
        ref += torch.einsum("bcp,bkp->ck", dout[i:i + 512].reshape(-1, 32, 81).double(), patches)
    rms = float(ref.pow(2).mean().sqrt())
    e6 = float((res["1"][0] - ref).pow(2).mean().sqrt()) / rms
    e32 = float((res["0"][0] - ref).pow(2).mean().sqrt()) / rms
    assert e6 <= 1.5 * e32 + 1e-7, (e6, e32)
    dbref = dout.double().sum(dim=(0, 2, 3))
    close("db vs fp64", res["1"][1].float(), dbref, 1e-5 * float(dbref.abs().max()) + 1e-4, 1e-5)


@pytest.mark.parametrize("B,A", [(2048 + 55, 3), (4096, 4), (2500, 1)])
def test_conv2_backward_passes_form_their_dout_from_the_rank_a_head(B, A):
    """a2c_conv2d_bwd_data_lanemask_rank / a2c_conv2d_bwd_weight_rank (A3CModel's update): dOut = (dl . Wc) * (a2 > 0) formed inside
    the two bf16-pipe kernels while they stage a sample == the same kernels reading the tensor a2c_small_n_bwd_data_bits wrote,
    bit for bit (dX, dW, db); strided dl rows, ragged batch, 1 / 3 / 4 logits."""
    ops = _ops()
    d = ops.conv_desc(16, 20, 20, 32, 4, 2, 0)
    assert ops.conv_bwd_rank_supported(d, A, B) and not ops.conv_bwd_rank_supported(d, 5, B) and not ops.conv_bwd_rank_supported(d, A, 64)
    gen = torch.Generator().manual_seed(23)
    w = ((torch.rand(32, 16, 4, 4, generator=gen) - 0.5) * 0.2).to(DEV)
    dlb = (torch.rand(B, A + 1, generator=gen) - 0.5).to(DEV)                # [dl | dv] rows: stride A + 1
    Wc = (torch.rand(A + 1, 2592, generator=gen) - 0.5).to(DEV)
    a2 = torch.relu(torch.rand(B, 2592, generator=gen) - 0.5)
    a1 = torch.relu(torch.rand(B, 6400, generator=gen) - 0.3).to(DEV)
    mb = torch.from_numpy(np.packbits((a2 > 0).numpy().astype(np.uint8), axis=1, bitorder="little")).to(DEV)
    lm = torch.zeros(B, 100, dtype=torch.int64, device=DEV)
    ops.lanemask_from_act(a1, lm)
    wb = torch.empty(ops.conv_prep_floats(d, 1), device=DEV)
    ops.conv_prep(d, 1, w, wb)
    ws = torch.empty(ops.conv_bwd_weight_ws_bytes(d, B) // 4 + 1, device=DEV)
    da2 = torch.empty(B, 32, 9, 9, device=DEV)
    ops.small_n_bwd_data_bits(dlb, A + 1, Wc, da2, 2592, mb, B, A, 2592)
    assert float(da2.abs().max()) > 0
    dX0 = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
    ops.conv_bwd_data_lanemask(d, da2, wb, lm, dX0, B)
    dW0, db0 = torch.empty(32, 16, 4, 4, device=DEV), torch.empty(32, device=DEV)
    ops.conv_bwd_weight(d, a1.data_ptr(), 6400, da2, dW0, db0, B, ws)
    dX1 = torch.full((B, 16, 20, 20), float("nan"), device=DEV)
    ops.conv_bwd_data_lanemask_rank(d, dlb, A + 1, A, Wc, mb, mb.stride(0), wb, lm, dX1, B)
    dW1, db1 = torch.full((32, 16, 4, 4), float("nan"), device=DEV), torch.full((32,), float("nan"), device=DEV)
    ops.conv_bwd_weight_rank(d, a1.data_ptr(), 6400, dlb, A + 1, A, Wc, mb, mb.stride(0), dW1, db1, B, ws)
    torch.cuda.synchronize()
    assert torch.equal(dX0, dX1)
    assert torch.equal(dW0, dW1) and torch.equal(db0, db1)


@pytest.mark.parametrize("M,N", [(257, 3), (4096, 4), (5, 8)])
def test_small_n_bwd_data_with_mask_bits_equals_the_float_mask_path(M, N):
    """a2c_small_n_bwd_data_bits (A3CModel's da2 = (dl . Wc[:A]) * (a2 > 0) from the ring kernel's a2 mask bits) == the same
    product through a2c_gemm_f32 with the fp32 activation as the mask, bit for bit; strided dy rows (dl is a view of [dl | dv])"""
    ops = _ops()
    K = 2592
    gen = torch.Generator().manual_seed(15)
    dyb = (torch.rand(M, N + 1, generator=gen) - 0.5).to(DEV)
    W = (torch.rand(N, K, generator=gen) - 0.5).to(DEV)
    act = torch.relu(torch.rand(M, K, generator=gen) - 0.5)
    act[0, :3] = torch.tensor([0.0, -0.0, float("nan")])
    mb = torch.from_numpy(np.packbits((act > 0).numpy().astype(np.uint8), axis=1, bitorder="little")).to(DEV)
    got = torch.full((M, K), float("nan"), device=DEV)
    ops.small_n_bwd_data_bits(dyb[:, :N], dyb.stride(0), W, got, K, mb, M, N, K)
    want = torch.full((M, K), float("nan"), device=DEV)
    actd = act.to(DEV)
    ops.gemm(0, 0, M, K, N, dyb.data_ptr(), dyb.stride(0), W.data_ptr(), K, want.data_ptr(), K, mask_ptr=actd.data_ptr(), ldmask=K)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and not bool(torch.isnan(got).any())
    ref = (dyb[:, :N].double().cpu() @ W.double().cpu()) * (act > 0)
    close("vs fp64", got, ref, 1e-6, 1e-5)


@pytest.mark.parametrize("B,vmax", [(3, 2), (40, 256), (300, 256)])
def test_conv2_backward_data_fused_with_the_first_layers_weight_gradient_from_frames(B, vmax):
    """round 6, GRUModel's conv2 over conv1 (models.py:570-590): a2c_conv2d_bwd_data_w1_frames keeps layer 2's masked input
    gradient in LDS and takes the first layer's weight gradient from it on the bf16 pipe (three exact bf16 pieces x exact uint8
    pixels, fp32 accumulation) -- against the two unfused launches (a2c_conv2d_bwd_data_signs, then
    a2c_conv2d_bwd_weight_frames) at re-association tolerance, and no further from the fp64 gradient than they are.
    B = 3 / 40 / 300: fewer bands than CUs, about one, several per workgroup; fresh episodes (nvalid < 4) mixed in."""
    ops = _ops()
    d2, d1 = ops.conv_desc(16, 84, 84, 24, 3, 2, 1), ops.conv_desc(4, 84, 84, 16, 3, 1, 1)
    nb = ops.conv_bwd_data_w1_frames_ws_bytes(d2, d1, B)
    assert nb > 0 and ops.conv_bwd_data_w1_frames_ws_bytes(ops.conv_desc(24, 42, 42, 32, 3, 2, 1), d1, B) == 0
    T, HW = 8, 84 * 84
    R = (B + T - 1) // T
    rng = np.random.default_rng(4400 + B)
    Fs = rng.integers(0, vmax, size=(R, T + 4, HW), dtype=np.uint8)
    nv = np.where(rng.random(R * T) < 0.8, 4, rng.integers(1, 5, size=(R * T,))).astype(np.int32)
    Fd, nvd = torch.from_numpy(Fs).to(DEV), torch.from_numpy(nv).to(DEV)
    w2 = rnd((24, 16, 3, 3), 31) / 12.0
    dout = (rnd((B, 24, 42, 42), 32) * torch.from_numpy(rng.lognormal(0, 1, size=(B, 1, 1, 1)).astype(np.float32))).to(DEV)
    a1pos = torch.from_numpy(rng.random((B, 16, 84, 84)) < 0.6)
    words = _sign_words(a1pos).to(DEV)
    wb = torch.empty(ops.conv_prep_floats(d2, 1), device=DEV)
    ops.conv_prep(d2, 1, w2.to(DEV), wb)
    # unfused
    din = torch.full((B, 16, 84, 84), float("nan"), device=DEV)
    ops.conv_bwd_data_signs(d2, dout, wb, words, din, B)
    dW_u, db_u = torch.full((16, 4, 3, 3), float("nan"), device=DEV), torch.full((16,), float("nan"), device=DEV)
    ws_u = torch.empty((ops.conv_bwd_weight_ws_bytes(d1, B) + 3) // 4, device=DEV)
    ops.conv_bwd_weight_frames(d1, Fd, Fd.stride(0), T, nvd, din, dW_u, db_u, B, ws_u)
    # fused
    dW_f, db_f = torch.full((16, 4, 3, 3), float("nan"), device=DEV), torch.full((16,), float("nan"), device=DEV)
    ws_f = torch.empty((nb + 3) // 4, device=DEV)
    ops.conv_bwd_data_w1_frames(d2, dout, wb, words, d1, Fd, Fd.stride(0), T, nvd, dW_f, db_f, B, ws_f)
    dW_g, db_g = torch.empty_like(dW_f), torch.empty_like(db_f)
    ops.conv_bwd_data_w1_frames(d2, dout, wb, words, d1, Fd, Fd.stride(0), T, nvd, dW_g, db_g, B, ws_f)
    torch.cuda.synchronize()
    assert torch.equal(dW_f, dW_g) and torch.equal(db_f, db_g)              # deterministic
    # fp64: da1 = conv_transpose(dout, w2) * (a1 > 0); dW1 = conv1's weight gradient over the stacked frames
    da1 = F.conv_transpose2d(dout.cpu().double(), w2.double(), stride=2, padding=1, output_padding=1) * a1pos
    xs = torch.zeros(B, 4, 84, 84, dtype=torch.float64)
    for n in range(B):
        r, t = divmod(n, T)
        for c in range(4):
            if c >= 4 - nv[n]:
                xs[n, c] = torch.from_numpy(Fs[r, t + c].astype(np.float64)).reshape(84, 84)
    wt = torch.zeros(16, 4, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xs, wt, padding=1).backward(da1)
    want, wantb = wt.grad, da1.sum((0, 2, 3))
    rms = float(want.pow(2).mean().sqrt())
    e_f = float((dW_f.cpu().double() - want).pow(2).mean().sqrt()) / rms
    e_u = float((dW_u.cpu().double() - want).pow(2).mean().sqrt()) / rms
    print(f"B={B} vmax={vmax}: rms(fused - fp64)/rms = {e_f:.2e}, rms(unfused - fp64)/rms = {e_u:.2e}")
    assert e_f <= 2 * e_u + 2e-7, (e_f, e_u)
    close("dW1 fused vs unfused", dW_f, dW_u, 2e-6 * float(want.abs().max()), 1e-5)
    close("db1", db_f, wantb, 2e-6 * float(wantb.abs().max()) + 1e-6, 2e-6)
    close("db1 unfused", db_u, wantb, 2e-6 * float(wantb.abs().max()) + 1e-6, 2e-6)


# ------------------------------------------------------------------ GRU gates / LayerNorm
def test_gru_kernels_vs_autograd():
    ops = _ops()
    B, h = 37, 64
    gx, gh, rhu = rnd((B, 3 * h), 100, -2, 2), rnd((B, 2 * h), 101, -2, 2), rnd((B, h), 102, -2, 2)
    b, hin = rnd((3, 1, h), 103) * 0.1, rnd((B, h), 104)
    t = [v.clone().requires_grad_(True) for v in (gx, gh, rhu, hin)]
    z = torch.sigmoid(t[0][:, :h] + t[1][:, :h] + b[0])
    r = torch.sigmoid(t[0][:, h:2 * h] + t[1][:, h:] + b[1])
    rh = r * t[3]
    c = torch.tanh(t[0][:, 2 * h:] + t[2] + b[2])
    hn = z * t[3] + (1 - z) * c
    dev = lambda v: v.detach().to(DEV).contiguous()
    zd, rd, rhd, cd, hnd = (torch.empty(B, h, device=DEV) for _ in range(5))
    ops.gru_gates(dev(gx), dev(gh), dev(b), dev(hin), zd, rd, rhd)
    ops.gru_out(dev(gx), dev(rhu), dev(b), dev(hin), zd, cd, hnd)
    close("z", zd, z); close("r", rd, r); close("rh", rhd, rh); close("c", cd, c); close("hn", hnd, hn)
    # backward: d(hn) given, with d(rhu) == dc_pre flowing back through rh (rh_u = rh Wh2 is outside: use identity)
    g = rnd((B, h), 105)
    dcp, dz, dh = (torch.empty(B, h, device=DEV) for _ in range(3))
    ops.gru_out_bwd(dev(g), dev(hin), zd, cd, dcp, dz, dh)
    close("dc_pre", dcp, g * (1 - z) * (1 - c * c))
    d_rh = rnd((B, h), 106)
    dzp, drp = torch.empty(B, h, device=DEV), torch.empty(B, h, device=DEV)
    ops.gru_gates_bwd(dev(d_rh), dz, dev(hin), zd, rd, dzp, drp, dh)
    close("dz_pre", dzp, g * (hin - c) * z * (1 - z))
    close("dr_pre", drp, d_rh * hin * r * (1 - r))
    close("dh", dh, g * z + d_rh * r)
    # the BPTT form: dh_new + carry * (1 - done) as the incoming gradient == mask_rows + add + gru_out_bwd, bit for bit; in place
    # (carry IS the dh buffer, as the unroll uses it)
    carry = rnd((B, h), 107)
    T = 5
    dones = (rnd((B, T), 108, 0, 1) < 0.4).float()
    t_ = 3
    gm = dev(g) + dev(carry) * (1 - dones.to(DEV)[:, t_:t_ + 1])
    ref = [torch.empty(B, h, device=DEV) for _ in range(3)]
    ops.gru_out_bwd(gm.contiguous(), dev(hin), zd, cd, *ref)
    dcp2, dz2, dh2 = torch.empty(B, h, device=DEV), torch.empty(B, h, device=DEV), dev(carry).clone()
    dd = dones.to(DEV)
    ops.gru_out_bwd_carry(dev(g), dh2, dd.data_ptr() + 4 * t_, T, dev(hin), zd, cd, dcp2, dz2, dh2)
    assert torch.equal(dcp2, ref[0]) and torch.equal(dz2, ref[1]) and torch.equal(dh2, ref[2])
    assert float(dones.sum()) > 0 and float((1 - dones).sum()) > 0


def test_layernorm_vs_torch():
    ops = _ops()
    rows, n = 53, 200
    x, w, b = rnd((rows, n), 110, -2, 2), 1 + 0.1 * rnd((n,), 111), 0.1 * rnd((n,), 112)
    xr, wr, br = (v.clone().requires_grad_(True) for v in (x, w, b))
    y = F.layer_norm(xr, (n,), wr, br)
    g = rnd((rows, n), 113)
    y.backward(g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    yd, mean, rstd = torch.empty(rows, n, device=DEV), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_fwd(xd, wd, bd, yd, mean, rstd)
    close("ln fwd", yd, y, 2e-6, 1e-5)
    dx, dwr = torch.empty(rows, n, device=DEV), torch.empty(rows, n, device=DEV)
    ops.layernorm_bwd(g.to(DEV), xd, wd, mean, rstd, dx, dwr)
    close("ln dx", dx, xr.grad, 2e-6, 1e-5)
    close("ln dw", dwr.sum(0), wr.grad, 1e-5, 1e-5)


# ------------------------------------------------------------------ clip + optimiser
@pytest.mark.parametrize("kind", ["RMSprop", "Adam"])
def test_clip_optimizer_vs_torch(kind):
    ops = _ops()
    n = 10007
    p0 = rnd((n,), 120)
    ref_p = p0.clone().requires_grad_(True)
    opt = getattr(torch.optim, kind)([ref_p], lr=1e-4)
    pd = torch.zeros(n + 1, device=DEV)[:n]           # keep 16 B alignment of the base
    pd.copy_(p0)
    gd = torch.zeros(n, device=DEV)
    s1, s2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sumsq = torch.zeros(1, dtype=torch.float64, device=DEV)
    norm = torch.zeros(1, device=DEV)
    for step in range(1, 4):
        g = rnd((n,), 121 + step) * (0.02 if step == 2 else 0.001)     # step 2 clips, others do not
        ref_p.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
        opt.step()
        gd.copy_(g)
        ops.gradnorm_sq(gd, sumsq)
        if kind == "RMSprop":
            ops.clip_rmsprop(pd, gd, s1, sumsq, 0.5, 1e-4, 0.99, 1e-8, norm)
        else:
            ops.clip_adam(pd, gd, s1, s2, sumsq, 0.5, 1e-4, 0.9, 0.999, 1e-8, step, norm)
        assert norm.item() == pytest.approx(float(tn), rel=2e-6)
        close(f"clipped grad step {step}", gd, ref_p.grad, 1e-10, 2e-6)
        close(f"params step {step}", pd, ref_p.detach(), 2e-7, 1e-6)


def test_torch_ops_shim_matches_oracle():
    """torch.ops.a2c_mi355x.* (custom-op registration over the C ABI) on the HIP dispatch key"""
    from a2c_amd import ops as aops
    o = aops.load_torch_ops()
    n_seg, T, A = 7, 12, 4
    N = n_seg * T
    x, r = rnd((N,), 1), rnd((N,), 2)
    d = (rnd((N,), 3, 0, 1) < 0.2).float()
    d[T - 1::T] = 1
    y = o.discount(x.to(DEV), d.to(DEV), 0.99, n_seg)
    assert torch.equal(y.cpu(), O.discount(x, d, 0.99))
    adv, ret = o.gae_returns(x.to(DEV), r.to(DEV), d.to(DEV), 0.9702, 0.99, n_seg)
    assert torch.equal(adv.cpu(), O.discount(x, d, 0.9702)) and torch.equal(ret.cpu(), O.discount(r, d, 0.99))
    logits, u = rnd((33, A), 4, -3, 3), rnd((33,), 5, 0, 0.999)
    acts = o.softmax_sample(logits.to(DEV), u.to(DEV))
    assert torch.equal(acts.cpu(), O.sample_action(F.softmax(logits, -1), u).long())
    prev = rnd((3, 4, 64), 6)
    f8 = (rnd((3, 64), 7, 0, 1) * 255).to(torch.uint8)
    rst = torch.tensor([0., 1., 0.])
    out = o.frame_stack_push(f8.to(DEV), rst.to(DEV), prev.to(DEV)).cpu()
    want = torch.cat([prev[:, 1:], f8.float()[:, None]], 1)
    want[1, :3] = 0
    assert torch.equal(out, want)
    assert torch.equal(o.frame_stack_push(f8.float().to(DEV), rst.to(DEV), prev.to(DEV)).cpu(), want)
    w, b, xin = rnd((5, 16), 8), rnd((5,), 9), rnd((9, 16), 10)
    close("linear", o.linear(xin.to(DEV), w.to(DEV), b.to(DEV), True), F.relu(F.linear(xin, w, b)), 1e-6, 1e-5)


def test_torch_ops_cover_the_model_and_optimiser_families():
    """the torch.ops.a2c_mi355x side door for the remaining section-8b families (conv2d fwd / bwd_data / bwd_weight, gemm nn / tn,
    GRU cell fwd / bwd, LayerNorm fwd / bwd, rollout_record_, clip_adam_): tensors in, tensors out, against torch on the CPU"""
    from a2c_amd import ops as aops
    o = aops.load_torch_ops()
    # conv (models.py:98,119,312): GRUModel's 16 -> 24 stride-2 layer at a small batch
    x, w, b = rnd((3, 16, 20, 24), 1, 0, 1), rnd((24, 16, 3, 3), 2) / 12.0, rnd((24,), 3) * 0.1
    y = o.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 2, 1, True)
    ref = F.relu(F.conv2d(x, w, b, stride=2, padding=1))
    close("conv2d_fwd", y, ref, 2e-6, 1e-5)
    dout = rnd(tuple(ref.shape), 4)
    xr, wr = x.clone().double().requires_grad_(True), w.clone().double().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=2, padding=1).backward(dout.double())
    mask = rnd(tuple(x.shape), 5)
    din = o.conv2d_bwd_data(dout.to(DEV), w.to(DEV), mask.to(DEV), 20, 24, 2, 1)
    close("conv2d_bwd_data", din, xr.grad * (mask > 0), 2e-6, 1e-5)
    close("conv2d_bwd_data no mask", o.conv2d_bwd_data(dout.to(DEV), w.to(DEV), None, 20, 24, 2, 1), xr.grad, 2e-6, 1e-5)
    dW, db = o.conv2d_bwd_weight(x.to(DEV), dout.to(DEV), 3, 2, 1)
    close("conv2d_bwd_weight", dW, wr.grad, 1e-5 * float(wr.grad.abs().max()), 1e-5)
    close("conv2d_bwd_weight bias", db, dout.double().sum((0, 2, 3)), 1e-5, 1e-5)
    with pytest.raises((RuntimeError, NotImplementedError), match="CPU"):      # only the HIP dispatch key is registered
        o.conv2d_fwd(x, w, b, 2, 1, True)
    # dense
    a, bm = rnd((37, 52), 6), rnd((52, 29), 7)
    close("gemm_nn", o.gemm_nn(a.to(DEV), bm.to(DEV)), a.double() @ bm.double(), 2e-6, 1e-5)
    at_, bt = rnd((300, 24), 8), rnd((300, 40), 9)
    close("gemm_tn", o.gemm_tn(at_.to(DEV), bt.to(DEV)), at_.double().t() @ bt.double(), 2e-5, 1e-5)
    # GRU cell (models.py:465-476)
    B, nin, hd = 5, 48, 32
    xg, hg = rnd((B, nin), 10), rnd((B, hd), 11)
    Wx, Wh, bb = rnd((3, nin, hd), 12) / 7, rnd((3, hd, hd), 13) / 6, rnd((3, 1, hd), 14) * 0.1
    hr = hg.clone().double().requires_grad_(True)
    Wxd, Whd, bd = Wx.double(), Wh.double(), bb.double()
    z = torch.sigmoid(xg.double() @ Wxd[0] + hr @ Whd[0] + bd[0])
    r = torch.sigmoid(xg.double() @ Wxd[1] + hr @ Whd[1] + bd[1])
    c = torch.tanh(xg.double() @ Wxd[2] + (r * hr) @ Whd[2] + bd[2])
    hn = z * hr + (1 - z) * c
    outs = o.gru_cell_fwd(xg.to(DEV), hg.to(DEV), Wx.to(DEV), Wh.to(DEV), bb.to(DEV))
    for name, got, want in zip(("h_new", "z", "r", "c"), outs, (hn, z, r, c)):
        close("gru " + name, got, want.detach(), 2e-6, 1e-5)
    dhn = rnd((B, hd), 15)
    hn.backward(dhn.double())
    dzp, drp, dcp, dh = o.gru_cell_bwd(dhn.to(DEV), hg.to(DEV), outs[1], outs[2], outs[3], Wh.to(DEV))
    dh_full = dh.cpu().double() + dzp.cpu().double() @ Whd[0].t() + drp.cpu().double() @ Whd[1].t()     # through the z / r gates
    close("gru dh", dh_full, hr.grad, 5e-6, 1e-5)
    # LayerNorm (models.py:392)
    xl, wl, bl = rnd((9, 200), 16), rnd((200,), 17, 0.5, 1.5), rnd((200,), 18)
    xlr, wlr = xl.clone().double().requires_grad_(True), wl.clone().double().requires_grad_(True)
    yl = F.layer_norm(xlr, (200,), wlr, bl.double())
    dyl = rnd((9, 200), 19)
    yl.backward(dyl.double())
    yg, mean, rstd = o.layernorm_fwd(xl.to(DEV), wl.to(DEV), bl.to(DEV))
    close("layernorm_fwd", yg, yl.detach(), 2e-6, 1e-5)
    dxl, dwr = o.layernorm_bwd(dyl.to(DEV), xl.to(DEV), wl.to(DEV), mean, rstd)
    close("layernorm dx", dxl, xlr.grad, 5e-6, 1e-5)
    close("layernorm dw", dwr.sum(0), wlr.grad, 1e-5, 1e-5)
    # bookkeeping of one env step (runner.py:212-232), two steps so that a delta is written
    T, Bn = 4, 3
    rewards, dones, deltas = (torch.zeros(Bn * T, device=DEV) for _ in range(3))
    vprev = torch.zeros(Bn, device=DEV)
    rew0, done0, v0 = torch.tensor([0., 1., 0.], device=DEV), torch.tensor([0., 0., 1.], device=DEV), torch.tensor([.5, .25, -1.], device=DEV)
    o.rollout_record_(rew0, done0, v0, vprev, rewards, dones, deltas, T, 0, 0, 0.99, True)
    v1 = torch.tensor([.1, .2, .3], device=DEV)
    o.rollout_record_(torch.zeros(Bn, device=DEV), torch.zeros(Bn, device=DEV), v1, vprev, rewards, dones, deltas, T, 1, 0, 0.99, True)
    assert rewards.view(Bn, T)[:, 0].tolist() == [0., 1., 0.] and dones.view(Bn, T)[:, 0].tolist() == [0., 1., 1.]   # Pong: rew != 0 -> done
    want = [0. + 0.99 * .1 * 1 - .5, 1. + 0. - .25, 0. + 0. + 1.]
    close("deltas", deltas.view(Bn, T)[:, 0], torch.tensor(want), 1e-6, 1e-6)
    # clip + Adam (updater.py:129-132, 226-229)
    p0, g0 = rnd((1000,), 20), rnd((1000,), 21)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=1e-3)
    pd, gd, m1, m2 = p0.to(DEV), g0.to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in (1, 2):
        pr.grad = g0.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 0.5)
        opt.step()
        gd.copy_(g0)
        norm = o.clip_adam_(pd, gd, m1, m2, step, 0.5, 1e-3)
        assert float(norm) == pytest.approx(float(tn), rel=2e-6)
        close(f"adam params step {step}", pd, pr.detach(), 2e-7, 1e-6)


@pytest.mark.parametrize("B,signs,strided", [(256, 2, True), (300, 0, False), (65, 1, False), (130, 3, True)])
def test_conv_chain_equals_the_per_layer_launches_bit_for_bit(B, signs, strided):
    """a2c_conv2d_fwd_chain (one launch: a workgroup walks one sample through GRUModel's conv2 .. conv5, models.py:570-636)
    against four a2c_conv2d_fwd[_signs] launches: same tiles, same summation order -> every activation and every sign word
    bit for bit; the per-layer results against torch.  strided: outputs are rows of wider buffers (the rollout writes rows
    slot * T + t of the update's activation stash); B = 300: more samples than CUs."""
    ops = _ops()
    specs = [(16, 84, 84, 24, 3, 2, 1), (24, 42, 42, 32, 3, 2, 1), (32, 21, 21, 48, 3, 2, 1), (48, 11, 11, 64, 3, 2, 1)]
    descs = [ops.conv_desc(*s) for s in specs]
    ch = ops.ConvChain(descs)
    assert ch.ok
    assert not ops.ConvChain(descs[:3]).ok and not ops.ConvChain([ops.conv_desc(16, 84, 84, 24, 3, 1, 1)] + descs[1:]).ok
    x = torch.relu(rnd((B, 16, 84, 84), 880)).to(DEV)
    ws, bs, wfs = [], [], []
    for i, (d, s) in enumerate(zip(descs, specs)):
        w = (rnd((s[3], s[0], 3, 3), 881 + i) / (s[0] * 9) ** 0.5 * 1.6).to(DEV)
        b = (rnd((s[3],), 891 + i) * 0.1).to(DEV)
        wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
        ops.conv_prep(d, 0, w, wf)
        ws.append(w); bs.append(b); wfs.append(wf)
    mult = 3 if strided else 1          # row stride of the output buffers, in samples
    nsw = [ops.conv_sign_words(d) for d in descs]        # signs = number of leading layers that leave sign words (0 .. 3)

    def run(chain):
        outs = [torch.full((B * mult, d.Cout, d.OH, d.OW), float("nan"), device=DEV) for d in descs]
        sg = [torch.full((B * mult, nsw[i]), -7, dtype=torch.int32, device=DEV) for i in range(signs)]
        n = [d.Cout * d.OH * d.OW for d in descs]
        optr = [o.data_ptr() + 4 * nn_ * (1 if strided else 0) for o, nn_ in zip(outs, n)]      # row 1, 1 + mult, ...
        sgp = {i: (sg[i].data_ptr() + 4 * nsw[i] * (1 if strided else 0), mult * nsw[i]) for i in range(signs)}
        if chain:
            ch.fwd(x.data_ptr(), 16 * 84 * 84, wfs, bs, optr, [mult * v for v in n], B, signs=sgp or None)
        else:
            src, sbs = x.data_ptr(), 16 * 84 * 84
            for i, d in enumerate(descs):
                if i in sgp:
                    ops.conv_fwd_signs(d, src, sbs, wfs[i], bs[i], True, optr[i], sgp[i][0], sgp[i][1], B, out_bstride=mult * n[i])
                else:
                    ops.conv_fwd(d, src, sbs, wfs[i], bs[i], True, optr[i], B, out_bstride=mult * n[i])
                src, sbs = optr[i], mult * n[i]
        torch.cuda.synchronize()
        return outs, sg
    o_ref, sg_ref = run(False)
    o_ch, sg_ch = run(True)
    for i in range(4):
        a, b = o_ch[i].view(torch.int32), o_ref[i].view(torch.int32)           # (NaN rows between the strided rows compare as bits)
        assert torch.equal(a, b), i
    for i in range(signs):
        assert torch.equal(sg_ch[i], sg_ref[i]), i
        rows = sg_ch[i][1::mult] if strided else sg_ch[i]
        assert torch.equal(rows.cpu(), _sign_words((o_ch[i][1::mult] if strided else o_ch[i]).cpu() > 0)), i
    ref = x.cpu()
    for i, s in enumerate(specs):
        ref = F.relu(F.conv2d(ref, ws[i].cpu(), bs[i].cpu(), stride=2, padding=1))
        got = o_ch[i][1::mult] if strided else o_ch[i]
        close(f"chain layer {i}", got, ref, 1e-5, 1e-5)


def test_conv_chain_argument_validation():
    ops = _ops()
    from a2c_amd import _lib
    descs = [ops.conv_desc(16, 84, 84, 24, 3, 2, 1), ops.conv_desc(24, 42, 42, 32, 3, 2, 1), ops.conv_desc(32, 21, 21, 48, 3, 2, 1),
             ops.conv_desc(48, 11, 11, 64, 3, 2, 1)]
    ch = ops.ConvChain(descs)
    wfs = [torch.zeros(ops.conv_prep_floats(d, 0), device=DEV) for d in descs]
    bs = [torch.zeros(d.Cout, device=DEV) for d in descs]
    outs = [torch.zeros(2, d.Cout, d.OH, d.OW, device=DEV) for d in descs]
    x = torch.zeros(2, 16, 84, 84, device=DEV)
    n = [d.Cout * d.OH * d.OW for d in descs]
    ch.fwd(x.data_ptr(), 16 * 84 * 84, wfs, bs, [o.data_ptr() for o in outs], n, 2)          # fine
    with pytest.raises(_lib.A2CKernelError):       # output stride smaller than a sample
        ch.fwd(x.data_ptr(), 16 * 84 * 84, wfs, bs, [o.data_ptr() for o in outs], [n[0], n[1] - 4, n[2], n[3]], 2)
    with pytest.raises(_lib.A2CKernelError):       # misaligned source
        ch.fwd(x.data_ptr() + 4, 16 * 84 * 84, wfs, bs, [o.data_ptr() for o in outs], n, 2)
    with pytest.raises(_lib.A2CKernelError):       # sign rows narrower than the layer's sign words
        ch.fwd(x.data_ptr(), 16 * 84 * 84, wfs, bs, [o.data_ptr() for o in outs], n, 2, signs={0: (outs[0].data_ptr(), 8)})


@pytest.mark.parametrize("B,xs,hd,inplace", [(256, 256, 256, True), (37, 256, 256, False), (300, 64, 96, False)])
@pytest.mark.parametrize("k4", [True, False])
def test_gru_cell_fwd_two_launches_equal_the_five_bit_for_bit(B, xs, hd, inplace, k4, monkeypatch):
    """a2c_gru_cell_fwd (models.py:465-476 at rollout batch: two launches) against the path it replaces -- a2c_gemm_f32
    (x [Wx0|Wx1|Wx2], h [Wh0|Wh1]) + a2c_gru_gates + a2c_gemm_f32 ((r*h) Wh2) + a2c_gru_out -- bit for bit with the
    four-way K split (A2C_GRU_K4=1: same K split, MFMA order and order of additions), within fp32 rounding of it with the
    default eight-way split (the same sums in another order), and against the reference's formula in torch; x as rows of a
    wider buffer; h_new written over h (the in-place rollout step)."""
    ops = _ops()
    monkeypatch.setenv("A2C_GRU_K4", "1" if k4 else "0")
    x_wide = rnd((B, xs + 8), 951, 0, 1).to(DEV)
    x = x_wide[:, :xs]
    h0 = rnd((B, hd), 952).to(DEV)
    Wx, Wh, b = (rnd((3, xs, hd), 953) / xs ** 0.5).to(DEV), (rnd((3, hd, hd), 954) / hd ** 0.5).to(DEV), (rnd((3, hd), 955) * 0.1).to(DEV)
    WxC = torch.cat([Wx[g] for g in range(3)], 1).contiguous()
    WhC = torch.cat([Wh[g] for g in range(2)], 1).contiguous()
    mk = lambda *s: torch.full(s, float("nan"), device=DEV)
    # the five launches
    gx, gh, z1, r1, rh1, rhu, c1, hn1 = mk(B, 3 * hd), mk(B, 2 * hd), mk(B, hd), mk(B, hd), mk(B, hd), mk(B, hd), mk(B, hd), mk(B, hd)
    ops.gemm(0, 0, B, 3 * hd, xs, x.data_ptr(), x.stride(0), WxC.data_ptr(), 3 * hd, gx.data_ptr(), 3 * hd)
    ops.gemm(0, 0, B, 2 * hd, hd, h0.data_ptr(), hd, WhC.data_ptr(), 2 * hd, gh.data_ptr(), 2 * hd)
    ops.gru_gates(gx, gh, b, h0, z1, r1, rh1)
    ops.gemm(0, 0, B, hd, hd, rh1.data_ptr(), hd, Wh[2].data_ptr(), hd, rhu.data_ptr(), hd)
    ops.gru_out(gx, rhu, b, h0, z1, c1, hn1)
    # the two
    gx2, z2, r2, rh2, c2 = mk(B, 3 * hd), mk(B, hd), mk(B, hd), mk(B, hd), mk(B, hd)
    h_io = h0.clone()
    hn2 = h_io if inplace else mk(B, hd)
    ops.gru_cell_fwd(x, h_io, WxC, WhC, Wh[2], b, gx2, z2, r2, rh2, c2, hn2)
    torch.cuda.synchronize()
    for name, a, bb in (("z", z2, z1), ("r", r2, r1), ("rh", rh2, rh1), ("c", c2, c1), ("hn", hn2, hn1)):
        if k4:
            assert torch.equal(a, bb), name
        else:
            close(name, a, bb, 4e-6, 0)
    if k4:
        assert torch.equal(gx2[:, 2 * hd:], gx[:, 2 * hd:])
    else:
        close("gx", gx2[:, 2 * hd:], gx[:, 2 * hd:], 4e-6, 0)
    xc, hc = x.cpu(), h0.cpu()
    zt = torch.sigmoid(xc.mm(Wx[0].cpu()) + hc.mm(Wh[0].cpu()) + b[0].cpu())
    rt = torch.sigmoid(xc.mm(Wx[1].cpu()) + hc.mm(Wh[1].cpu()) + b[1].cpu())
    ct = torch.tanh(xc.mm(Wx[2].cpu()) + (rt * hc).mm(Wh[2].cpu()) + b[2].cpu())
    close("h_new", hn2, zt * hc + (1 - zt) * ct, 1e-5, 1e-5)
    if not inplace:
        assert torch.equal(h_io, h0)
    ops.gru_cell_fwd(x, h0, WxC, WhC, Wh[2], b, gx2, z2, r2, rh2, None, mk(B, hd))      # c is optional
    from a2c_amd import _lib
    with pytest.raises(_lib.A2CKernelError):
        ops.gru_cell_fwd(x, h0[:, :hd - 8].contiguous(), WxC, WhC, Wh[2], b, gx2, z2, r2, rh2, c2, mk(B, hd))     # hdim % 32


@pytest.mark.parametrize("M,N,K,sk,kc", [(256, 2000, 28224, 32, True), (200, 300, 1000, 4, True), (520, 700, 330, 1, False)])
def test_gemm_x6_from_prebuilt_images(M, N, K, sk, kc):
    """a2c_gemm_x6_split + a2c_gemm_x6_images (ConvModel's resize_emb at rollout batch: the weight image built once per update):
    C = relu(A B^T + bias) from two three-piece bf16 panel images, six exact products per element pair, fp32 sums -- against
    fp64 no worse than 1.5 x the fp32 MFMA kernel's error; both source orientations, ragged rows / K, K splits; bit-identical
    run to run."""
    ops = _ops()
    a = (rnd((M, K), 41) * (rnd((M, K), 42) > 0.2)).to(DEV)          # (ReLU-like zeros)
    b = (rnd((N, K), 43) * 0.05).to(DEV)
    bias = rnd((N,), 44).to(DEV)
    ia = torch.empty(ops.gemm_x6_image_bytes(M, K) // 2, dtype=torch.int16, device=DEV)
    ib = torch.empty(ops.gemm_x6_image_bytes(N, K) // 2, dtype=torch.int16, device=DEV)
    if kc:
        ops.gemm_x6_split(a.data_ptr(), K, M, K, True, ia)
        ops.gemm_x6_split(b.data_ptr(), K, N, K, True, ib)
    else:
        at, bt = a.t().contiguous(), b.t().contiguous()
        ops.gemm_x6_split(at.data_ptr(), M, M, K, False, ia)
        ops.gemm_x6_split(bt.data_ptr(), N, N, K, False, ib)
    ws = torch.empty(max(1, sk * M * N), device=DEV)
    outs = []
    for _ in range(2):
        c = torch.full((M, N + 8), float("nan"), device=DEV)
        ops.gemm_x6_images(M, N, K, ia, ib, c.data_ptr(), N + 8, bias=bias, relu=True, splitk=sk, ws=ws if sk > 1 else None)
        assert torch.isnan(c[:, N:]).all()
        outs.append(c[:, :N].clone())
    assert torch.equal(outs[0], outs[1])
    want = torch.relu(a.double().cpu() @ b.double().cpu().t() + bias.double().cpu())
    c32 = torch.empty(M, N, device=DEV)
    import os
    old = os.environ.get("A2C_GEMM_X9")
    os.environ["A2C_GEMM_X9"] = "0"
    try:
        ops.gemm(0, 1, M, N, K, a.data_ptr(), K, b.data_ptr(), K, c32.data_ptr(), N, bias=bias, relu=True)
    finally:
        if old is None:
            del os.environ["A2C_GEMM_X9"]
        else:
            os.environ["A2C_GEMM_X9"] = old
    rms = float(want.pow(2).mean().sqrt())
    e6 = float((outs[0].double().cpu() - want).pow(2).mean().sqrt()) / rms
    e32 = float((c32.double().cpu() - want).pow(2).mean().sqrt()) / rms
    assert e6 <= 1.5 * e32 + 1e-7, (e6, e32)
    from a2c_amd import _lib
    with pytest.raises(_lib.A2CKernelError):             # K splits without their slabs
        ops.gemm_x6_images(M, N, K, ia, ib, c32.data_ptr(), N, splitk=2, ws=None)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("tA,tB,M,N,K", [(0, 1, 1280, 1100, 2000), (0, 0, 1152, 900, 1000), (1, 0, 520, 1030, 777)])
def test_gemm_bf16_x9_path_is_the_fp32_product(tA, tB, M, N, K, mode, monkeypatch):
    """A2C_GEMM_X9=1 (opt-in): both operands split into three bf16 pieces (exact), nine exact piece products per element
    pair, fp32 accumulation -- against fp64 no less accurate than the fp32 MFMA kernels (1.5 x + 1e-7 of the rms), for every
    operand orientation, ragged sizes, bias + ReLU + mask in the epilogue; without the extra workspace the call falls back."""
    ops = _ops()
    a = rnd((K, M) if tA else (M, K), 31).to(DEV)
    b = rnd((N, K) if tB else (K, N), 32).to(DEV)
    bias, mask = rnd((N,), 33).to(DEV), rnd((M, N), 34).to(DEV)
    A64 = (a.t() if tA else a).double().cpu()
    B64 = (b.t() if tB else b).double().cpu()
    want = torch.relu(A64 @ B64 + bias.double().cpu()) * (mask.cpu() > 0)
    res = {}
    for x9 in (mode, "0"):
        monkeypatch.setenv("A2C_GEMM_X9", x9)
        nb = ops.gemm_ws_bytes(M, N, 1, K)
        assert (nb > 0) == (x9 == mode)
        ws = torch.empty(max(1, (nb + 3) // 4), device=DEV)
        c = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(tA, tB, M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), N, bias=bias, relu=True,
                 mask_ptr=mask.data_ptr(), ldmask=N, ws=ws)
        res[x9] = c.cpu().double()
    rms = float(want.pow(2).mean().sqrt())
    e9, e32 = (float((res[k] - want).pow(2).mean().sqrt()) / rms for k in (mode, "0"))
    assert e9 <= 1.5 * e32 + 1e-7, (e9, e32)
    assert not torch.equal(res[mode], res["0"])          # (the x 9 kernel did run)
    monkeypatch.setenv("A2C_GEMM_X9", mode)              # no room for the images: the fp32 kernel, bit for bit
    c = torch.empty(M, N, device=DEV)
    ops.gemm(tA, tB, M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), N, bias=bias, relu=True,
             mask_ptr=mask.data_ptr(), ldmask=N)
    assert torch.equal(c.cpu().double(), res["0"])


@pytest.mark.parametrize("B,hd,with_carry", [(256, 256, True), (37, 256, False), (300, 96, True)])
@pytest.mark.parametrize("k4", [True, False])
def test_gru_cell_bwd_two_launches_equal_the_five_bit_for_bit(B, hd, with_carry, k4, monkeypatch):
    """a2c_gru_cell_bwd (one step of the BPTT unroll's backward, updater.py:139-169) against the five launches it replaces --
    a2c_gru_out_bwd[_carry], a2c_gemm_f32 (dc_pre Wh2^T), a2c_gru_gates_bwd, two accumulating a2c_gemm_f32 -- bit for bit
    with the four-way K split (A2C_GRU_K4=1), within fp32 rounding with the default eight-way split, and against autograd
    through the reference's cell formula (models.py:465-476)."""
    ops = _ops()
    monkeypatch.setenv("A2C_GRU_K4", "1" if k4 else "0")
    mk = lambda *s: torch.full(s, float("nan"), device=DEV)
    dhn, carry, h0 = rnd((B, hd), 961).to(DEV), rnd((B, hd), 962).to(DEV), rnd((B, hd), 963).to(DEV)
    dones = (rnd((B, 3), 964, 0, 1) < 0.3).float().to(DEV)          # done of row b at dones[b * 3 + 1]
    x = rnd((B, hd), 965).to(DEV)
    Wx, Wh, b = (rnd((3, hd, hd), 966) / hd ** 0.5).to(DEV), (rnd((3, hd, hd), 967) / hd ** 0.5).to(DEV), (rnd((3, hd), 968) * 0.1).to(DEV)
    z = torch.sigmoid(x @ Wx[0] + h0 @ Wh[0] + b[0])
    r = torch.sigmoid(x @ Wx[1] + h0 @ Wh[1] + b[1])
    c = torch.tanh(x @ Wx[2] + (r * h0) @ Wh[2] + b[2])
    dptr, dstride = dones.data_ptr() + 4, 3
    # the five launches
    dcp1, dz1, drh1, dzp1, drp1, dh1 = (mk(B, hd) for _ in range(6))
    if with_carry:
        ops.gru_out_bwd_carry(dhn, carry, dptr, dstride, h0, z, c, dcp1, dz1, dh1)
    else:
        ops.gru_out_bwd(dhn, h0, z, c, dcp1, dz1, dh1)
    ops.gemm(0, 1, B, hd, hd, dcp1.data_ptr(), hd, Wh[2].data_ptr(), hd, drh1.data_ptr(), hd)
    ops.gru_gates_bwd(drh1, dz1, h0, z, r, dzp1, drp1, dh1)
    ops.gemm(0, 1, B, hd, hd, dzp1.data_ptr(), hd, Wh[0].data_ptr(), hd, dh1.data_ptr(), hd, accumulate=True)
    ops.gemm(0, 1, B, hd, hd, drp1.data_ptr(), hd, Wh[1].data_ptr(), hd, dh1.data_ptr(), hd, accumulate=True)
    # the two
    dcp2, dz2, dzp2, drp2, dh2 = (mk(B, hd) for _ in range(5))
    ops.gru_cell_bwd(dhn, carry if with_carry else None, dptr if with_carry else 0, dstride, h0, z, r, c, Wh, dcp2, dz2, dzp2, drp2, dh2)
    torch.cuda.synchronize()
    for name, a, bb in (("dc_pre", dcp2, dcp1), ("dz", dz2, dz1), ("dz_pre", dzp2, dzp1), ("dr_pre", drp2, drp1), ("dh", dh2, dh1)):
        if k4:
            assert torch.equal(a, bb), name
        else:
            close(name, a, bb, 4e-6 * float(bb.abs().max()), 0)
    # autograd: dL/dh_in for L = sum(h_new * g), g = dhn + carry * (1 - done)
    hc = h0.cpu().double().requires_grad_(True)
    Wxd, Whd, bd, xd = Wx.cpu().double(), Wh.cpu().double(), b.cpu().double(), x.cpu().double()
    zt = torch.sigmoid(xd @ Wxd[0] + hc @ Whd[0] + bd[0])
    rt = torch.sigmoid(xd @ Wxd[1] + hc @ Whd[1] + bd[1])
    ct = torch.tanh(xd @ Wxd[2] + (rt * hc) @ Whd[2] + bd[2])
    hn = zt * hc + (1 - zt) * ct
    g = dhn.cpu().double() + (carry.cpu().double() * (1 - dones[:, 1].cpu().double()).unsqueeze(1) if with_carry else 0)
    (hn * g).sum().backward()
    close("dh vs autograd", dh2, hc.grad, 2e-5 * float(hc.grad.abs().max()), 1e-5)
    from a2c_amd import _lib
    with pytest.raises(_lib.A2CKernelError):        # the carry must not alias dh
        ops.gru_cell_bwd(dhn, dh2, dptr, dstride, h0, z, r, c, Wh, dcp2, dz2, dzp2, drp2, dh2)
