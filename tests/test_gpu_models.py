"""-m gpu: model classes, Runner and Updater of a2c_amd (HIP path through the C ABI) against
(1) the outputs recorded from the reference (tests/golden/*.npz) and (2) the CPU oracle on the
same closed-form inputs.  fp32 tolerance 1e-5 (north star), stated per assertion."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import (MODEL_CASES, ROLLOUT_CASES, UPDATE_CASES, CHECKPOINT_CASES, hashf, base_hyps, synth_shared,  # noqa: E402
                   sample_idx)
from test_gpu_kernels import close  # noqa: E402

DEV = "cuda"
torch.set_num_threads(4)


def make_net(kind, ss, A, h):
    import a2c_amd
    net = getattr(a2c_amd.models, kind)(list(ss), A, h_size=h, bnorm=False)
    sd = O.formula_state_dict(kind, ss, A, h)
    assert set(net.state_dict().keys()) == set(sd.keys())
    net.load_state_dict(sd)
    shapes, _ = O.param_shapes(kind, ss, A, h)
    prim = [k for k in shapes if "running" not in k and "num_batches" not in k]
    assert [n for n, _ in net.named_parameters()] == prim      # optimiser index order == reference
    return net


@pytest.mark.parametrize("i", range(len(MODEL_CASES)), ids=[f"{c[0]}-{c[1]}" for c in MODEL_CASES])
def test_model_forward_golden_and_grads(golden, i):
    g = golden["g4_model_forward"]
    kind, ss, A, h, B = MODEL_CASES[i]
    net = make_net(kind, ss, A, h)
    assert sum(p.numel() for p in net.parameters()) == int(g[f"nparams{i}"])
    x = torch.from_numpy(O.formula_frames(B, ss, seed=400 + i, binary=(len(ss) == 3 and ss[-1] == 84)))
    hin = torch.from_numpy(hashf(B * h, 450 + i, -1, 1).reshape(B, h)) if net.is_recurrent else None
    with torch.no_grad():
        out = net(x, hin) if net.is_recurrent else net(x)
    close("val", out[0], g[f"val{i}"], 1e-5, 1e-5)
    close("pi", out[1], g[f"pi{i}"], 1e-5, 1e-5)
    if net.is_recurrent:
        close("h", out[2], g[f"h{i}"], 1e-5, 1e-5)
    # gradients through the public autograd bridge vs the oracle (torch CPU autograd)
    onet = O.OracleNet(kind, ss, A, h)
    gv = torch.from_numpy(hashf(B, 460 + i, -1, 1).reshape(B, 1))
    gp = torch.from_numpy(hashf(B * A, 470 + i, -1, 1).reshape(B, A))
    gh = torch.from_numpy(hashf(B * h, 480 + i, -1, 1).reshape(B, h))
    if net.is_recurrent:
        hin_o = hin.clone().requires_grad_(True)
        v, p, hn = onet(x, hin_o)
        ((v * gv).sum() + (p * gp).sum() + (hn * gh).sum()).backward()
    else:
        v, p = onet(x)
        ((v * gv).sum() + (p * gp).sum()).backward()
    net.req_grads(True)
    for q in net.parameters():
        q.grad = None
    if net.is_recurrent:
        hin_d = hin.to(DEV).requires_grad_(True)
        v2, p2, h2 = net(x, hin_d)
        ((v2 * gv.to(DEV)).sum() + (p2 * gp.to(DEV)).sum() + (h2 * gh.to(DEV)).sum()).backward()
        close("d h_in", hin_d.grad, hin_o.grad, 1e-6, 1e-4)
    else:
        v2, p2 = net(x)
        ((v2 * gv.to(DEV)).sum() + (p2 * gp.to(DEV)).sum()).backward()
    for (n, q), (n2, q2) in zip(net.named_parameters(), onet.named_parameters()):
        assert n == n2
        if q2.grad is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
            continue
        scale = float(q2.grad.abs().max()) + 1e-12
        close(f"grad {n}", q.grad, q2.grad, 2e-5 * scale, 2e-5)


def test_state_dict_roundtrip_and_alias_keys():
    net = make_net("A3CModel", (4, 84, 84), 3, 256)
    net._ensure_device()
    sd = net.state_dict()
    for k in ("conv1.0.weight", "convs.0.0.weight", "features.0.0.weight"):
        assert k in sd
    assert sd["conv1.0.weight"].data_ptr() == sd["features.0.0.weight"].data_ptr()
    net2 = make_net("A3CModel", (4, 84, 84), 3, 256)
    net2.load_state_dict({k: v.cpu() for k, v in sd.items()})
    x = torch.from_numpy(O.formula_frames(2, (4, 84, 84), seed=1, binary=True))
    with torch.no_grad():
        a, b = net(x), net2(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_no_cpu_fallback():
    from a2c_amd import ops
    with pytest.raises(RuntimeError):
        ops.discount_rows(torch.zeros(4), torch.zeros(4), 0.9, 1, 4)


# ------------------------------------------------------------------ Runner
def _fake_pool(ekws):
    from a2c_amd.runner import HostEnvPool
    return HostEnvPool([O.FakeEnv(**k) for k in ekws])


def _datas(N, ss, recurrent, h=256, actions_on_host=True):
    D = dict(states=torch.zeros(N, *ss, device=DEV), deltas=torch.zeros(N, device=DEV),
             rewards=torch.zeros(N, device=DEV), dones=torch.zeros(N, device=DEV),
             actions=torch.zeros(N).long() if actions_on_host else torch.zeros(N, dtype=torch.int64, device=DEV))
    if recurrent:
        D["h_states"] = torch.zeros(N, h, device=DEV)
    return D


@pytest.mark.parametrize("fused_step", [True, False], ids=["step_kernel", "layered"])
@pytest.mark.parametrize("case", ROLLOUT_CASES, ids=[c[0] for c in ROLLOUT_CASES])
def test_runner_rollout_golden(golden, case, fused_step, monkeypatch):
    """one env playing consecutive slots, against the trace recorded from the reference Runner
    (A3CModel: through the one-launch step kernel and through the per-layer path)"""
    import queue
    if not fused_step:
        if case[1] != "A3CModel":
            pytest.skip("only A3CModel has a one-launch step kernel")
        monkeypatch.setenv("A2C_NO_FUSED_STEP", "1")
    from a2c_amd.runner import Runner
    g = golden["g5_rollout"]
    name, kind, env_type, T, n_slots, ekw, A = case
    hyps = base_hyps(env_type=env_type, n_tsteps=T, n_rollouts=n_slots, action_shift=1 if "Pong" in env_type else 0,
                     n_envs=1)
    ss = (4, 84, 84)
    net = make_net(kind, ss, A, 256)
    N = T * n_slots
    D = _datas(N, ss, net.is_recurrent)
    us = torch.from_numpy(g[f"{name}_uniforms"]).to(DEV)
    cnt = [0]

    def uniform_fn(t, B, env0):
        u = us[cnt[0]:cnt[0] + 1]
        cnt[0] += 1
        return u
    rq = queue.Queue(1)
    rq.put(-1)
    r = Runner(D, hyps, None, None, rq, env_pool=_fake_pool([ekw]), uniform_fn=uniform_fn)
    for idx in range(n_slots):
        r.rollout(net, idx, hyps)
    torch.cuda.synchronize()
    assert np.array_equal(D["actions"].numpy(), g[f"{name}_actions"])
    assert np.array_equal(D["dones"].cpu().numpy(), g[f"{name}_dones"])
    close("rewards", D["rewards"], g[f"{name}_rewards"], 1e-5, 1e-5)
    close("deltas", D["deltas"], g[f"{name}_deltas"], 1e-5, 1e-5)
    fs = D["states"].reshape(N, 4, -1).double().sum(2).cpu().numpy()
    assert np.array_equal(fs, g[f"{name}_state_frame_sums"])
    if net.is_recurrent:
        close("h_states", D["h_states"], g[f"{name}_h_states"], 1e-5, 1e-5)
        close("h_bookmark", r.h, g[f"{name}_h_bookmark"], 1e-5, 1e-5)
    assert abs(rq.get() - float(g[f"{name}_avg_rew"])) < 1e-9
    assert float(r.bookmark.double().sum()) == float(g[f"{name}_bookmark_sum"])


@pytest.mark.parametrize("kind,dev_actions,fused_step", [("A3CModel", False, True), ("A3CModel", True, True),
                                                        ("A3CModel", False, False), ("GRUModel", True, False)])
def test_runner_batched_vs_oracle(kind, dev_actions, fused_step, monkeypatch):
    """B envs in lock-step (one batched forward per step) == B independent batch-1 oracle runners"""
    from a2c_amd.runner import Runner
    if not fused_step:
        monkeypatch.setenv("A2C_NO_FUSED_STEP", "1")
    else:
        assert make_net(kind, (4, 84, 84), 4, 256)._step_supported()
    B, T, A, ss = 5, 7, 4, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0" if kind == "A3CModel" else "FakeBreakout", n_tsteps=T, n_rollouts=2 * B,
                     action_shift=0, n_envs=B)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    N = 2 * B * T
    D = _datas(N, ss, net.is_recurrent, actions_on_host=not dev_actions)
    us = torch.from_numpy(hashf(2 * T * B, 900, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd_ = [0]
    r = Runner(D, hyps, None, None, None, env_pool=_fake_pool(ekws),
               uniform_fn=lambda t, Bn, env0: usd[rnd_[0], t, env0:env0 + Bn].contiguous())
    # oracle: env j plays slot j, then slot B+j
    Do = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N), dones=torch.zeros(N),
              actions=torch.zeros(N).long())
    if net.is_recurrent:
        Do["h_states"] = torch.zeros(N, 256)
    for j in range(B):
        seq = iter([float(us[k, t, j]) for k in range(2) for t in range(T)])
        sr = O.SlotRunner(O.FakeEnv(**ekws[j]), Do, hyps, uniform_fn=lambda seq=seq: next(seq))
        sr.start(onet)
        sr.rollout(onet, j)
        sr.rollout(onet, B + j)
    for rnd_[0] in range(2):
        r.rollout(net, list(range(rnd_[0] * B, (rnd_[0] + 1) * B)), hyps)
    torch.cuda.synchronize()
    assert torch.equal(D["actions"].cpu(), Do["actions"])
    assert torch.equal(D["dones"].cpu(), Do["dones"])
    assert torch.equal(D["states"].cpu(), Do["states"])
    close("rewards", D["rewards"], Do["rewards"], 1e-5, 1e-5)
    close("deltas", D["deltas"], Do["deltas"], 1e-5, 1e-5)
    if net.is_recurrent:
        close("h_states", D["h_states"], Do["h_states"], 1e-5, 1e-5)


def test_a3c_wide_action_space_falls_back_to_plain_gemms():
    """18 actions (full Atari set): no skinny-head / composed-head / step kernels; forward, sampling
    through the Runner and one update against the oracle"""
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    A, ss, B, T = 18, (4, 84, 84), 3, 4
    net = make_net("A3CModel", ss, A, 256)
    assert not net._step_supported() and not net._fused_sampling
    onet = O.OracleNet("A3CModel", ss, A, 256)
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    ekws = [dict(env_id=j, rew_period=3, done_period=5 + j) for j in range(B)]
    N = B * T
    D = _datas(N, ss, False, actions_on_host=False)
    us = torch.from_numpy(hashf(T * B, 901, 0, 1).reshape(T, B))
    usd = us.to(DEV)
    r = Runner(D, hyps, None, None, None, env_pool=_fake_pool(ekws), uniform_fn=lambda t, Bn, e0: usd[t, e0:e0 + Bn].contiguous())
    r.rollout(net, list(range(B)), hyps)
    Do = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N), dones=torch.zeros(N),
              actions=torch.zeros(N).long())
    for j in range(B):
        seq = iter([float(us[t, j]) for t in range(T)])
        sr = O.SlotRunner(O.FakeEnv(**ekws[j]), Do, hyps, uniform_fn=lambda seq=seq: next(seq))
        sr.start(onet)
        sr.rollout(onet, j)
    torch.cuda.synchronize()
    assert torch.equal(D["actions"].cpu(), Do["actions"])
    close("deltas", D["deltas"], Do["deltas"], 1e-5, 1e-5)
    info = Updater(net, hyps).update_model(D)
    oinfo = O.OracleUpdater(onet, hyps).update_model(Do)
    for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy"):
        assert info[k] == pytest.approx(float(oinfo[k]), rel=3e-5, abs=2e-6), k
    for (n, p), (_, q) in zip(net.named_parameters(), onet.named_parameters()):
        close(f"param {n}", p.detach(), q.detach(), 3e-5, 1e-5)


@pytest.mark.parametrize("kind", ["A3CModel", "GRUModel"])
def test_three_stacked_frames(kind):
    """n_frame_stack = 3 (the reference's shipped hyperparams.json): the first conv layer has 3 input
    planes and runs on the zero-padded 4-plane copy.  Rollout through the Runner + one update vs the oracle."""
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    A, ss, B, T = 4, (3, 84, 84), 3, 4
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, n_frame_stack=3)
    ekws = [dict(env_id=j, rew_period=3, done_period=5 + j) for j in range(B)]
    N = B * T
    D = _datas(N, ss, net.is_recurrent, actions_on_host=False)
    us = torch.from_numpy(hashf(T * B, 902, 0, 1).reshape(T, B))
    usd = us.to(DEV)
    r = Runner(D, hyps, None, None, None, env_pool=_fake_pool(ekws), uniform_fn=lambda t, Bn, e0: usd[t, e0:e0 + Bn].contiguous())
    r.rollout(net, list(range(B)), hyps)
    Do = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N), dones=torch.zeros(N),
              actions=torch.zeros(N).long())
    if net.is_recurrent:
        Do["h_states"] = torch.zeros(N, 256)
    for j in range(B):
        seq = iter([float(us[t, j]) for t in range(T)])
        sr = O.SlotRunner(O.FakeEnv(**ekws[j]), Do, hyps, uniform_fn=lambda seq=seq: next(seq))
        sr.start(onet)
        sr.rollout(onet, j)
    torch.cuda.synchronize()
    assert torch.equal(D["actions"].cpu(), Do["actions"])
    assert torch.equal(D["states"].cpu(), Do["states"])
    close("deltas", D["deltas"], Do["deltas"], 1e-5, 1e-5)
    info = Updater(net, hyps).update_model(D)
    oinfo = O.OracleUpdater(onet, hyps).update_model(Do)
    for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy"):
        assert info[k] == pytest.approx(float(oinfo[k]), rel=3e-5, abs=2e-6), k
    # one RMSprop step moves a weight by lr * g / (sqrt(0.01 g^2) + eps) ~ lr * 10 * sign(g) = 1e-3 whatever
    # |g| is: for a noise-level gradient fp32 does not pin the sign, so single elements may differ by up to
    # two steps; everything else must agree to 3e-5
    for (n, p), (_, q) in zip(net.named_parameters(), onet.named_parameters()):
        diff = (p.detach().cpu() - q.detach()).abs()
        assert float(diff.max()) <= 2.1e-3, (n, float(diff.max()))
        n_off = int((diff > 3e-5 + 1e-5 * q.detach().abs()).sum())
        assert n_off <= max(1, int(2e-3 * diff.numel())), (n, n_off, diff.numel())


def test_a3c_step_kernel_matches_layered_ops():
    """a2c_a3c_step == frame_stack_push + conv/conv/composed heads + softmax_sample + record/bootstrap"""
    from a2c_amd import ops
    B, T, A, ss = 9, 5, 6, (4, 84, 84)
    net = make_net("A3CModel", ss, A, 256)
    net._ensure_device()
    st = ops.stream()
    net._refresh(st)
    S = 4 * 84 * 84
    prev = (torch.from_numpy(hashf(B * S, 31, 0, 1).reshape(B, S)) < 0.3).float().to(DEV)
    frame = (torch.from_numpy(hashf(B * 7056, 32, 0, 1).reshape(B, 7056)) < 0.3).float().to(DEV)
    reset = torch.zeros(B, device=DEV)
    reset[2] = 1
    u = torch.from_numpy(hashf(B, 33, 0, 1)).to(DEV)
    rew = torch.tensor([0, 1, 0, -1, 0, 0, 2, 0, 0], dtype=torch.float32, device=DEV)
    done = torch.tensor([0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=torch.float32, device=DEV)
    N = B * T
    for t_rec, boot in ((2, 0), (T - 1, 1)):
        mk = lambda seed: torch.from_numpy(hashf(N, seed, -1, 1)).to(DEV)
        bufs = {k: [mk(40 + i), mk(40 + i)] for i, k in enumerate(("rewards", "dones", "deltas"))}
        for k in (0, 1):
            bufs["dones"][k].copy_((bufs["dones"][k] > 0.5).float())
        vp = [torch.from_numpy(hashf(B, 50, -1, 1)).to(DEV) for _ in range(2)]
        hb_prev = torch.from_numpy(hashf(B * (A + 1), 51, -1, 1).reshape(B, A + 1)).to(DEV)
        # layered reference on device
        want_state = torch.empty(B, S, device=DEV)
        ops.rollout_post(rew, done, hb_prev[:, A].data_ptr(), A + 1, vp[0], bufs["rewards"][0], bufs["dones"][0],
                         bufs["deltas"][0], T, t_rec, 0, 0.99, True, frame, reset, prev.data_ptr(), S,
                         want_state.data_ptr(), S, B, 4, 7056, st)
        acts0 = torch.zeros(B, dtype=torch.int64, device=DEV)
        out = net._fwd(want_state.data_ptr(), S, B, "roll", st, False, sampler=(u, acts0.data_ptr(), 1))
        want_heads = torch.cat([out["logits"], out["vals"][:, None]], 1).clone()
        if boot:
            ops.rollout_bootstrap(out["vals"].data_ptr(), out["vals"].stride(0), vp[0], bufs["rewards"][0],
                                  bufs["dones"][0], bufs["deltas"][0], B, T, 0, 0.99, st)
        # one launch
        hb, _, _ = net._heads("roll", B)
        hb.copy_(hb_prev)
        got_state = torch.empty(B, S, device=DEV)
        acts1 = torch.zeros(B, dtype=torch.int64, device=DEV)
        net._step(B, st, prev=prev.data_ptr(), prev_stride=S, frame_new=frame.data_ptr(), reset_mask=reset.data_ptr(),
                  out=got_state.data_ptr(), out_stride=S, u=u.data_ptr(), actions=acts1.data_ptr(), act_stride=1,
                  rew=rew.data_ptr(), done=done.data_ptr(), val_prev=vp[1].data_ptr(),
                  rewards=bufs["rewards"][1].data_ptr(), dones=bufs["dones"][1].data_ptr(),
                  deltas=bufs["deltas"][1].data_ptr(), T=T, t_rec=t_rec, slot0=0, gamma=0.99, pong=1, bootstrap=boot)
        torch.cuda.synchronize()
        assert torch.equal(got_state, want_state)
        close("heads", hb, want_heads, 1e-5, 1e-5)             # fp32, different summation order
        assert torch.equal(acts1, acts0)
        assert torch.equal(vp[1], vp[0])
        assert torch.equal(bufs["dones"][1], bufs["dones"][0])
        if boot:          # bootstrap adds gamma * V computed by either path: tolerance of the heads
            close("rewards", bufs["rewards"][1], bufs["rewards"][0], 1e-5, 1e-5)
            close("deltas", bufs["deltas"][1], bufs["deltas"][0], 1e-5, 1e-5)
        else:
            assert torch.equal(bufs["rewards"][1], bufs["rewards"][0])
            assert torch.equal(bufs["deltas"][1], bufs["deltas"][0])
    # copy mode (step 0 of a slot): state = prev rows as they are, no bookkeeping
    got_state = torch.empty(B, S, device=DEV)
    net._step(B, st, prev=prev.data_ptr(), prev_stride=S, out=got_state.data_ptr(), out_stride=S)
    out = net._fwd(prev.data_ptr(), S, B, "upd", st, False)
    torch.cuda.synchronize()
    assert torch.equal(got_state, prev)
    close("vals", net._heads("roll", B)[2], out["vals"], 1e-5, 1e-5)
    # unsupported shapes are refused, not silently mis-computed
    assert not ops.a3c_step_supported(3, 84, 84, 6) and not ops.a3c_step_supported(4, 84, 84, 9)
    assert ops.a3c_step_supported(4, 84, 84, 6)


# ------------------------------------------------------------------ Updater
@pytest.mark.parametrize("case", UPDATE_CASES, ids=[c[0] for c in UPDATE_CASES])
def test_updater_golden(golden, case):
    from a2c_amd.updater import Updater
    g = golden["g6_update"]
    name, kind, ss, A, h, R_, T, opt, norm_advs, nstep, use_bptt, n_upd = case
    net = make_net(kind, ss, A, h)
    hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, norm_advs=norm_advs, use_nstep_rets=nstep,
                     use_bptt=use_bptt, h_size=h)
    upd = Updater(net, hyps)
    assert [n for n, _ in net.named_parameters()] == list(g[name + "_param_names"])
    for u in range(n_upd):
        D = synth_shared(kind, ss, A, h, R_, T, seed=700 + 10 * u, recurrent=net.is_recurrent)
        D = {k: (v.to(DEV) if k != "actions" else v) for k, v in D.items()}      # actions stay on the host
        info = upd.update_model(D)
        pre = f"{name}_u{u}_"
        for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy"):
            assert info[k] == pytest.approx(float(g[pre + k]), rel=3e-5, abs=2e-6), (k, info[k], float(g[pre + k]))
        # GradNorm: torch's CPU clip_grad_norm_ sums squares in fp32, which is itself ~1.4e-4 low on
        # multi-million-element tensors (checked against the fp64 norm of the reference's own grads:
        # DESIGN.md "Numerics"); the HIP reduction accumulates in fp64.  Hence the wider tolerance.
        assert info["GradNorm"] == pytest.approx(float(g[pre + "GradNorm"]), rel=3e-4), info["GradNorm"]
        b = upd._bufs
        assert np.array_equal(b["advs"].cpu().numpy(), g[pre + "advs_raw"])          # scans are bit-exact
        if not nstep:
            assert np.array_equal(b["rets"].cpu().numpy(), g[pre + "returns"])
        gnorm = float(g[pre + "GradNorm"])
        for j, (n, p) in enumerate(net.named_parameters()):
            idx = torch.from_numpy(sample_idx(p.numel()))
            has = bool(g[pre + "has_grad"][j])
            assert (n not in net._unused_params) == has, n
            if has:
                gr = net.G(n)
                want_n = float(g[pre + "grad_norms"][j])
                assert float(gr.double().norm()) == pytest.approx(want_n, rel=4e-4, abs=1e-6 * gnorm + 1e-9), n
                # update 0: identical weights on both sides.  Later updates start from weights that differ by the first
                # RMSprop / Adam step's amplification of gradient noise (see below): their sampled gradients agree to ~5e-4
                close(f"grad samples {n}", gr.reshape(-1)[idx.to(DEV)], g[pre + "grad_samples"][j],
                      2e-5 * max(want_n / max(p.numel(), 1) ** 0.5, 1e-7) + 1e-9, 4e-4 if u == 0 else 1e-3)
            assert float(p.detach().double().norm()) == pytest.approx(float(g[pre + "param_norms"][j]), rel=1e-5), n
            # a first RMSprop/Adam step moves every weight by ~lr*10 / ~lr regardless of |g|, and the
            # direction of a noise-level gradient is not pinned by fp32: tolerance = 2 steps of lr*10
            close(f"param samples {n}", p.detach().reshape(-1)[idx.to(DEV)], g[pre + "param_samples"][j], 3e-5, 1e-5)


@pytest.mark.parametrize("case", CHECKPOINT_CASES, ids=[c[0] for c in CHECKPOINT_CASES])
def test_resume_from_reference_written_checkpoint(golden, case):
    """tests/golden/g7_*_{net,optim}.p were written by the REFERENCE's Updater.save_model
    (updater.py:211-219) after one update; loading them here (training.py's resume path:
    net.load_state_dict / optim.load_state_dict) and running the next update must land on the
    weights the reference reached."""
    from a2c_amd.updater import Updater
    g = golden["g7_checkpoint"]
    name, kind, ss, A, h, R_, T, opt, use_bptt = case
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import a2c_amd
    net = getattr(a2c_amd.models, kind)(list(ss), A, h_size=h, bnorm=False)
    net.load_state_dict(torch.load(os.path.join(gdir, f"g7_{name}_net.p"), weights_only=False))
    hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, use_bptt=use_bptt, h_size=h)
    upd = Updater(net, hyps)
    upd.optim.load_state_dict(torch.load(os.path.join(gdir, f"g7_{name}_optim.p"), weights_only=False))
    D = synth_shared(kind, ss, A, h, R_, T, seed=920, recurrent=net.is_recurrent)
    D = {k: (v.to(DEV) if k != "actions" else v) for k, v in D.items()}
    info = upd.update_model(D)
    for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy"):
        assert info[k] == pytest.approx(float(g[f"{name}_{k}"]), rel=3e-5, abs=2e-6), k
    assert info["GradNorm"] == pytest.approx(float(g[f"{name}_GradNorm"]), rel=3e-4)
    assert [n for n, _ in net.named_parameters()] == list(g[name + "_param_names"])
    for n, p in net.named_parameters():
        close(f"param {n}", p.detach(), g[f"{name}_param_{n}"], 3e-5, 1e-5)     # two optimiser steps of lr*10 (see above)


def test_updater_optimizer_state_dict_matches_torch_layout():
    from a2c_amd.updater import Updater
    kind, ss, A, h = "A3CModel", (4, 84, 84), 3, 256
    for opt in ("RMSprop", "Adam"):
        net = make_net(kind, ss, A, h)
        hyps = base_hyps(n_tsteps=4, n_rollouts=2, optim_type=opt)
        upd = Updater(net, hyps)
        D = synth_shared(kind, ss, A, h, 2, 4, seed=700, recurrent=False)
        upd.update_model({k: v.to(DEV) for k, v in D.items()})
        sd = upd.optim.state_dict()
        onet = O.OracleNet(kind, ss, A, h)
        oupd = O.OracleUpdater(onet, hyps)
        oupd.update_model(D)
        ref = oupd.optim.state_dict()
        assert sd["param_groups"][0]["params"] == ref["param_groups"][0]["params"]
        assert set(sd["state"].keys()) == set(ref["state"].keys())       # emb_bnorm has no state
        for k in ref["state"]:
            assert set(ref["state"][k].keys()) == set(sd["state"][k].keys())
            for s in ref["state"][k]:
                if s == "step":
                    assert float(sd["state"][k][s]) == float(ref["state"][k][s])
                else:
                    close(f"{opt} state {k}.{s}", sd["state"][k][s], ref["state"][k][s], 1e-9, 2e-3)
        # reload into a fresh updater and into torch itself
        upd2 = Updater(make_net(kind, ss, A, h), hyps)
        upd2.optim.load_state_dict(sd)
        assert upd2.optim._steps == 1
        getattr(torch.optim, opt)(onet.parameters(), lr=1e-4).load_state_dict(
            {"state": {k: {s: (v.cpu() if torch.is_tensor(v) else v) for s, v in st.items()}
                       for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]})


# ------------------------------------------------------------------ step kernel at scale
@pytest.mark.parametrize("B", [300, 2048])
def test_a3c_step_kernel_many_envs_vs_layered_and_oracle(B, monkeypatch):
    """more workgroups than CUs (BASELINE config 5 runs 2048 envs): the one-launch step kernel against the
    per-layer path on every env, and against the oracle on sampled envs"""
    from a2c_amd.runner import Runner
    T, A, ss = 3, 4, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=2 + j % 3, done_period=3 + j % 5) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    us = torch.from_numpy(hashf(T * B, 4242, 0, 1).reshape(T, B))
    usd = us.to(DEV)

    out = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("A2C_NO_FUSED_STEP", "1")
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        from a2c_amd.runner import HostEnvPool
        pool = HostEnvPool([O.FakeEnv(**k) for k in ekws])
        r = Runner(D, hyps, None, None, None, env_pool=pool, uniform_fn=lambda t, Bn, e0: usd[t, e0:e0 + Bn].contiguous())
        r.rollout(net, list(range(B)), hyps)
        torch.cuda.synchronize()
        out[fused] = {k: v.cpu() for k, v in D.items()}
    for k in ("states", "actions", "dones"):
        assert torch.equal(out[True][k], out[False][k]), k
    close("rewards", out[True]["rewards"], out[False]["rewards"], 1e-5, 1e-5)      # last step: + gamma * V(bootstrap)
    close("deltas", out[True]["deltas"], out[False]["deltas"], 1e-5, 1e-5)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    for j in (0, 255, 256, B - 1):
        Do = dict(states=torch.zeros(T, *ss), deltas=torch.zeros(T), rewards=torch.zeros(T), dones=torch.zeros(T),
                  actions=torch.zeros(T).long())
        it = iter([float(us[t, j]) for t in range(T)])
        sr = O.SlotRunner(O.FakeEnv(**ekws[j]), Do, hyps, uniform_fn=lambda it=it: next(it))
        sr.start(onet)
        sr.rollout(onet, 0)
        sl = slice(j * T, (j + 1) * T)
        assert torch.equal(out[True]["actions"][sl], Do["actions"]) and torch.equal(out[True]["states"][sl], Do["states"])
        close("deltas", out[True]["deltas"][sl], Do["deltas"], 1e-5, 1e-5)
