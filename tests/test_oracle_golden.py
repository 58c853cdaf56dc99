"""CPU: the oracle (oracle/a2c_oracle.py) against the outputs recorded from the
reference itself (tests/golden/*.npz, made by tests/golden/make_golden.py).
This is what pins the oracle; the -m gpu tests then compare the HIP path with it."""
import queue
from collections import deque

import numpy as np
import pytest
import torch

from oracle import a2c_oracle as O
from cases import (MODEL_CASES, ROLLOUT_CASES, UPDATE_CASES, CHECKPOINT_CASES, SAMPLE, hashf, base_hyps, synth_shared,
                   sample_idx)

torch.set_num_threads(1)


def test_g1_discount_bitexact(golden):
    g = golden["g1_discount"]
    for i in range(int(g["n_cases"])):
        x, d, f, y = g[f"x{i}"], g[f"d{i}"], float(g[f"g{i}"]), g[f"y{i}"]
        got = O.discount(torch.from_numpy(x), torch.from_numpy(d), f).numpy()
        assert np.array_equal(got, y), i
        assert np.array_equal(O.discount_np(x, d, f), y), i


def test_g1_known_answers():
    y = O.discount(torch.tensor([1., 1, 1, 1]), torch.tensor([0., 1, 0, 1]), 0.5)
    assert y.tolist() == [1.5, 1.0, 1.5, 1.0]


def test_g2_sample_action(golden):
    g = golden["g2_sample_action"]
    for i in range(int(g["n_cases"])):
        a = O.sample_action(torch.from_numpy(g[f"p{i}"]), torch.from_numpy(g[f"u{i}"]))
        assert np.array_equal(a.numpy(), g[f"a{i}"]), i
    assert g["a0"].item() == 0 and g["a1"].item() == 2 and g["a2"].item() == -1 and g["a3"].item() == 0


def test_g3_next_state(golden):
    g = golden["g3_next_state"]

    class E:
        k = 0

        def reset(self):
            self.k += 1
            return np.full((1, 2, 3), 7.0 * self.k)
    env, dq = E(), deque(maxlen=3)
    seq = [(None, True), (np.full((1, 2, 3), 9.0), False), (np.full((1, 2, 3), 11.0), False),
           (np.full((1, 2, 3), 13.0), True), (np.full((1, 2, 3), 15.0), False)]
    outs = np.stack([O.next_state(env, dq, o, r) for o, r in seq])
    assert outs.dtype == np.float64 == np.dtype(str(g["dtype"]))
    assert np.array_equal(outs, g["states"])


@pytest.mark.parametrize("i", range(len(MODEL_CASES)))
def test_g4_model_forward(golden, i):
    g = golden["g4_model_forward"]
    kind, ss, A, h, B = MODEL_CASES[i]
    net = O.OracleNet(kind, ss, A, h)
    assert sum(p.numel() for p in net.parameters()) == int(g[f"nparams{i}"])
    x = torch.from_numpy(O.formula_frames(B, ss, seed=400 + i, binary=(len(ss) == 3 and ss[-1] == 84)))
    with torch.no_grad():
        if net.is_recurrent:
            hin = torch.from_numpy(hashf(B * h, 450 + i, -1, 1).reshape(B, h))
            v, p, hn = net(x, hin)
            np.testing.assert_allclose(hn.numpy(), g[f"h{i}"], rtol=0, atol=1e-6)
        else:
            v, p = net(x)
    np.testing.assert_allclose(v.numpy(), g[f"val{i}"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(p.numpy(), g[f"pi{i}"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", ROLLOUT_CASES, ids=[c[0] for c in ROLLOUT_CASES])
def test_g5_rollout(golden, case):
    g = golden["g5_rollout"]
    name, kind, env_type, T, n_slots, ekw, A = case
    hyps = base_hyps(env_type=env_type, n_tsteps=T, n_rollouts=n_slots,
                     action_shift=1 if "Pong" in env_type else 0)
    ss = (4, 84, 84)
    net = O.OracleNet(kind, ss, A, 256)
    N = T * n_slots
    D = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N),
             actions=torch.zeros(N).long(), dones=torch.zeros(N))
    if net.is_recurrent:
        D["h_states"] = torch.zeros(N, 256)
    it = iter(g[f"{name}_uniforms"])
    r = O.SlotRunner(O.FakeEnv(**ekw), D, hyps, uniform_fn=lambda: float(next(it)))
    r.start(net)
    for idx in range(n_slots):
        r.rollout(net, idx)
    assert np.array_equal(D["actions"].numpy(), g[f"{name}_actions"])
    assert np.array_equal(D["dones"].numpy(), g[f"{name}_dones"])
    np.testing.assert_allclose(D["rewards"].numpy(), g[f"{name}_rewards"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(D["deltas"].numpy(), g[f"{name}_deltas"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(D["states"].reshape(N, 4, -1).double().sum(2).numpy(),
                               g[f"{name}_state_frame_sums"], rtol=0, atol=0)
    if net.is_recurrent:
        np.testing.assert_allclose(D["h_states"].numpy(), g[f"{name}_h_states"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(r.h.numpy(), g[f"{name}_h_bookmark"], rtol=0, atol=1e-6)
    assert abs(r.avg_rew - float(g[f"{name}_avg_rew"])) < 1e-12
    assert float(np.asarray(r.state).sum()) == float(g[f"{name}_bookmark_sum"])


@pytest.mark.parametrize("case", UPDATE_CASES, ids=[c[0] for c in UPDATE_CASES])
def test_g6_update(golden, case):
    g = golden["g6_update"]
    name, kind, ss, A, h, R_, T, opt, norm_advs, nstep, use_bptt, n_upd = case
    net = O.OracleNet(kind, ss, A, h)
    hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, norm_advs=norm_advs,
                     use_nstep_rets=nstep, use_bptt=use_bptt, h_size=h)
    upd = O.OracleUpdater(net, hyps)
    assert [n for n, _ in net.named_parameters()] == list(g[name + "_param_names"])
    for u in range(n_upd):
        D = synth_shared(kind, ss, A, h, R_, T, seed=700 + 10 * u, recurrent=net.is_recurrent)
        info, ex = upd.update_model(D, keep=True)
        pre = f"{name}_u{u}_"
        for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy", "GradNorm"):
            assert info[k] == pytest.approx(float(g[pre + k]), rel=2e-5, abs=1e-6), k
        assert np.array_equal(ex["advs_raw"].numpy(), g[pre + "advs_raw"])
        if not nstep:
            assert np.array_equal(ex["returns"].numpy(), g[pre + "returns"])
        for j, (n, p) in enumerate(net.named_parameters()):
            idx = sample_idx(p.numel())
            gr = ex["grads"][n]
            assert (gr is not None) == bool(g[pre + "has_grad"][j]), n
            if gr is not None:
                assert float(gr.double().norm()) == pytest.approx(float(g[pre + "grad_norms"][j]), rel=1e-4, abs=1e-9), n
                np.testing.assert_allclose(gr.reshape(-1)[idx].numpy(), g[pre + "grad_samples"][j], rtol=1e-4, atol=1e-7, err_msg=n)
            assert float(p.detach().double().norm()) == pytest.approx(float(g[pre + "param_norms"][j]), rel=1e-6), n
            np.testing.assert_allclose(p.detach().reshape(-1)[idx].numpy(), g[pre + "param_samples"][j], rtol=0, atol=2e-5, err_msg=n)


def _c_oracle():
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    lib = ctypes.CDLL(os.path.join(root, "oracle", "_ref", "liba2c_oracle.so"))
    lib.oracle_td_delta.restype = ctypes.c_float
    lib.oracle_td_delta.argtypes = [ctypes.c_float] * 5
    return lib, ctypes


@pytest.mark.parametrize("case", CHECKPOINT_CASES, ids=[c[0] for c in CHECKPOINT_CASES])
def test_g7_resume_from_reference_checkpoint(golden, case):
    """the oracle resumed from the checkpoint files the reference wrote (Updater.save_model,
    updater.py:211-219) reaches the reference's next weights"""
    import os
    g = golden["g7_checkpoint"]
    name, kind, ss, A, h, R_, T, opt, use_bptt = case
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sd = torch.load(os.path.join(gdir, f"g7_{name}_net.p"), weights_only=False)
    net = O.OracleNet(kind, ss, A, h, state_dict=sd)
    hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, use_bptt=use_bptt, h_size=h)
    upd = O.OracleUpdater(net, hyps)
    upd.optim.load_state_dict(torch.load(os.path.join(gdir, f"g7_{name}_optim.p"), weights_only=False))
    info = upd.update_model(synth_shared(kind, ss, A, h, R_, T, seed=920, recurrent=net.is_recurrent))
    for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy", "GradNorm"):
        assert float(info[k]) == pytest.approx(float(g[f"{name}_{k}"]), rel=1e-5, abs=1e-7), k
    assert [n for n, _ in net.named_parameters()] == list(g[name + "_param_names"])
    for n, p in net.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), g[f"{name}_param_{n}"], rtol=1e-5, atol=1e-6, err_msg=n)


def test_c_oracle_matches_golden(golden):
    """the C part of the oracle (oracle/a2c_oracle.c) on the same recorded reference outputs"""
    lib, ct = _c_oracle()
    fp = lambda a: a.ctypes.data_as(ct.c_void_p)
    g = golden["g1_discount"]
    for i in range(int(g["n_cases"])):
        x, d, y = np.ascontiguousarray(g[f"x{i}"]), np.ascontiguousarray(g[f"d{i}"]), g[f"y{i}"]
        out = np.empty_like(x)
        lib.oracle_discount(fp(x), fp(d), fp(out), ct.c_int64(len(x)), ct.c_float(float(g[f"g{i}"])))
        assert np.array_equal(out, y), i
    g = golden["g2_sample_action"]
    for i in range(int(g["n_cases"])):
        p, u = np.ascontiguousarray(g[f"p{i}"]), np.ascontiguousarray(g[f"u{i}"])
        out = np.empty(p.shape[0], np.float32)
        lib.oracle_sample_action(fp(p), fp(u), fp(out), ct.c_int64(p.shape[0]), ct.c_int(p.shape[1]))
        assert np.array_equal(out, g[f"a{i}"].reshape(-1)), i
    t = torch.tensor
    want = (t(0.5) + t(0.99) * t(0.3) * (1 - t(0.0)) - t(0.2)).item()
    assert lib.oracle_td_delta(0.5, 0.99, 0.3, 0.0, 0.2) == want


def test_stats_rollout_matches_the_reference(golden):
    """oracle restatement of StatsRunner.rollout (runner.py:274-314) vs the reference's own run (g8)"""
    from cases import STATS_CASES
    g = golden["g8_stats"]
    for (name, kind, env_type, n_eps, ekw, A) in STATS_CASES:
        hyps = base_hyps(env_type=env_type, n_test_eps=n_eps, action_shift=1 if "Pong" in env_type else 0)
        net = O.OracleNet(kind, (4, 84, 84), A, 256)
        us = g[f"{name}_uniforms"]
        it = iter(us)
        env = O.FakeEnv(**ekw)
        env.reset()                       # SequentialEnvironment.__init__'s probing reset (runner.py:45)
        avg = O.stats_rollout(net, env, hyps, n_eps, lambda: float(next(it)))
        assert avg == float(g[f"{name}_avg_rew"])
        assert next(it, None) is None     # consumed exactly the uniforms the reference drew


@pytest.mark.parametrize("name", ["gru_bptt_rms", "gru_rms", "conv_small_rms", "a3c_rms", "grufc_bptt_rms"])
def test_chunked_oracle_gradients_equal_the_full_batch_update(golden, name):
    """O.update_grads_chunked (the checker of the full-size GRU+BPTT test, where full-batch autograd does not fit host
    memory) against (a) the losses the REFERENCE reported for the case (g6) and (b) OracleUpdater's gradients on the whole
    batch, for chunks of 1 and 2 slots, in fp32 and fp64."""
    case = [c for c in UPDATE_CASES if c[0] == name][0]
    _, kind, ss, A, h, R_, T, opt, norm_advs, nstep, use_bptt, _ = case
    g = golden["g6_update"]
    hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, norm_advs=norm_advs, use_nstep_rets=nstep,
                     use_bptt=use_bptt, h_size=h, max_norm=1e9)          # no clipping: compare raw gradients
    net = O.OracleNet(kind, ss, A, h)
    D = synth_shared(kind, ss, A, h, R_, T, seed=700, recurrent=net.is_recurrent)
    _, extra = O.OracleUpdater(O.OracleNet(kind, ss, A, h), hyps).update_model(D, keep=True)
    for chunk, dt in ((1, torch.float32), (2, torch.float32), (2, torch.float64)):
        if dt == torch.float64:
            net = O.OracleNet(kind, ss, A, h, state_dict={k: v.double() for k, v in O.formula_state_dict(kind, ss, A, h).items()})
        info, grads = O.update_grads_chunked(net, D, hyps, chunk, dtype=dt)
        for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy"):
            ref = float(g[f"{name}_u0_{k}"])
            assert info[k] == pytest.approx(ref, rel=2e-5, abs=2e-6), (chunk, dt, k, info[k], ref)
        for n, gr in extra["grads"].items():
            if gr is None:
                assert grads[n] is None, n
                continue
            scale = float(gr.abs().max()) + 1e-12
            assert float((grads[n].float() - gr).abs().max()) <= 2e-5 * scale + 1e-7, (chunk, dt, n)
