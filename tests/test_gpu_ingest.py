"""-m gpu: the north-star ingest -- env stepping in host worker processes, frames through ONE pinned,
device-mapped region (uint8 when the preprocessor yields uint8) -- against the CPU oracle's batch-1
runners on the same deterministic envs (runner.py:174-248).  Covers both ingest modes of the Runner
(memcpy: hipMemcpyAsync per step behind a host hand-off, every model; relay: the same segments with the
hand-off done by the device, a2c_pool_publish_actions / a2c_pool_ingest; zero-copy: the persistent one-launch
rollout a2c_a3c_rollout), rollout -> update -> rollout sequences (stale derived weights would show), more envs
than CUs, fp32 frames, worker failure and the host time-out of the persistent kernel."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import F32FakeEnv, FailingEnv, U8FakeEnv, base_hyps, hashf  # noqa: E402
from test_gpu_kernels import close  # noqa: E402
from test_gpu_models import _datas, make_net  # noqa: E402

DEV = "cuda"


def _pool(cls, ekws, n_workers, **kw):
    from a2c_amd.hostpool import ProcessEnvPool
    return ProcessEnvPool(cls, len(ekws), env_kwargs=ekws, n_workers=n_workers, probe_reset=True, **kw)


def _oracle_rollouts(kind, onet, hyps, ekws, us, n_rounds, B, T, ss, env_cls=O.FakeEnv, updater=None):
    """env j plays slot j of every round; `updater` (oracle) runs between rounds when given"""
    N = B * T
    Do = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N), dones=torch.zeros(N),
              actions=torch.zeros(N).long())
    if onet.is_recurrent:
        Do["h_states"] = torch.zeros(N, onet.h_size)
    runners = []
    for j in range(B):
        seq = iter([float(us[k, t, j]) for k in range(n_rounds) for t in range(T)])
        sr = O.SlotRunner(env_cls(**ekws[j]), Do, hyps, uniform_fn=lambda seq=seq: next(seq))
        sr.start(onet)
        runners.append(sr)
    outs = []
    for k in range(n_rounds):
        for j in range(B):
            runners[j].rollout(onet, j)
        outs.append({n: v.clone() for n, v in Do.items()})
        if updater is not None and k + 1 < n_rounds:
            outs[-1]["info"] = updater.update_model(Do)
    return outs


def _compare_round(D, ref, recurrent, tol=1e-5):
    assert torch.equal(D["actions"].cpu(), ref["actions"])
    assert torch.equal(D["dones"].cpu(), ref["dones"])
    assert torch.equal(D["states"].cpu(), ref["states"])
    close("rewards", D["rewards"], ref["rewards"], tol, tol)
    close("deltas", D["deltas"], ref["deltas"], tol, tol)
    if recurrent:
        close("h_states", D["h_states"], ref["h_states"], tol, tol)


@pytest.mark.parametrize("kind,ingest,fused", [("A3CModel", "zero-copy", True), ("A3CModel", "memcpy", True),
                                               ("A3CModel", "memcpy", False), ("GRUModel", "memcpy", False),
                                               ("ConvModel", "memcpy", False), ("FCModel", "memcpy", False),
                                               ("A3CModel", "relay", True), ("A3CModel", "relay", False),
                                               ("GRUModel", "relay", False), ("ConvModel", "relay", False),
                                               ("FCModel", "relay", False)])
def test_process_pool_rollouts_match_oracle(kind, ingest, fused, monkeypatch):
    """two consecutive rounds of B slots (the second continues the envs, bookmarks and hidden states)"""
    from a2c_amd.runner import Runner
    if not fused:
        monkeypatch.setenv("A2C_NO_FUSED_STEP", "1")
    B, T, A, ss = 5, 6, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0" if kind == "A3CModel" else "FakeBreakout", n_tsteps=T, n_rollouts=B,
                     action_shift=0, n_envs=B, env_timeout_s=20.0)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    D = _datas(B * T, ss, net.is_recurrent, actions_on_host=(kind == "GRUModel" and ingest != "relay"))
    us = torch.from_numpy(hashf(2 * T * B, 901, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd = [0]
    pool = _pool(U8FakeEnv, ekws, 2, pong="Pong" in hyps["env_type"])
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        refs = _oracle_rollouts(kind, onet, hyps, ekws, us, 2, B, T, ss)
        for rnd[0] in range(2):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            _compare_round(D, refs[rnd[0]], net.is_recurrent)
        assert pool.frame_dtype == np.uint8 and pool.seq == 2 * T
        if ingest == "zero-copy":
            assert r._zero_copy_ok(net)
    finally:
        r.close()


@pytest.mark.parametrize("kind,ingest,bptt", [("A3CModel", "zero-copy", False), ("A3CModel", "memcpy", False),
                                              ("GRUModel", "memcpy", False), ("GRUModel", "relay", False),
                                              ("ConvModel", "relay", False), ("GRUModel", "relay", True),
                                              ("GRUFCModel", "relay", True)])
def test_rollout_update_rollout_matches_oracle(kind, ingest, bptt):
    """three rounds with update_model in between (training.py:163-165): the rollout after an optimiser step
    must use the NEW weights (derived inference weights -- composed heads, conv fragments -- re-built).
    With use_bptt the update is Updater.bptt (updater.py:139-169); for GRUModel behind the relay that is the path the
    benchmark's config 4 runs: every GRU cell, embedding row and heads row of the update comes from the rollout's stash
    (models.py bptt_forward "has nothing left to compute") -- compared here with the oracle directly."""
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    B, T = 4, 5
    if kind == "GRUFCModel":
        A, ss, h, env_cls = 2, (4, 4), 64, O.FakeEnv
        ekws = [dict(env_id=j, frame_shape=(1, 4), binary=False, rew_period=2 + j % 2, done_period=4 + j) for j in range(B)]
    else:
        A, ss, h, env_cls = 3, (4, 84, 84), 256, U8FakeEnv
        ekws = [dict(env_id=j, rew_period=2 + j % 2, done_period=4 + j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-2,
                     optim_type="RMSprop", use_bptt=bptt, h_size=h)
    net = make_net(kind, ss, A, h)
    onet = O.OracleNet(kind, ss, A, h)
    D = _datas(B * T, ss, net.is_recurrent, h=h, actions_on_host=False)
    us = torch.from_numpy(hashf(3 * T * B, 977, 0, 1).reshape(3, T, B))
    usd = us.to(DEV)
    rnd = [0]
    pool = _pool(env_cls, ekws, 2, pong=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    upd = Updater(net, hyps)
    try:
        refs = _oracle_rollouts(kind, onet, hyps, ekws, us, 3, B, T, ss, env_cls=O.FakeEnv,
                                updater=O.OracleUpdater(onet, hyps))
        for rnd[0] in range(3):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            if kind == "GRUModel" and bptt and ingest == "relay":
                assert net._cells_stashed(D["states"], B, T)      # the update below takes its cells from the rollout
            # lr = 1e-2 moves the weights by ~1e-2 per step: stale weights would change values by >> tolerance;
            # after two such steps the nets agree to ~1e-4 relative (RMSprop's 1/sqrt(v) amplifies 1e-7 gradient noise)
            tol = 1e-5 if rnd[0] == 0 else 2e-3
            assert torch.equal(D["states"].cpu(), refs[rnd[0]]["states"])
            assert torch.equal(D["dones"].cpu(), refs[rnd[0]]["dones"])
            close("deltas", D["deltas"], refs[rnd[0]]["deltas"], tol, tol)
            if net.is_recurrent:
                close("h_states", D["h_states"], refs[rnd[0]]["h_states"], tol, tol)
            mism = (D["actions"].cpu() != refs[rnd[0]]["actions"]).sum().item()
            assert mism == 0 if rnd[0] == 0 else mism <= 1, mism
            if rnd[0] < 2:
                info = upd.update_model(D)
                oi = refs[rnd[0]]["info"]
                for k in oi:
                    assert abs(info[k] - oi[k]) <= 1e-4 + 2e-3 * abs(oi[k]), (rnd[0], k, info[k], oi[k])
        # and the weights really moved: round 2's values differ from what the ORIGINAL weights give
        assert not torch.allclose(D["deltas"].cpu(), _oracle_rollouts(kind, O.OracleNet(kind, ss, A, h), hyps, ekws, us, 3, B,
                                                                      T, ss)[2]["deltas"], atol=1e-3)
    finally:
        r.close()


@pytest.mark.parametrize("ring_blocks", [True, False])
def test_zero_copy_more_envs_than_cus_and_sampled_oracle(ring_blocks, monkeypatch):
    """300 envs on 256 CUs (a2c_a3c_rollout): two interleaved blocks of the ring kernel one after the other (default), or
    -- A2C_RING_BLOCKS=0 -- ONE launch of the per-step body whose workgroups play several envs in turn; every env's data
    against the layered per-step path (the ring kernel sums conv1 plane-major: its values, hence its deltas, agree to fp32
    re-association, everything else bit for bit), sampled envs against the oracle"""
    from a2c_amd.runner import Runner
    monkeypatch.setenv("A2C_RING_BLOCKS", "1" if ring_blocks else "0")
    B, T, A, ss = 300, 4, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=2 + j % 3, done_period=3 + j % 7) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    us = torch.from_numpy(hashf(T * B, 31337, 0, 1).reshape(1, T, B))
    usd = us.to(DEV)
    out = {}
    for ingest in ("zero-copy", "memcpy", "relay"):
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        pool = _pool(U8FakeEnv, ekws, 6, pong=True)
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
                   uniform_fn=lambda t, Bn, env0: usd[0, t, env0:env0 + Bn].contiguous())
        try:
            r.rollout(net, list(range(B)), hyps)
            r.finish()
        finally:
            r.close()
        out[ingest] = {k: v.cpu() for k, v in D.items()}
    for k in ("states", "actions", "dones", "rewards", "deltas"):
        if k in ("deltas", "rewards") and ring_blocks:      # (rewards: the bootstrap adds gamma * V to a slot's last one)
            close(k + " (ring)", out["zero-copy"][k], out["memcpy"][k], 2e-6, 1e-5)
        else:
            assert torch.equal(out["zero-copy"][k], out["memcpy"][k]), k
        assert torch.equal(out["relay"][k], out["memcpy"][k]), k      # the device relay hands over the same bytes
    onet = O.OracleNet("A3CModel", ss, A, 256)
    for j in (0, 137, 255, 256, 299):
        Do = dict(states=torch.zeros(T, *ss), deltas=torch.zeros(T), rewards=torch.zeros(T), dones=torch.zeros(T),
                  actions=torch.zeros(T).long())
        it = iter([float(us[0, t, j]) for t in range(T)])
        sr = O.SlotRunner(O.FakeEnv(**ekws[j]), Do, hyps, uniform_fn=lambda it=it: next(it))
        sr.start(onet)
        sr.rollout(onet, 0)
        sl = slice(j * T, (j + 1) * T)
        assert torch.equal(out["zero-copy"]["actions"][sl], Do["actions"]) and torch.equal(out["zero-copy"]["states"][sl], Do["states"])
        close("deltas", out["zero-copy"]["deltas"][sl], Do["deltas"], 1e-5, 1e-5)


def test_fp32_frames_travel_as_fp32():
    """grey-level float frames (breakout_prep-like) keep the fp32 transport"""
    from a2c_amd.runner import Runner
    B, T, A, ss = 3, 5, 4, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=3, done_period=4 + j) for j in range(B)]
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    net = make_net("A3CModel", ss, A, 256)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False)
    us = torch.from_numpy(hashf(T * B, 77, 0, 1).reshape(1, T, B))
    usd = us.to(DEV)
    pool = _pool(F32FakeEnv, ekws, 2)
    r = Runner(D, hyps, None, None, None, env_pool=pool, uniform_fn=lambda t, Bn, env0: usd[0, t, env0:env0 + Bn].contiguous())
    try:
        ref = _oracle_rollouts("A3CModel", onet, hyps, ekws, us, 1, B, T, ss, env_cls=F32FakeEnv)[0]
        r.rollout(net, list(range(B)), hyps)
        r.finish()
        assert pool.frame_dtype == np.float32 and not r._zero_copy_ok(net)
        _compare_round(D, ref, False)
    finally:
        r.close()


def test_persistent_rollout_times_out_instead_of_hanging():
    """an env worker that dies mid-slot: the kernel's bounded wait sets the error flag and the launch ends"""
    from a2c_amd.runner import Runner
    B, T, A, ss = 4, 8, 3, (4, 84, 84)
    ekws = [dict(env_id=j, fail_at=3 if j == 1 else 10 ** 9) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, env_timeout_s=1.0)
    net = make_net("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False, actions_on_host=False)
    pool = _pool(FailingEnv, ekws, 2, pong=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy")
    try:
        r.rollout(net, list(range(B)), hyps)
        with pytest.raises((TimeoutError, RuntimeError)):
            r.finish()
    finally:
        r.close()


def test_device_relay_rollout_times_out_instead_of_hanging():
    """the same for the per-step rollout of the other models: the segments wait for the workers on the DEVICE
    (a2c_pool_ingest); a dead worker ends the first wait after the time-out and every later one at once"""
    from a2c_amd.runner import Runner
    B, T, A, ss = 4, 8, 3, (4, 84, 84)
    ekws = [dict(env_id=j, fail_at=3 if j == 1 else 10 ** 9) for j in range(B)]
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, env_timeout_s=1.0)
    net = make_net("ConvModel", ss, A, 256)
    D = _datas(B * T, ss, False, actions_on_host=False)
    pool = _pool(FailingEnv, ekws, 2, pong=False)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="relay")
    t0 = time.time()
    try:
        r.rollout(net, list(range(B)), hyps)
        with pytest.raises((TimeoutError, RuntimeError)):
            r.finish()
        assert time.time() - t0 < 15.0          # one 1 s wait, not one per remaining segment
    finally:
        r.close()


def test_rollout_sampler_never_returns_minus_one():
    """u = nextafter(1, 0): the fp32 cumsum of a softmax can stay below it; the rollout samplers return the
    last action (what -1 indexes in the loss, updater.py:104), utils.sample_action keeps the reference's -1"""
    from a2c_amd import ops, utils
    A, B = 3, 64
    logits = torch.from_numpy(hashf(B * A, 5, -3, 3).reshape(B, A)).to(DEV)
    u = torch.full((B,), float(np.nextafter(np.float32(1), np.float32(0))), device=DEV)
    acts = torch.zeros(B, dtype=torch.int64, device=DEV)
    ops.softmax_sample(logits, u, acts.data_ptr(), 1, B, A)
    torch.cuda.synchronize()
    assert int(acts.min()) >= 0 and int(acts.max()) == A - 1
    p = torch.tensor([[.3, .3, .3]])
    assert float(utils.sample_action(p, rand_nums=torch.tensor([.95]))) == -1.0


def test_full_size_headline_rollout_properties_and_action_flip_census():
    """BASELINE headline size (A3CModel, 256 envs x 128 steps) through the headline path (native env threads,
    zero-copy persistent kernel), checked by size-independent properties and a census of sampled actions:
      * every slot ends with done == 1; actions in range; rewards in {-1, 0, 1} (+ bootstrap on the last step);
      * frame stack: plane c of state t+1 == plane c+1 of state t unless the env was reset (utils.py:26-43);
      * the update's GAE / returns scans bit-exact against the sequential definition on sampled rows;
      * the oracle forward on 2,048 of the 32,768 states with the SAME uniforms samples the same action;
        a flip is only tolerated where the fp32 cumsum is within 1e-6 of the uniform (composed heads
        re-associate the logits), and at most 1e-4 of the samples may flip."""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    from a2c_amd.updater import Updater
    B, T, A, ss = 256, 128, 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    net = make_net("A3CModel", ss, A, 256)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False, actions_on_host=False)
    envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy")
    try:
        r.rollout(net, list(range(B)), hyps)
        r.finish()
        u = r._u_keep.cpu()                                    # (T, B) uniforms the kernel used
        st = D["states"].cpu().reshape(B, T, 4, -1)
        acts = D["actions"].cpu().reshape(B, T)
        dones, rews = D["dones"].cpu().reshape(B, T), D["rewards"].cpu().reshape(B, T)
        assert bool((dones[:, -1] == 1).all())
        assert int(acts.min()) >= 0 and int(acts.max()) <= A - 1
        assert set(np.unique(rews[:, :-1].numpy())) <= {-1.0, 0.0, 1.0}
        # env data == the tapes (step k of env j consumed frame k+1, reward k, done k)
        for j in (0, 17, 255):
            e = TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100)
            for t in range(1, T):
                assert np.array_equal(st[j, t, 3].numpy(), e.frames[t % (T + 1)].reshape(-1).astype(np.float32)), (j, t)
        # frame stack property on every env / step
        real_done = torch.from_numpy(np.stack([e.dones[:T] for e in envs]).astype(np.float32))
        for c in range(3):
            same = (st[:, 1:, c] == st[:, :-1, c + 1]).all(-1)
            zero = (st[:, 1:, c] == 0).all(-1)
            assert bool((same | (zero & (real_done[:, :T - 1] == 1))).all())
        # scans: bit-exact against the sequential definition (utils.py:63-79) on sampled rows
        upd = Updater(net, hyps)
        deltas = D["deltas"].cpu().reshape(B, T).numpy()
        rew_np, done_np = rews.numpy(), dones.numpy()
        info = upd.update_model(D)
        assert all(np.isfinite(v) for v in info.values())
        advs, rets = upd._bufs["advs"].cpu().reshape(B, T).numpy(), upd._bufs["rets"].cpu().reshape(B, T).numpy()
        for j in (0, 31, 128, 255):
            close("advs", advs[j], O.discount_np(deltas[j], done_np[j], hyps["gamma"] * hyps["lambda_"]), 0, 0)
            close("rets", rets[j], O.discount_np(rew_np[j], done_np[j], hyps["gamma"]), 0, 0)
        # action-flip census
        idx = (np.arange(2048, dtype=np.int64) * 2654435761 % (B * T)).astype(np.int64)
        with torch.no_grad():
            _, logits = onet(D["states"].cpu().reshape(B * T, *ss)[idx])
        p = torch.softmax(logits, -1)
        cs = torch.cumsum(p, -1)
        uu = u.t().reshape(-1)[idx]                            # sample e = slot*T + t  <->  u[t, slot]
        ref = (cs >= uu[:, None]).float().argmax(-1)
        ref[(cs < uu[:, None]).all(-1)] = A - 1
        got = acts.reshape(-1)[idx]
        flips = (ref != got).nonzero().flatten()
        assert len(flips) <= max(1, int(1e-4 * len(idx))), len(flips)
        for f in flips.tolist():
            assert float((cs[f] - uu[f]).abs().min()) < 1e-6, (f, cs[f], uu[f])
    finally:
        r.close()


@pytest.mark.parametrize("kind,ingest", [("A3CModel", "zero-copy"), ("A3CModel", "memcpy"), ("GRUModel", "memcpy"),
                                         ("ConvModel", "memcpy"), ("GRUModel", "relay")])
def test_update_from_stashed_rollout_activations_equals_recomputed(kind, ingest, monkeypatch):
    """The one-launch rollout step stashes conv1/conv2 activations of every state; update_model reads them instead
    of re-running the two conv forwards (same weights, same states: training.py:150-165).  Same update as the
    recomputed one to fp32 rounding (the step kernel's K-split conv2 re-associates the sums)."""
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    B, T, A, ss = 6, 5, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=2 + j % 2, done_period=4 + j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-3,
                     use_bptt=(kind == "GRUModel"))
    res = {}
    for stash in (True, False):
        if not stash:
            monkeypatch.setenv("A2C_NO_STASH", "1")
        net = make_net(kind, ss, A, 256)
        D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
        pool = _pool(U8FakeEnv, ekws, 2, pong=True)
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest)
        torch.manual_seed(5)          # same sampling uniforms in both runs
        try:
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            assert (net._stash is not None) == stash
            info = Updater(net, hyps).update_model(D)
            assert net._stash is None          # the optimiser step invalidated it
            res[stash] = (info, net._arena.train_grads().cpu().clone(), {k: v.cpu().clone() for k, v in D.items()})
        finally:
            r.close()
    for k in ("states", "actions", "rewards", "deltas"):
        assert torch.equal(res[True][2][k], res[False][2][k]), k
    for k in res[True][0]:
        assert res[True][0][k] == pytest.approx(res[False][0][k], rel=2e-6, abs=1e-8), k
    # the whole (clipped) gradient arena; parameters after an RMSprop step are NOT compared: g/(sqrt(v)+eps) turns
    # 1e-9 noise on a near-zero gradient into a visible step
    ga, gb = res[True][1], res[False][1]
    close("gradient arena", ga, gb, 2e-6 * float(gb.abs().max()), 1e-5)


def test_first_layer_weight_gradient_from_the_single_frame_store():
    """row f4: the persistent rollout kernel keeps ONE uint8 frame per env-step (T+4 per slot) + the count of real
    planes per state; update_model's conv1 weight gradient stacks the 4 frames on load.  Three rollout+update rounds
    (history copy between slots, resets inside and across slots) against the same run reading the fp32 states."""
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    B, T, A, ss = 5, 6, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=50, done_period=3 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-5)
    res = {}
    for frames in (True, False):
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        pool = _pool(U8FakeEnv, ekws, 2)
        r = Runner(D, dict(hyps, frame_store=frames), None, None, None, env_pool=pool, ingest="zero-copy")
        upd = Updater(net, hyps)
        torch.manual_seed(11)
        out = []
        try:
            for k in range(3):
                r.rollout(net, list(range(B)), hyps)
                r.finish()
                assert (net._stash_frames is not None) == frames
                if frames:     # the store reproduces the fp32 states exactly: stack-on-load == states
                    F_, nv, _ = net._stash_frames
                    st = D["states"].reshape(B, T, 4, -1)
                    for b in range(B):
                        for t in range(T):
                            n = int(nv[b * T + t])
                            want = F_[b, t:t + 4].float()
                            want[:4 - n] = 0
                            assert torch.equal(st[b, t], want), (k, b, t, n)
                upd.update_model(D)
                out.append(net._arena.train_grads().cpu().clone())
            res[frames] = out
        finally:
            r.close()
    for k in range(3):
        ga, gb = res[True][k], res[False][k]
        close(f"gradient arena, round {k}", ga, gb, 2e-6 * float(gb.abs().max()), 1e-5)


@pytest.mark.parametrize("ingest,no_ring", [("zero-copy", False), ("zero-copy", True), ("relay", False)])
def test_device_side_polls_across_the_2_31_step_counter_wrap(ingest, no_ring, monkeypatch):
    """rec carries the step number modulo 2^31 (a2c_hostpool.h): the ring kernel, the persistent per-step kernel
    (A2C_NO_RING=1; also what runs with more envs than CUs) and the relay's a2c_pool_ingest must all compare modulo
    2^31 -- an env that has taken 2^31 - 3 steps keeps being served across the wrap, and the granule value
    0xffffffff'xxxxxxxx (seq = 2^31 - 1 with done = 1) is an answer, not a time-out."""
    from a2c_amd.runner import Runner
    if no_ring:
        monkeypatch.setenv("A2C_NO_RING", "1")
    B, T, A, ss = 3, 4, 3, (4, 84, 84)
    s0 = (1 << 31) - 6
    # done_period 5: env step number 5 (granule seq 2^31 - 1) reports done = 1 -> high word 0xffffffff
    ekws = [dict(env_id=j, rew_period=2 + j, done_period=5) for j in range(B)]
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, env_timeout_s=20.0)
    net = make_net("A3CModel", ss, A, 256)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False, actions_on_host=False)
    us = torch.from_numpy(hashf(3 * T * B, 2231, 0, 1).reshape(3, T, B))
    usd = us.to(DEV)
    rnd = [0]
    pool = _pool(U8FakeEnv, ekws, 2, pong=False, seq_start=s0)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        refs = _oracle_rollouts("A3CModel", onet, hyps, ekws, us, 3, B, T, ss)
        for rnd[0] in range(3):                       # 12 env steps: seq 2^31 - 6 ... 2^31 + 6
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            _compare_round(D, refs[rnd[0]], False)
        assert pool.seq == s0 + 3 * T
    finally:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bits,recurrent,t", [(True, True, 3), (True, False, 0), (False, True, 7)])
def test_ingest_with_post_equals_ingest_then_post(bits, recurrent, t):
    """a2c_pool_ingest_post (the workgroup that fetched env b's answer also does env b's bookkeeping of the env step it belongs to,
    runner.py:208-232, and the hidden row of the next step, runner.py:201,219-221) against a2c_pool_ingest[_bits] followed by
    a2c_rollout_post_frames: every output bit for bit (same arithmetic, one launch instead of two)."""
    from a2c_amd import ops
    B, T, HW, hd, slot0, seq0, k = 37, 8, 7056, 256, 2, 1000, 5
    g = torch.Generator().manual_seed(77 + t)
    rew = torch.where(torch.rand(B, generator=g) < 0.3, torch.randn(B, generator=g), torch.zeros(B))
    dn = (torch.rand(B, generator=g) < 0.3)
    rec = (((seq0 + k) & 0x7fffffff) << 33) | (dn.long() << 32) | rew.view(torch.int32).long().bitwise_and(0xffffffff)
    fbytes = HW // 8 if bits else HW
    fs = (fbytes + 15) // 16 * 16                              # slots of the pinned region are 16-byte multiples
    frames = torch.randint(0, 256, (B, fs), dtype=torch.uint8, generator=g)
    vals = torch.randn(B, 3, generator=g)                       # the value head's output is column 1 of a wider row
    h_src = torch.randn(B, hd, generator=g)

    def run(fused):
        d = lambda x: x.clone().to(DEV)
        recd, frd, seq = d(rec), d(frames), torch.tensor([seq0], dtype=torch.int32, device=DEV)
        err = torch.zeros(1, dtype=torch.int32, device=DEV)
        r_o, d_o = torch.full((B,), -9.0, device=DEV), torch.full((B,), -9.0, device=DEV)
        out = torch.zeros(B, HW + 16, dtype=torch.uint8, device=DEV)
        gen = torch.Generator().manual_seed(5)
        val_prev, rewards, dones, deltas = (d(torch.randn(n, generator=gen)) for n in (B, (slot0 + B) * T, (slot0 + B) * T, (slot0 + B) * T))
        done_eff = torch.full((B,), -9.0, device=DEV)
        h = d(h_src) if recurrent else None
        hs = d(torch.randn(B, hd, generator=gen)) if recurrent else None        # where the previous step left its hidden rows
        h_rows = torch.full((B, T, hd), -9.0, device=DEV) if recurrent else None
        nv_rows = torch.full(((slot0 + B) * T,), -9, dtype=torch.int32, device=DEV)
        nv_carry = d(torch.randint(1, 5, (B,), dtype=torch.int32, generator=gen))
        post = (vals.to(DEV).data_ptr() + 4, 3, val_prev, rewards, dones, deltas, T, t, slot0, 0.99, True, done_eff, h,
                h_rows.data_ptr() + 4 * (t + 1) * hd if (recurrent and t + 1 < T) else 0, T * hd, hs.data_ptr() if recurrent else 0, nv_rows,
                nv_carry.data_ptr())
        v_keep = vals.to(DEV)
        post = (v_keep.data_ptr() + 4,) + post[1:]
        if fused:
            ops.pool_ingest_post(bits, recd.data_ptr(), frd.data_ptr(), fs, HW if bits else fbytes, B, seq, k, 10 ** 8, err,
                                 r_o, d_o, out.data_ptr(), HW + 16, *post)
        else:
            if bits:
                ops.pool_ingest_bits(recd.data_ptr(), frd.data_ptr(), fs, HW, B, seq, k, 10 ** 8, err, r_o, d_o, out.data_ptr(), HW + 16)
            else:
                ops.pool_ingest(recd.data_ptr(), frd.data_ptr(), fs, fbytes, B, seq, k, 10 ** 8, err, r_o, d_o, out.data_ptr(), HW + 16)
            ops.rollout_post_frames(r_o, d_o, *post[:11], B, *post[11:])
        torch.cuda.synchronize()
        assert int(err) == 0
        res = dict(rew=r_o, done=d_o, out=out, val_prev=val_prev, rewards=rewards, dones=dones, deltas=deltas, done_eff=done_eff,
                   nv_rows=nv_rows, nv_carry=nv_carry)
        if recurrent:
            res.update(h=h, h_rows=h_rows)
        return {kk: vv.cpu() for kk, vv in res.items()}

    a, b = run(True), run(False)
    for kk in a:
        assert torch.equal(a[kk].view(torch.uint8) if a[kk].dtype != torch.uint8 else a[kk],
                           b[kk].view(torch.uint8) if b[kk].dtype != torch.uint8 else b[kk]), kk
    assert torch.equal(a["rew"], rew) and torch.equal(a["done"], dn.float())
    e = (slot0 + torch.arange(B)) * T + t
    d_eff = torch.where(rew != 0, torch.ones(B), dn.float())          # Pong: a point ends the episode for the bookkeeping
    assert torch.equal(a["dones"][e], d_eff) and torch.equal(a["rewards"][e], rew) and torch.equal(a["done_eff"], d_eff)
    if recurrent:
        assert bool((a["h"][d_eff != 0] == 0).all())
        if t + 1 < T:       # (the last step of a slot has no next row)
            assert bool((a["h_rows"][:, t + 1][d_eff != 0] == 0).all()) and torch.equal(a["h_rows"][:, t + 1], a["h"])


def test_ring_kernel_leaves_lane_masks_and_the_update_reads_them(monkeypatch):
    """round 6: beside its a1 / a2 stash rows the ring kernel writes their MASK BITS (include/a2c_mi355x.h: a1_lanemask_rows,
    a2_maskbit_rows); the update's conv2 backward-data takes 800 B per sample as its ReLU mask instead of the 25.6 KB
    activation row (a2c_conv2d_bwd_data_lanemask), and da2 = (dl . Wc) * (a2 > 0) 324 B instead of 10.4 KB
    (a2c_small_n_bwd_data_bits).  64 envs x 32 steps (2,048 rows: the streaming kernel's smallest batch): the masks the
    kernel wrote == a2c_lanemask_from_act of the stashed activations, and the update's gradient arena and infos are
    bit-identical to the same update reading the float mask (A2C_NO_LANEMASK=1)."""
    from a2c_amd import ops
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    from a2c_amd.updater import Updater
    B, T, A, ss = 64, 32, 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-5)
    monkeypatch.setenv("A2C_BWD_X6", "0")        # both updates on the fp32 MFMA kernels: what is compared is the mask plumbing
    res = {}
    for mode in ("lanemask", "float"):
        if mode == "float":
            monkeypatch.setenv("A2C_NO_LANEMASK", "1")
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 20) for j in range(B)]
        pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True, frame_bits=True)
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy")
        torch.manual_seed(11)
        try:
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            assert net._stash is not None and bool(getattr(net, "_stash_lm", False)) == (mode == "lanemask")
            if mode == "lanemask":
                ws = net.ws("train")
                a1 = ws.get("a1", (B * T,) + net._c1.out_shape)
                lm = ws.get("a1_lm", (B * T, 100), dtype=torch.int64)
                want = torch.zeros_like(lm)
                ops.lanemask_from_act(a1, want)
                torch.cuda.synchronize()
                assert torch.equal(lm, want) and int((lm != 0).sum()) > 0
                a2 = ws.get("a2", (B * T,) + net._c2.out_shape)
                mb = ws.get("a2_mb", (B * T, net.flat_size // 8), dtype=torch.uint8)
                bits = np.packbits((a2.reshape(B * T, -1) > 0).cpu().numpy().astype(np.uint8), axis=1, bitorder="little")
                assert np.array_equal(mb.cpu().numpy(), bits) and int(bits.sum()) > 0
            info = Updater(net, hyps).update_model(D)
            res[mode] = (info, net._arena.train_grads().cpu().clone())
        finally:
            r.close()
    assert res["lanemask"][0] == res["float"][0]
    assert torch.equal(res["lanemask"][1], res["float"][1])
