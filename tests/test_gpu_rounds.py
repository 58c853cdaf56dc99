"""-m gpu: what round 3 added on the host-ingest side, against the CPU oracle:
  * the PACKED transport (one bit per pixel for binary preprocessors, include/a2c_hostpool.h A2C_FRAME_BITS) through
    every ingest mode -- the kernels must see exactly the uint8 {0,1} planes of the uint8 transport;
  * fp32 frames whose size is not a multiple of 16 B (padded pool slots, dense kernel rows);
  * any number of slots per rollout: n_rollouts that is not a multiple of n_envs is played in lock-step rounds with
    envs falling out of step with each other (the reference's shipped hyperparams.json: 45 slots on 11 envs);
  * train() on that key set, two epochs, weights and losses against oracle SlotRunner + OracleUpdater."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import PongLikeEnv, U8FakeEnv, base_hyps, hashf  # noqa: E402
from test_gpu_kernels import close  # noqa: E402
from test_gpu_models import _datas, make_net  # noqa: E402
from test_gpu_ingest import _compare_round, _oracle_rollouts, _pool  # noqa: E402

DEV = "cuda"


# ---------------------------------------------------------------------------------------------- packed transport
@pytest.mark.parametrize("kind,ingest", [("A3CModel", "zero-copy"), ("A3CModel", "relay"), ("A3CModel", "memcpy"),
                                         ("GRUModel", "relay"), ("ConvModel", "memcpy"), ("FCModel", "relay")])
def test_packed_bits_rollouts_match_oracle(kind, ingest):
    """binary uint8 envs behind a frame_bits pool: two consecutive rounds, every buffer against the oracle"""
    from a2c_amd.runner import Runner
    B, T, A, ss = 5, 6, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0" if kind == "A3CModel" else "FakeBreakout", n_tsteps=T, n_rollouts=B,
                     action_shift=0, n_envs=B, env_timeout_s=20.0)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
    us = torch.from_numpy(hashf(2 * T * B, 4242, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd = [0]
    pool = _pool(U8FakeEnv, ekws, 2, pong="Pong" in hyps["env_type"], frame_bits=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        refs = _oracle_rollouts(kind, onet, hyps, ekws, us, 2, B, T, ss)
        for rnd[0] in range(2):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            _compare_round(D, refs[rnd[0]], net.is_recurrent)
        assert pool.transport == "bits" and pool.header.frame_bytes == 882 and r.bits and pool.seq == 2 * T
        if ingest == "zero-copy":
            assert r._zero_copy_ok(net)
    finally:
        r.close()


def test_packed_bits_equals_uint8_transport_more_envs_than_cus():
    """native env threads, 300 envs x 4 steps, persistent kernel: the packed and the uint8 transport leave bit-identical
    rollout buffers (states, actions, rewards, dones, deltas)"""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    B, T, A, ss = 300, 4, 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    usd = torch.from_numpy(hashf(2 * T * B, 99, 0, 1).reshape(2, T, B)).to(DEV)
    out = {}
    for bits in (False, True):
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        envs = [TapeEnv(env_id=j, length=2 * T + 1, p_done=0.1) for j in range(B)]
        pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True, frame_bits=bits)
        rnd = [0]
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy",
                   uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
        try:
            res = []
            for rnd[0] in range(2):
                r.rollout(net, list(range(B)), hyps)
                r.finish()
                res.append({k: v.cpu().clone() for k, v in D.items()})
            out[bits] = res
        finally:
            r.close()
    for k in range(2):
        for name in ("states", "actions", "rewards", "dones", "deltas"):
            assert torch.equal(out[True][k][name], out[False][k][name]), (k, name)
    assert float(out[True][1]["states"].sum()) > 0


# ---------------------------------------------------------------------------------------------- padded fp32 slots
@pytest.mark.parametrize("ingest", ["memcpy", "relay"])
def test_fp32_vector_frames_with_padded_pool_slots(ingest):
    """vector observations of 3 floats (12 B): the pool's slots are 16 B apart, the fp32 frame-stack kernels read
    dense rows -- every env b > 0 would read misplaced frames without the compaction"""
    from a2c_amd.runner import Runner
    B, T, A, L = 5, 7, 2, 3
    ss = (4, L)
    ekws = [dict(env_id=j, frame_shape=(1, L), rew_period=3, done_period=4 + j, binary=False) for j in range(B)]
    hyps = base_hyps(env_type="FakeCart", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, h_size=64)
    net = make_net("FCModel", ss, A, 64)
    onet = O.OracleNet("FCModel", ss, A, 64)
    D = _datas(B * T, ss, False, actions_on_host=False)
    us = torch.from_numpy(hashf(2 * T * B, 31, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd = [0]
    pool = _pool(O.FakeEnv, ekws, 2)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        refs = _oracle_rollouts("FCModel", onet, hyps, ekws, us, 2, B, T, ss)
        for rnd[0] in range(2):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            _compare_round(D, refs[rnd[0]], False)
        assert pool.frame_dtype == np.float32 and r.fstride == 16 and r.d_dense is not None
    finally:
        r.close()


# ---------------------------------------------------------------------------------------------- lock-step rounds
class _EnvCounterUniforms:
    """uniform_fn of a Runner whose envs fall out of step: env e's n-th sampled action uses table[e, n], whatever slot
    and round it is played in (the oracle's per-env SlotRunner draws from the same row in the same order)"""

    def __init__(self, table):
        self.table = table                      # (n_envs, K) cpu float32
        self.dev = table.to(DEV)
        self.cnt = np.zeros(table.shape[0], dtype=np.int64)

    def __call__(self, t, B, env0):
        idx = torch.from_numpy(self.cnt[env0:env0 + B].copy()).to(DEV)
        u = self.dev[env0:env0 + B].gather(1, idx[:, None]).reshape(B).contiguous()
        self.cnt[env0:env0 + B] += 1
        return u


def _oracle_rounds(kind, onet, hyps, envs, table, R, T, ss, n_epochs, updater=None):
    """epoch: the sorted slot list 0..R-1 in rounds of len(envs); slot k of a round is played by env k"""
    B, N = len(envs), R * T
    Do = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N), dones=torch.zeros(N),
              actions=torch.zeros(N).long())
    if onet.is_recurrent:
        Do["h_states"] = torch.zeros(N, onet.h_size)
    runners = []
    for j, env in enumerate(envs):
        it = iter(table[j].tolist())
        sr = O.SlotRunner(env, Do, hyps, uniform_fn=lambda it=it: float(next(it)))
        sr.start(onet)
        runners.append(sr)
    outs = []
    for ep in range(n_epochs):
        for slot in range(R):
            runners[slot % B].rollout(onet, slot)
        outs.append({n: v.clone() for n, v in Do.items()})
        if updater is not None:
            outs[-1]["info"] = updater.update_model(Do)
            outs[-1]["params"] = [p.detach().clone() for p in onet.parameters()]
    return outs


@pytest.mark.parametrize("kind,ingest", [("A3CModel", "zero-copy"), ("A3CModel", "relay"), ("GRUModel", "relay"),
                                         ("ConvModel", "memcpy")])
def test_rollout_rounds_when_n_rollouts_is_not_a_multiple_of_n_envs(kind, ingest):
    """7 slots on 3 envs, two epochs: rounds (0,1,2) (3,4,5) (6); in the second epoch env 0 is one slot ahead of the
    others, so round 1 splits into the blocks [env 0] and [envs 1, 2]"""
    from a2c_amd.runner import Runner
    B, R, T, A, ss = 3, 7, 4, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=2 + j, done_period=5 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=R, action_shift=0, n_envs=B)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    D = _datas(R * T, ss, net.is_recurrent, actions_on_host=False)
    table = torch.from_numpy(hashf(B * 8 * T, 515, 0, 0.999).reshape(B, 8 * T))     # (u = 1.0 is the sampler's -1 fall-through)
    uf = _EnvCounterUniforms(table)
    pool = _pool(U8FakeEnv, ekws, 2, pong=True, frame_bits=(ingest != "memcpy"))
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest, uniform_fn=uf)
    try:
        refs = _oracle_rounds(kind, onet, hyps, [O.FakeEnv(**kw) for kw in ekws], table, R, T, ss, 2)
        for ep in range(2):
            r.rollout(net, list(range(R)), hyps)
            r.finish()
            _compare_round(D, refs[ep], net.is_recurrent)
        assert pool.seq_env.tolist() == [6 * T, 4 * T, 4 * T]
        with pytest.raises(ValueError):
            pool.seq                                   # the envs are out of step with each other
    finally:
        r.close()


# ---------------------------------------------------------------------------------------------- train(), reference key set
# the key set of the reference's shipped training_scripts/hyperparams.json (values that matter to the path unchanged:
# FCModel, 12 steps, 11 envs, 45 rollouts, 3 stacked frames, RMSprop); env_type is a Pong-named fake
REFERENCE_HYPS = dict(exp_name="fcmodel", seed=0, model="FCModel", env_type="FakePong-v0", prep_fxn="pong_prep",
                      optim_type="RMSprop", max_tsteps=4e7, n_tsteps=12, n_test_eps=21, n_envs=11, n_frame_stack=3,
                      n_rollouts=45, n_past_rews=25, h_size=256, grid_size=15, unit_size=4, n_foods=2, lr=0.0001,
                      lr_low=1e-12, lambda_=0.98, gamma=0.99, gamma_high=0.995, val_coef=0.5, entr_coef=0.005,
                      entr_coef_low=0.001, max_norm=0.5, resume=False, render=False, decay_lr=False, decay_entr=False,
                      use_nstep_rets=False, norm_advs=True, use_bnorm=False, use_bptt=False)


def test_train_on_the_reference_hyperparams_matches_oracle(tmp_path):
    """row f1: two epochs of train() with n_envs = 11, n_rollouts = 45 (process env pool, packed transport) against two
    epochs of oracle SlotRunners (env j plays slots j, 11+j, 22+j, 33+j, and env 0 slot 44) + OracleUpdater"""
    from a2c_amd.training import train
    hyps = dict(REFERENCE_HYPS, main_path=str(tmp_path), action_size=3, action_shift=0, frame_bits=True, n_env_workers=3)
    B, R, T, C, A, h = 11, 45, 12, 3, 3, 256
    ss = (C, 80, 80)
    table = torch.from_numpy(hashf(B * 16 * T, 2024, 0, 0.999).reshape(B, 16 * T))   # (u = 1.0 is the sampler's -1 fall-through)
    uf = _EnvCounterUniforms(table)
    got = []

    def on_epoch(epoch, updater, shared):
        got.append((dict(updater.info), {k: v.detach().cpu().clone() for k, v in shared.items()},
                    [p.detach().cpu().clone() for p in updater.net.parameters()]))

    import a2c_amd.models as M
    sd = O.formula_state_dict("FCModel", ss, A, h)
    orig = M.FCModel.__init__

    def patched(self, *a, **k):          # train() builds the net itself: start it from the formula weights
        orig(self, *a, **k)
        self.load_state_dict(sd)
    M.FCModel.__init__ = patched
    try:
        train(None, hyps, verbose=False, env_fn=PongLikeEnv, max_epochs=2, uniform_fn=uf, on_epoch=on_epoch)
    finally:
        M.FCModel.__init__ = orig
    assert len(got) == 2
    ohyps = base_hyps(**{k: v for k, v in hyps.items() if k in ("gamma", "lambda_", "n_tsteps", "n_rollouts", "n_frame_stack",
                                                                "val_coef", "entr_coef", "max_norm", "lr", "optim_type",
                                                                "norm_advs", "use_nstep_rets", "use_bptt", "env_type")},
                      action_shift=0, pi_coef=1.0)
    onet = O.OracleNet("FCModel", ss, A, h)
    envs = []
    for j in range(B):
        e = PongLikeEnv(j)
        envs.append(e)
    refs = _oracle_rounds("FCModel", onet, ohyps, envs, table, R, T, ss, 2, updater=O.OracleUpdater(onet, ohyps))
    for ep in range(2):
        info, D, params = got[ep]
        ref = refs[ep]
        tol = 1e-5 if ep == 0 else 2e-4          # epoch 2 plays with weights that went through one RMSprop step
        assert torch.equal(D["states"], ref["states"])
        assert torch.equal(D["dones"], ref["dones"])
        mism = int((D["actions"] != ref["actions"]).sum())
        assert mism == 0 if ep == 0 else mism <= 1, mism
        close("deltas", D["deltas"], ref["deltas"], tol, tol)
        for k in ref["info"]:
            # (GradNorm: torch's fp32 clip_grad_norm_ over 4.9 M elements is itself ~1e-4 off the fp64 sum the engine takes)
            rel = (2e-4 if k == "GradNorm" else 3e-5) if ep == 0 else 2e-3
            assert abs(info[k] - ref["info"][k]) <= 1e-5 + rel * abs(ref["info"][k]), (ep, k, info[k], ref["info"][k])
        pm = max(float((p - q).abs().max()) for p, q in zip(params, ref["params"]))
        assert pm < (5e-5 if ep == 0 else 3e-4), (ep, pm)
    folder = os.path.join(str(tmp_path), "fcmodel", "fcmodel_0")
    assert sorted(os.listdir(folder)) == ["best_net.p", "log.txt", "net.p", "optim.p"]
    log = open(os.path.join(folder, "log.txt")).read()
    assert f"Step:{R * T}" in log and f"Step:{2 * R * T}" in log


def test_ring_kernel_on_the_tagged_mirror_matches_oracle(monkeypatch):
    """A2C_TAGGED=1 (opt-in, DESIGN.md section 7: measured slower than poll-then-frame): the ring kernel fetches poll + frame +
    record of an env step with ONE 16-byte load per lane from the pool's self-validating mirror and validates the chunk tags
    by ballot; two rounds against the oracle, step counter crossing 65536 (the tags carry its low 16 bits)."""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    monkeypatch.setenv("A2C_TAGGED", "1")
    B, T, A, ss = 5, 7, 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    net = make_net("A3CModel", ss, A, 256)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False, actions_on_host=False)
    kws = [dict(env_id=j, length=2 * T + 1, p_done=0.15) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=2, pong=True, frame_bits=True, seq_start=65536 - 5)
    us = torch.from_numpy(hashf(2 * T * B, 3391, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd = [0]
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy",
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        assert pool.start().header.off_tagged != 0
        Do = dict(states=torch.zeros(B * T, *ss), deltas=torch.zeros(B * T), rewards=torch.zeros(B * T), dones=torch.zeros(B * T),
                  actions=torch.zeros(B * T).long())
        runners = []
        for j in range(B):
            seq = iter([float(us[k, t, j]) for k in range(2) for t in range(T)])
            sr = O.SlotRunner(TapeEnv(**kws[j]), Do, hyps, uniform_fn=lambda seq=seq: next(seq))
            sr.start(onet)
            runners.append(sr)
        for rnd[0] in range(2):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            for j in range(B):
                runners[j].rollout(onet, j)
            assert torch.equal(D["actions"].cpu(), Do["actions"]) and torch.equal(D["states"].cpu(), Do["states"])
            assert torch.equal(D["dones"].cpu(), Do["dones"])
            close("rewards", D["rewards"], Do["rewards"], 1e-5, 1e-5)
            close("deltas", D["deltas"], Do["deltas"], 1e-5, 1e-5)
    finally:
        r.close()


@pytest.mark.parametrize("kind,ingest,bits", [("A3CModel", "zero-copy", True), ("A3CModel", "zero-copy", False), ("GRUModel", "relay", True)])
def test_push_mirror_rollouts_match_oracle(kind, ingest, bits, monkeypatch):
    """The push mirror (default; A2C_PUSH=0 switches it off): the env worker threads also write every answer -- frame,
    sfence, rec granule, sfence -- straight into fine-grained DEVICE memory (a2c_push_buffer_alloc), and the ring kernel /
    the relay's ingest kernel poll and fetch there instead of over PCIe.  Two rounds against the oracle."""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    monkeypatch.setenv("A2C_PUSH", "1")
    B, T, A, ss = 5, 7, 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
    kws = [dict(env_id=j, length=2 * T + 1, p_done=0.15) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=2, pong=True, frame_bits=bits)
    us = torch.from_numpy(hashf(2 * T * B, 3392, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd = [0]
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        assert pool.start().push_ptr != 0 and pool.dev_rec == pool.push_ptr
        refs = _oracle_rollouts(kind, onet, hyps, kws, us, 2, B, T, ss, env_cls=TapeEnv)
        for rnd[0] in range(2):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            _compare_round(D, refs[rnd[0]], net.is_recurrent)
    finally:
        r.close()


# ---------------------------------------------------------------------------------------------- grey levels 0..255
# Round 5 moved conv1 of the ring kernel onto the bf16 matrix pipe (step.hip: every fp32 weight as three bf16 pieces, the
# uint8 pixel as ONE bf16 value).  The exactness argument holds for any uint8 pixel; breakout_prep (preprocessing.py:19-23)
# hands on grey levels 0..255, runner.py:199 casts them to float unchanged.  These tests put pixels above 1 in front of the
# oracle on that kernel: through the uint8 transport (the packed one refuses such frames), both head widths.
def _close_rel(name, got, want, rel, floor=0.0):
    """|got - want| <= floor + rel * (max |want| + |want|)"""
    close(name, got, want, floor + rel * float(want.abs().max()), rel)


def _fp32_value_noise(kind, ss, A, h, onet, states):
    """max |V_fp32 - V_fp64| of the ORACLE on these states.  Grey levels make conv1's sums ~1e2 times larger than binary
    frames do while the formula weights cancel them back to O(1) values: the reference's own fp32 forward is then 6e-6 off
    its fp64 evaluation on values of scale 0.5 (binary frames: 4e-8).  A 1e-5 comparison of two fp32 evaluations would test
    that noise; the tests allow 1e-5 of the scale + 4 x this measured noise (a delta holds two values)."""
    o64 = O.OracleNet(kind, ss, A, h, state_dict={k: v.double() for k, v in O.formula_state_dict(kind, ss, A, h).items()})
    with torch.no_grad():
        return float((onet(states)[0].double() - o64(states.double())[0]).abs().max())


@pytest.mark.parametrize("A,store", [(3, False), (4, False), (4, True)], ids=["A3", "A4", "A4-store"])
def test_ring_kernel_on_grey_uint8_frames_matches_oracle(A, store, monkeypatch):
    """A3CModel, zero-copy ring kernel (bf16-pipe conv1, the default), env frames = uint8 grey levels 0..255 over the uint8
    transport; three rounds with an update between them against O.SlotRunner + OracleUpdater: states / dones / actions
    exact, rewards (bootstrap = gamma * V) and deltas at 1e-5 of their scale on the first round.  "store": the single-frame
    uint8 store + lazy states (the timed layout).  Then the same first round with A2C_RING_F32=1 (fp32 MFMAs): the two
    forms of conv1 agree to fp32 re-association."""
    from cases import GreyFakeEnv, GreyU8FakeEnv
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    B, T, ss = 5, 6, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=2 + j % 3, done_period=4 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-3, env_timeout_s=20.0)
    if store:
        hyps.update(frame_store=True, lazy_states=True)
    us = torch.from_numpy(hashf(3 * T * B, 60606, 0, 1).reshape(3, T, B))
    usd = us.to(DEV)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    refs = _oracle_rollouts("A3CModel", onet, hyps, ekws, us, 3, B, T, ss, env_cls=GreyFakeEnv,
                            updater=O.OracleUpdater(onet, hyps))
    assert float(refs[0]["states"].max()) == 255.0 and float((refs[0]["states"] > 1).float().mean()) > 0.4
    noise = _fp32_value_noise("A3CModel", ss, A, 256, O.OracleNet("A3CModel", ss, A, 256), refs[0]["states"])
    first = {}
    for f32 in (False, True):
        if f32:
            monkeypatch.setenv("A2C_RING_F32", "1")
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        rnd = [0]
        pool = _pool(GreyU8FakeEnv, ekws, 2, pong=False)
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy",
                   uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
        upd = Updater(net, hyps)
        try:
            for rnd[0] in range(1 if f32 else 3):
                r.rollout(net, list(range(B)), hyps)
                r.finish()
                assert r._zero_copy_ok(net) and pool.transport == "u8" and not r.bits
                if store:
                    assert r._states_stale
                    r.materialize_states()
                ref = refs[rnd[0]]
                assert torch.equal(D["states"].cpu(), ref["states"])
                assert torch.equal(D["dones"].cpu(), ref["dones"])
                mism = int((D["actions"].cpu() != ref["actions"]).sum())
                assert mism == 0 if rnd[0] == 0 else mism <= 1, mism
                # later rounds: weights that went through RMSprop steps.  A first step moves every weight by ~10 lr whatever
                # |g| (g / sqrt(0.01 g^2)), so the SIGN noise of near-zero gradients becomes +-10 lr per weight on both
                # sides, and with grey inputs (pixels ~127) the values after an update are dominated by those steps: they
                # grow from O(1) to O(1e3) and agree to 5e-3 (measured) -- stale weights would be off by O(1) of that
                rel = 1e-5 if rnd[0] == 0 else 2e-2
                _close_rel("rewards", D["rewards"].cpu(), ref["rewards"], rel, 4 * noise)
                _close_rel("deltas", D["deltas"].cpu(), ref["deltas"], rel, 4 * noise)
                if rnd[0] == 0:
                    e = float((D["deltas"].cpu() - ref["deltas"]).abs().max())
                    print(f"[grey ring A={A} store={store} f32={f32}] max |delta - oracle| {e:.3e}, scale "
                          f"{float(ref['deltas'].abs().max()):.3g}, oracle fp32-vs-fp64 value noise {noise:.3e}")
                if rnd[0] == 0:
                    first[f32] = {k: v.cpu().clone() for k, v in D.items()}
                if rnd[0] < 2 and not f32:
                    info = upd.update_model(D)
                    oi = ref["info"]
                    for k in oi:
                        assert abs(info[k] - oi[k]) <= 1e-4 + (2e-3 if rnd[0] == 0 else 2e-2) * abs(oi[k]), (rnd[0], k, info[k], oi[k])
        finally:
            r.close()
    for k in ("states", "actions", "dones"):
        assert torch.equal(first[True][k], first[False][k]), k
    _close_rel("deltas bf16-pipe vs fp32 MFMA", first[False]["deltas"], first[True]["deltas"], 2e-6, 4 * noise)


def test_full_size_headline_on_grey_frames_action_census():
    """256 envs x 128 steps of grey-level (0..255) tapes through the headline path on the uint8 transport and the timed
    layout (ring kernel, single-frame store, lazy states): the tapes come back as the states' newest planes, the frame-stack
    property holds, and the oracle forward on 2,048 of the 32,768 states with the same uniforms samples the same actions
    (a flip only where the fp32 cumsum is within 1e-6 of the uniform, at most 1e-4 of the samples); the values behind the
    recorded deltas against the oracle's on those states at 1e-5 of their scale."""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    B, T, A, ss = 256, 128, 4, (4, 84, 84)
    hyps = base_hyps(env_type="Breakout-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, frame_store=True,
                     lazy_states=True)
    net = make_net("A3CModel", ss, A, 256)
    onet = O.OracleNet("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False, actions_on_host=False)
    envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100, grey=True) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=False, frame_bits=False)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy")
    try:
        r.rollout(net, list(range(B)), hyps)
        r.finish()
        assert r._states_stale and r._fstore is not None
        r.materialize_states()
        u = r._u_keep.cpu()
        st = D["states"].cpu().reshape(B, T, 4, -1)
        acts = D["actions"].cpu().reshape(B, T)
        dones = D["dones"].cpu().reshape(B, T)
        assert float(st.max()) == 255.0 and bool((dones[:, -1] == 1).all())
        assert int(acts.min()) >= 0 and int(acts.max()) <= A - 1
        for j in (0, 17, 255):
            for t in range(1, T):
                assert np.array_equal(st[j, t, 3].numpy(), envs[j].frames[t % (T + 1)].reshape(-1).astype(np.float32)), (j, t)
        real_done = torch.from_numpy(np.stack([e.dones[:T] for e in envs]).astype(np.float32))
        for c in range(3):
            same = (st[:, 1:, c] == st[:, :-1, c + 1]).all(-1)
            zero = (st[:, 1:, c] == 0).all(-1)
            assert bool((same | (zero & (real_done[:, :T - 1] == 1))).all())
        idx = (np.arange(2048, dtype=np.int64) * 2654435761 % (B * T)).astype(np.int64)
        with torch.no_grad():
            vals, logits = onet(D["states"].cpu().reshape(B * T, *ss)[idx])
        cs = torch.cumsum(torch.softmax(logits, -1), -1)
        uu = u.t().reshape(-1)[idx]
        ref = (cs >= uu[:, None]).float().argmax(-1)
        ref[(cs < uu[:, None]).all(-1)] = A - 1
        got = acts.reshape(-1)[idx]
        flips = (ref != got).nonzero().flatten()
        assert len(flips) <= max(1, int(1e-4 * len(idx))), len(flips)
        for f in flips.tolist():
            assert float((cs[f] - uu[f]).abs().min()) < 1e-6, (f, cs[f], uu[f])
        # values: delta[t] = r[t] + gamma * V[t+1] * (1 - d[t]) - V[t]  (runner.py:231) on 256 sampled interior steps
        base = np.array([e for e in idx.tolist() if e % T != T - 1][:256], dtype=np.int64)
        with torch.no_grad():
            v0 = onet(D["states"].cpu().reshape(B * T, *ss)[base])[0].reshape(-1)
            v1 = onet(D["states"].cpu().reshape(B * T, *ss)[base + 1])[0].reshape(-1)
        rew, dl, dn = D["rewards"].cpu()[base], D["deltas"].cpu()[base], D["dones"].cpu()[base]
        want = rew + hyps["gamma"] * v1 * (1.0 - dn) - v0
        scale = float(torch.maximum(v0.abs(), v1.abs()).max())
        err = float((dl - want).abs().max())
        noise = _fp32_value_noise("A3CModel", ss, A, 256, onet, D["states"].cpu().reshape(B * T, *ss)[base])
        print(f"[grey census] value scale {scale:.4g}, max |delta - oracle| {err:.3e} = {err / scale:.2e} of the scale; "
              f"oracle fp32-vs-fp64 value noise {noise:.3e}")
        assert err <= 1e-5 * scale + 4 * noise, (err, scale, noise)
    finally:
        r.close()
