"""-m gpu: the conv-stack models (ConvModel, GRUModel) at the batch sizes the benchmark's configs 2, 4 and 5 run.

The tile tuners (bwd_band_tuned, conv_fwd_tuned) and the batch-dependent GEMM routes pick kernel instantiations per
(layer, batch): at 256 envs those are NOT the instantiations the 3-6 env parity tests exercise.  Here:
  * rollouts at B = 256, T = 4 through the device relay: every env bit for bit against the run with the tuners off
    (A2C_NO_TUNE=1: the rule-sized tilings), envs {0, 137, 255} against the CPU oracle (runner.py:174-248);
  * the full BASELINE sizes (GRUModel + BPTT 256 x 128, ConvModel 32 x 64) checked through size-independent properties:
    every slot ends done, frame-stack shifts, scans bit-exact on sampled rows, and a census of the sampled actions
    against the oracle forward on 2,048 recorded states (same uniforms; at most 1e-4 of them may flip, and only where the
    fp32 cumsum is within 1e-6 of the uniform)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import U8FakeEnv, base_hyps, hashf  # noqa: E402
from test_gpu_kernels import close  # noqa: E402
from test_gpu_models import _datas, make_net  # noqa: E402
from test_gpu_ingest import _pool  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("kind", ["ConvModel", "GRUModel"])
def test_conv_stack_rollout_at_256_envs(kind, monkeypatch):
    from a2c_amd.runner import Runner
    B, T, A, ss = 256, 4, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=2 + j % 3, done_period=3 + j % 7) for j in range(B)]
    hyps = base_hyps(env_type="FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
    us = torch.from_numpy(hashf(2 * T * B, 20261, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    out = {}
    for tuned in (True, False):
        if not tuned:
            monkeypatch.setenv("A2C_NO_TUNE", "1")
        net = make_net(kind, ss, A, 256)
        D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
        pool = _pool(U8FakeEnv, ekws, 6, frame_bits=tuned)        # (and the packed transport against the uint8 one)
        rnd = [0]
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="relay",
                   uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
        res = []
        try:
            for rnd[0] in range(2):          # round 0 eager (the tuners measure), round 1 = the captured slot graph
                r.rollout(net, list(range(B)), hyps)
                r.finish()
                res.append({k: v.cpu().clone() for k, v in D.items()})
        finally:
            r.close()
        out[tuned] = res
    names = ("states", "actions", "dones", "rewards", "deltas") + (("h_states",) if kind == "GRUModel" else ())
    for k in range(2):
        for n in names:
            assert torch.equal(out[True][k][n], out[False][k][n]), (k, n)
    onet = O.OracleNet(kind, ss, A, 256)
    for j in (0, 137, 255):
        Do = dict(states=torch.zeros(T, *ss), deltas=torch.zeros(T), rewards=torch.zeros(T), dones=torch.zeros(T),
                  actions=torch.zeros(T).long())
        if onet.is_recurrent:
            Do["h_states"] = torch.zeros(T, 256)
        it = iter([float(us[k, t, j]) for k in range(2) for t in range(T)])
        sr = O.SlotRunner(O.FakeEnv(**ekws[j]), Do, hyps, uniform_fn=lambda it=it: next(it))
        sr.start(onet)
        for k in range(2):
            sr.rollout(onet, 0)
            sl = slice(j * T, (j + 1) * T)
            got = out[True][k]
            assert torch.equal(got["actions"][sl], Do["actions"]), (j, k)
            assert torch.equal(got["states"][sl], Do["states"]) and torch.equal(got["dones"][sl], Do["dones"]), (j, k)
            close("rewards", got["rewards"][sl], Do["rewards"], 1e-5, 1e-5)
            close("deltas", got["deltas"][sl], Do["deltas"], 1e-5, 1e-5)
            if onet.is_recurrent:
                close("h_states", got["h_states"][sl], Do["h_states"], 1e-5, 1e-5)


@pytest.mark.parametrize("kind,B,T,bptt", [("GRUModel", 256, 128, True), ("ConvModel", 32, 64, False)])
def test_full_size_conv_stack_rollout_properties_and_action_census(kind, B, T, bptt):
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    from a2c_amd.updater import Updater
    A, ss = 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, use_bptt=bptt)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
    envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True, frame_bits=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="relay")
    try:
        r.rollout(net, list(range(B)), hyps)
        r.finish()
        u = r._u_buf.cpu()                                     # (T, B) uniforms of the slot
        st = D["states"].cpu().reshape(B, T, 4, -1)
        acts = D["actions"].cpu().reshape(B, T)
        dones, rews = D["dones"].cpu().reshape(B, T), D["rewards"].cpu().reshape(B, T)
        assert bool((dones[:, -1] == 1).all())
        assert int(acts.min()) >= 0 and int(acts.max()) <= A - 1
        assert set(np.unique(rews[:, :-1].numpy())) <= {-1.0, 0.0, 1.0}
        for j in (0, B // 2 + 1, B - 1):                       # env data == the tapes
            e = TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100)
            for t in range(1, T):
                assert np.array_equal(st[j, t, 3].numpy(), e.frames[t % (T + 1)].reshape(-1).astype(np.float32)), (j, t)
        real_done = torch.from_numpy(np.stack([e.dones[:T] for e in envs]).astype(np.float32))
        for c in range(3):                                     # frame stack (utils.py:26-43)
            same = (st[:, 1:, c] == st[:, :-1, c + 1]).all(-1)
            zero = (st[:, 1:, c] == 0).all(-1)
            assert bool((same | (zero & (real_done[:, :T - 1] == 1))).all())
        if net.is_recurrent:                                   # hidden state restarts after a done (runner.py:219-220)
            hs = D["h_states"].cpu().reshape(B, T, -1)
            eff = dones[:, :-1] == 1
            assert bool((hs[:, 1:][eff] == 0).all()) and bool((hs[:, 0] == 0).all())
            assert float(hs.abs().max()) > 0
        # action census: oracle forward on 2,048 recorded (state, hidden state) pairs, same uniforms
        idx = (np.arange(2048, dtype=np.int64) * 2654435761 % (B * T)).astype(np.int64)
        xs = D["states"].cpu().reshape(B * T, *ss)[idx]
        with torch.no_grad():
            if net.is_recurrent:
                logits = onet(xs, D["h_states"].cpu()[idx])[1]
            else:
                logits = onet(xs)[1]
        cs = torch.cumsum(torch.softmax(logits, -1), -1)
        uu = u.t().reshape(-1)[idx]                            # sample e = slot*T + t  <->  u[t, slot]
        ref = (cs >= uu[:, None]).float().argmax(-1)
        ref[(cs < uu[:, None]).all(-1)] = A - 1
        got = acts.reshape(-1)[idx]
        flips = (ref != got).nonzero().flatten()
        assert len(flips) <= max(1, int(1e-4 * len(idx))), len(flips)
        for f in flips.tolist():
            assert float((cs[f] - uu[f]).abs().min()) < 1e-6, (f, cs[f], uu[f])
        # the update this config runs (BPTT from the rollout's cell stash for GRUModel): finite, scans bit-exact
        deltas = D["deltas"].cpu().reshape(B, T).numpy()
        rew_np, done_np = rews.numpy(), dones.numpy()
        upd = Updater(net, hyps)
        if bptt:
            assert net._cells_stashed(D["states"], B, T)
        info = upd.update_model(D)
        assert all(np.isfinite(v) for v in info.values()), info
        advs, rets = upd._bufs["advs"].cpu().reshape(B, T).numpy(), upd._bufs["rets"].cpu().reshape(B, T).numpy()
        for j in (0, B // 3, B - 1):
            close("advs", advs[j], O.discount_np(deltas[j], done_np[j], hyps["gamma"] * hyps["lambda_"]), 0, 0)
            close("rets", rets[j], O.discount_np(rew_np[j], done_np[j], hyps["gamma"]), 0, 0)
    finally:
        r.close()


@pytest.mark.parametrize("kind,bptt", [("ConvModel", False), ("GRUModel", True)])
def test_update_with_sign_word_masks_equals_float_masks_bit_for_bit(kind, bptt, monkeypatch):
    """The conv stacks carry the ReLU mask of the stride-2 layers' backward-data as sign words (one bit per activation,
    written by the rollout's forward next to the activation stash).  The same rollout + update with A2C_NO_SIGNS=1 (float
    masks read from the stashed activations) must leave the same weights, bit for bit: masking with a bit or with
    (activation > 0) is the same selection and the kernels keep the same summation order."""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    from a2c_amd.updater import Updater
    B, T, A, ss = 96, 8, 3, (4, 84, 84)
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, use_bptt=bptt)
    us = torch.from_numpy(hashf(2 * T * B, 771, 0, 0.999).reshape(2, T, B)).to(DEV)
    res = {}
    for signs in (True, False):
        if not signs:
            monkeypatch.setenv("A2C_NO_SIGNS", "1")
        net = make_net(kind, ss, A, 256)
        assert bool(net._sign_layers()) == signs
        D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
        envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 20) for j in range(B)]
        pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True, frame_bits=True)
        rnd = [0]
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="relay",
                   uniform_fn=lambda t, Bn, env0: us[rnd[0], t, env0:env0 + Bn].contiguous())
        upd = Updater(net, hyps)
        try:
            infos = []
            for rnd[0] in range(2):
                r.rollout(net, list(range(B)), hyps)
                r.finish()
                infos.append(upd.update_model(D))
        finally:
            r.close()
        res[signs] = ({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, infos)
    for k in res[True][0]:
        assert torch.equal(res[True][0][k], res[False][0][k]), k
    # the reported scalars are fp64 sums of per-workgroup partials added with atomics (loss.hip): equal up to the ORDER of
    # those additions, i.e. to ~1e-16 relative -- the weights above are the bit-for-bit statement
    for a, b in zip(res[True][1], res[False][1]):
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-12, abs=1e-15), k
