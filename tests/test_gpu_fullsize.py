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
    # (GRUModel conv4's backward-data onto its ODD 21 x 21 input has a sign-word kernel only -- its float-mask twin is the
    # generic band kernel, another order of the same sums -- so the bit-for-bit statement is made with that layer on the generic
    # kernel in both runs; a third run with the odd-image kernel is compared at fp32 tolerance below)
    monkeypatch.setenv("A2C_NO_ODD_BS", "1")
    for signs in (True, False, "odd"):
        if signs == "odd":
            if kind != "GRUModel":
                continue
            monkeypatch.delenv("A2C_NO_SIGNS", raising=False)
            monkeypatch.delenv("A2C_NO_ODD_BS", raising=False)
        if signs is False:
            monkeypatch.setenv("A2C_NO_SIGNS", "1")
        net = make_net(kind, ss, A, 256)
        assert bool(net._sign_layers()) == bool(signs)
        if signs == "odd":
            assert 2 in net._sign_layers()          # conv3's forward leaves the sign words conv4's backward reads
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
    # the reported scalars are fixed-order fp64 reductions (grid_sum_ordered: the last workgroup adds the partials in
    # workgroup order): bit for bit too
    assert res[True][1] == res[False][1]
    if "odd" in res:            # the odd-image kernel: same update up to fp32 re-association of one layer's input gradient
        for u, (a, b) in enumerate(zip(res["odd"][1], res[True][1])):
            for k in a:         # update 0: same weights, same rollout: only GradNorm sees the other kernel; update 1 follows other weights
                tol = (2e-5 if k == "GradNorm" else 1e-9) if u == 0 else 5e-3
                assert a[k] == pytest.approx(b[k], rel=tol, abs=1e-9 if u == 0 else 2e-5), (u, k)
        for k in res[True][0]:
            d = (res["odd"][0][k] - res[True][0][k]).abs()
            # two RMSprop steps of lr = 1e-4 amplify gradient noise to ~lr * 10 per step in the worst element
            assert float(d.max()) <= 2.5e-3 and float(d.mean()) <= 2e-5, (k, float(d.max()), float(d.mean()))


def test_full_size_gru_bptt_update_against_the_chunked_oracle():
    """BASELINE configs[3] -- the model the reference publishes (README.md:12-26): GRUModel + BPTT, 256 envs x 128 steps,
    on the layout bench.py times (relay ingest into the single-frame store, lazy fp32 states, BPTT from the rollout's cell
    stash).  The 37 ms update (c3w / c3bs / bwd_band instances at N = 32,768, 128 BPTT steps) against the oracle's SAME
    update (updater.py:63-169) evaluated in chunks of 16 slots (O.update_grads_chunked: exact, see its docstring) in fp64
    and in fp32:
      * Pi_Loss / ValLoss / Entropy / Loss at rel 3e-5 of the fp64 evaluation;
      * GradNorm against the fp64 norm at 2e-5;
      * EVERY gradient tensor (conv1..conv5, resize_emb, the GRU's W_x / W_h / b, both heads): rms deviation from the fp64
        gradient <= 4 x the deviation of torch's own fp32 evaluation + 5e-5 of the tensor's rms (the bar of the A3C / ConvModel
        full-size tests);
      * the update's recorded forward (heads rows of the BPTT buffers) on 512 (state, h) rows against the oracle's bptt."""
    import os
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    from a2c_amd.updater import Updater
    threads = max(4, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(threads)
    torch.manual_seed(20260107)
    kind, B, T, A, ss, h = "GRUModel", 256, 128, 3, (4, 84, 84), 256
    hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, use_bptt=True,
                     frame_store=True, lazy_states=True)
    net = make_net(kind, ss, A, h)
    D = _datas(B * T, ss, True, actions_on_host=False)
    envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True, frame_bits=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="relay")
    try:
        r.rollout(net, list(range(B)), hyps)
        r.finish()
        assert r._states_stale and net._cells_stashed(D["states"], B, T)
        upd = Updater(net, hyps)
        info = upd.update_model(D)
        assert r._states_stale                       # the update ran from the store + the stashes
        r.materialize_states()
        torch.cuda.synchronize()
        Do = {k: v.cpu().clone() for k, v in D.items()}
        hip_g = {n: net.G(n).cpu().double() for n, _ in net.named_parameters() if n not in net._unused_params}
    finally:
        r.close()
    coef = 1.0
    if info["GradNorm"] > hyps["max_norm"]:          # the optimiser kernel wrote the clipped gradients back (like torch)
        coef = hyps["max_norm"] / (info["GradNorm"] + 1e-6)
    o64 = O.OracleNet(kind, ss, A, h, state_dict={k: v.double() for k, v in O.formula_state_dict(kind, ss, A, h).items()})
    i64, g64 = O.update_grads_chunked(o64, Do, hyps, 16, dtype=torch.float64)
    i32, g32 = O.update_grads_chunked(O.OracleNet(kind, ss, A, h), Do, hyps, 16, dtype=torch.float32)
    for k in ("Loss", "Pi_Loss", "ValLoss", "Entropy"):
        assert info[k] == pytest.approx(i64[k], rel=3e-5, abs=2e-6), (k, info[k], i64[k], i32[k])
    assert info["GradNorm"] == pytest.approx(i64["GradNorm"], rel=2e-5), (info["GradNorm"], i64["GradNorm"], i32["GradNorm"])
    worst = {}
    for n, gx in g64.items():
        if gx is None:
            assert n in net._unused_params, n
            continue
        gh, gr = hip_g[n] / coef, g32[n].double()
        rms = float(gx.pow(2).mean().sqrt())
        e_hip, e_ref = float((gh - gx).pow(2).mean().sqrt()), float((gr - gx).pow(2).mean().sqrt())
        worst[n] = (e_hip / rms, e_ref / rms)
        assert e_hip <= 4 * e_ref + 5e-5 * rms, (n, e_hip / rms, e_ref / rms)
        assert float((gh - gx).abs().max()) <= 4 * float((gr - gx).abs().max()) + 5e-3 * rms, n
    print("rms(HIP - fp64) / rms, rms(torch fp32 - fp64) / rms per tensor:", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in worst.items()})
    # the forward the loss saw ([logits | value] rows the rollout's cell stash left for the update) vs the oracle's bptt on 4 whole slots
    slots = [0, 85, 170, 255]
    sel = torch.cat([torch.arange(s * T, (s + 1) * T) for s in slots])
    with torch.no_grad():
        ov, ol = O.bptt(O.OracleNet(kind, ss, A, h), Do["states"][sel], Do["h_states"][sel], Do["dones"][sel],
                        dict(hyps, n_rollouts=len(slots)))
    H = net._heads("train", B * T)[0].cpu()
    rows = H[sel]                                    # rollout-major: row slot * T + t
    close("logits at N=32768", rows[:, :A], ol, 2e-5, 1e-5)
    close("values at N=32768", rows[:, A], ov.reshape(-1), 2e-5, 1e-5)
    torch.set_num_threads(4)
