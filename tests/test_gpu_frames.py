"""-m gpu: row f4 of SURVEY.md section 8 -- preprocessing on the device and the single-frame uint8 store.

  * a2c_frame_prep_u8 (pong_prep / breakout_prep, preprocessing.py:11-23) bit-exact against the REFERENCE's recorded
    outputs (tests/golden/g9_preprocessing.npz) and against the oracle on random Atari-shaped frames;
  * a2c_frames_to_states against numpy (window + valid-plane count -> the reference's fp32 state, utils.py:26-43);
  * first-layer forward / weight gradient of the 3x3 stacks stacked ON LOAD from the store: bit-identical to the same
    kernels fed with the materialised fp32 states (and those are tested against torch in test_gpu_kernels.py);
  * end to end: relay rollouts of ConvModel / GRUModel (+BPTT) with hyps['frame_store'] -- states never written by the
    rollout -- against the plain path bit for bit, and update_model from the store against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import PREP_SEEDS, U8FakeEnv, atari_frame, base_hyps, hashf  # noqa: E402
from test_gpu_kernels import _ops, _sign_words, close, rnd  # noqa: E402

DEV = "cuda"


def test_device_prep_matches_the_reference_recorded_outputs(golden):
    ops = _ops()
    g = golden["g9_preprocessing"]
    raw = torch.from_numpy(np.stack([atari_frame(s) for s in PREP_SEEDS])).to(DEV)          # (n, 210, 160, 3) uint8
    n = raw.shape[0]
    for name, key in (("pong_prep", "pong"), ("breakout_prep", "breakout")):
        _, oh, ow = ops.prep_out_shape(name, 210, 160)
        out = torch.full((n, oh * ow + 16), 77, dtype=torch.uint8, device=DEV)              # rows of a wider buffer
        ops.frame_prep_u8(name, raw.data_ptr(), raw[0].numel(), 210, 160, 3, out.data_ptr(), out.stride(0), n)
        for i, s in enumerate(PREP_SEEDS):
            want = g[f"{key}{s}"]
            assert want.shape == (1, oh, ow)
            assert np.array_equal(out[i, :oh * ow].cpu().numpy().reshape(oh, ow), want[0]), (name, s)
        assert bool((out[:, oh * ow:] == 77).all())


def test_device_prep_random_frames_vs_oracle():
    ops = _ops()
    rng = np.random.default_rng(5)
    fr = rng.integers(0, 256, size=(37, 210, 160, 3), dtype=np.uint8)
    fr[:, ::3, ::5, 0] = 144
    fr[:, 1::3, 2::7, 0] = 109
    fr[:, 2::9, :, 0] = 0
    raw = torch.from_numpy(fr).to(DEV)
    for name, fn in (("pong_prep", O.pong_prep), ("breakout_prep", O.breakout_prep)):
        _, oh, ow = ops.prep_out_shape(name, 210, 160)
        out = torch.zeros((37, oh * ow), dtype=torch.uint8, device=DEV)
        ops.frame_prep_u8(name, raw.data_ptr(), raw[0].numel(), 210, 160, 3, out.data_ptr(), out.stride(0), 37)
        want = np.stack([fn(f.copy())[0] for f in fr])
        assert np.array_equal(out.cpu().numpy().reshape(37, oh, ow), want), name


def _store(R, T, HW, seed):
    """random frame store (R, T+4, HW) uint8 + valid-plane counts (R*T,) -> and the fp32 states they stand for"""
    rng = np.random.default_rng(seed)
    Fs = rng.integers(0, 3, size=(R, T + 4, HW), dtype=np.uint8)
    nv = rng.integers(1, 5, size=(R * T,), dtype=np.int32)
    st = np.zeros((R * T, 4, HW), np.float32)
    for r in range(R):
        for t in range(T):
            for c in range(4):
                if c >= 4 - nv[r * T + t]:
                    st[r * T + t, c] = Fs[r, t + c]
    return Fs, nv, st


def test_frames_to_states_and_store_begin():
    ops = _ops()
    R, T, HW = 5, 6, 84 * 84
    Fs, nv, st = _store(R, T, HW, 1)
    Fd, nvd = torch.from_numpy(Fs).to(DEV), torch.from_numpy(nv).to(DEV)
    out = torch.full((R * T, 4, HW), float("nan"), device=DEV)
    ops.frames_to_states(Fd.data_ptr(), Fd.stride(0), nvd.data_ptr(), T, out.data_ptr(), T * 4 * HW, R, T, 4, HW)
    assert np.array_equal(out.cpu().numpy(), st)
    # one time step of every slot (what a rollout materialises per env step): state t = 3
    row = torch.full((R, 4, HW), float("nan"), device=DEV)
    ops.frames_to_states(Fd.data_ptr() + 3 * HW, Fd.stride(0), nvd.data_ptr() + 4 * 3, T, row.data_ptr(), 4 * HW, R, 1, 4, HW)
    assert np.array_equal(row.cpu().numpy(), st.reshape(R, T, 4, HW)[:, 3])
    # slot start: the window's tail moves to its head, state 0 gets the carried count
    carry = torch.tensor([1, 4, 2, 3, 4], dtype=torch.int32, device=DEV)
    ops.frame_store_begin(Fd.data_ptr(), Fd.stride(0), T, 4, HW, nvd.data_ptr(), carry.data_ptr(), R)
    F2 = Fd.cpu().numpy()
    assert np.array_equal(F2[:, :4], Fs[:, T:T + 4]) and np.array_equal(F2[:, 4:], Fs[:, 4:])
    assert np.array_equal(nvd.cpu().numpy().reshape(R, T)[:, 0], carry.cpu().numpy())
    assert np.array_equal(nvd.cpu().numpy().reshape(R, T)[:, 1:], nv.reshape(R, T)[:, 1:])


@pytest.mark.parametrize("B,signs", [(3, False), (40, True), (300, False), (300, True)])
def test_first_layer_forward_stacked_on_load_equals_forward_from_states(B, signs):
    """c3_kernel (B <= 64, no signs) / c3s_kernel: the loader waves expand the uint8 window instead of LDS-DMA of the fp32 planes"""
    ops = _ops()
    d = ops.conv_desc(4, 84, 84, 16, 3, 1, 1)
    assert ops.conv_fwd_frames_supported(d)
    T, HW = 4, 84 * 84
    R = (B + T - 1) // T
    Fs, nv, st = _store(R, T, HW, 2 + B)
    Fd, nvd, xd = torch.from_numpy(Fs).to(DEV), torch.from_numpy(nv).to(DEV), torch.from_numpy(st).to(DEV)
    w = (rnd((16, 4, 3, 3), 11) / 6.0).to(DEV)
    bias = (rnd((16,), 12) * 0.1).to(DEV)
    wf = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    ops.conv_prep(d, 0, w, wf)
    nsw = ops.conv_sign_words(d)
    ref = torch.empty(B, 16, 84, 84, device=DEV)
    sref = torch.zeros((B, nsw), dtype=torch.int32, device=DEV)
    if signs:
        ops.conv_fwd_signs(d, xd.data_ptr(), 4 * HW, wf, bias, True, ref, sref.data_ptr(), nsw, B)
    else:
        ops.conv_fwd(d, xd.data_ptr(), 4 * HW, wf, bias, True, ref, B)
    close("fwd vs torch", ref[:3], F.relu(F.conv2d(torch.from_numpy(st[:3]).reshape(3, 4, 84, 84), w.cpu(), bias.cpu(), padding=1)),
          2e-6, 1e-5)
    # (a) update-style addressing: sample n = slot * T + t
    out = torch.full_like(ref, float("nan"))
    sg = torch.zeros((B, nsw), dtype=torch.int32, device=DEV)
    ops.conv_fwd_frames(d, Fd.data_ptr(), Fd.stride(0), T, nvd.data_ptr(), 1, wf, bias, True, out, B,
                        signs=(sg.data_ptr(), nsw) if signs else None)
    assert torch.equal(out, ref)
    if signs:
        assert torch.equal(sg, sref) and torch.equal(sg.cpu(), _sign_words(ref.cpu() > 0))
    # (b) rollout-style addressing: time step t of slots 0 .. R-1 (sample stride = slot stride, counts T apart)
    t = 2
    Rb = min(R, B)
    out2 = torch.full((Rb, 16, 84, 84), float("nan"), device=DEV)
    ops.conv_fwd_frames(d, Fd.data_ptr() + t * HW, Fd.stride(0), 1, nvd.data_ptr() + 4 * t, T, wf, bias, True, out2, Rb)
    idx = torch.arange(Rb, device=DEV) * T + t
    keep = idx < B
    assert torch.equal(out2[keep], ref[idx[keep]])


@pytest.mark.parametrize("B", [5, 700])
@pytest.mark.parametrize("old_instance", [False, True])
def test_first_layer_weight_gradient_stacked_on_load_equals_gradient_from_states(B, old_instance, monkeypatch):
    """both sources run the SAME kernel instance (the default: short bands, ring of three chunk images, the uint8 bytes of the
    chunk after next in registers; A2C_C3W_D2=1: round 3's tall-band instance, one chunk of lookahead): bit-identical; and
    the two instances -- two partitions of the same sums -- agree to fp32 re-association"""
    ops = _ops()
    if old_instance:
        monkeypatch.setenv("A2C_C3W_D2", "1")
    else:
        monkeypatch.delenv("A2C_C3W_D2", raising=False)
    d = ops.conv_desc(4, 84, 84, 16, 3, 1, 1)
    T, HW = 5, 84 * 84
    R = (B + T - 1) // T
    Fs, nv, st = _store(R, T, HW, 40 + B)
    Fd, nvd, xd = torch.from_numpy(Fs).to(DEV), torch.from_numpy(nv).to(DEV), torch.from_numpy(st).to(DEV)
    dout = rnd((B, 16, 84, 84), 41).to(DEV)
    ws = torch.empty((ops.conv_bwd_weight_ws_bytes(d, B) + 3) // 4, device=DEV)
    dW0, db0 = torch.empty(16, 4, 3, 3, device=DEV), torch.empty(16, device=DEV)
    dW1, db1 = torch.full_like(dW0, float("nan")), torch.full_like(db0, float("nan"))
    ops.conv_bwd_weight(d, xd.data_ptr(), 4 * HW, dout, dW0, db0, B, ws)
    ops.conv_bwd_weight_frames(d, Fd, Fd.stride(0), T, nvd, dout, dW1, db1, B, ws)
    assert torch.equal(dW1, dW0) and torch.equal(db1, db0)
    if old_instance:
        monkeypatch.delenv("A2C_C3W_D2")
    else:
        monkeypatch.setenv("A2C_C3W_D2", "1")
    dW2, db2 = torch.empty_like(dW0), torch.empty_like(db0)
    ops.conv_bwd_weight(d, xd.data_ptr(), 4 * HW, dout, dW2, db2, B, ws)
    close("the other fp32 instance", dW2, dW0, 2e-6 * float(dW0.abs().max()), 1e-5)
    if B == 5:
        x = torch.from_numpy(st[:B]).reshape(B, 4, 84, 84).double().requires_grad_(False)
        wt = torch.zeros(16, 4, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x, wt, padding=1).backward(dout.cpu().double())
        close("dW vs autograd", dW1, wt.grad, 1e-5 * float(wt.grad.abs().max()), 1e-5)


# ------------------------------------------------------------------ end to end: Runner / Updater
from cases import RawAtariEnv  # noqa: E402
from test_gpu_models import _datas, make_net  # noqa: E402
from test_gpu_ingest import _compare_round, _oracle_rollouts, _pool  # noqa: E402


def _run_engine(kind, hyps, ekws, usd, B, T, A, ss, h, n_rounds, env_cls=U8FakeEnv, check=None):
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater
    net = make_net(kind, ss, A, h)
    D = _datas(B * T, ss, net.is_recurrent, h=h, actions_on_host=False)
    rnd = [0]
    pool = _pool(env_cls, ekws, 2, pong="Pong" in hyps["env_type"])
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="relay",
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    upd = Updater(net, hyps)
    outs = []
    try:
        for rnd[0] in range(n_rounds):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            if check is not None:
                check(r, net, D, rnd[0])
            info = upd.update_model(D) if rnd[0] + 1 < n_rounds else None
            if check is not None and info is not None:
                check(r, net, D, -1)
            r.materialize_states()
            outs.append(({k: v.detach().cpu().clone() for k, v in D.items()}, info,
                         [p.detach().cpu().clone() for p in net.parameters()]))
    finally:
        r.close()
    return outs


@pytest.mark.parametrize("kind,bptt,lazy", [("ConvModel", False, False), ("ConvModel", False, True), ("GRUModel", True, True),
                                            ("GRUModel", False, False)])
def test_frame_store_rollouts_and_updates_equal_the_plain_path_bit_for_bit(kind, bptt, lazy, monkeypatch):
    """hyps['frame_store']: the relay rollout of the conv-stack nets keeps ONE uint8 frame per env step (the ingest writes it
    straight into the store), the first conv layer stacks its planes on load, the update's first-layer weight gradient
    reads the store; hyps['lazy_states']: the fp32 `states` rows are not written at all until somebody asks.  Same
    kernels, same values, same summation order => rollout buffers, infos and weights identical to the plain path, over
    three rounds with updates in between (resets inside and across slots); and the plain path is the one the oracle tests pin."""
    B, T, A, ss, h = 5, 6, 3, (4, 84, 84), 256
    ekws = [dict(env_id=j, rew_period=2 + j % 2, done_period=4 + j) for j in range(B)]
    base = dict(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-3, optim_type="RMSprop",
                use_bptt=bptt, h_size=h)
    us = torch.from_numpy(hashf(3 * T * B, 6151, 0, 1).reshape(3, T, B))
    usd = us.to(DEV)
    seen = dict(stale=0, frames=0)

    def check(r, net, D, k):
        assert getattr(r, "_fstore", None) is not None and getattr(r, "_fstore_ok", True)         # the store is live
        if k >= 0:
            assert net._stash_frames is not None
            seen["frames"] += 1
        if lazy:
            assert r._states_stale            # after the rollout AND after the update: nobody needed the fp32 rows
            seen["stale"] += 1
            if k == 0:
                assert float(D["states"].abs().sum()) == 0.0          # really never written

    # (GRUModel: round 6's fused pass -- layer 2's backward-data with the first layer's weight gradient from the frames, on the
    # bf16 pipe -- re-associates that one gradient; bit-for-bit equality with the plain path holds for the UNFUSED launches,
    # the fused pass has its own test below)
    monkeypatch.setenv("A2C_NO_W1_FRAMES", "1")
    plain = _run_engine(kind, base_hyps(**base), ekws, usd, B, T, A, ss, h, 3)
    store = _run_engine(kind, base_hyps(frame_store=True, lazy_states=lazy, **base), ekws, usd, B, T, A, ss, h, 3, check=check)
    monkeypatch.delenv("A2C_NO_W1_FRAMES")
    assert seen["frames"] == 3 and (not lazy or seen["stale"] == 5)
    for k in range(3):
        for n in plain[k][0]:
            assert torch.equal(plain[k][0][n], store[k][0][n]), (k, n)
        for name in (plain[k][1] or {}):        # (fp64 sums of per-workgroup partials, atomics: equal to the order of the additions)
            assert plain[k][1][name] == pytest.approx(store[k][1][name], rel=1e-12, abs=1e-15), (k, name)
        for a, b in zip(plain[k][2], store[k][2]):
            assert torch.equal(a, b), k
    # round 0 against the oracle (later rounds: test_rollout_update_rollout_matches_oracle pins the plain path)
    onet = O.OracleNet(kind, ss, A, h)
    ref = _oracle_rollouts(kind, onet, base_hyps(**base), ekws, us, 1, B, T, ss)[0]
    _compare_round({k: v for k, v in store[0][0].items()}, ref, onet.is_recurrent)


@pytest.mark.parametrize("bptt", [True, False])
def test_gru_update_with_the_fused_first_layer_gradient_equals_the_unfused_launches(bptt, monkeypatch):
    """GRUModel on the single-frame store: the update's conv2 backward-data + conv1 weight gradient as ONE launch
    (a2c_conv2d_bwd_data_w1_frames: the 16 x 84 x 84 input gradient of conv2 never reaches HBM) against the two launches it
    replaces (A2C_NO_W1_FRAMES=1) after identical rollouts: every gradient but conv1's bit-identical, conv1's weight / bias
    gradient at re-association tolerance (2e-6 of its scale), the reported infos to 1e-6 (GradNorm sums that gradient)."""
    B, T, A, ss, h = 5, 6, 3, (4, 84, 84), 256
    ekws = [dict(env_id=j, rew_period=2 + j % 2, done_period=4 + j) for j in range(B)]
    base = dict(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-3, optim_type="RMSprop",
                use_bptt=bptt, h_size=h, frame_store=True, lazy_states=True)
    usd = torch.from_numpy(hashf(2 * T * B, 6152, 0, 1).reshape(2, T, B)).to(DEV)
    res = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            monkeypatch.setenv("A2C_NO_W1_FRAMES", "1")
        grads, spans = [], []

        def check(r, net, D, k, grads=grads):
            if k == -1:                                         # right after the update: the arena still holds its gradients
                grads.append({n: net.G(n).detach().cpu().clone() for n, _ in net.named_parameters() if n not in net._unused_params})
        outs = _run_engine("GRUModel", base_hyps(**base), ekws, usd, B, T, A, ss, h, 2, check=check)
        res[mode] = (outs, grads)
    monkeypatch.delenv("A2C_NO_W1_FRAMES")
    (fo, fg), (uo, ug) = res["fused"], res["unfused"]
    for n in fo[0][0]:
        assert torch.equal(fo[0][0][n], uo[0][0][n]), n            # the same rollout
    assert len(fg) == 1 and len(ug) == 1
    for n in fg[0]:
        if n.startswith("convs.0.0"):
            sc = float(ug[0][n].abs().max())
            close(n, fg[0][n], ug[0][n], 2e-6 * sc, 1e-5)
            assert not torch.equal(fg[0][n], ug[0][n]) or sc == 0.0      # (the fused pass really ran: another summation order)
        else:
            assert torch.equal(fg[0][n], ug[0][n]), n
    for k in fo[0][1]:
        assert fo[0][1][k] == pytest.approx(uo[0][1][k], rel=1e-6, abs=1e-9), k


@pytest.mark.parametrize("prep,kind", [("pong_prep", "A3CModel"), ("breakout_prep", "FCModel")])
def test_device_preprocessing_equals_host_preprocessing(prep, kind):
    """hyps['device_prep']: the env workers hand on RAW 210 x 160 x 3 frames (100,800 B over the link) and
    a2c_frame_prep_u8 runs pong_prep / breakout_prep on the device; against the same envs preprocessed on the host
    (the reference's place for it, runner.py:61-69): identical rollout buffers."""
    B, T, A, h = 3, 5, 3, 64
    shape = {"pong_prep": (1, 80, 80), "breakout_prep": (1, 80, 72)}[prep]
    ss = (4,) + shape[1:]
    base = dict(env_type="FakePong-v0" if prep == "pong_prep" else "FakeBreakout", n_tsteps=T, n_rollouts=B, action_shift=0,
                n_envs=B, h_size=h)
    us = torch.from_numpy(hashf(2 * T * B, 881, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    mk = lambda p: [dict(env_id=j, rew_period=2 + j, done_period=4 + j, prep=p) for j in range(B)]
    host = _run_engine(kind, base_hyps(**base), mk(prep), usd, B, T, A, ss, h, 2, env_cls=RawAtariEnv)
    dev = _run_engine(kind, base_hyps(device_prep=prep, **base), mk(None), usd, B, T, A, ss, h, 2, env_cls=RawAtariEnv)
    for k in range(2):
        for n in host[k][0]:
            assert torch.equal(host[k][0][n], dev[k][0][n]), (k, n)
        for name in (host[k][1] or {}):
            assert host[k][1][name] == pytest.approx(dev[k][1][name], rel=1e-12, abs=1e-15), (k, name)
    st = dev[0][0]["states"]
    assert float(st.max()) == (1.0 if prep == "pong_prep" else 255.0) or float(st.max()) > 1.0


@pytest.mark.parametrize("no_ring,B,T", [(False, 7, 9), (True, 7, 9), (False, 300, 5)])
def test_a3c_persistent_rollout_on_the_frame_store_with_lazy_states(no_ring, B, T, monkeypatch):
    """hyps['frame_store'] + hyps['lazy_states'] on the headline path (zero-copy ring kernel, packed frames): the kernel
    writes ONE uint8 frame per env step (7 KB) instead of the 113 KB fp32 state row, update_model's first-layer weight
    gradient stacks the frames on load, `states` is expanded on demand.  Against the plain run: identical rollout buffers
    (after materialize_states), infos and weights to fp32 re-association of one kernel.  With A2C_NO_RING=1 (the per-step
    persistent body, which cannot skip the rows) lazy_states is ignored and the rows are there.  300 envs: more envs than
    CUs, two interleaved blocks of the ring kernel one after the other."""
    from a2c_amd.hostpool import ThreadEnvPool
    from a2c_amd.runner import Runner
    from a2c_amd.synthetic import TapeEnv
    from a2c_amd.updater import Updater
    if no_ring:
        monkeypatch.setenv("A2C_NO_RING", "1")
    A, ss = 3, (4, 84, 84)
    us = torch.from_numpy(hashf(3 * T * B, 99, 0, 1).reshape(3, T, B)).to(DEV)
    res = {}
    for mode in ("plain", "lazy"):
        hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, lr=1e-4,
                         **(dict(frame_store=True, lazy_states=True) if mode == "lazy" else {}))
        net = make_net("A3CModel", ss, A, 256)
        D = _datas(B * T, ss, False, actions_on_host=False)
        envs = [TapeEnv(env_id=j, length=3 * T + 1, p_done=1.0 / 6) for j in range(B)]
        pool = ThreadEnvPool.from_tape_envs(envs, n_threads=2 if B < 64 else 6, pong=True, frame_bits=True)
        rnd = [0]
        r = Runner(D, hyps, None, None, None, env_pool=pool, ingest="zero-copy",
                   uniform_fn=lambda t, Bn, env0: us[rnd[0], t, env0:env0 + Bn].contiguous())
        upd = Updater(net, hyps)
        out = []
        try:
            for rnd[0] in range(3):
                r.rollout(net, list(range(B)), hyps)
                r.finish()
                if mode == "lazy":
                    assert net._stash_frames is not None
                    assert bool(getattr(r, "_states_stale", False)) == (not no_ring)
                    if rnd[0] == 0 and not no_ring:
                        assert float(D["states"].abs().sum()) == 0.0           # never written by the rollout
                info = upd.update_model(D)
                if mode == "lazy" and not no_ring:
                    assert r._states_stale                                      # the update read the store, not the rows
                r.materialize_states()
                out.append(({k: v.detach().cpu().clone() for k, v in D.items()}, info,
                            [p.detach().cpu().clone() for p in net.parameters()]))
        finally:
            r.close()
        res[mode] = out
    for k in range(3):
        for n in res["plain"][k][0]:
            if k == 0 or n in ("states", "dones"):      # (later rounds: the nets differ by the re-association below)
                assert torch.equal(res["plain"][k][0][n], res["lazy"][k][0][n]), (k, n)
        for name in res["plain"][k][1]:
            assert res["plain"][k][1][name] == pytest.approx(res["lazy"][k][1][name], rel=1e-5, abs=1e-7), (k, name)
        if B < 64:
            for a, b in zip(res["plain"][k][2], res["lazy"][k][2]):
                close("weights", b, a, 2e-6, 1e-5)
        elif k == 0:
            # N = 1500 samples: RMSprop divides by sqrt(mean-square), so where a gradient element is at rounding-noise level the
            # step is up to lr / sqrt(1 - alpha) = 1e-3 whatever its size and the re-association of the first layer's weight
            # gradient decides it -- the first update must differ in that layer ONLY; later rounds start from nets that differ
            # there and are compared through their infos above
            for i, (a, b) in enumerate(zip(res["plain"][k][2], res["lazy"][k][2])):
                if i < 2:
                    close("conv1 weights", b, a, 1e-4, 1e-5)
                else:
                    assert torch.equal(a, b), i


@pytest.mark.parametrize("B,vmax,p4", [(5, 256, 0.25), (700, 256, 0.25), (700, 2, 0.9), (32768 // 8, 2, 0.97), (32768 // 8, 256, 1.0)])
def test_a3c_conv1_weight_gradient_on_the_bf16_pipe_is_the_fp32_sum(B, vmax, p4, monkeypatch):
    """A3CModel conv1 (8x8 / stride 4, models.py:36) weight gradient from the single-frame uint8 store.  Default: the bf16
    matrix pipe with every fp32 dOut value split into three bf16 pieces (exact) and the uint8 pixels as bf16 (exact): every
    product is exact, all sums are fp32 -- so the result must be the fp32 MFMA kernel's (A2C_WGRAD_F32=1) up to
    re-association, for ANY uint8 pixel values (vmax = 256) and for binary frames (vmax = 2), and no further from the
    fp64 gradient than the fp32 kernel is (rms over the tensor; 1.5 x + 1e-7 of the tensor's rms).
    Round 6: a workgroup keeps the planes of its current sample in LDS as a ring and loads ONE frame for a successor sample
    (same slot, all four planes real: nvalid == 4 with probability p4 here, so runs of successors, fresh episodes and slot
    starts alternate): bit-identical to loading every sample whole (A2C_WSB_NO_RING=1)."""
    ops = _ops()
    d = ops.conv_desc(4, 84, 84, 16, 8, 4, 0)
    T, HW = 8, 84 * 84
    R = (B + T - 1) // T
    rng = np.random.default_rng(900 + B + vmax)
    Fs = rng.integers(0, vmax, size=(R, T + 4, HW), dtype=np.uint8)
    nv = np.where(rng.random(R * T) < p4, 4, rng.integers(1, 5, size=(R * T,))).astype(np.int32)
    Fd, nvd = torch.from_numpy(Fs).to(DEV), torch.from_numpy(nv).to(DEV)
    dout = (rnd((B, 16, 20, 20), 77) * torch.from_numpy(rng.lognormal(0, 2, size=(B, 1, 1, 1)).astype(np.float32))).to(DEV)
    ws = torch.empty((ops.conv_bwd_weight_ws_bytes(d, B) + 3) // 4, device=DEV)
    res = {}
    for f32 in (False, True):
        if f32:
            monkeypatch.setenv("A2C_WGRAD_F32", "1")
        else:
            monkeypatch.delenv("A2C_WGRAD_F32", raising=False)
        dW, db = torch.full((16, 4, 8, 8), float("nan"), device=DEV), torch.full((16,), float("nan"), device=DEV)
        ops.conv_bwd_weight_frames(d, Fd, Fd.stride(0), T, nvd, dout, dW, db, B, ws)
        dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
        ops.conv_bwd_weight_frames(d, Fd, Fd.stride(0), T, nvd, dout, dW2, db2, B, ws)
        assert torch.equal(dW, dW2) and torch.equal(db, db2)          # deterministic
        if not f32:
            monkeypatch.setenv("A2C_WSB_NO_RING", "1")
            ops.conv_bwd_weight_frames(d, Fd, Fd.stride(0), T, nvd, dout, dW2, db2, B, ws)
            monkeypatch.delenv("A2C_WSB_NO_RING")
            assert torch.equal(dW, dW2) and torch.equal(db, db2)      # the plane ring changes what is loaded, not what is summed
        res[f32] = (dW.cpu().double(), db.cpu().double())
    # fp64 gradient: dW[co][c][ky][kx] = sum_n sum_px dOut[n][co][px] x[n][c][4 oy + ky][4 ox + kx]
    xs = torch.zeros(B, 4, 84, 84, dtype=torch.float64)
    for n in range(B):
        r, t = divmod(n, T)
        for c in range(4):
            if c >= 4 - nv[n]:
                xs[n, c] = torch.from_numpy(Fs[r, t + c].astype(np.float64)).reshape(84, 84)
    wt = torch.zeros(16, 4, 8, 8, dtype=torch.float64, requires_grad=True)
    do64 = dout.cpu().double()
    for n0 in range(0, B, 512):
        F.conv2d(xs[n0:n0 + 512], wt, stride=4).backward(do64[n0:n0 + 512])
    want, wantb = wt.grad, do64.sum((0, 2, 3))
    rms = float(want.pow(2).mean().sqrt())
    e_bf = float((res[False][0] - want).pow(2).mean().sqrt()) / rms
    e_32 = float((res[True][0] - want).pow(2).mean().sqrt()) / rms
    assert e_bf <= 1.5 * e_32 + 1e-7, (e_bf, e_32)
    assert float((res[False][0] - want).abs().max()) <= 2 * float((res[True][0] - want).abs().max()) + 1e-6 * rms
    close("db", res[False][1], wantb, 2e-6 * float(wantb.abs().max()) + 1e-6, 2e-6)
    close("db fp32 kernel", res[True][1], wantb, 2e-6 * float(wantb.abs().max()) + 1e-6, 2e-6)
    print(f"B={B} vmax={vmax}: rms(bf16x3 - fp64)/rms = {e_bf:.2e}, rms(fp32 MFMA - fp64)/rms = {e_32:.2e}")
