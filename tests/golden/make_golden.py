#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  It imports the
reference's a2c/utils.py, a2c/models.py, a2c/updater.py and a2c/runner.py by
file path under a stub ``a2c`` package (the real ``a2c/__init__.py`` pulls in
skimage / gym / ml_utils, none of which are installed), drives them with
closed-form inputs and weights, and writes their OUTPUTS as small ``.npz``
files.  No reference source text is stored: fixtures hold numbers only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The oracle (oracle/a2c_oracle.py) is used here only for its closed-form input
generators (formula_state_dict, formula_frames, FakeEnv) so that tests can
regenerate the same inputs without storing them.
"""
import importlib.util
import os
import queue
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/a2c"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import a2c_oracle as O  # noqa: E402  (input generators only)

torch.set_num_threads(1)            # fixed reduction order for reproducibility


# ---------------------------------------------------------------- reference import
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _FakeGymEnv:
    """gym-shaped wrapper round oracle.FakeEnv: raw (H,W) frames, .seed, .action_space.n"""

    def __init__(self, spec):
        self.inner = O.FakeEnv(**spec["env_kwargs"])
        self.action_space = types.SimpleNamespace(n=spec["n_actions"])

    def seed(self, s):
        pass

    def reset(self):
        return self.inner.reset()[0]

    def step(self, a):
        o, r, d, i = self.inner.step(a)
        return o[0], r, d, i


_ENV_SPECS = {}


def _gym_make(env_type):
    return _FakeGymEnv(_ENV_SPECS[env_type])


def load_reference():
    _stub("gym", make=_gym_make)
    _stub("ml_utils")
    _stub("ml_utils.utils", try_key=lambda d, k, default: d[k] if k in d else default)
    _stub("mlagents_envs")
    _stub("mlagents_envs.environment", UnityEnvironment=object)
    _stub("mlagents_envs.side_channel")
    _stub("mlagents_envs.side_channel.engine_configuration_channel", EngineConfigurationChannel=object)
    _stub("mlagents_envs.side_channel.environment_parameters_channel", EnvironmentParametersChannel=object)
    _stub("gym_unity")
    _stub("gym_unity.envs", UnityToGymWrapper=object)
    pkg = _stub("a2c")
    pkg.__path__ = []
    mods = {}
    for name in ("utils", "models", "updater", "runner"):
        spec = importlib.util.spec_from_file_location(f"a2c.{name}", os.path.join(REF, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"a2c.{name}"] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
        mods[name] = m
    return mods


R = load_reference()
from cases import (MODEL_CASES, ROLLOUT_CASES, UPDATE_CASES, CHECKPOINT_CASES, STATS_CASES, SAMPLE, hashf, base_hyps, synth_shared,  # noqa: E402
                   sample_idx, null_prep)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, torch_version=np.array(torch.__version__), **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


# ---------------------------------------------------------------- G1 discount
def g1():
    out = {}
    cases = []
    # hand-derivable
    cases.append((np.array([1, 1, 1, 1], np.float32), np.array([0, 1, 0, 1], np.float32), 0.5))
    cases.append((np.array([0.1, 0.2, 0.3], np.float32), np.array([0, 0, 1], np.float32), 0.99 * 0.98))
    for i, (n, T, g) in enumerate([(64, 8, 0.99), (96, 32, 0.9702), (128, 16, 0.5), (257, 257, 0.99)]):
        x = hashf(n, 10 + i, -1, 1)
        d = (hashf(n, 50 + i) < 0.1).astype(np.float32)
        d[T - 1::T] = 1.0
        cases.append((x, d, g))
    cases.append((hashf(40, 90, -1, 1), np.ones(40, np.float32), 0.99))     # all done
    cases.append((hashf(40, 91, -1, 1), np.zeros(40, np.float32), 0.99))    # no done at all
    cases.append((np.zeros(0, np.float32), np.zeros(0, np.float32), 0.99))  # empty
    for i, (x, d, g) in enumerate(cases):
        y = R["utils"].discount(torch.from_numpy(x), torch.from_numpy(d), g)
        out[f"x{i}"], out[f"d{i}"], out[f"g{i}"], out[f"y{i}"] = x, d, np.float64(g), y.numpy()
    out["n_cases"] = np.array(len(cases))
    save("g1_discount.npz", **out)


# ---------------------------------------------------------------- G2 sample_action
def g2():
    real_rand = torch.rand
    probs = [np.array([[.25, .25, .5]], np.float32), np.array([[.25, .25, .5]], np.float32),
             np.array([[.3, .3, .3]], np.float32), np.array([[0, .5, .5]], np.float32)]
    us = [np.array([.25], np.float32), np.array([.95], np.float32), np.array([.95], np.float32),
          np.array([0.], np.float32)]
    # batched random cases, A = 2,3,4,6,18
    for i, A in enumerate([2, 3, 4, 6, 18]):
        lg = hashf(64 * A, 200 + i, -3, 3).reshape(64, A)
        p = torch.softmax(torch.from_numpy(lg), -1).numpy()
        probs.append(p)
        us.append(hashf(64, 300 + i))
    out = {"n_cases": np.array(len(probs))}
    for i, (p, u) in enumerate(zip(probs, us)):
        torch.rand = lambda *shape, _u=u: torch.from_numpy(_u).reshape(shape)
        a = R["utils"].sample_action(torch.from_numpy(p))
        out[f"p{i}"], out[f"u{i}"], out[f"a{i}"] = p, u, a.numpy()
    torch.rand = real_rand
    save("g2_sample_action.npz", **out)


# ---------------------------------------------------------------- G3 next_state
def g3():
    from collections import deque

    class E:
        def __init__(self):
            self.k = 0

        def reset(self):
            self.k += 1
            return np.full((1, 2, 3), 7.0 * self.k)

    env, dq = E(), deque(maxlen=3)
    seq = [(None, True), (np.full((1, 2, 3), 9.0), False), (np.full((1, 2, 3), 11.0), False),
           (np.full((1, 2, 3), 13.0), True), (np.full((1, 2, 3), 15.0), False)]
    outs = [R["utils"].next_state(env, dq, obs=o, reset=r) for o, r in seq]
    save("g3_next_state.npz", states=np.stack(outs), dtype=np.array(str(outs[0].dtype)))


# ---------------------------------------------------------------- models
def ref_model(kind, state_shape, n_actions, h_size):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):     # "Flat Features Size" prints
        net = getattr(R["models"], kind)(list(state_shape), n_actions, h_size=h_size, bnorm=False)
    sd = O.formula_state_dict(kind, state_shape, n_actions, h_size)
    ref_sd = net.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), (kind, set(ref_sd) ^ set(sd))
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (kind, k, ref_sd[k].shape, sd[k].shape)
    net.load_state_dict(sd)
    # parameter order must match the oracle's (optimizer state indices depend on it)
    shapes, _ = O.param_shapes(kind, state_shape, n_actions, h_size)
    prim = [k for k in shapes if "running" not in k and "num_batches" not in k]
    assert [n for n, _ in net.named_parameters()] == prim, (kind, [n for n, _ in net.named_parameters()], prim)
    return net


def g4():
    out = {"n_cases": np.array(len(MODEL_CASES))}
    for i, (kind, ss, A, h, B) in enumerate(MODEL_CASES):
        net = ref_model(kind, ss, A, h)
        x = torch.from_numpy(O.formula_frames(B, ss, seed=400 + i, binary=(len(ss) == 3 and ss[-1] == 84)))
        with torch.no_grad():
            if net.is_recurrent:
                hin = torch.from_numpy(hashf(B * h, 450 + i, -1, 1).reshape(B, h))
                v, p, hn = net(x, hin)
                out[f"h{i}"] = hn.numpy()
            else:
                v, p = net(x)
        out[f"val{i}"], out[f"pi{i}"] = v.numpy(), p.numpy()
        out[f"nparams{i}"] = np.array(sum(q.numel() for q in net.parameters()))
    save("g4_model_forward.npz", **out)


# ---------------------------------------------------------------- G5 rollout trace
def g5():
    out = {}
    real_rand = torch.rand
    for (name, kind, env_type, T, n_slots, ekw, A) in ROLLOUT_CASES:
        _ENV_SPECS[env_type] = dict(env_kwargs=ekw, n_actions=A)
        hyps = base_hyps(env_type=env_type, n_tsteps=T, n_rollouts=n_slots,
                         action_shift=1 if "Pong" in env_type else 0)
        ss = (4, 84, 84)
        net = ref_model(kind, ss, A, 256)
        N = T * n_slots
        datas = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N),
                     actions=torch.zeros(N).long(), dones=torch.zeros(N))
        if net.is_recurrent:
            datas["h_states"] = torch.zeros(N, 256)
        rew_q = queue.Queue(1)
        rew_q.put(-1)
        runner = R["runner"].Runner(datas, hyps, None, None, rew_q)
        # body of Runner.run (runner.py:158-168) without its infinite loop
        runner.net = net
        runner.env = R["runner"].SequentialEnvironment(**hyps)
        state = R["utils"].next_state(runner.env, runner.obs_deque, obs=None, reset=True)
        runner.state_bookmark = state
        runner.h_bookmark = torch.zeros(1, net.h_size) if net.is_recurrent else None
        runner.ep_rew = 0
        for p in net.parameters():
            p.requires_grad = False
        us = hashf(N + 8, 500 + len(out))
        it = iter(us)
        torch.rand = lambda *shape: torch.tensor([next(it)], dtype=torch.float32).reshape(shape)
        for idx in range(n_slots):
            runner.rollout(net, idx, hyps)
        torch.rand = real_rand
        st = datas["states"]
        out[f"{name}_uniforms"] = us
        out[f"{name}_rewards"] = datas["rewards"].numpy()
        out[f"{name}_dones"] = datas["dones"].numpy()
        out[f"{name}_actions"] = datas["actions"].numpy()
        out[f"{name}_deltas"] = datas["deltas"].numpy()
        out[f"{name}_state_sums"] = st.reshape(N, -1).double().sum(1).numpy()
        out[f"{name}_state_frame_sums"] = st.reshape(N, 4, -1).double().sum(2).numpy()
        if net.is_recurrent:
            out[f"{name}_h_states"] = datas["h_states"].numpy()
            out[f"{name}_h_bookmark"] = runner.h_bookmark.numpy()
        out[f"{name}_avg_rew"] = np.array(rew_q.get())
        out[f"{name}_bookmark_sum"] = np.array(np.asarray(runner.state_bookmark).sum())
    save("g5_rollout.npz", **out)


# ---------------------------------------------------------------- G6/G7 update_model
def g6():
    out = {}
    for (name, kind, ss, A, h, R_, T, opt, norm_advs, nstep, use_bptt, n_upd) in UPDATE_CASES:
        net = ref_model(kind, ss, A, h)
        hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, norm_advs=norm_advs,
                         use_nstep_rets=nstep, use_bptt=use_bptt, h_size=h)
        upd = R["updater"].Updater(net, hyps)
        pnames = [n for n, _ in net.named_parameters()]
        for u in range(n_upd):
            D = synth_shared(kind, ss, A, h, R_, T, seed=700 + 10 * u, recurrent=net.is_recurrent)
            grads = {}
            real_step = upd.optim.step

            def step_and_snap(*a, **k):
                for n, p in net.named_parameters():
                    grads[n] = None if p.grad is None else p.grad.detach().clone()
                return real_step(*a, **k)
            upd.optim.step = step_and_snap
            info = upd.update_model(D)
            upd.optim.step = real_step
            pre = f"{name}_u{u}_"
            for k, v in info.items():
                out[pre + k] = np.array(float(v))
            advs = R["utils"].discount(D["deltas"], D["dones"], hyps["gamma"] * hyps["lambda_"])
            out[pre + "advs_raw"] = advs.numpy()
            if not nstep:
                out[pre + "returns"] = R["utils"].discount(D["rewards"], D["dones"], hyps["gamma"]).numpy()
            gn, gs, pn, ps, has = [], [], [], [], []
            for n, p in net.named_parameters():
                g = grads[n]
                has.append(g is not None)
                idx = sample_idx(p.numel())
                gn.append(0.0 if g is None else float(g.double().norm()))
                gs.append(np.zeros(SAMPLE, np.float32) if g is None else g.reshape(-1)[idx].numpy())
                pn.append(float(p.detach().double().norm()))
                ps.append(p.detach().reshape(-1)[idx].numpy().copy())
            out[pre + "grad_norms"] = np.array(gn)           # post-clip grads
            out[pre + "grad_samples"] = np.stack(gs)
            out[pre + "param_norms"] = np.array(pn)          # post-step params
            out[pre + "param_samples"] = np.stack(ps)
            out[pre + "has_grad"] = np.array(has)
        out[name + "_param_names"] = np.array(pnames)
    save("g6_update.npz", **out)


def g7():
    """Checkpoint resume: update once, Updater.save_model (updater.py:211-219), update again."""
    out = {}
    here = os.path.dirname(os.path.abspath(__file__))
    for (name, kind, ss, A, h, R_, T, opt, use_bptt) in CHECKPOINT_CASES:
        net = ref_model(kind, ss, A, h)
        hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, use_bptt=use_bptt, h_size=h)
        upd = R["updater"].Updater(net, hyps)
        upd.update_model(synth_shared(kind, ss, A, h, R_, T, seed=910, recurrent=net.is_recurrent))
        upd.save_model(os.path.join(here, f"g7_{name}_net.p"), os.path.join(here, f"g7_{name}_optim.p"))
        info = upd.update_model(synth_shared(kind, ss, A, h, R_, T, seed=920, recurrent=net.is_recurrent))
        for k, v in info.items():
            out[f"{name}_{k}"] = np.array(float(v))
        out[name + "_param_names"] = np.array([n for n, _ in net.named_parameters()])
        for n, p in net.named_parameters():
            out[f"{name}_param_{n}"] = p.detach().numpy().copy()
    save("g7_checkpoint.npz", **out)


# ---------------------------------------------------------------- G8 StatsRunner.rollout
def g8():
    out = {}
    real_rand = torch.rand
    for (name, kind, env_type, n_eps, ekw, A) in STATS_CASES:
        _ENV_SPECS[env_type] = dict(env_kwargs=ekw, n_actions=A)
        hyps = base_hyps(env_type=env_type, n_test_eps=n_eps, action_shift=1 if "Pong" in env_type else 0)
        net = ref_model(kind, (4, 84, 84), A, 256)
        sr = R["runner"].StatsRunner(hyps)
        us = hashf(400, 800 + len(out))
        used = [0]

        def fake_rand(*shape):
            used[0] += 1
            return torch.tensor([us[used[0] - 1]], dtype=torch.float32).reshape(shape)
        torch.rand = fake_rand
        with torch.no_grad():
            avg = sr.rollout(net)
        torch.rand = real_rand
        out[f"{name}_uniforms"] = us[:used[0]]
        out[f"{name}_avg_rew"] = np.array(float(avg))
    save("g8_stats.npz", **out)


def g9():
    """The reference's own pong_prep / breakout_prep / null_prep (preprocessing.py:8-23) on Atari-shaped frames.
    preprocessing.py imports skimage.color.rgb2grey, which is not installed: it is stubbed with the documented behaviour
    of scikit-image <= 0.18 for the ONLY call the file makes (a 2-D array is returned unchanged, `rgb2gray`:
    "if rgb.ndim == 2: return np.ascontiguousarray(rgb)"); a non-2-D input raises, so a change of the call would show.
    pong_prep and null_prep do not touch the stub: those vectors are the reference's, unconditionally."""
    from cases import atari_frame, PREP_SEEDS

    def rgb2grey(a):
        a = np.asarray(a)
        if a.ndim != 2:
            raise AssertionError("stub: the reference only ever passes a 2-D slice")
        return np.ascontiguousarray(a)
    _stub("skimage")
    _stub("skimage.color", rgb2grey=rgb2grey)
    spec = importlib.util.spec_from_file_location("a2c.preprocessing", os.path.join(REF, "preprocessing.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for s_ in PREP_SEEDS:
        out[f"pong{s_}"] = np.array(m.pong_prep(atari_frame(s_)))
        out[f"breakout{s_}"] = np.array(m.breakout_prep(atari_frame(s_)))
        out[f"null{s_}_shape"] = np.array(m.null_prep(atari_frame(s_)).shape)
        assert out[f"pong{s_}"].dtype == np.uint8 and out[f"breakout{s_}"].dtype == np.uint8
    save("g9_preprocessing.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for fn in sys.argv[1:]:
            globals()[fn]()
    else:
        g1(); g2(); g3(); g4(); g5(); g6(); g7(); g8(); g9()
