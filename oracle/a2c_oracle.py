"""CPU ORACLE for the A2C rollout+update hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This module restates, on the CPU and from scratch, the algorithm that
grantsrb/PyTorch-A2C runs on its rollout+update path, so that the MI355X HIP
kernels can be checked against it.  It is only ever imported by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; the
product package (``pytorch-a2c_amd/a2c_amd``) never imports it and fails
loudly when its HIP library is missing.

Pinning: the reference has no tests and no golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned by the fixtures under
``tests/golden/*.npz``.  Those were produced in the build container by
``tests/golden/make_golden.py``, which imports the reference's own
``a2c/utils.py``, ``a2c/models.py``, ``a2c/updater.py`` and ``a2c/runner.py``
from ``/root/reference`` (file by file, under stub ``gym``/``ml_utils``
modules) and records their outputs.  ``tests/test_oracle_golden.py`` checks
every function here against those recorded outputs.

The dense arithmetic (conv2d, linear, softmax, autograd, RMSprop/Adam) is the
reference's own third-party dependency: ``torch`` (unpinned in the reference's
requirements.txt:2; the fixtures were made with torch 2.10.0 CPU).  The oracle
calls the same torch CPU operators the reference calls, and restates everything
the reference itself wrote in Python around them.

Reference citations are ``file:line`` into /root/reference/a2c/.
"""
from __future__ import annotations

import math
from collections import OrderedDict, deque

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# utils.py
# --------------------------------------------------------------------------

def try_key(d, key, default):
    """utils.py:4-7."""
    return d[key] if key in d else default


def discount(array, dones, factor):
    """Reverse segmented discounted sum (utils.py:63-79).

    ``y[i] = x[i] + factor * (0 if dones[i] == 1 else y[i+1])``, ``y[N] = 0``;
    the reset is applied *before* the add, so a done step keeps its own value.
    The reference keeps ``running_sum`` as a 0-d fp32 tensor once the first
    element has been read: every step is one fp32 multiply then one fp32 add.
    Sequential per-element loop, like the reference (this is what the CPU
    baseline times).
    """
    n = len(array)
    out = torch.zeros(n)
    run = 0
    for i in range(n - 1, -1, -1):
        if dones[i] == 1:
            run = 0
        run = array[i] + factor * run
        out[i] = run
    return out


def discount_np(x, dones, factor):
    """Same recurrence on numpy fp32 arrays (fast checker for large N).

    Bit-identical to :func:`discount`: ``factor`` is rounded to fp32 once
    (python float * 0-d fp32 tensor), the product and the sum are each rounded
    to fp32.
    """
    x = np.asarray(x, dtype=np.float32)
    d = np.asarray(dones, dtype=np.float32)
    g = np.float32(factor)
    out = np.empty_like(x)
    run = np.float32(0.0)
    for i in range(len(x) - 1, -1, -1):
        if d[i] == 1:
            run = np.float32(0.0)
        run = np.float32(x[i] + np.float32(g * run))
        out[i] = run
    return out


def sample_action(pi, rand_nums=None):
    """Inverse-CDF sampler (utils.py:45-60).

    Running fp32 cumulative sum over the action axis in index order; the first
    index whose cumsum is ``>= u`` wins; if none does the result stays -1.
    ``rand_nums`` lets a test supply the uniforms the reference draws with
    ``torch.rand`` (utils.py:54).
    """
    pi = pi.detach().cpu().float()
    lead = pi.shape[:-1]
    if rand_nums is None:
        rand_nums = torch.rand(*lead)
    rand_nums = torch.as_tensor(rand_nums, dtype=torch.float32).reshape(lead)
    csum = torch.zeros(lead)
    acts = -torch.ones(lead)
    for a in range(pi.shape[-1]):
        csum = csum + pi[..., a]
        hit = (csum >= rand_nums) & (acts < 0)
        acts = torch.where(hit, torch.full_like(acts, float(a)), acts)
    return acts


def next_state(env, obs_deque, obs, reset):
    """Frame stack (utils.py:26-43): oldest frame first on axis 0.

    On reset the passed ``obs`` is discarded, ``maxlen-1`` zero frames and the
    ``env.reset()`` frame are pushed.  ``np.zeros`` makes the result float64.
    """
    if reset:
        obs = env.reset()
        for _ in range(obs_deque.maxlen - 1):
            obs_deque.append(np.zeros(obs.shape))
    obs_deque.append(obs)
    return np.concatenate(obs_deque, axis=0)


# --------------------------------------------------------------------------
# preprocessing.py (pure numpy slicing; skimage is absent so breakout's
# rgb2grey of a single channel is restated as the identity on 2-D input)
# --------------------------------------------------------------------------

def null_prep(pic):
    """preprocessing.py:8-9."""
    return pic[None]


def pong_prep(pic):
    """preprocessing.py:11-17: crop 35:195, 2x downsample of channel 0, binarise."""
    pic = pic[35:195]
    pic = pic[::2, ::2, 0].copy()
    pic[pic == 144] = 0
    pic[pic == 109] = 0
    pic[pic != 0] = 1
    return pic[None]


def breakout_prep(pic):
    """preprocessing.py:19-23: crop 35:195 x 8:-8, 2x downsample of channel 0, then skimage's rgb2grey -- which, for the
    2-D slice it is handed, returns the array unchanged in every scikit-image release that still has the name
    (<= 0.18; pinned by tests/golden/g9_preprocessing.npz under exactly that stub)."""
    pic = pic[35:195, 8:-8]
    pic = pic[::2, ::2, 0]
    return np.ascontiguousarray(pic)[None]


# --------------------------------------------------------------------------
# models.py  --  parameter shapes and functional forwards
# --------------------------------------------------------------------------

def _conv_out(n, k, p, s):
    return (n - k + 2 * p) // s + 1

# (c_out, ksize, stride, padding) per conv layer
A3C_CONVS = [(16, 8, 4, 0), (32, 4, 2, 0)]                                   # models.py:23-36
CONV_CONVS = [(16, 3, 1, 1), (24, 3, 1, 1), (32, 3, 2, 1), (64, 3, 2, 1)]    # models.py:201-240
GRU_CONVS = [(16, 3, 1, 1), (24, 3, 2, 1), (32, 3, 2, 1), (48, 3, 2, 1), (64, 3, 2, 1)]  # models.py:570-622
CONV_H = 2000                                                                # models.py:245


def _feat_shape(state_shape, convs):
    c, h, w = state_shape[-3:]
    for (co, k, s, p) in convs:
        h, w, c = _conv_out(h, k, p, s), _conv_out(w, k, p, s), co
    return c, h, w


def param_shapes(kind, state_shape, n_actions, h_size):
    """Ordered ``state_dict`` key -> shape for each reference model class.

    Key names include the aliases the reference's module tree produces
    (A3CModel registers the same conv blocks under ``convs.*``, ``conv1/2.*``
    and ``features.*``: models.py:21-38).  ``primary`` keys (the ones
    ``named_parameters()`` yields, i.e. the optimiser's parameter order) come
    first in each alias group.
    """
    cin = state_shape[-3] if len(state_shape) >= 3 else None
    sd = OrderedDict()
    alias = {}   # alias key -> primary key

    def conv_stack(convs):
        c = cin
        for i, (co, k, s, p) in enumerate(convs):
            sd[f"convs.{i}.0.weight"] = (co, c, k, k)
            sd[f"convs.{i}.0.bias"] = (co,)
            c = co

    if kind == "A3CModel":
        conv_stack(A3C_CONVS)
        for i in range(2):
            for t in ("weight", "bias"):
                alias[f"conv{i+1}.0.{t}"] = f"convs.{i}.0.{t}"
                alias[f"features.{i}.0.{t}"] = f"convs.{i}.0.{t}"
        flat = int(np.prod(_feat_shape(state_shape, A3C_CONVS)))
        sd["proj_matrx.weight"] = (h_size, flat); sd["proj_matrx.bias"] = (h_size,)
        sd["emb_bnorm.weight"] = (h_size,); sd["emb_bnorm.bias"] = (h_size,)
        sd["emb_bnorm.running_mean"] = (h_size,); sd["emb_bnorm.running_var"] = (h_size,)
        sd["emb_bnorm.num_batches_tracked"] = ()
        sd["pi.weight"] = (n_actions, h_size); sd["pi.bias"] = (n_actions,)
        sd["value.weight"] = (1, h_size); sd["value.bias"] = (1,)
    elif kind == "ConvModel":
        conv_stack(CONV_CONVS)
        for i in range(4):
            for t in ("weight", "bias"):
                alias[f"features.{i}.0.{t}"] = f"convs.{i}.0.{t}"
        flat = int(np.prod(_feat_shape(state_shape, CONV_CONVS)))
        sd["resize_emb.0.weight"] = (CONV_H, flat); sd["resize_emb.0.bias"] = (CONV_H,)
        sd["pi.0.weight"] = (h_size, CONV_H); sd["pi.0.bias"] = (h_size,)
        sd["pi.2.weight"] = (n_actions, h_size); sd["pi.2.bias"] = (n_actions,)
        sd["value.0.weight"] = (h_size, CONV_H); sd["value.0.bias"] = (h_size,)
        sd["value.2.weight"] = (1, h_size); sd["value.2.bias"] = (1,)
    elif kind == "GRUModel":
        conv_stack(GRU_CONVS)
        for i in range(5):
            for t in ("weight", "bias"):
                alias[f"features.{i}.0.{t}"] = f"convs.{i}.0.{t}"
        flat = int(np.prod(_feat_shape(state_shape, GRU_CONVS)))
        sd["resize_emb.0.weight"] = (h_size, flat); sd["resize_emb.0.bias"] = (h_size,)
        sd["gru.W_x"] = (3, h_size, h_size); sd["gru.W_h"] = (3, h_size, h_size)
        sd["gru.b"] = (3, 1, h_size)
        sd["pi.weight"] = (n_actions, h_size); sd["pi.bias"] = (n_actions,)
        sd["value.weight"] = (1, h_size); sd["value.bias"] = (1,)
    elif kind in ("FCModel", "GRUFCModel"):
        flat = int(np.prod(state_shape[-3:]))
        sd["base.0.weight"] = (h_size, flat); sd["base.0.bias"] = (h_size,)
        sd["base.2.weight"] = (h_size, h_size); sd["base.2.bias"] = (h_size,)
        if kind == "GRUFCModel":
            sd["gru.W_x"] = (3, h_size, h_size); sd["gru.W_h"] = (3, h_size, h_size)
            sd["gru.b"] = (3, 1, h_size)
        sd["action_out.weight"] = (n_actions, h_size); sd["action_out.bias"] = (n_actions,)
        sd["value_out.0.weight"] = (h_size,); sd["value_out.0.bias"] = (h_size,)
        sd["value_out.1.weight"] = (1, h_size); sd["value_out.1.bias"] = (1,)
        sd["value_out.2.weight"] = (1, 1); sd["value_out.2.bias"] = (1,)
    else:
        raise KeyError(kind)
    return sd, alias


def formula_tensor(shape, k, kind="weight"):
    """Closed-form deterministic fill (no RNG, no big fixture files).

    ``v[i] = scale * sin(0.37*i + 1.3*k)`` computed in float64 and rounded to
    fp32 once.  ``scale`` is ``1/sqrt(fan_in)`` for matrices/filters, 0.05 for
    biases; norm-layer weights are ``1 + 0.1*sin``.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    s = np.sin(0.37 * i + 1.3 * k)
    if kind == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if len(shape) == 3:           # GRU (3, in, h): x.mm(W[g]) -> fan_in = in
            fan_in = shape[1]
        v = s / math.sqrt(max(fan_in, 1))
    elif kind == "bias":
        v = 0.05 * s
    elif kind == "norm_weight":
        v = 1.0 + 0.1 * s
    else:
        raise KeyError(kind)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def formula_state_dict(kind, state_shape, n_actions, h_size):
    """Full reference-keyed state_dict (aliases included) filled by formula."""
    shapes, alias = param_shapes(kind, state_shape, n_actions, h_size)
    sd = OrderedDict()
    for k, (name, shape) in enumerate(shapes.items()):
        if name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        elif name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.startswith("emb_bnorm") or name.startswith("value_out.0"):
            sd[name] = formula_tensor(shape, k, "norm_weight" if name.endswith("weight") else "bias")
        elif name.endswith("bias") or name == "gru.b":
            sd[name] = formula_tensor(shape, k, "bias")
        else:
            sd[name] = formula_tensor(shape, k, "weight")
    for a, p in alias.items():
        sd[a] = sd[p]
    return sd


def formula_frames(n, shape, seed=0, binary=False):
    """Deterministic synthetic observations from an integer hash (exact).

    ``binary=True`` gives Pong-like {0,1} pixels with ~25% ones.
    """
    total = int(n * np.prod(shape))
    idx = np.arange(total, dtype=np.uint64)
    x = (idx * np.uint64(2654435761) + np.uint64(seed) * np.uint64(40503) + np.uint64(12345)) % np.uint64(1 << 32)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(2246822519) % np.uint64(1 << 32)
    x = (x ^ (x >> np.uint64(13)))
    b = ((x >> np.uint64(8)) & np.uint64(0xFF)).astype(np.float32)
    if binary:
        v = (b < 64).astype(np.float32)
    else:
        v = b / np.float32(255.0)
    return v.reshape((n,) + tuple(shape))


class OracleNet:
    """Functional restatement of the reference model classes (models.py).

    Holds leaf tensors keyed by the reference's primary state_dict names.
    ``forward`` mirrors: A3CModel.forward (models.py:60-90), ConvModel.forward
    (276-304), FCModel.forward (396-405), GRU.forward (465-476),
    GRUFCModel.forward (514-524), GRUModel.forward (648-677).
    """

    def __init__(self, kind, state_shape, n_actions, h_size, state_dict=None):
        self.kind = kind
        self.state_shape = list(state_shape)
        self.n_actions = n_actions
        self.h_size = h_size
        self.is_recurrent = kind in ("GRUModel", "GRUFCModel")
        shapes, self.alias = param_shapes(kind, state_shape, n_actions, h_size)
        if state_dict is None:
            state_dict = formula_state_dict(kind, state_shape, n_actions, h_size)
        self.p = OrderedDict()
        for name in shapes:
            t = state_dict[name].clone()
            if t.is_floating_point() and "running" not in name:
                t.requires_grad_(True)
            self.p[name] = t

    # parameters in the reference's ``net.parameters()`` order
    def parameters(self):
        return [t for n, t in self.p.items()
                if t.is_floating_point() and "running" not in n]

    def named_parameters(self):
        return [(n, t) for n, t in self.p.items()
                if t.is_floating_point() and "running" not in n]

    def req_grads(self, flag):
        for t in self.parameters():
            t.requires_grad_(flag)

    def _convs(self, x, specs):
        for i, (co, k, s, p) in enumerate(specs):
            x = F.relu(F.conv2d(x, self.p[f"convs.{i}.0.weight"], self.p[f"convs.{i}.0.bias"],
                                stride=s, padding=p))
        return x.reshape(x.shape[0], -1)

    def _gru(self, x, h):
        Wx, Wh, b = self.p["gru.W_x"], self.p["gru.W_h"], self.p["gru.b"]
        z = torch.sigmoid(x.mm(Wx[0]) + h.mm(Wh[0]) + b[0])
        r = torch.sigmoid(x.mm(Wx[1]) + h.mm(Wh[1]) + b[1])
        cand = torch.tanh(x.mm(Wx[2]) + (r * h).mm(Wh[2]) + b[2])
        return z * h + (1 - z) * cand

    def forward(self, x, h=None):
        P = self.p
        k = self.kind
        if k == "A3CModel":
            f = self._convs(x, A3C_CONVS)
            emb = F.linear(f, P["proj_matrx.weight"], P["proj_matrx.bias"])   # no activation (models.py:73)
            pi = F.linear(emb, P["pi.weight"], P["pi.bias"])
            val = F.linear(emb.detach(), P["value.weight"], P["value.bias"])  # detached (models.py:85)
            return val, pi
        if k == "ConvModel":
            f = self._convs(x, CONV_CONVS)
            emb = F.relu(F.linear(f, P["resize_emb.0.weight"], P["resize_emb.0.bias"]))
            pi = F.linear(F.relu(F.linear(emb, P["pi.0.weight"], P["pi.0.bias"])), P["pi.2.weight"], P["pi.2.bias"])
            val = F.linear(F.relu(F.linear(emb, P["value.0.weight"], P["value.0.bias"])), P["value.2.weight"], P["value.2.bias"])
            return val, pi
        if k == "GRUModel":
            f = self._convs(x, GRU_CONVS)   # "lerelu" resolves to ReLU (models.py:681-689)
            emb = F.relu(F.linear(f, P["resize_emb.0.weight"], P["resize_emb.0.bias"]))
            hn = self._gru(emb, h)
            pi = F.linear(hn, P["pi.weight"], P["pi.bias"])
            val = F.linear(hn, P["value.weight"], P["value.bias"])
            return val, pi, hn
        if k in ("FCModel", "GRUFCModel"):
            fx = x.reshape(len(x), -1)
            fx = F.linear(F.relu(F.linear(fx, P["base.0.weight"], P["base.0.bias"])), P["base.2.weight"], P["base.2.bias"])
            hn = None
            if k == "GRUFCModel":
                fx = hn = self._gru(fx, h)
            pi = F.linear(fx, P["action_out.weight"], P["action_out.bias"])
            v = F.layer_norm(fx, (self.h_size,), P["value_out.0.weight"], P["value_out.0.bias"])
            v = F.linear(F.linear(v, P["value_out.1.weight"], P["value_out.1.bias"]),
                         P["value_out.2.weight"], P["value_out.2.bias"])
            return (v, pi, hn) if hn is not None else (v, pi)
        raise KeyError(k)

    __call__ = forward


# --------------------------------------------------------------------------
# runner.py  --  batch-1 rollout of one slot (Runner.rollout, runner.py:174-248)
# --------------------------------------------------------------------------

class SlotRunner:
    """One env + its bookmarks, filling slot ``idx`` of the shared buffers.

    ``env`` needs ``reset() -> obs`` and ``step(a) -> (obs, rew, done, info)``
    returning already-prepped (1,H,W) frames (SequentialEnvironment.reset/step,
    runner.py:69-80).  ``uniform_fn()`` supplies the sampler's uniform.
    """

    def __init__(self, env, datas, hyps, uniform_fn=None):
        self.env, self.datas, self.hyps = env, datas, hyps
        self.obs_deque = deque(maxlen=hyps["n_frame_stack"])
        self.uniform_fn = uniform_fn
        self.ep_rew = 0.0
        self.avg_rew = -1.0          # rew_q initial value (training.py:106)
        self.env.reset()             # SequentialEnvironment.__init__ resets once to read raw_shape (runner.py:45)

    def start(self, net):
        """Body of Runner.run before its loop (runner.py:158-168)."""
        self.state = next_state(self.env, self.obs_deque, obs=None, reset=True)
        self.h = torch.zeros(1, net.h_size) if net.is_recurrent else None

    def get_action(self, logits):
        """runner.py:94-97."""
        probs = F.softmax(logits, dim=-1)
        u = None if self.uniform_fn is None else torch.tensor([self.uniform_fn()], dtype=torch.float32)
        return int(sample_action(probs.data, u).item())

    def rollout(self, net, idx):
        hyps, D = self.hyps, self.datas
        T = hyps["n_tsteps"]
        gamma = hyps["gamma"]
        s0 = idx * T
        state, h = self.state, self.h
        prev_val = None
        done = False
        with torch.no_grad():
            for i in range(T):
                D["states"][s0 + i] = torch.FloatTensor(state)
                x = D["states"][s0 + i].unsqueeze(0)
                if "h_states" in D:
                    D["h_states"][s0 + i] = h[0]
                    val, logits, h = net(x, h)
                else:
                    val, logits = net(x)
                a = self.get_action(logits)
                obs, rew, done, _ = self.env.step(a + hyps["action_shift"])
                self.ep_rew += rew
                reset = done
                if "Pong" in hyps["env_type"] and rew != 0:
                    done = True                                    # runner.py:213-214
                if done:
                    self.avg_rew = .99 * self.avg_rew + .01 * self.ep_rew
                    self.ep_rew = 0
                    if h is not None:
                        h = torch.zeros(1, net.h_size)
                D["rewards"][s0 + i] = rew
                D["dones"][s0 + i] = float(done)
                D["actions"][s0 + i] = a
                state = next_state(self.env, self.obs_deque, obs=obs, reset=reset)
                if i > 0:
                    pr, pd = D["rewards"][s0 + i - 1], D["dones"][s0 + i - 1]
                    D["deltas"][s0 + i - 1] = (pr + gamma * val * (1 - pd) - prev_val).reshape(())   # runner.py:231
                prev_val = val.squeeze()
            e = s0 + T - 1
            if not done:                                           # bootstrap, runner.py:237-244
                x = torch.FloatTensor(state).unsqueeze(0)
                val = net(x, h)[0] if "h_states" in D else net(x)[0]
                D["rewards"][e] += gamma * val.squeeze()
                D["dones"][e] = 1.
            D["deltas"][e] = D["rewards"][e] - prev_val            # runner.py:245
        self.state, self.h = state, h


# --------------------------------------------------------------------------
# updater.py  --  Updater.update_model / bptt (updater.py:33-169)
# --------------------------------------------------------------------------

def stats_rollout(net, env, hyps, n_episodes, uniform_fn):
    """Restates StatsRunner.rollout (runner.py:274-314): ``n_episodes`` episodes of ONE env played one after the
    other with the training sampler (softmax + inverse CDF, ``uniform_fn()`` replaces torch.rand), hidden state
    reset on done, Pong: an episode ends whenever a point is scored; returns total reward / episodes."""
    obs_deque = deque(maxlen=hyps["n_frame_stack"])
    state = next_state(env, obs_deque, obs=None, reset=True)
    h = torch.zeros(1, net.h_size) if net.is_recurrent else None
    ep_rew, ep_count = 0.0, 0
    with torch.no_grad():
        while ep_count < n_episodes:
            x = torch.FloatTensor(state)[None]
            if h is not None:
                val, logits, h = net(x, h)
            else:
                val, logits = net(x)
            probs = F.softmax(logits, dim=-1)
            action = int(sample_action(probs, torch.tensor([uniform_fn()], dtype=torch.float32)).item())
            obs, rew, done, _ = env.step(action + hyps["action_shift"])
            ep_rew += rew
            reset = done
            if "Pong" in hyps["env_type"] and rew != 0:
                done = True
            if done:
                ep_count += 1
                if h is not None:
                    h = torch.zeros(1, net.h_size)
            state = next_state(env, obs_deque, obs=obs, reset=reset)
    return ep_rew / ep_count


def bptt(net, states, h_states, dones, hyps):
    """updater.py:139-169."""
    R, T = hyps["n_rollouts"], hyps["n_tsteps"]
    hs = h_states.view(R, T, -1)[:, 0]
    xs = states.view(R, T, *states.shape[1:])
    keep = 1 - dones.view(R, T, 1)
    vals, logits = [], []
    for t in range(T):
        v, l, hs = net(xs[:, t], hs)
        hs = hs * keep[:, t]
        vals.append(v)
        logits.append(l.unsqueeze(1))
    return torch.cat(vals, dim=-1).view(-1), torch.cat(logits, dim=1).view(-1, logits[0].shape[-1])


class OracleUpdater:
    """Restates Updater (updater.py:9-229) around an :class:`OracleNet`."""

    def __init__(self, net, hyps):
        self.net, self.hyps = net, hyps
        self.optim = getattr(torch.optim, hyps["optim_type"])(net.parameters(), lr=hyps["lr"])  # updater.py:226-229
        self.ret_mean = self.ret_std = None
        self.info, self.norm = {}, 0

    def update_model(self, D, keep=False):
        hyps, net = self.hyps, self.net
        net.req_grads(True)
        advs = discount(D["deltas"].squeeze(), D["dones"].squeeze(), hyps["gamma"] * hyps["lambda_"])
        if "h_states" in D:
            if hyps["use_bptt"]:
                vals, logits = bptt(net, D["states"], D["h_states"], D["dones"], hyps)
            else:
                vals, logits, _ = net(D["states"], D["h_states"])
        else:
            vals, logits = net(D["states"])
        if hyps["use_nstep_rets"]:
            returns = advs + vals.data.squeeze()
        else:
            returns = discount(D["rewards"].squeeze(), D["dones"].squeeze(), hyps["gamma"])
        if try_key(hyps, "norm_returns", False):
            if self.ret_mean is None:
                self.ret_mean, self.ret_std = returns.mean(), returns.std()
            else:
                self.ret_mean = 0.01 * returns.mean() + 0.99 * self.ret_mean
                self.ret_std = 0.01 * returns.std() + 0.99 * self.ret_std
            returns = (returns - self.ret_mean) / (self.ret_std + 1e-6)
        raw_advs = advs
        if hyps["norm_advs"]:
            advs = (advs - advs.mean()) / (advs.std() + 1e-6)       # unbiased std, eps on std (updater.py:98)
        lsm = F.log_softmax(logits, dim=-1)
        log_ps = lsm[torch.arange(len(D["actions"])).long(), D["actions"]]
        entr = -hyps["entr_coef"] * (lsm * F.softmax(logits, dim=-1)).sum(-1).mean()
        pi_loss = hyps["pi_coef"] * -(log_ps * advs.squeeze()).mean()
        val_loss = hyps["val_coef"] * F.mse_loss(vals.squeeze(), returns)
        loss = pi_loss + val_loss - entr
        loss.backward()
        self.norm = torch.nn.utils.clip_grad_norm_(net.parameters(), hyps["max_norm"])
        extra = None
        if keep:
            extra = dict(advs_raw=raw_advs.detach().clone(), advs=advs.detach().clone(),
                         returns=returns.detach().clone(), vals=vals.detach().clone().reshape(-1),
                         logits=logits.detach().clone(),
                         grads={n: (None if p.grad is None else p.grad.detach().clone())
                                for n, p in net.named_parameters()})
        self.optim.step()
        self.optim.zero_grad()
        self.info = {"Loss": loss.item(), "Pi_Loss": pi_loss.item(), "ValLoss": val_loss.item(),
                     "Entropy": entr.item(), "GradNorm": float(self.norm)}
        return (self.info, extra) if keep else self.info


def update_grads_chunked(net, D, hyps, chunk_slots, dtype=torch.float32):
    """The loss and the gradients ``OracleUpdater.update_model`` backpropagates (updater.py:63-128), evaluated
    ``chunk_slots`` rollout slots at a time -- for sizes whose full-batch autograd graph does not fit host memory
    (GRUModel + BPTT at 256 x 128: 179 k activation floats x 32,768 samples).  Exact restatement, not an
    approximation: advantages, returns and the advantage normalisation are computed on the WHOLE batch first, exactly
    as the reference does (updater.py:70-98; fp32 scans whatever ``dtype``); every loss term is a mean over samples
    (updater.py:106, 124, 125), so Loss = sum over chunks of (sum over the chunk's samples) / N and the gradients of
    the chunks add up (``.grad`` accumulates).  BPTT unrolls each slot from its own first h_state (updater.py:139-169),
    so slots are independent.  ``dtype=torch.float64`` evaluates the forward / backward in double (the net must hold
    double parameters).  Not supported (asserted): use_nstep_rets, norm_returns -- their targets depend on the forward.

    Returns (info without the optimiser step, {name: grad}); GradNorm is the fp64 norm of the unclipped gradients."""
    assert not hyps["use_nstep_rets"] and not try_key(hyps, "norm_returns", False)
    R, T = hyps["n_rollouts"], hyps["n_tsteps"]
    N = R * T
    net.req_grads(True)
    for p_ in net.parameters():
        p_.grad = None
    advs = discount(D["deltas"].float().squeeze(), D["dones"].float().squeeze(), hyps["gamma"] * hyps["lambda_"])
    returns = discount(D["rewards"].float().squeeze(), D["dones"].float().squeeze(), hyps["gamma"])
    advs, returns = advs.to(dtype), returns.to(dtype)
    if hyps["norm_advs"]:
        advs = (advs - advs.mean()) / (advs.std() + 1e-6)
    tot = dict(Pi_Loss=0.0, ValLoss=0.0, Entropy=0.0)
    for r0 in range(0, R, chunk_slots):
        r1 = min(R, r0 + chunk_slots)
        sl = slice(r0 * T, r1 * T)
        x = D["states"][sl].to(dtype)
        if "h_states" in D:
            if hyps["use_bptt"]:
                vals, logits = bptt(net, x, D["h_states"][sl].to(dtype), D["dones"][sl].to(dtype),
                                    dict(hyps, n_rollouts=r1 - r0))
            else:
                vals, logits, _ = net(x, D["h_states"][sl].to(dtype))
        else:
            vals, logits = net(x)
        lsm = F.log_softmax(logits, dim=-1)
        acts = D["actions"][sl]
        log_ps = lsm[torch.arange(len(acts)).long(), acts]
        entr = -hyps["entr_coef"] * (lsm * F.softmax(logits, dim=-1)).sum(-1).sum() / N
        pi_loss = hyps["pi_coef"] * -(log_ps * advs[sl]).sum() / N
        val_loss = hyps["val_coef"] * ((vals.squeeze() - returns[sl]) ** 2).sum() / N
        (pi_loss + val_loss - entr).backward()
        for k, v in (("Pi_Loss", pi_loss), ("ValLoss", val_loss), ("Entropy", entr)):
            tot[k] += float(v.detach())
        del x, vals, logits, lsm, log_ps, entr, pi_loss, val_loss
    grads = {n: (None if p_.grad is None else p_.grad.detach().clone()) for n, p_ in net.named_parameters()}
    gn = float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values() if g is not None)))
    for p_ in net.parameters():
        p_.grad = None
    info = dict(Loss=tot["Pi_Loss"] + tot["ValLoss"] - tot["Entropy"], GradNorm=gn, **tot)
    return info, grads


# --------------------------------------------------------------------------
# deterministic fake env used by goldens, tests, smoke and the CPU baseline
# --------------------------------------------------------------------------

class FakeEnv:
    """Deterministic env emitting hash frames (1,H,W), scheduled rewards/dones.

    ``step`` ignores the action except for an action-dependent reward term so
    a wrong action shows up in the rewards.  All schedules are integer
    arithmetic on (env_id, step counter): reproducible anywhere.
    """

    def __init__(self, env_id=0, frame_shape=(1, 84, 84), rew_period=7, done_period=23, binary=True):
        self.env_id, self.shape = env_id, tuple(frame_shape)
        self.rew_period, self.done_period, self.binary = rew_period, done_period, binary
        self.t = 0
        self.n_resets = 0

    def _frame(self):
        return formula_frames(1, self.shape, seed=self.env_id * 100003 + self.t * 17 + self.n_resets * 7919,
                              binary=self.binary)[0].astype(np.float64)

    def reset(self):
        self.n_resets += 1
        return self._frame()

    def step(self, action):
        self.t += 1
        k = self.t + 3 * self.env_id
        rew = 0.0
        if k % self.rew_period == 0:
            rew = 1.0 if (k // self.rew_period + int(action)) % 2 == 0 else -1.0
        done = (k % self.done_period == 0)
        return self._frame(), rew, done, {}
