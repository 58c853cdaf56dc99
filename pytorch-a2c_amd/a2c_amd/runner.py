"""Drop-in for the reference's ``a2c/runner.py`` (``SequentialEnvironment``, ``Runner``,
``StatsRunner``), re-designed for one GPU process driving MANY envs in lock-step.

Reference (runner.py:109-248): one OS process per env, batch-1 forwards, every element of the
rollout buffers written one at a time across the host/device boundary.  Here ONE ``Runner``
owns all ``n_envs`` environments: per time step it runs a single batched forward over the B
envs, samples all actions on the device, steps the envs on the host (env stepping stays on the
CPU, frames go up through pinned staging buffers with async copies), and a handful of kernels
write the step's rows of the rollout-major buffers (frame stack -> ``states``, rewards, dones,
TD deltas, ``h_states``).  The ``shared_data`` layout (training.py:88-101) and the gate/stop
queue protocol are unchanged, so ``Updater`` and a ``train()``-style driver see the same data.

An env pool may instead be *device resident* (``DeviceEnvPool`` protocol: frames / rewards /
dones already in HBM, e.g. the synthetic benchmark env); then a whole n_tsteps rollout is
enqueued without a single host synchronisation and can be captured into one hipGraph.
"""
import os
import time
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .utils import cuda_if, next_state, sample_action, try_key


class SequentialEnvironment:
    """gym adapter with the reference's surface (runner.py:14-106).  ``gym`` is imported lazily;
    ``env_fn`` (optional, not in the reference) lets callers inject any object with
    ``reset()``/``step(a)`` returning raw observations."""

    def __init__(self, env_type, preprocessor, seed=time.time(), float_params=dict(), env_fn=None, **kwargs):
        self.env_type, self.preprocessor, self.seed, self.float_params = env_type, preprocessor, seed, float_params
        if env_fn is not None:
            self.env = env_fn()
        else:
            try:
                import gym
                self.env = gym.make(env_type)
                self.env.seed(self.seed)
            except Exception:
                raise NotImplementedError("Unity compatibility not implemented")
        self.is_gym = True
        self.raw_shape = self.env.reset().shape
        self.is_discrete = hasattr(self.env.action_space, "n")
        self.n = self.env.action_space.n if self.is_discrete else self.env.action_space.shape[0]

    def prep_obs(self, obs):
        return self.preprocessor(obs)

    def reset(self):
        return self.prep_obs(self.env.reset())

    def step(self, action):
        obs, rew, done, info = self.env.step(action)
        return self.prep_obs(obs), rew, done, info

    def render(self):
        return self.env.render()

    def get_action(self, preds, rand_nums=None):
        """softmax + inverse-CDF sample of one (or a batch of) logits rows (runner.py:94-97)."""
        if not (self.is_gym and self.is_discrete):
            raise NotImplementedError
        probs = F.softmax(preds.detach(), dim=-1)
        return int(sample_action(probs, rand_nums).item())


class HostEnvPool:
    """B host environments stepped one after the other on the CPU (the reference's env stepping,
    runner.py:208, stays on the host).  Observations are the already prepped (1,H,W) frames."""

    def __init__(self, envs, frame_shape=None):
        self.envs = list(envs)
        if frame_shape is None:
            # raw envs: one probing reset each, exactly what SequentialEnvironment.__init__ does
            # to read raw_shape (runner.py:45)
            frame_shape = np.asarray(self.envs[0].reset()).shape
            for e in self.envs[1:]:
                e.reset()
        self.frame_shape = tuple(frame_shape)

    def __len__(self):
        return len(self.envs)

    def reset(self, j):
        return np.asarray(self.envs[j].reset())

    def step(self, j, action):
        obs, rew, done, _ = self.envs[j].step(action)
        return np.asarray(obs), float(rew), bool(done)


class Runner:
    """Collects rollouts for ALL envs of this process into ``datas`` (reference signature:
    runner.py:110).  ``datas`` tensors should live on the device (``cuda_if`` them like
    training.py:94-101); ``actions`` may stay a host LongTensor like the reference's."""

    def __init__(self, datas, hyps, gate_q, stop_q, rew_q, env_pool=None, uniform_fn=None):
        self.hyps, self.datas = hyps, datas
        self.gate_q, self.stop_q, self.rew_q = gate_q, stop_q, rew_q
        self.obs_deque = deque(maxlen=hyps["n_frame_stack"])     # kept for API parity (single-env helpers)
        self.env_pool = env_pool
        self.uniform_fn = uniform_fn          # (t, B) -> device tensor (B,) of uniforms; default torch.rand
        self._ready = False

    # ------------------------------------------------------------------ set-up (body of run())
    def _make_pool(self):
        hyps = self.hyps
        n = int(try_key(hyps, "n_envs", 1))
        envs = []
        for j in range(n):
            kw = dict(hyps)
            kw["seed"] = try_key(hyps, "seed", 0) + j
            envs.append(SequentialEnvironment(**kw))
        shape = np.asarray(envs[0].prep_obs(np.zeros(envs[0].raw_shape, dtype=np.uint8))).shape
        return HostEnvPool(envs, frame_shape=shape)

    def start(self, net):
        """Everything Runner.run does before its loop (runner.py:158-168), for all envs."""
        self.net = net
        net._ensure_device()
        dev = net._dev
        hyps = self.hyps
        if self.env_pool is None:
            self.env_pool = self._make_pool()
        pool = self.env_pool
        B = self.B = len(pool)
        C = int(hyps["n_frame_stack"])
        fshape = tuple(pool.frame_shape)          # (1, H, W) or (1, L)
        self.C, self.HW = C, int(np.prod(fshape))
        self.state_shape = (C,) + fshape[1:]
        self.S = C * self.HW
        self.device_pool = hasattr(pool, "device_step")
        f32 = dict(dtype=torch.float32, device=dev)
        self.bookmark = torch.zeros((B, self.S), **f32)                  # state_bookmark of every env
        self.val_prev = torch.zeros(B, **f32)
        self.done_eff = torch.zeros(B, **f32)
        self.act_dev = torch.zeros(B, dtype=torch.int64, device=dev)
        self.h = torch.zeros((B, net.h_size), **f32) if net.is_recurrent else None   # h_bookmark
        self.ep_rew = np.zeros(B)
        if not self.device_pool:
            pin = torch.cuda.is_available()
            mk = lambda *s, dt=torch.float32: (torch.zeros(*s, dtype=dt).pin_memory() if pin else torch.zeros(*s, dtype=dt))
            self.h_frames, self.h_rew, self.h_done, self.h_reset = mk(B, self.HW), mk(B), mk(B), mk(B)
            self.h_act = mk(B, dt=torch.int64)
            self.d_frames = torch.zeros((B, self.HW), **f32)
            self.d_rew, self.d_done, self.d_reset = (torch.zeros(B, **f32) for _ in range(3))
            # initial state: next_state(reset=True) -> [0,..,0, env.reset()] (utils.py:37-42)
            for j in range(B):
                self.h_frames[j] = torch.from_numpy(np.asarray(pool.reset(j), dtype=np.float32).reshape(-1))
            self.d_frames.copy_(self.h_frames, non_blocking=True)
            ones = torch.ones(B, **f32)
            ops.frame_stack_push(self.d_frames, ones, self.bookmark.data_ptr(), self.S, self.bookmark.data_ptr(), self.S,
                                 B, C, self.HW)
        else:
            pool.start(self)
        for p in net.parameters():
            p.requires_grad = False
        self._ready = True

    def run(self, net):
        """Entry point with the reference's protocol: wait on gate_q, roll out, answer on stop_q."""
        self.start(net)
        while True:
            idxs = [self.gate_q.get()]
            while len(idxs) < self.B:
                try:
                    idxs.append(self.gate_q.get_nowait())
                except Exception:
                    break
            self.rollout(net, sorted(idxs), self.hyps)
            for i in idxs:
                self.stop_q.put(i)

    # ------------------------------------------------------------------ the rollout
    def rollout(self, net, idx, hyps):
        """Fill slot(s) ``idx`` (an int like the reference, or a list of slots: slot k of the list
        is played by env k).  Non-contiguous lists are split into contiguous runs."""
        if not self._ready:
            self.start(net)
        idxs = [idx] if isinstance(idx, int) else list(idx)
        j = 0
        while j < len(idxs):
            k = j
            while k + 1 < len(idxs) and idxs[k + 1] == idxs[k] + 1:
                k += 1
            self._rollout_block(net, idxs[j], j, k - j + 1, hyps)
            j = k + 1

    def _uniforms(self, t, B, env0):
        if self.uniform_fn is not None:
            return self.uniform_fn(t, B, env0)
        return torch.rand(B, device=self.net._dev, dtype=torch.float32)

    def _forward(self, net, x_ptr, bstride, B, env0, st, sampler=None):
        if net.is_recurrent:
            return net._fwd(x_ptr, bstride, B, "roll", st, False, h_in=self.h[env0:env0 + B])
        if sampler is not None and getattr(net, "_fused_sampling", False):
            return net._fwd(x_ptr, bstride, B, "roll", st, False, sampler=sampler)
        return net._fwd(x_ptr, bstride, B, "roll", st, False)

    def _rollout_block(self, net, slot0, env0, B, hyps):
        D, pool = self.datas, self.env_pool
        T, S, C, HW = int(hyps["n_tsteps"]), self.S, self.C, self.HW
        gamma = hyps["gamma"]
        pong = "Pong" in hyps["env_type"]
        shift = hyps["action_shift"]
        st = ops.stream()
        net._refresh(st)
        states, rewards, dones, deltas = D["states"], D["rewards"], D["dones"], D["deltas"]
        for name in ("states", "rewards", "dones", "deltas"):
            ops._chk(D[name], name)
        sp = lambda t: states.data_ptr() + 4 * (slot0 * T + t) * S        # states[slot0*T + t]
        bm = self.bookmark[env0:env0 + B]
        val_prev, done_eff = self.val_prev[env0:env0 + B], self.done_eff[env0:env0 + B]
        h = None if self.h is None else self.h[env0:env0 + B]
        acts_host_out = D["actions"] if not D["actions"].is_cuda else None
        # The per-step bookkeeping kernel only feeds later bookkeeping, so it CAN run on a side stream
        # next to the frame-stack kernel (two branches in the hipGraph).  Measured on MI355X the
        # fork/join costs more than the 4.8 us it hides (16.0 -> 17.8 ms per 256x128 epoch), so it is
        # opt-in (A2C_SIDE_STREAM=1).
        main = torch.cuda.current_stream()
        side = None
        if h is None and os.environ.get("A2C_SIDE_STREAM") == "1":
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=net._dev)
            side = self._side
        if h is None and side is None and getattr(net, "_step_supported", lambda: False)():
            return self._rollout_block_fused(net, slot0, env0, B, hyps, sp, bm, val_prev, acts_host_out, st)
        # state of step 0 = the bookmark left by the previous slot (runner.py:190)
        ops.copy_rows(bm.data_ptr(), S, sp(0), T * S, B, S, st)
        for t in range(T):
            if h is not None:                                              # h_states[e] = h (runner.py:201)
                hs = D["h_states"]
                ops.copy_rows(h.data_ptr(), h.shape[1], hs.data_ptr() + 4 * (slot0 * T + t) * h.shape[1],
                              T * h.shape[1], B, h.shape[1], st)
            u = self._uniforms(t, B, env0)
            act = self.act_dev[env0:env0 + B]
            if acts_host_out is None:      # device-resident actions buffer: write it in place
                a_ptr, a_stride = D["actions"].data_ptr() + 8 * (slot0 * T + t), T
            else:
                a_ptr, a_stride = act.data_ptr(), 1
            out = self._forward(net, sp(t), T * S, B, env0, st, sampler=(u, a_ptr, a_stride))
            logits, vals = out["logits"], out["vals"]
            if not out.get("sampled", False):
                ops.softmax_sample(logits, u, a_ptr, a_stride, B, net.output_space, st=st)
            if h is not None:
                ops.copy_rows(out["h"].data_ptr(), h.shape[1], h.data_ptr(), h.shape[1], B, h.shape[1], st)
            if self.device_pool:
                frames, rew, done, reset = pool.device_step(t, env0, B)
            else:
                frames, rew, done, reset = self._host_step(pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift,
                                                           acts_host_out, pong)
            nxt_ptr, nxt_stride = (sp(t + 1), T * S) if t + 1 < T else (bm.data_ptr(), S)
            if h is None and side is None and HW % 4 == 0 and S % 4 == 0 and os.environ.get("A2C_NO_POST_FUSE") != "1":
                # feed-forward net: bookkeeping + next frame stack in ONE launch
                ops.rollout_post(rew, done, vals.data_ptr(), vals.stride(0), val_prev, rewards, dones, deltas, T, t, slot0,
                                 gamma, pong, frames, reset, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
                continue
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ops.rollout_record(rew, done, vals.data_ptr(), vals.stride(0), val_prev, rewards, dones, deltas,
                                       done_eff, None, B, T, t, slot0, gamma, pong, side.cuda_stream)
            else:
                ops.rollout_record(rew, done, vals.data_ptr(), vals.stride(0), val_prev, rewards, dones, deltas, done_eff,
                                   h, B, T, t, slot0, gamma, pong, st)
            # next state (utils.next_state): into states[t+1], or the bookmark after the last step
            nxt_ptr, nxt_stride = (sp(t + 1), T * S) if t + 1 < T else (bm.data_ptr(), S)
            ops.frame_stack_push(frames, reset, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
            if side is not None:
                main.wait_stream(side)     # join before the next forward overwrites the heads buffer
        # bootstrap (runner.py:236-245): value of the state after the last step
        out = self._forward(net, bm.data_ptr(), S, B, env0, st)
        ops.rollout_bootstrap(out["vals"].data_ptr(), out["vals"].stride(0), val_prev, rewards, dones, deltas, B, T,
                              slot0, gamma, st)

    def _rollout_block_fused(self, net, slot0, env0, B, hyps, sp, bm, val_prev, acts_host_out, st):
        """The same slot with ONE launch per step (a2c_a3c_step): launch t writes state t (frame
        stack of the frame env step t-1 returned), records env step t-1, runs the policy and samples
        action t; launch T records step T-1, leaves the bookmark and bootstraps (runner.py:174-248)."""
        D, pool = self.datas, self.env_pool
        T, S = int(hyps["n_tsteps"]), self.S
        gamma, pong, shift = hyps["gamma"], "Pong" in hyps["env_type"], hyps["action_shift"]
        rec = dict(val_prev=val_prev.data_ptr(), rewards=D["rewards"].data_ptr(), dones=D["dones"].data_ptr(),
                   deltas=D["deltas"].data_ptr(), T=T, slot0=slot0, gamma=float(gamma), pong=int(pong))
        frames = rew = done = reset = None
        for t in range(T + 1):
            kw = dict(rec)
            if t == 0:        # state of step 0 = the bookmark left by the previous slot (runner.py:190)
                kw.update(prev=bm.data_ptr(), prev_stride=S, out=sp(0), out_stride=T * S)
            else:
                out_ptr, out_stride = (sp(t), T * S) if t < T else (bm.data_ptr(), S)
                kw.update(prev=sp(t - 1), prev_stride=T * S, frame_new=frames.data_ptr(), reset_mask=reset.data_ptr(),
                          out=out_ptr, out_stride=out_stride, rew=rew.data_ptr(), done=done.data_ptr(), t_rec=t - 1)
            if t == T:
                net._step(B, st, bootstrap=1, **kw)
                break
            u = self._uniforms(t, B, env0)
            act = self.act_dev[env0:env0 + B]
            if acts_host_out is None:
                a_ptr, a_stride = D["actions"].data_ptr() + 8 * (slot0 * T + t), T
            else:
                a_ptr, a_stride = act.data_ptr(), 1
            net._step(B, st, u=u.data_ptr(), actions=a_ptr, act_stride=a_stride, **kw)
            if self.device_pool:
                frames, rew, done, reset = pool.device_step(t, env0, B)
            else:
                frames, rew, done, reset = self._host_step(pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift,
                                                           acts_host_out, pong)
            for name, x in (("frames", frames), ("rew", rew), ("done", done), ("reset", reset)):
                ops._chk(x, name)

    def _host_step(self, pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift, acts_host_out, pong):
        """actions D2H -> env.step on the host -> frames/rewards/dones H2D through pinned buffers."""
        ha = self.h_act[env0:env0 + B]
        if a_stride == 1:
            ha.copy_(act, non_blocking=True)
        else:
            ha.copy_(self.datas["actions"][slot0 * T + t::T][:B], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        hf, hr, hd, hz = (x[env0:env0 + B] for x in (self.h_frames, self.h_rew, self.h_done, self.h_reset))
        for j in range(B):
            a = int(ha[j])
            obs, rew, done = pool.step(env0 + j, a + shift)
            self.ep_rew[env0 + j] += rew
            reset = done
            if pong and rew != 0:
                done = True
            if done and self.rew_q is not None:                            # runner.py:216-217
                self.rew_q.put(.99 * self.rew_q.get() + .01 * self.ep_rew[env0 + j])
            if done:
                self.ep_rew[env0 + j] = 0
            if reset:
                obs = pool.reset(env0 + j)
            hf[j] = torch.from_numpy(np.asarray(obs, dtype=np.float32).reshape(-1))
            hr[j], hd[j], hz[j] = rew, float(reset), float(reset)
            if acts_host_out is not None:
                acts_host_out[(slot0 + j) * T + t] = a
        df, dr, dd, dz = (x[env0:env0 + B] for x in (self.d_frames, self.d_rew, self.d_done, self.d_reset))
        df.copy_(hf, non_blocking=True)
        dr.copy_(hr, non_blocking=True)
        dd.copy_(hd, non_blocking=True)
        dz.copy_(hz, non_blocking=True)
        return df, dr, dd, dz


class StatsRunner:
    """Evaluation rollouts (runner.py:250-314): ``n_test_eps`` episodes of one env with the same
    sampler, run on the device net with batch 1 (not part of the rollout+update metric)."""

    def __init__(self, hyps, env=None):
        self.hyps = hyps
        self.env = env if env is not None else SequentialEnvironment(**hyps)
        self.obs_deque = deque(maxlen=hyps["n_frame_stack"])
        self.n_episodes = try_key(hyps, "n_test_eps", 15)

    def rollout(self, net):
        state = next_state(self.env, self.obs_deque, obs=None, reset=True)
        h = cuda_if(torch.zeros(1, net.h_size)) if net.is_recurrent else None
        ep_rew, ep_count = 0, 0
        with torch.no_grad():
            while ep_count < self.n_episodes:
                x = cuda_if(torch.FloatTensor(state))[None]
                if h is not None:
                    val, logits, h = net(x, h)
                else:
                    val, logits = net(x)
                action = self.env.get_action(logits)
                obs, rew, done, _ = self.env.step(action + self.hyps["action_shift"])
                ep_rew += rew
                reset = done
                if "Pong" in self.hyps["env_type"] and rew != 0:
                    done = True
                if done:
                    ep_count += 1
                    if h is not None:
                        h = cuda_if(torch.zeros(1, net.h_size))
                state = next_state(self.env, self.obs_deque, obs=obs, reset=reset)
        return ep_rew / ep_count
