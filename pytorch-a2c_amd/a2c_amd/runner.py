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

Env pools (``env_pool=``):
  * ``hostpool.ProcessEnvPool`` -- the production path: env worker PROCESSES behind one pinned,
    device-mapped region (include/a2c_hostpool.h), frames travel as uint8 when the preprocessor
    yields uint8.  Two ingest modes (``hyps['ingest']`` or ``Runner(..., ingest=)``):
      "zero-copy"  (A3CModel-shaped nets, uint8 frames) the whole slot is ONE persistent launch
                   (a2c_a3c_rollout): workgroups and env workers hand actions / frames to each other
                   through the pinned region, the host process is not in the loop;
      "memcpy"     (every model) per step: actions D2H -> workers step -> hipMemcpyAsync of the
                   frames block from the pinned region into HBM -> the step kernels.
  * ``HostEnvPool`` -- B env objects stepped serially in this process (tests, tiny runs).
  * a *device resident* pool (``device_step`` protocol: frames / rewards / dones already in HBM,
    e.g. a synthetic tape): a whole n_tsteps rollout is enqueued without a single host
    synchronisation and can be captured into one hipGraph.
"""
import os
import queue
import warnings
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


class _Frames:
    """Where the new frames of one env step are on the device: fp32 (B, HW) or uint8 (B, stride bytes)."""
    __slots__ = ("ptr32", "ptr8", "stride")

    def __init__(self, ptr32=0, ptr8=0, stride=0):
        self.ptr32, self.ptr8, self.stride = ptr32, ptr8, stride


class Runner:
    """Collects rollouts for ALL envs of this process into ``datas`` (reference signature:
    runner.py:110).  ``datas`` tensors should live on the device (``cuda_if`` them like
    training.py:94-101); ``actions`` may stay a host LongTensor like the reference's."""

    def __init__(self, datas, hyps, gate_q, stop_q, rew_q, env_pool=None, uniform_fn=None, ingest=None):
        self.hyps, self.datas = hyps, datas
        self.gate_q, self.stop_q, self.rew_q = gate_q, stop_q, rew_q
        self.obs_deque = deque(maxlen=hyps["n_frame_stack"])     # kept for API parity (single-env helpers)
        self.env_pool = env_pool
        self.uniform_fn = uniform_fn          # (t, B) -> device tensor (B,) of uniforms; default torch.rand
        self.ingest = ingest or try_key(hyps, "ingest", None)     # None: zero-copy when it applies, else memcpy
        self._ready = False
        self.error = None                     # exception that ended run() (training.train re-raises it)

    # ------------------------------------------------------------------ set-up (body of run())
    def _make_pool(self):
        """n_envs gym envs like the reference's n_envs processes (training.py:109-121): worker processes
        behind the pinned pool region; ``hyps['env_pool'] == 'serial'`` keeps them in this process."""
        hyps = self.hyps
        n = int(try_key(hyps, "n_envs", 1))
        kws = []
        for j in range(n):
            kw = {k: v for k, v in hyps.items() if k not in ("seed",)}
            kw["seed"] = try_key(hyps, "seed", 0) + j
            kws.append(kw)
        if try_key(hyps, "env_pool", "process") == "serial":
            envs = [SequentialEnvironment(**kw) for kw in kws]
            shape = np.asarray(envs[0].prep_obs(np.zeros(envs[0].raw_shape, dtype=np.uint8))).shape
            return HostEnvPool(envs, frame_shape=shape)
        from .hostpool import ProcessEnvPool
        return ProcessEnvPool(SequentialEnvironment, n, env_kwargs=kws, n_workers=try_key(hyps, "n_env_workers", None),
                              action_shift=hyps["action_shift"], pong="Pong" in hyps["env_type"],
                              rew_ema0=-1.0,
                              frame_bits=bool(try_key(hyps, "frame_bits", try_key(hyps, "prep_fxn", None) == "pong_prep")))

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
        # hyps['device_prep'] = "pong_prep" / "breakout_prep" (SURVEY.md 8 row f4): the envs hand on RAW (H, W, C) uint8
        # frames (host preprocessor = null_prep) and a2c_frame_prep_u8 crops / strides / binarises them on the device
        self.dev_prep = try_key(hyps, "device_prep", None)
        if self.dev_prep:
            if self.dev_prep not in ops.PREP_SPECS or len(fshape) < 3:
                raise ValueError("device_prep: 'pong_prep' or 'breakout_prep' on raw (H, W, C) frames")
            self.raw_dims = tuple(int(v) for v in fshape[-3:])
            fshape = ops.prep_out_shape(self.dev_prep, self.raw_dims[0], self.raw_dims[1])
        self.C, self.HW = C, int(np.prod(fshape))
        self.state_shape = (C,) + fshape[1:]
        self.S = C * self.HW
        self.device_pool = hasattr(pool, "device_step")
        self.proc_pool = hasattr(pool, "post_actions")
        f32 = dict(dtype=torch.float32, device=dev)
        self.bookmark = torch.zeros((B, self.S), **f32)                  # state_bookmark of every env
        self.val_prev = torch.zeros(B, **f32)
        self.done_eff = torch.zeros(B, **f32)
        self.act_dev = torch.zeros(B, dtype=torch.int64, device=dev)
        self.h = torch.zeros((B, net.h_size), **f32) if net.is_recurrent else None   # h_bookmark
        self.ep_rew = np.zeros(B)
        pin = torch.cuda.is_available()
        mk = lambda *s, dt=torch.float32: (torch.zeros(*s, dtype=dt).pin_memory() if pin else torch.zeros(*s, dtype=dt))
        ones = torch.ones(B, **f32)
        if self.proc_pool:
            # worker processes behind the pinned region: frame 0 of every env (its env.reset()) is already there
            pool.start()
            self.h_rew, self.h_done, self.h_act = mk(B), mk(B), mk(B, dt=torch.int64)
            self.np_rew, self.np_done, self.np_act = self.h_rew.numpy(), self.h_done.numpy(), self.h_act.numpy()
            self.d_rew, self.d_done = torch.zeros(B, **f32), torch.zeros(B, **f32)
            self.u8 = pool.frame_dtype == np.uint8
            self.bits = bool(getattr(pool, "frame_bits", False))      # packed transport: one bit per pixel over the link
            self.fstride = int(pool.header.frame_stride)              # bytes between the envs' slots in the pinned region
            if self.bits and self.HW % 16:
                raise ValueError("frame_bits transport: the frame size must be a multiple of 16 pixels")
            if self.dev_prep and (not self.u8 or self.bits or self.fstride % 16):
                raise ValueError("device_prep needs a uint8 (not packed) raw-frame pool")
            # HBM staging of one env step's frames: uint8 pixels (expanded from the packed slots on the way in), or fp32
            self.dstride = -(-self.HW // 16) * 16 if (self.bits or self.dev_prep) else self.fstride
            self.d_frames = torch.zeros((B, self.dstride), dtype=torch.uint8, device=dev)
            self.d_raw = torch.zeros((B, self.fstride), dtype=torch.uint8, device=dev) if self.dev_prep else None
            self.d_packed = torch.zeros((B, self.fstride), dtype=torch.uint8, device=dev) if self.bits else None
            # fp32 frames whose size is not a multiple of 16 B sit `fstride` bytes apart; the fp32 frame-stack kernels
            # read dense (B, HW) rows: compact them with one strided row copy
            self.d_dense = torch.zeros((B, self.HW), **f32) if (not self.u8 and self.fstride != 4 * self.HW) else None
            self.rollout_err = torch.zeros(1, dtype=torch.int32, device=dev)
            pool.set_phase(1)
            pool.wait_frames(pool.seq_start)
            fr = self._pool_frames_h2d(pool, 0, B, ops.stream())
            if fr.ptr8:
                ops.frame_stack_push_u8(fr.ptr8, fr.stride, ones, self.bookmark.data_ptr(), self.S, self.bookmark.data_ptr(),
                                        self.S, B, C, self.HW)
            else:
                ops.frame_stack_push(_Ptr(fr.ptr32), ones, self.bookmark.data_ptr(), self.S, self.bookmark.data_ptr(), self.S,
                                     B, C, self.HW)
            torch.cuda.current_stream().synchronize()
        elif not self.device_pool:
            self.h_frames, self.h_rew, self.h_done, self.h_reset = mk(B, self.HW), mk(B), mk(B), mk(B)
            self.h_act = mk(B, dt=torch.int64)
            self.np_frames, self.np_rew, self.np_done, self.np_act = (x.numpy() for x in (self.h_frames, self.h_rew,
                                                                                             self.h_done, self.h_act))
            self.d_frames = torch.zeros((B, self.HW), **f32)
            self.d_rew, self.d_done, self.d_reset = (torch.zeros(B, **f32) for _ in range(3))
            # initial state: next_state(reset=True) -> [0,..,0, env.reset()] (utils.py:37-42)
            for j in range(B):
                self.np_frames[j] = np.asarray(pool.reset(j), dtype=np.float32).reshape(-1)
            self.d_frames.copy_(self.h_frames, non_blocking=True)
            ops.frame_stack_push(self.d_frames, ones, self.bookmark.data_ptr(), self.S, self.bookmark.data_ptr(), self.S,
                                 B, C, self.HW)
        else:
            pool.start(self)
        for p in net.parameters():
            p.requires_grad = False
        self._ready = True

    def run(self, net):
        """Entry point with the reference's protocol: wait on gate_q, roll out, answer on stop_q.
        An exception ends the loop, is kept in ``self.error`` and is answered with ``None`` tokens on
        stop_q so that the consumer wakes up (the reference's dead runner process deadlocks its
        ``stop_q.get()`` forever)."""
        try:
            self.start(net)
            while True:
                idxs = [self.gate_q.get()]
                if idxs[0] is None:        # shutdown token
                    return
                # the driver re-opens the gate with ALL n_rollouts tokens in one go (training.py:122-123, 174-175): take
                # the whole epoch, so that the slot -> env assignment below is the same every epoch.  Fewer tokens
                # (a caller that trickles them) still work: whatever arrived is played in lock-step rounds.
                n_tok = int(try_key(self.hyps, "n_rollouts", None) or self.B)
                stop = False
                # The rest of the epoch's tokens: wait for them (a partial batch would be played with a different
                # slot -> env mapping, which defeats the activation stash and captures a second set of slot graphs) --
                # but never forever: after hyps["gate_timeout_s"] without a further token (default 5 s; 0.05 s when the
                # caller did not say how many slots an epoch has, i.e. no "n_rollouts" key) whatever arrived is played,
                # so a caller that trickles fewer tokens, or two Runners sharing one gate_q, make progress and their
                # stop_q consumers wake up.
                gate_to = try_key(self.hyps, "gate_timeout_s", None)
                if gate_to is None:
                    gate_to = 5.0 if try_key(self.hyps, "n_rollouts", None) else 0.05
                while len(idxs) < n_tok:
                    try:
                        tok = self.gate_q.get(timeout=gate_to)
                    except queue.Empty:
                        # ONE Runner per gate_q is what performs: Runners sharing a gate_q (the reference's topology) each
                        # take part of the epoch's tokens, stall here once per epoch and play partial batches (which also
                        # re-capture slot graphs and defeat the activation stash).  Say so instead of stalling silently.
                        if not getattr(self, "_warned_partial", False):
                            self._warned_partial = True
                            warnings.warn("a2c_amd.Runner: %d of %d gate tokens after %.2f s -- playing a partial batch; use ONE "
                                          "Runner per gate_q (or set hyps['gate_timeout_s'])" % (len(idxs), n_tok, gate_to),
                                          RuntimeWarning)
                        break
                    if tok is None:
                        stop = True
                        break
                    idxs.append(tok)
                self.rollout(net, sorted(idxs), self.hyps)
                self.finish()
                for i in idxs:
                    self.stop_q.put(i)
                if stop:
                    return
        except BaseException as e:      # noqa: BLE001
            self.error = e
            for _ in range(int(try_key(self.hyps, "n_rollouts", 1))):
                try:
                    self.stop_q.put_nowait(None)
                except Exception:      # noqa: BLE001
                    break
            raise

    def finish(self):
        """Wait for the enqueued rollout work, surface a host-handshake timeout of the persistent kernel,
        publish the episode-reward EMA the env workers kept (runner.py:216) and let the workers idle."""
        torch.cuda.current_stream().synchronize()
        if getattr(self, "proc_pool", False):
            if not (self.env_pool.dev_phase and self.datas["actions"].is_cuda):
                self.env_pool.set_phase(0)
            self.check()
            if self.rew_q is not None:
                self.rew_q.get()
                self.rew_q.put(self.env_pool.rew_ema())

    def check(self):
        if getattr(self, "proc_pool", False) and int(self.rollout_err.item()):
            self.env_pool._check_workers()
            raise TimeoutError("a2c_a3c_rollout: an env worker did not answer within the time-out")

    def close(self):
        if getattr(self, "proc_pool", False):
            self.env_pool.close()

    # ------------------------------------------------------------------ the rollout
    def rollout(self, net, idx, hyps):
        """Fill slot(s) ``idx``: an int like the reference (runner.py:174), or a list of slots of ANY length.  The
        list is played in lock-step ROUNDS of at most n_envs slots: slot k of a round is played by env k (the
        reference hands slots to whichever of its n_envs runner processes is free, runner.py:169-172; n_rollouts
        need not be a multiple of n_envs -- its shipped hyperparams.json has 45 / 11).  Inside a round, runs of
        contiguous slots whose envs have taken the same number of steps go through ONE batched block."""
        if not self._ready:
            self.start(net)
        idxs = [idx] if isinstance(idx, int) else list(idx)
        N = self.datas["states"].shape[0]
        T_ = int(hyps["n_tsteps"])
        # rows a lazy rollout (hyps['lazy_states']) left in the single-frame store only: a partial refill keeps the others
        if getattr(self, "_states_stale", False):
            if idxs == list(range(N // T_)):
                self._states_stale = False
            else:
                self.materialize_states()
        # whatever an earlier rollout stashed in the net describes states this call overwrites
        net._stash = None
        net._stash_frames = None
        net._stash_lm = False
        self._lm_written = False
        if hasattr(net, "_cells_done"):
            net._cells_done = -1
        # a call that fills EVERY row of the rollout buffer: the step kernels may stash the conv activations
        # of each state for the update that follows (same weights, same states)
        self._stash_bufs = None
        self._frames_written = None
        if idxs == list(range(N // T_)) and N % T_ == 0:
            self._stash_bufs = net.stash_rows(self.datas["states"], N, T=T_)
        stash_all = self._stash_bufs is not None
        seqs = self.env_pool.seq_env if self.proc_pool else None
        # the env workers spin on their cmd granules only while a rollout is running: the phase word of the pool is
        # written FROM THE STREAM (in order with the rollout kernels -- the host may be enqueueing this rollout while the
        # previous update still runs), by the host when the region is not device mapped
        dev_phase = self.env_pool.dev_phase if (self.proc_pool and self.datas["actions"].is_cuda) else 0
        if dev_phase:
            ops.store_u32_system(dev_phase, 1)
        elif self.proc_pool:
            self.env_pool.set_phase(1)
        try:
            for r0 in range(0, len(idxs), self.B):
                rnd = idxs[r0:r0 + self.B]
                j = 0
                while j < len(rnd):
                    k = j
                    while k + 1 < len(rnd) and rnd[k + 1] == rnd[k] + 1 and (seqs is None or seqs[k + 1] == seqs[j]):
                        k += 1
                    self._rollout_block(net, rnd[j], j, k - j + 1, hyps)
                    stash_all = stash_all and self._stash_used
                    if self.proc_pool:
                        self.env_pool.advance(j, k - j + 1, T_)
                    j = k + 1
        except BaseException:
            # an env time-out, a lock-step violation or a failed capture must not leave the workers spinning on their cmd
            # granules until close(): back to the sleeping phase from the HOST (the stream may be unusable)
            if self.proc_pool:
                try:
                    self.env_pool.set_phase(0)
                except Exception:      # noqa: BLE001
                    pass
            raise
        if dev_phase:       # ... the last env step has been played: the workers sleep-poll through the update
            ops.store_u32_system(dev_phase, 0)
        if stash_all:
            if getattr(self, "_lm_written", False):      # (A3CModel, ring kernel: lane masks of a1 beside the stash)
                net.stash_commit(self.datas["states"], N, frames=getattr(self, "_frames_written", None), lanemask=True)
            else:
                net.stash_commit(self.datas["states"], N, frames=getattr(self, "_frames_written", None))

    def _uniforms(self, t, B, env0):
        if self.uniform_fn is not None:
            return self.uniform_fn(t, B, env0)
        return torch.rand(B, device=self.net._dev, dtype=torch.float32)

    def _forward(self, net, x_ptr, bstride, B, env0, st, sampler=None, stash=None):
        """stash = (bufs, row0, row_stride): conv-stack nets write this step's activations into the update's buffers"""
        kw = dict(stash=stash) if (stash is not None and isinstance(stash[0], list)) else {}
        if sampler is not None and getattr(net, "_fused_sampling", False):
            kw["sampler"] = sampler
        if net.is_recurrent:
            return net._fwd(x_ptr, bstride, B, "roll", st, False, h_in=self.h[env0:env0 + B], **kw)
        return net._fwd(x_ptr, bstride, B, "roll", st, False, **kw)

    def _zero_copy_ok(self, net):
        """the persistent one-launch rollout applies: A3CModel-shaped net, process pool with uint8 frames"""
        if not self.proc_pool or self.h is not None or self.ingest in ("memcpy", "relay") or getattr(self, "dev_prep", None):
            return False
        ok = (getattr(net, "_step_supported", lambda: False)() and self.u8 and self.HW % 16 == 0 and self.HW <= 8192
              and self.env_pool.dev_ptr != 0)
        if self.ingest == "zero-copy" and not ok:
            raise ValueError("ingest='zero-copy' needs an A3CModel-shaped net (a2c_a3c_step_supported) and a registered "
                             "process env pool with uint8 frames")
        return ok

    def _rollout_block(self, net, slot0, env0, B, hyps):
        D, pool = self.datas, self.env_pool
        T, S, C, HW = int(hyps["n_tsteps"]), self.S, self.C, self.HW
        gamma = hyps["gamma"]
        pong = "Pong" in hyps["env_type"]
        shift = hyps["action_shift"]
        st = ops.stream()
        if hasattr(net, "_set_rollout_batch"):
            net._set_rollout_batch(B)
        net._refresh(st)
        states, rewards, dones, deltas = D["states"], D["rewards"], D["dones"], D["deltas"]
        for name in ("states", "rewards", "dones", "deltas"):
            ops._chk(D[name], name)
        sp = lambda t: states.data_ptr() + 4 * (slot0 * T + t) * S        # states[slot0*T + t]
        bm = self.bookmark[env0:env0 + B]
        val_prev, done_eff = self.val_prev[env0:env0 + B], self.done_eff[env0:env0 + B]
        h = None if self.h is None else self.h[env0:env0 + B]
        acts_host_out = D["actions"] if not D["actions"].is_cuda else None
        self._stash_used = False
        if self._zero_copy_ok(net):
            return self._rollout_block_persistent(net, slot0, env0, B, hyps, bm, val_prev, acts_host_out, st)
        # The per-step bookkeeping kernel only feeds later bookkeeping, so it CAN run on a side stream
        # next to the frame-stack kernel (two branches in the hipGraph).  Measured on MI355X the
        # fork/join costs more than the 4.8 us it hides (16.0 -> 17.8 ms per 256x128 epoch), so it is
        # opt-in (A2C_SIDE_STREAM=1).
        if not self.device_pool and os.environ.get("A2C_NO_STEP_GRAPHS") != "1":
            return self._rollout_block_segmented(net, slot0, env0, B, hyps, sp, bm, val_prev, done_eff, h, acts_host_out)
        self._fstore_ok = False          # this slot is played by a path that does not keep the frame store: it goes stale
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
            if h is not None and out["h"].data_ptr() != h.data_ptr():      # (the GRU models update h in place)
                ops.copy_rows(out["h"].data_ptr(), h.shape[1], h.data_ptr(), h.shape[1], B, h.shape[1], st)
            fr, rew, done, reset = self._env_step(pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift, acts_host_out, pong)
            nxt_ptr, nxt_stride = (sp(t + 1), T * S) if t + 1 < T else (bm.data_ptr(), S)
            if h is None and side is None and HW % 4 == 0 and S % 4 == 0 and os.environ.get("A2C_NO_POST_FUSE") != "1":
                # feed-forward net: bookkeeping + next frame stack in ONE launch
                if fr.ptr8:
                    ops.rollout_post_u8(rew, done, vals.data_ptr(), vals.stride(0), val_prev, rewards, dones, deltas, T, t,
                                        slot0, gamma, pong, fr.ptr8, fr.stride, reset, sp(t), T * S, nxt_ptr, nxt_stride, B,
                                        C, HW, st)
                else:
                    ops.rollout_post(rew, done, vals.data_ptr(), vals.stride(0), val_prev, rewards, dones, deltas, T, t,
                                     slot0, gamma, pong, _Ptr(fr.ptr32), reset, sp(t), T * S, nxt_ptr, nxt_stride, B, C,
                                     HW, st)
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
            if fr.ptr8:
                ops.frame_stack_push_u8(fr.ptr8, fr.stride, reset, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
            else:
                ops.frame_stack_push(_Ptr(fr.ptr32), reset, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
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
        fr = rew = done = reset = None
        stash = self._stash_bufs
        self._stash_used = stash is not None
        for t in range(T + 1):
            kw = dict(rec)
            if stash is not None and t < T:      # conv activations of state (slot, t) -> row slot*T + t of the update's buffers
                n1, n2 = stash[0][0].numel(), stash[1][0].numel()
                kw.update(a1_out=stash[0].data_ptr() + 4 * (slot0 * T + t) * n1, a1_stride=T * n1,
                          a2_out=stash[1].data_ptr() + 4 * (slot0 * T + t) * n2, a2_stride=T * n2,
                          heads_out=stash[2].data_ptr() + 4 * (slot0 * T + t) * stash[2].stride(0),
                          heads_out_stride=T * stash[2].stride(0))
            if t == 0:        # state of step 0 = the bookmark left by the previous slot (runner.py:190)
                kw.update(prev=bm.data_ptr(), prev_stride=S, out=sp(0), out_stride=T * S)
            else:
                out_ptr, out_stride = (sp(t), T * S) if t < T else (bm.data_ptr(), S)
                kw.update(prev=sp(t - 1), prev_stride=T * S, reset_mask=reset.data_ptr(),
                          out=out_ptr, out_stride=out_stride, rew=rew.data_ptr(), done=done.data_ptr(), t_rec=t - 1)
                if fr.ptr8:
                    kw.update(frame_u8=fr.ptr8, frame_stride=fr.stride)
                else:
                    kw.update(frame_new=fr.ptr32)
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
            fr, rew, done, reset = self._env_step(pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift, acts_host_out, pong)
            for name, x in (("rew", rew), ("done", done), ("reset", reset)):
                ops._chk(x, name)

    # ------------------------------------------------------------------ host pools: one hipGraph per time step
    def _rollout_block_segmented(self, net, slot0, env0, B, hyps, sp, bm, val_prev, done_eff, h, acts_host_out):
        """Host env pools (memcpy ingest): the slot is T+1 device SEGMENTS separated by the one host hand-off of
        each env step.  Segment k = [H2D of what env step k-1 returned + its bookkeeping + frame stack] followed by
        [forward + sampling of state k + D2H of the actions] (k < T) or by the bootstrap (k == T).  A segment has
        no host dependency inside, so each one is captured into a hipGraph the first time it runs (all pointers --
        rollout buffer rows, workspaces, the pinned pool region, the slot's uniforms buffer -- are stable across
        epochs) and replayed afterwards: ONE launch per env step instead of 5-30 (runner.py:198-232 per step)."""
        D, pool = self.datas, self.env_pool
        T, S = int(hyps["n_tsteps"]), self.S
        shift, pong = hyps["action_shift"], "Pong" in hyps["env_type"]
        dev = net._dev
        # the slot's uniforms in ONE persistent buffer (rows are what the graph nodes read)
        if getattr(self, "_u_buf", None) is None or self._u_buf.shape != (T, self.B):
            self._u_buf = torch.zeros((T, self.B), dtype=torch.float32, device=dev)
        ub = self._u_buf
        if self.uniform_fn is not None:
            for t in range(T):
                ub[t, env0:env0 + B].copy_(self.uniform_fn(t, B, env0).reshape(B))
        else:
            ub[:, env0:env0 + B] = torch.rand((T, B), device=dev, dtype=torch.float32)
        # device relay (a2c_pool_publish_actions / a2c_pool_ingest): the segments hand actions and frames to the env
        # workers through the pinned region themselves, so the T+1 launches queue back to back with no host in between
        relay = (self.proc_pool and pool.dev_ptr != 0 and acts_host_out is None and self.ingest != "memcpy"
                 and self.fstride % 16 == 0 and os.environ.get("A2C_NO_RELAY") != "1")
        if self.ingest == "relay" and not relay:
            raise ValueError("ingest='relay' needs a registered process/thread env pool and a device-resident "
                             "datas['actions'] tensor")
        if relay:
            if getattr(self, "_seq_dev", None) is None:
                self._seq_dev = torch.zeros(1, dtype=torch.int32, device=dev)
                self._seq_pin = torch.zeros(64, dtype=torch.int32).pin_memory()      # ring: one entry per block in flight
                self._seq_i = 0
            i = self._seq_i % 64
            if i == 0 and self._seq_i:
                torch.cuda.current_stream().synchronize()       # the ring wraps: every earlier copy has been consumed
            self._seq_i += 1
            s32 = pool.seq_of(env0, B) & 0xffffffff               # the granules carry the step number modulo 2^32
            self._seq_pin[i] = s32 - (1 << 32) if s32 >= (1 << 31) else s32
            self._seq_dev.copy_(self._seq_pin[i:i + 1], non_blocking=True)
        stash = self._stash_bufs
        fused = h is None and getattr(net, "_step_supported", lambda: False)()
        if stash is not None and isinstance(stash, list) == fused:      # tuple (a1, a2): step kernel; list: conv-stack nets
            stash = None
        self._stash_used = stash is not None
        # Single-frame uint8 store (opt-in hyps['frame_store']): the ingest writes each new frame
        # straight into frames[slot][k+3], the first conv layer stacks its 4 planes on load, the fp32 `states` rows are
        # expanded from the store (per step, or on demand with hyps['lazy_states']) -- conv-stack nets behind the relay
        fs = None
        if relay and not fused and self.u8 and self.C == 4 and getattr(net, "_cl", None) and net._cl[0].frames_ok:
            fs = self._frame_store(T, D["states"].shape[0] // T, dev)
        if fs is None:
            self._fstore_ok = False      # this slot does not keep the store: it goes stale
        lazy = fs is not None and bool(try_key(hyps, "lazy_states", False))
        ctx = dict(net=net, slot0=slot0, env0=env0, B=B, T=T, hyps=hyps, sp=sp, bm=bm, val_prev=val_prev, done_eff=done_eff,
                   h=h, acts_host_out=acts_host_out, fused=fused, stash=stash, ub=ub, relay=relay, fs=fs, lazy=lazy)
        if fs is not None:
            if self._stash_used:
                self._frames_written = (fs[0], fs[1], T)
            if lazy:
                self._states_stale = True
                net._materialize_states = self.materialize_states
        graphs = None
        if try_key(hyps, "rollout_graphs", True) and torch.cuda.is_available():
            key = (id(net), slot0, env0, B, T, fused, stash is not None, acts_host_out is None, pong, relay, float(hyps["gamma"]),
                   fs is not None, lazy, tuple(D[k].data_ptr() for k in sorted(D) if D[k].is_cuda))
            cache = self.__dict__.setdefault("_seg_graphs", {})
            graphs = cache.setdefault(key, [None] * (T + 1))
        if graphs is not None and relay and os.environ.get("A2C_NO_SLOT_GRAPH") != "1":
            # device relay: nothing happens on the host between the segments, so the WHOLE slot (T+1 segments, a few
            # thousand kernel nodes) is ONE hipGraph -- first rollout eager (tuners), second captured, then replayed
            whole = cache.setdefault(key + ("slot",), [None])
            if whole[0] is None:
                whole[0] = "warm"
                for k in range(T + 1):
                    self._segment(k, ctx)
            else:
                if whole[0] == "warm":
                    try:
                        g = torch.cuda.CUDAGraph()
                        with ops.graph_capture(g):
                            for k in range(T + 1):
                                self._segment(k, ctx)
                        whole[0] = g
                    except Exception:      # noqa: BLE001 -- fall back to one graph per segment
                        torch.cuda.synchronize()
                        whole[0] = False
                if whole[0]:
                    whole[0].replay()
            if whole[0] is not False:
                self._mark_cells_done(net, stash, h, slot0, T, S, B)
                return
        for k in range(T + 1):
            if graphs is None:
                self._segment(k, ctx)
            else:
                if graphs[k] is None:          # first visit: eager (the conv tile tuners measure on eager calls only)
                    graphs[k] = "warm"
                    self._segment(k, ctx)
                    if k < T and not relay:
                        torch.cuda.current_stream().synchronize()
                        self._host_env_step(pool, env0, B, k, slot0, T, shift, acts_host_out, pong)
                    continue
                if graphs[k] == "warm":
                    try:
                        g = torch.cuda.CUDAGraph()
                        with ops.graph_capture(g):
                            self._segment(k, ctx)
                        graphs[k] = g
                    except Exception:      # noqa: BLE001 -- capture not possible here: run this slot eagerly
                        torch.cuda.synchronize()
                        graphs[k] = False
                if graphs[k]:
                    graphs[k].replay()
                else:
                    self._segment(k, ctx)
            if k < T and not relay:
                torch.cuda.current_stream().synchronize()
                self._host_env_step(pool, env0, B, k, slot0, T, shift, acts_host_out, pong)
        self._mark_cells_done(net, stash, h, slot0, T, S, B)

    def _mark_cells_done(self, net, stash, h, slot0, T, S, B):
        ro = getattr(net, "_roll_outputs", None)
        if ro and stash is not None and h is not None and self.HW % 4 == 0 and S % 4 == 0 and \
                os.environ.get("A2C_NO_FUSED_POST") != "1" and ro((stash, slot0 * T, T), B) is not None:
            net._cells_done = T - 1          # every step of every slot went through the cell stash (eagerly or replayed)

    def _post_frames_args(self, k, c, rew, done):
        """arguments of a2c_rollout_post_frames for env step k - 1 of this segment context (after rew, done; before the stream):
        where segment k - 1's forward left the values of its states and its new hidden rows -- the roll buffers, or rows of the
        update's buffers (GRUModel's cell stash) -- a pure function of (net, k), safe under graph replay"""
        net, slot0, env0, B, T, hyps = c["net"], c["slot0"], c["env0"], c["B"], c["T"], c["hyps"]
        D, h = self.datas, c["h"]
        hb, logits, vals = net._heads("roll", B)
        ro = getattr(net, "_roll_outputs", None)
        fused_post = h is not None and self.HW % 4 == 0 and self.S % 4 == 0 and os.environ.get("A2C_NO_FUSED_POST") != "1"
        prev = ro((c["stash"], slot0 * T + k - 1, T), B) if (ro and c["stash"] is not None and k > 0 and fused_post) else None
        v_ptr, v_ld, h_src = prev if prev else (vals.data_ptr(), vals.stride(0), 0)
        F_, nv_rows, nv_carry = c["fs"]
        hrow = 0
        if h is not None:
            hrow = D["h_states"].data_ptr() + 4 * (slot0 * T + k) * h.shape[1] if k < T else 0
        return (v_ptr, v_ld, c["val_prev"], D["rewards"], D["dones"], D["deltas"], T, k - 1, slot0, hyps["gamma"],
                "Pong" in hyps["env_type"], c["done_eff"], h, hrow, 0 if h is None else T * h.shape[1], h_src, nv_rows,
                nv_carry.data_ptr() + 4 * env0)

    def _segment(self, k, c):
        """device work of segment k (see _rollout_block_segmented); only enqueues, never waits on the host"""
        net, slot0, env0, B, T, hyps = c["net"], c["slot0"], c["env0"], c["B"], c["T"], c["hyps"]
        D, pool, S, C, HW = self.datas, self.env_pool, self.S, self.C, self.HW
        sp, bm, val_prev, done_eff, h = c["sp"], c["bm"], c["val_prev"], c["done_eff"], c["h"]
        gamma, pong = hyps["gamma"], "Pong" in hyps["env_type"]
        rewards, dones, deltas = D["rewards"], D["dones"], D["deltas"]
        st = ops.stream()
        fr = rew = done = None
        if k > 0 and c.get("relay"):       # what env step k-1 returned: waited for and fetched by the device itself
            rew, done = self.d_rew[env0:env0 + B], self.d_done[env0:env0 + B]
            fs, ds = self.fstride, self.dstride
            dst = self.d_frames.data_ptr() + env0 * ds
            if c.get("fs") is not None:      # single-frame store: the new frame lands in frames[slot][k+3], nowhere else
                F_ = c["fs"][0]
                ds = F_.stride(0)
                dst = F_.data_ptr() + slot0 * ds + (k + 3) * HW
            ticks = int(float(try_key(hyps, "env_timeout_s", 20.0)) * 1e8)
            post = self._post_frames_args(k, c, rew, done) if (c.get("fs") is not None and not self.dev_prep
                                                               and os.environ.get("A2C_NO_FUSED_POST_INGEST") != "1") else None
            if post is not None:     # the ingest workgroup of an env also does its bookkeeping (rollout_post_frames) of step k - 1
                ops.pool_ingest_post(self.bits, pool.dev_rec + 8 * env0, pool.dev_frames + env0 * fs, fs, HW if self.bits else fs, B,
                                     self._seq_dev, k, ticks, self.rollout_err, rew, done, dst, ds, *post, st)
                c["_post_done"] = k
            elif self.dev_prep:      # raw frames cross the link; crop / stride / binarise on the device into dst
                raw = self.d_raw.data_ptr() + env0 * fs
                ops.pool_ingest(pool.dev_rec + 8 * env0, pool.dev_frames + env0 * fs, fs, fs, B, self._seq_dev, k,
                                ticks, self.rollout_err, rew, done, raw, fs, st)
                ops.frame_prep_u8(self.dev_prep, raw, fs, *self.raw_dims, dst, ds, B, st)
            elif self.bits:
                ops.pool_ingest_bits(pool.dev_rec + 8 * env0, pool.dev_frames + env0 * fs, fs, HW, B, self._seq_dev, k,
                                     ticks, self.rollout_err, rew, done, dst, ds, st)
            else:
                ops.pool_ingest(pool.dev_rec + 8 * env0, pool.dev_frames + env0 * fs, fs, fs, B, self._seq_dev, k,
                                ticks, self.rollout_err, rew, done, dst, ds, st)
            fr = self._staged_frames(dst, env0, B, st) if c.get("fs") is None else None
        elif k > 0:     # what env step k-1 returned: pinned staging -> HBM
            rew, done = self.d_rew[env0:env0 + B], self.d_done[env0:env0 + B]
            rew.copy_(self.h_rew[env0:env0 + B], non_blocking=True)
            done.copy_(self.h_done[env0:env0 + B], non_blocking=True)
            if self.proc_pool:
                fr = self._pool_frames_h2d(pool, env0, B, st)
            else:
                df = self.d_frames[env0:env0 + B]
                df.copy_(self.h_frames[env0:env0 + B], non_blocking=True)
                fr = _Frames(ptr32=df.data_ptr())
        act = self.act_dev[env0:env0 + B]
        if c["acts_host_out"] is None:
            a_ptr, a_stride = D["actions"].data_ptr() + 8 * (slot0 * T + min(k, T - 1)), T
        else:
            a_ptr, a_stride = act.data_ptr(), 1
        if c["fused"]:      # a2c_a3c_step: record(k-1) + frame stack + forward + sample (+ bootstrap) in ONE launch
            kw = dict(val_prev=val_prev.data_ptr(), rewards=rewards.data_ptr(), dones=dones.data_ptr(), deltas=deltas.data_ptr(),
                      T=T, slot0=slot0, gamma=float(gamma), pong=int(pong))
            if k == 0:
                kw.update(prev=bm.data_ptr(), prev_stride=S, out=sp(0), out_stride=T * S)
            else:
                out_ptr, out_stride = (sp(k), T * S) if k < T else (bm.data_ptr(), S)
                kw.update(prev=sp(k - 1), prev_stride=T * S, reset_mask=done.data_ptr(), out=out_ptr, out_stride=out_stride,
                          rew=rew.data_ptr(), done=done.data_ptr(), t_rec=k - 1)
                kw.update(dict(frame_u8=fr.ptr8, frame_stride=fr.stride) if fr.ptr8 else dict(frame_new=fr.ptr32))
            if c["stash"] is not None and k < T:
                n1, n2 = c["stash"][0][0].numel(), c["stash"][1][0].numel()
                kw.update(a1_out=c["stash"][0].data_ptr() + 4 * (slot0 * T + k) * n1, a1_stride=T * n1,
                          a2_out=c["stash"][1].data_ptr() + 4 * (slot0 * T + k) * n2, a2_stride=T * n2)
                hs = c["stash"][2]
                kw.update(heads_out=hs.data_ptr() + 4 * (slot0 * T + k) * hs.stride(0), heads_out_stride=T * hs.stride(0))
            if k == T:
                net._step(B, st, bootstrap=1, **kw)
                return
            net._step(B, st, u=c["ub"][k, env0:env0 + B].data_ptr(), actions=a_ptr, act_stride=a_stride, **kw)
        else:
            hb, logits, vals = net._heads("roll", B)
            # where segment k-1's forward left the values of its states and its new hidden rows: the roll buffers, or
            # rows of the update's buffers (GRUModel's cell stash) -- a pure function of (net, k), safe under graph replay
            ro = getattr(net, "_roll_outputs", None)
            fused_post = h is not None and HW % 4 == 0 and S % 4 == 0 and os.environ.get("A2C_NO_FUSED_POST") != "1"
            prev = ro((c["stash"], slot0 * T + k - 1, T), B) if (ro and c["stash"] is not None and k > 0 and fused_post) else None
            v_ptr, v_ld, h_src = prev if prev else (vals.data_ptr(), vals.stride(0), 0)
            h_row_done = False
            if c.get("fs") is not None:
                F_, nv_rows, nv_carry = c["fs"]
                ss = F_.stride(0)
                f0 = F_.data_ptr() + slot0 * ss
                nvr, nvc = nv_rows.data_ptr() + 4 * slot0 * T, nv_carry.data_ptr() + 4 * env0
                if k == 0:  # state 0 = the window the previous slot ended with (the bookmark, runner.py:190)
                    ops.frame_store_begin(f0, ss, T, C, HW, nvr, nvc, B, st)
                elif c.get("_post_done") == k:      # ... done by the ingest launch of this segment (a2c_pool_ingest_post)
                    h_row_done = h is not None
                else:       # bookkeeping of env step k-1 (runner.py:212-232); the frame is already in the store
                    pa = self._post_frames_args(k, c, rew, done)
                    ops.rollout_post_frames(rew, done, *pa[:11], B, *pa[11:], st)
                    h_row_done = h is not None
                if k == T:  # the bookmark state stays materialised: any other rollout path can take over from here
                    ops.frames_to_states(f0 + T * HW, ss, nvc, 1, bm.data_ptr(), S, B, 1, C, HW, st)
                    net._frames_src = (f0 + T * HW, ss, 1, nvc, 1)
                else:
                    if not c["lazy"]:   # the reference's fp32 row of `states` (runner.py:199), expanded from the window
                        ops.frames_to_states(f0 + k * HW, ss, nvr + 4 * k, T, sp(k), T * S, B, 1, C, HW, st)
                    net._frames_src = (f0 + k * HW, ss, 1, nvr + 4 * k, T)
            elif k == 0:    # state of step 0 = the bookmark left by the previous slot (runner.py:190)
                ops.copy_rows(bm.data_ptr(), S, sp(0), T * S, B, S, st)
            else:           # bookkeeping of env step k-1 + the next state (utils.next_state)
                t = k - 1
                nxt_ptr, nxt_stride = (sp(k), T * S) if k < T else (bm.data_ptr(), S)
                if h is None and HW % 4 == 0 and S % 4 == 0:
                    if fr.ptr8:
                        ops.rollout_post_u8(rew, done, v_ptr, v_ld, val_prev, rewards, dones, deltas, T, t,
                                            slot0, gamma, pong, fr.ptr8, fr.stride, done, sp(t), T * S, nxt_ptr, nxt_stride, B,
                                            C, HW, st)
                    else:
                        ops.rollout_post(rew, done, v_ptr, v_ld, val_prev, rewards, dones, deltas, T, t,
                                         slot0, gamma, pong, _Ptr(fr.ptr32), done, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
                elif fused_post:
                    # recurrent nets: bookkeeping + frame stack + hidden-state reset + the h_states row in ONE launch
                    hs = D["h_states"]
                    hrow = hs.data_ptr() + 4 * (slot0 * T + k) * h.shape[1] if k < T else 0
                    ops.rollout_post_rec(rew, done, v_ptr, v_ld, val_prev, rewards, dones, deltas, T, t,
                                         slot0, gamma, pong, fr.ptr32 or 0, fr.ptr8 or 0, fr.stride if fr.ptr8 else 0, done,
                                         sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, done_eff, h, hrow, T * h.shape[1],
                                         h_src_ptr=h_src, st=st)
                    h_row_done = True
                else:
                    ops.rollout_record(rew, done, v_ptr, v_ld, val_prev, rewards, dones, deltas, done_eff,
                                       h, B, T, t, slot0, gamma, pong, st)
                    if fr.ptr8:
                        ops.frame_stack_push_u8(fr.ptr8, fr.stride, done, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
                    else:
                        ops.frame_stack_push(_Ptr(fr.ptr32), done, sp(t), T * S, nxt_ptr, nxt_stride, B, C, HW, st)
            if k == T:      # bootstrap (runner.py:236-245): value of the state after the last step
                try:
                    out = self._forward(net, bm.data_ptr(), S, B, env0, st)
                finally:
                    net._frames_src = None
                ops.rollout_bootstrap(out["vals"].data_ptr(), out["vals"].stride(0), val_prev, rewards, dones, deltas, B, T,
                                      slot0, gamma, st)
                return
            if h is not None and not h_row_done:                            # h_states[e] = h (runner.py:201)
                hs = D["h_states"]
                ops.copy_rows(h.data_ptr(), h.shape[1], hs.data_ptr() + 4 * (slot0 * T + k) * h.shape[1], T * h.shape[1], B,
                              h.shape[1], st)
            u = c["ub"][k, env0:env0 + B]
            net._cell_stash_ok = fused_post            # the cell stash needs the fused post kernel (h_src) of the next segment
            try:
                pub = None      # device relay: the heads kernel's sampling thread hands the action to the env worker itself
                if c.get("relay") and os.environ.get("A2C_NO_FUSED_PUBLISH") != "1":
                    pub = (pool.dev_cmd + 8 * env0, self._seq_dev, k)
                out = self._forward(net, sp(k), T * S, B, env0, st, sampler=(u, a_ptr, a_stride, pub),
                                    stash=None if c["stash"] is None else (c["stash"], slot0 * T + k, T))
            finally:
                net._frames_src = None
            if not out.get("sampled", False):
                ops.softmax_sample(out["logits"], u, a_ptr, a_stride, B, net.output_space, st=st)
            if out.get("h_next_src") is not None:      # cell stash: the next segment's post kernel reads h_new from there
                pass
            elif h is not None and out["h"].data_ptr() != h.data_ptr():    # (the GRU models update h in place)
                ops.copy_rows(out["h"].data_ptr(), h.shape[1], h.data_ptr(), h.shape[1], B, h.shape[1], st)
        if c.get("relay"):      # the sampled actions go to the env workers: cmd granules of env step k
            if not c["fused"] and out.get("published", False):
                return
            ops.pool_publish_actions(pool.dev_cmd + 8 * env0, a_ptr, a_stride, B, self._seq_dev, k, st)
            return
        # the sampled actions go to the host (pinned staging) at the tail of the segment
        ha = self.h_act[env0:env0 + B]
        if a_stride == 1:
            ha.copy_(act, non_blocking=True)
        else:
            if getattr(self, "_act_gather", None) is None:
                self._act_gather = torch.zeros(self.B, dtype=torch.int64, device=net._dev)
            ag = self._act_gather[env0:env0 + B]
            ag.copy_(D["actions"][slot0 * T + k::T][:B])
            ha.copy_(ag, non_blocking=True)

    def _host_env_step(self, pool, env0, B, t, slot0, T, shift, acts_host_out, pong):
        """host half of env step t (the segment's D2H of the actions has completed): hand the actions to the envs,
        wait for them, leave rewards / dones (/ frames) in the pinned staging the next segment copies up"""
        na = self.np_act[env0:env0 + B]
        if acts_host_out is not None:
            acts_host_out[slot0 * T + t:(slot0 + B) * T:T] = self.h_act[env0:env0 + B]
        if self.proc_pool:
            k = pool.seq_of(env0, B) + t
            pool.post_actions(na, env0=env0, seq=k)
            pool.wait_frames(k + 1, env0=env0, n=B, timeout=float(try_key(self.hyps, "env_timeout_s", 20.0)))
            pool.unpack(self.np_rew[env0:env0 + B], self.np_done[env0:env0 + B], env0=env0)
            return
        nf, nr, nd = (x[env0:env0 + B] for x in (self.np_frames, self.np_rew, self.np_done))
        for j in range(B):
            obs, rew, done = pool.step(env0 + j, int(na[j]) + shift)
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
            nf[j] = np.asarray(obs, dtype=np.float32).reshape(-1)
            nr[j], nd[j] = rew, float(reset)

    def _rollout_block_persistent(self, net, slot0, env0, B, hyps, bm, val_prev, acts_host_out, st):
        """The whole slot as ONE persistent launch (a2c_a3c_rollout): the workgroups and the env worker
        processes hand actions and uint8 frames to each other through the pinned pool region; nothing
        runs on the host of this process until the launch has finished."""
        D, pool = self.datas, self.env_pool
        T, S = int(hyps["n_tsteps"]), self.S
        dev = net._dev
        if self.uniform_fn is not None:
            u = torch.stack([self.uniform_fn(t, B, env0).reshape(B) for t in range(T)]).contiguous()
        else:
            u = torch.rand((T, B), device=dev, dtype=torch.float32)
        ops._chk(u, "uniforms")
        if acts_host_out is None:
            acts = D["actions"]
        else:
            if getattr(self, "_acts_dev", None) is None or self._acts_dev.numel() != D["actions"].numel():
                self._acts_dev = torch.zeros(D["actions"].numel(), dtype=torch.int64, device=dev)
            acts = self._acts_dev
        C, H, W = net.input_space[-3:]
        hb, _, _ = net._heads("roll", self.B)
        hb = hb[env0:env0 + B]
        P = net.P
        self._u_keep = u          # stays alive until the launch has consumed it
        timeout_s = float(try_key(hyps, "env_timeout_s", 20.0))
        # single-frame uint8 store (row f4): T+4 frames per slot, created with the bookmark in start(); the update's
        # first-layer weight gradient stacks the frames on load instead of reading the 4x duplicated fp32 states
        fs = self._frame_store(T, D["states"].shape[0] // T, dev)
        # only the ring kernel can leave the fp32 rows out (more envs than CUs: interleaved blocks of it, one after the
        # other); the library answers whether THIS call runs it (LDS budget, conv1 output size, weight alignment,
        # A2C_NO_RING / A2C_RING_BLOCKS) -- a shape that does not gets its rows written by the per-step body instead
        ring = ops.a3c_ring_supported(B, C, H, W, net.output_space, P("convs.0.0.weight").data_ptr())
        lazy = fs is not None and bool(try_key(hyps, "lazy_states", False)) and ring
        # ... and only the ring kernel writes the lane masks of the a1 stash rows (the update's conv2 backward-data mask)
        lm = self._stash_bufs[3] if (ring and self._stash_bufs is not None and len(self._stash_bufs) > 3) else None
        mb = self._stash_bufs[4] if (lm is not None and len(self._stash_bufs) > 4) else None
        ops.a3c_rollout(st, B=B, C=C, H=H, W=W, n_actions=net.output_space, states=D["states"].data_ptr(),
                        bookmark=bm.data_ptr(), wfrag1=net._c1.wf.data_ptr(), bias1=P("convs.0.0.bias").data_ptr(),
                        wfrag2=net._c2.wf.data_ptr(), bias2=P("convs.1.0.bias").data_ptr(), Wc=net._Wc.data_ptr(),
                        bc=net._bc.data_ptr(), heads=hb.data_ptr(), ldh=hb.stride(0), u=u.data_ptr(), u_stride=B,
                        actions=acts.data_ptr(), val_prev=val_prev.data_ptr(), rewards=D["rewards"].data_ptr(),
                        dones=D["dones"].data_ptr(), deltas=D["deltas"].data_ptr(), T=T, slot0=slot0,
                        gamma=float(hyps["gamma"]), pong=int("Pong" in hyps["env_type"]), cmd=pool.dev_cmd, rec=pool.dev_rec,
                        frames=pool.dev_frames, frame_stride=self.fstride, frame_bits=int(self.bits),
                        conv1_weight=P("convs.0.0.weight").data_ptr(),
                        seq0=pool.seq_of(env0, B) & 0xffffffff, env0=env0,
                        err=self.rollout_err.data_ptr(), timeout_ticks=int(timeout_s * 1e8),
                        a1_rows=0 if self._stash_bufs is None else self._stash_bufs[0].data_ptr(),
                        a2_rows=0 if self._stash_bufs is None else self._stash_bufs[1].data_ptr(),
                        heads_rows=0 if self._stash_bufs is None else self._stash_bufs[2].data_ptr(),
                        heads_rows_ld=0 if self._stash_bufs is None else self._stash_bufs[2].stride(0),
                        frame_store=0 if fs is None else fs[0].data_ptr(),
                        frame_store_slot_stride=0 if fs is None else fs[0].stride(0),
                        nvalid_rows=0 if fs is None else fs[1].data_ptr(),
                        nvalid_carry=0 if fs is None else fs[2][env0:env0 + B].data_ptr(), states_lazy=int(lazy),
                        tagged=getattr(pool, "dev_tagged", 0), tagged_stride=int(pool.header.tagged_stride),
                        tagged_chunks=int(pool.header.tagged_chunks), a1_lanemask_rows=0 if lm is None else lm.data_ptr(),
                        a2_maskbit_rows=0 if mb is None else mb.data_ptr())
        # every block of the call must have written them (a rollout of several blocks: all ring launches, or none counts)
        self._lm_written = (lm is not None) and (getattr(self, "_lm_written", False) or slot0 == 0)
        if lazy:            # (only after the launch was accepted: a refused call must not leave rows marked stale)
            self._states_stale = True
            net._materialize_states = self.materialize_states
        if fs is not None and self._stash_bufs is not None:
            self._frames_written = (fs[0], fs[1], T)
        self._stash_used = self._stash_bufs is not None
        if acts_host_out is not None:
            acts_host_out[slot0 * T:(slot0 + B) * T].copy_(acts[slot0 * T:(slot0 + B) * T])

    def materialize_states(self):
        """hyps['lazy_states']: the rollout kept one uint8 frame per env step (+ valid-plane counts) and did not write the
        fp32 ``states`` rows; this expands ALL of them (runner.py:199's layout and values, bit for bit) -- called by the
        net when an update needs the rows (no activation stash), by ``finish()`` unless hyps['lazy_states'], or by hand."""
        if not getattr(self, "_states_stale", False):
            return
        F_, nv_rows, _ = self._fstore
        R, T = F_.shape[0], F_.shape[1] - 4
        st = self.datas["states"]
        ops.frames_to_states(F_.data_ptr(), F_.stride(0), nv_rows.data_ptr(), T, st.data_ptr(), T * self.S, R, T, self.C, self.HW)
        self._states_stale = False

    def _frame_store(self, T, R, dev):
        """(frames uint8 (R, T+4, HW), nvalid_rows int32 (R*T,), nvalid_carry int32 (B,)) or None.  OPT-IN
        (hyps['frame_store'] / A2C_FRAME_STORE=1): on MI355X the first-layer weight gradient is matrix-bound, not
        HBM-bound, and runs 10 % SLOWER from the uint8 store (1.41 vs 1.27 ms at N = 32 768: the uint8 -> fp32
        expansion sits in the commit phase all waves wait on) although it reads 4x fewer bytes.  Valid only while
        EVERY env step since start() went through the persistent kernel (it is the only writer): the first slot finds
        frames[:, T+3] = the reset frame and one valid plane; any other rollout path switches the store off."""
        on = os.environ.get("A2C_FRAME_STORE") == "1" or bool(try_key(self.hyps, "frame_store", False))
        if not on or os.environ.get("A2C_NO_FRAME_STORE") == "1" or T < 4 or self.HW % 16 or \
                not getattr(self, "_fstore_ok", True):
            return None
        fs = getattr(self, "_fstore", None)
        if fs is None:
            if (self.env_pool.seq_env != self.env_pool.seq_start).any() or R != self.B:
                self._fstore_ok = False
                return None
            F_ = torch.zeros((R, T + 4, self.HW), dtype=torch.uint8, device=dev)
            F_[:, T + 3] = self.d_frames[:, :self.HW]          # frame 0 of every env (its env.reset()), still in HBM from start()
            fs = self._fstore = (F_, torch.zeros(R * T, dtype=torch.int32, device=dev),
                                 torch.ones(self.B, dtype=torch.int32, device=dev))
        return fs

    # ------------------------------------------------------------------ one env step of B envs
    def _env_step(self, pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift, acts_host_out, pong):
        """-> (_Frames, rew, done, reset) device views of what the envs returned for the actions just sampled"""
        if self.device_pool:
            frames, rew, done, reset = pool.device_step(t, env0, B)
            ops._chk(frames, "frames")
            return _Frames(ptr32=frames.data_ptr()), rew, done, reset
        if self.proc_pool:
            return self._pool_step(pool, act, a_stride, env0, B, t, slot0, T, acts_host_out)
        return self._host_step(pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift, acts_host_out, pong)

    def _actions_to_host(self, act, a_stride, env0, B, t, slot0, T):
        ha = self.h_act[env0:env0 + B]
        if a_stride == 1:
            ha.copy_(act, non_blocking=True)
        else:
            ha.copy_(self.datas["actions"][slot0 * T + t::T][:B], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return ha

    def _staged_frames(self, dst, env0, B, st):
        """_Frames of the B frames staged at ``dst`` (stride self.dstride): uint8 pixels, or fp32 (compacted to dense
        rows first when the pool's slots are padded: frame sizes that are not a multiple of 16 B)"""
        if self.u8:
            return _Frames(ptr8=dst, stride=self.dstride)
        if self.d_dense is None:
            return _Frames(ptr32=dst)
        dense = self.d_dense.data_ptr() + 4 * env0 * self.HW
        ops.copy_rows(dst, self.dstride // 4, dense, self.HW, B, self.HW, st)
        return _Frames(ptr32=dense)

    def _pool_frames_h2d(self, pool, env0, B, st):
        """hipMemcpyAsync of the frames block of envs env0..env0+B from the pinned region into HBM"""
        fs, ds = self.fstride, self.dstride
        dst = self.d_frames.data_ptr() + env0 * ds
        src = pool.region.base + pool.header.off_frames + env0 * fs
        if getattr(self, "dev_prep", None):
            raw = self.d_raw.data_ptr() + env0 * fs
            ops.memcpy_async(raw, src, B * fs, ops.H2D, st)
            ops.frame_prep_u8(self.dev_prep, raw, fs, *self.raw_dims, dst, ds, B, st)
        elif self.bits:     # the packed slots cross the link, one kernel expands them to the uint8 pixels the step kernels read
            pk = self.d_packed.data_ptr() + env0 * fs
            ops.memcpy_async(pk, src, B * fs, ops.H2D, st)
            ops.unpack_bits(pk, fs, dst, ds, B, self.HW, st)
        else:
            ops.memcpy_async(dst, src, B * fs, ops.H2D, st)
        return self._staged_frames(dst, env0, B, st)

    def _pool_step(self, pool, act, a_stride, env0, B, t, slot0, T, acts_host_out):
        """memcpy ingest with the process pool: actions D2H -> the workers step their envs in parallel ->
        frames H2D from the pinned region (uint8 when the pool carries uint8) + rewards / dones."""
        ha = self._actions_to_host(act, a_stride, env0, B, t, slot0, T)
        k = pool.seq_of(env0, B) + t
        na = self.np_act[env0:env0 + B]
        pool.post_actions(na, env0=env0, seq=k)
        if acts_host_out is not None:
            acts_host_out[slot0 * T + t:(slot0 + B) * T:T] = ha
        pool.wait_frames(k + 1, env0=env0, n=B, timeout=float(try_key(self.hyps, "env_timeout_s", 20.0)))
        pool.unpack(self.np_rew[env0:env0 + B], self.np_done[env0:env0 + B], env0=env0)
        st = ops.stream()
        fr = self._pool_frames_h2d(pool, env0, B, st)
        dr, dd = self.d_rew[env0:env0 + B], self.d_done[env0:env0 + B]
        dr.copy_(self.h_rew[env0:env0 + B], non_blocking=True)
        dd.copy_(self.h_done[env0:env0 + B], non_blocking=True)
        return fr, dr, dd, dd

    def _host_step(self, pool, act, a_ptr, a_stride, env0, B, t, slot0, T, shift, acts_host_out, pong):
        """serial in-process pool: actions D2H -> env.step one after the other -> frames/rewards/dones H2D
        through pinned buffers (numpy views of them: no per-element tensor writes)."""
        self._actions_to_host(act, a_stride, env0, B, t, slot0, T)
        na = self.np_act[env0:env0 + B]
        nf, nr, nd = (x[env0:env0 + B] for x in (self.np_frames, self.np_rew, self.np_done))
        for j in range(B):
            a = int(na[j])
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
            nf[j] = np.asarray(obs, dtype=np.float32).reshape(-1)
            nr[j], nd[j] = rew, float(reset)
        if acts_host_out is not None:
            acts_host_out[slot0 * T + t:(slot0 + B) * T:T] = self.h_act[env0:env0 + B]
        df, dr, dd = (x[env0:env0 + B] for x in (self.d_frames, self.d_rew, self.d_done))
        df.copy_(self.h_frames[env0:env0 + B], non_blocking=True)
        dr.copy_(self.h_rew[env0:env0 + B], non_blocking=True)
        dd.copy_(self.h_done[env0:env0 + B], non_blocking=True)
        return _Frames(ptr32=df.data_ptr()), dr, dd, dd


class _Ptr:
    """raw device address with the ``data_ptr()`` the ops wrappers ask for"""
    __slots__ = ("p",)

    def __init__(self, p):
        self.p = int(p)

    def data_ptr(self):
        return self.p


class StatsRunner:
    """Evaluation rollouts (runner.py:250-314) with the training sampler.

    The reference plays ``n_test_eps`` episodes of ONE env one after the other with batch-1 forwards on the main
    process, every epoch (training.py:167) -- with a millisecond-scale update that serial loop would dominate
    ``train()``.  Here the ``n_test_eps`` episodes are played by ``n_test_eps`` envs in LOCK-STEP: one batched
    forward + sampling launch per step for all of them (the one-launch step kernel for A3CModel-shaped nets),
    each env plays exactly one episode and the result is (sum of the episode rewards) / n_test_eps -- the same
    estimator over the same number of episodes.  ``StatsRunner(hyps, env=one_env)`` keeps the reference's serial
    loop (same episodes in the same order as the reference) for callers that own a single env object."""

    def __init__(self, hyps, env=None, envs=None, uniform_fn=None):
        self.hyps = hyps
        self.n_episodes = try_key(hyps, "n_test_eps", 15)
        self.uniform_fn = uniform_fn              # (t, E) -> (E,) device uniforms; default torch.rand
        self.obs_deque = deque(maxlen=hyps["n_frame_stack"])
        self.envs = list(envs) if envs is not None else None
        self.env = env
        if env is None and envs is None:
            seed = try_key(hyps, "seed", 0)
            self.envs = [SequentialEnvironment(**dict({k: v for k, v in hyps.items() if k != "seed"}, seed=seed + 10007 + j))
                         for j in range(self.n_episodes)]
            self.env = self.envs[0]               # the reference's attribute (training.py:61 reads stats_runner.env)

    def rollout(self, net):
        if self.envs is None:
            return self._rollout_serial(net)
        return self._rollout_batched(net)

    # ---- E envs, one episode each, in lock-step on the device
    def _rollout_batched(self, net, max_steps=None):
        hyps, envs = self.hyps, self.envs
        E = len(envs)
        net._ensure_device()
        dev = net._dev
        st = ops.stream()
        net._refresh(st)
        C = int(hyps["n_frame_stack"])
        shift, pong = hyps["action_shift"], "Pong" in hyps["env_type"]
        first = [np.asarray(e.reset(), dtype=np.float32) for e in envs]
        HW = first[0].size
        S = C * HW
        pin = torch.cuda.is_available()
        h_frames = torch.zeros(E, HW).pin_memory() if pin else torch.zeros(E, HW)
        np_frames = h_frames.numpy()
        for j in range(E):
            np_frames[j] = first[j].reshape(-1)
        f32 = dict(dtype=torch.float32, device=dev)
        d_frames = torch.zeros(E, HW, **f32)
        cur, nxt = torch.zeros(E, S, **f32), torch.zeros(E, S, **f32)
        ones, zeros = torch.ones(E, **f32), torch.zeros(E, **f32)
        act_dev = torch.zeros(E, dtype=torch.int64, device=dev)
        h = torch.zeros(E, net.h_size, **f32) if net.is_recurrent else None
        d_frames.copy_(h_frames, non_blocking=True)
        ops.frame_stack_push(d_frames, ones, cur.data_ptr(), S, cur.data_ptr(), S, E, C, HW, st)      # [0,..,0, env.reset()]
        fused = (not net.is_recurrent) and getattr(net, "_step_supported", lambda: False)()
        active = np.ones(E, dtype=bool)
        ep_rew = np.zeros(E)
        t = 0
        max_steps = max_steps or int(try_key(hyps, "max_eval_steps", 10 ** 6))
        while active.any() and t < max_steps:
            u = self.uniform_fn(t, E) if self.uniform_fn is not None else torch.rand(E, **f32)
            if fused:       # state t = push(state t-1, frame) + forward + sample in ONE launch
                kw = dict(prev=cur.data_ptr(), prev_stride=S) if t == 0 else \
                    dict(prev=cur.data_ptr(), prev_stride=S, frame_new=d_frames.data_ptr(), reset_mask=zeros.data_ptr(),
                         out=nxt.data_ptr(), out_stride=S)
                net._step(E, st, u=u.data_ptr(), actions=act_dev.data_ptr(), act_stride=1, **kw)
                if t > 0:
                    cur, nxt = nxt, cur
            else:
                if t > 0:
                    ops.frame_stack_push(d_frames, zeros, cur.data_ptr(), S, nxt.data_ptr(), S, E, C, HW, st)
                    cur, nxt = nxt, cur
                if net.is_recurrent:
                    out = net._fwd(cur.data_ptr(), S, E, "eval", st, False, h_in=h)
                else:
                    out = net._fwd(cur.data_ptr(), S, E, "eval", st, False, sampler=(u, act_dev.data_ptr(), 1)) \
                        if getattr(net, "_fused_sampling", False) else net._fwd(cur.data_ptr(), S, E, "eval", st, False)
                if not out.get("sampled", False):
                    ops.softmax_sample(out["logits"], u, act_dev.data_ptr(), 1, E, net.output_space, st=st)
                if net.is_recurrent:
                    h.copy_(out["h"])
            acts = act_dev.cpu().numpy()                       # the one host round trip of the step
            for j in np.nonzero(active)[0]:
                obs, rew, done, _ = envs[j].step(int(acts[j]) + shift)
                ep_rew[j] += rew
                if pong and rew != 0:
                    done = True
                if done:                                        # this env's episode is over: it stops playing
                    active[j] = False
                else:
                    np_frames[j] = np.asarray(obs, dtype=np.float32).reshape(-1)
            d_frames.copy_(h_frames, non_blocking=True)
            t += 1
        return float(ep_rew.sum()) / E

    # ---- the reference's loop, one env, batch 1 (runner.py:274-314)
    def _rollout_serial(self, net):
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
