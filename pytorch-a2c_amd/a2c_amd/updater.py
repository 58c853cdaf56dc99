"""Drop-in for the reference's ``a2c/updater.py``: ``Updater(net, hyps)`` with
``update_model(shared_data) -> info`` / ``bptt`` / ``save_model`` / ``new_lr`` / ``new_optim`` /
``print_statistics`` / ``log_statistics`` and the attributes ``net optim info norm hyps
is_discrete ret_mean ret_std`` (updater.py:9-229).

One ``update_model`` call enqueues, on one HIP stream and without host round trips until the
final read-back of five scalars:
  fused GAE + returns scan -> model forward (N = n_rollouts*n_tsteps samples, or the BPTT unroll)
  -> advantage statistics -> fused loss forward+backward -> model backward into the flat gradient
  arena -> [RCCL all-reduce when sharded] -> grad-norm reduction -> fused clip + RMSprop/Adam.
"""
import torch

from . import ops, optim as fused_optim
from .parallel import Shard
from .utils import try_key


def recurrent_key(shared_data):
    return "h_states" in shared_data


class _CutShard:
    """Shard stand-in used while capturing: every collective ends the graph being captured and starts the next"""

    def __init__(self, owner, real):
        self.o, self.real = owner, real
        self.rank, self.world, self.group = real.rank, real.world, real.group

    @property
    def active(self):
        return self.real.active

    def global_count(self, n):
        return self.real.global_count(n)

    def barrier(self):
        self.real.barrier()

    def allreduce_(self, t):
        if self.real.active:
            self.o._cut(t)
        return t


class _GraphedUpdate:
    def __init__(self, upd, shared_data):
        self.upd, self.graphs, self.colls = upd, [], []
        net = upd.net
        torch.cuda.synchronize()
        dirty, stash, cells = net._dirty, net._stash, getattr(net, "_cells_done", None)
        real = upd.shard
        upd.shard = _CutShard(self, real)
        self._ctx = None
        try:
            self._begin()
            self.dev, self.n_global = upd._enqueue_update(shared_data)
            self._end()
        except BaseException:
            if self._ctx is not None:
                try:
                    self._ctx.__exit__(None, None, None)
                except Exception:      # noqa: BLE001
                    pass
            torch.cuda.synchronize()
            raise
        finally:
            upd.shard = real
        self.prep_sig = net._prep_sig() if hasattr(net, "_prep_sig") else ""
        # the capture itself executed nothing: the net is still in the state the rollout left it in
        net._dirty, net._stash = dirty, stash
        if cells is not None:
            net._cells_done = cells
        upd.optim._steps -= 1

    def _stale_prep(self):
        net = self.upd.net
        return (net._prep_sig() if hasattr(net, "_prep_sig") else "") != self.prep_sig

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        self._ctx = ops.graph_capture(g)
        self._ctx.__enter__()
        self.graphs.append(g)

    def _end(self):
        self._ctx.__exit__(None, None, None)
        self._ctx = None

    def _cut(self, t):
        self._end()
        self.colls.append(t)
        self._begin()

    def replay(self):
        upd = self.upd
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.colls):
                upd.shard.allreduce_(self.colls[i])
        upd.optim._steps += 1
        upd.net.mark_dirty()
        # (the captured tail re-derived the inference weights: _enqueue_update's net._refresh -- unless the net derives more now)
        upd.net._dirty = self._stale_prep()
        return upd._finish_update(self.dev, self.n_global)

    __call__ = replay

    def replay_async(self):
        """replay() without the blocking read-back: -> token for Updater.collect(token) (the five scalars travel to a
        pinned host buffer behind the update on the stream; the caller may enqueue the NEXT rollout before collecting)"""
        upd = self.upd
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.colls):
                upd.shard.allreduce_(self.colls[i])
        upd.optim._steps += 1
        upd.net.mark_dirty()
        upd.net._dirty = self._stale_prep()
        return upd._post_async(self.dev, self.n_global)


class Updater:
    def __init__(self, net, hyps, shard=None):
        self.net = net
        self.hyps = hyps
        self.is_discrete = hyps["is_discrete"]
        if not self.is_discrete:
            raise NotImplementedError("a2c_amd: continuous action spaces are out of scope (see DESIGN.md)")
        self.shard = shard if shard is not None else Shard()
        self.optim = self.new_optim(hyps["lr"])
        self.info = {}
        self.flat_scan_fallback = False     # last update: a slot did not end with done == 1 -> flat scans were used
        self.norm = 0
        self.ret_mean = None
        self.ret_std = None
        self._bufs = None

    # ------------------------------------------------------------------ helpers
    def _dev(self, t, dtype=torch.float32):
        dev = self.net._dev
        if t.device != dev or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=dev, dtype=dtype).contiguous()
        return t

    def _buffers(self, N):
        b = self._bufs
        if b is None or b["N"] != N:
            dev = self.net._dev
            b = dict(N=N,
                     advs=torch.empty(N, device=dev), rets=torch.empty(N, device=dev),
                     vals_c=torch.empty(N, device=dev),
                     stats=torch.zeros(8, dtype=torch.float64, device=dev),   # [adv sums 0:2 | loss sums 2:5 | ret 5:7]
                     scratch=ops.new_reduce_scratch(dev),     # this updater's ticket + partials (never shared, never born in a capture)
                     err=torch.zeros(1, dtype=torch.int32, device=dev),
                     out5=torch.zeros(5, dtype=torch.float64, device=dev),
                     host=torch.zeros(8, dtype=torch.float64).pin_memory() if torch.cuda.is_available() else None)
            self._bufs = b
        return b

    # ------------------------------------------------------------------ the update
    def update_model(self, shared_data):
        """One A2C update (updater.py:53-137).  Everything up to the five scalars the reference reads
        back is enqueued on the stream without a host round trip (``_enqueue_update``: a launch-bound
        caller may capture that part into a hipGraph); ``_finish_update`` then syncs once."""
        return self._finish_update(*self._enqueue_update(shared_data))

    def _enqueue_update(self, shared_data):
        hyps, net, sh = self.hyps, self.net, self.shard
        net._ensure_device()
        net.req_grads(True)
        st = ops.stream()
        states = self._dev(shared_data["states"])
        rewards = self._dev(shared_data["rewards"]).reshape(-1)
        dones = self._dev(shared_data["dones"]).reshape(-1)
        deltas = self._dev(shared_data["deltas"]).reshape(-1)
        actions = self._dev(shared_data["actions"], torch.int64).reshape(-1)
        N = states.shape[0]
        T = int(hyps["n_tsteps"])
        if N % T:
            raise ValueError("len(states) is not a multiple of n_tsteps")
        R = N // T
        # global batch size: one blocking scalar all-reduce the first time a batch size is seen
        # (every rank steps with the same shapes, so the answer cannot change afterwards)
        cache = self.__dict__.setdefault("_n_global", {})
        if N not in cache:
            cache[N] = sh.global_count(N)
        n_global = cache[N]
        b = self._buffers(N)
        advs, rets, stats = b["advs"], b["rets"], b["stats"]
        if ops._TORCH_ABI_ON():     # A2C_TORCH_OPS=1: the launches below go through torch.ops.a2c_mi355x.abi_* (ops.TorchAbi),
            # which finds the tensor behind every address it is handed: tell it about the buffers that travel as raw rows
            for t in (states, rewards, dones, deltas, actions, net._arena.params, net._arena.grads,
                      shared_data.get("h_states") if recurrent_key(shared_data) else None, *[v for v in b.values() if torch.is_tensor(v)]):
                if t is not None and t.is_cuda:
                    ops.note_tensor(t)

        # advantages (gamma*lambda) and discounted returns (gamma) in one pass (updater.py:70-71, 86-88)
        with ops.span("gae_returns_scan"):
            ops.gae_returns(deltas, rewards, dones, hyps["gamma"] * hyps["lambda_"], hyps["gamma"], R, T, advs, rets,
                            err=b["err"], st=st)

        # forward pass (updater.py:73-80)
        recurrent = "h_states" in shared_data
        use_bptt = recurrent and bool(hyps["use_bptt"])
        if recurrent:
            h_states = self._dev(shared_data["h_states"])
            if use_bptt:
                net.bptt_forward(states, h_states, dones, R, T, "train", st)
            else:
                net._refresh(st)
                net._fwd(states.data_ptr(), states[0].numel(), N, "train", st, True, h_in=h_states)
        else:
            net._refresh(st)
            net._fwd(states.data_ptr(), states[0].numel(), N, "train", st, True)
        hb, logits, vals = net._heads("train", N)

        if hyps["use_nstep_rets"]:      # returns = advs + vals.data (updater.py:84)
            ops.copy_rows(vals.data_ptr(), vals.stride(0), b["vals_c"].data_ptr(), 1, N, 1, st)
            ops.add(advs, b["vals_c"], rets, st)
        if try_key(hyps, "norm_returns", False):
            self._norm_returns(rets, stats, n_global, st)

        adv_sums = None
        if hyps["norm_advs"]:           # (advs - mean)/(std + 1e-6) over the WHOLE batch (updater.py:97-98)
            adv_sums = stats[0:2]
            ops.moments(advs, adv_sums, st, scratch=b["scratch"])
            sh.allreduce_(adv_sums)

        db, dl, dv = net.dheads("train", N)
        ops.loss_fwd_bwd(logits, vals, actions, advs, rets, adv_sums, n_global, try_key(hyps, "pi_coef", 1.0), hyps["val_coef"],
                         hyps["entr_coef"], dl, dv, stats[2:5], st, scratch=b["scratch"])

        # backward into the flat gradient arena
        if use_bptt:
            net.bptt_backward(states, dones, R, T, "train", st)
        else:
            net._bwd(states.data_ptr(), states[0].numel(), N, "train", st)
        net._arena.attach_grads()
        if sh.active:                   # ONE all-reduce over xGMI per update: gradients + the three loss sums in its tail
            tail = net._arena.grad_tail()
            tail[:3].copy_(stats[2:5])                      # reporting only (updater.py:134-136): fp32 is plenty
            sh.allreduce_(net._arena.reduce_span())
            stats[2:5].copy_(tail[:3])

        # clip_grad_norm_ + optimiser step (updater.py:129-132), fused
        with ops.span("clip_optimizer"):
            self.optim.step(max_norm=hyps["max_norm"], st=st)
        self.optim.zero_grad()

        # the weights just changed: re-derive what the NEXT rollout needs from them (conv fragments, composed heads) here, at
        # the tail of the update's stream work (inside its hipGraph when captured), not on the host path between the
        # read-back below and the rollout kernel's launch, where the GPU would sit idle behind ~50 us of Python + 5 launches
        net._refresh(st)
        # five scalars for the host (the reference's .item() calls, updater.py:134-136), gathered by one kernel
        ops.check(ops.lib().a2c_pack_update_scalars(stats[2:5].data_ptr(), self.optim.grad_norm().data_ptr(),
                                                    b["err"].data_ptr(), b["out5"].data_ptr(), st), "a2c_pack_update_scalars")
        return b["out5"], n_global

    def capture_update(self, shared_data):
        """The enqueue half of update_model as hipGraphs: ONE graph on a single GPU; with a sharded update the captured
        stream is cut at every collective (advantage moments, gradient arena) -- graph, RCCL all-reduce on the stream,
        graph, ... -- so that N-GPU updates are not ~40 eager launches either and no collective sits inside a capture.
        Returns a callable ``replay() -> info`` (same effect as update_model on the same buffers); the caller must
        have run one eager update_model on these buffers first (workspaces, tuners, one-time hipMalloc / hipMemset of
        the GEMM and conv launchers: none of that is legal inside a capture)."""
        if not getattr(self.optim, "capture_safe", False):
            # Adam's bias correction takes the step count as a KERNEL ARGUMENT: a replay would apply the captured
            # step's correction for ever
            raise RuntimeError(f"a2c_amd: {type(self.optim).__name__}.step cannot be captured into a hipGraph "
                               "(its step count is a kernel argument); use update_model")
        if self._bufs is None:
            raise RuntimeError("a2c_amd: capture_update needs one eager update_model on these buffers first")
        return _GraphedUpdate(self, shared_data)

    def update_model_async(self, shared_data):
        """update_model without the blocking read-back of the five scalars: everything is enqueued, the scalars go to a
        pinned host buffer behind the update on the stream; ``collect(token)`` waits for THAT copy only and returns the info
        dict.  Lets a driver enqueue the next rollout (which needs nothing from the host) before it looks at the losses."""
        return self._post_async(*self._enqueue_update(shared_data))

    def _post_async(self, dev_vec, n_global):
        hb = self.__dict__.setdefault("_host_ring", [torch.zeros(8, dtype=torch.float64).pin_memory() for _ in range(4)])
        i = self.__dict__["_host_i"] = (self.__dict__.get("_host_i", -1) + 1) % len(hb)
        hb[i][:dev_vec.numel()].copy_(dev_vec, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return (ev, hb[i], n_global)

    def collect(self, token):
        ev, host, n_global = token
        ev.synchronize()
        return self._finish_host(host, n_global)

    def _finish_update(self, dev_vec, n_global):
        return self._finish_host(dev_vec.cpu(), n_global)

    def _finish_host(self, host, n_global):
        hyps = self.hyps
        # host[4] != 0: a slot did not end with done == 1 (not Runner data, runner.py:244); the scans then ran in
        # the reference's flat form on the device (a2c_gae_returns_fused), so the update is still the reference's
        self.flat_scan_fallback = bool(int(host[4]))
        s_pi, s_val, s_ent = (float(host[i]) for i in range(3))
        pi_loss = try_key(hyps, "pi_coef", 1.0) * -(s_pi / n_global)
        val_loss = hyps["val_coef"] * (s_val / n_global)
        entr_loss = -hyps["entr_coef"] * (s_ent / n_global)
        self.norm = float(host[3])
        self.info = {"Loss": pi_loss + val_loss - entr_loss, "Pi_Loss": pi_loss, "ValLoss": val_loss,
                     "Entropy": entr_loss, "GradNorm": self.norm}
        return self.info

    def _norm_returns(self, rets, stats, n_global, st):
        """EMA-normalised returns (updater.py:89-96, default off): needs the batch mean/std on the
        host to update the running values, hence one extra read-back."""
        sums = stats[5:7]
        ops.moments(rets, sums, st, scratch=self._bufs["scratch"])
        self.shard.allreduce_(sums)
        s0, s1 = (float(v) for v in sums.cpu())
        mean = s0 / n_global
        std = max((s1 - n_global * mean * mean) / (n_global - 1), 0.0) ** 0.5
        if self.ret_mean is None:
            self.ret_mean, self.ret_std = mean, std
        else:
            self.ret_mean = 0.01 * mean + 0.99 * self.ret_mean
            self.ret_std = 0.01 * std + 0.99 * self.ret_std
        # encode (ret_mean, ret_std) as moment sums of a virtual population so the same kernel applies them
        n = float(n_global)
        fake = torch.tensor([self.ret_mean * n, (self.ret_std ** 2) * (n - 1) + n * self.ret_mean ** 2],
                            dtype=torch.float64, device=rets.device)
        ops.normalize(rets, rets, fake, n_global, 1e-6, st)

    def bptt(self, states, h_states, dones):
        """updater.py:139-169: unrolled recurrent forward; returns (vals (N,), logits (N,A))."""
        hyps, net = self.hyps, self.net
        net._ensure_device()
        R, T = int(hyps["n_rollouts"]), int(hyps["n_tsteps"])
        vals, logits = net.bptt_forward(self._dev(states), self._dev(h_states), self._dev(dones).reshape(-1), R, T,
                                        "train", ops.stream())
        return vals.clone(), logits.clone()

    def gae(self, rewards, values, next_vals, dones, gamma, lambda_):
        """updater.py:172-189 (unused by the reference's own loop; kept for API parity)."""
        from .utils import discount
        deltas = rewards + gamma * next_vals * (1 - dones) - values
        return discount(deltas, dones, gamma * lambda_)

    # ------------------------------------------------------------------ bookkeeping API
    def print_statistics(self):
        nums = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in self.info.items()}
        print(" – ".join(k + ": " + str(round(v, 5)) for k, v in sorted(nums.items())))

    def log_statistics(self, log, T, reward, avg_action, best_avg_rew):
        nums = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in self.info.items()}
        arr = [k + ": " + (str(round(v, 5)) if "ntropy" not in k else str(v)) for k, v in nums.items()]
        arr += ["EpRew: " + str(reward), "AvgAction: " + str(avg_action), "BestRew:" + str(best_avg_rew)]
        log.write("Step:" + str(T) + " – " + " – ".join(arr) + "\n")
        log.flush()

    def save_model(self, net_file_name, optim_file_name):
        torch.save(self.net.state_dict(), net_file_name)
        if optim_file_name is not None:
            torch.save(self.optim.state_dict(), optim_file_name)

    def new_lr(self, new_lr):
        # like the reference: a new optimiser that then loads the old state (which, as in the
        # reference, also restores the old lr: updater.py:221-224)
        new_optim = self.new_optim(new_lr)
        new_optim.load_state_dict(self.optim.state_dict())
        self.optim = new_optim

    def new_optim(self, lr):
        return getattr(fused_optim, self.hyps["optim_type"])(self.net, lr=lr)
