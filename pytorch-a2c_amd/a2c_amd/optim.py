"""Fused clip + optimiser steps over the flat parameter arena (updater.py:129-132, 226-229).

``RMSprop`` / ``Adam`` are ``torch.optim.Optimizer`` subclasses whose ``state_dict()`` has the
same layout as ``torch.optim.RMSprop`` / ``torch.optim.Adam`` (param indices follow
``net.parameters()``, per-parameter ``step`` / ``square_avg`` / ``exp_avg`` / ``exp_avg_sq``), so
``optim.p`` checkpoints (training.py:127-128, updater.py:218-219) load either way.  The state
tensors are views into flat HBM buffers parallel to the net's arena; one kernel launch updates
every parameter (torch defaults otherwise: alpha .99 / betas (.9,.999), eps 1e-8, no momentum,
no weight decay, no amsgrad -- the reference only ever passes ``lr``).
"""
import torch

from . import ops


class _Fused(torch.optim.Optimizer):
    _state_names = ()
    capture_safe = False

    def __init__(self, net, lr, defaults):
        net._ensure_device()
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, **defaults))
        ar = net._arena
        self._flat = {k: torch.zeros(ar.n_train, dtype=torch.float32, device=ar.params.device)
                      for k in self._state_names}
        self._stats = torch.zeros(2, dtype=torch.float64, device=ar.params.device)   # [sum g^2, -]
        self._scratch = None         # this optimiser's own reduction scratch (ops.new_reduce_scratch), made on first use
        self._norm = torch.zeros(1, dtype=torch.float32, device=ar.params.device)
        self._steps = 0
        self._name_of = {id(p): n for n, p in net.named_parameters()}

    def _views(self, p):
        o, k, shp = self.net._arena.offsets[self._name_of[id(p)]]
        return {s: self._flat[s][o:o + k].view(shp) for s in self._state_names}

    def _publish_state(self):
        """Expose the flat state as torch-style per-parameter entries (lazily, like torch)."""
        ar = self.net._arena
        for p in self.param_groups[0]["params"]:
            if self._name_of[id(p)] not in ar.trainable:
                continue
            st = self.state[p]
            if "step" not in st:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st.update(self._views(p))
            st["step"].fill_(float(self._steps))

    def state_dict(self):
        if self._steps > 0:
            self._publish_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        ar = self.net._arena
        steps = 0
        for p in self.param_groups[0]["params"]:
            st = self.state.get(p)
            if not st:
                continue
            views = self._views(p)
            for s in self._state_names:
                views[s].copy_(st[s])
                st[s] = views[s]
            steps = max(steps, int(float(st["step"])))
            st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        self._steps = steps

    def zero_grad(self, set_to_none=False):
        # gradients live in the arena and are overwritten by every backward pass
        pass

    def grad_norm(self):
        """Pre-clip global gradient norm of the last step (device float tensor)."""
        return self._norm

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, st=None):
        ar = self.net._arena
        st = st if st is not None else ops.stream()
        g = ar.train_grads()
        if self._scratch is None or self._scratch.device != g.device:
            self._scratch = ops.new_reduce_scratch(g.device)
        ops.gradnorm_sq(g, self._stats[:1], st, scratch=self._scratch)
        self._steps += 1
        self._launch(ar.train_params(), g, float("1e30") if max_norm is None else float(max_norm), st)
        self.net.mark_dirty()


class RMSprop(_Fused):
    _state_names = ("square_avg",)
    capture_safe = True          # every argument of the step kernel is step-independent (hipGraph replays are exact)

    def __init__(self, net, lr=1e-2, alpha=0.99, eps=1e-8):
        super().__init__(net, lr, dict(alpha=alpha, eps=eps, weight_decay=0, momentum=0, centered=False,
                                       capturable=False, foreach=None, maximize=False, differentiable=False))

    def _launch(self, p, g, max_norm, st):
        grp = self.param_groups[0]
        ops.clip_rmsprop(p, g, self._flat["square_avg"], self._stats, max_norm, grp["lr"], grp["alpha"], grp["eps"],
                         self._norm, st)


class Adam(_Fused):
    _state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(net, lr, dict(betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                       foreach=None, capturable=False, differentiable=False, fused=None))

    def _launch(self, p, g, max_norm, st):
        grp = self.param_groups[0]
        ops.clip_adam(p, g, self._flat["exp_avg"], self._flat["exp_avg_sq"], self._stats, max_norm, grp["lr"],
                      grp["betas"][0], grp["betas"][1], grp["eps"], self._steps, self._norm, st)
